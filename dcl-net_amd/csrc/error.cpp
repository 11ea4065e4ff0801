// error.cpp -- thread-local last-error text of the C ABI.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/dclnet_hip.h"

static thread_local char g_err[512] = "";

void dcl_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" __attribute__((visibility("default"))) const char *dcl_last_error(void) { return g_err; }
extern "C" __attribute__((visibility("default"))) int dcl_abi_version(void) { return DCL_ABI_VERSION; }

#ifdef DCL_DIAG
// launch census of the diagnostic library (common.h): host stub -> launches since the last reset
#include <cxxabi.h>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>

static std::mutex g_census_mu;
static std::map<const void *, long long> g_census;

void dcl_diag_note_launch(const void *host_stub) {
  std::lock_guard<std::mutex> g(g_census_mu);
  ++g_census[host_stub];
}

extern "C" __attribute__((visibility("default"))) void dcl_debug_launch_census_reset(void) {
  std::lock_guard<std::mutex> g(g_census_mu);
  g_census.clear();
}

// "name<TAB>count<NEWLINE>" per kernel into buf (NUL-terminated, truncated at cap); returns the bytes the full text needs
extern "C" __attribute__((visibility("default"))) long long dcl_debug_launch_census(char *buf, long long cap) {
  std::map<std::string, long long> by_name;
  {
    std::lock_guard<std::mutex> g(g_census_mu);
    for (const auto &kv : g_census) {
      const char *raw = hipKernelNameRefByPtr(kv.first, nullptr);
      std::string name = raw ? raw : "?";
      int st = 0;
      char *dm = raw ? abi::__cxa_demangle(raw, nullptr, nullptr, &st) : nullptr;
      if (dm && st == 0) name = dm;
      free(dm);
      by_name[name] += kv.second;
    }
  }
  std::string text;
  for (const auto &kv : by_name) text += kv.first + "\t" + std::to_string(kv.second) + "\n";
  if (buf && cap > 0) {
    const size_t n = text.size() < (size_t)cap - 1 ? text.size() : (size_t)cap - 1;
    memcpy(buf, text.data(), n);
    buf[n] = 0;
  }
  return (long long)text.size() + 1;
}
#endif
