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
extern "C" __attribute__((visibility("default"))) int dcl_abi_version(void) { return 1; }
