// legacy_rng.cpp -- the loaders' point sampling draws, bit for bit, without numpy's per-element overhead.
//
// The reference's loaders draw the N observed points of a crop with the GLOBAL legacy generator:
//   np.random.choice(m, n_sample_observed_point, replace=False)        (YCBV/dataloader_test_YCBV.py:166-169,
//                                                                       LM/dataloader_test_LM.py:176-181)
// i.e. RandomState.permutation(m)[:n] = a Fisher-Yates shuffle of arange(m) from the top -- for i = m-1 .. 1: j =
// random_interval(i), swap(x[i], x[j]) -- whose index draws are MT19937 words masked to the smallest bit mask >= i and
// rejected while > i (numpy/random/mtrand.pyx: _shuffle_raw; legacy-distributions / distributions.c: random_interval).  numpy
// spends ~14.5 ns per element of m on it (three memcpy calls per swap, the generator behind a function pointer): 0.44 ms per
// 6-object frame in the crop builder, on the critical path of a serial eval loop.  This is the same walk on
// the same generator state as a tight loop over 32-bit indices: the caller hands over the state of np.random (get_state()),
// gets the first n entries of every permutation, and puts the advanced state back (set_state()), so a seeded run consumes
// the global stream exactly like the original loader.  Host code only (no GPU call).
#include "common.h"

namespace {

struct Mt {
  uint32_t *key;   // 624 words, numpy's layout
  int pos;
};

inline void mt_refill(Mt &s) {
  constexpr int N = 624, M = 397;
  constexpr uint32_t MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;
  uint32_t *mt = s.key;
  int kk = 0;
  uint32_t y;
  for (; kk < N - M; ++kk) {
    y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
    mt[kk] = mt[kk + M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
  }
  for (; kk < N - 1; ++kk) {
    y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
    mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
  }
  y = (mt[N - 1] & UPPER) | (mt[0] & LOWER);
  mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
  s.pos = 0;
}

inline uint32_t mt_next(Mt &s) {
  if (s.pos == 624) mt_refill(s);
  uint32_t y = s.key[s.pos++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

}  // namespace

// key624 / pos: np.random.get_state()[1] (uint32[624], updated in place) and [2] (in / out).  For each of the k objects:
// out[o * n .. o * n + n) = np.random.permutation(m[o])[:n] as int64 -- requires n <= m[o] < 2^31 (the loaders draw WITH
// replacement when m <= n: a different numpy routine, left to numpy).  scratch: max(m) int32.
DCL_API int dcl_legacy_permutation_heads(uint32_t *key624, int32_t *pos_io, const int32_t *m, int k, int n, int64_t *out,
                                         int32_t *scratch) {
  DCL_CHECK_ARG(key624 && pos_io && m && k >= 0 && n >= 0 && (out || k == 0 || n == 0) && scratch);
  DCL_CHECK_ARG(*pos_io >= 0 && *pos_io <= 624);
  Mt s{key624, *pos_io};
  for (int o = 0; o < k; ++o) {
    const int mo = m[o];
    DCL_CHECK_ARG(mo >= n && mo >= 1);
    int32_t *x = scratch;
    for (int i = 0; i < mo; ++i) x[i] = i;
    // One generator word per trip, branch-free: a rejected word (j > i: up to half of them just above a power of two) swaps
    // x[i] with itself and leaves i where it is -- the accept / reject branch of the textbook loop is unpredictable and cost
    // more than the generator.  mask = smallest bit mask >= i, recomputed when i falls to half of it.
    uint32_t i = (uint32_t)mo - 1, mask = 0;
    while (i >= 1) {
      if (mask == 0 || (mask >> 1) >= i) {
        mask = i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
      }
      uint32_t stop = mask >> 1;                                // i stays in (stop, mask] with this mask
      if (stop < 1) stop = 0;
      while (i > stop) {
        const uint32_t j = mt_next(s) & mask;
        const uint32_t ok = j <= i ? 1u : 0u;
        const uint32_t jj = ok ? j : i;
        const int32_t t = x[jj];
        x[jj] = x[i];
        x[i] = t;
        i -= ok;
      }
    }
    for (int i = 0; i < n; ++i) out[(size_t)o * n + i] = x[i];
  }
  *pos_io = s.pos;
  return 0;
}
