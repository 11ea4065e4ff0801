// ref_geometry_driver.cc -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" driver around the REFERENCE's own header-only rulebook code
// (libs/spconv/include/spconv/geometry.h: getValidOutPos :23-85,
// getIndicePairsConv :145-197, getIndicePairsSubM :243-293), compiled where it
// lies under /root/reference by oracle/Makefile into oracle/_ref/.  No
// reference source is copied; the only non-reference header on the include
// path is the genuine cuda_runtime_api.h that ships inside this image's triton
// wheel (tensorview.h includes it for type names only).
//
// Used to pin oracle/dclnet_oracle.c's rulebooks: the CPU functions below
// number output voxels in first-encounter order (geometry.h:181-187); the
// tests compare offset-wise (in, out-coordinate) pair SETS and the output
// coordinate SET, then check the oracle's GPU ordering rule separately.
#include <spconv/geometry.h>
#include <cstdint>
#include <vector>

extern "C" {

int ref_valid_out_pos(const int *pos, int ks, int st, int pad, int dil, const int *oshape, int *out) {
  int k[3] = {ks, ks, ks}, s[3] = {st, st, st}, p[3] = {pad, pad, pad}, d[3] = {dil, dil, dil};
  return spconv::getValidOutPos<int, 3>(pos, k, s, p, d, oshape, out);
}

// indices (V,4) int32; out_indices (>= 27*V,4); grid (batch*prod(oshape)) scratch;
// pairs [27][2][V]; indice_num[27].  Returns numActOut.
int ref_indice_pairs_conv(const int *indices, int V, int batch, const int *oshape, int ks, int st,
                          int pad, int dil, int *out_indices, int *pairs, int *indice_num) {
  int kv = ks * ks * ks;
  int svol = oshape[0] * oshape[1] * oshape[2];
  std::vector<int> grid((size_t)svol * batch, -1);
  for (size_t i = 0; i < (size_t)kv * 2 * V; ++i) pairs[i] = -1;
  for (int i = 0; i < kv; ++i) indice_num[i] = 0;
  int k[3] = {ks, ks, ks}, s[3] = {st, st, st}, p[3] = {pad, pad, pad}, d[3] = {dil, dil, dil};
  tv::TensorView<const int> tIn(indices, {V, 4});
  tv::TensorView<int> tOut(out_indices, {V * kv, 4});
  tv::TensorView<int> tGrid(grid.data(), {(int)grid.size()});
  tv::TensorView<int> tPairs(pairs, {kv, 2, V});
  tv::TensorView<int> tNum(indice_num, {kv});
  return spconv::getIndicePairsConv<int, int, 3>(tIn, tOut, tGrid, tPairs, tNum, k, s, p, d, oshape);
}

int ref_indice_pairs_subm(const int *indices, int V, int batch, const int *shape, int ks, int dil,
                          int *pairs, int *indice_num) {
  int kv = ks * ks * ks;
  int svol = shape[0] * shape[1] * shape[2];
  std::vector<int> grid((size_t)svol * batch, -1);
  for (size_t i = 0; i < (size_t)kv * 2 * V; ++i) pairs[i] = -1;
  for (int i = 0; i < kv; ++i) indice_num[i] = 0;
  int k[3] = {ks, ks, ks}, s[3] = {1, 1, 1}, p[3] = {ks / 2, ks / 2, ks / 2}, d[3] = {dil, dil, dil};
  tv::TensorView<const int> tIn(indices, {V, 4});
  tv::TensorView<int> tGrid(grid.data(), {(int)grid.size()});
  tv::TensorView<int> tPairs(pairs, {kv, 2, V});
  tv::TensorView<int> tNum(indice_num, {kv});
  return spconv::getIndicePairsSubM<int, int, 3>(tIn, tGrid, tPairs, tNum, k, s, p, d, shape);
}
}
