// voxelize_idx_host.cpp -- HOST voxel hashing (the reference's PG_OP.voxelize_idx,
// libs/pointgroup_ops/src/voxelize/voxelize.cpp:10-152, runs on the CPU inside DataLoader
// workers; so does this one).  Voxel ids follow first-encounter order of the points.
#include <string.h>
#include <unordered_map>
#include <vector>
#include "../../include/dclnet_hip.h"

void dcl_set_error(const char *fmt, ...);
#define API extern "C" __attribute__((visibility("default")))

namespace {
struct Key {
  int64_t v[4];
  bool operator==(const Key &o) const { return !memcmp(v, o.v, sizeof(v)); }
};
struct KeyHash {
  size_t operator()(const Key &k) const {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 4; ++i) {
      h ^= (uint64_t)k.v[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    }
    return (size_t)h;
  }
};
}  // namespace

API int dcl_voxelize_idx_count(const int64_t *coords, int n, int ncol, int batch_size, int mode,
                               int32_t *input_map, int32_t *n_active, int32_t *max_active) {
  (void)batch_size;
  if (!coords || !input_map || !n_active || !max_active || n < 0 || (ncol != 3 && ncol != 4) ||
      mode < 0 || mode > 4) {
    dcl_set_error("dcl_voxelize_idx_count: invalid argument (ncol must be 3|4, mode 0..4)");
    return DCL_EINVAL;
  }
  std::unordered_map<Key, int32_t, KeyHash> grid;
  grid.reserve((size_t)n * 2 + 16);
  std::vector<int32_t> count;
  for (int i = 0; i < n; ++i) {
    Key k;
    if (ncol == 4) {
      for (int j = 0; j < 4; ++j) k.v[j] = coords[(size_t)i * 4 + j];
    } else {
      k.v[0] = 0;
      for (int j = 0; j < 3; ++j) k.v[j + 1] = coords[(size_t)i * 3 + j];
    }
    auto it = grid.find(k);
    int32_t id;
    if (it == grid.end()) {
      id = (int32_t)count.size();
      grid.emplace(k, id);
      count.push_back(0);
    } else {
      id = it->second;
    }
    count[id] += 1;
    input_map[i] = id;
  }
  int32_t mx = 1;
  for (int32_t c : count) mx = c > mx ? c : mx;
  *n_active = (int32_t)count.size();
  // modes 0 (guaranteed unique), 1, 2 keep ONE point per voxel: maxActive == 1 (voxelize.cpp:111-138); the reference asserts
  // uniqueness in mode 0 (:120-124) -- reported as an error here instead of an abort
  if (mode == 0 && mx > 1) {
    dcl_set_error("dcl_voxelize_idx_count: mode 0 promises unique coordinates, but a voxel holds %d points", mx);
    return DCL_EINVAL;
  }
  *max_active = mode <= 2 ? 1 : mx;
  return 0;
}

API int dcl_voxelize_idx_fill(const int64_t *coords, int n, int ncol, const int32_t *input_map,
                              int n_active, int max_active, int64_t *output_coords,
                              int32_t *output_map) {
  return dcl_voxelize_idx_fill_mode(coords, n, ncol, input_map, n_active, max_active, 4, output_coords, output_map);
}

// mode 3 / 4: every point of the voxel, ascending (voxelize.cpp:139-149).  modes 0 / 1: the voxel's FIRST point
// (outputRows[i][0] / .front(), :120-131), mode 2: its LAST point (.back(), :132-137); rows are [1, point].
API int dcl_voxelize_idx_fill_mode(const int64_t *coords, int n, int ncol, const int32_t *input_map,
                                   int n_active, int max_active, int mode, int64_t *output_coords,
                                   int32_t *output_map) {
  if (!coords || !input_map || !output_coords || !output_map || n < 0 || (ncol != 3 && ncol != 4) || mode < 0 || mode > 4 ||
      (mode <= 2 && max_active != 1)) {
    dcl_set_error("dcl_voxelize_idx_fill: invalid argument");
    return DCL_EINVAL;
  }
  if (mode <= 2) {
    for (int i = 0; i < n; ++i) {
      const int v = input_map[i];
      if (v < 0 || v >= n_active) {
        dcl_set_error("dcl_voxelize_idx_fill: input_map[%d]=%d out of range", i, v);
        return DCL_EINVAL;
      }
      int32_t *row = output_map + (size_t)v * 2;
      if (row[0] == 0) memcpy(output_coords + (size_t)v * ncol, coords + (size_t)i * ncol, sizeof(int64_t) * ncol);
      if (row[0] == 0 || mode == 2) { row[0] = 1; row[1] = i; }
    }
    return 0;
  }
  const size_t stride = (size_t)max_active + 1;
  for (int i = 0; i < n; ++i) {
    int v = input_map[i];
    if (v < 0 || v >= n_active) {
      dcl_set_error("dcl_voxelize_idx_fill: input_map[%d]=%d out of range", i, v);
      return DCL_EINVAL;
    }
    int32_t *row = output_map + (size_t)v * stride;
    if (row[0] == 0) memcpy(output_coords + (size_t)v * ncol, coords + (size_t)i * ncol, sizeof(int64_t) * ncol);
    if (row[0] >= max_active) {
      dcl_set_error("dcl_voxelize_idx_fill: voxel %d exceeds max_active", v);
      return DCL_EINVAL;
    }
    row[0] += 1;
    row[row[0]] = i;
  }
  return 0;
}
