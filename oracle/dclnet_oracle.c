/*
 * dclnet_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle, CPU baseline).
 *
 * Plain-C restatement of the reference's native kernels on the DCL-Net forward
 * path.  Every function cites the reference file:line it follows (paths are
 * relative to the upstream repository root).  Nothing under dcl-net_amd/ may
 * import, link or execute this file: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker.
 *
 * Parity status: the reference ships no tests / golden vectors for these
 * kernels and its .cu sources cannot be compiled here (no nvcc), so the
 * restatements of the CUDA kernels are "parity unpinned" against the CUDA
 * binary.  What IS pinned:
 *   - the sparse-conv rulebook geometry, against the reference's own
 *     header-only C++ (libs/spconv/include/spconv/geometry.h) built by
 *     oracle/Makefile into oracle/_ref/ (tests/test_oracle_ref_geometry.py);
 *   - the Python graph above these kernels, against the reference's
 *     models/*.py imported in the build container (tests/golden/).
 *
 * Floating-point policy (build with -ffp-contract=off): squared distances
 * (a*a + b*b + c*c) and the 3-term interpolation sum (a*b + c*d + e*f) are
 * evaluated as fmaf(e,f, fmaf(a,b, c*d)) -- the association LLVM's scalar FMA
 * contraction (the rule nvcc's NVVM back end applies under its -fmad=true
 * default) gives those source expressions.  All other arithmetic is
 * uncontracted IEEE fp32 in source order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))
#ifdef _OPENMP
#include <omp.h>
#endif
/* threads of the row loops marked `omp parallel for` (bench.py's cpu_baseline leg sets the usable host cores; the
 * results do not depend on it: no reduction crosses threads) */
ORC_API int orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* ORC_FMA_POLICY selects how the two contractible source expressions of the CUDA kernels are associated.  The CUDA
 * binary cannot be run here, so the policy nvcc really applied is an assumption (policy 0); the alternates exist to
 * BOUND that assumption: tests/test_fma_policy.py counts the index picks that differ between the four builds and
 * checks that the pose does not move (oracle/Makefile builds liboracle_p1.so .. liboracle_p3.so from this same file).
 *   0  fmaf(dz,dz, fmaf(dx,dx, dy*dy))   -- LLVM/NVVM scalar contraction of  dx*dx + dy*dy + dz*dz   (the pinned policy)
 *   1  (dx*dx + dy*dy) + dz*dz            -- no contraction at all (nvcc -fmad=false)
 *   2  fmaf(dx,dx, fmaf(dy,dy, dz*dz))   -- the chain a compiler forms when it contracts from the right
 *   3  fmaf(dz,dz, fmaf(dy,dy, dx*dx))   -- left-to-right contraction with the FIRST product kept as the multiply:
 *                                           with 0 the two ways to contract (dx*dx + dy*dy) + dz*dz                    */
#ifndef ORC_FMA_POLICY
#define ORC_FMA_POLICY 0
#endif
ORC_API int orc_fma_policy(void) { return ORC_FMA_POLICY; }

static inline float dist2f(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
#if ORC_FMA_POLICY == 0
    return fmaf(dz, dz, fmaf(dx, dx, dy * dy));
#elif ORC_FMA_POLICY == 1
    return (dx * dx + dy * dy) + dz * dz;
#elif ORC_FMA_POLICY == 2
    return fmaf(dx, dx, fmaf(dy, dy, dz * dz));
#else
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
#endif
}

/* a*b + c*d + e*f under the same contraction policy as dist2f. */
static inline float wsum3(float a, float b, float c, float d, float e, float f) {
#if ORC_FMA_POLICY == 0
    return fmaf(e, f, fmaf(a, b, c * d));
#elif ORC_FMA_POLICY == 1
    return (a * b + c * d) + e * f;
#elif ORC_FMA_POLICY == 2
    return fmaf(a, b, fmaf(c, d, e * f));
#else
    return fmaf(e, f, fmaf(c, d, a * b));
#endif
}

/* ------------------------------------------------------------------------- */
/* a0. voxelize_idx  (libs/pointgroup_ops/src/voxelize/voxelize.cpp:58-152,
 *     :34-49).  First-encounter voxel ids, one key space per batch id.       */
/* ------------------------------------------------------------------------- */
typedef struct { int64_t k[4]; int32_t id; int32_t used; } vox_slot;

static uint64_t vox_hash(const int64_t *k) {
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < 4; ++i) { h ^= (uint64_t)k[i]; h *= 1099511628211ull; h ^= h >> 29; }
    return h;
}

/* pass 1: input_map[i] = voxel id of point i; returns nActive, *max_active. */
ORC_API int orc_voxelize_idx_pass1(const int64_t *coords, int n, int32_t *input_map,
                                   int32_t *max_active) {
    size_t cap = 16;
    while (cap < (size_t)n * 2 + 2) cap <<= 1;
    vox_slot *tab = (vox_slot *)calloc(cap, sizeof(vox_slot));
    int32_t *cnt = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
    int nActive = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t *k = coords + (size_t)i * 4;        /* [b,x,y,z]  voxelize.cpp:95-103 */
        size_t s = (size_t)vox_hash(k) & (cap - 1);
        for (;;) {
            if (!tab[s].used) {
                tab[s].used = 1; memcpy(tab[s].k, k, sizeof(int64_t) * 4);
                tab[s].id = nActive++;                     /* voxelize.cpp:104-107 */
                break;
            }
            if (!memcmp(tab[s].k, k, sizeof(int64_t) * 4)) break;
            s = (s + 1) & (cap - 1);
        }
        input_map[i] = tab[s].id;                          /* voxelize.cpp:110 */
        cnt[tab[s].id]++;
    }
    int32_t mx = 1;                                         /* voxelize.cpp:139-143 */
    for (int v = 0; v < nActive; ++v) if (cnt[v] > mx) mx = cnt[v];
    *max_active = mx;
    free(cnt); free(tab);
    return nActive;
}

/* modes 0 / 1 / 2 (voxelize.cpp:119-138): one point per voxel, rule row = [1, point]; mode 0 = outputRows[i][0] (unique by
 * contract), mode 1 = outputRows[i].front(), mode 2 = outputRows[i].back().  output_coords = coords of that point
 * (voxelize.cpp:34-49).  Buffers zeroed by the caller; output_map is (n_active, 2).                                */
ORC_API void orc_voxelize_idx_pass2_single(const int64_t *coords, int n, const int32_t *input_map, int mode,
                                           int64_t *output_coords, int32_t *output_map) {
    for (int i = 0; i < n; ++i) {
        int v = input_map[i];
        int32_t *row = output_map + (size_t)v * 2;
        if (row[0] == 0 || mode == 2) {
            row[0] = 1; row[1] = i;
            memcpy(output_coords + (size_t)v * 4, coords + (size_t)i * 4, sizeof(int64_t) * 4);
        }
    }
}

/* pass 2: output_map[v] = [count, p0, p1, ..., 0 pad]; output_coords[v] =
 * coords of p0 (voxelize.cpp:34-49, 144-149). Buffers must be zeroed.        */
ORC_API void orc_voxelize_idx_pass2(const int64_t *coords, int n, const int32_t *input_map,
                                    int n_active, int max_active, int64_t *output_coords,
                                    int32_t *output_map) {
    (void)n_active;
    for (int i = 0; i < n; ++i) {
        int v = input_map[i];
        int32_t *row = output_map + (size_t)v * (max_active + 1);
        if (row[0] == 0) memcpy(output_coords + (size_t)v * 4, coords + (size_t)i * 4, sizeof(int64_t) * 4);
        row[0] += 1;
        row[row[0]] = i;
    }
}

/* ------------------------------------------------------------------------- */
/* a1. voxelize_fp (libs/pointgroup_ops/src/voxelize/voxelize.cu:9-23):
 * out[plane] += multiplier * inp[plane], serial over the rule's points.      */
/* ------------------------------------------------------------------------- */
ORC_API void orc_voxelize_fp(const float *feats, const int32_t *rules, float *out, int n_rows,
                             int max_active, int n_planes, int average) {
    for (int row = 0; row < n_rows; ++row) {
        const int32_t *r = rules + (size_t)row * (max_active + 1);
        int nActive = r[0];
        float mult = (average && nActive > 0) ? 1.0f / (float)nActive : 1.0f;
        float *o = out + (size_t)row * n_planes;
        for (int p = 0; p < n_planes; ++p) o[p] = 0.0f;
        for (int i = 1; i <= nActive; ++i) {
            const float *inp = feats + (size_t)r[i] * n_planes;
            for (int p = 0; p < n_planes; ++p) o[p] = o[p] + mult * inp[p];
        }
    }
}

/* ------------------------------------------------------------------------- */
/* a3. rulebooks.  getValidOutPos: libs/spconv/include/spconv/geometry.h:23-85 */
/* ------------------------------------------------------------------------- */
static int valid_out_pos(const int *pos, int ks, int st, int pad, int dil, const int *oshape,
                         int *out /* [27][4]: o0,o1,o2,offset */) {
    int lowers[3], uppers[3], csize[3], counter[3] = {0, 0, 0};
    int numPoints = 1, pc = 0;
    for (int i = 0; i < 3; ++i) {
        lowers[i] = (pos[i] - (ks - 1) * dil - 1 + st + pad) / st;   /* geometry.h:40-44 */
        uppers[i] = (pos[i] + pad) / st;
        csize[i] = (uppers[i] - lowers[i]) / dil + 1;                  /* geometry.h:48-51 */
        numPoints *= csize[i];
    }
    for (int t = 0; t < numPoints; ++t) {
        int valid = 1, m = 1, offset = 0;
        for (int j = 2; j >= 0; --j) {                                  /* geometry.h:61-70 */
            int val = uppers[j] - counter[j] * dil;
            out[pc * 4 + j] = val;
            if (val < 0 || val > oshape[j] - 1) valid = 0;
            offset += m * (pos[j] - val * st + pad) / dil;
            m *= ks;
        }
        out[pc * 4 + 3] = offset;
        if (valid) ++pc;
        counter[2] += 1;
        for (int c = 2; c >= 0; --c)                                    /* geometry.h:76-81 */
            if (counter[c] == csize[c] && c > 0) { counter[c - 1] += 1; counter[c] = 0; }
    }
    return pc;
}

static int cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

/* Non-submanifold conv / pool rulebook in the reference's GPU order
 * (spconv_ops.h:104-135; indice.cu.h:24-65,112-148): candidate outputs are
 * linear indices ((b*S0+o0)*S1+o1)*S2+o2, the output list is their SORTED
 * UNIQUE set (torch::_unique, spconv_ops.h:126), out ids are ranks in it.
 * pairs: [27][2][V] (-1 padded), indice_num[27]; pair order inside an offset
 * is ascending input id here (unspecified on the GPU: atomicAdd).
 * Returns numActOut; out_indices needs capacity min(27*V, batch*S0*S1*S2).   */
ORC_API int orc_indice_pairs_conv(const int32_t *indices, int V, int batch, const int *oshape,
                                  int ks, int st, int pad, int dil, int32_t *out_indices,
                                  int32_t *pairs, int32_t *indice_num) {
    (void)batch;
    int kv = ks * ks * ks;
    for (size_t i = 0; i < (size_t)kv * 2 * V; ++i) pairs[i] = -1;     /* spconv_ops.h:55-57 */
    for (int k = 0; k < kv; ++k) indice_num[k] = 0;
    if (V == 0) return 0;
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)kv * V);
    size_t nc = 0;
    int vp[27 * 4];
    int svol = oshape[0] * oshape[1] * oshape[2];
    for (int j = 0; j < V; ++j) {
        const int32_t *p = indices + (size_t)j * 4;
        int pos[3] = {p[1], p[2], p[3]};
        int nv = valid_out_pos(pos, ks, st, pad, dil, oshape, vp);
        for (int i = 0; i < nv; ++i) {
            int off = vp[i * 4 + 3];
            int lin = (vp[i * 4] * oshape[1] + vp[i * 4 + 1]) * oshape[2] + vp[i * 4 + 2] + svol * p[0];
            int c = indice_num[off]++;                                   /* indice.cu.h:57-60 */
            pairs[((size_t)off * 2 + 0) * V + c] = j;
            pairs[((size_t)off * 2 + 1) * V + c] = lin;
            cand[nc++] = lin;
        }
    }
    qsort(cand, nc, sizeof(int32_t), cmp_i32);                           /* spconv_ops.h:126 */
    size_t nu = 0;
    for (size_t i = 0; i < nc; ++i) if (i == 0 || cand[i] != cand[i - 1]) cand[nu++] = cand[i];
    for (size_t r = 0; r < nu; ++r) {                                    /* indice.cu.h:121-127 */
        int lin = cand[r];
        int32_t *o = out_indices + r * 4;
        o[3] = lin % oshape[2]; lin /= oshape[2];
        o[2] = lin % oshape[1]; lin /= oshape[1];
        o[1] = lin % oshape[0]; lin /= oshape[0];
        o[0] = lin;
    }
    for (int k = 0; k < kv; ++k)                                          /* indice.cu.h:140-146 */
        for (int c = 0; c < indice_num[k]; ++c) {
            int32_t *slot = &pairs[((size_t)k * 2 + 1) * V + c];
            int32_t *f = (int32_t *)bsearch(slot, cand, nu, sizeof(int32_t), cmp_i32);
            *slot = (int32_t)(f - cand);
        }
    free(cand);
    return (int)nu;
}

/* Submanifold rulebook (indice.cu.h:150-208; geometry.h:243-293): out = in. */
ORC_API int orc_indice_pairs_subm(const int32_t *indices, int V, int batch, const int *shape,
                                  int ks, int dil, int32_t *pairs, int32_t *indice_num) {
    int kv = ks * ks * ks, pad = ks / 2, st = 1;                         /* spconv_ops.h:76-79 */
    for (size_t i = 0; i < (size_t)kv * 2 * V; ++i) pairs[i] = -1;
    for (int k = 0; k < kv; ++k) indice_num[k] = 0;
    if (V == 0) return 0;
    size_t svol = (size_t)shape[0] * shape[1] * shape[2];
    int32_t *grid = (int32_t *)malloc(sizeof(int32_t) * svol * batch);
    for (size_t i = 0; i < svol * batch; ++i) grid[i] = -1;
    for (int j = 0; j < V; ++j) {
        const int32_t *p = indices + (size_t)j * 4;
        grid[((size_t)p[1] * shape[1] + p[2]) * shape[2] + p[3] + svol * p[0]] = j;
    }
    int vp[27 * 4];
    for (int j = 0; j < V; ++j) {
        const int32_t *p = indices + (size_t)j * 4;
        int pos[3] = {p[1], p[2], p[3]};
        int nv = valid_out_pos(pos, ks, st, pad, dil, shape, vp);
        for (int i = 0; i < nv; ++i) {
            int off = vp[i * 4 + 3];
            size_t lin = ((size_t)vp[i * 4] * shape[1] + vp[i * 4 + 1]) * shape[2] + vp[i * 4 + 2] + svol * p[0];
            if (grid[lin] > -1) {                                          /* indice.cu.h:200-205 */
                int c = indice_num[off]++;
                pairs[((size_t)off * 2 + 0) * V + c] = j;
                pairs[((size_t)off * 2 + 1) * V + c] = grid[lin];
            }
        }
    }
    free(grid);
    return V;
}

/* ------------------------------------------------------------------------- */
/* a4. indiceConv forward (spconv_ops.h:253-349; reordering.cu.h:21-157):
 * out zeroed; subm: out = feat * W[argmax indiceNum] first (first maximum,
 * std::max_element :265-267); then for k ascending: gather rows, GEMM with
 * W[k] (Cin x Cout), scatter-add.  The GEMM's inner reduction order is
 * library-defined in the reference; here it is an ascending-ci fmaf chain
 * (what an fp32 MFMA does), then one fp32 add into out.                      */
/* ------------------------------------------------------------------------- */
ORC_API void orc_indice_conv(const float *feat, const float *filters, const int32_t *pairs,
                             const int32_t *indice_num, int V_in, int n_out, int cin, int cout,
                             int kv, int subm, float *out) {
    memset(out, 0, sizeof(float) * (size_t)n_out * cout);
    int kmax = 0;
    for (int k = 1; k < kv; ++k) if (indice_num[k] > indice_num[kmax]) kmax = k;
    /* OpenMP (bench.py's cpu_baseline leg runs this on all host cores): rows of one offset are independent -- an output
     * row occurs at most once per offset -- so the per-element summation order, hence every bit, is that of the serial loop */
    if (subm) {
        const float *W = filters + (size_t)kmax * cin * cout;
        #pragma omp parallel for schedule(static)
        for (int r = 0; r < n_out; ++r) {
            const float *f = feat + (size_t)r * cin;
            for (int co = 0; co < cout; ++co) {
                float acc = 0.0f;
                for (int ci = 0; ci < cin; ++ci) acc = fmaf(f[ci], W[(size_t)ci * cout + co], acc);
                out[(size_t)r * cout + co] = acc;
            }
        }
    }
    for (int k = 0; k < kv; ++k) {
        int nHot = indice_num[k];
        if (nHot <= 0 || (subm && k == kmax)) continue;                  /* spconv_ops.h:297-299 */
        const float *W = filters + (size_t)k * cin * cout;
        const int32_t *pin = pairs + ((size_t)k * 2 + 0) * V_in;
        const int32_t *pout = pairs + ((size_t)k * 2 + 1) * V_in;
        #pragma omp parallel for schedule(static)
        for (int c = 0; c < nHot; ++c) {
            const float *f = feat + (size_t)pin[c] * cin;
            float *o = out + (size_t)pout[c] * cout;
            for (int co = 0; co < cout; ++co) {
                float acc = 0.0f;
                for (int ci = 0; ci < cin; ++ci) acc = fmaf(f[ci], W[(size_t)ci * cout + co], acc);
                o[co] = o[co] + acc;                                      /* reordering.cu.h:99-157 */
            }
        }
    }
}

/* a6. indiceSummaryRF + indiceAvgPool (pool_ops.h:141-208; summaryRF.cu:26-41;
 * avgpool.cu:114-135): rf[o] = #pairs into o; out[o] += in[i] / (float)rf[o]
 * in ascending offset order.                                                 */
ORC_API void orc_indice_avgpool(const float *feat, const int32_t *pairs, const int32_t *indice_num,
                                int V_in, int n_out, int c, int kv, int32_t *rf, float *out) {
    memset(out, 0, sizeof(float) * (size_t)n_out * c);
    memset(rf, 0, sizeof(int32_t) * (size_t)n_out);
    for (int k = 0; k < kv; ++k) {
        const int32_t *pout = pairs + ((size_t)k * 2 + 1) * V_in;
        for (int i = 0; i < indice_num[k]; ++i) rf[pout[i]] += 1;        /* summaryRF.cu:39 */
    }
    for (int k = 0; k < kv; ++k) {
        const int32_t *pin = pairs + ((size_t)k * 2 + 0) * V_in;
        const int32_t *pout = pairs + ((size_t)k * 2 + 1) * V_in;
        #pragma omp parallel for schedule(static)
        for (int i = 0; i < indice_num[k]; ++i) {
            const float *f = feat + (size_t)pin[i] * c;
            float *o = out + (size_t)pout[i] * c;
            float d = (float)rf[pout[i]];
            for (int ch = 0; ch < c; ++ch) o[ch] = o[ch] + f[ch] / d;   /* avgpool.cu:130 */
        }
    }
}

/* ------------------------------------------------------------------------- */
/* a8. three_nn, flat with batch column (libs/pointnet_sp/src/
 * interpolate_gpu.cu:9-56).  Returns dist2 (float cast of the double bests:
 * 1e40 -> +inf) and idx.                                                     */
/* ------------------------------------------------------------------------- */
ORC_API void orc_three_nn_sp(int n, int m, const float *unknown, const float *known,
                             float *dist2, int32_t *idx) {
    #pragma omp parallel for schedule(static)
    for (int p = 0; p < n; ++p) {
        const float *u = unknown + (size_t)p * 4;
        float ub = u[0], ux = u[1], uy = u[2], uz = u[3];
        double best1 = 1e40, best2 = 1e40, best3 = 1e40;
        int b1 = 0, b2 = 0, b3 = 0;
        for (int k = 0; k < m; ++k) {
            const float *q = known + (size_t)k * 4;
            if (q[0] != ub) continue;                                      /* :36-38 */
            float d = dist2f(ux, uy, uz, q[1], q[2], q[3]);
            if (d < best1) { best3 = best2; b3 = b2; best2 = best1; b2 = b1; best1 = d; b1 = k; }
            else if (d < best2) { best3 = best2; b3 = b2; best2 = d; b2 = k; }
            else if (d < best3) { best3 = d; b3 = k; }
        }
        dist2[p * 3 + 0] = (float)best1; dist2[p * 3 + 1] = (float)best2; dist2[p * 3 + 2] = (float)best3;
        idx[p * 3 + 0] = b1; idx[p * 3 + 1] = b2; idx[p * 3 + 2] = b3;
    }
}

/* a10. three_interpolate, flat (libs/pointnet_sp/src/interpolate_gpu.cu:80-102):
 * points (M,C) -> out (N,C).                                                 */
ORC_API void orc_three_interpolate_sp(int c, int m, int n, const float *points, const int32_t *idx,
                                      const float *weight, float *out) {
    (void)m;
    #pragma omp parallel for schedule(static)
    for (int p = 0; p < n; ++p) {
        const int32_t *id = idx + (size_t)p * 3;
        const float *w = weight + (size_t)p * 3;
        for (int ch = 0; ch < c; ++ch)
            out[(size_t)p * c + ch] = wsum3(w[0], points[(size_t)id[0] * c + ch],
                                            w[1], points[(size_t)id[1] * c + ch],
                                            w[2], points[(size_t)id[2] * c + ch]);
    }
}

/* ------------------------------------------------------------------------- */
/* a17. libs/pointnet_lib primitives                                          */
/* ------------------------------------------------------------------------- */
/* ball_query (src/ball_query_gpu.cu:9-45); idx pre-zeroed by the caller
 * (pointnet2_utils.py:261).                                                  */
ORC_API void orc_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                            const float *xyz, int32_t *idx) {
    float r2 = radius * radius;
    for (int bs = 0; bs < b; ++bs)
        for (int p = 0; p < m; ++p) {
            const float *c = new_xyz + ((size_t)bs * m + p) * 3;
            const float *X = xyz + (size_t)bs * n * 3;
            int32_t *o = idx + ((size_t)bs * m + p) * nsample;
            int cnt = 0;
            for (int k = 0; k < n; ++k) {
                float d2 = dist2f(c[0], c[1], c[2], X[k * 3], X[k * 3 + 1], X[k * 3 + 2]);
                if (d2 < r2) {
                    if (cnt == 0) for (int l = 0; l < nsample; ++l) o[l] = k;   /* :35-39 */
                    o[cnt] = k;
                    if (++cnt >= nsample) break;
                }
            }
        }
}

/* group_points (src/group_points_gpu.cu:47-66): (B,C,N),(B,np,ns)->(B,C,np,ns) */
ORC_API void orc_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                              const int32_t *idx, float *out) {
    for (int bs = 0; bs < b; ++bs)
        for (int ch = 0; ch < c; ++ch)
            for (int p = 0; p < npoints; ++p)
                for (int s = 0; s < nsample; ++s)
                    out[(((size_t)bs * c + ch) * npoints + p) * nsample + s] =
                        points[((size_t)bs * c + ch) * n + idx[((size_t)bs * npoints + p) * nsample + s]];
}

/* gather_points (src/sampling_gpu.cu:8-24): (B,C,N),(B,M)->(B,C,M) */
ORC_API void orc_gather_points(int b, int c, int n, int m, const float *points, const int32_t *idx,
                               float *out) {
    for (int bs = 0; bs < b; ++bs)
        for (int ch = 0; ch < c; ++ch)
            for (int p = 0; p < m; ++p)
                out[((size_t)bs * c + ch) * m + p] = points[((size_t)bs * c + ch) * n + idx[(size_t)bs * m + p]];
}

/* opt_n_threads (src/cuda_utils.h:10-14): log() on doubles, truncated.       */
ORC_API int orc_fps_block_size(int work_size) {
    int pow_2 = (int)(log((double)work_size) / log(2.0));
    int t = 1 << pow_2;
    if (t > 1024) t = 1024;
    return t < 1 ? 1 : t;
}

/* furthest_point_sampling (src/sampling_gpu.cu:86-209): simulates the
 * thread-strided scan (strict >, earliest k per thread, best=-1, besti=0) and
 * the left-biased shared-memory tree; temp pre-filled with 1e10 by the caller
 * (pointnet2_utils.py:27) and updated in place.                              */
ORC_API void orc_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                         int32_t *idxs) {
    if (m <= 0) return;
    int T = orc_fps_block_size(n);
    float *dists = (float *)malloc(sizeof(float) * T);
    int32_t *dists_i = (int32_t *)malloc(sizeof(int32_t) * T);
    for (int bs = 0; bs < b; ++bs) {
        const float *X = dataset + (size_t)bs * n * 3;
        float *tp = temp + (size_t)bs * n;
        int32_t *out = idxs + (size_t)bs * m;
        int old = 0;
        out[0] = old;
        for (int j = 1; j < m; ++j) {
            float x1 = X[old * 3], y1 = X[old * 3 + 1], z1 = X[old * 3 + 2];
            for (int tid = 0; tid < T; ++tid) {
                int besti = 0; float best = -1.0f;
                for (int k = tid; k < n; k += T) {
                    float d = dist2f(X[k * 3], X[k * 3 + 1], X[k * 3 + 2], x1, y1, z1);
                    float d2 = d < tp[k] ? d : tp[k];                   /* min(d, temp[k]) */
                    tp[k] = d2;
                    besti = d2 > best ? k : besti;
                    best = d2 > best ? d2 : best;
                }
                dists[tid] = best; dists_i[tid] = besti;
            }
            for (int s = T >> 1; s >= 1; s >>= 1)                           /* :139-205 */
                for (int tid = 0; tid < s; ++tid) {
                    float v1 = dists[tid], v2 = dists[tid + s];
                    int i1 = dists_i[tid], i2 = dists_i[tid + s];
                    dists[tid] = v1 > v2 ? v1 : v2;                      /* max(v1,v2) */
                    dists_i[tid] = v2 > v1 ? i2 : i1;                    /* :86-91 */
                }
            old = dists_i[0];
            out[j] = old;
        }
    }
    free(dists); free(dists_i);
}

/* knn (src/interpolate_gpu.cu:9-57): k <= 200, strict-< insertion.           */
ORC_API void orc_knn(int b, int n, int m, int k, const float *unknown, const float *known,
                     float *dist2, int32_t *idx) {
    double best[200]; int besti[200];
    for (int bs = 0; bs < b; ++bs)
        for (int p = 0; p < n; ++p) {
            const float *u = unknown + ((size_t)bs * n + p) * 3;
            const float *K = known + (size_t)bs * m * 3;
            for (int i = 0; i < k; ++i) { best[i] = 1e40; besti[i] = 0; }
            for (int i = 0; i < m; ++i) {
                float d = dist2f(u[0], u[1], u[2], K[i * 3], K[i * 3 + 1], K[i * 3 + 2]);
                for (int j = 0; j < k; ++j)
                    if (d < best[j]) {
                        for (int l = k - 1; l > j; --l) { best[l] = best[l - 1]; besti[l] = besti[l - 1]; }
                        best[j] = d; besti[j] = i;
                        break;
                    }
            }
            for (int i = 0; i < k; ++i) {
                idx[((size_t)bs * n + p) * k + i] = besti[i];
                dist2[((size_t)bs * n + p) * k + i] = (float)best[i];
            }
        }
}

/* batched three_nn (src/interpolate_gpu.cu:81-124) */
ORC_API void orc_three_nn(int b, int n, int m, const float *unknown, const float *known,
                          float *dist2, int32_t *idx) {
    for (int bs = 0; bs < b; ++bs)
        for (int p = 0; p < n; ++p) {
            const float *u = unknown + ((size_t)bs * n + p) * 3;
            const float *K = known + (size_t)bs * m * 3;
            double best1 = 1e40, best2 = 1e40, best3 = 1e40;
            int b1 = 0, b2 = 0, b3 = 0;
            for (int k = 0; k < m; ++k) {
                float d = dist2f(u[0], u[1], u[2], K[k * 3], K[k * 3 + 1], K[k * 3 + 2]);
                if (d < best1) { best3 = best2; b3 = b2; best2 = best1; b2 = b1; best1 = d; b1 = k; }
                else if (d < best2) { best3 = best2; b3 = b2; best2 = d; b2 = k; }
                else if (d < best3) { best3 = d; b3 = k; }
            }
            size_t o = ((size_t)bs * n + p) * 3;
            dist2[o] = (float)best1; dist2[o + 1] = (float)best2; dist2[o + 2] = (float)best3;
            idx[o] = b1; idx[o + 1] = b2; idx[o + 2] = b3;
        }
}

/* batched three_interpolate (src/interpolate_gpu.cu:149-169): (B,C,M)->(B,C,N) */
ORC_API void orc_three_interpolate(int b, int c, int m, int n, const float *points,
                                   const int32_t *idx, const float *weight, float *out) {
    for (int bs = 0; bs < b; ++bs)
        for (int ch = 0; ch < c; ++ch) {
            const float *P = points + ((size_t)bs * c + ch) * m;
            for (int p = 0; p < n; ++p) {
                const int32_t *id = idx + ((size_t)bs * n + p) * 3;
                const float *w = weight + ((size_t)bs * n + p) * 3;
                out[((size_t)bs * c + ch) * n + p] = wsum3(w[0], P[id[0]], w[1], P[id[1]], w[2], P[id[2]]);
            }
        }
}
