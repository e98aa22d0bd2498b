// Attention encoder kernels for gfx950: what dgl GATConv.forward runs between its dense
// projections (reference call sites: GAT, spiro/dtdg/models/encoder/implicitTimeEncoder/
// staticGraphEncoder.py:170,174,190).
//
//   apply_edges(u_add_v('el','er')) + LeakyReLU + edge_softmax   -> gat_softmax_fwd_kernel
//   update_all(u_mul_e('ft','a'), sum)                           -> the fused aggregation
//                                                                   kernel, one launch per head
//   and their gradients: the per-edge dot products <dOut[dst], ft[src]> (sddmm_dot_kernel),
//   the softmax/LeakyReLU backward per destination row (gat_softmax_bwd_kernel) and the sum
//   of edge gradients per source row (segment_sum_kernel, on the by-source CSR).
//
// Per-edge arrays are head-major [H, E] in CSR order, so one head is one contiguous
// weight stream for the aggregation kernel.  Rows are reduced by groups of 16 lanes (four
// rows per wave); rows above the hub threshold get a whole 1024-thread workgroup.  All
// reductions run in a fixed order: results are bit-reproducible, no atomics.
// HBM-bound: E*H*4-byte streams plus one gather of el per edge; the dot-product kernel
// gathers one feature row per edge like the aggregation.

#include <cfloat>

#include "common.h"

namespace {

constexpr int kLight = 16;    // lanes per row, light rows
constexpr int kHeavy = 1024;  // threads per hub row
constexpr int kLightBlock = 256;

template <int G>
__device__ __forceinline__ float group_sum(float v, float* s_red) {
  if constexpr (G <= 64) {
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
  } else {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < G / 64; ++w) t += s_red[w];
    __syncthreads();
    return t;
  }
}

template <int G>
__device__ __forceinline__ float group_max(float v, float* s_red) {
  if constexpr (G <= 64) {
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
  } else {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = -FLT_MAX;
#pragma unroll
    for (int w = 0; w < G / 64; ++w) t = fmaxf(t, s_red[w]);
    __syncthreads();
    return t;
  }
}

// Which row this group of lanes owns and which of its lanes this thread is.  Light
// launch: rows in order, hub rows skipped; heavy launch: one listed hub row per workgroup.
template <int G>
__device__ __forceinline__ bool pick_row(const int32_t* __restrict__ indptr, int32_t n_rows,
                                         const int32_t* __restrict__ heavy_rows,
                                         int32_t heavy_thresh, int32_t* row, int32_t* lo,
                                         int32_t* hi, int* lane_g) {
  int32_t r;
  if constexpr (G == kHeavy) {
    r = heavy_rows[blockIdx.x];
    *lane_g = threadIdx.x;
  } else {
    r = static_cast<int32_t>(blockIdx.x) * (kLightBlock / G) + static_cast<int32_t>(threadIdx.x) / G;
    *lane_g = threadIdx.x % G;
  }
  *row = r;
  *lo = *hi = 0;
  if (r >= n_rows) return false;
  const int32_t a = indptr[r], b = indptr[r + 1];
  if (G != kHeavy && b - a > heavy_thresh) return false;
  *lo = a;
  *hi = b;
  return true;
}

template <int G>
__global__ __launch_bounds__(G == kHeavy ? kHeavy : kLightBlock) void gat_softmax_fwd_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int32_t n_rows,
    int64_t E, const float* __restrict__ el, const float* __restrict__ er, int32_t H, float slope,
    float* __restrict__ attn, float* __restrict__ raw, const int32_t* __restrict__ heavy_rows,
    int32_t heavy_thresh) {
  __shared__ float s_red[kHeavy / 64];
  int32_t r, lo, hi;
  int lane_g;
  const bool valid = pick_row<G>(indptr, n_rows, heavy_rows, heavy_thresh, &r, &lo, &hi, &lane_g);
  // Two passes per head.  Pass 1 gathers the logits (four independent gathers per lane in
  // flight), stores the scores and keeps a running (max, sum of exp) per lane, merged over
  // the group afterwards; pass 2 normalises.  The first score of every lane stays in a
  // register, so rows of up to G entries never read back what they wrote.
  const int32_t j_first = lo + lane_g;
  for (int h = 0; h < H; ++h) {
    float* a = attn + static_cast<size_t>(h) * E;
    const float erh = valid ? er[static_cast<size_t>(r) * H + h] : 0.f;
    float m = -FLT_MAX, l = 0.f, s_first = 0.f;
    constexpr int kBatch = G == kHeavy ? 8 : 4;
    for (int32_t j = j_first; j < hi; j += kBatch * G) {
      float x[kBatch];
#pragma unroll
      for (int q = 0; q < kBatch; ++q) {
        const int32_t jj = j + q * G;
        x[q] = jj < hi ? el[static_cast<size_t>(indices[jj]) * H + h] + erh : 0.f;
      }
#pragma unroll
      for (int q = 0; q < kBatch; ++q) {
        const int32_t jj = j + q * G;
        if (jj >= hi) break;
        if (raw) raw[static_cast<size_t>(h) * E + jj] = x[q];
        const float sc = x[q] > 0.f ? x[q] : x[q] * slope;
        if (jj == j_first) s_first = sc; else a[jj] = sc;
        if (sc > m) {
          l = l * expf(m - sc) + 1.0f;
          m = sc;
        } else {
          l += expf(sc - m);
        }
      }
    }
    const float big = group_max<G>(m, s_red);
    const float total = group_sum<G>(l * expf(m - big), s_red);
    const float inv = 1.0f / total;
    if (j_first < hi) a[j_first] = expf(s_first - big) * inv;
    for (int32_t j = j_first + G; j < hi; j += 4 * G) {
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = (j + q * G < hi) ? a[j + q * G] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (j + q * G < hi) a[j + q * G] = expf(v[q] - big) * inv;
    }
  }
}

// d_raw = a * (d_a - sum_k a_k d_a_k) * LeakyReLU'(raw);  d_er[r] = sum_j d_raw_j
template <int G>
__global__ __launch_bounds__(G == kHeavy ? kHeavy : kLightBlock) void gat_softmax_bwd_kernel(
    const int32_t* __restrict__ indptr, int32_t n_rows, int64_t E, const float* __restrict__ attn,
    const float* __restrict__ raw, const float* __restrict__ d_attn, int32_t H, float slope,
    float* __restrict__ d_raw, float* __restrict__ d_er, const int32_t* __restrict__ heavy_rows,
    int32_t heavy_thresh) {
  __shared__ float s_red[kHeavy / 64];
  int32_t r, lo, hi;
  int lane_g;
  const bool valid = pick_row<G>(indptr, n_rows, heavy_rows, heavy_thresh, &r, &lo, &hi, &lane_g);
  for (int h = 0; h < H; ++h) {
    const size_t base = static_cast<size_t>(h) * E;
    float t = 0.f;
    for (int32_t j = lo + lane_g; j < hi; j += G) t += attn[base + j] * d_attn[base + j];
    t = group_sum<G>(t, s_red);
    float acc = 0.f;
    for (int32_t j = lo + lane_g; j < hi; j += G) {
      const float ds = attn[base + j] * (d_attn[base + j] - t);
      const float de = raw[base + j] > 0.f ? ds : ds * slope;
      d_raw[base + j] = de;
      acc += de;
    }
    acc = group_sum<G>(acc, s_red);
    if (valid && lane_g == 0) d_er[static_cast<size_t>(r) * H + h] = acc;
  }
}

// out[r, h] = sum over the row's entries of vals[h, j]
template <int G>
__global__ __launch_bounds__(G == kHeavy ? kHeavy : kLightBlock) void segment_sum_kernel(
    const int32_t* __restrict__ indptr, int32_t n_rows, int64_t E, const float* __restrict__ vals,
    int32_t H, float* __restrict__ out, const int32_t* __restrict__ heavy_rows,
    int32_t heavy_thresh) {
  __shared__ float s_red[kHeavy / 64];
  int32_t r, lo, hi;
  int lane_g;
  const bool valid = pick_row<G>(indptr, n_rows, heavy_rows, heavy_thresh, &r, &lo, &hi, &lane_g);
  for (int h = 0; h < H; ++h) {
    const size_t base = static_cast<size_t>(h) * E;
    float acc = 0.f;
    for (int32_t j = lo + lane_g; j < hi; j += G) acc += vals[base + j];
    acc = group_sum<G>(acc, s_red);
    if (valid && lane_g == 0) out[static_cast<size_t>(r) * H + h] = acc;
  }
}

// ---- per-edge dot products ---------------------------------------------------------------
// out[h, j] = < A[row(j), h*D:(h+1)*D], B[indices[j], h*D:(h+1)*D] >.  Work is split by
// edge position, not by row (no reduction crosses edges), so hub rows need no special
// path: a wave owns kEdgesPerWave consecutive CSR positions, finds its first row by
// binary search and walks on, reloading its register copy of the A row when the row
// changes.  Lane l holds features l, l+64, ...: every row load is one coalesced 256-byte
// request per 64 features.  Heads are lane groups (D < 64, a power of two) or whole
// 64-feature tiles (D a multiple of 64).
constexpr int kEdgesPerWave = 256;
constexpr int kSddmmBlock = 256;
constexpr int kSddmmMaxHeads = 8;

template <int T>
__global__ __launch_bounds__(kSddmmBlock) void sddmm_dot_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int32_t n_rows,
    int64_t E, const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
    int32_t H, int32_t D, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * (kSddmmBlock / 64) + (threadIdx.x >> 6);
  const int64_t j0 = wave * kEdgesPerWave;
  if (j0 >= E) return;
  const int64_t j1 = (j0 + kEdgesPerWave < E) ? j0 + kEdgesPerWave : E;
  const int32_t F = H * D;
  const int seg = D < 64 ? D : 64;  // lanes that share a head inside one 64-feature tile
  const int tiles_per_head = D >= 64 ? D / 64 : 1;
  // first row of the wave: largest r with indptr[r] <= j0 (skips empty rows)
  int32_t r_first = 0;
  {
    int32_t lo = 0, hi = n_rows - 1;
    while (lo < hi) {
      const int32_t m = lo + ((hi - lo + 1) >> 1);
      if (indptr[m] <= j0) lo = m; else hi = m - 1;
    }
    r_first = lo;
  }
  // per-wave staging of the 64 results of a group (per head), stored coalesced afterwards
  __shared__ float s_out[kSddmmBlock / 64][kSddmmMaxHeads][64];
  float(*stage)[64] = s_out[threadIdx.x >> 6];
  float a[T];
  int32_t cur = -1;
  for (int64_t base = j0; base < j1; base += 64) {
    // one coalesced load of 64 neighbour ids; every lane finds the row of its own edge
    const int64_t jl = base + lane;
    const bool live = jl < j1;
    const int32_t my_src = live ? indices[jl] : 0;
    int32_t my_row = r_first;
    if (live) {
      int32_t lo = r_first, hi = n_rows - 1;
      while (lo < hi) {
        const int32_t m = lo + ((hi - lo + 1) >> 1);
        if (indptr[m] <= jl) lo = m; else hi = m - 1;
      }
      my_row = lo;
    }
    const int count = (j1 - base < 64) ? static_cast<int>(j1 - base) : 64;
    for (int k4 = 0; k4 < count; k4 += 4) {
      // four independent row gathers in flight before any of them is reduced
      float p[4][T];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kk = (k4 + q < count) ? k4 + q : count - 1;
        const float* brow = B + static_cast<size_t>(__builtin_amdgcn_readlane(my_src, kk)) * ldb;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const int f = lane + 64 * t;
          p[q][t] = brow[f < F ? f : 0];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (k4 + q >= count) break;
        const int32_t r = __builtin_amdgcn_readlane(my_row, k4 + q);
        if (r != cur) {  // wave-uniform: the register copy of the destination row changes
          cur = r;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const int f = lane + 64 * t;
            a[t] = f < F ? A[static_cast<size_t>(r) * lda + f] : 0.f;
          }
        }
        if (D >= 64) {
          // tiles of one head are summed first, then one wave reduction per head
#pragma unroll
          for (int t0 = 0; t0 < T; ++t0) {
            if (t0 % tiles_per_head != 0 || t0 * 64 >= F) continue;
            float v = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t)
              if (t >= t0 && t < t0 + tiles_per_head) v += a[t] * p[q][t];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) stage[t0 / tiles_per_head][k4 + q] = v;
          }
        } else {
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const int f = lane + 64 * t;
            float v = f < F ? a[t] * p[q][t] : 0.f;
            for (int off = seg / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off);
            if (f < F && (lane % seg) == 0) stage[f / D][k4 + q] = v;
          }
        }
      }
    }
    for (int h = 0; h < H; ++h)
      if (lane < count) out[static_cast<size_t>(h) * E + base + lane] = stage[h][lane];
  }
}

int check_rows(int64_t n_rows, int64_t E, int32_t H, const char* who) {
  SPMI_REQUIRE(n_rows >= 0 && E >= 0 && H >= 1 && H <= 64, SPMI_ERR_INVALID_ARG,
               "%s: bad sizes n_rows=%lld E=%lld H=%d", who, (long long)n_rows, (long long)E, H);
  SPMI_REQUIRE(n_rows < (int64_t{1} << 31) && E < (int64_t{1} << 31), SPMI_ERR_UNSUPPORTED,
               "%s: size beyond the int32 on-device limits", who);
  return SPMI_OK;
}

int check_heavy(const int32_t* heavy_rows, int32_t n_heavy, int32_t heavy_thresh, const char* who) {
  SPMI_REQUIRE(n_heavy >= 0 && (n_heavy == 0 || (heavy_rows && heavy_thresh >= 1)),
               SPMI_ERR_INVALID_ARG, "%s: hub-row list given without its buffer", who);
  return SPMI_OK;
}

unsigned light_grid(int64_t n_rows) {
  const int rows_per_block = kLightBlock / kLight;
  return static_cast<unsigned>((n_rows + rows_per_block - 1) / rows_per_block);
}

}  // namespace

extern "C" {

int spmi_gat_edge_softmax(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t E,
                          const float* el, const float* er, int32_t H, float slope, float* attn,
                          float* raw, const int32_t* heavy_rows, int32_t n_heavy,
                          int32_t heavy_thresh, void* stream) {
  if (int rc = check_rows(n_rows, E, H, "spmi_gat_edge_softmax")) return rc;
  if (int rc = check_heavy(heavy_rows, n_heavy, heavy_thresh, "spmi_gat_edge_softmax")) return rc;
  if (n_rows == 0 || E == 0) return SPMI_OK;
  SPMI_REQUIRE(indptr && indices && el && er && attn, SPMI_ERR_INVALID_ARG,
               "spmi_gat_edge_softmax: null pointer argument");
  hipStream_t st = spmi::as_stream(stream);
  const int32_t thresh = n_heavy > 0 ? heavy_thresh : INT32_MAX;
  gat_softmax_fwd_kernel<kLight><<<light_grid(n_rows), kLightBlock, 0, st>>>(
      indptr, indices, (int32_t)n_rows, E, el, er, H, slope, attn, raw, nullptr, thresh);
  SPMI_CHECK_LAUNCH("gat_softmax_fwd");
  if (n_heavy > 0) {
    gat_softmax_fwd_kernel<kHeavy><<<n_heavy, kHeavy, 0, st>>>(
        indptr, indices, (int32_t)n_rows, E, el, er, H, slope, attn, raw, heavy_rows, thresh);
    SPMI_CHECK_LAUNCH("gat_softmax_fwd(hub rows)");
  }
  return SPMI_OK;
}

int spmi_gat_edge_softmax_bwd(const int32_t* indptr, int64_t n_rows, int64_t E, const float* attn,
                              const float* raw, const float* d_attn, int32_t H, float slope,
                              float* d_raw, float* d_er, const int32_t* heavy_rows,
                              int32_t n_heavy, int32_t heavy_thresh, void* stream) {
  if (int rc = check_rows(n_rows, E, H, "spmi_gat_edge_softmax_bwd")) return rc;
  if (int rc = check_heavy(heavy_rows, n_heavy, heavy_thresh, "spmi_gat_edge_softmax_bwd")) return rc;
  if (n_rows == 0) return SPMI_OK;
  SPMI_REQUIRE(indptr && d_er && (E == 0 || (attn && raw && d_attn && d_raw)), SPMI_ERR_INVALID_ARG,
               "spmi_gat_edge_softmax_bwd: null pointer argument");
  hipStream_t st = spmi::as_stream(stream);
  const int32_t thresh = n_heavy > 0 ? heavy_thresh : INT32_MAX;
  gat_softmax_bwd_kernel<kLight><<<light_grid(n_rows), kLightBlock, 0, st>>>(
      indptr, (int32_t)n_rows, E, attn, raw, d_attn, H, slope, d_raw, d_er, nullptr, thresh);
  SPMI_CHECK_LAUNCH("gat_softmax_bwd");
  if (n_heavy > 0) {
    gat_softmax_bwd_kernel<kHeavy><<<n_heavy, kHeavy, 0, st>>>(
        indptr, (int32_t)n_rows, E, attn, raw, d_attn, H, slope, d_raw, d_er, heavy_rows, thresh);
    SPMI_CHECK_LAUNCH("gat_softmax_bwd(hub rows)");
  }
  return SPMI_OK;
}

int spmi_segment_sum(const int32_t* indptr, int64_t n_rows, int64_t E, const float* vals, int32_t H,
                     float* out, const int32_t* heavy_rows, int32_t n_heavy, int32_t heavy_thresh,
                     void* stream) {
  if (int rc = check_rows(n_rows, E, H, "spmi_segment_sum")) return rc;
  if (int rc = check_heavy(heavy_rows, n_heavy, heavy_thresh, "spmi_segment_sum")) return rc;
  if (n_rows == 0) return SPMI_OK;
  SPMI_REQUIRE(indptr && out && (E == 0 || vals), SPMI_ERR_INVALID_ARG,
               "spmi_segment_sum: null pointer argument");
  hipStream_t st = spmi::as_stream(stream);
  const int32_t thresh = n_heavy > 0 ? heavy_thresh : INT32_MAX;
  segment_sum_kernel<kLight><<<light_grid(n_rows), kLightBlock, 0, st>>>(
      indptr, (int32_t)n_rows, E, vals, H, out, nullptr, thresh);
  SPMI_CHECK_LAUNCH("segment_sum");
  if (n_heavy > 0) {
    segment_sum_kernel<kHeavy><<<n_heavy, kHeavy, 0, st>>>(indptr, (int32_t)n_rows, E, vals, H, out,
                                                           heavy_rows, thresh);
    SPMI_CHECK_LAUNCH("segment_sum(hub rows)");
  }
  return SPMI_OK;
}

int spmi_sddmm_dot_supported(int32_t H, int32_t D) {
  if (H < 1 || H > kSddmmMaxHeads || D < 1 || static_cast<int64_t>(H) * D > 512) return 0;
  if (D >= 64) return D % 64 == 0;
  return (D & (D - 1)) == 0;
}

int spmi_sddmm_dot(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t E,
                   const float* A, int64_t lda, const float* B, int64_t ldb, int32_t H, int32_t D,
                   float* out, void* stream) {
  if (int rc = check_rows(n_rows, E, H, "spmi_sddmm_dot")) return rc;
  SPMI_REQUIRE(spmi_sddmm_dot_supported(H, D), SPMI_ERR_UNSUPPORTED,
               "spmi_sddmm_dot: head width D=%d must be a power of two below 64 or a multiple of "
               "64, with H <= 8 and H*D <= 512 (H=%d)", D, H);
  const int64_t F = static_cast<int64_t>(H) * D;
  SPMI_REQUIRE(lda >= F && ldb >= F, SPMI_ERR_INVALID_ARG,
               "spmi_sddmm_dot: leading dimensions (%lld,%lld) smaller than H*D=%lld",
               (long long)lda, (long long)ldb, (long long)F);
  if (n_rows == 0 || E == 0) return SPMI_OK;
  SPMI_REQUIRE(indptr && indices && A && B && out, SPMI_ERR_INVALID_ARG,
               "spmi_sddmm_dot: null pointer argument");
  hipStream_t st = spmi::as_stream(stream);
  const int64_t waves = (E + kEdgesPerWave - 1) / kEdgesPerWave;
  const unsigned grid = static_cast<unsigned>((waves + kSddmmBlock / 64 - 1) / (kSddmmBlock / 64));
  const int tiles = static_cast<int>((F + 63) / 64);
#define SPMI_SDDMM(T)                                                                         \
  sddmm_dot_kernel<T><<<grid, kSddmmBlock, 0, st>>>(indptr, indices, (int32_t)n_rows, E, A, lda, \
                                                    B, ldb, H, D, out)
  if (tiles <= 1) SPMI_SDDMM(1);
  else if (tiles <= 2) SPMI_SDDMM(2);
  else if (tiles <= 4) SPMI_SDDMM(4);
  else SPMI_SDDMM(8);
#undef SPMI_SDDMM
  SPMI_CHECK_LAUNCH("sddmm_dot");
  return SPMI_OK;
}

}  // extern "C"
