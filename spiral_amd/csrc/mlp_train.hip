// Training step of the narrow layers of the output head: [Linear, BatchNorm1d (batch statistics),
// ReLU, Dropout] blocks of widths <= 32 over a million rows, forward and backward
// (spiro/dtdg/models/decoder/torch/simpleDecoder.py:60-72 under the training loop of
// examples/torch/citation_prediction.py:96-100).
//
// As torch modules every one of these layers is a handful of launches over [rows, 5..20]
// matrices (a library GEMM tiled for square problems, three BatchNorm kernels, clamp, dropout,
// and their backward counterparts), each far from memory speed: 3 ms of a 17 ms step for
// 0.3 GB of data.  Here a layer is one pass per direction plus two tiny launches:
//   forward   z_next = Linear(dropout(relu(bn(z)))) row by row, the batch statistics of z_next
//             collected on the way (shifted sums per wave, combined in float64);
//   backward  pass A: recompute the activations of a row, push the incoming gradient through
//             the Linear (dW, db as per-wave partial sums), the dropout mask and the ReLU, and
//             collect the two sums BatchNorm's backward needs; pass B: dz from those sums.
// One thread owns a row (its <= 32 values live in registers), so a wave's instruction stream is
// shared by 64 rows; weights sit in LDS and are read as broadcasts.  Every sum over rows is
// taken in a fixed order (lane, wave, workgroup, then a last kernel over the workgroups'
// partials): results are bit-reproducible.  The grids hold several waves per SIMD: a wave's
// rows are a chain of load, arithmetic and store with nothing overlapped, so the memory
// latency is covered by other waves.  All HBM-bound: rows * (wi + wo [+ wi mask]) * 4 bytes per pass.

#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxWidth = 32;
constexpr int kMaxBlocks = 1024;  // persistent grids: one set of partial sums per workgroup

// Small read-only parameters (weights, statistics) addressed through the constant address space:
// the compiler then fetches them with scalar loads into scalar registers, which the vector
// multiply-adds take as operands.  Only for memory that no kernel in flight writes.
using cfloat = const __attribute__((address_space(4))) float;
__device__ __forceinline__ cfloat* as_constant(const float* q) {
  return reinterpret_cast<cfloat*>(reinterpret_cast<uintptr_t>(q));
}
using f32x16 = float __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// The waves of a workgroup add their sums in wave order and the workgroup writes one set.
template <int NW, int S>
__device__ __forceinline__ void block_sums(float (*area)[S], float* __restrict__ dst) {
  __syncthreads();
  for (int v = threadIdx.x; v < S; v += NW * 64) {
    float t = area[0][v];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += area[w][v];
    dst[v] = t;
  }
}

// ---- column statistics of a matrix that another kernel wrote --------------------------------
// partial[block][0..W) = sum (z - c), partial[block][32..32+W) = sum (z - c)^2, c = row 0.
// Also copies the matrix into rows of P floats (zero past W) for the layer kernels.
template <int P>
__global__ __launch_bounds__(kThreads) void col_stats_kernel(const float* __restrict__ z, int64_t ld,
                                                              int64_t M, int W, float* __restrict__ partial,
                                                              float* __restrict__ packed) {
  __shared__ float area[kWaves][2 * kMaxWidth];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * kWaves + wv;
  const int n_waves = gridDim.x * kWaves;
  float s1[P], s2[P], c[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    s1[j] = s2[j] = 0.0f;
    c[j] = j < W ? z[j] : 0.0f;
  }
  for (int64_t row = static_cast<int64_t>(wave) * 64 + lane; row < M; row += static_cast<int64_t>(n_waves) * 64) {
    const float4* zr = reinterpret_cast<const float4*>(z + row * ld);
    float v[P];
#pragma unroll
    for (int q = 0; q < P / 4; ++q) {
      const float4 t = zr[q];
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int j = 0; j < P; ++j) {
      v[j] = j < W ? v[j] : 0.0f;
      const float d = v[j] - c[j];
      s1[j] += d;
      s2[j] += d * d;
    }
    float4* dst = reinterpret_cast<float4*>(packed + row * P);
#pragma unroll
    for (int q = 0; q < P / 4; ++q) dst[q] = float4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
  }
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const float a = wave_sum(s1[j]), b = wave_sum(s2[j]);
    if (lane == 0) {
      area[wv][j] = a;
      area[wv][kMaxWidth + j] = b;
    }
  }
  if (lane >= P && lane < kMaxWidth) area[wv][lane] = area[wv][kMaxWidth + lane] = 0.0f;
  block_sums<kWaves, 2 * kMaxWidth>(area, partial + static_cast<size_t>(blockIdx.x) * 2 * kMaxWidth);
}

// Sum over the waves' partials of 64 consecutive values, in a fixed order: 16 segments of waves
// are added up side by side (thread = value x segment, four loads in flight), then the
// segments in segment order.  Returns the total of value `threadIdx.x % 64` to the threads of
// segment 0 (threadIdx.x < 64); 1024 threads.
template <typename T>
__device__ __forceinline__ T sum_over_waves(const float* __restrict__ partial, int n_waves, int stride, int v,
                                            bool valid, T (*scratch)[64]) {
  const int seg = threadIdx.x >> 6, vl = threadIdx.x & 63;
  const int per = (n_waves + 15) / 16;
  const int w0 = seg * per, w1 = (w0 + per < n_waves) ? w0 + per : n_waves;
  T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  if (valid) {
    int w = w0;
    for (; w + 4 <= w1; w += 4) {
      a0 += static_cast<T>(partial[static_cast<size_t>(w) * stride + v]);
      a1 += static_cast<T>(partial[static_cast<size_t>(w + 1) * stride + v]);
      a2 += static_cast<T>(partial[static_cast<size_t>(w + 2) * stride + v]);
      a3 += static_cast<T>(partial[static_cast<size_t>(w + 3) * stride + v]);
    }
    for (; w < w1; ++w) a0 += static_cast<T>(partial[static_cast<size_t>(w) * stride + v]);
  }
  scratch[seg][vl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  T total = 0;
  if (seg == 0)
    for (int q = 0; q < 16; ++q) total += scratch[q][vl];
  return total;
}

// ---- statistics -> the affine form the layer kernels apply ----------------------------------
// One block: the shifted sums of the 2 x 32 columns are added over the waves in float64, then
// one thread per column writes mean, rstd (for the backward), scale = gamma * rstd,
// shift = beta - mean * scale, and moves the running statistics as torch does (unbiased
// variance, momentum).  pivot[j] is the value the sums were shifted by.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial, int n_waves,
                                                            const float* __restrict__ pivot, int64_t M, int W,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            float momentum, float* running_mean,
                                                            float* running_var, float* __restrict__ stats) {
  __shared__ double scratch[16][64];
  __shared__ double total[64];
  const int v = threadIdx.x & 63;
  const double t = sum_over_waves<double>(partial, n_waves, 2 * kMaxWidth, v, (v & 31) < W, scratch);
  if (threadIdx.x < 64) total[threadIdx.x] = t;
  __syncthreads();
  const int j = threadIdx.x;
  if (j >= W) return;
  const double dm = total[j] / static_cast<double>(M);
  const double mean = static_cast<double>(pivot[j]) + dm;
  double var = total[kMaxWidth + j] / static_cast<double>(M) - dm * dm;
  var = var < 0.0 ? 0.0 : var;
  const float rstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
  const float g = gamma ? gamma[j] : 1.0f, be = beta ? beta[j] : 0.0f;
  const float scale = g * rstd;
  stats[j] = static_cast<float>(mean);
  stats[kMaxWidth + j] = rstd;
  stats[2 * kMaxWidth + j] = scale;
  stats[3 * kMaxWidth + j] = be - static_cast<float>(mean) * scale;
  if (running_mean) running_mean[j] = (1.0f - momentum) * running_mean[j] + momentum * static_cast<float>(mean);
  if (running_var) {
    const double unbiased = M > 1 ? var * static_cast<double>(M) / static_cast<double>(M - 1) : var;
    running_var[j] = (1.0f - momentum) * running_var[j] + momentum * static_cast<float>(unbiased);
  }
}

// Intermediate matrices of the head (z, dy, dz, the incoming gradient g) are stored with their
// rows padded to 8, 16 or 32 floats, the padding holding zeros: rows are read and written as
// whole float4 vectors and no kernel below has a width test in its row loop.  Only the dropout
// factors keep torch's own [rows, width] layout (they come out of torch's generator).
struct LayerArgs {
  const float* z;      // [M, WI]: pre-normalisation input of the block
  int64_t M;
  int wi, wo;          // true widths; WI, WO (template) are the padded ones
  const float* stats;  // mean | rstd | scale | shift, 32 floats each (bn_finalize_kernel), zero past wi
  const float* mask;   // nullable [M, wi]: the dropout factors (0 or 1 / (1 - p))
  const float* W;      // [WO, WI]: weight of the Linear that follows, zero-padded
  const float* b;      // nullable [WO], zero-padded
  float* out;          // forward: [M, WO]
  float* partial;      // forward: nullable, column statistics of out; backward: the sums
  const float* g;      // backward: gradient of out [M, WO]
  float* dy;           // backward: [M, WI], gradient of the BatchNorm output
};

template <int W>
__device__ __forceinline__ void load_row(const float* base, int64_t row, float (&v)[W]) {
  const float4* src = reinterpret_cast<const float4*>(base + row * W);
#pragma unroll
  for (int q = 0; q < W / 4; ++q) {
    const float4 t = src[q];
    v[4 * q] = t.x;
    v[4 * q + 1] = t.y;
    v[4 * q + 2] = t.z;
    v[4 * q + 3] = t.w;
  }
}

template <int W>
__device__ __forceinline__ void store_row(float* base, int64_t row, const float (&v)[W]) {
  float4* dst = reinterpret_cast<float4*>(base + row * W);
#pragma unroll
  for (int q = 0; q < W / 4; ++q) dst[q] = float4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}

// The dropout factors of a row; past the true width the last factor is read again (its
// activation is zero there).
template <int WI>
__device__ __forceinline__ void load_mask(const LayerArgs& p, int64_t row, float (&m)[WI]) {
  if (p.mask) {
    const float* mr = p.mask + row * p.wi;
#pragma unroll
    for (int j = 0; j < WI; ++j) m[j] = mr[j < p.wi ? j : p.wi - 1];
  } else {
#pragma unroll
    for (int j = 0; j < WI; ++j) m[j] = 1.0f;
  }
}

// ---- forward of one block --------------------------------------------------------------------
// Weights, bias, scale and shift are read with wave-uniform addresses, i.e. by scalar loads
// into scalar registers that the multiply-adds take as operands: no vector register and no LDS
// traffic is spent on them.
template <int WI, int WO>
__global__ __launch_bounds__(kThreads) void bn_act_linear_fwd_kernel(LayerArgs p) {
  __shared__ float pivot_s[WO];
  cfloat* Wp = as_constant(p.W);
  cfloat* bp = as_constant(p.b);
  cfloat* sc = as_constant(p.stats + 2 * kMaxWidth);
  cfloat* sh = as_constant(p.stats + 3 * kMaxWidth);
  auto row_out = [&](int64_t row, float (&out)[WO]) {
    float a[WI], m[WI];
    load_row<WI>(p.z, row, a);
    load_mask<WI>(p, row, m);
#pragma unroll
    for (int j = 0; j < WI; ++j) a[j] = fmaxf(a[j] * sc[j] + sh[j], 0.0f) * m[j];
#pragma unroll
    for (int o = 0; o < WO; ++o) {
      float acc = p.b ? bp[o] : 0.0f;
#pragma unroll
      for (int j = 0; j < WI; ++j) acc += Wp[o * WI + j] * a[j];
      out[o] = acc;
    }
  };
  // the statistics of the output are shifted by its first row, which every block works out
  if (p.partial) {
    float first[WO];
    row_out(0, first);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int o = 0; o < WO; ++o) pivot_s[o] = first[o];
    }
    __syncthreads();
  }

  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * kWaves + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * kWaves;
  float s1[WO], s2[WO];
#pragma unroll
  for (int o = 0; o < WO; ++o) s1[o] = s2[o] = 0.0f;
  for (int64_t row = static_cast<int64_t>(wave) * 64 + lane; row < p.M; row += static_cast<int64_t>(n_waves) * 64) {
    // the weights are fetched again for every group of rows (scalar loads, cached): kept across
    // the loop they would be moved into WI * WO vector registers
    asm volatile("" ::: "memory");
    float out[WO];
    row_out(row, out);
    store_row<WO>(p.out, row, out);
    if (p.partial) {
#pragma unroll
      for (int o = 0; o < WO; ++o) {
        const float d = out[o] - pivot_s[o];
        s1[o] += d;
        s2[o] += d * d;
      }
    }
  }
  if (p.partial) {
    __shared__ float area[kWaves][2 * kMaxWidth];
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 0; o < WO; ++o) {
      const float a = wave_sum(s1[o]), b = wave_sum(s2[o]);
      if (lane == 0) {
        area[wv][o] = a;
        area[wv][kMaxWidth + o] = b;
      }
    }
    if (lane >= WO && lane < kMaxWidth) area[wv][lane] = area[wv][kMaxWidth + lane] = 0.0f;
    block_sums<kWaves, 2 * kMaxWidth>(area, p.partial + static_cast<size_t>(blockIdx.x) * 2 * kMaxWidth);
  }
}

// ---- backward, pass A -------------------------------------------------------------------------
// Partial sums, kSums floats per workgroup: d beta [32] | d gamma [32] | db [32] | dW [32 x 32]
// (dW[o][j] at 96 + o * 32 + j).  dW = g^T a, a [32 x 32] product over the wave's rows, runs on
// the fp32 matrix instruction: the rows of g and of the activations are staged in LDS row by
// row and read back two rows at a time in the instruction's operand layout (lane l: row l / 32
// of the pair, column l % 32).
constexpr int kSums = 3 * kMaxWidth + kMaxWidth * kMaxWidth;
constexpr int kBwdWaves = 2;  // the staging areas of more waves do not fit the static LDS limit

template <int WI, int WO>
__global__ __launch_bounds__(kBwdWaves * 64) void bn_act_linear_bwd_kernel(LayerArgs p) {
  __shared__ float stage_g[kBwdWaves][64][kMaxWidth + 1], stage_a[kBwdWaves][64][kMaxWidth + 1];
  __shared__ float area[kBwdWaves][kSums];
  cfloat* Wp = as_constant(p.W);
  cfloat* mean_s = as_constant(p.stats);
  cfloat* rstd_s = as_constant(p.stats + kMaxWidth);
  cfloat* sc_s = as_constant(p.stats + 2 * kMaxWidth);
  cfloat* sh_s = as_constant(p.stats + 3 * kMaxWidth);
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  // columns past the padded widths stay zero for the whole kernel
  for (int c = 0; c <= kMaxWidth; ++c) stage_g[wv][lane][c] = stage_a[wv][lane][c] = 0.0f;
  __syncthreads();
  const int wave = blockIdx.x * kBwdWaves + wv;
  const int n_waves = gridDim.x * kBwdWaves;
  const int lr = lane & 31, lh = lane >> 5;

  float d_beta[WI], d_gamma[WI], d_b[WO];
  f32x16 d_w;
#pragma unroll
  for (int j = 0; j < WI; ++j) d_beta[j] = d_gamma[j] = 0.0f;
#pragma unroll
  for (int o = 0; o < WO; ++o) d_b[o] = 0.0f;
#pragma unroll
  for (int e = 0; e < 16; ++e) d_w[e] = 0.0f;

  const int64_t n_groups = (p.M + 63) / 64;
  for (int64_t grp = wave; grp < n_groups; grp += n_waves) {
    asm volatile("" ::: "memory");  // weights are fetched again for every group, see the forward kernel
    const int64_t row = grp * 64 + lane;
    const bool live = row < p.M;
    const int64_t rr = live ? row : p.M - 1;  // rows past the end: a valid row, gradient zeroed
    float z[WI], m[WI], g[WO], dy[WI];
    load_row<WI>(p.z, rr, z);
    load_mask<WI>(p, rr, m);
    load_row<WO>(p.g, rr, g);
    if (!live) {
#pragma unroll
      for (int o = 0; o < WO; ++o) g[o] = 0.0f;
    }
#pragma unroll
    for (int o = 0; o < WO; ++o) {
      d_b[o] += g[o];
      stage_g[wv][lane][o] = g[o];
    }
    // gradient of the BatchNorm output: through the Linear, the dropout factors, the ReLU
#pragma unroll
    for (int j = 0; j < WI; ++j) {
      float da = 0.0f;
#pragma unroll
      for (int o = 0; o < WO; ++o) da += g[o] * Wp[o * WI + j];
      const float y = z[j] * sc_s[j] + sh_s[j];
      stage_a[wv][lane][j] = fmaxf(y, 0.0f) * m[j];
      dy[j] = y > 0.0f ? da * m[j] : 0.0f;
      d_beta[j] += dy[j];
      // x_hat from the statistics, not from y: gamma may be zero
      d_gamma[j] += dy[j] * ((z[j] - mean_s[j]) * rstd_s[j]);
    }
    if (live) store_row<WI>(p.dy, row, dy);
    __builtin_amdgcn_wave_barrier();
#pragma unroll 8
    for (int k2 = 0; k2 < 32; ++k2) {
      const float ga = stage_g[wv][2 * k2 + lh][lr];
      const float ab = stage_a[wv][2 * k2 + lh][lr];
      d_w = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, ab, d_w, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }

#pragma unroll
  for (int j = 0; j < WI; ++j) {
    const float s0 = wave_sum(d_beta[j]), s1 = wave_sum(d_gamma[j]);
    if (lane == 0) {
      area[wv][j] = s0;
      area[wv][kMaxWidth + j] = s1;
    }
  }
#pragma unroll
  for (int o = 0; o < WO; ++o) {
    const float s0 = wave_sum(d_b[o]);
    if (lane == 0) area[wv][2 * kMaxWidth + o] = s0;
  }
  if (lane >= WI && lane < kMaxWidth) area[wv][lane] = area[wv][kMaxWidth + lane] = 0.0f;
  if (lane >= WO && lane < kMaxWidth) area[wv][2 * kMaxWidth + lane] = 0.0f;
  // element e of lane l of the product: row (e&3) + 8 (e>>2) + 4 (l>>5) = o, column l & 31 = j
#pragma unroll
  for (int e = 0; e < 16; ++e)
    area[wv][3 * kMaxWidth + ((e & 3) + 8 * (e >> 2) + 4 * lh) * kMaxWidth + lr] = d_w[e];
  block_sums<kBwdWaves, kSums>(area, p.partial + static_cast<size_t>(blockIdx.x) * kSums);
}

// out[v] = sum over waves of partial[wave][v]; a block per 64 values
__global__ __launch_bounds__(1024) void sum_partials_kernel(const float* __restrict__ partial, int n_waves,
                                                             int stride, int n_values,
                                                             float* __restrict__ out) {
  __shared__ float scratch[16][64];
  const int v = blockIdx.x * 64 + (threadIdx.x & 63);
  const float t = sum_over_waves<float>(partial, n_waves, stride, v, v < n_values, scratch);
  if (threadIdx.x < 64 && v < n_values) out[v] = t;
}

// ---- backward, pass B: dz = scale * (dy - d_beta / M - x_hat * d_gamma / M) --------------------
// Past the true width scale is zero and so is dz.
template <int WP>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(const float* __restrict__ dy,
                                                                 const float* __restrict__ z, int64_t M,
                                                                 const float* __restrict__ stats,
                                                                 const float* __restrict__ sums,
                                                                 float* __restrict__ dz) {
  const int64_t row = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (row >= M) return;
  const float inv_m = 1.0f / static_cast<float>(M);
  float d[WP], x[WP];
  load_row<WP>(dy, row, d);
  load_row<WP>(z, row, x);
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    const float x_hat = (x[j] - stats[j]) * stats[kMaxWidth + j];
    d[j] = stats[2 * kMaxWidth + j] * (d[j] - sums[j] * inv_m - x_hat * sums[kMaxWidth + j] * inv_m);
  }
  store_row<WP>(dz, row, d);
}

// Workgroups of a persistent grid over groups of 64 rows: a group per wave while that stays
// within kMaxBlocks workgroups.
int grid_blocks(int64_t M, int waves_per_block) {
  const int64_t groups = (M + 63) / 64;
  const int64_t blocks = (groups + waves_per_block - 1) / waves_per_block;
  return static_cast<int>(blocks < kMaxBlocks ? blocks : kMaxBlocks);
}

bool pad_ok(int64_t w, int64_t pad) { return (pad == 8 || pad == 16 || pad == 32) && w >= 1 && w <= pad; }

template <template <int, int> class Launcher>
int dispatch(int pi, int po, const LayerArgs& a, hipStream_t st) {
#define SPMI_CASE(I, O) \
  if (pi == I && po == O) return Launcher<I, O>::run(a, st);
  SPMI_CASE(8, 8) SPMI_CASE(8, 16) SPMI_CASE(8, 32) SPMI_CASE(16, 8) SPMI_CASE(16, 16) SPMI_CASE(16, 32)
  SPMI_CASE(32, 8) SPMI_CASE(32, 16) SPMI_CASE(32, 32)
#undef SPMI_CASE
  return SPMI_ERR_UNSUPPORTED;
}

template <int WI, int WO>
struct FwdLauncher {
  static int run(const LayerArgs& a, hipStream_t st) {
    bn_act_linear_fwd_kernel<WI, WO><<<grid_blocks(a.M, kWaves), kThreads, 0, st>>>(a);
    return SPMI_OK;
  }
};
template <int WI, int WO>
struct BwdLauncher {
  static int run(const LayerArgs& a, hipStream_t st) {
    bn_act_linear_bwd_kernel<WI, WO><<<grid_blocks(a.M, kBwdWaves), kBwdWaves * 64, 0, st>>>(a);
    return SPMI_OK;
  }
};

int check_rows(const char* who, int64_t M) {
  SPMI_REQUIRE(M >= 1 && M < (int64_t{1} << 31), SPMI_ERR_INVALID_ARG, "%s: bad row count %lld", who, (long long)M);
  return SPMI_OK;
}

int check_layer(const char* who, int64_t M, int64_t wi, int64_t wi_pad, int64_t wo, int64_t wo_pad) {
  if (int rc = check_rows(who, M)) return rc;
  SPMI_REQUIRE(pad_ok(wi, wi_pad) && pad_ok(wo, wo_pad), SPMI_ERR_UNSUPPORTED,
               "%s: widths must be 1..32 in rows padded to 8, 16 or 32 (wi=%lld/%lld wo=%lld/%lld)", who,
               (long long)wi, (long long)wi_pad, (long long)wo, (long long)wo_pad);
  return SPMI_OK;
}

bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

}  // namespace

extern "C" {

size_t spmi_mlp_train_workspace_bytes(void) { return sizeof(float) * kMaxBlocks * kSums; }

int spmi_bn_col_stats(const float* z, int64_t ldz, int64_t M, int64_t W, int64_t w_pad, float* packed,
                      float* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_rows("spmi_bn_col_stats", M)) return rc;
  SPMI_REQUIRE(pad_ok(W, w_pad), SPMI_ERR_UNSUPPORTED, "spmi_bn_col_stats: width %lld in rows of %lld",
               (long long)W, (long long)w_pad);
  SPMI_REQUIRE(z && ws && packed && ldz >= w_pad && ldz % 4 == 0 && aligned16(z) && aligned16(packed),
               SPMI_ERR_INVALID_ARG, "spmi_bn_col_stats: null or misaligned pointer, or rows shorter than %lld",
               (long long)w_pad);
  SPMI_REQUIRE(ws_bytes >= spmi_mlp_train_workspace_bytes(), SPMI_ERR_WORKSPACE,
               "spmi_bn_col_stats: workspace %zu < %zu", ws_bytes, spmi_mlp_train_workspace_bytes());
  const int blocks = grid_blocks(M, kWaves);
  hipStream_t st = spmi::as_stream(stream);
  if (w_pad == 8) col_stats_kernel<8><<<blocks, kThreads, 0, st>>>(z, ldz, M, (int)W, ws, packed);
  else if (w_pad == 16) col_stats_kernel<16><<<blocks, kThreads, 0, st>>>(z, ldz, M, (int)W, ws, packed);
  else col_stats_kernel<32><<<blocks, kThreads, 0, st>>>(z, ldz, M, (int)W, ws, packed);
  SPMI_CHECK_LAUNCH("bn_col_stats");
  return SPMI_OK;
}

int spmi_bn_finalize(const float* ws, const float* pivot, int64_t M, int64_t W, const float* gamma,
                     const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                     float* stats, void* stream) {
  if (int rc = check_rows("spmi_bn_finalize", M)) return rc;
  SPMI_REQUIRE(W >= 1 && W <= kMaxWidth, SPMI_ERR_UNSUPPORTED, "spmi_bn_finalize: width %lld not in 1..32",
               (long long)W);
  SPMI_REQUIRE(ws && pivot && stats, SPMI_ERR_INVALID_ARG, "spmi_bn_finalize: null pointer");
  bn_finalize_kernel<<<1, 1024, 0, spmi::as_stream(stream)>>>(ws, grid_blocks(M, kWaves), pivot, M, (int)W,
                                                              gamma, beta, eps, momentum, running_mean,
                                                              running_var, stats);
  SPMI_CHECK_LAUNCH("bn_finalize");
  return SPMI_OK;
}

int spmi_bn_act_linear_fwd(const float* z, int64_t M, int64_t wi, int64_t wi_pad, const float* stats,
                           const float* mask, const float* W, const float* b, int64_t wo, int64_t wo_pad,
                           float* out, float* ws_stats, void* stream) {
  if (int rc = check_layer("spmi_bn_act_linear_fwd", M, wi, wi_pad, wo, wo_pad)) return rc;
  SPMI_REQUIRE(z && stats && W && out && aligned16(z) && aligned16(out), SPMI_ERR_INVALID_ARG,
               "spmi_bn_act_linear_fwd: null or misaligned pointer");
  LayerArgs a{};
  a.z = z; a.M = M; a.wi = (int)wi; a.wo = (int)wo; a.stats = stats; a.mask = mask;
  a.W = W; a.b = b; a.out = out; a.partial = ws_stats;
  const int rc = dispatch<FwdLauncher>((int)wi_pad, (int)wo_pad, a, spmi::as_stream(stream));
  SPMI_CHECK_LAUNCH("bn_act_linear_fwd");
  return rc;
}

int spmi_bn_act_linear_bwd(const float* z, int64_t M, int64_t wi, int64_t wi_pad, const float* stats,
                           const float* mask, const float* W, int64_t wo, int64_t wo_pad, const float* g,
                           float* dy, float* ws, size_t ws_bytes, float* sums, void* stream) {
  if (int rc = check_layer("spmi_bn_act_linear_bwd", M, wi, wi_pad, wo, wo_pad)) return rc;
  SPMI_REQUIRE(z && stats && W && g && dy && ws && sums && aligned16(z) && aligned16(g) && aligned16(dy),
               SPMI_ERR_INVALID_ARG, "spmi_bn_act_linear_bwd: null or misaligned pointer");
  SPMI_REQUIRE(ws_bytes >= spmi_mlp_train_workspace_bytes(), SPMI_ERR_WORKSPACE,
               "spmi_bn_act_linear_bwd: workspace %zu < %zu", ws_bytes, spmi_mlp_train_workspace_bytes());
  LayerArgs a{};
  a.z = z; a.M = M; a.wi = (int)wi; a.wo = (int)wo; a.stats = stats; a.mask = mask;
  a.W = W; a.g = g; a.dy = dy; a.partial = ws;
  hipStream_t st = spmi::as_stream(stream);
  const int rc = dispatch<BwdLauncher>((int)wi_pad, (int)wo_pad, a, st);
  SPMI_CHECK_LAUNCH("bn_act_linear_bwd");
  if (rc) return rc;
  sum_partials_kernel<<<(kSums + 63) / 64, 1024, 0, st>>>(ws, grid_blocks(M, kBwdWaves), kSums, kSums, sums);
  SPMI_CHECK_LAUNCH("mlp_train_sum_partials");
  return SPMI_OK;
}

int spmi_bn_bwd_apply(const float* dy, const float* z, int64_t M, int64_t W, int64_t w_pad, const float* stats,
                      const float* sums, float* dz, void* stream) {
  if (int rc = check_rows("spmi_bn_bwd_apply", M)) return rc;
  SPMI_REQUIRE(pad_ok(W, w_pad), SPMI_ERR_UNSUPPORTED, "spmi_bn_bwd_apply: width %lld in rows of %lld",
               (long long)W, (long long)w_pad);
  SPMI_REQUIRE(dy && z && stats && sums && dz && aligned16(dy) && aligned16(z) && aligned16(dz),
               SPMI_ERR_INVALID_ARG, "spmi_bn_bwd_apply: null or misaligned pointer");
  const unsigned blocks = static_cast<unsigned>((M + kThreads - 1) / kThreads);
  hipStream_t st = spmi::as_stream(stream);
  if (w_pad == 8) bn_bwd_apply_kernel<8><<<blocks, kThreads, 0, st>>>(dy, z, M, stats, sums, dz);
  else if (w_pad == 16) bn_bwd_apply_kernel<16><<<blocks, kThreads, 0, st>>>(dy, z, M, stats, sums, dz);
  else bn_bwd_apply_kernel<32><<<blocks, kThreads, 0, st>>>(dy, z, M, stats, sums, dz);
  SPMI_CHECK_LAUNCH("bn_bwd_apply");
  return SPMI_OK;
}

}  // extern "C"
