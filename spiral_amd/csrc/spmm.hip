// Fused normalised aggregation for gfx950:
//
//   Y[r,:] = act( dst_scale[r] * sum_{j in row r} w_j * X[indices[j],:] + bias )
//
// Replaces, in one kernel, the source scaling, update_all(copy_u|u_mul_e, sum),
// destination scaling, bias and activation of dgl GraphConv.forward as called at
// spiro/dtdg/models/encoder/implicitTimeEncoder/staticGraphEncoder.py:113,116.
//
// Mapping.  HBM-bound gather: every nonzero pulls one F-wide source row, so the
// kernel is built around whole-row, fully coalesced wave loads and many of them
// in flight.  One 64-lane wave owns a block of consecutive destination rows; the
// nonzeros of such a block are one contiguous CSR range, which the wave walks as
// a flat stream: 64 neighbour ids per coalesced index load, kept in a VGPR and
// broadcast with v_readlane (scalar row base + lane offset addressing), U source
// rows loaded before the first is consumed, row boundaries handled by
// wave-uniform scalar compares.  Short and empty rows therefore cost no idle
// load slots.  Rows heavier than `heavy_thresh` are skipped here and summed
// chunk-wise by whole workgroups (spmm_heavy_kernel + spmm_heavy_reduce_kernel)
// so that a hub row is spread over the chip.  No atomics anywhere: sums are in
// CSR order inside a row (fixed chunk order for hub rows), bit-reproducible.

#include "common.h"

#include <climits>
#include <mutex>

// Partial sums of hub-row chunks for spmi_gcn_spmm_static, the entry point for callers that
// can neither allocate nor hand in a workspace.  Every GPU the library runs on holds its own
// copy of this array, at its own address, until the process ends.  In host code the name is
// only a handle for hipGetSymbolAddress: the device address it returns is what kernels,
// memsets and copies take.
static __device__ __align__(256) float scratch[SPMI_STATIC_PARTIAL_FLOATS];

namespace {

constexpr int kThreads = 256;
constexpr int kWavesPerBlock = kThreads / 64;
constexpr int kUnroll = 16;  // source rows in flight per wave

template <int VEC>
struct VecOf;
template <>
struct VecOf<1> {
  using type = float;
};
template <>
struct VecOf<2> {
  using type = float __attribute__((ext_vector_type(2)));
};
template <>
struct VecOf<4> {
  using type = float __attribute__((ext_vector_type(4)));
};

// Feature rows are stored as float32 or bfloat16; sums are always float32.  A gathered
// row is kept as raw bits (Raw) until it is consumed: converting right at the load would
// put a use inside the load's predicated region and make the compiler wait for every
// load before issuing the next one, i.e. one row in flight instead of kUnroll.
template <int VEC, typename T>
struct RowOf {
  using Raw = typename VecOf<VEC>::type;  // float32: the row itself
};
template <int VEC>
struct RowOf<VEC, __bf16> {
  using Raw = __bf16 __attribute__((ext_vector_type(VEC)));
};
template <>
struct RowOf<1, __bf16> {
  using Raw = __bf16;
};

template <int VEC, typename T>
__device__ __forceinline__ typename RowOf<VEC, T>::Raw load_raw(const T* p) {
  return *reinterpret_cast<const typename RowOf<VEC, T>::Raw*>(p);
}
template <int VEC, typename T>
__device__ __forceinline__ typename VecOf<VEC>::type widen(typename RowOf<VEC, T>::Raw r) {
  using V = typename VecOf<VEC>::type;
  if constexpr (sizeof(T) == 4) {
    return r;
  } else if constexpr (VEC == 1) {
    return static_cast<float>(r);
  } else {
    return __builtin_convertvector(r, V);
  }
}
template <int VEC, typename T>
__device__ __forceinline__ void store_row(T* p, typename VecOf<VEC>::type v, bool last = true) {
  using V = typename VecOf<VEC>::type;
  if constexpr (sizeof(T) == 4) {
    // Finished rows are streamed out: they are not read again by this launch and should not
    // push the gathered rows out of the caches (step 4.007 -> 3.997 ms, 0.893 -> 0.897 of the
    // peak).  Partial sums that a later batch adds to (`last` false) are stored the usual way.
    if (last) __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
    else *reinterpret_cast<V*>(p) = v;
  } else if constexpr (VEC == 1) {
    *p = static_cast<T>(v);
  } else {
    using B = __bf16 __attribute__((ext_vector_type(VEC)));
    *reinterpret_cast<B*>(p) = __builtin_convertvector(v, B);
  }
}

struct SpmmParams {
  const int32_t* indptr;
  const int32_t* indices;
  const void* X;
  const float* src_scale;
  const float* edge_w;
  const float* dst_scale;
  const float* bias;
  void* Y;
  int64_t ldx, ldy;  // in elements
  int32_t n_rows;
  int32_t F;
  int32_t rows_per_wave;
  int32_t act;
  float slope;
  int32_t heavy_thresh;
  int32_t accumulate;  // add the sums to what Y holds (float32 Y only)
  int32_t finalize;    // apply dst_scale, bias and activation; 0 = store the raw sums
};

__device__ __forceinline__ int readlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

template <int VEC>
__device__ __forceinline__ typename VecOf<VEC>::type activate(typename VecOf<VEC>::type y, int act,
                                                              float slope) {
  if (act == SPMI_ACT_NONE) return y;
  const float neg = (act == SPMI_ACT_RELU) ? 0.0f : slope;
  if constexpr (VEC == 1) {
    return y > 0.0f ? y : y * neg;
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) y[i] = y[i] > 0.0f ? y[i] : y[i] * neg;
    return y;
  }
}

// Sum of w_j * X[indices[j]] over the CSR range [s,e) as seen by one wave, with a
// callback at every row boundary.  `row_end_of(row)` returns indptr of row+1 and
// `flush(row, acc)` consumes a finished row; both are wave-uniform.
template <int VEC, typename XT, typename RowEnd, typename Flush>
__device__ __forceinline__ void stream_range(const SpmmParams& p, const XT* __restrict__ xcol,
                                             int lane, int s, int e, int row,
                                             int row_last, RowEnd row_end_of, Flush flush) {
  using V = typename VecOf<VEC>::type;
  V acc = V(0.0f);
  int row_end = row_end_of(row);
  const bool weighted = (p.src_scale != nullptr) || (p.edge_w != nullptr);
  for (int base = s; base < e; base += 64) {
    const int cnt = min(64, e - base);
    int my = 0;
    float w = 1.0f;
    if (lane < cnt) {
      my = p.indices[base + lane];
      if (p.edge_w) w = p.edge_w[base + lane];
      if (p.src_scale) w *= p.src_scale[my];
    }
    for (int k = 0; k < cnt; k += kUnroll) {
      typename RowOf<VEC, XT>::Raw v[kUnroll];
      float wk[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int kk = min(k + u, cnt - 1);
        const int src = readlane_i(my, kk);
        wk[u] = weighted ? readlane_f(w, kk) : 1.0f;
        // unconditional: lanes past the last feature column read column 0 instead (xcol is
        // clamped by the caller), so no predicated region splits the batch of loads
        v[u] = load_raw<VEC, XT>(xcol + static_cast<int64_t>(src) * p.ldx);
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        if (k + u < cnt) {
          const int pos = base + k + u;
          while (pos == row_end) {
            flush(row, acc);
            acc = V(0.0f);
            ++row;
            row_end = row_end_of(row);
          }
          acc += wk[u] * widen<VEC, XT>(v[u]);
        }
      }
    }
  }
  while (row <= row_last) {
    flush(row, acc);
    acc = V(0.0f);
    ++row;
  }
}

template <int VEC, typename XT, typename YT>
__global__ __launch_bounds__(kThreads) void spmm_stream_kernel(SpmmParams p) {
  using V = typename VecOf<VEC>::type;
  const int lane = threadIdx.x & 63;
  const int wave =
      __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)));
  const int R = p.rows_per_wave;
  const int64_t r0_64 = static_cast<int64_t>(wave) * R;
  if (r0_64 >= p.n_rows) return;
  const int r0 = static_cast<int>(r0_64);
  const int nr = min(R, p.n_rows - r0);
  const int f = (static_cast<int>(blockIdx.y) * 64 + lane) * VEC;
  const bool fok = f < p.F;

  // lane i holds indptr[r0+i] (i <= nr) and the destination scale of row r0+i
  const int ip = p.indptr[r0 + min(lane, nr)];
  float dsl = 1.0f;
  if (p.dst_scale && lane < nr) dsl = p.dst_scale[r0 + lane];
  V bias = V(0.0f);
  if (p.bias && fok) bias = *reinterpret_cast<const V*>(p.bias + f);
  const int ip_next = __shfl_down(ip, 1);
  const bool heavy_lane = (lane < nr) && (ip_next - ip > p.heavy_thresh);
  const bool any_heavy = __ballot(heavy_lane) != 0ull;

  const XT* xcol = static_cast<const XT*>(p.X) + (fok ? f : 0);
  YT* ycol = static_cast<YT*>(p.Y) + f;
  auto row_end_of = [&](int row) { return readlane_i(ip, row + 1); };
  auto flush = [&](int row, V acc) {
    YT* yp = ycol + static_cast<int64_t>(r0 + row) * p.ldy;
    if constexpr (sizeof(YT) == 4) {
      if (p.accumulate && fok) acc += *reinterpret_cast<const V*>(yp);
    }
    V y = acc;
    if (p.finalize) {
      const float ds = readlane_f(dsl, row);
      y = activate<VEC>(acc * ds + bias, p.act, p.slope);
    }
    if (fok) store_row<VEC, YT>(yp, y, p.finalize);
  };

  if (!any_heavy) {
    stream_range<VEC, XT>(p, xcol, lane, readlane_i(ip, 0), readlane_i(ip, nr), 0, nr - 1,
                          row_end_of, flush);
  } else {
    for (int row = 0; row < nr; ++row) {
      const int rs = readlane_i(ip, row), re = readlane_i(ip, row + 1);
      if (re - rs > p.heavy_thresh) continue;  // summed by the heavy-row kernels
      stream_range<VEC, XT>(p, xcol, lane, rs, re, row, row, row_end_of, flush);
    }
  }
}

// One workgroup sums one chunk of a heavy row; its four waves take a quarter of
// the chunk each and are combined through LDS in wave order.
template <int VEC, typename XT>
__global__ __launch_bounds__(kThreads) void spmm_heavy_kernel(SpmmParams p,
                                                              const int32_t* __restrict__ heavy_rows,
                                                              const int32_t* __restrict__ chunk_ptr,
                                                              int n_heavy, int chunk,
                                                              float* __restrict__ partial) {
  using V = typename VecOf<VEC>::type;
  __shared__ V s_part[kWavesPerBlock][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int c = blockIdx.x;
  int lo = 0, hi = n_heavy;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_ptr[mid] <= c) lo = mid; else hi = mid;
  }
  const int row = heavy_rows[lo];
  const int k = c - chunk_ptr[lo];
  const int row_end = p.indptr[row + 1];
  const int cs = p.indptr[row] + k * chunk;
  const int ce = min(row_end, cs + chunk);
  const int per = (chunk + kWavesPerBlock - 1) / kWavesPerBlock;
  const int s = min(ce, cs + wave * per);
  const int e = min(ce, s + per);
  const int f = (static_cast<int>(blockIdx.y) * 64 + lane) * VEC;
  const bool fok = f < p.F;

  V total = V(0.0f);
  auto row_end_of = [&](int) { return INT_MAX; };
  auto flush = [&](int, V acc) { total = acc; };
  stream_range<VEC, XT>(p, static_cast<const XT*>(p.X) + (fok ? f : 0), lane, s, e, 0, 0, row_end_of,
                        flush);
  s_part[wave][lane] = total;
  __syncthreads();
  if (wave == 0 && fok) {
    V sum = s_part[0][lane];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; ++w) sum += s_part[w][lane];
    *reinterpret_cast<V*>(partial + static_cast<int64_t>(c) * p.F + f) = sum;
  }
}

// One workgroup finishes one heavy row: partial sums are added in a fixed order
// (wave w takes chunks w, w+4, ...; waves combined in wave order), then the
// same epilogue as the stream kernel.
template <int VEC, typename YT>
__global__ __launch_bounds__(kThreads) void spmm_heavy_reduce_kernel(
    SpmmParams p, const int32_t* __restrict__ heavy_rows, const int32_t* __restrict__ chunk_ptr,
    const float* __restrict__ partial) {
  using V = typename VecOf<VEC>::type;
  __shared__ V s_part[kWavesPerBlock][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int h = blockIdx.x;
  const int row = heavy_rows[h];
  const int c0 = chunk_ptr[h], c1 = chunk_ptr[h + 1];
  const int f = (static_cast<int>(blockIdx.y) * 64 + lane) * VEC;
  const bool fok = f < p.F;
  V acc = V(0.0f);
  if (fok) {
    const float* base = partial + f;
    int c = c0 + wave;
    for (; c + 3 * kWavesPerBlock < c1; c += 4 * kWavesPerBlock) {
      V a0 = *reinterpret_cast<const V*>(base + static_cast<int64_t>(c) * p.F);
      V a1 = *reinterpret_cast<const V*>(base + static_cast<int64_t>(c + kWavesPerBlock) * p.F);
      V a2 = *reinterpret_cast<const V*>(base + static_cast<int64_t>(c + 2 * kWavesPerBlock) * p.F);
      V a3 = *reinterpret_cast<const V*>(base + static_cast<int64_t>(c + 3 * kWavesPerBlock) * p.F);
      acc += a0;
      acc += a1;
      acc += a2;
      acc += a3;
    }
    for (; c < c1; c += kWavesPerBlock)
      acc += *reinterpret_cast<const V*>(base + static_cast<int64_t>(c) * p.F);
  }
  s_part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && fok) {
    V sum = s_part[0][lane];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; ++w) sum += s_part[w][lane];
    YT* yp = static_cast<YT*>(p.Y) + static_cast<int64_t>(row) * p.ldy + f;
    if constexpr (sizeof(YT) == 4) {
      if (p.accumulate) sum += *reinterpret_cast<const V*>(yp);
    }
    V y = sum;
    if (p.finalize) {
      const float ds = p.dst_scale ? p.dst_scale[row] : 1.0f;
      V bias = V(0.0f);
      if (p.bias) bias = *reinterpret_cast<const V*>(p.bias + f);
      y = activate<VEC>(sum * ds + bias, p.act, p.slope);
    }
    store_row<VEC, YT>(yp, y, p.finalize);
  }
}

// gP = gY * act'(Y); gb[f] += sum over rows of gP[:,f].  A workgroup takes a slab
// of rows and 64 columns; waves stride over the rows, lanes hold one column.
constexpr int kBwdRowsPerBlock = 512;
__global__ __launch_bounds__(kThreads) void act_bias_bwd_kernel(const float* __restrict__ gY,
                                                                const float* __restrict__ Y,
                                                                int64_t M, int F, int64_t ld,
                                                                int act, float slope,
                                                                float* __restrict__ gP,
                                                                float* __restrict__ gb,
                                                                float* __restrict__ partials,
                                                                const float* __restrict__ row_scale) {
  __shared__ float s_col[kWavesPerBlock][64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int f = blockIdx.y * 64 + lane;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * kBwdRowsPerBlock;
  const int64_t row1 = row0 + kBwdRowsPerBlock < M ? row0 + kBwdRowsPerBlock : M;
  const float neg = (act == SPMI_ACT_RELU) ? 0.0f : slope;
  float col = 0.0f;
  if (f < F) {
    for (int64_t r = row0 + wave; r < row1; r += kWavesPerBlock) {
      float g = gY[r * ld + f];
      if (act != SPMI_ACT_NONE) g = (Y[r * ld + f] > 0.0f) ? g : g * neg;
      gP[r * F + f] = row_scale ? g * row_scale[r] : g;
      col += g;
    }
  }
  if (gb || partials) {
    s_col[wave][lane] = col;
    __syncthreads();
    if (wave == 0 && f < F) {
      float sum = s_col[0][lane] + s_col[1][lane] + s_col[2][lane] + s_col[3][lane];
      if (partials) partials[static_cast<size_t>(blockIdx.x) * F + f] = sum;
      else atomicAdd(&gb[f], sum);
    }
  }
}

// The same with 16-byte accesses: a thread holds four columns, sixteen threads cover the 64
// columns of the block and a wave four rows per access, four accesses in flight.
__global__ __launch_bounds__(kThreads) void act_bias_bwd_v4_kernel(const float* __restrict__ gY,
                                                                   const float* __restrict__ Y,
                                                                   int64_t M, int F, int64_t ld,
                                                                   int act, float slope,
                                                                   float* __restrict__ gP,
                                                                   float* __restrict__ gb,
                                                                   float* __restrict__ partials,
                                                                   const float* __restrict__ row_scale) {
  __shared__ float4 s_col[16][16];
  const int cg = threadIdx.x & 15;   // column group: columns f .. f + 3
  const int rl = threadIdx.x >> 4;   // 0 .. 15: rows rl, rl + 16, ...
  const int f = blockIdx.y * 64 + 4 * cg;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * kBwdRowsPerBlock;
  const int64_t row1 = row0 + kBwdRowsPerBlock < M ? row0 + kBwdRowsPerBlock : M;
  const float neg = (act == SPMI_ACT_RELU) ? 0.0f : slope;
  float4 col = make_float4(0.f, 0.f, 0.f, 0.f);
  if (f < F) {
    constexpr int U = 4;
    for (int64_t r = row0 + rl; r < row1; r += 16 * U) {
      float4 g[U], y[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t rr = r + 16 * u < row1 ? r + 16 * u : r;  // past the slab: re-read, not stored
        g[u] = *reinterpret_cast<const float4*>(gY + rr * ld + f);
        if (act != SPMI_ACT_NONE) y[u] = *reinterpret_cast<const float4*>(Y + rr * ld + f);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (r + 16 * u >= row1) break;
        if (act != SPMI_ACT_NONE) {
          g[u].x = y[u].x > 0.0f ? g[u].x : g[u].x * neg;
          g[u].y = y[u].y > 0.0f ? g[u].y : g[u].y * neg;
          g[u].z = y[u].z > 0.0f ? g[u].z : g[u].z * neg;
          g[u].w = y[u].w > 0.0f ? g[u].w : g[u].w * neg;
        }
        col.x += g[u].x; col.y += g[u].y; col.z += g[u].z; col.w += g[u].w;
        if (row_scale) {
          const float rs = row_scale[r + 16 * u];
          g[u].x *= rs; g[u].y *= rs; g[u].z *= rs; g[u].w *= rs;
        }
        *reinterpret_cast<float4*>(gP + (r + 16 * u) * F + f) = g[u];
      }
    }
  }
  if (gb || partials) {
    s_col[rl][cg] = col;
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.y * 64 + static_cast<int>(threadIdx.x) < F) {
      const float* flat = reinterpret_cast<const float*>(&s_col[0][0]);  // [16 rows][64 columns]
      float sum = 0.0f;
#pragma unroll
      for (int q = 0; q < 16; ++q) sum += flat[q * 64 + threadIdx.x];
      const int fc = blockIdx.y * 64 + static_cast<int>(threadIdx.x);
      if (partials) partials[static_cast<size_t>(blockIdx.x) * F + fc] = sum;
      else atomicAdd(&gb[fc], sum);
    }
  }
}

// gb[f] = sum over the row slabs of their column sums, in slab order: each of the sixteen waves
// owns a sixteenth of the slabs and walks it front to back (lanes = columns, eight loads in
// flight), the wave sums are added in wave order.  No atomics: the same bits on every launch.
constexpr int kReduceWaves = 16;
__global__ __launch_bounds__(64 * kReduceWaves) void bias_grad_reduce_kernel(
    const float* __restrict__ partials, int n_slabs, int F, float* __restrict__ gb) {
  __shared__ float s_col[kReduceWaves][64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + lane;
  const int per = (n_slabs + kReduceWaves - 1) / kReduceWaves;
  const int s0 = min(wave * per, n_slabs), s1 = min(s0 + per, n_slabs);
  float sum = 0.0f;
  if (f < F) {
    int q = s0;
    for (; q + 8 <= s1; q += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partials[static_cast<size_t>(q + u) * F + f];
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; q < s1; ++q) sum += partials[static_cast<size_t>(q) * F + f];
  }
  s_col[wave][lane] = sum;
  __syncthreads();
  if (wave == 0 && f < F) {
    float total = s_col[0][lane];
#pragma unroll
    for (int w = 1; w < kReduceWaves; ++w) total += s_col[w][lane];
    gb[f] = total;
  }
}

bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

template <int VEC, typename XT, typename YT>
int launch_spmm(const SpmmParams& p, const int32_t* heavy_rows, const int32_t* chunk_ptr,
                int n_heavy, int n_chunks, int chunk, float* partial, hipStream_t st) {
  const unsigned fblocks = static_cast<unsigned>((p.F + 64 * VEC - 1) / (64 * VEC));
  const int64_t n_waves = (static_cast<int64_t>(p.n_rows) + p.rows_per_wave - 1) / p.rows_per_wave;
  const unsigned grid_x = static_cast<unsigned>((n_waves + kWavesPerBlock - 1) / kWavesPerBlock);
  spmm_stream_kernel<VEC, XT, YT><<<dim3(grid_x, fblocks), kThreads, 0, st>>>(p);
  SPMI_CHECK_LAUNCH("spmm_stream");
  if (n_heavy > 0) {
    spmm_heavy_kernel<VEC, XT><<<dim3(n_chunks, fblocks), kThreads, 0, st>>>(
        p, heavy_rows, chunk_ptr, n_heavy, chunk, partial);
    SPMI_CHECK_LAUNCH("spmm_heavy");
    spmm_heavy_reduce_kernel<VEC, YT><<<dim3(n_heavy, fblocks), kThreads, 0, st>>>(
        p, heavy_rows, chunk_ptr, partial);
    SPMI_CHECK_LAUNCH("spmm_heavy_reduce");
  }
  return SPMI_OK;
}


template <typename XT, typename YT>
int dispatch_spmm(SpmmParams& p, const int32_t* heavy_rows, const int32_t* heavy_chunk_ptr,
                  int32_t n_heavy, int32_t n_chunks, int32_t chunk, float* partial_ws,
                  hipStream_t st) {
  // widest per-lane access whose alignment holds for both the input and the output rows
  auto ok = [&](int vec) {
    const size_t ax = vec * sizeof(XT), ay = vec * sizeof(YT);
    return p.F % vec == 0 && p.ldx % vec == 0 && p.ldy % vec == 0 && aligned(p.X, ax) &&
           aligned(p.Y, ay) && (!p.bias || aligned(p.bias, vec * 4)) &&
           (!partial_ws || aligned(partial_ws, vec * 4));
  };
  if (p.F >= 256 && ok(4))
    return launch_spmm<4, XT, YT>(p, heavy_rows, heavy_chunk_ptr, n_heavy, n_chunks, chunk,
                                  partial_ws, st);
  if (p.F >= 128 && ok(2))
    return launch_spmm<2, XT, YT>(p, heavy_rows, heavy_chunk_ptr, n_heavy, n_chunks, chunk,
                                  partial_ws, st);
  return launch_spmm<1, XT, YT>(p, heavy_rows, heavy_chunk_ptr, n_heavy, n_chunks, chunk,
                                partial_ws, st);
}

}  // namespace

extern "C" {

int spmi_gcn_spmm_part(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t n_src,
                       int64_t E, const void* X, int x_dtype, int64_t ldx, int64_t F,
                       const float* src_scale, const float* edge_w, const float* dst_scale,
                       const float* bias, int act, float slope, void* Y, int y_dtype, int64_t ldy,
                       const int32_t* heavy_rows, const int32_t* heavy_chunk_ptr, int32_t n_heavy,
                       int32_t n_chunks, int32_t heavy_thresh, int32_t chunk, float* partial_ws,
                       int accumulate, int finalize, void* stream) {
  SPMI_REQUIRE((!accumulate && finalize) || y_dtype == SPMI_DTYPE_F32, SPMI_ERR_INVALID_ARG,
               "spmi_gcn_spmm_part: partial sums (accumulate / no finalize) need a float32 Y");
  SPMI_REQUIRE(n_rows >= 0 && n_src >= 0 && E >= 0 && F >= 1, SPMI_ERR_INVALID_ARG,
               "spmi_gcn_spmm: bad sizes n_rows=%lld n_src=%lld E=%lld F=%lld", (long long)n_rows,
               (long long)n_src, (long long)E, (long long)F);
  SPMI_REQUIRE(n_rows < (int64_t{1} << 31) && n_src < (int64_t{1} << 31) &&
                   E < (int64_t{1} << 31) && F < (int64_t{1} << 24),
               SPMI_ERR_UNSUPPORTED, "spmi_gcn_spmm: size beyond the int32 on-device limits");
  SPMI_REQUIRE(act >= SPMI_ACT_NONE && act <= SPMI_ACT_LEAKY_RELU, SPMI_ERR_INVALID_ARG,
               "spmi_gcn_spmm: bad activation code %d", act);
  SPMI_REQUIRE((x_dtype == SPMI_DTYPE_F32 || x_dtype == SPMI_DTYPE_BF16) &&
                   (y_dtype == SPMI_DTYPE_F32 || y_dtype == SPMI_DTYPE_BF16),
               SPMI_ERR_INVALID_ARG, "spmi_gcn_spmm: bad dtype code (%d, %d)", x_dtype, y_dtype);
  if (n_rows == 0) return SPMI_OK;
  SPMI_REQUIRE(indptr && Y, SPMI_ERR_INVALID_ARG, "spmi_gcn_spmm: null indptr/Y");
  SPMI_REQUIRE(E == 0 || (indices && X), SPMI_ERR_INVALID_ARG, "spmi_gcn_spmm: null indices/X");
  SPMI_REQUIRE(ldx >= F && ldy >= F, SPMI_ERR_INVALID_ARG,
               "spmi_gcn_spmm: leading dimensions (%lld,%lld) smaller than F=%lld", (long long)ldx,
               (long long)ldy, (long long)F);
  SPMI_REQUIRE(n_heavy >= 0 && n_chunks >= n_heavy, SPMI_ERR_INVALID_ARG,
               "spmi_gcn_spmm: bad heavy plan (%d rows, %d chunks)", n_heavy, n_chunks);
  if (n_heavy > 0) {
    SPMI_REQUIRE(heavy_rows && heavy_chunk_ptr && partial_ws && heavy_thresh >= 1 && chunk >= 1,
                 SPMI_ERR_INVALID_ARG, "spmi_gcn_spmm: heavy plan given without its buffers");
  }

  SpmmParams p{};
  p.indptr = indptr;
  p.indices = indices;
  p.X = X;
  p.src_scale = src_scale;
  p.edge_w = edge_w;
  p.dst_scale = dst_scale;
  p.bias = bias;
  p.Y = Y;
  p.ldx = ldx;
  p.ldy = ldy;
  p.n_rows = static_cast<int32_t>(n_rows);
  p.F = static_cast<int32_t>(F);
  p.act = act;
  p.slope = slope;
  p.heavy_thresh = (n_heavy > 0) ? heavy_thresh : INT_MAX;
  p.accumulate = accumulate != 0;
  p.finalize = finalize != 0;
  // about 96 nonzeros per wave keeps the index load and the row stream busy
  const int64_t avg_deg = (E + n_rows - 1) / n_rows;
  int rpw = 32;
  while (rpw > 2 && rpw * (avg_deg > 0 ? avg_deg : 1) > 128) rpw >>= 1;
  p.rows_per_wave = rpw;

  hipStream_t st = spmi::as_stream(stream);
  if (x_dtype == SPMI_DTYPE_F32 && y_dtype == SPMI_DTYPE_F32)
    return dispatch_spmm<float, float>(p, heavy_rows, heavy_chunk_ptr, n_heavy, n_chunks, chunk,
                                       partial_ws, st);
  if (x_dtype == SPMI_DTYPE_BF16 && y_dtype == SPMI_DTYPE_BF16)
    return dispatch_spmm<__bf16, __bf16>(p, heavy_rows, heavy_chunk_ptr, n_heavy, n_chunks, chunk,
                                         partial_ws, st);
  if (x_dtype == SPMI_DTYPE_BF16)
    return dispatch_spmm<__bf16, float>(p, heavy_rows, heavy_chunk_ptr, n_heavy, n_chunks, chunk,
                                        partial_ws, st);
  return dispatch_spmm<float, __bf16>(p, heavy_rows, heavy_chunk_ptr, n_heavy, n_chunks, chunk,
                                      partial_ws, st);
}

int spmi_gcn_spmm_ex(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t n_src,
                     int64_t E, const void* X, int x_dtype, int64_t ldx, int64_t F,
                     const float* src_scale, const float* edge_w, const float* dst_scale,
                     const float* bias, int act, float slope, void* Y, int y_dtype, int64_t ldy,
                     const int32_t* heavy_rows, const int32_t* heavy_chunk_ptr, int32_t n_heavy,
                     int32_t n_chunks, int32_t heavy_thresh, int32_t chunk, float* partial_ws,
                     void* stream) {
  return spmi_gcn_spmm_part(indptr, indices, n_rows, n_src, E, X, x_dtype, ldx, F, src_scale, edge_w,
                            dst_scale, bias, act, slope, Y, y_dtype, ldy, heavy_rows,
                            heavy_chunk_ptr, n_heavy, n_chunks, heavy_thresh, chunk, partial_ws, 0, 1,
                            stream);
}

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
float* g_scratch_ptr[kMaxDevices] = {nullptr};  // device address of `scratch`, per GPU
}  // namespace

int spmi_gcn_spmm_static(const int32_t* indptr, const int32_t* indices, int64_t n_rows,
                         int64_t n_src, int64_t E, const void* X, int x_dtype, int64_t ldx,
                         int64_t F, const float* src_scale, const float* edge_w,
                         const float* dst_scale, const float* bias, int act, float slope, void* Y,
                         int y_dtype, int64_t ldy, const int32_t* heavy_rows,
                         const int32_t* heavy_chunk_ptr, int32_t n_heavy, int32_t n_chunks,
                         int32_t heavy_thresh, int32_t chunk, void* stream) {
  SPMI_REQUIRE(n_heavy >= 0 && n_chunks >= n_heavy && F >= 1, SPMI_ERR_INVALID_ARG,
               "spmi_gcn_spmm_static: bad heavy plan (%d rows, %d chunks) or F=%lld", n_heavy,
               n_chunks, (long long)F);
  SPMI_REQUIRE(static_cast<int64_t>(n_chunks) * F <= int64_t{SPMI_STATIC_PARTIAL_FLOATS},
               SPMI_ERR_WORKSPACE,
               "spmi_gcn_spmm_static: %d hub-row chunks x %lld columns exceed the built-in scratch "
               "of %lld floats; use spmi_gcn_spmm_ex with a workspace",
               n_chunks, (long long)F, (long long)SPMI_STATIC_PARTIAL_FLOATS);
  float* partial = nullptr;
  if (n_heavy > 0) {
    int dev = -1;
    SPMI_CHECK_HIP(hipGetDevice(&dev));
    SPMI_REQUIRE(dev >= 0 && dev < kMaxDevices, SPMI_ERR_UNSUPPORTED,
                 "spmi_gcn_spmm_static: device ordinal %d beyond %d", dev, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
      void* addr = nullptr;
      SPMI_CHECK_HIP(hipGetSymbolAddress(&addr, HIP_SYMBOL(scratch)));
      SPMI_REQUIRE(addr != nullptr, SPMI_ERR_HIP, "spmi_gcn_spmm_static: scratch has no device address");
      g_scratch_ptr[dev] = static_cast<float*>(addr);
    }
    partial = g_scratch_ptr[dev];
  }
  // Every chunk's partial row is written by spmm_heavy_kernel before spmm_heavy_reduce_kernel
  // reads it, so nothing of the previous call's contents is ever read: no clearing needed.
  return spmi_gcn_spmm_ex(indptr, indices, n_rows, n_src, E, X, x_dtype, ldx, F, src_scale, edge_w,
                          dst_scale, bias, act, slope, Y, y_dtype, ldy, heavy_rows, heavy_chunk_ptr,
                          n_heavy, n_chunks, heavy_thresh, chunk, partial, stream);
}

int spmi_gcn_spmm(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t n_src,
                  int64_t E, const float* X, int64_t ldx, int64_t F, const float* src_scale,
                  const float* edge_w, const float* dst_scale, const float* bias, int act,
                  float slope, float* Y, int64_t ldy, const int32_t* heavy_rows,
                  const int32_t* heavy_chunk_ptr, int32_t n_heavy, int32_t n_chunks,
                  int32_t heavy_thresh, int32_t chunk, float* partial_ws, void* stream) {
  return spmi_gcn_spmm_ex(indptr, indices, n_rows, n_src, E, X, SPMI_DTYPE_F32, ldx, F, src_scale,
                          edge_w, dst_scale, bias, act, slope, Y, SPMI_DTYPE_F32, ldy, heavy_rows,
                          heavy_chunk_ptr, n_heavy, n_chunks, heavy_thresh, chunk, partial_ws,
                          stream);
}

static int act_bias_bwd_launch(const float* gY, const float* Y, int64_t M, int64_t F, int64_t ld, int act,
                               float slope, float* gP, float* gb, float* partials, const float* row_scale,
                               void* stream) {
  const unsigned gx = static_cast<unsigned>((M + kBwdRowsPerBlock - 1) / kBwdRowsPerBlock);
  const unsigned gy = static_cast<unsigned>((F + 63) / 64);
  const bool v4 = F % 4 == 0 && ld % 4 == 0 && aligned(gY, 16) && aligned(gP, 16) && (!Y || aligned(Y, 16));
  if (v4)
    act_bias_bwd_v4_kernel<<<dim3(gx, gy), kThreads, 0, spmi::as_stream(stream)>>>(
        gY, Y, M, static_cast<int>(F), ld, act, slope, gP, gb, partials, row_scale);
  else
    act_bias_bwd_kernel<<<dim3(gx, gy), kThreads, 0, spmi::as_stream(stream)>>>(
        gY, Y, M, static_cast<int>(F), ld, act, slope, gP, gb, partials, row_scale);
  SPMI_CHECK_LAUNCH("act_bias_bwd");
  if (partials) {
    bias_grad_reduce_kernel<<<gy, 64 * kReduceWaves, 0, spmi::as_stream(stream)>>>(partials, static_cast<int>(gx),
                                                                          static_cast<int>(F), gb);
    SPMI_CHECK_LAUNCH("bias_grad_reduce");
  }
  return SPMI_OK;
}

int spmi_act_bias_bwd(const float* gY, const float* Y, int64_t M, int64_t F, int64_t ld, int act,
                      float slope, float* gP, float* gb, void* stream) {
  SPMI_REQUIRE(M >= 0 && F >= 1 && ld >= F && F < (int64_t{1} << 24), SPMI_ERR_INVALID_ARG,
               "spmi_act_bias_bwd: bad sizes");
  SPMI_REQUIRE(act >= SPMI_ACT_NONE && act <= SPMI_ACT_LEAKY_RELU, SPMI_ERR_INVALID_ARG,
               "spmi_act_bias_bwd: bad activation code %d", act);
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(gY && gP && (act == SPMI_ACT_NONE || Y), SPMI_ERR_INVALID_ARG,
               "spmi_act_bias_bwd: null pointer");
  return act_bias_bwd_launch(gY, Y, M, F, ld, act, slope, gP, gb, nullptr, nullptr, stream);
}

size_t spmi_act_bias_bwd_workspace_bytes(int64_t M, int64_t F) {
  if (M < 0 || F < 1) return 0;
  return sizeof(float) * static_cast<size_t>((M + kBwdRowsPerBlock - 1) / kBwdRowsPerBlock) * static_cast<size_t>(F);
}

int spmi_act_bias_bwd_ordered(const float* gY, const float* Y, int64_t M, int64_t F, int64_t ld, int act,
                              float slope, const float* row_scale, float* gP, float* gb, void* ws,
                              size_t ws_bytes, void* stream) {
  SPMI_REQUIRE(M >= 0 && F >= 1 && ld >= F && F < (int64_t{1} << 24), SPMI_ERR_INVALID_ARG,
               "spmi_act_bias_bwd_ordered: bad sizes");
  SPMI_REQUIRE(act >= SPMI_ACT_NONE && act <= SPMI_ACT_LEAKY_RELU, SPMI_ERR_INVALID_ARG,
               "spmi_act_bias_bwd_ordered: bad activation code %d", act);
  if (M == 0) {
    if (gb) SPMI_CHECK_HIP(hipMemsetAsync(gb, 0, sizeof(float) * (size_t)F, spmi::as_stream(stream)));
    return SPMI_OK;
  }
  SPMI_REQUIRE(gY && gP && (!gb || ws) && (act == SPMI_ACT_NONE || Y), SPMI_ERR_INVALID_ARG,
               "spmi_act_bias_bwd_ordered: null pointer");
  SPMI_REQUIRE(!gb || ws_bytes >= spmi_act_bias_bwd_workspace_bytes(M, F), SPMI_ERR_WORKSPACE,
               "spmi_act_bias_bwd_ordered: workspace %zu < required %zu", ws_bytes,
               spmi_act_bias_bwd_workspace_bytes(M, F));
  return act_bias_bwd_launch(gY, Y, M, F, ld, act, slope, gP, gb, gb ? static_cast<float*>(ws) : nullptr,
                             row_scale, stream);
}

}  // extern "C"
