// Weight-gradient GEMM for gfx950: C[M,N] = A[K,M]^T B[K,N] with K = number of nodes
// (millions) and M, N = layer widths (tens to hundreds).
//
// What it replaces in the reference: the autograd of every dense layer on the path,
// dW = X^T dY, which torch evaluates with a library GEMM: GraphConv's weight
// (dgl GraphConv.forward, call sites staticGraphEncoder.py:113,116), the gate weights of
// nn.GRU / nn.LSTM / nn.RNN (implicitTimeModels.py:185-187,256-258,284-286) and the MLP's
// Linear layers (simpleDecoder.py:60-72) in the training loop of
// examples/torch/citation_prediction.py:96-100.  The library kernels are tiled for square
// problems and run this shape (tiny output, huge reduction) at a few percent of peak.
//
// Design: the reduction dimension is split over the grid.  A wave owns a 32 x (32*NT)
// block of C and a chunk of K; for the fp32 matrix instruction v_mfma_f32_32x32x2_f32 the
// operand layouts are "lane l holds A^T[m = l%32][k = l/32]" and "B[k = l/32][n = l%32]",
// which for row-major A[K,M] and B[K,N] are exactly two coalesced 128-byte row segments:
// operands stream from global memory straight into the matrix cores, no LDS, no
// transposition.  Eight k-pairs of loads are kept in flight.  Per-chunk partial blocks go
// to a workspace and a second kernel sums them in chunk order: fixed order, bit-reproducible,
// no atomics.  Exact fp32 products and fp32 accumulation (not a reduced-precision format).
// Bound: MFMA for wide outputs (2*K*M*N flop), HBM (K*(M+N)*4 bytes) for narrow ones.
// Further down: the same products from three exact bf16 pieces per value (six piece products
// per term, 2-3x faster) and, for outputs made of whole 128 x 128 blocks, from three fp16 planes
// per value under running power-of-two scales (three products per term).

#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kUnroll = 8;  // k-pairs per loop trip (16 rows of A and B)

template <int NT>
__global__ __launch_bounds__(kThreads) void gemm_tn_partial_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, int64_t K,
    int M, int N, int64_t k_chunk, float* __restrict__ partial, int m_strips, int n_tiles,
    float* __restrict__ sum_a_partial, float* __restrict__ sum_b_partial) {
  const int lane = threadIdx.x & 63;
  const int st = static_cast<int>(blockIdx.x) * kWaves + (threadIdx.x >> 6);
  if (st >= m_strips * n_tiles) return;
  const int m0 = (st % m_strips) * 32;
  const int n0 = (st / m_strips) * 32 * NT;
  const int64_t k_begin = static_cast<int64_t>(blockIdx.y) * k_chunk;
  const int64_t k_end = (k_begin + k_chunk < K) ? k_begin + k_chunk : K;
  const int lr = lane & 31, lh = lane >> 5;

  // Out-of-range columns read a valid column and are zeroed after the load, so that every
  // load is unconditional (a predicated load would serialise the loads in flight).
  // Addresses are "wave-uniform row base + 32-bit lane offset": one offset register per
  // operand column, the row bases live in scalar registers.
  const bool a_ok = m0 + lr < M;
  const uint32_t a_off = static_cast<uint32_t>(lh * lda + (a_ok ? m0 + lr : 0));
  bool b_ok[NT];
  uint32_t b_off[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    b_ok[t] = n0 + 32 * t + lr < N;
    b_off[t] = static_cast<uint32_t>(lh * ldb + (b_ok[t] ? n0 + 32 * t + lr : 0));
  }

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;

  // Column sums of the operands ride along: the waves of the first column tile add up what
  // they load of A, the waves of the first row strip what they load of B (one add per load).
  const bool want_sa = sum_a_partial != nullptr && n0 == 0;
  const bool want_sb = sum_b_partial != nullptr && m0 == 0;
  float sa = 0.0f, sb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) sb[t] = 0.0f;

  int64_t k = k_begin;
  for (; k + 2 * kUnroll <= k_end; k += 2 * kUnroll) {
    float a[kUnroll], b[kUnroll][NT];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const float* a_row = A + (k + 2 * u) * lda;
      const float* b_row = B + (k + 2 * u) * ldb;
      a[u] = a_row[a_off];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[u][t] = b_row[b_off[t]];
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const float av = a_ok ? a[u] : 0.0f;
      sa += av;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float bv = b_ok[t] ? b[u][t] : 0.0f;
        sb[t] += bv;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
      }
    }
  }
  for (; k < k_end; k += 2) {  // tail of the chunk: rows past k_end contribute zero
    const bool r_ok = k + lh < k_end;
    const int64_t kk = r_ok ? k : k - 1;  // k >= k_begin + 1 when lh == 1 overruns; stays inside
    const float av = (a_ok && r_ok) ? (A + kk * lda)[a_off] : 0.0f;
    sa += av;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float bv = (b_ok[t] && r_ok) ? (B + kk * ldb)[b_off[t]] : 0.0f;
      sb[t] += bv;
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
    }
  }
  if (want_sa) {  // even and odd rows of the chunk sit in the two lane halves
    sa += __shfl_xor(sa, 32);
    if (lh == 0 && a_ok) sum_a_partial[static_cast<size_t>(blockIdx.y) * M + m0 + lr] = sa;
  }
  if (want_sb) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float v = sb[t] + __shfl_xor(sb[t], 32);
      if (lh == 0 && b_ok[t]) sum_b_partial[static_cast<size_t>(blockIdx.y) * N + n0 + 32 * t + lr] = v;
    }
  }

  float* out = partial + static_cast<size_t>(blockIdx.y) * M * N;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = n0 + 32 * t + lr;
    if (n >= N) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      if (m < M) out[static_cast<size_t>(m) * N + n] = acc[t][e];
    }
  }
}

// The same block product on the bf16 matrix cores at fp32 accuracy: every fp32 operand value is
// split exactly into three bf16 pieces (v = hi + mid + lo) and a product is summed from the six
// piece products down to 2^-24 of its size (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid), with
// fp32 accumulation.  Six v_mfma_f32_32x32x16_bf16 (32 cycles, K = 16) stand where eight
// v_mfma_f32_32x32x2_f32 (64 cycles, K = 2) stood: 2.7x less matrix time, which moves these
// shapes from the fp32 matrix peak to the HBM stream of their operands.  A wave owns a
// (32 MT) x (32 NT) block of C, so a split piece of A feeds NT products and one of B feeds MT.
// Operand layout of the K = 16 instruction: lane l holds eight consecutive k of column l % 32,
// k = 8 (l / 32) + i: eight loads per operand column, each two coalesced 128-byte row segments,
// as before.  The loads of the next sixteen rows are issued before the current ones are split.
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3x8(const float (&v)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = static_cast<__bf16>(v[i]);
    const float r1 = v[i] - static_cast<float>(h);
    const __bf16 m = static_cast<__bf16>(r1);
    hi[i] = h;
    mid[i] = m;
    lo[i] = static_cast<__bf16>(r1 - static_cast<float>(m));
  }
}

template <int MT, int NT>
__global__ __launch_bounds__(kThreads, 2) void gemm_tn_partial_x3_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, int64_t K,
    int M, int N, int64_t k_chunk, float* __restrict__ partial, int m_groups, int n_tiles,
    float* __restrict__ sum_a_partial, float* __restrict__ sum_b_partial) {
  const int lane = threadIdx.x & 63;
  const int st = static_cast<int>(blockIdx.x) * kWaves + (threadIdx.x >> 6);
  if (st >= m_groups * n_tiles) return;
  const int m0 = (st % m_groups) * 32 * MT;
  const int n0 = (st / m_groups) * 32 * NT;
  const int64_t k_begin = static_cast<int64_t>(blockIdx.y) * k_chunk;
  const int64_t k_end = (k_begin + k_chunk < K) ? k_begin + k_chunk : K;
  const int lr = lane & 31, lh = lane >> 5;

  bool a_ok[MT], b_ok[NT];
  uint32_t a_off[MT], b_off[NT];
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    a_ok[s] = m0 + 32 * s + lr < M;
    a_off[s] = static_cast<uint32_t>(lh * 8 * lda + (a_ok[s] ? m0 + 32 * s + lr : 0));
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    b_ok[t] = n0 + 32 * t + lr < N;
    b_off[t] = static_cast<uint32_t>(lh * 8 * ldb + (b_ok[t] ? n0 + 32 * t + lr : 0));
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int s = 0; s < MT; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[s][t][e] = 0.0f;
  const bool want_sa = sum_a_partial != nullptr && n0 == 0;
  const bool want_sb = sum_b_partial != nullptr && m0 == 0;
  float sa[MT], sb[NT];
#pragma unroll
  for (int s = 0; s < MT; ++s) sa[s] = 0.0f;
#pragma unroll
  for (int t = 0; t < NT; ++t) sb[t] = 0.0f;

  // Rows past the chunk's end are read from its last row (valid memory) and zeroed.
  float a[MT][8], b[NT][8];
  // FULL: all sixteen rows of the step lie inside the chunk (every step but the last): row
  // bases are wave-uniform scalars and nothing is masked by row.
  auto load = [&](auto full_tag, int64_t k, float (&av)[MT][8], float (&bv)[NT][8]) {
    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int64_t row = k + u;  // this lane reads row + 8 lh (the lane half is in the offsets)
      if constexpr (!FULL) {
        const int64_t last = k_end - 1 - 8 * lh;
        row = row > last ? last : row;
      }
      const float* a_row = A + row * lda;
      const float* b_row = B + row * ldb;
#pragma unroll
      for (int s = 0; s < MT; ++s) av[s][u] = a_row[a_off[s]];
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[t][u] = b_row[b_off[t]];
    }
  };
  auto step = [&](auto full_tag, int64_t k, float (&av)[MT][8], float (&bv)[NT][8]) {
    constexpr bool FULL = decltype(full_tag)::value;
    bf16x8 ah[MT], am[MT], al[MT], bh[NT], bm[NT], bl[NT];
#pragma unroll
    for (int s = 0; s < MT; ++s) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        bool ok = a_ok[s];
        if constexpr (!FULL) ok = ok && (k + 8 * lh + u < k_end);
        av[s][u] = ok ? av[s][u] : 0.0f;
        sa[s] += av[s][u];
      }
      split3x8(av[s], ah[s], am[s], al[s]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        bool ok = b_ok[t];
        if constexpr (!FULL) ok = ok && (k + 8 * lh + u < k_end);
        bv[t][u] = ok ? bv[t][u] : 0.0f;
        sb[t] += bv[t][u];
      }
      split3x8(bv[t], bh[t], bm[t], bl[t]);
    }
#pragma unroll
    for (int s = 0; s < MT; ++s)
#pragma unroll
      for (int t = 0; t < NT; ++t) {  // smallest terms first
        acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[s], bm[t], acc[s][t], 0, 0, 0);
        acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], bh[t], acc[s][t], 0, 0, 0);
        acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl[t], acc[s][t], 0, 0, 0);
        acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[s], bh[t], acc[s][t], 0, 0, 0);
        acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bm[t], acc[s][t], 0, 0, 0);
        acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh[t], acc[s][t], 0, 0, 0);
      }
  };

  // full steps two at a time with the loads of the step after next in flight; k_chunk is a
  // multiple of 16, so only the last chunk of K can end in a partial step
  float a2[MT][8], b2[NT][8];
  const int64_t n_full = (k_end - k_begin) / 16;
  int64_t k = k_begin;
  int64_t i = 0;
  if (n_full > 0) load(std::true_type{}, k, a, b);
  for (; i + 2 <= n_full; i += 2, k += 32) {
    load(std::true_type{}, k + 16, a2, b2);
    step(std::true_type{}, k, a, b);
    if (i + 2 < n_full) load(std::true_type{}, k + 32, a, b);
    step(std::true_type{}, k + 16, a2, b2);
  }
  if (i < n_full) {
    step(std::true_type{}, k, a, b);
    k += 16;
  }
  if (k < k_end) {
    load(std::false_type{}, k, a, b);
    step(std::false_type{}, k, a, b);
  }

  if (want_sa) {
#pragma unroll
    for (int s = 0; s < MT; ++s) {
      const float v = sa[s] + __shfl_xor(sa[s], 32);
      if (lh == 0 && a_ok[s]) sum_a_partial[static_cast<size_t>(blockIdx.y) * M + m0 + 32 * s + lr] = v;
    }
  }
  if (want_sb) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float v = sb[t] + __shfl_xor(sb[t], 32);
      if (lh == 0 && b_ok[t]) sum_b_partial[static_cast<size_t>(blockIdx.y) * N + n0 + 32 * t + lr] = v;
    }
  }
  float* out = partial + static_cast<size_t>(blockIdx.y) * M * N;
#pragma unroll
  for (int s = 0; s < MT; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + 32 * t + lr;
      if (n >= N) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + 32 * s + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (m < M) out[static_cast<size_t>(m) * N + n] = acc[s][t][e];
      }
    }
}

// The piece kernel for outputs made of whole 128 x 128 blocks (the gate and layer weights of
// the path: 128 x 128, 384 x 128).  A wave owns one such block and a chunk of K.
//  * Which matrix column a lane's operand vector stands for is the kernel's choice as long as
//    the output is stored by the same rule, so a lane reads FOUR neighbouring columns of eight
//    rows (eight 16-byte loads per operand and K-step, each instruction two 512-byte row
//    segments) and the four 32-column tiles of the instruction grid are the columns = 0, 1, 2,
//    3 (mod 4): tile j of lane lr is column 4 lr + j.  Against the 32 x 32-tile kernel above:
//    a quarter of the load instructions, every split piece feeds four products instead of two
//    (the split is what bounds the small tiles), and an operand is re-read by M / 128 (N / 128)
//    waves instead of M / 64 (N / 64).
//  * A K-step has three phases: every operand value is split (all vector work; the loaded
//    registers are dead afterwards), the loads of the next step are issued into them, and the
//    96 matrix instructions run while those loads are in flight.  No loaded value is used
//    later than the split of its own step, so that nothing the compiler could turn into a
//    copy (which would wait for the load) crosses the bottom of the loop.
//  * The 256 accumulators take the whole accumulation register file: one wave per SIMD.
using f32x4 = float __attribute__((ext_vector_type(4)));
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));

// The fp16 planes of an operand tile as a lane holds it (eight rows of four columns), times a
// power of two f that brings it into fp16's range, as the bit patterns the kernel keeps its
// bf16 pieces in: p0 = fp16(v), p1 = p0 / 64, p2 = fp16((v - p0) * 64); plane register j holds
// column j of the eight rows.  The scaling runs as packed multiplies over the registers as
// they were loaded (column pairs are neighbours there), in place, first by f for p0 and then
// by 64 for the residual, which comes from one mixed-precision fused multiply-add per value:
// fp16(p0 * -64 + 64 v), the fp16 operand read as it is, one rounding.  64 v is exact and
// v - p0 has at most 13 significant bits, so the value is that of widening p0, subtracting,
// scaling and narrowing.  Six vector instructions per two values.
__device__ __forceinline__ void split_tile(f32x4 (&t)[8], float f, bf16x8 (&p0)[4], bf16x8 (&p1)[4],
                                           bf16x8 (&p2)[4]) {
  using f16x2 = _Float16 __attribute__((ext_vector_type(2)));
  uint32_t a[4][4];
#pragma unroll
  for (int u = 0; u < 8; ++u) t[u] *= f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a[j][i]) : "v"(t[2 * i][j]), "v"(t[2 * i + 1][j]));
#pragma unroll
  for (int u = 0; u < 8; ++u) t[u] *= 64.0f;
  const float m64 = -64.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t b[4], c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      b[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a[j][i]) * static_cast<_Float16>(0x1p-6f));
      // the low half first: the register's other half is whatever it held and is written next
      asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(c[i]) : "v"(a[j][i]), "v"(m64), "v"(t[2 * i][j]));
      asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(c[i]) : "v"(a[j][i]), "v"(m64), "v"(t[2 * i + 1][j]));
    }
    p0[j] = __builtin_bit_cast(bf16x8, (uint4{a[j][0], a[j][1], a[j][2], a[j][3]}));
    p1[j] = __builtin_bit_cast(bf16x8, (uint4{b[0], b[1], b[2], b[3]}));
    p2[j] = __builtin_bit_cast(bf16x8, (uint4{c[0], c[1], c[2], c[3]}));
  }
}

// max(m, |x|, |y|) in one instruction (the compiler's fmaxf chain spends one more per value on
// quieting signalling NaNs).  NaN operands are passed over.
__device__ __forceinline__ float max3_abs(float m, float x, float y) {
  asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(x), "v"(y));
  return m;
}

// Largest of the lanes' non-negative floats, as a wave-uniform bit pattern.  Row shifts and row
// broadcasts of the data-parallel primitives (no LDS crossbar): lane 63 ends up with the maximum.
__device__ __forceinline__ uint32_t wave_max_bits(float v) {
  int x = static_cast<int>(__float_as_uint(v));  // non-negative floats order like their bit patterns
#define SPMI_DPP_MAX(ctrl, row_mask)                                                        \
  {                                                                                         \
    const int o = __builtin_amdgcn_update_dpp(x, x, ctrl, row_mask, 0xf, false);            \
    x = o > x ? o : x;                                                                      \
  }
  SPMI_DPP_MAX(0x111, 0xf)  // row_shr:1
  SPMI_DPP_MAX(0x112, 0xf)  // row_shr:2
  SPMI_DPP_MAX(0x114, 0xf)  // row_shr:4
  SPMI_DPP_MAX(0x118, 0xf)  // row_shr:8: lane 15 of every row holds the row's maximum
  SPMI_DPP_MAX(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
  SPMI_DPP_MAX(0x143, 0xc)  // row_bcast:31 into rows 2 and 3
#undef SPMI_DPP_MAX
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(x, 63));
}

// One launch may carry a second product of the same shape over the same rows (`second`): the
// waves of one chunk of K then run side by side for both, so that columns the two products
// share (the r and z gate gradients of a GRU layer, read by dW_ih and by dW_hh) come from
// memory once.
struct TnProblem {
  const float* A;
  int64_t lda;
  const float* B;
  int64_t ldb;
  float* partial;
  float* sum_a_partial;
  float* sum_b_partial;
};

//
// F16 = false: three exact bf16 pieces per value, six piece products per term (no range limit).
// F16 = true: the fp16 planes of cell_split.hip (p0 = fp16(v), p1 = p0 / 64, p2 = fp16((v - p0)
// * 64); x * w = p0 p0 + p1 p2 + p2 p1), three products per term: half the matrix work, which is
// what bounds this kernel (1.18 TFLOP of piece products per GRU layer at 1M rows against 3 GB).
// fp16 has 30 binades and gradients of a mean loss over a million rows sit 20 below 1, so the
// operands are scaled by powers of two on the way: every K-step (16 rows) the wave takes the
// largest exponents eA, eB of its two operand tiles.  The first step that is not all zeros fixes
// the exponent S that every product of the wave's chunk carries (taken out when the partial
// block is written): S = 14 - eA - eB, the middle of the range.  Every step then scales A by
// 2^(pA - eA) and B by 2^(pB - eB) with pA + pB = eA + eB + S split evenly, which fits fp16 as
// long as the step's operands are no more than 2^14 times larger than those S was set by.  A
// step beyond that is left out of this pass over the chunk (both operands multiplied by zero,
// no branch around the matrix instructions); the wave writes what the pass summed to its
// partial block under the pass's S and goes over the chunk again with S set by the largest
// step it left, taking only the steps left (rare, wave-uniform: one pass unless a chunk's
// operands grow 2^14-fold after its first rows).  Steps far smaller than the one that set S
// lose relative, not absolute accuracy (the residual plane keeps what plane 0 drops): the
// error stays relative to the sum of absolute terms.  Domain: |a| |b| within 2^+-100.
template <bool F16>
__global__ __launch_bounds__(kThreads, 1) void gemm_tn_partial_x3_wide_kernel(
    TnProblem first, TnProblem second, int n_problems, int64_t K, int M, int N, int64_t k_chunk,
    int n_chunks) {
  const int lane = threadIdx.x & 63;
  const int m_blocks = M / 128, blocks = m_blocks * (N / 128);
  // wave-uniform by construction; said so, row bases stay in scalar registers
  const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * kWaves + (threadIdx.x >> 6));
  const int per_chunk = blocks * n_problems;
  const int chunk = w / per_chunk;
  if (chunk >= n_chunks) return;
  const int which = (w % per_chunk) / blocks, blk = w % blocks;
  const TnProblem& pr = which ? second : first;
  const float* __restrict__ A = pr.A;
  const float* __restrict__ B = pr.B;
  const int64_t lda = pr.lda, ldb = pr.ldb;
  float* __restrict__ partial = pr.partial;
  float* __restrict__ sum_a_partial = pr.sum_a_partial;
  float* __restrict__ sum_b_partial = pr.sum_b_partial;
  const int m0 = (blk % m_blocks) * 128, n0 = (blk / m_blocks) * 128;
  const int64_t k_begin = static_cast<int64_t>(chunk) * k_chunk;
  const int64_t k_end = (k_begin + k_chunk < K) ? k_begin + k_chunk : K;
  const int lr = lane & 31, lh = lane >> 5;
  const uint32_t a_off = static_cast<uint32_t>(lh * 8 * lda + m0 + 4 * lr);
  const uint32_t b_off = static_cast<uint32_t>(lh * 8 * ldb + n0 + 4 * lr);

  f32x16 acc[4][4];  // zeroed at the top of every pass over the chunk, written out at its end
  const bool want_sa = sum_a_partial != nullptr && n0 == 0;
  const bool want_sb = sum_b_partial != nullptr && m0 == 0;
  f32x4 sa = {}, sb = {};

  // Rows past the chunk's end are read from its last row (valid memory) and zeroed.
  auto load = [&](auto full_tag, int64_t k, f32x4 (&av)[8], f32x4 (&bv)[8]) {
    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int64_t row = k + u;  // this lane reads row + 8 lh (the lane half is in the offsets)
      if constexpr (!FULL) {
        const int64_t last = k_end - 1 - 8 * lh;
        row = row > last ? last : row;
      }
      av[u] = *reinterpret_cast<const f32x4*>(A + row * lda + a_off);
      bv[u] = *reinterpret_cast<const f32x4*>(B + row * ldb + b_off);
    }
  };
  // element e of lane l in tile (s, t): row index (e&3) + 8 (e>>2) + 4 lh and column index lr
  // of the instruction grid, i.e. C[m0 + 4 * row index + s][n0 + 4 lr + t].  `exponent`: the
  // power of two the accumulators carry (F16; taken out here, in two factors so that neither
  // leaves fp32's range); `add`: the block already holds an earlier part of the chunk's sum.
  float* out = partial + static_cast<size_t>(chunk) * M * N;
  auto write_partial = [&](int exponent, bool add) {
    const int h1 = exponent / 2, h2 = exponent - h1;
    const float u1 = __uint_as_float(static_cast<uint32_t>(127 - h1) << 23);
    const float u2 = __uint_as_float(static_cast<uint32_t>(127 - h2) << 23);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + 4 * ((e & 3) + 8 * (e >> 2) + 4 * lh) + s;
        f32x4 v = {acc[s][0][e], acc[s][1][e], acc[s][2][e], acc[s][3][e]};
        f32x4* dst = reinterpret_cast<f32x4*>(out + static_cast<size_t>(m) * N + n0 + 4 * lr);
        if constexpr (F16) v = v * u1 * u2;
        if (add) v += *dst;
        *dst = v;
      }
  };
  // F16: passes over the chunk.  A pass takes the steps whose exponent sum eA + eB lies above
  // `done_below` (what earlier passes took) and fits under the pass's S; the largest sum among
  // the steps it had to leave sets the S of the next pass.  One pass unless a chunk's operands
  // grow 2^14-fold after its first rows.
  int done_below = -(1 << 20), deferred_max = -(1 << 20);
  bool flushed = false;  // the partial block already holds the sums of earlier passes
  bool first_pass = true;  // the operand column sums are taken once
  bf16x8 ah[4], am[4], al[4], bh[4], bm[4], bl[4];  // F16: planes 0, 1, 2 as bit patterns
  constexpr int kNoScale = 1 << 20;
  int run_s = kNoScale;  // F16: the exponent S every product of the chunk carries (set by its first non-zero step)
  auto split_all = [&](auto full_tag, int64_t k, f32x4 (&av)[8], f32x4 (&bv)[8]) {
    constexpr bool FULL = decltype(full_tag)::value;
    if constexpr (!FULL) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k + 8 * lh + u >= k_end) {
          av[u] = f32x4{};
          bv[u] = f32x4{};
        }
    }
    if (first_pass) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        sa += av[u];
        sb += bv[u];
      }
    }
    if constexpr (F16) {
      // largest magnitudes of the two operand tiles of this K-step (NaN is ignored by the maximum
      // and multiplied like any other value)
      float fa_max = 0.0f, fb_max = 0.0f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        fa_max = max3_abs(max3_abs(fa_max, av[u][0], av[u][1]), av[u][2], av[u][3]);
        fb_max = max3_abs(max3_abs(fb_max, bv[u][0], bv[u][1]), bv[u][2], bv[u][3]);
      }
      const uint32_t ma = wave_max_bits(fa_max), mb = wave_max_bits(fb_max);
      // A tile of zeros adds nothing, a step that an earlier pass over the chunk has taken is
      // done, and a step whose operands do not fit under this pass's S waits for the next pass:
      // in all three cases both operands are multiplied by zero (no branch around the matrix
      // instructions; the other tile, unscaled, could leave fp16's range).
      int ea = static_cast<int>(ma >> 23) - 127, eb = static_cast<int>(mb >> 23) - 127;
      ea = ea > 100 ? 100 : (ea < -100 ? -100 : ea);
      eb = eb > 100 ? 100 : (eb < -100 ? -100 : eb);
      bool live = ma != 0 && mb != 0 && ea + eb > done_below;
      if (live && run_s == kNoScale) run_s = 14 - ea - eb;  // the pass's first products land mid-range
      if (live && ea + eb + run_s > 28) {
        live = false;
        deferred_max = ea + eb > deferred_max ? ea + eb : deferred_max;
      }
      const int total = ea + eb + (run_s == kNoScale ? 0 : run_s);
      int pa = total >> 1;
      pa = pa < -100 ? -100 : pa;
      int xb2 = total - pa - eb;
      xb2 = xb2 > 126 ? 126 : (xb2 < -126 ? -126 : xb2);
      const float fa = live ? __uint_as_float(static_cast<uint32_t>(127 + pa - ea) << 23) : 0.0f;
      const float fb = live ? __uint_as_float(static_cast<uint32_t>(127 + xb2) << 23) : 0.0f;
      split_tile(av, fa, ah, am, al);
      split_tile(bv, fb, bh, bm, bl);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[8], w8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v[u] = av[u][j];
          w8[u] = bv[u][j];
        }
        split3x8(v, ah[j], am[j], al[j]);
        split3x8(w8, bh[j], bm[j], bl[j]);
      }
    }
  };
  auto multiply = [&]() {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) {  // smallest terms first
        if constexpr (F16) {
          acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[s]),
                                                             __builtin_bit_cast(f16x8, bm[t]), acc[s][t], 0, 0, 0);
          acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, am[s]),
                                                             __builtin_bit_cast(f16x8, bl[t]), acc[s][t], 0, 0, 0);
          acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[s]),
                                                             __builtin_bit_cast(f16x8, bh[t]), acc[s][t], 0, 0, 0);
        } else {
          acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[s], bm[t], acc[s][t], 0, 0, 0);
          acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], bh[t], acc[s][t], 0, 0, 0);
          acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl[t], acc[s][t], 0, 0, 0);
          acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[s], bh[t], acc[s][t], 0, 0, 0);
          acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bm[t], acc[s][t], 0, 0, 0);
          acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh[t], acc[s][t], 0, 0, 0);
        }
      }
  };

  // k_chunk is a multiple of 16, so only the last chunk of K can end in a partial step
  f32x4 a[8], b[8], a2[8], b2[8];
  for (;;) {  // one trip unless F16 has to come back for steps it left (see done_below)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][t][e] = 0.0f;
    const int64_t n_full = (k_end - k_begin) / 16;
    int64_t k = k_begin;
    if (n_full > 0) load(std::true_type{}, k, a, b);
    int64_t i = 0;
    for (; i + 2 <= n_full; i += 2, k += 32) {
      // two steps per trip, two sets of load registers: a step's loads are issued a whole step
      // ahead.  No branch around the loads (the wait counts stay exact): the last trip re-reads
      // its own rows.
      load(std::true_type{}, k + 16, a2, b2);
      __builtin_amdgcn_sched_barrier(0);
      split_all(std::true_type{}, k, a, b);
      __builtin_amdgcn_sched_barrier(0);
      multiply();
      __builtin_amdgcn_sched_barrier(0);
      load(std::true_type{}, i + 2 < n_full ? k + 32 : k, a, b);
      __builtin_amdgcn_sched_barrier(0);
      split_all(std::true_type{}, k + 16, a2, b2);
      __builtin_amdgcn_sched_barrier(0);
      multiply();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (i < n_full) {
      split_all(std::true_type{}, k, a, b);
      multiply();
      k += 16;
    }
    if (k < k_end) {
      load(std::false_type{}, k, a, b);
      split_all(std::false_type{}, k, a, b);
      multiply();
    }
    // what this pass summed goes to the partial block (on top of earlier passes' sums)
    write_partial(F16 && run_s != kNoScale ? run_s : 0, flushed);
    if (!F16 || deferred_max == -(1 << 20)) break;
    // the next pass takes the steps this one had to leave
    flushed = true;
    first_pass = false;
    done_below = 28 - run_s;
    run_s = 14 - deferred_max;
    deferred_max = -(1 << 20);
  }

  if (want_sa) {  // rows k + 0..7 and k + 8..15 of every step sit in the two lane halves
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = sa[c] + __shfl_xor(sa[c], 32);
    if (lh == 0) *reinterpret_cast<f32x4*>(sum_a_partial + static_cast<size_t>(chunk) * M + m0 + 4 * lr) = v;
  }
  if (want_sb) {
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = sb[c] + __shfl_xor(sb[c], 32);
    if (lh == 0) *reinterpret_cast<f32x4*>(sum_b_partial + static_cast<size_t>(chunk) * N + n0 + 4 * lr) = v;
  }
}

// C = (accumulate ? C : 0) + sum over chunks.  L lanes share one output element: lane c of
// the group adds chunks c, c+L, ... and the group is folded by a shuffle tree, so the order
// of additions is fixed by the shape alone.  Inside a wave the 64/L elements are the fast
// index (neighbouring lanes read neighbouring addresses).
template <int L>
__global__ __launch_bounds__(kThreads) void gemm_tn_reduce_kernel(const float* __restrict__ partial,
                                                                  int n_chunks, int64_t mn, int N,
                                                                  float* __restrict__ C, int64_t ldc,
                                                                  int accumulate) {
  constexpr int EPW = 64 / L;  // elements per wave
  const int lane = threadIdx.x & 63;
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * kWaves + (threadIdx.x >> 6);
  const int64_t i = wave * EPW + (lane % EPW);
  const int cl = lane / EPW;
  float s0 = 0.f, s1 = 0.f;
  if (i < mn) {
    int c = cl;
    for (; c + L < n_chunks; c += 2 * L) {  // two independent loads per trip
      s0 += partial[static_cast<size_t>(c) * mn + i];
      s1 += partial[static_cast<size_t>(c + L) * mn + i];
    }
    if (c < n_chunks) s0 += partial[static_cast<size_t>(c) * mn + i];
  }
  float s = s0 + s1;
#pragma unroll
  for (int off = 32; off >= EPW; off >>= 1) s += __shfl_xor(s, off);
  if (i < mn && cl == 0) {
    float* dst = C + (i / N) * ldc + (i % N);
    *dst = accumulate ? *dst + s : s;
  }
}

struct Shape {
  int mt;         // 32-row strips per wave (bf16-piece kernel only)
  int nt;         // 32-column tiles per wave
  int m_strips;   // ceil(M / 32)
  int n_tiles;    // ceil(N / (32 * nt))
  int n_chunks;   // pieces of K
  int64_t k_chunk;
};

Shape plan(int64_t M, int64_t N, int64_t K, bool pieces) {
  Shape s{};
  s.mt = (pieces && M > 32) ? 2 : 1;
  s.nt = pieces ? (N > 32 ? 2 : 1) : (N > 64 ? 4 : (N > 32 ? 2 : 1));
  s.m_strips = static_cast<int>((M + 32 * s.mt - 1) / (32 * s.mt));
  s.n_tiles = static_cast<int>((N + 32 * s.nt - 1) / (32 * s.nt));
  const int64_t blocks = s.m_strips * s.n_tiles;
  // about eight waves per SIMD over the whole chip, at least 256 rows of K per wave
  int64_t chunks = 8192 / blocks;
  const int64_t max_chunks = (K + 255) / 256;
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  s.k_chunk = ((K + chunks - 1) / chunks + 15) / 16 * 16;
  s.n_chunks = static_cast<int>((K + s.k_chunk - 1) / s.k_chunk);
  return s;
}

// The 128 x 128-block kernel: one wave per SIMD over the whole chip, one block and one chunk
// of K per wave.
bool wide_shape(int64_t M, int64_t N) { return M % 128 == 0 && N % 128 == 0 && M * N <= 512 * 512; }

Shape plan_wide(int64_t M, int64_t N, int64_t K, int n_problems = 1) {
  Shape s{};
  s.mt = s.nt = 4;
  s.m_strips = static_cast<int>(M / 128);
  s.n_tiles = static_cast<int>(N / 128);
  const int64_t blocks = static_cast<int64_t>(s.m_strips) * s.n_tiles * n_problems;
  // one wave per SIMD over the whole chip
  int64_t chunks = 1024 / blocks;
  const int64_t max_chunks = (K + 255) / 256;
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  s.k_chunk = ((K + chunks - 1) / chunks + 15) / 16 * 16;
  s.n_chunks = static_cast<int>((K + s.k_chunk - 1) / s.k_chunk);
  return s;
}

int check_gemm_tn(int64_t M, int64_t N, int64_t K) {
  SPMI_REQUIRE(M >= 1 && N >= 1 && K >= 1 && M <= 4096 && N <= 4096 && K < (int64_t{1} << 40),
               SPMI_ERR_INVALID_ARG, "spmi_gemm_tn: bad sizes M=%lld N=%lld K=%lld (M, N <= 4096)",
               (long long)M, (long long)N, (long long)K);
  return SPMI_OK;
}

// C = (accumulate ? C : 0) + sum over the chunks' partial products, and the operand column sums
// by the same fixed-order reduction over vectors of M / N entries.
int reduce_chunks(const float* partial, const float* pa, const float* pb, int n_chunks, int64_t M, int64_t N,
                  float* C, int64_t ldc, int accumulate, float* col_sum_a, float* col_sum_b, hipStream_t st) {
  const int64_t mn = M * N;
  if (n_chunks >= 64) {
    const unsigned blocks = static_cast<unsigned>((mn + 4 * kWaves - 1) / (4 * kWaves));
    gemm_tn_reduce_kernel<16><<<blocks, kThreads, 0, st>>>(partial, n_chunks, mn, (int)N, C, ldc, accumulate);
  } else {
    const unsigned blocks = static_cast<unsigned>((mn + kThreads - 1) / kThreads);
    gemm_tn_reduce_kernel<1><<<blocks, kThreads, 0, st>>>(partial, n_chunks, mn, (int)N, C, ldc, accumulate);
  }
  SPMI_CHECK_LAUNCH("gemm_tn_reduce");
  const float* vec_src[2] = {pa, pb};
  float* vec_dst[2] = {col_sum_a, col_sum_b};
  const int64_t vec_len[2] = {M, N};
  for (int v = 0; v < 2; ++v) {
    if (!vec_dst[v]) continue;
    const int64_t len = vec_len[v];
    if (n_chunks >= 64) {
      const unsigned blocks = static_cast<unsigned>((len + 4 * kWaves - 1) / (4 * kWaves));
      gemm_tn_reduce_kernel<16><<<blocks, kThreads, 0, st>>>(vec_src[v], n_chunks, len, (int)len, vec_dst[v], len, 0);
    } else {
      const unsigned blocks = static_cast<unsigned>((len + kThreads - 1) / kThreads);
      gemm_tn_reduce_kernel<1><<<blocks, kThreads, 0, st>>>(vec_src[v], n_chunks, len, (int)len, vec_dst[v], len, 0);
    }
    SPMI_CHECK_LAUNCH("gemm_tn_reduce(sums)");
  }
  return SPMI_OK;
}

}  // namespace

extern "C" {

size_t spmi_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M < 1 || N < 1 || K < 1 || M > 4096 || N > 4096) return 0;
  // the two kernels cut K differently: room for the finer cut
  const Shape s0 = plan(M, N, K, false), s1 = plan(M, N, K, true);
  int chunks = s0.n_chunks > s1.n_chunks ? s0.n_chunks : s1.n_chunks;
  if (wide_shape(M, N)) {
    const int cw = plan_wide(M, N, K).n_chunks;
    chunks = cw > chunks ? cw : chunks;
  }
  return sizeof(float) * static_cast<size_t>(chunks) * (M * N + M + N);
}

int spmi_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t K, int64_t M,
                 int64_t N, float* C, int64_t ldc, int accumulate, void* ws, size_t ws_bytes,
                 void* stream) {
  return spmi_gemm_tn_sums(A, lda, B, ldb, K, M, N, C, ldc, accumulate, nullptr, nullptr, 1, ws,
                           ws_bytes, stream);
}

int spmi_gemm_tn_sums(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t K, int64_t M,
                      int64_t N, float* C, int64_t ldc, int accumulate, float* col_sum_a,
                      float* col_sum_b, int fp32_instruction, void* ws, size_t ws_bytes,
                      void* stream) {
  if (int rc = check_gemm_tn(M, N, K)) return rc;
  SPMI_REQUIRE(A && B && C && ws, SPMI_ERR_INVALID_ARG, "spmi_gemm_tn: null pointer argument");
  SPMI_REQUIRE(lda >= M && ldb >= N && ldc >= N, SPMI_ERR_INVALID_ARG,
               "spmi_gemm_tn: leading dimensions (%lld,%lld,%lld) too small", (long long)lda,
               (long long)ldb, (long long)ldc);
  SPMI_REQUIRE(ws_bytes >= spmi_gemm_tn_workspace_bytes(M, N, K), SPMI_ERR_WORKSPACE,
               "spmi_gemm_tn: workspace %zu < required %zu", ws_bytes,
               spmi_gemm_tn_workspace_bytes(M, N, K));
  const bool pieces = fp32_instruction != 1;
  const bool wide = pieces && wide_shape(M, N) && lda % 4 == 0 && ldb % 4 == 0 &&
                    ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15u) == 0;
  const Shape s = wide ? plan_wide(M, N, K) : plan(M, N, K, pieces);
  SPMI_REQUIRE(lda < (int64_t{1} << 27) && ldb < (int64_t{1} << 27), SPMI_ERR_UNSUPPORTED,
               "spmi_gemm_tn: leading dimension too large");
  hipStream_t st = spmi::as_stream(stream);
  float* partial = static_cast<float*>(ws);
  float* pa = col_sum_a ? partial + static_cast<size_t>(s.n_chunks) * M * N : nullptr;
  float* pb = col_sum_b ? partial + static_cast<size_t>(s.n_chunks) * (M * N + M) : nullptr;
  const dim3 grid((s.m_strips * s.n_tiles + kWaves - 1) / kWaves, s.n_chunks);
#define SPMI_GEMM_TN(NT)                                                                        \
  gemm_tn_partial_kernel<NT><<<grid, kThreads, 0, st>>>(A, lda, B, ldb, K, (int)M, (int)N,     \
                                                        s.k_chunk, partial, s.m_strips, s.n_tiles, pa, pb)
#define SPMI_GEMM_TN_X3(MT, NT)                                                                   \
  gemm_tn_partial_x3_kernel<MT, NT><<<grid, kThreads, 0, st>>>(A, lda, B, ldb, K, (int)M, (int)N, \
                                                               s.k_chunk, partial, s.m_strips,   \
                                                               s.n_tiles, pa, pb)
  if (wide) {
    const unsigned wgs = static_cast<unsigned>((s.m_strips * s.n_tiles * s.n_chunks + kWaves - 1) / kWaves);
    const TnProblem one{A, lda, B, ldb, partial, pa, pb};
    if (fp32_instruction == 2)
      gemm_tn_partial_x3_wide_kernel<true><<<wgs, kThreads, 0, st>>>(one, one, 1, K, (int)M, (int)N, s.k_chunk,
                                                                     s.n_chunks);
    else
      gemm_tn_partial_x3_wide_kernel<false><<<wgs, kThreads, 0, st>>>(one, one, 1, K, (int)M, (int)N, s.k_chunk,
                                                                      s.n_chunks);
  } else if (pieces) {
    if (s.mt == 2 && s.nt == 2) SPMI_GEMM_TN_X3(2, 2);
    else if (s.mt == 2) SPMI_GEMM_TN_X3(2, 1);
    else if (s.nt == 2) SPMI_GEMM_TN_X3(1, 2);
    else SPMI_GEMM_TN_X3(1, 1);
  } else if (s.nt == 4) SPMI_GEMM_TN(4);
  else if (s.nt == 2) SPMI_GEMM_TN(2);
  else SPMI_GEMM_TN(1);
#undef SPMI_GEMM_TN
#undef SPMI_GEMM_TN_X3
  SPMI_CHECK_LAUNCH("gemm_tn_partial");
  return reduce_chunks(partial, pa, pb, s.n_chunks, M, N, C, ldc, accumulate, col_sum_a, col_sum_b, st);
}

int spmi_gemm_tn_pair_supported(int64_t M, int64_t N) { return wide_shape(M, N) ? 1 : 0; }

size_t spmi_gemm_tn_pair_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (!wide_shape(M, N) || K < 1) return 0;
  const int chunks = plan_wide(M, N, K, 2).n_chunks;
  return 2 * sizeof(float) * static_cast<size_t>(chunks) * (M * N + M + N);
}

int spmi_gemm_tn_pair_sums(const float* A0, int64_t lda0, const float* B0, int64_t ldb0, const float* A1,
                           int64_t lda1, const float* B1, int64_t ldb1, int64_t K, int64_t M, int64_t N,
                           float* C0, int64_t ldc0, float* C1, int64_t ldc1, float* col_sum_a0,
                           float* col_sum_a1, int f16_planes, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_gemm_tn(M, N, K)) return rc;
  SPMI_REQUIRE(wide_shape(M, N), SPMI_ERR_UNSUPPORTED,
               "spmi_gemm_tn_pair_sums: M and N must be multiples of 128 with M*N <= 512*512 (M=%lld N=%lld)",
               (long long)M, (long long)N);
  SPMI_REQUIRE(A0 && B0 && A1 && B1 && C0 && C1 && ws, SPMI_ERR_INVALID_ARG,
               "spmi_gemm_tn_pair_sums: null pointer argument");
  SPMI_REQUIRE(lda0 >= M && lda1 >= M && ldb0 >= N && ldb1 >= N && ldc0 >= N && ldc1 >= N, SPMI_ERR_INVALID_ARG,
               "spmi_gemm_tn_pair_sums: a leading dimension is too small");
  SPMI_REQUIRE(lda0 % 4 == 0 && lda1 % 4 == 0 && ldb0 % 4 == 0 && ldb1 % 4 == 0 &&
                   ((reinterpret_cast<uintptr_t>(A0) | reinterpret_cast<uintptr_t>(B0) |
                     reinterpret_cast<uintptr_t>(A1) | reinterpret_cast<uintptr_t>(B1)) & 15u) == 0,
               SPMI_ERR_INVALID_ARG,
               "spmi_gemm_tn_pair_sums: operands must be 16-byte aligned with leading dimensions in multiples of 4");
  SPMI_REQUIRE(lda0 < (int64_t{1} << 27) && ldb0 < (int64_t{1} << 27) && lda1 < (int64_t{1} << 27) &&
                   ldb1 < (int64_t{1} << 27),
               SPMI_ERR_UNSUPPORTED, "spmi_gemm_tn_pair_sums: leading dimension too large");
  SPMI_REQUIRE(ws_bytes >= spmi_gemm_tn_pair_workspace_bytes(M, N, K), SPMI_ERR_WORKSPACE,
               "spmi_gemm_tn_pair_sums: workspace %zu < required %zu", ws_bytes,
               spmi_gemm_tn_pair_workspace_bytes(M, N, K));
  const Shape s = plan_wide(M, N, K, 2);
  hipStream_t st = spmi::as_stream(stream);
  const size_t per_problem = static_cast<size_t>(s.n_chunks) * (M * N + M + N);
  float* p0 = static_cast<float*>(ws);
  float* p1 = p0 + per_problem;
  float* sa0 = col_sum_a0 ? p0 + static_cast<size_t>(s.n_chunks) * M * N : nullptr;
  float* sa1 = col_sum_a1 ? p1 + static_cast<size_t>(s.n_chunks) * M * N : nullptr;
  const TnProblem first{A0, lda0, B0, ldb0, p0, sa0, nullptr};
  const TnProblem second{A1, lda1, B1, ldb1, p1, sa1, nullptr};
  const int64_t waves = static_cast<int64_t>(s.m_strips) * s.n_tiles * 2 * s.n_chunks;
  const unsigned wgs = static_cast<unsigned>((waves + kWaves - 1) / kWaves);
  if (f16_planes)
    gemm_tn_partial_x3_wide_kernel<true><<<wgs, kThreads, 0, st>>>(first, second, 2, K, (int)M, (int)N, s.k_chunk,
                                                                   s.n_chunks);
  else
    gemm_tn_partial_x3_wide_kernel<false><<<wgs, kThreads, 0, st>>>(first, second, 2, K, (int)M, (int)N, s.k_chunk,
                                                                    s.n_chunks);
  SPMI_CHECK_LAUNCH("gemm_tn_partial(pair)");
  if (int rc = reduce_chunks(p0, sa0, nullptr, s.n_chunks, M, N, C0, ldc0, 0, col_sum_a0, nullptr, st)) return rc;
  return reduce_chunks(p1, sa1, nullptr, s.n_chunks, M, N, C1, ldc1, 0, col_sum_a1, nullptr, st);
}

}  // extern "C"
