// One recurrent layer step as ONE kernel for gfx950, fp32 results from the bf16 matrix
// cores: every fp32 operand is split exactly into three bf16 pieces
// (v = hi + mid + lo, 3 x 8 significand bits = the 24 of fp32) and a product x*w is summed
// from the six piece products of weight above 2^-24 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi,
// mid*mid), each exact in the fp32 accumulator's input.  The dropped terms are below
// 3 * 2^-24 |x||w|, the size of fp32's own rounding, so the gate pre-activations agree with
// an fp32 FMA chain to a few units of fp32 rounding (measured against the CPU checker in the
// tests), while six v_mfma_f32_32x32x16_bf16 (32 cycles each, K = 16) replace eight
// v_mfma_f32_32x32x2_f32 (64 cycles each): 2.7x the matrix rate.
//
// A second operand format (MODE_F16X3) halves the matrix work again: three fp16 planes
//     p0 = fp16(v),   p1 = fp16(v * 2^-6),   p2 = fp16((v - p0) * 2^6)
// and three v_mfma_f32_32x32x16_f16 per term,
//     x*w  =  p0(x)*p0(w)  +  p1(x)*p2(w)  +  p2(x)*p1(w).
// With x = xh + xl, w = wh + wl (xh = p0(x), xl = x - xh exactly, |xl| <= 2^-11 |x|) the
// exact product is xh*wh + x*wl + xl*wh, and the last two terms are what the scaled planes
// carry (2^-6 * 2^6 = 1): each factor is rounded to 11 bits once, so a product is off by at
// most about 2^-20 |xw|.  The scaling keeps the small planes out of fp16's subnormal range
// for |v| down to about 2^-8; below that they lose relative but not absolute accuracy (the
// matrix cores keep subnormal fp16 inputs, scripts/probe/mfma_f16_denorm.hip).  Measured
// against float64 the sums are 3-8x closer than an fp32 FMA chain of the same length.
// Domain: the state and the weights within +-65504.  The rows of x may hold any finite value:
// unless the caller declares them bounded, a workgroup first reads its 128 rows of x, and a
// row whose largest magnitude reaches 2^15 is staged multiplied by the power of two that
// brings it to [2^13, 2^14); the accumulators of that row are multiplied by the inverse power
// after the last x term, before the first state term.  GraphConv outputs of graphs with
// count-valued features (degrees, citation counts) do get that large.
//
// Replaces the same reference lines as cell_fused.hip: torch.nn.GRU / LSTM / RNN as driven
// by SequentialDecoder.forward / LSTM.forward (spiro/dtdg/models/decoder/torch/
// sequentialDecoder/implicitTimeModels.py:144-158, :214-230) with NodeMemory.get_memory /
// update_memory (:48-54).
//
// Mapping (as in cell_fused.hip): a workgroup owns 128 batch rows and all hidden units of
// them; waves form a 4 x WC grid of 32-row groups x column groups, JSUB 32-unit column
// blocks x NACC accumulator sets per wave.  K runs in steps of 16 through a double-buffered
// LDS stage.  Weights arrive pre-split and pre-tiled ([K/16][3][2][G*H][8] 16-bit words: the B image
// of a K-step is one contiguous block); x and state rows are split while they are staged.
// Both operand images are [plane][k half][row or n][8 k]: the 32 lanes of a half read 512
// contiguous bytes with one ds_read_b128 (conflict-free), one read per plane.

#include <type_traits>

#include "common.h"

namespace {

constexpr int kRows = 128;
constexpr int kBK = 16;

using f32x16 = float __attribute__((ext_vector_type(16)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using bf16x4 = __bf16 __attribute__((ext_vector_type(4)));
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));
using f16x4 = _Float16 __attribute__((ext_vector_type(4)));

enum { MODE_BF16X3 = 0, MODE_F16X3 = 1 };

enum { CELL_GRU = 0, CELL_LSTM = 1, CELL_RNN = 2 };

template <int CELL>
struct Cell;
template <>
struct Cell<CELL_GRU> {
  static constexpr int NG = 3, NACC = 4;
  static constexpr int acc_x[3] = {0, 1, 2};
  static constexpr int acc_h[3] = {0, 1, 3};
};
template <>
struct Cell<CELL_LSTM> {
  static constexpr int NG = 4, NACC = 4;
  static constexpr int acc_x[4] = {0, 1, 2, 3};
  static constexpr int acc_h[4] = {0, 1, 2, 3};
};
template <>
struct Cell<CELL_RNN> {
  static constexpr int NG = 1, NACC = 1;
  static constexpr int acc_x[1] = {0};
  static constexpr int acc_h[1] = {0};
};

struct SplitArgs {
  const float* x;
  int64_t ldx;
  int I;
  const __bf16* ws_ih;  // [I/16][3][NG*H][16]
  const __bf16* ws_hh;  // [H/16][3][NG*H][16]
  const float* b_ih;
  const float* b_hh;
  float* h_mem;
  float* c_mem;
  const int64_t* ids;
  int64_t M;
  int H;
  float* h_out;
  float* c_out;
  int mem_write;
  float* saved;
  int x_bounded;  // the caller vouches for |x| < 2^15: skip the row-magnitude pass
};

// v_rcp_f32 (1 ulp).  __frcp_rn is the correctly rounded reciprocal: a full division sequence
// of a dozen instructions, twice per state element.
__device__ __forceinline__ float rcpf_(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float sigmoidf_(float v) { return rcpf_(1.0f + __expf(-v)); }
__device__ __forceinline__ float tanhf_(float v) {
  const float e = __expf(2.0f * (v > 15.0f ? 15.0f : v));  // NaN stays NaN
  return 1.0f - 2.0f * rcpf_(e + 1.0f);
}

// v = hi + mid + lo exactly (each difference of an fp32 and its bf16 rounding is exact).
__device__ __forceinline__ void split3(float v, __bf16& hi, __bf16& mid, __bf16& lo) {
  hi = static_cast<__bf16>(v);
  const float r1 = v - static_cast<float>(hi);
  mid = static_cast<__bf16>(r1);
  lo = static_cast<__bf16>(r1 - static_cast<float>(mid));
}

// p0 = fp16(v), p1 = fp16(v / 64), p2 = fp16((v - p0) * 64).
__device__ __forceinline__ void split_f16x3(float v, _Float16& p0, _Float16& p1, _Float16& p2) {
  p0 = static_cast<_Float16>(v);
  p1 = static_cast<_Float16>(v * 0x1p-6f);
  p2 = static_cast<_Float16>((v - static_cast<float>(p0)) * 0x1p6f);
}

// Weights [rows][cols] fp32 -> [cols/16][3 planes][2 k halves][rows][8] 16-bit words: for one
// K-step the plane images of all rows are one contiguous block in the order the LDS stage
// wants them, so the staging loads are fully coalesced.  A fragment read (lane (row, half)
// takes 16 bytes) then covers 512 contiguous bytes per half, which ds_read_b128 serves
// without bank conflicts; rows of 16 k (32 bytes, lanes 32 bytes apart) cost two-way
// conflicts on every read (SQ_LDS_BANK_CONFLICT was 38 % of the LDS-array cycles).
template <int MODE>
__global__ void split_weights_kernel(const float* __restrict__ w, int64_t rows, int64_t cols,
                                     int64_t row_stride, int64_t col_stride,
                                     uint16_t* __restrict__ out, uint32_t* __restrict__ amax_bits) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool inside = i < rows * cols;
  const int64_t n = inside ? i / cols : 0, k = inside ? i % cols : 0;
  const float v = w[n * row_stride + k * col_stride];
  if (amax_bits) {
    // largest |w| as its bit pattern: among non-negative floats the integer order is the
    // float order, with NaN above infinity, so a NaN weight is reported and not dropped
    uint32_t b = inside ? (__float_as_uint(v) & 0x7fffffffu) : 0u;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const uint32_t o = __shfl_xor(b, off);
      b = o > b ? o : b;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(amax_bits, b);
  }
  if (!inside) return;
  uint16_t q0, q1, q2;
  if constexpr (MODE == MODE_F16X3) {
    _Float16 p0, p1, p2;
    split_f16x3(v, p0, p1, p2);
    q0 = __builtin_bit_cast(uint16_t, p0);
    q1 = __builtin_bit_cast(uint16_t, p1);
    q2 = __builtin_bit_cast(uint16_t, p2);
  } else {
    __bf16 hi, mid, lo;
    split3(v, hi, mid, lo);
    q0 = __builtin_bit_cast(uint16_t, hi);
    q1 = __builtin_bit_cast(uint16_t, mid);
    q2 = __builtin_bit_cast(uint16_t, lo);
  }
  const int64_t base = (k / 16) * 3 * rows * 16 + ((k % 16) / 8) * rows * 8 + n * 8 + (k % 8);
  out[base] = q0;
  out[base + rows * 16] = q1;
  out[base + 2 * rows * 16] = q2;
}

// Plane 1 of eight fp16-plane weights from their plane 0: p1 = fp16(w / 64) = p0 / 64 exactly
// while p0 / 64 is a normal fp16 number (|w| >= 2^-8); below that the quotient is rounded once
// more, by at most 2^-25, in a term that is multiplied by the 2^-5 |x| of the x residual.  The
// weight image in memory keeps all three planes; the kernels here read two of them.
__device__ __forceinline__ uint4 plane1_of(const uint4& p0) {
  const f16x8 v = __builtin_bit_cast(f16x8, p0) * static_cast<_Float16>(0x1p-6f);
  return __builtin_bit_cast(uint4, v);
}

template <int APT, int BPT>
struct StageRegs {
  float4 a[APT];
  uint4 b[BPT];
  float sc[APT];  // power-of-two scale of the row each a[i] belongs to (1 unless the row is huge)
};

// Global loads of one K-step into registers.  Addresses are "wave-uniform base + small
// per-thread offset" wherever the rows are contiguous (x always; the state unless `ids`
// redirects rows), which keeps the loop free of 64-bit per-thread pointers.
template <int APT, int BPT, typename AOff>
__device__ __forceinline__ void load_stage(StageRegs<APT, BPT>& r, const float* abase,
                                           const AOff (&aoff)[APT], int k0, const __bf16* bbase,
                                           const int (&boff)[BPT], int brow0) {
#pragma unroll
  for (int i = 0; i < APT; ++i) r.a[i] = *reinterpret_cast<const float4*>(abase + aoff[i] + k0);
  const __bf16* b = bbase + static_cast<int64_t>(brow0) * 16;  // wave-uniform K-step base
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    // chunks past the tile read row 0 of the step (valid memory) and are not stored
    r.b[i] = *reinterpret_cast<const uint4*>(b + boff[i]);
  }
}

template <int MODE, int APT, int BPT, int B16, int THREADS, int A_PLANE, int A_BYTES>
__device__ __forceinline__ void store_stage(const StageRegs<APT, BPT>& r, char* stage,
                                            const int (&a_dst)[APT], int tid) {
#pragma unroll
  for (int i = 0; i < APT; ++i) {
    const float m = MODE == MODE_F16X3 ? r.sc[i] : 1.0f;
    const float v[4] = {r.a[i].x * m, r.a[i].y * m, r.a[i].z * m, r.a[i].w * m};
    if constexpr (MODE == MODE_F16X3) {
      f16x4 p0, p1, p2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        _Float16 h1, m1, l1;
        split_f16x3(v[q], h1, m1, l1);
        p0[q] = h1;
        p1[q] = m1;
        p2[q] = l1;
      }
      *reinterpret_cast<f16x4*>(stage + a_dst[i]) = p0;
      *reinterpret_cast<f16x4*>(stage + A_PLANE + a_dst[i]) = p1;
      *reinterpret_cast<f16x4*>(stage + 2 * A_PLANE + a_dst[i]) = p2;
    } else {
      bf16x4 hi, mid, lo;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        __bf16 h1, m1, l1;
        split3(v[q], h1, m1, l1);
        hi[q] = h1;
        mid[q] = m1;
        lo[q] = l1;
      }
      *reinterpret_cast<bf16x4*>(stage + a_dst[i]) = hi;
      *reinterpret_cast<bf16x4*>(stage + A_PLANE + a_dst[i]) = mid;
      *reinterpret_cast<bf16x4*>(stage + 2 * A_PLANE + a_dst[i]) = lo;
    }
  }
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int c = tid + i * THREADS;
    if ((i + 1) * THREADS <= B16 || c < B16) {
      if constexpr (MODE == MODE_F16X3) {
        // planes 0 and 2 were read (B16 = their chunks, B16 / 2 each); plane 1 is made here
        char* dst = stage + A_BYTES + c * 16;
        if (c < B16 / 2) {  // wave-uniform: a plane is a multiple of 64 chunks
          *reinterpret_cast<uint4*>(dst) = r.b[i];
          *reinterpret_cast<uint4*>(dst + (B16 / 2) * 16) = plane1_of(r.b[i]);
        } else {
          *reinterpret_cast<uint4*>(dst + (B16 / 2) * 16) = r.b[i];
        }
      } else {
        *reinterpret_cast<uint4*>(stage + A_BYTES + c * 16) = r.b[i];
      }
    }
  }
}

// The MFMAs of one K-step (16) of one wave: three A fragments, then per 32-column block
// three B fragments and the six piece products, smallest terms first.  The next block's
// fragments are read while the current block's MFMAs issue.
template <int MODE, int CELL, int JSUB, int JC, int A_PLANE, int B_PLANE, bool HPHASE>
__device__ __forceinline__ void mfma_stage(f32x16 (&acc)[JSUB][Cell<CELL>::NACC], const char* sa,
                                           const char* sb) {
  using C = Cell<CELL>;
  constexpr int NB = C::NG * JSUB;
  const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(sa);
  const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(sa + A_PLANE);
  const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(sa + 2 * A_PLANE);
  // Two waves per SIMD (JSUB == 2): the next block's fragments are read while the current
  // block's MFMAs issue.  Four waves per SIMD (JSUB == 1): the other waves cover the LDS
  // latency and the registers of a second fragment set are not affordable.
  constexpr bool PREFETCH = JSUB > 1;
  bf16x8 b_cur[3], b_nxt[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) b_cur[q] = *reinterpret_cast<const bf16x8*>(sb + q * B_PLANE);
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    if (PREFETCH && i + 1 < NB) {
      const int off = (((i + 1) / JSUB) * JC + ((i + 1) % JSUB) * 32) * 16;
#pragma unroll
      for (int q = 0; q < 3; ++q) b_nxt[q] = *reinterpret_cast<const bf16x8*>(sb + off + q * B_PLANE);
    }
    if (PREFETCH) __builtin_amdgcn_sched_barrier(0);
    const int g = i / JSUB, sub = i % JSUB;
    const int ai = HPHASE ? C::acc_h[g] : C::acc_x[g];
    if constexpr (MODE == MODE_F16X3) {
      acc[sub][ai] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a2),
                                                           __builtin_bit_cast(f16x8, b_cur[1]),
                                                           acc[sub][ai], 0, 0, 0);
      acc[sub][ai] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1),
                                                           __builtin_bit_cast(f16x8, b_cur[2]),
                                                           acc[sub][ai], 0, 0, 0);
      acc[sub][ai] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0),
                                                           __builtin_bit_cast(f16x8, b_cur[0]),
                                                           acc[sub][ai], 0, 0, 0);
    } else {
      acc[sub][ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b_cur[0], acc[sub][ai], 0, 0, 0);
      acc[sub][ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b_cur[2], acc[sub][ai], 0, 0, 0);
      acc[sub][ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b_cur[1], acc[sub][ai], 0, 0, 0);
      acc[sub][ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b_cur[0], acc[sub][ai], 0, 0, 0);
      acc[sub][ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b_cur[1], acc[sub][ai], 0, 0, 0);
      acc[sub][ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b_cur[0], acc[sub][ai], 0, 0, 0);
    }
    if (PREFETCH) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 3; ++q) b_cur[q] = b_nxt[q];
    } else if (i + 1 < NB) {
      const int off = (((i + 1) / JSUB) * JC + ((i + 1) % JSUB) * 32) * 16;
#pragma unroll
      for (int q = 0; q < 3; ++q) b_cur[q] = *reinterpret_cast<const bf16x8*>(sb + off + q * B_PLANE);
    }
  }
}

// One slice of the staging work of a K-step: slices 0 .. 5 APT - 1 split the x/state rows
// (two elements per slice) and store their three planes, the rest store the weight chunks.
// `k` is a constant after unrolling.  The last, partly filled weight chunk is stored
// unconditionally (threads past the tile write a dump row behind the stages).
template <int APT, int BPT, int B16, int THREADS, int A_PLANE, int A_BYTES>
__device__ __forceinline__ void stage_slice(int k, const StageRegs<APT, BPT>& r, f16x4 (&p0)[APT],
                                            f16x4 (&p1)[APT], f16x4 (&p2)[APT], char* stage,
                                            char* dump, const int (&a_dst)[APT], int tid) {
  if (k < 5 * APT) {
    const int i = k / 5, part = k % 5;
    if (part < 2) {
      const float v0 = (part == 0 ? r.a[i].x : r.a[i].z) * r.sc[i];
      const float v1 = (part == 0 ? r.a[i].y : r.a[i].w) * r.sc[i];
      _Float16 h, m, l;
      split_f16x3(v0, h, m, l);
      p0[i][2 * part] = h;
      p1[i][2 * part] = m;
      p2[i][2 * part] = l;
      split_f16x3(v1, h, m, l);
      p0[i][2 * part + 1] = h;
      p1[i][2 * part + 1] = m;
      p2[i][2 * part + 1] = l;
    } else if (part == 2) {
      *reinterpret_cast<f16x4*>(stage + a_dst[i]) = p0[i];
    } else if (part == 3) {
      *reinterpret_cast<f16x4*>(stage + A_PLANE + a_dst[i]) = p1[i];
    } else {
      *reinterpret_cast<f16x4*>(stage + 2 * A_PLANE + a_dst[i]) = p2[i];
    }
  } else {
    const int i = k - 5 * APT;
    const int c = tid + i * THREADS;
    const bool inside = (i + 1) * THREADS <= B16 || c < B16;
    char* dst = stage + A_BYTES + c * 16;
    if (c < B16 / 2) {  // plane 0 (wave-uniform): store it and the plane 1 made from it
      *reinterpret_cast<uint4*>(inside ? dst : dump) = r.b[i];
      *reinterpret_cast<uint4*>(inside ? dst + (B16 / 2) * 16 : dump) = plane1_of(r.b[i]);
    } else {            // plane 2 sits one plane further than its chunk number says
      *reinterpret_cast<uint4*>(inside ? dst + (B16 / 2) * 16 : dump) = r.b[i];
    }
  }
}

// One K-step of the pipelined loop (f16x3 only): the matrix instructions of step t with the
// staging of step t+1 placed BETWEEN them.  Waves issue in order, so staging overlaps matrix
// work only where the program order says so; the slices start after the first few MFMAs
// because the global loads of step t+1 were issued one step earlier and need that long.
// A fence after every 32-column block keeps the compiler from regrouping the sequence.
template <int CELL, int JSUB, int JC, int A_PLANE, int B_PLANE, bool HPHASE, int APT, int BPT,
          int B16, int THREADS, int A_BYTES>
__device__ __forceinline__ void piped_step(f32x16 (&acc)[JSUB][Cell<CELL>::NACC], const char* sa,
                                           const char* sb, const StageRegs<APT, BPT>& r,
                                           char* stage, char* dump, const int (&a_dst)[APT],
                                           int tid) {
  using C = Cell<CELL>;
  constexpr int NB = C::NG * JSUB;
  constexpr int SLOTS = 3 * NB;
  constexpr int SLICES = 5 * APT + BPT;
  constexpr int LEAD = SLOTS - SLICES < 0 ? 0 : (SLOTS - SLICES < 5 ? SLOTS - SLICES : 5);
  constexpr int LATE = SLOTS - LEAD;  // slots that carry staging work
  const f16x8 a0 = *reinterpret_cast<const f16x8*>(sa);
  const f16x8 a1 = *reinterpret_cast<const f16x8*>(sa + A_PLANE);
  const f16x8 a2 = *reinterpret_cast<const f16x8*>(sa + 2 * A_PLANE);
  f16x8 b_cur[3], b_nxt[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) b_cur[q] = *reinterpret_cast<const f16x8*>(sb + q * B_PLANE);
  f16x4 p0[APT], p1[APT], p2[APT];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    if (i + 1 < NB) {
      const int off = (((i + 1) / JSUB) * JC + ((i + 1) % JSUB) * 32) * 16;
#pragma unroll
      for (int q = 0; q < 3; ++q) b_nxt[q] = *reinterpret_cast<const f16x8*>(sb + off + q * B_PLANE);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int g = i / JSUB, sub = i % JSUB;
    const int ai = HPHASE ? C::acc_h[g] : C::acc_x[g];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const f16x8& fa = m == 0 ? a2 : (m == 1 ? a1 : a0);
      const f16x8& fb = m == 0 ? b_cur[1] : (m == 1 ? b_cur[2] : b_cur[0]);
      acc[sub][ai] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[sub][ai], 0, 0, 0);
      const int slot = 3 * i + m - LEAD;
      if (slot >= 0) {
#pragma unroll
        for (int k = slot * SLICES / LATE; k < (slot + 1) * SLICES / LATE; ++k)
          stage_slice<APT, BPT, B16, THREADS, A_PLANE, A_BYTES>(k, r, p0, p1, p2, stage, dump, a_dst, tid);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 3; ++q) b_cur[q] = b_nxt[q];
  }
}

// Gate math of one state element; `h` / `c` are the old state, the new one is returned.
template <int CELL>
__device__ __forceinline__ float gate_math(const float (&a)[Cell<CELL>::NACC], const float (&bi)[Cell<CELL>::NG],
                                           const float (&bhh)[Cell<CELL>::NG], float h, float& c,
                                           float (&sv)[6]) {
  if constexpr (CELL == CELL_GRU) {
    // r and z from one reciprocal: with p = 1 + e^-a, q = 1 + e^-b and i = 1/(p q),
    // r = q i and z = p i.  The exponents are clamped at 40 so that p q stays finite.
    const float ar = a[0] + (bi[0] + bhh[0]);
    const float az = a[1] + (bi[1] + bhh[1]);
    // (a NaN pre-activation is dropped by fminf here and carried by the n gate below, whose
    // clamp keeps it: x or h out of range makes a[2] or a[3] NaN along with a[0], a[1])
    const float pr = 1.0f + __expf(fminf(-ar, 40.0f));
    const float pz = 1.0f + __expf(fminf(-az, 40.0f));
    const float inv = rcpf_(pr * pz);
    const float r = pz * inv;
    const float z = pr * inv;
    const float hn = a[3] + bhh[2];
    const float n = tanhf_(a[2] + bi[2] + r * hn);
    sv[0] = r; sv[1] = z; sv[2] = n; sv[3] = hn; sv[4] = h;
    return n + z * (h - n);
  } else if constexpr (CELL == CELL_LSTM) {
    const float ig = sigmoidf_(a[0] + (bi[0] + bhh[0]));
    const float fg = sigmoidf_(a[1] + (bi[1] + bhh[1]));
    const float gg = tanhf_(a[2] + (bi[2] + bhh[2]));
    const float og = sigmoidf_(a[3] + (bi[3] + bhh[3]));
    sv[4] = c;
    c = fg * c + ig * gg;
    const float tc = tanhf_(c);
    sv[0] = ig; sv[1] = fg; sv[2] = gg; sv[3] = og; sv[5] = tc;
    return og * tc;
  } else {
    return tanhf_(a[0] + (bi[0] + bhh[0]));
  }
}

// Epilogue of a full tile whose state rows are contiguous: no per-element bounds test, the
// sixteen old-state values of a 32-column block loaded before the math, every address a
// uniform row base plus one 32-bit lane offset.  SAVE: the gate values go to `saved` for the
// backward pass (GRU r, z, n, hn, h; LSTM i, f, g, o, c, tanh(c')).
template <int CELL, int JSUB, int JW, bool SAVE>
__device__ __forceinline__ void epilogue_full_tile(const SplitArgs& p,
                                                   f32x16 (&acc)[JSUB][Cell<CELL>::NACC],
                                                   int64_t urow0, int jbase, int lr, int lh,
                                                   bool write_mem) {
  using C = Cell<CELL>;
  constexpr int NG = C::NG, NACC = C::NACC;
  const int H = p.H;
  const size_t rowb = static_cast<size_t>(H) * sizeof(float);
#pragma unroll
  for (int s = 0; s < JSUB; ++s) {
    const int j = jbase + s * 32 + lr;
    float bi[NG], bhh[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      bi[g] = p.b_ih ? p.b_ih[g * H + j] : 0.0f;
      bhh[g] = p.b_hh ? p.b_hh[g * H + j] : 0.0f;
    }
    const uint32_t lane_off = static_cast<uint32_t>(4 * lh * H + j) * 4u;  // bytes
    char* hm = reinterpret_cast<char*>(p.h_mem + urow0 * H);
    char* ho = reinterpret_cast<char*>(p.h_out + urow0 * H);
    char* cm = CELL == CELL_LSTM ? reinterpret_cast<char*>(p.c_mem + urow0 * H) : nullptr;
    char* co = CELL == CELL_LSTM ? reinterpret_cast<char*>(p.c_out + urow0 * H) : nullptr;
    float hold[16], cold[CELL == CELL_LSTM ? 16 : 1];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const size_t ro = ((e & 3) + 8 * (e >> 2)) * rowb;  // uniform
      if constexpr (CELL == CELL_GRU) hold[e] = *reinterpret_cast<const float*>(hm + ro + lane_off);
      if constexpr (CELL == CELL_LSTM) cold[e] = *reinterpret_cast<const float*>(cm + ro + lane_off);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const size_t ro = ((e & 3) + 8 * (e >> 2)) * rowb;
      float a[NACC];
#pragma unroll
      for (int q = 0; q < NACC; ++q) a[q] = acc[s][q][e];
      float c = CELL == CELL_LSTM ? cold[e] : 0.0f;
      float sv[6];
      const float hy = gate_math<CELL>(a, bi, bhh, CELL == CELL_GRU ? hold[e] : 0.0f, c, sv);
      if constexpr (SAVE && CELL != CELL_RNN) {
        constexpr int NS = CELL == CELL_GRU ? 5 : 6;
        char* sb = reinterpret_cast<char*>(p.saved + urow0 * (NS * H)) + NS * ro +
                   static_cast<uint32_t>(4 * lh * NS * H + j) * 4u;
#pragma unroll
        for (int q = 0; q < NS; ++q) *reinterpret_cast<float*>(sb + q * rowb) = sv[q];
      }
      *reinterpret_cast<float*>(ho + ro + lane_off) = hy;
      if constexpr (CELL == CELL_LSTM) *reinterpret_cast<float*>(co + ro + lane_off) = c;
      if (write_mem) {
        *reinterpret_cast<float*>(hm + ro + lane_off) = hy;
        if constexpr (CELL == CELL_LSTM) *reinterpret_cast<float*>(cm + ro + lane_off) = c;
      }
    }
  }
}

template <int MODE, int CELL, int JSUB, int WC, bool IDS, bool PIPE = false>
__global__ __launch_bounds__(256 * WC, JSUB == 1 ? 4 : 2) void split_cell_kernel(SplitArgs p) {
  using C = Cell<CELL>;
  constexpr int THREADS = 256 * WC;
  constexpr int NG = C::NG, NACC = C::NACC;
  constexpr int JW = JSUB * 32;               // hidden units per wave
  constexpr int JC = JW * WC;                 // hidden units per pass
  constexpr int NC = NG * JC;                 // B-tile rows (output columns)
  constexpr int ROWB = kBK * 2;               // bytes of one image row (16 bf16)
  constexpr int A_PLANE = kRows * ROWB;       // one piece of the A tile
  constexpr int B_PLANE = NC * ROWB;
  constexpr int A_BYTES = 3 * A_PLANE, B_BYTES = 3 * B_PLANE;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int APT = (kRows * kBK / 4) / THREADS;           // float4 of the A tile per thread
  // 16-byte chunks of the B tile that are READ per K-step: all three planes of the bf16 pieces,
  // planes 0 and 2 of the fp16 form (plane 1 is plane 0 / 64 and is made while staging, which
  // takes a third off the weight stream, the larger part of what a K-step reads)
  constexpr int B16 = (MODE == MODE_F16X3 ? 2 : 3) * (B_PLANE / 16);
  constexpr int BPT = (B16 + THREADS - 1) / THREADS;
  static_assert((kRows * kBK / 4) % THREADS == 0, "A tile must divide over the threads");

  extern __shared__ float4 smem4[];
  char* smem = reinterpret_cast<char*>(smem4);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: scalar registers
  const int rg = wave & 3;
  const int cg = wave >> 2;
  const int lr = lane & 31;
  const int lh = lane >> 5;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * kRows;
  const int H = p.H;
  const int GH = NG * H;

  // A staging: float4 number idx covers row idx / 4, k (idx % 4) * 4 .. +3
  using HOff = typename std::conditional<IDS, int64_t, int>::type;
  const float* xbase = p.x + row0 * p.ldx;
  const float* hbase = IDS ? p.h_mem : p.h_mem + row0 * H;
  int ax[APT];
  HOff ah[APT];
  int a_dst[APT];
#pragma unroll
  for (int i = 0; i < APT; ++i) {
    const int idx = tid + i * THREADS;
    const int c4 = (idx & 3) * 4;
    int trow = idx >> 2;
    a_dst[i] = ((c4 >> 3) * kRows + trow) * 16 + (c4 & 7) * 2;
    if (row0 + trow >= p.M) trow = static_cast<int>(p.M - 1 - row0);  // clamp: computed, not stored
    ax[i] = trow * static_cast<int>(p.ldx) + c4;
    if constexpr (IDS) ah[i] = p.ids[row0 + trow] * H + c4;
    else ah[i] = trow * H + c4;
  }

  const int a_rd = (lh * kRows + rg * 32 + lr) * 16;
  const int b_rd = A_BYTES + (lh * NC + cg * JW + lr) * 16;

  // Row magnitudes of x (fp16 planes only): the four threads that stage a row find its largest
  // |x|; a row that reaches 2^15 gets the power-of-two scale that brings it to [2^13, 2^14) and
  // its inverse is left in LDS for the epilogue.  Every other row keeps scale 1.
  float row_sc[APT];
  float* s_inv = reinterpret_cast<float*>(smem + 2 * STAGE + 1024);  // [kRows], behind the stages
  bool any_scaled = false;
#pragma unroll
  for (int i = 0; i < APT; ++i) row_sc[i] = 1.0f;
  if constexpr (MODE == MODE_F16X3) {
    if (!p.x_bounded) {
      const int n_steps = p.I / kBK;
      bool mine = false;
      float rmax[APT];
#pragma unroll
      for (int i = 0; i < APT; ++i) rmax[i] = 0.0f;
      for (int t0 = 0; t0 < n_steps; t0 += 4) {  // 4 * APT loads in flight
        float4 v[4][APT];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < APT; ++i)
            v[u][i] = t0 + u < n_steps ? *reinterpret_cast<const float4*>(xbase + ax[i] + (t0 + u) * kBK)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < APT; ++i)
            rmax[i] = fmaxf(rmax[i], fmaxf(fmaxf(fabsf(v[u][i].x), fabsf(v[u][i].y)),
                                           fmaxf(fabsf(v[u][i].z), fabsf(v[u][i].w))));
      }
#pragma unroll
      for (int i = 0; i < APT; ++i) {
        float m = rmax[i];
        m = fmaxf(m, __shfl_xor(m, 1));
        m = fmaxf(m, __shfl_xor(m, 2));
        const int ex = static_cast<int>((__float_as_uint(m) >> 23) & 0xffu);  // biased exponent
        const bool big = ex >= 127 + 15 && ex != 255;                         // finite, >= 2^15
        const int sh = big ? ex - (127 + 13) : 0;                             // m * 2^-sh in [2^13, 2^14)
        row_sc[i] = __uint_as_float(static_cast<uint32_t>(127 - sh) << 23);
        const int idx = tid + i * THREADS;
        if ((idx & 3) == 0) s_inv[idx >> 2] = __uint_as_float(static_cast<uint32_t>(127 + sh) << 23);
        mine |= big;
      }
      any_scaled = __syncthreads_or(mine);
    }
  }

  for (int j0 = 0; j0 < H; j0 += JC) {
    // B staging: chunk c -> (plane, k half) c / NC, tile row c % NC.  The weights are stored
    // [K/16][plane][k half][n][8], so a K-step advances every offset by 3*GH*16.
    int boff[BPT];
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int c = tid + i * THREADS;
      int ph, nl;  // plane * 2 + k half, tile row
      if constexpr (MODE == MODE_F16X3) {
        const int q = c / (2 * NC);  // 0: plane 0, 1: plane 2
        ph = 4 * q + (c - q * 2 * NC) / NC;
        nl = c % NC;
      } else {
        ph = c / NC;
        nl = c % NC;
      }
      const int brow = (c < B16) ? ph * GH + (nl / JC) * H + j0 + (nl % JC) : 0;
      boff[i] = brow * 8;
    }
    const int bstep = 3 * GH;  // 16-element rows per K-step

    f32x16 acc[JSUB][NACC];
#pragma unroll
    for (int s = 0; s < JSUB; ++s)
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][a][e] = 0.0f;

    StageRegs<APT, BPT> regs;
#pragma unroll
    for (int i = 0; i < APT; ++i) regs.sc[i] = row_sc[i];

    // Rows of x that were scaled down: their partial sums are scaled back between the x terms
    // and the state terms (a power of two, exact), so the state terms never see the factor.
    auto unscale_rows = [&]() {
      if (any_scaled) {  // rare
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float inv = s_inv[rg * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh];
#pragma unroll
          for (int sb = 0; sb < JSUB; ++sb)
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[sb][q][e] *= inv;
        }
      }
    };

    // The registers hold a tile of the state from here on: staged as it is.
    auto state_tile = [&]() {
#pragma unroll
      for (int i = 0; i < APT; ++i) regs.sc[i] = 1.0f;
    };

    const int tiles_x = p.I / kBK;
    const int tiles_h = H / kBK;
    if constexpr (PIPE) {
      static_assert(MODE == MODE_F16X3, "the pipelined loop is written for the fp16 planes");
      const int total = tiles_x + tiles_h;
      char* dump = smem + 2 * STAGE + (tid & 63) * 16;
      __syncthreads();  // the previous pass has finished reading both buffers
      load_stage<APT, BPT>(regs, xbase, ax, 0, p.ws_ih, boff, 0);
      store_stage<MODE, APT, BPT, B16, THREADS, A_PLANE, A_BYTES>(regs, smem, a_dst, tid);
      if (1 < tiles_x) load_stage<APT, BPT>(regs, xbase, ax, kBK, p.ws_ih, boff, bstep);
      else { load_stage<APT, BPT>(regs, hbase, ah, 0, p.ws_hh, boff, 0); state_tile(); }
      __syncthreads();
      int buf = 0;
      // two loops, one per phase: a single loop that picks the accumulator set by a branch
      // makes the compiler keep a second copy of all accumulators
      for (int s = 0; s < tiles_x; ++s) {
        const char* cur = smem + buf * STAGE;
        piped_step<CELL, JSUB, JC, A_PLANE, B_PLANE, false, APT, BPT, B16, THREADS, A_BYTES>(
            acc, cur + a_rd, cur + b_rd, regs, smem + (buf ^ 1) * STAGE, dump, a_dst, tid);
        const int s2 = s + 2;  // the registers are free again: fetch the step after next
        if (s2 < tiles_x) load_stage<APT, BPT>(regs, xbase, ax, s2 * kBK, p.ws_ih, boff, s2 * bstep);
        else if (s2 < total) {
          load_stage<APT, BPT>(regs, hbase, ah, (s2 - tiles_x) * kBK, p.ws_hh, boff, (s2 - tiles_x) * bstep);
          state_tile();
        }
        __syncthreads();
        buf ^= 1;
      }
      unscale_rows();
      for (int t = 0; t + 1 < tiles_h; ++t) {
        const char* cur = smem + buf * STAGE;
        piped_step<CELL, JSUB, JC, A_PLANE, B_PLANE, true, APT, BPT, B16, THREADS, A_BYTES>(
            acc, cur + a_rd, cur + b_rd, regs, smem + (buf ^ 1) * STAGE, dump, a_dst, tid);
        if (t + 2 < tiles_h) load_stage<APT, BPT>(regs, hbase, ah, (t + 2) * kBK, p.ws_hh, boff, (t + 2) * bstep);
        __syncthreads();
        buf ^= 1;
      }
      mfma_stage<MODE, CELL, JSUB, JC, A_PLANE, B_PLANE, true>(acc, smem + buf * STAGE + a_rd,
                                                               smem + buf * STAGE + b_rd);
    } else {
    __syncthreads();  // the previous pass has finished reading both buffers
    load_stage<APT, BPT>(regs, xbase, ax, 0, p.ws_ih, boff, 0);
    store_stage<MODE, APT, BPT, B16, THREADS, A_PLANE, A_BYTES>(regs, smem, a_dst, tid);
    __syncthreads();
    int buf = 0;
    for (int t = 0; t < tiles_x; ++t) {
      if (t + 1 < tiles_x) load_stage<APT, BPT>(regs, xbase, ax, (t + 1) * kBK, p.ws_ih, boff, (t + 1) * bstep);
      else { load_stage<APT, BPT>(regs, hbase, ah, 0, p.ws_hh, boff, 0); state_tile(); }
      mfma_stage<MODE, CELL, JSUB, JC, A_PLANE, B_PLANE, false>(acc, smem + buf * STAGE + a_rd,
                                                          smem + buf * STAGE + b_rd);
      store_stage<MODE, APT, BPT, B16, THREADS, A_PLANE, A_BYTES>(regs, smem + (buf ^ 1) * STAGE, a_dst, tid);
      __syncthreads();
      buf ^= 1;
    }
    unscale_rows();
    for (int t = 0; t < tiles_h; ++t) {
      const bool more = t + 1 < tiles_h;
      if (more) load_stage<APT, BPT>(regs, hbase, ah, (t + 1) * kBK, p.ws_hh, boff, (t + 1) * bstep);
      mfma_stage<MODE, CELL, JSUB, JC, A_PLANE, B_PLANE, true>(acc, smem + buf * STAGE + a_rd,
                                                         smem + buf * STAGE + b_rd);
      if (more)
        store_stage<MODE, APT, BPT, B16, THREADS, A_PLANE, A_BYTES>(regs, smem + (buf ^ 1) * STAGE, a_dst, tid);
      __syncthreads();
      buf ^= 1;
    }

    }
    // ---- epilogue: gates, state update, stores (same accumulator layout as the fp32 kernel).
    bool fast = false;
    if constexpr (!IDS) fast = row0 + kRows <= p.M;
    if (fast) {
      if (p.saved == nullptr || CELL == CELL_RNN)
        epilogue_full_tile<CELL, JSUB, JW, false>(p, acc, row0 + rg * 32, j0 + cg * JW, lr, lh,
                                                  p.mem_write && JC == H);
      else
        epilogue_full_tile<CELL, JSUB, JW, true>(p, acc, row0 + rg * 32, j0 + cg * JW, lr, lh,
                                                 p.mem_write && JC == H);
      continue;
    }
    // General path (last tile, redirected rows).
    // Row addresses are "uniform row base + lane offset", so that no per-element 64-bit
    // address is kept alive across the K loop.
#pragma unroll
    for (int s = 0; s < JSUB; ++s) {
      const int j = j0 + cg * JW + s * 32 + lr;
      float bi[NG], bhh[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        bi[g] = p.b_ih ? p.b_ih[g * H + j] : 0.0f;
        bhh[g] = p.b_hh ? p.b_hh[g * H + j] : 0.0f;
      }
      const int voff = 4 * lh * H + j;  // lane part of every address below
      const int64_t urow0 = row0 + rg * 32;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t urow = urow0 + ((e & 3) + 8 * (e >> 2));  // uniform part of the row
        const int64_t row = urow + 4 * lh;
        if (row >= p.M) continue;
        float* ho = p.h_out + urow * H;
        float* hm;
        float* cm = nullptr;
        if constexpr (IDS) {
          const int64_t mr = p.ids[row] * H + j;
          hm = p.h_mem + mr - voff;
          if constexpr (CELL == CELL_LSTM) cm = p.c_mem + mr - voff;
        } else {
          hm = p.h_mem + urow * H;
          if constexpr (CELL == CELL_LSTM) cm = p.c_mem + urow * H;
        }
        if constexpr (CELL == CELL_GRU) {
          const float h = hm[voff];
          // r and z from one reciprocal: with p = 1 + e^-a, q = 1 + e^-b and i = 1/(p q),
          // r = q i and z = p i.  The exponents are clamped at 40 so that p q stays finite
          // (sigmoid(-40) = 4e-18 is already 0 next to anything it is added to).
          const float ar = acc[s][0][e] + (bi[0] + bhh[0]);
          const float az = acc[s][1][e] + (bi[1] + bhh[1]);
          const float pr = 1.0f + __expf(ar < -40.0f ? 40.0f : -ar);
          const float pz = 1.0f + __expf(az < -40.0f ? 40.0f : -az);
          const float inv = rcpf_(pr * pz);
          const float r = pz * inv;
          const float z = pr * inv;
          const float hn = acc[s][3][e] + bhh[2];
          const float n = tanhf_(acc[s][2][e] + bi[2] + r * hn);
          const float hy = n + z * (h - n);
          ho[voff] = hy;
          if (p.saved) {
            float* sv = p.saved + urow * 5 * H + (4 * lh * 5 * H + j);
            sv[0] = r; sv[H] = z; sv[2 * H] = n; sv[3 * H] = hn; sv[4 * H] = h;
          }
          if (p.mem_write && JC == H) hm[voff] = hy;
        } else if constexpr (CELL == CELL_LSTM) {
          const float c = cm[voff];
          const float ig = sigmoidf_(acc[s][0][e] + (bi[0] + bhh[0]));
          const float fg = sigmoidf_(acc[s][1][e] + (bi[1] + bhh[1]));
          const float gg = tanhf_(acc[s][2][e] + (bi[2] + bhh[2]));
          const float og = sigmoidf_(acc[s][3][e] + (bi[3] + bhh[3]));
          const float cy = fg * c + ig * gg;
          const float tc = tanhf_(cy);
          const float hy = og * tc;
          ho[voff] = hy;
          (p.c_out + urow * H)[voff] = cy;
          if (p.saved) {
            float* sv = p.saved + urow * 6 * H + (4 * lh * 6 * H + j);
            sv[0] = ig; sv[H] = fg; sv[2 * H] = gg; sv[3 * H] = og; sv[4 * H] = c; sv[5 * H] = tc;
          }
          if (p.mem_write && JC == H) {
            hm[voff] = hy;
            cm[voff] = cy;
          }
        } else {
          const float hy = tanhf_(acc[s][0][e] + (bi[0] + bhh[0]));
          ho[voff] = hy;
          if (p.mem_write && JC == H) hm[voff] = hy;
        }
      }
    }
  }

  // With several column passes the state rows stay readable until the last pass is done.
  if (p.mem_write && JC != H) {
    __syncthreads();
    const int per_row = H / 4;
    for (int idx = tid; idx < kRows * per_row; idx += THREADS) {
      const int64_t row = row0 + idx / per_row;
      if (row >= p.M) break;
      const int c4 = (idx % per_row) * 4;
      const int64_t mrow = (p.ids ? p.ids[row] : row) * H + c4;
      *reinterpret_cast<float4*>(p.h_mem + mrow) =
          *reinterpret_cast<const float4*>(p.h_out + row * H + c4);
      if constexpr (CELL == CELL_LSTM)
        *reinterpret_cast<float4*>(p.c_mem + mrow) =
            *reinterpret_cast<const float4*>(p.c_out + row * H + c4);
    }
  }
}

template <int MODE, int CELL, int JSUB, int WC>
int launch_config(const SplitArgs& a, hipStream_t st) {
  constexpr int NC = Cell<CELL>::NG * JSUB * 32 * WC;
  constexpr bool PIPE = MODE == MODE_F16X3 && JSUB == 2;  // staging between the MFMAs: 3-4 % faster
  constexpr size_t lds = 2 * 3 * (kRows + NC) * (kBK * 2) + (MODE == MODE_F16X3 ? 1024 + 4 * kRows : 0);
  static_assert(lds <= 160 * 1024, "LDS stage does not fit");
  auto plain = split_cell_kernel<MODE, CELL, JSUB, WC, false, PIPE>;
  auto gathered = split_cell_kernel<MODE, CELL, JSUB, WC, true, PIPE>;
  SPMI_RAISE_LDS_LIMIT(plain, lds);
  SPMI_RAISE_LDS_LIMIT(gathered, lds);
  const unsigned grid = static_cast<unsigned>((a.M + kRows - 1) / kRows);
  if (a.ids) gathered<<<grid, 256 * WC, lds, st>>>(a);
  else plain<<<grid, 256 * WC, lds, st>>>(a);
  SPMI_CHECK_LAUNCH("split_cell");
  return SPMI_OK;
}

template <int MODE, int CELL>
int launch_split(const SplitArgs& a, hipStream_t st) {
  if (a.H % 128 == 0) return launch_config<MODE, CELL, 2, 2>(a, st);
  if (a.H % 64 == 0) return launch_config<MODE, CELL, 1, 2>(a, st);
  return launch_config<MODE, CELL, 1, 1>(a, st);
}

template <int MODE>
int split_weights(const float* w, int64_t rows, int64_t cols, int64_t row_stride, int64_t col_stride,
                  void* out, uint32_t* amax_bits, void* stream, const char* name) {
  SPMI_REQUIRE(rows >= 1 && cols >= 16 && cols % 16 == 0 && w && out, SPMI_ERR_INVALID_ARG,
               "%s: needs rows >= 1, cols a multiple of 16 and non-null pointers", name);
  SPMI_REQUIRE(row_stride >= 1 && col_stride >= 1, SPMI_ERR_INVALID_ARG, "%s: strides must be positive",
               name);
  const int64_t n = rows * cols;
  split_weights_kernel<MODE><<<static_cast<unsigned>((n + 255) / 256), 256, 0, spmi::as_stream(stream)>>>(
      w, rows, cols, row_stride, col_stride, static_cast<uint16_t*>(out), amax_bits);
  SPMI_CHECK_LAUNCH(name);
  return SPMI_OK;
}

template <int MODE>
int cell_split_fwd(const char* name, int cell, const float* x, int64_t ldx, int64_t I,
                   const void* ws_ih, const void* ws_hh, const float* b_ih, const float* b_hh,
                   float* h_mem, float* c_mem, const int64_t* ids, int64_t M, int64_t H,
                   float* h_out, float* c_out, int mem_write, float* saved, int x_bounded,
                   void* stream) {
  SPMI_REQUIRE(cell >= CELL_GRU && cell <= CELL_RNN, SPMI_ERR_INVALID_ARG, "%s: bad cell code %d",
               name, cell);
  SPMI_REQUIRE(M >= 0 && M < (int64_t{1} << 31), SPMI_ERR_INVALID_ARG, "%s: bad M=%lld", name,
               (long long)M);
  SPMI_REQUIRE(I >= 32 && I % 32 == 0 && H >= 32 && H % 32 == 0 && I <= 8192 && H <= 8192,
               SPMI_ERR_UNSUPPORTED, "%s: needs I %% 32 == 0 and H %% 32 == 0 (I=%lld H=%lld)", name,
               (long long)I, (long long)H);
  SPMI_REQUIRE(ldx < (int64_t{1} << 20), SPMI_ERR_UNSUPPORTED,
               "%s: leading dimension %lld too large", name, (long long)ldx);
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(x && ws_ih && ws_hh && h_mem && h_out && ldx >= I && ldx % 4 == 0,
               SPMI_ERR_INVALID_ARG, "%s: null pointer or bad ldx", name);
  SPMI_REQUIRE(cell != CELL_LSTM || (c_mem && c_out), SPMI_ERR_INVALID_ARG,
               "%s: LSTM needs c_mem and c_out", name);
  for (const void* q : {(const void*)x, ws_ih, ws_hh, (const void*)h_mem, (const void*)h_out,
                        (const void*)c_mem, (const void*)c_out})
    SPMI_REQUIRE(!q || (reinterpret_cast<uintptr_t>(q) & 15u) == 0, SPMI_ERR_INVALID_ARG,
                 "%s: pointers must be 16-byte aligned", name);
  SplitArgs a{};
  a.x = x; a.ldx = ldx; a.I = static_cast<int>(I);
  a.ws_ih = static_cast<const __bf16*>(ws_ih); a.ws_hh = static_cast<const __bf16*>(ws_hh);
  a.b_ih = b_ih; a.b_hh = b_hh;
  a.h_mem = h_mem; a.c_mem = c_mem; a.ids = ids; a.M = M; a.H = static_cast<int>(H);
  a.h_out = h_out; a.c_out = c_out; a.mem_write = mem_write; a.saved = saved;
  a.x_bounded = x_bounded;
  hipStream_t st = spmi::as_stream(stream);
  if (cell == CELL_GRU) return launch_split<MODE, CELL_GRU>(a, st);
  if (cell == CELL_LSTM) return launch_split<MODE, CELL_LSTM>(a, st);
  return launch_split<MODE, CELL_RNN>(a, st);
}

}  // namespace

extern "C" {

int spmi_split_bf16x3(const float* w, int64_t rows, int64_t cols, void* out, void* stream) {
  return split_weights<MODE_BF16X3>(w, rows, cols, cols, 1, out, nullptr, stream, "spmi_split_bf16x3");
}

int spmi_split_f16x3(const float* w, int64_t rows, int64_t cols, void* out, void* stream) {
  return split_weights<MODE_F16X3>(w, rows, cols, cols, 1, out, nullptr, stream, "spmi_split_f16x3");
}

int spmi_split_f16x3_strided(const float* w, int64_t rows, int64_t cols, int64_t row_stride,
                             int64_t col_stride, void* out, uint32_t* amax_bits, void* stream) {
  return split_weights<MODE_F16X3>(w, rows, cols, row_stride, col_stride, out, amax_bits, stream,
                                   "spmi_split_f16x3_strided");
}

int spmi_cell_split_fwd(int cell, const float* x, int64_t ldx, int64_t I, const void* ws_ih,
                        const void* ws_hh, const float* b_ih, const float* b_hh, float* h_mem,
                        float* c_mem, const int64_t* ids, int64_t M, int64_t H, float* h_out,
                        float* c_out, int mem_write, float* saved, void* stream) {
  return cell_split_fwd<MODE_BF16X3>("spmi_cell_split_fwd", cell, x, ldx, I, ws_ih, ws_hh, b_ih, b_hh,
                                     h_mem, c_mem, ids, M, H, h_out, c_out, mem_write, saved, 1, stream);
}

int spmi_cell_f16x3_fwd(int cell, const float* x, int64_t ldx, int64_t I, const void* ws_ih,
                        const void* ws_hh, const float* b_ih, const float* b_hh, float* h_mem,
                        float* c_mem, const int64_t* ids, int64_t M, int64_t H, float* h_out,
                        float* c_out, int mem_write, float* saved, int x_bounded, void* stream) {
  return cell_split_fwd<MODE_F16X3>("spmi_cell_f16x3_fwd", cell, x, ldx, I, ws_ih, ws_hh, b_ih, b_hh,
                                    h_mem, c_mem, ids, M, H, h_out, c_out, mem_write, saved, x_bounded,
                                    stream);
}

}  // extern "C"
