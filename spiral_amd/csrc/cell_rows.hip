// One GRU (and, further down, tanh RNN and LSTM) layer step over ALL node rows with the gate weights
// resident in LDS: the inference form of torch.nn.GRU as driven by SequentialDecoder.forward
// (spiro/dtdg/models/decoder/torch/sequentialDecoder/implicitTimeModels.py:144-158) when ids
// is every node, so that NodeMemory.get_memory / update_memory (:48-54) read one [M, H] slab
// and write another.
//
// Same arithmetic as the f16x3 mode of cell_split.hip (three fp16 planes per operand, three
// v_mfma_f32_32x32x16_f16 per term, fp32 accumulation, the same gate functions), different
// data movement.  cell_split.hip gives a workgroup 128 rows and ALL hidden units, so every
// workgroup streams the whole weight image (590 KB at I = H = 128) through LDS for 131 KB of
// rows, and the kernel is bound by that staging.  Here a workgroup owns 32 hidden units:
// their r, z, n columns of W_ih and W_hh (6 x 128 x 32 weights x 3 planes x 2 B = 144 KB)
// are loaded into LDS once and stay there, and the workgroup streams row tiles past them:
//  * x and the old state never touch LDS: the A fragment of the MFMA is eight consecutive k
//    of one row per lane, i.e. two float4 loads straight from global memory, split into the
//    three planes in registers (as in gemm_rows.hip);
//  * a wave owns a tile of 32 rows; its 64 row registers hold x of the tile, are refilled
//    K-step by K-step with the old state of the same tile while the x terms are multiplied,
//    and with x of the wave's next tile while the state terms are multiplied: 16 KB per wave
//    are in flight under the matrix work and there is no barrier in the loop;
//  * the H / 32 workgroups that hold the column slices of one row range run on the same XCD
//    (blockIdx -> XCD is round-robin) and walk the same tile sequence, so the rows they all
//    read come from HBM once and from that XCD's L2 afterwards.
// A row tile is read by several workgroups and written slice by slice, so the new state
// cannot overwrite the old one in place: h_out must not overlap h_in (the caller keeps two
// slabs per layer and swaps them).

#include <type_traits>

#include "common.h"

namespace {

using f32x16 = float __attribute__((ext_vector_type(16)));
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));

struct GruRowsArgs {
  const float* x;
  int64_t ldx;
  const uint4* ws_ih;  // [I/16][3][2][3H][8] fp16 (spmi_split_f16x3 of weight_ih [3H, I])
  const uint4* ws_hh;  // [H/16][3][2][3H][8] fp16
  const float* b_ih;
  const float* b_hh;
  const float* h_in;
  float* h_out;
  int64_t M;
  int x_bounded;
  float* saved;  // SAVE kernels: [M][4][H] = r, z, n, W_hn h + b_hn per row, the gate values a backward pass needs
};

__device__ __forceinline__ float rcpf_(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float tanhf_(float v) {
  const float e = __expf(2.0f * (v > 15.0f ? 15.0f : v));  // NaN stays NaN
  return 1.0f - 2.0f * rcpf_(e + 1.0f);
}

// The three planes of eight consecutive k of one row.  p1 is taken from p0 by a packed fp16
// multiply (equal to fp16(v / 64) wherever p0 / 64 is a normal fp16 number; below that, one
// more rounding of at most 2^-25) and the residual v - p0 by a mixed-precision fma that reads
// the fp16 value directly: 24 vector instructions for the eight values.
template <bool SCALED>
__device__ __forceinline__ void split8(const float4& lo, const float4& hi, float m, f16x8& p0,
                                       f16x8& p1, f16x8& p2) {
  float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  if constexpr (SCALED) {
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] *= m;
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) p0[q] = static_cast<_Float16>(v[q]);
  p1 = p0 * static_cast<_Float16>(0x1p-6f);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float d = __builtin_fmaf(static_cast<float>(p0[q]), -1.0f, v[q]);
    p2[q] = static_cast<_Float16>(d * 0x1p6f);
  }
}

constexpr int kGate = 3 * 1024;   // bytes of one gate's planes in one K-step: [plane 3][k half 2][32][8]
constexpr int kStep = 3 * kGate;  // bytes of one K-step: r, z, n

// The nine matrix instructions of one K-step, gate by gate: three plane products per gate,
// smallest terms first.  The weight fragments of a gate are read from LDS while the previous
// gate's instructions run: `bq` arrives holding the r-gate fragments of this step (read during
// the previous step) and leaves holding those of the step at `sb_next`.  NACC picks where the
// n gate goes (x terms and state terms of n are kept apart: the reset gate multiplies only
// the state part).  The fences keep the reads ahead of the instructions that cover them.
template <int NACC>
__device__ __forceinline__ void kstep_mfma(f32x16 (&acc)[4], const f16x8& p0, const f16x8& p1,
                                           const f16x8& p2, const char* sb, const char* sb_next,
                                           f16x8 (&bq)[3]) {
  f16x8 b1[3], b2[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) b1[q] = *reinterpret_cast<const f16x8*>(sb + kGate + q * 1024);
  __builtin_amdgcn_sched_barrier(0);
  acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2, bq[1], acc[0], 0, 0, 0);
  acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, bq[2], acc[0], 0, 0, 0);
  acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, bq[0], acc[0], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < 3; ++q) b2[q] = *reinterpret_cast<const f16x8*>(sb + 2 * kGate + q * 1024);
  __builtin_amdgcn_sched_barrier(0);
  acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2, b1[1], acc[1], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, b1[2], acc[1], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, b1[0], acc[1], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < 3; ++q) bq[q] = *reinterpret_cast<const f16x8*>(sb_next + q * 1024);
  __builtin_amdgcn_sched_barrier(0);
  acc[NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2, b2[1], acc[NACC], 0, 0, 0);
  acc[NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, b2[2], acc[NACC], 0, 0, 0);
  acc[NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, b2[0], acc[NACC], 0, 0, 0);
}

// KT = I / 16 = H / 16 K-steps per operand; H / 32 column slices.
// XB: the caller vouches for |x| < 2^15 (no magnitude pass, no scaling code).
// SAVE: the step is tracked for a backward pass: the gate values of every state element are
// stored as four [H]-wide planes per row (a store covers two 128-byte row segments, like the
// stores of the new state), the layout whose lines the backward kernels read whole.
template <int KT, bool XB, bool SAVE = false>
__global__ __launch_bounds__(512, 1) void gru_rows_kernel(GruRowsArgs p) {
  constexpr int H = KT * 16;
  constexpr int GH = 3 * H;
  constexpr int NS = H / 32;                   // column slices
  constexpr int B16 = 2 * KT * kStep / 16;     // 16-byte chunks of the slice image
  static_assert(KT % 2 == 0, "K-steps come in pairs");
  static_assert(32 % NS == 0, "slices of one row range must fit the workgroups of one XCD");
  extern __shared__ uint4 smem4[];
  const char* smem = reinterpret_cast<const char*>(smem4);

  // Workgroup b runs on XCD b % 8.  The NS slices of a row group sit on one XCD.
  const int b = blockIdx.x;
  const int slice = (b >> 3) % NS;
  const int group = (b & 7) + 8 * (b / (8 * NS));
  const int n_groups = gridDim.x / NS;
  const int j0 = slice * 32;

  const int tid = threadIdx.x;
  // LDS image [K-step: x steps, then state steps][gate][plane][k half][32][8]; the source
  // planes are [K-step][plane][k half][3H][8], so 32 consecutive chunks are contiguous.
  // K order: a lane reads 64 contiguous bytes of its row at a time (half of a 128-byte line,
  // all four loads back to back, so the line is fetched from L2 once), which feed two K-steps:
  // K-step 2u + v, lane half lh, slot i stands for k = 32 u + 16 lh + 8 v + i.  The image is
  // permuted to match: LDS (step 2u + v, half lh) <- source (step 2u + lh, half v).
  for (int c = tid; c < B16; c += 512) {
    const int n = c & 31, lh_ = (c >> 5) & 1, q = (c >> 6) % 3, g = (c / 192) % 3, t = c / 576;
    const uint4* src = t < KT ? p.ws_ih : p.ws_hh;
    const int s_ = t < KT ? t : t - KT;
    const int tt = (s_ & ~1) + lh_, hh = s_ & 1;
    smem4[c] = src[((tt * 3 + q) * 2 + hh) * GH + g * H + j0 + n];
  }
  __syncthreads();

  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31;
  const int lh = lane >> 5;
  const int64_t n_tiles = (p.M + 31) / 32;
  const int64_t stride = static_cast<int64_t>(n_groups) * 8;
  int64_t tile = static_cast<int64_t>(wave) * n_groups + group;
  if (tile >= n_tiles) return;

  const int j = j0 + lr;
  const float b_r = (p.b_ih ? p.b_ih[j] : 0.0f) + (p.b_hh ? p.b_hh[j] : 0.0f);
  const float b_z = (p.b_ih ? p.b_ih[H + j] : 0.0f) + (p.b_hh ? p.b_hh[H + j] : 0.0f);
  const float b_in = p.b_ih ? p.b_ih[2 * H + j] : 0.0f;
  const float b_hn = p.b_hh ? p.b_hh[2 * H + j] : 0.0f;

  // Rows past M are clamped to the last row: computed, never stored.
  auto clamp_row = [&](int64_t t) {
    const int64_t r = t * 32 + lr;
    return r < p.M ? r : p.M - 1;
  };

  // a[t][0..1]: the eight values of K-step t; the four float4 of a[2u], a[2u+1] are 64
  // contiguous bytes
  float4 a[KT][2];
  auto load_pair = [&](const float* src, int u) {
    const float4* q = reinterpret_cast<const float4*>(src + 32 * u);
    a[2 * u][0] = q[0];
    a[2 * u][1] = q[1];
    a[2 * u + 1][0] = q[2];
    a[2 * u + 1][1] = q[3];
  };
  {
    const float* src = p.x + clamp_row(tile) * p.ldx + 16 * lh;
#pragma unroll
    for (int u = 0; u < KT / 2; ++u) load_pair(src, u);
  }
  const char* sb = smem + lh * 512 + lr * 16;
  f16x8 bq[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) bq[q] = *reinterpret_cast<const f16x8*>(sb + q * 1024);
  const int voff = 4 * lh * H + j;

  // hy: the old state of the wave's 32 x 32 output block while a tile is worked on, its new
  // state from the gate math until it is stored.  Accumulator layout: element e of lane l is
  // row (e&3) + 8 (e>>2) + 4 (l>>5), column l&31.  The stores of a tile are issued in the
  // middle of the NEXT tile, so that at the top of the loop only the sixteen row loads of
  // that tile are outstanding, on the first pass and on every later one: the wait counts the
  // compiler derives for the loop head then never cover a store.
  float hy[16];
  int64_t prev_row0 = 0;

  auto store_tile = [&](int64_t row0) {
    float* ob = p.h_out + row0 * H;
    if (row0 + 32 <= p.M) {
#pragma unroll
      for (int e = 0; e < 16; ++e) ob[((e & 3) + 8 * (e >> 2)) * H + voff] = hy[e];
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ur = (e & 3) + 8 * (e >> 2);
        if (row0 + ur + 4 * lh < p.M) ob[ur * H + voff] = hy[e];
      }
    }
  };

  auto body = [&](auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const int64_t next = tile + stride;
    // the last iteration re-reads its own tile: valid memory, nothing is done with it
    const float* xn = p.x + clamp_row(next < n_tiles ? next : tile) * p.ldx + 16 * lh;
    const float* hc = p.h_in + clamp_row(tile) * H + 16 * lh;

    f32x16 acc[4];  // r, z, n (x terms), n (state terms)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[q][e] = 0.0f;

    // The x terms, optimistically unscaled: the K-steps start as their rows arrive, and the
    // largest magnitude of every row is collected on the way (unless the caller vouches for it).
    float m = 0.0f;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      if constexpr (!XB) {
        m = fmaxf(m, fmaxf(fmaxf(fabsf(a[t][0].x), fabsf(a[t][0].y)), fmaxf(fabsf(a[t][0].z), fabsf(a[t][0].w))));
        m = fmaxf(m, fmaxf(fmaxf(fabsf(a[t][1].x), fabsf(a[t][1].y)), fmaxf(fabsf(a[t][1].z), fabsf(a[t][1].w))));
      }
      f16x8 p0, p1, p2;
      split8<false>(a[t][0], a[t][1], 1.0f, p0, p1, p2);
      if (t & 1) load_pair(hc, t >> 1);
      // Keep the K-steps apart: left alone, the scheduler hoists every reload to the top of
      // the tile and then needs a second copy of the rows in registers.
      __builtin_amdgcn_sched_barrier(0);
      kstep_mfma<2>(acc, p0, p1, p2, sb + t * kStep, sb + (t + 1) * kStep, bq);
      __builtin_amdgcn_sched_barrier(0);
    }

    // A row of x whose largest magnitude reaches 2^15 (count-valued features do) must not go
    // through the fp16 planes as it is.  Rare, wave-uniform: the x terms of the whole tile are
    // done again from memory, every such row multiplied by the power of two that brings it to
    // [2^13, 2^14), and the partial sums are scaled back (exactly) before the state terms.
    if constexpr (!XB) {
      m = fmaxf(m, __shfl_xor(m, 32));
      const int ex = static_cast<int>((__float_as_uint(m) >> 23) & 0xffu);  // biased exponent
      const bool big = ex >= 127 + 15 && ex != 255;                         // finite, >= 2^15
      if (__any(big)) {
        const int sh = big ? ex - (127 + 13) : 0;
        const float up = __uint_as_float(static_cast<uint32_t>(127 - sh) << 23);
        const float down = __uint_as_float(static_cast<uint32_t>(127 + sh) << 23);
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[q][e] = 0.0f;
        const float* xr = p.x + clamp_row(tile) * p.ldx + 16 * lh;
        for (int t = 0; t < KT; ++t) {  // not unrolled: latency does not matter here, code size does
          const float4* q4 = reinterpret_cast<const float4*>(xr + 32 * (t >> 1)) + 2 * (t & 1);
          float4 lo = q4[0], hi = q4[1];
          f16x8 p0, p1, p2;
          split8<true>(lo, hi, up, p0, p1, p2);
          const char* sf = sb + t * kStep;
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            const f16x8 w0 = *reinterpret_cast<const f16x8*>(sf + g * kGate);
            const f16x8 w1 = *reinterpret_cast<const f16x8*>(sf + g * kGate + 1024);
            const f16x8 w2 = *reinterpret_cast<const f16x8*>(sf + g * kGate + 2048);
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2, w1, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, w2, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, w0, acc[g], 0, 0, 0);
          }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ur = (e & 3) + 8 * (e >> 2);
          // the inverse scale of row ur + 4 lh sits in lane ur + 4 lh (either half)
          const float d0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(down), ur));
          const float d1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(down), ur + 4));
          const float dn = lh ? d1 : d0;
          acc[0][e] *= dn;
          acc[1][e] *= dn;
          acc[2][e] *= dn;
        }
      }
    }

    const int64_t row0 = tile * 32;
    if constexpr (!FIRST) store_tile(prev_row0);
    __builtin_amdgcn_sched_barrier(0);
    {
      // rows past M read the last row (valid memory; their results are never stored)
      const float* hb = p.h_in + row0 * H;
      const int last = static_cast<int>(p.M - 1 - row0);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ur = (e & 3) + 8 * (e >> 2) + 4 * lh;
        hy[e] = hb[(ur < last ? ur : last) * H + j];
      }
    }
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int t = 0; t < KT; ++t) {
      f16x8 p0, p1, p2;
      split8<false>(a[t][0], a[t][1], 1.0f, p0, p1, p2);
      if (t & 1) load_pair(xn, t >> 1);
      __builtin_amdgcn_sched_barrier(0);
      kstep_mfma<3>(acc, p0, p1, p2, sb + (KT + t) * kStep, sb + (t + 1 < KT ? KT + t + 1 : 0) * kStep, bq);
      __builtin_amdgcn_sched_barrier(0);
    }

    float* sv = nullptr;
    bool sv_whole = true;
    if constexpr (SAVE) {
      sv = p.saved + row0 * (4 * H) + (4 * lh * 4 * H + j);  // element e: + ((e & 3) + 8 (e >> 2)) * 4 H
      sv_whole = row0 + 32 <= p.M;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      // r and z from one reciprocal: with p = 1 + e^-a, q = 1 + e^-b and i = 1/(p q),
      // r = q i and z = p i.  The exponents are clamped at 40 so that p q stays finite (a NaN
      // pre-activation is dropped by fminf here and carried by the n gate, whose clamp keeps it).
      const float ar = acc[0][e] + b_r;
      const float az = acc[1][e] + b_z;
      const float pr = 1.0f + __expf(fminf(-ar, 40.0f));
      const float pz = 1.0f + __expf(fminf(-az, 40.0f));
      const float inv = rcpf_(pr * pz);
      const float r = pz * inv;
      const float z = pr * inv;
      const float hn = acc[3][e] + b_hn;
      const float n = tanhf_(acc[2][e] + b_in + r * hn);
      hy[e] = n + z * (hy[e] - n);
      if constexpr (SAVE) {
        const int ur = (e & 3) + 8 * (e >> 2);
        if (sv_whole || row0 + ur + 4 * lh < p.M) {
          float* o = sv + ur * (4 * H);
          o[0] = r;
          o[H] = z;
          o[2 * H] = n;
          o[3 * H] = hn;
        }
      }
    }
    prev_row0 = row0;
    tile = next;
  };

  body(std::true_type{});
  while (tile < n_tiles) body(std::false_type{});
  store_tile(prev_row0);
}

// ---- the LSTM layer step on the same loop nest ---------------------------------------------
// torch.nn.LSTM with seq_len 1 as driven by LSTM.forward of the reference
// (implicitTimeModels.py:214-230) over every node: gates i, f, g, o from x W_ih^T + h W_hh^T,
// c' = f c + i g, h' = o tanh(c').  Four gates of a 32-unit slice in three planes would take
// 192 KB, so the operands travel as TWO planes, p0 = fp16(v) and the residual
// p2 = fp16((v - p0) * 2^6): the main product p0(x) p0(w) goes to one accumulator set, the two
// small products p2(x) p0(w) + p0(x) p2(w), both carrying the factor 2^6 that keeps the
// residuals out of fp16's subnormal range, to a second set that is divided by 2^6 once at the
// end.  Same three matrix instructions per term and the same sums as the three-plane form,
// 128 KB of weights in LDS, eight accumulators per wave.  Rows of x of any magnitude: the
// whole row of x is in registers when a tile starts, so a row whose largest magnitude reaches
// 2^15 is split multiplied by the power of two that brings it to [2^13, 2^14) and its
// accumulators are scaled back (exactly) before the first state term.  (A form without that
// pass over the row, for callers that vouch for |x| < 2^15, measured slower, 1.22 against
// 1.10 ms per layer at 1M rows: its K-steps wait for their rows one by one.  Not kept.)
constexpr int kGate2 = 2 * 1024;    // one gate's planes in one K-step: [p0, residual][k half 2][32][8]
constexpr int kStep4 = 4 * kGate2;  // i, f, g, o

struct LstmRowsArgs {
  const float* x;
  int64_t ldx;
  const uint4* ws_ih;  // [I/16][3][2][4H][8] fp16 (spmi_split_f16x3 of weight_ih [4H, I])
  const uint4* ws_hh;  // [H/16][3][2][4H][8] fp16
  const float* b_ih;
  const float* b_hh;
  const float* h_in;
  const float* c_in;
  float* h_out;
  float* c_out;
  float* saved;  // nullable: [M][6][H] = i, f, g, o, old cell state, tanh(new cell state)
  int64_t M;
  int x_bounded;
};

__device__ __forceinline__ float sigmoidf_(float v) { return rcpf_(1.0f + __expf(-v)); }

// The twelve matrix instructions of one K-step, gate by gate, the fragments of the next gate
// read from LDS while the current gate's instructions run (`bq`: gate i of this step on entry,
// gate i of the step at `sb_next` on exit).
__device__ __forceinline__ void kstep_lstm(f32x16 (&main)[4], f32x16 (&small)[4], const f16x8& p0,
                                           const f16x8& p2, const char* sb, const char* sb_next,
                                           f16x8 (&bq)[2]) {
  f16x8 bn[2];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const char* nx = g < 3 ? sb + (g + 1) * kGate2 : sb_next;
#pragma unroll
    for (int q = 0; q < 2; ++q) bn[q] = *reinterpret_cast<const f16x8*>(nx + q * 1024);
    __builtin_amdgcn_sched_barrier(0);
    small[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2, bq[0], small[g], 0, 0, 0);
    small[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, bq[1], small[g], 0, 0, 0);
    main[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, bq[0], main[g], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    bq[0] = bn[0];
    bq[1] = bn[1];
  }
}

// One wave per SIMD: eight accumulators, the row registers and the old cell state of a tile do
// not fit the 256 registers a wave gets when two share a SIMD.
constexpr int kLstmWaves = 4;

// RNN: the four column tiles are the four 32-unit blocks of ONE gate (the tanh RNN, H = 128):
// the workgroup holds every column, rows are read once.
// SAVE: the gate values a backward pass needs go to p.saved, in the layout lstm_fwd_kernel
// (gates.hip) writes, which is the one spmi_lstm_gate_bwd reads.
template <int KT, bool RNN, bool SAVE = false>
__global__ __launch_bounds__(kLstmWaves * 64, 1) void lstm_rows_kernel(LstmRowsArgs p) {
  constexpr int H = KT * 16;
  constexpr int GH = RNN ? H : 4 * H;
  constexpr int NS = RNN ? 1 : H / 32;
  static_assert(!RNN || H == 128, "four column tiles");
  static_assert(!(RNN && SAVE), "the tanh RNN's backward needs its output only");
  constexpr int B16 = 2 * KT * kStep4 / 16;
  static_assert(KT % 2 == 0, "K-steps come in pairs");
  static_assert(32 % NS == 0, "slices of one row range must fit the workgroups of one XCD");
  extern __shared__ uint4 smem4[];
  const char* smem = reinterpret_cast<const char*>(smem4);

  const int b = blockIdx.x;
  const int slice = (b >> 3) % NS;
  const int group = (b & 7) + 8 * (b / (8 * NS));
  const int n_groups = gridDim.x / NS;
  const int j0 = slice * 32;

  const int tid = threadIdx.x;
  // LDS image [K-step: x steps, then state steps][gate][p0, residual][k half][32][8] from planes 0
  // and 2 of the source's three, in the K order of gru_rows_kernel (64 contiguous bytes of a
  // row per lane and K-step pair).
  for (int c = tid; c < B16; c += kLstmWaves * 64) {
    const int n = c & 31, lh_ = (c >> 5) & 1, q = (c >> 6) & 1, g = (c >> 7) & 3, t = c >> 9;
    const uint4* src = t < KT ? p.ws_ih : p.ws_hh;
    const int s_ = t < KT ? t : t - KT;
    const int tt = (s_ & ~1) + lh_, hh = s_ & 1;
    smem4[c] = src[((tt * 3 + 2 * q) * 2 + hh) * GH + (RNN ? g * 32 : g * H + j0) + n];
  }
  __syncthreads();

  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31;
  const int lh = lane >> 5;
  const int64_t n_tiles = (p.M + 31) / 32;
  const int64_t stride = static_cast<int64_t>(n_groups) * kLstmWaves;
  int64_t tile = static_cast<int64_t>(wave) * n_groups + group;
  if (tile >= n_tiles) return;

  const int j = j0 + lr;
  float bias[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int col = RNN ? g * 32 + lr : g * H + j;
    bias[g] = (p.b_ih ? p.b_ih[col] : 0.0f) + (p.b_hh ? p.b_hh[col] : 0.0f);
  }

  // Rows past M are clamped to the last row: computed, never stored.
  auto clamp_row = [&](int64_t t) {
    const int64_t r = t * 32 + lr;
    return r < p.M ? r : p.M - 1;
  };

  float4 a[KT][2];
  auto load_pair = [&](const float* src, int u) {
    const float4* q = reinterpret_cast<const float4*>(src + 32 * u);
    a[2 * u][0] = q[0];
    a[2 * u][1] = q[1];
    a[2 * u + 1][0] = q[2];
    a[2 * u + 1][1] = q[3];
  };
  {
    const float* src = p.x + clamp_row(tile) * p.ldx + 16 * lh;
#pragma unroll
    for (int u = 0; u < KT / 2; ++u) load_pair(src, u);
  }
  const char* sb = smem + lh * 512 + lr * 16;
  f16x8 bq[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) bq[q] = *reinterpret_cast<const f16x8*>(sb + q * 1024);
  const int voff = 4 * lh * H + j;

  while (tile < n_tiles) {
    const int64_t next = tile + stride;
    // the last iteration re-reads its own tile: valid memory, nothing is done with it
    const float* xn = p.x + clamp_row(next < n_tiles ? next : tile) * p.ldx + 16 * lh;
    const float* hc = p.h_in + clamp_row(tile) * H + 16 * lh;
    const int64_t row0 = tile * 32;

    f32x16 acc[4], sml[4];  // i, f, g, o: main products, small products * 2^6
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[q][e] = sml[q][e] = 0.0f;

    float up, down;
    bool any_big;
    {
      float m = 0.0f;
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        m = fmaxf(m, fmaxf(fmaxf(fabsf(a[t][0].x), fabsf(a[t][0].y)), fmaxf(fabsf(a[t][0].z), fabsf(a[t][0].w))));
        m = fmaxf(m, fmaxf(fmaxf(fabsf(a[t][1].x), fabsf(a[t][1].y)), fmaxf(fabsf(a[t][1].z), fabsf(a[t][1].w))));
      }
      m = fmaxf(m, __shfl_xor(m, 32));
      const int ex = static_cast<int>((__float_as_uint(m) >> 23) & 0xffu);  // biased exponent
      const bool big = ex >= 127 + 15 && ex != 255;                         // finite, >= 2^15
      const int sh = big ? ex - (127 + 13) : 0;
      up = __uint_as_float(static_cast<uint32_t>(127 - sh) << 23);
      down = __uint_as_float(static_cast<uint32_t>(127 + sh) << 23);
      any_big = __any(big);
    }

#pragma unroll
    for (int t = 0; t < KT; ++t) {
      f16x8 p0, p1, p2;
      split8<true>(a[t][0], a[t][1], up, p0, p1, p2);
      if (t & 1) load_pair(hc, t >> 1);
      __builtin_amdgcn_sched_barrier(0);
      kstep_lstm(acc, sml, p0, p2, sb + t * kStep4, sb + (t + 1) * kStep4, bq);
      __builtin_amdgcn_sched_barrier(0);
    }

    if (any_big) {  // rare, wave-uniform
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ur = (e & 3) + 8 * (e >> 2);
        // the inverse scale of row ur + 4 lh sits in lane ur + 4 lh (either half)
        const float d0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(down), ur));
        const float d1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(down), ur + 4));
        const float dn = lh ? d1 : d0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          acc[g][e] *= dn;
          sml[g][e] *= dn;
        }
      }
    }

    // the old cell state of the wave's 32 x 32 output block in accumulator order (element e of
    // lane l: row (e&3) + 8 (e>>2) + 4 (l>>5), column l&31); rows past M read the last row
    float cy[16];
    if constexpr (!RNN) {
      const float* cb = p.c_in + row0 * H;
      const int last = static_cast<int>(p.M - 1 - row0);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ur = (e & 3) + 8 * (e >> 2) + 4 * lh;
        cy[e] = cb[(ur < last ? ur : last) * H + j];
      }
    }
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int t = 0; t < KT; ++t) {
      f16x8 p0, p1, p2;
      split8<false>(a[t][0], a[t][1], 1.0f, p0, p1, p2);
      if (t & 1) load_pair(xn, t >> 1);
      __builtin_amdgcn_sched_barrier(0);
      kstep_lstm(acc, sml, p0, p2, sb + (KT + t) * kStep4, sb + (t + 1 < KT ? KT + t + 1 : 0) * kStep4, bq);
      __builtin_amdgcn_sched_barrier(0);
    }

    float* hb = p.h_out + row0 * H;
    const bool whole = row0 + 32 <= p.M;
    if constexpr (RNN) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ur = (e & 3) + 8 * (e >> 2);
        if (whole || row0 + ur + 4 * lh < p.M) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            hb[ur * H + 4 * lh * H + g * 32 + lr] = tanhf_(__builtin_fmaf(sml[g][e], 0x1p-6f, acc[g][e]) + bias[g]);
        }
      }
    } else {
      float* cb = p.c_out + row0 * H;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float gi = sigmoidf_(__builtin_fmaf(sml[0][e], 0x1p-6f, acc[0][e]) + bias[0]);
        const float gf = sigmoidf_(__builtin_fmaf(sml[1][e], 0x1p-6f, acc[1][e]) + bias[1]);
        const float gg = tanhf_(__builtin_fmaf(sml[2][e], 0x1p-6f, acc[2][e]) + bias[2]);
        const float go = sigmoidf_(__builtin_fmaf(sml[3][e], 0x1p-6f, acc[3][e]) + bias[3]);
        const float c = gf * cy[e] + gi * gg;
        const float tc = tanhf_(c);
        const float h = go * tc;
        const int ur = (e & 3) + 8 * (e >> 2);
        if (whole || row0 + ur + 4 * lh < p.M) {
          cb[ur * H + voff] = c;
          hb[ur * H + voff] = h;
          if constexpr (SAVE) {
            float* sv = p.saved + (row0 + ur + 4 * lh) * (6 * H) + j;
            sv[0] = gi;
            sv[H] = gf;
            sv[2 * H] = gg;
            sv[3 * H] = go;
            sv[4 * H] = cy[e];
            sv[5 * H] = tc;
          }
        }
      }
    }
    tile = next;
  }
}

template <int KT>
int launch_lstm(const LstmRowsArgs& a, hipStream_t st) {
  constexpr size_t lds = static_cast<size_t>(2 * KT) * kStep4;
  static_assert(lds <= 160 * 1024, "slice image does not fit LDS");
  auto kern = a.saved ? lstm_rows_kernel<KT, false, true> : lstm_rows_kernel<KT, false>;
  SPMI_RAISE_LDS_LIMIT((lstm_rows_kernel<KT, false>), lds);  // each use keeps its own record
  SPMI_RAISE_LDS_LIMIT((lstm_rows_kernel<KT, false, true>), lds);
  constexpr int NS = KT / 2;
  const int64_t n_tiles = (a.M + 31) / 32;
  int64_t sets = (n_tiles + 8 * kLstmWaves - 1) / (8 * kLstmWaves);  // a set: 8 row groups of kLstmWaves tiles
  const int64_t max_sets = 256 / (8 * NS);
  if (sets > max_sets) sets = max_sets;
  kern<<<static_cast<unsigned>(sets * 8 * NS), kLstmWaves * 64, lds, st>>>(a);
  SPMI_CHECK_LAUNCH("lstm_rows");
  return SPMI_OK;
}

// ---- the tanh RNN layer step --------------------------------------------------------------
// torch.nn.RNN (tanh) with seq_len 1 as driven by SequentialDecoder.forward
// (implicitTimeModels.py:144-158) over every node: h' = tanh(x W_ih^T + b_ih + h W_hh^T + b_hh).
// One gate: in two planes all 128 columns fit LDS (128 KB), so the step runs on
// lstm_rows_kernel<KT, true> with the four 32-unit blocks of the gate in place of four gates:
// every row is read once and split once (0.33 ms per layer at 1M rows; a three-plane form with
// 64 columns per workgroup and two waves per SIMD, which reads the rows twice, took 0.38).
template <int KT>
int launch_rnn(const GruRowsArgs& a, hipStream_t st) {
  constexpr size_t lds = static_cast<size_t>(2 * KT) * kStep4;
  static_assert(lds <= 160 * 1024, "weight image does not fit LDS");
  LstmRowsArgs l{};
  l.x = a.x; l.ldx = a.ldx; l.ws_ih = a.ws_ih; l.ws_hh = a.ws_hh; l.b_ih = a.b_ih; l.b_hh = a.b_hh;
  l.h_in = a.h_in; l.h_out = a.h_out; l.M = a.M;
  auto kern = lstm_rows_kernel<KT, true>;
  SPMI_RAISE_LDS_LIMIT(kern, lds);
  const int64_t n_tiles = (a.M + 31) / 32;
  int64_t wgs = (n_tiles + kLstmWaves - 1) / kLstmWaves;  // kLstmWaves tiles in flight per workgroup
  wgs = (wgs + 7) / 8 * 8;                                // whole sets of 8 XCDs
  if (wgs > 256) wgs = 256;
  kern<<<static_cast<unsigned>(wgs), kLstmWaves * 64, lds, st>>>(l);
  SPMI_CHECK_LAUNCH("rnn_rows");
  return SPMI_OK;
}

template <int KT>
int launch(const GruRowsArgs& a, hipStream_t st) {
  constexpr size_t lds = static_cast<size_t>(2 * KT) * kStep;
  static_assert(lds <= 160 * 1024, "weight slice does not fit LDS");
  auto kern = a.saved ? (a.x_bounded ? gru_rows_kernel<KT, true, true> : gru_rows_kernel<KT, false, true>)
                      : (a.x_bounded ? gru_rows_kernel<KT, true> : gru_rows_kernel<KT, false>);
  SPMI_RAISE_LDS_LIMIT((gru_rows_kernel<KT, true>), lds);  // each use keeps its own record
  SPMI_RAISE_LDS_LIMIT((gru_rows_kernel<KT, false>), lds);
  SPMI_RAISE_LDS_LIMIT((gru_rows_kernel<KT, true, true>), lds);
  SPMI_RAISE_LDS_LIMIT((gru_rows_kernel<KT, false, true>), lds);
  // One workgroup per CU (the slice image takes most of its LDS); a small batch gets as many
  // row groups as it has tiles for, in whole sets of 8 XCDs x NS slices.
  constexpr int NS = KT / 2;
  const int64_t n_tiles = (a.M + 31) / 32;
  int64_t sets = (n_tiles + 63) / 64;  // a row group's workgroup has 8 waves, a set 8 groups
  const int64_t max_sets = 256 / (8 * NS);
  if (sets > max_sets) sets = max_sets;
  kern<<<static_cast<unsigned>(sets * 8 * NS), 512, lds, st>>>(a);
  SPMI_CHECK_LAUNCH("gru_rows");
  return SPMI_OK;
}

}  // namespace

extern "C" {

int spmi_gru_rows_supported(int64_t I, int64_t H) { return I == 128 && H == 128; }

static int gru_rows_fwd(const float* x, int64_t ldx, int64_t I, const void* ws_ih, const void* ws_hh,
                        const float* b_ih, const float* b_hh, const float* h_in, int64_t M, int64_t H,
                        float* h_out, float* saved, int x_bounded, void* stream) {
  SPMI_REQUIRE(spmi_gru_rows_supported(I, H), SPMI_ERR_UNSUPPORTED,
               "spmi_gru_rows_fwd: needs I = H = 128 (I=%lld H=%lld)", (long long)I, (long long)H);
  SPMI_REQUIRE(M >= 0 && M < (int64_t{1} << 31), SPMI_ERR_INVALID_ARG, "spmi_gru_rows_fwd: bad M=%lld",
               (long long)M);
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(x && ws_ih && ws_hh && h_in && h_out && ldx >= I && ldx % 4 == 0, SPMI_ERR_INVALID_ARG,
               "spmi_gru_rows_fwd: null pointer or bad ldx");
  SPMI_REQUIRE(ldx < (int64_t{1} << 24), SPMI_ERR_UNSUPPORTED,
               "spmi_gru_rows_fwd: leading dimension %lld too large", (long long)ldx);
  for (const void* q : {(const void*)x, ws_ih, ws_hh, (const void*)h_in, (const void*)h_out})
    SPMI_REQUIRE((reinterpret_cast<uintptr_t>(q) & 15u) == 0, SPMI_ERR_INVALID_ARG,
                 "spmi_gru_rows_fwd: pointers must be 16-byte aligned");
  {
    // every row is read by several workgroups and written in column slices: in place would
    // hand some of them the new state
    const uintptr_t i0 = reinterpret_cast<uintptr_t>(h_in), o0 = reinterpret_cast<uintptr_t>(h_out);
    const uintptr_t bytes = static_cast<uintptr_t>(M) * H * sizeof(float);
    SPMI_REQUIRE(i0 + bytes <= o0 || o0 + bytes <= i0, SPMI_ERR_INVALID_ARG,
                 "spmi_gru_rows_fwd: h_out overlaps h_in");
    const uintptr_t x0 = reinterpret_cast<uintptr_t>(x);
    const uintptr_t xbytes = (static_cast<uintptr_t>(M - 1) * ldx + I) * sizeof(float);
    SPMI_REQUIRE(x0 + xbytes <= o0 || o0 + bytes <= x0, SPMI_ERR_INVALID_ARG,
                 "spmi_gru_rows_fwd: h_out overlaps x");
    if (saved) {
      const uintptr_t s0 = reinterpret_cast<uintptr_t>(saved), sbytes = 4 * bytes;
      SPMI_REQUIRE((s0 & 15u) == 0, SPMI_ERR_INVALID_ARG, "spmi_gru_rows_fwd_saved: saved must be 16-byte aligned");
      SPMI_REQUIRE((s0 + sbytes <= o0 || o0 + bytes <= s0) && (s0 + sbytes <= i0 || i0 + bytes <= s0) &&
                       (s0 + sbytes <= x0 || x0 + xbytes <= s0),
                   SPMI_ERR_INVALID_ARG, "spmi_gru_rows_fwd_saved: saved overlaps x, h_in or h_out");
    }
  }
  GruRowsArgs a{};
  a.x = x; a.ldx = ldx;
  a.ws_ih = static_cast<const uint4*>(ws_ih); a.ws_hh = static_cast<const uint4*>(ws_hh);
  a.b_ih = b_ih; a.b_hh = b_hh; a.h_in = h_in; a.h_out = h_out; a.M = M; a.x_bounded = x_bounded;
  a.saved = saved;
  return launch<8>(a, spmi::as_stream(stream));
}

int spmi_gru_rows_fwd(const float* x, int64_t ldx, int64_t I, const void* ws_ih, const void* ws_hh,
                      const float* b_ih, const float* b_hh, const float* h_in, int64_t M, int64_t H,
                      float* h_out, int x_bounded, void* stream) {
  return gru_rows_fwd(x, ldx, I, ws_ih, ws_hh, b_ih, b_hh, h_in, M, H, h_out, nullptr, x_bounded, stream);
}

int spmi_gru_rows_fwd_saved(const float* x, int64_t ldx, int64_t I, const void* ws_ih, const void* ws_hh,
                            const float* b_ih, const float* b_hh, const float* h_in, int64_t M, int64_t H,
                            float* h_out, float* saved, int x_bounded, void* stream) {
  SPMI_REQUIRE(saved || M == 0, SPMI_ERR_INVALID_ARG, "spmi_gru_rows_fwd_saved: saved is null");
  return gru_rows_fwd(x, ldx, I, ws_ih, ws_hh, b_ih, b_hh, h_in, M, H, h_out, saved, x_bounded, stream);
}

int spmi_rnn_rows_supported(int64_t I, int64_t H) { return I == 128 && H == 128; }

int spmi_rnn_rows_fwd(const float* x, int64_t ldx, int64_t I, const void* ws_ih, const void* ws_hh,
                      const float* b_ih, const float* b_hh, const float* h_in, int64_t M, int64_t H,
                      float* h_out, int x_bounded, void* stream) {
  SPMI_REQUIRE(spmi_rnn_rows_supported(I, H), SPMI_ERR_UNSUPPORTED,
               "spmi_rnn_rows_fwd: needs I = H = 128 (I=%lld H=%lld)", (long long)I, (long long)H);
  SPMI_REQUIRE(M >= 0 && M < (int64_t{1} << 31), SPMI_ERR_INVALID_ARG, "spmi_rnn_rows_fwd: bad M=%lld",
               (long long)M);
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(x && ws_ih && ws_hh && h_in && h_out && ldx >= I && ldx % 4 == 0, SPMI_ERR_INVALID_ARG,
               "spmi_rnn_rows_fwd: null pointer or bad ldx");
  SPMI_REQUIRE(ldx < (int64_t{1} << 24), SPMI_ERR_UNSUPPORTED,
               "spmi_rnn_rows_fwd: leading dimension %lld too large", (long long)ldx);
  for (const void* q : {(const void*)x, ws_ih, ws_hh, (const void*)h_in, (const void*)h_out})
    SPMI_REQUIRE((reinterpret_cast<uintptr_t>(q) & 15u) == 0, SPMI_ERR_INVALID_ARG,
                 "spmi_rnn_rows_fwd: pointers must be 16-byte aligned");
  {
    const uintptr_t i0 = reinterpret_cast<uintptr_t>(h_in), o0 = reinterpret_cast<uintptr_t>(h_out);
    const uintptr_t bytes = static_cast<uintptr_t>(M) * H * sizeof(float);
    SPMI_REQUIRE(i0 + bytes <= o0 || o0 + bytes <= i0, SPMI_ERR_INVALID_ARG,
                 "spmi_rnn_rows_fwd: h_out overlaps h_in");
    const uintptr_t x0 = reinterpret_cast<uintptr_t>(x);
    const uintptr_t xbytes = (static_cast<uintptr_t>(M - 1) * ldx + I) * sizeof(float);
    SPMI_REQUIRE(x0 + xbytes <= o0 || o0 + bytes <= x0, SPMI_ERR_INVALID_ARG,
                 "spmi_rnn_rows_fwd: h_out overlaps x");
  }
  GruRowsArgs a{};
  a.x = x; a.ldx = ldx;
  a.ws_ih = static_cast<const uint4*>(ws_ih); a.ws_hh = static_cast<const uint4*>(ws_hh);
  a.b_ih = b_ih; a.b_hh = b_hh; a.h_in = h_in; a.h_out = h_out; a.M = M; a.x_bounded = x_bounded;
  return launch_rnn<8>(a, spmi::as_stream(stream));
}

int spmi_lstm_rows_supported(int64_t I, int64_t H) { return I == 128 && H == 128; }

static int lstm_rows_fwd(const float* x, int64_t ldx, int64_t I, const void* ws_ih, const void* ws_hh,
                         const float* b_ih, const float* b_hh, const float* h_in, const float* c_in,
                         int64_t M, int64_t H, float* h_out, float* c_out, float* saved, int x_bounded,
                         void* stream) {
  SPMI_REQUIRE(spmi_lstm_rows_supported(I, H), SPMI_ERR_UNSUPPORTED,
               "spmi_lstm_rows_fwd: needs I = H = 128 (I=%lld H=%lld)", (long long)I, (long long)H);
  SPMI_REQUIRE(M >= 0 && M < (int64_t{1} << 31), SPMI_ERR_INVALID_ARG, "spmi_lstm_rows_fwd: bad M=%lld",
               (long long)M);
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(x && ws_ih && ws_hh && h_in && c_in && h_out && c_out && ldx >= I && ldx % 4 == 0,
               SPMI_ERR_INVALID_ARG, "spmi_lstm_rows_fwd: null pointer or bad ldx");
  SPMI_REQUIRE(ldx < (int64_t{1} << 24), SPMI_ERR_UNSUPPORTED,
               "spmi_lstm_rows_fwd: leading dimension %lld too large", (long long)ldx);
  for (const void* q : {(const void*)x, ws_ih, ws_hh, (const void*)h_in, (const void*)c_in,
                        (const void*)h_out, (const void*)c_out})
    SPMI_REQUIRE((reinterpret_cast<uintptr_t>(q) & 15u) == 0, SPMI_ERR_INVALID_ARG,
                 "spmi_lstm_rows_fwd: pointers must be 16-byte aligned");
  {
    // h and x rows are read by several workgroups and written in column slices: the new hidden
    // state must not land on either.  The cell state is read and written by the same lane,
    // read first: c_out may be c_in.
    const uintptr_t bytes = static_cast<uintptr_t>(M) * H * sizeof(float);
    const uintptr_t xbytes = (static_cast<uintptr_t>(M - 1) * ldx + I) * sizeof(float);
    auto apart = [](uintptr_t a0, uintptr_t an, uintptr_t b0, uintptr_t bn) { return a0 + an <= b0 || b0 + bn <= a0; };
    const uintptr_t ho = reinterpret_cast<uintptr_t>(h_out), co = reinterpret_cast<uintptr_t>(c_out);
    const uintptr_t hi = reinterpret_cast<uintptr_t>(h_in), ci = reinterpret_cast<uintptr_t>(c_in);
    const uintptr_t x0 = reinterpret_cast<uintptr_t>(x);
    SPMI_REQUIRE(apart(ho, bytes, hi, bytes) && apart(ho, bytes, x0, xbytes) && apart(ho, bytes, ci, bytes),
                 SPMI_ERR_INVALID_ARG, "spmi_lstm_rows_fwd: h_out overlaps an input");
    SPMI_REQUIRE(apart(co, bytes, hi, bytes) && apart(co, bytes, x0, xbytes) && apart(co, bytes, ho, bytes) &&
                     (co == ci || apart(co, bytes, ci, bytes)),
                 SPMI_ERR_INVALID_ARG, "spmi_lstm_rows_fwd: c_out overlaps an input (other than being c_in)");
    if (saved) {
      const uintptr_t s0 = reinterpret_cast<uintptr_t>(saved), sbytes = 6 * bytes;
      SPMI_REQUIRE((s0 & 15u) == 0, SPMI_ERR_INVALID_ARG, "spmi_lstm_rows_fwd_saved: saved must be 16-byte aligned");
      SPMI_REQUIRE(apart(s0, sbytes, hi, bytes) && apart(s0, sbytes, ci, bytes) && apart(s0, sbytes, ho, bytes) &&
                       apart(s0, sbytes, co, bytes) && apart(s0, sbytes, x0, xbytes),
                   SPMI_ERR_INVALID_ARG, "spmi_lstm_rows_fwd_saved: saved overlaps x or a state matrix");
    }
  }
  LstmRowsArgs a{};
  a.x = x; a.ldx = ldx;
  a.ws_ih = static_cast<const uint4*>(ws_ih); a.ws_hh = static_cast<const uint4*>(ws_hh);
  a.b_ih = b_ih; a.b_hh = b_hh; a.h_in = h_in; a.c_in = c_in; a.h_out = h_out; a.c_out = c_out;
  a.saved = saved; a.M = M; a.x_bounded = x_bounded;
  return launch_lstm<8>(a, spmi::as_stream(stream));
}

int spmi_lstm_rows_fwd(const float* x, int64_t ldx, int64_t I, const void* ws_ih, const void* ws_hh,
                       const float* b_ih, const float* b_hh, const float* h_in, const float* c_in,
                       int64_t M, int64_t H, float* h_out, float* c_out, int x_bounded, void* stream) {
  return lstm_rows_fwd(x, ldx, I, ws_ih, ws_hh, b_ih, b_hh, h_in, c_in, M, H, h_out, c_out, nullptr, x_bounded,
                       stream);
}

int spmi_lstm_rows_fwd_saved(const float* x, int64_t ldx, int64_t I, const void* ws_ih, const void* ws_hh,
                             const float* b_ih, const float* b_hh, const float* h_in, const float* c_in,
                             int64_t M, int64_t H, float* h_out, float* c_out, float* saved, int x_bounded,
                             void* stream) {
  SPMI_REQUIRE(saved || M == 0, SPMI_ERR_INVALID_ARG, "spmi_lstm_rows_fwd_saved: saved is null");
  return lstm_rows_fwd(x, ldx, I, ws_ih, ws_hh, b_ih, b_hh, h_in, c_in, M, H, h_out, c_out, saved, x_bounded,
                       stream);
}

}  // extern "C"
