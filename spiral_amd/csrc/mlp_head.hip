// The output head in inference as ONE kernel for gfx950: every Linear (+ folded eval-mode
// BatchNorm) + ReLU of the reference's MLP (spiro/dtdg/models/decoder/torch/
// simpleDecoder.py:45-79, called from SequentialDecoder.forward,
// implicitTimeModels.py:157-158) applied row by row without writing an intermediate.
//
// Why: the head's widths are tens of units (the examples use 10, 20, 10, 5, 1), so the
// library path is a chain of ~15 launches that each stream a [N, width] matrix through
// HBM; here the [N, I] input is read once and only the [N, out] result is written.
// HBM-bound: N * (I + out) * 4 bytes.
//
// Two kernels.  For I a multiple of 32 up to 256, and for 320, 384, 448 and 512, the whole head
// runs on the fp16 matrix cores (mlp_head_mfma_kernel, below); other widths take the all-VALU
// kernel described next.
//
// Mapping (VALU kernel): a wave owns 64 rows, one per lane.  The input arrives in 32-column chunks
// (coalesced 128-byte row segments, the next chunk in flight while the current one is
// consumed) and is transposed through a padded, wave-private LDS tile so that lane r reads
// column k of row r conflict-free.  All weights sit in LDS (copied once per workgroup) and
// are read as broadcasts.  Layer 1 keeps its (zero-padded) outputs in registers; the
// remaining layers are a few hundred FMAs per row: activations ping-pong between two
// wave-private LDS tiles [unit][lane], weights in torch's own [out, in] layout.

#include "common.h"

namespace {

constexpr int kMaxRest = 7;       // layers after the first
constexpr int kMaxWidth = 32;     // widest layer output the kernel holds per lane
constexpr int kChunk = 32;        // input columns per staged chunk
constexpr int kPad = 65;          // LDS tile stride (floats): lane r, column k -> bank (k + r) % 32
constexpr int kWaves = 4;         // waves per workgroup
constexpr int kMaxW1 = 8192;      // floats of the padded layer-1 weight kept in LDS
constexpr int kMaxRestFloats = 4096;
constexpr int kWaveFloats = kChunk * kPad + kMaxWidth * 64;  // x tile (reused as act0) + act1

struct MlpArgs {
  const float* x;
  int64_t ldx;
  int64_t M;
  int I;
  const float* w1t;   // [I][O1P], zero-padded columns
  const float* b1;    // [O1P]
  int out1;           // real width of layer 1
  int o1p;            // padded width of layer 1 (8, 16 or 32): row length of w1t, length of b1
  int relu1;
  int n_rest;
  const float* rest;  // per layer: weight [out][in], then bias [out]
  int rest_floats;
  int in_w[kMaxRest];
  int out_w[kMaxRest];
  int relu[kMaxRest];
  int final_act;  // 0 none, 1 sigmoid, 2 relu, 3 tanh
  float* out;
  int64_t ldo;
};

__device__ __forceinline__ float final_activation(float v, int code) {
  if (code == 1) return 1.0f / (1.0f + expf(-v));
  if (code == 2) return fmaxf(v, 0.0f);
  if (code == 3) return tanhf(v);
  return v;
}

// Transposes one 64-row x 32-column chunk held as 8 float4 per lane into the wave's tile.
__device__ __forceinline__ void stage_chunk(float* xs, const float4 (&v)[8], int lane) {
  const int lcol = (lane & 7) * 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = i * 8 + (lane >> 3);
    xs[(lcol + 0) * kPad + r] = v[i].x;
    xs[(lcol + 1) * kPad + r] = v[i].y;
    xs[(lcol + 2) * kPad + r] = v[i].z;
    xs[(lcol + 3) * kPad + r] = v[i].w;
  }
}

template <int O1P>
__device__ __forceinline__ void layer1_chunk(float (&acc)[O1P], const float* xs, const float* w,
                                             int lane) {
#pragma unroll 8
  for (int k = 0; k < kChunk; ++k) {
    const float xv = xs[k * kPad + lane];
    float wv[O1P];  // one broadcast read of the weight row (every lane the same address)
#pragma unroll
    for (int o = 0; o < O1P; o += 4)
      *reinterpret_cast<float4*>(&wv[o]) = *reinterpret_cast<const float4*>(w + k * O1P + o);
#pragma unroll
    for (int o = 0; o < O1P; ++o) acc[o] = fmaf(xv, wv[o], acc[o]);
  }
}

template <int O1P>
__global__ __launch_bounds__(64 * kWaves) void mlp_head_kernel(MlpArgs p) {
  extern __shared__ float4 smem4[];
  float* s_w1 = reinterpret_cast<float*>(smem4);          // [I][O1P]
  float* s_rest = s_w1 + p.I * O1P;                        // packed later layers
  float* s_wave = s_rest + ((p.rest_floats + 3) & ~3);     // per-wave tiles
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* xs = s_wave + wave * kWaveFloats;
  float* act1 = xs + kChunk * kPad;

  // weights of every layer into LDS, once per workgroup
  for (int i = threadIdx.x * 4; i < p.I * O1P; i += 64 * kWaves * 4)
    *reinterpret_cast<float4*>(s_w1 + i) = *reinterpret_cast<const float4*>(p.w1t + i);
  for (int i = threadIdx.x; i < p.rest_floats; i += 64 * kWaves) s_rest[i] = p.rest[i];
  __syncthreads();

  const int64_t row0 = (static_cast<int64_t>(blockIdx.x) * kWaves + wave) * 64;
  if (row0 >= p.M) return;  // no barrier below

  // chunk loads: instruction i covers rows i*8 + lane/8, columns (lane%8)*4 .. +3
  const float* src[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int64_t r = row0 + i * 8 + (lane >> 3);
    if (r >= p.M) r = p.M - 1;  // clamp: computed, never stored
    src[i] = p.x + r * p.ldx + (lane & 7) * 4;
  }
  float acc[O1P];
#pragma unroll
  for (int o = 0; o < O1P; ++o) acc[o] = p.b1[o];

  // Two register sets, loop unrolled by two chunks: the loads of chunk c+1 are in flight
  // while chunk c is transposed and multiplied (no register copies between iterations).
  float4 va[8], vb[8];
  const int n_chunks = p.I / kChunk;
#pragma unroll
  for (int i = 0; i < 8; ++i) va[i] = *reinterpret_cast<const float4*>(src[i]);
  for (int c = 0; c < n_chunks; c += 2) {
    if (c + 1 < n_chunks) {
#pragma unroll
      for (int i = 0; i < 8; ++i) vb[i] = *reinterpret_cast<const float4*>(src[i] + (c + 1) * kChunk);
    }
    stage_chunk(xs, va, lane);
    layer1_chunk<O1P>(acc, xs, s_w1 + c * kChunk * O1P, lane);
    if (c + 1 < n_chunks) {
      if (c + 2 < n_chunks) {
#pragma unroll
        for (int i = 0; i < 8; ++i) va[i] = *reinterpret_cast<const float4*>(src[i] + (c + 2) * kChunk);
      }
      stage_chunk(xs, vb, lane);
      layer1_chunk<O1P>(acc, xs, s_w1 + (c + 1) * kChunk * O1P, lane);
    }
  }

  const bool live = row0 + lane < p.M;
  if (p.n_rest == 0) {
#pragma unroll
    for (int o = 0; o < O1P; ++o) {
      if (o < p.out1 && live) {
        const float v = p.relu1 ? fmaxf(acc[o], 0.0f) : acc[o];
        p.out[(row0 + lane) * p.ldo + o] = final_activation(v, p.final_act);
      }
    }
    return;
  }
  float* a_in = xs;  // the x tile is free now: it becomes the first activation tile
  float* a_out = act1;
#pragma unroll
  for (int o = 0; o < O1P; ++o)
    if (o < p.out1) a_in[o * 64 + lane] = p.relu1 ? fmaxf(acc[o], 0.0f) : acc[o];

  const float* w = s_rest;
  for (int l = 0; l < p.n_rest; ++l) {
    const int in_w = p.in_w[l], out_w = p.out_w[l];
    const float* b = w + in_w * out_w;
    const bool last = l + 1 == p.n_rest;
    for (int o = 0; o < out_w; ++o) {
      float s = b[o];
      const float* wr = w + o * in_w;  // broadcast reads: every lane the same address
      for (int k = 0; k < in_w; ++k) s = fmaf(a_in[k * 64 + lane], wr[k], s);
      if (p.relu[l]) s = fmaxf(s, 0.0f);
      if (last) {
        if (live) p.out[(row0 + lane) * p.ldo + o] = final_activation(s, p.final_act);
      } else {
        a_out[o * 64 + lane] = s;
      }
    }
    w = b + out_w;
    float* t = a_in;
    a_in = a_out;
    a_out = t;
  }
}

// ---- the whole head on the matrix cores ---------------------------------------------------
// Y1^T = W1 X^T with W1 as the A operand and the rows of x as the B operand of
// v_mfma_f32_32x32x16_f16: lane (n, h) of a wave supplies eight consecutive k of data row n,
// which is two float4 loads straight from global memory (no transpose through LDS), and
// receives 16 of the 32 (zero-padded) layer-1 outputs of that same row:
//     element e of lane (n, h)  =  unit (e & 3) + 8 (e >> 2) + 4 h  of row n.
// Every later layer is again W_l (A operand, <= 32 x 32, zero-padded) times the activations
// (B operand).  The contraction index of a matrix instruction can be numbered freely as long
// as both operands agree, so W_l's image is stored with its k slots in the order in which a
// lane already holds its activations (K-step t, slot 8 h + j  <->  element e = 8 t + j):
// the accumulator of one layer is the B operand of the next with no data movement at all.
// Operands travel as three fp16 planes (p0 = fp16(v), p1 = fp16(v/64), p2 = fp16((v - p0)*64);
// x*w = p0*p0 + p1*p2 + p2*p1), which keeps the sums as close to float64 as an fp32 FMA
// chain; inputs and activations must lie within +-65504.  A wave keeps the whole K of its 32
// rows in registers and refills each K-step's registers with the next tile's as soon as they
// are split, so its loads are in flight across the whole tail.
using f32x16 = float __attribute__((ext_vector_type(16)));
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));
constexpr int kMfmaWaves = 4;
constexpr int kPlane = 32 * 32;          // bytes: 32 units x 16 k slots (fp16)
constexpr int kKStep = 3 * kPlane;       // three planes
constexpr int kTailImage = 2 * kKStep;   // a later layer: two K-steps (32 inputs)

__device__ __forceinline__ void split_planes(float v, _Float16& p0, _Float16& p1, _Float16& p2) {
  p0 = static_cast<_Float16>(v);
  p1 = static_cast<_Float16>(v * 0x1p-6f);
  p2 = static_cast<_Float16>((v - static_cast<float>(p0)) * 0x1p6f);
}

// Eight values at once: p1 from p0 by a packed fp16 multiply (equal to fp16(v / 64) wherever
// p0 / 64 is a normal fp16 number; below that, one more rounding of at most 2^-25), the
// residual by a mixed-precision fma that reads the fp16 value directly.
__device__ __forceinline__ void split8(const float (&v)[8], f16x8& x0, f16x8& x1, f16x8& x2) {
#pragma unroll
  for (int q = 0; q < 8; ++q) x0[q] = static_cast<_Float16>(v[q]);
  x1 = x0 * static_cast<_Float16>(0x1p-6f);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float d = __builtin_fmaf(static_cast<float>(x0[q]), -1.0f, v[q]);
    x2[q] = static_cast<_Float16>(d * 0x1p6f);
  }
}

__device__ __forceinline__ f32x16 mfma3(const char* img, const f16x8& x0, const f16x8& x1,
                                        const f16x8& x2, f32x16 acc) {
  const f16x8 w0 = *reinterpret_cast<const f16x8*>(img);
  const f16x8 w1 = *reinterpret_cast<const f16x8*>(img + kPlane);
  const f16x8 w2 = *reinterpret_cast<const f16x8*>(img + 2 * kPlane);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, x1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x2, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, x0, acc, 0, 0, 0);
}

// KT K-steps of a row are in registers at a time; wider inputs (KC > 1 chunks of 16 KT inputs)
// pass through the same registers chunk by chunk while the first layer's sums stay in their
// accumulators.  The first layer's image of all KT KC K-steps stays in LDS, which then leaves
// room for one workgroup per CU: it has eight waves instead of four.
template <int KT, int KC>
__global__ __launch_bounds__(64 * (KC > 1 ? 2 : 1) * kMfmaWaves, 2) void mlp_head_mfma_kernel(MlpArgs p) {
  static_assert(KT % 2 == 0, "rows are read in pairs of K-steps");
  constexpr int NW = (KC > 1 ? 2 : 1) * kMfmaWaves;
  extern __shared__ float4 smem4[];
  char* s_l1 = reinterpret_cast<char*>(smem4);             // [KT KC][3][2][32][8] fp16
  char* s_tail = s_l1 + KT * KC * kKStep;                  // [n_rest][2][3][2][32][8] fp16
  float* s_bias = reinterpret_cast<float*>(s_tail + p.n_rest * kTailImage);  // [n_rest][32]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31;
  const int lh = lane >> 5;

  // W1 (given [I][o1p], zero-padded columns) -> planes in the A layout, once per workgroup
  for (int idx = tid; idx < KT * KC * 16 * 32; idx += 64 * NW) {
    const int k = idx >> 5, f = idx & 31;
    const float v = f < p.o1p ? p.w1t[k * p.o1p + f] : 0.0f;
    _Float16 q0, q1, q2;
    split_planes(v, q0, q1, q2);
    // a lane half reads 64 contiguous bytes of its row per pair of K-steps: inputs
    // 32 u + 16 half + 8 s + {0..7} sit in K-step 2 u + s, k half `half`
    const int step = 2 * (k >> 5) + ((k >> 3) & 1), half = (k >> 4) & 1;
    _Float16* dst = reinterpret_cast<_Float16*>(s_l1 + step * kKStep + half * 512 + f * 16) + (k & 7);
    dst[0] = q0;
    dst[kPlane / 2] = q1;
    dst[kPlane] = q2;
  }
  // later layers (given [out][in] then bias [out], fp32): slot kappa of K-step t holds input
  // unit (kappa & 3) + 8 (2 t + ((kappa & 7) >> 2)) + 4 (kappa >> 3)
  {
    const float* w = p.rest;
    for (int l = 0; l < p.n_rest; ++l) {
      const int in_w = p.in_w[l], out_w = p.out_w[l];
      for (int idx = tid; idx < 2 * 32 * 16; idx += 64 * NW) {
        const int t = idx >> 9, f = (idx >> 4) & 31, kappa = idx & 15;
        const int unit = (kappa & 3) + 8 * (2 * t + ((kappa & 7) >> 2)) + 4 * (kappa >> 3);
        const float v = (f < out_w && unit < in_w) ? w[f * in_w + unit] : 0.0f;
        _Float16 q0, q1, q2;
        split_planes(v, q0, q1, q2);
        _Float16* dst = reinterpret_cast<_Float16*>(s_tail + l * kTailImage + t * kKStep +
                                                    (kappa >> 3) * 512 + f * 16) + (kappa & 7);
        dst[0] = q0;
        dst[kPlane / 2] = q1;
        dst[kPlane] = q2;
      }
      if (tid < 32) s_bias[l * 32 + tid] = tid < out_w ? w[in_w * out_w + tid] : 0.0f;
      w += (in_w + 1) * out_w;
    }
  }
  __syncthreads();

  const int64_t n_tiles = (p.M + 31) / 32;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * NW;
  // Tiles are walked from the last row down: what the producer of x wrote last is what the
  // memory-side cache still holds when this kernel starts.
  int64_t tile = n_tiles - 1 - (static_cast<int64_t>(blockIdx.x) * NW + wave);
  if (tile < 0) return;  // no barrier below

  float b1[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int f = (e & 3) + 8 * (e >> 2) + 4 * lh;
    b1[e] = f < p.o1p ? p.b1[f] : 0.0f;
  }
  const int out_last = p.n_rest ? p.out_w[p.n_rest - 1] : p.out1;

  float4 a[KT][2];
  {
    int64_t r = tile * 32 + lr;
    r = r < p.M ? r : p.M - 1;  // clamp: computed, never stored
    const float* src = p.x + r * p.ldx + 16 * lh;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      a[t][0] = *reinterpret_cast<const float4*>(src + 32 * (t >> 1) + 8 * (t & 1));
      a[t][1] = *reinterpret_cast<const float4*>(src + 32 * (t >> 1) + 8 * (t & 1) + 4);
    }
  }
  const int frag = lh * 512 + lr * 16;  // this lane's 16 bytes inside a plane [k half][unit][8]

  while (tile >= 0) {
    const int64_t next = tile - stride;
    int64_t r_own = tile * 32 + lr;
    r_own = r_own < p.M ? r_own : p.M - 1;
    int64_t r_next = (next >= 0 ? next : tile) * 32 + lr;  // last pass re-reads its own tile
    r_next = r_next < p.M ? r_next : p.M - 1;

    f32x16 cur;
#pragma unroll
    for (int e = 0; e < 16; ++e) cur[e] = b1[e];
#pragma unroll 1  // unrolled, the chunks' loads are scheduled into each other and registers spill
    for (int c = 0; c < KC; ++c) {
      // what refills the registers: the next chunk of this tile, or the first of the next tile
      const float* src = c + 1 < KC ? p.x + r_own * p.ldx + 16 * lh + (c + 1) * 16 * KT
                                    : p.x + r_next * p.ldx + 16 * lh;
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const float v[8] = {a[t][0].x, a[t][0].y, a[t][0].z, a[t][0].w,
                            a[t][1].x, a[t][1].y, a[t][1].z, a[t][1].w};
        f16x8 x0, x1, x2;
        split8(v, x0, x1, x2);
        a[t][0] = *reinterpret_cast<const float4*>(src + 32 * (t >> 1) + 8 * (t & 1));
        a[t][1] = *reinterpret_cast<const float4*>(src + 32 * (t >> 1) + 8 * (t & 1) + 4);
        __builtin_amdgcn_sched_barrier(0);  // keeps the reloads from being hoisted over the tile
        cur = mfma3(s_l1 + (c * KT + t) * kKStep + frag, x0, x1, x2, cur);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (p.relu1) {
#pragma unroll
      for (int e = 0; e < 16; ++e) cur[e] = fmaxf(cur[e], 0.0f);
    }

    for (int l = 0; l < p.n_rest; ++l) {
      const char* img = s_tail + l * kTailImage + frag;
      f32x16 nxt;
      {  // bias of units 8 q + 4 lh + {0..3} = elements 4 q .. 4 q + 3
        const float* bl = s_bias + l * 32 + 4 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 bq = *reinterpret_cast<const float4*>(bl + 8 * q);
          nxt[4 * q + 0] = bq.x;
          nxt[4 * q + 1] = bq.y;
          nxt[4 * q + 2] = bq.z;
          nxt[4 * q + 3] = bq.w;
        }
      }
      f16x8 x0, x1, x2;
      {
        const float v[8] = {cur[0], cur[1], cur[2], cur[3], cur[4], cur[5], cur[6], cur[7]};
        split8(v, x0, x1, x2);
        nxt = mfma3(img, x0, x1, x2, nxt);
      }
      if (p.in_w[l] > 16) {  // uniform
        const float v[8] = {cur[8], cur[9], cur[10], cur[11], cur[12], cur[13], cur[14], cur[15]};
        split8(v, x0, x1, x2);
        nxt = mfma3(img + kKStep, x0, x1, x2, nxt);
      }
      if (p.relu[l]) {
#pragma unroll
        for (int e = 0; e < 16; ++e) nxt[e] = fmaxf(nxt[e], 0.0f);
      }
      cur = nxt;
    }

    const int64_t row = tile * 32 + lr;
    if (row < p.M) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (8 * q >= out_last) break;  // uniform: a one-unit output tests four elements, not 16
#pragma unroll
        for (int e = 4 * q; e < 4 * q + 4; ++e) {
          const int f = (e & 3) + 8 * q + 4 * lh;
          if (f < out_last) p.out[row * p.ldo + f] = final_activation(cur[e], p.final_act);
        }
      }
    }
    tile = next;
  }
}

template <int KT, int KC = 1>
int launch_mfma(const MlpArgs& a, hipStream_t st) {
  auto kernel = mlp_head_mfma_kernel<KT, KC>;
  constexpr int NW = (KC > 1 ? 2 : 1) * kMfmaWaves;
  const size_t lds = static_cast<size_t>(KT * KC) * kKStep + static_cast<size_t>(a.n_rest) * (kTailImage + 128);
  SPMI_RAISE_LDS_LIMIT(kernel, lds);
  const int64_t n_tiles = (a.M + 31) / 32;
  const int64_t wgs = (n_tiles + NW - 1) / NW;
  constexpr int64_t max_wgs = KC > 1 ? 256 : 512;  // what is resident at once
  kernel<<<static_cast<unsigned>(wgs < max_wgs ? wgs : max_wgs), 64 * NW, lds, st>>>(a);
  SPMI_CHECK_LAUNCH("mlp_head_mfma");
  return SPMI_OK;
}

size_t lds_bytes(int64_t I, int o1p, int rest_floats) {
  return sizeof(float) * (static_cast<size_t>(I) * o1p + ((rest_floats + 3) & ~3) +
                          static_cast<size_t>(kWaves) * kWaveFloats);
}

template <int O1P>
int launch(const MlpArgs& a, hipStream_t st) {
  auto kernel = mlp_head_kernel<O1P>;
  const size_t lds = lds_bytes(a.I, O1P, a.rest_floats);
  SPMI_RAISE_LDS_LIMIT(kernel, lds);  // raised again when a larger size shows up
  const int64_t rows_per_block = 64 * kWaves;
  const unsigned grid = static_cast<unsigned>((a.M + rows_per_block - 1) / rows_per_block);
  kernel<<<grid, 64 * kWaves, lds, st>>>(a);
  SPMI_CHECK_LAUNCH("mlp_head");
  return SPMI_OK;
}

// Input widths the matrix-core kernel is instantiated for (the switch in spmi_mlp_head_fwd).
bool mfma_input_width(int64_t I) {
  return (I % 32 == 0 && I >= 32 && I <= 256) || I == 320 || I == 384 || I == 448 || I == 512;
}

}  // namespace

extern "C" {

int spmi_mlp_head_supported(int64_t I, int32_t out1, int32_t n_rest, const int32_t* rest_out) {
  if (I < kChunk || I % kChunk || out1 < 1 || out1 > kMaxWidth || n_rest < 0 || n_rest > kMaxRest)
    return 0;
  const int o1p = out1 <= 8 ? 8 : (out1 <= 16 ? 16 : 32);
  // the matrix-core kernel keeps the first layer as fp16 planes of 32 padded units whatever
  // its width; the other kernel keeps it in fp32
  if (!mfma_input_width(I) && I * o1p > kMaxW1) return 0;
  int64_t floats = 0;
  int width = out1;
  for (int l = 0; l < n_rest; ++l) {
    if (!rest_out || rest_out[l] < 1 || rest_out[l] > kMaxWidth) return 0;
    floats += static_cast<int64_t>(width + 1) * rest_out[l];
    width = rest_out[l];
  }
  return floats <= kMaxRestFloats;
}

int spmi_mlp_head_fwd(const float* x, int64_t ldx, int64_t M, int64_t I, const float* w1t,
                      const float* b1, int32_t out1, int32_t relu1, int32_t n_rest,
                      const float* rest_packed, const int32_t* rest_out, const int32_t* rest_relu,
                      int32_t final_act, float* out, int64_t ldo, void* stream) {
  SPMI_REQUIRE(spmi_mlp_head_supported(I, out1, n_rest, rest_out), SPMI_ERR_UNSUPPORTED,
               "spmi_mlp_head_fwd: shape outside the kernel's limits (I=%lld, first width %d, %d "
               "more layers): see spmi_mlp_head_supported", (long long)I, out1, n_rest);
  SPMI_REQUIRE(M >= 0 && final_act >= 0 && final_act <= 3, SPMI_ERR_INVALID_ARG,
               "spmi_mlp_head_fwd: bad M or activation code");
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(x && w1t && b1 && out && ldx >= I && ldx % 4 == 0 &&
                   (reinterpret_cast<uintptr_t>(x) & 15u) == 0 &&
                   (reinterpret_cast<uintptr_t>(w1t) & 15u) == 0 && (n_rest == 0 || rest_packed),
               SPMI_ERR_INVALID_ARG, "spmi_mlp_head_fwd: null pointer, bad ldx or unaligned x / w1t");
  MlpArgs a{};
  a.x = x; a.ldx = ldx; a.M = M; a.I = static_cast<int>(I);
  a.w1t = w1t; a.b1 = b1; a.out1 = out1; a.relu1 = relu1; a.n_rest = n_rest; a.rest = rest_packed;
  int width = out1, floats = 0;
  for (int l = 0; l < n_rest; ++l) {
    a.in_w[l] = width; a.out_w[l] = rest_out[l]; a.relu[l] = rest_relu ? rest_relu[l] : 0;
    floats += (width + 1) * rest_out[l];
    width = rest_out[l];
  }
  a.rest_floats = floats;
  SPMI_REQUIRE(ldo >= width, SPMI_ERR_INVALID_ARG, "spmi_mlp_head_fwd: ldo %lld < output width %d",
               (long long)ldo, width);
  a.final_act = final_act; a.out = out; a.ldo = ldo;
  a.o1p = out1 <= 8 ? 8 : (out1 <= 16 ? 16 : 32);
  hipStream_t st = spmi::as_stream(stream);
  switch (I) {  // a wave holds up to 256 inputs of its rows in registers: the matrix-core kernel
    case 32: return launch_mfma<2>(a, st);
    case 64: return launch_mfma<4>(a, st);
    case 96: return launch_mfma<6>(a, st);
    case 128: return launch_mfma<8>(a, st);
    case 160: return launch_mfma<10>(a, st);
    case 192: return launch_mfma<12>(a, st);
    case 224: return launch_mfma<14>(a, st);
    case 256: return launch_mfma<16>(a, st);
    case 320: return launch_mfma<10, 2>(a, st);
    case 384: return launch_mfma<12, 2>(a, st);
    case 448: return launch_mfma<14, 2>(a, st);
    case 512: return launch_mfma<16, 2>(a, st);
    default: break;
  }
  if (out1 <= 8) return launch<8>(a, st);
  if (out1 <= 16) return launch<16>(a, st);
  return launch<32>(a, st);
}

}  // extern "C"
