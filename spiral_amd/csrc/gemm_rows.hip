// Y = act(X W^T + b) for a tall X [M, K] and a small weight [N, K]: the dense projection of
// GraphConv / SGConv / SAGEConv (`torch.matmul(feat, weight)` inside dgl GraphConv.forward,
// call sites spiro/dtdg/models/encoder/implicitTimeEncoder/staticGraphEncoder.py:113,116) and
// of the decoders' Linear layers (spiro/dtdg/models/decoder/torch/simpleDecoder.py:43-46).
//
// fp32 in, fp32 out, on the fp16 matrix cores with the three-plane operand format of
// cell_split.hip (p0 = fp16(v), p1 = fp16(v/64), p2 = fp16((v - p0)*64); x*w = p0*p0 + p1*p2 +
// p2*p1, fp32 accumulation): sums land closer to float64 than an fp32 FMA chain.  The rows of
// X may hold any finite fp32 value: a wave has the whole row (or, for K = 256 and 384, a chunk
// of 128 columns of it) in registers before it starts on it, so every row (chunk) is multiplied
// by a power of two that puts its largest magnitude in [2^13, 2^14) (exact), and the
// accumulators by the inverse power at the end (of the chunk).  Raw node
// features such as degree counts can exceed fp16's 65504; without the scaling such a row
// would come out as inf/NaN.  The weights must lie within +-65504 / 64.
//
// The shape makes this an HBM-bound kernel (M*(K+N)*4 bytes against 3*2*M*K*N flops on
// matrix cores that are ~15x faster than that), so the design is about the stream of X:
//  * the weight planes ([K/16][3][2][N][8] fp16, from spmi_split_f16x3) sit in LDS for the
//    lifetime of a workgroup (96 KB at K = N = 128) and are read as ds_read_b128 fragments;
//  * X never touches LDS: the A fragment of v_mfma_f32_32x32x16_f16 is eight consecutive k
//    of one row per lane, i.e. two float4 loads straight from global memory, split into the
//    three planes in registers;
//  * a wave owns 32 rows x all N columns and keeps up to 128 columns of its rows in registers
//    (64 VGPRs); as soon as the slice of K-step t is split, the same registers receive the
//    K-step t of the next chunk or of the wave's next tile, so 16 KB per wave (128 KB per CU)
//    are in flight while the matrix cores work;
//  * waves are independent after the weight load: no barrier in the loop;
//  * wide outputs (N > 128) and K = 256, 384 are cut into column slices of 128 (64 for
//    K = 256, 384) that fit LDS: the workgroups that hold the slices of one row range run on one XCD
//    (blockIdx -> XCD is round-robin) and walk the same tiles, so the rows they all read
//    come from HBM once and from that XCD's L2 afterwards.

#include <type_traits>

#include "common.h"

namespace {

using f32x16 = float __attribute__((ext_vector_type(16)));
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));

struct RowsArgs {
  const float* x;
  int64_t ldx;
  int64_t M;
  const uint4* wp;  // weight planes, [KT][3][2][N][8] fp16
  const float* bias;
  const float* row_scale;  // nullable: out row r is multiplied by row_scale[r] at the end
  int act;          // 0 none, 1 relu, 2 leaky_relu
  float slope;
  float* out;
  int64_t ldo;
  int n_total;  // columns of the whole weight; a workgroup computes NB * 32 of them
  int slices;   // n_total / (NB * 32)
  int accumulate;  // chunked kernels only: add what `out` holds before bias and activation
};

// The three planes of eight consecutive k of one row, the row's power-of-two scale applied on
// the way.  p1 is taken from p0 by a packed fp16 multiply (equal to fp16(v / 64) wherever
// p0 / 64 is a normal fp16 number; below that, one more rounding of at most 2^-25) and the
// residual v - p0 by a mixed-precision fma that reads the fp16 value directly.
__device__ __forceinline__ void split8(const float4& lo, const float4& hi, float m, f16x8& p0, f16x8& p1,
                                       f16x8& p2) {
  float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] *= m;
#pragma unroll
  for (int q = 0; q < 8; ++q) p0[q] = static_cast<_Float16>(v[q]);
  p1 = p0 * static_cast<_Float16>(0x1p-6f);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float d = __builtin_fmaf(static_cast<float>(p0[q]), -1.0f, v[q]);
    p2[q] = static_cast<_Float16>(d * 0x1p6f);
  }
}

// KT K-steps (16 KT columns) of a row are in registers at a time.  KC = 1: that is the whole
// row.  KC > 1: a row passes through the registers in KC chunks, each with a power-of-two
// scale of its own, and the chunks' sums are added up in a second accumulator set after the
// inverse scale; 64 row registers instead of 16 KT KC / 2 keep two waves on a SIMD for
// K = 256 and 384, where one wave with the whole row had nothing to run under its own waits.
template <int KT, int NB, int KC>
__global__ __launch_bounds__(KT > 8 ? 256 : 512, KT > 8 ? 1 : 2) void gemm_rows_kernel(RowsArgs p) {
  // More than 128 columns in registers at once: four waves per workgroup, one per SIMD, with
  // the whole register file; otherwise eight waves (two per SIMD).
  constexpr int NW = KT > 8 ? 4 : 8;
  constexpr int N = NB * 32;
  constexpr int PLANE = N * 32;           // bytes of one plane of one K-step
  constexpr int KSTEP = 3 * PLANE;        // bytes of one K-step
  constexpr int B16 = KT * KC * KSTEP / 16;  // 16-byte chunks of the weight image
  extern __shared__ uint4 smem4[];
  const char* smem = reinterpret_cast<const char*>(smem4);

  // Workgroup b runs on XCD b % 8; the column slices of one row group sit on one XCD.
  const int b = blockIdx.x;
  const int slice = (b >> 3) % p.slices;
  const int group = (b & 7) + 8 * (b / (8 * p.slices));
  const int n_groups = gridDim.x / p.slices;
  const int col0 = slice * N;

  const int tid = threadIdx.x;
  // image [K-step][plane][k half][n_total][8]: 16-byte chunk index ((t*3+q)*2+h) * n_total + n.
  // K order: a lane reads 64 contiguous bytes of its row at a time (half of a 128-byte line,
  // four loads back to back, so the line is fetched from L2 once), which feed two K-steps:
  // K-step 2u + v, lane half lh, slot i stands for k = 32 u + 16 lh + 8 v + i.  The image is
  // permuted to match: LDS (step 2u + v, half lh) <- source (step 2u + lh, half v).
  static_assert(KT % 2 == 0, "K-steps come in pairs");
  for (int c = tid; c < B16; c += NW * 64) {
    const int n = c % N, hh = (c / N) & 1, q = (c / (2 * N)) % 3, t = c / (6 * N);
    const int rest = (((t & ~1) + hh) * 3 + q) * 2 + (t & 1);
    smem4[c] = p.wp[static_cast<size_t>(rest) * p.n_total + col0 + n];
  }
  __syncthreads();

  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31;
  const int lh = lane >> 5;
  const int64_t n_tiles = (p.M + 31) / 32;
  const int64_t stride = static_cast<int64_t>(n_groups) * NW;
  // Tiles are walked from the last row down: the rows a producer (the aggregation of the layer
  // before) wrote last are the ones the memory-side cache still holds when this kernel starts.
  int64_t tile = n_tiles - 1 - (static_cast<int64_t>(group) * NW + wave);
  if (tile < 0) return;

  float bias[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bias[nb] = p.bias ? p.bias[col0 + nb * 32 + lr] : 0.0f;

  // Rows past M are clamped to the last row: computed, never stored.
  auto row_ptr = [&](int64_t t) {
    int64_t r = t * 32 + lr;
    r = r < p.M ? r : p.M - 1;
    return p.x + r * p.ldx + 16 * lh;
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
    const float* src = row_ptr(tile);
#pragma unroll
    for (int u = 0; u < KT / 2; ++u) load_pair(src, u);
  }
  const int b_rd = (lh * N + lr) * 16;  // [plane][k half][n][8]: 512 contiguous bytes per half

  while (tile >= 0) {
    const int64_t next = tile - stride;
    // the last iteration re-reads its own tile: valid memory, nothing is done with it
    const float* src_next = row_ptr(next >= 0 ? next : tile);
    const float* src_own = row_ptr(tile);

    f32x16 acc[NB];
    f32x16 tot[KC > 1 ? NB : 1];  // sums of the chunks already done, in the row's own units
    if constexpr (KC > 1) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) tot[nb][e] = 0.0f;
    }

    // the row scales of this tile, fetched under the matrix work (rows past M: clamped)
    float rs[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int64_t r = tile * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      r = r < p.M ? r : p.M - 1;
      rs[e] = p.row_scale ? p.row_scale[r] : 1.0f;
    }

    float up, down;
#pragma unroll 1
    for (int c = 0; c < KC; ++c) {
    // what refills the registers: the next chunk of this tile, or the first of the next tile
    const float* src = c + 1 < KC ? src_own + (c + 1) * 16 * KT : src_next;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nb][e] = 0.0f;

    // Power-of-two scale of this lane's row (the lane pair l, l ^ 32 holds one row): the
    // largest magnitude goes to [2^13, 2^14), a zero or non-finite row is left alone.
    {
      float m = 0.0f;
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        m = fmaxf(m, fmaxf(fmaxf(fabsf(a[t][0].x), fabsf(a[t][0].y)), fmaxf(fabsf(a[t][0].z), fabsf(a[t][0].w))));
        m = fmaxf(m, fmaxf(fmaxf(fabsf(a[t][1].x), fabsf(a[t][1].y)), fmaxf(fabsf(a[t][1].z), fabsf(a[t][1].w))));
      }
      m = fmaxf(m, __shfl_xor(m, 32));
      const int ex = static_cast<int>((__float_as_uint(m) >> 23) & 0xffu);  // biased exponent
      const bool ok = ex != 0 && ex != 255;                                   // normal and finite
      const int shift = ok ? 127 + 13 - ex : 0;                               // 2^shift * m in [2^13, 2^14)
      const int sh = shift > 100 ? 100 : (shift < -100 ? -100 : shift);
      up = __uint_as_float(static_cast<uint32_t>(127 + sh) << 23);
      down = __uint_as_float(static_cast<uint32_t>(127 - sh) << 23);
    }

#pragma unroll
    for (int t = 0; t < KT; ++t) {
      f16x8 p0, p1, p2;
      split8(a[t][0], a[t][1], up, p0, p1, p2);
      if (t & 1) load_pair(src, t >> 1);
      // Keep the K-steps apart: left alone, the scheduler hoists every reload to the top of
      // the tile and then needs a second copy of the whole row in registers.
      __builtin_amdgcn_sched_barrier(0);
      const char* sb = smem + (c * KT + t) * KSTEP + b_rd;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const f16x8 b0 = *reinterpret_cast<const f16x8*>(sb + nb * 32 * 16);
        const f16x8 b1 = *reinterpret_cast<const f16x8*>(sb + nb * 32 * 16 + PLANE);
        const f16x8 b2 = *reinterpret_cast<const f16x8*>(sb + nb * 32 * 16 + 2 * PLANE);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2, b1, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, b2, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, b0, acc[nb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (KC > 1) {  // this chunk's sums, back in the row's own units
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ur = (e & 3) + 8 * (e >> 2);
        const float d0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(down), ur));
        const float d1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(down), ur + 4));
        const float dn = lh ? d1 : d0;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) tot[nb][e] = __builtin_fmaf(acc[nb][e], dn, tot[nb][e]);
      }
    }
    }  // chunks

    // C layout of the 32x32 MFMA: element e of lane l is row (e&3) + 8 (e>>2) + 4 (l>>5),
    // column l&31: a store covers two 128-byte row segments.
    const int64_t row0 = tile * 32;
    float* obase = p.out + row0 * p.ldo + col0;  // wave-uniform
    const int voff = 4 * lh * static_cast<int>(p.ldo) + lr;
    // One decision per tile, not per element: the activation and whether every row of the tile
    // exists are wave-uniform, and asked inside the loops they cost a scalar branch or two for
    // each of the 16 NB results of a lane.
    auto finish = [&](auto act_tag, auto whole_tag, auto add_tag) {
      constexpr int ACT = decltype(act_tag)::value;
      constexpr bool WHOLE = decltype(whole_tag)::value;
      constexpr bool ADD = decltype(add_tag)::value;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ur = (e & 3) + 8 * (e >> 2);
        if constexpr (!WHOLE) {
          if (row0 + ur + 4 * lh >= p.M) continue;
        }
        float* o = obase + ur * p.ldo;
        // the inverse scale of output row ur + 4 lh sits in lane ur + 4 lh (either half)
        const float d0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(down), ur));
        const float d1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(down), ur + 4));
        const float dn = lh ? d1 : d0;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          float v = (KC > 1 ? tot[nb][e] : acc[nb][e] * dn) + bias[nb];
          if constexpr (ADD) v += o[voff + nb * 32];
          if constexpr (ACT == 1) v = fmaxf(v, 0.0f);
          else if constexpr (ACT == 2) v = v > 0.0f ? v : v * p.slope;
          o[voff + nb * 32] = v * rs[e];
        }
      }
    };
    auto finish_act = [&](auto whole_tag, auto add_tag) {
      if (p.act == 0) finish(std::integral_constant<int, 0>{}, whole_tag, add_tag);
      else if (p.act == 1) finish(std::integral_constant<int, 1>{}, whole_tag, add_tag);
      else finish(std::integral_constant<int, 2>{}, whole_tag, add_tag);
    };
    auto finish_rows = [&](auto add_tag) {
      if (row0 + 32 <= p.M) finish_act(std::true_type{}, add_tag);
      else finish_act(std::false_type{}, add_tag);
    };
    if constexpr (KC > 1) {
      if (p.accumulate) finish_rows(std::true_type{});
      else finish_rows(std::false_type{});
    } else {
      finish_rows(std::false_type{});
    }
    tile = next;
  }
}

template <int KT, int NB, int KC = 1>
int launch(const RowsArgs& a, hipStream_t st) {
  constexpr size_t lds = static_cast<size_t>(KT) * KC * 3 * NB * 32 * 32;
  static_assert(lds <= 160 * 1024, "weight image does not fit LDS");
  auto kern = gemm_rows_kernel<KT, NB, KC>;
  SPMI_RAISE_LDS_LIMIT(kern, lds);
  const int64_t n_tiles = (a.M + 31) / 32;
  const int per_cu = (lds * 2 <= 160 * 1024 && KT <= 8) ? 2 : 1;
  // row groups of NW tiles, in whole sets of 8 XCDs; every group has `slices` workgroups
  constexpr int NW = KT > 8 ? 4 : 8;
  int64_t sets = (n_tiles + 8 * NW - 1) / (8 * NW);
  int64_t max_sets = 256 * per_cu / (8 * a.slices);
  if (max_sets < 1) max_sets = 1;
  if (sets > max_sets) sets = max_sets;
  kern<<<static_cast<unsigned>(sets * 8 * a.slices), NW * 64, lds, st>>>(a);
  SPMI_CHECK_LAUNCH("gemm_rows");
  return SPMI_OK;
}

// ---- rows stored as bfloat16 -------------------------------------------------------------------
// The same projection for feature matrices kept in bfloat16 (EvolveGCN-O's storage option,
// spiro's staticGraphEncoder.py:113,116 call sites with half the feature bytes).  A bfloat16
// value IS a matrix-core operand: the eight consecutive k of a lane's row are one 16-byte
// load, with no split and no scaling (fp32's exponent range).  The weight stays an fp32
// quantity: its three exact bf16 pieces (spmi_split_bf16x3) give three products per term,
// x * w to fp32 accuracy, where a bfloat16 library GEMM rounds the weight to 8 bits first.
// Results leave as bfloat16 (rounded once, to nearest even) or float32.
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));

struct RowsBf16Args {
  const uint16_t* x;  // bfloat16 [M, K], row stride ldx (elements)
  int64_t ldx;
  int64_t M;
  const uint4* wp;  // weight pieces, [KT][3][2][N][8] bfloat16
  const float* bias;
  const float* row_scale;
  int act;
  float slope;
  void* out;  // bfloat16 or float32 [M, N], row stride ldo (elements)
  int64_t ldo;
  int out_bf16;
};

template <int KT, int NB>
__global__ __launch_bounds__(512, 2) void gemm_rows_bf16_kernel(RowsBf16Args p) {
  constexpr int NW = 8;
  constexpr int N = NB * 32;
  constexpr int PLANE = N * 32;
  constexpr int KSTEP = 3 * PLANE;
  constexpr int B16 = KT * KSTEP / 16;
  extern __shared__ uint4 smem4[];
  const char* smem = reinterpret_cast<const char*>(smem4);
  const int tid = threadIdx.x;
  // K order (KT a multiple of 4): a lane reads 64 contiguous bytes of its row at a time, four
  // 16-byte loads back to back that feed four K-steps, and the two lane halves share a
  // 128-byte line, which is then fetched from L2 once: K-step 4u + v, lane half lh, slot i
  // stands for k = 64 u + 32 lh + 8 v + i.  The weight image is permuted to match: LDS
  // (step 4u + v, half lh) <- source (step 4u + 2 lh + v / 2, half v % 2).
  constexpr bool GROUPED = KT % 4 == 0;
  for (int c = tid; c < B16; c += NW * 64) {
    int src = c;
    if constexpr (GROUPED) {
      const int n = c % N, hh = (c / N) & 1, q = (c / (2 * N)) % 3, t = c / (6 * N);
      const int s_ = (t & ~3) + 2 * hh + ((t & 3) >> 1), h_ = t & 1;
      src = ((s_ * 3 + q) * 2 + h_) * N + n;
    }
    smem4[c] = p.wp[src];
  }
  __syncthreads();

  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31;
  const int lh = lane >> 5;
  const int64_t n_tiles = (p.M + 31) / 32;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * NW;
  int64_t tile = static_cast<int64_t>(blockIdx.x) * NW + wave;
  if (tile >= n_tiles) return;

  float bias[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bias[nb] = p.bias ? p.bias[nb * 32 + lr] : 0.0f;

  // Rows past M are clamped to the last row: computed, never stored.
  auto row_ptr = [&](int64_t t) {
    int64_t r = t * 32 + lr;
    r = r < p.M ? r : p.M - 1;
    return p.x + r * p.ldx + (GROUPED ? 32 : 8) * lh;
  };
  // element offset of K-step t inside the lane's share of the row
  auto k_off = [](int t) { return GROUPED ? 64 * (t >> 2) + 8 * (t & 3) : 16 * t; };
  uint4 a[KT];
  {
    const uint16_t* src = row_ptr(tile);
#pragma unroll
    for (int t = 0; t < KT; ++t) a[t] = *reinterpret_cast<const uint4*>(src + k_off(t));
  }
  const int b_rd = (lh * N + lr) * 16;

  while (tile < n_tiles) {
    const int64_t next = tile + stride;
    // the last iteration re-reads its own tile: valid memory, nothing is done with it
    const uint16_t* src = row_ptr(next < n_tiles ? next : tile);
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nb][e] = 0.0f;
    float rs[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int64_t r = tile * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      r = r < p.M ? r : p.M - 1;
      rs[e] = p.row_scale ? p.row_scale[r] : 1.0f;
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      union { uint4 u; bf16x8 v; } av;
      av.u = a[t];
      if constexpr (GROUPED) {
        // the next tile's rows, a 64-byte span at a time, into the registers of the four
        // K-steps that have just been used (K-step t itself is in `av`)
        if ((t & 3) == 3) {
#pragma unroll
          for (int v = 0; v < 4; ++v) a[t - 3 + v] = *reinterpret_cast<const uint4*>(src + k_off(t - 3 + v));
        }
      } else {
        a[t] = *reinterpret_cast<const uint4*>(src + k_off(t));
      }
      __builtin_amdgcn_sched_barrier(0);
      const char* sb = smem + t * KSTEP + b_rd;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const bf16x8 hi = *reinterpret_cast<const bf16x8*>(sb + nb * 32 * 16);
        const bf16x8 mid = *reinterpret_cast<const bf16x8*>(sb + nb * 32 * 16 + PLANE);
        const bf16x8 lo = *reinterpret_cast<const bf16x8*>(sb + nb * 32 * 16 + 2 * PLANE);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av.v, lo, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av.v, mid, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av.v, hi, acc[nb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const int64_t row0 = tile * 32;
    if (p.out_bf16) {
      // The accumulator layout gives a lane one column of sixteen rows: stored from there, a
      // tile is 64 NB two-byte stores per lane.  The wave's 32 x N results go through an LDS
      // area of its own instead and leave as 16-byte pieces of whole rows.
      __bf16* stage = reinterpret_cast<__bf16*>(smem4) + (static_cast<size_t>(KT) * KSTEP) / 2 + wave * 32 * N;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ur = (e & 3) + 8 * (e >> 2) + 4 * lh;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          float v = acc[nb][e] + bias[nb];
          if (p.act == 1) v = fmaxf(v, 0.0f);
          else if (p.act == 2) v = v > 0.0f ? v : v * p.slope;
          stage[ur * N + nb * 32 + lr] = static_cast<__bf16>(v * rs[e]);
        }
      }
      __builtin_amdgcn_wave_barrier();
      constexpr int PER_ROW = N / 8;            // 16-byte pieces of a row
      constexpr int PIECES = 32 * PER_ROW / 64;  // per lane
      __bf16* obase = static_cast<__bf16*>(p.out);
#pragma unroll
      for (int q = 0; q < PIECES; ++q) {
        const int piece = q * 64 + lane;
        const int ur = piece / PER_ROW, c8 = piece % PER_ROW;
        const uint4 v = *reinterpret_cast<const uint4*>(stage + ur * N + c8 * 8);
        if (row0 + ur < p.M) *reinterpret_cast<uint4*>(obase + (row0 + ur) * p.ldo + c8 * 8) = v;
      }
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t r = row0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (r >= p.M) continue;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          float v = acc[nb][e] + bias[nb];
          if (p.act == 1) v = fmaxf(v, 0.0f);
          else if (p.act == 2) v = v > 0.0f ? v : v * p.slope;
          static_cast<float*>(p.out)[r * p.ldo + nb * 32 + lr] = v * rs[e];
        }
      }
    }
    tile = next;
  }
}

template <int KT, int NB>
int launch_bf16(const RowsBf16Args& a, hipStream_t st) {
  // weight image + a 32 x N bfloat16 output area per wave
  constexpr size_t lds = static_cast<size_t>(KT) * 3 * NB * 32 * 32 + 8 * 32 * NB * 32 * 2;
  static_assert(lds <= 160 * 1024, "weight image and output areas do not fit LDS");
  auto kern = gemm_rows_bf16_kernel<KT, NB>;
  SPMI_RAISE_LDS_LIMIT(kern, lds);
  const int64_t n_tiles = (a.M + 31) / 32;
  const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
  int64_t blocks = (n_tiles + 7) / 8;
  if (blocks > 256 * per_cu) blocks = 256 * per_cu;
  kern<<<static_cast<unsigned>(blocks), 512, lds, st>>>(a);
  SPMI_CHECK_LAUNCH("gemm_rows_bf16");
  return SPMI_OK;
}

template <int KT>
int launch_bf16_n(const RowsBf16Args& a, int64_t N, hipStream_t st) {
  switch (N) {
    case 32: return launch_bf16<KT, 1>(a, st);
    case 64: return launch_bf16<KT, 2>(a, st);
    case 96: return launch_bf16<KT, 3>(a, st);
    default: return launch_bf16<KT, 4>(a, st);
  }
}

// Columns per workgroup: the whole weight up to 128 columns, slices of 128 (K <= 128) or 64
// (K = 256, 384) beyond: what fits LDS as three planes.  K = 512 would leave 32 columns per
// workgroup (built and measured: level with the library GEMM, 3.7 ms at 1M x 512 x 512); it
// runs as two K = 256 launches instead, the second adding to the first one's output.
int slice_width(int64_t K, int64_t N) {
  if (K == 256 || K == 384) return N % 64 == 0 ? 64 : 0;
  if (N <= 128) return (N % 32 == 0) ? static_cast<int>(N) : 0;
  return N % 128 == 0 ? 128 : 0;
}

template <int KT, int KC = 1>
int launch_n(RowsArgs& a, int64_t K, int64_t N, hipStream_t st) {
  const int w = slice_width(K, N);
  a.n_total = static_cast<int>(N);
  a.slices = static_cast<int>(N / w);
  if constexpr (KC > 1) {
    return launch<KT, 2, KC>(a, st);
  } else {
    switch (w) {
      case 32: return launch<KT, 1>(a, st);
      case 64: return launch<KT, 2>(a, st);
      case 96: return launch<KT, 3>(a, st);
      default: return launch<KT, 4>(a, st);
    }
  }
}

}  // namespace

extern "C" {

int spmi_gemm_rows_supported(int64_t K, int64_t N) {
  if (N < 32 || N > 4096) return 0;
  if (K == 512) return slice_width(256, N) != 0;
  if (!(K == 32 || K == 64 || K == 128 || K == 256 || K == 384)) return 0;
  return slice_width(K, N) != 0;
}

int spmi_gemm_rows_bf16_supported(int64_t K, int64_t N) {
  return (K == 32 || K == 64 || K == 128) && N >= 32 && N <= 128 && N % 32 == 0;
}

int spmi_gemm_rows_bf16(const void* x, int64_t ldx, int64_t M, int64_t K, const void* w_pieces, int64_t N,
                        const float* bias, const float* row_scale, int act, float slope, void* out,
                        int64_t ldo, int out_bf16, void* stream) {
  SPMI_REQUIRE(spmi_gemm_rows_bf16_supported(K, N), SPMI_ERR_UNSUPPORTED,
               "spmi_gemm_rows_bf16: K must be 32, 64 or 128 and N 32..128 in steps of 32 (K=%lld N=%lld)",
               (long long)K, (long long)N);
  SPMI_REQUIRE(M >= 0 && M < (int64_t{1} << 31), SPMI_ERR_INVALID_ARG, "spmi_gemm_rows_bf16: bad M=%lld",
               (long long)M);
  SPMI_REQUIRE(act >= 0 && act <= 2, SPMI_ERR_INVALID_ARG, "spmi_gemm_rows_bf16: bad act %d", act);
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(x && w_pieces && out && ldx >= K && ldx % 8 == 0 && ldo >= N, SPMI_ERR_INVALID_ARG,
               "spmi_gemm_rows_bf16: null pointer or bad leading dimension (ldx a multiple of 8)");
  SPMI_REQUIRE(!out_bf16 || (ldo % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0), SPMI_ERR_INVALID_ARG,
               "spmi_gemm_rows_bf16: a bfloat16 result needs ldo a multiple of 8 and a 16-byte aligned pointer");
  SPMI_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15u) == 0 && (reinterpret_cast<uintptr_t>(w_pieces) & 15u) == 0,
               SPMI_ERR_INVALID_ARG, "spmi_gemm_rows_bf16: x and w_pieces must be 16-byte aligned");
  RowsBf16Args a{};
  a.x = static_cast<const uint16_t*>(x); a.ldx = ldx; a.M = M; a.wp = static_cast<const uint4*>(w_pieces);
  a.bias = bias; a.row_scale = row_scale; a.act = act; a.slope = slope; a.out = out; a.ldo = ldo;
  a.out_bf16 = out_bf16;
  hipStream_t st = spmi::as_stream(stream);
  if (K == 32) return launch_bf16_n<2>(a, N, st);
  if (K == 64) return launch_bf16_n<4>(a, N, st);
  return launch_bf16_n<8>(a, N, st);
}

int spmi_gemm_rows(const float* x, int64_t ldx, int64_t M, int64_t K, const void* w_planes, int64_t N,
                   const float* bias, const float* row_scale, int act, float slope, float* out,
                   int64_t ldo, void* stream) {
  SPMI_REQUIRE(spmi_gemm_rows_supported(K, N), SPMI_ERR_UNSUPPORTED,
               "spmi_gemm_rows: K must be 32, 64, 128 (N = 32..128 in steps of 32, or a multiple of 128) "
               "or 256, 384, 512 (N a multiple of 64) (K=%lld N=%lld)",
               (long long)K, (long long)N);
  SPMI_REQUIRE(M >= 0 && M < (int64_t{1} << 31), SPMI_ERR_INVALID_ARG, "spmi_gemm_rows: bad M=%lld",
               (long long)M);
  SPMI_REQUIRE(act >= 0 && act <= 2, SPMI_ERR_INVALID_ARG, "spmi_gemm_rows: bad act %d", act);
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(x && w_planes && out && ldx >= K && ldx % 4 == 0 && ldo >= N, SPMI_ERR_INVALID_ARG,
               "spmi_gemm_rows: null pointer or bad leading dimension");
  SPMI_REQUIRE(ldx < (int64_t{1} << 24) && ldo < (int64_t{1} << 24), SPMI_ERR_UNSUPPORTED,
               "spmi_gemm_rows: leading dimension too large");
  SPMI_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15u) == 0 &&
                   (reinterpret_cast<uintptr_t>(w_planes) & 15u) == 0,
               SPMI_ERR_INVALID_ARG, "spmi_gemm_rows: x and w_planes must be 16-byte aligned");
  RowsArgs a{};
  a.x = x; a.ldx = ldx; a.M = M; a.wp = static_cast<const uint4*>(w_planes); a.bias = bias; a.row_scale = row_scale;
  a.act = act; a.slope = slope; a.out = out; a.ldo = ldo;
  hipStream_t st = spmi::as_stream(stream);
  if (K == 32) return launch_n<2>(a, K, N, st);
  if (K == 64) return launch_n<4>(a, K, N, st);
  if (K == 128) return launch_n<8>(a, K, N, st);
  if (K == 256) return launch_n<8, 2>(a, K, N, st);  // wider rows pass in chunks of 128 columns
  if (K == 384) return launch_n<8, 3>(a, K, N, st);
  // K = 512: columns 0..255 first, plain sums; then columns 256..511 on top of them, with the
  // bias, the activation and the row scale.  The image is [K-step][plane][k half][N][8], so
  // the second half's K-steps start 16 * 6 * N chunks of 16 bytes in.
  RowsArgs lo = a;
  lo.bias = nullptr; lo.row_scale = nullptr; lo.act = 0;
  if (int rc = launch_n<8, 2>(lo, 256, N, st)) return rc;
  RowsArgs hi = a;
  hi.x = x + 256;
  hi.wp = a.wp + 16 * 6 * N;
  hi.accumulate = 1;
  return launch_n<8, 2>(hi, 256, N, st);
}

}  // extern "C"
