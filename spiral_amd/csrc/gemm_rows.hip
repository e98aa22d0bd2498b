// Y = act(X W^T + b) for a tall X [M, K] and a small weight [N, K]: the dense projection of
// GraphConv / SGConv / SAGEConv (`torch.matmul(feat, weight)` inside dgl GraphConv.forward,
// call sites spiro/dtdg/models/encoder/implicitTimeEncoder/staticGraphEncoder.py:113,116) and
// of the decoders' Linear layers (spiro/dtdg/models/decoder/torch/simpleDecoder.py:43-46).
//
// fp32 in, fp32 out, on the fp16 matrix cores with the three-plane operand format of
// cell_split.hip (p0 = fp16(v), p1 = fp16(v/64), p2 = fp16((v - p0)*64); x*w = p0*p0 + p1*p2 +
// p2*p1, fp32 accumulation): sums land closer to float64 than an fp32 FMA chain.  The rows of
// X may hold any finite fp32 value: a wave has the whole row in registers before it starts
// on it, so every row is multiplied by a power of two that puts its largest magnitude in
// [2^13, 2^14) (exact), and the accumulators by the inverse power at the end.  Raw node
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
//  * a wave owns 32 rows x all N columns and keeps the whole K of its rows in registers
//    (K/2 VGPRs); as soon as the slice of K-step t is split, the same registers receive the
//    K-step t of the wave's next tile, so 16 KB per wave (128 KB per CU) are in flight while
//    the matrix cores work;
//  * waves are independent after the weight load: no barrier in the loop;
//  * wide outputs (N > 128) and K = 256, 384 are cut into column slices of 128 (64 for
//    K = 256, 384) that fit LDS: the workgroups that hold the slices of one row range run on one XCD
//    (blockIdx -> XCD is round-robin) and walk the same tiles, so the rows they all read
//    come from HBM once and from that XCD's L2 afterwards.

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
};

__device__ __forceinline__ void split8(const float4& lo, const float4& hi, f16x8& p0, f16x8& p1,
                                       f16x8& p2) {
  const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const _Float16 h = static_cast<_Float16>(v[q]);
    p0[q] = h;
    p1[q] = static_cast<_Float16>(v[q] * 0x1p-6f);
    p2[q] = static_cast<_Float16>((v[q] - static_cast<float>(h)) * 0x1p6f);
  }
}

template <int KT, int NB>
__global__ __launch_bounds__(KT > 8 ? 256 : 512, KT > 8 ? 1 : 2) void gemm_rows_kernel(RowsArgs p) {
  // K = 256 keeps 128 registers of row data per lane: four waves per workgroup, one per SIMD,
  // with the whole register file; the smaller shapes run eight waves (two per SIMD).
  constexpr int NW = KT > 8 ? 4 : 8;
  constexpr int N = NB * 32;
  constexpr int PLANE = N * 32;           // bytes of one plane of one K-step
  constexpr int KSTEP = 3 * PLANE;        // bytes of one K-step
  constexpr int B16 = KT * KSTEP / 16;    // 16-byte chunks of the weight image
  extern __shared__ uint4 smem4[];
  const char* smem = reinterpret_cast<const char*>(smem4);

  // Workgroup b runs on XCD b % 8; the column slices of one row group sit on one XCD.
  const int b = blockIdx.x;
  const int slice = (b >> 3) % p.slices;
  const int group = (b & 7) + 8 * (b / (8 * p.slices));
  const int n_groups = gridDim.x / p.slices;
  const int col0 = slice * N;

  const int tid = threadIdx.x;
  // image [K-step][plane][k half][n_total][8]: 16-byte chunk index ((t*3+q)*2+h) * n_total + n
  for (int c = tid; c < B16; c += NW * 64) {
    const int n = c % N, rest = c / N;
    smem4[c] = p.wp[static_cast<size_t>(rest) * p.n_total + col0 + n];
  }
  __syncthreads();

  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31;
  const int lh = lane >> 5;
  const int64_t n_tiles = (p.M + 31) / 32;
  const int64_t stride = static_cast<int64_t>(n_groups) * NW;
  int64_t tile = static_cast<int64_t>(group) * NW + wave;
  if (tile >= n_tiles) return;

  float bias[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bias[nb] = p.bias ? p.bias[col0 + nb * 32 + lr] : 0.0f;

  // Rows past M are clamped to the last row: computed, never stored.
  auto row_ptr = [&](int64_t t) {
    int64_t r = t * 32 + lr;
    r = r < p.M ? r : p.M - 1;
    return p.x + r * p.ldx + 8 * lh;
  };

  float4 a[KT][2];
  {
    const float* src = row_ptr(tile);
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      a[t][0] = *reinterpret_cast<const float4*>(src + 16 * t);
      a[t][1] = *reinterpret_cast<const float4*>(src + 16 * t + 4);
    }
  }
  const int b_rd = (lh * N + lr) * 16;  // [plane][k half][n][8]: 512 contiguous bytes per half

  while (tile < n_tiles) {
    const int64_t next = tile + stride;
    // the last iteration re-reads its own tile: valid memory, nothing is done with it
    const float* src = row_ptr(next < n_tiles ? next : tile);

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nb][e] = 0.0f;

    // Power-of-two scale of this lane's row (the lane pair l, l ^ 32 holds one row): the
    // largest magnitude goes to [2^13, 2^14), a zero or non-finite row is left alone.
    float up, down;
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

    // the row scales of this tile, fetched under the matrix work (rows past M: clamped)
    float rs[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int64_t r = tile * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      r = r < p.M ? r : p.M - 1;
      rs[e] = p.row_scale ? p.row_scale[r] : 1.0f;
    }

#pragma unroll
    for (int t = 0; t < KT; ++t) {
      f16x8 p0, p1, p2;
      {
        float4 lo = a[t][0], hi = a[t][1];
        lo.x *= up; lo.y *= up; lo.z *= up; lo.w *= up;
        hi.x *= up; hi.y *= up; hi.z *= up; hi.w *= up;
        split8(lo, hi, p0, p1, p2);
      }
      a[t][0] = *reinterpret_cast<const float4*>(src + 16 * t);
      a[t][1] = *reinterpret_cast<const float4*>(src + 16 * t + 4);
      // Keep the K-steps apart: left alone, the scheduler hoists every reload to the top of
      // the tile and then needs a second copy of the whole row in registers.
      __builtin_amdgcn_sched_barrier(0);
      const char* sb = smem + t * KSTEP + b_rd;
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

    // C layout of the 32x32 MFMA: element e of lane l is row (e&3) + 8 (e>>2) + 4 (l>>5),
    // column l&31: a store covers two 128-byte row segments.
    const int64_t row0 = tile * 32;
    float* obase = p.out + row0 * p.ldo + col0;  // wave-uniform
    const int voff = 4 * lh * static_cast<int>(p.ldo) + lr;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ur = (e & 3) + 8 * (e >> 2);
      if (row0 + ur + 4 * lh >= p.M) continue;
      float* o = obase + ur * p.ldo;
      // the inverse scale of output row ur + 4 lh sits in lane ur + 4 lh (either half)
      const float d0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(down), ur));
      const float d1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(down), ur + 4));
      const float dn = lh ? d1 : d0;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float v = acc[nb][e] * dn + bias[nb];
        if (p.act == 1) v = fmaxf(v, 0.0f);
        else if (p.act == 2) v = v > 0.0f ? v : v * p.slope;
        o[voff + nb * 32] = v * rs[e];
      }
    }
    tile = next;
  }
}

template <int KT, int NB>
int launch(const RowsArgs& a, hipStream_t st) {
  constexpr size_t lds = static_cast<size_t>(KT) * 3 * NB * 32 * 32;
  static_assert(lds <= 160 * 1024, "weight image does not fit LDS");
  auto kern = gemm_rows_kernel<KT, NB>;
  static bool configured = false;
  if (!configured) {
    SPMI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured = true;
  }
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

// Columns per workgroup: the whole weight up to 128 columns, slices of 128 (K <= 128) or 64
// (K = 256, 384) beyond.
int slice_width(int64_t K, int64_t N) {
  if (K == 256 || K == 384) return N % 64 == 0 ? 64 : 0;
  if (N <= 128) return (N % 32 == 0) ? static_cast<int>(N) : 0;
  return N % 128 == 0 ? 128 : 0;
}

template <int KT>
int launch_n(RowsArgs& a, int64_t K, int64_t N, hipStream_t st) {
  const int w = slice_width(K, N);
  a.n_total = static_cast<int>(N);
  a.slices = static_cast<int>(N / w);
  if constexpr (KT > 8) {
    return launch<KT, 2>(a, st);
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
  if (!(K == 32 || K == 64 || K == 128 || K == 256 || K == 384) || N < 32 || N > 4096) return 0;
  return slice_width(K, N) != 0;
}

int spmi_gemm_rows(const float* x, int64_t ldx, int64_t M, int64_t K, const void* w_planes, int64_t N,
                   const float* bias, const float* row_scale, int act, float slope, float* out,
                   int64_t ldo, void* stream) {
  SPMI_REQUIRE(spmi_gemm_rows_supported(K, N), SPMI_ERR_UNSUPPORTED,
               "spmi_gemm_rows: K must be 32, 64, 128 (N = 32..128 in steps of 32, or a multiple of 128) "
               "or 256, 384 (N a multiple of 64) (K=%lld N=%lld)",
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
  if (K == 256) return launch_n<16>(a, K, N, st);
  return launch_n<24>(a, K, N, st);
}

}  // extern "C"
