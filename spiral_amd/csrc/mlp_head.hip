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
// Mapping: a wave owns 64 rows, one per lane.  The input arrives in 32-column chunks
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

size_t lds_bytes(int64_t I, int o1p, int rest_floats) {
  return sizeof(float) * (static_cast<size_t>(I) * o1p + ((rest_floats + 3) & ~3) +
                          static_cast<size_t>(kWaves) * kWaveFloats);
}

template <int O1P>
int launch(const MlpArgs& a, hipStream_t st) {
  auto kernel = mlp_head_kernel<O1P>;
  const size_t lds = lds_bytes(a.I, O1P, a.rest_floats);
  static size_t configured = 0;  // the dynamic-LDS limit is raised when a larger size shows up
  if (lds > configured) {
    SPMI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured = lds;
  }
  const int64_t rows_per_block = 64 * kWaves;
  const unsigned grid = static_cast<unsigned>((a.M + rows_per_block - 1) / rows_per_block);
  kernel<<<grid, 64 * kWaves, lds, st>>>(a);
  SPMI_CHECK_LAUNCH("mlp_head");
  return SPMI_OK;
}

}  // namespace

extern "C" {

int spmi_mlp_head_supported(int64_t I, int32_t out1, int32_t n_rest, const int32_t* rest_out) {
  if (I < kChunk || I % kChunk || out1 < 1 || out1 > kMaxWidth || n_rest < 0 || n_rest > kMaxRest)
    return 0;
  const int o1p = out1 <= 8 ? 8 : (out1 <= 16 ? 16 : 32);
  if (I * o1p > kMaxW1) return 0;
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
  hipStream_t st = spmi::as_stream(stream);
  if (out1 <= 8) return launch<8>(a, st);
  if (out1 <= 16) return launch<16>(a, st);
  return launch<32>(a, st);
}

}  // extern "C"
