// One recurrent layer step (sequence length 1, batch = nodes) as ONE kernel for gfx950:
// both gate GEMMs on the fp32 matrix cores with the gate math, the node-memory read and
// the in-place write-back fused into the epilogue.
//
// Replaces torch.nn.GRU / LSTM / RNN as driven by SequentialDecoder.forward / LSTM.forward
// (spiro/dtdg/models/decoder/torch/sequentialDecoder/implicitTimeModels.py:144-158,
// :214-230) together with NodeMemory.get_memory / update_memory (:48-54).
//
// Why one kernel: unfused, the two GEMMs write and the gate kernel re-reads
// 2 * N * G*H floats (3 GB per layer at N=1e6, H=128) for 0.5 GB of useful state; here the
// pre-activations never leave the accumulators.  Arithmetic is exact fp32
// (v_mfma_f32_32x32x2_f32: an fp32 FMA chain), not a reduced-precision format.
//
// Mapping.  A workgroup (4 waves) owns 128 consecutive batch rows and ALL hidden units
// of them, so that the state rows it overwrites are read by no other workgroup.
// Wave w holds rows 32w..32w+31; per pass it keeps JSUB 32-unit column blocks times NACC
// accumulator sets in registers (GRU: r, z, n_x, n_h; r and z accumulate over the
// concatenated K = I + H, so x W_ih^T and h W_hh^T share accumulators wherever the cell
// only needs their sum).  K runs in steps of 16 through LDS: the A tile (x rows, then
// state rows) padded to a 17-float stride and the B tile (pre-transposed weights,
// [k][gate*JC + unit]) are both read conflict-free with ds_read_b32, one A value feeding
// NG * JSUB MFMAs.  The next K-step's global loads are issued before the MFMAs of the
// current one.  MFMA-bound: 2*N*(I+H)*G*H flops against the 157 TFLOP/s fp32 matrix peak.

#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRows = 128;   // batch rows per workgroup
constexpr int kBK = 16;      // K-step
constexpr int kAStride = 17; // padded A-tile row stride (floats)

using f32x16 = float __attribute__((ext_vector_type(16)));

enum { CELL_GRU = 0, CELL_LSTM = 1, CELL_RNN = 2 };

template <int CELL>
struct Cell;
template <>
struct Cell<CELL_GRU> {
  static constexpr int NG = 3, NACC = 4;
  // accumulator fed by gate g in the x phase / in the h phase
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

struct FusedArgs {
  const float* x;
  int64_t ldx;
  int I;
  const float* wt_ih;  // [I][NG*H]  (weight_ih transposed)
  const float* wt_hh;  // [H][NG*H]
  const float* b_ih;   // nullable [NG*H]
  const float* b_hh;
  float* h_mem;        // [n_mem][H]
  float* c_mem;        // LSTM only
  const int64_t* ids;  // nullable
  int64_t M;
  int H;
  float* h_out;        // [M][H]
  float* c_out;        // LSTM only
  int mem_write;
  float* saved;        // nullable
};

// Gate non-linearities on the hardware exp2/rcp units (v_exp_f32, v_rcp_f32: about 1 ulp
// each).  Absolute error of both functions stays below 3e-7, which is what matters for a
// bounded gate value; the library expf/tanhf cost roughly ten times the instructions and
// the epilogue is the only VALU-heavy part of the kernel.
__device__ __forceinline__ float sigmoidf_(float v) {
  return __frcp_rn(1.0f + __expf(-v));
}
__device__ __forceinline__ float tanhf_(float v) {
  const float e = __expf(2.0f * fminf(v, 15.0f));  // tanh saturates to 1 in fp32 beyond ~9
  return (e - 1.0f) * __frcp_rn(e + 1.0f);
}

// MFMAs of one K-step for one wave: 8 k-pairs, each A value feeding NG * JSUB products.
template <int CELL, int JSUB, bool HPHASE>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[JSUB][Cell<CELL>::NACC], const float* ap,
                                          const float* bp) {
  using C = Cell<CELL>;
  constexpr int JC = JSUB * 32, NC = C::NG * JC;
#pragma unroll
  for (int kk = 0; kk < kBK / 2; ++kk) {
    const float a = ap[kk * 2];
#pragma unroll
    for (int g = 0; g < C::NG; ++g) {
      const int ai = HPHASE ? C::acc_h[g] : C::acc_x[g];
#pragma unroll
      for (int s = 0; s < JSUB; ++s) {
        const float b = bp[kk * 2 * NC + g * JC + s * 32];
        acc[s][ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[s][ai], 0, 0, 0);
      }
    }
  }
}

// Registers holding the next K-step's tiles while the current one is multiplied.
template <int BPT>
struct TileRegs {
  float4 a[2];
  float4 b[BPT];
};

// Global -> registers.  a0/a1: this thread's two A-row pointers (already offset to its
// 4-float column group); bsrc: weight matrix of the phase, advanced to the pass's columns.
template <int BPT, int B4>
__device__ __forceinline__ void load_tile(TileRegs<BPT>& r, const float* a0, const float* a1,
                                          const float* bsrc, const int (&boff)[BPT], int k0,
                                          int GH, int tid) {
  r.a[0] = *reinterpret_cast<const float4*>(a0 + k0);
  r.a[1] = *reinterpret_cast<const float4*>(a1 + k0);
  const float* b = bsrc + static_cast<int64_t>(k0) * GH;
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    if ((i + 1) * kThreads <= B4 || tid + i * kThreads < B4)
      r.b[i] = *reinterpret_cast<const float4*>(b + boff[i]);
  }
}

template <int BPT, int B4>
__device__ __forceinline__ void store_tile(const TileRegs<BPT>& r, float* sA, float* sB, int a_dst0,
                                           int a_dst1, int tid) {
  float* d0 = sA + a_dst0;
  d0[0] = r.a[0].x; d0[1] = r.a[0].y; d0[2] = r.a[0].z; d0[3] = r.a[0].w;
  float* d1 = sA + a_dst1;
  d1[0] = r.a[1].x; d1[1] = r.a[1].y; d1[2] = r.a[1].z; d1[3] = r.a[1].w;
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    if ((i + 1) * kThreads <= B4 || tid + i * kThreads < B4)
      *reinterpret_cast<float4*>(sB + (tid + i * kThreads) * 4) = r.b[i];
  }
}

template <int CELL, int JSUB>
__global__ __launch_bounds__(kThreads, JSUB <= 2 ? 2 : 1) void fused_cell_kernel(FusedArgs p) {
  using C = Cell<CELL>;
  constexpr int NG = C::NG, NACC = C::NACC;
  constexpr int JC = JSUB * 32;   // hidden units per pass
  constexpr int NC = NG * JC;     // B-tile columns
  constexpr int B4 = kBK * NC / 4;  // float4 per B tile
  constexpr int BPT = (B4 + kThreads - 1) / kThreads;

  __shared__ float sA[kRows * kAStride];
  __shared__ float sB[kBK * NC];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lr = lane & 31;  // row (A) / column (B, C) inside a 32-block
  const int lh = lane >> 5;  // which of the two k of an MFMA step
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * kRows;
  const int H = p.H;
  const int GH = NG * H;

  // A-tile loads: 128 rows x 16 floats = 512 float4, two per thread (rows tid/4, tid/4+64)
  const float* ax[2];
  const float* ah[2];
  int a_dst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int trow = (tid >> 2) + i * 64;
    int64_t r = row0 + trow;
    if (r >= p.M) r = p.M - 1;
    const int64_t mr = p.ids ? p.ids[r] : r;
    ax[i] = p.x + r * p.ldx + (tid & 3) * 4;
    ah[i] = p.h_mem + mr * H + (tid & 3) * 4;
    a_dst[i] = trow * kAStride + (tid & 3) * 4;
  }
  // B-tile loads: float4 number idx covers k = idx / (NC/4), columns (idx % (NC/4)) * 4
  int boff[BPT];
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int idx = tid + i * kThreads;
    const int k = idx / (NC / 4);
    const int col = (idx % (NC / 4)) * 4;
    boff[i] = k * GH + (col / JC) * H + (col % JC);
  }

  const int tiles_x = p.I / kBK;
  const int tiles_h = H / kBK;
  const float* ap = sA + (wave * 32 + lr) * kAStride + lh;
  const float* bp = sB + lh * NC + lr;

  for (int j0 = 0; j0 < H; j0 += JC) {
    f32x16 acc[JSUB][NACC];
#pragma unroll
    for (int s = 0; s < JSUB; ++s)
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][a][e] = 0.0f;

    const float* bx = p.wt_ih + j0;
    const float* bh = p.wt_hh + j0;
    TileRegs<BPT> regs;
    // Two plain loops (x phase, then h phase) rather than one loop with a phase branch
    // around the MFMAs: the accumulators then stay in place across iterations.
    load_tile<BPT, B4>(regs, ax[0], ax[1], bx, boff, 0, GH, tid);
    for (int t = 0; t < tiles_x; ++t) {
      __syncthreads();
      store_tile<BPT, B4>(regs, sA, sB, a_dst[0], a_dst[1], tid);
      __syncthreads();
      if (t + 1 < tiles_x) {
        load_tile<BPT, B4>(regs, ax[0], ax[1], bx, boff, (t + 1) * kBK, GH, tid);
      } else {
        load_tile<BPT, B4>(regs, ah[0], ah[1], bh, boff, 0, GH, tid);
      }
      mfma_step<CELL, JSUB, false>(acc, ap, bp);
    }
    for (int t = 0; t < tiles_h; ++t) {
      __syncthreads();
      store_tile<BPT, B4>(regs, sA, sB, a_dst[0], a_dst[1], tid);
      __syncthreads();
      if (t + 1 < tiles_h) load_tile<BPT, B4>(regs, ah[0], ah[1], bh, boff, (t + 1) * kBK, GH, tid);
      mfma_step<CELL, JSUB, true>(acc, ap, bp);
    }

    // ---- epilogue: gates, state update, stores ------------------------------------------
#pragma unroll
    for (int s = 0; s < JSUB; ++s) {
      const int j = j0 + s * 32 + lr;
      float bi[NG], bh[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        bi[g] = p.b_ih ? p.b_ih[g * H + j] : 0.0f;
        bh[g] = p.b_hh ? p.b_hh[g * H + j] : 0.0f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = row0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row >= p.M) continue;
        const int64_t mrow = (p.ids ? p.ids[row] : row) * H + j;
        const int64_t orow = row * H + j;
        if constexpr (CELL == CELL_GRU) {
          const float h = p.h_mem[mrow];
          const float r = sigmoidf_(acc[s][0][e] + (bi[0] + bh[0]));
          const float z = sigmoidf_(acc[s][1][e] + (bi[1] + bh[1]));
          const float hn = acc[s][3][e] + bh[2];
          const float n = tanhf_(acc[s][2][e] + bi[2] + r * hn);
          const float hy = n + z * (h - n);
          p.h_out[orow] = hy;
          if (p.saved) {
            float* sv = p.saved + row * 5 * H + j;
            sv[0] = r; sv[H] = z; sv[2 * H] = n; sv[3 * H] = hn; sv[4 * H] = h;
          }
          if (p.mem_write && JC == H) p.h_mem[mrow] = hy;
        } else if constexpr (CELL == CELL_LSTM) {
          const float c = p.c_mem[mrow];
          const float ig = sigmoidf_(acc[s][0][e] + (bi[0] + bh[0]));
          const float fg = sigmoidf_(acc[s][1][e] + (bi[1] + bh[1]));
          const float gg = tanhf_(acc[s][2][e] + (bi[2] + bh[2]));
          const float og = sigmoidf_(acc[s][3][e] + (bi[3] + bh[3]));
          const float cy = fg * c + ig * gg;
          const float tc = tanhf_(cy);
          const float hy = og * tc;
          p.h_out[orow] = hy;
          p.c_out[orow] = cy;
          if (p.saved) {
            float* sv = p.saved + row * 6 * H + j;
            sv[0] = ig; sv[H] = fg; sv[2 * H] = gg; sv[3 * H] = og; sv[4 * H] = c; sv[5 * H] = tc;
          }
          if (p.mem_write && JC == H) {
            p.h_mem[mrow] = hy;
            p.c_mem[mrow] = cy;
          }
        } else {
          const float hy = tanhf_(acc[s][0][e] + (bi[0] + bh[0]));
          p.h_out[orow] = hy;
          if (p.mem_write && JC == H) p.h_mem[mrow] = hy;
        }
      }
    }
  }

  // With several column passes the state rows stay readable until the last pass is
  // done; only then are they overwritten (this workgroup is the only one touching them).
  if (p.mem_write && JC != H) {
    __syncthreads();
    const int per_row = H / 4;
    for (int idx = tid; idx < kRows * per_row; idx += kThreads) {
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

template <int CELL>
int launch_fused(const FusedArgs& a, hipStream_t st) {
  const unsigned grid = static_cast<unsigned>((a.M + kRows - 1) / kRows);
  const int blocks32 = a.H / 32;
  // 64 hidden units per pass (128 accumulator registers) lets two workgroups share a CU,
  // so one workgroup's epilogue and loads run under the other's MFMAs; the price is that
  // the A tiles are read once per pass (from L2).
  if (blocks32 % 2 == 0) {
    fused_cell_kernel<CELL, 2><<<grid, kThreads, 0, st>>>(a);
  } else {
    fused_cell_kernel<CELL, 1><<<grid, kThreads, 0, st>>>(a);
  }
  SPMI_CHECK_LAUNCH("fused_cell");
  return SPMI_OK;
}

}  // namespace

extern "C" {

int spmi_cell_fused_supported(int64_t I, int64_t H) {
  return (I >= kBK && I % kBK == 0 && H >= 32 && H % 32 == 0) ? 1 : 0;
}

int spmi_cell_fused_fwd(int cell, const float* x, int64_t ldx, int64_t I, const float* wt_ih,
                        const float* wt_hh, const float* b_ih, const float* b_hh, float* h_mem,
                        float* c_mem, const int64_t* ids, int64_t M, int64_t H, float* h_out,
                        float* c_out, int mem_write, float* saved, void* stream) {
  SPMI_REQUIRE(cell >= CELL_GRU && cell <= CELL_RNN, SPMI_ERR_INVALID_ARG,
               "spmi_cell_fused_fwd: bad cell code %d", cell);
  SPMI_REQUIRE(M >= 0 && M < (int64_t{1} << 31), SPMI_ERR_INVALID_ARG,
               "spmi_cell_fused_fwd: bad M=%lld", (long long)M);
  SPMI_REQUIRE(spmi_cell_fused_supported(I, H), SPMI_ERR_UNSUPPORTED,
               "spmi_cell_fused_fwd: needs I %% 16 == 0 and H %% 32 == 0 (I=%lld H=%lld)",
               (long long)I, (long long)H);
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(x && wt_ih && wt_hh && h_mem && h_out && ldx >= I && ldx % 4 == 0,
               SPMI_ERR_INVALID_ARG, "spmi_cell_fused_fwd: null pointer or bad ldx");
  SPMI_REQUIRE(cell != CELL_LSTM || (c_mem && c_out), SPMI_ERR_INVALID_ARG,
               "spmi_cell_fused_fwd: LSTM needs c_mem and c_out");
  for (const void* q : {(const void*)x, (const void*)wt_ih, (const void*)wt_hh,
                        (const void*)h_mem, (const void*)h_out, (const void*)c_mem,
                        (const void*)c_out})
    SPMI_REQUIRE(!q || (reinterpret_cast<uintptr_t>(q) & 15u) == 0, SPMI_ERR_INVALID_ARG,
                 "spmi_cell_fused_fwd: pointers must be 16-byte aligned");
  FusedArgs a{};
  a.x = x; a.ldx = ldx; a.I = static_cast<int>(I);
  a.wt_ih = wt_ih; a.wt_hh = wt_hh; a.b_ih = b_ih; a.b_hh = b_hh;
  a.h_mem = h_mem; a.c_mem = c_mem; a.ids = ids; a.M = M; a.H = static_cast<int>(H);
  a.h_out = h_out; a.c_out = c_out; a.mem_write = mem_write; a.saved = saved;
  hipStream_t st = spmi::as_stream(stream);
  if (cell == CELL_GRU) return launch_fused<CELL_GRU>(a, st);
  if (cell == CELL_LSTM) return launch_fused<CELL_LSTM>(a, st);
  return launch_fused<CELL_RNN>(a, st);
}

}  // extern "C"
