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
// Mapping.  A workgroup owns 128 consecutive batch rows and ALL hidden units of them, so
// that the state rows it overwrites are read by no other workgroup.  Its waves form a
// 4 x WC grid: row group (32 rows) x column group; a wave keeps JSUB 32-unit column
// blocks times NACC accumulator sets in registers (GRU: r, z, n_x, n_h; r and z
// accumulate over the concatenated K = I + H, so x W_ih^T and h W_hh^T share
// accumulators wherever the cell only needs their sum).  With WC = 2 and JSUB = 2 a
// 128-unit state is one pass at 128 accumulator registers per wave, two waves per SIMD.
// K runs in steps of BK through a double-buffered LDS stage (one barrier per step): the
// A tile (x rows, then state rows) padded to an odd stride and the B tile
// (pre-transposed weights, [k][gate*JC + unit]) are both read conflict-free with
// ds_read_b32, one A value feeding NG * JSUB MFMAs.  The next step's global loads are
// issued before the MFMAs of the current one and land in the other LDS buffer after them.
// MFMA-bound: 2*N*(I+H)*G*H flops against the 157 TFLOP/s fp32 matrix peak.

#include "common.h"

namespace {

constexpr int kRows = 128;  // batch rows per workgroup (4 row groups of 32)

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
  const float e = __expf(2.0f * (v > 15.0f ? 15.0f : v));  // saturates beyond ~9; NaN stays NaN
  return (e - 1.0f) * __frcp_rn(e + 1.0f);
}

// MFMAs of one K-step for one wave: BK/2 k-pairs, each A value feeding NG * JSUB products.
// NC = columns of the workgroup's B tile, JC = hidden units per pass.
template <int CELL, int JSUB, int BK, int NC, int JC, bool HPHASE>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[JSUB][Cell<CELL>::NACC], const float* ap,
                                          const float* bp) {
  using C = Cell<CELL>;
  constexpr int NB = C::NG * JSUB;
  // Operands of k-pair kk+1 are read from LDS while the MFMAs of k-pair kk issue, so the
  // LDS latency is covered by this wave's own matrix work, not only by the other wave.
  float a_cur = ap[0], a_nxt = 0.0f;
  float b_cur[NB], b_nxt[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) b_cur[i] = bp[(i / JSUB) * JC + (i % JSUB) * 32];
#pragma unroll
  for (int kk = 0; kk < BK / 2; ++kk) {
    if (kk + 1 < BK / 2) {
      a_nxt = ap[(kk + 1) * 2];
#pragma unroll
      for (int i = 0; i < NB; ++i)
        b_nxt[i] = bp[(kk + 1) * 2 * NC + (i / JSUB) * JC + (i % JSUB) * 32];
    }
    // keep the reads above the MFMAs: left alone, the scheduler sinks each read to just
    // before its use (lowest register pressure) and the wave then waits on LDS every two MFMAs
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int g = i / JSUB, sub = i % JSUB;
      const int ai = HPHASE ? C::acc_h[g] : C::acc_x[g];
      acc[sub][ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[i], acc[sub][ai], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    a_cur = a_nxt;
#pragma unroll
    for (int i = 0; i < NB; ++i) b_cur[i] = b_nxt[i];
  }
}

// Registers holding the next K-step's tiles while the current one is multiplied.
template <int APT, int BPT>
struct TileRegs {
  float4 a[APT];
  float4 b[BPT];
};

template <int APT, int BPT, int B4, int THREADS>
__device__ __forceinline__ void load_tile(TileRegs<APT, BPT>& r, const float* const (&asrc)[APT],
                                          const float* bsrc, const int (&boff)[BPT], int k0,
                                          int GH, int tid) {
#pragma unroll
  for (int i = 0; i < APT; ++i) r.a[i] = *reinterpret_cast<const float4*>(asrc[i] + k0);
  const float* b = bsrc + static_cast<int64_t>(k0) * GH;
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    if ((i + 1) * THREADS <= B4 || tid + i * THREADS < B4)
      r.b[i] = *reinterpret_cast<const float4*>(b + boff[i]);
  }
}

template <int APT, int BPT, int B4, int THREADS>
__device__ __forceinline__ void store_tile(const TileRegs<APT, BPT>& r, float* sA, float* sB,
                                           const int (&a_dst)[APT], int tid) {
#pragma unroll
  for (int i = 0; i < APT; ++i) {
    float* d = sA + a_dst[i];
    d[0] = r.a[i].x; d[1] = r.a[i].y; d[2] = r.a[i].z; d[3] = r.a[i].w;
  }
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    if ((i + 1) * THREADS <= B4 || tid + i * THREADS < B4)
      *reinterpret_cast<float4*>(sB + (tid + i * THREADS) * 4) = r.b[i];
  }
}

template <int CELL, int JSUB, int WC, int BK>
__global__ __launch_bounds__(256 * WC, 2) void fused_cell_kernel(FusedArgs p) {
  using C = Cell<CELL>;
  constexpr int THREADS = 256 * WC;
  constexpr int NG = C::NG, NACC = C::NACC;
  constexpr int JW = JSUB * 32;          // hidden units per wave
  constexpr int JC = JW * WC;            // hidden units per pass
  constexpr int NC = NG * JC;            // B-tile columns
  constexpr int AS = BK + 1;             // padded A-tile row stride
  constexpr int A4 = kRows * BK / 4;     // float4 per A tile
  constexpr int B4 = BK * NC / 4;        // float4 per B tile
  constexpr int APT = A4 / THREADS;
  constexpr int BPT = (B4 + THREADS - 1) / THREADS;
  constexpr int A_FLOATS = kRows * AS, B_FLOATS = BK * NC;
  static_assert(A4 % THREADS == 0, "A tile must divide over the threads");

  // Two stages of [A tile | B tile].  Addresses are formed as offsets from the one LDS
  // array so that every access stays a ds_read/ds_write (a runtime-selected pointer
  // would degrade to flat loads, whose waits also cover the global prefetch).
  constexpr int STAGE = A_FLOATS + B_FLOATS;
  extern __shared__ float4 smem4[];
  float* smem = reinterpret_cast<float*>(smem4);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int rg = wave & 3;   // row group of this wave
  const int cg = wave >> 2;  // column group of this wave
  const int lr = lane & 31;  // row (A) / column (B, C) inside a 32-block
  const int lh = lane >> 5;  // which of the two k of an MFMA step
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * kRows;
  const int H = p.H;
  const int GH = NG * H;

  // A-tile loads: float4 number idx covers row idx / (BK/4), floats (idx % (BK/4)) * 4
  const float* ax[APT];
  const float* ah[APT];
  int a_dst[APT];
#pragma unroll
  for (int i = 0; i < APT; ++i) {
    const int idx = tid + i * THREADS;
    const int trow = idx / (BK / 4), c4 = (idx % (BK / 4)) * 4;
    int64_t r = row0 + trow;
    if (r >= p.M) r = p.M - 1;
    const int64_t mr = p.ids ? p.ids[r] : r;
    ax[i] = p.x + r * p.ldx + c4;
    ah[i] = p.h_mem + mr * H + c4;
    a_dst[i] = trow * AS + c4;
  }
  // B-tile loads: float4 number idx covers k = idx / (NC/4), columns (idx % (NC/4)) * 4
  int boff[BPT];
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int idx = tid + i * THREADS;
    const int k = idx / (NC / 4);
    const int col = (idx % (NC / 4)) * 4;
    boff[i] = k * GH + (col / JC) * H + (col % JC);
  }

  const int tiles_x = p.I / BK;
  const int tiles_h = H / BK;
  const int a_off = (rg * 32 + lr) * AS + lh;
  const int b_off = lh * NC + cg * JW + lr;

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
    TileRegs<APT, BPT> regs;
    // Two plain loops (x phase, then h phase) rather than one loop with a phase branch
    // around the MFMAs: the accumulators then stay in place across iterations.
    __syncthreads();  // the previous pass has finished reading both buffers
    load_tile<APT, BPT, B4, THREADS>(regs, ax, bx, boff, 0, GH, tid);
    store_tile<APT, BPT, B4, THREADS>(regs, smem, smem + A_FLOATS, a_dst, tid);
    __syncthreads();
    int buf = 0;
    for (int t = 0; t < tiles_x; ++t) {
      if (t + 1 < tiles_x) {
        load_tile<APT, BPT, B4, THREADS>(regs, ax, bx, boff, (t + 1) * BK, GH, tid);
      } else {
        load_tile<APT, BPT, B4, THREADS>(regs, ah, bh, boff, 0, GH, tid);
      }
      mfma_step<CELL, JSUB, BK, NC, JC, false>(acc, smem + buf * STAGE + a_off,
                                               smem + buf * STAGE + A_FLOATS + b_off);
      store_tile<APT, BPT, B4, THREADS>(regs, smem + (buf ^ 1) * STAGE,
                                        smem + (buf ^ 1) * STAGE + A_FLOATS, a_dst, tid);
      __syncthreads();
      buf ^= 1;
    }
    for (int t = 0; t < tiles_h; ++t) {
      const bool more = t + 1 < tiles_h;
      if (more) load_tile<APT, BPT, B4, THREADS>(regs, ah, bh, boff, (t + 1) * BK, GH, tid);
      mfma_step<CELL, JSUB, BK, NC, JC, true>(acc, smem + buf * STAGE + a_off,
                                              smem + buf * STAGE + A_FLOATS + b_off);
      if (more)
        store_tile<APT, BPT, B4, THREADS>(regs, smem + (buf ^ 1) * STAGE,
                                          smem + (buf ^ 1) * STAGE + A_FLOATS, a_dst, tid);
      __syncthreads();
      buf ^= 1;
    }

    // ---- epilogue: gates, state update, stores ------------------------------------------
#pragma unroll
    for (int s = 0; s < JSUB; ++s) {
      const int j = j0 + cg * JW + s * 32 + lr;
      float bi[NG], bhh[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        bi[g] = p.b_ih ? p.b_ih[g * H + j] : 0.0f;
        bhh[g] = p.b_hh ? p.b_hh[g * H + j] : 0.0f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = row0 + rg * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row >= p.M) continue;
        const int64_t mrow = (p.ids ? p.ids[row] : row) * H + j;
        const int64_t orow = row * H + j;
        if constexpr (CELL == CELL_GRU) {
          const float h = p.h_mem[mrow];
          const float r = sigmoidf_(acc[s][0][e] + (bi[0] + bhh[0]));
          const float z = sigmoidf_(acc[s][1][e] + (bi[1] + bhh[1]));
          const float hn = acc[s][3][e] + bhh[2];
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
          const float ig = sigmoidf_(acc[s][0][e] + (bi[0] + bhh[0]));
          const float fg = sigmoidf_(acc[s][1][e] + (bi[1] + bhh[1]));
          const float gg = tanhf_(acc[s][2][e] + (bi[2] + bhh[2]));
          const float og = sigmoidf_(acc[s][3][e] + (bi[3] + bhh[3]));
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
          const float hy = tanhf_(acc[s][0][e] + (bi[0] + bhh[0]));
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

template <int CELL, int JSUB, int WC, int BK>
int launch_config(const FusedArgs& a, hipStream_t st) {
  constexpr int NC = Cell<CELL>::NG * JSUB * 32 * WC;
  constexpr size_t lds = 2 * sizeof(float) * (kRows * (BK + 1) + BK * NC);
  static_assert(lds <= 160 * 1024, "LDS stage does not fit");
  auto kernel = fused_cell_kernel<CELL, JSUB, WC, BK>;
  SPMI_RAISE_LDS_LIMIT(kernel, lds);
  const unsigned grid = static_cast<unsigned>((a.M + kRows - 1) / kRows);
  kernel<<<grid, 256 * WC, lds, st>>>(a);
  SPMI_CHECK_LAUNCH("fused_cell");
  return SPMI_OK;
}

// K-step 16: measured equal or slightly ahead of 32 on every cell (the shorter step keeps
// the staged tiles small: 42 KB for the GRU at 128 hidden units), and it lets the LSTM's
// four gate column blocks fit the double-buffered stage too.
template <int CELL>
int launch_fused(const FusedArgs& a, hipStream_t st) {
  constexpr int BK = 16;
  if (a.H % 128 == 0) return launch_config<CELL, 2, 2, BK>(a, st);
  if (a.H % 64 == 0) return launch_config<CELL, 1, 2, BK>(a, st);
  return launch_config<CELL, 1, 1, BK>(a, st);
}

}  // namespace

extern "C" {

int spmi_cell_fused_supported(int64_t I, int64_t H) {
  return (I >= 32 && I % 32 == 0 && H >= 32 && H % 32 == 0) ? 1 : 0;
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
               "spmi_cell_fused_fwd: needs I %% 32 == 0 and H %% 32 == 0 (I=%lld H=%lld)",
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
