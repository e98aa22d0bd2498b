// Recurrent-cell gate kernels for gfx950, sequence length 1, batch = nodes.
//
// They replace the pointwise half of torch.nn.GRU / LSTM / RNN as driven by
// SequentialDecoder.forward and LSTM.forward
// (spiro/dtdg/models/decoder/torch/sequentialDecoder/implicitTimeModels.py:144-158,
// :214-230), fused with the NodeMemory gather memory[:, ids, :] (:52-54) and the
// detached write-back memory[:, ids, :] = h (:48-50).  The gate GEMMs stay library
// GEMMs in the caller; these kernels read their two outputs once, apply biases,
// sigmoid/tanh and the state update, and write the new state to the batch output
// and straight into the node-memory rows.  Pure streaming, HBM-bound: one thread
// handles VEC consecutive hidden units of one node with 16-byte accesses.
//
// Gate order follows torch: GRU r,z,n ; LSTM i,f,g,o.

#include "common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int VEC>
struct Pack {
  float v[VEC];
};

template <int VEC>
__device__ __forceinline__ Pack<VEC> ld(const float* p) {
  Pack<VEC> r;
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else {
    r.v[0] = *p;
  }
  return r;
}

template <int VEC>
__device__ __forceinline__ Pack<VEC> ld_or_zero(const float* p, int64_t off) {
  if (p) return ld<VEC>(p + off);
  Pack<VEC> r;
#pragma unroll
  for (int i = 0; i < VEC; ++i) r.v[i] = 0.0f;
  return r;
}

template <int VEC>
__device__ __forceinline__ void st(float* p, const Pack<VEC>& r) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  } else {
    *p = r.v[0];
  }
}

// Decodes a flat work index into (node m, first hidden unit j).
struct Elem {
  int64_t m;
  int j;
};
template <int VEC>
__device__ __forceinline__ bool decode(int64_t M, int H, Elem* e) {
  const int per_row = H / VEC;
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (idx >= M * per_row) return false;
  e->m = idx / per_row;
  e->j = static_cast<int>(idx % per_row) * VEC;
  return true;
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(const float* __restrict__ table,
                                                               int64_t ld_, const int64_t* ids,
                                                               int64_t M, int H,
                                                               float* __restrict__ out) {
  Elem e;
  if (!decode<VEC>(M, H, &e)) return;
  st<VEC>(out + e.m * H + e.j, ld<VEC>(table + ids[e.m] * ld_ + e.j));
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void gru_fwd_kernel(
    const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ b_ih,
    const float* __restrict__ b_hh, float* h_mem, const int64_t* __restrict__ ids, int64_t M, int H,
    float* __restrict__ h_out, int mem_write, float* __restrict__ saved) {
  Elem e;
  if (!decode<VEC>(M, H, &e)) return;
  const int64_t g0 = e.m * 3 * H + e.j;
  const int64_t mrow = (ids ? ids[e.m] : e.m) * H + e.j;
  const Pack<VEC> ir = ld<VEC>(gi + g0), iz = ld<VEC>(gi + g0 + H), in = ld<VEC>(gi + g0 + 2 * H);
  const Pack<VEC> hr = ld<VEC>(gh + g0), hz = ld<VEC>(gh + g0 + H), hn = ld<VEC>(gh + g0 + 2 * H);
  const Pack<VEC> bir = ld_or_zero<VEC>(b_ih, e.j), biz = ld_or_zero<VEC>(b_ih, H + e.j),
                  bin = ld_or_zero<VEC>(b_ih, 2 * H + e.j);
  const Pack<VEC> bhr = ld_or_zero<VEC>(b_hh, e.j), bhz = ld_or_zero<VEC>(b_hh, H + e.j),
                  bhn = ld_or_zero<VEC>(b_hh, 2 * H + e.j);
  const Pack<VEC> h = ld<VEC>(h_mem + mrow);
  Pack<VEC> r, z, n, hnb, hy;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    r.v[i] = sigmoidf_((ir.v[i] + bir.v[i]) + (hr.v[i] + bhr.v[i]));
    z.v[i] = sigmoidf_((iz.v[i] + biz.v[i]) + (hz.v[i] + bhz.v[i]));
    hnb.v[i] = hn.v[i] + bhn.v[i];
    n.v[i] = tanhf((in.v[i] + bin.v[i]) + r.v[i] * hnb.v[i]);
    hy.v[i] = n.v[i] + z.v[i] * (h.v[i] - n.v[i]);
  }
  st<VEC>(h_out + e.m * H + e.j, hy);
  if (saved) {
    float* s = saved + e.m * 5 * H + e.j;
    st<VEC>(s, r);
    st<VEC>(s + H, z);
    st<VEC>(s + 2 * H, n);
    st<VEC>(s + 3 * H, hnb);
    st<VEC>(s + 4 * H, h);
  }
  if (mem_write) st<VEC>(h_mem + mrow, hy);
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void gru_bwd_kernel(const float* __restrict__ g_out,
                                                           const float* __restrict__ saved,
                                                           int64_t M, int H,
                                                           float* __restrict__ g_gi,
                                                           float* __restrict__ g_gh,
                                                           float* __restrict__ g_h) {
  Elem e;
  if (!decode<VEC>(M, H, &e)) return;
  const float* s = saved + e.m * 5 * H + e.j;
  const Pack<VEC> g = ld<VEC>(g_out + e.m * H + e.j);
  const Pack<VEC> r = ld<VEC>(s), z = ld<VEC>(s + H), n = ld<VEC>(s + 2 * H),
                  hn = ld<VEC>(s + 3 * H), h = ld<VEC>(s + 4 * H);
  Pack<VEC> dar, daz, dan, danr, dh;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const float dn = g.v[i] * (1.0f - z.v[i]);
    const float dz = g.v[i] * (h.v[i] - n.v[i]);
    dan.v[i] = dn * (1.0f - n.v[i] * n.v[i]);
    daz.v[i] = dz * z.v[i] * (1.0f - z.v[i]);
    const float dr = dan.v[i] * hn.v[i];
    dar.v[i] = dr * r.v[i] * (1.0f - r.v[i]);
    danr.v[i] = dan.v[i] * r.v[i];
    dh.v[i] = g.v[i] * z.v[i];
  }
  const int64_t g0 = e.m * 3 * H + e.j;
  st<VEC>(g_gi + g0, dar);
  st<VEC>(g_gi + g0 + H, daz);
  st<VEC>(g_gi + g0 + 2 * H, dan);
  st<VEC>(g_gh + g0, dar);
  st<VEC>(g_gh + g0 + H, daz);
  st<VEC>(g_gh + g0 + 2 * H, danr);
  if (g_h) st<VEC>(g_h + e.m * H + e.j, dh);
}

// The same gate gradients as ONE matrix [M, 4H] = [d a_n | d a_r | d a_z | d a_n * r]: its
// columns 0..3H are the gradient of the input-side pre-activations (gate order n, r, z), its
// columns H..4H that of the state side (gate order r, z, n), so that the layer's backward
// reads what gru_bwd_kernel writes as two [M, 3H] matrices from two thirds of the bytes.
template <int VEC>
__global__ __launch_bounds__(kThreads) void gru_bwd_packed_kernel(const float* __restrict__ g_out,
                                                                  const float* __restrict__ saved,
                                                                  int64_t M, int H,
                                                                  float* __restrict__ g4,
                                                                  float* __restrict__ g_h) {
  Elem e;
  if (!decode<VEC>(M, H, &e)) return;
  const float* s = saved + e.m * 5 * H + e.j;
  const Pack<VEC> g = ld<VEC>(g_out + e.m * H + e.j);
  const Pack<VEC> r = ld<VEC>(s), z = ld<VEC>(s + H), n = ld<VEC>(s + 2 * H),
                  hn = ld<VEC>(s + 3 * H), h = ld<VEC>(s + 4 * H);
  Pack<VEC> dar, daz, dan, danr, dh;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {  // the same expressions, in the same order, as gru_bwd_kernel
    const float dn = g.v[i] * (1.0f - z.v[i]);
    const float dz = g.v[i] * (h.v[i] - n.v[i]);
    dan.v[i] = dn * (1.0f - n.v[i] * n.v[i]);
    daz.v[i] = dz * z.v[i] * (1.0f - z.v[i]);
    const float dr = dan.v[i] * hn.v[i];
    dar.v[i] = dr * r.v[i] * (1.0f - r.v[i]);
    danr.v[i] = dan.v[i] * r.v[i];
    dh.v[i] = g.v[i] * z.v[i];
  }
  const int64_t g0 = e.m * 4 * H + e.j;
  st<VEC>(g4 + g0, dan);
  st<VEC>(g4 + g0 + H, dar);
  st<VEC>(g4 + g0 + 2 * H, daz);
  st<VEC>(g4 + g0 + 3 * H, danr);
  if (g_h) st<VEC>(g_h + e.m * H + e.j, dh);
}

// The same packed gradients from gate values kept as four planes per row, [M][4][H] = r, z, n,
// W_hn h + b_hn (what the whole-table forward of cell_rows.hip writes), and the old state read
// from the matrix it came from.  A thread handles four consecutive hidden units of one node.
__global__ __launch_bounds__(kThreads) void gru_bwd_packed_q_kernel(const float* __restrict__ g_out,
                                                                    const float* __restrict__ saved,
                                                                    const float* __restrict__ h_old,
                                                                    int64_t ldh, int64_t M, int H,
                                                                    float* __restrict__ g4) {
  Elem e;
  if (!decode<4>(M, H, &e)) return;
  const Pack<4> g = ld<4>(g_out + e.m * H + e.j);
  const Pack<4> h = ld<4>(h_old + e.m * ldh + e.j);
  const float* s = saved + e.m * 4 * H + e.j;
  const Pack<4> rq = ld<4>(s), zq = ld<4>(s + H), nq = ld<4>(s + 2 * H), hq = ld<4>(s + 3 * H);
  Pack<4> dar, daz, dan, danr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // the same expressions, in the same order, as gru_bwd_kernel
    const float r = rq.v[i], z = zq.v[i], n = nq.v[i], hn = hq.v[i];
    const float dn = g.v[i] * (1.0f - z);
    const float dz = g.v[i] * (h.v[i] - n);
    dan.v[i] = dn * (1.0f - n * n);
    daz.v[i] = dz * z * (1.0f - z);
    const float dr = dan.v[i] * hn;
    dar.v[i] = dr * r * (1.0f - r);
    danr.v[i] = dan.v[i] * r;
  }
  const int64_t g0 = e.m * 4 * H + e.j;
  st<4>(g4 + g0, dan);
  st<4>(g4 + g0 + H, dar);
  st<4>(g4 + g0 + 2 * H, daz);
  st<4>(g4 + g0 + 3 * H, danr);
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void lstm_fwd_kernel(
    const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ b_ih,
    const float* __restrict__ b_hh, float* h_mem, float* c_mem, const int64_t* __restrict__ ids,
    int64_t M, int H, float* __restrict__ h_out, float* __restrict__ c_out, int mem_write,
    float* __restrict__ saved) {
  Elem e;
  if (!decode<VEC>(M, H, &e)) return;
  const int64_t g0 = e.m * 4 * H + e.j;
  const int64_t mrow = (ids ? ids[e.m] : e.m) * H + e.j;
  Pack<VEC> a[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const Pack<VEC> x = ld<VEC>(gi + g0 + k * H), y = ld<VEC>(gh + g0 + k * H);
    const Pack<VEC> bx = ld_or_zero<VEC>(b_ih, k * H + e.j), by = ld_or_zero<VEC>(b_hh, k * H + e.j);
#pragma unroll
    for (int i = 0; i < VEC; ++i) a[k].v[i] = (x.v[i] + bx.v[i]) + (y.v[i] + by.v[i]);
  }
  const Pack<VEC> c = ld<VEC>(c_mem + mrow);
  Pack<VEC> ig, fg, gg, og, cy, tc, hy;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    ig.v[i] = sigmoidf_(a[0].v[i]);
    fg.v[i] = sigmoidf_(a[1].v[i]);
    gg.v[i] = tanhf(a[2].v[i]);
    og.v[i] = sigmoidf_(a[3].v[i]);
    cy.v[i] = fg.v[i] * c.v[i] + ig.v[i] * gg.v[i];
    tc.v[i] = tanhf(cy.v[i]);
    hy.v[i] = og.v[i] * tc.v[i];
  }
  st<VEC>(h_out + e.m * H + e.j, hy);
  st<VEC>(c_out + e.m * H + e.j, cy);
  if (saved) {
    float* s = saved + e.m * 6 * H + e.j;
    st<VEC>(s, ig);
    st<VEC>(s + H, fg);
    st<VEC>(s + 2 * H, gg);
    st<VEC>(s + 3 * H, og);
    st<VEC>(s + 4 * H, c);
    st<VEC>(s + 5 * H, tc);
  }
  if (mem_write) {
    st<VEC>(h_mem + mrow, hy);
    st<VEC>(c_mem + mrow, cy);
  }
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void lstm_bwd_kernel(const float* __restrict__ gh_out,
                                                            const float* __restrict__ gc_out,
                                                            const float* __restrict__ saved,
                                                            int64_t M, int H,
                                                            float* __restrict__ g_gates,
                                                            float* __restrict__ g_c) {
  Elem e;
  if (!decode<VEC>(M, H, &e)) return;
  const float* s = saved + e.m * 6 * H + e.j;
  const Pack<VEC> gh = ld<VEC>(gh_out + e.m * H + e.j);
  const Pack<VEC> gc = ld_or_zero<VEC>(gc_out, e.m * H + e.j);
  const Pack<VEC> ig = ld<VEC>(s), fg = ld<VEC>(s + H), gg = ld<VEC>(s + 2 * H),
                  og = ld<VEC>(s + 3 * H), c = ld<VEC>(s + 4 * H), tc = ld<VEC>(s + 5 * H);
  Pack<VEC> di, df, dg, dou, dcp;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const float dc = gc.v[i] + gh.v[i] * og.v[i] * (1.0f - tc.v[i] * tc.v[i]);
    dou.v[i] = gh.v[i] * tc.v[i] * og.v[i] * (1.0f - og.v[i]);
    di.v[i] = dc * gg.v[i] * ig.v[i] * (1.0f - ig.v[i]);
    df.v[i] = dc * c.v[i] * fg.v[i] * (1.0f - fg.v[i]);
    dg.v[i] = dc * ig.v[i] * (1.0f - gg.v[i] * gg.v[i]);
    dcp.v[i] = dc * fg.v[i];
  }
  const int64_t g0 = e.m * 4 * H + e.j;
  st<VEC>(g_gates + g0, di);
  st<VEC>(g_gates + g0 + H, df);
  st<VEC>(g_gates + g0 + 2 * H, dg);
  st<VEC>(g_gates + g0 + 3 * H, dou);
  if (g_c) st<VEC>(g_c + e.m * H + e.j, dcp);
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void rnn_fwd_kernel(
    const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ b_ih,
    const float* __restrict__ b_hh, float* h_mem, const int64_t* __restrict__ ids, int64_t M, int H,
    float* __restrict__ h_out, int mem_write) {
  Elem e;
  if (!decode<VEC>(M, H, &e)) return;
  const int64_t g0 = e.m * H + e.j;
  const Pack<VEC> x = ld<VEC>(gi + g0), y = ld<VEC>(gh + g0);
  const Pack<VEC> bx = ld_or_zero<VEC>(b_ih, e.j), by = ld_or_zero<VEC>(b_hh, e.j);
  Pack<VEC> hy;
#pragma unroll
  for (int i = 0; i < VEC; ++i) hy.v[i] = tanhf((x.v[i] + bx.v[i]) + (y.v[i] + by.v[i]));
  st<VEC>(h_out + g0, hy);
  if (mem_write) st<VEC>(h_mem + (ids ? ids[e.m] : e.m) * H + e.j, hy);
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void rnn_bwd_kernel(const float* __restrict__ g_out,
                                                           const float* __restrict__ h_out,
                                                           int64_t M, int H,
                                                           float* __restrict__ g_gates) {
  Elem e;
  if (!decode<VEC>(M, H, &e)) return;
  const int64_t g0 = e.m * H + e.j;
  const Pack<VEC> g = ld<VEC>(g_out + g0), h = ld<VEC>(h_out + g0);
  Pack<VEC> d;
#pragma unroll
  for (int i = 0; i < VEC; ++i) d.v[i] = g.v[i] * (1.0f - h.v[i] * h.v[i]);
  st<VEC>(g_gates + g0, d);
}

int check_mh(int64_t M, int64_t H, const char* who) {
  SPMI_REQUIRE(M >= 0 && H >= 1 && H < (int64_t{1} << 24), SPMI_ERR_INVALID_ARG,
               "%s: bad sizes M=%lld H=%lld", who, (long long)M, (long long)H);
  SPMI_REQUIRE(M * H < (int64_t{1} << 40), SPMI_ERR_UNSUPPORTED, "%s: M*H too large", who);
  return SPMI_OK;
}

bool all_aligned16(std::initializer_list<const void*> ps) {
  for (const void* p : ps)
    if (p && (reinterpret_cast<uintptr_t>(p) & 15u)) return false;
  return true;
}

unsigned grid_for(int64_t M, int64_t H, int vec) {
  return static_cast<unsigned>((M * (H / vec) + kThreads - 1) / kThreads);
}

}  // namespace

#define SPMI_LAUNCH_VEC(kernel, vec4_ok, M, H, st, ...)                                  \
  do {                                                                                   \
    if (vec4_ok)                                                                         \
      kernel<4><<<grid_for(M, H, 4), kThreads, 0, st>>>(__VA_ARGS__);                    \
    else                                                                                 \
      kernel<1><<<grid_for(M, H, 1), kThreads, 0, st>>>(__VA_ARGS__);                    \
    SPMI_CHECK_LAUNCH(#kernel);                                                          \
  } while (0)

extern "C" {

int spmi_gather_rows(const float* table, int64_t ld_, const int64_t* ids, int64_t M, int64_t H,
                     float* out, void* stream) {
  if (int rc = check_mh(M, H, "spmi_gather_rows")) return rc;
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(table && ids && out && ld_ >= H, SPMI_ERR_INVALID_ARG,
               "spmi_gather_rows: null pointer or ld < H");
  const bool v4 = (H % 4 == 0) && (ld_ % 4 == 0) && all_aligned16({table, out});
  SPMI_LAUNCH_VEC(gather_rows_kernel, v4, M, H, spmi::as_stream(stream), table, ld_, ids, M,
                  static_cast<int>(H), out);
  return SPMI_OK;
}

int spmi_gru_gate_fwd(const float* gi, const float* gh, const float* b_ih, const float* b_hh,
                      float* h_mem, const int64_t* ids, int64_t M, int64_t H, float* h_out,
                      int mem_write, float* saved, void* stream) {
  if (int rc = check_mh(M, H, "spmi_gru_gate_fwd")) return rc;
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(gi && gh && h_mem && h_out, SPMI_ERR_INVALID_ARG, "spmi_gru_gate_fwd: null pointer");
  const bool v4 = (H % 4 == 0) && all_aligned16({gi, gh, b_ih, b_hh, h_mem, h_out, saved});
  SPMI_LAUNCH_VEC(gru_fwd_kernel, v4, M, H, spmi::as_stream(stream), gi, gh, b_ih, b_hh, h_mem, ids,
                  M, static_cast<int>(H), h_out, mem_write, saved);
  return SPMI_OK;
}

int spmi_gru_gate_bwd(const float* gh_out, const float* saved, int64_t M, int64_t H, float* g_gi,
                      float* g_gh, float* g_h, void* stream) {
  if (int rc = check_mh(M, H, "spmi_gru_gate_bwd")) return rc;
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(gh_out && saved && g_gi && g_gh, SPMI_ERR_INVALID_ARG,
               "spmi_gru_gate_bwd: null pointer");
  const bool v4 = (H % 4 == 0) && all_aligned16({gh_out, saved, g_gi, g_gh, g_h});
  SPMI_LAUNCH_VEC(gru_bwd_kernel, v4, M, H, spmi::as_stream(stream), gh_out, saved, M,
                  static_cast<int>(H), g_gi, g_gh, g_h);
  return SPMI_OK;
}

int spmi_gru_gate_bwd_packed(const float* gh_out, const float* saved, int64_t M, int64_t H, float* g4,
                             float* g_h, void* stream) {
  if (int rc = check_mh(M, H, "spmi_gru_gate_bwd_packed")) return rc;
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(gh_out && saved && g4, SPMI_ERR_INVALID_ARG, "spmi_gru_gate_bwd_packed: null pointer");
  const bool v4 = (H % 4 == 0) && all_aligned16({gh_out, saved, g4, g_h});
  SPMI_LAUNCH_VEC(gru_bwd_packed_kernel, v4, M, H, spmi::as_stream(stream), gh_out, saved, M,
                  static_cast<int>(H), g4, g_h);
  return SPMI_OK;
}

int spmi_gru_gate_bwd_packed_q(const float* gh_out, const float* saved, const float* h_old, int64_t ldh,
                               int64_t M, int64_t H, float* g4, void* stream) {
  if (int rc = check_mh(M, H, "spmi_gru_gate_bwd_packed_q")) return rc;
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(gh_out && saved && h_old && g4, SPMI_ERR_INVALID_ARG, "spmi_gru_gate_bwd_packed_q: null pointer");
  SPMI_REQUIRE(H % 4 == 0 && ldh >= H && ldh % 4 == 0 && all_aligned16({gh_out, saved, h_old, g4}),
               SPMI_ERR_INVALID_ARG,
               "spmi_gru_gate_bwd_packed_q: H and ldh must be multiples of 4 and the pointers 16-byte aligned");
  const int64_t work = M * (H / 4);
  const int64_t blocks = (work + kThreads - 1) / kThreads;
  SPMI_REQUIRE(blocks < (int64_t{1} << 31), SPMI_ERR_UNSUPPORTED, "spmi_gru_gate_bwd_packed_q: too many elements");
  gru_bwd_packed_q_kernel<<<static_cast<unsigned>(blocks), kThreads, 0, spmi::as_stream(stream)>>>(
      gh_out, saved, h_old, ldh, M, static_cast<int>(H), g4);
  SPMI_CHECK_LAUNCH("gru_bwd_packed_q");
  return SPMI_OK;
}

int spmi_lstm_gate_fwd(const float* gi, const float* gh, const float* b_ih, const float* b_hh,
                       float* h_mem, float* c_mem, const int64_t* ids, int64_t M, int64_t H,
                       float* h_out, float* c_out, int mem_write, float* saved, void* stream) {
  if (int rc = check_mh(M, H, "spmi_lstm_gate_fwd")) return rc;
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(gi && gh && h_mem && c_mem && h_out && c_out, SPMI_ERR_INVALID_ARG,
               "spmi_lstm_gate_fwd: null pointer");
  const bool v4 =
      (H % 4 == 0) && all_aligned16({gi, gh, b_ih, b_hh, h_mem, c_mem, h_out, c_out, saved});
  SPMI_LAUNCH_VEC(lstm_fwd_kernel, v4, M, H, spmi::as_stream(stream), gi, gh, b_ih, b_hh, h_mem,
                  c_mem, ids, M, static_cast<int>(H), h_out, c_out, mem_write, saved);
  return SPMI_OK;
}

int spmi_lstm_gate_bwd(const float* gh_out, const float* gc_out, const float* saved, int64_t M,
                       int64_t H, float* g_gates, float* g_c, void* stream) {
  if (int rc = check_mh(M, H, "spmi_lstm_gate_bwd")) return rc;
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(gh_out && saved && g_gates, SPMI_ERR_INVALID_ARG, "spmi_lstm_gate_bwd: null pointer");
  const bool v4 = (H % 4 == 0) && all_aligned16({gh_out, gc_out, saved, g_gates, g_c});
  SPMI_LAUNCH_VEC(lstm_bwd_kernel, v4, M, H, spmi::as_stream(stream), gh_out, gc_out, saved, M,
                  static_cast<int>(H), g_gates, g_c);
  return SPMI_OK;
}

int spmi_rnn_gate_fwd(const float* gi, const float* gh, const float* b_ih, const float* b_hh,
                      float* h_mem, const int64_t* ids, int64_t M, int64_t H, float* h_out,
                      int mem_write, void* stream) {
  if (int rc = check_mh(M, H, "spmi_rnn_gate_fwd")) return rc;
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(gi && gh && h_out && (!mem_write || h_mem), SPMI_ERR_INVALID_ARG,
               "spmi_rnn_gate_fwd: null pointer");
  const bool v4 = (H % 4 == 0) && all_aligned16({gi, gh, b_ih, b_hh, h_mem, h_out});
  SPMI_LAUNCH_VEC(rnn_fwd_kernel, v4, M, H, spmi::as_stream(stream), gi, gh, b_ih, b_hh, h_mem, ids,
                  M, static_cast<int>(H), h_out, mem_write);
  return SPMI_OK;
}

int spmi_rnn_gate_bwd(const float* gh_out, const float* h_out, int64_t M, int64_t H,
                      float* g_gates, void* stream) {
  if (int rc = check_mh(M, H, "spmi_rnn_gate_bwd")) return rc;
  if (M == 0) return SPMI_OK;
  SPMI_REQUIRE(gh_out && h_out && g_gates, SPMI_ERR_INVALID_ARG, "spmi_rnn_gate_bwd: null pointer");
  const bool v4 = (H % 4 == 0) && all_aligned16({gh_out, h_out, g_gates});
  SPMI_LAUNCH_VEC(rnn_bwd_kernel, v4, M, H, spmi::as_stream(stream), gh_out, h_out, M,
                  static_cast<int>(H), g_gates);
  return SPMI_OK;
}

}  // extern "C"
