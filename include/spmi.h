/*
 * spmi.h -- C-ABI of libspmi, the MI355X (gfx950) implementation of the
 * per-snapshot GCN message-passing path of spiro.dtdg.
 *
 * The reference (mcgill-cpslab/spiral, package `spiro`) has no FFI of its own:
 * on this path it calls DGL 0.8.0.post1 (poetry.lock:206-207), whose C++ does
 * the arithmetic.  Each entry point below names the reference call site whose
 * work it replaces.
 *
 * Conventions
 *   - every pointer marked "device" is caller-owned device memory on the
 *     current HIP device; the library never allocates, frees or synchronises
 *     in a launch entry point (all work is queued on `stream`);
 *   - workspaces are caller-provided, sized by the matching *_workspace_bytes;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - return value: 0 = queued successfully, negative = SPMI_ERR_* and
 *     spmi_last_error() describes the failure (thread-local string);
 *   - node ids at the boundary are int64 like the reference's ID_TYPE
 *     (spiro/dtdg/types.py:39); on-device CSR arrays are int32, which bounds
 *     one snapshot to N < 2^31 nodes and E < 2^31 edges (checked);
 *   - features are float32 like the reference's FEATURE_TYPE
 *     (spiro/dtdg/types.py:40), row-major with an explicit leading dimension
 *     (spmi_gcn_spmm_ex also accepts bfloat16 storage).
 */
#ifndef SPMI_H
#define SPMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPMI_OK 0
#define SPMI_ERR_INVALID_ARG (-1) /* null pointer, negative size, bad enum  */
#define SPMI_ERR_HIP (-2)         /* a HIP runtime call failed               */
#define SPMI_ERR_WORKSPACE (-3)   /* workspace too small                      */
#define SPMI_ERR_RANGE (-4)       /* a node id outside [0, N)                 */
#define SPMI_ERR_UNSUPPORTED (-5) /* size beyond the int32 on-device limits   */

/* activation codes of the fused epilogues */
#define SPMI_ACT_NONE 0
#define SPMI_ACT_RELU 1
#define SPMI_ACT_LEAKY_RELU 2 /* slope given separately (torch default 0.01) */

/* storage type of feature rows (sums are always float32) */
#define SPMI_DTYPE_F32 0
#define SPMI_DTYPE_BF16 1

/* degree -> norm modes (DGL GraphConv `norm`) */
#define SPMI_NORM_RSQRT 0 /* clamp(deg,1)^-0.5 : norm='both'            */
#define SPMI_NORM_RECIP 1 /* 1/clamp(deg,1)    : norm='right' / 'left'  */

int spmi_version(void);
const char* spmi_last_error(void);

/* ------------------------------------------------------------------------
 * Degrees.  Replaces DGLGraph.in_degrees() (spiro/dtdg/types.py:241) and the
 * in_degrees()/out_degrees() calls inside dgl GraphConv.forward reached from
 * staticGraphEncoder.py:113,116.  deg[v] = number of occurrences of v in
 * ids[0..E); multi-edges count every time.  Exact and reproducible: for
 * N <= 2^20 one radix pass groups the ids into buckets of 4096 nodes and a
 * workgroup per bucket counts them in LDS (integer LDS atomics only); larger N
 * take a keys-only radix sort followed by run boundaries (no atomics at all).
 * `status` (device, 1 x int32, caller-zeroed) is set non-zero when an id is
 * outside [0,N); the offending id is clamped so no access leaves the buffers.
 * ---------------------------------------------------------------------- */
size_t spmi_degree_workspace_bytes(int64_t E, int64_t N);
int spmi_degree(const int64_t* ids /*device [E]*/, int64_t E, int64_t N,
                int64_t* deg /*device [N]*/, void* ws, size_t ws_bytes,
                int32_t* status /*device [1]*/, void* stream);

/* ------------------------------------------------------------------------
 * COO -> CSR.  Replaces dgl.graph((src,dst), num_nodes) + DGL's lazy
 * COOToCSR, reached from spiro/dtdg/types.py:191.  Rows are grouped by
 * `key` (pass dst for the by-destination CSR used by the forward, src for
 * the by-source CSR used by the backward); inside a row the entries keep the
 * original edge order (stable), `indices[j]` is the other endpoint and
 * `eid[j]` the original edge position (eid may be NULL when edge ids are not
 * needed: one payload less travels through the sort).  indptr has N+1 entries.
 * Implemented as a stable LSB radix sort (<= 8-bit digits) + run boundaries.
 * ---------------------------------------------------------------------- */
size_t spmi_coo_to_csr_workspace_bytes(int64_t E, int64_t N);
int spmi_coo_to_csr(const int64_t* key /*device [E]*/, const int64_t* other /*device [E]*/,
                    int64_t E, int64_t N /*key range*/, int64_t N_other /*other range*/,
                    int32_t* indptr /*device [N+1]*/, int32_t* indices /*device [E]*/,
                    int32_t* eid /*nullable, device [E]*/, void* ws, size_t ws_bytes,
                    int32_t* status /*device [1]*/, void* stream);

/* ------------------------------------------------------------------------
 * Snapshot build in one call: COO -> CSR, the normaliser of the key side and
 * the load-balance plan of the aggregation.  Replaces the same reference
 * lines as spmi_coo_to_csr (dgl.graph + COOToCSR, spiro/dtdg/types.py:191),
 * spmi_norm_from_indptr (the in_degrees()/clamp/pow of dgl GraphConv.forward,
 * call sites staticGraphEncoder.py:113,116) and spmi_spmm_plan together; the
 * outputs are bit-identical to those three calls.  For 4096 < N <= 2^24 the
 * edges take stable radix passes on the TOP bits of the row id only (one pass
 * up to 2^20 rows, two beyond), after which every bucket of 4096 rows is
 * contiguous and in edge order; the rest is a stable counting sort inside the
 * buckets (segments of 8192 edges, wave-ballot ranking, writes inside a
 * cache-resident window), whose workgroups also write row pointers, normaliser
 * and hub rows: at 1M rows six launches and two passes over the edges instead
 * of sixteen launches and four passes.  Other N run the three calls as they are.
 *   norm_key  nullable, device [N]: f(row degree), mode as spmi_norm_from_indptr
 *   norm_other nullable, device [N_other]: the same f of the number of edges whose other end
 *             is the node (the source normaliser beside a by-destination CSR): the first pass
 *             reads both id arrays anyway, so the grouping of the other endpoints into buckets
 *             of 4096 nodes rides on it (one more histogram per tile, one unordered scatter
 *             from registers) and a counting kernel follows.  Only where
 *             spmi_snapshot_csr_norm_other_supported(E, N, N_other) (N and N_other <= 2^20);
 *             elsewhere pass NULL and call spmi_norm_from_ids32 on `indices`
 *   meta      device int32[3], ZEROED BY THE CALLER: [0] status (non-zero: an
 *             id outside its range, clamped), [1] hub rows, [2] chunks
 *   heavy_*   as spmi_spmm_plan; the order of heavy_rows is not specified
 * ---------------------------------------------------------------------- */
size_t spmi_snapshot_csr_workspace_bytes(int64_t E, int64_t N);
int spmi_snapshot_csr_norm_other_supported(int64_t E, int64_t N, int64_t N_other);
int spmi_snapshot_csr(const int64_t* key /*device [E]*/, const int64_t* other /*device [E]*/,
                      int64_t E, int64_t N /*key range*/, int64_t N_other /*other range*/,
                      int32_t* indptr /*device [N+1]*/, int32_t* indices /*device [E]*/,
                      int32_t* eid /*nullable, device [E]*/, float* norm_key /*nullable, device [N]*/,
                      float* norm_other /*nullable, device [N_other]*/,
                      int norm_mode, int32_t heavy_thresh, int32_t chunk,
                      int32_t* heavy_rows, int32_t* heavy_chunk_ptr, int64_t capacity, void* ws,
                      size_t ws_bytes, int32_t* meta /*device [3], zeroed*/, void* stream);

/* ------------------------------------------------------------------------
 * Cumulative snapshots.  Replaces the per-call rebuild of dgl.graph on the
 * slice `edges[:anchor[t]]` in VEInvariantDTDG.dispatcher
 * (spiro/dtdg/types.py:183-198) and the extra build of snapshot t-1 that
 * CitationGraph.dispatcher makes only to read in-degrees (types.py:239-242).
 * The time-sorted table is grouped once with spmi_coo_to_csr (with eid); as
 * that grouping is stable, the edges [0, e_cut) of the table form a prefix of
 * every row.  spmi_csr_prefix packs those row prefixes for the first n_rows
 * rows: the result equals spmi_coo_to_csr run on the first e_cut edges with
 * N = n_rows, bit for bit (indptr, indices and, when `eid` is given, eid).
 * `status` is raised when an edge of the prefix touches a row >= n_rows or a
 * column >= n_cols.  spmi_csr_prefix_degree returns only the prefix lengths
 * (the degrees of that snapshot).
 * ---------------------------------------------------------------------- */
size_t spmi_csr_prefix_workspace_bytes(int64_t n_rows);
int spmi_csr_prefix(const int32_t* indptr_all /*device [>= n_rows+1]*/,
                    const int32_t* indices_all, const int32_t* eid_all /*device [E_all]*/,
                    int64_t n_rows, int64_t n_cols, int64_t e_cut,
                    int32_t* indptr /*device [n_rows+1]*/, int32_t* indices /*device [e_cut]*/,
                    int32_t* eid /*nullable, device [e_cut]*/, void* ws, size_t ws_bytes,
                    int32_t* status /*device [1]*/, void* stream);
int spmi_csr_prefix_degree(const int32_t* indptr_all, const int32_t* eid_all, int64_t n_rows,
                           int64_t e_cut, int64_t* deg /*device [n_rows]*/, void* stream);

/* ------------------------------------------------------------------------
 * Degree -> normaliser.  Replaces `degs.float().clamp(min=1)` followed by
 * `torch.pow(degs, -0.5)` (norm='both') or `1.0/degs` (norm='right'/'left')
 * inside dgl GraphConv.forward (call sites staticGraphEncoder.py:113,116).
 * deg is read from a CSR indptr: deg[v] = indptr[v+1]-indptr[v].
 * ---------------------------------------------------------------------- */
int spmi_norm_from_indptr(const int32_t* indptr /*device [N+1]*/, int64_t N, int mode,
                          float* norm /*device [N]*/, void* stream);
/* Same normaliser from an int64 degree vector (the output of spmi_degree or
 * spmi_csr_prefix_degree), for the side of GraphConv whose CSR is not built. */
int spmi_norm_from_degree(const int64_t* deg /*device [N]*/, int64_t N, int mode,
                          float* norm /*device [N]*/, void* stream);

/* The normaliser of the side of GraphConv whose CSR is not built, straight from a column of
 * 32-bit node ids that the library itself produced and range-checked (the `indices` of the
 * other side's CSR): norm[v] = f(number of occurrences of v in ids), f as in
 * spmi_norm_from_indptr; deg (nullable, int64 [N]) receives the counts.  Same reference lines
 * as spmi_degree + spmi_norm_from_degree (the out_degrees() / clamp / pow of dgl
 * GraphConv.forward), half the bytes of a second pass over the caller's int64 ids and one
 * launch less.  Workspace: spmi_degree_workspace_bytes(E, N).  ids outside [0, N) are not
 * detected here (no access leaves the buffers; the counts are then meaningless). */
int spmi_norm_from_ids32(const int32_t* ids /*device [E]*/, int64_t E, int64_t N, int mode,
                         float* norm /*device [N]*/, int64_t* deg /*nullable, device [N]*/, void* ws,
                         size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Load-balance plan for the aggregation: rows whose degree exceeds
 * `heavy_thresh` are listed and cut into chunks of `chunk` nonzeros that
 * separate workgroups sum; the row-per-wavefront kernel skips them.
 * heavy_rows / heavy_chunk_ptr need capacity E/heavy_thresh + 2 entries.
 * counts[0] = number of heavy rows, counts[1] = number of chunks (device;
 * the host reads them back once per snapshot to size the partial buffer).
 * No counterpart in the reference: DGL's CPU SpMM is row-parallel OpenMP.
 * ---------------------------------------------------------------------- */
int spmi_spmm_plan(const int32_t* indptr /*device [N+1]*/, int64_t N, int32_t heavy_thresh,
                   int32_t chunk, int32_t* heavy_rows, int32_t* heavy_chunk_ptr,
                   int64_t capacity, int32_t* counts /*device [2]*/, void* stream);

/* ------------------------------------------------------------------------
 * Fused normalised aggregation.  Replaces, in one kernel, the
 * `feat * norm` source scaling, `graph.update_all(copy_u|u_mul_e, sum)`
 * (DGL SpMMCsr), the destination scaling, `+ bias` and the activation of
 * dgl GraphConv.forward (call sites staticGraphEncoder.py:113,116):
 *
 *   Y[r,:] = act( dst_scale[r] * sum_{j in row r} w_j * X[indices[j],:] + bias )
 *   w_j    = (edge_w ? edge_w[j] : 1) * (src_scale ? src_scale[indices[j]] : 1)
 *
 * The CSR has n_rows rows (this rank's destination block) and its indices
 * address rows of X in [0, n_src).  Sums run in CSR order inside a row for
 * rows up to heavy_thresh; heavier rows are summed chunk-wise in a fixed
 * order, so results are bit-reproducible run to run (no atomics).
 * The same entry point computes the backward aggregation when given the
 * by-source CSR with the two scale vectors exchanged.
 * n_heavy == 0 disables the heavy path (partial_ws may then be NULL).
 * ---------------------------------------------------------------------- */
int spmi_gcn_spmm(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t n_src,
                  int64_t E, const float* X, int64_t ldx, int64_t F,
                  const float* src_scale /*nullable [n_src]*/,
                  const float* edge_w /*nullable [E], CSR order*/,
                  const float* dst_scale /*nullable [n_rows]*/,
                  const float* bias /*nullable [F]*/, int act, float slope,
                  float* Y, int64_t ldy,
                  const int32_t* heavy_rows, const int32_t* heavy_chunk_ptr,
                  int32_t n_heavy, int32_t n_chunks, int32_t heavy_thresh, int32_t chunk,
                  float* partial_ws /*device [n_chunks*F]*/, void* stream);

/* Same aggregation with feature rows stored as float32 or bfloat16 (x_dtype / y_dtype =
 * SPMI_DTYPE_*; leading dimensions in elements).  Every product and sum is float32; a
 * bfloat16 output is rounded once, at the store.  The reference has no reduced-precision
 * path: this serves bf16 feature storage (half the gather bytes per edge). */
int spmi_gcn_spmm_ex(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t n_src,
                     int64_t E, const void* X, int x_dtype, int64_t ldx, int64_t F,
                     const float* src_scale, const float* edge_w, const float* dst_scale,
                     const float* bias, int act, float slope, void* Y, int y_dtype, int64_t ldy,
                     const int32_t* heavy_rows, const int32_t* heavy_chunk_ptr, int32_t n_heavy,
                     int32_t n_chunks, int32_t heavy_thresh, int32_t chunk, float* partial_ws,
                     void* stream);

/* The aggregation of spmi_gcn_spmm_ex in pieces, for a destination block whose source rows
 * arrive in several batches (the row-partitioned layer: the sum over the sources of batch k
 * runs while batch k+1 is still on the wire).  Each call covers the CSR of one batch of
 * sources.  accumulate != 0: the sums are added to what Y already holds (the raw partial
 * sums of the earlier calls); finalize == 0: Y receives the raw sums, without dst_scale,
 * bias and activation, which only the last call (finalize != 0) applies.  Partial sums
 * are float32: accumulate or finalize == 0 require y_dtype = SPMI_DTYPE_F32.  With
 * accumulate = 0 and finalize = 1 this is spmi_gcn_spmm_ex.  Same reference lines (the
 * update_all(copy_u, sum) of dgl GraphConv.forward); the reference has no partitioned
 * form. */
int spmi_gcn_spmm_part(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t n_src,
                       int64_t E, const void* X, int x_dtype, int64_t ldx, int64_t F,
                       const float* src_scale, const float* edge_w, const float* dst_scale,
                       const float* bias, int act, float slope, void* Y, int y_dtype, int64_t ldy,
                       const int32_t* heavy_rows, const int32_t* heavy_chunk_ptr, int32_t n_heavy,
                       int32_t n_chunks, int32_t heavy_thresh, int32_t chunk, float* partial_ws,
                       int accumulate, int finalize, void* stream);

/* The aggregation of spmi_gcn_spmm_ex for callers that cannot allocate device memory or
 * carry a workspace: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  No counterpart in the reference.  The partial sums of the hub-row
 * chunks go to an array built into the library (SPMI_STATIC_PARTIAL_FLOATS floats = 16 MiB,
 * held by every GPU the library runs on until the process ends) instead of `partial_ws`;
 * the call allocates and frees nothing and otherwise queues the same kernels, so results
 * are bit-identical to spmi_gcn_spmm_ex.  n_chunks * F beyond SPMI_STATIC_PARTIAL_FLOATS
 * is rejected with SPMI_ERR_WORKSPACE.  Every call on one GPU shares that GPU's array:
 * two calls that overlap in time on one GPU (different streams or host threads) race;
 * calls queued on one stream are safe.  The FIRST call on a device looks up the array's
 * address, which may make the runtime load the library's code object (and with it the
 * array's 16 MiB) at that moment: make it outside graph capture and before a region that
 * must not allocate.  Every later call on that device touches no allocator. */
#define SPMI_STATIC_PARTIAL_FLOATS (4 * 1024 * 1024)
int spmi_gcn_spmm_static(const int32_t* indptr, const int32_t* indices, int64_t n_rows,
                         int64_t n_src, int64_t E, const void* X, int x_dtype, int64_t ldx,
                         int64_t F, const float* src_scale, const float* edge_w,
                         const float* dst_scale, const float* bias, int act, float slope, void* Y,
                         int y_dtype, int64_t ldy, const int32_t* heavy_rows,
                         const int32_t* heavy_chunk_ptr, int32_t n_heavy, int32_t n_chunks,
                         int32_t heavy_thresh, int32_t chunk, void* stream);

/* ------------------------------------------------------------------------
 * Backward of the fused epilogue: gP = gY * act'(Y) and gb = column sums.
 * Replaces the autograd of `rst + bias` / activation in GraphConv.forward.
 * gb (nullable) must be zeroed by the caller; it is accumulated by one
 * atomic per workgroup column, so it is reproducible only to fp32 rounding.
 * ---------------------------------------------------------------------- */
int spmi_act_bias_bwd(const float* gY, const float* Y, int64_t M, int64_t F, int64_t ld,
                      int act, float slope, float* gP /*[M,F] ld=F*/, float* gb /*nullable [F]*/,
                      void* stream);
/* The same with a bias gradient that is the same bits on every launch: every slab of 512 rows
 * leaves its column sums in the workspace and a second kernel adds the slabs in slab order
 * (gb is written, not accumulated, and needs no zeroing; gb nullable, then no workspace).
 * row_scale (nullable, device [M]): gP[r,:] is stored multiplied by row_scale[r] -- the
 * destination normaliser, so that the backward aggregation that follows needs no per-edge
 * weight -- while gb sums the unscaled values.  Same reference lines. */
size_t spmi_act_bias_bwd_workspace_bytes(int64_t M, int64_t F);
int spmi_act_bias_bwd_ordered(const float* gY, const float* Y, int64_t M, int64_t F, int64_t ld,
                              int act, float slope, const float* row_scale /*nullable*/, float* gP,
                              float* gb /*nullable, device [F]*/, void* ws, size_t ws_bytes,
                              void* stream);

/* ------------------------------------------------------------------------
 * Recurrent cells, sequence length 1, batch = nodes.  Replace the pointwise
 * half of torch.nn.GRU / LSTM / RNN as called by SequentialDecoder.forward
 * and LSTM.forward (implicitTimeModels.py:144-158, 214-230), fused with the
 * NodeMemory gather `memory[:, ids, :]` (:52-54) and the detached write-back
 * `memory[:, ids, :] = h` (:48-50, :155-156).  The gate GEMMs
 * gi = x W_ih^T and gh = h W_hh^T are plain library GEMMs done by the caller.
 *
 * Gate order follows torch: GRU r,z,n ; LSTM i,f,g,o.
 *   h_mem / c_mem : one layer of the node memory, [n_mem_rows, H]
 *   ids           : nullable; row m of the batch uses memory row ids[m]
 *                   (NULL = identity, the full-snapshot case)
 *   h_out         : [M,H] new hidden state (input of the next layer / MLP)
 *   mem_write     : non-zero -> also store the new state into the memory rows
 *   saved         : nullable; when given, what the backward needs is stored
 *                   (GRU [M,5H]: r,z,n,hn,h_old ; LSTM [M,6H]: i,f,g,o,c_old,
 *                   tanh(c_new) ; RNN: nothing beyond h_out)
 * spmi_gather_rows gathers h_old rows into a dense [M,H] matrix for the
 * caller's h W_hh^T GEMM when ids is not the identity.
 * ---------------------------------------------------------------------- */
int spmi_gather_rows(const float* table, int64_t ld, const int64_t* ids, int64_t M, int64_t H,
                     float* out /*[M,H]*/, void* stream);
int spmi_gru_gate_fwd(const float* gi /*[M,3H]*/, const float* gh /*[M,3H]*/,
                      const float* b_ih /*nullable [3H]*/, const float* b_hh /*nullable [3H]*/,
                      float* h_mem, const int64_t* ids, int64_t M, int64_t H,
                      float* h_out, int mem_write, float* saved, void* stream);
int spmi_gru_gate_bwd(const float* gh_out /*[M,H] grad of h_out*/, const float* saved /*[M,5H]*/,
                      int64_t M, int64_t H, float* g_gi /*[M,3H]*/, float* g_gh /*[M,3H]*/,
                      float* g_h /*nullable [M,H]: direct path gh_out*z*/, void* stream);
/* The gate gradients of spmi_gru_gate_bwd as one matrix g4 [M,4H] = [d a_n | d a_r | d a_z |
 * d a_n * r]: columns 0..3H are g_gi with the gates in the order n, r, z, columns H..4H are
 * g_gh (r, z, n); a third less to write and to read back than the two matrices.  Replaces
 * the same autograd step of torch.nn.GRU (implicitTimeModels.py:185-187) as spmi_gru_gate_bwd. */
int spmi_gru_gate_bwd_packed(const float* gh_out /*[M,H]*/, const float* saved /*[M,5H]*/, int64_t M,
                             int64_t H, float* g4 /*[M,4H]*/, float* g_h /*nullable [M,H]*/,
                             void* stream);
/* The same packed gradients from the gate values spmi_gru_rows_fwd_saved keeps: saved [M, 4, H] =
 * the planes r, z, n, W_hn h + b_hn of every row, the old state read from h_old [M, H] (row stride
 * ldh), the matrix the forward read it from.  H and ldh multiples of 4, pointers 16-byte
 * aligned.  Same autograd step of torch.nn.GRU (implicitTimeModels.py:185-187). */
int spmi_gru_gate_bwd_packed_q(const float* gh_out /*[M,H]*/, const float* saved /*[M,4,H]*/,
                               const float* h_old, int64_t ldh, int64_t M, int64_t H,
                               float* g4 /*[M,4H]*/, void* stream);
int spmi_lstm_gate_fwd(const float* gi /*[M,4H]*/, const float* gh /*[M,4H]*/,
                       const float* b_ih, const float* b_hh, float* h_mem, float* c_mem,
                       const int64_t* ids, int64_t M, int64_t H, float* h_out, float* c_out,
                       int mem_write, float* saved, void* stream);
int spmi_lstm_gate_bwd(const float* gh_out /*[M,H]*/, const float* gc_out /*nullable [M,H]*/,
                       const float* saved /*[M,6H]*/, int64_t M, int64_t H,
                       float* g_gates /*[M,4H] = grad of gi and of gh*/,
                       float* g_c /*nullable [M,H]: grad of c_old*/, void* stream);
int spmi_rnn_gate_fwd(const float* gi /*[M,H]*/, const float* gh /*[M,H]*/, const float* b_ih,
                      const float* b_hh, float* h_mem, const int64_t* ids, int64_t M, int64_t H,
                      float* h_out, int mem_write, void* stream);
int spmi_rnn_gate_bwd(const float* gh_out /*[M,H]*/, const float* h_out /*[M,H]*/, int64_t M,
                      int64_t H, float* g_gates /*[M,H]*/, void* stream);

/* ------------------------------------------------------------------------
 * One whole recurrent layer step in one kernel: both gate GEMMs on the fp32
 * matrix cores (exact fp32 FMA chains), gate math, node-memory read and the
 * in-place write-back.  Same reference lines as the gate kernels above, plus
 * the torch.nn.GRU/LSTM/RNN call itself (implicitTimeModels.py:154,225).
 *   cell   : 0 = GRU, 1 = LSTM, 2 = RNN(tanh)
 *   x      : [M, I] with leading dimension ldx
 *   wt_ih  : weight_ih transposed, [I, G*H]; wt_hh: weight_hh transposed, [H, G*H]
 *            (G = 3 / 4 / 1; torch gate order)
 *   others : as for the gate kernels; `saved` has the same layout, so the gate
 *            backward kernels apply unchanged.
 * Shapes: I % 32 == 0 and H % 32 == 0 (spmi_cell_fused_supported tells);
 * other shapes go through library GEMMs + the gate kernels.
 * ---------------------------------------------------------------------- */
int spmi_cell_fused_supported(int64_t I, int64_t H);
int spmi_cell_fused_fwd(int cell, const float* x, int64_t ldx, int64_t I, const float* wt_ih,
                        const float* wt_hh, const float* b_ih, const float* b_hh, float* h_mem,
                        float* c_mem, const int64_t* ids, int64_t M, int64_t H, float* h_out,
                        float* c_out, int mem_write, float* saved, void* stream);

/* Same layer step with the gate GEMMs on the bf16 matrix cores at fp32 accuracy: every
 * fp32 operand is split exactly into three bf16 pieces (spmi_split_bf16x3 for the
 * weights, in-kernel for x and the state) and each product is summed from the six piece
 * products above 2^-24; pre-activations agree with an fp32 FMA chain to a few fp32
 * roundings.  ws_ih / ws_hh: weight_ih [G*H, I] / weight_hh [G*H, H] (torch's own
 * layout) passed through spmi_split_bf16x3, which writes [cols/16][3][2][rows][8] bfloat16
 * (K-step, piece, half of the K-step, row, k; 3 * rows * cols elements).  Other arguments as spmi_cell_fused_fwd, same shape limits. */
int spmi_split_bf16x3(const float* w /*device [rows, cols]*/, int64_t rows, int64_t cols,
                      void* out /*device bf16, 3*rows*cols*/, void* stream);
int spmi_cell_split_fwd(int cell, const float* x, int64_t ldx, int64_t I, const void* ws_ih,
                        const void* ws_hh, const float* b_ih, const float* b_hh, float* h_mem,
                        float* c_mem, const int64_t* ids, int64_t M, int64_t H, float* h_out,
                        float* c_out, int mem_write, float* saved, void* stream);

/* Same layer step (same reference lines as spmi_cell_fused_fwd) with half the matrix work
 * of spmi_cell_split_fwd: three fp16 matrix instructions per term instead of six bf16 ones.
 * An operand v travels as p0 = fp16(v), p1 = fp16(v / 64), p2 = fp16((v - p0) * 64) and
 * x*w = p0(x)*p0(w) + p1(x)*p2(w) + p2(x)*p1(w) with fp32 accumulation: each product is off
 * by at most about 2^-20 |x w|, and sums of 256 terms measure 3-8x closer to float64 than an
 * fp32 FMA chain.  Domain: the state and the weights within +-65504 (a larger operand turns
 * into infinity and its output row into inf/NaN; spmi_cell_split_fwd has no such limit).  The
 * rows of x may hold any finite value: with x_bounded = 0 a workgroup first reads its rows of
 * x and stages a row that reaches 2^15 scaled by a power of two (its largest entry lands in
 * [2^13, 2^14)); the partial sums are scaled back, exactly, before the state terms are added.
 * Entries of such a row resolve to 2^-44 of the row's largest.  x_bounded = 1 skips that pass
 * for callers that know |x| < 2^15, e.g. when x is the output of another recurrent layer.
 * Weights below 2^-9 in magnitude keep an absolute resolution of 2^-31 rather than the relative
 * one.  ws_ih / ws_hh come from spmi_split_f16x3 (same tiling as spmi_split_bf16x3:
 * [cols/16][3][2][rows][8] 16-bit words).  Arguments and shape limits as
 * spmi_cell_split_fwd. */
int spmi_split_f16x3(const float* w /*device [rows, cols]*/, int64_t rows, int64_t cols,
                       void* out /*device 16-bit words, 3*rows*cols*/, void* stream);
/* The same planes of a matrix given by strides (element (r, c) at w[r * row_stride + c *
 * col_stride], so a transposed view such as GraphConv's [in, out] weight needs no copy), and,
 * if amax_bits is not null, the largest |w| for the caller's domain check: the kernel takes
 * the maximum of *amax_bits (which the caller has set to 0) and the bit patterns of |w|, which
 * order like the values with NaN on top.  A training loop re-splits every weight after every
 * optimiser step (the reference's loop, examples/torch/citation_prediction.py:98-101, updates
 * them all); this keeps that to one launch per weight. */
int spmi_split_f16x3_strided(const float* w, int64_t rows, int64_t cols, int64_t row_stride,
                             int64_t col_stride, void* out, uint32_t* amax_bits /*nullable*/,
                             void* stream);
int spmi_cell_f16x3_fwd(int cell, const float* x, int64_t ldx, int64_t I, const void* ws_ih,
                        const void* ws_hh, const float* b_ih, const float* b_hh, float* h_mem,
                        float* c_mem, const int64_t* ids, int64_t M, int64_t H, float* h_out,
                        float* c_out, int mem_write, float* saved, int x_bounded, void* stream);

/* The GRU layer step of spmi_cell_f16x3_fwd (same reference lines: torch.nn.GRU with
 * seq_len 1 as driven by SequentialDecoder.forward, implicitTimeModels.py:144-158, with
 * NodeMemory.get_memory / update_memory, :48-54) for the case the hot loop runs: ids = every
 * node, no gate values kept.  Same operand format, same gate functions, results equal to
 * spmi_cell_f16x3_fwd's up to the order of the matrix instructions; different data movement:
 * a workgroup keeps the r, z, n weight columns of 32 hidden units in LDS for its lifetime and
 * streams row tiles of x and the old state straight from memory into the matrix cores, so
 * no weight is staged per tile.  Rows are read by H / 32 workgroups and written in column
 * slices, hence NOT in place: h_out [M, H] must overlap neither h_in [M, H] nor x; a caller
 * that owns the node memory keeps two slabs per layer and swaps them.  Supported: I = H = 128
 * (spmi_gru_rows_supported).  ws_ih / ws_hh from spmi_split_f16x3; b_ih / b_hh nullable;
 * x_bounded as in spmi_cell_f16x3_fwd. */
int spmi_gru_rows_supported(int64_t I, int64_t H);
int spmi_gru_rows_fwd(const float* x /*device [M, I], row stride ldx*/, int64_t ldx, int64_t I,
                      const void* ws_ih, const void* ws_hh, const float* b_ih, const float* b_hh,
                      const float* h_in /*device [M, H]*/, int64_t M, int64_t H,
                      float* h_out /*device [M, H]*/, int x_bounded, void* stream);
/* The same step for a caller that will run a backward pass (torch.nn.GRU under autograd in the
 * reference's training loop, examples/torch/citation_prediction.py:96-101): besides h_out it
 * writes saved [M, 4, H] = the planes r, z, n, W_hn h + b_hn of every row, the input of
 * spmi_gru_gate_bwd_packed_q.  saved must overlap none of x, h_in, h_out. */
int spmi_gru_rows_fwd_saved(const float* x, int64_t ldx, int64_t I, const void* ws_ih, const void* ws_hh,
                            const float* b_ih, const float* b_hh, const float* h_in, int64_t M, int64_t H,
                            float* h_out, float* saved /*device [M, 4, H]*/, int x_bounded, void* stream);

/* The tanh RNN layer step on the same loop nest: torch.nn.RNN with seq_len 1 as driven by
 * SequentialDecoder.forward (implicitTimeModels.py:144-158), ids = every node, nothing kept:
 * h' = tanh(x W_ih^T + b_ih + h W_hh^T + b_hh).  One gate: as two fp16 planes (see
 * spmi_lstm_rows_fwd) all 128 columns fit LDS, so every row is read and split once.  h_out must
 * overlap neither h_in nor x.  ws_ih / ws_hh from spmi_split_f16x3 of the [H, I] / [H, H]
 * weights; biases nullable; I = H = 128; rows of x may hold any finite value (x_bounded is
 * accepted and ignored). */
int spmi_rnn_rows_supported(int64_t I, int64_t H);
int spmi_rnn_rows_fwd(const float* x /*device [M, I], row stride ldx*/, int64_t ldx, int64_t I,
                      const void* ws_ih, const void* ws_hh, const float* b_ih, const float* b_hh,
                      const float* h_in /*device [M, H]*/, int64_t M, int64_t H,
                      float* h_out /*device [M, H]*/, int x_bounded, void* stream);

/* The LSTM layer step on the same loop nest: torch.nn.LSTM with seq_len 1 as driven by
 * LSTM.forward (implicitTimeModels.py:214-230) with the two NodeMemory tables (:48-54), for
 * ids = every node and no gate values kept.  Gates i, f, g, o; c' = f c + i g; h' = o tanh(c').
 * Operands as two fp16 planes (the value and its residual * 2^6), the two small products of a
 * term in an accumulator set of their own: the same three matrix instructions per term and the
 * same accuracy as spmi_cell_f16x3_fwd, with the four gates' columns of 32 hidden units (128 KB)
 * resident in LDS.  h_out must overlap neither h_in nor x nor c_in; c_out may be c_in (a lane
 * reads the cell state it later writes) but no other input.  ws_ih / ws_hh from
 * spmi_split_f16x3 of the [4H, I] / [4H, H] weights; biases nullable; I = H = 128.  Rows of x
 * may hold any finite value; x_bounded is accepted for symmetry with spmi_gru_rows_fwd and
 * changes nothing (the form without the magnitude pass measured slower). */
int spmi_lstm_rows_supported(int64_t I, int64_t H);
int spmi_lstm_rows_fwd(const float* x /*device [M, I], row stride ldx*/, int64_t ldx, int64_t I,
                       const void* ws_ih, const void* ws_hh, const float* b_ih, const float* b_hh,
                       const float* h_in /*device [M, H]*/, const float* c_in /*device [M, H]*/,
                       int64_t M, int64_t H, float* h_out /*device [M, H]*/,
                       float* c_out /*device [M, H]*/, int x_bounded, void* stream);
/* The same step for a caller that will run a backward pass (torch.nn.LSTM under autograd in
 * the reference's training loop, examples/torch/citation_prediction.py:96-101): besides h_out
 * and c_out it writes saved [M, 6, H] = the planes i, f, g, o, the old cell state and
 * tanh(c') of every row, the input of spmi_lstm_gate_bwd.  saved must overlap none of x, h_in,
 * c_in, h_out, c_out.  (The tanh RNN needs no such variant: spmi_rnn_gate_bwd reads the
 * output of spmi_rnn_rows_fwd.) */
int spmi_lstm_rows_fwd_saved(const float* x, int64_t ldx, int64_t I, const void* ws_ih, const void* ws_hh,
                             const float* b_ih, const float* b_hh, const float* h_in, const float* c_in,
                             int64_t M, int64_t H, float* h_out, float* c_out,
                             float* saved /*device [M, 6, H]*/, int x_bounded, void* stream);

/* ------------------------------------------------------------------------
 * Dense projection of a tall matrix: out = act(x W^T + bias), x [M, K] fp32, W [N, K].
 * Replaces `torch.matmul(feat, weight)` inside dgl GraphConv.forward (call sites
 * staticGraphEncoder.py:113,116; likewise SGConv / SAGEConv's fc layers) and the Linear
 * layers of the decoders (simpleDecoder.py:43-46) for K in {32, 64, 128} with
 * N in {32, 64, 96, 128} or any multiple of 128, and K = 256, 384 or 512 with N a multiple
 * of 64 (wide outputs are computed in column slices by workgroups that share an XCD's L2;
 * K = 512 is two launches, the second adding its half of the sums to `out`, which holds
 * the first half in between).  fp32 results from the fp16 matrix cores with the three-plane
 * operand format of spmi_cell_f16x3_fwd (same accuracy).  The rows of x may hold any finite
 * fp32 value (each row, in chunks of 128 columns, is scaled by a power of two before it is
 * split and its sums by the inverse; a non-finite entry makes its own row non-finite);
 * |W| <= 1023;
 * w_planes = spmi_split_f16x3(W, N, K).  The weight planes stay in LDS and x streams from
 * HBM straight into the matrix cores, so the kernel runs at memory speed:
 * M*(K+N)*4 bytes.  act: 0 none, 1 relu, 2 leaky_relu(slope).  bias nullable.
 * row_scale (nullable, device [M]): row r of the result is multiplied by row_scale[r]
 * last, which is how GraphConv applies its source normaliser (`feat_src * norm` in dgl
 * GraphConv.forward) without a pass of its own or a per-edge weight.
 * ---------------------------------------------------------------------- */
int spmi_gemm_rows_supported(int64_t K, int64_t N);
/* The same projection for x stored as bfloat16 (bf16 feature storage of EvolveGCN-O): x is an
 * operand as it is, w_pieces = spmi_split_bf16x3(W, N, K) keeps the weight at fp32 accuracy
 * (three products per term), out is bfloat16 (out_bf16 = 1, rounded once) or float32.  K in
 * {32, 64, 128}, N in {32, 64, 96, 128}; ldx (elements) a multiple of 8. */
int spmi_gemm_rows_bf16_supported(int64_t K, int64_t N);
int spmi_gemm_rows_bf16(const void* x /*device bf16 [M, K]*/, int64_t ldx, int64_t M, int64_t K,
                        const void* w_pieces, int64_t N, const float* bias, const float* row_scale,
                        int act, float slope, void* out, int64_t ldo, int out_bf16, void* stream);
int spmi_gemm_rows(const float* x /*device [M, K], row stride ldx*/, int64_t ldx, int64_t M, int64_t K,
                   const void* w_planes, int64_t N, const float* bias /*device [N] or null*/,
                   const float* row_scale /*device [M] or null*/, int act, float slope,
                   float* out /*device [M, N], row stride ldo*/, int64_t ldo, void* stream);

/* ------------------------------------------------------------------------
 * Attention encoder.  Replace what dgl GATConv.forward runs on the graph
 * (reference call sites: GAT in staticGraphEncoder.py:170,174,190):
 * apply_edges(u_add_v('el','er')), LeakyReLU(negative_slope), edge_softmax
 * over the incoming edges of every destination, and their gradients.  The
 * weighted sum update_all(u_mul_e('ft','a'), sum) is spmi_gcn_spmm with
 * edge_w = one head of `attn`, one launch per head on the column slice of
 * that head.
 *   per-edge arrays are head-major [H, E] in the order of the CSR given;
 *   el [n_src, H], er [n_rows, H]: attention logits of sources / destinations;
 *   raw (nullable): el[src]+er[dst] before LeakyReLU, kept for the backward;
 *   heavy_rows / n_heavy / heavy_thresh: the hub-row list of spmi_spmm_plan
 *   (rows above the threshold are reduced by a whole workgroup).
 * Rows without entries are left untouched by the softmax and get zero from
 * the backward and from spmi_segment_sum.
 * ---------------------------------------------------------------------- */
int spmi_gat_edge_softmax(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t E,
                          const float* el, const float* er, int32_t H, float slope,
                          float* attn /*[H,E]*/, float* raw /*nullable [H,E]*/,
                          const int32_t* heavy_rows, int32_t n_heavy, int32_t heavy_thresh,
                          void* stream);
/* d_raw = attn * (d_attn - sum_row(attn * d_attn)) * LeakyReLU'(raw); d_er = row sums of d_raw */
int spmi_gat_edge_softmax_bwd(const int32_t* indptr, int64_t n_rows, int64_t E, const float* attn,
                              const float* raw, const float* d_attn, int32_t H, float slope,
                              float* d_raw /*[H,E]*/, float* d_er /*[n_rows,H]*/,
                              const int32_t* heavy_rows, int32_t n_heavy, int32_t heavy_thresh,
                              void* stream);
/* out[r,h] = sum of vals[h, j] over the entries j of row r (d_el on the by-source CSR) */
int spmi_segment_sum(const int32_t* indptr, int64_t n_rows, int64_t E, const float* vals /*[H,E]*/,
                     int32_t H, float* out /*[n_rows,H]*/, const int32_t* heavy_rows,
                     int32_t n_heavy, int32_t heavy_thresh, void* stream);
/* out[h,j] = <A[row(j), h*D:(h+1)*D], B[indices[j], h*D:(h+1)*D]>: gradient of the
 * attention weights, A = d(output) rows of destinations, B = projected source rows.
 * D must be a power of two below 64 or a multiple of 64, H <= 8, H*D <= 512. */
int spmi_sddmm_dot_supported(int32_t H, int32_t D);
int spmi_sddmm_dot(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t E,
                   const float* A, int64_t lda, const float* B, int64_t ldb, int32_t H, int32_t D,
                   float* out /*[H,E]*/, void* stream);

/* ------------------------------------------------------------------------
 * Weight-gradient GEMM: C[M,N] (+)= A[K,M]^T B[K,N], row-major, K = number
 * of nodes, M and N = layer widths (<= 4096).  Replaces the library GEMM that
 * autograd runs for dW = X^T dY of every dense layer on the path in the
 * training loop (examples/torch/citation_prediction.py:96-100): GraphConv's
 * weight (staticGraphEncoder.py:113,116), the gate weights of nn.GRU/LSTM/RNN
 * (implicitTimeModels.py:185-187,256-258,284-286), the MLP's Linear layers
 * (simpleDecoder.py:60-72).  Exact fp32 products and sums on the matrix
 * cores; K is split over the grid and the partial blocks are summed in a
 * fixed order (bit-reproducible).  accumulate != 0 adds to C.
 * ---------------------------------------------------------------------- */
size_t spmi_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t K);
int spmi_gemm_tn(const float* A /*[K,M]*/, int64_t lda, const float* B /*[K,N]*/, int64_t ldb,
                 int64_t K, int64_t M, int64_t N, float* C /*[M,N]*/, int64_t ldc, int accumulate,
                 void* ws, size_t ws_bytes, void* stream);
/* The same product with the column sums of its operands on the side: col_sum_a[m] = sum_k A[k,m],
 * col_sum_b[n] = sum_k B[k,n] (each nullable).  In the training loop the bias gradient of a dense
 * layer is the column sum of the very matrix its weight gradient multiplies (db = sum_rows dY
 * next to dW = X^T dY: the autograd of `+ bias` in dgl GraphConv.forward, nn.GRU/LSTM/RNN and
 * nn.Linear, same reference lines as spmi_gemm_tn), so the sum costs one add per element
 * already in a register instead of another pass over a [nodes, width] matrix.  Fixed
 * summation order (bit-reproducible).  Workspace as for spmi_gemm_tn.
 * fp32_instruction != 0: products on the fp32 matrix instruction, as spmi_gemm_tn.
 * fp32_instruction == 0: products on the bf16 matrix cores at fp32 accuracy: every operand
 * value is split exactly into three bf16 pieces and a product is summed from the six piece
 * products above 2^-24 of its size, fp32 accumulation; no range limit beyond bfloat16's own
 * (fp32's exponent range); 2-3x faster on these shapes (HBM-bound instead of matrix-bound).
 * fp32_instruction == 2: as 0, except that outputs made of whole 128 x 128 blocks (M, N multiples
 * of 128, M * N <= 512 * 512, aligned operands) are multiplied in three fp16 planes per value
 * (three products per term, half the matrix work, fp32-chain accuracy) under power-of-two scales
 * that follow the operands' magnitudes through the rows: the error stays relative to the sum of
 * absolute terms for operands of any size with |a| |b| within 2^+-100. */
int spmi_gemm_tn_sums(const float* A /*[K,M]*/, int64_t lda, const float* B /*[K,N]*/, int64_t ldb,
                      int64_t K, int64_t M, int64_t N, float* C /*[M,N]*/, int64_t ldc,
                      int accumulate, float* col_sum_a /*nullable [M]*/,
                      float* col_sum_b /*nullable [N]*/, int fp32_instruction, void* ws,
                      size_t ws_bytes, void* stream);
/* Two such products of one shape over the same K rows in ONE launch: C0 = A0^T B0 and
 * C1 = A1^T B1 (bf16-piece products, col_sum_a0 / col_sum_a1 nullable).  The waves of a chunk
 * of K run side by side for both, so columns the operands share are read from memory once: the
 * two weight gradients of a GRU layer (implicitTimeModels.py:185-187 under autograd) are
 * dW_ih = G[:, 0:3H]^T x and dW_hh = G[:, H:4H]^T h over one packed gate-gradient matrix G
 * whose middle 2H columns both read (3 GB instead of 4 per layer at 1M rows, H = 128).
 * M and N multiples of 128, M * N <= 512 * 512 (spmi_gemm_tn_pair_supported); operands 16-byte
 * aligned, leading dimensions multiples of 4.  Bit-reproducible; the split of K differs from
 * the single product's, so the sums agree with it to fp32 rounding, not bit for bit.
 * f16_planes != 0: the arithmetic of spmi_gemm_tn_sums' fp32_instruction == 2. */
int spmi_gemm_tn_pair_supported(int64_t M, int64_t N);
size_t spmi_gemm_tn_pair_workspace_bytes(int64_t M, int64_t N, int64_t K);
int spmi_gemm_tn_pair_sums(const float* A0, int64_t lda0, const float* B0, int64_t ldb0, const float* A1,
                           int64_t lda1, const float* B1, int64_t ldb1, int64_t K, int64_t M, int64_t N,
                           float* C0, int64_t ldc0, float* C1, int64_t ldc1, float* col_sum_a0 /*nullable [M]*/,
                           float* col_sum_a1 /*nullable [M]*/, int f16_planes, void* ws, size_t ws_bytes,
                           void* stream);

/* ------------------------------------------------------------------------
 * Output head in inference, one kernel: the MLP of
 * spiro/dtdg/models/decoder/torch/simpleDecoder.py:45-79 ([Linear,
 * BatchNorm1d, ReLU, Dropout] x k + Linear) as called from
 * SequentialDecoder.forward (implicitTimeModels.py:157-158), in eval mode:
 * BatchNorm is folded into the preceding Linear by the caller, Dropout is the
 * identity.  x [M, I] is read once, out [M, width of the last layer] written.
 *   layer 1: w1t = weight transposed and zero-padded to [I][P], P = 8 / 16 / 32
 *            (the smallest that holds out1), b1 [P];
 *   layers 2..: rest_packed = per layer the weight in torch's [out, in]
 *            layout followed by its bias [out] (zeros if none), back to back,
 *            device memory; rest_out / rest_relu are HOST arrays of n_rest ints;
 *   relu flags per layer; final_act: 0 none, 1 sigmoid, 2 relu, 3 tanh.
 * Limits (spmi_mlp_head_supported): I a multiple of 32, widths <= 32, at most
 * 7 layers after the first; I * P <= 8192 unless I is one of the widths the
 * matrix-core kernel is built for (multiples of 32 up to 256, 320, 384, 448,
 * 512), which has no such bound.
 * ---------------------------------------------------------------------- */
int spmi_mlp_head_supported(int64_t I, int32_t out1, int32_t n_rest, const int32_t* rest_out);
int spmi_mlp_head_fwd(const float* x, int64_t ldx, int64_t M, int64_t I, const float* w1t,
                      const float* b1, int32_t out1, int32_t relu1, int32_t n_rest,
                      const float* rest_packed, const int32_t* rest_out, const int32_t* rest_relu,
                      int32_t final_act, float* out, int64_t ldo, void* stream);

/* ------------------------------------------------------------------------
 * Output head in training: the narrow [Linear, BatchNorm1d, ReLU, Dropout] blocks of the same
 * MLP (simpleDecoder.py:60-72) with batch statistics, as the loop of
 * examples/torch/citation_prediction.py:96-100 runs them, forward and backward; widths 1..32.
 * The matrices between the layers (z = a Linear's output, its gradient dz, the gradient dy of a
 * BatchNorm's output) are float32 [M, pad] with rows padded to pad = 8, 16 or 32 floats, the
 * padding holding zeros, 16-byte aligned.  `stats` is 128 floats per BatchNorm: mean | rstd |
 * scale = gamma * rstd | shift = beta - mean * scale (32 each, zero past the width).  `mask`
 * (nullable) holds the dropout factors [M, wi] (0 or 1 / (1 - p)), unpadded.  Sums over rows are
 * taken in a fixed order.  ws: spmi_mlp_train_workspace_bytes() bytes.
 *   spmi_bn_col_stats      shifted column sums of a z that another kernel wrote -> ws, and z
 *                          copied into rows of w_pad floats
 *   spmi_bn_finalize       ws -> stats; moves running_mean / running_var (nullable) with
 *                          `momentum` (unbiased variance), as BatchNorm1d.forward does.  pivot:
 *                          row 0 of the matrix the sums were taken of
 *   spmi_bn_act_linear_fwd out = Linear(dropout(relu(z * scale + shift))), W [wo_pad, wi_pad] and
 *                          b [wo_pad] (nullable) zero-padded like the rows;
 *                          ws_stats (nullable): the shifted column sums of out
 *   spmi_bn_act_linear_bwd g = gradient of that out -> dy (gradient of the BatchNorm output)
 *                          and sums[1120] = d beta [32] | d gamma [32] | db [32] | dW [32][32]
 *   spmi_bn_bwd_apply      dz = scale * (dy - d beta / M - x_hat * d gamma / M)
 * ---------------------------------------------------------------------- */
size_t spmi_mlp_train_workspace_bytes(void);
int spmi_bn_col_stats(const float* z /*[M, >= w_pad], row stride ldz*/, int64_t ldz, int64_t M,
                      int64_t W, int64_t w_pad, float* packed /*[M, w_pad]: z, zero past W*/,
                      float* ws, size_t ws_bytes, void* stream);
int spmi_bn_finalize(const float* ws, const float* pivot, int64_t M, int64_t W, const float* gamma,
                     const float* beta, float eps, float momentum, float* running_mean,
                     float* running_var, float* stats, void* stream);
int spmi_bn_act_linear_fwd(const float* z, int64_t M, int64_t wi, int64_t wi_pad, const float* stats,
                           const float* mask, const float* W, const float* b, int64_t wo,
                           int64_t wo_pad, float* out, float* ws_stats, void* stream);
int spmi_bn_act_linear_bwd(const float* z, int64_t M, int64_t wi, int64_t wi_pad, const float* stats,
                           const float* mask, const float* W, int64_t wo, int64_t wo_pad,
                           const float* g, float* dy, float* ws, size_t ws_bytes, float* sums,
                           void* stream);
int spmi_bn_bwd_apply(const float* dy, const float* z, int64_t M, int64_t W, int64_t w_pad,
                      const float* stats, const float* sums, float* dz, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPMI_H */
