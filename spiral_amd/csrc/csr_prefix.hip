// Cumulative snapshots for gfx950: the CSR of a time prefix of an edge table, cut out of
// one table-wide CSR instead of being sorted again.
//
// What it replaces in the reference: VEInvariantDTDG.dispatcher(t) slices the time-sorted
// edge table to `edges[:anchor[t]]` and builds a new dgl.graph from the slice on every
// call (spiro/dtdg/types.py:183-198), so DGL regroups the same edges again and again;
// CitationGraph.dispatcher builds snapshot t-1 once more only to read its in-degrees
// (types.py:239-242).
//
// The table-wide CSR is a stable grouping, so inside a row the edge positions (eid)
// ascend; the edges of the prefix [0, e_cut) therefore form a prefix of every row.  Per
// snapshot that leaves: one binary search per row (prefix length = degree), a scan
// (row pointers) and a gather that packs the row prefixes (indices / eid).  The packed
// arrays equal, bit for bit, what the radix sort produces on the prefix.  HBM-bound
// integer work: E_t * 8 B (read + write, twice with eid) + N * 12 B.

#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kRowsPerThread = 4;
constexpr int kRowsPerBlock = kThreads * kRowsPerThread;
constexpr int kOutPerThread = 8;
constexpr int kCopyTile = kThreads * kOutPerThread;
constexpr int kSpanMax = 2048;  // rows of a copy tile staged in LDS

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t t = __shfl_up(v, off);
    if (lane >= off) v += t;
  }
  return v;
}

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_wave,
                                                         uint32_t* total) {
  uint32_t incl = wave_inclusive_scan(v);
  if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t base = 0, sum = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    uint32_t c = s_wave[w];
    if (w < static_cast<int>(threadIdx.x >> 6)) base += c;
    sum += c;
  }
  *total = sum;
  __syncthreads();
  return base + incl - v;
}

// Number of entries of the row [lo, hi) whose edge position is below e_cut.
__device__ __forceinline__ int32_t row_prefix_len(const int32_t* __restrict__ eid, int32_t lo,
                                                  int32_t hi, int32_t e_cut) {
  if (lo == hi) return 0;
  if (eid[hi - 1] < e_cut) return hi - lo;
  int32_t a = lo, b = hi - 1;  // eid[b] >= e_cut; the first such position is in [a, b]
  while (a < b) {
    const int32_t m = a + ((b - a) >> 1);
    if (eid[m] < e_cut) a = m + 1; else b = m;
  }
  return a - lo;
}

// ---- kernel 1: prefix length of every row (+ per-block sums) --------------------------
template <bool DEG64>
__global__ __launch_bounds__(kThreads) void prefix_count_kernel(
    const int32_t* __restrict__ indptr_all, const int32_t* __restrict__ eid_all, int64_t n_rows,
    int32_t e_cut, int32_t* __restrict__ cnt, int64_t* __restrict__ deg64,
    uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t s_wave[kWaves];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kRowsPerBlock;
  uint32_t sum = 0;
#pragma unroll
  for (int j = 0; j < kRowsPerThread; ++j) {
    const int64_t r = base + j * kThreads + threadIdx.x;
    if (r < n_rows) {
      const int32_t c = row_prefix_len(eid_all, indptr_all[r], indptr_all[r + 1], e_cut);
      if constexpr (DEG64) deg64[r] = c; else cnt[r] = c;
      sum += static_cast<uint32_t>(c);
    }
  }
  if constexpr (!DEG64) {
    uint32_t total;
    block_exclusive_scan(sum, s_wave, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
  }
}

// ---- kernel 2: exclusive scan of the block sums (one workgroup) -----------------------
// block_sums[n_blocks] receives the grand total; status is raised when the total differs
// from e_cut (an edge of the prefix ends in a row outside [0, n_rows)).
__global__ __launch_bounds__(kThreads) void prefix_scan_sums_kernel(uint32_t* __restrict__ block_sums,
                                                                    int n_blocks, int32_t e_cut,
                                                                    int32_t* status) {
  __shared__ uint32_t s_wave[kWaves];
  __shared__ uint32_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < n_blocks; base += kThreads) {
    const int i = base + threadIdx.x;
    const uint32_t c = i < n_blocks ? block_sums[i] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan(c, s_wave, &total);
    const uint32_t carry = s_carry;
    if (i < n_blocks) block_sums[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) s_carry = carry + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    block_sums[n_blocks] = s_carry;
    if (s_carry != static_cast<uint32_t>(e_cut)) *status = 1;
  }
}

// ---- kernel 3: row pointers of the prefix ---------------------------------------------
__global__ __launch_bounds__(kThreads) void prefix_indptr_kernel(
    const int32_t* __restrict__ cnt, const uint32_t* __restrict__ block_sums, int n_blocks,
    int64_t n_rows, int32_t* __restrict__ indptr) {
  __shared__ uint32_t s_wave[kWaves];
  const int64_t first = static_cast<int64_t>(blockIdx.x) * kRowsPerBlock +
                        static_cast<int64_t>(threadIdx.x) * kRowsPerThread;
  uint32_t c[kRowsPerThread];
  uint32_t sum = 0;
#pragma unroll
  for (int j = 0; j < kRowsPerThread; ++j) {
    c[j] = (first + j < n_rows) ? static_cast<uint32_t>(cnt[first + j]) : 0u;
    sum += c[j];
  }
  uint32_t total;
  uint32_t run = block_sums[blockIdx.x] + block_exclusive_scan(sum, s_wave, &total);
#pragma unroll
  for (int j = 0; j < kRowsPerThread; ++j) {
    if (first + j < n_rows) indptr[first + j] = static_cast<int32_t>(run);
    run += c[j];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) indptr[n_rows] = static_cast<int32_t>(block_sums[n_blocks]);
}

// Largest i in [lo, hi] with ptr[i] <= j (ptr ascending, ptr[lo] <= j).
template <typename P>
__device__ __forceinline__ int32_t last_not_above(P ptr, int32_t lo, int32_t hi, int32_t j) {
  while (lo < hi) {
    const int32_t m = lo + ((hi - lo + 1) >> 1);
    if (ptr[m] <= j) lo = m; else hi = m - 1;
  }
  return lo;
}

// ---- kernel 4: pack the row prefixes ---------------------------------------------------
// A workgroup owns kCopyTile consecutive output positions.  The rows they belong to are
// looked up in LDS when the tile spans at most kSpanMax rows, in global memory otherwise
// (a tile crossing a long run of empty rows).
__global__ __launch_bounds__(kThreads) void prefix_copy_kernel(
    const int32_t* __restrict__ indptr_all, const int32_t* __restrict__ indices_all,
    const int32_t* __restrict__ eid_all, const int32_t* __restrict__ indptr, int64_t n_rows,
    int32_t n_cols, const uint32_t* __restrict__ total_ptr, int32_t* __restrict__ indices,
    int32_t* __restrict__ eid, int32_t* status) {
  __shared__ int32_t s_ptr[kSpanMax + 1];
  __shared__ int32_t s_start[kSpanMax];
  const int32_t total = static_cast<int32_t>(*total_ptr);
  const int32_t j0 = static_cast<int32_t>(blockIdx.x) * kCopyTile;
  if (j0 >= total) return;
  const int32_t j1 = min(j0 + kCopyTile, total);  // exclusive
  const int32_t last_row = static_cast<int32_t>(n_rows) - 1;
  const int32_t row_lo = last_not_above(indptr, 0, last_row, j0);
  const int32_t row_hi = last_not_above(indptr, row_lo, last_row, j1 - 1);
  const int32_t span = row_hi - row_lo + 1;
  const bool staged = span <= kSpanMax;
  if (staged) {
    for (int i = threadIdx.x; i <= span; i += kThreads) s_ptr[i] = indptr[row_lo + i];
    for (int i = threadIdx.x; i < span; i += kThreads) s_start[i] = indptr_all[row_lo + i];
    __syncthreads();
  }
  // Lane-consecutive output positions: coalesced stores, and loads that are contiguous
  // inside a row.  One search per position; consecutive positions of a lane are a whole
  // workgroup apart, so the previous row only bounds the search from below.
  bool bad = false;
  int32_t i = 0, r = row_lo;
#pragma unroll 2
  for (int k = 0; k < kOutPerThread; ++k) {
    const int32_t j = j0 + k * kThreads + static_cast<int32_t>(threadIdx.x);
    if (j >= j1) break;
    int32_t src;
    if (staged) {
      i = last_not_above(s_ptr, i, span - 1, j);
      src = s_start[i] + (j - s_ptr[i]);
    } else {
      r = last_not_above(indptr, r, row_hi, j);
      src = indptr_all[r] + (j - indptr[r]);
    }
    const int32_t v = indices_all[src];
    bad |= static_cast<uint32_t>(v) >= static_cast<uint32_t>(n_cols);
    indices[j] = v;
    if (eid) eid[j] = eid_all[src];
  }
  if (bad) *status = 1;
}

int check_prefix_args(int64_t n_rows, int64_t e_cut, const char* who) {
  SPMI_REQUIRE(n_rows >= 1 && n_rows < (int64_t{1} << 31) && e_cut >= 0 &&
                   e_cut < (int64_t{1} << 31),
               SPMI_ERR_INVALID_ARG, "%s: bad sizes n_rows=%lld e_cut=%lld", who,
               (long long)n_rows, (long long)e_cut);
  return SPMI_OK;
}

int n_row_blocks(int64_t n_rows) { return static_cast<int>((n_rows + kRowsPerBlock - 1) / kRowsPerBlock); }

}  // namespace

extern "C" {

size_t spmi_csr_prefix_workspace_bytes(int64_t n_rows) {
  if (n_rows < 1) return 0;
  return spmi::align_up(sizeof(int32_t) * (size_t)n_rows, 256) +
         spmi::align_up(sizeof(uint32_t) * (size_t)(n_row_blocks(n_rows) + 1), 256);
}

int spmi_csr_prefix(const int32_t* indptr_all, const int32_t* indices_all, const int32_t* eid_all,
                    int64_t n_rows, int64_t n_cols, int64_t e_cut, int32_t* indptr,
                    int32_t* indices, int32_t* eid, void* ws, size_t ws_bytes, int32_t* status,
                    void* stream) {
  if (int rc = check_prefix_args(n_rows, e_cut, "spmi_csr_prefix")) return rc;
  SPMI_REQUIRE(n_cols >= 1 && n_cols < (int64_t{1} << 31), SPMI_ERR_INVALID_ARG,
               "spmi_csr_prefix: bad n_cols=%lld", (long long)n_cols);
  SPMI_REQUIRE(indptr_all && indptr && status, SPMI_ERR_INVALID_ARG,
               "spmi_csr_prefix: null indptr_all/indptr/status");
  hipStream_t st = spmi::as_stream(stream);
  if (e_cut == 0) {
    SPMI_CHECK_HIP(hipMemsetAsync(indptr, 0, sizeof(int32_t) * (size_t)(n_rows + 1), st));
    return SPMI_OK;
  }
  SPMI_REQUIRE(indices_all && eid_all && indices && ws, SPMI_ERR_INVALID_ARG,
               "spmi_csr_prefix: null pointer argument");
  SPMI_REQUIRE(ws_bytes >= spmi_csr_prefix_workspace_bytes(n_rows), SPMI_ERR_WORKSPACE,
               "spmi_csr_prefix: workspace %zu < required %zu", ws_bytes,
               spmi_csr_prefix_workspace_bytes(n_rows));
  char* p = static_cast<char*>(ws);
  int32_t* cnt = reinterpret_cast<int32_t*>(p);
  uint32_t* sums =
      reinterpret_cast<uint32_t*>(p + spmi::align_up(sizeof(int32_t) * (size_t)n_rows, 256));
  const int nb = n_row_blocks(n_rows);
  prefix_count_kernel<false><<<nb, kThreads, 0, st>>>(indptr_all, eid_all, n_rows, (int32_t)e_cut,
                                                      cnt, nullptr, sums);
  SPMI_CHECK_LAUNCH("prefix_count");
  prefix_scan_sums_kernel<<<1, kThreads, 0, st>>>(sums, nb, (int32_t)e_cut, status);
  SPMI_CHECK_LAUNCH("prefix_scan_sums");
  prefix_indptr_kernel<<<nb, kThreads, 0, st>>>(cnt, sums, nb, n_rows, indptr);
  SPMI_CHECK_LAUNCH("prefix_indptr");
  const int tiles = static_cast<int>((e_cut + kCopyTile - 1) / kCopyTile);
  prefix_copy_kernel<<<tiles, kThreads, 0, st>>>(indptr_all, indices_all, eid_all, indptr, n_rows,
                                                 (int32_t)n_cols, sums + nb, indices, eid, status);
  SPMI_CHECK_LAUNCH("prefix_copy");
  return SPMI_OK;
}

int spmi_csr_prefix_degree(const int32_t* indptr_all, const int32_t* eid_all, int64_t n_rows,
                           int64_t e_cut, int64_t* deg, void* stream) {
  if (int rc = check_prefix_args(n_rows, e_cut, "spmi_csr_prefix_degree")) return rc;
  SPMI_REQUIRE(indptr_all && deg, SPMI_ERR_INVALID_ARG, "spmi_csr_prefix_degree: null pointer");
  hipStream_t st = spmi::as_stream(stream);
  if (e_cut == 0) {
    SPMI_CHECK_HIP(hipMemsetAsync(deg, 0, sizeof(int64_t) * (size_t)n_rows, st));
    return SPMI_OK;
  }
  SPMI_REQUIRE(eid_all, SPMI_ERR_INVALID_ARG, "spmi_csr_prefix_degree: null eid_all");
  prefix_count_kernel<true><<<n_row_blocks(n_rows), kThreads, 0, st>>>(
      indptr_all, eid_all, n_rows, (int32_t)e_cut, nullptr, deg, nullptr);
  SPMI_CHECK_LAUNCH("prefix_count(degree)");
  return SPMI_OK;
}

}  // extern "C"
