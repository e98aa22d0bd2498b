// Snapshot graph container kernels for gfx950: degrees, COO -> CSR, gcn norm and
// the load-balance plan of the aggregation.
//
// What they replace in the reference: dgl.graph((src,dst)) and DGL's lazy
// COOToCSR reached from spiro/dtdg/types.py:191, DGLGraph.in_degrees()
// (types.py:241) and the degree/norm lines of dgl GraphConv.forward (call sites
// spiro/dtdg/models/encoder/implicitTimeEncoder/staticGraphEncoder.py:113,116).
//
// Design: one stable LSB radix sort of (key = row id; values = other endpoint and,
// when wanted, edge position) with digits of at most 8 bits.  Per pass: a per-tile digit histogram, a
// per-digit scan over tiles, and a scatter that ranks keys inside the tile with
// wave ballots (stable, no atomics on the data path), stages the tile in LDS in
// digit order and writes each digit run coalesced.  Row pointers then fall out
// of the sorted keys as run boundaries, so degrees need no atomics either and
// every output is bit-reproducible.  All of it is HBM-bound integer work.

#include <algorithm>

#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kKeysPerThread = 8;
constexpr int kTile = kThreads * kKeysPerThread;  // keys per workgroup per pass
constexpr int kMaxDigits = 256;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// Keys enter as the caller's int64 node ids (first pass) or as the library's
// own validated uint32 keys (later passes).  An id outside [0,N) raises the
// status word and is clamped so that no later access leaves a buffer.
template <typename T>
__device__ __forceinline__ uint32_t checked_key(T raw, uint32_t n, int32_t* status) {
  if constexpr (sizeof(T) == 8) {
    if (static_cast<uint64_t>(raw) >= static_cast<uint64_t>(n)) {
      *status = 1;
      return n - 1;
    }
    return static_cast<uint32_t>(raw);
  } else {
    return static_cast<uint32_t>(raw);
  }
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  const int lane = lane_id();
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t t = __shfl_up(v, off);
    if (lane >= off) v += t;
  }
  return v;
}

// Exclusive scan of one value per thread over a 256-thread workgroup.
// s_wave needs kWaves entries; *total receives the workgroup sum.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_wave,
                                                         uint32_t* total) {
  uint32_t incl = wave_inclusive_scan(v);
  if (lane_id() == 63) s_wave[wave_id()] = incl;
  __syncthreads();
  uint32_t base = 0, sum = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    uint32_t c = s_wave[w];
    if (w < wave_id()) base += c;
    sum += c;
  }
  *total = sum;
  __syncthreads();
  return base + incl - v;
}

// The lanes of this wave row whose (valid) digit equals this lane's, as two 32-bit halves,
// and how many of them sit below this lane.  Per bit: one sign-extending bit-field extract
// (0 or all ones), one compare for the ballot, and xnor + and per half; the ranking of a
// radix pass is bound by exactly this vector work, not by memory.
struct WaveMatch {
  uint32_t lo, hi;      // lanes with the same digit
  uint32_t below;       // ... of which below this lane
  __device__ __forceinline__ uint32_t size() const { return __popc(lo) + __popc(hi); }
};

template <int BITS>
__device__ __forceinline__ WaveMatch wave_match(uint32_t d, bool valid) {
  // `d` must be 0 in lanes that are not valid (they then never raise a ballot bit)
  const uint64_t v = __ballot(valid);
  uint32_t diff_lo = 0, diff_hi = 0;   // lanes that differ from this one in some bit
#pragma unroll
  for (int bb = 0; bb < BITS; ++bb) {
    int32_t m;                           // bit set: all ones, else 0
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(d), "n"(bb));
    const uint64_t bal = __ballot(m != 0);
    diff_lo |= static_cast<uint32_t>(bal) ^ static_cast<uint32_t>(m);
    diff_hi |= static_cast<uint32_t>(bal >> 32) ^ static_cast<uint32_t>(m);
  }
  WaveMatch r;
  r.lo = static_cast<uint32_t>(v) & ~diff_lo;
  r.hi = static_cast<uint32_t>(v >> 32) & ~diff_hi;
  r.below = __builtin_amdgcn_mbcnt_hi(r.hi, __builtin_amdgcn_mbcnt_lo(r.lo, 0u));
  return r;
}

__device__ __forceinline__ WaveMatch wave_match_n(uint32_t d, bool valid, int bits) {
  switch (bits) {   // uniform: a pass has one digit width
    case 1: return wave_match<1>(d, valid);
    case 2: return wave_match<2>(d, valid);
    case 3: return wave_match<3>(d, valid);
    case 4: return wave_match<4>(d, valid);
    case 5: return wave_match<5>(d, valid);
    case 6: return wave_match<6>(d, valid);
    case 7: return wave_match<7>(d, valid);
    default: return wave_match<8>(d, valid);
  }
}

// ---- pass kernel 1: per-tile digit histogram ------------------------------------
template <typename KeyT>
__global__ __launch_bounds__(kThreads) void radix_hist_kernel(const KeyT* __restrict__ keys,
                                                              int64_t E, uint32_t N, int shift,
                                                              int bits, uint32_t* __restrict__ hist,
                                                              int n_tiles, int32_t* status) {
  __shared__ uint32_t s_hist[kMaxDigits];
  const int ndig = 1 << bits;
  const uint32_t mask = ndig - 1;
  for (int d = threadIdx.x; d < ndig; d += kThreads) s_hist[d] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile;
  // all loads of the thread first (positions past the end read the last key, which is then
  // not counted), so that they are in flight together
  KeyT raw[kKeysPerThread];
#pragma unroll
  for (int j = 0; j < kKeysPerThread; ++j) {
    const int64_t i = base + j * kThreads + threadIdx.x;
    raw[j] = keys[i < E ? i : E - 1];
  }
#pragma unroll
  for (int j = 0; j < kKeysPerThread; ++j) {
    const int64_t i = base + j * kThreads + threadIdx.x;
    if (i < E) {
      const uint32_t k = checked_key(raw[j], N, status);
      atomicAdd(&s_hist[(k >> shift) & mask], 1u);
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < ndig; d += kThreads)
    hist[static_cast<size_t>(d) * n_tiles + blockIdx.x] = s_hist[d];
}

// The first pass of a snapshot build reads both id arrays anyway (the scatter carries the
// other endpoint along), so the pass that groups the OTHER endpoints into buckets of 4096
// nodes for their degree count rides on it: this kernel makes both tile histograms in one
// read of the two int64 arrays.
__global__ __launch_bounds__(kThreads) void radix_hist_dual_kernel(
    const int64_t* __restrict__ keys, const int64_t* __restrict__ other, int64_t E, uint32_t N,
    uint32_t n_other, int shift, int bits, int shift2, int bits2, uint32_t* __restrict__ hist,
    uint32_t* __restrict__ hist2, int n_tiles, int32_t* status) {
  __shared__ uint32_t s_hist[kMaxDigits];
  __shared__ uint32_t s_hist2[kMaxDigits];
  const int ndig = 1 << bits, ndig2 = 1 << bits2;
  const uint32_t mask = ndig - 1, mask2 = ndig2 - 1;
  for (int d = threadIdx.x; d < kMaxDigits; d += kThreads) s_hist[d] = s_hist2[d] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile;
  int64_t raw[kKeysPerThread], raw2[kKeysPerThread];
#pragma unroll
  for (int j = 0; j < kKeysPerThread; ++j) {
    const int64_t i = base + j * kThreads + threadIdx.x;
    raw[j] = keys[i < E ? i : E - 1];
    raw2[j] = other[i < E ? i : E - 1];
  }
#pragma unroll
  for (int j = 0; j < kKeysPerThread; ++j) {
    const int64_t i = base + j * kThreads + threadIdx.x;
    if (i < E) {
      atomicAdd(&s_hist[(checked_key(raw[j], N, status) >> shift) & mask], 1u);
      atomicAdd(&s_hist2[(checked_key(raw2[j], n_other, status) >> shift2) & mask2], 1u);
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < ndig; d += kThreads)
    hist[static_cast<size_t>(d) * n_tiles + blockIdx.x] = s_hist[d];
  for (int d = threadIdx.x; d < ndig2; d += kThreads)
    hist2[static_cast<size_t>(d) * n_tiles + blockIdx.x] = s_hist2[d];
}

// ---- pass kernel 2: one workgroup per digit scans that digit's tile counts ------
// A wave owns a contiguous quarter of the row and walks it in strips of 64 consecutive
// counts, one per lane (a load or store instruction covers 256 contiguous bytes), eight
// strips in flight at a time; the running sum travels in a register.  With a million tiles'
// worth of edges a thread-contiguous split of the row (every lane of a load in a cache line of
// its own) took 0.16 ms per pass.
// Workgroups from `split` on scan the rows of a second histogram (the dual first pass).
__global__ __launch_bounds__(kThreads) void radix_scan_kernel(uint32_t* __restrict__ hist,
                                                              int n_tiles,
                                                              uint32_t* __restrict__ totals,
                                                              uint32_t* __restrict__ hist2 = nullptr,
                                                              uint32_t* __restrict__ totals2 = nullptr,
                                                              int split = 1 << 30) {
  __shared__ uint32_t s_wave[kWaves];
  constexpr int kStrips = 8;
  const bool second = static_cast<int>(blockIdx.x) >= split;
  const int digit = second ? static_cast<int>(blockIdx.x) - split : static_cast<int>(blockIdx.x);
  uint32_t* row = (second ? hist2 : hist) + static_cast<size_t>(digit) * n_tiles;
  const int lane = lane_id();
  const int wave = wave_id();
  const int span = ((n_tiles + kWaves - 1) / kWaves + 63) / 64 * 64;
  const int begin = min(wave * span, n_tiles);
  const int end = min(begin + span, n_tiles);

  uint32_t sum = 0;
  for (int i0 = begin; i0 < end; i0 += 64 * kStrips) {
    uint32_t c[kStrips];
#pragma unroll
    for (int j = 0; j < kStrips; ++j) {
      const int i = i0 + 64 * j + lane;
      c[j] = i < end ? row[i] : 0u;
    }
#pragma unroll
    for (int j = 0; j < kStrips; ++j) sum += c[j];
  }
  sum = wave_inclusive_scan(sum);  // lane 63: the wave's total
  if (lane == 63) s_wave[wave] = sum;
  __syncthreads();
  uint32_t run = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    const uint32_t c = s_wave[w];
    if (w < wave) run += c;
    total += c;
  }

  for (int i0 = begin; i0 < end; i0 += 64 * kStrips) {
    uint32_t c[kStrips];
#pragma unroll
    for (int j = 0; j < kStrips; ++j) {
      const int i = i0 + 64 * j + lane;
      c[j] = i < end ? row[i] : 0u;
    }
#pragma unroll
    for (int j = 0; j < kStrips; ++j) {
      const int i = i0 + 64 * j + lane;
      const uint32_t incl = wave_inclusive_scan(c[j]);
      if (i < end) row[i] = run + incl - c[j];
      run += __shfl(incl, 63);
    }
  }
  if (threadIdx.x == 0) (second ? totals2 : totals)[digit] = total;
}

// ---- pass kernel 3: stable scatter ------------------------------------------------
// NVAL 32-bit values travel with each key: value 0 is the other endpoint of the edge,
// value 1 the original edge position.  FIRST: keys and value 0 come from the caller's
// int64 arrays (range-checked) and value 1 is the position itself.
// What the dual first pass adds to the scatter: where the other endpoints go, grouped (in any
// order) by their bucket of 2^shift2 nodes.
struct OtherBuckets {
  uint32_t* keys_out;
  const uint32_t* hist;
  const uint32_t* totals;
  int shift, bits;
};

template <typename KeyT, bool FIRST, int NVAL, bool DUAL = false>
__global__ __launch_bounds__(kThreads) void radix_scatter_kernel(
    const KeyT* __restrict__ keys_in, const KeyT* __restrict__ other_in,
    const uint32_t* __restrict__ v0_in, const uint32_t* __restrict__ v1_in,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ v0_out, uint32_t* __restrict__ v1_out,
    int64_t E, uint32_t N, uint32_t n_other, int shift, int bits,
    const uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals, int n_tiles,
    int32_t* status, OtherBuckets ob = OtherBuckets{}) {
  __shared__ uint32_t s_cnt[kWaves][kMaxDigits];
  __shared__ uint32_t s_dstart[kMaxDigits];
  __shared__ uint32_t s_goff[kMaxDigits];
  __shared__ uint32_t s_wave[kWaves];
  __shared__ uint32_t s_keys[kTile];
  __shared__ uint32_t s_v0[NVAL >= 1 ? kTile : 1];
  __shared__ uint32_t s_v1[NVAL >= 2 ? kTile : 1];

  const int ndig = 1 << bits;
  const uint32_t mask = ndig - 1;
  const int lane = lane_id();
  const int wave = wave_id();
  const int tid = threadIdx.x;

  for (int d = tid; d < kWaves * kMaxDigits; d += kThreads) (&s_cnt[0][0])[d] = 0;
  __syncthreads();

  const int64_t tile_base = static_cast<int64_t>(blockIdx.x) * kTile;
  const int64_t wave_base = tile_base + static_cast<int64_t>(wave) * 64 * kKeysPerThread;
  const int64_t tile_left = E - tile_base;
  const int tile_count = tile_left < kTile ? static_cast<int>(tile_left) : kTile;

  uint32_t key[kKeysPerThread];
  uint32_t val0[kKeysPerThread];
  uint32_t val1[kKeysPerThread];
  uint32_t rank[kKeysPerThread];

#pragma unroll
  for (int j = 0; j < kKeysPerThread; ++j) {
    const int64_t i = wave_base + j * 64 + lane;
    const bool valid = i < E;
    uint32_t k = 0, a = 0, b = 0;
    if (valid) {
      k = checked_key(keys_in[i], N, status);
      if constexpr (NVAL >= 1) a = FIRST ? checked_key(other_in[i], n_other, status) : v0_in[i];
      if constexpr (NVAL >= 2) b = FIRST ? static_cast<uint32_t>(i) : v1_in[i];
    }
    key[j] = k;
    val0[j] = a;
    val1[j] = b;
    const uint32_t d = (k >> shift) & mask;
    // lanes of this wave row that hold the same digit
    const WaveMatch same = wave_match_n(d, valid, bits);
    uint32_t prev = 0;
    if (valid) prev = s_cnt[wave][d];
    rank[j] = prev + same.below;
    if (valid && same.below == 0) s_cnt[wave][d] = prev + same.size();
  }
  __syncthreads();

  // per digit: exclusive prefix over waves, tile total, tile-local digit start
  uint32_t tile_total_d = 0;
  if (tid < ndig) {
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      uint32_t c = s_cnt[w][tid];
      s_cnt[w][tid] = run;
      run += c;
    }
    tile_total_d = run;
  }
  uint32_t unused;
  const uint32_t dstart = block_exclusive_scan(tile_total_d, s_wave, &unused);
  const uint32_t gtot = (tid < ndig) ? totals[tid] : 0u;
  const uint32_t gbase = block_exclusive_scan(gtot, s_wave, &unused);
  if (tid < ndig) {
    s_dstart[tid] = dstart;
    s_goff[tid] = gbase + hist[static_cast<size_t>(tid) * n_tiles + blockIdx.x];
  }
  __syncthreads();

#pragma unroll
  for (int j = 0; j < kKeysPerThread; ++j) {
    const int64_t i = wave_base + j * 64 + lane;
    if (i < E) {
      const uint32_t d = (key[j] >> shift) & mask;
      const uint32_t pos = s_dstart[d] + s_cnt[wave][d] + rank[j];
      s_keys[pos] = key[j];
      if constexpr (NVAL >= 1) s_v0[pos] = val0[j];
      if constexpr (NVAL >= 2) s_v1[pos] = val1[j];
    }
  }
  __syncthreads();

#pragma unroll 4
  for (int j = 0; j < kKeysPerThread; ++j) {
    const int i = j * kThreads + tid;
    if (i < tile_count) {
      const uint32_t k = s_keys[i];
      const uint32_t d = (k >> shift) & mask;
      const uint32_t g = s_goff[d] + (static_cast<uint32_t>(i) - s_dstart[d]);
      keys_out[g] = k;
      if constexpr (NVAL >= 1) v0_out[g] = s_v0[i];
      if constexpr (NVAL >= 2) v1_out[g] = s_v1[i];
    }
  }

  if constexpr (DUAL && NVAL >= 1) {
    // the other endpoints of this tile (range-checked above, still in registers), grouped by
    // bucket without order: count, scan, next free slot per digit, staged in LDS, run by run out
    __shared__ uint32_t s_cursor2[kMaxDigits];
    __shared__ uint32_t s_dstart2[kMaxDigits];
    __shared__ uint32_t s_goff2[kMaxDigits];
    const int ndig2 = 1 << ob.bits;
    const uint32_t mask2 = ndig2 - 1;
    if (tid < ndig2) s_cursor2[tid] = 0;
    __syncthreads();   // also: every thread is done reading s_keys
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j)
      if (wave_base + j * 64 + lane < E) atomicAdd(&s_cursor2[(val0[j] >> ob.shift) & mask2], 1u);
    __syncthreads();
    const uint32_t dstart2 = block_exclusive_scan(tid < ndig2 ? s_cursor2[tid] : 0u, s_wave, &unused);
    const uint32_t gbase2 = block_exclusive_scan(tid < ndig2 ? ob.totals[tid] : 0u, s_wave, &unused);
    if (tid < ndig2) {
      s_dstart2[tid] = dstart2;
      s_cursor2[tid] = dstart2;
      s_goff2[tid] = gbase2 + ob.hist[static_cast<size_t>(tid) * n_tiles + blockIdx.x];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j)
      if (wave_base + j * 64 + lane < E)
        s_keys[atomicAdd(&s_cursor2[(val0[j] >> ob.shift) & mask2], 1u)] = val0[j];
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < kKeysPerThread; ++j) {
      const int i = j * kThreads + tid;
      if (i < tile_count) {
        const uint32_t k = s_keys[i];
        const uint32_t d = (k >> ob.shift) & mask2;
        ob.keys_out[s_goff2[d] + (static_cast<uint32_t>(i) - s_dstart2[d])] = k;
      }
    }
  }
}

// ---- grouping without order -------------------------------------------------------------
// Counting the ids of a bucket needs them side by side, in any order.  The same tile
// histogram and scan as a sort pass give every (tile, digit) its run in the output; inside the
// run a key takes the next free slot (an integer LDS counter per digit; the tile is staged in
// LDS in digit order and written out run by run), so no ranking is needed, and ranking is what
// a stable pass spends its time on.  Which key lands in which slot of its run varies from
// launch to launch; the counts made from the result do not.
template <typename KeyT>
__global__ __launch_bounds__(kThreads) void bucket_scatter_kernel(
    const KeyT* __restrict__ keys_in, uint32_t* __restrict__ keys_out, int64_t E, uint32_t N, int shift,
    int bits, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals, int n_tiles,
    int32_t* status) {
  __shared__ uint32_t s_cursor[kMaxDigits];
  __shared__ uint32_t s_dstart[kMaxDigits];
  __shared__ uint32_t s_goff[kMaxDigits];
  __shared__ uint32_t s_wave[kWaves];
  __shared__ uint32_t s_keys[kTile];
  const int ndig = 1 << bits;
  const uint32_t mask = ndig - 1;
  const int tid = threadIdx.x;
  if (tid < ndig) s_cursor[tid] = 0;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile;
  const int64_t left = E - base;
  const int tile_count = left < kTile ? static_cast<int>(left) : kTile;
  KeyT raw[kKeysPerThread];
#pragma unroll
  for (int j = 0; j < kKeysPerThread; ++j) {
    const int64_t i = base + j * kThreads + tid;
    raw[j] = keys_in[i < E ? i : E - 1];
  }
  __syncthreads();
  uint32_t key[kKeysPerThread];
#pragma unroll
  for (int j = 0; j < kKeysPerThread; ++j) {
    key[j] = 0;
    if (j * kThreads + tid < tile_count) {
      key[j] = checked_key(raw[j], N, status);
      atomicAdd(&s_cursor[(key[j] >> shift) & mask], 1u);
    }
  }
  __syncthreads();
  uint32_t unused;
  const uint32_t dstart = block_exclusive_scan(tid < ndig ? s_cursor[tid] : 0u, s_wave, &unused);
  const uint32_t gbase = block_exclusive_scan(tid < ndig ? totals[tid] : 0u, s_wave, &unused);
  if (tid < ndig) {
    s_dstart[tid] = dstart;
    s_cursor[tid] = dstart;
    s_goff[tid] = gbase + hist[static_cast<size_t>(tid) * n_tiles + blockIdx.x];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kKeysPerThread; ++j)
    if (j * kThreads + tid < tile_count) s_keys[atomicAdd(&s_cursor[(key[j] >> shift) & mask], 1u)] = key[j];
  __syncthreads();
#pragma unroll 4
  for (int j = 0; j < kKeysPerThread; ++j) {
    const int i = j * kThreads + tid;
    if (i < tile_count) {
      const uint32_t k = s_keys[i];
      const uint32_t d = (k >> shift) & mask;
      keys_out[s_goff[d] + (static_cast<uint32_t>(i) - s_dstart[d])] = k;
    }
  }
}

// ---- row pointers from sorted keys (+ gather of the other endpoint) ---------------
// Position i starts every row in (keys[i-1], keys[i]].  Short gaps are written by
// the lane itself; long gaps (many consecutive empty rows) are filled by the
// whole wave so that one lane never writes a long run alone.
__device__ __forceinline__ void wave_fill_gaps(int32_t* indptr, int64_t start, int64_t count,
                                               int32_t value) {
  const int lane = lane_id();
  uint64_t m = __ballot(count > 0);
  while (m) {
    const int src_lane = __ffsll(static_cast<unsigned long long>(m)) - 1;
    m &= m - 1;
    const int64_t s = __shfl(start, src_lane);
    const int64_t n = __shfl(count, src_lane);
    const int32_t v = __shfl(value, src_lane);
    for (int64_t r = lane; r < n; r += 64) indptr[s + r] = v;
  }
}

// `shift`: boundaries of (key >> shift), e.g. of buckets of 2^shift rows.
__global__ __launch_bounds__(kThreads) void csr_finalize_kernel(const uint32_t* __restrict__ keys,
                                                                int64_t E, int64_t N,
                                                                int32_t* __restrict__ indptr, int shift = 0) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  const bool valid = i < E;
  int64_t k = 0, prev = -1;
  if (valid) {
    k = keys[i] >> shift;
    if (i > 0) prev = keys[i - 1] >> shift;
  }
  int64_t gap_start = prev + 1;
  int64_t gap = valid ? (k - prev) : 0;
  if (gap > 0 && gap <= 4) {
    for (int64_t r = 0; r < gap; ++r) indptr[gap_start + r] = static_cast<int32_t>(i);
    gap = 0;
  }
  wave_fill_gaps(indptr, gap_start, gap, static_cast<int32_t>(i));
  // rows after the last key (and indptr[N]) point at E
  const bool last = valid && (i == E - 1);
  wave_fill_gaps(indptr, k + 1, last ? (N - k) : 0, static_cast<int32_t>(E));
}

__global__ void indptr_to_degree_kernel(const int32_t* __restrict__ indptr, int64_t N,
                                        int64_t* __restrict__ deg) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < N) deg[i] = static_cast<int64_t>(indptr[i + 1]) - static_cast<int64_t>(indptr[i]);
}

__global__ void norm_kernel(const int32_t* __restrict__ indptr, int64_t N, int mode,
                            float* __restrict__ norm) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < N) {
    float d = static_cast<float>(indptr[i + 1] - indptr[i]);
    d = fmaxf(d, 1.0f);
    norm[i] = (mode == SPMI_NORM_RSQRT) ? (1.0f / sqrtf(d)) : (1.0f / d);
  }
}

// ---- degrees by one bucket pass + LDS counters ---------------------------------------
// Counting needs no order inside a bucket: after ONE scatter pass on the top bits the keys of
// a 4096-node bucket are contiguous, and a workgroup counts their low 12 bits in LDS
// (integer LDS atomics: exact, order-independent) and writes that bucket's degrees with
// plain stores.  Two passes over the keys instead of the three full passes of the sort.
constexpr int kCountLow = 12;
constexpr int kCountThreads = 1024;

__global__ __launch_bounds__(kCountThreads) void bucket_count_kernel(
    const uint32_t* __restrict__ keys, const uint32_t* __restrict__ totals,
    const int32_t* __restrict__ bstart, int64_t N, int64_t* __restrict__ deg, float* __restrict__ norm,
    int mode) {
  __shared__ uint32_t s_hist[1 << kCountLow];
  __shared__ uint32_t s_start;
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  for (int j = tid; j < (1 << kCountLow); j += kCountThreads) s_hist[j] = 0;
  if (tid == 0) s_start = 0;
  __syncthreads();
  // where the bucket's keys are: from the bucket starts when there are more than 256 buckets,
  // else from the digit totals of the one pass that grouped them
  if (!bstart && tid < b) atomicAdd(&s_start, totals[tid]);  // b <= 255 < blockDim
  __syncthreads();
  const uint32_t start = bstart ? static_cast<uint32_t>(bstart[b]) : s_start;
  const uint32_t count = bstart ? static_cast<uint32_t>(bstart[b + 1] - bstart[b]) : totals[b];
  for (uint32_t i = tid; i < count; i += kCountThreads)
    atomicAdd(&s_hist[keys[start + i] & ((1u << kCountLow) - 1u)], 1u);
  __syncthreads();
  const int64_t node0 = static_cast<int64_t>(b) << kCountLow;
  for (int j = tid; j < (1 << kCountLow); j += kCountThreads) {
    if (node0 + j >= N) continue;
    if (deg) deg[node0 + j] = s_hist[j];
    if (norm) {  // the normaliser of this degree, in the same pass
      const float d = fmaxf(static_cast<float>(s_hist[j]), 1.0f);
      norm[node0 + j] = (mode == SPMI_NORM_RSQRT) ? (1.0f / sqrtf(d)) : (1.0f / d);
    }
  }
}

__global__ void norm_from_degree_kernel(const int64_t* __restrict__ deg, int64_t N, int mode,
                                        float* __restrict__ norm) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < N) {
    float d = fmaxf(static_cast<float>(deg[i]), 1.0f);
    norm[i] = (mode == SPMI_NORM_RSQRT) ? (1.0f / sqrtf(d)) : (1.0f / d);
  }
}

// ---- aggregation plan ---------------------------------------------------------------
constexpr int kPlanThreads = 1024;
constexpr int kPlanWaves = kPlanThreads / 64;
constexpr int kPlanRows = 16;  // rows per thread of plan_collect_kernel

// Exclusive scan of one value per thread over a 1024-thread workgroup (s_wave: kPlanWaves
// entries; *total receives the workgroup sum).
__device__ __forceinline__ uint32_t plan_block_scan(uint32_t v, uint32_t* s_wave, uint32_t* total) {
  const uint32_t incl = wave_inclusive_scan(v);
  if (lane_id() == 63) s_wave[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t base = 0, sum = 0;
#pragma unroll
  for (int w = 0; w < kPlanWaves; ++w) {
    const uint32_t c = s_wave[w];
    if (w < static_cast<int>(threadIdx.x >> 6)) base += c;
    sum += c;
  }
  *total = sum;
  __syncthreads();
  return base + incl - v;
}

// Lists the rows with more than heavy_thresh nonzeros.  A workgroup looks at 16384 rows,
// compacts its hub rows among its threads and appends them with ONE atomic: with a threshold
// of 128 a power-law snapshot of 1M rows has thousands of them, spread evenly, and an append
// per row (or per wave, which is the same thing when they are 200 rows apart) queued 44 us
// of atomics on one counter.  The list is in arrival order of the workgroups; results of the
// aggregation do not depend on that order (each row is reduced on its own).
__global__ __launch_bounds__(kPlanThreads) void plan_collect_kernel(
    const int32_t* __restrict__ indptr, int64_t N, int32_t heavy_thresh,
    int32_t* __restrict__ heavy_rows, int64_t capacity, int32_t* counts) {
  __shared__ uint32_t s_wave[kPlanWaves];
  __shared__ int32_t s_base;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * (kPlanThreads * kPlanRows);
  uint32_t flags = 0;
#pragma unroll
  for (int i = 0; i < kPlanRows; ++i) {
    const int64_t r = row0 + i * kPlanThreads + threadIdx.x;
    if (r < N && indptr[r + 1] - indptr[r] > heavy_thresh) flags |= 1u << i;
  }
  uint32_t total;
  const uint32_t mine = static_cast<uint32_t>(__popc(flags));
  const uint32_t ex = plan_block_scan(mine, s_wave, &total);
  if (total == 0) return;
  if (threadIdx.x == 0) s_base = atomicAdd(&counts[0], static_cast<int32_t>(total));
  __syncthreads();
  int64_t pos = static_cast<int64_t>(s_base) + ex;
#pragma unroll
  for (int i = 0; i < kPlanRows; ++i) {
    if (flags & (1u << i)) {
      if (pos < capacity) heavy_rows[pos] = static_cast<int32_t>(row0 + i * kPlanThreads + threadIdx.x);
      ++pos;
    }
  }
}

// Turns the collected list into chunk offsets: heavy_chunk_ptr = exclusive scan of
// ceil(deg/chunk) in list order.
__global__ __launch_bounds__(kPlanThreads) void plan_finish_kernel(
    const int32_t* __restrict__ indptr, int32_t chunk, const int32_t* __restrict__ heavy_rows,
    int32_t* __restrict__ heavy_chunk_ptr, int64_t capacity, int32_t* counts) {
  __shared__ uint32_t s_wave[kPlanWaves];
  __shared__ uint32_t s_carry;
  const int64_t collected = counts[0];
  const int n = static_cast<int>(collected < capacity ? collected : capacity);
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += kPlanThreads) {
    const int i = base + threadIdx.x;
    uint32_t c = 0;
    if (i < n) {
      const int32_t r = heavy_rows[i];
      const int32_t deg = indptr[r + 1] - indptr[r];
      c = static_cast<uint32_t>((deg + chunk - 1) / chunk);
    }
    uint32_t total;
    const uint32_t ex = plan_block_scan(c, s_wave, &total);
    const uint32_t carry = s_carry;
    if (i < n) heavy_chunk_ptr[i] = static_cast<int32_t>(carry + ex);
    __syncthreads();
    if (threadIdx.x == 0) s_carry = carry + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    heavy_chunk_ptr[n] = static_cast<int32_t>(s_carry);
    counts[0] = n;
    counts[1] = static_cast<int32_t>(s_carry);
  }
}

// ---- bucket stage: stable grouping inside buckets of 4096 rows -----------------------------
// After ONE stable scatter pass on the top bits of the row id the edges of a bucket of 4096
// consecutive rows are contiguous and still in edge order.  What is left is a stable counting
// sort on 12 bits inside every bucket, and that needs no further pass over the whole array:
// a bucket is cut into segments of 8192 edges, one workgroup (four waves) each;
//   seg_count_kernel   counts the rows of its segment in LDS and writes the 4096 counts
//                      (16 bits each) as one row of a table;
//   seg_place_kernel   adds up the table rows of its bucket (row totals -> row pointers, and
//                      the counts of the segments in front of its own), ranks its 8192 edges
//                      with wave ballots (stable) and writes every edge straight to its final
//                      place.  The writes of a bucket fall into a window of a few hundred KB,
//                      which the L2 holds; consecutive segments run on the same XCD.
// The first segment of a bucket also writes the bucket's row pointers, its normaliser and its
// rows above the hub threshold, so run boundaries, the normaliser pass and the hub-row scan of
// the three-pass sort are gone with two of its three passes.
constexpr int kLow = 12;
constexpr int kRows = 1 << kLow;                  // rows per bucket
constexpr int kRowWords = kRows / 2;              // a table row: 4096 16-bit counts in 2048 words
constexpr int kSegThreads = 256;
constexpr int kSegWaves = kSegThreads / 64;
constexpr int kSegKeys = 32;                      // edges per thread
constexpr int kSeg = kSegThreads * kSegKeys;      // edges per segment
constexpr int kRowsPerThread = kRows / kSegThreads;   // 16 rows = 8 table words = 2 x 16 bytes
constexpr int kFastBuckets = 256;    // one pass on the top bits, segment layout worked out per workgroup
constexpr int kMaxBuckets = 4096;    // several passes on the top bits, segment layout by a kernel

struct SegLoc {
  int bucket, seg_in_bucket;
  uint32_t n_seg, seg_base, bstart, bcount;
};

__device__ __forceinline__ uint32_t seg_block_scan(uint32_t v, uint32_t* s_wave, uint32_t* total) {
  const uint32_t incl = wave_inclusive_scan(v);
  if (lane_id() == 63) s_wave[wave_id()] = incl;
  __syncthreads();
  uint32_t base = 0, sum = 0;
#pragma unroll
  for (int w = 0; w < kSegWaves; ++w) {
    const uint32_t c = s_wave[w];
    if (w < wave_id()) base += c;
    sum += c;
  }
  *total = sum;
  __syncthreads();
  return base + incl - v;
}

// Where the buckets and their segments are.  Up to 256 buckets (one pass on the top bits) every
// workgroup works it out from that pass's digit totals; beyond that (several passes on the top
// bits) the bucket starts come from the sorted keys and a small kernel lays the segments out.
struct SegMap {
  const uint32_t* totals;     // <= 256 buckets: edges per bucket
  const int32_t* bstart;      // otherwise: first edge of every bucket, [n_buckets + 1]
  const uint32_t* seg_base;   //            first segment of every bucket, [n_buckets + 1]
  const uint2* desc;          //            (bucket, segment inside it) of every segment
  int n_buckets, grid;
};

// Which segment of which bucket this workgroup owns.  Every bucket has at least one segment
// (an empty bucket still has row pointers to write).  Workgroup numbers are dealt so that
// consecutive segments run on one XCD (workgroups go round the 8 XCDs).  False: no such segment.
__device__ __forceinline__ bool locate_segment(const SegMap& m, SegLoc* s_loc, uint32_t* s_wave) {
  const int seg = static_cast<int>(blockIdx.x & 7) * (m.grid >> 3) + static_cast<int>(blockIdx.x >> 3);
  const int tid = threadIdx.x;
  if (m.desc) {   // uniform
    if (static_cast<uint32_t>(seg) >= m.seg_base[m.n_buckets]) return false;
    if (tid == 0) {
      const uint2 d = m.desc[seg];
      s_loc->bucket = static_cast<int>(d.x);
      s_loc->seg_in_bucket = static_cast<int>(d.y);
      s_loc->seg_base = m.seg_base[d.x];
      s_loc->n_seg = m.seg_base[d.x + 1] - m.seg_base[d.x];
      s_loc->bstart = static_cast<uint32_t>(m.bstart[d.x]);
      s_loc->bcount = static_cast<uint32_t>(m.bstart[d.x + 1] - m.bstart[d.x]);
    }
    __syncthreads();
    return true;
  }
  const uint32_t cnt = tid < m.n_buckets ? m.totals[tid] : 0u;
  const uint32_t segs = tid < m.n_buckets ? max(1u, (cnt + kSeg - 1) / kSeg) : 0u;
  uint32_t total_cnt, total_segs;
  const uint32_t bstart = seg_block_scan(cnt, s_wave, &total_cnt);
  const uint32_t sbase = seg_block_scan(segs, s_wave, &total_segs);
  if (tid < m.n_buckets && static_cast<uint32_t>(seg) >= sbase && static_cast<uint32_t>(seg) < sbase + segs) {
    s_loc->bucket = tid;
    s_loc->seg_in_bucket = seg - static_cast<int>(sbase);
    s_loc->n_seg = segs;
    s_loc->seg_base = sbase;
    s_loc->bstart = bstart;
    s_loc->bcount = cnt;
  }
  __syncthreads();
  return static_cast<uint32_t>(seg) < total_segs;
}

// More than 256 buckets: first segment of every bucket (a scan of ceil(count / 8192), at least
// one each) and the (bucket, segment) pair of every segment.  One workgroup; buckets are few.
__global__ __launch_bounds__(1024) void seg_layout_kernel(const int32_t* __restrict__ bstart, int n_buckets,
                                                          uint32_t* __restrict__ seg_base,
                                                          uint2* __restrict__ desc) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < n_buckets; b0 += 1024) {
    const int b = b0 + threadIdx.x;
    uint32_t segs = 0;
    if (b < n_buckets) {
      const uint32_t cnt = static_cast<uint32_t>(bstart[b + 1] - bstart[b]);
      segs = max(1u, (cnt + kSeg - 1) / kSeg);
    }
    const uint32_t incl = wave_inclusive_scan(segs);
    if (lane_id() == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const uint32_t c = s_wave[w];
      if (w < static_cast<int>(threadIdx.x >> 6)) base += c;
      sum += c;
    }
    const uint32_t first = s_carry + base + incl - segs;
    if (b < n_buckets) {
      seg_base[b] = first;
      for (uint32_t j = 0; j < segs; ++j) desc[first + j] = make_uint2(static_cast<uint32_t>(b), j);
    }
    __syncthreads();
    if (threadIdx.x == 0) s_carry += sum;
    __syncthreads();
  }
  if (threadIdx.x == 0) seg_base[n_buckets] = s_carry;
}

__global__ __launch_bounds__(kSegThreads) void seg_count_kernel(const uint32_t* __restrict__ keys,
                                                                SegMap map, uint32_t* __restrict__ table) {
  __shared__ uint32_t s_hist[kRowWords];
  __shared__ uint32_t s_wave[kSegWaves];
  __shared__ SegLoc s_loc;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < kRowWords / kSegThreads; ++i) s_hist[i * kSegThreads + tid] = 0;
  if (!locate_segment(map, &s_loc, s_wave)) return;
  const SegLoc L = s_loc;
  const uint32_t lo = L.bstart + static_cast<uint32_t>(L.seg_in_bucket) * kSeg;
  const uint32_t hi = min(lo + kSeg, L.bstart + L.bcount);
  uint32_t k[kSegKeys];
#pragma unroll
  for (int j = 0; j < kSegKeys; ++j) {
    const uint32_t i = lo + j * kSegThreads + tid;
    k[j] = i < hi ? keys[i] : 0xffffffffu;
  }
#pragma unroll
  for (int j = 0; j < kSegKeys; ++j) {
    if (k[j] != 0xffffffffu) {
      const uint32_t r = k[j] & (kRows - 1);
      atomicAdd(&s_hist[r >> 1], 1u << (16 * (r & 1)));   // at most 8192 per half: no carry
    }
  }
  __syncthreads();
  uint4* row = reinterpret_cast<uint4*>(table + static_cast<size_t>(L.seg_base + L.seg_in_bucket) * kRowWords);
#pragma unroll
  for (int i = 0; i < kRowWords / 4 / kSegThreads; ++i)
    row[i * kSegThreads + tid] = reinterpret_cast<const uint4*>(s_hist)[i * kSegThreads + tid];
}

template <int NVAL>
__global__ __launch_bounds__(kSegThreads) void seg_place_kernel(
    const uint32_t* __restrict__ keys, const uint32_t* __restrict__ v0_in, const uint32_t* __restrict__ v1_in,
    SegMap map, const uint32_t* __restrict__ table, int64_t E, int64_t N, int32_t* __restrict__ indptr, uint32_t* __restrict__ v0_out,
    uint32_t* __restrict__ v1_out, float* __restrict__ norm, int mode, int32_t heavy_thresh,
    int32_t* __restrict__ heavy_rows, int64_t capacity, int32_t* counts) {
  __shared__ __align__(16) uint16_t s_cnt[kSegWaves][kRows];
  __shared__ __align__(16) uint32_t s_off[kRows];
  __shared__ uint32_t s_wave[kSegWaves];
  __shared__ SegLoc s_loc;
  __shared__ int32_t s_base;
  const int tid = threadIdx.x;
  const int lane = lane_id();
  const int wave = wave_id();
  {
    uint4* z = reinterpret_cast<uint4*>(&s_cnt[0][0]);
#pragma unroll
    for (int i = 0; i < static_cast<int>(sizeof(s_cnt) / 16) / kSegThreads; ++i)
      z[i * kSegThreads + tid] = make_uint4(0, 0, 0, 0);
  }
  if (!locate_segment(map, &s_loc, s_wave)) return;
  const int n_buckets = map.n_buckets;
  const SegLoc L = s_loc;
  const uint32_t lo = L.bstart + static_cast<uint32_t>(L.seg_in_bucket) * kSeg;
  const uint32_t hi = min(lo + kSeg, L.bstart + L.bcount);

  // this thread's edges: a wave owns 2048 consecutive edges, 64 at a time (edge order)
  uint32_t key[kSegKeys];
#pragma unroll
  for (int g = 0; g < kSegKeys; ++g) {
    const uint32_t i = lo + wave * (64 * kSegKeys) + g * 64 + lane;
    key[g] = i < hi ? keys[i] : 0u;
  }

  // rows 16 tid .. 16 tid + 15 of the bucket: totals over all its segments, and over those in
  // front of this one (four table rows in flight at a time)
  constexpr int R = kRowsPerThread;
  uint32_t tot[R], bef[R];
#pragma unroll
  for (int i = 0; i < R; ++i) tot[i] = bef[i] = 0;
  const uint4* rows = reinterpret_cast<const uint4*>(table + static_cast<size_t>(L.seg_base) * kRowWords) + 2 * tid;
  for (uint32_t q0 = 0; q0 < L.n_seg; q0 += 4) {
    uint4 w[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t q = min(q0 + u, L.n_seg - 1);
      w[u][0] = rows[static_cast<size_t>(q) * (kRowWords / 4)];
      w[u][1] = rows[static_cast<size_t>(q) * (kRowWords / 4) + 1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (q0 + u < L.n_seg) {
        const uint32_t c[8] = {w[u][0].x, w[u][0].y, w[u][0].z, w[u][0].w, w[u][1].x, w[u][1].y, w[u][1].z, w[u][1].w};
        const bool front = q0 + u < static_cast<uint32_t>(L.seg_in_bucket);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const uint32_t c_lo = c[i] & 0xffffu, c_hi = c[i] >> 16;
          tot[2 * i] += c_lo;
          tot[2 * i + 1] += c_hi;
          if (front) {
            bef[2 * i] += c_lo;
            bef[2 * i + 1] += c_hi;
          }
        }
      }
    }
  }
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < R; ++i) sum += tot[i];
  uint32_t unused;
  uint32_t run = L.bstart + seg_block_scan(sum, s_wave, &unused);
  const bool first_seg = L.seg_in_bucket == 0;   // uniform over the workgroup
  const int64_t row0 = (static_cast<int64_t>(L.bucket) << kLow) + R * tid;
  uint32_t flags = 0;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    s_off[R * tid + i] = run + bef[i];
    if (first_seg) {
      // row pointers, normaliser and hub rows of this bucket
      const int64_t r = row0 + i;
      if (r < N) {
        indptr[r] = static_cast<int32_t>(run);
        if (norm) {
          const float d = fmaxf(static_cast<float>(tot[i]), 1.0f);
          norm[r] = (mode == SPMI_NORM_RSQRT) ? (1.0f / sqrtf(d)) : (1.0f / d);
        }
        if (heavy_rows && static_cast<int32_t>(tot[i]) > heavy_thresh) flags |= 1u << i;
      }
    }
    run += tot[i];
  }
  if (first_seg) {
    if (L.bucket == n_buckets - 1 && tid == 0) indptr[N] = static_cast<int32_t>(E);
    if (heavy_rows) {
      uint32_t total;
      const uint32_t ex = seg_block_scan(static_cast<uint32_t>(__popc(flags)), s_wave, &total);
      if (total) {
        if (tid == 0) s_base = atomicAdd(&counts[0], static_cast<int32_t>(total));
        __syncthreads();
        int64_t pos = static_cast<int64_t>(s_base) + ex;
#pragma unroll
        for (int i = 0; i < R; ++i) {
          if (flags & (1u << i)) {
            if (pos < capacity) heavy_rows[pos] = static_cast<int32_t>(row0 + i);
            ++pos;
          }
        }
      }
    }
  }

  // rank inside the wave's 2048 edges: same row, earlier position (ballots: stable, no atomics).
  // From here on key[g] holds (row inside the bucket) | (rank << 12).
#pragma unroll
  for (int g = 0; g < kSegKeys; ++g) {
    const uint32_t i = lo + wave * (64 * kSegKeys) + g * 64 + lane;
    const bool valid = i < hi;
    const uint32_t d = key[g] & (kRows - 1);
    const WaveMatch same = wave_match<kLow>(d, valid);
    uint32_t prev = 0;
    if (valid) prev = s_cnt[wave][d];
    key[g] = d | ((prev + same.below) << kLow);
    if (valid && same.below == 0) s_cnt[wave][d] = static_cast<uint16_t>(prev + same.size());
  }
  // the values that travel with the edges: fetched now, so that they arrive under the scan
  // below (holding them from the start costs a third of the kernel's registers)
  uint32_t val0[kSegKeys], val1[kSegKeys];
#pragma unroll
  for (int g = 0; g < kSegKeys; ++g) {
    const uint32_t i = lo + wave * (64 * kSegKeys) + g * 64 + lane;
    const bool valid = i < hi;
    if constexpr (NVAL >= 1) val0[g] = valid ? v0_in[i] : 0u;
    if constexpr (NVAL >= 2) val1[g] = valid ? v1_in[i] : 0u;
  }
  __syncthreads();
  // per row: counts of the waves -> offsets of the waves (a wave's edges follow those of the
  // waves below it)
  {
    uint32_t acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = 0;
#pragma unroll
    for (int w = 0; w < kSegWaves; ++w) {
      uint4* p = reinterpret_cast<uint4*>(&s_cnt[w][R * tid]);
#pragma unroll
      for (int h = 0; h < R / 8; ++h) {
        const uint4 c = p[h];
        const uint32_t v[4] = {c.x, c.y, c.z, c.w};
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o[i] = acc[8 * h + 2 * i] | (acc[8 * h + 2 * i + 1] << 16);
          acc[8 * h + 2 * i] += v[i] & 0xffffu;
          acc[8 * h + 2 * i + 1] += v[i] >> 16;
        }
        p[h] = make_uint4(o[0], o[1], o[2], o[3]);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int g = 0; g < kSegKeys; ++g) {
    const uint32_t i = lo + wave * (64 * kSegKeys) + g * 64 + lane;
    if (i < hi) {
      const uint32_t d = key[g] & (kRows - 1);
      const uint32_t pos = s_off[d] + s_cnt[wave][d] + (key[g] >> kLow);
      if constexpr (NVAL >= 1) v0_out[pos] = val0[g];
      if constexpr (NVAL >= 2) v1_out[pos] = val1[g];
    }
  }
}

// ---- host-side sort driver ------------------------------------------------------------
struct SortShape {
  int passes;
  int pass_bits[4];
  int n_tiles;
};

SortShape sort_shape(int64_t E, int64_t N) {
  SortShape s{};
  int bits = 1;
  while ((int64_t{1} << bits) < N) ++bits;
  s.passes = (bits + 7) / 8;
  const int base = bits / s.passes, rem = bits % s.passes;
  for (int p = 0; p < s.passes; ++p) s.pass_bits[p] = base + (p < rem ? 1 : 0);
  s.n_tiles = static_cast<int>((E + kTile - 1) / kTile);
  return s;
}

struct SortBuffers {
  uint32_t* keys[2];
  uint32_t* v0[2];  // v0[0] is the caller's indices output
  uint32_t* v1[2];  // v1[0] is the caller's eid output
  uint32_t* hist;
  uint32_t* totals;
};

// Sorts E keys (with NVAL values each); the result lands in keys[0] / v0[0] / v1[0].
template <int NVAL>
int run_sort(const int64_t* keys_in, const int64_t* other_in, int64_t E, int64_t N, int64_t N_other,
             const SortShape& sh, const SortBuffers& b, int32_t* status, hipStream_t stream) {
  int shift = 0;
  for (int p = 0; p < sh.passes; ++p) {
    const int bits = sh.pass_bits[p];
    const int out = (sh.passes - 1 - p) & 1;
    const int in = out ^ 1;
    const dim3 grid(sh.n_tiles), block(kThreads);
    if (p == 0) {
      radix_hist_kernel<int64_t><<<grid, block, 0, stream>>>(keys_in, E, (uint32_t)N, shift, bits,
                                                             b.hist, sh.n_tiles, status);
    } else {
      radix_hist_kernel<uint32_t><<<grid, block, 0, stream>>>(b.keys[in], E, (uint32_t)N, shift,
                                                              bits, b.hist, sh.n_tiles, status);
    }
    SPMI_CHECK_LAUNCH("radix_hist");
    radix_scan_kernel<<<dim3(1 << bits), block, 0, stream>>>(b.hist, sh.n_tiles, b.totals);
    SPMI_CHECK_LAUNCH("radix_scan");
    if (p == 0) {
      radix_scatter_kernel<int64_t, true, NVAL><<<grid, block, 0, stream>>>(
          keys_in, other_in, nullptr, nullptr, b.keys[out], b.v0[out], b.v1[out], E, (uint32_t)N,
          (uint32_t)N_other, shift, bits, b.hist, b.totals, sh.n_tiles, status);
    } else {
      radix_scatter_kernel<uint32_t, false, NVAL><<<grid, block, 0, stream>>>(
          b.keys[in], nullptr, b.v0[in], b.v1[in], b.keys[out], b.v0[out], b.v1[out], E,
          (uint32_t)N, (uint32_t)N_other, shift, bits, b.hist, b.totals, sh.n_tiles, status);
    }
    SPMI_CHECK_LAUNCH("radix_scatter");
    shift += bits;
  }
  return SPMI_OK;
}

int check_sizes(int64_t E, int64_t N, const char* who) {
  SPMI_REQUIRE(E >= 0 && N >= 1, SPMI_ERR_INVALID_ARG, "%s: need E >= 0 and N >= 1 (E=%lld N=%lld)",
               who, (long long)E, (long long)N);
  SPMI_REQUIRE(E < (int64_t{1} << 31) && N < (int64_t{1} << 31), SPMI_ERR_UNSUPPORTED,
               "%s: E=%lld or N=%lld exceeds the int32 on-device CSR limit", who, (long long)E,
               (long long)N);
  return SPMI_OK;
}

size_t sort_scratch_bytes(int64_t E, int64_t N) {
  const SortShape sh = sort_shape(E, N);
  return spmi::align_up(sizeof(uint32_t) * kMaxDigits * (size_t)sh.n_tiles, 256) +
         spmi::align_up(sizeof(uint32_t) * kMaxDigits, 256);
}

// Counts per node by grouping the ids into buckets of 4096 nodes and counting every bucket in
// LDS (N <= 2^24).  Up to 256 buckets: one grouping pass, in any order.  Up to 4096: two passes
// on the bucket number, the first in any order, the second stable (it must keep the first one's
// grouping), then the bucket starts from the sorted keys.
template <typename KeyT>
static int degree_by_buckets(const KeyT* ids, int64_t E, int64_t N, int64_t* deg, float* norm, int mode,
                             uint32_t* keys_a, uint32_t* keys_b, int32_t* bstart, uint32_t* hist,
                             uint32_t* totals, int32_t* status, hipStream_t st) {
  const int n_tiles = static_cast<int>((E + kTile - 1) / kTile);
  const int n_buckets = static_cast<int>((N + (1 << kCountLow) - 1) >> kCountLow);
  int bits = 1;
  while ((1 << bits) < n_buckets) ++bits;
  const dim3 grid(n_tiles), block(kThreads);
  if (bits <= 8) {
    radix_hist_kernel<KeyT><<<grid, block, 0, st>>>(ids, E, (uint32_t)N, kCountLow, bits, hist, n_tiles, status);
    SPMI_CHECK_LAUNCH("radix_hist(degree)");
    radix_scan_kernel<<<dim3(1 << bits), block, 0, st>>>(hist, n_tiles, totals);
    SPMI_CHECK_LAUNCH("radix_scan(degree)");
    bucket_scatter_kernel<KeyT><<<grid, block, 0, st>>>(ids, keys_a, E, (uint32_t)N, kCountLow, bits, hist,
                                                        totals, n_tiles, status);
    SPMI_CHECK_LAUNCH("bucket_scatter(degree)");
    bucket_count_kernel<<<n_buckets, kCountThreads, 0, st>>>(keys_a, totals, nullptr, N, deg, norm, mode);
    SPMI_CHECK_LAUNCH("bucket_count");
    return SPMI_OK;
  }
  const int lo_bits = bits / 2, hi_bits = bits - lo_bits;
  radix_hist_kernel<KeyT><<<grid, block, 0, st>>>(ids, E, (uint32_t)N, kCountLow, lo_bits, hist, n_tiles, status);
  SPMI_CHECK_LAUNCH("radix_hist(degree)");
  radix_scan_kernel<<<dim3(1 << lo_bits), block, 0, st>>>(hist, n_tiles, totals);
  SPMI_CHECK_LAUNCH("radix_scan(degree)");
  bucket_scatter_kernel<KeyT><<<grid, block, 0, st>>>(ids, keys_b, E, (uint32_t)N, kCountLow, lo_bits, hist,
                                                      totals, n_tiles, status);
  SPMI_CHECK_LAUNCH("bucket_scatter(degree)");
  radix_hist_kernel<uint32_t><<<grid, block, 0, st>>>(keys_b, E, (uint32_t)N, kCountLow + lo_bits, hi_bits, hist,
                                                      n_tiles, nullptr);
  SPMI_CHECK_LAUNCH("radix_hist(degree)");
  radix_scan_kernel<<<dim3(1 << hi_bits), block, 0, st>>>(hist, n_tiles, totals);
  SPMI_CHECK_LAUNCH("radix_scan(degree)");
  radix_scatter_kernel<uint32_t, false, 0><<<grid, block, 0, st>>>(
      keys_b, nullptr, nullptr, nullptr, keys_a, nullptr, nullptr, E, (uint32_t)N, 1u, kCountLow + lo_bits,
      hi_bits, hist, totals, n_tiles, nullptr);
  SPMI_CHECK_LAUNCH("radix_scatter(degree)");
  csr_finalize_kernel<<<static_cast<int>((E + kThreads - 1) / kThreads), kThreads, 0, st>>>(
      keys_a, E, n_buckets, bstart, kCountLow);
  SPMI_CHECK_LAUNCH("bucket_bounds(degree)");
  bucket_count_kernel<<<n_buckets, kCountThreads, 0, st>>>(keys_a, nullptr, bstart, N, deg, norm, mode);
  SPMI_CHECK_LAUNCH("bucket_count");
  return SPMI_OK;
}

constexpr int64_t kBucketDegreeMaxN = int64_t{1} << (kCountLow + 12);   // 4096 buckets of 4096 nodes

}  // namespace

extern "C" {

size_t spmi_coo_to_csr_workspace_bytes(int64_t E, int64_t N) {
  if (E < 0 || N < 1) return 0;
  const size_t e4 = spmi::align_up(sizeof(uint32_t) * (size_t)E, 256);
  return 4 * e4 + sort_scratch_bytes(E, N);
}

int spmi_coo_to_csr(const int64_t* key, const int64_t* other, int64_t E, int64_t N,
                    int64_t N_other, int32_t* indptr, int32_t* indices, int32_t* eid, void* ws,
                    size_t ws_bytes, int32_t* status, void* stream) {
  if (int rc = check_sizes(E, N, "spmi_coo_to_csr")) return rc;
  SPMI_REQUIRE(N_other >= 1 && N_other < (int64_t{1} << 31), SPMI_ERR_INVALID_ARG,
               "spmi_coo_to_csr: bad N_other=%lld", (long long)N_other);
  SPMI_REQUIRE(indptr && status, SPMI_ERR_INVALID_ARG, "spmi_coo_to_csr: null indptr/status");
  hipStream_t st = spmi::as_stream(stream);
  if (E == 0) {
    SPMI_CHECK_HIP(hipMemsetAsync(indptr, 0, sizeof(int32_t) * (size_t)(N + 1), st));
    return SPMI_OK;
  }
  SPMI_REQUIRE(key && other && indices && ws, SPMI_ERR_INVALID_ARG,
               "spmi_coo_to_csr: null pointer argument");
  SPMI_REQUIRE(ws_bytes >= spmi_coo_to_csr_workspace_bytes(E, N), SPMI_ERR_WORKSPACE,
               "spmi_coo_to_csr: workspace %zu < required %zu", ws_bytes,
               spmi_coo_to_csr_workspace_bytes(E, N));
  const SortShape sh = sort_shape(E, N);
  const size_t e4 = spmi::align_up(sizeof(uint32_t) * (size_t)E, 256);
  char* p = static_cast<char*>(ws);
  SortBuffers b{};
  b.keys[0] = reinterpret_cast<uint32_t*>(p);
  b.keys[1] = reinterpret_cast<uint32_t*>(p + e4);
  b.v0[0] = reinterpret_cast<uint32_t*>(indices);
  b.v0[1] = reinterpret_cast<uint32_t*>(p + 2 * e4);
  b.v1[0] = reinterpret_cast<uint32_t*>(eid);
  b.v1[1] = reinterpret_cast<uint32_t*>(p + 3 * e4);
  b.hist = reinterpret_cast<uint32_t*>(p + 4 * e4);
  b.totals = reinterpret_cast<uint32_t*>(
      p + 4 * e4 + spmi::align_up(sizeof(uint32_t) * kMaxDigits * (size_t)sh.n_tiles, 256));
  const int rc = eid ? run_sort<2>(key, other, E, N, N_other, sh, b, status, st)
                     : run_sort<1>(key, other, E, N, N_other, sh, b, status, st);
  if (rc) return rc;
  const int grid = static_cast<int>((E + kThreads - 1) / kThreads);
  csr_finalize_kernel<<<grid, kThreads, 0, st>>>(b.keys[0], E, N, indptr);
  SPMI_CHECK_LAUNCH("csr_finalize");
  return SPMI_OK;
}

// True when the bucket-stage build applies: more than one bucket of 4096 rows and at most 4096
// of them (N <= 2^24).  Up to 256 buckets the top bits take one radix pass, beyond that two.
static bool bucket_build_applies(int64_t E, int64_t N) {
  return E > 0 && N > kRows && N <= (static_cast<int64_t>(kMaxBuckets) << kLow);
}

static int bucket_build_grid(int64_t E, int n_buckets) {
  const int64_t segs = (E + kSeg - 1) / kSeg + n_buckets;
  return static_cast<int>((segs + 7) / 8 * 8);
}

static size_t bucket_layout_bytes(int64_t E, int n_buckets) {
  return spmi::align_up(sizeof(int32_t) * (size_t)(n_buckets + 1), 256) * 2 +
         spmi::align_up(sizeof(uint2) * (size_t)bucket_build_grid(E, n_buckets), 256);
}

size_t spmi_snapshot_csr_workspace_bytes(int64_t E, int64_t N) {
  if (E < 0 || N < 1) return 0;
  size_t bytes = spmi_coo_to_csr_workspace_bytes(E, N);
  if (bucket_build_applies(E, N)) {
    const int n_buckets = static_cast<int>((N + kRows - 1) >> kLow);
    const size_t e4 = spmi::align_up(sizeof(uint32_t) * (size_t)E, 256);
    const size_t table = sizeof(uint32_t) * kRowWords * (size_t)bucket_build_grid(E, n_buckets);
    // 5 e4: two key buffers, one buffer per value, the grouped other endpoints of the dual pass;
    // twice the sort scratch: the dual pass keeps a second tile histogram
    bytes = std::max(bytes, 5 * e4 + 2 * sort_scratch_bytes(E, N) + bucket_layout_bytes(E, n_buckets) + table);
  }
  return bytes;
}

int spmi_snapshot_csr_norm_other_supported(int64_t E, int64_t N, int64_t N_other) {
  // the other side's normaliser rides on the build when the build has ONE pass on the top bits
  // (N <= 2^20) and the other side has at most 256 buckets of 4096 nodes
  return bucket_build_applies(E, N) && N <= (static_cast<int64_t>(kFastBuckets) << kLow) && N_other >= 1 &&
         N_other <= (int64_t{1} << (kCountLow + 8));
}

int spmi_snapshot_csr(const int64_t* key, const int64_t* other, int64_t E, int64_t N, int64_t N_other,
                      int32_t* indptr, int32_t* indices, int32_t* eid, float* norm_key, float* norm_other,
                      int norm_mode, int32_t heavy_thresh, int32_t chunk, int32_t* heavy_rows, int32_t* heavy_chunk_ptr,
                      int64_t capacity, void* ws, size_t ws_bytes, int32_t* meta, void* stream) {
  if (int rc = check_sizes(E, N, "spmi_snapshot_csr")) return rc;
  SPMI_REQUIRE(meta && indptr, SPMI_ERR_INVALID_ARG, "spmi_snapshot_csr: null indptr/meta");
  SPMI_REQUIRE((!norm_key && !norm_other) || norm_mode == SPMI_NORM_RSQRT || norm_mode == SPMI_NORM_RECIP,
               SPMI_ERR_INVALID_ARG, "spmi_snapshot_csr: bad norm_mode %d", norm_mode);
  SPMI_REQUIRE(!norm_other || spmi_snapshot_csr_norm_other_supported(E, N, N_other), SPMI_ERR_UNSUPPORTED,
               "spmi_snapshot_csr: norm_other is not available for E=%lld N=%lld N_other=%lld "
               "(see spmi_snapshot_csr_norm_other_supported)", (long long)E, (long long)N, (long long)N_other);
  SPMI_REQUIRE(heavy_rows && heavy_chunk_ptr && heavy_thresh >= 1 && chunk >= 1 && capacity >= 2,
               SPMI_ERR_INVALID_ARG, "spmi_snapshot_csr: bad plan arguments");
  SPMI_REQUIRE(ws_bytes >= spmi_snapshot_csr_workspace_bytes(E, N), SPMI_ERR_WORKSPACE,
               "spmi_snapshot_csr: workspace %zu < required %zu", ws_bytes,
               spmi_snapshot_csr_workspace_bytes(E, N));
  hipStream_t st = spmi::as_stream(stream);
  if (!bucket_build_applies(E, N)) {
    // a single bucket, or more than 4096 of them: the radix sort on every bit and its separate kernels
    if (int rc = spmi_coo_to_csr(key, other, E, N, N_other, indptr, indices, eid, ws, ws_bytes, meta, stream))
      return rc;
    if (norm_key)
      if (int rc = spmi_norm_from_indptr(indptr, N, norm_mode, norm_key, stream)) return rc;
    return spmi_spmm_plan(indptr, N, heavy_thresh, chunk, heavy_rows, heavy_chunk_ptr, capacity, meta + 1,
                          stream);
  }
  SPMI_REQUIRE(N_other >= 1 && N_other < (int64_t{1} << 31), SPMI_ERR_INVALID_ARG,
               "spmi_snapshot_csr: bad N_other=%lld", (long long)N_other);
  SPMI_REQUIRE(key && other && indices && ws, SPMI_ERR_INVALID_ARG, "spmi_snapshot_csr: null pointer argument");
  int bits = 1;
  while ((int64_t{1} << bits) < N) ++bits;
  const int top = bits - kLow;                      // 1..12
  const int passes = top <= 8 ? 1 : 2;
  const int pass_bits[2] = {passes == 1 ? top : top / 2, top - top / 2};
  const int n_buckets = static_cast<int>((N + kRows - 1) >> kLow);
  const int n_tiles = static_cast<int>((E + kTile - 1) / kTile);
  const size_t e4 = spmi::align_up(sizeof(uint32_t) * (size_t)E, 256);
  char* p = static_cast<char*>(ws);
  // the last pass must leave its result in workspace buffers (the bucket stage writes `indices`
  // and `eid`); an earlier pass may use those two arrays as its output
  uint32_t* keys_ab[2] = {reinterpret_cast<uint32_t*>(p), reinterpret_cast<uint32_t*>(p + e4)};
  uint32_t* v0_ab[2] = {reinterpret_cast<uint32_t*>(p + 2 * e4), reinterpret_cast<uint32_t*>(indices)};
  uint32_t* v1_ab[2] = {reinterpret_cast<uint32_t*>(p + 3 * e4), reinterpret_cast<uint32_t*>(eid)};
  char* q = p + 4 * e4;
  uint32_t* hist = reinterpret_cast<uint32_t*>(q);
  const size_t hist_bytes = spmi::align_up(sizeof(uint32_t) * kMaxDigits * (size_t)n_tiles, 256);
  uint32_t* totals = reinterpret_cast<uint32_t*>(q + hist_bytes);
  q += sort_scratch_bytes(E, N);
  uint32_t* hist2 = reinterpret_cast<uint32_t*>(q);
  uint32_t* totals2 = reinterpret_cast<uint32_t*>(q + hist_bytes);
  q += sort_scratch_bytes(E, N);
  uint32_t* other_grouped = reinterpret_cast<uint32_t*>(q);
  q += e4;
  const size_t nb_bytes = spmi::align_up(sizeof(int32_t) * (size_t)(n_buckets + 1), 256);
  int32_t* bstart = reinterpret_cast<int32_t*>(q);
  uint32_t* seg_base = reinterpret_cast<uint32_t*>(q + nb_bytes);
  uint2* desc = reinterpret_cast<uint2*>(q + 2 * nb_bytes);
  uint32_t* table = reinterpret_cast<uint32_t*>(q + bucket_layout_bytes(E, n_buckets));
  int32_t* status = meta;
  const dim3 grid(n_tiles), block(kThreads);
  // the dual first pass: the other endpoints grouped into buckets of 4096 nodes on the side
  const int n_buckets2 = static_cast<int>((N_other + (1 << kCountLow) - 1) >> kCountLow);
  int bits2 = 1;
  while ((1 << bits2) < n_buckets2) ++bits2;
  OtherBuckets ob{other_grouped, hist2, totals2, kCountLow, bits2};
  int shift = kLow;
  for (int ps = 0; ps < passes; ++ps) {
    const int out = (passes - 1 - ps) & 1, in = out ^ 1;
    const int pb = pass_bits[ps];
    if (ps == 0 && norm_other) {
      radix_hist_dual_kernel<<<grid, block, 0, st>>>(key, other, E, (uint32_t)N, (uint32_t)N_other, shift, pb,
                                                     kCountLow, bits2, hist, hist2, n_tiles, status);
      SPMI_CHECK_LAUNCH("radix_hist_dual");
      radix_scan_kernel<<<dim3((1 << pb) + (1 << bits2)), block, 0, st>>>(hist, n_tiles, totals, hist2, totals2,
                                                                          1 << pb);
      SPMI_CHECK_LAUNCH("radix_scan(dual)");
      if (eid) {
        radix_scatter_kernel<int64_t, true, 2, true><<<grid, block, 0, st>>>(
            key, other, nullptr, nullptr, keys_ab[out], v0_ab[out], v1_ab[out], E, (uint32_t)N, (uint32_t)N_other,
            shift, pb, hist, totals, n_tiles, status, ob);
      } else {
        radix_scatter_kernel<int64_t, true, 1, true><<<grid, block, 0, st>>>(
            key, other, nullptr, nullptr, keys_ab[out], v0_ab[out], nullptr, E, (uint32_t)N, (uint32_t)N_other,
            shift, pb, hist, totals, n_tiles, status, ob);
      }
      SPMI_CHECK_LAUNCH("radix_scatter(dual)");
      shift += pb;
      continue;
    }
    if (ps == 0) {
      radix_hist_kernel<int64_t><<<grid, block, 0, st>>>(key, E, (uint32_t)N, shift, pb, hist, n_tiles, status);
    } else {
      radix_hist_kernel<uint32_t><<<grid, block, 0, st>>>(keys_ab[in], E, (uint32_t)N, shift, pb, hist, n_tiles,
                                                          status);
    }
    SPMI_CHECK_LAUNCH("radix_hist(bucket)");
    radix_scan_kernel<<<dim3(1 << pb), block, 0, st>>>(hist, n_tiles, totals);
    SPMI_CHECK_LAUNCH("radix_scan(bucket)");
    if (ps == 0 && eid) {
      radix_scatter_kernel<int64_t, true, 2><<<grid, block, 0, st>>>(
          key, other, nullptr, nullptr, keys_ab[out], v0_ab[out], v1_ab[out], E, (uint32_t)N, (uint32_t)N_other,
          shift, pb, hist, totals, n_tiles, status);
    } else if (ps == 0) {
      radix_scatter_kernel<int64_t, true, 1><<<grid, block, 0, st>>>(
          key, other, nullptr, nullptr, keys_ab[out], v0_ab[out], nullptr, E, (uint32_t)N, (uint32_t)N_other,
          shift, pb, hist, totals, n_tiles, status);
    } else if (eid) {
      radix_scatter_kernel<uint32_t, false, 2><<<grid, block, 0, st>>>(
          keys_ab[in], nullptr, v0_ab[in], v1_ab[in], keys_ab[out], v0_ab[out], v1_ab[out], E, (uint32_t)N,
          (uint32_t)N_other, shift, pb, hist, totals, n_tiles, status);
    } else {
      radix_scatter_kernel<uint32_t, false, 1><<<grid, block, 0, st>>>(
          keys_ab[in], nullptr, v0_ab[in], nullptr, keys_ab[out], v0_ab[out], nullptr, E, (uint32_t)N,
          (uint32_t)N_other, shift, pb, hist, totals, n_tiles, status);
    }
    SPMI_CHECK_LAUNCH("radix_scatter(bucket)");
    shift += pb;
  }
  const uint32_t* keys_b = keys_ab[0];
  const uint32_t* v0_b = v0_ab[0];
  const uint32_t* v1_b = v1_ab[0];
  const int sgrid = bucket_build_grid(E, n_buckets);
  SegMap map{};
  map.n_buckets = n_buckets;
  map.grid = sgrid;
  if (n_buckets <= kFastBuckets && passes == 1) {
    map.totals = totals;
  } else {
    csr_finalize_kernel<<<static_cast<int>((E + kThreads - 1) / kThreads), kThreads, 0, st>>>(
        keys_b, E, n_buckets, bstart, kLow);
    SPMI_CHECK_LAUNCH("bucket_bounds");
    seg_layout_kernel<<<1, 1024, 0, st>>>(bstart, n_buckets, seg_base, desc);
    SPMI_CHECK_LAUNCH("seg_layout");
    map.bstart = bstart;
    map.seg_base = seg_base;
    map.desc = desc;
  }
  seg_count_kernel<<<sgrid, kSegThreads, 0, st>>>(keys_b, map, table);
  SPMI_CHECK_LAUNCH("seg_count");
  if (eid) {
    seg_place_kernel<2><<<sgrid, kSegThreads, 0, st>>>(
        keys_b, v0_b, v1_b, map, table, E, N, indptr, reinterpret_cast<uint32_t*>(indices),
        reinterpret_cast<uint32_t*>(eid), norm_key, norm_mode, heavy_thresh, heavy_rows, capacity - 1, meta + 1);
  } else {
    seg_place_kernel<1><<<sgrid, kSegThreads, 0, st>>>(
        keys_b, v0_b, nullptr, map, table, E, N, indptr, reinterpret_cast<uint32_t*>(indices), nullptr, norm_key,
        norm_mode, heavy_thresh, heavy_rows, capacity - 1, meta + 1);
  }
  SPMI_CHECK_LAUNCH("seg_place");
  plan_finish_kernel<<<1, kPlanThreads, 0, st>>>(indptr, chunk, heavy_rows, heavy_chunk_ptr, capacity - 1,
                                             meta + 1);
  SPMI_CHECK_LAUNCH("plan_finish(bucket)");
  if (norm_other) {
    bucket_count_kernel<<<n_buckets2, kCountThreads, 0, st>>>(other_grouped, totals2, nullptr, N_other, nullptr,
                                                              norm_other, norm_mode);
    SPMI_CHECK_LAUNCH("bucket_count(other)");
  }
  return SPMI_OK;
}

size_t spmi_degree_workspace_bytes(int64_t E, int64_t N) {
  if (E < 0 || N < 1) return 0;
  const size_t e4 = spmi::align_up(sizeof(uint32_t) * (size_t)E, 256);
  return 2 * e4 + spmi::align_up(sizeof(int32_t) * (size_t)(N + 1), 256) + sort_scratch_bytes(E, N);
}

int spmi_degree(const int64_t* ids, int64_t E, int64_t N, int64_t* deg, void* ws, size_t ws_bytes,
                int32_t* status, void* stream) {
  if (int rc = check_sizes(E, N, "spmi_degree")) return rc;
  SPMI_REQUIRE(deg && status, SPMI_ERR_INVALID_ARG, "spmi_degree: null deg/status");
  hipStream_t st = spmi::as_stream(stream);
  if (E == 0) {
    SPMI_CHECK_HIP(hipMemsetAsync(deg, 0, sizeof(int64_t) * (size_t)N, st));
    return SPMI_OK;
  }
  SPMI_REQUIRE(ids && ws, SPMI_ERR_INVALID_ARG, "spmi_degree: null pointer argument");
  SPMI_REQUIRE(ws_bytes >= spmi_degree_workspace_bytes(E, N), SPMI_ERR_WORKSPACE,
               "spmi_degree: workspace %zu < required %zu", ws_bytes,
               spmi_degree_workspace_bytes(E, N));
  SortShape sh = sort_shape(E, N);
  const size_t e4 = spmi::align_up(sizeof(uint32_t) * (size_t)E, 256);
  const size_t np = spmi::align_up(sizeof(int32_t) * (size_t)(N + 1), 256);
  char* p = static_cast<char*>(ws);
  SortBuffers b{};
  b.keys[0] = reinterpret_cast<uint32_t*>(p);
  b.keys[1] = reinterpret_cast<uint32_t*>(p + e4);
  b.v0[0] = b.v0[1] = b.v1[0] = b.v1[1] = nullptr;
  int32_t* indptr = reinterpret_cast<int32_t*>(p + 2 * e4);
  b.hist = reinterpret_cast<uint32_t*>(p + 2 * e4 + np);
  b.totals = reinterpret_cast<uint32_t*>(
      p + 2 * e4 + np + spmi::align_up(sizeof(uint32_t) * kMaxDigits * (size_t)sh.n_tiles, 256));
  if (N <= kBucketDegreeMaxN)
    return degree_by_buckets<int64_t>(ids, E, N, deg, nullptr, 0, b.keys[0], b.keys[1], indptr, b.hist, b.totals,
                                      status, st);
  if (int rc = run_sort<0>(ids, nullptr, E, N, 1, sh, b, status, st)) return rc;
  const int grid = static_cast<int>((E + kThreads - 1) / kThreads);
  csr_finalize_kernel<<<grid, kThreads, 0, st>>>(b.keys[0], E, N, indptr);
  SPMI_CHECK_LAUNCH("csr_finalize(degree)");
  indptr_to_degree_kernel<<<(int)((N + 255) / 256), 256, 0, st>>>(indptr, N, deg);
  SPMI_CHECK_LAUNCH("indptr_to_degree");
  return SPMI_OK;
}

int spmi_norm_from_ids32(const int32_t* ids, int64_t E, int64_t N, int mode, float* norm,
                         int64_t* deg, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_sizes(E, N, "spmi_norm_from_ids32")) return rc;
  SPMI_REQUIRE(norm, SPMI_ERR_INVALID_ARG, "spmi_norm_from_ids32: null norm");
  SPMI_REQUIRE(mode == SPMI_NORM_RSQRT || mode == SPMI_NORM_RECIP, SPMI_ERR_INVALID_ARG,
               "spmi_norm_from_ids32: bad mode %d", mode);
  SPMI_REQUIRE(ws && (E == 0 || ids), SPMI_ERR_INVALID_ARG, "spmi_norm_from_ids32: null pointer argument");
  SPMI_REQUIRE(ws_bytes >= spmi_degree_workspace_bytes(E, N), SPMI_ERR_WORKSPACE,
               "spmi_norm_from_ids32: workspace %zu < required %zu", ws_bytes,
               spmi_degree_workspace_bytes(E, N));
  hipStream_t st = spmi::as_stream(stream);
  const SortShape sh = sort_shape(E, N);
  const size_t e4 = spmi::align_up(sizeof(uint32_t) * (size_t)E, 256);
  const size_t np = spmi::align_up(sizeof(int32_t) * (size_t)(N + 1), 256);
  char* p = static_cast<char*>(ws);
  SortBuffers b{};
  b.keys[0] = reinterpret_cast<uint32_t*>(p);
  b.keys[1] = reinterpret_cast<uint32_t*>(p + e4);
  int32_t* indptr = reinterpret_cast<int32_t*>(p + 2 * e4);
  b.hist = reinterpret_cast<uint32_t*>(p + 2 * e4 + np);
  b.totals = reinterpret_cast<uint32_t*>(
      p + 2 * e4 + np + spmi::align_up(sizeof(uint32_t) * kMaxDigits * (size_t)sh.n_tiles, 256));
  const uint32_t* keys = reinterpret_cast<const uint32_t*>(ids);
  const dim3 grid(sh.n_tiles), block(kThreads);
  int32_t* no_status = nullptr;  // 32-bit keys are never range-checked (checked_key passes them through)
  if (E > 0 && N <= kBucketDegreeMaxN)
    return degree_by_buckets<uint32_t>(keys, E, N, deg, norm, mode, b.keys[0], b.keys[1], indptr, b.hist,
                                       b.totals, no_status, st);
  if (E == 0) {
    SPMI_CHECK_HIP(hipMemsetAsync(indptr, 0, sizeof(int32_t) * (size_t)(N + 1), st));
  } else {
    // keys-only sort of the 32-bit ids, then run boundaries
    int shift = 0;
    for (int q = 0; q < sh.passes; ++q) {
      const int bits = sh.pass_bits[q];
      const int out = (sh.passes - 1 - q) & 1;
      const uint32_t* in = q == 0 ? keys : b.keys[out ^ 1];
      radix_hist_kernel<uint32_t><<<grid, block, 0, st>>>(in, E, (uint32_t)N, shift, bits, b.hist,
                                                          sh.n_tiles, no_status);
      SPMI_CHECK_LAUNCH("radix_hist(ids32)");
      radix_scan_kernel<<<dim3(1 << bits), block, 0, st>>>(b.hist, sh.n_tiles, b.totals);
      SPMI_CHECK_LAUNCH("radix_scan(ids32)");
      radix_scatter_kernel<uint32_t, false, 0><<<grid, block, 0, st>>>(
          in, nullptr, nullptr, nullptr, b.keys[out], nullptr, nullptr, E, (uint32_t)N, 1u, shift, bits,
          b.hist, b.totals, sh.n_tiles, no_status);
      SPMI_CHECK_LAUNCH("radix_scatter(ids32)");
      shift += bits;
    }
    const int fgrid = static_cast<int>((E + kThreads - 1) / kThreads);
    csr_finalize_kernel<<<fgrid, kThreads, 0, st>>>(b.keys[0], E, N, indptr);
    SPMI_CHECK_LAUNCH("csr_finalize(ids32)");
  }
  norm_kernel<<<(int)((N + 255) / 256), 256, 0, st>>>(indptr, N, mode, norm);
  SPMI_CHECK_LAUNCH("norm(ids32)");
  if (deg) {
    indptr_to_degree_kernel<<<(int)((N + 255) / 256), 256, 0, st>>>(indptr, N, deg);
    SPMI_CHECK_LAUNCH("indptr_to_degree(ids32)");
  }
  return SPMI_OK;
}

int spmi_norm_from_indptr(const int32_t* indptr, int64_t N, int mode, float* norm, void* stream) {
  SPMI_REQUIRE(indptr && norm, SPMI_ERR_INVALID_ARG, "spmi_norm_from_indptr: null pointer");
  SPMI_REQUIRE(N >= 0 && N < (int64_t{1} << 31), SPMI_ERR_INVALID_ARG,
               "spmi_norm_from_indptr: bad N=%lld", (long long)N);
  SPMI_REQUIRE(mode == SPMI_NORM_RSQRT || mode == SPMI_NORM_RECIP, SPMI_ERR_INVALID_ARG,
               "spmi_norm_from_indptr: bad mode %d", mode);
  if (N == 0) return SPMI_OK;
  norm_kernel<<<(int)((N + 255) / 256), 256, 0, spmi::as_stream(stream)>>>(indptr, N, mode, norm);
  SPMI_CHECK_LAUNCH("norm");
  return SPMI_OK;
}

int spmi_norm_from_degree(const int64_t* deg, int64_t N, int mode, float* norm, void* stream) {
  SPMI_REQUIRE(deg && norm, SPMI_ERR_INVALID_ARG, "spmi_norm_from_degree: null pointer");
  SPMI_REQUIRE(N >= 0 && N < (int64_t{1} << 31), SPMI_ERR_INVALID_ARG,
               "spmi_norm_from_degree: bad N=%lld", (long long)N);
  SPMI_REQUIRE(mode == SPMI_NORM_RSQRT || mode == SPMI_NORM_RECIP, SPMI_ERR_INVALID_ARG,
               "spmi_norm_from_degree: bad mode %d", mode);
  if (N == 0) return SPMI_OK;
  norm_from_degree_kernel<<<(int)((N + 255) / 256), 256, 0, spmi::as_stream(stream)>>>(deg, N, mode,
                                                                                      norm);
  SPMI_CHECK_LAUNCH("norm_from_degree");
  return SPMI_OK;
}

int spmi_spmm_plan(const int32_t* indptr, int64_t N, int32_t heavy_thresh, int32_t chunk,
                   int32_t* heavy_rows, int32_t* heavy_chunk_ptr, int64_t capacity,
                   int32_t* counts, void* stream) {
  SPMI_REQUIRE(indptr && heavy_rows && heavy_chunk_ptr && counts, SPMI_ERR_INVALID_ARG,
               "spmi_spmm_plan: null pointer");
  SPMI_REQUIRE(N >= 0 && N < (int64_t{1} << 31) && heavy_thresh >= 1 && chunk >= 1 &&
                   capacity >= 2,
               SPMI_ERR_INVALID_ARG, "spmi_spmm_plan: bad sizes");
  hipStream_t st = spmi::as_stream(stream);
  SPMI_CHECK_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), st));
  if (N > 0) {
    const int64_t per_wg = kPlanThreads * kPlanRows;
    plan_collect_kernel<<<(int)((N + per_wg - 1) / per_wg), kPlanThreads, 0, st>>>(
        indptr, N, heavy_thresh, heavy_rows, capacity - 1, counts);
    SPMI_CHECK_LAUNCH("plan_collect");
  }
  plan_finish_kernel<<<1, kPlanThreads, 0, st>>>(indptr, chunk, heavy_rows, heavy_chunk_ptr,
                                             capacity - 1, counts);
  SPMI_CHECK_LAUNCH("plan_finish");
  return SPMI_OK;
}

}  // extern "C"
