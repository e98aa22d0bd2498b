// Probe: how fast can a kernel shaped like the row-streaming kernels (one workgroup per CU, a
// wave owns 32-row tiles, 16 KB of loads in flight per wave, rows refilled K-step by K-step)
// READ a [M, 128] float matrix when it does nothing else, (a) with the operand-order loads of
// those kernels (a lane reads 64 contiguous bytes of ITS row: an instruction touches 32 rows)
// and (b) with whole-line loads (an instruction reads 1 KB of consecutive memory)?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 row_loads.hip -o _bin/row_loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <bool COALESCED, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void read_rows(const float* __restrict__ x, int64_t M, float* out) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t n_tiles = M / 32;
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  int64_t tile = (int64_t)wave * gridDim.x + blockIdx.x;
  if (tile >= n_tiles) return;
  float4 a[16];
  auto src_of = [&](int64_t t) {
    return COALESCED ? x + t * 32 * 128 + lane * 4 : x + (t * 32 + lr) * 128 + 16 * lh;
  };
  auto off = [&](int i) {  // float offset of register i
    return COALESCED ? i * 256 : 32 * (i >> 2) + 4 * (i & 3);
  };
  {
    const float* s = src_of(tile);
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = *reinterpret_cast<const float4*>(s + off(i));
  }
  float acc = 0.0f;
  while (tile < n_tiles) {
    const int64_t next = tile + stride;
    const float* s = src_of(next < n_tiles ? next : tile);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc += a[i].x + a[i].y + a[i].z + a[i].w;
      a[i] = *reinterpret_cast<const float4*>(s + off(i));
      __builtin_amdgcn_sched_barrier(0);
    }
    tile = next;
  }
  if (acc == 12345.678f) out[threadIdx.x] = acc;
}

template <bool C, int W>
void run(const float* x, int64_t M, float* out, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    (void)hipEventRecord(e0);
    read_rows<C, W><<<256, W * 64>>>(x, M, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("%-34s %d waves/CU: %.3f ms, %.2f TB/s\n", name, W, best, M * 512.0 / (best * 1e-3) / 1e12);
}

int main() {
  const int64_t M = 1 << 20;
  float *x, *out;
  (void)hipMalloc(&x, M * 512 * 4);  // four matrices' worth, so that repeats do not sit in the caches
  (void)hipMalloc(&out, 4096);
  (void)hipMemset(x, 0, M * 512 * 4);
  for (int r = 0; r < 2; ++r) {
    run<false, 4>(x, M, out, "operand order (lane = row)");
    run<false, 8>(x, M, out, "operand order (lane = row)");
    run<false, 16>(x, M, out, "operand order (lane = row)");
    run<true, 4>(x, M, out, "whole lines (1 KB per instruction)");
    run<true, 8>(x, M, out, "whole lines (1 KB per instruction)");
    run<true, 16>(x, M, out, "whole lines (1 KB per instruction)");
  }
  return 0;
}
