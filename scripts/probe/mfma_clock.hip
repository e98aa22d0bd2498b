// Diagnostic: the rate at which v_mfma_f32_32x32x16_f16 issues chip-wide, against wall time and
// against s_memtime, with 8 waves per CU (two per SIMD) on every CU: what clock the matrix
// pipe really runs at under a dense fp16 load.  hipcc --offload-arch=gfx950 -O3 mfma_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = float __attribute__((ext_vector_type(16)));
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));
template <int FILL, int BLOCKED>
__global__ __launch_bounds__(512, 1) void spin(int iters, float* out, long long* ticks) {
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
  f16x8 a, b;
  for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(threadIdx.x * 0.001f + q); b[q] = (_Float16)(q * 0.5f); }
  float f[8];
  for (int q = 0; q < 8; ++q) f[q] = threadIdx.x * 0.5f + q;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (BLOCKED) {
      // all the vector work of 16 matrix instructions in one block, then the matrix instructions
#pragma unroll
      for (int k = 0; k < 16 * FILL; ++k) f[k & 7] = __builtin_fmaf(f[k & 7], 1.0001f, 0.5f);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q], 0, 0, 0);
#pragma unroll
          for (int k = 0; k < FILL; ++k) f[k & 7] = __builtin_fmaf(f[k & 7], 1.0001f, 0.5f);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int q = 0; q < 4; ++q) for (int e = 0; e < 16; ++e) s += acc[q][e];
  for (int q = 0; q < 8; ++q) s += f[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; long long* ticks;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, void (*k)(int, float*, long long*)) {
    for (int rep = 0; rep < 2; ++rep) {
      const int iters = 20000;  // 16 MFMAs per iteration per wave
      hipEventRecord(e0);
      k<<<256, 512>>>(iters, out, ticks);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      const double mfma_per_simd = 2.0 * 16 * iters;  // two waves per SIMD
      printf("%-28s wall %.3f ms  ticks %lld  %.1f ns per MFMA per SIMD  %.0f TFLOP/s\n", name, ms, t,
             ms * 1e6 / mfma_per_simd, mfma_per_simd * 1024 * 32768.0 / (ms * 1e-3) / 1e12);
    }
  };
  run("bare", spin<0, 0>);
  run("3 fma after each MFMA", spin<3, 0>);
  run("5 fma after each MFMA", spin<5, 0>);
  run("8 fma after each MFMA", spin<8, 0>);
  run("3 fma per MFMA, in a block", spin<3, 1>);
  run("5 fma per MFMA, in a block", spin<5, 1>);
  run("8 fma per MFMA, in a block", spin<8, 1>);
  return 0;
}
