// Probe: cycles per v_mfma_f32_32x32x16_f16 in a bare stream, every CU busy, and the chip-wide
// rate that results.  Variants: waves per SIMD (1 or 2), operands random or zero, 8 or 2
// accumulators.  hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_rate.hip -o _bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512, 2) void probe(const f16x8* a, const f16x8* b, float* out, unsigned long long* ticks, int iters, int active_waves) {
  const int wave = threadIdx.x >> 6;
  if (wave >= active_waves) return;
  f16x8 av[4], bv[4];
  for (int i = 0; i < 4; ++i) { av[i] = a[threadIdx.x * 4 + i]; bv[i] = b[threadIdx.x * 4 + i]; }
  f32x16 acc[NACC];
  for (int q = 0; q < NACC; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0.0f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u)
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[u & 3], bv[(u >> 2) & 3], acc[u % NACC], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.0f;
  for (int q = 0; q < NACC; ++q) for (int e = 0; e < 16; ++e) s += acc[q][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
  const int blocks = 256, threads = 512, iters = 2000;
  std::vector<_Float16> ha(threads * 32), hb(threads * 32);
  f16x8 *a, *b; float* out; unsigned long long* ticks;
  hipMalloc(&a, threads * 64); hipMalloc(&b, threads * 64); hipMalloc(&out, blocks * threads * 4); hipMalloc(&ticks, blocks * 8 * 8);
  for (int zero = 0; zero < 2; ++zero) {
    for (size_t i = 0; i < ha.size(); ++i) { ha[i] = zero ? (_Float16)0.0f : (_Float16)((rand() % 2001 - 1000) / 1000.0f); hb[i] = zero ? (_Float16)0.0f : (_Float16)((rand() % 2001 - 1000) / 10000.0f); }
    hipMemcpy(a, ha.data(), threads * 64, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), threads * 64, hipMemcpyHostToDevice);
    for (int waves : {4, 8}) for (int nacc : {8, 2}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipMemset(ticks, 0, blocks * 64);
        hipEventRecord(e0);
        if (nacc == 8) probe<8><<<blocks, threads>>>(a, b, out, ticks, iters, waves);
        else probe<2><<<blocks, threads>>>(a, b, out, ticks, iters, waves);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      std::vector<unsigned long long> ht(blocks * 8);
      hipMemcpy(ht.data(), ticks, blocks * 64, hipMemcpyDeviceToHost);
      double sum = 0; int n = 0; for (auto t : ht) if (t) { sum += (double)t; ++n; }
      const double per_wave = sum / n / (iters * 32.0);
      printf("%s operands, %d waves/WG (%d per SIMD), %d accumulators: %.1f ticks per MFMA per wave = %.1f per SIMD, kernel %.3f ms, %.0f TFLOP/s\n",
             zero ? "zero" : "random", waves, waves / 4, nacc, per_wave, per_wave / (waves / 4), best,
             2.0 * 32 * 32 * 16 * 32.0 * iters * waves * blocks / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
