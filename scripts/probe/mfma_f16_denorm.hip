// Does v_mfma_f32_32x32x16_f16 keep subnormal fp16 inputs?  A = 2^-20 (subnormal in fp16)
// in every element, B = 2^10: each output is 16 * 2^-10 = 2^-6 if inputs are kept, 0 if flushed.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = float __attribute__((ext_vector_type(16)));
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));
__global__ void k(float a, float b, float* out) {
  f16x8 va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = static_cast<_Float16>(a); vb[i] = static_cast<_Float16>(b); }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
  float* d; float h = -1.0f;
  hipMalloc(&d, 4);
  k<<<1, 64>>>(0x1p-20f, 0x1p10f, d);
  hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  printf("subnormal A: got %g, kept = %g, flushed = 0\n", h, 0x1p-6);
  k<<<1, 64>>>(0x1p10f, 0x1p-20f, d);
  hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  printf("subnormal B: got %g, kept = %g, flushed = 0\n", h, 0x1p-6);
  k<<<1, 64>>>(0x1p-20f, 0x1p-20f, d);
  hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  printf("both subnormal: got %g, exact = %g\n", h, 16 * 0x1p-40);
  return 0;
}
