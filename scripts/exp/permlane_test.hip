// gfx950's v_permlane16_swap / v_permlane32_swap as the 16- and 32-lane levels of a wave butterfly, against __shfl_xor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__device__ __forceinline__ float xsum16(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float xsum32(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__global__ void k(const float* in, float* o1, float* o2, float* o3, float* o4) {
  const int t = threadIdx.x;
  float v = in[t];
  o1[t] = v + __shfl_xor(v, 16, 64);
  o2[t] = xsum16(v);
  o3[t] = v + __shfl_xor(v, 32, 64);
  o4[t] = xsum32(v);
}
int main() {
  float h[64], *d, *o[4], r[4][64];
  for (int i = 0; i < 64; ++i) h[i] = sinf(i * 1.7f) * 100.f + i;
  hipMalloc(&d, 256); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  for (int k2 = 0; k2 < 4; ++k2) hipMalloc(&o[k2], 256);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o[0], o[1], o[2], o[3]);
  for (int k2 = 0; k2 < 4; ++k2) hipMemcpy(r[k2], o[k2], 256, hipMemcpyDeviceToHost);
  int bad16 = 0, bad32 = 0;
  for (int i = 0; i < 64; ++i) { bad16 += r[0][i] != r[1][i]; bad32 += r[2][i] != r[3][i]; }
  printf("xor16 mismatches %d, xor32 mismatches %d\n", bad16, bad32);
  for (int i = 0; i < 64; i += 9) printf("lane %2d: shfl16 %g swap16 %g | shfl32 %g swap32 %g\n", i, r[0][i], r[1][i], r[2][i], r[3][i]);
  return bad16 || bad32;
}
