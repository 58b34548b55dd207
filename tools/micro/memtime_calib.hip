// Is s_memtime a constant-rate clock?  Spin until 2e6 ticks have passed, idle and under MFMA load, and time it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <bool LOAD>
__global__ void spin(unsigned long long ticks, float* out) {
  f32x16 acc = {0};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.01f + i); b[i] = (__bf16)(1.f + 0.1f * i); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < ticks) {
    if (LOAD) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    } else {
      __builtin_amdgcn_s_sleep(8);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <bool LOAD>
void run(int threads, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(spin<LOAD>, dim3(256), dim3(threads), 0, 0, 100000ull, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL(spin<LOAD>, dim3(256), dim3(threads), 0, 0, 2000000ull, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s, %d threads/WG: 2e6 ticks took %.3f ms -> %.1f MHz\n", LOAD ? "MFMA load" : "idle", threads, ms, 2e6 / (ms * 1e3));
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<false>(256, out); run<true>(256, out); run<true>(512, out); run<false>(512, out);
  return 0;
}
