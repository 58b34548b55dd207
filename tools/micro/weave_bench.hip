// What one MFMA slot costs a wave when two waves share a SIMD: loop of { v_mfma_f32_32x32x16_f16 on one accumulator chain;
// K independent vector fillers; R ds_read_b128 } -- cycles per slot per wave, 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/weave_bench.hip -o tools/micro/weave_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int K, int R>
__global__ void bench(long long* out, float* sink) {
  __shared__ f32x4 tile[2048];
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) tile[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  f32x16 acc = {};
  f16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(threadIdx.x * 0.01f + i); y[i] = (_Float16)(i - threadIdx.x * 0.02f); }
  f32x4 l = {};
  long long t0 = clock64();
  for (int it = 0; it < 512; ++it) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < K; ++k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[k % 8]) : "v"(v[(k + 1) % 8]), "v"(v[(k + 3) % 8]));
#pragma unroll
    for (int r = 0; r < R; ++r) {
      f32x4 t = tile[(threadIdx.x + 64 * r + it) & 2047];
      asm volatile("" : "+v"(t));
      l += t;
    }
    asm volatile("" : "+v"(acc));
  }
  long long t1 = clock64();
  if (threadIdx.x % 64 == 0) out[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
  float s = acc[0] + l[0];
  for (int i = 0; i < 8; ++i) s += v[i];
  sink[threadIdx.x] = s;
}

template <int K, int R>
void run(long long* d_out, float* sink) {
  for (int threads : {256, 512, 1024}) {
    hipLaunchKernelGGL((bench<K, R>), dim3(256), dim3(threads), 0, 0, d_out, sink);
    long long h[8];
    (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    printf("fillers %d reads %d  waves/SIMD=%d  cycles per slot: %.1f\n", K, R, threads / 256, (double)h[0] / 512);
  }
}

int main() {
  long long* d_out; float* sink;
  (void)hipMalloc(&d_out, 256 * 8 * 8); (void)hipMalloc(&sink, 4096);
  run<0, 0>(d_out, sink); run<4, 0>(d_out, sink);
  return 0;
}
