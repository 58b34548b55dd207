// Does VALU work issue under MFMAs on gfx950?  Per iteration a wave issues 4 independent
// v_mfma_f32_32x32x2_f32 (64 cycles of matrix pipe each) and NV "filler" instructions of one kind
// between them.  If fillers co-issue, time/iteration stays 4 x 64 cycles until the fillers' own
// issue time exceeds it.  Build: hipcc --offload-arch=gfx950 -O3 coissue_bench.hip -o coissue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int NV>
__global__ __launch_bounds__(512) void bench(float* out, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  float f[8];
  double d[8];
  for (int i = 0; i < 8; ++i) { f[i] = x + i; d[i] = x + 2.0 * i; }
  const float c = 0.999f;
  const double dc = 1.5;
  __shared__ float lds[2048];
  lds[threadIdx.x] = x; lds[threadIdx.x + 512] = x; lds[threadIdx.x + 1024] = x; lds[threadIdx.x + 1536] = x;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (KIND != 9) {
        if (m == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
        if (m == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
        if (m == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y));
        if (m == 3) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(x), "v"(y));
      }
#pragma unroll
      for (int v = 0; v < NV / 4; ++v) {
        const int r = v & 7;
        if (KIND == 1 || KIND == 9) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[r]) : "v"(c));
        if (KIND == 2) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[r]) : "v"(dc));
        if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(f[r]));
        if (KIND == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[r]) : "v"(dc));
        if (KIND == 5) asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(f[r]) : "v"((int)(threadIdx.x * 4)));
        if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[r]) : "v"(c));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
  for (int i = 0; i < 8; ++i) s += f[i] + (float)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NV>
void run(const char* name, int waves_per_simd, float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256), block(256 * waves_per_simd);
  bench<KIND, NV><<<grid, block>>>(out, 100);
  hipEventRecord(e0);
  bench<KIND, NV><<<grid, block>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // cycles per iteration per SIMD at 2.4 GHz nominal (the real clock is lower under MFMA load)
  printf("%-14s fillers/iter %3d  waves/SIMD %d : %8.1f ns/iter  (%6.0f cyc @2.4GHz; MFMA-only floor %d x 256)\n", name, NV,
         waves_per_simd, ms * 1e6 / iters, ms * 1e6 / iters * 2.4, waves_per_simd);
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  for (int w = 1; w <= 2; ++w) {
    run<1, 0>("mfma only", w, out);
    run<9, 32>("f32 fma alone", w, out);
    run<1, 16>("f32 fma", w, out); run<1, 32>("f32 fma", w, out); run<1, 64>("f32 fma", w, out); run<1, 128>("f32 fma", w, out);
    run<2, 16>("f64 max", w, out); run<2, 32>("f64 max", w, out); run<2, 64>("f64 max", w, out);
    run<4, 16>("f64 fma", w, out); run<4, 32>("f64 fma", w, out);
    run<3, 16>("f32 exp", w, out); run<3, 32>("f32 exp", w, out);
    run<6, 32>("cndmask", w, out); run<6, 64>("cndmask", w, out);
    run<5, 16>("ds_read+wait", w, out);
  }
  return 0;
}
