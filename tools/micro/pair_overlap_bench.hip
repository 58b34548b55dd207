// Two waves on one SIMD: do the vector instructions of one wave issue under the MFMAs of the other?
//   mode 0  every wave: NM MFMAs (two accumulators alternating) per iteration, nothing else
//   mode 1  every wave: NV v_fma_f32 per iteration, nothing else
//   mode 2  waves 0-3 the MFMAs, waves 4-7 (their SIMD partners) the vector work
//   mode 3  every wave both, in blocks: NM MFMAs, then NV vector instructions
//   mode 4  every wave both, woven: after every MFMA NV / NM vector instructions
//   mode 5  as 4, one accumulator (every MFMA depends on the one before)
//   mode 6  as 4, and an s_nop 7 after the vector instructions (the wave stays away from the issue port)
//   mode 7  as 4, and two s_nop 7
// hipcc --offload-arch=gfx950 -O3 tools/micro/pair_overlap_bench.hip -o /tmp/pair_overlap_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NM, int NV>
__global__ __launch_bounds__(512) void bench(float* out, int iters) {
  f32x16 a0 = {0}, a1 = {0};
  bf16x8 x = {}, y = {};
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 1e-3f + i;
  const float c = 0.999f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool do_m = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
  const bool do_v = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 4) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        if ((m & 1) && MODE != 5) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
#pragma unroll
        for (int v = 0; v < NV / NM; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[v & 7]) : "v"(c));
        if (MODE == 6) asm volatile("s_nop 7");
        if (MODE == 7) asm volatile("s_nop 7\n\ts_nop 7");
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          if (m & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
        }
      }
      if (do_v) {
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[v & 7]) : "v"(c));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NM, int NV>
void run(const char* name, int threads, float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  bench<MODE, NM, NV><<<256, threads>>>(out, 100);
  hipEventRecord(e0);
  bench<MODE, NM, NV><<<256, threads>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s NM %2d NV %3d waves/SIMD %d : %8.1f ns/iter\n", name, NM, NV, threads / 256, ms * 1e6 / iters);
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<0, 12, 0>("mfma only", 256, out);
  run<0, 12, 0>("mfma only", 512, out);
  run<1, 0, 60>("valu only", 256, out);
  run<1, 0, 60>("valu only", 512, out);
  run<2, 24, 120>("A mfma (24) | B valu (120)", 512, out);
  run<2, 24, 60>("A mfma (24) | B valu (60)", 512, out);
  run<2, 24, 240>("A mfma (24) | B valu (240)", 512, out);
  run<3, 12, 60>("blocks 12 mfma + 60 valu", 256, out);
  run<3, 12, 60>("blocks 12 mfma + 60 valu", 512, out);
  run<3, 6, 30>("blocks 6 mfma + 30 valu", 512, out);
  run<3, 2, 10>("blocks 2 mfma + 10 valu", 512, out);
  run<4, 12, 60>("woven 1 mfma + 5 valu", 256, out);
  run<4, 12, 60>("woven 1 mfma + 5 valu", 512, out);
  run<4, 12, 36>("woven 1 mfma + 3 valu", 256, out);
  run<4, 12, 36>("woven 1 mfma + 3 valu", 512, out);
  run<4, 12, 24>("woven 1 mfma + 2 valu", 512, out);
  run<4, 12, 72>("woven 1 mfma + 6 valu", 256, out);
  run<4, 12, 72>("woven 1 mfma + 6 valu", 512, out);
  run<5, 12, 0>("dependent mfma only", 256, out);
  run<5, 12, 0>("dependent mfma only", 512, out);
  run<5, 12, 36>("dependent woven 1 + 3", 256, out);
  run<5, 12, 36>("dependent woven 1 + 3", 512, out);
  run<5, 12, 72>("dependent woven 1 + 6", 512, out);
  run<6, 12, 36>("woven 1 + 3 + nop7", 256, out);
  run<6, 12, 36>("woven 1 + 3 + nop7", 512, out);
  run<7, 12, 36>("woven 1 + 3 + 2 nop7", 256, out);
  run<7, 12, 36>("woven 1 + 3 + 2 nop7", 512, out);
  run<6, 12, 48>("woven 1 + 4 + nop7", 512, out);
  run<7, 12, 48>("woven 1 + 4 + 2 nop7", 512, out);
  return 0;
}
