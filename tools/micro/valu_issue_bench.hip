// Issue cost (cycles per wave64 instruction) of the vector instructions the kNN's selection is made of, with one and
// with two waves per SIMD, alone and beside v_mfma_f32_32x32x16_f16 of the same wave.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_issue_bench.hip -o tools/micro/valu_issue_bench && tools/micro/valu_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define REP8(x) x x x x x x x x
#define OP_MED3 asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(a0) : "v"(a1), "v"(a2), "v"(a3));
#define OP_MAX3 asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(a0) : "v"(a1), "v"(a2), "v"(a3));
#define OP_ALIGN asm volatile("v_alignbit_b32 %0, %1, %2, 31" : "=v"(a0) : "v"(a1), "v"(a2));
#define OP_SUB asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a0) : "v"(a1), "v"(a2));
#define OP_MAX asm volatile("v_max_f32 %0, %1, %2" : "=v"(a0) : "v"(a1), "v"(a2));
#define OP_CND asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a0) : "v"(a1), "v"(a2));
#define OP_MAXF64 asm volatile("v_max_f64 %0, %1, %2" : "=v"(d0) : "v"(d1), "v"(d2));

template <int WHICH, bool MFMA>
__global__ void bench(long long* out, float* sink) {
  float a0 = threadIdx.x, a1 = a0 * 2.f, a2 = a0 + 3.f, a3 = a0 - 1.f;
  double d0 = a0, d1 = a1, d2 = a2;
  f32x16 acc = {};
  f16x8 x = {}, y = {};
  long long t0 = clock64();
  for (int it = 0; it < 256; ++it) {
    if (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc, 0, 0, 0);
    if (WHICH == 0) { REP8(OP_MED3) }
    if (WHICH == 1) { REP8(OP_MAX3) }
    if (WHICH == 2) { REP8(OP_ALIGN) }
    if (WHICH == 3) { REP8(OP_SUB) }
    if (WHICH == 4) { REP8(OP_MAX) }
    if (WHICH == 5) { REP8(OP_CND) }
    if (WHICH == 6) { REP8(OP_MAXF64) }
    if (WHICH == 7) { }
  }
  long long t1 = clock64();
  if (threadIdx.x % 64 == 0) out[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
  sink[threadIdx.x] = a0 + acc[0] + (float)d0;
}

template <int W, bool M>
void run(const char* name, int threads, long long* d_out, float* sink) {
  hipLaunchKernelGGL((bench<W, M>), dim3(256), dim3(threads), 0, 0, d_out, sink);
  long long h[8];
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-10s %s waves/SIMD=%d  cycles per iteration (8 ops%s): %.1f\n", name, M ? "+MFMA" : "     ", threads / 256,
         M ? " + 1 MFMA" : "", (double)h[0] / 256);
}

int main() {
  long long* d_out; float* sink;
  hipMalloc(&d_out, 256 * 8 * 8); hipMalloc(&sink, 4096);
  for (int threads : {256, 512}) {
#define R(W, n) run<W, false>(n, threads, d_out, sink); run<W, true>(n, threads, d_out, sink);
    R(0, "v_med3") R(1, "v_max3") R(2, "alignbit") R(3, "v_sub") R(4, "v_max") R(5, "cndmask") R(6, "max_f64") R(7, "none")
  }
  return 0;
}
