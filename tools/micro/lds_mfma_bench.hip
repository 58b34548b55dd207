// What rate does the matrix pipe reach when every MFMA operand comes from LDS the way the step's kernels fetch it?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../samble_amd/csrc lds_mfma_bench.hip -o lds_mfma_bench
// One workgroup per CU, WAVES waves (4 = one per SIMD, 8 = two), a 24 KB operand tile in LDS; every wave runs
// `iters` tiles of 48 MFMAs (8 steps x 6 split-bf16 products) with its A operand read from LDS per step
// (3 x ds_read_b128) in the given prefetch DEPTH, B operand in registers.  Prints s_memtime ticks (= shader cycles,
// tools/micro/memtime_calib.hip) per MFMA per SIMD, the rate, and the clock the ticks imply.
// Measured: one wave per SIMD 32.3-35 ticks per MFMA = 285-341 TFLOP/s fp32-equivalent at 1.75-2.1 GHz; two waves per
// SIMD 337-354 TFLOP/s with the clock at 1.0-1.1 GHz by the same counter -- on this data the chip delivers ~2.1 PFLOP/s
// of bf16 MFMA work whichever way it is issued (power), and trades clock for issue density.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "tri_dev.h"
using namespace samble;

template <int WAVES, int DEPTH, int NV>
__global__ __launch_bounds__(WAVES * 64) void bench(const unsigned* seed, float* out, int iters, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  for (int e = tid; e < kTriTile / 4; e += WAVES * 64) {
    unsigned x = seed[e & 255] * 2654435761u + e * 40503u;
    // bf16 pairs with moderate exponents
    reinterpret_cast<unsigned*>(smem_c)[e] = (x & 0x007F007Fu) | 0x3F003F00u | ((x >> 3) & 0x80008000u);
  }
  __syncthreads();
  Tri b0, b1;
  for (int w = 0; w < 4; ++w) {
    b0.h[w] = 0x3F803F80u + lane; b0.m[w] = 0x3B003B00u + w; b0.l[w] = 0x37003700u;
    b1.h[w] = 0x3F003F80u + lane; b1.m[w] = 0x3B803B00u + w; b1.l[w] = 0x37803700u;
  }
  f32x16 acc[4];
  for (int d = 0; d < 4; ++d) acc[d] = zero16();
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = tid + i;
  auto fetch = [&](int i) {
    const char* ap = smem_c + tri_tr_off(32 * (i & 3) + lo, 2 * (i >> 2) + h, 0);
    return Tri{*reinterpret_cast<const u32x4*>(ap), *reinterpret_cast<const u32x4*>(ap + 2048),
               *reinterpret_cast<const u32x4*>(ap + 4096)};
  };
  auto use = [&](int i, const Tri& a) {
    acc[i & 3] = mfma_tri(a, (i >> 2) ? b1 : b0, acc[i & 3]);
#pragma unroll
    for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[v & 7]) : "v"(0.999f));
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (DEPTH == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) use(i, fetch(i));
    } else if (DEPTH == 3) {
      tri_pipelined3<8>(fetch, use);
    } else {
      tri_pipelined<8>(fetch, use);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int d = 0; d < 4; ++d)
    for (int r = 0; r < 16; ++r) s += acc[d][r];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * WAVES * 64 + tid] = s;
  if (blockIdx.x == 0 && lane == 0) cyc[tid >> 6] = t1 - t0;

}

template <int WAVES, int DEPTH, int NV>
void run(const unsigned* seed, float* out, unsigned long long* cyc) {
  const int iters = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(bench<WAVES, DEPTH, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, kTriTile);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  bench<WAVES, DEPTH, NV><<<256, WAVES * 64, kTriTile>>>(seed, out, 100, cyc);
  hipEventRecord(e0);
  bench<WAVES, DEPTH, NV><<<256, WAVES * 64, kTriTile>>>(seed, out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long hc[8 + 512];
  hipMemcpy(hc, cyc, sizeof hc, hipMemcpyDeviceToHost);

  const double mfma_per_simd = (double)iters * 48 * (WAVES / 4);
  printf("waves/SIMD=%d prefetch=%d NV=%d: %.3f ms, %.1f ticks per MFMA per SIMD (wave 0), %.0f TFLOP/s fp32-equivalent, clock %.2f GHz\n",
         WAVES / 4, DEPTH, NV, ms, (double)hc[0] / mfma_per_simd, (double)iters * 48 * WAVES * 256 * 32768.0 / 6 / (ms * 1e-3) / 1e12,
         (double)hc[0] / (ms * 1e6));
}

int main() {
  unsigned* seed; float* out; unsigned long long* cyc;
  hipMalloc(&seed, 1024); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8 * (8 + 512));
  std::vector<unsigned> hs(256);
  for (int i = 0; i < 256; ++i) hs[i] = (unsigned)rand() * 2654435761u + i;
  hipMemcpy(seed, hs.data(), 1024, hipMemcpyHostToDevice);
  run<4, 0, 0>(seed, out, cyc); run<4, 2, 0>(seed, out, cyc); run<4, 3, 0>(seed, out, cyc);
  run<8, 0, 0>(seed, out, cyc); run<8, 2, 0>(seed, out, cyc); run<8, 3, 0>(seed, out, cyc);
  run<4, 2, 3>(seed, out, cyc); run<8, 2, 3>(seed, out, cyc);
  run<4, 2, 6>(seed, out, cyc); run<8, 2, 6>(seed, out, cyc);
  return 0;
}
