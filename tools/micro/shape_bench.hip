// Round 6, verdict item 2: would the sampled-row kernels gain from the 16x16x32 MFMA shape (2 048 row blocks of 16 rows:
// two waves per SIMD at half the accumulators) over 32x32x16 (1 024 row blocks, one wave per SIMD)?  Before rewriting
// tri_dev.h's image order and two 1 100-line kernels, the part of the question a loop can answer: with every A operand
// read from LDS the way the kernels fetch it (three 16-byte planes per k-step, 2 k-steps ahead), the B operand in
// registers, six split-bf16 products per k-step and NV vector instructions per k-step beside them --
//   shape 32:  v_mfma_f32_32x32x16_bf16, a step = 3 x ds_read_b128 + 6 MFMAs (32 cycles each) + NV fillers
//   shape 16:  v_mfma_f32_16x16x32_bf16, a step = 3 x ds_read_b128 + 6 MFMAs (16 cycles each) + NV / 2 fillers
// i.e. per FLOP the 16-shape reads twice the operand bytes from LDS (1 KB A per instruction of half the flops) and runs
// the same vector work.  1 or 2 waves per SIMD each.  Prints the fp32-equivalent rate (executed bf16 flops / 6) on RANDOM
// operands, the shader clock the s_memtime ticks imply, and ticks per MFMA per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../samble_amd/csrc shape_bench.hip -o shape_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "tri_dev.h"
using namespace samble;

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4v mfma16(u32x4 a, u32x4 b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int SHAPE, int WAVES, int NV>
__global__ __launch_bounds__(WAVES * 64) void bench(const unsigned* seed, float* out, int iters, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int e = tid; e < kTriTile / 4; e += WAVES * 64) {
    unsigned x = seed[e & 255] * 2654435761u + e * 40503u;
    reinterpret_cast<unsigned*>(smem_c)[e] = (x & 0x007F007Fu) | 0x3F003F00u | ((x >> 3) & 0x80008000u);  // random bf16 pairs
  }
  __syncthreads();
  Tri b0, b1;
  for (int w = 0; w < 4; ++w) {
    const unsigned r = seed[(lane * 4 + w) & 255];
    b0.h[w] = (r & 0x007F007Fu) | 0x3F803F80u;
    b0.m[w] = ((r >> 5) & 0x007F007Fu) | 0x3B003B00u;
    b0.l[w] = ((r >> 9) & 0x007F007Fu) | 0x37003700u;
    b1.h[w] = ((r >> 2) & 0x007F007Fu) | 0x3F003F80u;
    b1.m[w] = ((r >> 7) & 0x007F007Fu) | 0x3B803B00u;
    b1.l[w] = ((r >> 11) & 0x007F007Fu) | 0x37803700u;
  }
  f32x16 acc32[4];
  f32x4v acc16[8];
  for (int d = 0; d < 4; ++d) acc32[d] = zero16();
  for (int d = 0; d < 8; ++d) acc16[d] = f32x4v{0.f, 0.f, 0.f, 0.f};
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = tid + i;
  // eight k-steps per "tile"; the three planes of a step's A operand: conflict-free 16-byte reads, chunk-major
  auto fetch = [&](int i) {
    const char* ap = smem_c + ((i * 3) * 64 + lane) * 16;
    return Tri{*reinterpret_cast<const u32x4*>(ap), *reinterpret_cast<const u32x4*>(ap + 1024),
               *reinterpret_cast<const u32x4*>(ap + 2048)};
  };
  auto use = [&](int i, const Tri& a) {
    const Tri& b = (i >> 2) ? b1 : b0;
    if (SHAPE == 32) {
      acc32[i & 3] = mfma_tri(a, b, acc32[i & 3]);
    } else {
      f32x4v c = acc16[i & 7];
      c = mfma16(a.m, b.m, c);
      c = mfma16(a.h, b.l, c);
      c = mfma16(a.l, b.h, c);
      c = mfma16(a.h, b.m, c);
      c = mfma16(a.m, b.h, c);
      c = mfma16(a.h, b.h, c);
      acc16[i & 7] = c;
    }
    constexpr int nv = SHAPE == 32 ? NV : NV / 2;
#pragma unroll
    for (int v = 0; v < nv; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[v & 7]) : "v"(0.999f));
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) tri_pipelined<8>(fetch, use);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int d = 0; d < 4; ++d)
    for (int r = 0; r < 16; ++r) s += acc32[d][r];
  for (int d = 0; d < 8; ++d)
    for (int r = 0; r < 4; ++r) s += acc16[d][r];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * WAVES * 64 + tid] = s;
  if (blockIdx.x == 0 && lane == 0) cyc[tid >> 6] = t1 - t0;
}

template <int SHAPE, int WAVES, int NV>
void run(const unsigned* seed, float* out, unsigned long long* cyc) {
  const int iters = SHAPE == 32 ? 4000 : 8000;   // the same flops per wave
  hipFuncSetAttribute(reinterpret_cast<const void*>(bench<SHAPE, WAVES, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, kTriTile);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  bench<SHAPE, WAVES, NV><<<256, WAVES * 64, kTriTile>>>(seed, out, 100, cyc);
  hipEventRecord(e0);
  bench<SHAPE, WAVES, NV><<<256, WAVES * 64, kTriTile>>>(seed, out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long hc[16];
  hipMemcpy(hc, cyc, sizeof hc, hipMemcpyDeviceToHost);
  const double flops_per_mfma = SHAPE == 32 ? 32768.0 : 16384.0;
  const double mfma_per_simd = (double)iters * 48 * (WAVES / 4);
  printf("shape %dx%d waves/SIMD=%d NV=%d: %.3f ms, %.1f ticks per MFMA per SIMD, %.0f TFLOP/s fp32-equivalent, clock %.2f GHz, "
         "LDS operand reads per SIMD %.1f GB/s\n",
         SHAPE, SHAPE, WAVES / 4, NV, ms, (double)hc[0] / mfma_per_simd,
         (double)iters * 48 * WAVES * 256 * flops_per_mfma / 6 / (ms * 1e-3) / 1e12, (double)hc[0] / (ms * 1e6),
         (double)iters * 8 * 3 * 1024.0 * (WAVES / 4) / (ms * 1e-3) / 1e9);
}

int main() {
  unsigned* seed;
  float* out;
  unsigned long long* cyc;
  hipMalloc(&seed, 1024);
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 8 * 16);
  std::vector<unsigned> hs(256);
  for (int i = 0; i < 256; ++i) hs[i] = (unsigned)rand() * 2654435761u + i;
  hipMemcpy(seed, hs.data(), 1024, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<32, 4, 0>(seed, out, cyc);
    run<16, 4, 0>(seed, out, cyc);
    run<16, 8, 0>(seed, out, cyc);
    run<32, 4, 24>(seed, out, cyc);   // ~ the sampled-row kernels' vector load: 4 instructions per 32x32x16 MFMA
    run<32, 8, 24>(seed, out, cyc);
    run<16, 4, 24>(seed, out, cyc);
    run<16, 8, 24>(seed, out, cyc);
    run<32, 4, 48>(seed, out, cyc);
    run<16, 8, 48>(seed, out, cyc);
  }
  return 0;
}
