// Accuracy and rate of fp32-equivalent products built from 16-bit MFMAs on gfx950.
//   method 0: v_mfma_f32_32x32x2_f32 (the fp32 matrix instruction the round-1 kernels use)
//   method 1: bf16 x 6   a = h + m + l (three bf16), products hh, hm, mh, hl, lh, mm
//   method 2: bf16 x 3   hh, hm, mh
//   method 3: fp16 x 3   a = h + l (two fp16), products hh, hl, lh
//   method 4: fp16 x 3 with l carried as l * 2^11 in its own accumulator (no fp16 denormals)
// Accuracy: one wave computes C(32x32) = A(32x128) B(32x128)^T per method; the host compares with fp64.
// Rate: register-resident loops of the MFMA mix of each method (+ NV filler v_fma_f32 per MFMA).
// Build: hipcc --offload-arch=gfx950 -O3 split_mfma_bench.hip -o split_mfma_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split_bf16(float a, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)a;
  const float r1 = a - (float)h;
  m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  l = (__bf16)r2;
}

template <int METHOD>
__global__ __launch_bounds__(64) void tile_kernel(const float* A, const float* B, float* C) {
  const int lane = threadIdx.x, rc = lane & 31, hl = lane >> 5;
  f32x16 acc = {0}, acc2 = {0};
  if (METHOD == 0) {
    for (int k = 0; k < 128; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rc * 128 + k + hl], B[rc * 128 + k + hl], acc, 0, 0, 0);
  } else if (METHOD == 1 || METHOD == 2) {
    for (int k = 0; k < 128; k += 16) {
      bf16x8 ah, am, al, bh, bm, bl;
      for (int i = 0; i < 8; ++i) {
        __bf16 h, m, l;
        split_bf16(A[rc * 128 + k + 8 * hl + i], h, m, l);
        ah[i] = h; am[i] = m; al[i] = l;
        split_bf16(B[rc * 128 + k + 8 * hl + i], h, m, l);
        bh[i] = h; bm[i] = m; bl[i] = l;
      }
      if (METHOD == 1) {  // small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
  } else {
    const float sc = (METHOD == 4) ? 2048.f : 1.f;
    for (int k = 0; k < 128; k += 16) {
      f16x8 ah, al, bh, bl;
      for (int i = 0; i < 8; ++i) {
        float a = A[rc * 128 + k + 8 * hl + i], b = B[rc * 128 + k + 8 * hl + i];
        ah[i] = (_Float16)a; al[i] = (_Float16)((a - (float)ah[i]) * sc);
        bh[i] = (_Float16)b; bl[i] = (_Float16)((b - (float)bh[i]) * sc);
      }
      if (METHOD == 4) {
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    }
    if (METHOD == 4)
      for (int r = 0; r < 16; ++r) acc[r] += acc2[r] * (1.f / 2048.f);
  }
  // D layout: col = lane & 31 (B row), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (A row)
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * hl) * 32 + rc] = acc[r];
}

template <int NMFMA, bool F16, int NV, int NACC = 4>
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters) {
  f32x16 acc[4] = {{0}, {0}, {0}, {0}};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(1.f + i * 0.01f); }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x + i;
  const float c = 0.999f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NMFMA; ++m) {
      if (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m % NACC]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m % NACC]) : "v"(a), "v"(b));
#pragma unroll
      for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[v & 7]) : "v"(c));
    }
  }
  float s = 0;
  for (int m = 0; m < 4; ++m)
    for (int i = 0; i < 16; ++i) s += acc[m][i];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Same loop on RANDOM operands (the chip lowers its clock under bf16 MFMA load on random data), for the two
// bf16 shapes: 6 MFMAs of 32x32x16 (32 cycles each) vs 12 of 16x16x32 (16 cycles each) per iteration.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool SMALL>
__global__ __launch_bounds__(512) void rate_random_kernel(const unsigned* seed, float* out, int iters) {
  f32x16 acc[4] = {{0}, {0}, {0}, {0}};
  f32x4v acs[4] = {{0}, {0}, {0}, {0}};
  bf16x8 a[4], b[4];
  unsigned h = seed[threadIdx.x & 255] ^ (blockIdx.x * 2654435761u);
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u;
      a[k][i] = (__bf16)(((int)(h >> 8) & 0xffff) * (1.f / 32768.f) - 1.f);
      h = h * 1664525u + 1013904223u;
      b[k][i] = (__bf16)(((int)(h >> 8) & 0xffff) * (1.f / 32768.f) - 1.f);
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < (SMALL ? 12 : 6); ++m) {
      if (SMALL) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acs[m & 3]) : "v"(a[m & 3]), "v"(b[(m + 1) & 3]));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a[m & 3]), "v"(b[(m + 1) & 3]));
    }
  }
  float s = 0;
  for (int m = 0; m < 4; ++m) {
    for (int i = 0; i < 16; ++i) s += acc[m][i];
    for (int i = 0; i < 4; ++i) s += acs[m][i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool SMALL>
void run_rate_random(const unsigned* seed, int waves_per_simd, float* out) {
  const int iters = 200000;  // long enough for the clock governor to settle
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256), block(256 * waves_per_simd);
  rate_random_kernel<SMALL><<<grid, block>>>(seed, out, 1000);
  hipEventRecord(e0);
  rate_random_kernel<SMALL><<<grid, block>>>(seed, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)iters * 6 * 32768.0 * 256 * 4 * waves_per_simd;  // both shapes: 6 x 32768 flop per iteration
  const double cyc = (double)iters * 6 * 32;                                     // matrix-pipe cycles per wave
  printf("random data, %s, waves/SIMD=%d: %.2f ms  %.0f TFLOP/s executed (%.0f fp32-equivalent at 6 products)  pipe clock >= %.2f GHz\n",
         SMALL ? "16x16x32" : "32x32x16", waves_per_simd, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 6,
         cyc * waves_per_simd / (ms * 1e-3) / 1e9);
}

template <bool F16, int NV, int NACC = 4>
void run_rate(const char* name, int waves_per_simd, float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256), block(256 * waves_per_simd);
  rate_kernel<6, F16, NV, NACC><<<grid, block>>>(out, 100);
  hipEventRecord(e0);
  rate_kernel<6, F16, NV, NACC><<<grid, block>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)iters * 6 * 256 * 4 * waves_per_simd;
  const double tf = mfmas * 32768.0 / (ms * 1e-3) / 1e12;
  printf("%-6s chains=%d NV=%2d waves/SIMD=%d  %.3f ms  %.0f TFLOP/s hardware  (%.0f fp32-equivalent at 6 products, %.0f at 3)\n", name,
         NACC, NV, waves_per_simd, ms, tf, tf / 6, tf / 3);
}

template <int METHOD>
void run_acc(const char* name, const std::vector<float>& hA, const std::vector<float>& hB, float* dA, float* dB, float* dC,
             int trials) {
  double se = 0, sr = 0, mx = 0, sab = 0;
  for (int t = 0; t < trials; ++t) {
    const float* A = hA.data() + (size_t)t * 4096;
    const float* B = hB.data() + (size_t)t * 4096;
    hipMemcpy(dA, A, 4096 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B, 4096 * 4, hipMemcpyHostToDevice);
    tile_kernel<METHOD><<<1, 64>>>(dA, dB, dC);
    float hC[1024];
    hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double ref = 0, ab = 0;
        for (int k = 0; k < 128; ++k) {
          ref += (double)A[i * 128 + k] * B[j * 128 + k];
          ab += std::fabs((double)A[i * 128 + k] * B[j * 128 + k]);
        }
        const double e = hC[i * 32 + j] - ref;
        se += e * e; sr += ref * ref; sab += ab * ab;
        mx = std::fmax(mx, std::fabs(e) / ab);
      }
  }
  printf("%-22s rms err / rms C = %.3e   rms err / rms sum|ab| = %.3e   max err / sum|ab| = %.3e\n", name, std::sqrt(se / sr),
         std::sqrt(se / sab), mx);
}

int main() {
  const int trials = 16;
  std::vector<float> hA((size_t)trials * 4096), hB((size_t)trials * 4096);
  srand(1);
  auto gauss = []() {
    double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
    return std::sqrt(-2 * std::log(u)) * std::cos(6.283185307179586 * v);
  };
  for (int t = 0; t < trials; ++t) {
    // trial scale sweeps 1e-6 .. 1e4 so range problems (fp16) show up
    const double sa = std::pow(10.0, -6 + (t % 6) * 2), sb = std::pow(10.0, (t % 3) - 1);
    for (int i = 0; i < 4096; ++i) { hA[(size_t)t * 4096 + i] = (float)(gauss() * sa); hB[(size_t)t * 4096 + i] = (float)(gauss() * sb); }
  }
  float *dA, *dB, *dC, *out;
  hipMalloc(&dA, 4096 * 4); hipMalloc(&dB, 4096 * 4); hipMalloc(&dC, 1024 * 4); hipMalloc(&out, 256 * 1024 * 4);
  printf("all 16 trials (row scales 1e-6..1e4):\n");
  run_acc<0>("fp32 MFMA", hA, hB, dA, dB, dC, trials);
  run_acc<1>("bf16 x 6", hA, hB, dA, dB, dC, trials);
  run_acc<2>("bf16 x 3", hA, hB, dA, dB, dC, trials);
  run_acc<3>("fp16 x 3", hA, hB, dA, dB, dC, trials);
  run_acc<4>("fp16 x 3 scaled lo", hA, hB, dA, dB, dC, trials);
  // unit-scale only (trial 3: sa = 1, sb = 0.1 ... pick t with sa == 1: t % 6 == 3)
  std::vector<float> uA, uB;
  for (int t = 0; t < trials; ++t)
    if (t % 6 == 3) { uA.insert(uA.end(), hA.begin() + (size_t)t * 4096, hA.begin() + (size_t)(t + 1) * 4096); uB.insert(uB.end(), hB.begin() + (size_t)t * 4096, hB.begin() + (size_t)(t + 1) * 4096); }
  printf("unit-scale trials only:\n");
  run_acc<0>("fp32 MFMA", uA, uB, dA, dB, dC, (int)(uA.size() / 4096));
  run_acc<1>("bf16 x 6", uA, uB, dA, dB, dC, (int)(uA.size() / 4096));
  run_acc<3>("fp16 x 3", uA, uB, dA, dB, dC, (int)(uA.size() / 4096));
  run_acc<4>("fp16 x 3 scaled lo", uA, uB, dA, dB, dC, (int)(uA.size() / 4096));
  for (int w = 1; w <= 2; ++w) {
    run_rate<false, 0>("bf16", w, out);
    run_rate<false, 4>("bf16", w, out);
    run_rate<false, 6>("bf16", w, out);
    run_rate<false, 8>("bf16", w, out);
    run_rate<false, 12>("bf16", w, out);
    run_rate<false, 0, 1>("bf16", w, out);
    run_rate<false, 4, 1>("bf16", w, out);
    run_rate<false, 8, 1>("bf16", w, out);
    run_rate<false, 0, 2>("bf16", w, out);
    run_rate<false, 4, 2>("bf16", w, out);
    run_rate<false, 8, 2>("bf16", w, out);
    run_rate<true, 0>("f16", w, out);
    run_rate<true, 6>("f16", w, out);
  }
  unsigned* seed;
  hipMalloc(&seed, 256 * 4);
  std::vector<unsigned> hs(256);
  for (int i = 0; i < 256; ++i) hs[i] = (unsigned)rand() * 2654435761u + (unsigned)i;
  hipMemcpy(seed, hs.data(), 256 * 4, hipMemcpyHostToDevice);
  for (int w = 1; w <= 2; ++w) {
    run_rate_random<false>(seed, w, out);
    run_rate_random<true>(seed, w, out);
  }
  return 0;
}
