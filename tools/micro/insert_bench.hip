// Microbenchmark: cost of one sorted-list insertion (K = 32) per lane, three encodings.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
template <int KN> __device__ __forceinline__ void ins_f64(double (&L)[KN], double x) {
#pragma unroll
  for (int s = KN - 1; s > 0; --s) L[s] = fmin(L[s], fmax(L[s - 1], x));
  L[0] = fmin(L[0], x);
}
template <int KN> __device__ __forceinline__ void ins_f32(float (&Lw)[KN], int (&Lj)[KN], float x, int xj) {
  bool c_hi = x < Lw[KN - 1];  // c[s]
#pragma unroll
  for (int s = KN - 1; s > 0; --s) {
    const bool c_lo = x < Lw[s - 1];  // c[s-1]
    Lj[s] = c_lo ? Lj[s - 1] : (c_hi ? xj : Lj[s]);
    Lw[s] = __builtin_amdgcn_fmed3f(Lw[s], Lw[s - 1], x);
    c_hi = c_lo;
  }
  Lj[0] = c_hi ? xj : Lj[0];
  Lw[0] = fminf(Lw[0], x);
}
__global__ void k_f64(const float* in, double* out, int iters) {
  double L[32];
  for (int s = 0; s < 32; ++s) L[s] = 1e30;
  float v = in[threadIdx.x];
  for (int it = 0; it < iters; ++it) {
    v = v * 1.0001f + 0.37f; if (v > 1000.f) v -= 999.f;
    ins_f64<32>(L, __longlong_as_double(__double_as_longlong((double)v) | (long long)it));
  }
  double s = 0; for (int i = 0; i < 32; ++i) s += L[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_f32(const float* in, double* out, int iters) {
  float Lw[32]; int Lj[32];
  for (int s = 0; s < 32; ++s) { Lw[s] = 1e30f; Lj[s] = 0; }
  float v = in[threadIdx.x];
  for (int it = 0; it < iters; ++it) {
    v = v * 1.0001f + 0.37f; if (v > 1000.f) v -= 999.f;
    ins_f32<32>(Lw, Lj, v, it);
  }
  double s = 0; for (int i = 0; i < 32; ++i) s += Lw[i] + Lj[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* in; double* out; const int blocks = 1024, threads = 256, iters = 2000;
  hipMalloc(&in, threads * 4); hipMalloc(&out, blocks * threads * 8);
  hipMemset(in, 0, threads * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      if (which == 0) hipLaunchKernelGGL(k_f64, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
      else hipLaunchKernelGGL(k_f32, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      // waves per SIMD = blocks*4/1024 ; cycles per insertion per wave at 2.2 GHz
      double waves_per_simd = blocks * 4.0 / 1024.0;
      if (rep) printf("%s: %.3f ms  -> %.0f cycles per wave-insertion (at 2.2 GHz, %g waves/SIMD serialised)\n",
                      which ? "f32 med3+cndmask" : "f64 min/max     ", ms, ms * 1e-3 * 2.2e9 / (iters * waves_per_simd), waves_per_simd);
    }
  }
  return 0;
}
