// nn.BatchNorm1d.forward in training mode on a channel-major (B, C, N) tensor -- the two BatchNorms of an attention layer
// (reference models/attention.py:187-192: x = bn1(x + attention(x)); x = bn2(x + ff(x))), forward only: the backward stays
// the aten entry the module dispatches to (miopen_batch_norm_backward), which takes the mean and 1 / sqrt(var + eps) this
// forward saves.
//
// Why: at (32, 128, 2048) the library's training forward is one workgroup per channel (128 workgroups for 256 CUs, each
// walking its channel's 32 rows twice): 44 us where the tensor's three passes (read, read, write: 100 MB) take ~20 at the
// memory rate.  Here:
//   bn_stats   grid (C, S): workgroup (c, s) sums x and x^2 over the clouds b = s, s + S, ... of channel c in float64
//              (float4 loads of whole rows) -> one partial pair; S is chosen so that C S ~ 1 024 workgroups
//   bn_apply   grid (C, B): workgroup (c, b) adds the channel's S partials in index order (every workgroup the same sums:
//              deterministic, no finalize launch), forms mean, invstd, scale = gamma invstd, shift = beta - mean scale in
//              float64 and writes y = x scale + shift for its row; the b = 0 workgroups also write the saved statistics
//              and the running estimates (momentum update with the unbiased variance, as torch does)
#include "samble_dev.h"

namespace samble {

constexpr int kBnThreads = 256;

__device__ __forceinline__ double bn_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(kBnThreads) void bn_stats_kernel(const float* __restrict__ x, int B, int C, int N,
                                                              double* __restrict__ part) {
  __shared__ double red[2][kBnThreads / 64];
  const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y, tid = threadIdx.x;
  double a0 = 0.0, a1 = 0.0;
  const bool vec = (N & 3) == 0 && ((uintptr_t)x & 15) == 0;
  for (int b = s; b < B; b += S) {
    const float* row = x + ((long)b * C + c) * N;
    if (vec) {
      const f32x4* r4 = reinterpret_cast<const f32x4*>(row);
      for (int i = tid; i < (N >> 2); i += kBnThreads) {
        const f32x4 v = r4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a0 += (double)v[e];
          a1 += (double)v[e] * (double)v[e];
        }
      }
    } else {
      for (int i = tid; i < N; i += kBnThreads) {
        const double v = (double)row[i];
        a0 += v;
        a1 += v * v;
      }
    }
  }
  a0 = bn_wave_sum(a0);
  a1 = bn_wave_sum(a1);
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = a0;
    red[1][tid >> 6] = a1;
  }
  __syncthreads();
  if (tid == 0) {
    double t0 = red[0][0], t1 = red[1][0];
#pragma unroll
    for (int w = 1; w < kBnThreads / 64; ++w) {
      t0 += red[0][w];
      t1 += red[1][w];
    }
    part[((long)c * S + s) * 2] = t0;
    part[((long)c * S + s) * 2 + 1] = t1;
  }
}

__global__ __launch_bounds__(kBnThreads) void bn_apply_kernel(const float* __restrict__ x, int B, int C, int N,
                                                              const double* __restrict__ part, int S,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float eps, float momentum, float* running_mean,
                                                              float* running_var, float* __restrict__ out,
                                                              float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  double t0 = 0.0, t1 = 0.0;
  for (int s = 0; s < S; ++s) {  // (uniform addresses: broadcast loads; the same order in every workgroup)
    t0 += part[((long)c * S + s) * 2];
    t1 += part[((long)c * S + s) * 2 + 1];
  }
  const double E = (double)B * (double)N;
  const double mean = t0 / E;
  double var = t1 / E - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const double g = gamma ? (double)gamma[c] : 1.0, bt = beta ? (double)beta[c] : 0.0;
  const float sc = (float)(g * invstd), sh = (float)(bt - mean * g * invstd);
  if (b == 0 && tid == 0) {
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)invstd;
    if (running_mean) running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
    if (running_var) {
      const double unbiased = E > 1.0 ? var * E / (E - 1.0) : var;
      running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
  }
  const float* row = x + ((long)b * C + c) * N;
  float* orow = out + ((long)b * C + c) * N;
  if ((N & 3) == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
    const f32x4* r4 = reinterpret_cast<const f32x4*>(row);
    f32x4* o4 = reinterpret_cast<f32x4*>(orow);
    for (int i = tid; i < (N >> 2); i += kBnThreads) {
      f32x4 v = r4[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc, sh);
      o4[i] = v;
    }
  } else {
    for (int i = tid; i < N; i += kBnThreads) orow[i] = fmaf(row[i], sc, sh);
  }
}

}  // namespace samble

using namespace samble;

// slices of the batch per channel in the statistics pass
static int bn_slices(int B, int C) {
  int S = 1024 / (C > 0 ? C : 1);
  S = S < 1 ? 1 : S;
  return S > B ? B : S;
}

extern "C" size_t samble_bn_train_ws_bytes(int B, int C) { return (size_t)C * bn_slices(B, C) * 2 * sizeof(double); }

extern "C" int samble_launch_bn_train_fwd(const float* x, int B, int C, int N, const float* gamma, const float* beta, float eps,
                                          float momentum, float* running_mean, float* running_var, float* out, float* save_mean,
                                          float* save_invstd, void* ws, hipStream_t s) {
  const int S = bn_slices(B, C);
  Timed timed(kT_bn_fwd, s);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(C, S), dim3(kBnThreads), 0, s, x, B, C, N, (double*)ws);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(C, B), dim3(kBnThreads), 0, s, x, B, C, N, (const double*)ws, S, gamma, beta, eps,
                     momentum, running_mean, running_var, out, save_mean, save_invstd);
  return (int)hipGetLastError();
}
