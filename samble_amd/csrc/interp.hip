// Inverse-distance interpolation of coarse features onto a finer point set (reference models/upsample.py:181-213:
// select_neighbors_interpolate + the weighted sum over the K = 3 nearest coarse points), forward and backward, as two
// gather kernels instead of the (B, C, N, K) neighbour tensor and a dozen elementwise / reduce launches over it
// (profiles/r04_block_seg: ~290 us forward at N = 2048, of which 116 us one reduction over a 100 MB product).
//
//   forward   out[b][c][n] = sum_k w[b][n][k] feat[b][c][idx[b][n][k]],   w = (1 / (d + 1e-8)) / sum_k (1 / (d + 1e-8))
//   backward  dfeat[b][c][j] = sum over the (n, k) with idx[b][n][k] = j of w[b][n][k] g[b][c][n], in the order of the
//             inverse neighbour lists (ascending edge): deterministic, no atomics; on point-major rows (two tile
//             transposes around a row gather).  d (xyz distances) carries no gradient
//             on this path.
#include "samble_dev.h"

namespace samble {

constexpr int kIK = 8;  // most neighbours per point these kernels take

// workgroup = 64 consecutive points x 4 waves; wave g takes the channels c = g, g + 4, ... (consecutive lanes =
// consecutive points: coalesced stores; eight channels' gathers in flight per trip)
__global__ __launch_bounds__(256) void interp_fwd_kernel(const float* __restrict__ feat, int C, int M,
                                                         const int* __restrict__ idx, const float* __restrict__ dist,
                                                         int N, int K, float* __restrict__ w_out,
                                                         float* __restrict__ out) {
  int chunk, b;
  xcd_assign(chunk, b);  // a cloud's workgroups on one XCD: the coarse rows they gather are that cloud's
  const int n = chunk * 64 + (threadIdx.x & 63), cg = threadIdx.x >> 6;
  if (n >= N) return;
  int j[kIK];
  float w[kIK];
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < kIK; ++k) {
    j[k] = 0;
    w[k] = 0.f;
    if (k < K) {
      j[k] = idx[((long)b * N + n) * K + k];
      w[k] = 1.f / (dist[((long)b * N + n) * K + k] + 1e-8f);
      tot += w[k];
    }
  }
#pragma unroll
  for (int k = 0; k < kIK; ++k)
    if (k < K) {
      w[k] = w[k] / tot;
      if (cg == 0) w_out[((long)b * N + n) * K + k] = w[k];
    }
  const float* fb = feat + (long)b * C * M;
  float* ob = out + (long)b * C * N + n;
  for (int c0 = cg; c0 < C; c0 += 32) {
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = 0.f;
      const int c = c0 + 4 * u;
#pragma unroll
      for (int k = 0; k < kIK; ++k)
        if (k < K && c < C) a[u] = fmaf(w[k], fb[(long)c * M + j[k]], a[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (c0 + 4 * u < C) ob[(long)(c0 + 4 * u) * N] = a[u];
  }
}

// (B, R, S) -> (B, S, R) through a 64 x 64 LDS tile: channel-major <-> point-major rows
__global__ __launch_bounds__(256) void interp_transpose_kernel(const float* __restrict__ src, int R, int S,
                                                               float* __restrict__ dst) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, r0 = blockIdx.y * 64, s0 = blockIdx.x * 64, tid = threadIdx.x;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    tile[r][c] = (r0 + r < R && s0 + c < S) ? src[((long)b * R + r0 + r) * S + s0 + c] : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < 64 * 64; e += 256) {
    const int c = e >> 6, r = e & 63;
    if (s0 + c < S && r0 + r < R) dst[((long)b * S + s0 + c) * R + r0 + r] = tile[r][c];
  }
}

// backward on POINT-MAJOR rows (a channel-major gather costs a 64-byte sector per 4-byte value: 1.6 GB of L2 traffic
// for the 196 k edges of a step, 136 us): wave = one coarse point j, lane = channels lane, lane + 64, ...; its incoming
// edges e = (n, k) from the inverse lists (order / offsets of samble_inverse_neighbors over the (B, N, K) table, targets
// b * N + j), four 512-byte rows in flight, summed in list order
__global__ __launch_bounds__(256) void interp_bwd_rows_kernel(const float* __restrict__ gt, int C, int N,
                                                              const float* __restrict__ w, const int* __restrict__ order,
                                                              const int* __restrict__ offsets, int K, int M,
                                                              float* __restrict__ dft) {
  int chunk, b;
  xcd_assign(chunk, b);
  const int j = chunk * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= M) return;
  const long t = (long)b * N + j;
  const int e0 = offsets[t], e1 = offsets[t + 1];
  for (int c = lane; c < C; c += 64) {
    float a = 0.f;
    int q = e0;
    for (; q + 4 <= e1; q += 4) {
      int e[4];
      float we[4], v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) e[u] = order[q + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        we[u] = w[e[u]];
        v[u] = gt[(long)(e[u] / K) * C + c];   // (e / K = b * N + n: the row of the point-major gradient)
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) a = fmaf(we[u], v[u], a);
    }
    for (; q < e1; ++q) {
      const int e = order[q];
      a = fmaf(w[e], gt[(long)(e / K) * C + c], a);
    }
    dft[((long)b * M + j) * C + c] = a;
  }
}

}  // namespace samble

using namespace samble;

extern "C" int samble_launch_interp_fwd(const float* feat, int B, int C, int M, const int* idx, const float* dist, int N,
                                        int K, float* w, float* out, hipStream_t s) {
  hipLaunchKernelGGL(interp_fwd_kernel, dim3((N + 63) / 64, B), dim3(256), 0, s, feat, C, M, idx, dist, N, K, w, out);
  return (int)hipGetLastError();
}

extern "C" size_t samble_interp_bwd_ws_bytes(int B, int C, int N, int M) { return (size_t)B * C * ((size_t)N + M) * sizeof(float); }

extern "C" int samble_launch_interp_bwd(const float* g, int B, int C, int N, const float* w, const int* order,
                                        const int* offsets, int K, int M, float* dfeat, void* ws, hipStream_t s) {
  float* gt = (float*)ws;                   // (B, N, C)
  float* dft = gt + (size_t)B * N * C;      // (B, M, C)
  hipLaunchKernelGGL(interp_transpose_kernel, dim3((N + 63) / 64, (C + 63) / 64, B), dim3(256), 0, s, g, C, N, gt);
  hipLaunchKernelGGL(interp_bwd_rows_kernel, dim3((M + 3) / 4, B), dim3(256), 0, s, gt, C, N, w, order, offsets, K, M, dft);
  hipLaunchKernelGGL(interp_transpose_kernel, dim3((C + 63) / 64, (M + 63) / 64, B), dim3(256), 0, s, dft, M, C, dfeat);
  return (int)hipGetLastError();
}
