// Pairwise distance + k-nearest-neighbour build (reference utils/ops.py:17-44).
//
// The reference centres both sets on the query set's mean, divides by one positive scalar per
// cloud and takes topk of -cdist.  Centring and an isotropic positive scale do not change the
// ranking, so neighbours are ranked on  key(i,j) = |b_j|^2 - 2 <a_i, b_j>  (= d^2 - |a_i|^2) of the
// raw channel-major inputs; the reference-normalised distance is rebuilt only for the K winners.
//
//   rownorm       |x_n|^2 per point (also per-cloud channel sums for the reference's scale)
//   gram_keys     key matrix, fp32 MFMA: workgroup = 128 keys x 128 queries, wave = 64 x 64,
//                 channels staged through LDS 32 at a time straight from the (B,C,N) layout
//                 (rows of 128 consecutive points = 512 contiguous bytes).  Written key-major so
//                 the select kernel reads it coalesced.
//   select_rows   one lane per query streams its column of keys: candidates below the lane's
//                 current K-th best go to a per-lane LDS queue; when a lane's queue fills the wave
//                 drains queues into per-lane sorted K-lists held in registers.
// Bound: gram_keys fp32 MFMA; select_rows VALU.  (Round-1 structure: the key matrix round-trips
// through HBM; DESIGN.md lists fusing the select into the MFMA tile loop as the next step.)
#include "samble_dev.h"

namespace samble {

__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ x, long bs, int C, int N,
                                                      float* __restrict__ norms) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float* p = x + (long)b * bs + n;
  float acc = 0.f;
  int c = 0;
  for (; c + 8 <= C; c += 8) {  // 8 independent loads in flight; the sum keeps channel order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(long)(c + u) * N];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = fmaf(v[u], v[u], acc);
  }
  for (; c < C; ++c) {
    float v = p[(long)c * N];
    acc = fmaf(v, v, acc);
  }
  norms[(long)b * N + n] = acc;
}

// per-cloud channel means of a channel-major (B, C, N) set (the reference centres both sets on the QUERY
// set's mean, utils/ops.py:23-25): one wave per (cloud, channel), fixed order
__global__ __launch_bounds__(256) void cloud_mean_kernel(const float* __restrict__ x, long bs, int C, int N,
                                                         float* __restrict__ mean) {
  const int b = blockIdx.y, c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  const float* p = x + (long)b * bs + (long)c * N;
  float s = 0.f;
  int n = 0;
  if ((N & 3) == 0 && (bs & 3) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0) {
    // 16-byte loads, 8 in flight per lane (a row of 2048 floats is one trip); a lane sums its values in index order
    for (; n + 8 * 256 <= N; n += 8 * 256) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + n + 256 * u + 4 * lane);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
    }
    for (; n + 256 <= N; n += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + n + 4 * lane);
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
  }
  for (n += lane; n < N; n += 64) s += p[n];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) mean[b * C + c] = s / (float)N;
}

// centred copy xc[b][c][n] = x[b][c][n] - mean[b][c] (contiguous (B, C, N)): the Gram form of the squared
// distance, |a|^2 + |b|^2 - 2 a.b, cancels catastrophically when the cloud sits far from the origin
__global__ __launch_bounds__(256) void center_cm_kernel(const float* __restrict__ x, long bs, int C, int N,
                                                        const float* __restrict__ mean, float* __restrict__ xc) {
  const int b = blockIdx.z, c = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
  if (n < N) xc[((long)b * C + c) * N + n] = x[(long)b * bs + (long)c * N + n] - mean[b * C + c];
}

// keyT[b][j][i] = bnorm[j] - 2 * sum_c xk[b][c][j] * xq[b][c][i]
__global__ __launch_bounds__(256, 2) void gram_keys_kernel(const float* __restrict__ xq, long q_bs, int Nq,
                                                           const float* __restrict__ xk, long k_bs, int Nk, int C,
                                                           const float* __restrict__ knorm,
                                                           float* __restrict__ keyT) {
  __shared__ __attribute__((aligned(16))) float As[32 * 128];  // [channel][key]
  __shared__ __attribute__((aligned(16))) float Bs[32 * 128];  // [channel][query]
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  const int b = blockIdx.z;
  const int j0 = blockIdx.y * 128, i0 = blockIdx.x * 128;
  const int wj = (wave >> 1) * 64, wi = (wave & 1) * 64;
  const float* xkb = xk + (long)b * k_bs;
  const float* xqb = xq + (long)b * q_bs;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[a][c] = zero16();

  const bool k_vec = ((Nk & 3) == 0) && (j0 + 128 <= Nk);
  const bool q_vec = ((Nq & 3) == 0) && (i0 + 128 <= Nq);
  for (int c0 = 0; c0 < C; c0 += 32) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int e = tid + 256 * it;  // float4 slot: 32 channels x 32 float4
      const int c = e >> 5, p4 = (e & 31) * 4;
      f32x4 av = {0.f, 0.f, 0.f, 0.f}, bv = {0.f, 0.f, 0.f, 0.f};
      if (c0 + c < C) {
        const float* ar = xkb + (long)(c0 + c) * Nk + j0 + p4;
        const float* br = xqb + (long)(c0 + c) * Nq + i0 + p4;
        if (k_vec) {
          av = *reinterpret_cast<const f32x4*>(ar);
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (j0 + p4 + u < Nk) av[u] = ar[u];
        }
        if (q_vec) {
          bv = *reinterpret_cast<const f32x4*>(br);
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (i0 + p4 + u < Nq) bv[u] = br[u];
        }
      }
      *reinterpret_cast<f32x4*>(&As[c * 128 + p4]) = av;
      *reinterpret_cast<f32x4*>(&Bs[c * 128 + p4]) = bv;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int c = 2 * kk + h;
      const float a0 = As[c * 128 + wj + lo], a1 = As[c * 128 + wj + 32 + lo];
      const float b0 = Bs[c * 128 + wi + lo], b1 = Bs[c * 128 + wi + 32 + lo];
      acc[0][0] = mfma32(a0, b0, acc[0][0]);
      acc[0][1] = mfma32(a0, b1, acc[0][1]);
      acc[1][0] = mfma32(a1, b0, acc[1][0]);
      acc[1][1] = mfma32(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
  float* out = keyT + (long)b * Nk * Nq;
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + wj + 32 * jt + crow(r, h);
      if (j >= Nk) continue;
      const float bn = knorm[(long)b * Nk + j];
#pragma unroll
      for (int itq = 0; itq < 2; ++itq) {
        const int i = i0 + wi + 32 * itq + lo;
        if (i < Nq) out[(long)j * Nq + i] = fmaf(-2.f, acc[jt][itq][r], bn);
      }
    }
  }
}

// Exact (a-b)^2 keys for tiny channel counts (xyz: C = 3): no cancellation, no MFMA.
__global__ __launch_bounds__(256) void smallc_keys_kernel(const float* __restrict__ xq, long q_bs, int Nq,
                                                          const float* __restrict__ xk, long k_bs, int Nk, int C,
                                                          float* __restrict__ keyT) {
  const int b = blockIdx.z;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int j = blockIdx.y;
  if (i >= Nq) return;
  float acc = 0.f;
  for (int c = 0; c < C; ++c) {
    const float d = xq[(long)b * q_bs + (long)c * Nq + i] - xk[(long)b * k_bs + (long)c * Nk + j];
    acc = fmaf(d, d, acc);
  }
  keyT[((long)b * Nk + j) * Nq + i] = acc;
}

constexpr int kQueue = 16;

template <int KN>
__device__ __forceinline__ void sorted_insert(float (&bk)[KN], int (&bi)[KN], float v, int j) {
#pragma unroll
  for (int s = KN - 1; s > 0; --s) {
    const bool shift = v < bk[s - 1];
    const bool here = (!shift) && (v < bk[s]);
    bk[s] = shift ? bk[s - 1] : (here ? v : bk[s]);
    bi[s] = shift ? bi[s - 1] : (here ? j : bi[s]);
  }
  if (v < bk[0]) {
    bk[0] = v;
    bi[0] = j;
  }
}

// One lane per query.  idx_out (B,Nq,KN) nearest first; key_out optional (B,Nq,KN) raw keys.
template <int KN>
__global__ __launch_bounds__(256) void select_rows_kernel(const float* __restrict__ keyT, int Nq, int Nk,
                                                          int* __restrict__ idx_out, float* __restrict__ key_out) {
  __shared__ float qk[kQueue * 256];
  __shared__ int qi[kQueue * 256];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + tid;
  const bool valid = i < Nq;
  const float* col = keyT + (long)b * Nk * Nq + (valid ? i : 0);

  float bk[KN];
  int bi[KN];
#pragma unroll
  for (int s = 0; s < KN; ++s) {
    bk[s] = __builtin_huge_valf();
    bi[s] = 0;
  }
  int cnt = 0;
  auto drain = [&]() {
    for (int s = 0; s < kQueue; ++s) {
      if (!__any(s < cnt)) break;
      const float v = (s < cnt) ? qk[s * 256 + tid] : __builtin_huge_valf();
      const int j = qi[s * 256 + tid];
      sorted_insert<KN>(bk, bi, v, j);
    }
    cnt = 0;
  };
  for (int j = 0; j < Nk; ++j) {
    const float v = valid ? col[(long)j * Nq] : __builtin_huge_valf();
    if (v < bk[KN - 1]) {
      qk[cnt * 256 + tid] = v;
      qi[cnt * 256 + tid] = j;
      ++cnt;
    }
    if (__any(cnt == kQueue)) drain();
  }
  drain();
  if (valid) {
    int* io = idx_out + ((long)b * Nq + i) * KN;
#pragma unroll
    for (int s = 0; s < KN; ++s) io[s] = bi[s];
    if (key_out) {
      float* ko = key_out + ((long)b * Nq + i) * KN;
#pragma unroll
      for (int s = 0; s < KN; ++s) ko[s] = bk[s];
    }
  }
}

// packed (w bits | index) doubles: a sorted K-list step is one v_max_f64 + v_min_f64 per slot, ties by index
template <int KN>
__device__ __forceinline__ void insert_packed(double (&L)[KN], double x) {
#pragma unroll
  for (int s = KN - 1; s > 0; --s) L[s] = fmin(L[s], fmax(L[s - 1], x));
  L[0] = fmin(L[0], x);
}

__device__ __forceinline__ double pack_wj(float w, unsigned int j) {
  return __longlong_as_double(__double_as_longlong((double)w) | (long long)j);
}

// ------------------------------------------------------------------------------------------------
// Fused exact-distance + top-K for tiny channel counts (xyz, C <= 8): EdgeConv's first layer and the
// interpolation upsampling.  Lane = one query with its coordinates in registers; the key set streams
// through LDS in chunks (every lane reads the same key: LDS broadcast); d^2 = sum_c (a_c - b_c)^2 is
// accumulated exactly like smallc_keys_kernel; candidates below the lane's bound go to a per-lane LDS
// queue and are inserted into the packed-double sorted list (w bits | index: ties by ascending index)
// when any lane's queue is full.  No (B, Nk, Nq) key matrix (537 MB at B = 32, N = 2048).
// ------------------------------------------------------------------------------------------------
#ifndef SAMBLE_SMALLC_SPLIT
#define SAMBLE_SMALLC_SPLIT 1  // K <= 8: the keys of a chunk dealt to the four waves of a workgroup (0: A/B builds)
#endif
constexpr int kSmallChunk = 512;  // keys per LDS chunk
constexpr int kSmallQueue = 16;

template <int KN>
__global__ __launch_bounds__(256) void knn_smallc_fused_kernel(const float* __restrict__ xq, long q_bs, int Nq,
                                                               const float* __restrict__ xk, long k_bs, int Nk, int C,
                                                               int* __restrict__ idx_out, float* __restrict__ key_out) {
  __shared__ __attribute__((aligned(16))) float kx[8 * kSmallChunk];
  __shared__ float qw[kSmallQueue * 256];
  __shared__ unsigned short qj[kSmallQueue * 256];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int i = blockIdx.x * 256 + tid;
  const bool valid = i < Nq;
  float q[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) q[c] = (c < C && valid) ? xq[(long)b * q_bs + (long)c * Nq + i] : 0.f;
  double L[KN];
#pragma unroll
  for (int s = 0; s < KN; ++s) L[s] = __builtin_huge_val();
  float thr = __builtin_huge_valf();
  int cnt = 0;
  auto drain = [&]() {
    for (int s = 0; s < kSmallQueue; ++s) {
      if (!__any(s < cnt)) break;
      const double xd = (s < cnt) ? pack_wj(qw[s * 256 + tid], qj[s * 256 + tid]) : __builtin_huge_val();
      insert_packed<KN>(L, xd);
    }
    cnt = 0;
    thr = (float)L[KN - 1];  // index bits are far below half a float ulp: exactly the K-th w (or +inf)
  };
  for (int j0 = 0; j0 < Nk; j0 += kSmallChunk) {
    const int nj = min(kSmallChunk, Nk - j0);
    __syncthreads();  // previous chunk fully consumed
    for (int c = 0; c < C; ++c)
      for (int e = tid; e < nj; e += 256) kx[c * kSmallChunk + e] = xk[(long)b * k_bs + (long)c * Nk + j0 + e];
    __syncthreads();
    // eight keys per trip: their distances first (broadcast 16-byte LDS reads, no branch), then ONE vote on the
    // smallest of them -- once the lists hold K entries almost every group of eight fails it as a whole and the per-key
    // path below (same arithmetic, keys in ascending order: same lists) is skipped.  (One key per trip with a branch and
    // a vote each took 187 us for the interpolation's two searches: a dependent LDS round trip per key.)
    const int nj8 = nj & ~7;
    for (int jj = 0; jj < nj8; jj += 8) {
      float acc8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc8[u] = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c < C) {
          const f32x4 k0 = *reinterpret_cast<const f32x4*>(kx + c * kSmallChunk + jj);
          const f32x4 k1 = *reinterpret_cast<const f32x4*>(kx + c * kSmallChunk + jj + 4);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float d0 = q[c] - k0[u], d1 = q[c] - k1[u];
            acc8[u] = fmaf(d0, d0, acc8[u]);
            acc8[4 + u] = fmaf(d1, d1, acc8[4 + u]);
          }
        }
      }
      const float m8 = fminf(fminf(fminf(acc8[0], acc8[1]), fminf(acc8[2], acc8[3])),
                             fminf(fminf(acc8[4], acc8[5]), fminf(acc8[6], acc8[7])));
      if (!__any(valid && m8 <= thr)) continue;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (valid && acc8[u] <= thr) {
          qw[cnt * 256 + tid] = acc8[u];
          qj[cnt * 256 + tid] = (unsigned short)(j0 + jj + u);
          ++cnt;
        }
        if (__any(cnt == kSmallQueue)) drain();
      }
    }
    for (int jj = nj8; jj < nj; ++jj) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c < C) {
          const float d = q[c] - kx[c * kSmallChunk + jj];
          acc = fmaf(d, d, acc);
        }
      }
      if (valid && acc <= thr) {
        qw[cnt * 256 + tid] = acc;
        qj[cnt * 256 + tid] = (unsigned short)(j0 + jj);
        ++cnt;
      }
      if (__any(cnt == kSmallQueue)) drain();
    }
  }
  drain();
  if (valid) {
    int* io = idx_out + ((long)b * Nq + i) * KN;
#pragma unroll
    for (int s = 0; s < KN; ++s) io[s] = (int)(__double_as_longlong(L[s]) & 0x1FFFFFFFll);
    if (key_out) {
      float* ko = key_out + ((long)b * Nq + i) * KN;
#pragma unroll
      for (int s = 0; s < KN; ++s) ko[s] = (float)L[s];
    }
  }
}

// The same search with the KEYS of every chunk dealt to the four waves of a workgroup (K <= 8: the interpolation's
// K = 3).  A lane per query gives a batch of 32 x 2048 queries 1 024 waves -- one per SIMD, each walking all keys behind
// its own LDS round trips (89 us for a search whose arithmetic is microseconds).  Here a workgroup is 64 queries x 4
// waves, wave w scans keys 128 w .. 128 w + 127 of each 512-key chunk into its own sorted list, and wave 0 merges the
// four lists of a query (packed (w, index) doubles: the same K entries in the same order as one list over all keys).
template <int KN>
__global__ __launch_bounds__(256) void knn_smallc_split_kernel(const float* __restrict__ xq, long q_bs, int Nq,
                                                               const float* __restrict__ xk, long k_bs, int Nk, int C,
                                                               int* __restrict__ idx_out, float* __restrict__ key_out) {
  constexpr int Q = 8;  // queue slots per thread
  __shared__ __attribute__((aligned(16))) float kx[8 * kSmallChunk];  // later: the four lists of every query (doubles)
  __shared__ float qw[Q * 256];
  __shared__ unsigned short qj[Q * 256];
  static_assert(4 * 64 * KN * 8 <= 8 * kSmallChunk * 4, "merge buffer must fit the key chunk");
  const int tid = threadIdx.x, b = blockIdx.y, wave = tid >> 6, lane = tid & 63;
  const int i = blockIdx.x * 64 + lane;
  const bool valid = i < Nq;
  float q[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) q[c] = (c < C && valid) ? xq[(long)b * q_bs + (long)c * Nq + i] : 0.f;
  double L[KN];
#pragma unroll
  for (int s = 0; s < KN; ++s) L[s] = __builtin_huge_val();
  float thr = __builtin_huge_valf();
  int cnt = 0;
  auto drain = [&]() {
    for (int s = 0; s < Q; ++s) {
      if (!__any(s < cnt)) break;
      const double xd = (s < cnt) ? pack_wj(qw[s * 256 + tid], qj[s * 256 + tid]) : __builtin_huge_val();
      insert_packed<KN>(L, xd);
    }
    cnt = 0;
    thr = (float)L[KN - 1];
  };
  for (int j0 = 0; j0 < Nk; j0 += kSmallChunk) {
    const int nj = min(kSmallChunk, Nk - j0);
    __syncthreads();  // previous chunk fully consumed
    for (int c = 0; c < C; ++c)
      for (int e = tid; e < nj; e += 256) kx[c * kSmallChunk + e] = xk[(long)b * k_bs + (long)c * Nk + j0 + e];
    __syncthreads();
    const int w0 = min(wave * (kSmallChunk / 4), nj), w1 = min(w0 + kSmallChunk / 4, nj);  // this wave's keys of the chunk
    const int n8 = w0 + ((w1 - w0) & ~7);
    for (int jj = w0; jj < n8; jj += 8) {
      float acc8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc8[u] = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c < C) {
          const f32x4 k0 = *reinterpret_cast<const f32x4*>(kx + c * kSmallChunk + jj);
          const f32x4 k1 = *reinterpret_cast<const f32x4*>(kx + c * kSmallChunk + jj + 4);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float d0 = q[c] - k0[u], d1 = q[c] - k1[u];
            acc8[u] = fmaf(d0, d0, acc8[u]);
            acc8[4 + u] = fmaf(d1, d1, acc8[4 + u]);
          }
        }
      }
      const float m8 = fminf(fminf(fminf(acc8[0], acc8[1]), fminf(acc8[2], acc8[3])),
                             fminf(fminf(acc8[4], acc8[5]), fminf(acc8[6], acc8[7])));
      if (!__any(valid && m8 <= thr)) continue;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (valid && acc8[u] <= thr) {
          qw[cnt * 256 + tid] = acc8[u];
          qj[cnt * 256 + tid] = (unsigned short)(j0 + jj + u);
          ++cnt;
        }
        if (__any(cnt == Q)) drain();
      }
    }
    for (int jj = n8; jj < w1; ++jj) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c < C) {
          const float d = q[c] - kx[c * kSmallChunk + jj];
          acc = fmaf(d, d, acc);
        }
      }
      if (valid && acc <= thr) {
        qw[cnt * 256 + tid] = acc;
        qj[cnt * 256 + tid] = (unsigned short)(j0 + jj);
        ++cnt;
      }
      if (__any(cnt == Q)) drain();
    }
  }
  drain();
  __syncthreads();  // the last chunk is consumed: its LDS carries the lists now
  double* mg = reinterpret_cast<double*>(kx);  // [wave][KN][64 lanes]
#pragma unroll
  for (int s = 0; s < KN; ++s) mg[(wave * KN + s) * 64 + lane] = L[s];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 1; w < 4; ++w)
#pragma unroll
      for (int s = 0; s < KN; ++s) insert_packed<KN>(L, mg[(w * KN + s) * 64 + lane]);
    if (valid) {
      int* io = idx_out + ((long)b * Nq + i) * KN;
#pragma unroll
      for (int s = 0; s < KN; ++s) io[s] = (int)(__double_as_longlong(L[s]) & 0x1FFFFFFFll);
      if (key_out) {
        float* ko = key_out + ((long)b * Nq + i) * KN;
#pragma unroll
        for (int s = 0; s < KN; ++s) ko[s] = (float)L[s];
      }
    }
  }
}

template <int KN>
static void launch_smallc_fused(const float* xq, long q_bs, int Nq, const float* xk, long k_bs, int Nk, int B, int C,
                                int* idx, float* keys, hipStream_t s) {
  if constexpr (KN <= 8) {
    if (SAMBLE_SMALLC_SPLIT) {
      hipLaunchKernelGGL(knn_smallc_split_kernel<KN>, dim3((Nq + 63) / 64, B), dim3(256), 0, s, xq, q_bs, Nq, xk, k_bs, Nk, C,
                         idx, keys);
      return;
    }
  }
  hipLaunchKernelGGL(knn_smallc_fused_kernel<KN>, dim3((Nq + 255) / 256, B), dim3(256), 0, s, xq, q_bs, Nq, xk, k_bs, Nk,
                     C, idx, keys);
}

extern "C" int samble_launch_knn_stream(const float*, long, int, const float*, long, int, int, int, int, const float*,
                                        int*, float*, hipStream_t);

// Per-cloud scale of the reference (utils/ops.py:27): mean over channels of the unbiased std over
// points of the query set.  One workgroup per cloud; double accumulation, fixed order.
__global__ __launch_bounds__(256) void knn_scale_kernel(const float* __restrict__ x, long bs, int C, int N,
                                                        float* __restrict__ scale_out) {
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  double total = 0.0;
  for (int c = 0; c < C; ++c) {
    const float* p = x + (long)b * bs + (long)c * N;
    double s = 0.0, ss = 0.0;
    for (int n = tid; n < N; n += 256) {
      const double v = p[n];
      s += v;
      ss += v * v;
    }
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    const double sum = red[0];
    __syncthreads();
    red[tid] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    const double sumsq = red[0];
    __syncthreads();
    const double mean = sum / N;
    double var = (sumsq - N * mean * mean) / (N - 1);
    if (var < 0) var = 0;
    total += sqrt(var);
  }
  if (tid == 0) scale_out[b] = (float)(total / C);
}

// dist_out[b][i][s] = sqrt(max(key + |a_i|^2, 0)) / scale_b   (positive, reference-normalised)
__global__ void knn_dist_kernel(const float* __restrict__ keys, const float* __restrict__ qnorm,
                                const float* __restrict__ scale, int Nq, int KN, int add_norm,
                                float* __restrict__ dist) {
  const int b = blockIdx.y;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)Nq * KN) return;
  const int i = (int)(e / KN);
  float d2 = keys[(long)b * Nq * KN + e] + (add_norm ? qnorm[(long)b * Nq + i] : 0.f);
  d2 = fmaxf(d2, 0.f);
  dist[(long)b * Nq * KN + e] = sqrtf(d2) / scale[b];
}

}  // namespace samble

using namespace samble;

template <int KN>
static void launch_select(const float* keyT, int B, int Nq, int Nk, int* idx, float* keys, hipStream_t s) {
  hipLaunchKernelGGL(select_rows_kernel<KN>, dim3((Nq + 255) / 256, B), dim3(256), 0, s, keyT, Nq, Nk, idx, keys);
}


// variant (include/samble.h SAMBLE_KNN_*): bit 0 = fp32-MFMA stream kernel instead of the split-bf16 one,
// bit 1 = the two-kernel path (key matrix through HBM) -- also the fallback for shapes the fused kernels
// do not take
constexpr int kVarF32 = 1, kVarTwoKernel = 2;

// workspace layout: [keyT B*Nk*Nq][knorm B*Nk][qnorm B*Nq][scale B][keys B*Nq*K][mean B*C][operand images]
extern "C" int samble_knn_duo_supported(int C, int K, int Nk);
// Channel count of the fp16 matrix-core kernel's operand images for a C-channel problem, 0 = not its shape.  Point
// sets of fewer than 64 channels (xyz: C = 3) are padded with zero channels: the 64-channel kernel takes 135 us where
// the exact small-C kernel takes 550 (B = 32, N = 2 048, K = 32) -- selection, not the products, is what costs.
static int knn_duo_channels(int C, int K, int Nk, int variant) {
  if (variant & (kVarF32 | kVarTwoKernel)) return 0;
  const int Cd = C == 128 ? 128 : (C >= 1 && C <= 64) ? 64 : 0;
  return (Cd && samble_knn_duo_supported(Cd, K, Nk) && Nk >= 2 * K) ? Cd : 0;
}
static bool knn_uses_fused(int C, int K, int Nk, int variant) {
  if (knn_duo_channels(C, K, Nk, variant)) return true;
  return !(variant & kVarTwoKernel) && (C == 128 || C == 64) && (K == 32 || K == 16) && Nk <= 65536 && Nk >= 2 * K;
}
static bool knn_uses_small_fused(int C, int K, int Nk, int variant) {
  return C <= 8 && !(variant & kVarTwoKernel) && Nk <= 65536 && Nk >= K && !knn_duo_channels(C, K, Nk, variant);
}
extern "C" size_t samble_knn_duo_image_bytes(int B, int C, int N);
extern "C" int samble_launch_knn_duo_prep(const float* xq, long q_bs, int Nq, const float* xk, long k_bs, int Nk, int B,
                                          int Cin, int C, float* mean, float* amax, float* inv_scale, void* qimg, void* kimg,
                                          float* qnorm, float* knorm, hipStream_t s);
extern "C" int samble_launch_knn_duo(const void* qimg, int Nq, const void* kimg, int Nk, int B, int C, int K,
                                     const float* qnorm, const float* knorm, const float* inv_scale, int* idx, float* d2,
                                     hipStream_t s);
// two fp16 planes per operand on the matrix cores (knn_duo.hip): the default for C in {64, 128}, K in {16, 32}
static bool knn_uses_duo(int C, int K, int Nk, int variant) { return knn_duo_channels(C, K, Nk, variant) != 0; }

extern "C" int samble_launch_cloud_mean(const float* x, long bs, int C, int N, int B, float* mean, hipStream_t s) {
  hipLaunchKernelGGL(cloud_mean_kernel, dim3((C + 3) / 4, B), dim3(256), 0, s, x, bs, C, N, mean);
  return (int)hipGetLastError();
}

// the key matrix (B*Nk*Nq floats) is only needed by the two-kernel path; the MFMA paths that form the Gram in
// fp32 (C >= 16) work on centred copies of the two sets
static bool knn_centres_copy(int C, int K, int Nk, int variant) { return C > 8 && !knn_uses_duo(C, K, Nk, variant); }
static size_t knn_base_floats(int B, int C, int Nq, int Nk, int K, int variant) {
  const size_t key_matrix =
      (knn_uses_fused(C, K, Nk, variant) || knn_uses_small_fused(C, K, Nk, variant)) ? 0 : (size_t)B * Nk * Nq;
  const size_t centred = knn_centres_copy(C, K, Nk, variant) ? (size_t)B * C * ((size_t)Nq + Nk) : 0;
  // (mean B*C; the duo path: + the key set's scratch mean, two per-channel extents, 1/scale per cloud)
  const int Cd = knn_duo_channels(C, K, Nk, variant), Cw = Cd > C ? Cd : C;
  const size_t n = key_matrix + (size_t)B * Nk + (size_t)B * Nq + (size_t)B + (size_t)B * Nq * K + (size_t)4 * B * Cw + B +
                   centred + 64;
  return (n + 63) & ~(size_t)63;  // what follows (operand images) stays 256-byte aligned
}

// duo path: + the fp16 operand images of the two point sets (knn_duo.hip)
extern "C" size_t samble_knn_ws_floats(int B, int C, int Nq, int Nk, int K, int variant) {
  size_t n = knn_base_floats(B, C, Nq, Nk, K, variant);
  if (const int Cd = knn_duo_channels(C, K, Nk, variant))
    n += (samble_knn_duo_image_bytes(B, Cd, Nq) + samble_knn_duo_image_bytes(B, Cd, Nk)) / 4;
  return n;
}

extern "C" int samble_launch_knn(const float* xq, long q_bs, int Nq, const float* xk, long k_bs, int Nk, int B, int C,
                                 int K, int variant, int* idx_out, float* dist_out, float* ws, hipStream_t stream) {
  const bool fused = knn_uses_fused(C, K, Nk, variant);
  const bool small_fused = knn_uses_small_fused(C, K, Nk, variant);
  float* keyT = ws;
  float* knorm = keyT + ((fused || small_fused) ? 0 : (size_t)B * Nk * Nq);
  float* qnorm = knorm + (size_t)B * Nk;
  float* scale = qnorm + (size_t)B * Nq;
  float* keys = scale + B;
  float* mean = keys + (size_t)B * Nq * K;
  const bool smallc = C <= 8;
  const float* xq0 = xq;  // the reference-normalised distances are rebuilt from the caller's points
  const long q_bs0 = q_bs;
  if (knn_centres_copy(C, K, Nk, variant)) {
    float* cq = mean + (size_t)B * C;
    const bool same = xq == xk && Nq == Nk && q_bs == k_bs;
    float* ck = same ? cq : cq + (size_t)B * C * Nq;
    int rc = samble_launch_cloud_mean(xq, q_bs, C, Nq, B, mean, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(center_cm_kernel, dim3((Nq + 255) / 256, C, B), dim3(256), 0, stream, xq, q_bs, C, Nq, mean, cq);
    if (!same)
      hipLaunchKernelGGL(center_cm_kernel, dim3((Nk + 255) / 256, C, B), dim3(256), 0, stream, xk, k_bs, C, Nk, mean, ck);
    xq = cq;
    q_bs = (long)C * Nq;
    xk = ck;
    k_bs = (long)C * Nk;
  }
  float* kout = dist_out ? keys : nullptr;
  bool have_qnorm = false;
  if (fused) {
    int rc = 0;
    if (const int Cd = knn_duo_channels(C, K, Nk, variant)) {
      // fp16 matrix cores on two-plane operands: centred, scaled images + norms of the point sets first (one
      // image if the sets coincide); Cd > C: zero channels behind the C real ones
      char* kimg = reinterpret_cast<char*>(ws + knn_base_floats(B, C, Nq, Nk, K, variant));
      const bool same = xq == xk && Nq == Nk && q_bs == k_bs;
      char* qimg = same ? kimg : kimg + samble_knn_duo_image_bytes(B, Cd, Nk);
      float* amax = mean + (size_t)2 * B * Cd;
      float* inv_scale = mean + (size_t)4 * B * Cd;
      rc = samble_launch_knn_duo_prep(xq, q_bs, Nq, same ? nullptr : xk, k_bs, Nk, B, C, Cd, mean, amax, inv_scale, qimg,
                                      kimg, qnorm, knorm, stream);
      have_qnorm = true;
      if (!rc)
        rc = samble_launch_knn_duo(qimg, Nq, kimg, Nk, B, Cd, K, same ? knorm : qnorm, knorm, inv_scale, idx_out, kout,
                                   stream);
    } else {
      hipLaunchKernelGGL(rownorm_kernel, dim3((Nk + 255) / 256, B), dim3(256), 0, stream, xk, k_bs, C, Nk, knorm);
      rc = samble_launch_knn_stream(xq, q_bs, Nq, xk, k_bs, Nk, B, C, K, knorm, idx_out, kout, stream);
    }
    if (rc) return rc;
  } else {
    if (small_fused) {
      Timed timed(kT_knn_small, stream);
      bool ok = true;
      switch (K) {
        case 1: launch_smallc_fused<1>(xq, q_bs, Nq, xk, k_bs, Nk, B, C, idx_out, kout, stream); break;
        case 3: launch_smallc_fused<3>(xq, q_bs, Nq, xk, k_bs, Nk, B, C, idx_out, kout, stream); break;
        case 8: launch_smallc_fused<8>(xq, q_bs, Nq, xk, k_bs, Nk, B, C, idx_out, kout, stream); break;
        case 16: launch_smallc_fused<16>(xq, q_bs, Nq, xk, k_bs, Nk, B, C, idx_out, kout, stream); break;
        case 20: launch_smallc_fused<20>(xq, q_bs, Nq, xk, k_bs, Nk, B, C, idx_out, kout, stream); break;
        case 32: launch_smallc_fused<32>(xq, q_bs, Nq, xk, k_bs, Nk, B, C, idx_out, kout, stream); break;
        case 40: launch_smallc_fused<40>(xq, q_bs, Nq, xk, k_bs, Nk, B, C, idx_out, kout, stream); break;
        case 64: launch_smallc_fused<64>(xq, q_bs, Nq, xk, k_bs, Nk, B, C, idx_out, kout, stream); break;
        default: ok = false;
      }
      if (!ok) return -22;
      goto selected;
    }
    if (smallc) {
      hipLaunchKernelGGL(smallc_keys_kernel, dim3((Nq + 255) / 256, Nk, B), dim3(256), 0, stream, xq, q_bs, Nq, xk,
                         k_bs, Nk, C, keyT);
    } else {
      hipLaunchKernelGGL(rownorm_kernel, dim3((Nk + 255) / 256, B), dim3(256), 0, stream, xk, k_bs, C, Nk, knorm);
      hipLaunchKernelGGL(gram_keys_kernel, dim3((Nq + 127) / 128, (Nk + 127) / 128, B), dim3(256), 0, stream, xq,
                         q_bs, Nq, xk, k_bs, Nk, C, knorm, keyT);
    }
    switch (K) {
      case 1: launch_select<1>(keyT, B, Nq, Nk, idx_out, kout, stream); break;
      case 3: launch_select<3>(keyT, B, Nq, Nk, idx_out, kout, stream); break;
      case 8: launch_select<8>(keyT, B, Nq, Nk, idx_out, kout, stream); break;
      case 16: launch_select<16>(keyT, B, Nq, Nk, idx_out, kout, stream); break;
      case 20: launch_select<20>(keyT, B, Nq, Nk, idx_out, kout, stream); break;
      case 32: launch_select<32>(keyT, B, Nq, Nk, idx_out, kout, stream); break;
      case 40: launch_select<40>(keyT, B, Nq, Nk, idx_out, kout, stream); break;
      case 64: launch_select<64>(keyT, B, Nq, Nk, idx_out, kout, stream); break;
      default: return -22;
    }
  }
selected:
  if (dist_out) {
    // keys of the fused kernels are full squared distances d^2; the two-kernel path returns d^2 - |a|^2
    if (!smallc && !fused && !have_qnorm)
      hipLaunchKernelGGL(rownorm_kernel, dim3((Nq + 255) / 256, B), dim3(256), 0, stream, xq, q_bs, C, Nq, qnorm);
    hipLaunchKernelGGL(knn_scale_kernel, dim3(B), dim3(256), 0, stream, xq0, q_bs0, C, Nq, scale);
    const long tot = (long)Nq * K;
    hipLaunchKernelGGL(knn_dist_kernel, dim3((unsigned)((tot + 255) / 256), B), dim3(256), 0, stream, keys, qnorm, scale,
                       Nq, K, (smallc || fused) ? 0 : 1, dist_out);
  }
  return (int)hipGetLastError();
}
