// Per-point sampling score from the kNN-sparsified attention map
// (reference models/downsample.py:300-344) and the per-cloud z-score (utils/ops.py:450-452).
//
// The reference builds two dense (B,N,N) tensors (mask, A*mask) and reduces them.  Only K entries
// per row are non-zero, so here each (row i, neighbour j) pair recomputes
//     A_ij = exp(scale * <Q_i, K_j> - lse_i)
// from the forward's log-sum-exp and the column sums are accumulated EXACTLY: A_ij in [0,1] is
// converted to 44-bit fixed point and added with integer atomics (LDS first, then one coalesced
// flush per workgroup), so the result does not depend on arrival order and the final conversion
// back to fp32 is the single rounding of the exact sum.  The kNN in-degree (sparse_num,
// downsample.py:311) is counted in the same pass.
#include "samble_dev.h"
#pragma clang fp contract(off)

namespace samble {

constexpr float kFix = 17592186044416.f;      // 2^44
constexpr float kUnfix = 1.f / 17592186044416.f;

enum ScoreMode { kColSum = 0, kColAvg = 1, kColSqr = 2, kRowSum = 3, kRowStd = 4 };

// grid (ceil(N/64), B), 256 threads: 8 half-waves, each walks rows r0+hw, r0+hw+8, ...
// DIRECT: clouds whose N column accumulators (12 bytes each) do not fit a workgroup's LDS (N > 12 800): the same integer
// atomics go to the cloud's global accumulators one by one -- the same sums bit for bit (integers), slower
constexpr size_t kScoreLdsMax = 150 * 1024;
static inline bool score_direct(int N) { return (size_t)N * 12 > kScoreLdsMax; }

template <bool DIRECT>
__global__ __launch_bounds__(256) void sparse_score_kernel(const float* __restrict__ Q, long q_bs, long q_rs,
                                                           const float* __restrict__ K, long k_bs, long k_rs,
                                                           const float* __restrict__ lse,
                                                           const int* __restrict__ nn, int N, int KN, float scale,
                                                           unsigned long long* __restrict__ colacc,
                                                           int* __restrict__ indeg, float* __restrict__ rowstat,
                                                           int row_mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sraw[];
  const int tid = threadIdx.x;
  // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MB
  // L2), so give every XCD its own clouds: the K rows a cloud's workgroups gather (1 MB) are then
  // re-read from that XCD's L2 instead of the Infinity Cache.  Speed only; any mapping is correct.
  int b, chunk;
  xcd_assign(chunk, b);
  unsigned long long* acc = DIRECT ? colacc + (long)b * N : reinterpret_cast<unsigned long long*>(sraw);
  int* cnt = DIRECT ? indeg + (long)b * N : reinterpret_cast<int*>(reinterpret_cast<unsigned long long*>(sraw) + N);
  if (!DIRECT) {
    for (int n = tid; n < N; n += 256) {
      acc[n] = 0ull;
      cnt[n] = 0;
    }
    __syncthreads();
  }
  const int hw = tid >> 5, c = tid & 31;
  const int r0 = chunk * 64;
  const float* Qb = Q + (long)b * q_bs;
  const float* Kb = K + (long)b * k_bs;
  for (int i = r0 + hw; i < min(r0 + 64, N); i += 8) {
    const f32x4 qv = *reinterpret_cast<const f32x4*>(Qb + (long)i * q_rs + 4 * c);
    const float li = lse[(long)b * N + i];
    const int* ni = nn + ((long)b * N + i) * KN;
    float rs = 0.f;
    double rs_d = 0.0, rss_d = 0.0;
    for (int k0 = 0; k0 < KN; k0 += 4) {
      int j[4];
      f32x4 kv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        j[u] = (k0 + u < KN) ? ni[k0 + u] : ni[0];
        kv[u] = *reinterpret_cast<const f32x4*>(Kb + (long)j[u] * k_rs + 4 * c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float s = qv[0] * kv[u][0] + qv[1] * kv[u][1] + qv[2] * kv[u][2] + qv[3] * kv[u][3];
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (k0 + u < KN) {
          const float a = __expf(s * scale - li);
          rs += a;
          rs_d += (double)a;
          rss_d += (double)a * (double)a;
          if (c == 0) {
            atomicAdd(&acc[j[u]], (unsigned long long)__float2ll_rn(a * kFix));
            atomicAdd(&cnt[j[u]], 1);
          }
        }
      }
    }
    if (rowstat && c == 0) {
      float v = rs;
      if (row_mode == kRowStd) {  // unbiased std over the K picked entries (torch.std default)
        const double mean = rs_d / KN;
        v = (float)sqrt(fmax((rss_d - KN * mean * mean) / (KN - 1), 0.0));
      }
      rowstat[(long)b * N + i] = v;
    }
  }
  if (DIRECT) return;
  __syncthreads();
  for (int n = tid; n < N; n += 256) {
    const int cn = cnt[n];
    if (cn) {
      atomicAdd(&colacc[(long)b * N + n], acc[n]);
      atomicAdd(&indeg[(long)b * N + n], cn);
    }
  }
}

// Same statistics with A_ij read from the logit map of attn_stats (attn_map.hip) instead of being
// recomputed: A_ij = exp(S_ij - lse_i), one lane per (row, neighbour) pair.  The reference gathers these
// very entries from its dense map (downsample.py:300-307), so this is also the closer restatement.
// grid (ceil(N/64), B), 256 threads: 8 half-waves, each walks rows r0+hw, r0+hw+8, ...
// COMPACT: `smap` is the (B, N, KN) array of neighbour logits of attn_stats_nl (entry k of row i = the logit of
// neighbour nn[i][k], nn in ascending-index order) instead of the (B, N, ld) logit map.
template <bool COMPACT, bool DIRECT>
__global__ __launch_bounds__(256) void sparse_score_map_kernel(const float* __restrict__ smap, int ld,
                                                               const float* __restrict__ lse,
                                                               const int* __restrict__ nn, int N, int KN,
                                                               unsigned long long* __restrict__ colacc,
                                                               int* __restrict__ indeg, float* __restrict__ rowstat,
                                                               int row_mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sraw[];
  __shared__ double rsum[8][8][2];  // row-statistic partials of (half-wave, row) across neighbour chunks
  const int tid = threadIdx.x;
  int b, chunk;
  xcd_assign(chunk, b);
  unsigned long long* acc = DIRECT ? colacc + (long)b * N : reinterpret_cast<unsigned long long*>(sraw);
  int* cnt = DIRECT ? indeg + (long)b * N : reinterpret_cast<int*>(reinterpret_cast<unsigned long long*>(sraw) + N);
  if (!DIRECT) {
    for (int n = tid; n < N; n += 256) {
      acc[n] = 0ull;
      cnt[n] = 0;
    }
    __syncthreads();
  }
  const int hw = tid >> 5, c = tid & 31;
  const int r0 = chunk * 64;
  // the 8 rows of a half-wave are independent: their neighbour-list loads, then their scattered map reads,
  // are issued together (two global latencies per 32-neighbour chunk instead of sixteen)
  for (int k0 = 0; k0 < KN; k0 += 32) {
    const int k = k0 + c;
    int j[8];
    float li[8], sv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(r0 + hw + 8 * u, N - 1);
      j[u] = nn[((long)b * N + i) * KN + min(k, KN - 1)];
      li[u] = lse[(long)b * N + i];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(r0 + hw + 8 * u, N - 1);
      sv[u] = COMPACT ? smap[((long)b * N + i) * KN + min(k, KN - 1)] : smap[((long)b * N + i) * ld + j[u]];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = r0 + hw + 8 * u;
      double rs_d = 0.0, rss_d = 0.0;
      if (i < N && k < KN) {
        const float a = __expf(sv[u] - li[u]);
        atomicAdd(&acc[j[u]], (unsigned long long)__float2ll_rn(a * kFix));
        atomicAdd(&cnt[j[u]], 1);
        rs_d = (double)a;
        rss_d = (double)a * (double)a;
      }
      if (rowstat) {  // (KN <= 32 in every shipped config; larger K accumulates over the chunks)
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
          rs_d += __shfl_xor(rs_d, off, 64);
          rss_d += __shfl_xor(rss_d, off, 64);
        }
        if (c == 0 && i < N) {
          if (k0 == 0) {
            rsum[hw][u][0] = rs_d;
            rsum[hw][u][1] = rss_d;
          } else {
            rsum[hw][u][0] += rs_d;
            rsum[hw][u][1] += rss_d;
          }
          if (k0 + 32 >= KN) {
            const double tot = rsum[hw][u][0], tot2 = rsum[hw][u][1];
            float v = (float)tot;
            if (row_mode == kRowStd) {  // unbiased std over the K picked entries (torch.std default)
              const double mean = tot / KN;
              v = (float)sqrt(fmax((tot2 - KN * mean * mean) / (KN - 1), 0.0));
            }
            rowstat[(long)b * N + i] = v;
          }
        }
      }
    }
  }
  if (DIRECT) return;
  __syncthreads();
  for (int n = tid; n < N; n += 256) {
    const int cn = cnt[n];
    if (cn) {
      atomicAdd(&colacc[(long)b * N + n], acc[n]);
      atomicAdd(&indeg[(long)b * N + n], cn);
    }
  }
}

// One workgroup per cloud: score from the exact column sums, NaN -> 0, then z-score with the
// reference's operation order  z = (s - mean) / std  (population std), mean and std each the
// correctly rounded fp32 of a double-precision two-pass reduction in a fixed tree order.
__global__ __launch_bounds__(256) void finalize_score_kernel(const unsigned long long* __restrict__ colacc,
                                                             const int* __restrict__ indeg,
                                                             const float* __restrict__ rowstat, int N, int mode,
                                                             float* __restrict__ score, float* __restrict__ z) {
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  double part = 0.0;
  for (int n = tid; n < N; n += 256) {
    float s;
    if (mode >= kRowSum) {
      s = rowstat[(long)b * N + n];
    } else {
      const float sum = __ll2float_rn((long long)colacc[(long)b * N + n]) * kUnfix;
      const float num = (float)indeg[(long)b * N + n] + 1e-8f;
      s = sum;
      if (mode == kColAvg) s = sum / num;
      if (mode == kColSqr) s = sum / num / num;
    }
    if (s != s) s = 0.f;
    score[(long)b * N + n] = s;
    part += (double)s;
  }
  red[tid] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const double mean_d = red[0] / N;
  __syncthreads();
  part = 0.0;
  for (int n = tid; n < N; n += 256) {
    const double d = (double)score[(long)b * N + n] - mean_d;
    part += d * d;
  }
  red[tid] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float mean_f = (float)mean_d;
  const float std_f = (float)sqrt(red[0] / N);
  for (int n = tid; n < N; n += 256) z[(long)b * N + n] = (score[(long)b * N + n] - mean_f) / std_f;
}

// z-score only (a caller that already has the score, e.g. a stage-wise parity test)
__global__ __launch_bounds__(256) void zscore_kernel(const float* __restrict__ score, int N, float* __restrict__ z) {
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  double part = 0.0;
  for (int n = tid; n < N; n += 256) part += (double)score[(long)b * N + n];
  red[tid] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const double mean_d = red[0] / N;
  __syncthreads();
  part = 0.0;
  for (int n = tid; n < N; n += 256) {
    const double d = (double)score[(long)b * N + n] - mean_d;
    part += d * d;
  }
  red[tid] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float mean_f = (float)mean_d;
  const float std_f = (float)sqrt(red[0] / N);
  for (int n = tid; n < N; n += 256) z[(long)b * N + n] = (score[(long)b * N + n] - mean_f) / std_f;
}

}  // namespace samble

using namespace samble;

// ws: [colacc B*N u64][indeg B*N i32][rowstat B*N f32]
extern "C" size_t samble_score_ws_bytes(int B, int N) { return (size_t)B * N * 16 + 64; }

extern "C" int samble_launch_sparse_score(const float* Q, long q_bs, long q_rs, const float* K, long k_bs, long k_rs,
                                          const float* lse, const int* nn, int B, int N, int KN, float scale, int mode,
                                          float* score, float* z, int* indeg_out, void* ws, hipStream_t stream) {
  if (mode < 0 || mode > kRowStd) return -22;
  unsigned long long* colacc = reinterpret_cast<unsigned long long*>(ws);
  int* indeg = reinterpret_cast<int*>(colacc + (size_t)B * N);
  float* rowstat = reinterpret_cast<float*>(indeg + (size_t)B * N);
  hipError_t e = hipMemsetAsync(ws, 0, (size_t)B * N * 12, stream);
  if (e != hipSuccess) return (int)e;
  const bool direct = score_direct(N);
  const size_t lds = direct ? 0 : (size_t)N * 12;
  auto kern = direct ? sparse_score_kernel<true> : sparse_score_kernel<false>;
  if (lds > 64 * 1024) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  Timed timed(kT_sparse_score, stream);
  hipLaunchKernelGGL(kern, dim3((N + 63) / 64, B), dim3(256), lds, stream, Q, q_bs, q_rs, K, k_bs, k_rs,
                     lse, nn, N, KN, scale, colacc, indeg, mode >= kRowSum ? rowstat : nullptr, mode);
  hipLaunchKernelGGL(finalize_score_kernel, dim3(B), dim3(256), 0, stream, colacc, indeg, rowstat, N, mode, score, z);
  if (indeg_out) {
    e = hipMemcpyAsync(indeg_out, indeg, (size_t)B * N * sizeof(int), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return (int)e;
  }
  return (int)hipGetLastError();
}

extern "C" int samble_launch_sparse_score_map(const float* smap, int ld, const float* lse, const int* nn, int B, int N,
                                              int KN, int mode, float* score, float* z, int* indeg_out, void* ws,
                                              hipStream_t stream) {
  if (mode < 0 || mode > kRowStd) return -22;
  unsigned long long* colacc = reinterpret_cast<unsigned long long*>(ws);
  int* indeg = reinterpret_cast<int*>(colacc + (size_t)B * N);
  float* rowstat = reinterpret_cast<float*>(indeg + (size_t)B * N);
  hipError_t e = hipSuccess;
  Timed timed(kT_sparse_score, stream);
  if (smap) {  // null: the accumulators were filled by attn_stats_nl_tri (attn_tri.hip)
    e = hipMemsetAsync(ws, 0, (size_t)B * N * 12, stream);
    if (e != hipSuccess) return (int)e;
    const bool direct = score_direct(N);
    const size_t lds = direct ? 0 : (size_t)N * 12;
    auto kern = ld == 0 ? (direct ? sparse_score_map_kernel<true, true> : sparse_score_map_kernel<true, false>)   // ld == 0: compact logits
                        : (direct ? sparse_score_map_kernel<false, true> : sparse_score_map_kernel<false, false>);
    if (lds > 64 * 1024) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((N + 63) / 64, B), dim3(256), lds, stream, smap, ld, lse, nn, N, KN, colacc, indeg,
                       mode >= kRowSum ? rowstat : nullptr, mode);
  }
  hipLaunchKernelGGL(finalize_score_kernel, dim3(B), dim3(256), 0, stream, colacc, indeg, rowstat, N, mode, score, z);
  if (indeg_out) {
    e = hipMemcpyAsync(indeg_out, indeg, (size_t)B * N * sizeof(int), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return (int)e;
  }
  return (int)hipGetLastError();
}

// the accumulation pass alone, for the fused select chain (chain.hip): zeroes `zero_bytes` of ws (accumulators and
// whatever follows them: the chain's histograms and barrier counters), then gathers the map entries
// (ld == 0: `smap` holds the compact neighbour logits (B, N, KN) of attn_stats_nl, nn their ascending index lists)
extern "C" int samble_launch_sparse_score_map_acc(const float* smap, int ld, const float* lse, const int* nn, int B, int N,
                                                  int KN, int mode, void* ws, size_t zero_bytes, hipStream_t stream) {
  if (mode < 0 || mode > kRowStd) return -22;
  unsigned long long* colacc = reinterpret_cast<unsigned long long*>(ws);
  int* indeg = reinterpret_cast<int*>(colacc + (size_t)B * N);
  float* rowstat = reinterpret_cast<float*>(indeg + (size_t)B * N);
  hipError_t e = hipMemsetAsync(ws, 0, zero_bytes, stream);
  if (e != hipSuccess) return (int)e;
  const size_t lds = (size_t)N * 12;   // (the chain takes N <= 12 800 only: samble_chain_supported)
  auto kern = ld == 0 ? sparse_score_map_kernel<true, false> : sparse_score_map_kernel<false, false>;
  if (lds > 64 * 1024) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  Timed timed(kT_sparse_score, stream);
  hipLaunchKernelGGL(kern, dim3((N + 63) / 64, B), dim3(256), lds, stream, smap, ld, lse, nn, N, KN, colacc, indeg,
                     mode >= kRowSum ? rowstat : nullptr, mode);
  return (int)hipGetLastError();
}

namespace samble {
// Neighbour lists for the map-free forward (attn_stats_nl_tri_kernel): per query the K neighbour indices in
// ASCENDING order (rank by counting: the K indices of a row are distinct) and one 32-bit mask per (query, tile of
// 32 point keys): bit k set <=> key 32 t + k is a neighbour.  masks (B, T, N), T = ceil(N / 32): the word of a
// (tile, 32 consecutive queries) is one coalesced 128-byte line.
// Workgroup = 64 queries x 4 waves; wave p owns a quarter of every query's list (entries KN/4 p ..) and a quarter of
// the mask tiles: a thread ranks its KN/4 entries against the query's KN (all of them read back from an LDS tile),
// ORs their bits into the query's column of mask words in LDS and stores its 16 tiles of the column.  (One thread per
// query -- 1024 compares and 64 stores in a row on one wave per SIMD -- took 21 us at B = 32, N = 2048.)
constexpr int kNnpQ = 64, kNnpThreads = 4 * kNnpQ;
template <int KN>
__global__ __launch_bounds__(kNnpThreads) void nn_prepare_kernel(const int* __restrict__ nn, int N, int T,
                                                                 int* __restrict__ nn_sorted, unsigned* __restrict__ masks,
                                                                 f32x4* __restrict__ clear, long clear_quads) {
  constexpr int PER = KN / 4;
  if (clear) {  // the statistics pass that follows accumulates into a zeroed workspace: cleared here, not by a launch
    const long total = (long)gridDim.x * gridDim.y * kNnpThreads;
    for (long e = ((long)blockIdx.y * gridDim.x + blockIdx.x) * kNnpThreads + threadIdx.x; e < clear_quads; e += total)
      clear[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __shared__ int vals[kNnpQ][KN + 1];     // the lists as loaded, then in ascending order
  __shared__ unsigned words[64][kNnpQ];   // 64 mask tiles x the 64 queries
  const int b = blockIdx.y, tid = threadIdx.x;
  const int q = tid & 63, part = tid >> 6;
  const int i0 = blockIdx.x * kNnpQ, i = i0 + q;
  const bool live = i < N;
  int v[PER];
  {
    const int* row = nn + ((long)b * N + min(i, N - 1)) * KN + PER * part;
#pragma unroll
    for (int k4 = 0; k4 < PER / 4; ++k4) {
      const int4 o = *reinterpret_cast<const int4*>(row + 4 * k4);
      v[4 * k4] = o.x;
      v[4 * k4 + 1] = o.y;
      v[4 * k4 + 2] = o.z;
      v[4 * k4 + 3] = o.w;
    }
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) vals[q][PER * part + k] = v[k];
  __syncthreads();
  int rank[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) rank[k] = 0;
#pragma unroll
  for (int k2 = 0; k2 < KN; ++k2) {
    const int o = vals[q][k2];
#pragma unroll
    for (int k = 0; k < PER; ++k) rank[k] += (o < v[k]) ? 1 : 0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; ++k) vals[q][rank[k]] = v[k];
  __syncthreads();
  {  // the sorted lists leave as whole lines: KN / 4 consecutive lanes cover one query's list
    constexpr int kQuads = KN / 4;
#pragma unroll
    for (int e = tid; e < kNnpQ * kQuads; e += kNnpThreads) {
      const int r = e / kQuads, c4 = e % kQuads;
      if (i0 + r < N) {
        const int* src = &vals[r][4 * c4];
        const int4 o = {src[0], src[1], src[2], src[3]};
        *reinterpret_cast<int4*>(nn_sorted + ((long)b * N + i0 + r) * KN + 4 * c4) = o;
      }
    }
  }
  for (int t0 = 0; t0 < T; t0 += 64) {
#pragma unroll
    for (int t = 0; t < 16; ++t) words[16 * part + t][q] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int t = (v[k] >> 5) - t0;
      if (t >= 0 && t < 64) atomicOr(&words[t][q], 1u << (v[k] & 31));
    }
    __syncthreads();
    unsigned w[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) w[t] = words[16 * part + t][q];
    if (live) {
#pragma unroll
      for (int t = 0; t < 16; ++t)
        if (t0 + 16 * part + t < T) masks[((long)b * T + t0 + 16 * part + t) * N + i] = w[t];
    }
    __syncthreads();
  }
}
}  // namespace samble

extern "C" int samble_launch_nn_prepare(const int* nn, int B, int N, int KN, int* nn_sorted, unsigned* masks,
                                        void* clear, size_t clear_bytes, hipStream_t stream) {
  const int T = (N + 31) / 32;
  if ((reinterpret_cast<size_t>(nn) | reinterpret_cast<size_t>(nn_sorted)) & 15) return -22;  // 16-byte row pieces
  if (KN != 32 && KN != 16) return -22;
  if (clear && ((reinterpret_cast<size_t>(clear) | clear_bytes) & 15)) {  // odd piece: a memset of its own
    hipError_t e = hipMemsetAsync(clear, 0, clear_bytes, stream);
    if (e != hipSuccess) return (int)e;
    clear = nullptr;
  }
  f32x4* cl = clear_bytes ? reinterpret_cast<f32x4*>(clear) : nullptr;
  const long quads = (long)(clear_bytes / 16);
  const dim3 grid((N + kNnpQ - 1) / kNnpQ, B);
  Timed timed(kT_nn_prepare, stream);
  if (KN == 32)
    hipLaunchKernelGGL(nn_prepare_kernel<32>, grid, dim3(kNnpThreads), 0, stream, nn, N, T, nn_sorted, masks, cl, quads);
  else
    hipLaunchKernelGGL(nn_prepare_kernel<16>, grid, dim3(kNnpThreads), 0, stream, nn, N, T, nn_sorted, masks, cl, quads);
  return (int)hipGetLastError();
}

// score = stat with NaN -> 0 (models/downsample.py:342), then the z-score
__global__ __launch_bounds__(256) void clean_kernel(const float* __restrict__ stat, long n, float* __restrict__ score) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < n) {
    const float v = stat[e];
    score[e] = (v != v) ? 0.f : v;
  }
}

extern "C" int samble_launch_stat_score(const float* stat, int B, int N, float* score, float* z, hipStream_t stream) {
  const long n = (long)B * N;
  hipLaunchKernelGGL(clean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, stat, n, score);
  hipLaunchKernelGGL(zscore_kernel, dim3(B), dim3(256), 0, stream, score, N, z);
  return (int)hipGetLastError();
}

extern "C" int samble_launch_zscore(const float* score, int B, int N, float* z, hipStream_t stream) {
  hipLaunchKernelGGL(zscore_kernel, dim3(B), dim3(256), 0, stream, score, N, z);
  return (int)hipGetLastError();
}
