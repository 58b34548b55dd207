// Score-bin partition, per-bin count allocation and the top-k / Boltzmann-random / uniform
// selection that produces the sampled index set (reference utils/ops.py:174-236, 385-619;
// models/downsample.py:264-284), plus the row gather that emits x_ds (downsample.py:242-252).
//
//   batch_quantiles   nb-1 order statistics of ALL B*N z-scores (the reference sorts the whole
//                     batch, ops.py:185-189): 4-pass 8-bit radix select, all ranks at once.
//   bin_assign        per cloud: membership bits (lower[t] <= z < upper[t]), per-bin counts,
//                     masked mean of the token logits -> bin weights.
//   alloc_counts      the reference's float water-filling, same operation order and the same
//                     WHOLE-BATCH early exit (ops.py:403-430), one workgroup for the batch,
//                     no host sync.
//   bin_select        per (cloud, bin): selection key per member, bitonic sort of 64-bit
//                     (descending key, ascending index) composites in LDS, first k emitted.
//   gather_rows       x_ds[b,:,m] = O[b, idx[b,m], :]  (LDS transpose to the (B,C,M) layout).
// All integer outputs are exact functions of the fp32 inputs; every reduction has a fixed order.
#include "select_dev.h"

namespace samble {

// ------------------------------------------------------------------------------------------------
// batch quantiles
// ------------------------------------------------------------------------------------------------
// Radix select on the order-preserving key, digits of 11 / 11 / 10 bits from the top, all nb-1 ranks
// at once.  z-scores cluster (sign and exponent bits nearly constant), so most lanes of a wave hit the
// same few counters and same-address LDS atomics serialise: every counter is therefore kept in kRep
// copies on consecutive banks, lane l adds to copy l % kRep (8x fewer collisions), and the scans sum
// the copies.  Histograms are indexed by the REVERSED digit so that ascending bin order = descending
// value.  A thread's loads of a sweep are all in flight together.
constexpr int kRep = 8;

__global__ __launch_bounds__(1024) void batch_quantiles_kernel(const float* __restrict__ z, long n, int nb,
                                                               float* __restrict__ out) {
  extern __shared__ unsigned int qsm[];
  unsigned int* hist = qsm;                       // max(2048, (nb-1) * 1024 / 2 ...) x kRep, see launcher
  __shared__ unsigned int scanbuf[1024];
  __shared__ unsigned int prefix[kMaxBins];
  __shared__ unsigned int rem[kMaxBins];
  const int tid = threadIdx.x;
  const int rep = tid & (kRep - 1);
  const int nq = nb - 1;
  if (tid < nq) {
    // rank into the DESCENDING order: fp32 arithmetic then truncation (utils/ops.py:182-183)
    const float frac = (float)(tid + 1) / (float)nb;
    rem[tid] = (unsigned int)(int)(frac * (float)n);
    prefix[tid] = 0u;
  }
  // ---- pass 0: top 11 bits, one histogram shared by every rank
  for (int e = tid; e < 2048 * kRep; e += 1024) hist[e] = 0u;
  __syncthreads();
  for (long e0 = 0; e0 < n; e0 += 16 * 1024) {  // 16 loads in flight per thread
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const long e = e0 + u * 1024 + tid;
      v[u] = (e < n) ? z[e] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (e0 + u * 1024 + tid < n) atomicAdd(&hist[(2047u - (ordered_bits(v[u]) >> 21)) * kRep + rep], 1u);
  }
  __syncthreads();
  {
    unsigned int loc[2], ts = 0u;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      unsigned int c = 0u;
#pragma unroll
      for (int r8 = 0; r8 < kRep; ++r8) c += hist[(2 * tid + u) * kRep + r8];
      loc[u] = c;
      ts += c;
    }
    const unsigned int excl = block_scan_incl(ts, scanbuf, tid) - ts;
    unsigned int rr[kMaxBins];
#pragma unroll
    for (int t = 0; t < kMaxBins; ++t) rr[t] = (t < nq) ? rem[t] : 0u;
    __syncthreads();  // every thread has read the ranks before the (single) owner of each rewrites it
#pragma unroll
    for (int t = 0; t < kMaxBins; ++t) {
      if (t >= nq) continue;
      unsigned int c = excl;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (c <= rr[t] && rr[t] < c + loc[u]) {  // exactly one (thread, u) matches
          prefix[t] = (2047u - (unsigned int)(2 * tid + u)) << 21;
          rem[t] = rr[t] - c;
        }
        c += loc[u];
      }
    }
    __syncthreads();
  }
  // ---- pass 1: next 11 bits (2048 bins), pass 2: last 10 bits (1024 bins); one histogram per rank, kRep2 copies
  for (int pass = 1; pass <= 2; ++pass) {
    const int shift = (pass == 1) ? 10 : 0;
    const int bits = (pass == 1) ? 11 : 10;
    const int nbin = 1 << bits;
    constexpr int kRep2 = 2;
    for (int e = tid; e < nq * nbin * kRep2; e += 1024) hist[e] = 0u;
    __syncthreads();
    unsigned int want[kMaxBins];
#pragma unroll
    for (int t = 0; t < kMaxBins; ++t) want[t] = (t < nq) ? (prefix[t] >> (shift + bits)) : 0xFFFFFFFFu;
    for (long e0 = 0; e0 < n; e0 += 16 * 1024) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const long e = e0 + u * 1024 + tid;
        v[u] = (e < n) ? z[e] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (e0 + u * 1024 + tid >= n) continue;
        const unsigned int key = ordered_bits(v[u]);
        const unsigned int hi = key >> (shift + bits);
        const unsigned int dig = (unsigned int)(nbin - 1) - ((key >> shift) & (unsigned int)(nbin - 1));
#pragma unroll
        for (int t = 0; t < kMaxBins; ++t)
          if (t < nq && hi == want[t]) atomicAdd(&hist[(t * nbin + dig) * kRep2 + (tid & (kRep2 - 1))], 1u);
      }
    }
    __syncthreads();
    unsigned int rr[kMaxBins];
#pragma unroll
    for (int t = 0; t < kMaxBins; ++t) rr[t] = (t < nq) ? rem[t] : 0u;
    for (int t = 0; t < nq; ++t) {
      // nbin / 1024 bins per thread (2 in pass 1, 1 in pass 2)
      unsigned int loc[2] = {0u, 0u}, ts = 0u;
      const int per = nbin >> 10;
      for (int u = 0; u < per; ++u) {
        unsigned int c = 0u;
#pragma unroll
        for (int r2 = 0; r2 < kRep2; ++r2) c += hist[(t * nbin + per * tid + u) * kRep2 + r2];
        loc[u] = c;
        ts += c;
      }
      const unsigned int incl = block_scan_incl(ts, scanbuf, tid);
      unsigned int c = incl - ts;
      for (int u = 0; u < per; ++u) {
        if (c <= rr[t] && rr[t] < c + loc[u]) {  // exactly one (thread, u) matches; rem/prefix[t] are read by nobody else now
          prefix[t] |= ((unsigned int)(nbin - 1) - (unsigned int)(per * tid + u)) << shift;
          rem[t] = rr[t] - c;
        }
        c += loc[u];
      }
    }
    __syncthreads();
  }
  if (tid < nq) out[tid] = from_ordered_bits(prefix[tid]);
}

// LEVEL 0: histogram of the top 11 bits.  LEVEL 1/2: resolve the previous level, publish it, histogram the next digit
// of the values that match each rank's prefix.  LEVEL 3: resolve level 2 and write the answers.
template <int LEVEL>
__global__ __launch_bounds__(1024) void qsel_kernel(const float* __restrict__ z, long n, int nb,
                                                    unsigned int* __restrict__ ws, float* __restrict__ out) {
  extern __shared__ unsigned int qsm[];  // LDS histogram of this level
  __shared__ unsigned int scanbuf[2 * 16 * kMaxBins];
  __shared__ unsigned int prefix[kMaxBins];
  __shared__ unsigned int rem[kMaxBins];
  const int tid = threadIdx.x;
  const int nq = nb - 1;
  if (LEVEL > 0) {
    qsel_resolve<LEVEL - 1>(ws, nq, n, nb, prefix, rem, scanbuf);
    if (blockIdx.x == 0 && tid < kMaxBins) {  // identical in every workgroup; one publishes
      ws[kQState + 16 * (LEVEL - 1) + tid] = prefix[tid];
      ws[kQState + 16 * (LEVEL - 1) + 8 + tid] = rem[tid];
    }
    if (LEVEL == 3) {
      if (tid < nq) out[tid] = from_ordered_bits(prefix[tid]);
      return;
    }
  }
  constexpr int bits = (LEVEL == 2) ? 10 : 11, nbin = 1 << bits;
  constexpr int shift = (LEVEL == 0) ? 21 : (LEVEL == 1) ? 10 : 0;
  const int nh = (LEVEL == 0) ? 1 : nq;
  for (int e = tid; e < nh * nbin; e += 1024) qsm[e] = 0u;
  unsigned int want[kMaxBins];
#pragma unroll
  for (int t = 0; t < kMaxBins; ++t) want[t] = (LEVEL > 0 && t < nq) ? (prefix[t] >> (shift + bits)) : 0xFFFFFFFFu;
  __syncthreads();
  const long per_wg = (n + gridDim.x - 1) / gridDim.x;
  const long lo = blockIdx.x * per_wg, hi_e = min(n, lo + per_wg);
  for (long e0 = lo; e0 < hi_e; e0 += 8 * 1024) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long e = e0 + u * 1024 + tid;
      v[u] = (e < hi_e) ? z[e] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (e0 + u * 1024 + tid >= hi_e) continue;
      const unsigned int key = ordered_bits(v[u]);
      const unsigned int dig = (unsigned int)(nbin - 1) - ((key >> shift) & (unsigned int)(nbin - 1));
      if (LEVEL == 0) {
        atomicAdd(&qsm[dig], 1u);
      } else {
        const unsigned int hi = key >> (shift + bits);
#pragma unroll
        for (int t = 0; t < kMaxBins; ++t)
          if (t < nq && hi == want[t]) atomicAdd(&qsm[t * nbin + dig], 1u);
      }
    }
  }
  __syncthreads();
  unsigned int* gh = ws + (LEVEL == 0 ? kQH0 : LEVEL == 1 ? kQH1 : kQH2);
  for (int e = tid; e < nh * nbin; e += 1024) {
    const unsigned int c = qsm[e];
    if (c) atomicAdd(&gh[(LEVEL == 2) ? (e / nbin) * 1024 + (e % nbin) : (LEVEL == 1) ? (e / nbin) * 2048 + (e % nbin) : e], c);
  }
}

// ------------------------------------------------------------------------------------------------
// boundary state update (reference utils/ops.py:201-233) in one launch instead of five tiny tensor ops:
// first != 0: upper = [+inf, q...], lower = [q..., -inf]; else both are blended IN PLACE,
// mixed = old * mu + one_minus_mu * q (two fp32 products, then the sum: this file is built without
// FMA contraction, like the reference's separate multiply / add kernels)
// ------------------------------------------------------------------------------------------------
__global__ void blend_boundaries_kernel(const float* __restrict__ quant, float* __restrict__ upper,
                                        float* __restrict__ lower, int nb, float mu, float one_minus_mu, int first) {
  const int t = threadIdx.x;
  if (t >= nb - 1) return;
  float v = quant[t];
  // (a state of NaN was allocated and never written -- a first call whose fused chain gave up: no state)
  if (!first && upper[1] != upper[1]) first = 1;
  if (!first) {
    const float a = upper[t + 1] * mu;
    const float b = one_minus_mu * v;
    v = a + b;
  }
  upper[t + 1] = v;
  lower[t] = v;
  if (first && t == 0) {
    upper[0] = __builtin_huge_valf();
    lower[nb - 1] = -__builtin_huge_valf();
  }
}

// ------------------------------------------------------------------------------------------------
// bin membership + weights: one workgroup per cloud
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bin_assign_kernel(const float* __restrict__ z, const float* __restrict__ tok,
                                                          int nt, const float* __restrict__ upper,
                                                          const float* __restrict__ lower, int N, int nb,
                                                          int relu_first, unsigned char* __restrict__ member,
                                                          int* __restrict__ cap, float* __restrict__ w_pre,
                                                          float* __restrict__ w) {
  __shared__ double rsum[kMaxBins][16];
  __shared__ int rcnt[kMaxBins][16];
  bin_assign_body(blockIdx.x, z, tok, nt, upper, lower, N, nb, relu_first, member, cap, w_pre, w, rsum, rcnt);
}

// ------------------------------------------------------------------------------------------------
// count allocation: ONE workgroup, thread b = cloud b (B <= 1024)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void alloc_counts_kernel(const float* __restrict__ w, const int* __restrict__ cap,
                                                            int B, int nb, int M, int* __restrict__ counts) {
  alloc_counts_body(w, cap, B, nb, M, counts);
}

// ------------------------------------------------------------------------------------------------
// per-(cloud, bin) selection
// ------------------------------------------------------------------------------------------------
#ifdef SAMBLE_STAMPS  // scratch builds only (tools/scratch)
__device__ unsigned long long g_select_stamps[16];
#define STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_select_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
enum SampleMode { kTopk = 0, kUniform = 1, kRandom = 2, kTopRaw = 3, kBottomRaw = 4 };

// Exp(1) draw of element `elem` under a (seed, offset) Philox state: Philox4x32-10 with the counter laid out as
// curand / torch lay it out (offset in 128-bit blocks in the low words, the element as the subsequence in the high
// ones), first output word, 23 random bits into the open interval (0, 1), -log.  What torch.multinomial draws inside
// (an Exp(1) tensor the size of its input) without the tensor: models/downsample.py:346-362 via utils/ops.py:516-597.
struct PhiloxState {
  unsigned long long seed, offset;  // offset in 32-bit outputs (a multiple of 4), as torch's generator counts it
};
__device__ inline float exp1_draw(PhiloxState st, unsigned long long elem) {
  const unsigned long long blk = st.offset >> 2;
  unsigned c0 = (unsigned)blk, c1 = (unsigned)(blk >> 32), c2 = (unsigned)elem, c3 = (unsigned)(elem >> 32);
  unsigned k0 = (unsigned)st.seed, k1 = (unsigned)(st.seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    c1 = (unsigned)p1;
    c3 = (unsigned)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const float u = ((float)(c0 >> 9) + 0.5f) * 1.1920928955078125e-07f;  // (0, 1): 2^-23 (i + 1/2)
  return -logf(u);
}

// the (rows, N) Exp(1) matrix a seeded bin_select draws from, written out (tests: the seeded select against the
// select on this tensor, and the distribution itself)
__global__ __launch_bounds__(256) void exp1_noise_kernel(PhiloxState st, long n, float* __restrict__ out) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < n) out[e] = exp1_draw(st, (unsigned long long)e);
}
enum TempMode { kTempFixed = 0, kTempCount = 1 };  // count: inv_T = members / temp_div

// dynamic LDS: NP composites (u64) ; grid (nb, B), 1024 threads
__global__ __launch_bounds__(1024) void bin_select_kernel(const float* __restrict__ score, const float* __restrict__ z,
                                                          const unsigned char* __restrict__ member,
                                                          const int* __restrict__ counts,
                                                          const float* __restrict__ noise, PhiloxState philox, int N,
                                                          int NP, int nb, int M, int mode, int temp_mode, float temp,
                                                          long long* __restrict__ idx_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long comp[];
  __shared__ double red[1024];
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const unsigned int bit = 1u << t;
  const unsigned char* mb = member + (long)b * N;
  STAMP(0);

  // Boltzmann normaliser: sum over members of exp(tanh(z) * inv_T), fixed-order double tree
  float inv_t = temp;
  float psum = 1.f;
  int members = 0;
  if (mode == kRandom || mode == kUniform) {
    double part = 0.0;
    int pc = 0;
    if (temp_mode == kTempCount) {
      for (int n = tid; n < N; n += 1024) pc += (mb[n] & bit) ? 1 : 0;
      red[tid] = (double)pc;
      __syncthreads();
      for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
      }
      members = (int)red[0];
      __syncthreads();
      inv_t = (float)members / temp;
    }
    if (mode == kRandom) {
      for (int n = tid; n < N; n += 1024)
        if (mb[n] & bit) part += (double)expf(tanhf(z[(long)b * N + n]) * inv_t);
      red[tid] = part;
      __syncthreads();
      for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
      }
      psum = (float)red[0];
      __syncthreads();
    }
  }
  // the Exp(1) draw of (bin row b nb + t, point n): the caller's tensor, or drawn here
  const long nrow = ((long)b * nb + t) * N;
  auto nzv = [&](int n) { return noise ? noise[nrow + n] : exp1_draw(philox, (unsigned long long)(nrow + n)); };
  STAMP(1);
  int off = 0;
  for (int u = 0; u < t; ++u) off += counts[b * nb + u];
  const int kt = counts[b * nb + t];
  // composite of a member: (descending key, index) -- unique, so ranks are a permutation
  auto composite = [&](int n) {
    float key;
    if (mode == kTopRaw) {
      key = score[(long)b * N + n];  // plain topk(M) of the score (DownSampleGlobal)
    } else if (mode == kBottomRaw) {
      key = -score[(long)b * N + n];  // topk(.., largest=False)
    } else if (mode == kTopk) {
      key = score[(long)b * N + n] + 1e-8f;
    } else if (mode == kUniform) {
      key = 1.f / nzv(n);
    } else {
      float p = expf(tanhf(z[(long)b * N + n]) * inv_t) / psum;
      if (p != p) p = 1e-8f;
      key = p / nzv(n);
    }
    return ((unsigned long long)(~ordered_bits(key)) << 32) | (unsigned int)n;
  };
  // ---- the usual case: the bin holds at least its count.  The members are compacted into LDS (any order) and
  // every member finds its rank by counting the smaller composites (all lanes read the same words: LDS broadcast,
  // no barrier inside): m^2 / 1024 steps per thread, m ~ N / nb -- against 66 barrier-separated passes of a bitonic
  // network over all N points.
  __shared__ int n_members;
  if (tid == 0) n_members = 0;
  __syncthreads();
  for (int n = tid; n < N; n += 1024) {
    if (mb[n] & bit) comp[atomicAdd(&n_members, 1)] = composite(n);
  }
  __syncthreads();
  const int m = n_members;
  if (kt <= m) {
    if (tid == 0) comp[m] = ~0ull;  // pad to an even count (m < NP, or m == NP == N even)
    __syncthreads();
    STAMP(2);
    const int m2 = (m + 1) & ~1;
    for (int i = tid; i < m; i += 1024) {
      const unsigned long long ci = comp[i];
      int rank = 0;
      int j = 0;
      // 16 composites per trip: the eight reads go out together (one LDS latency per trip, not per pair)
      for (; j + 16 <= m2; j += 16) {
        ulonglong2 c2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c2[u] = *reinterpret_cast<const ulonglong2*>(&comp[j + 2 * u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) rank += ((c2[u].x < ci) ? 1 : 0) + ((c2[u].y < ci) ? 1 : 0);
      }
      for (; j < m2; j += 2) {
        const ulonglong2 c2 = *reinterpret_cast<const ulonglong2*>(&comp[j]);
        rank += (c2.x < ci) ? 1 : 0;
        rank += (c2.y < ci) ? 1 : 0;
      }
      if (rank < kt && off + rank < M) idx_out[(long)b * M + off + rank] = (long long)(ci & 0xFFFFFFFFull);
    }
    STAMP(3);
    return;
  }
  // ---- degenerate cloud (NaN scores leave every bin empty and bin 0 is handed all M picks): non-members sort behind
  // every member, in index order, so that the surplus slots receive valid, distinct point indices, as the
  // reference's sort of the masked zeros does (utils/ops.py:486-503)
  __syncthreads();
  for (int n = tid; n < NP; n += 1024) {
    unsigned long long c = (n < N) ? (0xFFFFFFFF00000000ull | (unsigned int)n) : ~0ull;
    if (n < N && (mb[n] & bit)) c = composite(n);
    comp[n] = c;
  }
  __syncthreads();
  for (int k = 2; k <= NP; k <<= 1) {  // bitonic sort ascending
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int e = tid; e < NP; e += 1024) {
        const int partner = e ^ j;
        if (partner > e) {
          const unsigned long long a = comp[e], c2 = comp[partner];
          const bool up = ((e & k) == 0);
          if ((a > c2) == up) {
            comp[e] = c2;
            comp[partner] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int s = tid; s < kt; s += 1024) {
    if (off + s < M) idx_out[(long)b * M + off + s] = (long long)(comp[s] & 0xFFFFFFFFull);
  }
}

// ------------------------------------------------------------------------------------------------
// row gather with transpose to (B, 128, M)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ O, long o_bs, long o_rs,
                                                          const long long* __restrict__ idx, int M,
                                                          float* __restrict__ out) {
  __shared__ float tile[128 * 33];
  const int b = blockIdx.y, m0 = blockIdx.x * 32, tid = threadIdx.x;
  const int sub = tid >> 5, l32 = tid & 31;
  for (int rr = sub; rr < 32; rr += 8) {
    const int m = m0 + rr;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m < M) {
      const long row = idx[(long)b * M + m];
      v = *reinterpret_cast<const f32x4*>(O + (long)b * o_bs + row * o_rs + 4 * l32);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) tile[(4 * l32 + u) * 33 + rr] = v[u];
  }
  __syncthreads();
  float* ob = out + (long)b * 128 * M;
  for (int e = tid; e < 128 * 32; e += 256) {
    const int d = e >> 5, mm = e & 31;
    if (m0 + mm < M) ob[(long)d * M + m0 + mm] = tile[d * 33 + mm];
  }
}

// point-set gather by sampled index: out[b][c][m] = pcd[b][c][idx[b][m]]  (utils/ops.py:136-145)
__global__ void gather_points_kernel(const float* __restrict__ pcd, int C, int N, const long long* __restrict__ idx,
                                     int M, float* __restrict__ out) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  out[((long)b * C + c) * M + m] = pcd[((long)b * C + c) * N + idx[(long)b * M + m]];
}

// ------------------------------------------------------------------------------------------------
// farthest point sampling (reference utils/ops.py:622-643): npoint rounds of {record the current
// farthest point, distance[n] = min(distance[n], |xyz_n - c|^2), farthest = argmax(distance)}.
// One workgroup per cloud, the cloud's points and running distances in registers (PER per thread: 8 up to N = 8192,
// 16 up to 16 384; PER = 0: longer clouds, up to 32 768 points, keep the running distances in LDS and re-read the points
// from the L2 every round), the argmax by wave shuffles + one LDS exchange; ties go to the smallest index (torch.max on
// the CPU returns the first maximum).  The squared distance is summed in the reference's order
// ((dx^2 + dy^2) + dz^2, no FMA contraction: this file is built with -ffp-contract=off).
// ------------------------------------------------------------------------------------------------
constexpr int kFpsMaxN = 32768;

template <int PER>
__global__ __launch_bounds__(1024) void fps_kernel(const float* __restrict__ xyz,  // (B,3,N) channel-major
                                                   const long long* __restrict__ start, int N, int npoint,
                                                   long long* __restrict__ out) {
  extern __shared__ float fps_dist[];   // PER == 0: the N running distances
  __shared__ float wbest[16];
  __shared__ int widx[16];
  __shared__ int cur_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* xb = xyz + (long)b * 3 * N;
  constexpr int kFpsPer = PER ? PER : 1;
  float px[kFpsPer], py[kFpsPer], pz[kFpsPer], dist[kFpsPer];
  if (PER) {
#pragma unroll
    for (int u = 0; u < kFpsPer; ++u) {
      const int n = tid + 1024 * u;
      const bool v = n < N;
      px[u] = v ? xb[n] : 0.f;
      py[u] = v ? xb[N + n] : 0.f;
      pz[u] = v ? xb[2 * N + n] : 0.f;
      dist[u] = v ? 1e10f : -1.f;  // padding can never win the argmax (distances are >= 0)
    }
  } else {
    for (int n = tid; n < N; n += 1024) fps_dist[n] = 1e10f;   // (each thread only ever touches its own entries)
  }
  int cur = (int)start[b];
  for (int it = 0; it < npoint; ++it) {
    if (tid == 0) out[(long)b * npoint + it] = cur;
    const float cx = xb[cur], cy = xb[N + cur], cz = xb[2 * N + cur];
    float best = -2.f;
    int bidx = 0x7fffffff;
    if (PER) {
#pragma unroll
      for (int u = 0; u < kFpsPer; ++u) {
        const float dx = px[u] - cx, dy = py[u] - cy, dz = pz[u] - cz;
        const float d = (dx * dx + dy * dy) + dz * dz;
        if (d < dist[u]) dist[u] = d;
        const int n = tid + 1024 * u;
        if (dist[u] > best) {  // ascending n inside a thread: strict > keeps the smallest index
          best = dist[u];
          bidx = n;
        }
      }
    } else {
#pragma unroll 4
      for (int n = tid; n < N; n += 1024) {
        const float dx = xb[n] - cx, dy = xb[N + n] - cy, dz = xb[2 * N + n] - cz;
        const float d = (dx * dx + dy * dy) + dz * dz;
        float dn = fps_dist[n];
        if (d < dn) fps_dist[n] = dn = d;
        if (dn > best) {
          best = dn;
          bidx = n;
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float ob = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(bidx, off, 64);
      if (ob > best || (ob == best && oi < bidx)) {
        best = ob;
        bidx = oi;
      }
    }
    if (lane == 0) {
      wbest[wv] = best;
      widx[wv] = bidx;
    }
    __syncthreads();
    if (tid < 64) {
      float b2 = (tid < 16) ? wbest[tid] : -2.f;
      int i2 = (tid < 16) ? widx[tid] : 0x7fffffff;
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(b2, off, 64);
        const int oi = __shfl_xor(i2, off, 64);
        if (ob > b2 || (ob == b2 && oi < i2)) {
          b2 = ob;
          i2 = oi;
        }
      }
      if (tid == 0) cur_s = i2;
    }
    __syncthreads();
    cur = cur_s;
  }
}

// ------------------------------------------------------------------------------------------------
// neighbour grouping (reference utils/ops.py:47-65, 83-112): out (B, C or 2C, N, K) from x (B,C,N) and
// the neighbour lists nn (B,N,K).  mode 0 neighbor: x_j; 1 diff: x_j - x_i; 2 center_neighbor: [x_i ; x_j];
// 3 center_diff: [x_i ; x_j - x_i].  One thread per output element of the neighbour half, K innermost
// (coalesced writes, 4-byte gathers served by the L2); the centre half is a broadcast of x.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_gather_kernel(const float* __restrict__ x, const int* __restrict__ nn,
                                                           int C, int N, int K, int mode, float* __restrict__ out) {
  const int b = blockIdx.z, c = blockIdx.y;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;  // n * K + k
  if (e >= (long)N * K) return;
  const int n = (int)(e / K);
  const float* xc = x + ((long)b * C + c) * N;
  const int j = nn[(long)b * N * K + e];
  const float ctr = xc[n];
  float v = xc[j];
  if (mode == 1 || mode == 3) v -= ctr;
  const int CO = (mode >= 2) ? 2 * C : C;
  const long plane = (long)N * K;
  float* ob = out + (long)b * CO * plane;
  if (mode >= 2) {
    ob[(long)c * plane + e] = ctr;
    ob[(long)(C + c) * plane + e] = v;
  } else {
    ob[(long)c * plane + e] = v;
  }
}

}  // namespace samble

using namespace samble;

extern "C" size_t samble_quantiles_ws_bytes(void) { return (size_t)kQWords * sizeof(unsigned int); }

extern "C" int samble_launch_batch_quantiles(const float* z, long n, int nb, float* out, void* ws, hipStream_t s) {
  if (nb < 2 || nb > kMaxBins) return -22;
  if (ws && n >= 16384) {  // multi-workgroup select
    unsigned int* w = reinterpret_cast<unsigned int*>(ws);
    hipError_t e = hipMemsetAsync(w, 0, (size_t)kQWords * sizeof(unsigned int), s);
    if (e != hipSuccess) return (int)e;
    const int G = (int)((n + 2047) / 2048 > 128 ? 128 : (n + 2047) / 2048);
    {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(qsel_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              64 * 1024);
      if (e != hipSuccess) return (int)e;
    }
    Timed timed(kT_quantiles, s);
    hipLaunchKernelGGL(qsel_kernel<0>, dim3(G), dim3(1024), 2048 * 4, s, z, n, nb, w, out);
    hipLaunchKernelGGL(qsel_kernel<1>, dim3(G), dim3(1024), (size_t)(nb - 1) * 2048 * 4, s, z, n, nb, w, out);
    hipLaunchKernelGGL(qsel_kernel<2>, dim3(G), dim3(1024), (size_t)(nb - 1) * 1024 * 4, s, z, n, nb, w, out);
    hipLaunchKernelGGL(qsel_kernel<3>, dim3(1), dim3(1024), 16, s, z, n, nb, w, out);
    return (int)hipGetLastError();
  }
  // dynamic LDS: max(pass 0: 2048 bins x 8 copies, pass 1: (nb-1) x 2048 bins x 2 copies) counters
  size_t words = 2048 * 8;
  if ((size_t)(nb - 1) * 2048 * 2 > words) words = (size_t)(nb - 1) * 2048 * 2;
  const size_t lds = words * sizeof(unsigned int);
  {  // per call: cheap, and correct for every device / thread (no process-wide 'done' flag)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(batch_quantiles_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  Timed timed(kT_quantiles, s);
  hipLaunchKernelGGL(batch_quantiles_kernel, dim3(1), dim3(1024), lds, s, z, n, nb, out);
  return (int)hipGetLastError();
}

extern "C" int samble_launch_blend_boundaries(const float* quant, float* upper, float* lower, int nb, float mu,
                                              float one_minus_mu, int first, hipStream_t s) {
  hipLaunchKernelGGL(blend_boundaries_kernel, dim3(1), dim3(64), 0, s, quant, upper, lower, nb, mu, one_minus_mu, first);
  return (int)hipGetLastError();
}

#ifdef SAMBLE_STAMPS
extern "C" __attribute__((visibility("default"))) int samble_scratch_select_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(samble::g_select_stamps), sizeof(unsigned long long) * 16);
}
#endif

extern "C" int samble_launch_bin_assign(const float* z, const float* tok, int nt, const float* upper,
                                        const float* lower, int B, int N, int nb, int relu_first,
                                        unsigned char* member, int* cap, float* w_pre, float* w, hipStream_t s) {
  if (nb < 1 || nb > kMaxBins || (nt != 1 && nt != nb)) return -22;
  Timed timed(kT_bin_assign, s);
  hipLaunchKernelGGL(bin_assign_kernel, dim3(B), dim3(1024), 0, s, z, tok, nt, upper, lower, N, nb, relu_first, member,
                     cap, w_pre, w);
  return (int)hipGetLastError();
}

extern "C" int samble_launch_alloc_counts(const float* w, const int* cap, int B, int nb, int M, int* counts,
                                          hipStream_t s) {
  if (B > 1024 || nb > kMaxBins) return -22;
  Timed timed(kT_alloc_counts, s);
  hipLaunchKernelGGL(alloc_counts_kernel, dim3(1), dim3(((B + 63) / 64) * 64), 0, s, w, cap, B, nb, M, counts);  // thread = cloud
  return (int)hipGetLastError();
}

extern "C" int samble_launch_bin_select(const float* score, const float* z, const unsigned char* member,
                                        const int* counts, const float* noise, unsigned long long seed,
                                        unsigned long long offset, int B, int N, int nb, int M, int mode,
                                        int temp_mode, float temp, long long* idx_out, hipStream_t s) {
  // noise null (uniform / random): the kernel draws Exp(1) itself under the Philox state (seed, offset)
  if (mode < 0 || mode > kBottomRaw) return -22;
  int NP = 1;
  while (NP < N) NP <<= 1;
  const size_t lds = (size_t)(NP + 2) * 8;  // + the pad word of the rank-by-counting path
  if (lds > 144 * 1024) return -27;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bin_select_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  Timed timed(kT_bin_select, s);
  hipLaunchKernelGGL(bin_select_kernel, dim3(nb, B), dim3(1024), lds, s, score, z, member, counts, noise,
                     PhiloxState{seed, offset}, N, NP, nb, M, mode, temp_mode, temp, idx_out);
  return (int)hipGetLastError();
}

extern "C" int samble_launch_exp1_noise(unsigned long long seed, unsigned long long offset, long n, float* out,
                                        hipStream_t s) {
  hipLaunchKernelGGL(exp1_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, PhiloxState{seed, offset}, n,
                     out);
  return (int)hipGetLastError();
}

extern "C" int samble_launch_group_gather(const float* x, const int* nn, int B, int C, int N, int K, int mode, float* out,
                                          hipStream_t s) {
  hipLaunchKernelGGL(group_gather_kernel, dim3((unsigned)(((long)N * K + 255) / 256), C, B), dim3(256), 0, s, x, nn, C, N,
                     K, mode, out);
  return (int)hipGetLastError();
}

extern "C" int samble_launch_fps(const float* xyz, const long long* start, int B, int N, int npoint, long long* out,
                                 hipStream_t s) {
  if (N > kFpsMaxN) return -22;
  if (N <= 1024 * 8) {
    hipLaunchKernelGGL(fps_kernel<8>, dim3(B), dim3(1024), 0, s, xyz, start, N, npoint, out);
  } else if (N <= 1024 * 16) {
    hipLaunchKernelGGL(fps_kernel<16>, dim3(B), dim3(1024), 0, s, xyz, start, N, npoint, out);
  } else {
    const size_t lds = (size_t)N * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fps_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fps_kernel<0>, dim3(B), dim3(1024), lds, s, xyz, start, N, npoint, out);
  }
  return (int)hipGetLastError();
}

extern "C" int samble_launch_gather_rows(const float* O, long o_bs, long o_rs, const long long* idx, int B, int M,
                                         float* out, hipStream_t s) {
  hipLaunchKernelGGL(gather_rows_kernel, dim3((M + 31) / 32, B), dim3(256), 0, s, O, o_bs, o_rs, idx, M, out);
  return (int)hipGetLastError();
}

extern "C" int samble_launch_gather_points(const float* pcd, int B, int C, int N, const long long* idx, int M,
                                           float* out, hipStream_t s) {
  Timed timed(kT_gather, s);
  hipLaunchKernelGGL(gather_points_kernel, dim3((M + 255) / 256, C, B), dim3(256), 0, s, pcd, C, N, idx, M, out);
  return (int)hipGetLastError();
}
