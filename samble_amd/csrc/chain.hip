// The integer tail of the sampler as TWO launches instead of eleven (reference models/downsample.py:309-344,
// utils/ops.py:174-236, 385-464):
//
//   score_quantiles   one workgroup per cloud (all co-resident): score and z-score of the cloud from the exact
//                     column sums of sparse_score(_map) [what finalize_score_kernel does], then the nb-1 batch
//                     quantiles of ALL B*N z-scores by the same three-level radix select as qsel_kernel<0..3> --
//                     each workgroup histograms its own cloud's values (they never leave its registers), the
//                     per-level global histograms are combined with integer atomics, and the workgroups meet at a
//                     grid barrier per level.  Replaces memset + finalize + 4 qsel launches + the in-degree copy.
//   [the reference's all-reduce of the quantiles over the ranks sits here: torch.distributed, utils/ops.py:191-199]
//   bin_plan          one workgroup per cloud: boundary state update (first call: the quantiles; later: momentum blend,
//                     ops.py:201-233), bin membership + bin weights of the cloud, a grid barrier, then the count
//                     allocation of the whole batch (every workgroup runs it -- it is B x nb numbers -- and keeps its
//                     cloud's row).  Replaces blend_boundaries + bin_assign + alloc_counts.
//
// Every arithmetic step is the shared device function the stand-alone kernels use (select_dev.h), thread-to-point
// mapping and summation orders included: the two paths give the same integers.
//
// Grid barrier (MI355X_MICROARCH.md, "barrier-counter"): monotonic counter in the workspace (zeroed by the
// launcher's memset), lane 0 of each workgroup: release fence, atomic add, relaxed agent-scope poll with s_sleep,
// acquire fence; __syncthreads on both sides.  All B <= 256 workgroups are resident (1024 threads, one per CU),
// which the launchers check against the device's CU count.
namespace samble {
#ifdef SAMBLE_STAMPS  // scratch builds only (tools/scratch): s_memtime marks of workgroup 0, read back by the harness
__device__ unsigned long long g_chain_stamps[64];
#define STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_chain_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
}  // namespace samble
#define BA_STAMP(i) STAMP(i)
#include "select_dev.h"

namespace samble {



constexpr float kUnfixC = 1.f / 17592186044416.f;  // 2^-44: the fixed-point scale of the score accumulators (score.hip)
enum { kColSumC = 0, kColAvgC = 1, kColSqrC = 2, kRowSumC = 3 };

// RELEASE: the workgroups hand over plain stores (needs the L2 write-back of an agent-scope release fence: a few
// microseconds); false when everything exchanged went through agent-scope atomics (performed at the memory side:
// each wave only has to wait until its own have been issued and acknowledged).
//
// Bounded and clean: the barrier assumes that all B workgroups are resident at once, which the launcher can only infer
// (CU count, nothing else on the device).  If a co-tenant, a CU mask or a second stream keeps some of them off the chip,
// the poll gives up after `budget` rounds (default ~1 s), raises the chain's TIMEOUT word and the workgroup LEAVES the
// kernel through chain_bail (valid, harmless outputs for its cloud); every other workgroup sees the word in its own poll
// and leaves the same way, workgroups that start later leave at their first barrier.  Nothing traps: the context stays
// alive, the host reads the word (samble_select_chain_status_async) and falls back to the stage kernels.
// Returns true when the barrier completed (uniform over the workgroup).
// how a barrier gives up: after `budget` poll rounds, telling the caller's mailbox (pinned host memory, may be null)
struct ChainCtl {
  unsigned int budget;
  int* host_status;
};

template <bool RELEASE>
__device__ __forceinline__ bool grid_barrier(unsigned int* counter, unsigned int target, unsigned int* flag,
                                             const ChainCtl ctl, int* ok_lds) {
  const unsigned int budget = ctl.budget;
  int* const host_status = ctl.host_status;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (RELEASE) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (budget 0xFFFFFFFF: fault injection -- the barrier gives up without polling; include/samble.h `spin_budget`)
    bool ok = budget != 0xFFFFFFFFu && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
    unsigned spins = 0;
    while (ok && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      ++spins;
      if (spins > budget || ((spins & 63u) == 0u && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
        ok = false;
      else
        __builtin_amdgcn_s_sleep(1);
    }
    if (ok) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // the caller's mailbox in pinned host memory (include/samble.h `host_status`): raised here, in the one path that
      // needs it, instead of a device-to-host copy of the word behind every launch
      if (host_status) __hip_atomic_store(host_status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    *ok_lds = ok ? 1 : 0;
  }
  __syncthreads();
  return *ok_lds != 0;
}

// What a workgroup leaves behind when the chain gave up (TIMEOUT word raised): every point of its cloud in bin 0 and
// all M picks from that bin -- integers that the kernels downstream (bin_select, the row gather) can run on without
// touching memory they do not own.  The step's results are meaningless; the host learns from the word.
__device__ inline void chain_bail(int b, int N, int nb, int M, unsigned char* __restrict__ member, int* cap, float* w_pre,
                                  float* w, int* counts) {
  for (int n = threadIdx.x; n < N; n += blockDim.x) member[(long)b * N + n] = 1;
  if ((int)threadIdx.x < nb) {
    const int t = threadIdx.x;
    cap[b * nb + t] = t == 0 ? N : 0;
    w_pre[b * nb + t] = 0.f;
    w[b * nb + t] = 0.f;
    counts[b * nb + t] = t == 0 ? M : 0;
  }
}

// chain workspace (uint32 words): [quantile histograms and state: kQWords][barrier counters: 8][TIMEOUT word][pad: 7]
constexpr int kChainBar = kQWords;
constexpr int kChainFlag = kQWords + 8;
constexpr int kChainWords = kQWords + 16;

// LDS of the chain's two bodies (static part; the histograms / score buffer are dynamic)
struct ChainLds {
  double red[258];
  unsigned int scanbuf[2 * 16 * kMaxBins];
  unsigned int prefix[kMaxBins];
  unsigned int rem[kMaxBins];
  double rsum[kMaxBins][16];
  int rcnt[kMaxBins][16];
  float up_s[kMaxBins], lo_s[kMaxBins];
  int counts_s[1024 * kMaxBins / 8];  // B x nb <= 1024 ints (B <= 128 at nb = 8)
  int ok;
};

// score + z of cloud b, then (want_q) the nb-1 batch quantiles: their ordered bits end up in L.prefix[0..nb-2] of EVERY
// workgroup.  PT = ceil(N / 1024) values per thread (point n = tid + 1024 k).  false: the grid barrier gave up.
template <int PT>
__device__ __forceinline__ bool score_quantiles_body(ChainLds& L, unsigned int* qsm, int b, int B,
                                                     const unsigned long long* __restrict__ colacc,
                                                     const int* __restrict__ indeg, const float* __restrict__ rowstat,
                                                     int N, int mode, int nb, float* __restrict__ score,
                                                     float* __restrict__ z, int* __restrict__ indeg_out,
                                                     unsigned int* __restrict__ cws, bool want_q, const ChainCtl budget) {
  double* red = L.red;
  unsigned int* scanbuf = L.scanbuf;
  unsigned int* prefix = L.prefix;
  unsigned int* rem = L.rem;
  const int tid = threadIdx.x;
  const long n_all = (long)B * N;
  const int nq = nb - 1;
  STAMP(0);

  // ---- score + z of this cloud: the arithmetic and summation order of finalize_score_kernel (score.hip): the
  // first 256 threads own points n = tid, tid + 256, ... for the two double-precision reductions
  float* sbuf = reinterpret_cast<float*>(qsm);  // N floats, free again before the histograms
  {
    // every input word of the thread's PT points is requested before any of them is used (a runtime-bound loop waited
    // for each point's loads in turn: stamped 4.7 k cycles for two points)
    unsigned long long ca[PT];
    int dg[PT];
    float rs[PT];
#pragma unroll
    for (int k = 0; k < PT; ++k) {
      ca[k] = 0ull;
      dg[k] = 0;
      rs[k] = 0.f;
    }
    if (mode >= kRowSumC) {
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        const long at = (long)b * N + min(tid + 1024 * k, N - 1);
        rs[k] = rowstat[at];
        if (indeg_out) dg[k] = indeg[at];
      }
    } else {
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        const long at = (long)b * N + min(tid + 1024 * k, N - 1);
        ca[k] = colacc[at];
        dg[k] = indeg[at];
      }
    }
#pragma unroll
    for (int k = 0; k < PT; ++k) {
      const int n = tid + 1024 * k;
      float s;
      if (mode >= kRowSumC) {
        s = rs[k];
      } else {
        const float sum = __ll2float_rn((long long)ca[k]) * kUnfixC;
        const float num = (float)dg[k] + 1e-8f;
        s = sum;
        if (mode == kColAvgC) s = sum / num;
        if (mode == kColSqrC) s = sum / num / num;
      }
      if (s != s) s = 0.f;
      if (n < N) {
        sbuf[n] = s;
        score[(long)b * N + n] = s;
        if (indeg_out) indeg_out[(long)b * N + n] = dg[k];
      }
    }
  }
  __syncthreads();
  STAMP(1);
  // two double-precision reductions in the summation order of finalize_score_kernel: thread t < 256 sums points
  // t, t + 256, ... in order, then the pairwise tree o = 128, 64, ..., 1 (element i takes element i + o).  The tree's
  // last six levels run inside wave 0 on shuffles (a + b is commutative bit for bit), so three barriers per
  // reduction instead of nine.
  constexpr int PQ = 4 * PT;
  float sv[PQ];
#pragma unroll
  for (int k = 0; k < PQ; ++k) sv[k] = sbuf[min((tid & 255) + 256 * k, N - 1)];
  auto tree = [&](double part, int slot) -> double {
    if (tid < 256) red[tid] = part;
    __syncthreads();
    if (tid < 128) red[tid] = red[tid] + red[tid + 128];  // nobody else touches elements tid and tid + 128 here
    __syncthreads();
    if (tid < 64) {
      part = red[tid] + red[tid + 64];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
      if (tid == 0) red[256 + slot] = part;
    }
    __syncthreads();
    return red[256 + slot];
  };
  double part = 0.0;
#pragma unroll
  for (int k = 0; k < PQ; ++k)
    if ((tid & 255) + 256 * k < N) part += (double)sv[k];
  const double mean_d = tree(part, 0) / N;
  part = 0.0;
#pragma unroll
  for (int k = 0; k < PQ; ++k)
    if ((tid & 255) + 256 * k < N) {
      const double d = (double)sv[k] - mean_d;
      part += d * d;
    }
  const double var_sum = tree(part, 1);
  const float mean_f = (float)mean_d;
  const float std_f = (float)sqrt(var_sum / N);
  unsigned int key[PT];
#pragma unroll
  for (int k = 0; k < PT; ++k) {
    const int n = tid + 1024 * k;
    key[k] = 0u;
    if (n < N) {
      const float zv = (sbuf[n] - mean_f) / std_f;
      z[(long)b * N + n] = zv;
      key[k] = ordered_bits(zv);
    }
  }
  __syncthreads();  // sbuf is free
  STAMP(2);
  if (!want_q) return true;  // static boundaries: no quantiles wanted (uniform over the grid)

  // ---- batch quantiles: three levels of digits (11 / 11 / 10 bits), all nb-1 ranks at once
  unsigned int* bar = cws + kChainBar;
#pragma unroll
  for (int level = 0; level < 3; ++level) {
    const int bits = (level == 2) ? 10 : 11, nbin = 1 << bits;
    const int shift = (level == 0) ? 21 : (level == 1) ? 10 : 0;
    const int nh = (level == 0) ? 1 : nq;
    STAMP(3 + 5 * level);
    for (int e = tid; e < nh * nbin; e += 1024) qsm[e] = 0u;
    unsigned int want[kMaxBins];
#pragma unroll
    for (int t = 0; t < kMaxBins; ++t) want[t] = (level > 0 && t < nq) ? (prefix[t] >> (shift + bits)) : 0xFFFFFFFFu;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PT; ++k) {
      if (tid + 1024 * k >= N) continue;
      const unsigned int dig = (unsigned int)(nbin - 1) - ((key[k] >> shift) & (unsigned int)(nbin - 1));
      if (level == 0) {
        atomicAdd(&qsm[dig], 1u);
      } else {
        const unsigned int hi = key[k] >> (shift + bits);
#pragma unroll
        for (int t = 0; t < kMaxBins; ++t)
          if (t < nq && hi == want[t]) atomicAdd(&qsm[t * nbin + dig], 1u);
      }
    }
    __syncthreads();
    STAMP(4 + 5 * level);
    unsigned int* gh = cws + (level == 0 ? kQH0 : level == 1 ? kQH1 : kQH2);
    for (int e = tid; e < nh * nbin; e += 1024) {
      const unsigned int c = qsm[e];
      if (c) atomicAdd(&gh[(level == 2) ? (e / nbin) * 1024 + (e % nbin) : (level == 1) ? (e / nbin) * 2048 + (e % nbin) : e], c);
    }
    STAMP(5 + 5 * level);
    // the histograms were combined by atomics: no release needed
    if (!grid_barrier<false>(bar, (unsigned int)B * (level + 1), cws + kChainFlag, budget, &L.ok)) return false;
    STAMP(6 + 5 * level);
    if (level == 0) qsel_resolve<0>(cws, nq, n_all, nb, prefix, rem, scanbuf, /*keep_state=*/true);
    else if (level == 1) qsel_resolve<1>(cws, nq, n_all, nb, prefix, rem, scanbuf, /*keep_state=*/true);
    else qsel_resolve<2>(cws, nq, n_all, nb, prefix, rem, scanbuf, /*keep_state=*/true);
    STAMP(7 + 5 * level);
  }
  return true;
}

// boundary state update (first call: the quantiles; later: momentum blend, ops.py:201-233), bin membership + bin
// weights of the cloud, a grid barrier, then the count allocation of the whole batch.
// Boundaries in (1,1,1,nb) layout: upper[0] = +inf, upper[t] = q[t-1]; lower[t] = q[t], lower[nb-1] = -inf.
// The nb-1 quantiles: quant (global; null with qbits null = static boundaries) or qbits (their ordered bits in LDS).
__device__ __forceinline__ bool bin_plan_body(ChainLds& L, int b, int B, const float* __restrict__ z,
                                              const float* __restrict__ tok, int nt, const float* __restrict__ quant,
                                              const float* __restrict__ quant_div, const unsigned int* qbits,
                                              float* upper, float* lower, int first,
                                              float mu, float one_minus_mu, int N, int nb, int relu_first, int M,
                                              unsigned char* __restrict__ member, int* cap, float* w_pre, float* w,
                                              int* __restrict__ counts, unsigned int* __restrict__ cws,
                                              const ChainCtl budget) {
  const int tid = threadIdx.x;
  float* up_s = L.up_s;
  float* lo_s = L.lo_s;
  const bool have_q = quant != nullptr || qbits != nullptr;
  // quant_div: the rank average's divisor (reference utils/ops.py:199 `bin_boundaries / world_size`, a true division) --
  // the all-reduced count of ranks whose quantiles are valid, i.e. the world size unless a rank's chain gave up
  const float qd = (quant && quant_div) ? *quant_div : 1.f;
  auto q_at = [&](int t) { return quant ? (quant_div ? quant[t] / qd : quant[t]) : from_ordered_bits(qbits[t]); };
  STAMP(20);
  // a state that was allocated but never written (the first call's chain gave up: the host fills new state with NaN)
  // counts as no state: the quantiles initialise it, as on a first call
  if (have_q && nb >= 2 && upper[1] != upper[1]) first = 1;
  // ---- boundary state (blend_boundaries_kernel's arithmetic: two fp32 products, then the sum; no FMA in this file)
  if (tid < nb) {
    float u = upper[tid], l = lower[tid];
    if (have_q) {
      if (tid >= 1) {
        float v = q_at(tid - 1);
        if (!first) v = u * mu + one_minus_mu * v;
        u = v;
      } else {
        u = first ? __builtin_huge_valf() : u;
      }
      if (tid < nb - 1) {
        float v = q_at(tid);
        if (!first) v = upper[tid + 1] * mu + one_minus_mu * v;
        l = v;
      } else {
        l = first ? -__builtin_huge_valf() : l;
      }
    }
    up_s[tid] = u;
    lo_s[tid] = l;
  }
  __syncthreads();
  STAMP(21);
  bin_assign_body(b, z, tok, nt, up_s, lo_s, N, nb, relu_first, member, cap, w_pre, w, L.rsum, L.rcnt);
  STAMP(22);
  // every workgroup has read the old state and published its cloud's (w, cap): now the state may be overwritten
  // and the whole batch's counts allocated
  if (!grid_barrier<true>(cws + kChainBar + 1, (unsigned int)B, cws + kChainFlag, budget, &L.ok)) return false;
  STAMP(23);
  if (b == 0 && have_q && tid < nb) {
    upper[tid] = up_s[tid];
    lower[tid] = lo_s[tid];
  }
  int* counts_s = L.counts_s;
  if (B <= 64) {  // one wave, lane = cloud, no barriers; the shipped bin counts fully unrolled
    if (nb == 6) alloc_counts_wave<6>(w, cap, B, nb, M, counts_s);
    else if (nb == 4) alloc_counts_wave<4>(w, cap, B, nb, M, counts_s);
    else alloc_counts_wave<0>(w, cap, B, nb, M, counts_s);
  }
  else alloc_counts_lanes(w, cap, B, nb, M, counts_s);          // B <= 128: eight lanes per cloud
  __syncthreads();
  STAMP(24);
  if (tid < nb) counts[b * nb + tid] = counts_s[b * nb + tid];
  return true;
}

template <int PT>
__global__ __launch_bounds__(1024) void score_quantiles_kernel(const unsigned long long* __restrict__ colacc,
                                                               const int* __restrict__ indeg,
                                                               const float* __restrict__ rowstat, int N, int mode,
                                                               int nb, float* __restrict__ score,
                                                               float* __restrict__ z, int* __restrict__ indeg_out,
                                                               unsigned int* __restrict__ cws,
                                                               float* __restrict__ quant_out, const ChainCtl budget) {
  extern __shared__ unsigned int qsm[];  // per-level histogram of this cloud (up to (nb-1) x 2048 words)
  __shared__ ChainLds L;
  const int b = blockIdx.x, B = gridDim.x;
  // (a give-up leaves the TIMEOUT word raised: bin_plan_kernel, which follows on the stream, bails on it)
  // quant_out: nb floats -- the nb-1 quantiles and a validity count (1) that rides through the ranks' all-reduce with
  // them; a give-up leaves all nb at zero: the sum over the ranks then carries only the healthy ranks' quantiles and
  // their number, which bin_plan_kernel divides by
  if (!score_quantiles_body<PT>(L, qsm, b, B, colacc, indeg, rowstat, N, mode, nb, score, z, indeg_out, cws,
                                quant_out != nullptr, budget)) {
    if (quant_out && b == 0 && (int)threadIdx.x < nb) quant_out[threadIdx.x] = 0.f;
    return;
  }
  if (quant_out && b == 0 && (int)threadIdx.x < nb)
    quant_out[threadIdx.x] = (int)threadIdx.x < nb - 1 ? from_ordered_bits(L.prefix[threadIdx.x]) : 1.f;
}

__global__ __launch_bounds__(1024) void bin_plan_kernel(const float* __restrict__ z, const float* __restrict__ tok,
                                                        int nt, const float* __restrict__ quant,
                                                        const float* __restrict__ quant_div, float* upper,
                                                        float* lower, int first, float mu, float one_minus_mu, int N,
                                                        int nb, int relu_first, int M,
                                                        unsigned char* __restrict__ member, int* cap, float* w_pre,
                                                        float* w, int* __restrict__ counts,
                                                        unsigned int* __restrict__ cws, const ChainCtl budget) {
  __shared__ ChainLds L;
  const int b = blockIdx.x, B = gridDim.x;
  const bool dead = __hip_atomic_load(cws + kChainFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;  // uniform
  if (dead || !bin_plan_body(L, b, B, z, tok, nt, quant, quant_div, nullptr, upper, lower, first, mu, one_minus_mu, N, nb, relu_first,
                             M, member, cap, w_pre, w, counts, cws, budget))
    chain_bail(b, N, nb, M, member, cap, w_pre, w, counts);
}

// Both bodies in ONE launch, for a single rank (no all-reduce of the quantiles stands between them): every workgroup
// holds the quantiles in LDS after the third level, so nothing is re-read and no launch boundary is paid.
template <int PT>
__global__ __launch_bounds__(1024) void select_chain_kernel(const unsigned long long* __restrict__ colacc,
                                                            const int* __restrict__ indeg,
                                                            const float* __restrict__ rowstat, int N, int mode, int nb,
                                                            float* __restrict__ score, float* __restrict__ z,
                                                            int* __restrict__ indeg_out, unsigned int* __restrict__ cws,
                                                            float* __restrict__ quant_out, const float* __restrict__ tok,
                                                            int nt, float* upper, float* lower, int first, float mu,
                                                            float one_minus_mu, int relu_first, int M,
                                                            unsigned char* __restrict__ member, int* cap, float* w_pre,
                                                            float* w, int* __restrict__ counts, const ChainCtl budget) {
  extern __shared__ unsigned int qsm[];
  __shared__ ChainLds L;
  const int b = blockIdx.x, B = gridDim.x;
  const bool want_q = quant_out != nullptr;
  bool ok = score_quantiles_body<PT>(L, qsm, b, B, colacc, indeg, rowstat, N, mode, nb, score, z, indeg_out, cws, want_q,
                                     budget);
  if (ok) {
    if (want_q && b == 0 && (int)threadIdx.x < nb - 1) quant_out[threadIdx.x] = from_ordered_bits(L.prefix[threadIdx.x]);
    // (this workgroup wrote its cloud's z itself: the barrier inside the body orders those stores before its loads)
    __syncthreads();
    ok = bin_plan_body(L, b, B, z, tok, nt, nullptr, nullptr, want_q ? L.prefix : nullptr, upper, lower, first, mu, one_minus_mu, N,
                       nb, relu_first, M, member, cap, w_pre, w, counts, cws, budget);
  }
  if (!ok) chain_bail(b, N, nb, M, member, cap, w_pre, w, counts);
}

}  // namespace samble

using namespace samble;

static int resident_workgroups(int* out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  int cus = 0;
  e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return (int)e;
  *out = cus;  // one 1024-thread workgroup per CU
  return 0;
}

#ifdef SAMBLE_STAMPS
extern "C" __attribute__((visibility("default"))) int samble_scratch_chain_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_stamps), sizeof(unsigned long long) * 64);
}
#endif

extern "C" size_t samble_chain_ws_bytes(void) { return (size_t)kChainWords * sizeof(unsigned int); }

// 1 if the fused chain takes this shape (else the caller uses the stand-alone kernels)
extern "C" int samble_chain_supported(int B, int N, int nb) {
  int cus = 0;
  if (resident_workgroups(&cus)) return 0;
  // (2 B <= CUs: head-room for a second rank or another stream sharing the device -- co-residency is inferred, not
  // guaranteed; the barrier's poll is bounded for the cases this does not catch)
  // (N: the score pass in front of the chain keeps a cloud's N column accumulators, 12 bytes each, in LDS)
  return B >= 1 && 2 * B <= cus && B * nb <= 1024 && B <= 128 && N >= 1 && (size_t)N * 12 <= 150 * 1024 && nb >= 2 &&
         nb <= kMaxBins;
}

// poll rounds a grid barrier waits before it gives up: the caller's `spin_budget`, 0 = the default (~1 s)
static inline ChainCtl chain_budget(unsigned int spin_budget, int* host_status) {
  return ChainCtl{spin_budget ? spin_budget : (1u << 20), host_status};
}

extern "C" size_t samble_chain_flag_offset(void) { return (size_t)kChainFlag * sizeof(unsigned int); }

static size_t chain_dyn_lds(int N, int nb) {
  size_t lds = (size_t)(nb - 1) * 2048 * 4;
  if ((size_t)N * 4 > lds) lds = (size_t)N * 4;
  if (lds < 2048 * 4) lds = 2048 * 4;
  return lds;
}

// cws must have been zeroed on the stream (the score launcher's memset covers it)
extern "C" int samble_launch_score_quantiles(const void* colacc, const int* indeg, const float* rowstat, int B, int N,
                                             int mode, int nb, float* score, float* z, int* indeg_out, void* cws,
                                             float* quant_out, unsigned int spin_budget, int* host_status, hipStream_t s) {
  const ChainCtl g_chain_budget = chain_budget(spin_budget, host_status);
  const int pt = (N + 1023) / 1024;
  const size_t lds = chain_dyn_lds(N, nb);
  Timed timed(kT_quantiles, s);
#define SAMBLE_SQ(PT)                                                                                              \
  {                                                                                                                \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(score_quantiles_kernel<PT>),                  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                     \
    if (e != hipSuccess) return (int)e;                                                                            \
    hipLaunchKernelGGL(score_quantiles_kernel<PT>, dim3(B), dim3(1024), lds, s, (const unsigned long long*)colacc, \
                       indeg, rowstat, N, mode, nb, score, z, indeg_out, (unsigned int*)cws, quant_out,            \
                       g_chain_budget);                                                                            \
  }
  if (pt <= 1) SAMBLE_SQ(1)
  else if (pt <= 2) SAMBLE_SQ(2)
  else if (pt <= 4) SAMBLE_SQ(4)
  else if (pt <= 8) SAMBLE_SQ(8)
  else SAMBLE_SQ(16)
#undef SAMBLE_SQ
  return (int)hipGetLastError();
}

extern "C" int samble_launch_bin_plan(const float* z, const float* tok, int nt, const float* quant,
                                      const float* quant_div, float* upper,
                                      float* lower, int first, float mu, float one_minus_mu, int B, int N, int nb,
                                      int relu_first, int M, unsigned char* member, int* cap, float* w_pre, float* w,
                                      int* counts, void* cws, unsigned int spin_budget, int* host_status, hipStream_t s) {
  const ChainCtl g_chain_budget = chain_budget(spin_budget, host_status);
  Timed timed(kT_bin_assign, s);
  hipLaunchKernelGGL(bin_plan_kernel, dim3(B), dim3(1024), 0, s, z, tok, nt, quant, quant_div, upper, lower, first, mu,
                     one_minus_mu, N, nb, relu_first, M, member, cap, w_pre, w, counts, (unsigned int*)cws,
                     g_chain_budget);
  return (int)hipGetLastError();
}

// score_quantiles + bin_plan as one launch (single rank: nothing is exchanged between them)
extern "C" int samble_launch_select_chain(const void* colacc, const int* indeg, const float* rowstat, int B, int N,
                                          int mode, int nb, float* score, float* z, int* indeg_out, void* cws,
                                          float* quant_out, const float* tok, int nt, float* upper, float* lower,
                                          int first, float mu, float one_minus_mu, int relu_first, int M,
                                          unsigned char* member, int* cap, float* w_pre, float* w, int* counts,
                                          unsigned int spin_budget, int* host_status, hipStream_t s) {
  const ChainCtl g_chain_budget = chain_budget(spin_budget, host_status);
  const int pt = (N + 1023) / 1024;
  const size_t lds = chain_dyn_lds(N, nb);
  Timed timed(kT_quantiles, s);
#define SAMBLE_SC(PT)                                                                                              \
  {                                                                                                                \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(select_chain_kernel<PT>),                     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                     \
    if (e != hipSuccess) return (int)e;                                                                            \
    hipLaunchKernelGGL(select_chain_kernel<PT>, dim3(B), dim3(1024), lds, s, (const unsigned long long*)colacc,    \
                       indeg, rowstat, N, mode, nb, score, z, indeg_out, (unsigned int*)cws, quant_out, tok, nt,   \
                       upper, lower, first, mu, one_minus_mu, relu_first, M, member, cap, w_pre, w, counts,        \
                       g_chain_budget);                                                                            \
  }
  if (pt <= 1) SAMBLE_SC(1)
  else if (pt <= 2) SAMBLE_SC(2)
  else if (pt <= 4) SAMBLE_SC(4)
  else if (pt <= 8) SAMBLE_SC(8)
  else SAMBLE_SC(16)
#undef SAMBLE_SC
  return (int)hipGetLastError();
}
