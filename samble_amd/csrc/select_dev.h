// Device functions shared by the stand-alone select kernels (select.hip) and the fused select chain (chain.hip):
// the batch-quantile radix select's resolve step, the bin assignment and the count allocation.  One definition,
// so that both paths produce the same integers from the same fp32 inputs.
#pragma once
#include "samble_dev.h"
#pragma clang fp contract(off)

#ifndef BA_STAMP
#define BA_STAMP(i) do { } while (0)
#endif

namespace samble {

constexpr int kMaxBins = 8;

// inclusive scan of one value per thread over the 1024-thread block: shuffles inside a wave, the 16
// wave totals through LDS (buf: >= 16 words); two barriers
__device__ __forceinline__ unsigned int block_scan_incl(unsigned int v, unsigned int* buf, int tid) {
  const int lane = tid & 63, wv = tid >> 6;
  unsigned int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned int up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  __syncthreads();  // previous users of buf are done
  if (lane == 63) buf[wv] = incl;
  __syncthreads();
  unsigned int base = 0u;
  for (int w2 = 0; w2 < wv; ++w2) base += buf[w2];
  return base + incl;
}

// ---- multi-workgroup version of the same radix select (one CU sweeping B*N values is VALU-bound:
// ~30 instructions per value on 4 SIMDs).  Three sweep kernels of G workgroups + a one-workgroup
// finish; level L's sweep first resolves level L-1 from its global histogram (every workgroup does
// that redundantly and writes the identical result: no tickets, no fences), then histograms its own
// slice in LDS and adds the non-empty counters to the global histogram of level L.
// ws (uint32): hist0[2048] | hist1[7*2048] | hist2[7*1024] | 3 x state {prefix[8], rem[8]}; zeroed by the launcher.
constexpr int kQH0 = 0, kQH1 = 2048, kQH2 = 2048 + 7 * 2048, kQState = kQH2 + 7 * 1024, kQWords = kQState + 3 * 16;  // state: one slot per resolved level

// Inclusive prefix sum over the 64 lanes of a wave in DPP: Hillis-Steele inside each row of 16 (row_shr 1,2,4,8, lanes
// without a source read 0), then row_bcast15 into rows 1 and 3 and row_bcast31 into rows 2 and 3.  Seven VALU
// instructions and no LDS crossbar traffic (the __shfl_up version was 6 dependent ds_bpermute per rank).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned int qsel_dpp(unsigned int x) {
  return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ unsigned int qsel_wave_scan(unsigned int x) {
  x += qsel_dpp<0x111, 0xF>(x);
  x += qsel_dpp<0x112, 0xF>(x);
  x += qsel_dpp<0x114, 0xF>(x);
  x += qsel_dpp<0x118, 0xF>(x);
  x += qsel_dpp<0x142, 0xA>(x);
  x += qsel_dpp<0x143, 0xC>(x);
  return x;
}

// resolve the ranks of level `level` (0,1,2) from its global histogram; all 1024 threads; result in prefix/rem (LDS)
// keep_state: prefix / rem of the previous level are already in LDS (the fused chain resolves every level in the
// same workgroup); otherwise they are read from the state block 0 of the previous sweep published in ws.
template <int level>
__device__ inline void qsel_resolve(const unsigned int* ws, int nq, long n, int nb, unsigned int* prefix,
                                    unsigned int* rem, unsigned int* scanbuf, bool keep_state = false) {
  const int tid = threadIdx.x;
  if (keep_state && level > 0) {
  } else if (level == 0) {
    if (tid < kMaxBins) {
      const float frac = (float)(tid + 1) / (float)nb;  // fp32 arithmetic then truncation (utils/ops.py:182-183)
      rem[tid] = (tid < nq) ? (unsigned int)(int)(frac * (float)n) : 0u;
      prefix[tid] = 0u;
    }
  } else if (tid < kMaxBins) {
    prefix[tid] = ws[kQState + 16 * (level - 1) + tid];  // published by block 0 of the previous sweep
    rem[tid] = ws[kQState + 16 * (level - 1) + 8 + tid];
  }
  __syncthreads();
  constexpr int bits = (level == 2) ? 10 : 11, nbin = 1 << bits, per = nbin >> 10;
  constexpr int shift = (level == 0) ? 21 : (level == 1) ? 10 : 0;
  unsigned int rr[kMaxBins], pp[kMaxBins];
#pragma unroll
  for (int t = 0; t < kMaxBins; ++t) {
    rr[t] = rem[t];
    pp[t] = prefix[t];
  }
  // all nq ranks resolved by ONE scan round (two barriers): every thread carries the nq per-rank partial counts
  // side by side (level 0 has one histogram for all ranks).  scanbuf: 2 x 16 x kMaxBins words.
  unsigned int loc[kMaxBins][2], ts[kMaxBins], incl[kMaxBins];
  // every histogram word this thread needs, loaded unconditionally (rank index clamped) before any of them is used:
  // inside the per-rank branches each load was waited for in turn (stamped: 15 k cycles at level 1 against 5 k at
  // level 0, which reads one histogram)
  unsigned int raw[kMaxBins][2];
  if (level == 1) BA_STAMP(40);
#pragma unroll
  for (int t = 0; t < (level == 0 ? 1 : kMaxBins); ++t) {
    const int tt = min(t, max(nq - 1, 0));
    const unsigned int* h = ws + (level == 0 ? kQH0 : level == 1 ? kQH1 + tt * 2048 : kQH2 + tt * 1024);
#pragma unroll
    for (int u = 0; u < per; ++u) raw[t][u] = h[per * tid + u];
  }
#pragma unroll
  for (int t = 0; t < kMaxBins; ++t) {
    loc[t][0] = loc[t][1] = ts[t] = 0u;
    if (t < nq && (level > 0 || t == 0)) {
#pragma unroll
      for (int u = 0; u < per; ++u) {
        loc[t][u] = raw[t][u];
        ts[t] += loc[t][u];
      }
    }
  }
  if (level == 0) {
#pragma unroll
    for (int t = 1; t < kMaxBins; ++t) {
      loc[t][0] = loc[0][0];
      loc[t][1] = loc[0][1];
      ts[t] = ts[0];
    }
  }
  if (level == 1) BA_STAMP(41);
  {
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int t = 0; t < kMaxBins; ++t) incl[t] = (level == 0 && t > 0) ? 0u : qsel_wave_scan(ts[t]);
    if (level == 1) BA_STAMP(42);
    if (lane == 63) {
#pragma unroll
      for (int t = 0; t < kMaxBins; ++t) scanbuf[wv * kMaxBins + t] = incl[t];
    }
    __syncthreads();
    if (level == 1) BA_STAMP(43);
    // exclusive scan of the 16 wave totals of every rank: thread (t, w) = tid 16 t + w, one DPP row each
    if (tid < 16 * kMaxBins) {
      const int w = tid & 15, t = tid >> 4;
      const unsigned int mine = scanbuf[w * kMaxBins + t];
      unsigned int acc = mine;  // w is the lane's position in its DPP row of 16
      acc += qsel_dpp<0x111, 0xF>(acc);
      acc += qsel_dpp<0x112, 0xF>(acc);
      acc += qsel_dpp<0x114, 0xF>(acc);
      acc += qsel_dpp<0x118, 0xF>(acc);
      scanbuf[16 * kMaxBins + w * kMaxBins + t] = acc - mine;
    }
    __syncthreads();
    if (level == 1) BA_STAMP(44);
#pragma unroll
    for (int t = 0; t < (level == 0 ? 1 : kMaxBins); ++t) incl[t] += scanbuf[16 * kMaxBins + wv * kMaxBins + t];
    if (level == 0) {
#pragma unroll
      for (int t = 1; t < kMaxBins; ++t) incl[t] = incl[0];
    }
  }
#pragma unroll
  for (int t = 0; t < kMaxBins; ++t) {
    unsigned int c = incl[t] - ts[t];
#pragma unroll
    for (int u = 0; u < per; ++u) {
      if (t < nq && c <= rr[t] && rr[t] < c + loc[t][u]) {  // exactly one (thread, u) matches
        prefix[t] = pp[t] | (((unsigned int)(nbin - 1) - (unsigned int)(per * tid + u)) << shift);
        rem[t] = rr[t] - c;
      }
      c += loc[t][u];
    }
  }
  __syncthreads();
}

// Sum over the 64 lanes of a wave, delivered in lane 63: quad swaps, half-row and row mirrors, then the row broadcasts
// of gfx9 DPP (row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3).  A fixed tree: run-to-run identical.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int x) {
  return __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ int wave_sum_i32(int x) {
  x += dpp_i32<0xB1, 0xF>(x);   // quad_perm [1,0,3,2]
  x += dpp_i32<0x4E, 0xF>(x);   // quad_perm [2,3,0,1]
  x += dpp_i32<0x141, 0xF>(x);  // row_half_mirror
  x += dpp_i32<0x140, 0xF>(x);  // row_mirror
  x += dpp_i32<0x142, 0xA>(x);  // row_bcast15 -> rows 1, 3
  x += dpp_i32<0x143, 0xC>(x);  // row_bcast31 -> rows 2, 3
  return x;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double x) {
  const long long u = __double_as_longlong(x);
  const int lo = dpp_i32<CTRL, ROW_MASK>((int)u), hi = dpp_i32<CTRL, ROW_MASK>((int)(u >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);  // (rows masked off receive +0.0)
}
__device__ __forceinline__ double wave_sum_f64(double x) {
  x += dpp_f64<0xB1, 0xF>(x);
  x += dpp_f64<0x4E, 0xF>(x);
  x += dpp_f64<0x141, 0xF>(x);
  x += dpp_f64<0x140, 0xF>(x);
  x += dpp_f64<0x142, 0xA>(x);
  x += dpp_f64<0x143, 0xC>(x);
  return x;
}

// bin membership + weights of cloud b (reference utils/ops.py:454-463, models/downsample.py:264-284); the whole
// 1024-thread workgroup takes part; rsum / rcnt: LDS scratch
__device__ inline void bin_assign_body(int b, const float* __restrict__ z, const float* __restrict__ tok, int nt,
                                       const float* upper, const float* lower, int N, int nb, int relu_first,
                                       unsigned char* __restrict__ member, int* cap, float* w_pre, float* w,
                                       double (*rsum)[16], int (*rcnt)[16]) {
  // 1024 threads per cloud (two points each at N = 2048: the per-point token-logit loads are latency
  // bound); per-bin sums in double: lane tree inside a wave, then the 16 wave partials in index order
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float up[kMaxBins], lo[kMaxBins];
  double ps[kMaxBins];
  int pc[kMaxBins];
#pragma unroll
  for (int t = 0; t < kMaxBins; ++t) {
    up[t] = (t < nb) ? upper[t] : 0.f;
    lo[t] = (t < nb) ? lower[t] : 0.f;
    ps[t] = 0.0;
    pc[t] = 0;
  }
  for (int n = tid; n < N; n += 1024) {
    const float zv = z[(long)b * N + n];
    // the point's token logits first, all of them and unconditionally: loads inside the per-bin branches would be
    // waited for one after the other (stamped: six memory latencies per point, 26 k cycles for the cloud)
    float lgs[kMaxBins];
#pragma unroll
    for (int t = 0; t < kMaxBins; ++t) lgs[t] = (t < nt) ? tok[((long)b * N + n) * nt + t] : 0.f;
    unsigned int bits = 0;
#pragma unroll
    for (int t = 0; t < kMaxBins; ++t) {
      if (t < nb && zv < up[t] && zv >= lo[t]) {
        bits |= 1u << t;
        float lg = nt == 1 ? lgs[0] : lgs[t];
        if (relu_first) lg = fmaxf(lg, 0.f);
        ps[t] += (double)lg;
        pc[t] += 1;
      }
    }
    member[(long)b * N + n] = (unsigned char)bits;
  }
  BA_STAMP(30);
  // wave sums by DPP (vector ALU only; lane 63 ends up with the total): 16 waves x 8 bins x 6 xor-shuffle steps of a
  // double and an int were 336 ds_bpermute per wave -- the LDS crossbar of the CU, not the arithmetic, set the pace
  // (stamped: 7-15 k cycles of the body's 22 k)
#pragma unroll
  for (int t = 0; t < kMaxBins; ++t) {
    if (t >= nb) break;
    const double v = wave_sum_f64(ps[t]);
    const int c = wave_sum_i32(pc[t]);
    if (lane == 63) {
      rsum[t][wv] = v;
      rcnt[t][wv] = c;
    }
  }
  BA_STAMP(31);
  __syncthreads();
  BA_STAMP(32);
  if (tid < nb) {
    double v = 0.0;
    int c = 0;
    for (int w2 = 0; w2 < 16; ++w2) {
      v += rsum[tid][w2];
      c += rcnt[tid][w2];
    }
    const float pre = (float)v / ((float)c + 1e-8f);
    cap[b * nb + tid] = c;
    w_pre[b * nb + tid] = pre;
    w[b * nb + tid] = relu_first ? pre : fmaxf(pre, 0.f);
  }
}

// Sum of n <= 8 floats in the order ATen's scalar reduction path uses for a short contiguous row
// (4 interleaved partial sums, tail onto partial 0, partials folded left to right).
__device__ __forceinline__ float short_row_sum(const float (&a)[kMaxBins], int n) {
  float part[4] = {0.f, 0.f, 0.f, 0.f};
  const int q = n >> 2;
  for (int i = 0; i < q; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) part[k] = __fadd_rn(part[k], a[4 * i + k]);
  for (int i = 4 * q; i < n; ++i) part[0] = __fadd_rn(part[0], a[i]);
  float s = part[0];
#pragma unroll
  for (int k = 1; k < 4; ++k) s = __fadd_rn(s, part[k]);
  return s;
}

// count allocation of the whole batch (reference utils/ops.py:385-432): thread b of the workgroup = cloud b
// (B <= blockDim.x), every thread of the workgroup must call it (whole-batch early exit by __syncthreads_and)
__device__ inline void alloc_counts_body(const float* w, const int* cap, int B, int nb, int M, int* counts) {
  const int b = threadIdx.x;
  const bool live = b < B;
  float p[kMaxBins], chosen[kMaxBins], capf[kMaxBins];
  int capi[kMaxBins];
#pragma unroll
  for (int t = 0; t < kMaxBins; ++t) {
    capi[t] = (live && t < nb) ? cap[b * nb + t] : 0;
    capf[t] = (float)capi[t];
    const float wt = (live && t < nb) ? w[b * nb + t] : 0.f;
    p[t] = __fadd_rn(__fmul_rn(wt, capf[t]), 1e-10f);
    chosen[t] = 0.f;
  }
  for (int round = 0; round < nb; ++round) {
    const float s = short_row_sum(p, nb);
#pragma unroll
    for (int t = 0; t < kMaxBins; ++t) p[t] = __fdiv_rn(p[t], s);
    const float left = __fsub_rn((float)M, short_row_sum(chosen, nb));
    const int done = (!live) || (left == 0.f);
    if (__syncthreads_and(done)) break;  // whole-batch early exit (utils/ops.py:409)
#pragma unroll
    for (int t = 0; t < kMaxBins; ++t) {
      float c = __fadd_rn(chosen[t], __fmul_rn(p[t], left));
      const bool sat = c >= capf[t];
      chosen[t] = sat ? capf[t] : c;
      p[t] = __fmul_rn(p[t], sat ? 0.f : 1.f);
    }
  }
  if (!live) return;
  int k[kMaxBins];
  int total = 0, best = 0;
  long bestv = 0;
#pragma unroll
  for (int t = 0; t < kMaxBins; ++t) {
    k[t] = (t < nb) ? (int)chosen[t] : 0;
    total += k[t];
  }
  for (int t = 0; t < nb; ++t) {
    const long room = (long)capi[t] - k[t];
    if (t == 0 || room > bestv) {
      bestv = room;
      best = t;
    }
  }
  for (int t = 0; t < nb; ++t) counts[b * nb + t] = k[t] + (t == best ? (M - total) : 0);
}

// B <= 64: ONE wave runs the serial allocation (lane = cloud) and the whole-batch exit is a wave vote -- no
// workgroup barrier inside the rounds (each costs more than a round's arithmetic).  The other waves just return;
// the caller synchronises afterwards.  Same arithmetic as alloc_counts_body.
// NB: the bin count when it is known at compile time (every loop unrolls: no dynamic indexing of the per-bin
// registers), 0 = take `nb` at run time.  Same operations in the same order either way.
template <int NB>
__device__ __forceinline__ float short_row_sum_t(const float (&a)[kMaxBins], int n_rt) {
  const int n = NB ? NB : n_rt;
  float part[4] = {0.f, 0.f, 0.f, 0.f};
  const int q = n >> 2;
#pragma unroll
  for (int i = 0; i < kMaxBins / 4; ++i)
    if (i < q) {
#pragma unroll
      for (int k = 0; k < 4; ++k) part[k] = __fadd_rn(part[k], a[4 * i + k]);
    }
#pragma unroll
  for (int i = 0; i < kMaxBins; ++i)
    if (i >= 4 * q && i < n) part[0] = __fadd_rn(part[0], a[i]);
  float s = part[0];
#pragma unroll
  for (int k = 1; k < 4; ++k) s = __fadd_rn(s, part[k]);
  return s;
}

template <int NB>
__device__ inline void alloc_counts_wave(const float* w, const int* cap, int B, int nb_rt, int M, int* counts) {
  if (threadIdx.x >= 64) return;
  const int nb = NB ? NB : nb_rt;
  const int b = threadIdx.x;
  const bool live = b < B;
  float p[kMaxBins], chosen[kMaxBins], capf[kMaxBins];
  int capi[kMaxBins];
#pragma unroll
  for (int t = 0; t < kMaxBins; ++t) {
    capi[t] = (live && t < nb) ? cap[b * nb + t] : 0;
    capf[t] = (float)capi[t];
    const float wt = (live && t < nb) ? w[b * nb + t] : 0.f;
    p[t] = __fadd_rn(__fmul_rn(wt, capf[t]), 1e-10f);
    chosen[t] = 0.f;
  }
#pragma unroll
  for (int round = 0; round < kMaxBins; ++round) {
    if (round >= nb) break;
    const float s = short_row_sum_t<NB>(p, nb);
#pragma unroll
    for (int t = 0; t < kMaxBins; ++t) p[t] = __fdiv_rn(p[t], s);
    const float left = __fsub_rn((float)M, short_row_sum_t<NB>(chosen, nb));
    const int done = (!live) || (left == 0.f);
    if (__all(done)) break;  // whole-batch early exit (utils/ops.py:409)
#pragma unroll
    for (int t = 0; t < kMaxBins; ++t) {
      float c = __fadd_rn(chosen[t], __fmul_rn(p[t], left));
      const bool sat = c >= capf[t];
      chosen[t] = sat ? capf[t] : c;
      p[t] = __fmul_rn(p[t], sat ? 0.f : 1.f);
    }
  }
  if (!live) return;
  int k[kMaxBins];
  int total = 0, best = 0;
  long bestv = 0;
#pragma unroll
  for (int t = 0; t < kMaxBins; ++t) {
    k[t] = (t < nb) ? (int)chosen[t] : 0;
    total += k[t];
  }
#pragma unroll
  for (int t = 0; t < kMaxBins; ++t) {
    const long room = (long)capi[t] - k[t];
    if (t < nb && (t == 0 || room > bestv)) {
      bestv = room;
      best = t;
    }
  }
#pragma unroll
  for (int t = 0; t < kMaxBins; ++t)
    if (t < nb) counts[b * nb + t] = k[t] + (t == best ? (M - total) : 0);
}

// The same allocation with EIGHT lanes per cloud (lane = 8 b + t owns bin t; B <= blockDim.x / 8): the serial
// version is one long dependent chain per cloud (six IEEE divisions per round on one wave); here a round is one
// division per lane and two 8-lane gathers.  Every sum is formed by short_row_sum on the gathered row, in every
// lane alike: the same arithmetic in the same order, the same integers.  Every thread of the workgroup must call it.
__device__ inline void alloc_counts_lanes(const float* w, const int* cap, int B, int nb, int M, int* counts) {
  const int tid = threadIdx.x, b = tid >> 3, t = tid & 7, lane = tid & 63, g0 = lane & ~7;
  const bool live = b < B;
  const bool mine = live && t < nb;
  const int capi = mine ? cap[b * nb + t] : 0;
  const float capf = (float)capi;
  float p = __fadd_rn(__fmul_rn(mine ? w[b * nb + t] : 0.f, capf), 1e-10f);
  float chosen = 0.f;
  auto row = [&](float v, float (&out)[kMaxBins]) {
#pragma unroll
    for (int u = 0; u < kMaxBins; ++u) out[u] = __shfl(v, g0 + u, 64);
  };
  float r[kMaxBins];
  for (int round = 0; round < nb; ++round) {
    row(p, r);
    const float s = short_row_sum(r, nb);
    p = __fdiv_rn(p, s);
    row(chosen, r);
    const float left = __fsub_rn((float)M, short_row_sum(r, nb));
    const int done = (!live) || (left == 0.f);
    if (__syncthreads_and(done)) break;  // whole-batch early exit (utils/ops.py:409)
    const float c = __fadd_rn(chosen, __fmul_rn(p, left));
    const bool sat = c >= capf;
    chosen = sat ? capf : c;
    p = __fmul_rn(p, sat ? 0.f : 1.f);
  }
  const int k = (t < nb) ? (int)chosen : 0;
  int total = 0, best = 0;
  long bestv = 0;
#pragma unroll
  for (int u = 0; u < kMaxBins; ++u) {
    const int ku = __shfl(k, g0 + u, 64), cu = __shfl(capi, g0 + u, 64);
    total += ku;
    const long room = (long)cu - ku;
    if (u < nb && (u == 0 || room > bestv)) {
      bestv = room;
      best = u;
    }
  }
  if (mine) counts[b * nb + t] = k + (t == best ? (M - total) : 0);
}

}  // namespace samble
