// Fused Gram + top-K kNN on the bf16 matrix cores with split fp32 operands (tri_dev.h).
//
// Same selection as knn_stream_kernel (knn_stream.hip): the accumulator of lane (i, h) holds
// G[j][i] - |b_j|^2/2 for 16 keys of query i, w = |a_i|^2/2 - acc is the ranking key, candidates that
// pass the running bound wait in a per-lane LDS ring, per-lane sorted K-lists of packed doubles
// (w bits | index), halves merged at the end.  What is specific to this kernel:
//   * the Gram product is 48 bf16 MFMAs of 32 cycles per tile and the vector ALU is free while they run,
//     so the selection's vector work is what has to shrink: a SEED PASS over all key tiles with the
//     leading partial product only (h x h planes: 8 MFMAs per tile, 1/6 of the exact product) gives every
//     query a proven lower bound of its K-th best accumulator value BEFORE the exact pass starts:
//     each half-lane keeps the ceil(K/2) largest of its per-tile maxima (one sorted-insertion per tile);
//     T = min over the two halves of the ceil(K/2)-th largest: at least K keys have an approximate value
//     >= T, hence an exact value >= T - e, e = bound of |approximate - exact| (bf16 rounding of both
//     operands: 2^-7 |a||b| by Cauchy-Schwarz, taken with max_j |b_j|).  The exact pass starts with its
//     filter at T - e instead of -inf: ~30 insertions per lane instead of ~143 (370 in lockstep);
//   * the points are CENTRED on the query set's mean first (the reference does the same, utils/ops.py:23-25):
//     ranking is shift-invariant in exact arithmetic, and the centred Gram form does not cancel;
//   * key tiles arrive as operand images by LDS-DMA (no staging registers: the 96 registers of the
//     query operand and the 64 of the K-list leave none to spare).
#include <type_traits>

#include "tri_dev.h"

namespace samble {

constexpr int kCapT = 32;   // ring slots per lane (power of two, >= 16: a tile adds at most 16)
constexpr int kKeepT = 8;   // extra insertion steps (all lanes in lockstep) only while a ring of the wave holds more
constexpr int kSeedDepth = 6;            // LDS ring of the seed pass: tiles of the h plane (8 KB each), 2 used + 4 ahead
constexpr int kSeedTile = kTriTile / 3;  // bytes of one plane of a tile
// |approximate - exact| <= kSeedErr |a| max_j |b_j|: bf16 rounding of both operands (2 x 2^-8 + 2^-16, summed over
// the channels by Cauchy-Schwarz) + fp32 accumulation of either product (< 2^-15); 0.0085 leaves 8 % of slack
constexpr float kSeedErr = 0.0085f;

// channel-major fp32 (B, 128, N) -> RM operand image of the CENTRED points (rows = points, contraction =
// channels) and their squared norms (the same rounded differences that the image holds, fixed order)
__global__ __launch_bounds__(256) void tri_split_cm_kernel(const float* __restrict__ x, long bs, int N,
                                                           const float* __restrict__ mean,
                                                           char* __restrict__ img_all, float* __restrict__ norm_all) {
  __shared__ float part[16][33];
  const int tile = blockIdx.x, b = blockIdx.y, ntiles = gridDim.x, tid = threadIdx.x;
  const float* xb = x + (long)b * bs;
  const float* mb = mean + b * 128;
  char* img = img_all + ((long)b * ntiles + tile) * kTriTile;
  for (int e = tid; e < 512; e += 256) {
    const int r = e & 31, g = e >> 5, n = tile * 32 + r;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (n < N) ? xb[(long)(8 * g + i) * N + n] - mb[8 * g + i] : 0.f;
    const Tri t = tri_split8(v);
    *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 0)) = t.h;
    *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 1)) = t.m;
    *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 2)) = t.l;
    float p = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) p = fmaf(v[i], v[i], p);
    part[g][r] = p;
  }
  __syncthreads();
  if (tid < 32) {
    float sacc = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) sacc += part[g][tid];
    const int n = tile * 32 + tid;
    if (n < N) norm_all[(long)b * N + n] = sacc;
  }
}

// v_max_f64 / v_min_f64 as written: fmax / fmin come with a canonicalising `v_max_f64 x, x, x` per list slot under
// the kernel's IEEE mode (quieting signalling NaNs that cannot occur here: the packed values are finite or +inf) --
// three double-precision instructions per slot and insertion instead of two, plus register copies.
__device__ __forceinline__ double max_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double min_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int KN>
__device__ __forceinline__ void insert_packed_t(double (&L)[KN], double x) {
#pragma unroll
  for (int s = KN - 1; s > 0; --s) L[s] = min_f64(L[s], max_f64(L[s - 1], x));
  L[0] = min_f64(L[0], x);
}

__device__ __forceinline__ double pack_wj_t(float w, unsigned int j) {
  return __longlong_as_double(__double_as_longlong((double)w) | (long long)j);
}

// value held by the partner half-lane (lane ^ 32): v_permlane32_swap exchanges the upper half of its first
// register with the lower half of its second one -- no LDS crossbar, no wait (h = lane >> 5)
__device__ __forceinline__ unsigned partner32(unsigned v, int h) {
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return h ? r[0] : r[1];
}
__device__ __forceinline__ double partner64(double v, int h) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = partner32((unsigned)u, h), hi = partner32((unsigned)(u >> 32), h);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// SEED: run the seed pass (Nk * 4 bytes of key norms must fit the candidate ring's LDS).
// Workgroup = 8 waves = 256 queries, two waves per SIMD (w and w + 4).  The two waves of a SIMD run the tile's two
// phases in OPPOSITE order: the low wave [Gram product of tile t][filter + insertion step of tile t], the high
// wave [filter + insertion step of tile t-1][Gram product of tile t] -- each wave's vector work runs while its
// partner owns the matrix pipe, a wave needs one accumulator, and every wave does the same fixed amount of
// selection work per tile (no wave waits at the tile barrier for another one's drain).
template <int KN, bool SEED>
__global__ __launch_bounds__(512, 2) void knn_tri_kernel(const char* __restrict__ Qimg, int Nq,
                                                         const char* __restrict__ Kimg, int Nk,
                                                         const float* __restrict__ qnorm,
                                                         const float* __restrict__ knorm, int* __restrict__ idx_out,
                                                         float* __restrict__ d2_out) {
  constexpr int NT = 512, NW = 8;
  constexpr int KH = (KN + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  float* bns = reinterpret_cast<float*>(smem_c + 2 * kTriTile);  // 2 x 32 accumulator start values -|b_j|^2/2
  float* qa = bns + 68;                                          // kCapT x NT ring: accumulator values
  unsigned short* qj = reinterpret_cast<unsigned short*>(qa + kCapT * NT);  // ring: key codes 16 t + r

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int i = chunk * (32 * NW) + wave * 32 + lo;
  const bool ivalid = i < Nq;
  const int qrow = min(i, Nq - 1);
  const int qtiles = (Nq + 31) / 32, ntiles = (Nk + 31) / 32;
  const char* Kb = Kimg + (long)b * ntiles * kTriTile;
  const float* knb = knorm + (long)b * Nk;

#ifdef SAMBLE_KNN_STAMP
  long long st_t0 = clock64(), st_seed = 0, st_prod = 0, st_drain = 0, st_bar = 0, st_steps = 0, st_tmp;
  long long st_f = 0, st_a = 0, st_i = 0, st_c = 0, st_x;
#define XB() st_x = clock64()
#define XE(acc) acc += clock64() - st_x
#define STAMP_BEGIN() st_tmp = clock64()
#define STAMP_END(acc) acc += clock64() - st_tmp
#else
#define STAMP_BEGIN()
#define STAMP_END(acc)
#define XB()
#define XE(acc)
#endif
  // query operand: the h plane now (all the seed pass needs), the m and l planes after it -- 64 registers that
  // the seed loop would otherwise push into the accumulation registers
  u32x4 q[24];
  const u32x4* qp = reinterpret_cast<const u32x4*>(Qimg + ((long)b * qtiles + (qrow >> 5)) * kTriTile +
                                                   tri_rm_off(qrow & 31, h, 0));
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) q[3 * ks] = qp[192 * ks];
  const float an = qnorm[(long)b * Nq + qrow];
  const float half_an = 0.5f * an;

  // ---- seed pass: lower bound of this query's K-th best accumulator value --------------------------------
  float cut = -__builtin_huge_valf();
  if (SEED) {
    float* nrm = qa;                                   // all Nk key norms (the ring is not in use yet)
    float* red = bns;                                  // 8 partial maxima
    // h plane of tile t: chunk (g, r) of the plane sits at ((3 g) * 32 + r) * 16 in the image tile; thread
    // tid = 32 g + r fetches it, the plane lands compact (chunk tid at tid * 16)
    auto glds_h = [&](int t) {
      const char* gt = Kb + (long)min(t, ntiles - 1) * kTriTile + ((3 * (tid >> 5)) * 32 + (tid & 31)) * 16;
      char* lt = smem_c + (t % kSeedDepth) * kSeedTile + wave * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gt,
                                       (__attribute__((address_space(3))) void*)lt, 16, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < kSeedDepth - 2; ++t) glds_h(t);
    float bmax = 0.f;
    for (int j = tid; j < ntiles * 32; j += NT) {
      const float v = (j < Nk) ? knb[j] : 0.f;
      nrm[j] = -0.5f * v;  // the accumulator's start value
      bmax = fmaxf(bmax, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, o, 64));
    if (lane == 0) red[wave] = bmax;
    float G[KH];  // the KH largest per-tile maxima of this half-lane, descending
#pragma unroll
    for (int s = 0; s < KH; ++s) G[s] = -__builtin_huge_valf();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // one tile of the seed pass: leading partial product, maximum over the lane's 16 keys, sorted insertion
    auto seed_tile = [&](int t, auto masked_c) {
      constexpr bool MASKED = decltype(masked_c)::value;  // the last tile: padding keys past Nk must not count
      f32x16 acc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(nrm + t * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * g + e] = v4[e];
      }
      const u32x4* lp = reinterpret_cast<const u32x4*>(smem_c + (t % kSeedDepth) * kSeedTile) + 32 * h + lo;
#pragma unroll
#ifndef SAMBLE_KNN_SEEDABL
#define SAMBLE_KNN_SEEDABL 0
#endif
      for (int ks = 0; ks < 8; ++ks) {
        if (SAMBLE_KNN_SEEDABL & 1) acc[ks] += __uint_as_float(lp[64 * ks][0] ^ q[3 * ks][1]);
        else acc = mfma_bf(lp[64 * ks], q[3 * ks], acc);
      }
      if (MASKED) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * 32 + crow(r, h) >= Nk) acc[r] = -__builtin_huge_valf();
      }
      float gm = __builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), acc[2]);  // v_max3_f32
#pragma unroll
      for (int r = 3; r < 15; r += 2) gm = __builtin_fmaxf(__builtin_fmaxf(gm, acc[r]), acc[r + 1]);
      gm = fmaxf(gm, acc[15]);
      // sorted insertion into the descending list: G[s] <- max(G[s], min(G[s-1], gm)) = median(G[s-1], G[s], gm)
#pragma unroll
      for (int s = KH - 1; s > 0 && !(SAMBLE_KNN_SEEDABL & 2); --s) G[s] = __builtin_amdgcn_fmed3f(G[s - 1], G[s], gm);
      G[0] = fmaxf(G[0], gm);
    };
    // two FULL tiles per barrier; tiles t+2 .. t+5 are in flight or landed while t, t+1 are used
    const int nfull = Nk / 32;
    int t = 0;
    for (; t + 2 <= nfull; t += 2) {
      if (!(SAMBLE_KNN_SEEDABL & 4)) {
        glds_h(t + kSeedDepth - 2);  // their slots were read one iteration ago
        glds_h(t + kSeedDepth - 1);
      }
      seed_tile(t, std::false_type{});
      seed_tile(t + 1, std::false_type{});
      // tiles t+2, t+3 have landed for this wave once at most the two youngest DMAs remain
      asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // the remaining one or two tiles (the last one may hold padding keys); their DMAs were issued above
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (; t < ntiles; ++t) seed_tile(t, std::true_type{});
    float bm = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) bm = fmaxf(bm, red[w]);
    const float T = fminf(G[KH - 1], __shfl_xor(G[KH - 1], 32, 64));
    const float err = kSeedErr * sqrtf(an) * sqrtf(bm) * 1.0001f;
    const float c = T - err;
    cut = c - fabsf(c) * 0x1p-21f - 0x1p-100f;  // (-inf stays -inf: fewer than KH tiles per half)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // seed ring, norms, red are free
  }
#ifdef SAMBLE_KNN_STAMP
  st_seed = clock64() - st_t0;
#endif
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    q[3 * ks + 1] = qp[192 * ks + 32];
    q[3 * ks + 2] = qp[192 * ks + 64];
  }

  auto glds = [&](int t, int buf) {
    const char* gt = Kb + (long)min(t, ntiles - 1) * kTriTile;
    char* lt = smem_c + buf * kTriTile;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gt + (tid + 512 * k) * 16),
                                       (__attribute__((address_space(3))) void*)(lt + (wave * 64 + 512 * k) * 16), 16, 0,
                                       0);
  };
  glds(0, 0);
  if (tid < 64) bns[tid] = (tid < 32 && tid < Nk) ? knb[tid] : 0.f;  // |b_j|^2 as stored; halved and negated where it is read

  // The K-list of a query is ONE sorted list split by rank over its two half-lanes: lane (i, 0) keeps the KH best
  // (ranks 0 .. KH-1), lane (i, 1) ranks KH .. 2 KH - 1 (32 registers per lane instead of 64).  Packed doubles
  // (w bits | index: one v_max_f64 + v_min_f64 per slot, ties by index).  An insertion step handles one candidate
  // of the ROW (the low half's ring first): lane 0 inserts it, the element that falls off lane 0's list goes to
  // lane 1, which inserts it one step later (`pend`).  No merge at the end; the row's K-th best is lane 1's last.
  double L[KH];
#pragma unroll
  for (int s = 0; s < KH; ++s) L[s] = __builtin_huge_val();
  double pend = __builtin_huge_val();
  // A candidate passes when acc >= cut, cut a shade BELOW |a|^2/2 - bound: the filter lets through a
  // superset of {w <= bound} whatever the rounding of the two subtractions (an extra candidate only costs
  // an insertion that falls off the list); w itself is formed when a candidate is inserted.
  int head = 0, tail = 0;  // ring positions of this lane (monotonic; slot = position & (kCapT-1))

  auto insert_step = [&]() {
    const bool v_own = head < tail;
    const int slot = (head & (kCapT - 1)) * NT + tid;
    const unsigned code = qj[slot];
    const float w = fmaxf(half_an - qa[slot], 0.f);
    const unsigned j = (code >> 4) * 32 + crow(code & 15, h);
    const double x_own = v_own ? pack_wj_t(w, j) : __builtin_huge_val();
    const double x_p = partner64(x_own, h);                // the partner half-lane's candidate (+inf: none)
    const bool v_p = x_p < __builtin_huge_val();
    // lane 0 inserts the row's candidate (its own first), lane 1 the element that left lane 0 last step
    const double ins = h == 0 ? (v_own ? x_own : x_p) : pend;
    const double y = max_f64(L[KH - 1], ins);              // what leaves this lane's list
    insert_packed_t<KH>(L, ins);
    const double y_p = partner64(y, h);
    pend = h == 1 ? y_p : __builtin_huge_val();
    head += (h == 0 ? v_own : (v_own && !v_p)) ? 1 : 0;
  };
  auto update_cut = [&]() {
    // lane 1's last entry bounds the row's K-th best from above (exactly it, once `pend` has been inserted)
    const double mine = L[KH - 1];
    const double theirs = partner64(mine, h);
    const float thr = (float)(h == 1 ? mine : theirs);  // the index bits are far below half a float ulp
    const float c = half_an - thr;
    cut = fmaxf(cut, c - fabsf(c) * 0x1p-21f - 0x1p-100f);
  };
  // Gram product of the tile in buffer `buf` (norm slot `buf`): accumulator starts at -|b_j|^2/2 of its 16 keys
  auto products = [&](int buf) {
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(bns + buf * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[4 * g + e] = -0.5f * v4[e];
    }
    const u32x4* lp = reinterpret_cast<const u32x4*>(smem_c + buf * kTriTile + tri_rm_off(lo, h, 0));
    // operand reads two k-steps ahead of their MFMAs (the compiler's own placement: 44-51 cycles per MFMA, stamped
    // in attn_stats)
    tri_pipelined<8>([&](int ks) { return Tri{lp[192 * ks], lp[192 * ks + 32], lp[192 * ks + 64]}; },
                     [&](int ks, const Tri& a) {
                       const Tri bq = {q[3 * ks], q[3 * ks + 1], q[3 * ks + 2]};
                       acc = mfma_tri(a, bq, acc);
                     });
    return acc;
  };
  // filter of tile tt's accumulator into the ring + ONE insertion step (~0.8 candidates per row and tile arrive
  // against the one served) + the bound; extra steps only while a ring of the wave is filling up
  // Selection work of one tile (accumulator `acc` of tile tt):
  //   filter  d_r = acc_r - cut, sign bits collected by v_alignbit: two vector instructions per element, no
  //           per-element address arithmetic (~2.5 % of the elements pass after the seed);
  //   append  only the lanes that have passing elements copy them (value picked out of the 16 registers by a
  //           select tree -- no LDS round trip -- + key code) into their candidate ring, one per lane and
  //           iteration (typically 2-3 iterations per tile);
  //   insert  ONE insertion step (~0.8 candidates per row and tile arrive against the one served), more only
  //           while a ring of the wave is filling up; then the bound.
  const unsigned validbits_last = [&]() {  // bit (15 - r) set: key crow(r, h) of the LAST tile exists
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) m |= ((ntiles - 1) * 32 + crow(r, h) < Nk) ? (1u << (15 - r)) : 0u;
    return m;
  }();
  auto select = [&](const f32x16 acc, int tt) {
    XB();
    unsigned sb = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = acc[r] - cut;  // >= 0 (sign bit clear) <=> passes; exact in sign
      sb = __builtin_amdgcn_alignbit(sb, __float_as_uint(d), 31);  // sb = (sb << 1) | sign(d)
    }
    unsigned bits = ~sb & 0xFFFFu;  // bit (15 - r): element r passes
    if (tt == ntiles - 1) bits &= validbits_last;
    // room for this tile's candidates (at most 16 of them)
    while (__any(tail - head + (int)__popc(bits) > kCapT)) insert_step();
    XE(st_f);
    XB();
    while (__any(bits != 0)) {
      const int pos = __builtin_ctz(bits | 0x10000u);
      const int r = 15 - pos;  // (-1 for a lane without candidates: it writes nothing)
      // acc[r] by a select tree over the four bits of r
      const bool b0 = r & 1, b1 = r & 2, b2 = r & 4;
      const float e0 = b0 ? acc[1] : acc[0], e1 = b0 ? acc[3] : acc[2], e2 = b0 ? acc[5] : acc[4];
      const float e3 = b0 ? acc[7] : acc[6], e4 = b0 ? acc[9] : acc[8], e5 = b0 ? acc[11] : acc[10];
      const float e6 = b0 ? acc[13] : acc[12], e7 = b0 ? acc[15] : acc[14];
      const float f0 = b1 ? e1 : e0, f1 = b1 ? e3 : e2, f2 = b1 ? e5 : e4, f3 = b1 ? e7 : e6;
      const float v2[2] = {b2 ? f1 : f0, b2 ? f3 : f2};
      const float val = (r & 8) ? v2[1] : v2[0];
      if (bits != 0) {
        const int slot = (tail & (kCapT - 1)) * NT + tid;
        qa[slot] = val;
        qj[slot] = (unsigned short)(16 * tt + r);
        ++tail;
      }
      bits &= bits - 1;
    }
    XE(st_a);
    XB();
    // (a fixed step per tile beats purely adaptive drains, 262 vs 288-302 us: every wave does the same work.  One
    // every other tile: 246 against 236 us.  The step woven into products() behind the MFMAs: no change, 237-240 --
    // the two waves of a SIMD share its vector issue, and the other wave is in its selection phase then.)
    insert_step();
    while (__any(tail - head > kKeepT)) {
      insert_step();
#ifdef SAMBLE_KNN_STAMP
      ++st_steps;
#endif
    }
    XE(st_i);
    XB();
    update_cut();
    XE(st_c);
  };
  __syncthreads();  // tile 0 and its norms have landed

  const bool mfma_first = wave < 4;
  f32x16 acc;  // high waves: the accumulator of the previous tile, filtered at the start of the next iteration
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1, nxt = cur ^ 1;
    glds(t + 1, nxt);  // buffer nxt was last read one iteration ago
    // the next tile's 32 key norms go to LDS by DMA as well.  (As a register load consumed at the end of the iteration
    // the compiler moved it into wave 0's `tid < 32` branch and waited for it there with vmcnt(0) -- behind the three
    // tile pieces just issued: wave 0 sat out their whole latency at the top of every tile, and the other seven
    // waves waited for it at the barrier.)  Keys past Nk keep a stale slot: their candidates are masked (last tile).
    if (wave == 0) {
      const int jn = t * 32 + 32 + lane;
      if (lane < 32 && jn < Nk)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(knb + jn),
                                         (__attribute__((address_space(3))) void*)(bns + nxt * 32), 4, 0, 0);
    }
    STAMP_BEGIN();
    if (mfma_first) {
      acc = products(cur);
      __builtin_amdgcn_sched_barrier(0);
      STAMP_END(st_prod);
      STAMP_BEGIN();
      select(acc, t);
      STAMP_END(st_drain);
    } else {
      if (t > 0) select(acc, t - 1);
      __builtin_amdgcn_sched_barrier(0);
      STAMP_END(st_drain);
      STAMP_BEGIN();
      acc = products(cur);
      STAMP_END(st_prod);
    }
    STAMP_BEGIN();
    __syncthreads();
    STAMP_END(st_bar);
  }
  if (!mfma_first) select(acc, ntiles - 1);
#ifdef SAMBLE_KNN_STAMP
  const long long st_loop = clock64() - st_t0;
#endif
  while (__any(tail > head)) insert_step();
  insert_step();  // lane 1 takes in the last element handed over

#ifdef SAMBLE_KNN_STAMP
  if (d2_out && lane == 0) {  // diagnostic build: the distance output carries the stamps of (cloud, chunk, wave)
    float* o = d2_out + ((long)b * Nq + chunk * (32 * NW) + wave * 32) * KN;
    o[0] = (float)st_seed; o[1] = (float)st_prod; o[2] = (float)st_drain; o[3] = (float)st_bar;
    o[4] = (float)st_loop; o[5] = (float)(clock64() - st_t0); o[6] = (float)st_steps; o[7] = (float)tail;
    o[8] = (float)st_f; o[9] = (float)st_a; o[10] = (float)st_i; o[11] = (float)st_c;
  }
  if (d2_out) return;
#endif
  // ranks KH h .. KH h + KH - 1 of query i sit in lane (i, h): nearest first, ties by ascending index
  if (ivalid) {
    int* io = idx_out + ((long)b * Nq + i) * KN + KH * h;
    float* dout = d2_out ? d2_out + ((long)b * Nq + i) * KN + KH * h : nullptr;
#pragma unroll
    for (int s2 = 0; s2 < KH; ++s2) {
      if (KH * h + s2 < KN) {
        io[s2] = (int)(__double_as_longlong(L[s2]) & 0x1FFFFFFFll);
        if (dout) dout[s2] = 2.f * (float)L[s2];
      }
    }
  }
}

template <int KN>
static int launch_knn_tri(const char* qimg, int Nq, const char* kimg, int Nk, int B, const float* qnorm,
                          const float* knorm, int* idx, float* d2, hipStream_t s) {
  constexpr int NT = 512;
  const size_t lds = (size_t)2 * kTriTile + 68 * 4 + (size_t)kCapT * NT * 4 + (size_t)kCapT * NT * 2;
  const bool seed = (size_t)((Nk + 31) / 32) * 32 * 4 <= (size_t)kCapT * NT * 6;
  auto kern = seed ? knn_tri_kernel<KN, true> : knn_tri_kernel<KN, false>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess) return (int)e;
  Timed timed(kT_knn, s);
  hipLaunchKernelGGL(kern, dim3((Nq + 255) / 256, B), dim3(NT), lds, s, qimg, Nq, kimg, Nk, qnorm, knorm, idx, d2);
  return (int)hipGetLastError();
}

}  // namespace samble

using namespace samble;

extern "C" int samble_launch_cloud_mean(const float* x, long bs, int C, int N, int B, float* mean, hipStream_t s);

// image bytes for one point set of a (B, 128, N) cloud batch
extern "C" size_t samble_knn_tri_image_bytes(int B, int N) { return (size_t)B * ((N + 31) / 32) * kTriTile; }

// centred operand images and squared norms of the two point sets (xk == nullptr: the key set is the query set)
extern "C" int samble_launch_knn_tri_prep(const float* xq, long q_bs, int Nq, const float* xk, long k_bs, int Nk, int B,
                                          float* mean, void* qimg, void* kimg, float* qnorm, float* knorm,
                                          hipStream_t s) {
  Timed timed(kT_knn_prep, s);
  const int rc = samble_launch_cloud_mean(xq, q_bs, 128, Nq, B, mean, s);
  if (rc) return rc;
  if (!xk) {
    hipLaunchKernelGGL(tri_split_cm_kernel, dim3((Nk + 31) / 32, B), dim3(256), 0, s, xq, q_bs, Nk, mean, (char*)kimg,
                       knorm);
  } else {
    hipLaunchKernelGGL(tri_split_cm_kernel, dim3((Nq + 31) / 32, B), dim3(256), 0, s, xq, q_bs, Nq, mean, (char*)qimg,
                       qnorm);
    hipLaunchKernelGGL(tri_split_cm_kernel, dim3((Nk + 31) / 32, B), dim3(256), 0, s, xk, k_bs, Nk, mean, (char*)kimg,
                       knorm);
  }
  return (int)hipGetLastError();
}

// C = 128, K in {16, 32}
extern "C" int samble_launch_knn_tri(const void* qimg, int Nq, const void* kimg, int Nk, int B, int K, const float* qnorm,
                                     const float* knorm, int* idx, float* d2, hipStream_t s) {
  if (K == 32) return launch_knn_tri<32>((const char*)qimg, Nq, (const char*)kimg, Nk, B, qnorm, knorm, idx, d2, s);
  if (K == 16) return launch_knn_tri<16>((const char*)qimg, Nq, (const char*)kimg, Nk, B, qnorm, knorm, idx, d2, s);
  return -22;
}
