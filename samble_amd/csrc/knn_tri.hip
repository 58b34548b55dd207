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

constexpr int kCapT = 32;   // ring slots per lane (power of two); a tile adds at most 16
constexpr int kKeepT = 5;   // a wave inserts (all lanes in lockstep) until its fullest ring holds <= kKeepT entries
constexpr int kSeedDepth = 4;            // LDS ring of the seed pass: tiles of the h plane (8 KB each)
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

template <int KN>
__device__ __forceinline__ void insert_packed_t(double (&L)[KN], double x) {
#pragma unroll
  for (int s = KN - 1; s > 0; --s) L[s] = fmin(L[s], fmax(L[s - 1], x));
  L[0] = fmin(L[0], x);
}

__device__ __forceinline__ double pack_wj_t(float w, unsigned int j) {
  return __longlong_as_double(__double_as_longlong((double)w) | (long long)j);
}

// SEED: run the seed pass (Nk * 4 bytes of key norms must fit the candidate ring's LDS: Nk <= 16384)
template <int KN, bool SEED>
__global__ __launch_bounds__(512, 2) void knn_tri_kernel(const char* __restrict__ Qimg, int Nq,
                                                         const char* __restrict__ Kimg, int Nk,
                                                         const float* __restrict__ qnorm,
                                                         const float* __restrict__ knorm, int* __restrict__ idx_out,
                                                         float* __restrict__ d2_out) {
  constexpr int NT = 512, NW = 8;
  constexpr int KH = (KN + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  float* bns = reinterpret_cast<float*>(smem_c + 2 * kTriTile);  // 2 x 32 key norms
  int* vote = reinterpret_cast<int*>(bns + 64);                  // 2 overflow flags (tile parity), padded to 16 B
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

  u32x4 q[24];
  {
    const u32x4* qp = reinterpret_cast<const u32x4*>(Qimg + ((long)b * qtiles + (qrow >> 5)) * kTriTile +
                                                     tri_rm_off(qrow & 31, h, 0));
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      q[3 * ks] = qp[192 * ks];
      q[3 * ks + 1] = qp[192 * ks + 32];
      q[3 * ks + 2] = qp[192 * ks + 64];
    }
  }
  const float an = qnorm[(long)b * Nq + qrow];
  const float half_an = 0.5f * an;

  // ---- seed pass: lower bound of this query's K-th best accumulator value --------------------------------
  float cut0 = -__builtin_huge_valf();
  if (SEED) {
    float* nrm = qa;                                   // all Nk key norms (the ring is not in use yet)
    float* red = bns;                                  // 8 partial maxima
    // h plane of tile t: chunk (g, r) of the plane sits at ((3 g) * 32 + r) * 16 in the image tile; thread
    // tid = 32 g + r fetches it, the plane lands compact (chunk tid at tid * 16)
    auto glds_h = [&](int t) {
      const char* gt = Kb + (long)min(t, ntiles - 1) * kTriTile + ((3 * (tid >> 5)) * 32 + (tid & 31)) * 16;
      char* lt = smem_c + (t & (kSeedDepth - 1)) * kSeedTile + wave * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gt,
                                       (__attribute__((address_space(3))) void*)lt, 16, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < kSeedDepth - 1; ++t) glds_h(t);
    float bmax = 0.f;
    for (int j = tid; j < ntiles * 32; j += NT) {
      const float v = (j < Nk) ? knb[j] : 0.f;
      nrm[j] = v;
      bmax = fmaxf(bmax, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, o, 64));
    if (lane == 0) red[wave] = bmax;
    float G[KH];  // the KH largest per-tile maxima of this half-lane, descending
#pragma unroll
    for (int s = 0; s < KH; ++s) G[s] = -__builtin_huge_valf();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int t = 0; t < ntiles; ++t) {
      glds_h(t + kSeedDepth - 1);  // its slot was read one iteration ago
      f32x16 acc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(nrm + t * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * g + e] = -0.5f * v4[e];
      }
      const u32x4* lp = reinterpret_cast<const u32x4*>(smem_c + (t & (kSeedDepth - 1)) * kSeedTile) + 32 * h + lo;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) acc = mfma_bf(lp[64 * ks], q[3 * ks], acc);
      if ((t + 1) * 32 > Nk) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * 32 + crow(r, h) >= Nk) acc[r] = -__builtin_huge_valf();
      }
      float gm = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
#pragma unroll
      for (int r = 4; r < 16; r += 4) gm = fmaxf(gm, fmaxf(fmaxf(acc[r], acc[r + 1]), fmaxf(acc[r + 2], acc[r + 3])));
#pragma unroll
      for (int s = KH - 1; s > 0; --s) G[s] = fmaxf(G[s], fminf(G[s - 1], gm));
      G[0] = fmaxf(G[0], gm);
      // tile t+1 has landed for this wave once at most the two youngest DMAs (t+2, t+3) are in flight
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kSeedDepth - 2) : "memory");
    }
    float bm = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) bm = fmaxf(bm, red[w]);
    const float T = fminf(G[KH - 1], __shfl_xor(G[KH - 1], 32, 64));
    const float err = kSeedErr * sqrtf(an) * sqrtf(bm) * 1.0001f;
    const float c = T - err;
    cut0 = c - fabsf(c) * 0x1p-21f - 0x1p-100f;  // (-inf stays -inf: fewer than KH tiles per half)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // seed ring, norms, red are free
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
  if (tid < 32) bns[tid] = (tid < Nk) ? knb[tid] : 0.f;
  if (tid < 2) vote[tid] = 0;

  double L[KN];
#pragma unroll
  for (int s = 0; s < KN; ++s) L[s] = __builtin_huge_val();
  // A candidate passes when acc >= cut, cut a shade BELOW |a|^2/2 - bound: the filter lets through a
  // superset of {w <= bound} whatever the rounding of the two subtractions (an extra candidate only costs
  // an insertion that falls off the list); w itself is formed when a candidate is inserted.
  float cut = cut0;
  int head = 0, tail = 0;  // ring positions of this lane (monotonic; slot = position & (kCapT-1))

  auto insert_step = [&]() {
    const bool valid = head < tail;
    const int slot = (head & (kCapT - 1)) * NT + tid;
    const unsigned code = qj[slot];
    const float w = fmaxf(half_an - qa[slot], 0.f);
    const unsigned j = (code >> 4) * 32 + crow(code & 15, h);
    const double xd = valid ? pack_wj_t(w, j) : __builtin_huge_val();
    insert_packed_t<KN>(L, xd);
    head += valid ? 1 : 0;
  };
  auto update_cut = [&]() {
    const double mid = L[KH - 1];
    const double pmid = __shfl_xor(mid, 32, 64);
    const double lim = fmin(L[KN - 1], fmax(mid, pmid));
    const float thr = (float)lim;  // the index bits are far below half a float ulp: this is exactly lim's w
    const float c = half_an - thr;
    cut = fmaxf(cut0, c - fabsf(c) * 0x1p-21f - 0x1p-100f);
  };
  __syncthreads();  // tile 0 and its norms have landed

  // The two waves of a SIMD (w and w + 4) run the tile's two phases in opposite order -- insertion steps
  // (vector ALU) first in one, Gram product (matrix pipe) first in the other -- so the pipes overlap
  // although every wave issues in order and the workgroup meets at a barrier every tile.
  const bool mfma_first = wave < 4;
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1, nxt = cur ^ 1;
    const int j0 = t * 32;
    glds(t + 1, nxt);  // buffer nxt was last read one iteration ago
    const int jn = j0 + 32 + (tid & 31);
    const float nb = (jn < Nk) ? knb[jn] : 0.f;
    // every tile each WAVE inserts until its fullest ring holds at most kKeepT entries (lanes insert in
    // lockstep, so a step is well used only while most lanes have a candidate)
    auto drain_steps = [&]() {
      if (__any(tail - head > kKeepT)) {
        do insert_step();
        while (__any(tail - head > kKeepT));
        update_cut();
      }
    };
    if (!mfma_first) {
      drain_steps();
      __builtin_amdgcn_sched_barrier(0);
    }
    // accumulator starts at -|b_j|^2/2 of its 16 keys (rows crow(r, h))
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(bns + cur * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[4 * g + e] = -0.5f * v4[e];
    }
    const u32x4* lp = reinterpret_cast<const u32x4*>(smem_c + cur * kTriTile + tri_rm_off(lo, h, 0));
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const Tri a = {lp[192 * ks], lp[192 * ks + 32], lp[192 * ks + 64]};
      const Tri bq = {q[3 * ks], q[3 * ks + 1], q[3 * ks + 2]};
      acc = mfma_tri(a, bq, acc);
    }
    const bool tail_tile = j0 + 32 > Nk;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      bool pass = acc[r] >= cut;
      if (tail_tile) pass = pass && (j0 + crow(r, h) < Nk);
      const int slot = (tail & (kCapT - 1)) * NT + tid;
      qa[slot] = acc[r];
      qj[slot] = (unsigned short)(16 * t + r);
      tail += pass ? 1 : 0;
    }
    if (mfma_first) {
      __builtin_amdgcn_sched_barrier(0);
      drain_steps();
    }
    if (tid < 32) bns[nxt * 32 + tid] = nb;
    // the tile barrier doubles as the overflow vote (a ring may take 16 more entries next tile): one flag
    // per tile parity, set by any wave that has a full ring, cleared two tiles later
    if (__any(tail - head > kCapT - 16)) vote[cur] = 1;
    if (tid == 0) vote[nxt] = 0;
    __syncthreads();
    if (vote[cur]) {
      while (__any(tail - head > 4)) insert_step();
      update_cut();
    }
  }
  while (__any(tail > head)) insert_step();

  // merge the two halves of every query through LDS (the whole dynamic region is free now)
  double* mg = reinterpret_cast<double*>(smem_c);
  __syncthreads();
#pragma unroll
  for (int s = 0; s < KN; ++s) mg[s * NT + tid] = L[s];
  __syncthreads();
  if (h == 0 && ivalid) {
    int pa = 0, pb = 0;
    double va = mg[tid], vb = mg[tid + 32];
    int* io = idx_out + ((long)b * Nq + i) * KN;
    float* dout = d2_out ? d2_out + ((long)b * Nq + i) * KN : nullptr;
    for (int k = 0; k < KN; ++k) {
      const bool take = va <= vb;
      const double o = take ? va : vb;
      io[k] = (int)(__double_as_longlong(o) & 0x1FFFFFFFll);
      if (dout) dout[k] = 2.f * (float)o;
      if (take) {
        ++pa;
        va = (pa < KN) ? mg[pa * NT + tid] : __builtin_huge_val();
      } else {
        ++pb;
        vb = (pb < KN) ? mg[pb * NT + tid + 32] : __builtin_huge_val();
      }
    }
  }
}

template <int KN>
static int launch_knn_tri(const char* qimg, int Nq, const char* kimg, int Nk, int B, const float* qnorm,
                          const float* knorm, int* idx, float* d2, hipStream_t s) {
  constexpr int NT = 512;
  size_t lds = (size_t)2 * kTriTile + 68 * 4 + (size_t)kCapT * NT * 4 + (size_t)kCapT * NT * 2;
  const size_t merge = (size_t)KN * NT * 8;
  if (merge > lds) lds = merge;
  const bool seed = (size_t)((Nk + 31) / 32) * 32 * 4 <= (size_t)kCapT * NT * 4;
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
