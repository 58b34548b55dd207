// Fused Gram + top-K kNN on the bf16 matrix cores with split fp32 operands (tri_dev.h).
//
// Same selection as knn_stream_kernel (knn_stream.hip): the accumulator of lane (i, h) holds
// G[j][i] - |b_j|^2/2 for 16 keys of query i, w = |a_i|^2/2 - acc is the ranking key, candidates that
// pass the running bound wait in a per-lane LDS ring, per-lane sorted K-lists of packed doubles
// (w bits | index), halves merged at the end.  What changes with the bf16 MFMA:
//   * the Gram product is 48 MFMAs of 32 cycles per tile instead of 64 of 64, and the vector ALU is free
//     while they run -- the kernel becomes bound by the selection's vector work, so
//   * the ring is drained by a FIXED number of branch-free insertion steps per tile (the compiler can
//     schedule them around the MFMAs; no workgroup-wide drain loops except on ring overflow), and
//   * key tiles arrive as operand images by LDS-DMA (no staging registers: the 96 registers of the
//     query operand and the 64 of the K-list leave none to spare).
#include <type_traits>

#include "tri_dev.h"

extern "C" void samble_time_begin(int, hipStream_t);
extern "C" void samble_time_end(int, hipStream_t);

namespace samble {

constexpr int kCapT = 32;  // ring slots per lane (power of two); a tile adds at most 16
int g_knn_tri_steps = 205;  // 1..4: fixed insertion steps per tile; 8..99: step budget (tile t gets budget / (t + 1));
                            // 101..: timing ablations; >= 200: per-wave drain down to (value - 200) entries [default]
int g_knn_tri = 1;         // 0: use the fp32-MFMA stream kernel

// channel-major fp32 (B, 128, N) -> RM operand image of the points (rows = points, contraction = channels)
__global__ __launch_bounds__(256) void tri_split_cm_kernel(const float* __restrict__ x, long bs, int N,
                                                           char* __restrict__ img_all) {
  const int tile = blockIdx.x, b = blockIdx.y, ntiles = gridDim.x, tid = threadIdx.x;
  const float* xb = x + (long)b * bs;
  char* img = img_all + ((long)b * ntiles + tile) * kTriTile;
  for (int e = tid; e < 512; e += 256) {
    const int r = e & 31, g = e >> 5, n = tile * 32 + r;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (n < N) ? xb[(long)(8 * g + i) * N + n] : 0.f;
    const Tri t = tri_split8(v);
    *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 0)) = t.h;
    *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 1)) = t.m;
    *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 2)) = t.l;
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

// ABL (timing-only ablations, wrong results): 1 = no matrix products, 2 = no insertions
// STEPS > 0: that many insertion steps per tile.  STEPS == 0, budget >= 200 (default): every tile each WAVE
// inserts until its fullest ring holds at most budget - 200 entries (lanes insert in lockstep, so a step is
// well used only while most lanes have a candidate: measured optimum 4-6 left over; a workgroup-wide
// drain loop or a fixed per-tile count both cost ~10 %).  STEPS == 0, budget < 200: budget / (t + 1) steps
// in tile t (a new key enters a K-list that has seen n keys with probability ~K/n).
template <int KN, int STEPS, int ABL = 0>
__global__ __launch_bounds__(512, 2) void knn_tri_kernel(const char* __restrict__ Qimg, int Nq,
                                                         const char* __restrict__ Kimg, int Nk,
                                                         const float* __restrict__ qnorm,
                                                         const float* __restrict__ knorm, int* __restrict__ idx_out,
                                                         float* __restrict__ d2_out, int budget) {
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

  auto glds = [&](int t, int buf) {
    const char* gt = Kb + (long)((ABL & 4) ? 0 : min(t, ntiles - 1)) * kTriTile;
    char* lt = smem_c + buf * kTriTile;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gt + (tid + 512 * k) * 16),
                                       (__attribute__((address_space(3))) void*)(lt + (wave * 64 + 512 * k) * 16), 16, 0,
                                       0);
  };
  glds(0, 0);

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
  const float half_an = 0.5f * qnorm[(long)b * Nq + qrow];
  if (tid < 32) bns[tid] = (tid < Nk) ? knb[tid] : 0.f;
  if (tid < 2) vote[tid] = 0;

  double L[KN];
#pragma unroll
  for (int s = 0; s < KN; ++s) L[s] = __builtin_huge_val();
  // A candidate passes when acc >= cut, cut a shade BELOW |a|^2/2 - bound: the filter lets through a
  // superset of {w <= bound} whatever the rounding of the two subtractions (an extra candidate only costs
  // an insertion that falls off the list); w itself is formed when a candidate is inserted.
  float cut = -__builtin_huge_valf();
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
    cut = c - fabsf(c) * 0x1p-21f - 0x1p-100f;
  };
  __syncthreads();  // tile 0 and its norms have landed

  // The two waves of a SIMD (w and w + 4) run the tile's two phases in opposite order -- insertion steps
  // (vector ALU) first in one, Gram product (matrix pipe) first in the other -- so the pipes overlap
  // although every wave issues in order and the workgroup meets at a barrier every tile.
  const bool mfma_first = wave < 4;
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1, nxt = cur ^ 1;
    const int j0 = t * 32;
    if (!(ABL & 16)) glds(t + 1, nxt);  // buffer nxt was last read one iteration ago
    const int jn = j0 + 32 + (tid & 31);
    const float nb = (jn < Nk) ? knb[jn] : 0.f;
    auto drain_steps = [&]() {
      if (!(ABL & 2)) {
        if (STEPS > 0) {
#pragma unroll
          for (int s = 0; s < STEPS; ++s) insert_step();
        } else {
          if (budget >= 200) {  // adaptive: this wave inserts until its fullest ring is down to budget - 200 entries
            const int keep = budget - 200;
            while (__any(tail - head > keep)) insert_step();
          } else {
            const int n = max(min(budget / (t + 1), 32), 1);
            for (int s = 0; s < n; ++s) insert_step();
          }
        }
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
      if (ABL & 1) acc[ks] += __uint_as_float(a.h[0] ^ bq.l[1] ^ a.m[1] ^ a.l[2]);
      else acc = mfma_tri(a, bq, acc);
    }
    const bool tail_tile = j0 + 32 > Nk;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      bool pass = acc[r] >= cut;
      if (tail_tile) pass = pass && (j0 + crow(r, h) < Nk);
      const int slot = (tail & (kCapT - 1)) * NT + tid;
      if (!(ABL & 8)) {
        qa[slot] = acc[r];
        qj[slot] = (unsigned short)(16 * t + r);
      } else if (acc[r] == 12345.f) qa[slot] = 0.f;
      tail += pass ? 1 : 0;
    }
    if (mfma_first) {
      __builtin_amdgcn_sched_barrier(0);
      drain_steps();
    }
    if (ABL & 2) head = tail;
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
  auto kern = knn_tri_kernel<KN, 0>;
  switch (g_knn_tri_steps) {
    case 3: kern = knn_tri_kernel<KN, 3>; break;
    case 1: kern = knn_tri_kernel<KN, 1>; break;
    case 2: kern = knn_tri_kernel<KN, 2>; break;
    case 4: kern = knn_tri_kernel<KN, 4>; break;
    case 101: kern = knn_tri_kernel<KN, 3, 1>; break;
    case 102: kern = knn_tri_kernel<KN, 3, 2>; break;
    case 103: kern = knn_tri_kernel<KN, 3, 3>; break;
    case 107: kern = knn_tri_kernel<KN, 3, 7>; break;
    case 111: kern = knn_tri_kernel<KN, 3, 11>; break;
    case 115: kern = knn_tri_kernel<KN, 3, 15>; break;
    case 119: kern = knn_tri_kernel<KN, 3, 19>; break;
    case 116: kern = knn_tri_kernel<KN, 3, 16>; break;
    default: break;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess) return (int)e;
  samble_time_begin(4, s);
  hipLaunchKernelGGL(kern, dim3((Nq + 255) / 256, B), dim3(NT), lds, s, qimg, Nq, kimg, Nk, qnorm, knorm, idx, d2,
                     g_knn_tri_steps);
  samble_time_end(4, s);
  return (int)hipGetLastError();
}

}  // namespace samble

using namespace samble;

extern "C" int samble_knn_tri_enabled() { return g_knn_tri; }
extern "C" __attribute__((visibility("default"))) void samble_knn_tri_config(int enabled, int steps) {
  g_knn_tri = enabled;
  if (steps > 0) g_knn_tri_steps = steps;
}

// image bytes for one point set of a (B, 128, N) cloud batch
extern "C" size_t samble_knn_tri_image_bytes(int B, int N) { return (size_t)B * ((N + 31) / 32) * kTriTile; }

extern "C" int samble_launch_tri_split_cm(const float* x, long bs, int B, int N, void* img, hipStream_t s) {
  hipLaunchKernelGGL(tri_split_cm_kernel, dim3((N + 31) / 32, B), dim3(256), 0, s, x, bs, N, (char*)img);
  return (int)hipGetLastError();
}

// C = 128, K in {16, 32}
extern "C" int samble_launch_knn_tri(const void* qimg, int Nq, const void* kimg, int Nk, int B, int K, const float* qnorm,
                                     const float* knorm, int* idx, float* d2, hipStream_t s) {
  if (K == 32) return launch_knn_tri<32>((const char*)qimg, Nq, (const char*)kimg, Nk, B, qnorm, knorm, idx, d2, s);
  if (K == 16) return launch_knn_tri<16>((const char*)qimg, Nq, (const char*)kimg, Nk, B, qnorm, knorm, idx, d2, s);
  return -22;
}
