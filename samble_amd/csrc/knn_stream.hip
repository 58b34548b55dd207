// Fused Gram + top-K kNN, software-pipelined (reference utils/ops.py:35-43: cdist + topk).
//
// Same algorithm and the same exact (w, index) ordering as knn_fused_kernel (knn.hip) -- the
// accumulator of lane (i, h) holds G[j][i] - |b_j|^2/2 for 16 keys of query i, w = |a_i|^2/2 - acc
// is the ranking key, per-lane sorted K-lists of packed doubles (w bits | index), bound =
// min(own K-th, max of the two halves' ceil(K/2)-th), halves merged at the end -- but scheduled so
// for this chip:
//   * the candidate filter is branch-free (unconditional ring-buffer write, tail += pass);
//   * candidates wait in a per-lane LDS ring; lanes insert in lockstep, so the rings are drained on a
//     workgroup-wide vote (a ring could overflow) and only down to a few entries: an insertion step is
//     well used while most lanes still have candidates, the stragglers wait for the next vote;
//   * workgroup = 8 waves = 256 queries (one per CU, two waves per SIMD): a cloud's key tiles are
//     staged half as often as with 128-query workgroups.
// fp32 MFMA and VALU do not overlap on gfx950, so nothing is gained by issuing the next tile's MFMAs
// over the selection (tried: it only cost registers); the selection's VALU time is what is left.
#include <type_traits>

#include "samble_dev.h"

namespace samble {

constexpr int kCap = 32;     // ring slots per lane (power of two); a tile adds at most 16
constexpr int kKeepStream = 6;  // drain policy (see full_drain)

template <int KN>
__device__ __forceinline__ void insert_packed2(double (&L)[KN], double x) {
#pragma unroll
  for (int s = KN - 1; s > 0; --s) L[s] = fmin(L[s], fmax(L[s - 1], x));
  L[0] = fmin(L[0], x);
}

__device__ __forceinline__ double pack_wj2(float w, unsigned int j) {
  return __longlong_as_double(__double_as_longlong((double)w) | (long long)j);
}

template <int C, int KN, int NW>
__global__ __launch_bounds__(64 * NW, 2) void knn_stream_kernel(const float* __restrict__ xq, long q_bs, int Nq,
                                                                const float* __restrict__ xk, long k_bs, int Nk,
                                                                const float* __restrict__ knorm,
                                                                int* __restrict__ idx_out, float* __restrict__ d2_out,
                                                                int g_keep) {
  constexpr int H = C / 2;            // MFMA steps; lane half h consumes channels H*h .. H*h+H-1
  constexpr int TILE = C * 32;        // floats per key tile, [channel][32 keys]
  constexpr int NT = 64 * NW;
  constexpr int LOADS = TILE / 4 / NT;
  constexpr int KH = (KN + 1) / 2;
  static_assert(LOADS >= 1 && TILE % (4 * NT) == 0, "tile must split evenly over the workgroup");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tiles = smem;                               // 2 x TILE
  float* bns = smem + 2 * TILE;                      // 2 x 32 key norms
  float* qw = bns + 64;                              // kCap x NT
  unsigned short* qj = reinterpret_cast<unsigned short*>(qw + kCap * NT);

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int i = chunk * (32 * NW) + wave * 32 + lo;
  const bool ivalid = i < Nq;
  const float* xkb = xk + (long)b * k_bs;
  const float* knb = knorm + (long)b * Nk;

  float q[H];
  float an = 0.f;
#pragma unroll
  for (int kk = 0; kk < H; ++kk) {
    q[kk] = xq[(long)b * q_bs + (long)(H * h + kk) * Nq + min(i, Nq - 1)];
    an = fmaf(q[kk], q[kk], an);
  }
  an += wave_xor32(an);
  const float half_an = 0.5f * an;

  double L[KN];
#pragma unroll
  for (int s = 0; s < KN; ++s) L[s] = __builtin_huge_val();
  float thr = __builtin_huge_valf();
  int head = 0, tail = 0;  // ring positions of this lane (monotonic; slot = position & (kCap-1))

  const bool vec = (Nk & 3) == 0;
  f32x4 stage[LOADS];
  float stage_bn = 0.f;
  auto issue = [&](int j0, auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;  // whole tile inside the key set and 16-byte aligned rows
#pragma unroll
    for (int it = 0; it < LOADS; ++it) {
      const int e = tid + NT * it;
      const int c = e >> 3, p4 = (e & 7) * 4;
      const float* src = xkb + (long)c * Nk + j0 + p4;
      if (FULL) {
        stage[it] = *reinterpret_cast<const f32x4*>(src);
      } else {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (vec && j0 + p4 + 3 < Nk) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (j0 + p4 + u < Nk) v[u] = src[u];
        }
        stage[it] = v;
      }
    }
    const int jn = j0 + (tid & 31);
    stage_bn = FULL ? knb[jn] : ((jn < Nk) ? knb[jn] : 0.f);
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int it = 0; it < LOADS; ++it) {
      const int e = tid + NT * it;
      *reinterpret_cast<f32x4*>(tiles + buf * TILE + (e >> 3) * 32 + (e & 7) * 4) = stage[it];
    }
    bns[buf * 32 + (tid & 31)] = stage_bn;  // every wave writes the same 32 values
  };
  auto product = [&](int buf) {
    const float* xs = tiles + buf * TILE + (H * h) * 32 + lo;
    f32x16 acc = zero16();
#pragma unroll
    for (int kk = 0; kk < H; ++kk) acc = mfma32(xs[kk * 32], q[kk], acc);
    return mfma32(h == 0 ? bns[buf * 32 + lo] : 0.f, h == 0 ? -0.5f : 0.f, acc);
  };
  auto insert_step = [&]() {
    const bool valid = head < tail;
    const int slot = (head & (kCap - 1)) * NT + tid;
    const double xd = valid ? pack_wj2(qw[slot], qj[slot]) : __builtin_huge_val();
    insert_packed2<KN>(L, xd);
    head += valid ? 1 : 0;
  };
  auto update_thr = [&]() {
    const double mid = L[KH - 1];
    const double pmid = __shfl_xor(mid, 32, 64);
    const double lim = fmin(L[KN - 1], fmax(mid, pmid));
    thr = (float)lim;  // the index bits are far below half a float ulp: this is exactly lim's w
  };
  // Lanes insert in lockstep, so a step is only well used while most lanes still have candidates:
  // drain down to `keep` entries in the fullest ring, not to empty (the rest waits for the next vote).
  auto full_drain = [&](int keep) {
    if (keep < 0) {  // timing ablation: no insertions at all (wrong results)
      head = tail;
      return;
    }
    while (__any(tail - head > keep)) insert_step();
    update_thr();
  };

  const int ntiles = (Nk + 31) / 32;
  using T = std::true_type;
  using F = std::false_type;
  issue(0, F{});
  commit(0);
  __syncthreads();

  // one tile: Gram product (64 + 1 MFMAs), branch-free candidate filter, staging of tile t+1, vote.
  // (A software pipeline that issued tile t+1's MFMAs over this filter bought nothing: fp32 MFMA and
  // VALU do not overlap on gfx950, and the second accumulator pushed the kernel into scratch spills.)
  auto body = [&](int t, int cur, auto fast_c) {
    constexpr bool FAST = decltype(fast_c)::value;
    const int nxt = cur ^ 1;
    const int j0 = t * 32;
    if (FAST) issue(j0 + 32, T{});
    else if (t + 1 < ntiles) issue(j0 + 32, F{});
    const f32x16 acc = product(cur);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + crow(r, h);
      const float w = fmaxf(half_an - acc[r], 0.f);
      bool pass = w <= thr;
      if (!FAST) pass = pass && (j < Nk);
      const int slot = (tail & (kCap - 1)) * NT + tid;
      qw[slot] = w;
      qj[slot] = (unsigned short)j;
      tail += pass ? 1 : 0;
    }
    if (FAST || t + 1 < ntiles) commit(nxt);
    // the tile barrier doubles as the overflow vote: a ring may take 16 more entries next tile
    if (__syncthreads_or(tail - head > kCap - 16)) full_drain(g_keep);
  };
  // fast iterations: tiles t and t+1 are full tiles (unguarded loads, no index checks)
  const int n_fast = vec ? max(Nk / 32 - 1, 0) : 0;
  int t = 0, cur = 0;
  for (; t < n_fast; ++t) {
    body(t, cur, T{});
    cur ^= 1;
  }
  for (; t < ntiles; ++t) {
    body(t, cur, F{});
    cur ^= 1;
  }
  full_drain(g_keep < 0 ? -1 : 0);

  // merge the two halves of every query through LDS (the whole dynamic region is free now)
  double* mg = reinterpret_cast<double*>(smem);
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

template <int C, int KN, int NW>
static int launch_stream(const float* xq, long q_bs, int Nq, const float* xk, long k_bs, int Nk, int B,
                         const float* knorm, int* idx, float* d2, hipStream_t s) {
  constexpr int NT = 64 * NW;
  size_t lds = (size_t)(2 * C * 32 + 64 + kCap * NT) * 4 + (size_t)kCap * NT * 2;
  const size_t merge = (size_t)KN * NT * 8;
  if (merge > lds) lds = merge;
  auto kern = knn_stream_kernel<C, KN, NW>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess) return (int)e;
  Timed timed(kT_knn, s);
  hipLaunchKernelGGL(kern, dim3((Nq + 32 * NW - 1) / (32 * NW), B), dim3(NT), lds, s, xq, q_bs, Nq, xk, k_bs, Nk, knorm,
                     idx, d2, kKeepStream);
  return (int)hipGetLastError();
}

}  // namespace samble

using namespace samble;

// C in {64,128}, K in {16,32}; 256-query workgroups when they still fill the chip, else 128-query ones
extern "C" int samble_launch_knn_stream(const float* xq, long q_bs, int Nq, const float* xk, long k_bs, int Nk, int B,
                                        int C, int K, const float* knorm, int* idx, float* d2, hipStream_t s) {
  const bool big = (long)B * ((Nq + 255) / 256) >= 200;
#define SAMBLE_KS(CC, KK)                                                                                       \
  if (C == CC && K == KK)                                                                                       \
    return big ? launch_stream<CC, KK, 8>(xq, q_bs, Nq, xk, k_bs, Nk, B, knorm, idx, d2, s)                     \
               : launch_stream<CC, KK, 4>(xq, q_bs, Nq, xk, k_bs, Nk, B, knorm, idx, d2, s);
  SAMBLE_KS(128, 32)
  SAMBLE_KS(128, 16)
  SAMBLE_KS(64, 32)
  SAMBLE_KS(64, 16)
#undef SAMBLE_KS
  return -22;
}
