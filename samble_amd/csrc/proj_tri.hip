// QKV projection forward and its input gradient on the bf16 matrix cores with split fp32 operands
// (tri_dev.h); same results contract as proj_fwd_kernel / proj_dx_kernel of proj.hip (fp32-equivalent
// products, other rounding).  dW: proj_dw_tri_kernel below (its contraction runs over the points: both
// operands are transposed through LDS and split in registers).
//
//   forward   qkv[n][o] = sum_c W[o][c] x[c][n]      A: W row image tile (32 outputs, LDS ring), B: the point's
//                                                    128 channels, read channel-major and split in registers
//   dx        dx[c][n]  = sum_o W[o][c] dqkv[n][o]   A: W transposed image tile (32 outputs = the contraction
//                                                    rows), B: 16 outputs of the point's gradient row per k-step,
//                                                    read in the image's element order and split in registers
// One workgroup = 8 waves = 256 points, two waves per SIMD; the 12 W tiles (295 KB as an image) stream through
// a ring of 4 LDS slots by LDS-DMA, three tiles ahead, every wait counted.
#include "tri_dev.h"

namespace samble {

constexpr int kPO = 384, kPTiles = kPO / 32, kPDepth = 4;
constexpr int kProjTriLds = kPDepth * kTriTile;

__device__ __forceinline__ void pglds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// The operand images of the attention passes (row images of Q, K and -- for the backward -- V; transposed images of V
// and -- for the backward -- K; layouts: tri_dev.h), written by the projection itself: every FULL tile of 32 points from
// the accumulators, the token rows' own tile (N a multiple of 32) by the workgroup that copies them; only a ragged
// last point tile goes through tri_split_qkv.  Nothing reads the 101 MB of fp32 rows back.
struct ProjImages {
  char* q_rm;   // null: no images at all
  char* k_rm;
  char* v_tr;
  char* k_tr;   // null: no backward images
  char* v_rm;
  int ktiles;   // tiles per cloud of the K / V images (rows N + nt); the Q image has ceil(N / 32)
  int q_only;   // the fp32 K / V columns of the point rows are not written where the images cover the tile
};
constexpr int kPXt = 36;  // row stride (floats) of a wave's 32 x 32 transpose tile

// What the projection needs from the weights and the tokens, in ONE launch (they were two of ~5 us each, both far
// below the cost of a launch): workgroups 0..11 write the row image of W (tile = 32 output rows; tri_split_kernel's
// bytes), workgroups 12.. the token rows tokqkv[t][o] = sum_c W[o][c] tokens[c][t] (proj_tok_fwd_kernel's arithmetic:
// one wave per output row, lanes across the channels, independent shuffle trees for the 8 token sums).
__global__ __launch_bounds__(256) void proj_prologue_kernel(const ProjW W, const float* __restrict__ tokens,
                                                            int nt, char* __restrict__ wimg, char* __restrict__ wtr,
                                                            float* __restrict__ tokqkv) {
  const int tid = threadIdx.x;
  if (blockIdx.x < kPTiles) {
    if (wtr && kLinDuo) {
      // the transposed image as well, for the backward's input gradient (no image launch there) -- in the form
      // lin_dx_duo_kernel reads: two fp16 planes under the tile's power-of-two scale, 2^-e in the spare slot; byte for byte
      // what lin_images_duo_kernel (linear.hip) writes for this tile
      __shared__ float red[4];
      char* img = wtr + (long)blockIdx.x * kTriTile;
      float v[2][8], amax = 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = tid + 256 * k, d = e & 127, cg = e >> 7, s2 = cg >> 1, hh = cg & 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          v[k][i] = W.row(blockIdx.x * 32 + 16 * s2 + 8 * (i >> 2) + 4 * hh + (i & 3))[d];
          amax = fmaxf(amax, fabsf(v[k][i]));
        }
      }
      amax = wave_max64(amax);
      if ((tid & 63) == 0) red[tid >> 6] = amax;
      __syncthreads();
      float sc, inv;
      duo_scale_for(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), sc, inv);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = tid + 256 * k, d = e & 127, cg = e >> 7;
        u32x4 hp, lp;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          unsigned hw, lw;
          duo_split2(v[k][2 * w] * sc, v[k][2 * w + 1] * sc, hw, lw);
          hp[w] = hw;
          lp[w] = lw;
        }
        *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 0)) = hp;
        *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 1)) = lp;
        *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 2)) = (d == 0 && cg == 0) ? u32x4{__float_as_uint(inv), 0u, 0u, 0u}
                                                                                    : u32x4{0u, 0u, 0u, 0u};
      }
    } else if (wtr) {  // (three-plane build: proj_dx_tri_kernel's image)
      char* img = wtr + (long)blockIdx.x * kTriTile;
      for (int e = tid; e < 512; e += 256) {
        const int d = e & 127, cg = e >> 7, s = cg >> 1, hh = cg & 1;
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = W.row(blockIdx.x * 32 + 16 * s + 8 * (i >> 2) + 4 * hh + (i & 3))[d];
        const Tri t = tri_split8(x);
        *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 0)) = t.h;
        *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 1)) = t.m;
        *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 2)) = t.l;
      }
    }
    char* img = wimg + (long)blockIdx.x * kTriTile;
    for (int e = tid; e < 512; e += 256) {
      const int r = e & 31, g = e >> 5, row = blockIdx.x * 32 + r;
      const f32x4* p = reinterpret_cast<const f32x4*>(W.row(row) + 8 * g);
      const f32x4 a = p[0], bb = p[1];
      const float x[8] = {a[0], a[1], a[2], a[3], bb[0], bb[1], bb[2], bb[3]};
      const Tri t = tri_split8(x);
      *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 0)) = t.h;
      *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 1)) = t.m;
      *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 2)) = t.l;
    }
    return;
  }
  if (nt <= 0) return;
  const int lane = tid & 63;
  const int o = (blockIdx.x - kPTiles) * 4 + (tid >> 6);
  const float w0 = W.row(o)[lane], w1 = W.row(o)[lane + 64];
  float p[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float tk0 = (t < nt) ? tokens[lane * nt + t] : 0.f;
    const float tk1 = (t < nt) ? tokens[(lane + 64) * nt + t] : 0.f;
    p[t] = fmaf(w0, tk0, w1 * tk1);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
    for (int t = 0; t < 8; ++t) p[t] += __shfl_xor(p[t], off, 64);
#pragma unroll
  for (int t = 0; t < 8; ++t)
    if (lane == 0 && t < nt) tokqkv[t * kPO + o] = p[t];
}

template <bool IMG>
__global__ __launch_bounds__(512, 2) void proj_fwd_tri_kernel(const float* __restrict__ x, long x_bs, int N,
                                                              const float* __restrict__ tokqkv, int nt,
                                                              const char* __restrict__ Wimg,  // row image of W (384 rows)
                                                              float* __restrict__ qkv, long o_bs, long o_rs,
                                                              const ProjImages im) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  constexpr int D = kPDepth;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  // points past N-1 are clamped: those lanes recompute and rewrite point N-1's row bit for bit (no predicated store)
  const int n = min(chunk * 256 + wave * 32 + lo, N - 1);
  auto stage = [&](int t) {
    const char* gt = Wimg + (long)min(t, kPTiles - 1) * kTriTile;
    char* lt = smem_c + (t % D) * kTriTile;
#pragma unroll
    for (int k = 0; k < 3; ++k) pglds16(gt + (tid + 512 * k) * 16, lt + (wave * 64 + 512 * k) * 16);
  };
#pragma unroll
  for (int t = 0; t < D - 1; ++t) stage(t);
  if (chunk == 0) {  // this cloud's copy of the token rows
    for (int e = tid; e < nt * kPO; e += 512) qkv[(long)b * o_bs + (long)(N + e / kPO) * o_rs + (e % kPO)] = tokqkv[e];
    if (IMG && nt > 0 && (N & 31) == 0) {
      // ... and, when the token rows have an image tile of their own, that tile of the K / V images (rows nt.. zero)
      const long toff = ((long)b * im.ktiles + (N >> 5)) * kTriTile;
      // row-image tiles of K and (for the backward) V, in their logit form like the point tiles below (tri_dev.h): pass
      // `which` has the whole workgroup on one image, one chunk triple per thread; the tile's largest value through the
      // (still unused) transpose area of the LDS
      float* red = reinterpret_cast<float*>(smem_c + D * kTriTile);
      for (int which = 1; which <= 2; ++which) {
        if (which == 2 && !im.v_rm) break;  // (uniform)
        const int r = tid & 31, g = tid >> 5;
        float v8[8], amax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          v8[i] = (r < nt) ? tokqkv[r * kPO + 128 * which + 8 * g + i] : 0.f;
          amax = fmaxf(amax, fabsf(v8[i]));
        }
        amax = wave_max64(amax);
        __syncthreads();  // (the previous pass's readers of red)
        if (lane == 0) red[wave] = amax;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 8; ++w) amax = fmaxf(amax, red[w]);
        float sc, inv;
        duo_scale_for(amax, sc, inv);
        u32x4 hw, lw;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          unsigned a1, a2;
          duo_split2(v8[2 * w] * sc, v8[2 * w + 1] * sc, a1, a2);
          hw[w] = a1;
          lw[w] = a2;
        }
        char* img = (which == 1 ? im.k_rm : im.v_rm) + toff;
        *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 0)) = hw;
        *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 1)) = lw;
        if (tid == 0) *reinterpret_cast<u32x4*>(img + kDuoScaleSlot) = u32x4{__float_as_uint(inv), kDuoTag, 0u, 0u};
      }
      for (int e = tid; e < 1024; e += 512) {  // transposed-image chunks of V and (for the backward) K
        const int which = e >> 9, d = e & 127, cg = (e >> 7) & 3, s2 = cg >> 1, hh = cg & 1;
        if (which == 1 && !im.k_tr) continue;
        float v8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = 16 * s2 + 8 * (i >> 2) + 4 * hh + (i & 3);
          v8[i] = (row < nt) ? tokqkv[row * kPO + (which ? 128 : 256) + d] : 0.f;
        }
        const Tri t3 = tri_split8(v8);
        char* img = (which ? im.k_tr : im.v_tr) + toff;
        *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 0)) = t3.h;
        *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 1)) = t3.m;
        *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 2)) = t3.l;
      }
    }
  }
  u32x4 xq[24];  // this point's channels as the B operand: k-step ks, half h <-> channels 16 ks + 8 h + e
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = x[(long)b * x_bs + (long)(16 * ks + 8 * h + e) * N + n];
    const Tri t3 = tri_split8(v);
    xq[3 * ks] = t3.h;
    xq[3 * ks + 1] = t3.m;
    xq[3 * ks + 2] = t3.l;
  }
  float* orow = qkv + (long)b * o_bs + (long)n * o_rs + 4 * h;
  // images: this wave's 32 points are one image tile when all of them exist (wave-uniform)
  const int n0 = chunk * 256 + wave * 32;
  const bool full = IMG && n0 + 31 < N;
  const int ptile = n0 >> 5, qtiles = (N + 31) >> 5;
  float* xt = reinterpret_cast<float*>(smem_c + D * kTriTile) + wave * (32 * kPXt);  // [output of the tile][point]
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // iteration t: tile t+3 into the slot of tile t-1, product of tile t, its 4 row stores; VM operations
  // younger than tile t+1's DMA at the end of the iteration: stores(t-2) 4 + 2 x (3 + 4) = 18 (image stores only
  // add to that: counting low is the safe side)
  // The K row image (and the V row image of the backward) leaves in its LOGIT form (tri_dev.h: two fp16 planes per
  // 32-point tile under the tile's own power-of-two scale).  A wave's 32 points ARE one tile, its 128 K channels come out
  // of iterations t = 4..7 (V: 8..11): their
  // values wait in registers (kst, compile-time indices: those four iterations are written out below) until the last of
  // them knows the tile's largest |k|.
  float kst[4][2][8];
  auto body = [&](int t, auto kc_c) {
    constexpr int KC = decltype(kc_c)::value;  // 0..3: K channel block tc of a FULL tile with images; -1: everything else
    stage(t + D - 1);
    const u32x4* lp = reinterpret_cast<const u32x4*>(smem_c + (t % D) * kTriTile + tri_rm_off(lo, h, 0));
    f32x16 acc = zero16();  // D[row = output 32 t + crow(r, h)][col = point]
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const Tri a = {lp[192 * ks], lp[192 * ks + 32], lp[192 * ks + 64]};
      const Tri bq = {xq[3 * ks], xq[3 * ks + 1], xq[3 * ks + 2]};
      acc = mfma_tri(a, bq, acc);
    }
    if (t < 4 || !(IMG && im.q_only && full)) {  // (wave-uniform; 67 of the kernel's 392 MB at B=32, N=2048)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        *reinterpret_cast<f32x4*>(orow + t * 32 + 8 * g) = o;
      }
    }
    if (full) {
      const int which = t >> 2, tc = t & 3;  // 0 Q, 1 K, 2 V; channels 32 tc .. 32 tc + 31 of it
      char* rm = which == 0 ? im.q_rm + ((long)b * qtiles + ptile) * kTriTile
                            : (which == 1 ? im.k_rm : im.v_rm) + ((long)b * im.ktiles + ptile) * kTriTile;
      char* tr = (which == 1 ? im.k_tr : im.v_tr) + ((long)b * im.ktiles + ptile) * kTriTile;
      if (which == 0 || which == 1 || im.v_rm) {
        // row image: a chunk = 8 consecutive channels of one point.  Lane (point, h) holds channels 8 g + 4 h .. + 3 of
        // the groups g = 0..3; one v_permlane32_swap per register pair hands lane h = 0 the whole groups 0 and 1 and
        // lane h = 1 the whole groups 2 and 3
        float c8[2][8];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {  // groups (pr, pr + 2)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[4 * pr + e]), __float_as_uint(acc[4 * (pr + 2) + e]),
                                                             false, false);
            // r2[0] = {h0's group pr, h0's group pr+2}, r2[1] = {h1's group pr, h1's group pr+2} (lower | upper half)
            c8[pr][e] = __uint_as_float(r2[0]);      // channels 0..3 of this lane's group (own for h = 0, partner's for h = 1)
            c8[pr][4 + e] = __uint_as_float(r2[1]);  // channels 4..7
          }
        }
        if constexpr (KC >= 0) {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int e = 0; e < 8; ++e) kst[KC][pr][e] = c8[pr][e];
          if constexpr (KC == 3) {  // the tile is complete: its exponent, then the two fp16 planes of all 8 groups of this lane
            float amax = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(kst[c][pr][e]));
            amax = wave_max64(amax);
            float sc, inv;
            duo_scale_for(amax, sc, inv);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int pr = 0; pr < 2; ++pr) {
                const int g = 4 * c + pr + 2 * h;
                u32x4 hw, lw;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                  unsigned a, b2;
                  duo_split2(kst[c][pr][2 * w] * sc, kst[c][pr][2 * w + 1] * sc, a, b2);
                  hw[w] = a;
                  lw[w] = b2;
                }
                *reinterpret_cast<u32x4*>(rm + tri_rm_off(lo, g, 0)) = hw;
                *reinterpret_cast<u32x4*>(rm + tri_rm_off(lo, g, 1)) = lw;
              }
            // (the third piece slots stay unwritten: dead space of the image, except the tile's 2^-e)
            if (lane == 0) *reinterpret_cast<u32x4*>(rm + kDuoScaleSlot) = u32x4{__float_as_uint(inv), kDuoTag, 0u, 0u};
          }
        } else {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const int g = 4 * tc + pr + 2 * h;  // 8-channel group inside the 128 channels
            const Tri t3 = tri_split8(c8[pr]);
            *reinterpret_cast<u32x4*>(rm + tri_rm_off(lo, g, 0)) = t3.h;
            *reinterpret_cast<u32x4*>(rm + tri_rm_off(lo, g, 1)) = t3.m;
            *reinterpret_cast<u32x4*>(rm + tri_rm_off(lo, g, 2)) = t3.l;
          }
        }
      }
      if (which == 2 || (which == 1 && im.k_tr)) {
        // transposed image: a chunk = 8 POINTS of one channel: through the wave's LDS tile (wave-private, in-order LDS)
#pragma unroll
        for (int r = 0; r < 16; ++r) xt[crow(r, h) * kPXt + lo] = acc[r];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = lane + 64 * u, c = e & 31, cg = e >> 5, s2 = cg >> 1, hh = cg & 1;
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(xt + c * kPXt + 16 * s2 + 4 * hh);
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(xt + c * kPXt + 16 * s2 + 8 + 4 * hh);
          const float v8[8] = {a4[0], a4[1], a4[2], a4[3], b4[0], b4[1], b4[2], b4[3]};
          const Tri t3 = tri_split8(v8);
          const int d = 32 * tc + c;
          *reinterpret_cast<u32x4*>(tr + tri_tr_off(d, cg, 0)) = t3.h;
          *reinterpret_cast<u32x4*>(tr + tri_tr_off(d, cg, 1)) = t3.m;
          *reinterpret_cast<u32x4*>(tr + tri_tr_off(d, cg, 2)) = t3.l;
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  using std::integral_constant;
  for (int t = 0; t < 4; ++t) body(t, integral_constant<int, -1>{});
  if (full) {  // (wave-uniform)
    body(4, integral_constant<int, 0>{});
    body(5, integral_constant<int, 1>{});
    body(6, integral_constant<int, 2>{});
    body(7, integral_constant<int, 3>{});
  } else {
    for (int t = 4; t < 8; ++t) body(t, integral_constant<int, -1>{});
  }
  if (full && im.v_rm) {  // the V row image (the backward's dP = dO V^T) leaves the same way
    body(8, integral_constant<int, 0>{});
    body(9, integral_constant<int, 1>{});
    body(10, integral_constant<int, 2>{});
    body(11, integral_constant<int, 3>{});
  } else {
    for (int t = 8; t < kPTiles; ++t) body(t, integral_constant<int, -1>{});
  }
}

#ifdef SAMBLE_STAMPS  // scratch builds only (tools/scratch, tools/proj_stamps.py): workgroup (0,0), every wave, tiles 5 and 6
__device__ unsigned long long g_proj_stamps[8 * 2 * 4];
#define XSTAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (t == 5 || t == 6)) \
  g_proj_stamps[(wave * 2 + (t - 5)) * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define XSTAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(512, 2) void proj_dx_tri_kernel(const float* __restrict__ dqkv, long g_bs, long g_rs,
                                                             const char* __restrict__ Wtr,  // transposed image of W
                                                             int N, float* __restrict__ dx, long dx_bs, const float* res) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  constexpr int D = kPDepth;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int n = min(blockIdx.x * 256 + wave * 32 + lo, N - 1);
  const bool own = blockIdx.x * 256 + wave * 32 + lo < N;   // (with `res`, which may alias dx: the clamped lanes do not store)
  const float* grow = dqkv + (long)b * g_bs + (long)n * g_rs + 4 * h;
  auto stage = [&](int t) {
    const char* gt = Wtr + (long)min(t, kPTiles - 1) * kTriTile;
    char* lt = smem_c + (t % D) * kTriTile;
#pragma unroll
    for (int k = 0; k < 3; ++k) pglds16(gt + (tid + 512 * k) * 16, lt + (wave * 64 + 512 * k) * 16);
  };
  // this lane's 16 gradient values of tile t in the transposed image's element order:
  // k-step s, element e <-> output 32 t + 16 s + 8 (e >> 2) + 4 h + (e & 3)
  // The loads are written in assembly so that the compiler does not count them: with its own bookkeeping it put an
  // s_waitcnt vmcnt(0) at the top of the loop (for the rows loaded one iteration earlier) -- behind the three W pieces
  // just issued, so every tile sat out a DMA round trip (stamped: 2000-3000 of a tile's 6000 cycles; tools/isa_waits.py
  // finds such waits).  The hand-counted wait at the end of the iteration names the registers, which orders their use.
  auto load_g = [&](int t, f32x4 (&dst)[4]) {
    const float* p = grow + min(t, kPTiles - 1) * 32;
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:32\n\t"
                 "global_load_dwordx4 %2, %4, off offset:64\n\tglobal_load_dwordx4 %3, %4, off offset:96"
                 : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3])
                 : "v"(p)
                 : "memory");
  };
#pragma unroll
  for (int t = 0; t < D - 1; ++t) stage(t);
  f32x16 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = zero16();
  // Software pipeline: tile t's products run on operands split one iteration earlier; tile t+1's rows (loaded one
  // iteration earlier still) are split pair by pair behind the MFMAs -- the two waves of a SIMD run this loop in step,
  // and with the split in front of the products both sat on the vector ALU, then both on the matrix pipe.
  f32x4 gn[4], gnn[4];
  Tri bg[2], nb[2];
  load_g(0, gnn);
  load_g(1, gn);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier"
               : "+v"(gnn[0]), "+v"(gnn[1]), "+v"(gnn[2]), "+v"(gnn[3]), "+v"(gn[0]), "+v"(gn[1]), "+v"(gn[2]), "+v"(gn[3])::"memory");
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const float v[8] = {gnn[2 * ks][0], gnn[2 * ks][1], gnn[2 * ks][2], gnn[2 * ks][3],
                        gnn[2 * ks + 1][0], gnn[2 * ks + 1][1], gnn[2 * ks + 1][2], gnn[2 * ks + 1][3]};
    bg[ks] = tri_split8(v);
  }

  for (int t = 0; t < kPTiles; ++t) {
    XSTAMP(0);
    load_g(t + 2, gnn);  // 4 loads, then the 3 DMA pieces: the wait below leaves exactly those 3 in flight
    stage(t + D - 1);
    const char* wt = smem_c + (t % D) * kTriTile;
    auto fetch = [&](int i) {
      const char* ap = wt + tri_tr_off(32 * (i & 3) + lo, 2 * (i >> 2) + h, 0);
      return Tri{*reinterpret_cast<const u32x4*>(ap), *reinterpret_cast<const u32x4*>(ap + 2048),
                 *reinterpret_cast<const u32x4*>(ap + 4096)};
    };
    Tri a0 = fetch(0), a1 = fetch(1), a2 = fetch(2);  // W operands three groups ahead of their MFMAs
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // k-step i >> 2, channel tile i & 3; pair i of tile t+1's values beside it
      const int ks = i >> 2, ct = i & 3;
      Tri a3 = a2;
      if (i + 3 < 8) a3 = fetch(i + 3);
      __builtin_amdgcn_sched_barrier(0);
      acc[ct] = mfma_tri(a0, bg[ks], acc[ct]);
      unsigned hh, mm, ll;
      tri_split2(gn[2 * ks + (ct >> 1)][2 * (ct & 1)], gn[2 * ks + (ct >> 1)][2 * (ct & 1) + 1], hh, mm, ll);
      nb[ks].h[ct] = hh;
      nb[ks].m[ct] = mm;
      nb[ks].l[ct] = ll;
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      a0 = a1;
      a1 = a2;
      a2 = a3;
    }
#ifdef SAMBLE_STAMPS
    XSTAMP(1);
    asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" : "+v"(gnn[0]), "+v"(gnn[1]), "+v"(gnn[2]), "+v"(gnn[3])::"memory");
    XSTAMP(2);
    asm volatile("s_barrier" ::: "memory");
    XSTAMP(3);
#else
    asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" : "+v"(gnn[0]), "+v"(gnn[1]), "+v"(gnn[2]), "+v"(gnn[3])::"memory");
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) gn[i] = gnn[i];
    bg[0] = nb[0];
    bg[1] = nb[1];
  }
  // rows past N-1 hold point N-1's column again: same values to the same address
  float* ob = dx + (long)b * dx_bs + n;
  if (res) {  // (may be dx itself) the gradient that reaches x beside this product: all loads first, then the stores
    const float* rb = res + (long)b * dx_bs + n;
    float rv[4][16];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[ct][r] = rb[(long)(32 * ct + crow(r, h)) * N];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (own) ob[(long)(32 * ct + crow(r, h)) * N] = rv[ct][r] + acc[ct][r];
    return;
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
    for (int r = 0; r < 16; ++r) ob[(long)(32 * ct + crow(r, h)) * N] = acc[ct][r];
  }
}

// ------------------------------------------------------------------------------------------------
// dW partials on the split scheme: dW[o][c] = sum_n dqkv[n][o] x[c][n].  The contraction runs over the POINTS, so
// both operands are transposed on the fly: a 32-point tile of dqkv ([n][384 o]) and of x ([128 c][n]) goes through
// LDS as in proj_dw_kernel (proj.hip); lane (o or c, h) then reads its 8 points of a k-step column-wise and splits
// them into the three bf16 planes in registers.  Workgroup = (cloud, 256-point chunk), 8 waves = two per SIMD;
// wave w owns output rows 96 (w >> 1) .. +95 and channels 64 (w & 1) .. +63: six accumulator tiles, per 32 points
// 2 k-steps x (3 + 2 operand splits, 36 MFMAs).
// ------------------------------------------------------------------------------------------------
constexpr int kDwTriPts = 256;
constexpr int kDwTriGS = 388, kDwTriXS = 33;  // row strides: dqkv tile rows 16-byte aligned, column reads conflict-free
constexpr int kDwTriBuf = kTile * kDwTriGS + 128 * kDwTriXS;
constexpr int kDwTriLds = 2 * kDwTriBuf * 4;

__global__ __launch_bounds__(512, 2) void proj_dw_tri_kernel(const float* __restrict__ dqkv, long g_bs, long g_rs,
                                                             const float* __restrict__ x, long x_bs, int N,
                                                             float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  float* smem = reinterpret_cast<float*>(smem_c);
  constexpr int GS = kDwTriGS, XS = kDwTriXS, BUF = kDwTriBuf;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  const int og = wave >> 1, ch = wave & 1;
  const int b = blockIdx.y;
  const int n0 = blockIdx.x * kDwTriPts;

  f32x16 acc[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[a][c] = zero16();

  f32x4 gst[6];  // 32 rows x 96 float4 = 3072 float4 / 512 threads
  float xst[8];  // 128 channels x 32 points = 4096 floats / 512 threads
  auto issue = [&](int nn0) {
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int e = tid + 512 * it;
      const int r = e / 96, c4 = (e % 96) * 4;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      gst[it] = (nn0 + r < N) ? *reinterpret_cast<const f32x4*>(dqkv + (long)b * g_bs + (long)(nn0 + r) * g_rs + c4) : z4;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int e = tid + 512 * it;
      const int c = e >> 5, pnt = e & 31;
      xst[it] = (nn0 + pnt < N) ? x[(long)b * x_bs + (long)c * N + nn0 + pnt] : 0.f;
    }
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int e = tid + 512 * it;
      const int r = e / 96, c4 = (e % 96) * 4;
      *reinterpret_cast<f32x4*>(buf + r * GS + c4) = gst[it];
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int e = tid + 512 * it;
      buf[kTile * GS + (e >> 5) * XS + (e & 31)] = xst[it];
    }
  };
  constexpr int ntiles = kDwTriPts / kTile;
  issue(n0);
  commit(smem);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    float* cur = smem + (t & 1) * BUF;
    float* nxt = smem + ((t & 1) ^ 1) * BUF;
    if (t + 1 < ntiles) issue(n0 + (t + 1) * kTile);
    const float* gt = cur;
    const float* xt = cur + kTile * GS;
    // D[row = o][col = c] += sum_n dqkv[n][o] * x[c][n]; k-step ks covers points 16 ks .. 16 ks + 15, lane half h
    // the 8 points 16 ks + 8 h + e
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Tri bq[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = xt[(64 * ch + 32 * ct + lo) * XS + 16 * ks + 8 * h + e];
        bq[ct] = tri_split8(v);
      }
#pragma unroll
      for (int ot = 0; ot < 3; ++ot) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gt[(16 * ks + 8 * h + e) * GS + 96 * og + 32 * ot + lo];
        const Tri a = tri_split8(v);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[ot][ct] = mfma_tri(a, bq[ct], acc[ot][ct]);
      }
    }
    if (t + 1 < ntiles) commit(nxt);
    __syncthreads();
  }
  float* out = part + ((long)b * gridDim.x + blockIdx.x) * kPO * 128;
#pragma unroll
  for (int ot = 0; ot < 3; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = 96 * og + 32 * ot + crow(r, h);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) out[(long)o * 128 + 64 * ch + 32 * ct + lo] = acc[ot][ct][r];
    }
}

}  // namespace samble

using namespace samble;

extern "C" int samble_launch_tri_split(const float* src, long bs, long rs, int B, int rows, void* rm, void* tr,
                                       hipStream_t stream);

// image bytes of W (384 x 128), either kind
extern "C" size_t samble_proj_tri_image_bytes() { return (size_t)kPTiles * kTriTile; }

extern "C" int samble_launch_tri_split_qkv_tiles(const float* qkv, long bs, long rs, int B, int N, int nt, int tile0, void* qimg,
                                                 void* kimg, void* vimg, void* ktr, void* vrm, hipStream_t stream);
extern "C" int samble_launch_k_to_duo(void* kimg, int B, int rows, int tile0, hipStream_t stream);

// images (q_rm non-null): the five operand images of (B, N + nt, 384) = [Q | K | V] are written as well -- the full
// 32-point tiles by the projection kernel, the rest (token rows, ragged end) by a tri_split_qkv launch over those tiles
extern "C" int samble_launch_proj_fwd_tri(const float* x, long x_bs, int B, int N, const float* tokens, float* tokqkv,
                                          int nt, const float* W, const float* Wk, const float* Wv, void* wimg,
                                          void* wtr_out, float* qkv, long o_bs, long o_rs, void* q_rm,
                                          void* k_rm, void* v_tr, void* k_tr, void* v_rm, int q_only, hipStream_t s) {
  const int lds_img = kProjTriLds + 8 * 32 * kPXt * 4;
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(proj_fwd_tri_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kProjTriLds);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(proj_fwd_tri_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_img);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(proj_dx_tri_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kProjTriLds);
    if (e != hipSuccess) return (int)e;
  }
  int rc = 0;
  hipLaunchKernelGGL(proj_prologue_kernel, dim3(kPTiles + kPO / 4), dim3(256), 0, s, proj_w(W, Wk, Wv), tokens, nt,
                     (char*)wimg, (char*)wtr_out, tokqkv);
  const ProjImages im{(char*)q_rm, (char*)k_rm, (char*)v_tr, (char*)k_tr, (char*)v_rm, (N + nt + 31) / 32, q_only};
  {
    Timed timed(kT_proj_fwd, s);
    if (q_rm)
      hipLaunchKernelGGL(proj_fwd_tri_kernel<true>, dim3((N + 255) / 256, B), dim3(512), lds_img, s, x, x_bs, N, tokqkv, nt,
                         (const char*)wimg, qkv, o_bs, o_rs, im);
    else
      hipLaunchKernelGGL(proj_fwd_tri_kernel<false>, dim3((N + 255) / 256, B), dim3(512), kProjTriLds, s, x, x_bs, N, tokqkv,
                         nt, (const char*)wimg, qkv, o_bs, o_rs, im);
  }
  if (q_rm && (N & 31)) {  // a ragged last point tile (+ the token rows behind it): the split kernel over those tiles
    rc = samble_launch_tri_split_qkv_tiles(qkv, o_bs, o_rs, B, N, nt, N / 32, q_rm, k_rm, v_tr, k_tr, v_rm, s);
    if (rc) return rc;
  }
  // the K and V row images leave in their logit form (tri_dev.h): the full point tiles and a token tile of its own from
  // the kernel, the tiles of a ragged end (written as three planes by the split launch above) by the conversion kernel
  if (k_rm && (N & 31)) {  // (N % 32 == 0: the kernel wrote the token tile in that form as well)
    rc = samble_launch_k_to_duo(k_rm, B, N + nt, N / 32, s);
    if (rc) return rc;
    if (v_rm) return samble_launch_k_to_duo(v_rm, B, N + nt, N / 32, s);
  }
  return (int)hipGetLastError();
}

extern "C" int samble_launch_linear_images(const float* W, int O, void* rm, void* tr, int transposed, hipStream_t s);
extern "C" int samble_launch_linear_dx_as(const float* g, long g_bs, long g_rs, const void* w_tr, int O, int B, int Cin, int N,
                                          float* dx, long dx_bs, const float* residual, hipStream_t s, int timing_id);
extern "C" int samble_launch_proj_dx_tri(const float* dqkv, long g_bs, long g_rs, const float* W, void* wtr, int have_wtr,
                                         int B, int N, float* dx, long dx_bs, const float* residual, hipStream_t s) {
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(proj_dx_tri_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kProjTriLds);
    if (e != hipSuccess) return (int)e;
  }
  if (kLinDuo) {
    // two fp16 planes: csrc/linear.hip's lin_dx kernel with 12 output tiles (3 matrix instructions per product instead of
    // 6, and its launch geometry for short clouds); the image: the forward's prologue, or linear.hip's image kernel here
    if (!have_wtr) {
      const int rc = samble_launch_linear_images(W, kPO, nullptr, wtr, 0, s);
      if (rc) return rc;
    }
    return samble_launch_linear_dx_as(dqkv, g_bs, g_rs, wtr, kPO, B, 128, N, dx, dx_bs, residual, s, kT_proj_dx);
  }
  if (!have_wtr) {  // (the forward's prologue wrote it otherwise)
    const int rc = samble_launch_tri_split(W, 0, 128, 1, kPO, nullptr, wtr, s);
    if (rc) return rc;
  }
  Timed timed(kT_proj_dx, s);
  hipLaunchKernelGGL(proj_dx_tri_kernel, dim3((N + 255) / 256, B), dim3(512), kProjTriLds, s, dqkv, g_bs, g_rs,
                     (const char*)wtr, N, dx, dx_bs, residual);
  return (int)hipGetLastError();
}

// per-chunk partials of dW (same layout and chunking as proj_dw_kernel: the fixed-order reduce of proj.hip follows)
extern "C" int samble_launch_proj_dw_tri(const float* dqkv, long g_bs, long g_rs, const float* x, long x_bs, int B, int N,
                                         float* part, hipStream_t s) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(proj_dw_tri_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kDwTriLds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(proj_dw_tri_kernel, dim3((N + kDwTriPts - 1) / kDwTriPts, B), dim3(512), kDwTriLds, s, dqkv, g_bs,
                     g_rs, x, x_bs, N, part);
  return (int)hipGetLastError();
}

#ifdef SAMBLE_STAMPS
extern "C" __attribute__((visibility("default"))) int samble_scratch_proj_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(samble::g_proj_stamps), sizeof(unsigned long long) * 64);
}
#endif
