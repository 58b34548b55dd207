// Fused EdgeConv body (reference models/embedding.py:7-39): for every point i and its K = 32 nearest
// neighbours j,   out_i = max_k LReLU(BN2(W2 . LReLU(BN1(W1 . [x_i ; x_j - x_i])))).
//
// The reference materialises five (B, 64, N, K) tensors per layer (537 MB each at B = 32, N = 2048).
// Here none exists in the forward pass:
//   * a 1x1 conv is linear, so  W1 . [x_i ; x_j - x_i] = a_i + b_j  with two per-POINT projections
//     a = (W1c - W1d) x, b = W1d x (every group_type of utils/ops.py:83-112 has this form); the caller
//     folds the BatchNorm-1 affine into them (a', b'), so an edge's hidden vector is LReLU(a'_i + b'_j);
//   * BatchNorm-1's batch statistics over all B.N.K edges follow from per-point sums: the caller needs
//     S_i = sum_k b_j and Q_i = sum_k b_j^2 (edge_gather_sums), the rest is closed form;
//   * the 64 -> 64 conv2 is the one real GEMM over edges: one wave = one point = 32 edges, 64 fp32 MFMAs;
//   * LReLU o BN2 is monotone per channel, so max_k commutes with it: only max_k / min_k of the raw
//     conv2 output per (point, channel) and the channel sums for BN2's statistics leave the kernel.
// Backward (edge_mlp_bwd) recomputes the edge tensors tile by tile and emits the gradient of the
// pre-activation (a'_i + b'_j) per edge plus the dW2 partials; everything else is closed form on
// per-point tensors (samble_amd/embedding.py).
#include "tri_dev.h"

// 1: the two sweeps on v_mfma_f32_32x32x2_f32 (the round-2 kernels; scratch builds for A/B runs)
// timing-only ablations of edge_mlp_bwd_tri (wrong results; tools/bench_edge_mlp.py): 1 no du stores, 2 no dW2 product,
// 4 no dh product, 8 no dusum butterflies
#ifndef SAMBLE_EDGE_ABL
#define SAMBLE_EDGE_ABL 0
#endif
#ifndef SAMBLE_EDGE_F32
#define SAMBLE_EDGE_F32 0
#endif
// 1 (default since round 5): the two sweeps' products on TWO fp16 planes per operand (tri_dev.h "duo", 3 matrix
// instructions per k-step instead of 6) under power-of-two scales -- one per W2 image, one per point for its 32 x 64 tile
// of hidden vectors, one per point for its tile of dy; dW2, which accumulates over the points, takes every point's product
// from a zero accumulator and adds it to the fp32 totals.  0: the three-bf16-plane kernels of round 4 (A/B)
#ifndef SAMBLE_EDGE_DUO
#define SAMBLE_EDGE_DUO 1
#endif

namespace samble {

constexpr int kEC = 64;   // channels of the hidden and output features
constexpr int kEK = 32;   // neighbours per point = edges per wave

__device__ __forceinline__ float lrelu(float v) { return fmaxf(v, 0.2f * v); }

// sum over the 32 lanes of each half of the wave, in every lane of rows 1 and 3 (lanes 16..31, 48..63): quad swaps, half-row
// and row mirrors (every lane of a row of 16 then holds the row's sum, formed in the same order), row broadcast 15
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float edge_dpp_f(float x) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ float half_wave_sum(float x) {
  x += edge_dpp_f<0xB1, 0xF>(x);   // quad_perm [1,0,3,2]
  x += edge_dpp_f<0x4E, 0xF>(x);   // quad_perm [2,3,0,1]
  x += edge_dpp_f<0x141, 0xF>(x);  // row_half_mirror
  x += edge_dpp_f<0x140, 0xF>(x);  // row_mirror
  x += edge_dpp_f<0x142, 0xA>(x);  // row_bcast15 -> rows 1, 3 (rows 0, 2 receive 0: their value is not used)
  return x;
}

// S[p][c] = sum_k bp[j(p,k)][c],  Q[p][c] = sum_k bp[j(p,k)][c]^2     (p = b*N + i)
// half-wave per point, lane = 2 channels; grid-stride over points
__global__ __launch_bounds__(256) void edge_gather_sums_kernel(const float* __restrict__ bp, const int* __restrict__ nn,
                                                               int N, long npoints, float* __restrict__ S,
                                                               float* __restrict__ Q) {
  const int hw = threadIdx.x >> 5, c2 = threadIdx.x & 31;
  for (long p = (long)blockIdx.x * 8 + hw; p < npoints; p += (long)gridDim.x * 8) {
    const long cloud = p / N;
    const int* ni = nn + p * kEK;
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll 8
    for (int k = 0; k < kEK; ++k) {
      const int j = ni[k];
      const float2 v = *reinterpret_cast<const float2*>(bp + (cloud * N + j) * kEC + 2 * c2);
      s0 += v.x;
      s1 += v.y;
      q0 = fmaf(v.x, v.x, q0);
      q1 = fmaf(v.y, v.y, q1);
    }
    *reinterpret_cast<float2*>(S + p * kEC + 2 * c2) = make_float2(s0, s1);
    *reinterpret_cast<float2*>(Q + p * kEC + 2 * c2) = make_float2(q0, q1);
  }
}

// forward sweep.  ap/bp (npoints, 64): BN1-folded projections; W2 (64 out, 64 in) row-major.
// ymax/ymin (npoints, 64): max_k / min_k of y = W2 h over the point's edges; part (nwaves, 2, 64) doubles:
// per-wave sums of y and y^2 (BN2 batch statistics), summed by the caller in a fixed order.
__global__ __launch_bounds__(256, 2) void edge_mlp_fwd_kernel(const float* __restrict__ ap, const float* __restrict__ bp,
                                                              const int* __restrict__ nn,
                                                              const float* __restrict__ W2, int N, long npoints,
                                                              float* __restrict__ ymax, float* __restrict__ ymin,
                                                              unsigned char* __restrict__ kmax,
                                                              unsigned char* __restrict__ kmin,
                                                              double* __restrict__ part) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  const long gw = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
  // B operand: lane (o = lo, h) holds W2[32 ot + lo][32 h + kk]
  float w2r[2][32];
#pragma unroll
  for (int ot = 0; ot < 2; ++ot) {
    const f32x4* wp = reinterpret_cast<const f32x4*>(W2 + (long)(32 * ot + lo) * kEC + 32 * h);
#pragma unroll
    for (int q4 = 0; q4 < 8; ++q4) {
      const f32x4 v = wp[q4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w2r[ot][4 * q4 + e] = v[e];
    }
  }
  double s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0};
  for (long p = gw; p < npoints; p += nw) {
    const long cloud = p / N;
    const int j = nn[p * kEK + lo];
    const f32x4* av = reinterpret_cast<const f32x4*>(ap + p * kEC + 32 * h);
    const f32x4* bv = reinterpret_cast<const f32x4*>(bp + (cloud * N + j) * kEC + 32 * h);
    float hv[32];
#pragma unroll
    for (int q4 = 0; q4 < 8; ++q4) {
      const f32x4 a4 = av[q4], b4 = bv[q4];
#pragma unroll
      for (int e = 0; e < 4; ++e) hv[4 * q4 + e] = lrelu(a4[e] + b4[e]);
    }
    // D[row = edge][col = out channel] = sum_c h[edge][c] W2[o][c]
    f32x16 acc[2] = {zero16(), zero16()};
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
      acc[0] = mfma32(hv[kk], w2r[0][kk], acc[0]);
      acc[1] = mfma32(hv[kk], w2r[1][kk], acc[1]);
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
      float mx = acc[ot][0], mn = acc[ot][0], s = 0.f, q = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        mx = fmaxf(mx, acc[ot][r]);
        mn = fminf(mn, acc[ot][r]);
        s += acc[ot][r];
        q = fmaf(acc[ot][r], acc[ot][r], q);
      }
      mx = fmaxf(mx, wave_xor32(mx));
      mn = fminf(mn, wave_xor32(mn));
      s += wave_xor32(s);
      q += wave_xor32(q);
      // the edge that attains the extremum (first one on ties): the backward routes the gradient by index
      int kx = 99, kn = 99;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        kx = min(kx, acc[ot][r] == mx ? crow(r, h) : 99);
        kn = min(kn, acc[ot][r] == mn ? crow(r, h) : 99);
      }
      kx = xor32_min(kx);
      kn = xor32_min(kn);
      if (h == 0) {
        ymax[p * kEC + 32 * ot + lo] = mx;
        ymin[p * kEC + 32 * ot + lo] = mn;
        kmax[p * kEC + 32 * ot + lo] = (unsigned char)kx;
        kmin[p * kEC + 32 * ot + lo] = (unsigned char)kn;
      }
      s1[ot] += (double)s;
      s2[ot] += (double)q;
    }
  }
  if (h == 0 && gw < nw) {
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
      part[(gw * 2 + 0) * kEC + 32 * ot + lo] = s1[ot];
      part[(gw * 2 + 1) * kEC + 32 * ot + lo] = s2[ot];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward sweep.  Per point (wave) it recomputes h and y^T = W2 h^T (edges on lanes), forms
//     dy[k][o] = c0[o] + c1[o] y[k][o] + (k == argext(i,o) ? sdv[i][o] : 0)
// (BatchNorm-2's dense correction plus the gradient that arrives at the arg-max / arg-min edge; c0,
// c1 are per-channel constants and sdv = sc2 * LReLU'(v) * g per (point, channel), from the caller),
// then  dh^T = W2^T dy^T  (accumulator-as-operand), du = dh * LReLU'(u)  -> written per edge
// (npoints, 32, 64), and  dW2 += dy^T h  through two LDS tiles.  dW2 partials per wave.
// ------------------------------------------------------------------------------------------------
constexpr int kEwPad = 68;  // LDS row stride of 64-float rows (16-byte aligned, odd multiple of 16 bytes)

__global__ __launch_bounds__(512, 2) void edge_mlp_bwd_kernel(const float* __restrict__ ap, const float* __restrict__ bp,
                                                              const int* __restrict__ nn,
                                                              const float* __restrict__ W2,
                                                              const unsigned char* __restrict__ kext,  // (npoints,64) arg-max or arg-min edge per the sign of gamma2
                                                              const float* __restrict__ sdv,    // (npoints,64)
                                                              const float* __restrict__ c0c1,   // (2,64)
                                                              int N, long npoints, float* __restrict__ du,
                                                              float* __restrict__ dusum,  // (npoints,64) sum_k du, or null
                                                              float* __restrict__ dw2part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* w2s = smem;                         // [64 o][kEwPad]   W2, read as rows (A operand) and as columns
  float* cst = w2s + kEC * kEwPad;           // c0[64], c1[64]
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  float* hts = cst + 2 * kEC + wave * (2 * kEK * kEwPad);  // this wave's [32 edges][68] h tile
  float* dys = hts + kEK * kEwPad;                         // and [32 edges][68] dy tile
  for (int e = tid; e < kEC * kEC; e += 512) w2s[(e >> 6) * kEwPad + (e & 63)] = W2[e];
  if (tid < 2 * kEC) cst[tid] = c0c1[tid];
  __syncthreads();
  const long gw = (long)blockIdx.x * 8 + wave, nw = (long)gridDim.x * 8;  // 8 waves: two per SIMD
  f32x16 dw[2][2];  // dW2 tile [ot][ct]: rows = o (lanes of A), cols = c
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) dw[a][c] = zero16();

  for (long p = gw; p < npoints; p += nw) {
    const long cloud = p / N;
    const int j = nn[p * kEK + lo];
    const float* arow = ap + p * kEC;
    const float* brow = bp + (cloud * N + j) * kEC;
    // B operand of y^T: lane (edge k = lo, h) holds h[k][32 h + kk]; also parked in LDS as h[k][c]
    float hv[32];
#pragma unroll
    for (int q4 = 0; q4 < 8; ++q4) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 32 * h + 4 * q4);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(brow + 32 * h + 4 * q4);
      f32x4 h4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h4[e] = lrelu(a4[e] + b4[e]);
        hv[4 * q4 + e] = h4[e];
      }
      *reinterpret_cast<f32x4*>(hts + lo * kEwPad + 32 * h + 4 * q4) = h4;
    }
    // y^T tile ot: D[row = o][col = edge] = sum_c W2[o][c] h[edge][c]; A = W2 rows from LDS
    f32x16 yt[2] = {zero16(), zero16()};
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
      const f32x4* wp = reinterpret_cast<const f32x4*>(w2s + (32 * ot + lo) * kEwPad + 32 * h);
#pragma unroll
      for (int q4 = 0; q4 < 8; ++q4) {
        const f32x4 w4 = wp[q4];
#pragma unroll
        for (int e = 0; e < 4; ++e) yt[ot] = mfma32(w4[e], hv[4 * q4 + e], yt[ot]);
      }
    }
    // dy^T in place: register r of tile ot <-> channel o = 32 ot + crow(r, h), lane <-> edge
    f32x16 dyt[2];
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uchar4 k4 = *reinterpret_cast<const uchar4*>(kext + p * kEC + 32 * ot + 8 * g + 4 * h);
        const int kk4[4] = {k4.x, k4.y, k4.z, k4.w};
        const f32x4 sdv4 = *reinterpret_cast<const f32x4*>(sdv + p * kEC + 32 * ot + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const int o = 32 * ot + 8 * g + 4 * h + e;
          const float y = yt[ot][r];
          dyt[ot][r] = fmaf(cst[kEC + o], y, cst[o]) + (lo == kk4[e] ? sdv4[e] : 0.f);  // edges = lanes here
        }
      }
      // park dy as [edge][o] for the dW2 product
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 d4 = {dyt[ot][4 * g], dyt[ot][4 * g + 1], dyt[ot][4 * g + 2], dyt[ot][4 * g + 3]};
        *reinterpret_cast<f32x4*>(dys + lo * kEwPad + 32 * ot + 8 * g + 4 * h) = d4;
      }
    }
    // dh^T tile ct: D[row = c][col = edge] = sum_o W2[o][c] dy[edge][o]; reduced index o = the register axis of dy^T
    f32x16 dht[2] = {zero16(), zero16()};
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float* wrow = w2s + (32 * ot + crow(t, h)) * kEwPad + lo;
        dht[0] = mfma32(wrow[0], dyt[ot][t], dht[0]);
        dht[1] = mfma32(wrow[32], dyt[ot][t], dht[1]);
      }
    }
    // du = dh * LReLU'(u), u = a' + b' at channel c = 32 ct + crow(r, h); written per edge
    float* durow = du + (p * kEK + lo) * kEC;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      float rs = 0.f;  // lanes 16..31 of each half: the point's sum over its 32 edges for channel 32 ct + crow(lo - 16, h)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 32 * ct + 8 * g + 4 * h);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(brow + 32 * ct + 8 * g + 4 * h);
        f32x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = dht[ct][4 * g + e] * ((a4[e] + b4[e]) > 0.f ? 1.f : 0.2f);
        *reinterpret_cast<f32x4*>(durow + 32 * ct + 8 * g + 4 * h) = o4;
        if (dusum) {  // (uniform) the edges are the lanes: a fixed DPP butterfly over the 32 lanes of the half
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = half_wave_sum(o4[e]);
            rs = (lo == 16 + 4 * g + e) ? t : rs;
          }
        }
      }
      if (dusum && lo >= 16) dusum[p * kEC + 32 * ct + crow(lo - 16, h)] = rs;
    }
    // dW2[o][c] += sum_edge dy[edge][o] h[edge][c]: both operands read from the wave's LDS tiles by rows
    // (edge pair of MFMA step t = rows crow(t,0), crow(t,1)); same-wave LDS traffic needs no barrier
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float* dr = dys + crow(t, h) * kEwPad + lo;
      const float* hr = hts + crow(t, h) * kEwPad + lo;
      const float d0 = dr[0], d1 = dr[32], h0 = hr[0], h1 = hr[32];
      dw[0][0] = mfma32(d0, h0, dw[0][0]);
      dw[0][1] = mfma32(d0, h1, dw[0][1]);
      dw[1][0] = mfma32(d1, h0, dw[1][0]);
      dw[1][1] = mfma32(d1, h1, dw[1][1]);
    }
  }
  // per-wave dW2 partial (64 x 64): tile [ot][ct] register r, lane (c = lo, h) <-> o = 32 ot + crow(r,h)
  if (gw < nw) {
    float* outp = dw2part + gw * kEC * kEC;
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) outp[(32 * ot + crow(r, h)) * kEC + 32 * ct + lo] = dw[ot][ct][r];
  }
}


// ------------------------------------------------------------------------------------------------
// The same two sweeps on the bf16 matrix cores with split fp32 operands (tri_dev.h: three bf16 planes per operand, six
// products per k-step: fp32-equivalent sums at 32 cycles per instruction where v_mfma_f32_32x32x2_f32 takes 64 for an
// eighth of the depth).  W2 waits in LDS as operand images:
//   image 1, fragment (ot, ks): lane (o = lo, half h) holds W2[32 ot + lo][32 h + 8 ks + e], e = 0..7 -- the channels a
//            lane of the edge tile holds in hv[8 ks + e] (the contraction order is free as long as both operands
//            agree, so the hidden vector never moves between lanes);
//   image 2, fragment (ct, 2 ot + gp): lane (c = lo, half h) holds W2[o(i)][32 ct + lo], o(i) = 32 ot + 16 gp +
//            8 (i >> 2) + 4 h + (i & 3): the output channels that registers 8 gp .. 8 gp + 7 of an accumulator tile
//            stand for (accumulator-as-operand: dy^T goes into the next product without leaving its lanes).
// The backward forms y in BOTH orientations (the operands swapped: 48 more instructions on a pipe that is not the
// bound): y^T[o][edge] feeds dh^T = W2^T dy^T as before, y[edge][o] puts the edges of dy on the register axis -- the
// A operand of dW2 += dy^T h as it stands.  Only the h tile still crosses LDS (column reads for dW2's B operand).
// ------------------------------------------------------------------------------------------------
constexpr int kEImg = 8 * 3 * 1024;  // bytes of one W2 image: 8 fragments x 3 planes x 64 lanes x 16 B

constexpr bool kEdgeDuo = SAMBLE_EDGE_DUO != 0;
// a wave-uniform float into a scalar register (the backward sweep sits at the 256-register limit of two waves per SIMD)
__device__ __forceinline__ float uniform_f(float x) {
  return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(x)));
}
struct EdgeDuoFrag {
  u32x4 h, l;
};
__device__ __forceinline__ EdgeDuoFrag edge_frag_duo(const char* img, int frag, int lane) {
  const u32x4* pp = reinterpret_cast<const u32x4*>(img + frag * 3072) + lane;
  return EdgeDuoFrag{pp[0], pp[64]};
}
// both images as two fp16 planes of W2 x 2^e (in the slots of the first two planes); returns 2^-e (uniform over the
// workgroup; `red`: 8 words of LDS scratch).  512 threads, thread = (fragment, lane) as edge_build_images
__device__ __forceinline__ float edge_build_images_duo(const float* __restrict__ W2, char* img1, char* img2, int tid,
                                                       bool want2, float* red) {
  const int frag = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  float v1[8], v2[8];
  {
    const int ot = frag >> 2, ks = frag & 3;
    const float* src = W2 + (32 * ot + lo) * kEC + 32 * h + 8 * ks;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v1[e] = a0[e];
      v1[4 + e] = a1[e];
    }
  }
  float amax = 0.f;  // image 1's fragments cover every element of W2 exactly once
#pragma unroll
  for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v1[e]));
  amax = wave_amax64(amax);
  if (lane == 0) red[frag] = amax;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) amax = fmaxf(amax, red[k]);
  float sw, inv_w;
  duo_scale_for(amax, sw, inv_w);
  {
    u32x4 hp, lp;
    duo_split8s(v1, sw, hp, lp);
    u32x4* d = reinterpret_cast<u32x4*>(img1 + frag * 3072) + lane;
    d[0] = hp;
    d[64] = lp;
  }
  if (want2) {
    const int ct = frag >> 2, ot = (frag >> 1) & 1, gp = frag & 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) v2[i] = W2[(32 * ot + 16 * gp + 8 * (i >> 2) + 4 * h + (i & 3)) * kEC + 32 * ct + lo];
    u32x4 hp, lp;
    duo_split8s(v2, sw, hp, lp);
    u32x4* d = reinterpret_cast<u32x4*>(img2 + frag * 3072) + lane;
    d[0] = hp;
    d[64] = lp;
  }
  return inv_w;
}

__device__ __forceinline__ Tri edge_frag(const char* img, int frag, int lane) {
  const u32x4* pp = reinterpret_cast<const u32x4*>(img + frag * 3072) + lane;
  return Tri{pp[0], pp[64], pp[128]};
}

// 512 threads: thread = (fragment, lane) of each image
__device__ __forceinline__ void edge_build_images(const float* __restrict__ W2, char* img1, char* img2, int tid, bool want2) {
  const int frag = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  {
    const int ot = frag >> 2, ks = frag & 3;
    const float* src = W2 + (32 * ot + lo) * kEC + 32 * h + 8 * ks;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    const Tri t = tri_split8(v);
    u32x4* d = reinterpret_cast<u32x4*>(img1 + frag * 3072) + lane;
    d[0] = t.h;
    d[64] = t.m;
    d[128] = t.l;
  }
  if (want2) {
    const int ct = frag >> 2, ot = (frag >> 1) & 1, gp = frag & 1;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = W2[(32 * ot + 16 * gp + 8 * (i >> 2) + 4 * h + (i & 3)) * kEC + 32 * ct + lo];
    const Tri t = tri_split8(v);
    u32x4* d = reinterpret_cast<u32x4*>(img2 + frag * 3072) + lane;
    d[0] = t.h;
    d[64] = t.m;
    d[128] = t.l;
  }
}

__global__ __launch_bounds__(512, 2) void edge_mlp_fwd_tri_kernel(const float* __restrict__ ap, const float* __restrict__ bp,
                                                                  const int* __restrict__ nn,
                                                                  const float* __restrict__ W2, int N, long npoints,
                                                                  float* __restrict__ ymax, float* __restrict__ ymin,
                                                                  unsigned char* __restrict__ kmax,
                                                                  unsigned char* __restrict__ kmin,
                                                                  double* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) char img1[kEImg];
  __shared__ float red8[8];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  const long gw = (long)blockIdx.x * 8 + wave, nw = (long)gridDim.x * 8;
  float inv_w = 1.f;
  if (kEdgeDuo) inv_w = edge_build_images_duo(W2, img1, nullptr, tid, false, red8);
  else edge_build_images(W2, img1, nullptr, tid, false);
  __syncthreads();
  double s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0};
  for (long p = gw; p < npoints; p += nw) {
    const long cloud = p / N;
    const int j = nn[p * kEK + lo];
    const f32x4* av = reinterpret_cast<const f32x4*>(ap + p * kEC + 32 * h);
    const f32x4* bv = reinterpret_cast<const f32x4*>(bp + (cloud * N + j) * kEC + 32 * h);
    // D[row = edge][col = out channel] = sum_c h[edge][c] W2[o][c]
    f32x16 acc[2] = {zero16(), zero16()};
    if (kEdgeDuo) {
      float hv[32];
      float amax = 0.f;
#pragma unroll
      for (int q4 = 0; q4 < 8; ++q4) {
        const f32x4 a4 = av[q4], b4 = bv[q4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hv[4 * q4 + e] = lrelu(a4[e] + b4[e]);
          amax = fmaxf(amax, fabsf(hv[4 * q4 + e]));
        }
      }
      float sh, inv_h;
      duo_scale_for(wave_amax64(amax), sh, inv_h);   // one scale for the point's 32 x 64 tile of hidden vectors
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const float v[8] = {hv[8 * ks], hv[8 * ks + 1], hv[8 * ks + 2], hv[8 * ks + 3],
                            hv[8 * ks + 4], hv[8 * ks + 5], hv[8 * ks + 6], hv[8 * ks + 7]};
        u32x4 hh, hl;
        duo_split8s(v, sh, hh, hl);
        const EdgeDuoFrag w0 = edge_frag_duo(img1, ks, lane), w1 = edge_frag_duo(img1, 4 + ks, lane);
        acc[0] = mfma_duo(hh, hl, w0.h, w0.l, acc[0]);
        acc[1] = mfma_duo(hh, hl, w1.h, w1.l, acc[1]);
      }
      const float sc = inv_h * inv_w;   // exact: powers of two
#pragma unroll
      for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ot][r] *= sc;
    } else {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 a0 = av[2 * ks], a1 = av[2 * ks + 1], b0 = bv[2 * ks], b1 = bv[2 * ks + 1];
      const float v[8] = {lrelu(a0[0] + b0[0]), lrelu(a0[1] + b0[1]), lrelu(a0[2] + b0[2]), lrelu(a0[3] + b0[3]),
                          lrelu(a1[0] + b1[0]), lrelu(a1[1] + b1[1]), lrelu(a1[2] + b1[2]), lrelu(a1[3] + b1[3])};
      const Tri ht = tri_split8(v);
      acc[0] = mfma_tri(ht, edge_frag(img1, ks, lane), acc[0]);
      acc[1] = mfma_tri(ht, edge_frag(img1, 4 + ks, lane), acc[1]);
    }
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
      float mx = acc[ot][0], mn = acc[ot][0], s = 0.f, q = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        mx = fmaxf(mx, acc[ot][r]);
        mn = fminf(mn, acc[ot][r]);
        s += acc[ot][r];
        q = fmaf(acc[ot][r], acc[ot][r], q);
      }
      mx = fmaxf(mx, wave_xor32(mx));
      mn = fminf(mn, wave_xor32(mn));
      s += wave_xor32(s);
      q += wave_xor32(q);
      // the edge that attains the extremum (first one on ties): the backward routes the gradient by index
      int kx = 99, kn = 99;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        kx = min(kx, acc[ot][r] == mx ? crow(r, h) : 99);
        kn = min(kn, acc[ot][r] == mn ? crow(r, h) : 99);
      }
      kx = xor32_min(kx);
      kn = xor32_min(kn);
      if (h == 0) {
        ymax[p * kEC + 32 * ot + lo] = mx;
        ymin[p * kEC + 32 * ot + lo] = mn;
        kmax[p * kEC + 32 * ot + lo] = (unsigned char)kx;
        kmin[p * kEC + 32 * ot + lo] = (unsigned char)kn;
      }
      s1[ot] += (double)s;
      s2[ot] += (double)q;
    }
  }
  if (h == 0 && gw < nw) {
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
      part[(gw * 2 + 0) * kEC + 32 * ot + lo] = s1[ot];
      part[(gw * 2 + 1) * kEC + 32 * ot + lo] = s2[ot];
    }
  }
}

constexpr int kETriBwdLds = 2 * kEImg + 2 * kEC * 4 + 8 * kEK * kEwPad * 4;  // 119 KB: one workgroup of 8 waves per CU

#ifdef SAMBLE_STAMPS  // scratch builds only (tools/bench_edge_mlp.py --stamps): workgroup 0, every wave, its 3rd and 4th point
__device__ unsigned long long g_edge_stamps[8 * 2 * 10];
#define EDGE_STAMP(i)                                                                                           \
  do {                                                                                                          \
    if (blockIdx.x == 0 && lane == 0 && (pcount == 2 || pcount == 3))                                           \
      g_edge_stamps[(wave * 2 + (pcount - 2)) * 10 + (i)] = __builtin_amdgcn_s_memtime();                       \
  } while (0)
#else
#define EDGE_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(512) void edge_mlp_bwd_tri_kernel(const float* __restrict__ ap, const float* __restrict__ bp,
                                                               const int* __restrict__ nn,
                                                               const float* __restrict__ W2,
                                                               const unsigned char* __restrict__ kext,
                                                               const float* __restrict__ sdv,
                                                               const float* __restrict__ c0c1, int N, long npoints,
                                                               float* __restrict__ du, float* __restrict__ dusum,
                                                               float* __restrict__ dw2part) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  char* img1 = smem_c;
  char* img2 = smem_c + kEImg;
  float* cst = reinterpret_cast<float*>(smem_c + 2 * kEImg);  // c0[64], c1[64]
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  float* hts = cst + 2 * kEC + wave * (kEK * kEwPad);  // this wave's [32 edges][68] h tile
  float inv_w = 1.f;
  if (kEdgeDuo) inv_w = uniform_f(edge_build_images_duo(W2, img1, img2, tid, true, cst));  // (cst as scratch: rewritten below)
  else edge_build_images(W2, img1, img2, tid, true);
  __syncthreads();
  if (tid < 2 * kEC) cst[tid] = c0c1[tid];
  __syncthreads();
  const long gw = (long)blockIdx.x * 8 + wave, nw = (long)gridDim.x * 8;  // 8 waves: two per SIMD
  f32x16 dw[2][2];  // dW2 tile [ot][ct]: rows = o, cols = c
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) dw[a][c] = zero16();

  int pcount = -1;
  float sh_run = 1.2676506e30f, sdy_run = 1.2676506e30f;   // 2^100: duo_scale_for's largest
  if (kEdgeDuo) {
  // ---- the same sweep on two fp16 planes: one scale for the point's tile of hidden vectors (s_h), one for its tile of dy
  // (s_dy: y^T and y hold the same 32 x 64 values, one maximum serves both layouts), W2's from the image build
  for (long p = gw; p < npoints; p += nw) {
    ++pcount;
    // (the W2 fragments are loop-invariant LDS reads: 16 x 8 registers that the compiler would otherwise hoist out of the
    // loop and then spill -- 119 spills; they are to be read where they are used)
    asm volatile("" ::: "memory");
    EDGE_STAMP(0);
    const long cloud = p / N;
    const int j = nn[p * kEK + lo];
    const float* arow = ap + p * kEC;
    const float* brow = bp + (cloud * N + j) * kEC;
    float amax = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {  // the tile goes to LDS as it is formed; its split waits for the tile's maximum
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(arow + 32 * h + 8 * ks), a1 = *reinterpret_cast<const f32x4*>(arow + 32 * h + 8 * ks + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(brow + 32 * h + 8 * ks), b1 = *reinterpret_cast<const f32x4*>(brow + 32 * h + 8 * ks + 4);
      const f32x4 h0 = {lrelu(a0[0] + b0[0]), lrelu(a0[1] + b0[1]), lrelu(a0[2] + b0[2]), lrelu(a0[3] + b0[3])};
      const f32x4 h1 = {lrelu(a1[0] + b1[0]), lrelu(a1[1] + b1[1]), lrelu(a1[2] + b1[2]), lrelu(a1[3] + b1[3])};
      *reinterpret_cast<f32x4*>(hts + lo * kEwPad + 32 * h + 8 * ks) = h0;
      *reinterpret_cast<f32x4*>(hts + lo * kEwPad + 32 * h + 8 * ks + 4) = h1;
#pragma unroll
      for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(h0[e]), fabsf(h1[e])));
    }
    float s_h, inv_h;
    duo_scale_for(wave_amax64(amax), s_h, inv_h);
    s_h = uniform_f(s_h);
    inv_h = uniform_f(inv_h);
    // y in both orientations, ONE AT A TIME (together with the dW2 accumulators they do not fit 256 registers beside
    // the operands: 111 spills; the price is the hidden tile split twice): first y[edge][o] = D[row = edge][col = o]
    f32x16 yt2[2] = {zero16(), zero16()};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(hts + lo * kEwPad + 32 * h + 8 * ks);       // this lane's own row
      const f32x4 h1 = *reinterpret_cast<const f32x4*>(hts + lo * kEwPad + 32 * h + 8 * ks + 4);
      const float v[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      u32x4 hh, hl;
      duo_split8s(v, s_h, hh, hl);
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) {
        const EdgeDuoFrag w = edge_frag_duo(img1, 4 * ot + ks, lane);
        yt2[ot] = mfma_duo(hh, hl, w.h, w.l, yt2[ot]);
      }
    }
    EDGE_STAMP(1);
    __builtin_amdgcn_sched_barrier(0);
    // dy = c0 + c1 y + [edge == kext] sdv in place; y = accumulator x 2^-(e_h + e_w)
    const float sc_y = uniform_f(inv_h * inv_w);
    float dmax = 0.f;
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {  // yt2: register r <-> edge crow(r, h), lane <-> channel o = 32 ot + lo
      const int o = 32 * ot + lo;
      const int ke = kext[p * kEC + o];
      const float sv = sdv[p * kEC + o], c0 = cst[o], c1 = cst[kEC + o];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        yt2[ot][r] = fmaf(c1, yt2[ot][r] * sc_y, c0) + (crow(r, h) == ke ? sv : 0.f);
        dmax = fmaxf(dmax, fabsf(yt2[ot][r]));
      }
    }
    float s_dy, inv_dy;
    duo_scale_for(wave_amax64(dmax), s_dy, inv_dy);   // (y^T below holds the same 32 x 64 values: one maximum serves both)
    s_dy = uniform_f(s_dy);
    inv_dy = uniform_f(inv_dy);
    EDGE_STAMP(2);
    __builtin_amdgcn_sched_barrier(0);
    // dW2[o][c] += sum_edge dy[edge][o] h[edge][c] ACCUMULATES over the wave's points, so its two operands run under
    // scales that only ever shrink: the smallest (= for the largest tile so far) of the points' scales.  When a point
    // brings a larger tile the accumulators are rescaled by the exact power of two (uniform over the wave, rare after
    // the first few points); a point with a smaller tile is split under a scale larger tiles chose -- what that loses is
    // relative to a sum those tiles dominate
    if (s_h < sh_run || s_dy < sdy_run) {
      const float ratio = uniform_f((fminf(s_h, sh_run) / sh_run) * (fminf(s_dy, sdy_run) / sdy_run));
      sh_run = uniform_f(fminf(s_h, sh_run));
      sdy_run = uniform_f(fminf(s_dy, sdy_run));
#pragma unroll
      for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 16; ++r) dw[ot][ct][r] *= ratio;
    }
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      u32x4 bqh[2], bql[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = hts[(16 * kp + 8 * (i >> 2) + 4 * h + (i & 3)) * kEwPad + 32 * ct + lo];
        duo_split8s(v, sh_run, bqh[ct], bql[ct]);
      }
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = yt2[ot][8 * kp + i];
        u32x4 ah, al;
        duo_split8s(v, sdy_run, ah, al);
        dw[ot][0] = mfma_duo(ah, al, bqh[0], bql[0], dw[ot][0]);
        dw[ot][1] = mfma_duo(ah, al, bqh[1], bql[1], dw[ot][1]);
      }
    }
    EDGE_STAMP(3);
    __builtin_amdgcn_sched_barrier(0);
    // ... then y^T[o][edge] = D[row = o][col = edge] and its dy
    f32x16 yt1[2] = {zero16(), zero16()};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(hts + lo * kEwPad + 32 * h + 8 * ks);
      const f32x4 h1 = *reinterpret_cast<const f32x4*>(hts + lo * kEwPad + 32 * h + 8 * ks + 4);
      const float v[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      u32x4 hh, hl;
      duo_split8s(v, s_h, hh, hl);
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) {
        const EdgeDuoFrag w = edge_frag_duo(img1, 4 * ot + ks, lane);
        yt1[ot] = mfma_duo(w.h, w.l, hh, hl, yt1[ot]);
      }
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // yt1: register r <-> channel o = 32 ot + crow(r, h), lane <-> edge
        const uchar4 k4 = *reinterpret_cast<const uchar4*>(kext + p * kEC + 32 * ot + 8 * g + 4 * h);
        const int kk4[4] = {k4.x, k4.y, k4.z, k4.w};
        const f32x4 sdv4 = *reinterpret_cast<const f32x4*>(sdv + p * kEC + 32 * ot + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const int o = 32 * ot + 8 * g + 4 * h + e;
          yt1[ot][r] = fmaf(cst[kEC + o], yt1[ot][r] * sc_y, cst[o]) + (lo == kk4[e] ? sdv4[e] : 0.f);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // dh^T tile ct: D[row = c][col = edge] = sum_o W2[o][c] dy[edge][o]; k-step (ot, gp) = registers 8 gp .. + 7 of yt1[ot]
    f32x16 dht[2] = {zero16(), zero16()};
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = yt1[ot][8 * gp + i];
        u32x4 bh, bl;
        duo_split8s(v, s_dy, bh, bl);
        const EdgeDuoFrag w0 = edge_frag_duo(img2, 2 * ot + gp, lane), w1 = edge_frag_duo(img2, 4 + 2 * ot + gp, lane);
        dht[0] = mfma_duo(w0.h, w0.l, bh, bl, dht[0]);
        dht[1] = mfma_duo(w1.h, w1.l, bh, bl, dht[1]);
      }
    }
    EDGE_STAMP(4);
    __builtin_amdgcn_sched_barrier(0);   // (phases stay apart: hoisted across them, the next phase's operands spill)
    // du = dh * LReLU'(u) (the scales of dh ride on the two slopes), written per edge; see the three-plane sweep below
    const float slope1 = uniform_f(inv_w * inv_dy), slope02 = uniform_f(0.2f * inv_w * inv_dy);
    float* durow0 = du + p * (kEK * kEC);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 h4 = *reinterpret_cast<const f32x4*>(hts + lo * kEwPad + 32 * ct + 8 * g + 4 * h);
        f32x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = dht[ct][4 * g + e] * (h4[e] > 0.f ? slope1 : slope02);
        *reinterpret_cast<f32x4*>(hts + lo * kEwPad + 32 * ct + 8 * g + 4 * h) = o4;
      }
    }
    EDGE_STAMP(5);
    __builtin_amdgcn_sched_barrier(0);   // (phases stay apart: hoisted across them, the next phase's operands spill)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(hts + (4 * it + (lane >> 4)) * kEwPad + 4 * (lane & 15));
      *reinterpret_cast<f32x4*>(durow0 + 256 * it + 4 * lane) = v;
    }
    EDGE_STAMP(6);
    if (dusum) {  // (uniform)
      float rs = 0.f;
#pragma unroll
      for (int e = 0; e < kEK; ++e) rs += hts[e * kEwPad + lane];
      dusum[p * kEC + lane] = rs;
    }
    EDGE_STAMP(7);
  }
  } else
  for (long p = gw; p < npoints; p += nw) {
    ++pcount;
    EDGE_STAMP(0);
    const long cloud = p / N;
    const int j = nn[p * kEK + lo];
    const float* arow = ap + p * kEC;
    const float* brow = bp + (cloud * N + j) * kEC;
    // y in both orientations: yt1 = D[row = o][col = edge], yt2 = D[row = edge][col = o]
    f32x16 yt1[2] = {zero16(), zero16()}, yt2[2] = {zero16(), zero16()};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(arow + 32 * h + 8 * ks), a1 = *reinterpret_cast<const f32x4*>(arow + 32 * h + 8 * ks + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(brow + 32 * h + 8 * ks), b1 = *reinterpret_cast<const f32x4*>(brow + 32 * h + 8 * ks + 4);
      const f32x4 h0 = {lrelu(a0[0] + b0[0]), lrelu(a0[1] + b0[1]), lrelu(a0[2] + b0[2]), lrelu(a0[3] + b0[3])};
      const f32x4 h1 = {lrelu(a1[0] + b1[0]), lrelu(a1[1] + b1[1]), lrelu(a1[2] + b1[2]), lrelu(a1[3] + b1[3])};
      *reinterpret_cast<f32x4*>(hts + lo * kEwPad + 32 * h + 8 * ks) = h0;
      *reinterpret_cast<f32x4*>(hts + lo * kEwPad + 32 * h + 8 * ks + 4) = h1;
      const float v[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      const Tri ht = tri_split8(v);
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) {
        const Tri w = edge_frag(img1, 4 * ot + ks, lane);
        yt1[ot] = mfma_tri(w, ht, yt1[ot]);
        yt2[ot] = mfma_tri(ht, w, yt2[ot]);
      }
    }
    EDGE_STAMP(1);
    // dy = c0 + c1 y + [edge == kext] sdv, in place, in both layouts
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // yt1: register r <-> channel o = 32 ot + crow(r, h), lane <-> edge
        const uchar4 k4 = *reinterpret_cast<const uchar4*>(kext + p * kEC + 32 * ot + 8 * g + 4 * h);
        const int kk4[4] = {k4.x, k4.y, k4.z, k4.w};
        const f32x4 sdv4 = *reinterpret_cast<const f32x4*>(sdv + p * kEC + 32 * ot + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const int o = 32 * ot + 8 * g + 4 * h + e;
          yt1[ot][r] = fmaf(cst[kEC + o], yt1[ot][r], cst[o]) + (lo == kk4[e] ? sdv4[e] : 0.f);
        }
      }
      {  // yt2: register r <-> edge crow(r, h), lane <-> channel o = 32 ot + lo
        const int o = 32 * ot + lo;
        const int ke = kext[p * kEC + o];
        const float sv = sdv[p * kEC + o], c0 = cst[o], c1 = cst[kEC + o];
#pragma unroll
        for (int r = 0; r < 16; ++r) yt2[ot][r] = fmaf(c1, yt2[ot][r], c0) + (crow(r, h) == ke ? sv : 0.f);
      }
    }
    EDGE_STAMP(2);
    // (dW2 BEFORE dh: y[edge][o] dies here instead of living through the dh product -- 548 -> 468 us alone, round 4)
    // dW2[o][c] += sum_edge dy[edge][o] h[edge][c]: A = registers 8 kp .. + 7 of yt2[ot] (edges 16 kp + 8 (i >> 2) + 4 h +
    // (i & 3)), B = the same edges of the h tile's column c (same-wave LDS traffic needs no barrier)
#pragma unroll
    for (int kp = 0; kp < ((SAMBLE_EDGE_ABL & 2) ? 0 : 2); ++kp) {
      Tri bq[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = hts[(16 * kp + 8 * (i >> 2) + 4 * h + (i & 3)) * kEwPad + 32 * ct + lo];
        bq[ct] = tri_split8(v);
      }
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = yt2[ot][8 * kp + i];
        const Tri a = tri_split8(v);
        dw[ot][0] = mfma_tri(a, bq[0], dw[ot][0]);
        dw[ot][1] = mfma_tri(a, bq[1], dw[ot][1]);
      }
    }
    EDGE_STAMP(3);
    // dh^T tile ct: D[row = c][col = edge] = sum_o W2[o][c] dy[edge][o]; k-step (ot, gp) = registers 8 gp .. + 7 of yt1[ot]
    f32x16 dht[2] = {zero16(), zero16()};
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = yt1[ot][8 * gp + i];
        const Tri bq = tri_split8(v);
        if (SAMBLE_EDGE_ABL & 4) {
          dht[0][gp] += __uint_as_float(bq.h[0] ^ bq.m[1] ^ bq.l[2]);
          continue;
        }
        dht[0] = mfma_tri(edge_frag(img2, 2 * ot + gp, lane), bq, dht[0]);
        dht[1] = mfma_tri(edge_frag(img2, 4 + 2 * ot + gp, lane), bq, dht[1]);
      }
    }
    EDGE_STAMP(4);
    // du = dh * LReLU'(u), u = a' + b' at channel c = 32 ct + crow(r, h); written per edge.  The point's sum over its 32
    // edges (dusum): the tile goes through the wave's LDS tile (the h tile has been read by now) and lane = channel adds
    // its column in edge order -- 72 instructions where the DPP butterflies over the lanes took 400
    float* durow0 = du + p * (kEK * kEC);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // LReLU'(u) from the sign of h = LReLU(u), this edge's row of the h tile (then overwritten by du: same lane, same
        // address, LDS operations of a wave execute in order)
        const f32x4 h4 = *reinterpret_cast<const f32x4*>(hts + lo * kEwPad + 32 * ct + 8 * g + 4 * h);
        f32x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = dht[ct][4 * g + e] * (h4[e] > 0.f ? 1.f : 0.2f);
        *reinterpret_cast<f32x4*>(hts + lo * kEwPad + 32 * ct + 8 * g + 4 * h) = o4;
      }
    }
    EDGE_STAMP(5);
    // the point's 32 x 64 block of du is 8 KB of consecutive addresses: out of the tile as whole lines, 1 KB per
    // instruction (16-byte pieces straight from the accumulator lanes would touch 32 lines per instruction, a quarter
    // of each)
    if (!(SAMBLE_EDGE_ABL & 1)) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(hts + (4 * it + (lane >> 4)) * kEwPad + 4 * (lane & 15));
        *reinterpret_cast<f32x4*>(durow0 + 256 * it + 4 * lane) = v;
      }
    }
    EDGE_STAMP(6);
    if (dusum && !(SAMBLE_EDGE_ABL & 8)) {  // (uniform)
      float rs = 0.f;
#pragma unroll
      for (int e = 0; e < kEK; ++e) rs += hts[e * kEwPad + lane];
      dusum[p * kEC + lane] = rs;
    }
    EDGE_STAMP(7);
  }
  // per-wave dW2 partial (64 x 64): tile [ot][ct] register r, lane (c = lo, h) <-> o = 32 ot + crow(r,h)
  if (gw < nw) {
    const float un = kEdgeDuo ? (1.f / sh_run) * (1.f / sdy_run) : 1.f;   // (two steps: the product of the scales may not be a float)
    float* outp = dw2part + gw * kEC * kEC;
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) outp[(32 * ot + crow(r, h)) * kEC + 32 * ct + lo] = dw[ot][ct][r] * un;
  }
}

}  // namespace samble

using namespace samble;

#ifdef SAMBLE_STAMPS
extern "C" __attribute__((visibility("default"))) int samble_scratch_edge_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(samble::g_edge_stamps), sizeof(unsigned long long) * 160);
}
#endif

extern "C" int samble_edge_waves(void) { return 2048; }  // persistent waves per sweep (512 workgroups x 4)

extern "C" int samble_launch_edge_gather_sums(const float* bp, const int* nn, int B, int N, float* S, float* Q,
                                              hipStream_t s) {
  const long np = (long)B * N;
  Timed timed(kT_edge_sums, s);
  hipLaunchKernelGGL(edge_gather_sums_kernel, dim3(2048), dim3(256), 0, s, bp, nn, N, np, S, Q);
  return (int)hipGetLastError();
}

extern "C" int samble_launch_edge_mlp_fwd(const float* ap, const float* bp, const int* nn, const float* W2, int B, int N,
                                          float* ymax, float* ymin, unsigned char* kmax, unsigned char* kmin,
                                          double* part, hipStream_t s) {
  const long np = (long)B * N;
  Timed timed(kT_edge_fwd, s);
  if (SAMBLE_EDGE_F32)
    hipLaunchKernelGGL(edge_mlp_fwd_kernel, dim3(samble_edge_waves() / 4), dim3(256), 0, s, ap, bp, nn, W2, N, np, ymax, ymin,
                       kmax, kmin, part);
  else
    hipLaunchKernelGGL(edge_mlp_fwd_tri_kernel, dim3(samble_edge_waves() / 8), dim3(512), 0, s, ap, bp, nn, W2, N, np, ymax,
                       ymin, kmax, kmin, part);
  return (int)hipGetLastError();
}

extern "C" int samble_launch_edge_mlp_bwd(const float* ap, const float* bp, const int* nn, const float* W2,
                                          const unsigned char* yext, const float* sdv, const float* c0c1, int B, int N, float* du,
                                          float* dusum, float* dw2part, hipStream_t s) {
  const long np = (long)B * N;
  const size_t lds = (size_t)(kEC * kEwPad + 2 * kEC + 8 * 2 * kEK * kEwPad) * sizeof(float);  // 157 KB: one workgroup per CU
  {  // per call: cheap, and correct for every device / thread (no process-wide 'done' flag)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(edge_mlp_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(edge_mlp_bwd_tri_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kETriBwdLds);
    if (e != hipSuccess) return (int)e;
  }
  Timed timed(kT_edge_bwd, s);
  if (SAMBLE_EDGE_F32)
    hipLaunchKernelGGL(edge_mlp_bwd_kernel, dim3(samble_edge_waves() / 8), dim3(512), lds, s, ap, bp, nn, W2, yext, sdv, c0c1,
                       N, np, du, dusum, dw2part);
  else
    hipLaunchKernelGGL(edge_mlp_bwd_tri_kernel, dim3(samble_edge_waves() / 8), dim3(512), kETriBwdLds, s, ap, bp, nn, W2, yext,
                       sdv, c0c1, N, np, du, dusum, dw2part);
  return (int)hipGetLastError();
}
