// 1x1 convolutions over 128 input channels around the neighbour / sampler kernels, on the bf16 matrix cores with split
// fp32 operands (tri_dev.h: six partial products per fp32 product, fp32 accumulation):
//
//   reference                                                       here
//   models/attention.py:187-192  ff = Conv1d 128->512, LeakyReLU(0.2), Conv1d 512->128   lin_fwd (+ leaky), lin_dx;
//                                (and its autograd)                                      backward: lin_fwd (+ mask), lin_dx, lin_dw x 2
//   models/cls_model.py:136      conv_list[i](x).max(dim=-1): Conv1d 128->1024 + max    lin_fwd<AMAX>: the (B, 1024, N) tensor
//                                over the points                                         (268 MB at N=2048) is never written;
//                                (and its autograd: only the arg-max column counts)      backward: amax_bwd (sparse, ordered)
//
// Same tile machinery as the QKV projection (proj_tri.hip), for any output width O that is a multiple of 32:
//   lin_fwd   out[n][o] = sum_c W[o][c] x[c][n]      x channel-major (B,128,N) -> point-major rows (B,N,O); A: W row-image
//             tiles through a ring of 4 LDS slots by LDS-DMA, B: the point's channels split in registers
//   lin_dx    dx[c][n]  = sum_o W[o][c] g[n][o]      point-major (B,N,O) -> channel-major (B,128,N); A: W transposed-image
//             tiles, B: 16 outputs of the point's row per k-step
//   lin_dw    dW[o][c]  = sum_n g[n][o] x[c][n]      per-workgroup partials (256 outputs x 128 channels over 512 points),
//             summed in a fixed order by lin_sum_parts: deterministic, no float atomics
#include "tri_dev.h"

namespace samble {

// Round 5: the products run on TWO fp16 planes per operand (tri_dev.h "duo": 3 matrix instructions per k-step instead of
// 6) under power-of-two scales that are exact to take out again:
//   weights      per 32-row image tile (the tile's largest |w| lands just under 2^13; 2^-e rides in the tile's spare slot);
//   lin_fwd      the point's 128 channels under the point's own scale (contraction complete inside one tile: one scale pair);
//   lin_dx       the point's 32 gradient values of output tile t under their own scale; every tile's product starts from
//                a ZERO accumulator and is added to the fp32 totals by the vector ALU, times 2^-(e_w + e_g) -- the sum
//                over the tiles is an fp32 sum, as in the reference;
//   lin_dw       the same per 32-point tile: g and x blocks under wave-uniform scales, fp32 totals.
// 22 significant bits per operand, products of the planes exact in fp32: measured against float64 like the three-plane
// kernels (tests/test_gpu_linear.py, same tolerances).  -DSAMBLE_LIN_DUO=0 builds the three-bf16-plane kernels (A/B).
// (SAMBLE_LIN_DUO / kLinDuo: tri_dev.h)

#ifndef SAMBLE_LIN_FILL
#define SAMBLE_LIN_FILL 256
#endif
constexpr int kLinFillWgs = SAMBLE_LIN_FILL;   // workgroups a launch should reach before it stops slicing its outputs
constexpr int kLinDepth = 4;
constexpr int kLinLds = kLinDepth * kTriTile;
enum { kLinPlain = 0, kLinLeaky = 1, kLinMask = 2, kLinAmax = 3, kLinLeakyBits = 4, kLinMaskBits = 5 };
constexpr float kLeakySlope = 0.2f;

__device__ __forceinline__ void lin_glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// max over the 32 lanes of each half of the wave (DPP: quad swaps, half-row and row mirrors, row broadcast), delivered in
// lanes 31 and 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float lin_dpp_f(float x, float old) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(x), CTRL,
                                                               ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float half_wave_max(float x) {
  x = fmaxf(x, lin_dpp_f<0xB1, 0xF>(x, x));   // quad_perm [1,0,3,2]
  x = fmaxf(x, lin_dpp_f<0x4E, 0xF>(x, x));   // quad_perm [2,3,0,1]
  x = fmaxf(x, lin_dpp_f<0x141, 0xF>(x, x));  // row_half_mirror
  x = fmaxf(x, lin_dpp_f<0x140, 0xF>(x, x));  // row_mirror: every lane of a row of 16 holds the row's max
  x = fmaxf(x, lin_dpp_f<0x142, 0xA>(x, x));  // row_bcast15 -> rows 1, 3 (lanes of rows 0, 2 keep their value: old = x)
  return x;                                    // lanes 16..31 / 48..63 hold the half's max
}

// max over all 64 lanes, in every lane (four DPP steps inside the rows of 16, v_permlane16_swap / v_permlane32_swap across them)
__device__ __forceinline__ float wave_max_all(float x) {
  x = fmaxf(x, lin_dpp_f<0xB1, 0xF>(x, x));
  x = fmaxf(x, lin_dpp_f<0x4E, 0xF>(x, x));
  x = fmaxf(x, lin_dpp_f<0x141, 0xF>(x, x));
  x = fmaxf(x, lin_dpp_f<0x140, 0xF>(x, x));
  {
    const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(p[0]), __uint_as_float(p[1]));
  }
  return xor32_max(x);
}

// eight fp32 values x s -> the two fp16 planes of one 16-byte operand chunk
__device__ __forceinline__ void duo_split8(const float (&v)[8], float s, u32x4& hp, u32x4& lp) {
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    unsigned hw, lw;
    duo_split2(v[2 * w] * s, v[2 * w + 1] * s, hw, lw);
    hp[w] = hw;
    lp[w] = lw;
  }
}

// EPI  kLinPlain: out = W x            kLinLeaky: out = leaky(W x)
//      kLinMask:  out = (W x) * (ref > 0 ? 1 : 0.2)   (ref: same layout as out -- the activation the forward kept)
//      kLinAmax:  no out: per 32-point tile the maximum of every output row over the tile's points and the first point
//                 that reaches it -> pmax / parg [(b, tile, o)]
//      kLinLeakyBits / kLinMaskBits: the leaky / mask pair with the activation's SIGN carried as one bit per value instead of
//                 re-reading the activation (the backward's mask pass moved 134 MB of `ref` at N = 2048, O = 512 to learn
//                 4 MB of signs): `bits` = uint16 words [(b, output tile, lane half h)][point], bit 4 g + e <-> output
//                 8 g + 4 h + e of the tile (the accumulator's register order), written by LeakyBits (out > 0), read by
//                 MaskBits -- the same products, the same selects: bit for bit the kLinLeaky / kLinMask results
//
// CM (plain epilogue only): the output leaves CHANNEL-major, out[b][o][n] (o_rs = the stride between output channels) --
// the accumulator's columns are the points, so a register's 32 lanes write one 128-byte line; accum: out += W x (the second
// half of a 256-channel input: models/upsample.py:150 `res_conv(cat(up, interpolated))` without the concatenation)
template <int EPI, bool CM = false>
__global__ __launch_bounds__(512, 2) void lin_fwd_tri_kernel(const float* __restrict__ x, long x_bs, int Cin, int N,
                                                             const char* __restrict__ Wimg, int otiles, int O,
                                                             float* __restrict__ out, long o_bs, long o_rs,
                                                             const float* __restrict__ ref, float* __restrict__ pmax,
                                                             int* __restrict__ parg, int accum,
                                                             unsigned short* __restrict__ bits) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  constexpr int D = kLinDepth;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  // points past N-1 are clamped: those lanes recompute and rewrite point N-1's row bit for bit (no predicated store)
  const int n = min(chunk * 256 + wave * 32 + lo, N - 1);
  const bool own = chunk * 256 + wave * 32 + lo < N;   // (CM with accum: clamped lanes do not store -- a read-modify-write)
  // grid.z slices the output tiles (otiles = tiles per slice): small launches (the blocks' coarse levels) fill the chip
  const int t0 = blockIdx.z * otiles;
  Wimg += (long)t0 * kTriTile;
  auto stage = [&](int t) {
    const char* gt = Wimg + (long)min(t, otiles - 1) * kTriTile;
    char* lt = smem_c + (t % D) * kTriTile;
#pragma unroll
    for (int k = 0; k < 3; ++k) lin_glds16(gt + (tid + 512 * k) * 16, lt + (wave * 64 + 512 * k) * 16);
  };
#pragma unroll
  for (int t = 0; t < D - 1; ++t) stage(t);
  // this point's channels as the B operand: k-step ks, half h <-> channels 16 ks + 8 h + e; three bf16 planes
  // (xq[3 ks + piece]) or two fp16 planes under the point's scale (xq[2 ks + plane], x_inv = 2^-e)
  u32x4 xq[kLinDuo ? 16 : 24];
  float x_inv = 1.f;
  if (kLinDuo) {
    float xv[64];
    float amax = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = 16 * ks + 8 * h + e;   // (channels Cin .. 127 do not exist: zeros, as in W's image)
        xv[8 * ks + e] = c < Cin ? x[(long)b * x_bs + (long)c * N + n] : 0.f;
        amax = fmaxf(amax, fabsf(xv[8 * ks + e]));
      }
    amax = xor32_max(amax);   // lanes lo and lo + 32 hold the two halves of one point
    float sx;
    duo_scale_for(amax, sx, x_inv);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const float v[8] = {xv[8 * ks], xv[8 * ks + 1], xv[8 * ks + 2], xv[8 * ks + 3],
                          xv[8 * ks + 4], xv[8 * ks + 5], xv[8 * ks + 6], xv[8 * ks + 7]};
      duo_split8(v, sx, xq[2 * ks], xq[2 * ks + 1]);
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = 16 * ks + 8 * h + e;
        v[e] = c < Cin ? x[(long)b * x_bs + (long)c * N + n] : 0.f;
      }
      const Tri t3 = tri_split8(v);
      xq[3 * ks] = t3.h;
      xq[3 * ks + 1] = t3.m;
      xq[3 * ks + 2] = t3.l;
    }
  }
  // amax (operands swapped: the points are the accumulator's ROWS): the scales of the points crow(r, h) of this wave's tile
  float xinv_r[16];
  if (kLinDuo && EPI == kLinAmax) {
#pragma unroll
    for (int r = 0; r < 16; ++r) xinv_r[r] = __shfl(x_inv, crow(r, h), 64);
  }
  float* orow = (EPI == kLinAmax) ? nullptr
                : CM              ? out + (long)b * o_bs + n + (long)t0 * 32 * o_rs
                                  : out + (long)b * o_bs + (long)n * o_rs + 4 * h + t0 * 32;
  const float* rrow = (EPI == kLinMask) ? ref + (long)b * o_bs + (long)n * o_rs + 4 * h + t0 * 32 : nullptr;
  // sign words of this lane: [(b, tile, h)][point]
  unsigned short* brow = (EPI == kLinLeakyBits || EPI == kLinMaskBits) ? bits + (((long)b * (O / 32) + t0) * 2 + h) * N + n : nullptr;
  unsigned bf = 0, bnx = 0;
  if (EPI == kLinMaskBits) bf = brow[0];
  const long ptile = (long)b * (gridDim.x * 8) + chunk * 8 + wave;   // (b, 32-point tile) of this wave
  const int n_first = chunk * 256 + wave * 32;
  f32x4 rf[4];
  if (EPI == kLinMask) {
#pragma unroll
    for (int g = 0; g < 4; ++g) rf[g] = *reinterpret_cast<const f32x4*>(rrow + 8 * g);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // iteration t: [mask: the activation words of tile t+1], tile t+3 into the slot of tile t-1, the product of tile t, its
  // stores.  VM operations younger than tile t+1's DMA at the end of iteration t (in-order retirement): plain / leaky
  // 4 + 2 x (3 + 4) = 18; mask 4 + 2 x (4 + 3 + 4) = 26; amax 2 + 2 x (3 + 2) = 12; CM 16 + 2 x (3 + 16) = 54 (accum: more).
  // Counting LOW is the safe side.
  float res_m = 0.f;
  int res_a = 0;
  for (int t = 0; t < otiles; ++t) {
    f32x4 rn[4];
    if (EPI == kLinMask) {
      const float* rp = rrow + min(t + 1, otiles - 1) * 32;
#pragma unroll
      for (int g = 0; g < 4; ++g) rn[g] = *reinterpret_cast<const f32x4*>(rp + 8 * g);
    }
    if (EPI == kLinMaskBits) bnx = brow[(long)min(t + 1, otiles - 1) * 2 * N];
    stage(t + D - 1);
    const u32x4* lp = reinterpret_cast<const u32x4*>(smem_c + (t % D) * kTriTile + tri_rm_off(lo, h, 0));
    f32x16 acc = zero16();  // D[row = output 32 t + crow(r, h)][col = point]
    float w_inv = 1.f;      // duo: 2^-e of this weight tile
    if (kLinDuo) w_inv = *reinterpret_cast<const float*>(smem_c + (t % D) * kTriTile + kDuoScaleSlot);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (kLinDuo) {
        const u32x4 ah = lp[192 * ks], al = lp[192 * ks + 32];
        if (EPI == kLinAmax) {  // mfma_duo(a, x)'s three products in its order, each with its operands swapped (see below)
          acc = mfma_hf(xq[2 * ks], al, acc);
          acc = mfma_hf(xq[2 * ks + 1], ah, acc);
          acc = mfma_hf(xq[2 * ks], ah, acc);
        } else {
          acc = mfma_duo(ah, al, xq[2 * ks], xq[2 * ks + 1], acc);
        }
        continue;
      }
      const Tri a = {lp[192 * ks], lp[192 * ks + 32], lp[192 * ks + 64]};
      const Tri bq = {xq[kLinDuo ? 0 : 3 * ks], xq[kLinDuo ? 0 : 3 * ks + 1], xq[kLinDuo ? 0 : 3 * ks + 2]};
      // amax: the operands swapped -- D[row = point crow(r, h)][col = output 32 t + lo] -- so that the maximum over the
      // tile's points runs over a lane's REGISTERS (with the points on the lanes it took a DPP butterfly, two readlanes
      // and a ballot per output row: 400 instructions per tile beside 48 MFMAs)
      if (EPI == kLinAmax) {  // mfma_tri(a, bq)'s six products in its order, each with its operands swapped: the same sums
        acc = mfma_bf(bq.m, a.m, acc);
        acc = mfma_bf(bq.l, a.h, acc);
        acc = mfma_bf(bq.h, a.l, acc);
        acc = mfma_bf(bq.m, a.h, acc);
        acc = mfma_bf(bq.h, a.m, acc);
        acc = mfma_bf(bq.h, a.h, acc);
      } else {
        acc = mfma_tri(a, bq, acc);
      }
    }
    if (kLinDuo) {  // take the scales out: exact (powers of two)
      const float sc = w_inv * x_inv;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] *= (EPI == kLinAmax) ? w_inv * xinv_r[r] : sc;
    }
    if (EPI == kLinAmax) {
      float m = acc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
      int am = 99;  // the first of this half's 16 points that reaches m (crow(r, h) ascends with r); none (NaN): 99
#pragma unroll
      for (int r = 15; r >= 0; --r) am = (acc[r] == m) ? crow(r, h) : am;
      // NaN propagates like torch's conv(x).max(dim=-1) (and the stock fallback): value NaN, argument = the first NaN
      int an = 99;
#pragma unroll
      for (int r = 15; r >= 0; --r) an = (acc[r] != acc[r]) ? crow(r, h) : an;
      if (an != 99) {
        m = __builtin_nanf("");
        am = an;
      }
      const auto pm = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
      const auto pa = __builtin_amdgcn_permlane32_swap((unsigned)am, (unsigned)am, false, false);
      const float mo = __uint_as_float(h ? pm[0] : pm[1]);  // the other half's maximum and argument
      const int ao = (int)(h ? pa[0] : pa[1]);
      const bool n_me = m != m, n_ot = mo != mo;
      res_m = (n_me || n_ot) ? __builtin_nanf("") : fmaxf(m, mo);
      const int a_me = (n_me || (!n_ot && m == res_m)) ? am : 99, a_ot = (n_ot || (!n_me && mo == res_m)) ? ao : 99;
      res_a = min(a_me, a_ot);
      res_a = res_a == 99 ? 0 : res_a;
      if (h == 0) {  // one coalesced line per array
        const long at = ptile * O + (t0 + t) * 32 + lo;
        pmax[at] = res_m;
        parg[at] = min(n_first + res_a, N - 1);
      }
      asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else if (CM) {
      float* oc = orow + (long)(t * 32 + 4 * h) * o_rs;   // acc[4 g + e]: output 32 t + 8 g + 4 h + e
      if (accum) {
        float prev[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) prev[r] = oc[(long)(8 * (r >> 2) + (r & 3)) * o_rs];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += prev[r];
      }
      // without `accum` the clamped lanes store too (point N-1's value again, as the point-major path does): a wave wholly
      // past N in a tail chunk must ISSUE its 16 stores, or vmcnt(54) below is satisfied before tile t+1's DMA has landed
      // (only 2 x 3 DMA pieces would be younger).  With `accum` such a wave has issued the 16 `prev` loads instead.
      if (own || !accum) {
#pragma unroll
        for (int r = 0; r < 16; ++r) oc[(long)(8 * (r >> 2) + (r & 3)) * o_rs] = acc[r];
      }
      asm volatile("s_waitcnt vmcnt(54) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        if (EPI == kLinLeaky || EPI == kLinLeakyBits) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (EPI == kLinLeakyBits) bf |= (o[e] > 0.f ? 1u : 0u) << (4 * g + e);
            o[e] = o[e] > 0.f ? o[e] : kLeakySlope * o[e];
          }
        }
        if (EPI == kLinMask) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rf[g][e] > 0.f ? o[e] : kLeakySlope * o[e];
        }
        if (EPI == kLinMaskBits) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = ((bf >> (4 * g + e)) & 1u) ? o[e] : kLeakySlope * o[e];
        }
        *reinterpret_cast<f32x4*>(orow + t * 32 + 8 * g) = o;
      }
      if (EPI == kLinLeakyBits) {
        if (own) brow[(long)t * 2 * N] = (unsigned short)bf;
        bf = 0;
      }
      if (EPI == kLinMaskBits) bf = bnx;
      if (EPI == kLinMask) {
#pragma unroll
        for (int g = 0; g < 4; ++g) rf[g] = rn[g];
        asm volatile("s_waitcnt vmcnt(26) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
  }
}

// (b, tile, o) partial maxima -> y[b][o] = max over the tiles, arg[b][o] = the first point that reaches it (tiles in
// ascending point order: the first tile with the maximum wins)
__global__ __launch_bounds__(256) void lin_amax_reduce_kernel(const float* __restrict__ pmax, const int* __restrict__ parg,
                                                              int ntiles, int O, float* __restrict__ y,
                                                              int* __restrict__ arg) {
  const int b = blockIdx.y, o = blockIdx.x * 256 + threadIdx.x;
  if (o >= O) return;
  const float* pm = pmax + (long)b * ntiles * O + o;
  const int* pa = parg + (long)b * ntiles * O + o;
  float m = pm[0];
  int a = pa[0];
  for (int t = 1; t < ntiles; ++t) {
    const float v = pm[(long)t * O];
    const int av = pa[(long)t * O];
    // (a NaN partial wins over every number and keeps the first tile that has one: torch's max)
    if (m == m && (v != v || v > m || (v == m && av < a))) {
      m = v;
      a = av;
    }
  }
  y[(long)b * O + o] = m;
  arg[(long)b * O + o] = a;
}

// dx[c][n] = sum_o W[o][c] g[n][o] (proj_dx_tri_kernel with a run-time tile count)
__global__ __launch_bounds__(512, 2) void lin_dx_tri_kernel(const float* __restrict__ g, long g_bs, long g_rs,
                                                            const char* __restrict__ Wtr, int otiles, int Cin, int N,
                                                            float* dx, long dx_bs, const float* res) {  // (`res` may be `dx`: no __restrict__)
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  constexpr int D = kLinDepth;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int n = min(chunk * 256 + wave * 32 + lo, N - 1);
  // (lanes past N-1 repeat point N-1: harmless for a plain store of the same value, NOT for an in-place accumulation --
  // with `res` only the point's own lane stores)
  const bool own = chunk * 256 + wave * 32 + lo < N;
  const float* grow = g + (long)b * g_bs + (long)n * g_rs + 4 * h;
  auto stage = [&](int t) {
    const char* gt = Wtr + (long)min(t, otiles - 1) * kTriTile;
    char* lt = smem_c + (t % D) * kTriTile;
#pragma unroll
    for (int k = 0; k < 3; ++k) lin_glds16(gt + (tid + 512 * k) * 16, lt + (wave * 64 + 512 * k) * 16);
  };
  // this lane's 16 gradient values of tile t in the transposed image's element order (k-step s, element e <-> output
  // 32 t + 16 s + 8 (e >> 2) + 4 h + (e & 3)); loads in assembly so that the compiler does not count them (it would wait
  // for them with vmcnt(0) behind the DMA pieces just issued): the hand-counted wait below names the registers
  auto load_g = [&](int t, f32x4 (&dst)[4]) {
    const float* p = grow + min(t, otiles - 1) * 32;
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
  for (int t = 0; t < otiles; ++t) {
    load_g(t + 2, gnn);  // 4 loads, then the P DMA pieces: the wait below leaves exactly those P in flight
    stage(t + D - 1);
    const char* wt = smem_c + (t % D) * kTriTile;
    auto fetch = [&](int i) {
      const char* ap = wt + tri_tr_off(32 * (i & 3) + lo, 2 * (i >> 2) + h, 0);
      return Tri{*reinterpret_cast<const u32x4*>(ap), *reinterpret_cast<const u32x4*>(ap + 2048),
                 *reinterpret_cast<const u32x4*>(ap + 4096)};
    };
    Tri a0 = fetch(0), a1 = fetch(1), a2 = fetch(2);
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // k-step i >> 2, channel tile i & 3; pair i of tile t+1's values split beside it
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
    asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" : "+v"(gnn[0]), "+v"(gnn[1]), "+v"(gnn[2]), "+v"(gnn[3])::"memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) gn[i] = gnn[i];
    bg[0] = nb[0];
    bg[1] = nb[1];
  }
  float* ob = dx + (long)b * dx_bs + n;  // rows past N-1 hold point N-1's column again: same values, same address
  if (res) {
    // every residual value first, then the stores: `res` may be `dx` itself, so the compiler must keep each load in front
    // of the stores before it -- interleaved, that is 64 dependent round trips at the end of every wave
    const float* rb = res + (long)b * dx_bs + n;
    float rv[4][16];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 32 * ct + crow(r, h);
        rv[ct][r] = c < Cin ? rb[(long)c * N] : 0.f;
      }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 32 * ct + crow(r, h);
        if (c < Cin && own) ob[(long)c * N] = rv[ct][r] + acc[ct][r];
      }
    return;
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    if (32 * ct >= Cin) break;   // (uniform)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = 32 * ct + crow(r, h);
      if (c < Cin) ob[(long)c * N] = acc[ct][r];
    }
  }
}

// lin_dx on two fp16 planes.  Per output tile t: the point's 32 gradient values (16 per lane half) under their own
// power-of-two scale, the W^T tile under the image tile's; the tile's product (two k-steps) starts from a ZERO accumulator
// per channel tile and is added to the fp32 totals x 2^-(e_w + e_g) by the vector ALU one step later, beside the next
// channel tile's MFMAs.  Channel tile outer, k-step inner, so that only two temporaries are alive at a time.
// res (may be null, may be dx itself): a tensor of dx's layout added on the way out -- the residual of the layer this
// product closes (forward: y + W2 h), or the gradient that arrives beside it (backward: in place)
// NW waves per workgroup (32 points each): 8, or 4 / 2 when 8 would leave CUs without a workgroup (the blocks' coarse
// levels: a wave's work does not shrink with N, the waves per CU do)
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void lin_dx_duo_kernel(const float* __restrict__ g, long g_bs, long g_rs,
                                                               const char* __restrict__ Wtr, int otiles, int Cin, int N,
                                                               float* dx, long dx_bs, const float* res) {  // (`res` may be `dx`: no __restrict__)
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  constexpr int D = kLinDepth, NT = 64 * NW, P = kTriTile / 16 / NT;   // P: 16-byte DMA pieces per thread and tile
  static_assert(P == 3 || P == 6 || P == 12, "8, 4 or 2 waves");
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int n = min(chunk * (32 * NW) + wave * 32 + lo, N - 1);
  // (lanes past N-1 repeat point N-1: harmless for a plain store of the same value, NOT for an in-place accumulation --
  // with `res` only the point's own lane stores)
  const bool own = chunk * (32 * NW) + wave * 32 + lo < N;
  const float* grow = g + (long)b * g_bs + (long)n * g_rs + 4 * h;
  auto stage = [&](int t) {
    const char* gt = Wtr + (long)min(t, otiles - 1) * kTriTile;
    char* lt = smem_c + (t % D) * kTriTile;
#pragma unroll
    for (int k = 0; k < P; ++k) lin_glds16(gt + (tid + NT * k) * 16, lt + (wave * 64 + NT * k) * 16);
  };
  auto load_g = [&](int t, f32x4 (&dst)[4]) {  // (assembly loads: lin_dx_tri_kernel explains why)
    const float* p = grow + min(t, otiles - 1) * 32;
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:32\n\t"
                 "global_load_dwordx4 %2, %4, off offset:64\n\tglobal_load_dwordx4 %3, %4, off offset:96"
                 : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3])
                 : "v"(p)
                 : "memory");
  };
  // scale of a lane pair's 32 values (this lane's 16, the other half's 16): s = 2^e with amax s in [2^12, 2^13)
  auto scale_of = [&](const f32x4 (&v)[4], float& sc, float& inv) {
    float amax = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(v[q][e]));
    amax = xor32_max(amax);
    duo_scale_for(amax, sc, inv);
  };
#pragma unroll
  for (int t = 0; t < D - 1; ++t) stage(t);
  f32x16 tot[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) tot[ct] = zero16();
  f32x4 gn[4], gnn[4];
  u32x4 bh[2], bl[2], nh[2], nl[2];  // planes of tile t / tile t + 1, k-steps 0 and 1
  load_g(0, gnn);
  load_g(1, gn);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier"
               : "+v"(gnn[0]), "+v"(gnn[1]), "+v"(gnn[2]), "+v"(gnn[3]), "+v"(gn[0]), "+v"(gn[1]), "+v"(gn[2]), "+v"(gn[3])::"memory");
  float g_inv;
  {
    float s0;
    scale_of(gnn, s0, g_inv);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float v[8] = {gnn[2 * ks][0], gnn[2 * ks][1], gnn[2 * ks][2], gnn[2 * ks][3],
                          gnn[2 * ks + 1][0], gnn[2 * ks + 1][1], gnn[2 * ks + 1][2], gnn[2 * ks + 1][3]};
      duo_split8(v, s0, bh[ks], bl[ks]);
    }
  }
  for (int t = 0; t < otiles; ++t) {
    load_g(t + 2, gnn);  // 4 loads, then the P DMA pieces: the wait below leaves exactly those P in flight
    stage(t + D - 1);
    const char* wt = smem_c + (t % D) * kTriTile;
    const float sc = *reinterpret_cast<const float*>(wt + kDuoTrScaleSlot) * g_inv;  // 2^-(e_w + e_g), this lane's point
    float s_next, g_inv_next;
    scale_of(gn, s_next, g_inv_next);  // tile t + 1's values (past the last tile: a repeat, never used)
    auto fetch_h = [&](int i) { return *reinterpret_cast<const u32x4*>(wt + tri_tr_off(32 * (i >> 1) + lo, 2 * (i & 1) + h, 0)); };
    auto fetch_l = [&](int i) { return *reinterpret_cast<const u32x4*>(wt + tri_tr_off(32 * (i >> 1) + lo, 2 * (i & 1) + h, 1)); };
    u32x4 a0h = fetch_h(0), a0l = fetch_l(0), a1h = fetch_h(1), a1l = fetch_l(1), a2h = fetch_h(2), a2l = fetch_l(2);
    f32x16 tmp[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // channel tile i >> 1, k-step i & 1
      const int ct = i >> 1, ks = i & 1;
      u32x4 a3h = a2h, a3l = a2l;
      if (i + 3 < 8) {
        a3h = fetch_h(i + 3);
        a3l = fetch_l(i + 3);
      }
      __builtin_amdgcn_sched_barrier(0);
      tmp[ct] = mfma_duo(a0h, a0l, bh[ks], bl[ks], ks ? tmp[ct] : zero16());
      {  // pair i of tile t + 1's values -> its plane words
        const int k2 = i >> 2, w = i & 3;
        unsigned hw, lw;
        duo_split2(gn[2 * k2 + (w >> 1)][2 * (w & 1)] * s_next, gn[2 * k2 + (w >> 1)][2 * (w & 1) + 1] * s_next, hw, lw);
        nh[k2][w] = hw;
        nl[k2][w] = lw;
      }
      if (ks == 0 && ct >= 1) {  // the channel tile finished one step ago joins the totals
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[ct - 1][r] = fmaf(tmp[ct - 1][r], sc, tot[ct - 1][r]);
      }
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      a0h = a1h; a0l = a1l;
      a1h = a2h; a1l = a2l;
      a2h = a3h; a2l = a3l;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) tot[3][r] = fmaf(tmp[3][r], sc, tot[3][r]);
    if (P == 3)
      asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" : "+v"(gnn[0]), "+v"(gnn[1]), "+v"(gnn[2]), "+v"(gnn[3])::"memory");
    else if (P == 6)
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" : "+v"(gnn[0]), "+v"(gnn[1]), "+v"(gnn[2]), "+v"(gnn[3])::"memory");
    else
      asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" : "+v"(gnn[0]), "+v"(gnn[1]), "+v"(gnn[2]), "+v"(gnn[3])::"memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) gn[i] = gnn[i];
    bh[0] = nh[0]; bh[1] = nh[1];
    bl[0] = nl[0]; bl[1] = nl[1];
    g_inv = g_inv_next;
  }
  float* ob = dx + (long)b * dx_bs + n;  // rows past N-1 hold point N-1's column again: same values, same address
  if (res) {
    // every residual value first, then the stores: `res` may be `dx` itself, so the compiler must keep each load in front
    // of the stores before it -- interleaved, that is 64 dependent round trips at the end of every wave
    const float* rb = res + (long)b * dx_bs + n;
    float rv[4][16];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 32 * ct + crow(r, h);
        rv[ct][r] = c < Cin ? rb[(long)c * N] : 0.f;
      }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 32 * ct + crow(r, h);
        if (c < Cin && own) ob[(long)c * N] = rv[ct][r] + tot[ct][r];
      }
    return;
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    if (32 * ct >= Cin) break;   // (uniform)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = 32 * ct + crow(r, h);
      if (c < Cin) ob[(long)c * N] = tot[ct][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// lin_chain: the two 1x1 convolutions of a feed-forward layer in ONE sweep over the points --
//     mid[n][o]  = epi(sum_c Wa[o][c] x[c][n])          (lin_fwd; written out for the weight gradients that need it)
//     out[c][n]  = sum_o Wb[o][c] mid[n][o]  (+ res)     (lin_dx)
// forward (models/attention.py:187-192 `ff`): Wa = W1, epi = LeakyReLU + sign words, Wb = W2^T;  backward: Wa = W2^T,
// epi = the sign words' mask, Wb = W1 (the input gradient).  The separate kernels are memory-bound (~4 TB/s), and the
// second one's only input is what the first just wrote: 134 MB at B = 32, N = 2048, H = 512 that this kernel does not
// read back.  It can do so because the transposed weight image already IS in the accumulator's order (tri_tr_off's chunk
// (k-step s, half h) holds the outputs crow(8 s + i, h): lin_dx's B operand was laid out that way on purpose), so product 1's
// accumulator -- lane (point, h): outputs crow(r, h), r = 0..15 -- is product 2's B operand as it stands: scale (lane
// pair's largest value), split into two fp16 planes, multiply.  Per hidden tile: 24 + 24 matrix instructions; W tiles of
// both matrices through one ring of three slot pairs (144 KB); the point's output accumulates in fp32 as in lin_dx_duo.
// ------------------------------------------------------------------------------------------------
constexpr int kChainDepth = 3;
constexpr int kChainLds = kChainDepth * 2 * kTriTile;
static_assert(kChainLds <= 160 * 1024, "LDS budget");

// NW waves per workgroup: 8, or 4 / 2 for short clouds (the two waves of a SIMD run their phases one after the other: a
// wave alone on its SIMD is twice as fast, and at N <= 1024 there are idle CUs to spread them over)
template <int EPI, int NW>  // kLinLeakyBits | kLinMaskBits
__global__ __launch_bounds__(64 * NW, 2) void lin_chain_kernel(const float* __restrict__ x, long x_bs, int N,
                                                           const char* __restrict__ Wa_rm, const char* __restrict__ Wb_tr,
                                                           int otiles, float* __restrict__ mid, long m_bs, long m_rs,
                                                           unsigned short* __restrict__ bits, float* out,   // (`res` may be `out`)
                                                           long o_bs, const float* res) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  constexpr int D = kChainDepth, NT = 64 * NW, P = kTriTile / 16 / NT;   // P: 16-byte DMA pieces per thread, tile and matrix
  static_assert(P == 3 || P == 6 || P == 12, "8, 4 or 2 waves");
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int n = min(chunk * (32 * NW) + wave * 32 + lo, N - 1);
  const bool own = chunk * (32 * NW) + wave * 32 + lo < N;
  auto stage = [&](int t) {
    const long tt = min(t, otiles - 1);
    char* slot = smem_c + (t % D) * 2 * kTriTile;
#pragma unroll
    for (int k = 0; k < P; ++k) lin_glds16(Wa_rm + tt * kTriTile + (tid + NT * k) * 16, slot + (wave * 64 + NT * k) * 16);
#pragma unroll
    for (int k = 0; k < P; ++k)
      lin_glds16(Wb_tr + tt * kTriTile + (tid + NT * k) * 16, slot + kTriTile + (wave * 64 + NT * k) * 16);
  };
  stage(0);
  stage(1);
  // the point's 128 channels: two fp16 planes under the point's own scale (lin_fwd's B operand)
  u32x4 xq[16];
  float x_inv;
  {
    float xv[64];
    float amax = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xv[8 * ks + e] = x[(long)b * x_bs + (long)(16 * ks + 8 * h + e) * N + n];
        amax = fmaxf(amax, fabsf(xv[8 * ks + e]));
      }
    amax = xor32_max(amax);
    float sx;
    duo_scale_for(amax, sx, x_inv);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const float v[8] = {xv[8 * ks], xv[8 * ks + 1], xv[8 * ks + 2], xv[8 * ks + 3],
                          xv[8 * ks + 4], xv[8 * ks + 5], xv[8 * ks + 6], xv[8 * ks + 7]};
      duo_split8(v, sx, xq[2 * ks], xq[2 * ks + 1]);
    }
  }
  float* mrow = mid ? mid + (long)b * m_bs + (long)n * m_rs + 4 * h : nullptr;
  unsigned short* brow = bits + (((long)b * otiles) * 2 + h) * N + n;   // sign words [(b, tile, h)][point]
  unsigned bf = 0, bnx = 0;
  if (EPI == kLinMaskBits) bf = brow[0];
  f32x16 tot[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) tot[ct] = zero16();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // iteration t: [mask: tile t+1's sign word], tiles t+2 into the slot pair of tile t-1, product 1, its epilogue and
  // stores, product 2.  VM operations younger than tile t+1's DMA at the end of iteration t: stores(t-1) 4 + DMA 6 +
  // stores(t) 4 = 14 (sign-word traffic only adds to that: counting low is the safe side)
  for (int t = 0; t < otiles; ++t) {
    if (EPI == kLinMaskBits) bnx = brow[(long)min(t + 1, otiles - 1) * 2 * N];
    stage(t + 2);
    const char* slot = smem_c + (t % D) * 2 * kTriTile;
    const u32x4* lp = reinterpret_cast<const u32x4*>(slot + tri_rm_off(lo, h, 0));
    f32x16 acc = zero16();  // D[row = hidden unit 32 t + crow(r, h)][col = point]
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) acc = mfma_duo(lp[192 * ks], lp[192 * ks + 32], xq[2 * ks], xq[2 * ks + 1], acc);
    const float sc1 = *reinterpret_cast<const float*>(slot + kDuoScaleSlot) * x_inv;
    unsigned word = 0;
    float amax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r] * sc1;
      if (EPI == kLinLeakyBits) {
        word |= (v > 0.f ? 1u : 0u) << r;
        v = v > 0.f ? v : kLeakySlope * v;
      } else {
        v = ((bf >> r) & 1u) ? v : kLeakySlope * v;
      }
      acc[r] = v;
      amax = fmaxf(amax, fabsf(v));
    }
    if (mrow) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(mrow + t * 32 + 8 * g) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    }
    if (EPI == kLinLeakyBits && own) brow[(long)t * 2 * N] = (unsigned short)word;
    if (EPI == kLinMaskBits) bf = bnx;
    // product 2: the accumulator as the B operand (k-step s: registers 8 s .. 8 s + 7), under the lane pair's scale
    amax = xor32_max(amax);
    float s2, g_inv;
    duo_scale_for(amax, s2, g_inv);
    u32x4 bh[2], bl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float v[8] = {acc[8 * ks], acc[8 * ks + 1], acc[8 * ks + 2], acc[8 * ks + 3],
                          acc[8 * ks + 4], acc[8 * ks + 5], acc[8 * ks + 6], acc[8 * ks + 7]};
      duo_split8(v, s2, bh[ks], bl[ks]);
    }
    const char* wt = slot + kTriTile;
    const float sc2 = *reinterpret_cast<const float*>(wt + kDuoTrScaleSlot) * g_inv;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      f32x16 tmp = zero16();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const u32x4 ah = *reinterpret_cast<const u32x4*>(wt + tri_tr_off(32 * ct + lo, 2 * ks + h, 0));
        const u32x4 al = *reinterpret_cast<const u32x4*>(wt + tri_tr_off(32 * ct + lo, 2 * ks + h, 1));
        tmp = mfma_duo(ah, al, bh[ks], bl[ks], tmp);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[ct][r] = fmaf(tmp[r], sc2, tot[ct][r]);
    }
    // with `mid`: 4 + 2 P + 4 younger operations (14 / 20 / 32).  Without it (mid == NULL: no stores are issued) only this
    // iteration's 2 P DMA pieces are younger than tile t+1's -- 6 / 12 / 24 -- and the larger count would let the barrier
    // release the waves onto a slot whose DMA has not landed.  `mrow` is a kernel argument: a wave-uniform branch.
    if (mrow) {
      if (P == 3)
        asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else if (P == 6)
        asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      if (P == 3)
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else if (P == 6)
        asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
  float* ob = out + (long)b * o_bs + n;
  if (res) {  // (may be `out` itself; a channel tile's 16 loads, then its 16 stores: 16 registers, not 64)
    const float* rb = res + (long)b * o_bs + n;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = rb[(long)(32 * ct + crow(r, h)) * N];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (own) ob[(long)(32 * ct + crow(r, h)) * N] = rv[r] + tot[ct][r];
    }
    return;
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (own) ob[(long)(32 * ct + crow(r, h)) * N] = tot[ct][r];
}

// dW partials: workgroup = (512-point chunk, cloud, block of 128 OT outputs); wave w owns output rows 32 OT (w >> 1) .. of
// the block and channels 64 (w & 1) .. +63: 2 OT accumulator tiles.  Both operands are transposed through LDS (the
// contraction runs over the points) and split in registers, as in proj_dw_tri_kernel.
// OT = output tiles per wave: 2 (blocks of 256 outputs) or 1 (blocks of 128: narrow layers)
constexpr int kLdwPts = 512, kLdwXS = 33;
// points per workgroup: 512, halved (down to 128) while the launch would leave CUs without a workgroup -- the coarse levels
// of the blocks (N = 1024, 512) ran 128 / 64 workgroups of the same length as N = 2048's 256.  A function of the shape only:
// the partial sums' grouping, hence the result's bits, are fixed per shape.
static int ldw_points(int B, int N, int O, int ob) {
  int pts = kLdwPts;
  while (pts > 128 && (long)B * ((N + pts - 1) / pts) * (O / ob) < 256) pts >>= 1;
  return pts;
}
// GCM: g arrives channel-major, (B, O, N) with g_rs = the stride between output channels (the weight gradient of a
// channel-major -> channel-major convolution): its tile is staged like x's, [output][point] with row stride 33
template <int OT, bool GCM = false>
struct Ldw {
  static constexpr int kOB = 128 * OT;
  static constexpr int kGS = kOB + 4;  // row stride: g tile rows 16-byte aligned, column reads conflict-free
  static constexpr int kG = GCM ? kOB * kLdwXS : kTile * kGS;
  static constexpr int kBuf = kG + 128 * kLdwXS;
  static constexpr int kLds = 2 * kBuf * 4;
};

template <int OT, bool GCM = false>
__global__ __launch_bounds__(512, 2) void lin_dw_tri_kernel(const float* __restrict__ g, long g_bs, long g_rs,
                                                            const float* __restrict__ x, long x_bs, int Cin, int N, int O,
                                                            float* __restrict__ part, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  float* smem = reinterpret_cast<float*>(smem_c);
  constexpr int GS = Ldw<OT, GCM>::kGS, XS = kLdwXS, BUF = Ldw<OT, GCM>::kBuf, kLdwOB = Ldw<OT, GCM>::kOB, GOFF = Ldw<OT, GCM>::kG;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  const int og = wave >> 1, ch = wave & 1;
  const int b = blockIdx.y, o0 = blockIdx.z * kLdwOB;
  const int n0 = blockIdx.x * ntiles * kTile;
  f32x16 acc[OT][2];
#pragma unroll
  for (int a = 0; a < OT; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[a][c] = zero16();
  f32x4 gst[2 * OT];  // 32 rows x (32 OT) float4 / 512 threads
  float gsc[GCM ? 8 * OT : 1];  // GCM: (128 OT) outputs x 32 points / 512 threads
  float xst[8];  // 128 channels x 32 points = 4096 floats / 512 threads
  auto issue = [&](int nn0) {
    if (GCM) {
#pragma unroll
      for (int it = 0; it < 8 * OT; ++it) {
        const int e = tid + 512 * it;
        gsc[GCM ? it : 0] = (nn0 + (e & 31) < N) ? g[(long)b * g_bs + (long)(o0 + (e >> 5)) * g_rs + nn0 + (e & 31)] : 0.f;
      }
    } else
#pragma unroll
    for (int it = 0; it < 2 * OT; ++it) {
      const int e = tid + 512 * it;
      const int r = e / (32 * OT), c4 = (e % (32 * OT)) * 4;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      gst[it] = (nn0 + r < N) ? *reinterpret_cast<const f32x4*>(g + (long)b * g_bs + (long)(nn0 + r) * g_rs + o0 + c4) : z4;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int e = tid + 512 * it;
      const int c = e >> 5, pnt = e & 31;
      xst[it] = (nn0 + pnt < N && c < Cin) ? x[(long)b * x_bs + (long)c * N + nn0 + pnt] : 0.f;
    }
  };
  auto commit = [&](float* buf) {
    if (GCM) {
#pragma unroll
      for (int it = 0; it < 8 * OT; ++it) {
        const int e = tid + 512 * it;
        buf[(e >> 5) * XS + (e & 31)] = gsc[GCM ? it : 0];
      }
    } else
#pragma unroll
    for (int it = 0; it < 2 * OT; ++it) {
      const int e = tid + 512 * it;
      const int r = e / (32 * OT), c4 = (e % (32 * OT)) * 4;
      *reinterpret_cast<f32x4*>(buf + r * GS + c4) = gst[it];
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int e = tid + 512 * it;
      buf[GOFF + (e >> 5) * XS + (e & 31)] = xst[it];
    }
  };
  // g's tile element (point p of the tile, output j of the block)
  auto gat = [&](const float* gt, int p, int j) -> float { return GCM ? gt[j * XS + p] : gt[p * GS + j]; };
  issue(n0);
  commit(smem);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    float* cur = smem + (t & 1) * BUF;
    float* nxt = smem + ((t & 1) ^ 1) * BUF;
    if (t + 1 < ntiles) issue(n0 + (t + 1) * kTile);
    const float* gt = cur;
    const float* xt = cur + GOFF;
    if (kLinDuo) {
      // both k-steps' operands of the tile under two wave-uniform scales (the block of g this wave multiplies, its block
      // of x), every (output tile, channel tile) product from a zero accumulator, fp32 totals by the vector ALU
      u32x4 ah[2][OT], al[2][OT], bh[2][2], bl[2][2];
      float inv_a, inv_b;
      {
        float v[2][OT][8];
        float amax = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int ot = 0; ot < OT; ++ot)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              v[ks][ot][e] = gat(gt, 16 * ks + 8 * h + e, 32 * OT * og + 32 * ot + lo);
              amax = fmaxf(amax, fabsf(v[ks][ot][e]));
            }
        float sa;
        duo_scale_for(wave_max_all(amax), sa, inv_a);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int ot = 0; ot < OT; ++ot) duo_split8(v[ks][ot], sa, ah[ks][ot], al[ks][ot]);
      }
      {
        float v[2][2][8];
        float amax = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              v[ks][ct][e] = xt[(64 * ch + 32 * ct + lo) * XS + 16 * ks + 8 * h + e];
              amax = fmaxf(amax, fabsf(v[ks][ct][e]));
            }
        float sb;
        duo_scale_for(wave_max_all(amax), sb, inv_b);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) duo_split8(v[ks][ct], sb, bh[ks][ct], bl[ks][ct]);
      }
      const float sc = inv_a * inv_b;
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          f32x16 tmp = mfma_duo(ah[0][ot], al[0][ot], bh[0][ct], bl[0][ct], zero16());
          tmp = mfma_duo(ah[1][ot], al[1][ot], bh[1][ct], bl[1][ct], tmp);
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ot][ct][r] = fmaf(tmp[r], sc, acc[ot][ct][r]);
        }
    } else
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // points 16 ks .. 16 ks + 15; lane half h: the 8 points 16 ks + 8 h + e
      Tri bq[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = xt[(64 * ch + 32 * ct + lo) * XS + 16 * ks + 8 * h + e];
        bq[ct] = tri_split8(v);
      }
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gat(gt, 16 * ks + 8 * h + e, 32 * OT * og + 32 * ot + lo);
        const Tri a = tri_split8(v);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[ot][ct] = mfma_tri(a, bq[ct], acc[ot][ct]);
      }
    }
    if (t + 1 < ntiles) commit(nxt);
    __syncthreads();
  }
  float* outp = part + ((long)b * gridDim.x + blockIdx.x) * O * 128;
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o0 + 32 * OT * og + 32 * ot + crow(r, h);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) outp[(long)o * 128 + 64 * ch + 32 * ct + lo] = acc[ot][ct][r];
    }
}

// out[e] = sum over the nparts partial blocks (each n4 float4) in a fixed order: 16 groups of threads take the parts
// p = g, g + 16, ... (16 loads in flight each), then the 16 group sums are added in index order
// (tr_O > 0: the (tr_O, 128) sum leaves transposed, out[c][o] -- the second FFN convolution's weight gradient in the
// layout its parameter has)
__global__ __launch_bounds__(256) void lin_sum_parts_kernel(const float* __restrict__ part, int nparts, long n4,
                                                            float* __restrict__ out, int tr_O) {
  __shared__ f32x4 red[16][17];
  const int e4l = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long e4 = (long)blockIdx.x * 16 + e4l;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (e4 < n4) {
    const f32x4* p4 = reinterpret_cast<const f32x4*>(part) + e4;
    for (int p0 = g; p0 < nparts; p0 += 16 * 16) {
      f32x4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int p = p0 + 16 * u;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        v[u] = (p < nparts) ? p4[(long)p * n4] : z4;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
  }
  red[g][e4l] = s;
  __syncthreads();
  if (g == 0 && e4 < n4) {
    f32x4 tot = red[0][e4l];
#pragma unroll
    for (int k = 1; k < 16; ++k) tot += red[k][e4l];
    if (tr_O) {
      const long o = e4 >> 5, c = 4 * (e4 & 31);
#pragma unroll
      for (int k = 0; k < 4; ++k) out[(c + k) * tr_O + o] = tot[k];
    } else {
      reinterpret_cast<f32x4*>(out)[e4] = tot;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward of y[b][o] = max_n (W x)[o][n] (models/cls_model.py:136 and its autograd): only the arg-max column of each
// (cloud, output) carries gradient --
//   dx[b][:, n] = sum over the outputs o with arg[b][o] = n of gy[b][o] W[o][:]     (every other column: 0)
//   dW[o][:]    = sum over the clouds b of gy[b][o] x[b][:, arg[b][o]]
// amax_sort: one workgroup per cloud groups the O outputs by their arg-max point (counting sort in LDS; order inside a
// group: ascending output); amax_dx: a wave per group sums its W rows in that order and writes the point's 128 gradient
// values; amax_dw: a wave per output sums the clouds' arg-max columns in ascending cloud order.  No atomics: run-to-run
// identical.
// (Exact ties of the maximum go to the lowest point index; torch.amax's backward splits them evenly.)
// ------------------------------------------------------------------------------------------------
// workspace of the backward, per cloud: [start offsets of the points' groups: N + 1][outputs grouped by point: O]
// [distinct points, ascending: O][number of groups: 1] (ints)
__global__ __launch_bounds__(1024) void amax_sort_kernel(int N, const int* __restrict__ arg, int O, int* __restrict__ wsi) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  int* cnt = reinterpret_cast<int*>(smem_c);        // N + 1 words: points' group sizes, then their start offsets
  int* abl = cnt + ((N + 4) & ~3);                  // O words (16-byte aligned): this cloud's arg-max points
  __shared__ int wtot[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int* ab = arg + (long)b * O;
  int* start_g = wsi + (long)b * (N + 2 + 2 * O);
  int* ord_g = start_g + N + 1;
  int* pts_g = ord_g + O;
  for (int n = tid; n <= N; n += 1024) cnt[n] = 0;
  for (int o = tid; o < O; o += 1024) abl[o] = min(max(ab[o], 0), N - 1);
  __syncthreads();
  for (int o = tid; o < O; o += 1024) atomicAdd(&cnt[abl[o]], 1);
  __syncthreads();
  // exclusive scan of cnt[0..N) in place; each thread owns a contiguous run of per = ceil(N / 1024) points
  const int per = (N + 1023) / 1024, lo_n = min(tid * per, N), hi_n = min(lo_n + per, N);
  int run = 0, distinct = 0;
  for (int n = lo_n; n < hi_n; ++n) {
    run += cnt[n];
    distinct += cnt[n] > 0;
  }
  int incl = run, dincl = distinct;
#pragma unroll
  for (int ofs = 1; ofs < 64; ofs <<= 1) {
    const int u = __shfl_up(incl, ofs, 64), du = __shfl_up(dincl, ofs, 64);
    if (lane >= ofs) {
      incl += u;
      dincl += du;
    }
  }
  if (lane == 63) wtot[wave] = incl | (dincl << 16);   // (O <= 8192, N <= 32767)
  __syncthreads();
  int base = 0, dbase = 0;
  for (int w = 0; w < wave; ++w) {
    base += wtot[w] & 0xFFFF;
    dbase += wtot[w] >> 16;
  }
  int start = base + incl - run, dpos = dbase + dincl - distinct;
  for (int n = lo_n; n < hi_n; ++n) {
    const int c = cnt[n];
    cnt[n] = start;
    start_g[n] = start;
    if (c > 0) pts_g[dpos++] = n;
    start += c;
  }
  if (tid == 1023) {
    start_g[N] = O;
    pts_g[O] = dbase + dincl;   // number of groups
  }
  __syncthreads();
  // placement, ascending output inside a group.  Round 6: every output takes the next free slot of its group with an LDS
  // atomic (any order); its place in the list is then the number of its group's members with a smaller output index --
  // counted over the GROUP (a few members; the whole list only when every output of a degenerate cloud points at one
  // point), where rounds 4-5 counted over all earlier outputs: O^2 / 16 LDS reads per cloud, 27 us of a 78 us backward.
  // The lists are the same run to run: the count does not depend on the slots the atomics handed out.
  int mine[8];                                      // O <= 8192: at most 8 outputs per thread; their arg-max points
#pragma unroll
  for (int u = 0; u < 8; ++u) mine[u] = (tid + 1024 * u < O) ? abl[tid + 1024 * u] : -1;
  __syncthreads();                                  // abl is free now: it becomes the (unordered) grouped list
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (mine[u] >= 0) abl[atomicAdd(&cnt[mine[u]], 1)] = tid + 1024 * u;   // cnt[n]: group n's start, bumped to its end
  __syncthreads();                                  // (also makes the owners' start_g stores visible to the workgroup)
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (mine[u] < 0) continue;
    const int o = tid + 1024 * u, s0 = start_g[mine[u]], s1 = cnt[mine[u]];
    int before = 0;
    for (int i = s0; i < s1; ++i) before += abl[i] < o ? 1 : 0;
    ord_g[s0 + before] = o;
  }
}

// dx: one wave per group (= one arg-max point of one cloud), lane = two channels; the group's members in ascending
// output order, four W rows in flight
__global__ __launch_bounds__(256) void amax_dx_kernel(int N, const float* __restrict__ gy, const float* __restrict__ W,
                                                      int O, const int* __restrict__ wsi, float* __restrict__ dx,
                                                      long dx_bs) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int* start_g = wsi + (long)b * (N + 2 + 2 * O);
  const int* ord_g = start_g + N + 1;
  const int* pts_g = ord_g + O;
  const int ngroups = pts_g[O];
  const float* gb = gy + (long)b * O;
  for (int gi = blockIdx.x * 4 + (threadIdx.x >> 6); gi < ngroups; gi += gridDim.x * 4) {
    const int n = pts_g[gi];
    const int s0 = start_g[n], s1 = start_g[n + 1];
    float a0 = 0.f, a1 = 0.f;
    int k = s0;
    for (; k + 4 <= s1; k += 4) {
      int o[4];
      float gv[4], w0[4], w1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) o[u] = ord_g[k + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        gv[u] = gb[o[u]];
        w0[u] = W[(long)o[u] * 128 + lane];
        w1[u] = W[(long)o[u] * 128 + lane + 64];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = fmaf(gv[u], w0[u], a0);
        a1 = fmaf(gv[u], w1[u], a1);
      }
    }
    for (; k < s1; ++k) {
      const int o = ord_g[k];
      const float gv = gb[o];
      a0 = fmaf(gv, W[(long)o * 128 + lane], a0);
      a1 = fmaf(gv, W[(long)o * 128 + lane + 64], a1);
    }
    // += : the column of a point is this wave's alone, so the caller may hand over a buffer that already holds ANOTHER
    // gradient of the same tensor (the sampler's dx behind a pooled head: no zero fill, no add launch); on zeros it is a store
    float* d0 = dx + (long)b * dx_bs + (long)lane * N + n;
    float* d1 = dx + (long)b * dx_bs + (long)(lane + 64) * N + n;
    const float p0 = *d0, p1 = *d1;
    *d0 = p0 + a0;
    *d1 = p1 + a1;
  }
}

// dW[o][c] = sum over the clouds (ascending) of gy[b][o] x[b][c][arg[b][o]]: one wave per output, lane = two channels,
// eight clouds' columns in flight
__global__ __launch_bounds__(256) void amax_dw_kernel(const float* __restrict__ x, long x_bs, int B, int N,
                                                      const int* __restrict__ arg, const float* __restrict__ gy, int O,
                                                      float* __restrict__ dW) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= O) return;
  float a0 = 0.f, a1 = 0.f;
  int b = 0;
  for (; b + 8 <= B; b += 8) {
    float gv[8], x0[8], x1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = min(max(arg[(long)(b + u) * O + o], 0), N - 1);
      gv[u] = gy[(long)(b + u) * O + o];
      x0[u] = x[(long)(b + u) * x_bs + (long)lane * N + n];
      x1[u] = x[(long)(b + u) * x_bs + (long)(lane + 64) * N + n];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = fmaf(gv[u], x0[u], a0);
      a1 = fmaf(gv[u], x1[u], a1);
    }
  }
  for (; b < B; ++b) {
    const int n = min(max(arg[(long)b * O + o], 0), N - 1);
    const float gv = gy[(long)b * O + o];
    a0 = fmaf(gv, x[(long)b * x_bs + (long)lane * N + n], a0);
    a1 = fmaf(gv, x[(long)b * x_bs + (long)(lane + 64) * N + n], a1);
  }
  dW[(long)o * 128 + lane] = a0;
  dW[(long)o * 128 + lane + 64] = a1;
}

// duo images of one 32-row tile of W (O x 128 row-major) per workgroup
// (trans_O > 0: W arrives TRANSPOSED, (128, trans_O) row-major -- the second FFN convolution's weight as the module holds
// it -- and the images are those of its transpose (trans_O, 128): no transposing copy in front of this launch)
// (W2 != null: a second matrix in the same launch -- workgroups tiles1 .. take its tiles: the two weights of a feed-forward
// layer, 2 x 16 workgroups of latency instead of two launches of it)
__global__ __launch_bounds__(256) void lin_images_duo_kernel(const float* __restrict__ W, char* __restrict__ rm,
                                                             char* __restrict__ tr, int trans_O, int tiles1,
                                                             const float* __restrict__ W2, char* __restrict__ rm2,
                                                             char* __restrict__ tr2, int trans_O2) {
  __shared__ float wt[32][129];
  __shared__ float red[4];
  int tile = blockIdx.x;
  const int tid = threadIdx.x;
  if (tile >= tiles1) {  // (uniform)
    tile -= tiles1;
    W = W2;
    rm = rm2;
    tr = tr2;
    trans_O = trans_O2;
  }
  float mx = 0.f;
  for (int e = tid; e < 32 * 128; e += 256) {
    const float v = trans_O ? W[(long)(e & 127) * trans_O + tile * 32 + (e >> 7)] : W[((long)tile * 32 + (e >> 7)) * 128 + (e & 127)];
    wt[e >> 7][e & 127] = v;
    mx = fmaxf(mx, fabsf(v));
  }
  mx = wave_max_all(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  float sc, inv;
  duo_scale_for(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), sc, inv);
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const u32x4 tag4 = {__float_as_uint(inv), 0u, 0u, 0u};
  if (rm) {
    char* img = rm + (long)tile * kTriTile;
    for (int e = tid; e < 512; e += 256) {
      const int r = e & 31, g = e >> 5;
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = wt[r][8 * g + i];
      u32x4 hp, lp;
      duo_split8(v, sc, hp, lp);
      *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 0)) = hp;
      *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 1)) = lp;
      *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 2)) = (r == 0 && g == 0) ? tag4 : zero4;
    }
  }
  if (tr) {
    char* img = tr + (long)tile * kTriTile;
    for (int e = tid; e < 512; e += 256) {
      const int d = e & 127, cg = e >> 7, s2 = cg >> 1, hh = cg & 1;
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = wt[16 * s2 + 8 * (i >> 2) + 4 * hh + (i & 3)][d];
      u32x4 hp, lp;
      duo_split8(v, sc, hp, lp);
      *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 0)) = hp;
      *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 1)) = lp;
      *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 2)) = (d == 0 && cg == 0) ? tag4 : zero4;
    }
  }
}

}  // namespace samble

using namespace samble;

extern "C" int samble_launch_tri_split(const float* src, long bs, long rs, int B, int rows, void* rm, void* tr,
                                       hipStream_t stream);

extern "C" size_t samble_linear_image_bytes_impl(int O) { return (size_t)((O + 31) / 32) * kTriTile; }

// row image and / or transposed image of W (O x 128, row-major): three bf16 planes, or (duo) two fp16 planes of the
// tile x 2^e in the slots of the first two pieces, 2^-e in the tile's spare slot (kDuoScaleSlot / kDuoTrScaleSlot)
extern "C" int samble_launch_linear_images(const float* W, int O, void* rm, void* tr, int transposed, hipStream_t s) {
  if (!kLinDuo) return transposed ? (int)hipErrorNotSupported : samble_launch_tri_split(W, 0, 128, 1, O, rm, tr, s);
  hipLaunchKernelGGL(lin_images_duo_kernel, dim3(O / 32), dim3(256), 0, s, W, (char*)rm, (char*)tr, transposed ? O : 0, O / 32,
                     nullptr, nullptr, nullptr, 0);
  return (int)hipGetLastError();
}
// two matrices in one launch (two-plane build only): W1 (O1, 128) as it is, W2t (128, O2) transposed
extern "C" int samble_launch_linear_images_pair(const float* W1, int O1, void* rm1, void* tr1, const float* W2t, int O2,
                                                void* rm2, void* tr2, hipStream_t s) {
  if (!kLinDuo) return (int)hipErrorNotSupported;
  hipLaunchKernelGGL(lin_images_duo_kernel, dim3(O1 / 32 + O2 / 32), dim3(256), 0, s, W1, (char*)rm1, (char*)tr1, 0, O1 / 32, W2t,
                     (char*)rm2, (char*)tr2, O2);
  return (int)hipGetLastError();
}
extern "C" int samble_linear_is_duo(void) { return kLinDuo ? 1 : 0; }

// output-tile slices of a forward launch: doubled while the launch leaves CUs without a workgroup
static int lin_fwd_slices(int wgs, int otiles) {
  int z = 1;
  while (wgs * z < kLinFillWgs && otiles % (2 * z) == 0) z *= 2;
  return z;
}

extern "C" int samble_launch_linear_fwd(const float* x, long x_bs, int B, int Cin, int N, const void* w_rm, int O, int epi,
                                        const float* ref, float* out, long o_bs, long o_rs, hipStream_t s) {
  const void* fns[6] = {reinterpret_cast<const void*>(lin_fwd_tri_kernel<kLinPlain>),
                        reinterpret_cast<const void*>(lin_fwd_tri_kernel<kLinLeaky>),
                        reinterpret_cast<const void*>(lin_fwd_tri_kernel<kLinMask>),
                        nullptr,
                        reinterpret_cast<const void*>(lin_fwd_tri_kernel<kLinLeakyBits>),
                        reinterpret_cast<const void*>(lin_fwd_tri_kernel<kLinMaskBits>)};
  hipError_t e = hipFuncSetAttribute(fns[epi], hipFuncAttributeMaxDynamicSharedMemorySize, kLinLds);
  if (e != hipSuccess) return (int)e;
  const int Z = lin_fwd_slices((N + 255) / 256 * B, O / 32);
  const dim3 grid((N + 255) / 256, B, Z);
  Timed timed(kT_lin_fwd, s);
  if (epi == kLinPlain)
    hipLaunchKernelGGL(lin_fwd_tri_kernel<kLinPlain>, grid, dim3(512), kLinLds, s, x, x_bs, Cin, N, (const char*)w_rm, O / 32 / Z, O,
                       out, o_bs, o_rs, nullptr, nullptr, nullptr, 0, nullptr);
  else if (epi == kLinLeaky)
    hipLaunchKernelGGL(lin_fwd_tri_kernel<kLinLeaky>, grid, dim3(512), kLinLds, s, x, x_bs, Cin, N, (const char*)w_rm, O / 32 / Z, O,
                       out, o_bs, o_rs, nullptr, nullptr, nullptr, 0, nullptr);
  else if (epi == kLinMask)
    hipLaunchKernelGGL(lin_fwd_tri_kernel<kLinMask>, grid, dim3(512), kLinLds, s, x, x_bs, Cin, N, (const char*)w_rm, O / 32 / Z, O,
                       out, o_bs, o_rs, ref, nullptr, nullptr, 0, nullptr);
  else if (epi == kLinLeakyBits)   // (`ref` carries the sign words: written here, read by kLinMaskBits)
    hipLaunchKernelGGL(lin_fwd_tri_kernel<kLinLeakyBits>, grid, dim3(512), kLinLds, s, x, x_bs, Cin, N, (const char*)w_rm,
                       O / 32 / Z, O, out, o_bs, o_rs, nullptr, nullptr, nullptr, 0,
                       reinterpret_cast<unsigned short*>(const_cast<float*>(ref)));
  else
    hipLaunchKernelGGL(lin_fwd_tri_kernel<kLinMaskBits>, grid, dim3(512), kLinLds, s, x, x_bs, Cin, N, (const char*)w_rm,
                       O / 32 / Z, O, out, o_bs, o_rs, nullptr, nullptr, nullptr, 0,
                       reinterpret_cast<unsigned short*>(const_cast<float*>(ref)));
  return (int)hipGetLastError();
}

// out (B, O, N) channel-major = W x [+ out]
extern "C" int samble_launch_linear_fwd_cm(const float* x, long x_bs, int B, int Cin, int N, const void* w_rm, int O,
                                           int accumulate, float* out, long o_bs, hipStream_t s) {
  const void* fn = reinterpret_cast<const void*>(lin_fwd_tri_kernel<kLinPlain, true>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLinLds);
  if (e != hipSuccess) return (int)e;
  Timed timed(kT_lin_fwd, s);
  const int Z = lin_fwd_slices((N + 255) / 256 * B, O / 32);
  hipLaunchKernelGGL((lin_fwd_tri_kernel<kLinPlain, true>), dim3((N + 255) / 256, B, Z), dim3(512), kLinLds, s, x, x_bs, Cin, N,
                     (const char*)w_rm, O / 32 / Z, O, out, o_bs, (long)N, nullptr, nullptr, nullptr, accumulate, nullptr);
  return (int)hipGetLastError();
}

// workspace: partial maxima + indices of the ceil(N / 256) * 8 point tiles of every cloud
extern "C" size_t samble_linear_amax_ws_bytes(int B, int N, int O) {
  return (size_t)B * ((N + 255) / 256) * 8 * O * 8;
}

extern "C" int samble_launch_linear_amax(const float* x, long x_bs, int B, int N, const void* w_rm, int O, float* y,
                                         int* arg, void* ws, hipStream_t s) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lin_fwd_tri_kernel<kLinAmax>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLinLds);
  if (e != hipSuccess) return (int)e;
  const int chunks = (N + 255) / 256, ntiles = chunks * 8;
  float* pmax = (float*)ws;
  int* parg = (int*)(pmax + (size_t)B * ntiles * O);
  Timed timed(kT_lin_amax, s);
  const int Z = lin_fwd_slices(chunks * B, O / 32);
  hipLaunchKernelGGL(lin_fwd_tri_kernel<kLinAmax>, dim3(chunks, B, Z), dim3(512), kLinLds, s, x, x_bs, 128, N, (const char*)w_rm,
                     O / 32 / Z, O, nullptr, 0, 0, nullptr, pmax, parg, 0, nullptr);
  hipLaunchKernelGGL(lin_amax_reduce_kernel, dim3((O + 255) / 256, B), dim3(256), 0, s, pmax, parg, ntiles, O, y, arg);
  return (int)hipGetLastError();
}

extern "C" int samble_launch_linear_dx_as(const float* g, long g_bs, long g_rs, const void* w_tr, int O, int B, int Cin, int N,
                                          float* dx, long dx_bs, const float* residual, hipStream_t s, int timing_id);
extern "C" int samble_launch_linear_dx(const float* g, long g_bs, long g_rs, const void* w_tr, int O, int B, int Cin, int N,
                                       float* dx, long dx_bs, const float* residual, hipStream_t s) {
  return samble_launch_linear_dx_as(g, g_bs, g_rs, w_tr, O, B, Cin, N, dx, dx_bs, residual, s, kT_lin_dx);
}
// (timing_id: the projection's input gradient runs on this kernel too, under its own measurement id)
extern "C" int samble_launch_linear_dx_as(const float* g, long g_bs, long g_rs, const void* w_tr, int O, int B, int Cin, int N,
                                          float* dx, long dx_bs, const float* residual, hipStream_t s, int timing_id) {
  if (kLinDuo) {
    int nw = 8;
    while (nw > 2 && (long)((N + 32 * nw - 1) / (32 * nw)) * B < kLinFillWgs) nw >>= 1;
    const void* fn = nw == 8   ? reinterpret_cast<const void*>(lin_dx_duo_kernel<8>)
                     : nw == 4 ? reinterpret_cast<const void*>(lin_dx_duo_kernel<4>)
                               : reinterpret_cast<const void*>(lin_dx_duo_kernel<2>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLinLds);
    if (e != hipSuccess) return (int)e;
    Timed timed(timing_id, s);
    const dim3 grid((N + 32 * nw - 1) / (32 * nw), B);
    if (nw == 8)
      hipLaunchKernelGGL(lin_dx_duo_kernel<8>, grid, dim3(512), kLinLds, s, g, g_bs, g_rs, (const char*)w_tr, O / 32, Cin, N, dx,
                         dx_bs, residual);
    else if (nw == 4)
      hipLaunchKernelGGL(lin_dx_duo_kernel<4>, grid, dim3(256), kLinLds, s, g, g_bs, g_rs, (const char*)w_tr, O / 32, Cin, N, dx,
                         dx_bs, residual);
    else
      hipLaunchKernelGGL(lin_dx_duo_kernel<2>, grid, dim3(128), kLinLds, s, g, g_bs, g_rs, (const char*)w_tr, O / 32, Cin, N, dx,
                         dx_bs, residual);
    return (int)hipGetLastError();
  }
  const void* fn = reinterpret_cast<const void*>(lin_dx_tri_kernel);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLinLds);
  if (e != hipSuccess) return (int)e;
  Timed timed(timing_id, s);   // (three-plane build)
  hipLaunchKernelGGL(lin_dx_tri_kernel, dim3((N + 255) / 256, B), dim3(512), kLinLds, s, g, g_bs, g_rs, (const char*)w_tr,
                     O / 32, Cin, N, dx, dx_bs, residual);
  return (int)hipGetLastError();
}

// mid (B, N, H) point-major rows (may be null), bits: samble_linear_sign_bytes(B, N, H) of sign words (written by the leaky
// form, read by the mask form), out / res (B, 128, N) channel-major
extern "C" int samble_launch_linear_chain(const float* x, long x_bs, int B, int N, const void* wa_rm, const void* wb_tr, int H,
                                          int epi, float* mid, long m_bs, long m_rs, void* bits, float* out, long o_bs,
                                          const float* res, hipStream_t s) {
  if (!kLinDuo) return (int)hipErrorNotSupported;
  int nw = 8;
  while (nw > 2 && (long)((N + 32 * nw - 1) / (32 * nw)) * B < kLinFillWgs) nw >>= 1;
  const dim3 grid((N + 32 * nw - 1) / (32 * nw), B);
  Timed timed(kT_lin_chain, s);
#define SAMBLE_CHAIN_LAUNCH(E, W)                                                                                              \
  do {                                                                                                                         \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lin_chain_kernel<E, W>),                                  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kChainLds);                                \
    if (e != hipSuccess) return (int)e;                                                                                        \
    hipLaunchKernelGGL((lin_chain_kernel<E, W>), grid, dim3(64 * W), kChainLds, s, x, x_bs, N, (const char*)wa_rm,             \
                       (const char*)wb_tr, H / 32, mid, m_bs, m_rs, (unsigned short*)bits, out, o_bs, res);                    \
  } while (0)
  if (epi == kLinLeakyBits) {
    if (nw == 8) SAMBLE_CHAIN_LAUNCH(kLinLeakyBits, 8);
    else if (nw == 4) SAMBLE_CHAIN_LAUNCH(kLinLeakyBits, 4);
    else SAMBLE_CHAIN_LAUNCH(kLinLeakyBits, 2);
  } else {
    if (nw == 8) SAMBLE_CHAIN_LAUNCH(kLinMaskBits, 8);
    else if (nw == 4) SAMBLE_CHAIN_LAUNCH(kLinMaskBits, 4);
    else SAMBLE_CHAIN_LAUNCH(kLinMaskBits, 2);
  }
#undef SAMBLE_CHAIN_LAUNCH
  return (int)hipGetLastError();
}

extern "C" size_t samble_linear_dw_ws_bytes(int B, int N, int O) {
  const int pts = ldw_points(B, N, O, O % 256 == 0 ? 256 : 128);
  return (size_t)B * ((N + pts - 1) / pts) * O * 128 * sizeof(float);
}

extern "C" int samble_launch_linear_dw(const float* g, long g_bs, long g_rs, const float* x, long x_bs, int B, int Cin, int N,
                                       int O, float* dW, int transposed, void* ws, hipStream_t s, int g_cm) {
  // (the workspace query assumes blocks of 256 outputs where O allows: at least as many chunks as the g_cm launch makes)
  const int pts = ldw_points(B, N, O, (O % 256 == 0 && !g_cm) ? 256 : 128), chunks = (N + pts - 1) / pts, nt = pts / kTile;
  Timed timed(kT_lin_dw, s);
  if (g_cm) {   // g (B, O, N): blocks of 128 outputs (two workgroups per CU)
    constexpr int lds = Ldw<1, true>::kLds;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lin_dw_tri_kernel<1, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((lin_dw_tri_kernel<1, true>), dim3(chunks, B, O / 128), dim3(512), lds, s, g, g_bs, g_rs, x, x_bs, Cin,
                       N, O, (float*)ws, nt);
  } else if (O % 256 == 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lin_dw_tri_kernel<2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Ldw<2>::kLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(lin_dw_tri_kernel<2>, dim3(chunks, B, O / 256), dim3(512), Ldw<2>::kLds, s, g, g_bs, g_rs, x, x_bs,
                       Cin, N, O, (float*)ws, nt);
  } else {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lin_dw_tri_kernel<1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Ldw<1>::kLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(lin_dw_tri_kernel<1>, dim3(chunks, B, O / 128), dim3(512), Ldw<1>::kLds, s, g, g_bs, g_rs, x, x_bs,
                       Cin, N, O, (float*)ws, nt);
  }
  const long n4 = (long)O * 128 / 4;
  hipLaunchKernelGGL(lin_sum_parts_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, s, (const float*)ws, B * chunks,
                     n4, dW, transposed ? O : 0);
  return (int)hipGetLastError();
}

extern "C" size_t samble_amax_bwd_ws_bytes(int B, int N, int O) { return (size_t)B * ((size_t)N + 2 + 2 * (size_t)O) * sizeof(int); }

// dx: zeros, or another gradient of the same tensor, on entry -- the arg-max columns are ADDED to (one wave per column)
extern "C" int samble_launch_amax_bwd(const float* x, long x_bs, int B, int N, const int* arg, const float* gy,
                                      const float* W, int O, float* dx, long dx_bs, float* dW, void* ws, hipStream_t s) {
  const size_t lds = (((size_t)N + 4) & ~(size_t)3) * sizeof(int) + (size_t)O * sizeof(int);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(amax_sort_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  Timed timed(kT_lin_amax_bwd, s);
  hipLaunchKernelGGL(amax_sort_kernel, dim3(B), dim3(1024), lds, s, N, arg, O, (int*)ws);
  hipLaunchKernelGGL(amax_dx_kernel, dim3(64, B), dim3(256), 0, s, N, gy, W, O, (const int*)ws, dx, dx_bs);   // (round 6: 64, was 16: 2-3 groups per wave instead of ~10 dependent chains)
  hipLaunchKernelGGL(amax_dw_kernel, dim3((O + 3) / 4), dim3(256), 0, s, x, x_bs, B, N, arg, gy, O, dW);
  return (int)hipGetLastError();
}
