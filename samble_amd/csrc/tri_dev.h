// fp32-equivalent matrix products on the bf16 matrix cores of gfx950 ("tri" operands).
//
// An fp32 value a is carried as three bf16 pieces a = h + m + l (h = bf16(a), m = bf16(a - h),
// l = bf16(a - h - m); the two subtractions are exact in fp32, so the three pieces hold all 24
// significand bits).  A product of two such operands keeps the six partial products whose weight is
// >= 2^-16 of the leading one:
//     a b ~= hh + (hm + mh) + (hl + lh + mm)          dropped: ml, lm, ll  (<= 2^-23 |a||b|)
// each of them one v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Measured against fp64 on random
// data (tools/micro/split_mfma_bench.hip): rms error 1.5e-7 of rms(C) for a 128-deep contraction, the
// fp32 instruction v_mfma_f32_32x32x2_f32 gives 2.0e-7 -- the scheme IS fp32 arithmetic for this
// path's purposes, at 6 x 32 = 192 matrix-pipe cycles per 32x32x16 block instead of 8 x 64 = 512,
// and unlike the fp32 MFMA the bf16 MFMA leaves the vector ALU free while it runs.
//
// Operand maps of v_mfma_f32_32x32x16_bf16 (cdna_hip_programming.md, fragment layout): lane l = (x =
// l & 31, h = l >> 5) holds A[row x][k = 8h + e] and B[k = 8h + e][col x], e = 0..7 (16 bytes);
// C/D as the fp32 instruction: D[row crow(r, h)][col x] in register r.
//
// Operand IMAGES in memory (HBM and LDS use the same bytes, tiles are copied verbatim).  A tile covers
// 32 rows of the matrix and is kTriTile = 24576 bytes = 1536 chunks of 16 bytes (8 bf16), CHUNK-MAJOR:
// chunk (c, r) at (c * R + r) * 16, so the 64 lanes of an operand read (r = lane & 31 or the channel)
// touch consecutive 16-byte words: conflict-free ds_read_b128 without padding, coalesced global reads.
//   RM image of a (rows x 128) matrix, contraction over the 128 channels: R = 32 tile rows,
//     c = 3 (channel / 8) + piece (48 per tile).  k-step ks (0..7) of a lane in half h uses channel
//     group 2 ks + h: three ds_read_b128 (h, m, l).
//   TR image of the same matrix, contraction over the 32 ROWS of a tile (P V, dS^T Q ...): R = 128
//     channels, c = 3 (2 s + h) + piece (12 per tile); the 8 elements of chunk group 2 s + h are the
//     tile rows j = 16 s + 8 (e >> 2) + 4 h + (e & 3), e = 0..7 -- the order in which a 32x32
//     accumulator presents its rows when it is used as the other operand (accumulator-as-operand).
#pragma once
#include <type_traits>
#include <utility>

#include "samble_dev.h"

// Non-temporal hints on the M-row maps of the backward (P written by attn_rows_rc_tri, read by bwd_dq_pm_tri and the dV
// kernel; dS written by bwd_dq_pm_tri, read by the dK kernel): bit 0 the P stores, 1 the dQ kernel's P loads, 2 its dS
// stores, 3 the key-stationary kernels' map loads.  The value shipped is the one the same-box A/B picked.
#ifndef SAMBLE_MAP_NT
#define SAMBLE_MAP_NT 9
#endif

#ifndef SAMBLE_KACC_DK_FIRST
#define SAMBLE_KACC_DK_FIRST 1
#endif

namespace samble {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kTriTile = 32 * 128 * 6;          // bytes per 32-row tile, either image kind
constexpr int kTriTileChunks = kTriTile / 16;   // 1536
// byte offset inside an RM tile of (row r, channel group g = channel / 8, piece p)
__host__ __device__ constexpr int tri_rm_off(int r, int g, int p) { return ((3 * g + p) * 32 + r) * 16; }
// byte offset inside a TR tile of (channel d, chunk group cg = 2 s + h, piece p)
__host__ __device__ constexpr int tri_tr_off(int d, int cg, int p) { return ((3 * cg + p) * 128 + d) * 16; }

struct Tri {
  u32x4 h, m, l;
};

__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// c += a b over one 16-deep k-step, six partial products, smallest first
__device__ __forceinline__ f32x16 mfma_tri(const Tri& a, const Tri& b, f32x16 c) {
#ifndef SAMBLE_TRI_HALF  // (timing-only scratch builds, tools/tri_half.sh: what three products per k-step would cost)
  c = mfma_bf(a.m, b.m, c);
  c = mfma_bf(a.h, b.l, c);
  c = mfma_bf(a.l, b.h, c);
#endif
  c = mfma_bf(a.h, b.m, c);
  c = mfma_bf(a.m, b.h, c);
  c = mfma_bf(a.h, b.h, c);
  return c;
}

// ---- two fp16 planes for the LOGIT products (round 3) ---------------------------------------------------------
// S = Q K^T is the one product of the attention passes that is formed tile by tile with nothing accumulated across
// tiles, so each operand block can carry its own power-of-two scale: a K row-image tile is rewritten IN PLACE as two
// fp16 planes of the tile's values x 2^e (e: max |k| of the tile lands just under 2^12), h = fp16(x), l = fp16(x - h),
// in the slots of the h and m pieces; the l slot of (group 0, row 0) holds 2^-e as a float.  A query row is converted
// in registers, scale from its own row.  Three products (lh + hl + hh; the dropped ll <= 2^-22 |a||b|) instead of
// six, and the accumulator x 2^-(e_q + e_k) -- exact -- is the logit's fp32 numerator as before.  22 significant
// bits per operand: the sampled indices are as close to the oracle's as any other fp32 evaluation of the sums
// (tools/experiments/duo_identity_probe.py); P V and the backward stay on three bf16 planes.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_hf(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_duo(u32x4 ah, u32x4 al, u32x4 bh, u32x4 bl, f32x16 c) {
  c = mfma_hf(al, bh, c);
  c = mfma_hf(ah, bl, c);
  c = mfma_hf(ah, bh, c);
  return c;
}
constexpr int kDuoScaleSlot = ((3 * 0 + 2) * 32 + 0) * 16;  // tri_rm_off(0, 0, 2): where a converted K tile keeps 2^-e
// ... and, in the word behind it, this tag: two bf16 NaNs, which the l plane of a three-plane image (the residual of a
// finite value) cannot hold -- how samble_tri_k_logit_form tells a converted tile from a raw one (the conversion is
// idempotent: a tile that carries the tag is left alone)
constexpr unsigned kDuoTag = 0xFFC07FC0u;
// csrc/linear.hip's 1x1-convolution kernels (and, through lin_dx, the projection's input gradient) run on two fp16 planes;
// -DSAMBLE_LIN_DUO=0 builds the three-bf16-plane kernels instead (A/B runs).  Both files that write those weight images
// (linear.hip, proj_tri.hip's prologue) follow this switch.
#ifndef SAMBLE_LIN_DUO
#define SAMBLE_LIN_DUO 1
#endif
constexpr bool kLinDuo = SAMBLE_LIN_DUO != 0;
// 2^e with amax x 2^e in [2^12, 2^13); the exponent is clamped so that 2^e and 2^-e are normal.  A tile of zeros
// (amax = 0 or denormal) gets the LARGEST exponent: its 2^-e never is the maximum over a cloud's tiles
__device__ __forceinline__ void duo_scale_for(float amax, float& s, float& inv) {
  const int ex = (int)((__float_as_uint(amax) >> 23) & 0xFFu);
  const int se = ex == 0 ? 100 : max(-100, min(100, 12 - (ex - 127)));
  s = __uint_as_float((unsigned)(127 + se) << 23);
  inv = __uint_as_float((unsigned)(127 - se) << 23);
}
// two scaled fp32 values -> the packed words of the two fp16 planes
__device__ __forceinline__ void duo_split2(float x0, float x1, unsigned& h, unsigned& l) {
  const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;  // round to nearest even
  const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);  // exact subtractions
  h = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
  l = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
}
// eight fp32 values x s -> the two fp16 planes of one 16-byte operand chunk
__device__ __forceinline__ void duo_split8s(const float (&v)[8], float s, u32x4& hp, u32x4& lp) {
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    unsigned hw, lw;
    duo_split2(v[2 * w] * s, v[2 * w + 1] * s, hw, lw);
    hp[w] = hw;
    lp[w] = lw;
  }
}
// largest value over the 64 lanes, in every lane
__device__ __forceinline__ float wave_amax64(float x) {
  return wave_max64(x);
}
// ---- products that ACCUMULATE over tiles (out^T += A_tile^T X_tile: dV^T += dO^T P, dK^T += Q^T dS) ------------------
// The A tiles (transposed images of the sampled rows) carry per-tile scales 2^e_t; their X blocks are split in the
// kernel anyway, so X is multiplied by 2^(13 - (e_t - e_min)) x xs before ITS split (xs = 1 for probabilities, a power of
// two that puts the cloud's largest |dS| under 1 for dS): every term of the sum carries 2^(13 + e_min) xs, taken out once
// at the end.  A tile of small values (large e_t) gets its X scaled less: what that loses is relative to a contribution
// that is small already.  The tiles' 2^-e_t are read once, lane l keeping those of tiles l, l + 64, ... (kDuoTsRegs
// registers: ntiles <= 64 kDuoTsRegs), and the loop gets tile t's factor by a lane permute -- a load per tile would join
// the hand-counted vmcnt queue of the DMA rings (and v_readlane of a register a load wrote draws a compiler-placed
// vmcnt(0) into the loop).
constexpr int kDuoTrScaleSlot = ((3 * 0 + 2) * 128 + 0) * 16;  // tri_tr_off(0, 0, 2): a transposed-image tile's 2^-e
constexpr int kDuoTsRegs = 8;
struct DuoTileScales {
  float inv[kDuoTsRegs];
  float rcp_max;
  __device__ __forceinline__ void load(const char* __restrict__ img, int ntiles, int lane) {
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < kDuoTsRegs; ++j) {
      const int t = 64 * j + lane;
      inv[j] = t < ntiles ? *reinterpret_cast<const float*>(img + (long)t * kTriTile + kDuoTrScaleSlot) : 0.f;
      mx = fmaxf(mx, inv[j]);
    }
    mx = wave_max64(mx);
    rcp_max = 1.f / mx;  // (mx = 2^-e_min: a power of two, the reciprocal is exact)
#pragma unroll
    for (int j = 0; j < kDuoTsRegs; ++j) {  // (into registers a vector-ALU instruction wrote: see above)
      float c;
      asm volatile("v_mov_b32 %0, %1" : "=v"(c) : "v"(inv[j]));
      inv[j] = c;
    }
  }
  __device__ __forceinline__ float x_scale(int t) const {  // 2^(13 - (e_t - e_min)); t wave-uniform
    float v = inv[0];
#pragma unroll
    for (int j = 1; j < kDuoTsRegs; ++j) v = (t >> 6) == j ? inv[j] : v;
    return 8192.f * __shfl(v, t & 63, 64) * rcp_max;
  }
  __device__ __forceinline__ float unscale() const { return (1.f / 8192.f) / rcp_max; }  // 2^-(13 + e_min)
};

// the 8 fp32 values of one 16-byte chunk triple (h, m, l pieces of 8 consecutive channels): exact sums
__device__ __forceinline__ void tri_chunk_values(u32x4 hh, u32x4 mm, u32x4 ll, float (&x)[8]) {
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    x[2 * w] = (__uint_as_float(hh[w] << 16) + __uint_as_float(mm[w] << 16)) + __uint_as_float(ll[w] << 16);
    x[2 * w + 1] = (__uint_as_float(hh[w] & 0xFFFF0000u) + __uint_as_float(mm[w] & 0xFFFF0000u)) +
                   __uint_as_float(ll[w] & 0xFFFF0000u);
  }
}
// this lane's half of a query row (q[3 ks + piece], as loaded from the Q row image) -> two fp16 planes qd[2 ks + plane]
// under the ROW's own scale (both halves of the row agree on it: lanes x and x + 32), unscale = 2^-e
__device__ __forceinline__ void duo_q_from_tri(const u32x4 (&q)[24], u32x4 (&qd)[16], float& unscale) {
  float x[64];
  float amax = 0.f;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    float v[8];
    tri_chunk_values(q[3 * ks], q[3 * ks + 1], q[3 * ks + 2], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      x[8 * ks + e] = v[e];
      amax = fmaxf(amax, fabsf(v[e]));
    }
  }
  amax = xor32_max(amax);
  float s;
  duo_scale_for(amax, s, unscale);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned hw, lw;
      duo_split2(x[8 * ks + 2 * w] * s, x[8 * ks + 2 * w + 1] * s, hw, lw);
      qd[2 * ks][w] = hw;
      qd[2 * ks + 1][w] = lw;
    }
}

__device__ __forceinline__ unsigned bf16_bits(float x) {  // round to nearest even
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x);
}
__device__ __forceinline__ float bf16_value(unsigned bits) { return __uint_as_float(bits << 16); }

// two fp32 values -> one packed word of bf16 (lo = a, hi = b), round to nearest even: ONE v_cvt_pk_bf16_f32
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf16_pack2(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// split two fp32 values into the packed (lo = x0, hi = x1) words of the three planes: 3 packed conversions, 4 bit
// operations to widen the halves again, 4 exact subtractions (scalar conversions + packing cost 18 instructions;
// these are issued beside MFMAs that hold the same issue port)
__device__ __forceinline__ void tri_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = bf16_pack2(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);
  m = bf16_pack2(r0, r1);
  const float q0 = r0 - __uint_as_float(m << 16), q1 = r1 - __uint_as_float(m & 0xFFFF0000u);
  l = bf16_pack2(q0, q1);
}

__device__ __forceinline__ Tri tri_split8(const float (&x)[8]) {
  Tri t;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned h, m, l;
    tri_split2(x[2 * i], x[2 * i + 1], h, m, l);
    t.h[i] = h;
    t.m[i] = m;
    t.l[i] = l;
  }
  return t;
}

// A run of NSTEP k-steps whose LDS operand `fetch(i)` is requested two steps before `use(i, operand)`
// issues its six MFMAs.  For kernels that run ONE wave per SIMD: nothing else hides the LDS latency
// between a ds_read and the MFMA that consumes it, and left to itself the compiler places each read
// right in front of its use (62 s_waitcnt per 144 MFMAs in the first build of the dK/dV kernel).
// Same with the operand requested THREE steps ahead (+12 registers): for kernels whose two waves per SIMD hit the
// LDS together (stamped in bwd_kacc_tri: 45 cycles per MFMA with both waves in their product phase)
template <int NSTEP, class Fetch, class Use>
__device__ __forceinline__ void tri_pipelined3(Fetch fetch, Use use) {
  Tri a0 = fetch(0), a1 = fetch(NSTEP > 1 ? 1 : 0), a2 = fetch(NSTEP > 2 ? 2 : 0);
#pragma unroll
  for (int i = 0; i < NSTEP; ++i) {
    Tri a3 = a2;
    if (i + 3 < NSTEP) a3 = fetch(i + 3);
    __builtin_amdgcn_sched_barrier(0);
    use(i, a0);
    __builtin_amdgcn_sched_barrier(0);
    a0 = a1;
    a1 = a2;
    a2 = a3;
  }
}

template <int NSTEP, class Fetch, class Use>
__device__ __forceinline__ void tri_pipelined(Fetch fetch, Use use) {
  Tri a0 = fetch(0), a1 = fetch(NSTEP > 1 ? 1 : 0);
#pragma unroll
  for (int i = 0; i < NSTEP; ++i) {
    Tri a2 = a1;
    if (i + 2 < NSTEP) a2 = fetch(i + 2);
    __builtin_amdgcn_sched_barrier(0);
    use(i, a0);
    __builtin_amdgcn_sched_barrier(0);
    a0 = a1;
    a1 = a2;
  }
}

// Staging of one image tile (1536 16-byte chunks, copied verbatim) by NT threads: loads issued early,
// committed to LDS later.  Conflict-free 16-byte LDS stores (consecutive lanes, consecutive chunks).
template <int NT>
struct TriStage {
  static constexpr int kPer = kTriTileChunks / NT;
  static_assert(kPer * NT == kTriTileChunks, "thread count must divide the tile");
  u32x4 v[kPer];
  __device__ __forceinline__ void issue(const char* __restrict__ tile, int tid) {
    const u32x4* s = reinterpret_cast<const u32x4*>(tile);
#pragma unroll
    for (int i = 0; i < kPer; ++i) v[i] = s[tid + NT * i];
  }
  __device__ __forceinline__ void commit(char* __restrict__ lds, int tid) const {
    u32x4* d = reinterpret_cast<u32x4*>(lds);
#pragma unroll
    for (int i = 0; i < kPer; ++i) d[tid + NT * i] = v[i];
  }
};

// LDS reads whose wait is placed by hand.  The compiler counts lgkmcnt itself, but not across the branches of the
// woven selection and not the way the weave needs it: it hoists a step's prefetch reads above the step's first MFMA
// and then waits for ALL of them (lgkmcnt(0)) to get at that MFMA's operands -- an exposed LDS round trip in front of
// every group of MFMAs (tools/micro/weave_bench.hip: +45..70 cycles per MFMA slot).  An asm read is opaque to it: the
// destination counts as written at the asm statement, so every consumer sits behind an explicit
// `s_waitcnt lgkmcnt(n)` (n = reads issued later that may still be in flight; LDS returns in order) and an empty asm
// that "rewrites" the registers (uses cannot rise above it).
template <int OFF>
__device__ __forceinline__ u32x4 lds_ld128(unsigned addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
#define DUO_LGKM_WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

}  // namespace samble
