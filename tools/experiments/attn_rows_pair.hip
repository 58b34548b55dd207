// EXPERIMENT (round 3), not part of the library.  State of commit 1f6d637: the library's K row image was three bf16
// planes then; since the logit products moved to two fp16 planes (tri_dev.h) the one-wave kernel reads its K image
// differently and rows_pair.py's bitwise comparison no longer applies (timings and stamps still do).
// What it was: pass 2 of the map-free forward with TWO waves per SIMD -- a pair of
// waves shares 32 sampled rows.  Bit-identical to attn_rows_rc_tri_kernel (x_ds and the P map), and slower:
// 221 us against 201 us in the step (tools/experiments/run_rows_pair.sh).  Kept because the measurements explain why
// a second wave per SIMD does not buy this kernel anything:
//   * the SIMD's matrix pipe goes to the OLDER wave whenever both waves have an MFMA ready.  Stamped
//     (tools/experiments/rows_pair.py stamps): wave A issues its 48 MFMAs of a tile back to back while B sits at its
//     first one, then B runs alone and A waits at the tile's barrier.  The pair is a serial schedule A, then B.
//   * vector instructions of one wave do issue under the other's MFMAs (tools/micro/pair_overlap_bench.hip: 24 MFMAs
//     | 120 v_fma in 398 ns against 356 ns for the MFMAs alone) -- but a wave that is stuck at an MFMA cannot reach
//     its vector instructions (in-order issue), so the overlap only happens by accident of the program order.
//   * what the pair does save (half the operand reads and DMA pieces per wave) is less than what it adds: the vector
//     work of the P tile is done twice, and the exchange + first operand reads at the top of each tile (~850 cycles
//     of the leading wave, nothing under them) come on top of 2 x 48 MFMAs at 38 cycles.  4 850 cycles per tile
//     against the one-wave kernel's 5 350 in isolation, level in the step.
//
// The design: attn_rows_rc_tri_kernel (attn_tri.hip) runs one wave per SIMD -- B M / 32 = 1 024 row blocks is all
// there is, and its 364 registers allow nothing else.  Here a workgroup is 8 waves = 128 rows; waves w and w + 4 sit
// on one SIMD and split the work of the same 32 rows so that
//   * the bits stay what they are: the 128-deep logit sum is ONE chain of 8 k-steps x 6 products; wave A (w < 4) runs
//     k-steps 0-3 of tile t+2 and leaves the accumulator in LDS, wave B picks it up one iteration later, runs k-steps
//     4-7 and has S -- same MFMAs in the same order as the one-wave kernel (the alternative, each wave summing its 64
//     channels and adding the halves, changes the order and would have to be mirrored in pass 1);
//   * P V splits by OUTPUT channel: A accumulates channels 0-63, B 64-127 (no exchange, same order per element);
//   * both waves exponentiate and split the P tile (B from its registers, A from the S that B left in LDS): the
//     vector work is duplicated, but each wave has half the MFMAs, half the operand reads and half the DMA pieces of
//     the one-wave kernel to hide behind it, and the partner's MFMAs to hide under;
//   * B writes the P map rows (16-byte pieces, straight from the registers).
// Per wave 130 registers less than the one-wave kernel (half the Q operand, two of the four output accumulators).
// LDS: K as half tiles (channels 0-63 for A, 64-127 for B: different tiles at any time) 2 x 2 x 12 KB, V 2 x 24 KB,
// the two exchanges 2 x (2 x 4 x 4 KB) = exactly 160 KB.  One barrier per tile.
// Iteration t (t = -2 .. ntiles):   A: chain 0-3 of tile t+2   B: chain 4-7 of tile t+1   both: P of tile t, P V of tile t-1.
#include "../../samble_amd/csrc/tri_dev.h"

#ifndef SAMBLE_PR_ABL
#define SAMBLE_PR_ABL 0  // timing-only ablations (tools/experiments/run_rows_pair.sh): 1 no vector work, 2 no barrier, 4 no DMA,
#endif                   // 8 no chain MFMAs, 16 no P V MFMAs, 32 no exchange writes

namespace samble {

constexpr int kPrHalf = kTriTile / 2;                  // one channel half of a row-image tile (groups 0-7 / 8-15)
constexpr int kPrKlo = 0;                              // [2][12 KB]
constexpr int kPrKhi = 2 * kPrHalf;                    // [2][12 KB]
constexpr int kPrV = 4 * kPrHalf;                      // [2][24 KB]
constexpr int kPrXlo = kPrV + 2 * kTriTile;            // [2][4 pairs][4 KB]: A's accumulator after k-step 3
constexpr int kPrXs = kPrXlo + 2 * 4 * 4096;           // [2][4 pairs][4 KB]: the finished logits of B
constexpr int kPrLds = kPrXs + 2 * 4 * 4096;
static_assert(kPrLds == 160 * 1024, "LDS budget of the pair kernel");

__device__ __forceinline__ const char* pr_uniform_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)v), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi32 << 32) | lo32);
}

#ifdef SAMBLE_STAMPS  // scratch builds only: s_memtime marks of workgroup 0, tile 20, waves 0 and 4 (one pair)
__device__ unsigned long long g_pr_stamps[2 * 16];
#define PR_STAMP(i) asm volatile("s_memtime %0" : "=s"(stamp[i]))
#else
#define PR_STAMP(i) do { } while (0)
#endif

template <bool PMAP>
__global__ __launch_bounds__(512, 2) void attn_rows_pair_tri_kernel(const char* __restrict__ Qimg,
                                                                    const char* __restrict__ Kimg,
                                                                    const char* __restrict__ Vtr,
                                                                    const float* __restrict__ lse,
                                                                    const long long* __restrict__ idx, int N, int NK,
                                                                    int M, float scale, float* __restrict__ xds,
                                                                    float* __restrict__ pmap, int ld) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  const int pair = wave & 3, c = wave >> 2;  // c = 0: wave A (channels 0-63), 1: wave B
  int chunk, b;
  xcd_assign(chunk, b);
  const int m0 = chunk * 128 + pair * 32;
  const int mrow = m0 + lo;
  const bool mvalid = mrow < M;
  const int mc = mvalid ? mrow : M - 1;  // rows past M-1 recompute and rewrite row M-1's values (same bytes)
  const long row = idx[(long)b * M + mc];
  const float my_lse = lse[(long)b * N + row];
  const int qtiles = (N + kTile - 1) / kTile, ntiles = (NK + kTile - 1) / kTile;
  const char* Kb = Kimg + (long)b * ntiles * kTriTile;
  const char* Vb = Vtr + (long)b * ntiles * kTriTile;
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(smem_c);

  // Q operand of this wave's four k-steps (4 c .. 4 c + 3)
  u32x4 q[12];
  {
    const u32x4* qp = reinterpret_cast<const u32x4*>(Qimg + ((long)b * qtiles + (int)(row >> 5)) * kTriTile +
                                                     tri_rm_off((int)(row & 31), h, 0));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      q[3 * j] = qp[192 * (4 * c + j)];
      q[3 * j + 1] = qp[192 * (4 * c + j) + 32];
      q[3 * j + 2] = qp[192 * (4 * c + j) + 64];
    }
  }
  // DMA of iteration t: K_lo(t+3) -> lo slot (t+3)&1, K_hi(t+2) -> hi slot t&1 (1 536 chunks of 16 B: 3 per thread, the
  // first 768 the lo half), V(t) -> V slot t&1 (3 per thread).  Tiles past either end: clamped (landed, never used).
  auto clampt = [&](int t) { return min(max(t, 0), ntiles - 1); };
  auto dma = [&](int t) {
    const char* klo = pr_uniform_ptr(Kb + (long)clampt(t + 3) * kTriTile);
    const char* khi = pr_uniform_ptr(Kb + (long)clampt(t + 2) * kTriTile + kPrHalf);
    const char* vsrc = pr_uniform_ptr(Vb + (long)clampt(t) * kTriTile);
    char* dlo = smem_c + kPrKlo + ((t + 3) & 1) * kPrHalf;
    char* dhi = smem_c + kPrKhi + (t & 1) * kPrHalf;
    char* dv = smem_c + kPrV + (t & 1) * kTriTile;
    // piece 0: chunks 0..511 of the lo half; piece 1: waves 0-3 chunks 512..767 of lo, waves 4-7 chunks 0..255 of hi;
    // piece 2: chunks 256..767 of hi
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(klo + tid * 16),
                                     (__attribute__((address_space(3))) void*)(dlo + wave * 1024), 16, 0, 0);
    if (wave < 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(klo + (512 + tid) * 16),
                                       (__attribute__((address_space(3))) void*)(dlo + (8 + wave) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(khi + (tid - 256) * 16),
                                       (__attribute__((address_space(3))) void*)(dhi + (wave - 4) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(khi + (256 + tid) * 16),
                                     (__attribute__((address_space(3))) void*)(dhi + (4 + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc + (tid + 512 * k) * 16),
                                       (__attribute__((address_space(3))) void*)(dv + (wave + 8 * k) * 1024), 16, 0, 0);
  };
  // prologue: K_lo(0) for A's first chain (t = -2: tile 0); everything else arrives through the loop's own DMA
  {
    const char* klo = Kb;
    char* dlo = smem_c + kPrKlo;  // slot 0
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(klo + tid * 16),
                                     (__attribute__((address_space(3))) void*)(dlo + wave * 1024), 16, 0, 0);
    if (wave < 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(klo + (512 + tid) * 16),
                                       (__attribute__((address_space(3))) void*)(dlo + (8 + wave) * 1024), 16, 0, 0);
  }
  f32x16 oacc[2] = {zero16(), zero16()};
  f32x16 s_cur = zero16();  // B: the logits of tile t (its chain of iteration t-1); A: read from the exchange
  Tri bp[2];                // P^T fragments (two k-steps of 16 keys) of tile t-1
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) bp[ks] = Tri{u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
  const unsigned xoff = (unsigned)(pair * 4096 + lane * 16);  // this pair's block, lane-linear 16-byte words (x 4: + 1 KB)
  // the first iterations multiply tiles that do not exist by P = 0: the V slots must hold finite numbers by then
  for (int k = tid; k < 2 * kTriTile / 16; k += 512) reinterpret_cast<u32x4*>(smem_c + kPrV)[k] = u32x4{0, 0, 0, 0};
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // the compiler cannot see that wait: have it settle its own count for the prologue's loads HERE, not in the loop
  // (where its merged state would cost a vmcnt(0) behind every iteration's DMA)
#pragma unroll
  for (int j = 0; j < 12; ++j) asm volatile("" : "+v"(q[j]));
  {
    float l = my_lse;
    asm volatile("" : "+v"(l));
  }

  for (int t = -2; t <= ntiles; ++t) {
    const bool vec_valid = t >= 0 && t < ntiles;
#ifdef SAMBLE_STAMPS
    unsigned long long stamp[16];
#endif
    PR_STAMP(0);
    // ---- the chain's start: A from zero, B from A's accumulator of the previous iteration; A's logits from B's ----
    const int tch = c == 0 ? t + 2 : t + 1;   // tile of this wave's chain
    f32x16 ch = zero16();
    const unsigned xlo_r = lds0 + kPrXlo + (unsigned)(((t + 1) & 1) * 16384) + xoff;  // tile t+1's lo accumulator
    const unsigned xs_r = lds0 + kPrXs + (unsigned)((t & 1) * 16384) + xoff;          // tile t's logits
    u32x4 xin[4];
    if (c == 1) {
      xin[0] = lds_ld128<0>(xlo_r); xin[1] = lds_ld128<1024>(xlo_r); xin[2] = lds_ld128<2048>(xlo_r); xin[3] = lds_ld128<3072>(xlo_r);
    } else {
      xin[0] = lds_ld128<0>(xs_r); xin[1] = lds_ld128<1024>(xs_r); xin[2] = lds_ld128<2048>(xs_r); xin[3] = lds_ld128<3072>(xs_r);
    }
    // operands of the first k-step / the first P V step
    const unsigned ka = lds0 + (c == 0 ? kPrKlo : kPrKhi) + (unsigned)((tch & 1) * kPrHalf) + (unsigned)((96 * h + lo) * 16);
    const unsigned va = lds0 + kPrV + (unsigned)(((t - 1) & 1) * kTriTile) + (unsigned)((32 * 2 * c + lo) * 16) +
                        (unsigned)(3 * h * 128 * 16);  // tri_tr_off(32 (2 c) + lo, h, 0); step i: + (dt & 1) 512 + ks 12288
    Tri k0 = {lds_ld128<0>(ka), lds_ld128<512>(ka), lds_ld128<1024>(ka)}, k1 = k0;
    Tri v0 = {lds_ld128<0>(va), lds_ld128<2048>(va), lds_ld128<4096>(va)}, v1 = v0;
    DUO_LGKM_WAIT(0);
    PR_STAMP(1);
    asm volatile("" : "+v"(xin[0]), "+v"(xin[1]), "+v"(xin[2]), "+v"(xin[3]), "+v"(k0.h), "+v"(k0.m), "+v"(k0.l), "+v"(v0.h),
                      "+v"(v0.m), "+v"(v0.l));
    if (c == 1) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) ch[4 * g + e] = __uint_as_float(xin[g][e]);
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) s_cur[4 * g + e] = __uint_as_float(xin[g][e]);
    }
    if (!vec_valid || t == ntiles - 1) {  // no such tile / padding keys of the last tile: P = 0 there
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (!vec_valid || t * kTile + crow(r, h) >= NK) s_cur[r] = -__builtin_huge_valf();
    }
    if constexpr (!(SAMBLE_PR_ABL & 4)) dma(t);
    PR_STAMP(2);
    float* pout = PMAP ? pmap + ((long)b * M + min(m0 + lo, M - 1)) * ld + t * kTile + 4 * h : nullptr;
    Tri bn[2];
    if constexpr (SAMBLE_PR_ABL & 1) bn[0] = bn[1] = Tri{u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
    float p[16];
    float va0 = 0.f, va1 = 0.f;
    unsigned vh = 0, vm = 0;
    // ---- 48 slots: even = chain MFMA, odd = P V MFMA; the vector work and the operand reads spread between them ----
    static_for<0, 48>([&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      constexpr int m = s >> 1;  // MFMA number within its stream
      constexpr int step = m / 6, prod = m % 6;
      if constexpr (s % 6 == 0 && s > 0) PR_STAMP(2 + s / 6);
      if constexpr ((s & 1) == 0 && (SAMBLE_PR_ABL & 8)) {
      } else if constexpr ((s & 1) == 1 && (SAMBLE_PR_ABL & 16)) {
      } else if constexpr ((s & 1) == 0) {  // chain k-step `step` (of this wave's four), product `prod`
        if constexpr (prod == 0 && step > 0) {
          DUO_LGKM_WAIT(0);  // (the operands of this k-step and of P V step `step`: read a dozen slots ago)
          asm volatile("" : "+v"(k0.h), "+v"(k0.m), "+v"(k0.l), "+v"(v0.h), "+v"(v0.m), "+v"(v0.l));
        }
        const Tri bq = {q[3 * step], q[3 * step + 1], q[3 * step + 2]};
        if constexpr (prod == 0) ch = mfma_bf(k0.m, bq.m, ch);
        if constexpr (prod == 1) ch = mfma_bf(k0.h, bq.l, ch);
        if constexpr (prod == 2) ch = mfma_bf(k0.l, bq.h, ch);
        if constexpr (prod == 3) ch = mfma_bf(k0.h, bq.m, ch);
        if constexpr (prod == 4) ch = mfma_bf(k0.m, bq.h, ch);
        if constexpr (prod == 5) ch = mfma_bf(k0.h, bq.h, ch);
      } else {  // P V step `step`: k-step step >> 1 of the 32 keys, output block 2 c + (step & 1)
        constexpr int ks = step >> 1, dtl = step & 1;
        if constexpr (prod == 0) oacc[dtl] = mfma_bf(v0.m, bp[ks].m, oacc[dtl]);
        if constexpr (prod == 1) oacc[dtl] = mfma_bf(v0.h, bp[ks].l, oacc[dtl]);
        if constexpr (prod == 2) oacc[dtl] = mfma_bf(v0.l, bp[ks].h, oacc[dtl]);
        if constexpr (prod == 3) oacc[dtl] = mfma_bf(v0.h, bp[ks].m, oacc[dtl]);
        if constexpr (prod == 4) oacc[dtl] = mfma_bf(v0.m, bp[ks].h, oacc[dtl]);
        if constexpr (prod == 5) oacc[dtl] = mfma_bf(v0.h, bp[ks].h, oacc[dtl]);
      }
      // operand reads one step ahead: issued early in the step, consumed 10 slots later
      if constexpr (s % 12 == 2 && step < 3) {
        k1 = Tri{lds_ld128<3072 * (step + 1)>(ka), lds_ld128<3072 * (step + 1) + 512>(ka),
                 lds_ld128<3072 * (step + 1) + 1024>(ka)};
      }
      if constexpr (s % 12 == 3 && step < 3) {
        constexpr int nks = (step + 1) >> 1, ndt = (step + 1) & 1;
        v1 = Tri{lds_ld128<nks * 12288 + ndt * 512>(va), lds_ld128<nks * 12288 + ndt * 512 + 2048>(va),
                 lds_ld128<nks * 12288 + ndt * 512 + 4096>(va)};
      }
      if constexpr (s % 12 == 11) {  // rotate at the end of a step
        k0 = k1;
        v0 = v1;
      }
      // vector work on tile t, a few instructions behind EVERY MFMA (a wave that bunches them leaves the matrix pipe
      // idle whenever its partner is not ready -- and the partner, the younger wave, gets the pipe only when this one
      // does not want it): slice i = logits 2 i, 2 i + 1 in the six slots 6 i .. 6 i + 5: scale, - lse, exp, and the
      // three planes one at a time
      if constexpr (!(SAMBLE_PR_ABL & 1)) {
#pragma clang fp contract(off)  // the statistics pass formed lse from round(s * scale): no fma here
        constexpr int i = s / 6, st = s % 6, r0 = 2 * i, r1 = 2 * i + 1;
        if constexpr (st == 0) {
          va0 = s_cur[r0];
          va1 = s_cur[r1];
          asm volatile("" : "+v"(va0), "+v"(va1));
          va0 = va0 * scale;
          va1 = va1 * scale;
        }
        if constexpr (st == 1) {
          va0 = va0 - my_lse;
          va1 = va1 - my_lse;
        }
        if constexpr (st == 2) {
          p[r0] = va0 = __expf(va0);
          p[r1] = va1 = __expf(va1);
        }
        if constexpr (st == 3) {
          vh = bf16_pack2(va0, va1);
          va0 = va0 - __uint_as_float(vh << 16);
          va1 = va1 - __uint_as_float(vh & 0xFFFF0000u);
        }
        if constexpr (st == 4) {
          vm = bf16_pack2(va0, va1);
          va0 = va0 - __uint_as_float(vm << 16);
          va1 = va1 - __uint_as_float(vm & 0xFFFF0000u);
        }
        if constexpr (st == 5) {
          unsigned vl = bf16_pack2(va0, va1);
          asm volatile("" : "+v"(vl));
          bn[i >> 2].h[i & 3] = vh;
          bn[i >> 2].m[i & 3] = vm;
          bn[i >> 2].l[i & 3] = vl;
          if constexpr (PMAP && (i & 1) == 1) {  // B: four consecutive keys of the row as one 16-byte piece
            if (c == 1 && vec_valid) {
              constexpr int g = i >> 1;
              *reinterpret_cast<f32x4*>(pout + 8 * g) = f32x4{p[4 * g], p[4 * g + 1], p[4 * g + 2], p[4 * g + 3]};
            }
          }
        }
        asm volatile("" : "+v"(va0), "+v"(va1), "+v"(vh), "+v"(vm));
      }
      asm volatile("" : "+v"(ch), "+v"(oacc[0]), "+v"(oacc[1]) : : "memory");
      __builtin_amdgcn_sched_barrier(0);
    });
    PR_STAMP(10);
    // ---- hand the chain on: A's accumulator of tile t+2 / B's finished logits of tile t+1 ----
    if constexpr (!(SAMBLE_PR_ABL & 32)) {
      const unsigned xw = lds0 + (c == 0 ? kPrXlo + (unsigned)((t & 1) * 16384) : kPrXs + (unsigned)(((t + 1) & 1) * 16384)) + xoff;
      float* xp = reinterpret_cast<float*>(smem_c + (xw - lds0));
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(xp + 256 * g) = f32x4{ch[4 * g], ch[4 * g + 1], ch[4 * g + 2], ch[4 * g + 3]};
    }
    if (c == 1) s_cur = ch;
    bp[0] = bn[0];
    bp[1] = bn[1];
    // the DMA of this iteration has to have landed before anyone reads it (B's four map stores are younger)
    if (PMAP && c == 1 && vec_valid) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SAMBLE_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PR_STAMP(11);
    asm volatile("s_barrier" ::: "memory");
    PR_STAMP(12);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && pair == 0 && t == 20)
      for (int i = 0; i < 13; ++i) g_pr_stamps[c * 16 + i] = stamp[i];
#else
    if constexpr (SAMBLE_PR_ABL & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
  }
  if (mvalid) {
    float* ob = xds + (long)b * 128 * M + mrow;
#pragma unroll
    for (int dtl = 0; dtl < 2; ++dtl) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ob[(long)(32 * (2 * c + dtl) + crow(r, h)) * M] = oacc[dtl][r];
    }
  }
}

}  // namespace samble

using namespace samble;

extern "C" __attribute__((visibility("default"))) int samble_experiment_attn_rows_pair_tri(const void* qimg, const void* kimg, const void* v_tr_image, const float* lse,
                                                const long long* idx, int B, int N, int nt, int M, float scale, float* xds,
                                                float* pmap, int ld, hipStream_t stream) {
  for (const void* f : {reinterpret_cast<const void*>(attn_rows_pair_tri_kernel<false>),
                        reinterpret_cast<const void*>(attn_rows_pair_tri_kernel<true>)}) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kPrLds);
    if (e != hipSuccess) return (int)e;
  }
  if (pmap)
    hipLaunchKernelGGL(attn_rows_pair_tri_kernel<true>, dim3((M + 127) / 128, B), dim3(512), kPrLds, stream,
                       (const char*)qimg, (const char*)kimg, (const char*)v_tr_image, lse, idx, N, N + nt, M, scale, xds,
                       pmap, ld);
  else
    hipLaunchKernelGGL(attn_rows_pair_tri_kernel<false>, dim3((M + 127) / 128, B), dim3(512), kPrLds, stream,
                       (const char*)qimg, (const char*)kimg, (const char*)v_tr_image, lse, idx, N, N + nt, M, scale, xds,
                       pmap, ld);
  return (int)hipGetLastError();
}
#ifdef SAMBLE_STAMPS
extern "C" __attribute__((visibility("default"))) int samble_scratch_pr_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(samble::g_pr_stamps), sizeof(unsigned long long) * 32);
}
#endif
