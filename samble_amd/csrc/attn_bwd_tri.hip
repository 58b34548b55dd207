// Backward of the two-pass attention on the bf16 matrix cores with split fp32 operands (tri_dev.h),
// reading S from the logit map.  Two kernels, each with every product in its natural orientation (no
// transposition of dS, no partial-dQ slabs and no reduction pass as in attn_rows_bwd.hip):
//
//   bwd_dkdv_tri   key-stationary, wave = 32 point keys (lane = key).  Per tile of 32 sampled rows:
//                    dP   = dO_tile V_keys^T          A: dO RM image (LDS), B: V rows (registers)
//                    P    = exp(S - lse), dS = P (dP - delta) scale           [rows on registers]
//                    dV^T += dO_tile^T P              A: dO TR image (LDS), B: P  (accumulator-as-operand)
//                    dK^T += Q_tile^T  dS             A: Q  TR image (LDS), B: dS (accumulator-as-operand)
//   bwd_dq_tri     query-stationary, wave = 32 sampled rows (lane = row).  Per tile of 32 keys (points AND
//                  token keys: the map holds their logits, the images their rows):
//                    dP^T = V_tile dO_rows^T          A: V RM image (LDS), B: dO rows (registers)
//                    dS^T = P^T (dP^T - delta) scale                          [keys on registers]
//                    dQ^T += K_tile^T dS^T            A: K TR image (LDS), B: dS^T (accumulator-as-operand)
// 5 matrix products per (rows x keys) tile instead of 4 -- dP is formed in both orientations -- which
// the 2.6x matrix rate pays for several times over; dQ rows are written once, in place.
// Operand tiles arrive by LDS-DMA; all waits are counted by hand (see attn_tri.hip).
#include <type_traits>

#include "tri_dev.h"

namespace samble {

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// the same with the non-temporal hint (aux bit 1): bytes that are read once (the M-row maps)
__device__ __forceinline__ void glds16_nt(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 2);
}
__device__ __forceinline__ void glds4(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}

// registers 8 ks .. 8 ks + 7 of a 32x32 accumulator-layout tile -> the tri fragment of k-step ks
__device__ __forceinline__ Tri tri_from_acc(const float (&x)[16], int ks) {
  Tri t;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    unsigned hh, mm, ll;
    tri_split2(x[8 * ks + 2 * w], x[8 * ks + 2 * w + 1], hh, mm, ll);
    t.h[w] = hh;
    t.m[w] = mm;
    t.l[w] = ll;
  }
  return t;
}

// out[dt] (channels 32 dt .. 32 dt + 31 x this lane's column) += TRtile^T x frag, both k-steps
template <int DEPTH = 2>
__device__ __forceinline__ void mma_tr_x_acc(const char* __restrict__ tr_tile, int lo, int h, const float (&x)[16],
                                             f32x16 (&out)[4]) {
  const Tri b0 = tri_from_acc(x, 0), b1 = tri_from_acc(x, 1);
  auto fetch = [&](int i) {  // step i: k-step i >> 2, channel block i & 3
    const char* ap = tr_tile + tri_tr_off(32 * (i & 3) + lo, 2 * (i >> 2) + h, 0);
    return Tri{*reinterpret_cast<const u32x4*>(ap), *reinterpret_cast<const u32x4*>(ap + 2048),
               *reinterpret_cast<const u32x4*>(ap + 4096)};
  };
  auto use = [&](int i, const Tri& a) { out[i & 3] = mfma_tri(a, (i >> 2) ? b1 : b0, out[i & 3]); };
  if (DEPTH == 3) {
    tri_pipelined3<8>(fetch, use);
    return;
  }
  tri_pipelined<8>(
      [&](int i) {  // step i: k-step i >> 2, channel block i & 3
        const char* ap = tr_tile + tri_tr_off(32 * (i & 3) + lo, 2 * (i >> 2) + h, 0);
        return Tri{*reinterpret_cast<const u32x4*>(ap), *reinterpret_cast<const u32x4*>(ap + 2048),
                   *reinterpret_cast<const u32x4*>(ap + 4096)};
      },
      [&](int i, const Tri& a) { out[i & 3] = mfma_tri(a, (i >> 2) ? b1 : b0, out[i & 3]); });
}

// the same on a transposed image in its two-fp16-plane form (tri_dev.h, "products that accumulate over tiles"): x is
// multiplied by f (the tile's 2^(13 - (e_t - e_min)) x xs) before its split, three products per k-step
__device__ __forceinline__ Tri duo_from_acc(const float (&x)[16], int ks, float f) {
  Tri t;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    unsigned hh, ll;
    duo_split2(x[8 * ks + 2 * w] * f, x[8 * ks + 2 * w + 1] * f, hh, ll);
    t.h[w] = hh;
    t.m[w] = ll;
  }
  t.l = u32x4{0, 0, 0, 0};
  return t;
}
__device__ __forceinline__ void mma_tr_x_acc_duo(const char* __restrict__ tr_tile, int lo, int h, const float (&x)[16], float f,
                                                 f32x16 (&out)[4]) {
  const Tri b0 = duo_from_acc(x, 0, f), b1 = duo_from_acc(x, 1, f);
  tri_pipelined3<8>(
      [&](int i) {  // step i: k-step i >> 2, channel block i & 3
        const char* ap = tr_tile + tri_tr_off(32 * (i & 3) + lo, 2 * (i >> 2) + h, 0);
        return Tri{*reinterpret_cast<const u32x4*>(ap), *reinterpret_cast<const u32x4*>(ap + 2048), u32x4{0, 0, 0, 0}};
      },
      [&](int i, const Tri& a) {
        const Tri& bb = (i >> 2) ? b1 : b0;
        out[i & 3] = mfma_duo(a.h, a.m, bb.h, bb.m, out[i & 3]);
      });
}

// acc(32x32) = RMtile(rows from LDS) x reg(24 operand registers of this lane's row)^T
__device__ __forceinline__ f32x16 mma_rm_x_regs(const char* __restrict__ rm_tile, int lo, int h, const u32x4 (&q)[24]) {
  const u32x4* lp = reinterpret_cast<const u32x4*>(rm_tile + tri_rm_off(lo, h, 0));
  f32x16 acc = zero16();
  tri_pipelined<8>([&](int ks) { return Tri{lp[192 * ks], lp[192 * ks + 32], lp[192 * ks + 64]}; },
                   [&](int ks, const Tri& a) {
                     const Tri bq = {q[3 * ks], q[3 * ks + 1], q[3 * ks + 2]};
                     acc = mfma_tri(a, bq, acc);
                   });
  return acc;
}

// the same with both operands in their two-fp16-plane form (tri_dev.h): the tile from LDS carries 2^-e in its scale
// slot, the register operand (planes h, l in q[3 ks], q[3 ks + 1], as load_rm_row fetches them from such an image) its
// own tile's 2^-e in q_inv; three products per k-step, the accumulator x 2^-(e_a + e_q) is exact
__device__ __forceinline__ f32x16 mma_rm_x_regs_duo(const char* __restrict__ rm_tile, int lo, int h, const u32x4 (&q)[24],
                                                    float q_inv) {
  const u32x4* lp = reinterpret_cast<const u32x4*>(rm_tile + tri_rm_off(lo, h, 0));
  const float f = q_inv * *reinterpret_cast<const float*>(rm_tile + kDuoScaleSlot);
  f32x16 acc = zero16();
  tri_pipelined<8>([&](int ks) { return Tri{lp[192 * ks], lp[192 * ks + 32], u32x4{0, 0, 0, 0}}; },
                   [&](int ks, const Tri& a) { acc = mfma_duo(a.h, a.m, q[3 * ks], q[3 * ks + 1], acc); });
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] *= f;
  return acc;
}
// 2^-e of the tile that holds `row` of a row image in that form
__device__ __forceinline__ float rm_tile_inv(const char* __restrict__ img, long tiles_per_cloud, int b, int row) {
  return *reinterpret_cast<const float*>(img + ((long)b * tiles_per_cloud + (row >> 5)) * kTriTile + kDuoScaleSlot);
}

__device__ __forceinline__ void load_rm_row(const char* __restrict__ img, long tiles_per_cloud, int b, int row, int h,
                                            u32x4 (&q)[24]) {
  const u32x4* qp = reinterpret_cast<const u32x4*>(img + ((long)b * tiles_per_cloud + (row >> 5)) * kTriTile +
                                                   tri_rm_off(row & 31, h, 0));
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    q[3 * ks] = qp[192 * ks];
    q[3 * ks + 1] = qp[192 * ks + 32];
    q[3 * ks + 2] = qp[192 * ks + 64];
  }
}

// ------------------------------------------------------------------------------------------------
// dQ: one workgroup = 4 waves = 128 sampled rows; key tiles (V RM + K TR images, 48 KB) double-buffered
// ------------------------------------------------------------------------------------------------
constexpr int kDqStage = 2 * kTriTile + 4 * 4096;  // V RM tile, K TR tile, S slots of the 4 waves
constexpr int kDqXt = 36;                          // row stride (floats) of a wave's 32x32 transpose tile
constexpr int kDqLds = 2 * kDqStage + 4 * 32 * kDqXt * 4;

struct DqTriArgs {
  const float* smap;
  int ld;
  const float* lse_s;   // (B, M) of the sampled rows
  const float* delta;   // (B, M)
  const char* dO_rm;    // image of the sampled rows' dO (M rows)
  const char* V_rm;     // image of V (N + nt rows)
  const char* K_tr;     // image of K (N + nt rows)
  const long long* idx;
  int N, NK, M;
  float scale;
  float* dQ;
  long dq_bs, dq_rs;
  float* dsmap;  // optional (B, M, ld): dS of the sampled rows, for the key-stationary kernels
  unsigned* ds_amax;  // optional (B): raised to the bits of the cloud's largest |dS| (what scales dS for two-plane dK products)
};

// PMAP: a.smap is the P map of the SAMPLED rows (B, M, ld) written by attn_rows_rc_tri_kernel -- P is read, not
// re-exponentiated, and the row needs no index indirection
template <int ABL, bool PMAP>  // ABL: timing-only ablations (wrong results): 1 = tiles staged once, 2 = no matrix products
__global__ __launch_bounds__(256) void bwd_dq_tri_kernel(const DqTriArgs a) {
  float ds_max = 0.f;  // this lane's largest |dS| (a.ds_amax)
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  constexpr int NW = 4;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int M = a.M, N = a.N;
  const int mrow = chunk * (32 * NW) + wave * 32 + lo;
  const bool mvalid = mrow < M;
  const int mc = mvalid ? mrow : M - 1;
  const long row = a.idx[(long)b * M + mc];
  const float my_lse = a.lse_s[(long)b * M + mc], my_delta = a.delta[(long)b * M + mc];
  const float* srow = a.smap + (PMAP ? (long)b * M + mc : (long)b * N + row) * a.ld + 4 * h;
  const int ntiles = (a.NK + kTile - 1) / kTile, mtiles = (M + kTile - 1) / kTile;
  const char* Vb = a.V_rm + (long)b * ntiles * kTriTile;
  const char* Kb = a.K_tr + (long)b * ntiles * kTriTile;

  const float* prow8[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = min(chunk * (32 * NW) + wave * 32 + 8 * k + (lane >> 3), M - 1);
    prow8[k] = a.smap + ((long)b * M + r) * a.ld + 4 * ((lane & 7) ^ (lane >> 3));
  }
  auto stage = [&](int t) {  // 16 DMA pieces per thread
    const int tt = min(t, ntiles - 1);
    char* st = smem_c + (t & 1) * kDqStage;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      glds16(Vb + (long)tt * kTriTile + (tid + 256 * k) * 16, st + (wave * 64 + 256 * k) * 16);
      glds16(Kb + (long)tt * kTriTile + (tid + 256 * k) * 16, st + kTriTile + (wave * 64 + 256 * k) * 16);
    }
    if (PMAP) {
      // the wave's 32 x 32 block of the P map as whole 128-byte lines: piece k = rows 8k .. 8k+7, lane (r8 = lane >> 3,
      // c = lane & 7) fetches 16-byte block c ^ r8 of its row (swizzle on the source: the rows' blocks land on
      // distinct banks for the row-per-lane reads below)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        glds16(prow8[k] + tt * kTile, st + 2 * kTriTile + wave * 4096 + k * 1024);
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) glds16(srow + tt * kTile + 8 * g, st + 2 * kTriTile + wave * 4096 + g * 1024);
    }
  };
  stage(0);
  u32x4 go[24];
  load_rm_row(a.dO_rm, mtiles, b, mc, h, go);
  const float do_inv = rm_tile_inv(a.dO_rm, mtiles, b, mc);  // (wave-uniform: the wave's 32 rows are one tile)
  f32x16 oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) oacc[dt] = zero16();
  const float scale = a.scale;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  for (int t = 0; t < ntiles; ++t) {
    const char* st = smem_c + (t & 1) * kDqStage;
    if (!(ABL & 1)) stage(t + 1);  // into the other stage, whose reads ended before the last barrier
    f32x16 dp;
    if (ABL & 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = __uint_as_float(go[r][0]);
    } else dp = mma_rm_x_regs_duo(st, lo, h, go, do_inv);  // dP^T: rows = keys crow(r, h), column = this lane's row
    const char* sw = st + 2 * kTriTile + wave * 4096;
    const f32x4* sp = reinterpret_cast<const f32x4*>(sw + lane * 16);
    float ds[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v4 = PMAP ? *reinterpret_cast<const f32x4*>(sw + (lo >> 3) * 1024 + (lo & 7) * 128 +
                                                              (((2 * g + h) ^ (lo & 7)) << 4))
                            : sp[64 * g];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        // columns past N + nt hold -inf (P map: 0): P = 0
        ds[r] = (PMAP ? v4[e] : __expf(v4[e] - my_lse)) * (dp[r] - my_delta) * scale;
      }
    }
    if (a.ds_amax) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) ds_max = __builtin_fmaxf(__builtin_fmaxf(ds_max, fabsf(ds[r])), fabsf(ds[r + 1]));
    }
    if (a.dsmap) {  // dS tile -> map rows as full 128-byte lines (8 lanes per row), through the wave's LDS tile
      float* xt = reinterpret_cast<float*>(smem_c + 2 * kDqStage) + wave * (32 * kDqXt);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 o = {ds[4 * g], ds[4 * g + 1], ds[4 * g + 2], ds[4 * g + 3]};
        *reinterpret_cast<f32x4*>(xt + lo * kDqXt + 8 * g + 4 * h) = o;
      }
      const int m0 = chunk * (32 * NW) + wave * 32;
#pragma unroll
      for (int k8 = 0; k8 < 4; ++k8) {
        const int rr = (lane >> 3) + 8 * k8;
        const f32x4 o = *reinterpret_cast<const f32x4*>(xt + rr * kDqXt + 4 * (lane & 7));
        const int mr = min(m0 + rr, M - 1);  // rows past M-1 rewrite row M-1's values of this wave (same bytes)
        *reinterpret_cast<f32x4*>(a.dsmap + ((long)b * M + mr) * a.ld + t * kTile + 4 * (lane & 7)) = o;
      }
    }
    if (ABL & 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r & 3][r] += ds[r];
    } else mma_tr_x_acc(st + kTriTile, lo, h, ds, oacc);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (a.ds_amax) {  // (non-negative floats order as their bits)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ds_max = fmaxf(ds_max, __shfl_xor(ds_max, o, 64));
    if (lane == 0) atomicMax(a.ds_amax + b, __float_as_uint(ds_max));
  }
  if (mvalid) {
    float* orow = a.dQ + (long)b * a.dq_bs + row * a.dq_rs + 4 * h;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 o = {oacc[dt][4 * g], oacc[dt][4 * g + 1], oacc[dt][4 * g + 2], oacc[dt][4 * g + 3]};
        *reinterpret_cast<f32x4*>(orow + 32 * dt + 8 * g) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dQ from the P map, woven (the structure of attn_rows_rc_tri_kernel, attn_tri.hip): this kernel runs one wave per
// SIMD, so an iteration issues three INDEPENDENT streams k-step by k-step --
//     dP^T of tile t+1 (48 MFMAs)  |  dQ^T += K^T dS^T of tile t-1 (48 MFMAs)  |  dS of tile t = P (dP - delta) scale
//     and its three-plane split (vector work), + the dS tile of t-1 on its way to the dS map
// instead of running dP -> dS -> dQ of one tile back to back behind a drained queue.  V row tiles, K transposed
// tiles and the waves' P blocks through 2-deep LDS rings (everything staged in iteration t is waited for at its
// end: the 16 pieces go out in the first k-steps), every wait counted.  Same products in the same order as
// bwd_dq_tri_kernel<.., true>: bit-identical dQ and dS map.
// ------------------------------------------------------------------------------------------------
constexpr int kDqpLds = 4 * kTriTile + 2 * 4 * 4096 + 4 * 4096;  // V ring, K ring, P slots, transpose tiles: 144 KB

__global__ __launch_bounds__(256) void bwd_dq_pm_tri_kernel(const DqTriArgs a) {
  float ds_max = 0.f;  // this lane's largest |dS| (a.ds_amax)
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  constexpr int NW = 4;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int M = a.M;
  const int m0 = chunk * (32 * NW) + wave * 32;
  const int mrow = m0 + lo;
  const bool mvalid = mrow < M;
  const int mc = mvalid ? mrow : M - 1;
  const long row = a.idx[(long)b * M + mc];
  const float my_delta = a.delta[(long)b * M + mc];
  const float scale = a.scale;
  const int ntiles = (a.NK + kTile - 1) / kTile, mtiles = (M + kTile - 1) / kTile;
  const char* Vb = a.V_rm + (long)b * ntiles * kTriTile;
  const char* Kb = a.K_tr + (long)b * ntiles * kTriTile;
  char* vring = smem_c;
  char* kring = smem_c + 2 * kTriTile;
  char* pslots = smem_c + 4 * kTriTile + wave * 4096;  // + (t & 1) * 16384
  char* xt = smem_c + 4 * kTriTile + 2 * 16384 + wave * 4096;
  // the wave's 32 x 32 block of the P map as whole 128-byte lines: piece k = rows 8k .. 8k+7, lane (r8 = lane >> 3,
  // c = lane & 7) fetches 16-byte block c ^ r8 of its row (swizzle on the source)
  const float* prow8[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = min(m0 + 8 * k + (lane >> 3), M - 1);
    prow8[k] = a.smap + (map_cloud(b, 0) * M + r) * a.ld + 4 * ((lane & 7) ^ (lane >> 3));
  }
  auto stage_tile = [&](const char* img, char* ring, int t) {  // 6 pieces per thread
    const char* src = img + (long)min(t, ntiles - 1) * kTriTile;
    char* dst = ring + (t & 1) * kTriTile;
#pragma unroll
    for (int k = 0; k < 6; ++k) glds16(src + (tid + 256 * k) * 16, dst + (wave * 64 + 256 * k) * 16);
  };
  auto stage_p = [&](int t) {  // 4 pieces per thread
    const int tt = min(t, ntiles - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (SAMBLE_MAP_NT & 2) glds16_nt(prow8[k] + tt * kTile, pslots + (t & 1) * 16384 + k * 1024);
      else glds16(prow8[k] + tt * kTile, pslots + (t & 1) * 16384 + k * 1024);
    }
  };
  stage_tile(Vb, vring, 0);
  stage_tile(Vb, vring, 1);
  stage_p(0);
  u32x4 go[24];
  load_rm_row(a.dO_rm, mtiles, b, mc, h, go);
  const float do_inv = rm_tile_inv(a.dO_rm, mtiles, b, mc);  // (wave-uniform: the wave's 32 rows are one tile)
  f32x16 oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) oacc[dt] = zero16();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  f32x16 dp_cur = mma_rm_x_regs_duo(vring, lo, h, go, do_inv), dp_nxt;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // V slot 0 is restaged by iteration 0
  Tri bp[2];  // dS^T fragments of the tile whose dQ product is due (tile t-1): none yet
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) bp[ks] = Tri{u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
  float dsprev[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) dsprev[r] = 0.f;

  // iteration t = 0 .. ntiles: dP of tile t+1, dS of tile t, dQ product and dS-map rows of tile t-1
  auto step = [&](int t, auto last_c, auto first_c) {
    constexpr bool LAST = decltype(last_c)::value, FIRST = decltype(first_c)::value;
    // operand reads first
    const u32x4* lp = reinterpret_cast<const u32x4*>(vring + ((t + 1) & 1) * kTriTile + tri_rm_off(lo, h, 0));
    // dP of tile t+1: V tile and dO row as two fp16 planes each; 2^-(e_dO + e_V) comes out of the finished sum
    const float vf = do_inv * *reinterpret_cast<const float*>(vring + ((t + 1) & 1) * kTriTile + kDuoScaleSlot);
    // FIRST (t = 0): no dS yet, K tile 0 is only arriving: the dQ products are left out (any stand-in operand would have
    // to be finite in all three pieces, and the third piece slots of a two-plane image are unwritten memory)
    const char* kt = FIRST ? vring + kTriTile : kring + ((t - 1) & 1) * kTriTile;
    auto fetch_v = [&](int ks) { return Tri{lp[192 * ks], lp[192 * ks + 32], u32x4{0, 0, 0, 0}}; };
    auto fetch_k = [&](int i) {  // step i: k-step i >> 2, channel block i & 3
      const char* ap = kt + tri_tr_off(32 * (i & 3) + lo, 2 * (i >> 2) + h, 0);
      return Tri{*reinterpret_cast<const u32x4*>(ap), *reinterpret_cast<const u32x4*>(ap + 2048),
                 *reinterpret_cast<const u32x4*>(ap + 4096)};
    };
    Tri v0 = fetch_v(0), v1 = fetch_v(1), k0 = fetch_k(0), k1 = fetch_k(1);
    f32x4 p4[4];  // this lane's 16 P values of tile t (row lo, columns 8g + 4h .. + 3)
    {
      const char* sw = pslots + (t & 1) * 16384;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        p4[g] = *reinterpret_cast<const f32x4*>(sw + (lo >> 3) * 1024 + (lo & 7) * 128 + (((2 * g + h) ^ (lo & 7)) << 4));
    }
    float* dsout = a.dsmap + map_cloud(b, 1) * M * a.ld + max(t - 1, 0) * kTile + 4 * (lane & 7);
    f32x4 po[4];
    Tri bn[2];
    float ds[16];
    dp_nxt = zero16();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      Tri v2 = v1, k2 = k1;
      if (i + 2 < 8) {
        v2 = fetch_v(i + 2);
        k2 = fetch_k(i + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!LAST) dp_nxt = mfma_duo(v0.h, v0.m, go[3 * i], go[3 * i + 1], dp_nxt);
      if (!FIRST) oacc[i & 3] = mfma_tri(k0, bp[i >> 2], oacc[i & 3]);
      {  // slice i of the vector work on tile t: elements 2 i, 2 i + 1
#pragma clang fp contract(off)  // dS is what the map holds: the split must start from the ROUNDED product, not fuse into it
        const int r0 = 2 * i, r1 = 2 * i + 1;
        float x0 = dp_cur[r0], x1 = dp_cur[r1];
        asm volatile("" : "+v"(x0), "+v"(x1));  // pins the slice inside this k-step's scheduling region
        // columns past N + nt hold P = 0
        ds[r0] = LAST ? 0.f : p4[r0 >> 2][r0 & 3] * (x0 - my_delta) * scale;
        ds[r1] = LAST ? 0.f : p4[r1 >> 2][r1 & 3] * (x1 - my_delta) * scale;
        ds_max = __builtin_fmaxf(__builtin_fmaxf(ds_max, fabsf(ds[r0])), fabsf(ds[r1]));  // (v_max3 with |.| modifiers)
        unsigned hh, mm, ll;
        tri_split2(ds[r0], ds[r1], hh, mm, ll);
        asm volatile("" : "+v"(hh), "+v"(mm), "+v"(ll));
        bn[i >> 2].h[i & 3] = hh;
        bn[i >> 2].m[i & 3] = mm;
        bn[i >> 2].l[i & 3] = ll;
      }
      if (!LAST) {  // this iteration's 16 DMA pieces in the first k-steps: they are waited for at its end
        if (i == 0) stage_tile(Vb, vring, t + 2);   // slot of V tile t: read in iteration t-1
        if (i == 1) stage_tile(Kb, kring, t);       // slot of K tile t-2: read in iteration t-1
        if (i == 2) stage_p(t + 1);                 // slot of P block t-1: read at the top of iteration t-1
      }
      // the previous tile's dS -> map rows as full 128-byte lines through the wave's own 4 KB of LDS (XOR-swizzled
      // 16-byte blocks; LDS operations of a wave execute in order).  Iteration 0 writes zeros over tile 0's place.
      if (i == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 o = {dsprev[4 * g], dsprev[4 * g + 1], dsprev[4 * g + 2], dsprev[4 * g + 3]};
          *reinterpret_cast<f32x4*>(xt + lo * 128 + (((2 * g + h) ^ (lo & 7)) << 4)) = o;
        }
      }
      if (i == 3) {
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
          const int rr = (lane >> 3) + 8 * k8;
          po[k8] = *reinterpret_cast<const f32x4*>(xt + rr * 128 + (((lane & 7) ^ (rr & 7)) << 4));
        }
      }
      if (i >= 6) {
#pragma unroll
        for (int k8 = 2 * (i - 6); k8 < 2 * (i - 6) + 2; ++k8) {
          const int mr = min(m0 + (lane >> 3) + 8 * k8, M - 1);  // rows past M-1 rewrite row M-1's values (same bytes)
          if (SAMBLE_MAP_NT & 4) __builtin_nontemporal_store(po[k8], reinterpret_cast<f32x4*>(dsout + (long)mr * a.ld));
          else *reinterpret_cast<f32x4*>(dsout + (long)mr * a.ld) = po[k8];
        }
      }
#pragma unroll
      for (int m = 0; m < (LAST ? 6 : FIRST ? 3 : 9); ++m) {  // 3 dP + 6 dQ products per k-step
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, LAST ? 4 : FIRST ? 8 : 3, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      v0 = v1;
      v1 = v2;
      k0 = k1;
      k1 = k2;
    }
    // the 16 pieces of this iteration must have landed; younger than them: its 4 stores
    if (!LAST) asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int r = 0; r < 16; ++r) dp_cur[r] = dp_nxt[r] * vf;
    bp[0] = bn[0];
    bp[1] = bn[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) dsprev[r] = ds[r];
  };
  step(0, std::false_type{}, std::true_type{});
  for (int t = 1; t < ntiles; ++t) step(t, std::false_type{}, std::false_type{});
  step(ntiles, std::true_type{}, std::false_type{});
  if (a.ds_amax) {  // (non-negative floats order as their bits)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ds_max = fmaxf(ds_max, __shfl_xor(ds_max, o, 64));
    if (lane == 0) atomicMax(a.ds_amax + b, __float_as_uint(ds_max));
  }
  if (mvalid) {
    float* orow = a.dQ + (long)b * a.dq_bs + row * a.dq_rs + 4 * h;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 o = {oacc[dt][4 * g], oacc[dt][4 * g + 1], oacc[dt][4 * g + 2], oacc[dt][4 * g + 3]};
        *reinterpret_cast<f32x4*>(orow + 32 * dt + 8 * g) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dK, dV (and the column sums of dS): one workgroup = 4 waves = 128 point keys; tiles of 32 sampled rows
// (dO RM, dO TR, Q TR images, 72 KB) double-buffered, their lse / delta / row ids in three small slots
// ------------------------------------------------------------------------------------------------
constexpr int kKvStage = 3 * kTriTile;
constexpr int kKvMeta = 512;  // per slot: lse[32], delta[32], idx[32] (int64)
constexpr int kKvLds = 2 * kKvStage + 3 * kKvMeta;

struct KvTriArgs {
  const float* smap;
  int ld;
  const float* lse_s;
  const float* delta;
  const char* dO_rm;
  const char* dO_tr;
  const char* Q_tr;   // images of the sampled rows (M rows, zero padded to whole tiles)
  const char* V_rm;   // image of V (N + nt rows)
  const long long* idx;
  int N, NK, M;
  float scale;
  float* dK;
  long dk_bs, dk_rs;
  float* dV;
  long dv_bs, dv_rs;
  float* cs;  // optional (B, N + nt)
};

template <bool CS>
__global__ __launch_bounds__(256) void bwd_dkdv_tri_kernel(const KvTriArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int N = a.N, M = a.M, ld = a.ld;
  const int j = chunk * 128 + wave * 32 + lo;
  const bool jvalid = j < N;
  const int jc = min(j, N - 1);
  const int ktiles = (a.NK + kTile - 1) / kTile, mtiles = (M + kTile - 1) / kTile;
  const char* Gr = a.dO_rm + (long)b * mtiles * kTriTile;
  const char* Gt = a.dO_tr + (long)b * mtiles * kTriTile;
  const char* Qt = a.Q_tr + (long)b * mtiles * kTriTile;
  char* meta = smem_c + 2 * kKvStage;
  const float* scol = a.smap + (long)b * N * ld + jc;

  auto stage_tiles = [&](int t) {  // 18 DMA pieces per thread
    const int tt = min(t, mtiles - 1);
    char* st = smem_c + (t & 1) * kKvStage;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int off = (tid + 256 * k) * 16, loff = (wave * 64 + 256 * k) * 16;
      glds16(Gr + (long)tt * kTriTile + off, st + loff);
      glds16(Gt + (long)tt * kTriTile + off, st + kTriTile + loff);
      glds16(Qt + (long)tt * kTriTile + off, st + 2 * kTriTile + loff);
    }
  };
  auto stage_meta = [&](int t) {  // 2 pieces per thread; the four waves write the same bytes
    const int i0 = min(t, mtiles - 1) * 32;
    char* ms = meta + (t % 3) * kKvMeta;
    const int ii = min(i0 + lo, M - 1);
    glds4((h ? a.delta : a.lse_s) + (long)b * M + ii, ms);                        // lane -> float lane
    glds4(reinterpret_cast<const int*>(a.idx + (long)b * M + min(i0 + (lane >> 1), M - 1)) + (lane & 1), ms + 256);
  };
  stage_tiles(0);
  stage_meta(0);
  stage_meta(1);
  u32x4 vr[24];
  load_rm_row(a.V_rm, ktiles, b, jc, h, vr);
  const float v_inv = rm_tile_inv(a.V_rm, ktiles, b, jc);  // (wave-uniform: the wave's 32 keys are one tile)
  f32x16 dv[4], dk[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dv[dt] = dk[dt] = zero16();
  float csum = 0.f;
  const float scale = a.scale;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // this lane's 16 rows crow(r, h) = 8g + 4h + e of a [32]-float array: four 16-byte LDS reads
  auto rows16 = [&](const float* base, float (&dst)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(base + 8 * g + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[4 * g + e] = v4[e];
    }
  };
  auto load_s = [&](int t, float (&dst)[16]) {  // S[sampled row][this lane's key] of tile t (its meta has landed)
    const long long* sel = reinterpret_cast<const long long*>(meta + (t % 3) * kKvMeta + 256);
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[r] = scol[sel[crow(r, h)] * ld];
  };
  // One block of 48 MFMAs: out[dt] += TRtile^T x frag (8 steps of six MFMAs, operands fetched two steps
  // ahead), with `fill(i)` -- about 24 vector instructions of OTHER work -- placed in the MFMAs' shadow
  // step by step: this wave is alone on its SIMD, nothing else would run there.
  auto block = [&](const char* tr_tile, const Tri (&f)[2], f32x16 (&out)[4], auto fill) {
    tri_pipelined<8>(
        [&](int i) {
          const char* ap = tr_tile + tri_tr_off(32 * (i & 3) + lo, 2 * (i >> 2) + h, 0);
          return Tri{*reinterpret_cast<const u32x4*>(ap), *reinterpret_cast<const u32x4*>(ap + 2048),
                     *reinterpret_cast<const u32x4*>(ap + 4096)};
        },
        [&](int i, const Tri& aa) {
          out[i & 3] = mfma_tri(aa, f[i >> 2], out[i & 3]);
          fill(i);
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          }
        });
  };
  float p[16], lv[16], dl[16];
  Tri pf[2];
  {  // P of tile 0 and its planes
    float s0[16];
    load_s(0, s0);
    rows16(reinterpret_cast<const float*>(meta), lv);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __expf(s0[r] - lv[r]);
      p[r] = (crow(r, h) < M) ? pv : 0.f;
    }
    pf[0] = tri_from_acc(p, 0);
    pf[1] = tri_from_acc(p, 1);
  }

  // tile t:  dP (48 MFMAs)  |  dV += dO^T P (48) under which dS and its planes are formed  |
  //          dK += Q^T dS (48) under which the NEXT tile's P and its planes are formed (logits fetched a tile ahead)
  for (int t = 0; t < mtiles; ++t) {
    const char* st = smem_c + (t & 1) * kKvStage;
    const float* Lt = reinterpret_cast<const float*>(meta + (t % 3) * kKvMeta);
    const float* Ln = reinterpret_cast<const float*>(meta + ((t + 1) % 3) * kKvMeta);
    float sn[16];
    load_s(t + 1, sn);  // rows past M-1 are clamped in the meta slot; their P is masked below
    stage_tiles(t + 1);
    stage_meta(t + 2);
    rows16(Lt + 32, dl);
    rows16(Ln, lv);
    const f32x16 dp = mma_rm_x_regs_duo(st, lo, h, vr, v_inv);  // dP: rows = sampled rows crow(r, h), column = this lane's key
    Tri df[2];
    float ds[16];
    block(st + kTriTile, pf, dv, [&](int i) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int r = 2 * i + e;
        ds[r] = jvalid ? p[r] * (dp[r] - dl[r]) * scale : 0.f;
        if (CS) csum += ds[r];
      }
      unsigned hh, mm, ll;
      tri_split2(ds[2 * i], ds[2 * i + 1], hh, mm, ll);
      df[i >> 2].h[i & 3] = hh;
      df[i >> 2].m[i & 3] = mm;
      df[i >> 2].l[i & 3] = ll;
    });
    const int i1 = (t + 1) * kTile;
    block(st + 2 * kTriTile, df, dk, [&](int i) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int r = 2 * i + e;
        const float pv = __expf(sn[r] - lv[r]);
        p[r] = (i1 + crow(r, h) < M) ? pv : 0.f;
      }
      unsigned hh, mm, ll;
      tri_split2(p[2 * i], p[2 * i + 1], hh, mm, ll);
      pf[i >> 2].h[i & 3] = hh;
      pf[i >> 2].m[i & 3] = mm;
      pf[i >> 2].l[i & 3] = ll;
    });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (CS) {
    const float ctot = csum + wave_xor32(csum);
    if (jvalid && h == 0) a.cs[(long)b * a.NK + j] = ctot;
  }
  if (jvalid) {
    float* vrow = a.dV + (long)b * a.dv_bs + (long)j * a.dv_rs + 4 * h;
    float* krow = a.dK + (long)b * a.dk_bs + (long)j * a.dk_rs + 4 * h;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 ov = {dv[dt][4 * g], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]};
        const f32x4 ok = {dk[dt][4 * g], dk[dt][4 * g + 1], dk[dt][4 * g + 2], dk[dt][4 * g + 3]};
        *reinterpret_cast<f32x4*>(vrow + 32 * dt + 8 * g) = ov;
        *reinterpret_cast<f32x4*>(krow + 32 * dt + 8 * g) = ok;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Key-stationary accumulation with dS taken from the map bwd_dq_tri wrote (4 products per tile in all,
// nothing formed twice):   MODE 0: dV^T += dO^T P,  P = exp(S - lse) from the logit map
//                          MODE 1: dK^T += Q^T dS,  dS read back (and its column sums for l2 scoring)
// One workgroup = 8 waves = 256 point keys, two waves per SIMD; per tile of 32 sampled rows one
// transposed image tile (24 KB, ring of 4 by LDS-DMA, three tiles ahead) and 16 map values per lane,
// fetched one tile ahead.  48 MFMAs per tile and wave, 64 accumulator registers.
// ------------------------------------------------------------------------------------------------
constexpr int kAccMapSlots = 3, kAccTrSlots = 2, kAccMetaSlots = 4;
constexpr int kAccMap = 32 * 256 * 4;  // 32 sampled rows x the workgroup's 256 keys, fp32
constexpr int kAccMeta = 512;          // per slot: lse[32] twice (MODE 0), idx[32] (int64) at +256
constexpr int kAccLds = kAccTrSlots * kTriTile + kAccMapSlots * kAccMap + kAccMetaSlots * kAccMeta;

#ifdef SAMBLE_STAMPS  // scratch builds only: s_memtime marks of workgroup 0, tiles 10 and 11, every wave
__device__ unsigned long long g_ka_stamps[8 * 2 * 8];
#define KA_STAMP(i)                                                                                        \
  do {                                                                                                     \
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (t == 10 || t == 11))                           \
      g_ka_stamps[(wave * 2 + (t - 10)) * 8 + (i)] = __builtin_amdgcn_s_memtime();                         \
  } while (0)
#else
#define KA_STAMP(i) do { } while (0)
#endif
struct KaccArgs {
  const float* map;    // MODE 0: smap (B, N, ld);  MODE 1: dsmap (B, M, ld)
  int ld;
  const float* lse_s;  // MODE 0
  const char* tr;      // transposed image of the sampled rows: dO (MODE 0) / Q (MODE 1)
  const long long* idx;
  int N, NK, M;
  float* out;          // dV / dK rows
  long o_bs, o_rs;
  float* cs;           // MODE 1, optional (B, N + nt)
  int duo;             // the transposed image is in its two-fp16-plane form (bwd_prep_tri with ds_amax)
  const unsigned* x_amax;  // duo, map = dS: (B) bits of the cloud's largest |dS|; null: the map values are <= 1 (P)
};
// scales of the two-plane accumulation (tri_dev.h): the factor of tile t's map values is ts.x_scale(t) * xs, the finished
// sum comes out x out_un
struct KaccDuo {
  DuoTileScales ts;
  float xs, out_un;
  __device__ __forceinline__ void init(const KaccArgs& a, const char* Tb, int mtiles, int b, int lane) {
    ts.load(Tb, mtiles, lane);
    xs = 1.f;
    if (a.x_amax) {
      float s2, inv2;
      duo_scale_for(__uint_as_float(a.x_amax[b]), s2, inv2);  // |dS| s2 < 2^13
      xs = s2 * (1.f / 8192.f);
    }
    out_un = ts.unscale() / xs;
  }
  __device__ __forceinline__ float factor(int t) const { return ts.x_scale(t) * xs; }
};

// The map block of a tile (32 rows x 256 keys = 32 pieces of 1 KB, each one contiguous kilobyte of a map
// row) comes by LDS-DMA two tiles ahead, the image tile one tile ahead (it is shared by the cloud's
// workgroups: L2).  Every operation of the loop is a DMA piece, all waits are counted: per iteration a
// thread issues 3 image pieces, (MODE 0) 2 meta pieces, then 4 map pieces -- `vmcnt(4)` at the end
// leaves exactly the next-but-one map block in flight.
template <int MODE, bool CS>
__global__ __launch_bounds__(512, 2) void bwd_kacc_tri_kernel(const KaccArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int N = a.N, M = a.M, ld = a.ld;
  const int j = chunk * 256 + wave * 32 + lo;
  const bool jvalid = j < N;
  const int mtiles = (M + kTile - 1) / kTile;
  const char* Tb = a.tr + (long)b * mtiles * kTriTile;
  char* mapring = smem_c + kAccTrSlots * kTriTile;
  char* meta = mapring + kAccMapSlots * kAccMap;
  // this lane's 16 bytes of a 1 KB row piece; past the row's end the piece is pulled back inside it (those
  // keys are >= N: masked below)
  const float* mapb = a.map + (long)b * (MODE == 0 ? N : M) * ld + min(chunk * 256 + lane * 4, ld - 4);

  auto stage_tr = [&](int t) {  // 3 pieces per thread
    const int tt = min(t, mtiles - 1);
    char* st = smem_c + (t % kAccTrSlots) * kTriTile;
#pragma unroll
    for (int k = 0; k < 3; ++k) glds16(Tb + (long)tt * kTriTile + (tid + 512 * k) * 16, st + (wave * 64 + 512 * k) * 16);
  };
  auto stage_meta = [&](int t) {  // MODE 0 only: 2 pieces per thread, the eight waves write the same bytes
    const int i0 = min(t, mtiles - 1) * 32;
    char* ms = meta + (t % kAccMetaSlots) * kAccMeta;
    glds4(a.lse_s + (long)b * M + min(i0 + lo, M - 1), ms);  // lanes 32..63 repeat lanes 0..31
    glds4(reinterpret_cast<const int*>(a.idx + (long)b * M + min(i0 + (lane >> 1), M - 1)) + (lane & 1), ms + 256);
  };
  auto stage_map = [&](int t) {  // 4 pieces per thread: wave w brings rows 4w .. 4w+3 of the tile
    const int tt = min(t, mtiles - 1);
    char* ms = mapring + (t % kAccMapSlots) * kAccMap;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 4 * wave + q;
      long row;
      if (MODE == 0) row = reinterpret_cast<const long long*>(meta + (tt % kAccMetaSlots) * kAccMeta + 256)[i];
      else row = min(tt * 32 + i, M - 1);
      glds16(mapb + row * ld, ms + i * 1024);
    }
  };
  if (MODE == 0) {
    stage_meta(0);
    stage_meta(1);
    stage_meta(2);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  stage_tr(0);
  stage_map(0);
  stage_map(1);
  f32x16 acc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) acc[dt] = zero16();
  float csum = 0.f;
  KaccDuo kd;
  if (a.duo) kd.init(a, Tb, mtiles, b, lane);  // (uniform)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // Vector instructions share the SIMD's issue port with the MFMAs (48 per tile here), so the loop carries as few
  // as it can: keys past N need no masking (their outputs are never stored), sampled rows past M exist only in the
  // last tile (TAIL), and the 16 map reads use immediate offsets off one address.
  auto step = [&](int t, auto tail_c) {
    constexpr bool TAIL = decltype(tail_c)::value;
    KA_STAMP(0);
    const char* st = smem_c + (t % kAccTrSlots) * kTriTile;
    stage_tr(t + 1);                     // slot of tile t-1
    if (MODE == 0) stage_meta(t + 3);    // slot of tile t-1; needed by stage_map(t + 3) in the next iteration
    stage_map(t + 2);                    // slot of tile t-1; the 4 youngest operations
    KA_STAMP(1);
    const float* mp = reinterpret_cast<const float*>(mapring + (t % kAccMapSlots) * kAccMap) + wave * 32 + lo + 1024 * h;
    const int i0 = t * kTile;
    float x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = mp[(8 * (r >> 2) + (r & 3)) * 256];  // row crow(r, h) = 8 (r >> 2) + 4 h + (r & 3)
    if (MODE == 0) {
      const float* Lt = reinterpret_cast<const float*>(meta + (t % kAccMetaSlots) * kAccMeta);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(Lt + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float pv = __expf(x[r] - l4[e]);
          x[r] = (TAIL && i0 + crow(r, h) >= M) ? 0.f : pv;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (TAIL) x[r] = (i0 + crow(r, h) < M) ? x[r] : 0.f;  // rows past M: the clamped copies of row M-1
        if (CS) csum += x[r];
      }
    }
    KA_STAMP(2);
    if (a.duo) mma_tr_x_acc_duo(st, lo, h, x, kd.factor(t), acc);
    else mma_tr_x_acc<3>(st, lo, h, x, acc);
    KA_STAMP(3);
#ifdef SAMBLE_STAMPS
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    KA_STAMP(4);
    asm volatile("s_barrier" ::: "memory");
    KA_STAMP(5);
#else
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
  };
  for (int t = 0; t < mtiles - 1; ++t) step(t, std::false_type{});
  step(mtiles - 1, std::true_type{});
  if (MODE == 1 && CS) {
    const float ctot = csum + wave_xor32(csum);
    if (jvalid && h == 0) a.cs[(long)b * a.NK + j] = ctot;
  }
  if (jvalid) {
    const float un = a.duo ? kd.out_un : 1.f;  // (x 1 is exact: the three-plane results are unchanged)
    float* orow = a.out + (long)b * a.o_bs + (long)j * a.o_rs + 4 * h;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 o = {acc[dt][4 * g] * un, acc[dt][4 * g + 1] * un, acc[dt][4 * g + 2] * un, acc[dt][4 * g + 3] * un};
        *reinterpret_cast<f32x4*>(orow + 32 * dt + 8 * g) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same accumulation from an M-row map (P or dS: MODE 1 above), woven.  In the kernel above the two waves of a SIMD
// run in step -- both split their map values (vector work), then both issue their 48 MFMAs -- so the matrix pipe
// idles through the splits and the vector ALU through the products (SQ_VALU_MFMA_COEXEC_CYCLES: 3.6 % of the busy
// cycles; per tile 2 x 1000 + 2 x 1536 cycles, one after the other).  Here every wave brings ITS OWN 32 keys x 32 rows
// of the map block (four pieces of eight 128-byte row segments: nobody else reads them, so no barrier stands between
// their arrival and their use) and splits tile t+1's values k-step by k-step behind the MFMAs of tile t.
// Same products in the same order: bit-identical dV / dK.
// ------------------------------------------------------------------------------------------------
constexpr int kAccPmMapSlots = 2, kAccPmTrSlots = 3;
constexpr int kAccPmLds = kAccPmTrSlots * kTriTile + kAccPmMapSlots * kAccMap;

template <bool CS, bool DUO>  // DUO: the transposed image in its two-fp16-plane form, three products per k-step (a.duo)
__global__ __launch_bounds__(512, 2) void bwd_kacc_pm_tri_kernel(const KaccArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int N = a.N, M = a.M, ld = a.ld;
  const int j = chunk * 256 + wave * 32 + lo;
  const bool jvalid = j < N;
  const int mtiles = (M + kTile - 1) / kTile;
  const char* Tb = a.tr + (long)b * mtiles * kTriTile;
  char* mapring = smem_c + kAccPmTrSlots * kTriTile;
  // lane l of a piece: row l >> 3 of its eight rows, 16-byte chunk l & 7 of the wave's 128 bytes (past the row's end
  // the chunk is pulled back inside it: those keys are >= N, their outputs are never stored)
  const float* mapb = a.map + map_cloud(b) * M * ld + min(chunk * 256 + wave * 32 + 4 * (lane & 7), ld - 4);
  const int prow = lane >> 3;

  // one piece of the image tile t (k = 0..2) / of the wave's map block of tile t (q = 0..3: rows 8 q .. 8 q + 7)
  auto tr_piece = [&](int t, int k) {
    const int tt = min(t, mtiles - 1);
    char* st = smem_c + (t % kAccPmTrSlots) * kTriTile;
    glds16(Tb + (long)tt * kTriTile + (tid + 512 * k) * 16, st + (wave * 64 + 512 * k) * 16);
  };
  auto map_piece = [&](int t, int q) {
    const int tt = min(t, mtiles - 1);
    char* ms = mapring + (t % kAccPmMapSlots) * kAccMap + wave * 4096;
    if (SAMBLE_MAP_NT & 8) glds16_nt(mapb + (long)min(tt * 32 + 8 * q + prow, M - 1) * ld, ms + q * 1024);
    else glds16(mapb + (long)min(tt * 32 + 8 * q + prow, M - 1) * ld, ms + q * 1024);
  };
  float csum = 0.f;
  // the wave's block of tile t: [32 rows][32 keys]; this lane's 16 values in accumulator order (row crow(r, h), key lo)
  auto read_x = [&](int t, float (&x)[16]) {
    const float* xp = reinterpret_cast<const float*>(mapring + (t % kAccPmMapSlots) * kAccMap + wave * 4096) + lo + 128 * h;
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = xp[(8 * (r >> 2) + (r & 3)) * 32];
  };
  auto mask_tail = [&](int t, float (&x)[16]) {  // rows past M: the clamped copies of row M-1
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = (t * kTile + crow(r, h) < M) ? x[r] : 0.f;
  };
#pragma unroll
  for (int k = 0; k < 3; ++k) tr_piece(0, k);
#pragma unroll
  for (int q = 0; q < 4; ++q) map_piece(0, q);
  // (order as inside the loop: a tile's map block, then its image tile)
#pragma unroll
  for (int q = 0; q < 4; ++q) map_piece(1, q);
#pragma unroll
  for (int k = 0; k < 3; ++k) tr_piece(1, k);
  f32x16 acc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) acc[dt] = zero16();
  KaccDuo kd;
  if (DUO) kd.init(a, Tb, mtiles, b, lane);
  asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // all but image tile 1
  Tri bcur[2], bnext[2];
  {
    float x0[16];
    read_x(0, x0);
    if (mtiles == 1) mask_tail(0, x0);
    if (CS) {
#pragma unroll
      for (int r = 0; r < 16; ++r) csum += x0[r];
    }
    if (DUO) {
      const float f0 = kd.factor(0);
      bcur[0] = duo_from_acc(x0, 0, f0);
      bcur[1] = duo_from_acc(x0, 1, f0);
    } else {
      bcur[0] = tri_from_acc(x0, 0);
      bcur[1] = tri_from_acc(x0, 1);
    }
  }

  auto step = [&](int t, auto next_tail_c, auto last_c) {
    constexpr bool NEXT_TAIL = decltype(next_tail_c)::value, LAST = decltype(last_c)::value;
    const char* st = smem_c + (t % kAccPmTrSlots) * kTriTile;
    // Outstanding here: tile t+1's map block (4 pieces of the previous iteration's k-steps 0-3), then image tile t+1
    // (3 pieces, k-steps 4-6).  The block -- this wave's own pieces, nobody else reads them -- must have landed.
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    float x[16];
    if (!LAST) {
      read_x(t + 1, x);
      if (NEXT_TAIL) mask_tail(t + 1, x);
    }
    const float f1 = DUO ? kd.factor(min(t + 1, mtiles - 1)) : 1.f;  // tile t+1's map values: x f1 before their split
    auto fetch = [&](int i) {  // step i: k-step i >> 2, channel block i & 3
      const char* ap = st + tri_tr_off(32 * (i & 3) + lo, 2 * (i >> 2) + h, 0);
      if (DUO) return Tri{*reinterpret_cast<const u32x4*>(ap), *reinterpret_cast<const u32x4*>(ap + 2048), u32x4{0, 0, 0, 0}};
      return Tri{*reinterpret_cast<const u32x4*>(ap), *reinterpret_cast<const u32x4*>(ap + 2048),
                 *reinterpret_cast<const u32x4*>(ap + 4096)};
    };
    Tri a0 = fetch(0), a1 = fetch(1), a2 = fetch(2);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      Tri a3 = a2;
      if (i + 3 < 8) a3 = fetch(i + 3);
      __builtin_amdgcn_sched_barrier(0);
      if (DUO) acc[i & 3] = mfma_duo(a0.h, a0.m, bcur[i >> 2].h, bcur[i >> 2].m, acc[i & 3]);
      else acc[i & 3] = mfma_tri(a0, bcur[i >> 2], acc[i & 3]);
      if (!LAST) {  // pair c of tile t+1's values -> word c & 3 of the fragment of k-step c >> 2; one k-step behind
                    // the reads (k-step 0 has only its MFMAs: the values are still on their way from LDS)
#pragma unroll
        for (int c = (i == 0 ? 8 : i - 1); c < (i == 7 ? 8 : i); ++c) {
          unsigned hh, mm, ll = 0;
          if (DUO) duo_split2(x[2 * c] * f1, x[2 * c + 1] * f1, hh, mm);
          else tri_split2(x[2 * c], x[2 * c + 1], hh, mm, ll);
          bnext[c >> 2].h[c & 3] = hh;
          bnext[c >> 2].m[c & 3] = mm;
          bnext[c >> 2].l[c & 3] = ll;
          if (CS) csum += x[2 * c] + x[2 * c + 1];
        }
      }
      // this iteration's 7 DMA pieces, one per k-step (issuing them in a row at the top kept both waves of the SIMD
      // off the matrix pipe together): tile t+2's map block into the slot of tile t (split one iteration ago), then
      // image tile t+2 into the slot of tile t-1
      if (i < 4) map_piece(t + 2, i);
      else if (i < 7) tr_piece(t + 2, i - 4);
#pragma unroll
      for (int m = 0; m < (DUO ? 3 : 6); ++m) {  // the weave: an MFMA, then its share of the vector work
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, DUO ? 7 : 3, 0);
        if (m == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      a0 = a1;
      a1 = a2;
      a2 = a3;
    }
    bcur[0] = bnext[0];
    bcur[1] = bnext[1];
    // image tile t+1 (the previous iteration's pieces: 7 younger ones) must have landed for everybody
    asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  for (int t = 0; t < mtiles - 2; ++t) step(t, std::false_type{}, std::false_type{});
  if (mtiles >= 2) step(mtiles - 2, std::true_type{}, std::false_type{});
  step(mtiles - 1, std::false_type{}, std::true_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the clamped pieces of the tiles past the end)
  if (CS) {
    const float ctot = csum + wave_xor32(csum);
    if (jvalid && h == 0) a.cs[(long)b * a.NK + j] = ctot;
  }
  if (jvalid) {
    const float un = a.duo ? kd.out_un : 1.f;  // (x 1 is exact: the three-plane results are unchanged)
    float* orow = a.out + (long)b * a.o_bs + (long)j * a.o_rs + 4 * h;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 o = {acc[dt][4 * g] * un, acc[dt][4 * g + 1] * un, acc[dt][4 * g + 2] * un, acc[dt][4 * g + 3] * un};
        *reinterpret_cast<f32x4*>(orow + 32 * dt + 8 * g) = o;
      }
    }
  }
}

}  // namespace samble

using namespace samble;

#ifdef SAMBLE_STAMPS
extern "C" __attribute__((visibility("default"))) int samble_scratch_ka_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(samble::g_ka_stamps), sizeof(unsigned long long) * 128);
}
#endif

extern "C" size_t samble_bwd_tri_dsmap_floats(int B, int N, int M) { return (size_t)B * M * (32 * ((N + 8 + 31) / 32)); }
// the dS map (B, M, ld) + behind it one word per cloud: the largest |dS| (what scales dS for the two-plane dK products)
extern "C" size_t samble_bwd_tri_dsmap_bytes(int B, int N, int M) {
  return samble_bwd_tri_dsmap_floats(B, N, M) * sizeof(float) + (((size_t)B * 4 + 255) & ~(size_t)255);
}

extern "C" int samble_launch_bwd_tri(const float* smap, int ld, const float* lse_s, const float* delta, const void* dO_rm,
                                     const void* dO_tr, const void* Q_tr, const void* V_rm, const void* K_tr,
                                     const long long* idx, int B, int N, int nt, int M, float scale, float* dQ, long dq_bs,
                                     long dq_rs, float* dK, long dk_bs, long dk_rs, float* dV, long dv_bs, long dv_rs,
                                     float* cs, float* dsmap, int fused_dkdv, int pmap, unsigned* ds_amax,
                                     hipStream_t stream) {
  // ds_amax != null (dS-map variants only): bwd_prep_tri wrote dO_tr / Q_tr as two fp16 planes per tile and cleared the
  // clouds' slots; the dQ kernel raises them to the largest |dS|, the key-stationary kernels run three products
  // pmap != 0: smap is the P map (B, M, ld) of the sampled rows (attn_rows_rc_tri); needs the dS-map variant
  // fused_dkdv == 0 (default): the dQ kernel writes a dS map and dV / dK accumulate from the maps (4 products per
  // tile); != 0: fused dP / dV / dK kernel (5 products, no dS map)
  {  // per call: cheap, and correct for every device / thread (no process-wide 'done' flag)
    hipError_t e = hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_dq_tri_kernel<0, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kDqLds);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_dq_pm_tri_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kDqpLds);
    if (e != hipSuccess) return (int)e;
    for (const void* f : {reinterpret_cast<const void*>(bwd_dkdv_tri_kernel<false>), reinterpret_cast<const void*>(bwd_dkdv_tri_kernel<true>)}) {
      e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kKvLds);
      if (e != hipSuccess) return (int)e;
    }
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_kacc_tri_kernel<0, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kAccLds);
    if (e != hipSuccess) return (int)e;
    for (const void* f : {reinterpret_cast<const void*>(bwd_kacc_pm_tri_kernel<false, false>), reinterpret_cast<const void*>(bwd_kacc_pm_tri_kernel<true, false>),
                          reinterpret_cast<const void*>(bwd_kacc_pm_tri_kernel<false, true>), reinterpret_cast<const void*>(bwd_kacc_pm_tri_kernel<true, true>)}) {
      e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kAccPmLds);
      if (e != hipSuccess) return (int)e;
    }
  }
  const bool use_map = !fused_dkdv && dsmap;
  if (pmap && !use_map) return (int)hipErrorInvalidValue;
  const DqTriArgs dq{smap, ld, lse_s, delta, (const char*)dO_rm, (const char*)V_rm, (const char*)K_tr, idx, N, N + nt, M,
                     scale, dQ, dq_bs, dq_rs, use_map ? dsmap : nullptr, use_map ? ds_amax : nullptr};
  const int duo = (use_map && ds_amax) ? 1 : 0;
  {
    Timed timed(kT_bwd_dq, stream);
    if (pmap) hipLaunchKernelGGL(bwd_dq_pm_tri_kernel, dim3((M + 127) / 128, B), dim3(256), kDqpLds, stream, dq);
    else hipLaunchKernelGGL((bwd_dq_tri_kernel<0, false>), dim3((M + 127) / 128, B), dim3(256), kDqLds, stream, dq);
  }
  if (use_map) {
    const KaccArgs av{smap, ld, lse_s, (const char*)dO_tr, idx, N, N + nt, M, dV, dv_bs, dv_rs, nullptr, duo, nullptr};
    const KaccArgs ak{dsmap, ld, nullptr, (const char*)Q_tr, idx, N, N + nt, M, dK, dk_bs, dk_rs, cs, duo, duo ? ds_amax : nullptr};
    const dim3 grid((N + 255) / 256, B);
    auto launch_dv = [&] {
      Timed timed(kT_bwd_dv, stream);
      if (pmap) {  // P is there already: the M-row-map kernel on (P map, dO^T)
        if (duo) hipLaunchKernelGGL((bwd_kacc_pm_tri_kernel<false, true>), grid, dim3(512), kAccPmLds, stream, av);
        else hipLaunchKernelGGL((bwd_kacc_pm_tri_kernel<false, false>), grid, dim3(512), kAccPmLds, stream, av);
      } else {
        hipLaunchKernelGGL((bwd_kacc_tri_kernel<0, false>), grid, dim3(512), kAccLds, stream, av);
      }
    };
    auto launch_dk = [&] {
      Timed timed(kT_bwd_dk, stream);
      if (cs) {
        if (duo) hipLaunchKernelGGL((bwd_kacc_pm_tri_kernel<true, true>), grid, dim3(512), kAccPmLds, stream, ak);
        else hipLaunchKernelGGL((bwd_kacc_pm_tri_kernel<true, false>), grid, dim3(512), kAccPmLds, stream, ak);
      } else {
        if (duo) hipLaunchKernelGGL((bwd_kacc_pm_tri_kernel<false, true>), grid, dim3(512), kAccPmLds, stream, ak);
        else hipLaunchKernelGGL((bwd_kacc_pm_tri_kernel<false, false>), grid, dim3(512), kAccPmLds, stream, ak);
      }
    };
    // dK first: the dS map is what the dQ kernel wrote last
    if (SAMBLE_KACC_DK_FIRST) {
      launch_dk();
      launch_dv();
    } else {
      launch_dv();
      launch_dk();
    }
  } else {
    const KvTriArgs kv{smap, ld, lse_s, delta, (const char*)dO_rm, (const char*)dO_tr, (const char*)Q_tr, (const char*)V_rm,
                       idx, N, N + nt, M, scale, dK, dk_bs, dk_rs, dV, dv_bs, dv_rs, cs};
    Timed timed(kT_bwd_dv, stream);
    if (cs) hipLaunchKernelGGL(bwd_dkdv_tri_kernel<true>, dim3((N + 127) / 128, B), dim3(256), kKvLds, stream, kv);
    else hipLaunchKernelGGL(bwd_dkdv_tri_kernel<false>, dim3((N + 127) / 128, B), dim3(256), kKvLds, stream, kv);
  }
  return (int)hipGetLastError();
}
