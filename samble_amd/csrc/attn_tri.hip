// The attention passes of attn_map.hip on the bf16 matrix cores with split ("tri") fp32 operands
// (tri_dev.h): same map layout, same statistics, same orientation of every tile, 2.6x the matrix
// rate and the vector ALU free under the MFMAs.  Q, K (and V) arrive as operand images written by
// tri_split_kernel (or by the projection's epilogue).
#include <type_traits>

#include "tri_dev.h"

namespace samble {

// ------------------------------------------------------------------------------------------------
// fp32 rows (B, R, 128) with strides -> RM and / or TR operand images, one workgroup per 32-row tile.
// Rows >= R of the last tile are zeros.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tri_split_kernel(const float* __restrict__ src, long bs, long rs, int R,
                                                        char* __restrict__ rm, char* __restrict__ tr) {
  const int tile = blockIdx.x, b = blockIdx.y, ntiles = gridDim.x, tid = threadIdx.x;
  const float* sb = src + (long)b * bs;
  if (rm) {
    char* img = rm + ((long)b * ntiles + tile) * kTriTile;
    for (int e = tid; e < 512; e += 256) {
      const int r = e & 31, g = e >> 5, row = tile * 32 + r;
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = 0.f;
      if (row < R) {
        const f32x4* p = reinterpret_cast<const f32x4*>(sb + (long)row * rs + 8 * g);
        const f32x4 a = p[0], bb = p[1];
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = a[i]; x[4 + i] = bb[i]; }
      }
      const Tri t = tri_split8(x);
      *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 0)) = t.h;
      *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 1)) = t.m;
      *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 2)) = t.l;
    }
  }
  if (tr) {
    char* img = tr + ((long)b * ntiles + tile) * kTriTile;
    for (int e = tid; e < 512; e += 256) {
      const int d = e & 127, cg = e >> 7, s = cg >> 1, hh = cg & 1;
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = tile * 32 + 16 * s + 8 * (i >> 2) + 4 * hh + (i & 3);
        x[i] = (row < R) ? sb[(long)row * rs + d] : 0.f;
      }
      const Tri t = tri_split8(x);
      *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 0)) = t.h;
      *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 1)) = t.m;
      *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 2)) = t.l;
    }
  }
}

// One launch for the three operands of the attention passes: qkv rows (B, N+nt, 384) = [Q | K | V] ->
// RM image of Q (tiles of the N point rows), RM image of K and TR image of V (N+nt rows).  One workgroup
// per 32-row tile; every qkv row is read once.
__global__ __launch_bounds__(256) void tri_split_qkv_kernel(const float* __restrict__ qkv, long bs, long rs, int N,
                                                            int NK, char* __restrict__ qimg, char* __restrict__ kimg,
                                                            char* __restrict__ vimg, char* __restrict__ ktr,
                                                            char* __restrict__ vrm, int tile0) {
  const int tile = tile0 + blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int qtiles = (N + 31) / 32, ntiles = (NK + 31) / 32;
  const float* sb = qkv + (long)b * bs;
  for (int e = tid; e < 1536; e += 256) {  // RM chunks of Q (e < 512), K and (for the backward) V
    const int which = e >> 9, r = e & 31, g = (e >> 5) & 15, row = tile * 32 + r;
    if (which == 0 && tile >= qtiles) continue;
    if (which == 2 && !vrm) continue;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.f;
    if (row < (which == 0 ? N : NK)) {  // the Q image covers the N point rows only
      const f32x4* p = reinterpret_cast<const f32x4*>(sb + (long)row * rs + 128 * which + 8 * g);
      const f32x4 a = p[0], bb = p[1];
#pragma unroll
      for (int i = 0; i < 4; ++i) { x[i] = a[i]; x[4 + i] = bb[i]; }
    }
    const Tri t = tri_split8(x);
    char* img = which == 0 ? qimg + ((long)b * qtiles + tile) * kTriTile
                           : (which == 1 ? kimg : vrm) + ((long)b * ntiles + tile) * kTriTile;
    *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 0)) = t.h;
    *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 1)) = t.m;
    *reinterpret_cast<u32x4*>(img + tri_rm_off(r, g, 2)) = t.l;
  }
  for (int e = tid; e < 1024; e += 256) {  // TR chunks of V (e < 512) and (for the backward) K
    const int which = e >> 9, d = e & 127, cg = (e >> 7) & 3, s = cg >> 1, hh = cg & 1;
    if (which == 1 && !ktr) continue;
    char* img = (which ? ktr : vimg) + ((long)b * ntiles + tile) * kTriTile;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = tile * 32 + 16 * s + 8 * (i >> 2) + 4 * hh + (i & 3);
      x[i] = (row < NK) ? sb[(long)row * rs + (which ? 128 : 256) + d] : 0.f;
    }
    const Tri t = tri_split8(x);
    *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 0)) = t.h;
    *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 1)) = t.m;
    *reinterpret_cast<u32x4*>(img + tri_tr_off(d, cg, 2)) = t.l;
  }
}

// ------------------------------------------------------------------------------------------------
// pass 1 (attn_stats_kernel of attn_map.hip, same structure): one workgroup = 8 waves = 256 query rows,
// K image tiles triple-buffered in LDS and fetched two tiles ahead, S^T orientation (keys on the
// register axis, queries on lanes), wave-private transpose tile for 128-byte map stores.
// ------------------------------------------------------------------------------------------------
constexpr int kStPadT = 36;

// One tile step = the 48 MFMAs of tile t+1's S product + the softmax statistics / map store of tile t.
// The loop body must stay ONE basic block: the staged tile is waited for with a counted vmcnt (stores
// share the counter on gfx9), and any branch in the body makes the compiler fall back to waiting for
// the loads it has just issued (measured: 2900 cycles per tile instead of the MFMA time).
// ABL (timing-only ablations, wrong outputs): 1 = no map stores, 2 = no matrix products
template <int ABL>
__device__ __forceinline__ void stats_products(const char* __restrict__ Kn, int lo, int h, const u32x4 (&qd)[16],
                                               float q_scale, f32x16& s_nxt, float& sc_nxt) {
  // Kn: a K tile converted to two fp16 planes (tri_k_to_duo_kernel): h in the first, l in the second piece slot of
  // each channel group; qd: this lane's query row as two fp16 planes (duo_q_from_tri)
  const u32x4* lp = reinterpret_cast<const u32x4*>(Kn + tri_rm_off(lo, h, 0));  // group 2 ks + h: + ks * 192 chunks
  // q_scale = 2^-e_q x the logit scale 1/sqrt(D); x the tile's 2^-e_k (both factors of two: exact) = the ONE factor that
  // turns the raw accumulator into the logit -- a single rounding, the bits round(S 2^-(e_q + e_k) / sqrt(D)) in every
  // kernel that forms logits
  sc_nxt = q_scale * *reinterpret_cast<const float*>(Kn + kDuoScaleSlot);
  s_nxt = zero16();
  if (ABL & 16) {  // timing only (wrong logits): half the operand reads, every pair of k-steps shares one
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) {
      const u32x4 ah = lp[192 * 2 * kp], al = lp[192 * 2 * kp + 32];
      s_nxt = mfma_duo(ah, al, qd[4 * kp], qd[4 * kp + 1], s_nxt);
      s_nxt = mfma_duo(ah, al, qd[4 * kp + 2], qd[4 * kp + 3], s_nxt);
    }
  } else if (ABL & 8) {  // the compiler's own placement of the operand reads
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const u32x4 ah = lp[192 * ks], al = lp[192 * ks + 32];
      if (ABL & 2) s_nxt[ks] += __uint_as_float(ah[0] ^ qd[2 * ks + 1][1]);
      else s_nxt = mfma_duo(ah, al, qd[2 * ks], qd[2 * ks + 1], s_nxt);
    }
  } else {
    // operand reads two k-steps ahead of their MFMAs (stamped: 44-51 cycles per MFMA with the compiler's placement)
    tri_pipelined<8>([&](int ks) { return Tri{lp[192 * ks], lp[192 * ks + 32], u32x4{0, 0, 0, 0}}; },
                     [&](int ks, const Tri& a) {
                       if (ABL & 2) s_nxt[ks] += __uint_as_float(a.h[0] ^ qd[2 * ks + 1][1]);
                       else s_nxt = mfma_duo(a.h, a.m, qd[2 * ks], qd[2 * ks + 1], s_nxt);
                     });
  }
}

template <bool TAIL, bool L2, int ABL>
__device__ __forceinline__ void stats_epilogue(int lo, int h, f32x16& s_cur, float scale, float* __restrict__ xt,
                                               float* __restrict__ gdst, const int (&roff)[4], int j0, int N, int NK,
                                               float* __restrict__ tokrow, float& m, float& l, float qb,
                                               const float (&kb)[16]) {
  const int lane = lo + 32 * h;
  float mt = kNegInf, ps = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {  // scale, tile max; registers 4g .. 4g+3 (4 consecutive keys of this lane's row) -> transpose tile
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * g + e;
      // dot: <q,k>/sqrt(D);  l2: -|q-k|^2/sqrt(D) = (2<q,k> - |q|^2 - |k|^2)/sqrt(D), qb / kb = the norms x scale
      float v = L2 ? fmaf(s_cur[r], 2.f * scale, -qb) - kb[r] : s_cur[r] * scale;
      if (TAIL) {
        const int j = j0 + crow(r, h);
        if (j >= NK) v = kNegInf;
        if (j >= N && j < NK) tokrow[j - N] = v;
      }
      s_cur[r] = v;
      mt = fmaxf(mt, v);
    }
    const f32x4 o = {s_cur[4 * g], s_cur[4 * g + 1], s_cur[4 * g + 2], s_cur[4 * g + 3]};
    *reinterpret_cast<f32x4*>(xt + lo * kStPadT + 8 * g + 4 * h) = o;
  }
#pragma unroll
  for (int k8 = 0; k8 < 4; ++k8) {  // map store: 8 lanes cover one row's 32 keys (128 contiguous bytes)
    const int row = (lane >> 3) + 8 * k8;
    const f32x4 o = *reinterpret_cast<const f32x4*>(xt + row * kStPadT + 4 * (lane & 7));
    if (!(ABL & 1)) *reinterpret_cast<f32x4*>(gdst + roff[k8]) = o;
    else if (o[0] == 12345.f) gdst[roff[k8]] = o[1];
  }
  mt = fmaxf(mt, wave_xor32(mt));  // running max (branch-free rescale of the running sum)
  const float mnew = fmaxf(m, mt);
  l *= __expf(m - mnew);
  m = mnew;
  // exp(v - m) as exp2(v log2e - m log2e): one fused multiply-add and the exponential instead of subtract, multiply,
  // exponential (the sum only enters lse; both statistics kernels form it the same way)
  const float m2 = m * 1.4426950408889634f;
#pragma unroll
  for (int r = 0; r < 16; ++r) ps += __builtin_amdgcn_exp2f(fmaf(s_cur[r], 1.4426950408889634f, -m2));
  l += ps;
}

// K tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers), kStatsDepth tiles
// ahead, into a ring of kStatsDepth buffers.  Why so deep: loads, LDS-DMA and stores retire in issue
// order on one counter (vmcnt), so waiting for a tile also waits for every OLDER map store -- and the map
// stream (545 MB at B=32, N=2048) keeps ~3 tiles of stores per CU in flight at HBM write latency.  With
// the tile issued 4 iterations before it is needed the wait is `vmcnt(18)`: only stores at least two
// iterations old have to have landed.  (Register staging one tile ahead made every wave wait for its
// previous tile's stores: 54% of wave time in s_waitcnt, matrix pipe 56% busy.)
constexpr int kStatsDepth = 4;

// A K tile in its logit form (tri_dev.h) holds two live planes of its three: per channel group g (96 chunks) the 64
// chunks of planes h and l, then 32 dead ones -- except chunk 64 of group 0, which keeps the tile's 2^-e.  The DMA
// moves the live chunks only (wave w: groups w and w + 8, 1 KB each) and the scale word as a 4-byte piece: three
// VM operations per wave and tile as before (the counted waits do not change), a third less through the fabric.
__device__ __forceinline__ void glds_tile(const char* __restrict__ gtile, char* lds_tile, int tid, int wave) {
  const int lane = tid & 63;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int g = wave + 8 * i;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gtile + (g * 96 + lane) * 16),
                                     (__attribute__((address_space(3))) void*)(lds_tile + g * 96 * 16), 16, 0, 0);
  }
  // (every wave issues it -- the same word to the same place -- so that all waves count the same operations)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gtile + kDuoScaleSlot + lane * 4),
                                   (__attribute__((address_space(3))) void*)(lds_tile + kDuoScaleSlot), 4, 0, 0);
}

template <bool L2, int ABL = 0>
__global__ __launch_bounds__(512, 2) void attn_stats_tri_kernel(const char* __restrict__ Qimg, const char* __restrict__ Kimg,
                                                                int N, int NK, float scale, float* __restrict__ smap,
                                                                int ld, float* __restrict__ lse, float* __restrict__ tok,
                                                                int nt, const float* __restrict__ qn,
                                                                const float* __restrict__ kn) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  constexpr int NW = 8, D = kStatsDepth;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int qtiles = (N + kTile - 1) / kTile, ntiles = (NK + kTile - 1) / kTile;
  // rows past N are clamped to row N-1 (they recompute and rewrite its values bit for bit): every store
  // stays unpredicated
  const int qrow = min(chunk * (32 * NW) + wave * 32 + lo, N - 1);
  const char* Kb = Kimg + (long)b * ntiles * kTriTile;
  auto tile_ptr = [&](int t) { return Kb + (long)((ABL & 8) ? (t & 1) : min(t, ntiles - 1)) * kTriTile; };  // past the end: the last tile again, unused
  auto buf_ptr = [&](int t) { return smem_c + (t & (D - 1)) * kTriTile; };
  static_assert((D & (D - 1)) == 0, "ring size must be a power of two");
#pragma unroll
  for (int t = 0; t < D; ++t) glds_tile(tile_ptr(t), buf_ptr(t), tid, wave);

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
  const int row0 = chunk * (32 * NW) + wave * 32;
  float* xt = reinterpret_cast<float*>(smem_c + D * kTriTile) + wave * (kTile * kStPadT);
  float* knl = reinterpret_cast<float*>(smem_c + D * kTriTile) + NW * kTile * kStPadT;  // l2: |k_j|^2 scale, ld floats
  float* tokrow = tok + ((long)b * N + qrow) * nt;
  float m = kNegInf, l = 0.f;
  float qb = 0.f, kb[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) kb[r] = 0.f;
  if (L2) {
    qb = qn[(long)b * N + qrow] * scale;
    for (int j = tid; j < ld; j += 512) knl[j] = kn[(long)b * ld + j] * scale;
  }
  auto load_kb = [&](int tile) {
    if (L2) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(knl + tile * kTile + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) kb[4 * g + e] = v4[e];
      }
    }
  };

  int roff[4];
#pragma unroll
  for (int k8 = 0; k8 < 4; ++k8) roff[k8] = min(row0 + (lane >> 3) + 8 * k8, N - 1) * ld + 4 * (lane & 7);
  float* gdst = smap + (long)b * N * ld;

  // all D prologue tiles (and this wave's Q rows) have landed; from here on the waits are counted
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  u32x4 qd[16];  // the query row as two fp16 planes under its own scale (tri_dev.h)
  float q_unscale;
  duo_q_from_tri(q, qd, q_unscale);
  const float q_scale = q_unscale * scale;  // (exact: q_unscale is a power of two)
  float sc_cur, sc_nxt;
  f32x16 s_cur, s_nxt;
  stats_products<ABL>(buf_ptr(0), lo, h, qd, q_scale, s_cur, sc_cur);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // buffer 0 is restaged by iteration 0

  // iteration t: restage the buffer of tile t (read one iteration ago) with tile t+D, products of tile
  // t+1, statistics / map store of tile t, then retire tile t+2's DMA.  VM operations issued after that
  // DMA: stores of iteration t+2-D, then (3 DMA + 4 stores) per later iteration = 4 + 7 (D - 2).
  const bool pfirst = wave < 4;
  auto step = [&](int t, auto tail_c) {
    constexpr bool TAIL = decltype(tail_c)::value;
    const int j0 = t * kTile;
    glds_tile(tile_ptr(t + D), buf_ptr(t), tid, wave);
    load_kb(t);
    // the two waves of a SIMD (w and w + 4) take the two phases in opposite order, so one's vector work
    // and stores run under the other's MFMAs (same order: both in the MFMA phase, then both out of it)
    if (pfirst || (ABL & 4)) {
      stats_products<ABL>(buf_ptr(t + 1), lo, h, qd, q_scale, s_nxt, sc_nxt);
      __builtin_amdgcn_sched_barrier(0);
      stats_epilogue<TAIL, L2, ABL>(lo, h, s_cur, sc_cur, xt, gdst + j0, roff, j0, N, NK, tokrow, m, l, qb, kb);
    } else {
      stats_epilogue<TAIL, L2, ABL>(lo, h, s_cur, sc_cur, xt, gdst + j0, roff, j0, N, NK, tokrow, m, l, qb, kb);
      __builtin_amdgcn_sched_barrier(0);
      stats_products<ABL>(buf_ptr(t + 1), lo, h, qd, q_scale, s_nxt, sc_nxt);
    }
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"((ABL & 1) ? 3 * (D - 2) : 4 + 7 * (D - 2))
                 : "memory");
    s_cur = s_nxt;
    sc_cur = sc_nxt;
  };
  const int n_full = min(N / kTile, ntiles);  // tiles without token / padding columns
  int t = 0;
  for (; t < n_full; ++t) step(t, std::false_type{});
  for (; t < ntiles; ++t) step(t, std::true_type{});
  const float ltot = l + wave_xor32(l);
  if (h == 0) lse[(long)b * N + qrow] = m + __logf(ltot);
}

// ------------------------------------------------------------------------------------------------
// pass 1 WITHOUT the logit map ("nl" = neighbour logits): the sparse_* score modes only touch the K kNN entries of
// a row, so the pass keeps the softmax statistics (lse), the token logits and -- instead of streaming all N x (N+nt)
// logits to HBM (545 MB at B = 32, N = 2048) -- just the K logits S[i][j], j in kNN(i), in ascending-j order
// (nl (B, N, K): 8 MB).  Which keys of a tile are neighbours of a query comes as one 32-bit mask per (query, tile)
// (nn_prepare_kernel, score.hip); the logits sit in this lane's accumulator registers, so the extraction is a bit
// test, a popcount for the slot and a predicated 4-byte LDS write per register (a row of K slots per query in LDS,
// flushed once at the end: predicated GLOBAL stores would sit behind exec-zero branches, and a store that may or
// may not issue cannot be on a hand-counted vmcnt queue).  The sampled rows' logits are recomputed by pass 2
// (attn_rows_rc_tri_kernel).  Same products, same order as attn_stats_tri_kernel: the statistics and every
// extracted logit are bit-identical to that kernel's.
// ------------------------------------------------------------------------------------------------
// Score accumulators of the sparse_* modes (score.hip: sparse_score_map_kernel does the same from a logit array):
// with nn_sorted non-null the flush of a row also forms A_ij = exp(S_ij - lse_i) of its K neighbours and adds it to
// the fixed-point column sums / in-degrees (order-free integer atomics) and, for the row modes, writes the row
// statistic -- the separate score pass and the neighbour-logit array are then not needed at all.
struct NlScoreArgs {
  const int* nn_sorted;          // (B, N, KN) ascending neighbour indices, or null: no accumulation
  unsigned long long* colacc;    // (B, N) sums of round(A * 2^44)
  int* indeg;                    // (B, N)
  float* rowstat;                // (B, N) or null (column modes)
  int row_std;                   // row modes: 0 = sum, 1 = unbiased std over the K entries
};
constexpr float kNlFix = 17592186044416.f;  // 2^44, as score.hip

#ifdef SAMBLE_STAMPS  // scratch builds only: s_memtime marks of workgroup 0, tiles 20 and 21, every wave
__device__ unsigned long long g_nl_stamps[8 * 2 * 8];
#define NL_STAMP(i)                                                                                        \
  do {                                                                                                     \
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (t == 20 || t == 21))                           \
      g_nl_stamps[(wave * 2 + (t - 20)) * 8 + (i)] = __builtin_amdgcn_s_memtime();                         \
  } while (0)
#else
#define NL_STAMP(i) do { } while (0)
#endif
#ifndef SAMBLE_NL_ABL
#define SAMBLE_NL_ABL 0  // scratch builds: stats_products' timing-only ablations inside attn_stats_nl_tri_kernel
#endif
constexpr int kNlStride = 33;  // words per query row in LDS (K <= 32; odd: rows on distinct banks)
constexpr int kStatsNlLds = kStatsDepth * kTriTile + kStatsDepth * 2048 + 256 * kNlStride * 4;

template <bool TAIL>
__device__ __forceinline__ void stats_nl_epilogue(int h, f32x16& s_cur, float scale, unsigned mask, int& cnt,
                                                  float* nlrow, int j0, int N, int NK,
                                                  float* __restrict__ tokrow, float& m, float& l) {
  float mt = kNegInf, ps = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = s_cur[r] * scale;
    const int kk = crow(r, h);
    if (TAIL) {
      const int j = j0 + kk;
      if (j >= NK) v = kNegInf;
      if (j >= N && j < NK) tokrow[j - N] = v;
    }
    s_cur[r] = v;
    mt = fmaxf(mt, v);
    if (TAIL && ((mask >> kk) & 1u)) nlrow[cnt + (int)__popc(mask & ((1u << kk) - 1u))] = v;
  }
  if (!TAIL) {
    // neighbour logits of this tile: a lane holds 16 of the tile's 32 keys and on average a quarter of a neighbour
    // among them, so instead of 16 predicated writes (8 instructions each, issued whether or not the bit is set:
    // 128 of the epilogue's 208) the lanes walk their set bits -- the fullest lane of the wave sets the trip count
    // (about 2) -- and pick the value out of the 16 registers by a select tree
    unsigned sub = __builtin_amdgcn_ubfe(mask, 4 * h, 4) | (__builtin_amdgcn_ubfe(mask, 8 + 4 * h, 4) << 4) |
                   (__builtin_amdgcn_ubfe(mask, 16 + 4 * h, 4) << 8) | (__builtin_amdgcn_ubfe(mask, 24 + 4 * h, 4) << 12);
    if (SAMBLE_NL_ABL & 32) sub = 0;  // (timing only: no neighbour extraction)
    while (__any(sub != 0)) {
      const int r = __builtin_ctz(sub | 0x10000u) & 15;  // (lanes without a bit left pick register 0 and write nothing)
      const bool b0 = r & 1, b1 = r & 2, b2 = r & 4;
      const float e0 = b0 ? s_cur[1] : s_cur[0], e1 = b0 ? s_cur[3] : s_cur[2], e2 = b0 ? s_cur[5] : s_cur[4];
      const float e3 = b0 ? s_cur[7] : s_cur[6], e4 = b0 ? s_cur[9] : s_cur[8], e5 = b0 ? s_cur[11] : s_cur[10];
      const float e6 = b0 ? s_cur[13] : s_cur[12], e7 = b0 ? s_cur[15] : s_cur[14];
      const float f0 = b1 ? e1 : e0, f1 = b1 ? e3 : e2, f2 = b1 ? e5 : e4, f3 = b1 ? e7 : e6;
      const float g0 = b2 ? f1 : f0, g1 = b2 ? f3 : f2;
      const float val = (r & 8) ? g1 : g0;
      const int kk = 8 * (r >> 2) + 4 * h + (r & 3);
      if (sub != 0) nlrow[cnt + (int)__popc(mask & ((1u << kk) - 1u))] = val;
      sub &= sub - 1;
    }
  }
  cnt += (int)__popc(mask);
  mt = fmaxf(mt, wave_xor32(mt));  // running max (branch-free rescale of the running sum)
  const float mnew = fmaxf(m, mt);
  l *= __expf(m - mnew);
  m = mnew;
  // exp(v - m) as exp2(v log2e - m log2e): one fused multiply-add and the exponential instead of subtract, multiply,
  // exponential (the sum only enters lse; both statistics kernels form it the same way)
  const float m2 = m * 1.4426950408889634f;
#pragma unroll
  for (int r = 0; r < 16; ++r)
    ps += (SAMBLE_NL_ABL & 64) ? fmaf(s_cur[r], 1.4426950408889634f, -m2) : __builtin_amdgcn_exp2f(fmaf(s_cur[r], 1.4426950408889634f, -m2));
  l += ps;
}

__global__ __launch_bounds__(512, 2) void attn_stats_nl_tri_kernel(const char* __restrict__ Qimg,
                                                                   const char* __restrict__ Kimg, int N, int NK,
                                                                   float scale, const unsigned* __restrict__ masks,
                                                                   int KN, float* __restrict__ nl,
                                                                   float* __restrict__ lse, float* __restrict__ tok,
                                                                   int nt, const NlScoreArgs sc) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  constexpr int NW = 8, D = kStatsDepth;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int qtiles = (N + kTile - 1) / kTile, ntiles = (NK + kTile - 1) / kTile;
  const int mtiles = qtiles;  // mask words per query: one per tile of POINT keys
  // rows past N are clamped to row N-1 (they recompute and rewrite its values bit for bit)
  const int qrow = min(chunk * (32 * NW) + wave * 32 + lo, N - 1);
  const char* Kb = Kimg + (long)b * ntiles * kTriTile;
  auto tile_ptr = [&](int t) { return Kb + (long)min(t, ntiles - 1) * kTriTile; };  // past the end: the last tile again, unused
  auto buf_ptr = [&](int t) { return smem_c + (t & (D - 1)) * kTriTile; };
  // the mask word of (this query, tile t) rides along with tile t through the ring, by LDS-DMA as well: a
  // register-returning load in the loop would make the compiler drain the whole queue (vmcnt(0)) every tile
  const unsigned* mrow = masks + (long)b * mtiles * N + qrow;  // + t * N: the word of tile t
  char* mring = smem_c + D * kTriTile;                        // D slots of 512 words
  auto stage = [&](int t) {
    glds_tile(tile_ptr(t), buf_ptr(t), tid, wave);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(mrow + (long)min(t, mtiles - 1) * N),
                                     (__attribute__((address_space(3))) void*)(mring + (t & (D - 1)) * 2048 + wave * 256),
                                     4, 0, 0);
  };
#pragma unroll
  for (int t = 0; t < D; ++t) stage(t);

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
  float* tokrow = tok + ((long)b * N + qrow) * nt;
  // neighbour logits of this wave's 32 queries: rows of kNlStride words behind the mask ring (both half-waves fill
  // the row of their query; the slots are disjoint)
  float* nlrow = reinterpret_cast<float*>(mring + D * 2048) + (wave * 32 + lo) * kNlStride;
  float m = kNegInf, l = 0.f;
  int cnt = 0;
  auto mask_of = [&](int t) {  // tiles of token / padding keys only: no neighbours
    const unsigned w = *reinterpret_cast<const unsigned*>(mring + (t & (D - 1)) * 2048 + tid * 4);
    return t < mtiles ? w : 0u;
  };

  // all D prologue tiles (and this wave's Q rows) have landed; from here on the waits are counted
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  u32x4 qd[16];  // the query row as two fp16 planes under its own scale (tri_dev.h)
  float q_unscale;
  duo_q_from_tri(q, qd, q_unscale);
  const float q_scale = q_unscale * scale;  // (exact: q_unscale is a power of two)
  float sc_cur, sc_nxt;
  f32x16 s_cur, s_nxt;
  stats_products<SAMBLE_NL_ABL>(buf_ptr(0), lo, h, qd, q_scale, s_cur, sc_cur);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // buffer 0 is restaged by iteration 0

  // iteration t: restage the slot of tile t (read one iteration ago) with tile t+D and its mask word, products of
  // tile t+1, statistics / neighbour logits of tile t, then retire tile t+2's DMA.  VM operations issued after
  // that DMA for certain: 3 + 1 DMA pieces per later iteration = 4 (D - 2).  (The tail tiles' predicated token
  // stores may add to that: counting low is the safe side, vmcnt(n) = at most n pending.)
  const bool pfirst = wave < 4;
  auto step = [&](int t, auto tail_c) {
    constexpr bool TAIL = decltype(tail_c)::value;
    const int j0 = t * kTile;
    NL_STAMP(0);
    const unsigned mask_cur = mask_of(t);  // before its slot is restaged
    stage(t + D);
    NL_STAMP(1);
    if (pfirst) {
      stats_products<SAMBLE_NL_ABL>(buf_ptr(t + 1), lo, h, qd, q_scale, s_nxt, sc_nxt);
      __builtin_amdgcn_sched_barrier(0);
      NL_STAMP(2);
      stats_nl_epilogue<TAIL>(h, s_cur, sc_cur, mask_cur, cnt, nlrow, j0, N, NK, tokrow, m, l);
      NL_STAMP(3);
    } else {
      stats_nl_epilogue<TAIL>(h, s_cur, sc_cur, mask_cur, cnt, nlrow, j0, N, NK, tokrow, m, l);
      __builtin_amdgcn_sched_barrier(0);
      NL_STAMP(2);
      stats_products<SAMBLE_NL_ABL>(buf_ptr(t + 1), lo, h, qd, q_scale, s_nxt, sc_nxt);
      NL_STAMP(3);
    }
#ifdef SAMBLE_STAMPS
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(4 * (D - 2)) : "memory");
    NL_STAMP(4);
    asm volatile("s_barrier" ::: "memory");
    NL_STAMP(5);
#else
#if defined(SAMBLE_NL_NOBAR)  // scratch builds, timing only (races on the ring): what the per-tile barrier costs
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(4 * (D - 2)) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(4 * (D - 2)) : "memory");
#endif
#endif
    s_cur = s_nxt;
    sc_cur = sc_nxt;
  };
  const int n_full = min(N / kTile, ntiles);  // tiles without token / padding columns
  int t = 0;
  for (; t < n_full; ++t) step(t, std::false_type{});
  for (; t < ntiles; ++t) step(t, std::true_type{});
  const float ltot = l + wave_xor32(l);
  const float my_lse = m + __logf(ltot);
  if (h == 0) lse[(long)b * N + qrow] = my_lse;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the row was written by this wave's two halves only
  if (nl) {
    float* nlout = nl + ((long)b * N + qrow) * KN;
    for (int k = h; k < KN; k += 2) nlout[k] = nlrow[k];
  }
  const bool live = chunk * (32 * NW) + wave * 32 + lo < N;  // clamped duplicates of row N-1 must not count twice
  if (!sc.nn_sorted) return;  // (uniform)
  // Column sums: scattered device-scope atomics from every lane would each be a memory-side transaction (measured:
  // +250 us); instead the workgroup's 256 rows x K entries meet in LDS accumulators laid over the tile ring (free
  // now: N x 12 bytes, N <= 8192), and one coalesced sweep adds the non-empty columns to the cloud's totals.
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // the ring's last (unused) DMA pieces have landed
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem_c);
  int* deg = reinterpret_cast<int*>(acc + N);
  for (int n = tid; n < N; n += 512) {
    acc[n] = 0ull;
    deg[n] = 0;
  }
  __syncthreads();
  if (live) {
    const int* jrow = sc.nn_sorted + ((long)b * N + qrow) * KN;
    for (int k = h; k < KN; k += 2) {
      const float a = __expf(nlrow[k] - my_lse);
      const int j = jrow[k];
      atomicAdd(&acc[j], (unsigned long long)__float2ll_rn(a * kNlFix));
      atomicAdd(&deg[j], 1);
    }
  }
  __syncthreads();
  for (int n = tid; n < N; n += 512) {
    const int cn = deg[n];
    if (cn) {
      atomicAdd(&sc.colacc[(long)b * N + n], acc[n]);
      atomicAdd(&sc.indeg[(long)b * N + n], cn);
    }
  }
  if (live) {
    if (sc.rowstat && h == 0) {
      // the K entries in double, summed in the order of sparse_score_map_kernel's 32-lane xor butterfly (entry k on
      // lane k, zeros past K): same partial sums, bit for bit
#pragma clang fp contract(off)
      double x[32], x2[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const double a = k < KN ? (double)__expf(nlrow[k] - my_lse) : 0.0;
        x[k] = a;
        x2[k] = a * a;
      }
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1)
#pragma unroll
        for (int c = 0; c < off; ++c) {
          x[c] += x[c + off];
          x2[c] += x2[c + off];
        }
      const double tot = x[0], tot2 = x2[0];
      float v = (float)tot;
      if (sc.row_std) {
        const double mean = tot / KN;
        v = (float)sqrt(fmax((tot2 - KN * mean * mean) / (KN - 1), 0.0));
      }
      sc.rowstat[(long)b * N + qrow] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// pass 2 (attn_rows_kernel of attn_map.hip): one workgroup = 4 waves = 128 SAMPLED rows of one cloud.
// Lane (i, h) takes its row's 16 logits per key tile from the map, P = exp(S - lse) lands in the
// accumulator layout, is split into three bf16 planes in registers (accumulator-as-operand: registers
// 8s .. 8s+7 are the fragment of k-step s) and feeds O^T += V_tile^T P^T: 48 MFMAs per tile against the
// TR image of V (LDS ring filled by LDS-DMA, two tiles ahead).  The logits go through LDS-DMA as well
// (a wave-private 4 KB slot per tile, read back by the lane that addressed it), so that every in-flight
// memory operation of the loop is on the one hand-counted vmcnt queue.
// ------------------------------------------------------------------------------------------------
constexpr int kRowsDepth = 3;                                        // tiles in LDS: t (in use), t+1, t+2 (in flight)
constexpr int kRowsLds = kRowsDepth * (kTriTile + 4 * 4096);         // V ring + S slots of the 4 waves

__global__ __launch_bounds__(256) void attn_rows_tri_kernel(const float* __restrict__ smap, int ld,
                                                            const float* __restrict__ lse,
                                                            const char* __restrict__ Vtr,
                                                            const long long* __restrict__ idx, int N, int NK, int M,
                                                            float* __restrict__ xds) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  constexpr int NW = 4, D = kRowsDepth;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int mrow = chunk * (32 * NW) + wave * 32 + lo;
  const bool mvalid = mrow < M;
  const long row = idx[(long)b * M + (mvalid ? mrow : M - 1)];
  const float my_lse = lse[(long)b * N + row];
  const float* srow = smap + ((long)b * N + row) * ld + 4 * h;
  const int ntiles = (NK + kTile - 1) / kTile;
  const char* Vb = Vtr + (long)b * ntiles * kTriTile;
  char* sslot = smem_c + D * kTriTile + wave * (D * 4096);  // this wave's S slots

  // tile t -> ring slot t % D: 6 DMA pieces of the V image per thread + 4 of this wave's logits
  auto stage = [&](int t) {
    const int tt = min(t, ntiles - 1);  // past the end: the last tile again, unused
    const int slot = t % D;
    const char* gt = Vb + (long)tt * kTriTile;
    char* lt = smem_c + slot * kTriTile;
#pragma unroll
    for (int k = 0; k < 6; ++k)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gt + (tid + 256 * k) * 16),
                                       (__attribute__((address_space(3))) void*)(lt + (wave * 64 + 256 * k) * 16), 16, 0,
                                       0);
#pragma unroll
    for (int g = 0; g < 4; ++g)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srow + tt * kTile + 8 * g),
                                       (__attribute__((address_space(3))) void*)(sslot + slot * 4096 + g * 1024), 16, 0,
                                       0);
  };
  stage(0);
  stage(1);

  f32x16 oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) oacc[dt] = zero16();
  asm volatile("s_waitcnt vmcnt(10)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // tile 0 landed

  for (int t = 0; t < ntiles; ++t) {
    const int slot = t % D;
    stage(t + 2);  // into the slot of tile t-1, whose reads ended before the last barrier
    const f32x4* sp = reinterpret_cast<const f32x4*>(sslot + slot * 4096 + lane * 16);
    float p[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v4 = sp[64 * g];
#pragma unroll
      for (int e = 0; e < 4; ++e) p[4 * g + e] = __expf(v4[e] - my_lse);
    }
    const char* vt = smem_c + slot * kTriTile;
    Tri bp[2];  // P^T fragments of the two k-steps: elements e <-> registers 8 ks + e
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        unsigned hh, mm, ll;
        tri_split2(p[8 * ks + 2 * w], p[8 * ks + 2 * w + 1], hh, mm, ll);
        bp[ks].h[w] = hh;
        bp[ks].m[w] = mm;
        bp[ks].l[w] = ll;
      }
    }
    tri_pipelined<8>(
        [&](int i) {  // step i: k-step i >> 2, channel block i & 3
          const char* ap = vt + tri_tr_off(32 * (i & 3) + lo, 2 * (i >> 2) + h, 0);
          return Tri{*reinterpret_cast<const u32x4*>(ap), *reinterpret_cast<const u32x4*>(ap + 2048),
                     *reinterpret_cast<const u32x4*>(ap + 4096)};
        },
        [&](int i, const Tri& a) { oacc[i & 3] = mfma_tri(a, bp[i >> 2], oacc[i & 3]); });
    // tile t+1 (staged one iteration ago) must have landed before anyone reads it; the 10 pieces issued
    // in this iteration stay in flight
    asm volatile("s_waitcnt vmcnt(10)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (mvalid) {
    float* ob = xds + (long)b * 128 * M + mrow;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ob[(long)(32 * dt + crow(r, h)) * M] = oacc[dt][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// pass 2 WITHOUT the logit map: the M sampled rows recompute their logits (the same 48 MFMAs per tile and the same
// operand images as pass 1: S comes out bit-identical), P = exp(S - lse) feeds O^T += V^T P^T as in
// attn_rows_tri_kernel, and -- when the backward will run -- the P tile goes out to a P MAP of the SAMPLED rows
// only, pmap (B, M, ld): what bwd_dq_tri / bwd_kacc_tri read (no exp, no row indirection there).  Per tile a
// K row-image tile and a V transposed-image tile by LDS-DMA, ring of 3, every wait counted: per iteration a thread
// issues 12 DMA pieces and (PMAP) 4 stores.
// ------------------------------------------------------------------------------------------------
// a pointer the compiler can keep in scalar registers (the value IS wave-uniform; this tells it so)
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)v), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi32 << 32) | lo32);
}

constexpr int kRcDepth = 3;
constexpr int kRcLds = kRcDepth * 2 * kTriTile + 4 * 4096;  // K ring, V ring, one 32 x 32 fp32 tile per wave

// This kernel runs ONE wave per SIMD (32 768 sampled rows / 32), so nothing but its own instruction stream can
// fill the matrix pipe while it exponentiates and splits P.  Measured (tools/micro/split_mfma_bench.hip): a bf16
// MFMA holds the SIMD's issue port for 16 of its 32 cycles and up to 4 vector instructions fit under the rest, so
// an iteration is three INDEPENDENT streams woven together, one k-step at a time:
//     logit products of tile t+1 (48 MFMAs)  |  P V of tile t-1 (48 MFMAs)  |  vector work on tile t (exp, mask,
//     three-plane split: one slice of two logits per k-step, ~3 vector instructions per MFMA)
// K tiles run two tiles ahead of the V tiles through their rings; every wait is counted.
#ifndef SAMBLE_RC_ABL
#define SAMBLE_RC_ABL 0  // timing-only ablations (wrong results): 1 no in-loop DMA, 2 no P V MFMAs, 4 no logit MFMAs
#endif
#ifdef SAMBLE_STAMPS  // scratch builds only (tools/scratch): s_memtime marks of workgroup 0, tiles 20 and 21
__device__ unsigned long long g_rc_stamps[4 * 2 * 16];
#define RC_STAMP(i)                                                                                        \
  do {                                                                                                     \
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (t == 20 || t == 21))                           \
      g_rc_stamps[(wave * 2 + (t - 20)) * 16 + (i)] = __builtin_amdgcn_s_memtime();                        \
  } while (0)
#else
#define RC_STAMP(i) do { } while (0)
#endif
template <bool PMAP>
__global__ __launch_bounds__(256) void attn_rows_rc_tri_kernel(const char* __restrict__ Qimg,
                                                               const char* __restrict__ Kimg,
                                                               const char* __restrict__ Vtr,
                                                               const float* __restrict__ lse,
                                                               const long long* __restrict__ idx, int N, int NK, int M,
                                                               float scale, float* __restrict__ xds,
                                                               float* __restrict__ pmap, int ld) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  constexpr int NW = 4, D = kRcDepth;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int mrow = chunk * (32 * NW) + wave * 32 + lo;
  const bool mvalid = mrow < M;
  const int mc = mvalid ? mrow : M - 1;  // rows past M-1 recompute and rewrite row M-1's values (same bytes)
  const long row = idx[(long)b * M + mc];
  const float my_lse = lse[(long)b * N + row];
  const int qtiles = (N + kTile - 1) / kTile, ntiles = (NK + kTile - 1) / kTile;
  const char* Kb = Kimg + (long)b * ntiles * kTriTile;
  const char* Vb = Vtr + (long)b * ntiles * kTriTile;
  char* kring = smem_c;
  char* vring = smem_c + D * kTriTile;

  auto stage = [&](const char* img, char* ring, int t) {  // tile t -> slot t % D: 6 DMA pieces per thread
    const int tt = min(t, ntiles - 1);                      // past the end: the last tile again, unused
    char* lt = ring + (t % D) * kTriTile;
#pragma unroll
    for (int k = 0; k < 6; ++k)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(img + (long)tt * kTriTile + (tid + 256 * k) * 16),
          (__attribute__((address_space(3))) void*)(lt + (wave * 64 + 256 * k) * 16), 16, 0, 0);
  };
  stage(Kb, kring, 0);
  stage(Kb, kring, 1);
  stage(Vb, vring, 0);
  stage(Kb, kring, 2);
  u32x4 q[24];
  {
    const u32x4* qp = reinterpret_cast<const u32x4*>(Qimg + ((long)b * qtiles + (int)(row >> 5)) * kTriTile +
                                                     tri_rm_off((int)(row & 31), h, 0));
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      q[3 * ks] = qp[192 * ks];
      q[3 * ks + 1] = qp[192 * ks + 32];
      q[3 * ks + 2] = qp[192 * ks + 64];
    }
  }
  f32x16 oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) oacc[dt] = zero16();
  const int m0 = chunk * (32 * NW) + wave * 32;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  u32x4 qd[16];  // the sampled row as two fp16 planes under its own scale: the SAME conversion as in pass 1, so the
  float q_unscale;  // recomputed logits are the bits lse was formed from
  duo_q_from_tri(q, qd, q_unscale);
  const float q_scale = q_unscale * scale;  // (exact: q_unscale is a power of two)
  float sc_cur;
  f32x16 s_cur, s_nxt;
  stats_products<0>(kring, lo, h, qd, q_scale, s_cur, sc_cur);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // K slot 0 is restaged by iteration 0
  Tri bp[2];  // P^T fragments (two k-steps of 16 keys) of the tile whose P V is due: tile t-1; none yet
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) bp[ks] = Tri{u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};

  // iteration t = 0 .. ntiles: logits of tile t+1, P of tile t, P V of tile t-1 (t = 0: zeros against tile 0;
  // t = ntiles: P is all padding, the logits are not used)
  unsigned voff[6];  // byte offset of this thread's piece k inside a tile (loop-invariant)
#pragma unroll
  for (int k = 0; k < 6; ++k) voff[k] = (unsigned)(tid * 16 + 4096 * k);
  float pprev[16];  // P of the previous tile: its map rows leave under the NEXT iteration's MFMAs
#pragma unroll
  for (int r = 0; r < 16; ++r) pprev[r] = 0.f;
  char* xt = smem_c + 2 * D * kTriTile + wave * 4096;  // this wave's 32 x 32 fp32 transpose tile

  // Everything an iteration issues besides MFMAs is spread over its eight k-steps, so that it runs in the MFMAs'
  // shadow (stamped, tools/rc_stamps.py: at the top / bottom of the loop body the 12 DMA pieces cost 780 cycles, the
  // P tile's way out 640, the first operand fetch 500 -- of 6 100): operand reads first; two DMA pieces per k-step
  // in steps 0-5; the previous tile's P goes through the wave's LDS tile in steps 0 / 2 and out in steps 6 / 7.
  // LAST: the extra step after the last tile (only P V of tile ntiles-1); TAILK: tile t may hold padding keys
  auto step = [&](int t, auto last_c, auto tail_c) {
    constexpr bool LAST = decltype(last_c)::value;
    constexpr bool TAILK = decltype(tail_c)::value;
    RC_STAMP(0);
    const char* vt = vring + (max(t - 1, 0) % D) * kTriTile;
    const u32x4* lp = reinterpret_cast<const u32x4*>(kring + ((t + 1) % D) * kTriTile + tri_rm_off(lo, h, 0));
    auto fetch_k = [&](int ks) { return Tri{lp[192 * ks], lp[192 * ks + 32], u32x4{0, 0, 0, 0}}; };  // fp16 planes h, l
    // 2^-(e_q + e_k) of tile t+1 (its slot is not restaged before the next barrier)
    const float sc_nxt = q_scale * *reinterpret_cast<const float*>(kring + ((t + 1) % D) * kTriTile + kDuoScaleSlot);
    auto fetch_v = [&](int i) {  // step i: k-step i >> 2, channel block i & 3
      const char* ap = vt + tri_tr_off(32 * (i & 3) + lo, 2 * (i >> 2) + h, 0);
      return Tri{*reinterpret_cast<const u32x4*>(ap), *reinterpret_cast<const u32x4*>(ap + 2048),
                 *reinterpret_cast<const u32x4*>(ap + 4096)};
    };
    Tri k0 = fetch_k(0), k1 = fetch_k(1), v0 = fetch_v(0), v1 = fetch_v(1);
    // DMA piece i of K tile t+3 (slot of K tile t: its reads ended before the last barrier) and of V tile t+1
    // (slot of V tile t-2: likewise); past the end: the last tile again, unused
    const char* ksrc = uniform_ptr(Kb + (long)min(t + 3, ntiles - 1) * kTriTile);  // scalar base + 32-bit lane offset:
    const char* vsrc = uniform_ptr(Vb + (long)min(t + 1, ntiles - 1) * kTriTile);  // no 64-bit vector address arithmetic
    char* kdst = kring + ((t + 3) % D) * kTriTile + wave * 1024;
    char* vdst = vring + ((t + 1) % D) * kTriTile + wave * 1024;
    auto piece = [&](int k) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksrc + voff[k]),
                                       (__attribute__((address_space(3))) void*)(kdst + 4096 * k), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc + voff[k]),
                                       (__attribute__((address_space(3))) void*)(vdst + 4096 * k), 16, 0, 0);
    };
    float* pout = PMAP ? pmap + map_cloud(b, 0) * M * ld + max(t - 1, 0) * kTile + 4 * (lane & 7) : nullptr;
    f32x4 po[4];
    Tri bn[2];
    float p[16];
    s_nxt = zero16();
    RC_STAMP(1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      Tri k2 = k1, v2 = v1;
      if (i + 2 < 8) {
        k2 = fetch_k(i + 2);
        v2 = fetch_v(i + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!LAST) {
        if (SAMBLE_RC_ABL & 4) s_nxt[i] += __uint_as_float(k0.h[0] ^ qd[2 * i + 1][1] ^ k0.m[1]);
        else s_nxt = mfma_duo(k0.h, k0.m, qd[2 * i], qd[2 * i + 1], s_nxt);
      }
      if (SAMBLE_RC_ABL & 2) oacc[i & 3][i] += __uint_as_float(v0.h[0] ^ bp[i >> 2].l[1] ^ v0.m[1] ^ v0.l[2] ^ bp[i >> 2].h[1] ^ bp[i >> 2].m[1]);
      else oacc[i & 3] = mfma_tri(v0, bp[i >> 2], oacc[i & 3]);
      {  // slice i of the vector work on tile t: logits 2 i, 2 i + 1
#pragma clang fp contract(off)  // the statistics pass formed lse from round(s * scale): no fma here
        const int r0 = 2 * i, r1 = 2 * i + 1;
        float x0 = s_cur[r0], x1 = s_cur[r1];
        asm volatile("" : "+v"(x0), "+v"(x1));  // pins the slice inside this k-step's scheduling region (with the one below)
        const float e0 = __expf(x0 * sc_cur - my_lse), e1 = __expf(x1 * sc_cur - my_lse);
        p[r0] = LAST ? 0.f : (TAILK && t * kTile + crow(r0, h) >= NK) ? 0.f : e0;  // padding keys of the last tile
        p[r1] = LAST ? 0.f : (TAILK && t * kTile + crow(r1, h) >= NK) ? 0.f : e1;
        unsigned hh, mm, ll;
        tri_split2(p[r0], p[r1], hh, mm, ll);
        asm volatile("" : "+v"(hh), "+v"(mm), "+v"(ll));
        bn[i >> 2].h[i & 3] = hh;
        bn[i >> 2].m[i & 3] = mm;
        bn[i >> 2].l[i & 3] = ll;
      }
      if (!LAST && !(SAMBLE_RC_ABL & 1) && i < 6) piece(i);
      if (PMAP) {
        // the previous tile's P -> map rows as full 128-byte lines (8 lanes per row) through the wave's own 4 KB of
        // LDS; 16-byte block c of row r sits at block c ^ (r & 7): conflict-free both ways without padding.  LDS
        // operations of a wave execute in order: no wait between the writes and the reads.  (Iteration 0 writes
        // zeros over tile 0's place; iteration 1 overwrites them.)
        if (i == 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 o = {pprev[4 * g], pprev[4 * g + 1], pprev[4 * g + 2], pprev[4 * g + 3]};
            *reinterpret_cast<f32x4*>(xt + lo * 128 + (((2 * g + h) ^ (lo & 7)) << 4)) = o;
          }
        }
        if (i == 2) {
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
            if (SAMBLE_MAP_NT & 1) __builtin_nontemporal_store(po[k8], reinterpret_cast<f32x4*>(pout + (long)mr * ld));
            else *reinterpret_cast<f32x4*>(pout + (long)mr * ld) = po[k8];
          }
        }
      }
      // the weave: one MFMA, then its share of the step's vector instructions; memory operations float
#pragma unroll
      for (int m = 0; m < (LAST ? 6 : 9); ++m) {  // 3 logit + 6 P V products per k-step
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, LAST ? 6 : 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      RC_STAMP(2 + i);
      k0 = k1;
      k1 = k2;
      v0 = v1;
      v1 = v2;
    }
    // K tile t+2 and V tile t (their pieces went out in the PREVIOUS iteration's k-steps 0-5) must have landed before
    // anyone reads them.  Younger than those pieces: that iteration's 4 stores, this iteration's 12 pieces and 4 stores
    RC_STAMP(10);
#ifdef SAMBLE_STAMPS
    if (!LAST) {
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(PMAP ? 20 : 12) : "memory");
      RC_STAMP(11);
      asm volatile("s_barrier" ::: "memory");
    }
#else
    if (!LAST) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(PMAP ? 20 : 12) : "memory");
#endif
    RC_STAMP(12);
    s_cur = s_nxt;
    sc_cur = sc_nxt;
    bp[0] = bn[0];
    bp[1] = bn[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) pprev[r] = p[r];
  };
  for (int t = 0; t < ntiles - 1; ++t) step(t, std::false_type{}, std::false_type{});
  step(ntiles - 1, std::false_type{}, std::true_type{});
  step(ntiles, std::true_type{}, std::false_type{});  // P V of the last tile
  if (mvalid) {
    float* ob = xds + (long)b * 128 * M + mrow;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ob[(long)(32 * dt + crow(r, h)) * M] = oacc[dt][r];
    }
  }
}

}  // namespace samble

using namespace samble;

extern "C" size_t samble_tri_image_size(int B, int rows, int transposed) {
  const size_t tiles = (size_t)B * ((rows + 31) / 32);
  (void)transposed;
  return tiles * kTriTile;
}

extern "C" int samble_launch_tri_split(const float* src, long bs, long rs, int B, int rows, void* rm, void* tr,
                                       hipStream_t stream) {
  hipLaunchKernelGGL(tri_split_kernel, dim3((rows + 31) / 32, B), dim3(256), 0, stream, src, bs, rs, rows, (char*)rm,
                     (char*)tr);
  return (int)hipGetLastError();
}


extern "C" int samble_launch_attn_stats_tri(const void* qimg, const void* kimg, int B, int N, int nt, float scale,
                                            float* smap, int ld, float* lse, float* tok, const float* qn, const float* kn,
                                            hipStream_t stream) {
  {  // per call: cheap, and correct for every device / thread (no process-wide 'done' flag)
    for (const void* f : {reinterpret_cast<const void*>(attn_stats_tri_kernel<false>),
                          reinterpret_cast<const void*>(attn_stats_tri_kernel<true>)}) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
    }
  }
  const size_t lds = kStatsDepth * kTriTile + 8 * kTile * kStPadT * sizeof(float) + ((qn && kn) ? (size_t)ld * 4 : 0);
  if (lds > 160 * 1024) return -22;
  auto kern = (qn && kn) ? attn_stats_tri_kernel<true> : attn_stats_tri_kernel<false>;
  Timed timed(kT_attn_stats, stream);
  hipLaunchKernelGGL(kern, dim3((N + 255) / 256, B), dim3(512), lds, stream, (const char*)qimg, (const char*)kimg, N,
                     N + nt, scale, smap, ld, lse, tok, nt, qn, kn);
  return (int)hipGetLastError();
}

extern "C" int samble_launch_attn_rows_tri(const float* smap, int ld, const float* lse, const void* v_tr_image,
                                           const long long* idx, int B, int N, int nt, int M, float* xds,
                                           hipStream_t stream) {
  {  // per call: cheap, and correct for every device / thread (no process-wide 'done' flag)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_rows_tri_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kRowsLds);
    if (e != hipSuccess) return (int)e;
  }
  Timed timed(kT_attn_rows, stream);
  hipLaunchKernelGGL(attn_rows_tri_kernel, dim3((M + 127) / 128, B), dim3(256), kRowsLds, stream, smap, ld, lse,
                     (const char*)v_tr_image, idx, N, N + nt, M, xds);
  return (int)hipGetLastError();
}

namespace samble {
// A K row-image tile -> its LOGIT form, in place (tri_dev.h, "two fp16 planes for the logit products"): the tile's
// 32 x 128 values x 2^e as fp16 h / l planes in the h / m piece slots, 2^-e in the l slot of (group 0, row 0).
// One workgroup per tile; a thread owns two (row, channel group) chunk triples and rewrites only those.
__global__ __launch_bounds__(256) void tri_k_to_duo_kernel(char* __restrict__ img, int ntiles, int tile0) {
  __shared__ float wmax[4];
  char* tile = img + ((long)blockIdx.y * ntiles + tile0 + blockIdx.x) * kTriTile;
  const int tid = threadIdx.x;
  if (*reinterpret_cast<const unsigned*>(tile + kDuoScaleSlot + 4) == kDuoTag) return;  // converted already (uniform)
  float x[2][8];
  float amax = 0.f;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int p = tid + 256 * u, r = p & 31, g = p >> 5;
    const u32x4* c = reinterpret_cast<const u32x4*>(tile + tri_rm_off(r, g, 0));
    tri_chunk_values(c[0], c[32], c[64], x[u]);
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(x[u][e]));
  }
  amax = wave_max64(amax);
  if ((tid & 63) == 0) wmax[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  float sc, inv;
  duo_scale_for(amax, sc, inv);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int p = tid + 256 * u, r = p & 31, g = p >> 5;
    u32x4 hw, lw;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned a, b2;
      duo_split2(x[u][2 * w] * sc, x[u][2 * w + 1] * sc, a, b2);
      hw[w] = a;
      lw[w] = b2;
    }
    u32x4* c = reinterpret_cast<u32x4*>(tile + tri_rm_off(r, g, 0));
    c[0] = hw;
    c[32] = lw;
    if (p == 0) c[64] = u32x4{__float_as_uint(inv), kDuoTag, 0u, 0u};
  }
}
}  // namespace samble

// the K row image of (B, rows, 128) in place -> its logit form (what attn_stats_tri / attn_stats_nl_tri /
// attn_rows_rc_tri read); samble_launch_tri_split_qkv and samble_launch_proj_fwd_tri end with it
extern "C" int samble_launch_k_to_duo(void* kimg, int B, int rows, int tile0, hipStream_t stream) {
  const int ntiles = (rows + 31) / 32;  // tiles tile0 .. ntiles-1 of every cloud (the projection writes the full point
  if (tile0 >= ntiles) return 0;        // tiles in this form itself: only the token / ragged tiles are left to it)
  hipLaunchKernelGGL(tri_k_to_duo_kernel, dim3(ntiles - tile0, B), dim3(256), 0, stream, (char*)kimg, ntiles, tile0);
  return (int)hipGetLastError();
}

extern "C" int samble_launch_tri_split_qkv(const float* qkv, long bs, long rs, int B, int N, int nt, void* qimg, void* kimg,
                                           void* vimg, void* ktr, void* vrm, hipStream_t stream) {
  Timed timed(kT_tri_split, stream);
  hipLaunchKernelGGL(tri_split_qkv_kernel, dim3((N + nt + 31) / 32, B), dim3(256), 0, stream, qkv, bs, rs, N, N + nt,
                     (char*)qimg, (char*)kimg, (char*)vimg, (char*)ktr, (char*)vrm, 0);
  if (kimg) {
    const int rc = samble_launch_k_to_duo(kimg, B, N + nt, 0, stream);
    if (rc) return rc;
  }
  if (vrm) return samble_launch_k_to_duo(vrm, B, N + nt, 0, stream);  // (the backward's dP = dO V^T: same form)
  return (int)hipGetLastError();
}

// the same for the tiles tile0 .. only (the projection kernel writes the images of the full point tiles itself)
extern "C" int samble_launch_tri_split_qkv_tiles(const float* qkv, long bs, long rs, int B, int N, int nt, int tile0, void* qimg,
                                                 void* kimg, void* vimg, void* ktr, void* vrm, hipStream_t stream) {
  const int ntiles = (N + nt + 31) / 32;
  if (tile0 >= ntiles) return 0;
  hipLaunchKernelGGL(tri_split_qkv_kernel, dim3(ntiles - tile0, B), dim3(256), 0, stream, qkv, bs, rs, N, N + nt,
                     (char*)qimg, (char*)kimg, (char*)vimg, (char*)ktr, (char*)vrm, tile0);
  return (int)hipGetLastError();
}

#ifdef SAMBLE_STAMPS
extern "C" __attribute__((visibility("default"))) int samble_scratch_nl_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(samble::g_nl_stamps), sizeof(unsigned long long) * 128);
}
extern "C" __attribute__((visibility("default"))) int samble_scratch_rc_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(samble::g_rc_stamps), sizeof(unsigned long long) * 128);
}
#endif

// pass 1 without the map: lse, token logits and the K neighbour logits per row (nl (B, N, KN), ascending-index order)
// nn_sorted / acc_ws non-null: also accumulate the sparse_* score statistics of `score_mode` (score.hip's modes)
// into acc_ws = [colacc B*N u64][indeg B*N i32][rowstat B*N f32], after zeroing zero_bytes of it (0: the caller
// cleared it, samble_launch_nn_prepare does on request)
extern "C" int samble_launch_attn_stats_nl_tri(const void* qimg, const void* kimg, int B, int N, int nt, float scale,
                                               const unsigned* masks, int KN, float* nl, float* lse, float* tok,
                                               const int* nn_sorted, int score_mode, void* acc_ws, size_t zero_bytes,
                                               hipStream_t stream) {
  if (KN < 1 || KN > 32) return (int)hipErrorInvalidValue;
  const size_t lds = kStatsNlLds;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_stats_nl_tri_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  NlScoreArgs sc{nullptr, nullptr, nullptr, nullptr, 0};
  if (nn_sorted && acc_ws) {
    if (score_mode < 0 || score_mode > 4 || (size_t)N * 12 > (size_t)kStatsDepth * kTriTile) return (int)hipErrorInvalidValue;
    if (zero_bytes) {
      e = hipMemsetAsync(acc_ws, 0, zero_bytes, stream);
      if (e != hipSuccess) return (int)e;
    }
    unsigned long long* colacc = reinterpret_cast<unsigned long long*>(acc_ws);
    int* indeg = reinterpret_cast<int*>(colacc + (size_t)B * N);
    float* rowstat = reinterpret_cast<float*>(indeg + (size_t)B * N);
    sc = NlScoreArgs{nn_sorted, colacc, indeg, score_mode >= 3 ? rowstat : nullptr, score_mode == 4};
  }
  Timed timed(kT_attn_stats, stream);
  hipLaunchKernelGGL(attn_stats_nl_tri_kernel, dim3((N + 255) / 256, B), dim3(512), lds, stream, (const char*)qimg,
                     (const char*)kimg, N, N + nt, scale, masks, KN, nl, lse, tok, nt, sc);
  return (int)hipGetLastError();
}

// pass 2 without the map: x_ds of the M sampled rows, and (pmap != null) their P map (B, M, ld) for the backward
extern "C" int samble_launch_attn_rows_rc_tri(const void* qimg, const void* kimg, const void* v_tr_image,
                                              const float* lse, const long long* idx, int B, int N, int nt, int M,
                                              float scale, float* xds, float* pmap, int ld, hipStream_t stream) {
  for (const void* f : {reinterpret_cast<const void*>(attn_rows_rc_tri_kernel<false>),
                        reinterpret_cast<const void*>(attn_rows_rc_tri_kernel<true>)}) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kRcLds);
    if (e != hipSuccess) return (int)e;
  }
  Timed timed(kT_attn_rows, stream);
  if (pmap)
    hipLaunchKernelGGL(attn_rows_rc_tri_kernel<true>, dim3((M + 127) / 128, B), dim3(256), kRcLds, stream,
                       (const char*)qimg, (const char*)kimg, (const char*)v_tr_image, lse, idx, N, N + nt, M, scale, xds,
                       pmap, ld);
  else
    hipLaunchKernelGGL(attn_rows_rc_tri_kernel<false>, dim3((M + 127) / 128, B), dim3(256), kRcLds, stream,
                       (const char*)qimg, (const char*)kimg, (const char*)v_tr_image, lse, idx, N, N + nt, M, scale, xds,
                       pmap, ld);
  return (int)hipGetLastError();
}
