// Two-pass attention forward with the logit map kept in HBM (reference models/downsample.py:139-153
// and 242-252).
//
// Why materialise what flash attention avoids: on MI355X the path runs in exact fp32, where the
// matrix cores give 157 TFLOP/s against 8 TB/s of HBM3E, i.e. 20 flop per byte.  Recomputing one
// logit costs 2*D = 256 flop, reloading it costs 4 bytes = 80 flop-equivalents, and the 288 GB of
// HBM make the (B,N,N+nt) map (538 MB at B=32, N=2048; 4.3 GB for the N=8192 stress case) a
// non-issue.  So S = scale * Q K^T is computed ONCE:
//   pass 1  attn_stats   all N rows: S tiles -> HBM, running max / sum-exp -> lse, token logits
//   (score, bins, selection run on lse / S; they decide WHICH M rows are sampled)
//   pass 2  attn_rows    the M sampled rows only: P = exp(S - lse) straight from the map, O = P V,
//                        written channel-major as the module output x_ds (B, D, M)
//   backward             reads the same M rows again instead of recomputing S (attn_bwd.hip)
// Matrix work per cloud: 2N(N+nt)D + 2M(N+nt)D forward (= the algorithmic count of SURVEY 8d; the
// single-pass kernel of attn_fwd.hip executes 4N(N+nt)D), 4 products instead of 5 backward.
//
// Map layout: row-major (B, N, ld), ld = 32 * ceil((N+nt)/32); columns N..N+nt-1 are the token
// logits, columns >= N+nt hold -inf so consumers need no tail masks (exp(-inf - lse) = 0).
#include "samble_dev.h"

namespace samble {

// ------------------------------------------------------------------------------------------------
// pass 1: one workgroup = NW waves = 32*NW query rows; K tiles (32 keys) triple-buffered in LDS and
// fetched two tiles ahead.  S^T orientation (keys on the accumulator's register axis, queries on lanes):
// the row statistics are lane-local.  The S tile is transposed through a wave-private LDS tile and
// written to the map as full 128-byte lines (8 lanes per row).
// ------------------------------------------------------------------------------------------------
// Order of work inside a tile step: the 64 MFMAs of tile t+1's S product are issued BEFORE the softmax /
// map-store work of tile t, so the latency of their LDS operand reads and of the accumulator is not
// exposed (VALU work itself does not overlap fp32 MFMAs on gfx950: tools/micro/coissue_bench.hip; the
// gain measured for this ordering was 8%).  Tile t+1 is resident while tile t+2 is being staged.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifndef SAMBLE_MAP_STORE_AUX
#define SAMBLE_MAP_STORE_AUX 0
#endif
constexpr int kStoreAux = SAMBLE_MAP_STORE_AUX;  // cache policy of the map stores: 0 plain, 2 nt, 16 sc1, 18 nt sc1
constexpr int kStPad = 36;  // row stride (floats) of a wave's 32x32 transpose tile: 16-byte aligned, 9 x 16 B (odd)

template <bool TAIL, int ABL, bool L2>
__device__ __forceinline__ void stats_step(const float* __restrict__ Kn, int lo, int h, const float (&q)[64],
                                           f32x16& s_cur, f32x16& s_nxt, float scale, float* __restrict__ xt,
                                           float* __restrict__ gdst, __amdgpu_buffer_rsrc_t rsrc,
                                           const int (&roff)[4], int j0, int N, int NK,
                                           float* __restrict__ tokrow, float& m, float& l, float qb,
                                           const float (&kb)[16]) {
  const f32x4* lp = reinterpret_cast<const f32x4*>(Kn + lo * kLdsPad + 64 * h);
  const int lane = lo + 32 * h;
  float mt = kNegInf, ps = 0.f;
  s_nxt = zero16();
#pragma unroll
  for (int q4 = 0; q4 < 16; ++q4) {
    const f32x4 a = lp[q4];
#pragma unroll
    for (int e = 0; e < 4; ++e) s_nxt = mfma32(a[e], q[4 * q4 + e], s_nxt);
    if (q4 < 4) {  // scale, tile max; registers 4*q4 .. 4*q4+3 (4 consecutive keys of this lane's row) -> transpose tile
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * q4 + e;
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
      const f32x4 o = {s_cur[4 * q4], s_cur[4 * q4 + 1], s_cur[4 * q4 + 2], s_cur[4 * q4 + 3]};
      *reinterpret_cast<f32x4*>(xt + lo * kStPad + 8 * q4 + 4 * h) = o;
    } else if (q4 == 4) {  // running max (branch-free rescale of the running sum)
      mt = fmaxf(mt, wave_xor32(mt));
      const float mnew = fmaxf(m, mt);
      l *= __expf(m - mnew);
      m = mnew;
    } else if (q4 == 5) {
      // map store, row-major: 8 lanes cover one row's 32 keys (128 contiguous bytes), 8 rows per
      // instruction.  Written straight from the accumulator layout (lane = row, 16 bytes each) a store
      // instruction touches 64 different lines and the L1's per-line write requests delayed the K-tile
      // loads of the whole CU (20% of the kernel).
      if (ABL != 1) {
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
          const int row = (lane >> 3) + 8 * k8;
          const f32x4 o = *reinterpret_cast<const f32x4*>(xt + row * kStPad + 4 * (lane & 7));
          if (ABL == 3) *reinterpret_cast<f32x4*>(gdst - j0 + roff[k8]) = o;  // same lines every tile
          else if (ABL == 4) *reinterpret_cast<f32x4*>(gdst - j0 + (roff[k8] / 4) % 8192 * 4) = o;  // compact scratch
          else if (kStoreAux == 0) *reinterpret_cast<f32x4*>(gdst + roff[k8]) = o;
          else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsrc, (roff[k8] + j0) * 4, 0, kStoreAux);
        }
      }
    } else if (q4 < 14) {  // two exps per step
      ps += __expf(s_cur[2 * (q4 - 6)] - m);
      ps += __expf(s_cur[2 * (q4 - 6) + 1] - m);
    }
  }
  l += ps;
}

// ABL (timing-only ablations, wrong outputs): 1 = no map stores, 2 = no tile staging
template <int NW, int ABL = 0, bool L2 = false>
__global__ __launch_bounds__(64 * NW, 2) void attn_stats_kernel(const float* __restrict__ Q, long q_bs, long q_rs,
                                                                const float* __restrict__ K, long k_bs, long k_rs,
                                                                int N, int NK, float scale, float* __restrict__ smap,
                                                                int ld, float* __restrict__ lse,
                                                                float* __restrict__ tok, int nt,
                                                                const float* __restrict__ qn,
                                                                const float* __restrict__ kn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kBuf = kTile * kLdsPad;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  // rows past N (last workgroup of a ragged cloud) are clamped to row N-1: those lanes recompute and
  // rewrite row N-1's values bit for bit, which keeps every store unpredicated -- the loop body is
  // one basic block, so the compiler can wait for the staged tile with a COUNTED vmcnt instead of
  // draining the map stores too (stores share vmcnt on gfx9; the drain cost 20% of this kernel)
  const int qrow = min(chunk * (32 * NW) + wave * 32 + lo, N - 1);
  const float* Kb = K + (long)b * k_bs;

  float q[64];
  load_row_half(Q + (long)b * q_bs + (long)qrow * q_rs, h, q);
  // rows of this wave, clamped like qrow: row index of the wave's first row, then min() per row
  const int row0 = chunk * (32 * NW) + wave * 32;
  float* xt = smem + 3 * kBuf + wave * (kTile * kStPad);
  float* tokrow = tok + ((long)b * N + qrow) * nt;
  float m = kNegInf, l = 0.f;

  const int ntiles = (NK + kTile - 1) / kTile;
  TileRegsT<64 * NW> kr;
  tile_load_issue(kr, Kb, k_rs, 0, NK, tid);
  tile_store_lds(kr, smem, kLdsPad, tid);
  tile_load_issue(kr, Kb, k_rs, kTile, NK, tid);  // rows >= NK read as zeros
  tile_store_lds(kr, smem + kBuf, kLdsPad, tid);
  __syncthreads();
  f32x16 s_cur = mma_rows_x_regs(smem, kLdsPad, lo, h, q, zero16());
  f32x16 s_nxt;
  // l2 scoring: |q_i|^2 x scale of this lane's row and |k_j|^2 x scale of its 16 keys per tile (kn is
  // zero-padded to ld columns by the caller), the next tile's fetched one iteration ahead
  float qb = 0.f, kb_cur[16], kb_nxt[16];
  const float* knb = L2 ? kn + (long)b * ld + 4 * h : nullptr;
  auto load_kb = [&](int tile, float (&dst)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(knb + tile * kTile + 8 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[4 * g + e] = v4[e] * scale;
    }
  };
  if (L2) {
    qb = qn[(long)b * N + qrow] * scale;
    load_kb(0, kb_cur);
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) kb_cur[r] = kb_nxt[r] = 0.f;
  }

  // destination of the transposed store: lane L writes rows (L>>3)+8k of the wave's 32; rows past N-1
  // carry row N-1's values (clamped q above) and are folded onto row N-1
  int roff[4];
#pragma unroll
  for (int k8 = 0; k8 < 4; ++k8) roff[k8] = min(row0 + (lane >> 3) + 8 * k8, N - 1) * ld + 4 * (lane & 7);
  float* gdst = smap + (long)b * N * ld;
  // the same cloud as a buffer resource (wave-uniform by construction), for stores with a cache policy
  const unsigned long long gaddr = reinterpret_cast<unsigned long long>(gdst);
  const unsigned ghi = (unsigned)__builtin_amdgcn_readfirstlane((int)(gaddr >> 32));
  const unsigned glo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)gaddr);
  float* gu = reinterpret_cast<float*>(((unsigned long long)ghi << 32) | (unsigned long long)glo);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(gu, 0, (unsigned)((long)N * ld * 4), 0x00020000);
  int cur = 0;  // LDS buffer of tile t
  // K tiles are fetched TWO iterations ahead (registers kr2 / kr, alternating) and committed to LDS one
  // iteration before use: with the 538 MB map streaming out through the L2 the tile loads take longer
  // than one iteration (tools/ablate_attn_stats.py), and a stalled tile stalls all 8 waves at the barrier.
  TileRegsT<64 * NW> kr2;
  tile_load_issue(kr, Kb, k_rs, 2 * kTile, NK, tid);  // tile 2 in flight
  // main loop: pairs of full point tiles whose prefetches (up to tile t+4) are full tiles too
  const int n_main = max(min(N / kTile, ntiles - 4), 0) & ~1;
  int t = 0;
  for (; t < n_main; t += 2) {
    {
      const int nxt = (cur == 2) ? 0 : cur + 1, nn2 = (nxt == 2) ? 0 : nxt + 1;
      const int j0 = t * kTile;
      if (ABL != 2) tile_load_issue(kr2, Kb, k_rs, j0 + 3 * kTile, NK, tid);
      if (L2) load_kb(t + 1, kb_nxt);
      stats_step<false, ABL, L2>(smem + nxt * kBuf, lo, h, q, s_cur, s_nxt, scale, xt, gdst + j0, rsrc, roff, j0, N, NK, tokrow, m, l, qb, kb_cur);
      if (ABL != 2) tile_store_lds(kr, smem + nn2 * kBuf, kLdsPad, tid);
      __syncthreads();
      s_cur = s_nxt;
      cur = nxt;
      if (L2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) kb_cur[r] = kb_nxt[r];
      }
    }
    {
      const int nxt = (cur == 2) ? 0 : cur + 1, nn2 = (nxt == 2) ? 0 : nxt + 1;
      const int j0 = (t + 1) * kTile;
      if (ABL != 2) tile_load_issue(kr, Kb, k_rs, j0 + 3 * kTile, NK, tid);
      if (L2) load_kb(t + 2, kb_nxt);
      stats_step<false, ABL, L2>(smem + nxt * kBuf, lo, h, q, s_cur, s_nxt, scale, xt, gdst + j0, rsrc, roff, j0, N, NK, tokrow, m, l, qb, kb_cur);
      if (ABL != 2) tile_store_lds(kr2, smem + nn2 * kBuf, kLdsPad, tid);
      __syncthreads();
      s_cur = s_nxt;
      cur = nxt;
      if (L2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) kb_cur[r] = kb_nxt[r];
      }
    }
  }
  const int t_end = t;
  for (; t < ntiles; ++t) {  // the last point tiles, token keys and padding: fetch distance 1 again
    const int nxt = (cur == 2) ? 0 : cur + 1, nn2 = (nxt == 2) ? 0 : nxt + 1;
    const int j0 = t * kTile;
    if (t > t_end && t + 2 < ntiles) tile_load_issue(kr, Kb, k_rs, j0 + 2 * kTile, NK, tid);  // t_end + 2 is in kr already
    if (L2 && t + 1 < ntiles) load_kb(t + 1, kb_nxt);
    if (j0 + kTile > N)
      stats_step<true, ABL, L2>(smem + nxt * kBuf, lo, h, q, s_cur, s_nxt, scale, xt, gdst + j0, rsrc, roff, j0, N, NK, tokrow, m, l, qb, kb_cur);
    else
      stats_step<false, ABL, L2>(smem + nxt * kBuf, lo, h, q, s_cur, s_nxt, scale, xt, gdst + j0, rsrc, roff, j0, N, NK, tokrow, m, l, qb, kb_cur);
    if (t + 2 < ntiles) tile_store_lds(kr, smem + nn2 * kBuf, kLdsPad, tid);
    __syncthreads();
    s_cur = s_nxt;
    cur = nxt;
    if (L2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) kb_cur[r] = kb_nxt[r];
    }
  }
  const float ltot = l + wave_xor32(l);
  if (h == 0) lse[(long)b * N + qrow] = m + __logf(ltot);
}

// ------------------------------------------------------------------------------------------------
// pass 2: one workgroup = NW waves = 32*NW SAMPLED rows of one cloud; V tiles (32 keys) double-
// buffered in LDS.  Lane (i, h) reads its row's 4 x 16 bytes of the map per tile (prefetched one
// tile ahead), P = exp(S - lse) lands in the accumulator layout the product wants (reduced index =
// keys on the register axis) and feeds O^T += V_tile^T P^T (64 MFMA).  O^T registers are channels
// x queries-on-lanes: the channel-major module output (B, D, M) is written with 128-byte runs.
// ------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_rows_kernel(const float* __restrict__ smap, int ld,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ V, long v_bs, long v_rs,
                                                               const long long* __restrict__ idx, int N, int NK,
                                                               int M, float* __restrict__ xds) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kBuf = kTile * 128;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int mrow = chunk * (32 * NW) + wave * 32 + lo;
  const bool mvalid = mrow < M;
  const long row = idx[(long)b * M + (mvalid ? mrow : M - 1)];
  const float my_lse = lse[(long)b * N + row];
  const float* srow = smap + ((long)b * N + row) * ld + 4 * h;
  const float* Vb = V + (long)b * v_bs;

  f32x16 oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) oacc[dt] = zero16();

  const int ntiles = (NK + kTile - 1) / kTile;
  TileRegsT<64 * NW> vr;
  f32x4 sv[4], sn[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) sv[g] = *reinterpret_cast<const f32x4*>(srow + 8 * g);
  tile_load_issue(vr, Vb, v_rs, 0, NK, tid);
  tile_store_lds(vr, smem, 128, tid);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    float* Vc = smem + (t & 1) * kBuf;
    float* Vn = smem + ((t & 1) ^ 1) * kBuf;
    const int j0 = t * kTile;
    if (t + 1 < ntiles) {
      tile_load_issue(vr, Vb, v_rs, j0 + kTile, NK, tid);
#pragma unroll
      for (int g = 0; g < 4; ++g) sn[g] = *reinterpret_cast<const f32x4*>(srow + j0 + kTile + 8 * g);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __expf(sv[r >> 2][r & 3] - my_lse);
      // O^T tile dt, row rho <-> channel 4 rho + dt: lane lo's four A operands are one 16-byte LDS read
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(Vc + crow(r, h) * 128 + 4 * lo);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oacc[dt] = mfma32(a4[dt], p, oacc[dt]);
    }
    if (t + 1 < ntiles) {
      tile_store_lds(vr, Vn, 128, tid);
#pragma unroll
      for (int g = 0; g < 4; ++g) sv[g] = sn[g];
    }
    __syncthreads();
  }
  if (mvalid) {
    float* ob = xds + (long)b * 128 * M + mrow;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ob[(long)(4 * crow(r, h) + dt) * M] = oacc[dt][r];
    }
  }
}

}  // namespace samble

using namespace samble;

extern "C" int samble_attn_map_ld(int N, int nt) { return 32 * ((N + nt + 31) / 32); }

extern "C" int samble_launch_attn_stats(const float* Q, long q_bs, long q_rs, const float* K, long k_bs, long k_rs, int B,
                                        int N, int nt, float scale, float* smap, int ld, float* lse, float* tok,
                                        const float* qn, const float* kn, hipStream_t stream) {
  constexpr int NW = 8;
  const size_t lds = (3 * kTile * kLdsPad + NW * kTile * kStPad) * sizeof(float);
  auto kern = (qn && kn) ? attn_stats_kernel<NW, 0, true> : attn_stats_kernel<NW, 0>;
  Timed timed(kT_attn_stats, stream);
  hipLaunchKernelGGL(kern, dim3((N + 32 * NW - 1) / (32 * NW), B), dim3(64 * NW), lds, stream, Q, q_bs, q_rs, K, k_bs,
                     k_rs, N, N + nt, scale, smap, ld, lse, tok, nt, qn, kn);
  return (int)hipGetLastError();
}

extern "C" int samble_launch_attn_rows(const float* smap, int ld, const float* lse, const float* V, long v_bs, long v_rs,
                                       const long long* idx, int B, int N, int nt, int M, float* xds,
                                       hipStream_t stream) {
  constexpr int NW = 4;
  const size_t lds = 2 * kTile * 128 * sizeof(float);
  Timed timed(kT_attn_rows, stream);
  hipLaunchKernelGGL(attn_rows_kernel<NW>, dim3((M + 32 * NW - 1) / (32 * NW), B), dim3(64 * NW), lds, stream, smap, ld,
                     lse, V, v_bs, v_rs, idx, N, N + nt, M, xds);
  return (int)hipGetLastError();
}
