// Flash-style attention backward for the SAMBLE sampler.  Autograd of the reference
// (models/downsample.py:139-147 + 242-252) reduces exactly to an M-query cross-attention over the
// N point keys + nt token keys (SURVEY.md section 3 iii): only the M sampled rows of the attention
// map receive gradient.  With S = scale * Q K^T, P = exp(S - lse), dP = dO V^T,
// delta_i = sum_d dO_id O_id, dS = P o (dP - delta):
//     dQ_i = scale * sum_j dS_ij K_j      dK_j = scale * sum_i dS_ij Q_i      dV_j = sum_i P_ij dO_i
//
// Four kernels, all deterministic (no float atomics):
//   bwd_prep        gather the sampled Q rows / lse, transpose the upstream gradient (B,D,M)->(B,M,D),
//                   delta = rowsum(dO o O_sampled); while the rows are in registers, also P / dS
//                   against the nt (<= 8) token keys -> per-workgroup partial dK_tok / dV_tok
//   bwd_dq          query-stationary: one wave = 32 sampled rows, K/V stream through LDS; recomputes
//                   S and dP per tile (2x64 MFMA) and accumulates dQ^T (64 MFMA); scatters rows to dQ
//   bwd_dkdv        key-stationary over the N POINT keys: one wave = 32 keys with K,V rows and the
//                   dK^T/dV^T accumulators in registers; sampled Q / dO tiles stream through LDS
//                   (S, dP, dV^T, dK^T = 4x64 MFMA per tile)
//   bwd_tokens_reduce  fixed-order sum of those partials (token keys stay out of the MFMA grids,
//                   which keeps them a whole number of rounds: N/32 key waves per cloud, not N/32 + 1)
#include "tri_dev.h"

#ifndef SAMBLE_PREP_NT
#define SAMBLE_PREP_NT 1  // the dQ clear of bwd_prep_tri: read again only by the projection backward, three map kernels later (A/B -0.7 %)
#endif

namespace samble {

// ------------------------------------------------------------------------------------------------
// prep: one workgroup = 32 sampled rows of one cloud
// ------------------------------------------------------------------------------------------------
// The same preparation for the split-bf16 backward (one workgroup = one tile of 32 sampled rows), laid out so that
// nothing waits on a chain of dependent gathers or on cross-lane reductions: the dO^T and Q^T tiles sit in LDS
// ([channel][row], 33-float rows), all 32 row gathers are in flight together, the (row, token) logits are one
// thread each (32 x 8 = the workgroup), and every token-gradient partial is owned by one thread (fixed order).
// Outputs as bwd_prep_kernel with dO_rm / dO_tr / Q_tr given: lse_s, delta, tok_part, (L2) cs_part, the images.
#ifdef SAMBLE_STAMPS  // scratch builds only (tools/scratch, tools/prep_stamps.py): first and last workgroup of the grid
__device__ unsigned long long g_prep_stamps[32];
#define PSTAMP(i) do { if (threadIdx.x == 0) { if (blockIdx.x == 0 && blockIdx.y == 0) g_prep_stamps[i] = __builtin_amdgcn_s_memtime(); \
  if (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1) g_prep_stamps[16 + i] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif
template <bool L2>
__global__ __launch_bounds__(256) void bwd_prep_tri_kernel(const float* __restrict__ Q, long q_bs, long q_rs,
                                                           const float* __restrict__ K, long k_bs, long k_rs,
                                                           const float* __restrict__ V, long v_bs, long v_rs,
                                                           const float* __restrict__ Oc, const float* __restrict__ lse,
                                                           const long long* __restrict__ idx,
                                                           const float* __restrict__ g, int N, int nt, int M,
                                                           float scale, float* __restrict__ lse_s,
                                                           float* __restrict__ delta, float* __restrict__ tok_part,
                                                           float* __restrict__ cs_part, char* __restrict__ dO_rm,
                                                           char* __restrict__ dO_tr, char* __restrict__ Q_tr,
                                                           float* __restrict__ dQ, long dq_bs, long dq_rs,
                                                           unsigned* __restrict__ ds_amax) {
  // ds_amax != null: the two TRANSPOSED images leave as two fp16 planes per tile under the tile's own power-of-two
  // scale (tri_dev.h, "products that accumulate over tiles": what bwd_kacc_tri / bwd_kacc_pm_tri read), and the cloud's
  // slot of ds_amax (largest |dS|, raised by the dQ kernel that runs next) is cleared here
  __shared__ float gt[128 * 33];  // dO^T tile
  __shared__ float qt[128 * 33];  // Q^T tile of the gathered rows
  // 40 704 bytes of LDS in all, so that four workgroups fit a CU and the grid's 4 x 256 workgroups are one round
  __shared__ __attribute__((aligned(16))) float tk[8][128];  // the token keys, later the token values
  __shared__ float ps[32][9], dss[32][9];  // P and dS of (row, token); ps doubles as the delta partials before that
  float (*dred)[32] = reinterpret_cast<float (*)[32]>(&ps[0][0]);  // [8][32]
  __shared__ float lrow_s[32], delta_s[32];
  __shared__ long long rows[32];
  const int b = blockIdx.y, m0 = blockIdx.x * 32, tid = threadIdx.x;
  const float* gb = g + (long)b * 128 * M;
  const float* ob = Oc + (long)b * 128 * M;
  PSTAMP(0);
  if (dQ) {
    // the rows of dQ that were not sampled carry no gradient: this workgroup clears its share of the cloud's N rows
    // (all of them: the dQ kernel, which runs after this one, then writes the sampled rows) -- was a launch of its own
    const int per = (N + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * per, r1 = min(N, r0 + per);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < (r1 - r0) * 32; e += 256)
      if (SAMBLE_PREP_NT) __builtin_nontemporal_store(z4, reinterpret_cast<f32x4*>(dQ + (long)b * dq_bs + (long)(r0 + (e >> 5)) * dq_rs + 4 * (e & 31)));
      else *reinterpret_cast<f32x4*>(dQ + (long)b * dq_bs + (long)(r0 + (e >> 5)) * dq_rs + 4 * (e & 31)) = z4;
  }
  if (tid < 32) rows[tid] = idx[(long)b * M + min(m0 + tid, M - 1)];
  f32x4 vtok = {0.f, 0.f, 0.f, 0.f};  // this thread's four token-value words: into tk once the keys are done with
  {
    const int t = tid >> 5, c = 4 * (tid & 31);
    f32x4 ktok = {0.f, 0.f, 0.f, 0.f};
    if (t < nt) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {  // (row strides need not be multiples of four)
        ktok[u] = K[(long)b * k_bs + (long)(N + t) * k_rs + c + u];
        vtok[u] = V[(long)b * v_bs + (long)(N + t) * v_rs + c + u];
      }
    }
    *reinterpret_cast<f32x4*>(&tk[t][c]) = ktok;
  }
  // dO tile and delta = sum_c dO O: each thread 16 channels of one row, the 8 channel groups in a fixed order below
  float dpart = 0.f;
  {
    const int mm = tid & 31;
    const bool ok = m0 + mm < M;
    float gv[16], ov[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int d = (tid >> 5) + 8 * i;
      gv[i] = ok ? gb[(long)d * M + m0 + mm] : 0.f;
      ov[i] = ok ? ob[(long)d * M + m0 + mm] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      gt[((tid >> 5) + 8 * i) * 33 + mm] = gv[i];
      dpart = fmaf(gv[i], ov[i], dpart);
    }
  }
  dred[tid >> 5][tid & 31] = dpart;
  __syncthreads();
  PSTAMP(1);
  {  // the 32 row gathers: every half-wave has its 4 rows in flight at once
    const int sub = tid >> 5, l32 = tid & 31;
    f32x4 qv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = sub + 8 * u;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      qv[u] = (m0 + rr < M) ? *reinterpret_cast<const f32x4*>(Q + (long)b * q_bs + rows[rr] * q_rs + 4 * l32) : z4;
    }
    float lr = 0.f;
    if (tid < 32) lr = lse[(long)b * N + rows[tid]];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = sub + 8 * u;
#pragma unroll
      for (int c = 0; c < 4; ++c) qt[(4 * l32 + c) * 33 + rr] = qv[u][c];
    }
    if (tid < 32) {
      float part = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) part += dred[w8][tid];
      lrow_s[tid] = lr;
      delta_s[tid] = part;
      if (m0 + tid < M) {
        delta[(long)b * M + m0 + tid] = part;
        lse_s[(long)b * M + m0 + tid] = lr;
      }
    }
  }
  __syncthreads();
  PSTAMP(2);
  const int ntiles_m = gridDim.x;
  {  // operand images of the tile (layouts: tri_dev.h); rows past M-1 are zeros in both tiles.
    // The ROW image of dO (the dQ kernels' dP = dO V^T) always leaves as two fp16 planes under the tile's power-of-two
    // scale (tri_dev.h, the logit form); the two TRANSPOSED images likewise when ds_amax is given, else as three planes.
    char* irm = dO_rm + ((long)b * ntiles_m + blockIdx.x) * kTriTile;
    char* itr = dO_tr + ((long)b * ntiles_m + blockIdx.x) * kTriTile;
    char* qtr = Q_tr + ((long)b * ntiles_m + blockIdx.x) * kTriTile;
    __shared__ float amx[2][4];
    if (ds_amax && blockIdx.x == 0 && tid == 0) ds_amax[b] = 0u;
    float ax = 0.f, ay = 0.f;
    for (int e = tid; e < 128 * 32; e += 256) {
      ax = fmaxf(ax, fabsf(gt[(e >> 5) * 33 + (e & 31)]));
      ay = fmaxf(ay, fabsf(qt[(e >> 5) * 33 + (e & 31)]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ax = fmaxf(ax, __shfl_xor(ax, o, 64));
      ay = fmaxf(ay, __shfl_xor(ay, o, 64));
    }
    if ((tid & 63) == 0) {
      amx[0][tid >> 6] = ax;
      amx[1][tid >> 6] = ay;
    }
    __syncthreads();
    ax = fmaxf(fmaxf(amx[0][0], amx[0][1]), fmaxf(amx[0][2], amx[0][3]));
    ay = fmaxf(fmaxf(amx[1][0], amx[1][1]), fmaxf(amx[1][2], amx[1][3]));
    float sx, ix, sy, iy;
    duo_scale_for(ax, sx, ix);
    duo_scale_for(ay, sy, iy);
    auto duo8 = [&](const float (&v)[8], float sc, u32x4& hw, u32x4& lw) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        unsigned a1, a2;
        duo_split2(v[2 * w] * sc, v[2 * w + 1] * sc, a1, a2);
        hw[w] = a1;
        lw[w] = a2;
      }
    };
    for (int e = tid; e < 512; e += 256) {
      {
        const int r = e & 31, gq = e >> 5;
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = gt[(8 * gq + i) * 33 + r];
        u32x4 hw, lw;
        duo8(x, sx, hw, lw);
        *reinterpret_cast<u32x4*>(irm + tri_rm_off(r, gq, 0)) = hw;
        *reinterpret_cast<u32x4*>(irm + tri_rm_off(r, gq, 1)) = lw;
        if (e == 0) *reinterpret_cast<u32x4*>(irm + kDuoScaleSlot) = u32x4{__float_as_uint(ix), kDuoTag, 0u, 0u};
      }
      const int d = e & 127, cg = e >> 7;
      float x[8], y[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int o = d * 33 + 16 * (cg >> 1) + 8 * (i >> 2) + 4 * (cg & 1) + (i & 3);
        x[i] = gt[o];
        y[i] = qt[o];
      }
      if (ds_amax) {  // (uniform)
        u32x4 xh, xl, yh, yl;
        duo8(x, sx, xh, xl);
        duo8(y, sy, yh, yl);
        *reinterpret_cast<u32x4*>(itr + tri_tr_off(d, cg, 0)) = xh;
        *reinterpret_cast<u32x4*>(itr + tri_tr_off(d, cg, 1)) = xl;
        *reinterpret_cast<u32x4*>(qtr + tri_tr_off(d, cg, 0)) = yh;
        *reinterpret_cast<u32x4*>(qtr + tri_tr_off(d, cg, 1)) = yl;
        if (e == 0) {
          *reinterpret_cast<u32x4*>(itr + kDuoTrScaleSlot) = u32x4{__float_as_uint(ix), kDuoTag, 0u, 0u};
          *reinterpret_cast<u32x4*>(qtr + kDuoTrScaleSlot) = u32x4{__float_as_uint(iy), 0u, 0u, 0u};
        }
      } else {
        const Tri t3 = tri_split8(x), u3 = tri_split8(y);
        *reinterpret_cast<u32x4*>(itr + tri_tr_off(d, cg, 0)) = t3.h;
        *reinterpret_cast<u32x4*>(itr + tri_tr_off(d, cg, 1)) = t3.m;
        *reinterpret_cast<u32x4*>(itr + tri_tr_off(d, cg, 2)) = t3.l;
        *reinterpret_cast<u32x4*>(qtr + tri_tr_off(d, cg, 0)) = u3.h;
        *reinterpret_cast<u32x4*>(qtr + tri_tr_off(d, cg, 1)) = u3.m;
        *reinterpret_cast<u32x4*>(qtr + tri_tr_off(d, cg, 2)) = u3.l;
      }
    }
  }
  PSTAMP(3);
  if (nt <= 0) return;  // (uniform)
  {  // token logits: thread = (row r, token t)
    const int r = tid & 31, t = tid >> 5;
    float st = 0.f, dpt = 0.f, qq = 0.f, kk = 0.f;
    // the token words as 16-byte broadcast reads; channels in order (one fma chain per sum, as before)
#pragma unroll 4
    for (int c4 = 0; c4 < 32; ++c4) {
      const f32x4 k4 = *reinterpret_cast<const f32x4*>(&tk[t][4 * c4]);
      float q4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q4[u] = qt[(4 * c4 + u) * 33 + r];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        st = fmaf(q4[u], k4[u], st);
        if (L2) {
          qq = fmaf(q4[u], q4[u], qq);
          kk = fmaf(k4[u], k4[u], kk);
        }
      }
    }
    __syncthreads();
    *reinterpret_cast<f32x4*>(&tk[tid >> 5][4 * (tid & 31)]) = vtok;
    __syncthreads();
#pragma unroll 4
    for (int c4 = 0; c4 < 32; ++c4) {
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(&tk[t][4 * c4]);
      float g4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) g4[u] = gt[(4 * c4 + u) * 33 + r];
#pragma unroll
      for (int u = 0; u < 4; ++u) dpt = fmaf(g4[u], v4[u], dpt);
    }
    if (L2) st = 2.f * st - qq - kk;  // -|q - k_tok|^2
    const bool ok = t < nt && m0 + r < M;
    const float p = ok ? __expf(st * scale - lrow_s[r]) : 0.f;
    ps[r][t] = p;
    dss[r][t] = p * (dpt - delta_s[r]) * scale;
  }
  __syncthreads();
  PSTAMP(4);
  float* outp = tok_part + ((long)b * gridDim.x + blockIdx.x) * 2 * 8 * 128;
  // [dK | dV][token][channel]: dK_t += dS_rt q_r, dV_t += P_rt dO_r, rows ascending.  Thread = channel tid & 127 and the
  // tokens (tid >> 7) + 2 i: the channel's 32 row values are read once per tile and serve its four tokens
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const int c = tid & 127;
    const float* x = which ? &gt[c * 33] : &qt[c * 33];
    float xr[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) xr[r] = x[r];
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 32; ++r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = (tid >> 7) + 2 * i;
        a[i] = fmaf((which ? ps : dss)[r][t], xr[r], a[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) outp[which * 1024 + ((tid >> 7) + 2 * i) * 128 + c] = a[i];
  }
  PSTAMP(5);
  if (L2 && tid < 8) {  // column sums of dS over this workgroup's rows, per token
    float a = 0.f;
    for (int r = 0; r < 32; ++r) a += dss[r][tid];
    cs_part[((long)b * gridDim.x + blockIdx.x) * 8 + tid] = a;
  }
}

template <bool L2>  // L2: token logits are -|q-k|^2 and the dS column sums of the token keys are produced
__global__ __launch_bounds__(256) void bwd_prep_kernel(const float* __restrict__ Q, long q_bs, long q_rs,
                                                       const float* __restrict__ K, long k_bs, long k_rs,
                                                       const float* __restrict__ V, long v_bs, long v_rs,
                                                       const float* __restrict__ O, const float* __restrict__ Oc,
                                                       const float* __restrict__ lse,
                                                       const long long* __restrict__ idx,
                                                       const float* __restrict__ g,  // (B,128,M)
                                                       int N, int nt, int M, float scale, float* __restrict__ Qs,
                                                       float* __restrict__ dO, float* __restrict__ lse_s,
                                                       float* __restrict__ delta, float* __restrict__ tok_part,
                                                       float* __restrict__ slab, int nslab, int tok_slab, int l2,
                                                       float* __restrict__ cs_part, char* __restrict__ dO_rm,
                                                       char* __restrict__ dO_tr, char* __restrict__ Q_tr) {
  // dO_rm / dO_tr / Q_tr != null (split-bf16 backward): the operand images of this tile of 32 sampled rows
  // are written from the LDS tiles right here, and the fp32 copies Qs / dO are not
  __shared__ float gt[128 * 33];
  __shared__ float red[4][2][8][128];  // [wave][dK|dV][token][channel]
  __shared__ float dred[8][32];
  __shared__ float csred[8][8];
  const int b = blockIdx.y, m0 = blockIdx.x * 32, tid = threadIdx.x;
  const float* gb = g + (long)b * 128 * M;
  // O either as rows of the all-rows forward output (O, gathered by idx below) or as the sampled rows'
  // channel-major output of attn_rows (Oc = x_ds, (B,128,M)): then delta is reduced right here, each
  // thread over its 16 channels of one row, the 8 channel groups in a fixed order afterwards
  float dpart = 0.f;
  for (int e = tid; e < 128 * 32; e += 256) {
    int d = e >> 5, mm = e & 31;
    const float gv = (m0 + mm < M) ? gb[(long)d * M + m0 + mm] : 0.f;
    gt[d * 33 + mm] = gv;
    if (Oc && m0 + mm < M) dpart = fmaf(gv, Oc[(long)b * 128 * M + (long)d * M + m0 + mm], dpart);
  }
  dred[tid >> 5][tid & 31] = dpart;
  float* qt = &red[0][0][0][0];  // [channel][33]: the gathered Q rows, transposed (red is not in use before the end)
  if (Q_tr)
    for (int e = tid; e < 128 * 33; e += 256) qt[e] = 0.f;
  __syncthreads();
  const int ntiles_m = gridDim.x;
  if (dO_rm) {  // dO tile -> row image and transposed image (layouts: tri_dev.h), rows past M-1 are zeros in gt
    char* irm = dO_rm + ((long)b * ntiles_m + blockIdx.x) * kTriTile;
    char* itr = dO_tr + ((long)b * ntiles_m + blockIdx.x) * kTriTile;
    for (int e = tid; e < 512; e += 256) {
      {
        const int r = e & 31, gq = e >> 5;
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = gt[(8 * gq + i) * 33 + r];
        const Tri t3 = tri_split8(x);
        *reinterpret_cast<u32x4*>(irm + tri_rm_off(r, gq, 0)) = t3.h;
        *reinterpret_cast<u32x4*>(irm + tri_rm_off(r, gq, 1)) = t3.m;
        *reinterpret_cast<u32x4*>(irm + tri_rm_off(r, gq, 2)) = t3.l;
      }
      {
        const int d = e & 127, cg = e >> 7;
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = gt[d * 33 + 16 * (cg >> 1) + 8 * (i >> 2) + 4 * (cg & 1) + (i & 3)];
        const Tri t3 = tri_split8(x);
        *reinterpret_cast<u32x4*>(itr + tri_tr_off(d, cg, 0)) = t3.h;
        *reinterpret_cast<u32x4*>(itr + tri_tr_off(d, cg, 1)) = t3.m;
        *reinterpret_cast<u32x4*>(itr + tri_tr_off(d, cg, 2)) = t3.l;
      }
    }
  }
  const int sub = tid >> 5, l32 = tid & 31;  // 8 half-waves, each one row at a time
  // token keys / values: this lane's 4 channels of each of the nt (<= 8) rows
  f32x4 kt[8], vt[8], ak[8], av[8];
  float ktt[8], acs[8];  // l2 scoring: |k_tok|^2; column sums of dS over this half-wave's rows
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    kt[t] = (t < nt) ? *reinterpret_cast<const f32x4*>(K + (long)b * k_bs + (long)(N + t) * k_rs + 4 * l32) : z4;
    vt[t] = (t < nt) ? *reinterpret_cast<const f32x4*>(V + (long)b * v_bs + (long)(N + t) * v_rs + 4 * l32) : z4;
    ak[t] = z4;
    av[t] = z4;
    acs[t] = 0.f;
    ktt[t] = 0.f;
    if (L2) {
      float kk = kt[t][0] * kt[t][0] + kt[t][1] * kt[t][1] + kt[t][2] * kt[t][2] + kt[t][3] * kt[t][3];
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) kk += __shfl_xor(kk, off, 64);
      ktt[t] = kk;
    }
  }
  for (int rr = sub; rr < 32; rr += 8) {
    const int m = m0 + rr;
    if (m >= M) continue;  // uniform per half-wave
    const long row = idx[(long)b * M + m];
    const f32x4 qv = *reinterpret_cast<const f32x4*>(Q + (long)b * q_bs + row * q_rs + 4 * l32);
    const f32x4 z4r = {0.f, 0.f, 0.f, 0.f};
    const f32x4 ov = Oc ? z4r : *reinterpret_cast<const f32x4*>(O + ((long)b * N + row) * 128 + 4 * l32);
    const float lrow = lse[(long)b * N + row];
    f32x4 dv = {gt[(4 * l32 + 0) * 33 + rr], gt[(4 * l32 + 1) * 33 + rr], gt[(4 * l32 + 2) * 33 + rr],
                gt[(4 * l32 + 3) * 33 + rr]};
    if (Q_tr) {
#pragma unroll
      for (int u = 0; u < 4; ++u) qt[(4 * l32 + u) * 33 + rr] = qv[u];
    } else {
      *reinterpret_cast<f32x4*>(Qs + ((long)b * M + m) * 128 + 4 * l32) = qv;
      *reinterpret_cast<f32x4*>(dO + ((long)b * M + m) * 128 + 4 * l32) = dv;
    }
    float part = dv[0] * ov[0] + dv[1] * ov[1] + dv[2] * ov[2] + dv[3] * ov[3];
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
    if (Oc) {
      part = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) part += dred[w8][rr];
    }
    if (l32 == 0) {
      delta[(long)b * M + m] = part;
      lse_s[(long)b * M + m] = lrow;
    }
    // token keys: P and dS of this row against each token, accumulated into this lane's channels;
    // their share of dQ of this row (sum_t dS_t K_tok[t]) goes to the extra dQ slab
    f32x4 dqt = {0.f, 0.f, 0.f, 0.f};
    float qq = 0.f;
    if (L2) {
      qq = qv[0] * qv[0] + qv[1] * qv[1] + qv[2] * qv[2] + qv[3] * qv[3];
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) qq += __shfl_xor(qq, off, 64);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t < nt) {
        float st = qv[0] * kt[t][0] + qv[1] * kt[t][1] + qv[2] * kt[t][2] + qv[3] * kt[t][3];
        float dpt = dv[0] * vt[t][0] + dv[1] * vt[t][1] + dv[2] * vt[t][2] + dv[3] * vt[t][3];
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
          st += __shfl_xor(st, off, 64);
          dpt += __shfl_xor(dpt, off, 64);
        }
        if (L2) st = 2.f * st - qq - ktt[t];  // -|q - k_tok|^2
        const float p = __expf(st * scale - lrow);
        const float ds = p * (dpt - part) * scale;
        if (L2) acs[t] += ds;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          av[t][u] = fmaf(p, dv[u], av[t][u]);
          ak[t][u] = fmaf(ds, qv[u], ak[t][u]);
          dqt[u] = fmaf(ds, kt[t][u], dqt[u]);
        }
      }
    }
    if (slab)  // slab (b, tok_slab): the token keys' share of dQ
      *reinterpret_cast<f32x4*>(slab + (((long)b * nslab + tok_slab) * M + m) * 128 + 4 * l32) = dqt;
  }
  if (Q_tr) {  // the gathered Q tile -> transposed image, then the LDS region goes back to `red`
    __syncthreads();
    char* itr = Q_tr + ((long)b * ntiles_m + blockIdx.x) * kTriTile;
    for (int e = tid; e < 512; e += 256) {
      const int d = e & 127, cg = e >> 7;
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = qt[d * 33 + 16 * (cg >> 1) + 8 * (i >> 2) + 4 * (cg & 1) + (i & 3)];
      const Tri t3 = tri_split8(x);
      *reinterpret_cast<u32x4*>(itr + tri_tr_off(d, cg, 0)) = t3.h;
      *reinterpret_cast<u32x4*>(itr + tri_tr_off(d, cg, 1)) = t3.m;
      *reinterpret_cast<u32x4*>(itr + tri_tr_off(d, cg, 2)) = t3.l;
    }
    __syncthreads();
  }
  if (nt > 0) {
    // the two half-waves of a wave first (register exchange), then the 4 waves through LDS, in a
    // fixed order; one partial per workgroup
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float k2 = ak[t][u] + __shfl_xor(ak[t][u], 32, 64);
        const float v2 = av[t][u] + __shfl_xor(av[t][u], 32, 64);
        if ((tid & 32) == 0) {
          red[tid >> 6][0][t][4 * l32 + u] = k2;
          red[tid >> 6][1][t][4 * l32 + u] = v2;
        }
      }
    }
    if (L2 && l32 == 0) {
#pragma unroll
      for (int t = 0; t < 8; ++t) csred[sub][t] = acs[t];
    }
    __syncthreads();
    float* outp = tok_part + ((long)b * gridDim.x + blockIdx.x) * 2 * 8 * 128;
    for (int e = tid; e < 2 * 8 * 128; e += 256) {
      float sacc = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) sacc += (&red[w4][0][0][0])[e];
      outp[e] = sacc;
    }
    if (L2 && tid < 8) {  // column sums of dS over this workgroup's rows, per token
      float sacc = 0.f;
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) sacc += csred[s8][tid];
      cs_part[((long)b * gridDim.x + blockIdx.x) * 8 + tid] = sacc;
    }
  }
}

// dK / dV of the nt token rows: fixed-order sum of the per-workgroup partials of bwd_prep.
// grid (16 = dK|dV x token, B), 128 threads = channels
__global__ __launch_bounds__(128) void bwd_tokens_reduce_kernel(const float* __restrict__ tok_part, int nparts, int N,
                                                                int nt, float* __restrict__ dK, long dk_bs, long dk_rs,
                                                                float* __restrict__ dV, long dv_bs, long dv_rs,
                                                                const float* __restrict__ cs_part,
                                                                float* __restrict__ cs, float* __restrict__ dq_tok,
                                                                long dq_bs, long dq_rs) {
  // dq_tok (optional) = dQ where it has nt rows behind its N point rows (the [Q|K|V] gradient block of the
  // projection): token rows are keys only, their dQ is zero
  const int b = blockIdx.y, which = blockIdx.x >> 3, t = blockIdx.x & 7, d = threadIdx.x;
  if (blockIdx.x == 16) {  // l2 scoring: column sums of dS for the token keys -> cs[b][N + t]
    if (cs && d < nt) {
      float sacc = 0.f;
      for (int p = 0; p < nparts; ++p) sacc += cs_part[((long)b * nparts + p) * 8 + d];
      cs[(long)b * (N + nt) + N + d] = sacc;
    }
    return;
  }
  if (t >= nt) return;
  if (dq_tok && which == 0) dq_tok[(long)b * dq_bs + (long)(N + t) * dq_rs + d] = 0.f;
  const float* src = tok_part + (long)b * nparts * 2 * 8 * 128 + (which * 8 + t) * 128 + d;
  float sacc = 0.f;
  int p = 0;
  for (; p + 8 <= nparts; p += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(long)(p + u) * 2 * 8 * 128];
#pragma unroll
    for (int u = 0; u < 8; ++u) sacc += v[u];
  }
  for (; p < nparts; ++p) sacc += src[(long)p * 2 * 8 * 128];
  if (which == 0) dK[(long)b * dk_bs + (long)(N + t) * dk_rs + d] = sacc;
  else dV[(long)b * dv_bs + (long)(N + t) * dv_rs + d] = sacc;
}

// ------------------------------------------------------------------------------------------------
// Fused backward over the N point keys: S, dP, dV, dK and dQ from ONE recomputation of S / dP
// (5 MFMA products per tile instead of the 7 of bwd_dq + bwd_dkdv).
//
// Key-stationary like bwd_dkdv (wave = 32 keys, K/V rows and dK^T/dV^T accumulators in registers).
// For dQ the four waves write their 32x32 dS sub-tiles side by side into one LDS tile
// dS_all[32 queries][128 keys]; wave w then computes the 32-channel slice
//     dQ[:, 32w..32w+31] = dS_all x K[keys of this workgroup][32w..32w+31]
// with that slice of K in registers (64 MFMA, reduced index = the workgroup's 128 keys), so no
// cross-wave reduction exists.  The dS tile is double buffered and consumed one iteration later,
// which costs no extra barrier.  Each key block writes its dQ contribution to its own slab;
// bwd_dq_reduce sums the slabs (and the token-key slab from bwd_prep) in a fixed order and scatters
// the rows: deterministic, no float atomics.
// ------------------------------------------------------------------------------------------------
constexpr int kFusedTile = 2 * kTile * kLdsPad + 2 * kTile;  // Q tile, dO tile, lse[32], delta[32]
constexpr int kFusedLdsFloats = 2 * kFusedTile + 2 * kTile * kLdsPad;

__global__ __launch_bounds__(256, 1) void bwd_fused_kernel(const float* __restrict__ Qs, const float* __restrict__ dO,
                                                           const float* __restrict__ lse_s,
                                                           const float* __restrict__ delta,
                                                           const float* __restrict__ K, long k_bs, long k_rs,
                                                           const float* __restrict__ V, long v_bs, long v_rs, int N,
                                                           int M, float scale, float* __restrict__ dK, long dk_bs,
                                                           long dk_rs, float* __restrict__ dV, long dv_bs, long dv_rs,
                                                           float* __restrict__ slab, int nslab) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dsbuf = smem + 2 * kFusedTile;  // 2 x [32][kLdsPad]
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int j = chunk * 128 + wave * 32 + lo;
  const bool jvalid = j < N;
  const float* Qb = Qs + (long)b * M * 128;
  const float* Gb = dO + (long)b * M * 128;

  float kreg[64], vreg[64], kcol[64];
  if (jvalid) {
    load_row_half(K + (long)b * k_bs + (long)j * k_rs, h, kreg);
    load_row_half(V + (long)b * v_bs + (long)j * v_rs, h, vreg);
  } else {
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      kreg[i] = 0.f;
      vreg[i] = 0.f;
    }
  }
  // this wave's channel slice of the workgroup's 128 K rows: lane (d, h) holds K[key 64h+kk][32w + d]
#pragma unroll
  for (int kk = 0; kk < 64; ++kk) {
    const int jj = chunk * 128 + 64 * h + kk;
    kcol[kk] = (jj < N) ? K[(long)b * k_bs + (long)jj * k_rs + 32 * wave + lo] : 0.f;
  }
  f32x16 dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    dk[dt] = zero16();
    dv[dt] = zero16();
  }

  const int ntiles = (M + kTile - 1) / kTile;
  TileRegs qr, gr;
  float st = 0.f;
  auto issue = [&](int i0) {
    tile_load_issue(qr, Qb, 128, i0, M, tid);
    tile_load_issue(gr, Gb, 128, i0, M, tid);
    if (tid < 64) {
      const int ii = i0 + (tid & 31);
      const float* src = (tid < 32) ? lse_s : delta;
      st = (ii < M) ? src[(long)b * M + ii] : 0.f;
    }
  };
  auto commit = [&](float* buf) {
    tile_store_lds(qr, buf, kLdsPad, tid);
    tile_store_lds(gr, buf + kTile * kLdsPad, kLdsPad, tid);
    if (tid < 64) buf[2 * kTile * kLdsPad + tid] = st;
  };
  issue(0);
  commit(smem);
  __syncthreads();

  float* myslab = slab + ((long)b * nslab + chunk) * M * 128;
  for (int t = 0; t <= ntiles; ++t) {
    float* cur = smem + (t & 1) * kFusedTile;
    float* nxt = smem + ((t & 1) ^ 1) * kFusedTile;
    const int i0 = t * kTile;
    if (t + 1 < ntiles) issue(i0 + kTile);
    if (t < ntiles) {
      const float* Qt = cur;
      const float* Gt = cur + kTile * kLdsPad;
      const float* Lt = cur + 2 * kTile * kLdsPad;
      const float* Dt = Lt + kTile;
      float* dsw = dsbuf + (t & 1) * kTile * kLdsPad + 32 * wave + lo;
      f32x16 s = mma_rows_x_regs(Qt, kLdsPad, lo, h, kreg, zero16());   // S  (queries x keys)
      f32x16 dp = mma_rows_x_regs(Gt, kLdsPad, lo, h, vreg, zero16());  // dP
      const bool tail = (i0 + kTile > M);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ir = crow(r, h);
        float p = __expf(s[r] * scale - Lt[ir]);
        if (tail && (i0 + ir >= M)) p = 0.f;
        const float ds = p * (dp[r] - Dt[ir]) * scale;
        dsw[ir * kLdsPad] = ds;  // dS_all[query][this wave's 32 key columns]
        mma_tileT_step(Gt, kLdsPad, lo, h, r, p, dv);
        mma_tileT_step(Qt, kLdsPad, lo, h, r, ds, dk);
      }
    }
    if (t > 0) {
      // dQ slice of the previous tile: rows = its 32 queries, reduced index = this workgroup's 128 keys
      const float* dsr = dsbuf + ((t - 1) & 1) * kTile * kLdsPad;
      const f32x16 dqa = mma_rows_x_regs(dsr, kLdsPad, lo, h, kcol, zero16());
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (t - 1) * kTile + crow(r, h);
        if (m < M) myslab[(long)m * 128 + 32 * wave + lo] = dqa[r];
      }
    }
    if (t + 1 < ntiles) commit(nxt);
    __syncthreads();
  }
  if (jvalid) {
    float* krow = dK + (long)b * dk_bs + (long)j * dk_rs;
    float* vrow = dV + (long)b * dv_bs + (long)j * dv_rs;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        f32x4 a = {dk[dt][4 * gq], dk[dt][4 * gq + 1], dk[dt][4 * gq + 2], dk[dt][4 * gq + 3]};
        f32x4 c = {dv[dt][4 * gq], dv[dt][4 * gq + 1], dv[dt][4 * gq + 2], dv[dt][4 * gq + 3]};
        *reinterpret_cast<f32x4*>(krow + 32 * dt + 8 * gq + 4 * h) = a;
        *reinterpret_cast<f32x4*>(vrow + 32 * dt + 8 * gq + 4 * h) = c;
      }
    }
  }
}

// dQ row of sampled point m = sum over the key-block slabs (+ the token slab), fixed order; scattered
// to row idx[m] of dQ.  grid (ceil(M*32/256), B): one thread per (row, 4-channel group)
__global__ __launch_bounds__(256) void bwd_dq_reduce_kernel(const float* __restrict__ slab, int nslab,
                                                            const long long* __restrict__ idx, int M,
                                                            float* __restrict__ dQ, long dq_bs, long dq_rs) {
  const int b = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int m = e >> 5, c4 = e & 31;
  if (m >= M) return;
  const float* src = slab + (long)b * nslab * M * 128 + (long)m * 128 + 4 * c4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int sidx = 0;
  for (; sidx + 4 <= nslab; sidx += 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (long)(sidx + u) * M * 128);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u];
  }
  for (; sidx < nslab; ++sidx) acc += *reinterpret_cast<const f32x4*>(src + (long)sidx * M * 128);
  const long row = idx[(long)b * M + m];
  *reinterpret_cast<f32x4*>(dQ + (long)b * dq_bs + row * dq_rs + 4 * c4) = acc;
}

// ------------------------------------------------------------------------------------------------
// dQ: query-stationary
// ------------------------------------------------------------------------------------------------
constexpr int kDqLdsFloats = 2 * (2 * kTile * kLdsPad);
constexpr int kDqWaves = 4;  // 4 waves = 128 sampled rows per workgroup (8 waves at 2 per SIMD spill: 1.54 vs 1.07 ms)

template <int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void bwd_dq_kernel(const float* __restrict__ Qs, const float* __restrict__ dO,
                                                        const float* __restrict__ lse_s,
                                                        const float* __restrict__ delta,
                                                        const float* __restrict__ K, long k_bs, long k_rs,
                                                        const float* __restrict__ V, long v_bs, long v_rs,
                                                        const long long* __restrict__ idx, int N, int NK, int M,
                                                        float scale, float* __restrict__ dQ, long dq_bs, long dq_rs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kBuf = 2 * kTile * kLdsPad;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int m = chunk * (32 * NW) + wave * 32 + lo;
  const bool mvalid = m < M;
  const float* Kb = K + (long)b * k_bs;
  const float* Vb = V + (long)b * v_bs;

  float q[64], go[64];
  float my_lse = 0.f, my_delta = 0.f;
  if (mvalid) {
    load_row_half(Qs + ((long)b * M + m) * 128, h, q);
    load_row_half(dO + ((long)b * M + m) * 128, h, go);
    my_lse = lse_s[(long)b * M + m];
    my_delta = delta[(long)b * M + m];
  } else {
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      q[i] = 0.f;
      go[i] = 0.f;
    }
  }
  f32x16 dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = zero16();

  const int ntiles = (NK + kTile - 1) / kTile;
  TileRegsT<64 * NW> kr, vr;
  tile_load_issue(kr, Kb, k_rs, 0, NK, tid);
  tile_load_issue(vr, Vb, v_rs, 0, NK, tid);
  tile_store_lds(kr, smem, kLdsPad, tid);
  tile_store_lds(vr, smem + kTile * kLdsPad, kLdsPad, tid);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    float* Kc = smem + (t & 1) * kBuf;
    float* Vc = Kc + kTile * kLdsPad;
    float* Kn = smem + ((t & 1) ^ 1) * kBuf;
    float* Vn = Kn + kTile * kLdsPad;
    const int j0 = t * kTile;
    if (t + 1 < ntiles) {
      tile_load_issue(kr, Kb, k_rs, j0 + kTile, NK, tid);
      tile_load_issue(vr, Vb, v_rs, j0 + kTile, NK, tid);
    }
    f32x16 s = mma_rows_x_regs(Kc, kLdsPad, lo, h, q, zero16());    // S^T  (keys x queries)
    f32x16 dp = mma_rows_x_regs(Vc, kLdsPad, lo, h, go, zero16());  // dP^T
    const bool tail = (j0 + kTile > NK);
    // dQ^T[d][i] += sum_j K[j][d] dS[i][j]; dS of register r is produced right before its MFMAs
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float p = __expf(s[r] * scale - my_lse);
      if (tail && (j0 + crow(r, h) >= NK)) p = 0.f;
      const float ds = p * (dp[r] - my_delta) * scale;  // dS^T, scale of S folded in
      mma_tileT_step(Kc, kLdsPad, lo, h, r, ds, dq);
    }
    if (t + 1 < ntiles) {
      tile_store_lds(kr, Kn, kLdsPad, tid);
      tile_store_lds(vr, Vn, kLdsPad, tid);
    }
    __syncthreads();
  }
  if (mvalid) {
    const long row = idx[(long)b * M + m];
    float* orow = dQ + (long)b * dq_bs + row * dq_rs;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        f32x4 o = {dq[dt][4 * gq], dq[dt][4 * gq + 1], dq[dt][4 * gq + 2], dq[dt][4 * gq + 3]};
        *reinterpret_cast<f32x4*>(orow + 32 * dt + 8 * gq + 4 * h) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dK, dV for the N point keys: key-stationary
// ------------------------------------------------------------------------------------------------
constexpr int kDkvLdsFloats = 2 * (2 * kTile * kLdsPad + 2 * kTile);

__global__ __launch_bounds__(256, 1) void bwd_dkdv_kernel(const float* __restrict__ Qs, const float* __restrict__ dO,
                                                          const float* __restrict__ lse_s,
                                                          const float* __restrict__ delta,
                                                          const float* __restrict__ K, long k_bs, long k_rs,
                                                          const float* __restrict__ V, long v_bs, long v_rs, int N,
                                                          int M, float scale, float* __restrict__ dK, long dk_bs,
                                                          long dk_rs, float* __restrict__ dV, long dv_bs,
                                                          long dv_rs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kBuf = 2 * kTile * kLdsPad + 2 * kTile;  // Q tile, dO tile, lse[32], delta[32]
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int j = chunk * 128 + wave * 32 + lo;
  const bool jvalid = j < N;
  const float* Qb = Qs + (long)b * M * 128;
  const float* Gb = dO + (long)b * M * 128;

  float kreg[64], vreg[64];
  if (jvalid) {
    load_row_half(K + (long)b * k_bs + (long)j * k_rs, h, kreg);
    load_row_half(V + (long)b * v_bs + (long)j * v_rs, h, vreg);
  } else {
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      kreg[i] = 0.f;
      vreg[i] = 0.f;
    }
  }
  f32x16 dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    dk[dt] = zero16();
    dv[dt] = zero16();
  }

  const int ntiles = (M + kTile - 1) / kTile;
  TileRegs qr, gr;
  float st = 0.f;  // threads 0..31 stage lse, 32..63 stage delta of the next query tile
  auto issue = [&](int i0) {
    tile_load_issue(qr, Qb, 128, i0, M, tid);
    tile_load_issue(gr, Gb, 128, i0, M, tid);
    if (tid < 64) {
      const int ii = i0 + (tid & 31);
      const float* src = (tid < 32) ? lse_s : delta;
      st = (ii < M) ? src[(long)b * M + ii] : 0.f;
    }
  };
  auto commit = [&](float* buf) {
    tile_store_lds(qr, buf, kLdsPad, tid);
    tile_store_lds(gr, buf + kTile * kLdsPad, kLdsPad, tid);
    if (tid < 64) buf[2 * kTile * kLdsPad + tid] = st;
  };
  issue(0);
  commit(smem);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    float* cur = smem + (t & 1) * kBuf;
    float* nxt = smem + ((t & 1) ^ 1) * kBuf;
    const float* Qt = cur;
    const float* Gt = cur + kTile * kLdsPad;
    const float* Lt = cur + 2 * kTile * kLdsPad;
    const float* Dt = Lt + kTile;
    const int i0 = t * kTile;
    if (t + 1 < ntiles) issue(i0 + kTile);

    f32x16 s = mma_rows_x_regs(Qt, kLdsPad, lo, h, kreg, zero16());   // S  (queries x keys)
    f32x16 dp = mma_rows_x_regs(Gt, kLdsPad, lo, h, vreg, zero16());  // dP
    const bool tail = (i0 + kTile > M);
    // dV^T[d][j] += sum_i dO[i][d] P[i][j];  dK^T[d][j] += sum_i Q[i][d] dS[i][j]: P / dS of register r
    // are produced right before the eight MFMAs that consume them
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ir = crow(r, h);
      float p = __expf(s[r] * scale - Lt[ir]);
      if (tail && (i0 + ir >= M)) p = 0.f;
      const float ds = p * (dp[r] - Dt[ir]) * scale;
      mma_tileT_step(Gt, kLdsPad, lo, h, r, p, dv);
      mma_tileT_step(Qt, kLdsPad, lo, h, r, ds, dk);
    }
    if (t + 1 < ntiles) commit(nxt);
    __syncthreads();
  }
  if (jvalid) {
    float* krow = dK + (long)b * dk_bs + (long)j * dk_rs;
    float* vrow = dV + (long)b * dv_bs + (long)j * dv_rs;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        f32x4 a = {dk[dt][4 * gq], dk[dt][4 * gq + 1], dk[dt][4 * gq + 2], dk[dt][4 * gq + 3]};
        f32x4 c = {dv[dt][4 * gq], dv[dt][4 * gq + 1], dv[dt][4 * gq + 2], dv[dt][4 * gq + 3]};
        *reinterpret_cast<f32x4*>(krow + 32 * dt + 8 * gq + 4 * h) = a;
        *reinterpret_cast<f32x4*>(vrow + 32 * dt + 8 * gq + 4 * h) = c;
      }
    }
  }
}

}  // namespace samble

using namespace samble;

extern "C" int samble_launch_bwd_rows(const float*, const float*, const float*, const float*, const float*, long, long,
                                      const float*, long, long, int, int, int, float, float*, long, long, float*, long,
                                      long, float*, int, const float*, int, const long long*, float*, int, hipStream_t);

extern "C" size_t samble_attn_bwd_slab_floats(int B, int N, int M) {
  return (size_t)B * ((N + 127) / 128 + 1) * M * 128;
}

extern "C" size_t samble_tri_image_size(int, int, int);
extern "C" int samble_launch_bwd_tri(const float*, int, const float*, const float*, const void*, const void*, const void*,
                                     const void*, const void*, const long long*, int, int, int, int, float, float*, long,
                                     long, float*, long, long, float*, long, long, float*, float*, int, int, unsigned*,
                                     hipStream_t);

extern "C" size_t samble_bwd_tri_dsmap_floats(int B, int N, int M);
extern "C" int samble_launch_attn_bwd(const float* Q, long q_bs, long q_rs, const float* K, long k_bs, long k_rs,
                                      const float* V, long v_bs, long v_rs, const float* O, const float* Oc,
                                      const float* smap, int ld, const float* lse, const long long* idx,
                                      const float* g, int B, int N, int nt, int M, float scale, float* Qs, float* dOb,
                                      float* lse_s, float* delta, float* tok_part, float* slab, float* dQ, long dq_bs,
                                      long dq_rs, float* dK, long dk_bs, long dk_rs, float* dV, long dv_bs, long dv_rs,
                                      int l2, float* cs, float* cs_part, const void* k_tr_image, const void* v_rm_image,
                                      void* img_ws, int variant, int zero_dq, hipStream_t stream) {
  // zero_dq bit 0: the caller left clearing dQ's N rows to bwd_prep_tri (samble_attn_bwd_prep_clears_dq says when it may);
  // bit 1: dQ has nt token rows behind the N point rows, cleared by bwd_tokens_reduce
  // variant (include/samble.h): single-pass path: 1 = the two-kernel backward (bwd_dq + bwd_dkdv, 7 products)
  // instead of the fused one (5); split-bf16 map path: 1 = fused dP / dV / dK kernel instead of the dS map,
  // 2 = smap is the P map (B, M, ld) of the sampled rows (attn_rows_rc_tri) instead of the logit map; 2 + 4 = the same
  // on the un-woven dQ kernel (A/B)
  // k_tr_image / v_rm_image / img_ws != null (map path only): the split-bf16 kernels of attn_bwd_tri.hip
  // l2 != 0 (map path only): token logits are -|q-k|^2 and cs (B, N+nt) receives the column sums of dS
  // O (B,N,128) rows of the single-pass forward, or Oc (B,128,M) = x_ds of attn_rows; smap (B,N,ld) =
  // the logit map of attn_stats (then S is read, not recomputed) or null
  const size_t lds_dq = kDqLdsFloats * sizeof(float), lds_dkv = kDkvLdsFloats * sizeof(float);
  const size_t lds_fused = kFusedLdsFloats * sizeof(float);
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_dq_kernel<kDqWaves>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dq);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_dkdv_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dkv);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_fused_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fused);
    if (e != hipSuccess) return (int)e;
  }
  const int NK = N + nt;
  const int nparts = (M + 31) / 32;
  const int kb = (N + 127) / 128;
  const bool tri = smap && k_tr_image && v_rm_image && img_ws;
  const bool fused = (!(variant & 1) || smap) && !tri;
  float* tok_slab = (fused && nt > 0) ? slab + (size_t)kb * M * 128 : nullptr;  // per cloud: slab index kb
  // (the token slab of cloud b sits at slab + (b * (kb + 1) + kb) * M * 128: pass the base, prep adds b * M * 128
  //  only, so give it a view with the cloud stride folded in below)
  // split-bf16 backward: images of the sampled rows (dO row + transposed, Q transposed, whole tiles of 32 rows),
  // written by bwd_prep itself; then the dS map
  const size_t img = tri ? samble_tri_image_size(B, M, 0) : 0;
  char* dO_rm = tri ? (char*)img_ws : nullptr;
  char* dO_tr = tri ? dO_rm + img : nullptr;
  char* Q_tr = tri ? dO_tr + img : nullptr;
  // the dS-map backward (variant bit 0 clear) reads its transposed images as two fp16 planes per tile; the cloud's
  // largest |dS| lives behind the dS map (samble_bwd_tri_dsmap_bytes counts it)
  unsigned* ds_amax = (tri && Oc && !(variant & 1))
                          ? reinterpret_cast<unsigned*>(Q_tr + img + samble_bwd_tri_dsmap_floats(B, N, M) * sizeof(float))
                          : nullptr;
  if (tri && !Oc) return (int)hipErrorInvalidValue;  // (the split-bf16 backward is the map pipeline's: x_ds is there)
  if (tri && Oc) {
    Timed timed(kT_bwd_prep, stream);
    hipLaunchKernelGGL(l2 ? bwd_prep_tri_kernel<true> : bwd_prep_tri_kernel<false>, dim3(nparts, B), dim3(256), 0, stream, Q,
                       q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs, Oc, lse, idx, g, N, nt, M, scale, lse_s, delta, tok_part,
                       l2 ? cs_part : nullptr, dO_rm, dO_tr, Q_tr, (zero_dq & 1) ? dQ : nullptr, dq_bs, dq_rs, ds_amax);
  } else {
    Timed timed(kT_bwd_prep, stream);
    hipLaunchKernelGGL(l2 ? bwd_prep_kernel<true> : bwd_prep_kernel<false>, dim3(nparts, B), dim3(256), 0, stream, Q, q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs,
                     O, Oc, lse, idx, g, N, nt, M, scale, Qs, dOb, lse_s, delta, tok_part, fused ? slab : nullptr,
                     fused ? kb + 1 : 0, kb, l2, l2 ? cs_part : nullptr, dO_rm, dO_tr, Q_tr);
  }
  if (tri) {
    float* dsmap = reinterpret_cast<float*>(Q_tr + img);  // (B, M, ld) after the three images
    const int rc = samble_launch_bwd_tri(smap, ld, lse_s, delta, dO_rm, dO_tr, Q_tr, v_rm_image, k_tr_image, idx, B, N, nt, M,
                                         scale, dQ, dq_bs, dq_rs, dK, dk_bs, dk_rs, dV, dv_bs, dv_rs, l2 ? cs : nullptr,
                                         dsmap, variant & 1, variant & 6, ds_amax, stream);
    if (rc) return rc;
  } else if (fused) {
    if (smap) {
      const int rc = samble_launch_bwd_rows(Qs, dOb, lse_s, delta, K, k_bs, k_rs, V, v_bs, v_rs, B, N, M, scale, dK, dk_bs,
                                            dk_rs, dV, dv_bs, dv_rs, slab, kb + 1, smap, ld, idx, l2 ? cs : nullptr, nt,
                                            stream);
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL(bwd_fused_kernel, dim3(kb, B), dim3(256), lds_fused, stream, Qs, dOb, lse_s, delta, K, k_bs, k_rs,
                         V, v_bs, v_rs, N, M, scale, dK, dk_bs, dk_rs, dV, dv_bs, dv_rs, slab, kb + 1);
    }
    hipLaunchKernelGGL(bwd_dq_reduce_kernel, dim3((M * 32 + 255) / 256, B), dim3(256), 0, stream, slab, kb + 1, idx, M,
                       dQ, dq_bs, dq_rs);
  } else {
    hipLaunchKernelGGL(bwd_dq_kernel<kDqWaves>, dim3((M + 32 * kDqWaves - 1) / (32 * kDqWaves), B), dim3(64 * kDqWaves),
                       lds_dq, stream, Qs, dOb, lse_s, delta, K, k_bs, k_rs, V, v_bs, v_rs, idx, N, NK, M, scale, dQ,
                       dq_bs, dq_rs);
    hipLaunchKernelGGL(bwd_dkdv_kernel, dim3(kb, B), dim3(256), lds_dkv, stream, Qs, dOb, lse_s, delta, K, k_bs, k_rs, V,
                       v_bs, v_rs, N, M, scale, dK, dk_bs, dk_rs, dV, dv_bs, dv_rs);
  }
  (void)tok_slab;
  if (nt > 0)
    hipLaunchKernelGGL(bwd_tokens_reduce_kernel, dim3(l2 ? 17 : 16, B), dim3(128), 0, stream, tok_part, nparts, N, nt, dK,
                       dk_bs, dk_rs, dV, dv_bs, dv_rs, l2 ? cs_part : nullptr, l2 ? cs : nullptr, (zero_dq & 2) ? dQ : nullptr,
                       dq_bs, dq_rs);
  return (int)hipGetLastError();
}

#ifdef SAMBLE_STAMPS
extern "C" __attribute__((visibility("default"))) int samble_scratch_prep_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(samble::g_prep_stamps), sizeof(unsigned long long) * 32);
}
#endif

// true when samble_launch_attn_bwd with these arguments runs bwd_prep_tri_kernel, which can clear dQ on its way
extern "C" int samble_attn_bwd_prep_clears_dq(const float* smap, const float* Oc, const void* k_tr_image,
                                              const void* v_rm_image, const void* img_ws) {
  return smap && Oc && k_tr_image && v_rm_image && img_ws;
}
