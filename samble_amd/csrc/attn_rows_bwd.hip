// Backward of the two-pass attention over the N point keys, reading S from the logit map
// (attn_map.hip) instead of recomputing it: dP, dV, dK and the dQ slabs = 4 matrix products per
// (32 sampled rows x 32 keys) tile.  Same mathematics and the same deterministic dQ-slab scheme as
// bwd_fused_kernel (attn_bwd.hip); what differs is the schedule.  One wave per SIMD runs here (the
// key-stationary operands fill the register file), so nothing but the wave's own instruction order
// can overlap VALU / LDS latency with the matrix pipe:
//
//   per tile t      A. dP(t)   = dO_tile x V_keys^T                       64 MFMA
//                   B. dQ(t-1) = dS_all(t-1) x K_cols                     64 MFMA, with the softmax-
//                      backward VALU of tile t (P = exp(S - lse), dS = P (dP - delta) scale, dS -> LDS)
//                      issued between them
//                   C. slab stores of dQ(t-1), then dV += P^T dO, dK += dS^T Q   128 MFMA
//                   D. commit of tile t+1 (its global loads were issued before A), barrier
//
// The slab stores sit ~8000 cycles before the commit's vmcnt wait (memory operations of a wave
// retire in order, so a store issued right before the wait -- the previous schedule -- is paid in
// full), and the middle iterations contain no branch, which lets the compiler count that wait.
#include <type_traits>

#include "samble_dev.h"

namespace samble {

constexpr int kRbTile = 2 * kTile * kLdsPad + 2 * kTile;  // Q tile, dO tile, lse[32], delta[32]
constexpr int kRbLdsFloats = 2 * kRbTile + 2 * kTile * kLdsPad;

struct RowsBwdArgs {
  const float* Qs;
  const float* dO;
  const float* lse_s;
  const float* delta;
  const float* K;
  long k_bs, k_rs;
  const float* V;
  long v_bs, v_rs;
  int N, M;
  float scale;
  float* dK;
  long dk_bs, dk_rs;
  float* dV;
  long dv_bs, dv_rs;
  float* slab;
  int nslab;
  const float* smap;
  int ld;
  const long long* idx;
};

__global__ __launch_bounds__(256, 1) void bwd_rows_kernel(const RowsBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dsbuf = smem + 2 * kRbTile;                                 // 2 x [32][kLdsPad]
  int* sel = reinterpret_cast<int*>(dsbuf + 2 * kTile * kLdsPad);    // the cloud's M sampled row ids
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int N = a.N, M = a.M;
  const int j = chunk * 128 + wave * 32 + lo;
  const bool jvalid = j < N;
  const float* Qb = a.Qs + (long)b * M * 128;
  const float* Gb = a.dO + (long)b * M * 128;

  float vreg[64], kcol[64];
  load_row_half(a.V + (long)b * a.v_bs + (long)min(j, N - 1) * a.v_rs, h, vreg);
  // this wave's channel slice of the workgroup's 128 K rows: lane (d, h) holds K[key 64h+kk][32w + d]
#pragma unroll
  for (int kk = 0; kk < 64; ++kk) {
    const int jj = chunk * 128 + 64 * h + kk;
    const float x = a.K[(long)b * a.k_bs + (long)min(jj, N - 1) * a.k_rs + 32 * wave + lo];
    kcol[kk] = (jj < N) ? x : 0.f;
  }
  f32x16 dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    dk[dt] = zero16();
    dv[dt] = zero16();
  }
  for (int i = tid; i < M; i += 256) sel[i] = (int)a.idx[(long)b * M + i];

  const int ntiles = (M + kTile - 1) / kTile;
  const float* scol = a.smap + (long)b * N * a.ld + min(j, a.ld - 1);
  const int ld = a.ld;
  TileRegs qr, gr;
  float st;
  float sv[16], sn[16];
  // tile loads: rows past M-1 are clamped (their P is masked to zero below), lse/delta by all threads
  auto issue = [&](int i0) {
#pragma unroll
    for (int i = 0; i < TileRegs::kPer; ++i) {
      const int e = tid + 256 * i, r = e >> 5, c4 = e & 31;
      const long row = min(i0 + r, M - 1);
      qr.v[i] = *reinterpret_cast<const f32x4*>(Qb + row * 128 + 4 * c4);
      gr.v[i] = *reinterpret_cast<const f32x4*>(Gb + row * 128 + 4 * c4);
    }
    const int ii = min(i0 + (tid & 31), M - 1);
    st = ((tid & 32) ? a.delta : a.lse_s)[(long)b * M + ii];
  };
  auto commit = [&](float* buf) {
    tile_store_lds(qr, buf, kLdsPad, tid);
    tile_store_lds(gr, buf + kTile * kLdsPad, kLdsPad, tid);
    buf[2 * kTile * kLdsPad + (tid & 63)] = st;  // the 4 waves write the same 64 values
  };
  auto load_s = [&](int i0, float (&dst)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[r] = scol[sel[min(i0 + crow(r, h), M - 1)] * ld];
  };
  issue(0);
  commit(smem);
  __syncthreads();
  load_s(0, sv);

  float* myslab = a.slab + ((long)b * a.nslab + chunk) * M * 128 + 32 * wave + lo;
  const float scale = a.scale;

  // one iteration; PREV: tile t-1 exists (its dQ is due), CUR: tile t exists, NEXT: tile t+1 exists
  auto body = [&](int t, auto prev_c, auto cur_c, auto next_c) {
    constexpr bool PREV = decltype(prev_c)::value, CUR = decltype(cur_c)::value, NEXT = decltype(next_c)::value;
    float* cur = smem + (t & 1) * kRbTile;
    float* nxt = smem + ((t & 1) ^ 1) * kRbTile;
    const int i0 = t * kTile;
    if (NEXT) {
      issue(i0 + kTile);
      load_s(i0 + kTile, sn);
    }
    const float* Qt = cur;
    const float* Gt = cur + kTile * kLdsPad;
    const float* Lt = cur + 2 * kTile * kLdsPad;
    const float* Dt = Lt + kTile;
    f32x16 dp = zero16();
    if (CUR) dp = mma_rows_x_regs(Gt, kLdsPad, lo, h, vreg, zero16());  // A. dP (queries x keys)
    // B. dQ of the previous tile (rows = its 32 queries, reduced index = this workgroup's 128 keys)
    //    with the softmax backward of this tile in between
    f32x16 dqa = zero16();
    const f32x4* lpd = reinterpret_cast<const f32x4*>(dsbuf + ((t - 1) & 1) * kTile * kLdsPad + lo * kLdsPad + 64 * h);
    float* dsw = dsbuf + (t & 1) * kTile * kLdsPad + 32 * wave + lo;
    float p[16], ds[16];
#pragma unroll
    for (int q4 = 0; q4 < 16; ++q4) {
      if (PREV) {
        const f32x4 av = lpd[q4];
#pragma unroll
        for (int e = 0; e < 4; ++e) dqa = mfma32(av[e], kcol[4 * q4 + e], dqa);
      }
      if (CUR) {
        const int ir = crow(q4, h);
        float pv = __expf(sv[q4] - Lt[ir]);
        pv = (i0 + ir < M) ? pv : 0.f;
        p[q4] = pv;
        ds[q4] = jvalid ? pv * (dp[q4] - Dt[ir]) * scale : 0.f;
        dsw[ir * kLdsPad] = ds[q4];  // dS_all[query][this wave's 32 key columns]
      }
    }
    // C. slab stores of dQ(t-1) first (far ahead of the wait in D), then dV and dK
    if (PREV) {
      float* srow = myslab + (long)(i0 - kTile) * 128;
      if (CUR) {  // tile t-1 is a full tile
#pragma unroll
        for (int r = 0; r < 16; ++r) srow[crow(r, h) * 128] = dqa[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (i0 - kTile + crow(r, h) < M) srow[crow(r, h) * 128] = dqa[r];
      }
    }
    if (CUR) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        mma_tileT_step(Gt, kLdsPad, lo, h, r, p[r], dv);
        mma_tileT_step(Qt, kLdsPad, lo, h, r, ds[r], dk);
      }
    }
    // D.
    if (NEXT) {
      commit(nxt);
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = sn[r];
    }
    __syncthreads();
  };
  using T = std::true_type;
  using F = std::false_type;
  if (ntiles == 1) {
    body(0, F{}, T{}, F{});
  } else {
    body(0, F{}, T{}, T{});
    int t = 1;
    for (; t + 1 < ntiles; ++t) body(t, T{}, T{}, T{});
    body(t, T{}, T{}, F{});
  }
  body(ntiles, T{}, F{}, F{});

  if (jvalid) {
    float* krow = a.dK + (long)b * a.dk_bs + (long)j * a.dk_rs;
    float* vrow = a.dV + (long)b * a.dv_bs + (long)j * a.dv_rs;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        f32x4 x = {dk[dt][4 * gq], dk[dt][4 * gq + 1], dk[dt][4 * gq + 2], dk[dt][4 * gq + 3]};
        f32x4 c = {dv[dt][4 * gq], dv[dt][4 * gq + 1], dv[dt][4 * gq + 2], dv[dt][4 * gq + 3]};
        *reinterpret_cast<f32x4*>(krow + 32 * dt + 8 * gq + 4 * h) = x;
        *reinterpret_cast<f32x4*>(vrow + 32 * dt + 8 * gq + 4 * h) = c;
      }
    }
  }
}

}  // namespace samble

using namespace samble;

extern "C" int samble_launch_bwd_rows(const float* Qs, const float* dOb, const float* lse_s, const float* delta,
                                      const float* K, long k_bs, long k_rs, const float* V, long v_bs, long v_rs, int B,
                                      int N, int M, float scale, float* dK, long dk_bs, long dk_rs, float* dV, long dv_bs,
                                      long dv_rs, float* slab, int nslab, const float* smap, int ld,
                                      const long long* idx, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_rows_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const size_t lds = kRbLdsFloats * sizeof(float) + (size_t)M * 4;
  if (lds > 160 * 1024) return -22;
  RowsBwdArgs a{Qs, dOb, lse_s, delta, K, k_bs, k_rs, V, v_bs, v_rs, N, M, scale, dK, dk_bs, dk_rs,
                dV, dv_bs, dv_rs, slab, nslab, smap, ld, idx};
  hipLaunchKernelGGL(bwd_rows_kernel, dim3((N + 127) / 128, B), dim3(256), lds, stream, a);
  return (int)hipGetLastError();
}
