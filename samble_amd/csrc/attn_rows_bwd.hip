// Backward of the two-pass attention over the N point keys, reading S from the logit map
// (attn_map.hip) instead of recomputing it: dP, dV, dK and the dQ slabs = 4 matrix products per
// (32 sampled rows x 32 keys) tile.  Same mathematics and the same deterministic dQ-slab scheme as
// bwd_fused_kernel (attn_bwd.hip).
//
// Schedule.  On gfx950 the fp32 MFMA runs on the SIMD's own fp32 lanes: VALU instructions do not
// issue under it (tools/micro/coissue_bench.hip: time = MFMA time + VALU time), a lone wave on a
// SIMD pays ~8% issue bubbles between MFMAs and ~7.5 cycles per VALU instruction instead of ~4.3.
// So the kernel is built for TWO waves per SIMD: a workgroup owns 128 keys and has 8 waves, two
// per 32-key block with different roles, a software pipeline one tile deep between them:
//   role A (waves 0-3), tile t:    dP = dO_tile x V_keys^T (V rows in registers), P = exp(S - lse)
//                                  from the map, dS = P (dP - delta) scale -> LDS, dV += P^T dO
//   role B (waves 4-7), tile t-1:  dK += dS^T Q (dS read back from LDS), dQ slab = dS_all x K_cols
//                                  (this wave's 32-channel slice of the block's K rows in registers)
// 128 MFMAs per role and tile, ~200 registers per wave, one barrier per tile; Q / dO tiles are
// triple-buffered in LDS (B consumes a tile one iteration after A), staged by all 512 threads.
#include <type_traits>

#include "samble_dev.h"

namespace samble {

constexpr int kRbTile = 2 * kTile * kLdsPad + 2 * kTile;  // Q tile, dO tile, lse[32], delta[32]
constexpr int kRbLdsFloats = 3 * kRbTile + 2 * kTile * kLdsPad;

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
  float* cs;  // optional (B, N+nt): column sums of dS over the sampled rows (needed by l2 scoring)
  int nt;
};

template <bool CS>  // CS: also accumulate the column sums of dS (asm "l2")
__global__ __launch_bounds__(512, 2) void bwd_rows_kernel(const RowsBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dsbuf = smem + 3 * kRbTile;                                 // 2 x [32][kLdsPad]
  int* sel = reinterpret_cast<int*>(dsbuf + 2 * kTile * kLdsPad);    // the cloud's M sampled row ids
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  const bool roleA = wave < 4;
  const int kw = wave & 3;  // 32-key block of this wave inside the workgroup's 128 keys
  int chunk, b;
  xcd_assign(chunk, b);
  const int N = a.N, M = a.M;
  const int j = chunk * 128 + kw * 32 + lo;
  const bool jvalid = j < N;
  const float* Qb = a.Qs + (long)b * M * 128;
  const float* Gb = a.dO + (long)b * M * 128;

  // key-stationary operands: role A holds its keys' V rows, role B the 32-channel slice kw of the
  // workgroup's 128 K rows (lane (d, h) holds K[key 64h+kk][32 kw + d])
  float kv[64];
  if (roleA) {
    load_row_half(a.V + (long)b * a.v_bs + (long)min(j, N - 1) * a.v_rs, h, kv);
  } else {
#pragma unroll
    for (int kk = 0; kk < 64; ++kk) {
      const int jj = chunk * 128 + 64 * h + kk;
      const float x = a.K[(long)b * a.k_bs + (long)min(jj, N - 1) * a.k_rs + 32 * kw + lo];
      kv[kk] = (jj < N) ? x : 0.f;
    }
  }
  f32x16 acc[4];  // role A: dV^T, role B: dK^T  (channels x this wave's 32 keys)
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) acc[dt] = zero16();
  for (int i = tid; i < M; i += 512) sel[i] = (int)a.idx[(long)b * M + i];

  const int ntiles = (M + kTile - 1) / kTile;
  const float* scol = a.smap + (long)b * N * a.ld + min(j, a.ld - 1);
  const int ld = a.ld;
  TileRegsT<512> qr, gr;
  float st;
  float sv[16];
  // tile loads: rows past M-1 are clamped (their P is masked to zero below), lse/delta by all threads
  auto issue = [&](int i0) {
#pragma unroll
    for (int i = 0; i < TileRegsT<512>::kPer; ++i) {
      const int e = tid + 512 * i, r = e >> 5, c4 = e & 31;
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
    buf[2 * kTile * kLdsPad + (tid & 63)] = st;  // the 8 waves write the same 64 values
  };
  auto load_s = [&](int i0, float (&dst)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[r] = scol[sel[min(i0 + crow(r, h), M - 1)] * ld];
  };
  issue(0);
  commit(smem);
  __syncthreads();
  if (roleA) load_s(0, sv);

  float* myslab = a.slab + ((long)b * a.nslab + chunk) * M * 128 + 32 * kw + lo;
  float csum = 0.f;  // role A: sum over the sampled rows of dS for this lane's key (its half of the rows)
  const float scale = a.scale;

  // iteration `it`: role A works on tile it (ACUR), role B on tile it-1 (BPREV); tile it+1 is staged (NEXT)
  auto body = [&](int it, int bufA, auto acur_c, auto bprev_c, auto next_c) {
    constexpr bool ACUR = decltype(acur_c)::value, BPREV = decltype(bprev_c)::value, NEXT = decltype(next_c)::value;
    const int bufN = (bufA == 2) ? 0 : bufA + 1;  // tile it+1
    const int bufB = (bufN == 2) ? 0 : bufN + 1;  // tile it-1
    const int i0 = it * kTile;
    if (NEXT) issue(i0 + kTile);
    if (roleA) {
      if (ACUR) {
        const float* cur = smem + bufA * kRbTile;
        const float* Gt = cur + kTile * kLdsPad;
        const float* Lt = cur + 2 * kTile * kLdsPad;
        const float* Dt = Lt + kTile;
        const f32x16 dp = mma_rows_x_regs(Gt, kLdsPad, lo, h, kv, zero16());  // dP (queries x keys)
        float* dsw = dsbuf + (it & 1) * kTile * kLdsPad + 32 * kw + lo;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ir = crow(r, h);
          float p = __expf(sv[r] - Lt[ir]);
          if (NEXT) sv[r] = scol[sel[min(i0 + kTile + ir, M - 1)] * ld];  // refill in place for tile it+1
          p = (i0 + ir < M) ? p : 0.f;
          const float ds = jvalid ? p * (dp[r] - Dt[ir]) * scale : 0.f;
          dsw[ir * kLdsPad] = ds;  // dS_all[query][this wave's 32 key columns]
          if (CS) csum += ds;
          {  // dV^T += dO^T P; accumulator tile dt, row rho <-> channel 4 rho + dt: one 16-byte LDS read per step
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(Gt + ir * kLdsPad + 4 * lo);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[dt] = mfma32(a4[dt], p, acc[dt]);
          }
        }
      }
    } else {
      if (BPREV) {
        const float* Qt = smem + bufB * kRbTile;
        const float* dsr = dsbuf + ((it - 1) & 1) * kTile * kLdsPad;
        // dQ slab of tile it-1: rows = its 32 queries, reduced index = the workgroup's 128 keys
        const f32x16 dqa = mma_rows_x_regs(dsr, kLdsPad, lo, h, kv, zero16());
        float* srow = myslab + (long)(i0 - kTile) * 128;
        if (ACUR) {  // tile it-1 is a full tile
#pragma unroll
          for (int r = 0; r < 16; ++r) srow[crow(r, h) * 128] = dqa[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (i0 - kTile + crow(r, h) < M) srow[crow(r, h) * 128] = dqa[r];
        }
        // dK^T += Q^T dS, dS of this wave's 32 keys read back from the LDS tile
        const float* dsc = dsr + 32 * kw + lo;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(Qt + crow(r, h) * kLdsPad + 4 * lo);
          const float dsv = dsc[crow(r, h) * kLdsPad];
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) acc[dt] = mfma32(a4[dt], dsv, acc[dt]);
        }
      }
    }
    if (NEXT) commit(smem + bufN * kRbTile);
    __syncthreads();
  };
  using T = std::true_type;
  using F = std::false_type;
  int bufA = 0;
  auto adv = [&]() { bufA = (bufA == 2) ? 0 : bufA + 1; };
  if (ntiles == 1) {
    body(0, bufA, T{}, F{}, F{});
    adv();
  } else {
    body(0, bufA, T{}, F{}, T{});
    adv();
    int it = 1;
    for (; it + 1 < ntiles; ++it) {
      body(it, bufA, T{}, T{}, T{});
      adv();
    }
    body(it, bufA, T{}, T{}, F{});
    adv();
  }
  body(ntiles, bufA, F{}, T{}, F{});

  if (CS && roleA) {
    const float ctot = csum + wave_xor32(csum);
    if (jvalid && h == 0) a.cs[(long)b * (N + a.nt) + j] = ctot;
  }
  if (jvalid) {
    float* orow = roleA ? a.dV + (long)b * a.dv_bs + (long)j * a.dv_rs : a.dK + (long)b * a.dk_bs + (long)j * a.dk_rs;
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // acc[dt][r] = channel 4 crow(r,h) + dt of this lane's key
      f32x4 x = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
      *reinterpret_cast<f32x4*>(orow + 4 * crow(r, h)) = x;
    }
  }
}

}  // namespace samble

using namespace samble;

extern "C" int samble_launch_bwd_rows(const float* Qs, const float* dOb, const float* lse_s, const float* delta,
                                      const float* K, long k_bs, long k_rs, const float* V, long v_bs, long v_rs, int B,
                                      int N, int M, float scale, float* dK, long dk_bs, long dk_rs, float* dV, long dv_bs,
                                      long dv_rs, float* slab, int nslab, const float* smap, int ld,
                                      const long long* idx, float* cs, int nt, hipStream_t stream) {
  {  // per call: cheap, and correct for every device / thread (no process-wide 'done' flag)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_rows_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_rows_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  const size_t lds = kRbLdsFloats * sizeof(float) + (size_t)M * 4;
  if (lds > 160 * 1024) return -22;
  RowsBwdArgs a{Qs, dOb, lse_s, delta, K, k_bs, k_rs, V, v_bs, v_rs, N, M, scale, dK, dk_bs, dk_rs,
                dV, dv_bs, dv_rs, slab, nslab, smap, ld, idx, cs, nt};
  Timed timed(kT_bwd_rows_f32, stream);
  if (cs) hipLaunchKernelGGL(bwd_rows_kernel<true>, dim3((N + 127) / 128, B), dim3(512), lds, stream, a);
  else hipLaunchKernelGGL(bwd_rows_kernel<false>, dim3((N + 127) / 128, B), dim3(512), lds, stream, a);
  return (int)hipGetLastError();
}
