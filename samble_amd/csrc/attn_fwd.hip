// Flash-style attention forward for the SAMBLE sampler (reference models/downsample.py:139-153,
// 242-252): for every one of the N query points, softmax over the N point keys AND the nt bin-token
// keys of (Q K^T / sqrt(D)), times V.  The (B,1,N,N+nt) map of the reference is never materialised;
// per row the kernel keeps the running max / sum-exp and writes
//     O   (B,N,D)   = softmax(S) V          (row i is what the reference gathers for a sampled i)
//     lse (B,N)     = log sum_j exp(S_ij)   (lets later kernels rebuild any A_ij = exp(S_ij - lse_i))
//     tok (B,N,nt)  = S[:, N:N+nt]          (attention_bins_beforesoftmax, downsample.py:149-152)
//
// Mapping: one workgroup = 8 waves = 256 query rows (one workgroup per CU, two waves per SIMD), one
// wave = 32 rows.  Keys/values stream through LDS in 32-row tiles shared by the 8 waves (the tile
// reads of a launch are 16 MB per cloud x its 8 workgroups; with 4-wave workgroups twice that, and
// the ablation of tools/ablate_attn_fwd.py showed the tile load latency exposed) (double buffered, loads for tile t+1 issued before the
// MFMAs of tile t).  Per tile and wave: S^T = K_tile Q^T (64 MFMA, queries on the lane axis so the
// row statistics are lane-local), softmax in registers, O^T += V_tile^T P^T (64 MFMA) with the S
// accumulator registers fed back directly as MFMA B operands.  Bound: fp32 MFMA (157 TFLOP/s).
#include "samble_dev.h"

namespace samble {

constexpr int kFwdLdsFloats = 2 * (kTile * kLdsPad + kTile * 128);

// ABL (timing-only ablation builds, wrong outputs): 0 = real kernel, 1 = softmax skipped,
// 2 = tile staging skipped (no global loads / LDS commits), 3 = staging and barrier skipped,
// 4 = global loads kept but LDS commit skipped, 5 = LDS commit kept but global loads skipped
template <int ABL, int NW = 8>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_kernel(
    const float* __restrict__ Q, long q_bs, long q_rs, const float* __restrict__ K, long k_bs, long k_rs,
    const float* __restrict__ V, long v_bs, long v_rs, int N, int NK, float scale, float* __restrict__ O,
    float* __restrict__ lse, float* __restrict__ tok, int nt, float* __restrict__ row_std) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kBuf = kTile * kLdsPad + kTile * 128;  // one K tile + one V tile

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int qrow = chunk * (32 * NW) + wave * 32 + lo;
  const bool qvalid = qrow < N;

  const float* Kb = K + (long)b * k_bs;
  const float* Vb = V + (long)b * v_bs;

  float q[64];
  if (qvalid) {
    load_row_half(Q + (long)b * q_bs + (long)qrow * q_rs, h, q);
  } else {
#pragma unroll
    for (int i = 0; i < 64; ++i) q[i] = 0.f;
  }

  f32x16 oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) oacc[dt] = zero16();
  float m = kNegInf, l = 0.f;
  // idx_mode row_std (models/downsample.py:319-320): first and second moments of the POINT columns
  // of each row, kept relative to the running max like l
  const bool want_std = row_std != nullptr;
  float sp1 = 0.f, sp2 = 0.f;

  const int ntiles = (NK + kTile - 1) / kTile;
  TileRegsT<64 * NW> kr, vr;
  tile_load_issue(kr, Kb, k_rs, 0, NK, tid);
  tile_load_issue(vr, Vb, v_rs, 0, NK, tid);
  tile_store_lds(kr, smem, kLdsPad, tid);
  tile_store_lds(vr, smem + kTile * kLdsPad, 128, tid);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    float* Kc = smem + (t & 1) * kBuf;
    float* Vc = Kc + kTile * kLdsPad;
    float* Kn = smem + ((t & 1) ^ 1) * kBuf;
    float* Vn = Kn + kTile * kLdsPad;
    const int j0 = t * kTile;
    if ((ABL < 2 || ABL == 4) && t + 1 < ntiles) {
      tile_load_issue(kr, Kb, k_rs, j0 + kTile, NK, tid);
      tile_load_issue(vr, Vb, v_rs, j0 + kTile, NK, tid);
    }
    // S^T tile: rows = keys of this tile, cols = this wave's 32 queries
    f32x16 s = mma_rows_x_regs(Kc, kLdsPad, lo, h, q, zero16());

    if (ABL == 1) {
#pragma unroll
      for (int t16 = 0; t16 < 16; ++t16) mma_tileT_step(Vc, 128, lo, h, t16, s[t16], oacc);
    } else {
    const bool tail = (j0 + kTile > N);  // tile holds token keys and/or padding (wave-uniform)
    float mt = kNegInf;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = s[r] * scale;
      if (tail) {
        const int j = j0 + crow(r, h);
        if (j >= NK) v = kNegInf;
        if (j >= N && j < NK && qvalid) tok[((long)b * N + qrow) * nt + (j - N)] = v;
      }
      s[r] = v;
      mt = fmaxf(mt, v);
    }
    mt = fmaxf(mt, wave_xor32(mt));
    const float mnew = fmaxf(m, mt);
    if (__any(mnew != m)) {  // the running max moved for some row of this wave: rescale (rare after the first tiles)
      const float alpha = __expf(m - mnew);
      l *= alpha;
      sp1 *= alpha;
      sp2 *= alpha * alpha;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oacc[dt] *= alpha;
      m = mnew;
    }
    // O^T += V_tile^T P^T, one accumulator register (= 2 key rows) at a time: the exp of register
    // t+1 issues while the four MFMAs of register t run
    float ps = 0.f;
#pragma unroll
    for (int t16 = 0; t16 < 16; ++t16) {
      const float p = __expf(s[t16] - m);
      ps += p;
      if (want_std && (j0 + crow(t16, h) < N)) {
        sp1 += p;
        sp2 = fmaf(p, p, sp2);
      }
      mma_tileT_step(Vc, 128, lo, h, t16, p, oacc);
    }
    l += ps;
    }

    if (ABL == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(kr.v[i][0]), "v"(vr.v[i][0]));
    }
    if ((ABL < 2 || ABL == 5) && t + 1 < ntiles) {
      tile_store_lds(kr, Kn, kLdsPad, tid);
      tile_store_lds(vr, Vn, 128, tid);
    }
    if (ABL < 3) __syncthreads();
  }

  const float ltot = l + wave_xor32(l);
  const float inv = 1.f / ltot;
  if (qvalid) {
    float* orow = O + ((long)b * N + qrow) * 128;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 o = {oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv,
                   oacc[dt][4 * g + 3] * inv};
        *reinterpret_cast<f32x4*>(orow + 32 * dt + 8 * g + 4 * h) = o;
      }
    }
    if (h == 0) lse[(long)b * N + qrow] = m + __logf(ltot);
    if (want_std) {  // unbiased std over the N point columns of A = p / l  (torch.std default)
      const float s1 = (sp1 + wave_xor32(sp1)) * inv, s2 = (sp2 + wave_xor32(sp2)) * inv * inv;
      const float mean = s1 / N;
      if (h == 0) row_std[(long)b * N + qrow] = sqrtf(fmaxf((s2 - N * mean * mean) / (N - 1), 0.f));
    }
  }
}

// idx_mode col_sum (models/downsample.py:315-318): column sums of the point-to-point attention block,
// colsum_j = sum_i exp(scale <Q_i, K_j> - lse_i).  Key-stationary: one wave = 32 keys (rows in
// registers), the N query rows stream through LDS; needs the forward's lse.  Fixed summation order.
__global__ __launch_bounds__(256, 2) void attn_colsum_kernel(const float* __restrict__ Q, long q_bs, long q_rs,
                                                             const float* __restrict__ K, long k_bs, long k_rs,
                                                             const float* __restrict__ lse, int N, float scale,
                                                             float* __restrict__ colsum) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kBuf = kTile * kLdsPad + kTile;  // Q tile + lse[32]
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  const int j = chunk * 128 + wave * 32 + lo;
  const bool jvalid = j < N;
  float kreg[64];
  if (jvalid) {
    load_row_half(K + (long)b * k_bs + (long)j * k_rs, h, kreg);
  } else {
#pragma unroll
    for (int i = 0; i < 64; ++i) kreg[i] = 0.f;
  }
  const float* Qb = Q + (long)b * q_bs;
  const int ntiles = (N + kTile - 1) / kTile;
  TileRegs qr;
  float st = 0.f;
  auto issue = [&](int i0) {
    tile_load_issue(qr, Qb, q_rs, i0, N, tid);
    if (tid < 32) st = (i0 + tid < N) ? lse[(long)b * N + i0 + tid] : 0.f;
  };
  auto commit = [&](float* buf) {
    tile_store_lds(qr, buf, kLdsPad, tid);
    if (tid < 32) buf[kTile * kLdsPad + tid] = st;
  };
  issue(0);
  commit(smem);
  __syncthreads();
  float sum = 0.f;
  for (int t = 0; t < ntiles; ++t) {
    float* cur = smem + (t & 1) * kBuf;
    float* nxt = smem + ((t & 1) ^ 1) * kBuf;
    const float* Lt = cur + kTile * kLdsPad;
    const int i0 = t * kTile;
    if (t + 1 < ntiles) issue(i0 + kTile);
    f32x16 s = mma_rows_x_regs(cur, kLdsPad, lo, h, kreg, zero16());  // S (queries x keys)
    const bool tail = i0 + kTile > N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ir = crow(r, h);
      float p = __expf(s[r] * scale - Lt[ir]);
      if (tail && (i0 + ir >= N)) p = 0.f;
      sum += p;
    }
    if (t + 1 < ntiles) commit(nxt);
    __syncthreads();
  }
  sum += wave_xor32(sum);
  if (jvalid && h == 0) colsum[(long)b * N + j] = sum;
}

}  // namespace samble

using namespace samble;

extern "C" int samble_launch_attn_colsum(const float* Q, long q_bs, long q_rs, const float* K, long k_bs, long k_rs,
                                         const float* lse, int B, int N, float scale, float* colsum,
                                         hipStream_t stream) {
  const size_t lds = 2 * (kTile * kLdsPad + kTile) * sizeof(float);
  hipLaunchKernelGGL(attn_colsum_kernel, dim3((N + 127) / 128, B), dim3(256), lds, stream, Q, q_bs, q_rs, K, k_bs, k_rs,
                     lse, N, scale, colsum);
  return (int)hipGetLastError();
}

extern "C" int samble_launch_attn_fwd(const float* Q, long q_bs, long q_rs, const float* K, long k_bs, long k_rs,
                                      const float* V, long v_bs, long v_rs, int B, int N, int NK, float scale, float* O,
                                      float* lse, float* tok, int nt, float* row_std, hipStream_t stream) {
  const size_t lds = kFwdLdsFloats * sizeof(float);
  auto kern = attn_fwd_kernel<0>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid((N + 255) / 256, B);
  Timed timed(kT_attn_fwd, stream);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, Q, q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs, N, NK, scale, O,
                     lse, tok, nt, row_std);
  return (int)hipGetLastError();
}
