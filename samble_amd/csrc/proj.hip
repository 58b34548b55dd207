// QKV projection of the sampler (reference models/downsample.py:116-137: three bias-free 1x1 Conv1d
// on x and on x || bin_tokens) and its backward, fp32 MFMA.
//
//   qkv[b][n][o] = sum_c W[o][c] * xt[b][c][n]     o in [0,3D): rows of W = [Wq; Wk; Wv], D = C = 128
//                                                   n < N: point n of x (B,C,N channel-major)
//                                                   n >= N: bin token n-N (tokens (C,nt), shared by the batch)
//   proj_fwd     wave = 32 points held in registers (coalesced channel-major loads), W streams
//                through LDS in 32-row tiles; output rows are point-major [Q|K|V] (what the
//                attention kernels read).  The nt token rows are batch-independent: computed once
//                (proj_tok_fwd) and copied into each cloud by its first workgroup, so x is never
//                concatenated or copied and the MFMA grid stays N/128 workgroups per cloud.
//   proj_dx      dx[b][c][n] = sum_o W[o][c] dqkv[b][n][o]: wave = 32 points, dqkv rows in registers
//                128 outputs at a time, W^T from LDS; writes channel-major dx directly.
//   proj_dw      dW[o][c] = sum_{b,n} dqkv[b][n][o] xt[b][c][n]: workgroup = 256 points of one cloud,
//                full 384x128 partial in registers (4 waves x 12 tiles), partials summed in a fixed
//                order by proj_dw_reduce (deterministic, no float atomics).
//   proj_tok_bwd dtokens[c][t] and the token rows' share of dW (nt <= 8 rows: VALU).
#include <type_traits>

#include "samble_dev.h"

namespace samble {

constexpr int kC = 128;
constexpr int kO = 384;

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
constexpr int kProjLdsFloats = 2 * kTile * kLdsPad;

// token rows are the same for every cloud: tokqkv[t][o] = sum_c W[o][c] tokens[c][t], computed once.
// One wave per output row o (lanes span the channels: coalesced 512-byte reads of the W row), the 8 token sums of
// a wave are independent shuffle trees.
__global__ __launch_bounds__(256) void proj_tok_fwd_kernel(const float* __restrict__ tokens, int nt,
                                                           const float* __restrict__ W, float* __restrict__ tokqkv) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const float w0 = W[(long)o * kC + lane], w1 = W[(long)o * kC + lane + 64];
  float p[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float tk0 = (t < nt) ? tokens[lane * nt + t] : 0.f;
    const float tk1 = (t < nt) ? tokens[(lane + 64) * nt + t] : 0.f;
    p[t] = fmaf(w0, tk0, w1 * tk1);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
    for (int t = 0; t < 8; ++t) p[t] += __shfl_xor(p[t], off, 64);
#pragma unroll
  for (int t = 0; t < 8; ++t)
    if (lane == 0 && t < nt) tokqkv[t * kO + o] = p[t];
}

__global__ __launch_bounds__(256, 2) void proj_fwd_kernel(const float* __restrict__ x, long x_bs, int N,
                                                          const float* __restrict__ tokqkv, int nt,
                                                          const float* __restrict__ W,  // (384,128) row-major
                                                          float* __restrict__ qkv, long o_bs, long o_rs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  int chunk, b;
  xcd_assign(chunk, b);
  // points past N-1 (last workgroup of a ragged cloud) are clamped: those lanes recompute and rewrite
  // point N-1's row bit for bit, so no store is predicated, the tile loop has no branch and the staged
  // W tile is awaited with a counted vmcnt instead of draining the output stores as well
  const int n = min(chunk * 128 + wave * 32 + lo, N - 1);

  if (chunk == 0) {  // this cloud's copy of the token rows
    for (int e = tid; e < nt * kO; e += 256)
      qkv[(long)b * o_bs + (long)(N + e / kO) * o_rs + (e % kO)] = tokqkv[e];
  }
  float xr[64];
#pragma unroll
  for (int kk = 0; kk < 64; ++kk) xr[kk] = x[(long)b * x_bs + (long)(64 * h + kk) * N + n];
  TileRegs wr;
  tile_load_issue(wr, W, kC, 0, kO, tid);
  tile_store_lds(wr, smem, kLdsPad, tid);
  __syncthreads();
  float* orow = qkv + (long)b * o_bs + (long)n * o_rs + 4 * h;
  auto body = [&](int t, auto next_c) {
    constexpr bool NEXT = decltype(next_c)::value;
    float* cur = smem + (t & 1) * kTile * kLdsPad;
    float* nxt = smem + ((t & 1) ^ 1) * kTile * kLdsPad;
    if (NEXT) {  // rows (t+1)*32 .. +31 of W always exist: unguarded loads
#pragma unroll
      for (int i = 0; i < TileRegs::kPer; ++i) {
        const int e = tid + 256 * i, r = e >> 5, c4 = e & 31;
        wr.v[i] = *reinterpret_cast<const f32x4*>(W + (long)((t + 1) * kTile + r) * kC + 4 * c4);
      }
    }
    // D[row = output o][col = point n]
    f32x16 acc = mma_rows_x_regs(cur, kLdsPad, lo, h, xr, zero16());
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
      *reinterpret_cast<f32x4*>(orow + t * kTile + 8 * g) = o;
    }
    if (NEXT) tile_store_lds(wr, nxt, kLdsPad, tid);
    __syncthreads();
  };
  for (int t = 0; t + 1 < kO / kTile; ++t) body(t, std::true_type{});
  body(kO / kTile - 1, std::false_type{});
}

// ------------------------------------------------------------------------------------------------
// dx
// ------------------------------------------------------------------------------------------------
constexpr int kDxStage = 64;                       // outputs (rows of W) per pipeline stage
constexpr int kDxLdsFloats = 2 * kDxStage * 128;   // two stages of W, [o][c]

// dx[b][c][n] = sum_o W[o][c] dqkv[b][n][o]: wave = 32 points, the 384-long contraction in 6 stages of
// 64 outputs.  W stage s+1 and the points' dqkv slice for it are fetched (registers) while stage s
// multiplies; one barrier per stage; clamped points instead of predicates (branch-free stage loop).
__global__ __launch_bounds__(256, 2) void proj_dx_kernel(const float* __restrict__ dqkv, long g_bs, long g_rs,
                                                         const float* __restrict__ W, int N,
                                                         float* __restrict__ dx, long dx_bs) {
  extern __shared__ __attribute__((aligned(16))) float Ws[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int n = min(blockIdx.x * 128 + wave * 32 + lo, N - 1);
  const float* grow = dqkv + (long)b * g_bs + (long)n * g_rs + 32 * h;  // this lane half's 32 outputs of a stage

  f32x16 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = zero16();
  f32x4 wreg[8], gcur[8], gnxt[8];
  auto issue = [&](int st, f32x4 (&g)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i, r = e >> 5, c4 = (e & 31) * 4;  // 64 rows x 32 float4
      wreg[i] = *reinterpret_cast<const f32x4*>(W + (long)(st * kDxStage + r) * kC + c4);
      g[i] = *reinterpret_cast<const f32x4*>(grow + st * kDxStage + 4 * i);
    }
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i, r = e >> 5, c4 = (e & 31) * 4;
      *reinterpret_cast<f32x4*>(buf + r * 128 + c4) = wreg[i];
    }
  };
  issue(0, gcur);
  commit(Ws);
  __syncthreads();
  auto body = [&](int st, auto next_c) {
    constexpr bool NEXT = decltype(next_c)::value;
    const float* cur = Ws + (st & 1) * kDxStage * 128;
    if (NEXT) issue(st + 1, gnxt);
    // D[row = channel c][col = point n] += sum_o W[o][c] * dqkv[n][o];  o = 32h + kk within the stage
    const float* wp = cur + (32 * h) * 128 + lo;
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
      const float bval = gcur[kk >> 2][kk & 3];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = mfma32(wp[kk * 128 + 32 * ct], bval, acc[ct]);
    }
    if (NEXT) {
      commit(Ws + ((st & 1) ^ 1) * kDxStage * 128);
#pragma unroll
      for (int i = 0; i < 8; ++i) gcur[i] = gnxt[i];
    }
    __syncthreads();
  };
  constexpr int kStages = kO / kDxStage;
  for (int st = 0; st + 1 < kStages; ++st) body(st, std::true_type{});
  body(kStages - 1, std::false_type{});
  float* out = dx + (long)b * dx_bs + n;  // clamped points rewrite point N-1's values
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(long)(32 * ct + crow(r, h)) * N] = acc[ct][r];
}

// ------------------------------------------------------------------------------------------------
// dW partials: workgroup = (cloud b, 256-point chunk); wave w owns output rows 96w .. 96w+95
// ------------------------------------------------------------------------------------------------
constexpr int kDwPts = 256;
constexpr int kDwLdsFloats = 2 * (kTile * 388 + kC * 33);  // dqkv tile [32 n][384 o] + x tile [128 c][32 n]

__global__ __launch_bounds__(256, 1) void proj_dw_kernel(const float* __restrict__ dqkv, long g_bs, long g_rs,
                                                         const float* __restrict__ x, long x_bs, int N,
                                                         float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int GS = 388;  // row stride of the dqkv tile (rows 16-byte aligned, column reads conflict-free)
  constexpr int XS = 33;
  constexpr int BUF = kTile * GS + kC * XS;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int n0 = blockIdx.x * kDwPts;

  f32x16 acc[3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = zero16();

  f32x4 gst[12];  // 32 rows x 96 float4 = 3072 float4 / 256 threads
  float xst[16];  // 128 channels x 32 points = 4096 floats / 256 threads
  auto issue = [&](int nn0) {
#pragma unroll
    for (int it = 0; it < 12; ++it) {
      const int e = tid + 256 * it;
      const int r = e / 96, c4 = (e % 96) * 4;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      gst[it] = (nn0 + r < N) ? *reinterpret_cast<const f32x4*>(dqkv + (long)b * g_bs + (long)(nn0 + r) * g_rs + c4) : z4;
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int e = tid + 256 * it;
      const int c = e >> 5, p = e & 31;
      xst[it] = (nn0 + p < N) ? x[(long)b * x_bs + (long)c * N + nn0 + p] : 0.f;
    }
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int it = 0; it < 12; ++it) {
      const int e = tid + 256 * it;
      const int r = e / 96, c4 = (e % 96) * 4;
      *reinterpret_cast<f32x4*>(buf + r * GS + c4) = gst[it];
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int e = tid + 256 * it;
      buf[kTile * GS + (e >> 5) * XS + (e & 31)] = xst[it];
    }
  };
  const int ntiles = kDwPts / kTile;
  issue(n0);
  commit(smem);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    float* cur = smem + (t & 1) * BUF;
    float* nxt = smem + ((t & 1) ^ 1) * BUF;
    if (t + 1 < ntiles) issue(n0 + (t + 1) * kTile);
    const float* gt = cur;
    const float* xt = cur + kTile * GS;
    // D[row = o][col = c] += sum_n dqkv[n][o] * x[c][n];  n = 16h + kk within the tile
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int nn = 16 * h + kk;
      float a[3], bb[4];
#pragma unroll
      for (int ot = 0; ot < 3; ++ot) a[ot] = gt[nn * GS + 96 * wave + 32 * ot + lo];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) bb[ct] = xt[(32 * ct + lo) * XS + nn];
#pragma unroll
      for (int ot = 0; ot < 3; ++ot)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ot][ct] = mfma32(a[ot], bb[ct], acc[ot][ct]);
    }
    if (t + 1 < ntiles) commit(nxt);
    __syncthreads();
  }
  float* out = part + ((long)b * gridDim.x + blockIdx.x) * kO * kC;
#pragma unroll
  for (int ot = 0; ot < 3; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = 96 * wave + 32 * ot + crow(r, h);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) out[(long)o * kC + 32 * ct + lo] = acc[ot][ct][r];
    }
}

// dW[o][c] = sum over the per-chunk partials (fixed order) + the token rows' share
//            sum_t gsum[t][o] * tokens[c][t]
// 49 152 outputs x (B * chunks) partials = 50 MB at B = 32: bandwidth work, so every load must be in flight at
// once.  Workgroup = one output row (32 float4 columns = 512 contiguous bytes of every partial) x 8 partial groups: thread
// (e4, g) sums partials g, g + 8, g + 16, ... (16 independent 16-byte loads in flight per round), the 8 group sums are
// added in index order through LDS.  (Round 4: 16 columns x 16 groups read 256-byte pieces: 17.1 us against the row form.)
// The token rows ride along (no launch of their own): every workgroup forms the gsum[t][o] of ITS output row o =
// sum_b dqkv[b][N+t][o] (index order) while its partial loads are in flight, and workgroups kO .. kO + nt - 1 are
// the token workgroups: all of gsum[t][.] for one token t, then dtokens[c][t] = sum_o W[o][c] gsum[t][o] in three
// parts of 128 outputs combined in part order (the arithmetic of the former proj_tok_bwd_kernel, bit for bit).
struct TokGrad {
  const float* dqkv;  // (B, N + nt, 384) gradient block: the token rows are read
  long g_bs, g_rs;
  int B, N;
  ProjW W;
  float* dtok;        // (128, nt)
};
__device__ __forceinline__ float tok_gsum(const TokGrad& tg, int t, int o) {
  const float* src = tg.dqkv + (long)(tg.N + t) * tg.g_rs + o;
  float s = 0.f;
  int b = 0;
  for (; b + 32 <= tg.B; b += 32) {  // 32 loads in flight, summed in index order
    float v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = src[(long)(b + u) * tg.g_bs];
#pragma unroll
    for (int u = 0; u < 32; ++u) s += v[u];
  }
  for (; b + 8 <= tg.B; b += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(long)(b + u) * tg.g_bs];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; b < tg.B; ++b) s += src[(long)b * tg.g_bs];
  return s;
}

__global__ __launch_bounds__(256) void proj_dw_reduce_kernel(const float* __restrict__ part, int nparts,
                                                             const TokGrad tg, const float* __restrict__ tokens, int nt,
                                                             float* __restrict__ dW) {
  __shared__ f32x4 red[8][33];
  __shared__ float gs[kO];
  __shared__ float ps[3][kC];
  if (blockIdx.x >= kO) {  // a token workgroup
    const int t = blockIdx.x - kO, tid = threadIdx.x;
    for (int o = tid; o < kO; o += 256) gs[o] = tok_gsum(tg, t, o);
    __syncthreads();
    // thread = (channel c, half): half 0 takes parts 0 and 2 of the 384 outputs, half 1 part 1; 16 W loads in flight
    const int c = tid & (kC - 1);
    for (int part3 = tid >> 7; part3 < 3; part3 += 2) {
      float acc = 0.f;
      for (int o0 = 128 * part3; o0 < 128 * part3 + 128; o0 += 64) {  // (one weight: 64 loads in flight)
        float wv[64];
        const float* wp = tg.W.row(o0) + c;
#pragma unroll
        for (int u = 0; u < 64; ++u) wv[u] = wp[u * kC];
#pragma unroll
        for (int u = 0; u < 64; ++u) acc = fmaf(wv[u], gs[o0 + u], acc);
      }
      ps[part3][c] = acc;
    }
    __syncthreads();
    if (tid < kC) tg.dtok[tid * nt + t] = (ps[0][tid] + ps[1][tid]) + ps[2][tid];
    return;
  }
  // one output ROW (128 floats = 512 contiguous bytes of every partial) per workgroup: thread (e4, g) = float4 column e4 of
  // the row, partials g, g + 8, g + 16, ... with 16 loads in flight; the 8 group sums are added in index order through LDS
  const int e4l = threadIdx.x & 31, g = threadIdx.x >> 5;
  if (threadIdx.x < nt) gs[threadIdx.x] = tok_gsum(tg, threadIdx.x, blockIdx.x);
  const int e4 = blockIdx.x * 32 + e4l;  // float4 column of the (384 x 128) matrix; grid covers it exactly
  const f32x4* p4 = reinterpret_cast<const f32x4*>(part) + e4;
  constexpr long kStride4 = (long)kO * kC / 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int p0 = g; p0 < nparts; p0 += 8 * 16) {
    f32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int p = p0 + 8 * u;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      v[u] = (p < nparts) ? p4[(long)p * kStride4] : z4;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  red[g][e4l] = s;
  __syncthreads();
  if (g == 0) {
    f32x4 tot = red[0][e4l];
#pragma unroll
    for (int k = 1; k < 8; ++k) tot += red[k][e4l];
    const int c = 4 * e4l;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      for (int t = 0; t < nt; ++t) tot[j] = fmaf(gs[t], tokens[(c + j) * nt + t], tot[j]);
    reinterpret_cast<f32x4*>(dW)[e4] = tot;
  }
}

}  // namespace samble

using namespace samble;

extern "C" int samble_launch_proj_fwd_tri(const float*, long, int, int, const float*, float*, int, const float*, const float*,
                                          const float*, void*, void*,
                                          float*, long, long, void*, void*, void*, void*, void*, int, hipStream_t);
extern "C" int samble_launch_proj_dx_tri(const float*, long, long, const float*, void*, int, int, int, float*, long, const float*, hipStream_t);
extern "C" int samble_launch_proj_dw_tri(const float*, long, long, const float*, long, int, int, float*, hipStream_t);

// wimg != null: room for the row image of W -> the split-bf16 kernel (proj_tri.hip)
extern "C" int samble_launch_proj_fwd(const float* x, long x_bs, int B, int N, const float* tokens, int nt,
                                      const float* W, const float* Wk, const float* Wv, float* qkv, long o_bs, long o_rs,
                                      float* ws, void* wimg, void* const* images, int q_only, void* wtr_out, hipStream_t s) {
  // Wk / Wv non-null (split-bf16 path only): W = Wq and the three 128 x 128 weights are tensors of their own
  const size_t lds = kProjLdsFloats * sizeof(float);
  float* tokqkv = ws;  // 8 x 384 floats
  if (wimg)  // images: {q_rm, k_rm, v_tr, k_tr | null, v_rm | null} or null; the token rows come with the W image
    return samble_launch_proj_fwd_tri(x, x_bs, B, N, tokens, tokqkv, nt, W, Wk, Wv, wimg, wtr_out, qkv, o_bs, o_rs, images ? images[0] : nullptr,
                                      images ? images[1] : nullptr, images ? images[2] : nullptr,
                                      images ? images[3] : nullptr, images ? images[4] : nullptr, images ? q_only : 0, s);
  if (Wk) return (int)hipErrorInvalidValue;
  if (nt > 0) hipLaunchKernelGGL(proj_tok_fwd_kernel, dim3(kO / 4), dim3(256), 0, s, tokens, nt, W, tokqkv);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(proj_fwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  Timed timed(kT_proj_fwd, s);
  hipLaunchKernelGGL(proj_fwd_kernel, dim3((N + 127) / 128, B), dim3(256), lds, s, x, x_bs, N, tokqkv, nt, W, qkv, o_bs,
                     o_rs);
  return (int)hipGetLastError();
}

extern "C" size_t samble_proj_bwd_ws_floats(int B, int N) {
  const size_t nparts = (size_t)B * ((N + kDwPts - 1) / kDwPts);
  return nparts * kO * kC + (size_t)kO * kC + 64;
}

extern "C" int samble_launch_proj_bwd(const float* dqkv, long g_bs, long g_rs, const float* x, long x_bs, int B, int N,
                                      const float* tokens, int nt, const float* W, const float* Wk, const float* Wv,
                                      float* dx, long dx_bs, float* dW, float* dtok, float* ws, void* wtr,
                                      const void* wtr_ready, const float* dx_residual, hipStream_t s) {
  if (dx_residual && !(dx && wtr)) return (int)hipErrorInvalidValue;  // (the split-bf16 kernel carries the epilogue)
  // Wk / Wv non-null: W = Wq, three tensors -- only with wtr_ready (nothing else reads W as one block then)
  if (Wk && !(wtr && wtr_ready)) return (int)hipErrorInvalidValue;
  // wtr_ready: the transposed image of W as the forward's prologue wrote it (then wtr is not used)
  const size_t lds_dx = kDxLdsFloats * sizeof(float), lds_dw = kDwLdsFloats * sizeof(float);
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(proj_dw_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dw);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(proj_dx_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dx);
    if (e != hipSuccess) return (int)e;
  }
  const int chunks = (N + kDwPts - 1) / kDwPts;
  float* part = ws;
  if (dx && wtr) {  // room for the transposed image of W -> the split-bf16 kernel
    const int rc = samble_launch_proj_dx_tri(dqkv, g_bs, g_rs, W, wtr_ready ? const_cast<void*>(wtr_ready) : wtr,
                                             wtr_ready != nullptr, B, N, dx, dx_bs, dx_residual, s);
    if (rc) return rc;
  } else if (dx) {
    Timed timed(kT_proj_dx, s);
    hipLaunchKernelGGL(proj_dx_kernel, dim3((N + 127) / 128, B), dim3(256), lds_dx, s, dqkv, g_bs, g_rs, W, N, dx, dx_bs);
  }
  if (dW) {
    Timed timed(kT_proj_dw, s);
    if (wtr) {  // split-bf16 mode
      const int rc = samble_launch_proj_dw_tri(dqkv, g_bs, g_rs, x, x_bs, B, N, part, s);
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL(proj_dw_kernel, dim3(chunks, B), dim3(256), lds_dw, s, dqkv, g_bs, g_rs, x, x_bs, N, part);
    }
    const TokGrad tg{dqkv, g_bs, g_rs, B, N, proj_w(W, Wk, Wv), dtok};
    hipLaunchKernelGGL(proj_dw_reduce_kernel, dim3(kO + nt), dim3(256), 0, s, part, B * chunks, tg, tokens, nt,
                       dW);
  }
  return (int)hipGetLastError();
}
