// Global self-attention over H heads of depth D = C / H for the feature-learning layer Point2PointAttention
// (reference models/attention.py:253-355): per head  softmax((mul * q k^T + bias_j) / sqrt(D)) v  for every one of the
// N points, forward and backward, without the (B, H, N, N) map.
//
//   asm "dot":  mul = 1,  bias = 0                      (attention.py:341)
//   asm "l2":   mul = 2,  bias_j = -|k_j|^2             (attention.py:343, utils/ops.py l2_global: -|q - k|^2; the row
//   asm "l2+":  mul = -2, bias_j = +|k_j|^2              term |q_i|^2 is constant along a softmax row and drops out)
//
// A wave owns 32 rows of ONE head (32 queries in the forward and the dQ kernel, 32 keys in the dK / dV kernel); the
// contraction over the head's D channels is D / 2 k-steps of v_mfma_f32_32x32x2_f32 (true fp32 products: no operand
// splitting on this secondary path).  Depths below 32 are padded with zero channels, 64 and 128 take two / four
// accumulator blocks (NB = ceil(D / 32)).  Everything is computed transposed (S^T = K Q^T, O^T += V^T P^T, ...) so
// that a query's softmax statistics live in one lane pair and the finished S / P / dS accumulators are MFMA B
// operands as they are (samble_dev.h: operand layouts).  The tiles of the other side (32 rows x D channels) go
// through LDS, shared by the workgroup's four waves; no pipelining beyond the next tile's loads in flight -- the layer
// is not selected by any shipped configuration, it only must not cost H times the work any more.
#include "samble_dev.h"

namespace samble {

struct HeadsArgs {
  const float* Q; long q_bs, q_rs;   // (B, N, H * D) rows, head h = columns h * D ..
  const float* K; long k_bs, k_rs;
  const float* V; long v_bs, v_rs;
  const float* bias;                 // (B, H, N) or null
  float mul, scale;
  int N, H, D;
  float* O; long o_bs, o_rs;         // forward: output rows; backward: the forward's output
  float* lse;                        // (B, H, N)
  // backward
  const float* dO; long g_bs, g_rs;
  float* delta;                      // (B, H, N) workspace: rowsum(dO * O) per head
  float* dQ; long dq_bs, dq_rs;
  float* dK; long dk_bs, dk_rs;
  float* dV; long dv_bs, dv_rs;
  float* cs;                         // (B, H, N) column sums of dS (what d bias is), or null
};

template <int NB>
struct HeadsGeom {
  static constexpr int DP = 32 * NB;      // padded depth
  static constexpr int KS = DP / 2;       // k-steps of a contraction over the channels: step i = channels i, i + KS
  static constexpr int LD = DP + 1;       // LDS row stride of a 32-row tile (odd: both access patterns conflict-free)
  static constexpr int kTileFloats = 32 * LD;
};

// the 32 x D tile rows row0 .. row0 + 31 of X (head column offset already applied) -> registers of the 256 threads
template <int NB>
struct HeadsTileRegs {
  f32x4 v[NB];
};
template <int NB>
__device__ __forceinline__ void heads_tile_load(HeadsTileRegs<NB>& r, const float* __restrict__ X, long rs, int row0, int N,
                                                int D, int tid) {
  const int row = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int c = 32 * nb + c4;
    r.v[nb] = (row0 + row < N && c < D) ? *reinterpret_cast<const f32x4*>(X + (long)(row0 + row) * rs + c)
                                        : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
template <int NB>
__device__ __forceinline__ void heads_tile_store(const HeadsTileRegs<NB>& r, float* __restrict__ tile, int tid) {
  const int row = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[row * HeadsGeom<NB>::LD + 32 * nb + c4 + e] = r.v[nb][e];
}

// the lane's B operand of a contraction over the channels: row `row` of X, channels i + KS * h (16-byte loads)
template <int NB>
__device__ __forceinline__ void heads_row_regs(const float* __restrict__ X, long rs, int row, int N, int D, int h,
                                               float (&out)[HeadsGeom<NB>::KS]) {
  constexpr int KS = HeadsGeom<NB>::KS;
#pragma unroll
  for (int i4 = 0; i4 < KS / 4; ++i4) {
    const int c = KS * h + 4 * i4;
    const f32x4 v = (row < N && c < D) ? *reinterpret_cast<const f32x4*>(X + (long)row * rs + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) out[4 * i4 + e] = v[e];
  }
}

// acc (32 tile rows x 32 lane columns) = tile (rows x channels) . regs^T: A = tile[lo][i + KS h], B = regs[i]
template <int NB>
__device__ __forceinline__ f32x16 heads_rows_x_regs(const float* __restrict__ tile, int lo, int h,
                                                    const float (&regs)[HeadsGeom<NB>::KS]) {
  constexpr int KS = HeadsGeom<NB>::KS, LD = HeadsGeom<NB>::LD;
  f32x16 acc = zero16();
  const float* row = tile + lo * LD + KS * h;
#pragma unroll
  for (int i = 0; i < KS; ++i) acc = mfma32(row[i], regs[i], acc);
  return acc;
}
// acc[nb] (channels 32 nb .. x lane columns) += tile^T (channels x 32 tile rows) . x: A = tile[crow(r, h)][32 nb + lo],
// B = x[r] (a finished 32 x 32 accumulator whose rows are the tile's rows)
template <int NB>
__device__ __forceinline__ void heads_tileT_x_acc(const float* __restrict__ tile, int lo, int h, const f32x16& x,
                                                  f32x16 (&acc)[NB]) {
  constexpr int LD = HeadsGeom<NB>::LD;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float* row = tile + crow(r, h) * LD + lo;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma32(row[32 * nb], x[r], acc[nb]);
  }
}
// acc^T rows = channels 32 nb + crow(r, h), column = the lane's row `row` of an (., H * D) matrix
template <int NB>
__device__ __forceinline__ void heads_store_T(float* __restrict__ X, long rs, int row, int D, int h, const f32x16 (&acc)[NB],
                                              float f) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int c = 32 * nb + 8 * r4 + 4 * h;
      if (c < D)
        *reinterpret_cast<f32x4*>(X + (long)row * rs + c) =
            f32x4{acc[nb][4 * r4] * f, acc[nb][4 * r4 + 1] * f, acc[nb][4 * r4 + 2] * f, acc[nb][4 * r4 + 3] * f};
    }
}

__device__ __forceinline__ float heads_other_half(float v) { return __shfl_xor(v, 32, 64); }

// ---------------------------------------------------------------------------------------------------------------
// forward: grid (ceil(N / 128), H, B), 256 threads; wave = 32 queries of head blockIdx.y
// ---------------------------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256) void attn_heads_fwd_kernel(const HeadsArgs a) {
  using G = HeadsGeom<NB>;
  __shared__ float Kt[G::kTileFloats], Vt[G::kTileFloats], bt[32];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z, N = a.N, D = a.D;
  const int qrow = blockIdx.x * 128 + wave * 32 + lo;
  const float* Kb = a.K + (long)b * a.k_bs + head * D;
  const float* Vb = a.V + (long)b * a.v_bs + head * D;
  const float* bias = a.bias ? a.bias + ((long)b * a.H + head) * N : nullptr;
  float q[G::KS];
  heads_row_regs<NB>(a.Q + (long)b * a.q_bs + head * D, a.q_rs, qrow, N, D, h, q);
  f32x16 oacc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) oacc[nb] = zero16();
  float m = kNegInf, l = 0.f;
  const int ntiles = (N + 31) / 32;
  HeadsTileRegs<NB> kr, vr;
  heads_tile_load<NB>(kr, Kb, a.k_rs, 0, N, D, tid);
  heads_tile_load<NB>(vr, Vb, a.v_rs, 0, N, D, tid);
  float br = (bias && tid < 32 && tid < N) ? bias[tid] : 0.f;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();  // the previous tile's reads are done
    heads_tile_store<NB>(kr, Kt, tid);
    heads_tile_store<NB>(vr, Vt, tid);
    if (tid < 32) bt[tid] = br;
    __syncthreads();
    if (t + 1 < ntiles) {
      heads_tile_load<NB>(kr, Kb, a.k_rs, 32 * (t + 1), N, D, tid);
      heads_tile_load<NB>(vr, Vb, a.v_rs, 32 * (t + 1), N, D, tid);
      br = (bias && tid < 32 && 32 * (t + 1) + tid < N) ? bias[32 * (t + 1) + tid] : 0.f;
    }
    f32x16 s = heads_rows_x_regs<NB>(Kt, lo, h, q);  // S^T: rows = keys of the tile, column = the lane's query
    float mt = kNegInf;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 * t + crow(r, h);
      s[r] = key < N ? (a.mul * s[r] + bt[crow(r, h)]) * a.scale : kNegInf;
      mt = fmaxf(mt, s[r]);
    }
    mt = fmaxf(mt, heads_other_half(mt));
    const float mnew = fmaxf(m, mt);   // finite from the first tile on (key 0 exists)
    const float corr = __expf(m - mnew);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __expf(s[r] - mnew);
      ps += s[r];
    }
    ps += heads_other_half(ps);
    l = l * corr + ps;
    m = mnew;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[nb][r] *= corr;
    heads_tileT_x_acc<NB>(Vt, lo, h, s, oacc);  // O^T += V^T P^T
  }
  if (qrow < N) {
    heads_store_T<NB>(a.O + (long)b * a.o_bs + head * D, a.o_rs, qrow, D, h, oacc, 1.f / l);
    if (h == 0) a.lse[((long)b * a.H + head) * N + qrow] = m + __logf(l);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward, query-stationary: delta, dQ.  grid (ceil(N / 128), H, B); wave = 32 queries
// ---------------------------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256) void attn_heads_dq_kernel(const HeadsArgs a) {
  using G = HeadsGeom<NB>;
  __shared__ float Kt[G::kTileFloats], Vt[G::kTileFloats], bt[32];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z, N = a.N, D = a.D;
  const int qrow = blockIdx.x * 128 + wave * 32 + lo;
  const float* Kb = a.K + (long)b * a.k_bs + head * D;
  const float* Vb = a.V + (long)b * a.v_bs + head * D;
  const float* bias = a.bias ? a.bias + ((long)b * a.H + head) * N : nullptr;
  float q[G::KS], go[G::KS];
  heads_row_regs<NB>(a.Q + (long)b * a.q_bs + head * D, a.q_rs, qrow, N, D, h, q);
  heads_row_regs<NB>(a.dO + (long)b * a.g_bs + head * D, a.g_rs, qrow, N, D, h, go);
  float delta;
  {
    float o[G::KS];
    heads_row_regs<NB>(a.O + (long)b * a.o_bs + head * D, a.o_rs, qrow, N, D, h, o);
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < G::KS; ++i) d = fmaf(go[i], o[i], d);
    delta = d + heads_other_half(d);
  }
  const long srow = ((long)b * a.H + head) * N + min(qrow, N - 1);
  const float my_lse = a.lse[srow];
  if (qrow < N && h == 0) a.delta[srow] = delta;
  f32x16 dq[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) dq[nb] = zero16();
  const int ntiles = (N + 31) / 32;
  HeadsTileRegs<NB> kr, vr;
  heads_tile_load<NB>(kr, Kb, a.k_rs, 0, N, D, tid);
  heads_tile_load<NB>(vr, Vb, a.v_rs, 0, N, D, tid);
  float br = (bias && tid < 32 && tid < N) ? bias[tid] : 0.f;
  const float gscale = a.scale * a.mul;  // d logit / d (q . k)
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    heads_tile_store<NB>(kr, Kt, tid);
    heads_tile_store<NB>(vr, Vt, tid);
    if (tid < 32) bt[tid] = br;
    __syncthreads();
    if (t + 1 < ntiles) {
      heads_tile_load<NB>(kr, Kb, a.k_rs, 32 * (t + 1), N, D, tid);
      heads_tile_load<NB>(vr, Vb, a.v_rs, 32 * (t + 1), N, D, tid);
      br = (bias && tid < 32 && 32 * (t + 1) + tid < N) ? bias[32 * (t + 1) + tid] : 0.f;
    }
    f32x16 s = heads_rows_x_regs<NB>(Kt, lo, h, q);        // S^T
    const f32x16 dp = heads_rows_x_regs<NB>(Vt, lo, h, go);  // dP^T = V dO^T
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 * t + crow(r, h);
      const float p = key < N ? __expf((a.mul * s[r] + bt[crow(r, h)]) * a.scale - my_lse) : 0.f;
      s[r] = p * (dp[r] - delta) * gscale;                 // dS^T (w.r.t. q . k)
    }
    heads_tileT_x_acc<NB>(Kt, lo, h, s, dq);               // dQ^T += K^T dS^T
  }
  if (qrow < N) heads_store_T<NB>(a.dQ + (long)b * a.dq_bs + head * D, a.dq_rs, qrow, D, h, dq, 1.f);
}

// ---------------------------------------------------------------------------------------------------------------
// backward, key-stationary: dK, dV, column sums of dS.  grid (ceil(N / 128), H, B); wave = 32 keys
// ---------------------------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256) void attn_heads_dkv_kernel(const HeadsArgs a) {
  using G = HeadsGeom<NB>;
  __shared__ float Qt[G::kTileFloats], Gt[G::kTileFloats], lt[32], dt[32];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lo = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z, N = a.N, D = a.D;
  const int krow = blockIdx.x * 128 + wave * 32 + lo;
  const float* Qb = a.Q + (long)b * a.q_bs + head * D;
  const float* Gb = a.dO + (long)b * a.g_bs + head * D;
  const long sbase = ((long)b * a.H + head) * N;
  float k[G::KS], v[G::KS];
  heads_row_regs<NB>(a.K + (long)b * a.k_bs + head * D, a.k_rs, krow, N, D, h, k);
  heads_row_regs<NB>(a.V + (long)b * a.v_bs + head * D, a.v_rs, krow, N, D, h, v);
  const float my_bias = (a.bias && krow < N) ? a.bias[sbase + krow] : 0.f;
  f32x16 dk[NB], dv[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) dk[nb] = dv[nb] = zero16();
  float cs = 0.f;
  const int ntiles = (N + 31) / 32;
  HeadsTileRegs<NB> qr, gr;
  heads_tile_load<NB>(qr, Qb, a.q_rs, 0, N, D, tid);
  heads_tile_load<NB>(gr, Gb, a.g_rs, 0, N, D, tid);
  float lr = (tid < 32 && tid < N) ? a.lse[sbase + tid] : 0.f, dr = (tid < 32 && tid < N) ? a.delta[sbase + tid] : 0.f;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    heads_tile_store<NB>(qr, Qt, tid);
    heads_tile_store<NB>(gr, Gt, tid);
    if (tid < 32) {
      lt[tid] = lr;
      dt[tid] = dr;
    }
    __syncthreads();
    if (t + 1 < ntiles) {
      heads_tile_load<NB>(qr, Qb, a.q_rs, 32 * (t + 1), N, D, tid);
      heads_tile_load<NB>(gr, Gb, a.g_rs, 32 * (t + 1), N, D, tid);
      const int i = 32 * (t + 1) + tid;
      lr = (tid < 32 && i < N) ? a.lse[sbase + i] : 0.f;
      dr = (tid < 32 && i < N) ? a.delta[sbase + i] : 0.f;
    }
    f32x16 s = heads_rows_x_regs<NB>(Qt, lo, h, k);        // S: rows = queries of the tile, column = the lane's key
    f32x16 dp = heads_rows_x_regs<NB>(Gt, lo, h, v);       // dP = dO V^T
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = 32 * t + crow(r, h);
      const float p = i < N ? __expf((a.mul * s[r] + my_bias) * a.scale - lt[crow(r, h)]) : 0.f;
      s[r] = p;
      dp[r] = p * (dp[r] - dt[crow(r, h)]) * a.scale;      // dS w.r.t. the logit's numerator (mul q.k + bias)
      cs += dp[r];
    }
    heads_tileT_x_acc<NB>(Gt, lo, h, s, dv);               // dV^T += dO^T P
    heads_tileT_x_acc<NB>(Qt, lo, h, dp, dk);              // dK^T += Q^T dS
  }
  cs += heads_other_half(cs);
  if (krow < N) {
    heads_store_T<NB>(a.dK + (long)b * a.dk_bs + head * D, a.dk_rs, krow, D, h, dk, a.mul);
    heads_store_T<NB>(a.dV + (long)b * a.dv_bs + head * D, a.dv_rs, krow, D, h, dv, 1.f);
    if (a.cs && h == 0) a.cs[sbase + krow] = cs;
  }
}

}  // namespace samble

using namespace samble;

template <int NB>
static int launch_heads(const HeadsArgs& a, int B, bool backward, hipStream_t s) {
  const dim3 grid((a.N + 127) / 128, a.H, B);
  if (!backward) {
    hipLaunchKernelGGL(attn_heads_fwd_kernel<NB>, grid, dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL(attn_heads_dq_kernel<NB>, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(attn_heads_dkv_kernel<NB>, grid, dim3(256), 0, s, a);
  }
  return (int)hipGetLastError();
}

static int launch_heads_any(const HeadsArgs& a, int B, bool backward, hipStream_t s) {
  const int D = a.D;
  if (D < 4 || D > 128 || (D & 3) || a.H < 1 || a.N < 1 || B < 1) return (int)hipErrorInvalidValue;
  if (D <= 32) return launch_heads<1>(a, B, backward, s);
  if (D <= 64) return launch_heads<2>(a, B, backward, s);
  return launch_heads<4>(a, B, backward, s);
}

extern "C" int samble_launch_attn_heads_fwd(const float* Q, long q_bs, long q_rs, const float* K, long k_bs, long k_rs,
                                            const float* V, long v_bs, long v_rs, const float* bias, float mul, float scale,
                                            int B, int N, int H, int D, float* O, long o_bs, long o_rs, float* lse,
                                            hipStream_t s) {
  HeadsArgs a{};
  a.Q = Q; a.q_bs = q_bs; a.q_rs = q_rs;
  a.K = K; a.k_bs = k_bs; a.k_rs = k_rs;
  a.V = V; a.v_bs = v_bs; a.v_rs = v_rs;
  a.bias = bias; a.mul = mul; a.scale = scale;
  a.N = N; a.H = H; a.D = D;
  a.O = O; a.o_bs = o_bs; a.o_rs = o_rs;
  a.lse = lse;
  return launch_heads_any(a, B, false, s);
}

extern "C" int samble_launch_attn_heads_bwd(const float* Q, long q_bs, long q_rs, const float* K, long k_bs, long k_rs,
                                            const float* V, long v_bs, long v_rs, const float* bias, float mul, float scale,
                                            int B, int N, int H, int D, const float* O, long o_bs, long o_rs,
                                            const float* lse, const float* dO, long g_bs, long g_rs, float* delta,
                                            float* dQ, long dq_bs, long dq_rs, float* dK, long dk_bs, long dk_rs, float* dV,
                                            long dv_bs, long dv_rs, float* cs, hipStream_t s) {
  HeadsArgs a{};
  a.Q = Q; a.q_bs = q_bs; a.q_rs = q_rs;
  a.K = K; a.k_bs = k_bs; a.k_rs = k_rs;
  a.V = V; a.v_bs = v_bs; a.v_rs = v_rs;
  a.bias = bias; a.mul = mul; a.scale = scale;
  a.N = N; a.H = H; a.D = D;
  a.O = const_cast<float*>(O); a.o_bs = o_bs; a.o_rs = o_rs;
  a.lse = const_cast<float*>(lse);
  a.dO = dO; a.g_bs = g_bs; a.g_rs = g_rs;
  a.delta = delta;
  a.dQ = dQ; a.dq_bs = dq_bs; a.dq_rs = dq_rs;
  a.dK = dK; a.dk_bs = dk_bs; a.dk_rs = dk_rs;
  a.dV = dV; a.dv_bs = dv_bs; a.dv_rs = dv_rs;
  a.cs = cs;
  return launch_heads_any(a, B, true, s);
}
