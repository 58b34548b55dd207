// Everything of the fused EdgeConv (csrc/edgeconv.hip, reference models/embedding.py:7-39) that is not its two MLP
// sweeps: the closed forms of the two BatchNorms (batch statistics over the B N K edges from per-point sums, their
// backward corrections), the activation, and the per-point gradients -- as a handful of HIP kernels over (points, 64)
// rows instead of ~250 torch launches per layer and step (profiles/r04a: 5 of the block's 14.8 ms were this glue and the
// host-bound gaps between its tiny kernels).
//
// Per-channel sums over all points are formed in DOUBLE, deterministically: every workgroup owns a contiguous run of
// points and writes one partial row per statistic, a one-workgroup kernel adds the partials in index order and derives
// the constants the next sweep needs (no host round trip, no atomics).
//
// Notation (embedding.py of this package): a_i, b_i the two per-point projections of conv1; an edge (i, k) with
// j = nn[i][k] has u = a_i + b_j; S_i = sum_k b_j, Q_i = sum_k b_j^2; BN1: z = gamma1 (u - mu1) / sig1 + beta1;
// conv2 output y; BN2: v = gamma2 (ext - mu2) / sig2 + beta2 on the per-point extremum ext of y (max or min by the
// sign of gamma2: LeakyReLU o BN2 is monotone), out = LeakyReLU(v).
#include "samble_dev.h"

namespace samble {

constexpr int kGC = 64;    // channels
constexpr int kGK = 32;    // neighbours per point
constexpr int kGParts = 1024;

// layout of the per-layer constants block `cst` (floats) and statistics block `st` (doubles)
enum { kCstSc1 = 0, kCstSh1 = 64, kCstSc2 = 128, kCstSh2 = 192, kCstC0 = 256, kCstC1 = 320, kCstM1p = 384, kCstM2p = 448,
       kCstWords = 512 };
enum { kStMu1 = 0, kStSig1 = 64, kStMu2 = 128, kStSig2 = 192, kStWords = 256 };

// per-workgroup partial of NS statistics: thread (rl = tid >> 5, c2 = tid & 31) holds channels 2 c2, 2 c2 + 1
template <int NS>
__device__ __forceinline__ void chan_partial_store(const double (&acc)[NS][2], double* __restrict__ part) {
  __shared__ double red[8][NS][kGC];
  const int rl = threadIdx.x >> 5, c2 = threadIdx.x & 31;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    red[rl][s][2 * c2] = acc[s][0];
    red[rl][s][2 * c2 + 1] = acc[s][1];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < NS * kGC; e += 256) {
    const int s = e / kGC, c = e % kGC;
    double v = red[0][s][c];
#pragma unroll
    for (int r = 1; r < 8; ++r) v += red[r][s][c];
    part[((long)blockIdx.x * NS + s) * kGC + c] = v;
  }
}

// sums of the workgroup partials of two statistics, for the finalize kernels: 8 workgroups of 256 threads, workgroup w
// owns the channels 8 w .. 8 w + 7 (a single workgroup pulled the whole 1-2 MB of partials through one CU: 16-31 us per
// launch, eight launches per block step).  Thread (channel tid & 7, group tid >> 3): the group's partials p = g, g + 32, ...
// (loads in flight eight deep), then the 32 group sums in index order.  Returns the channel; the totals are valid in the
// threads tid < 8.  Fixed order: deterministic.
constexpr int kFinWgs = 8, kFinThreads = 256;
__device__ __forceinline__ int chan_totals2(const double* __restrict__ part, int nparts, double& t0, double& t1) {
  __shared__ double red[32][2][8];
  const int cl = threadIdx.x & 7, g = threadIdx.x >> 3, c = blockIdx.x * 8 + cl;
  double v0 = 0.0, v1 = 0.0;
#pragma unroll 8
  for (int p = g; p < nparts; p += 32) {
    v0 += part[((long)p * 2 + 0) * kGC + c];
    v1 += part[((long)p * 2 + 1) * kGC + c];
  }
  red[g][0][cl] = v0;
  red[g][1][cl] = v1;
  __syncthreads();
  t0 = red[0][0][cl];
  t1 = red[0][1][cl];
#pragma unroll
  for (int k = 1; k < 32; ++k) {
    t0 += red[k][0][cl];
    t1 += red[k][1][cl];
  }
  return c;
}

// nn.SyncBatchNorm (the reference trainer converts every BatchNorm, train_modelnet.py:245-246): the two totals of a
// finalize kernel and the edge count as ONE block of 2 * 64 + 1 doubles -- [t0 | t1 | E] -- that the caller all-reduces
// over its process group between a launcher's "sums" phase (statistics kernel + this fold) and its "apply" phase (the
// finalize kernel reading the block instead of the partials, then the elementwise kernels).  Gradients of gamma / beta
// stay this rank's sums (DistributedDataParallel averages parameter gradients itself): mode 1 / 2 write them here.
constexpr int kPoolWords = 2 * kGC + 1;
__global__ __launch_bounds__(kFinThreads) void edge_fold_totals_kernel(const double* __restrict__ part, int nparts, double E,
                                                                      double* __restrict__ pooled, int mode,
                                                                      const double* __restrict__ st,
                                                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
  double t0, t1;
  const int c = chan_totals2(part, nparts, t0, t1);
  if (threadIdx.x >= 8) return;
  pooled[c] = t0;
  pooled[kGC + c] = t1;
  if (c == 0) pooled[2 * kGC] = E;
  if (mode == 2) {          // BN2's backward: d gamma2 = sum dv yhat, d beta2 = sum dv
    dgamma[c] = (float)t1;
    dbeta[c] = (float)t0;
  } else if (mode == 1) {   // BN1's backward: d gamma1 = sum du zhat = (sum du u - mu1 sum du) / sig1, d beta1 = sum du
    dgamma[c] = (float)((t1 - st[kStMu1 + c] * t0) / st[kStSig1 + c]);
    dbeta[c] = (float)t0;
  }
}
// the totals a finalize kernel works from: the all-reduced block when there is one, the partials otherwise
__device__ __forceinline__ int chan_totals_or_pooled(const double* __restrict__ part, int nparts, const double* __restrict__ pooled,
                                                     double& t0, double& t1, double& E) {
  if (pooled) {
    const int c = blockIdx.x * 8 + (threadIdx.x & 7);
    t0 = pooled[c];
    t1 = pooled[kGC + c];
    E = pooled[2 * kGC];
    return c;
  }
  return chan_totals2(part, nparts, t0, t1);
}

// ---- forward 1: S, Q of every point (edge_gather_sums) and the BN1 edge sums  sum u = K a + S,  sum u^2 = K a^2 + 2 a S + Q
// (a, bp: rows of 64 floats at a stride of `rs` floats -- the two halves of one (points, 128) projection output)
__global__ __launch_bounds__(256) void edge_sums_stats_kernel(const float* __restrict__ a, const float* __restrict__ bp,
                                                              long rs, const int* __restrict__ nn, int N, long npoints,
                                                              float* __restrict__ S, float* __restrict__ Q,
                                                              double* __restrict__ part) {
  const int hw = threadIdx.x >> 5, c2 = threadIdx.x & 31;
  double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  // XCD x (workgroups x, x + 8, ...) takes the clouds x, x + 8, ...: the 32 rows a point gathers are its cloud's, and an
  // XCD's L2 then holds its own clouds' b rows (0.5 MB each) instead of a slice of every cloud of the batch
  const long nclouds = npoints / N;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const bool by_xcd = (gridDim.x & 7) == 0 && nclouds * N == npoints;
  const long mine = by_xcd ? ((nclouds - xcd + 7) / 8) * N : npoints;          // points of this XCD's clouds
  const long per = (mine + (by_xcd ? per_xcd : (int)gridDim.x) - 1) / (by_xcd ? per_xcd : (int)gridDim.x);
  const long u0 = (long)(by_xcd ? slot : (int)blockIdx.x) * per, u1 = min(u0 + per, mine);
  for (long u = u0 + hw; u < u1; u += 8) {
    const long p = by_xcd ? ((u / N) * 8 + xcd) * N + u % N : u;
    const long cloud = p / N;
    const int* ni = nn + p * kGK;
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll 8
    for (int k = 0; k < kGK; ++k) {
      const int j = ni[k];
      const float2 v = *reinterpret_cast<const float2*>(bp + (cloud * N + j) * rs + 2 * c2);
      s0 += v.x;
      s1 += v.y;
      q0 = fmaf(v.x, v.x, q0);
      q1 = fmaf(v.y, v.y, q1);
    }
    *reinterpret_cast<float2*>(S + p * kGC + 2 * c2) = make_float2(s0, s1);
    *reinterpret_cast<float2*>(Q + p * kGC + 2 * c2) = make_float2(q0, q1);
    const float2 av = *reinterpret_cast<const float2*>(a + p * rs + 2 * c2);
    const double a0 = av.x, a1 = av.y;
    acc[0][0] += kGK * a0 + (double)s0;
    acc[0][1] += kGK * a1 + (double)s1;
    acc[1][0] += kGK * a0 * a0 + 2.0 * a0 * (double)s0 + (double)q0;
    acc[1][1] += kGK * a1 * a1 + 2.0 * a1 * (double)s1 + (double)q1;
  }
  chan_partial_store<2>(acc, part);
}

// running statistics of nn.BatchNorm2d in training mode (biased batch variance -> unbiased running variance)
__device__ __forceinline__ void bn_running_update(float* rmean, float* rvar, int c, double mu, double var, double E,
                                                  float momentum) {
  if (!rmean) return;
  rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * mu);
  rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * (var * E / (E - 1.0)));
}

// ---- forward 2: BN1 constants from the partials: mu1, sig1; sc1 = gamma1 / sig1, sh1 = beta1 - mu1 sc1
__global__ __launch_bounds__(kFinThreads) void edge_bn1_finalize_kernel(const double* __restrict__ part, int nparts, double E,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float eps, float* __restrict__ cst, double* __restrict__ st,
                                                               float* rmean, float* rvar, float momentum, long long* nbt,
                                                               const double* __restrict__ pooled) {
  double t0, t1;
  const int c = chan_totals_or_pooled(part, nparts, pooled, t0, t1, E);
  if (threadIdx.x >= 8) return;
  const double mu = t0 / E;
  double var = t1 / E - mu * mu;
  var = var < 0.0 ? 0.0 : var;
  const double sig = sqrt(var + (double)eps);
  const double sc = (double)gamma[c] / sig;
  st[kStMu1 + c] = mu;
  st[kStSig1 + c] = sig;
  cst[kCstSc1 + c] = (float)sc;
  cst[kCstSh1 + c] = (float)((double)beta[c] - mu * sc);
  bn_running_update(rmean, rvar, c, mu, var, E, momentum);
  if (c == 0 && nbt) *nbt += 1;   // nn.BatchNorm2d.num_batches_tracked (int64), without a launch of its own
}

// ---- forward 2b: the BN1-folded projections the MLP sweeps read: a' = a sc1 + sh1, b' = b sc1 (two roundings each,
// like the elementwise expressions they replace)
__global__ __launch_bounds__(256) void edge_fold_kernel(const float* __restrict__ a, const float* __restrict__ bp, long rs,
                                                        const float* __restrict__ cst, long n4, float* __restrict__ ap,
                                                        float* __restrict__ bpo) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;  // one float4 of a row (16 per row)
  if (e >= n4) return;
  const int c = 4 * (int)(e & 15);
  const f32x4 av = *reinterpret_cast<const f32x4*>(a + (e >> 4) * rs + c), bv = *reinterpret_cast<const f32x4*>(bp + (e >> 4) * rs + c);
  f32x4 oa, ob;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    oa[u] = __fadd_rn(__fmul_rn(av[u], cst[kCstSc1 + c + u]), cst[kCstSh1 + c + u]);
    ob[u] = __fmul_rn(bv[u], cst[kCstSc1 + c + u]);
  }
  reinterpret_cast<f32x4*>(ap)[e] = oa;
  reinterpret_cast<f32x4*>(bpo)[e] = ob;
}

// ---- forward 3: BN2 constants from edge_mlp_fwd's per-wave sums of y and y^2: mu2, sig2; sc2 = gamma2 / sig2, sh2 = beta2 - mu2 sc2
__global__ __launch_bounds__(kFinThreads) void edge_bn2_finalize_kernel(const double* __restrict__ part, int nparts, double E,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float eps, float* __restrict__ cst, double* __restrict__ st,
                                                               float* rmean, float* rvar, float momentum, long long* nbt,
                                                               const double* __restrict__ pooled) {
  double t0, t1;
  const int c = chan_totals_or_pooled(part, nparts, pooled, t0, t1, E);
  if (threadIdx.x >= 8) return;
  const double mu = t0 / E;
  double var = t1 / E - mu * mu;
  var = var < 0.0 ? 0.0 : var;
  const double sig = sqrt(var + (double)eps);
  const double sc = (double)gamma[c] / sig;
  st[kStMu2 + c] = mu;
  st[kStSig2 + c] = sig;
  cst[kCstSc2 + c] = (float)sc;
  cst[kCstSh2 + c] = (float)((double)beta[c] - mu * sc);
  bn_running_update(rmean, rvar, c, mu, var, E, momentum);
  if (c == 0 && nbt) *nbt += 1;   // nn.BatchNorm2d.num_batches_tracked (int64), without a launch of its own
}

// ---- forward 4: ext = the extremum the sign of gamma2 selects, its edge, out = LeakyReLU(sc2 ext + sh2) written
// CHANNEL-MAJOR (B, 64, N) through a 64 x 64 LDS tile (the layers downstream hold features that way)
__global__ __launch_bounds__(256) void edge_out_kernel(const float* __restrict__ ymax, const float* __restrict__ ymin,
                                                       const unsigned char* __restrict__ kmax,
                                                       const unsigned char* __restrict__ kmin,
                                                       const float* __restrict__ gamma2, const float* __restrict__ cst, int N,
                                                       float* __restrict__ ext, unsigned char* __restrict__ kext,
                                                       float* __restrict__ out) {
  __shared__ float tile[64][65];
  const int b = blockIdx.y, n0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63, n = n0 + r;
    float v = 0.f;
    if (n < N) {
      const long at = ((long)b * N + n) * kGC + c;
      const bool up = gamma2[c] >= 0.f;
      const float x = up ? ymax[at] : ymin[at];
      ext[at] = x;
      kext[at] = up ? kmax[at] : kmin[at];
      v = fmaf(x, cst[kCstSc2 + c], cst[kCstSh2 + c]);
      v = fmaxf(v, 0.2f * v);
    }
    tile[r][c] = v;
  }
  __syncthreads();
  for (int e = tid; e < 64 * 64; e += 256) {
    const int c = e >> 6, r = e & 63, n = n0 + r;
    if (n < N) out[((long)b * kGC + c) * N + n] = tile[r][c];
  }
}

// ---- backward 1: g (B, 64, N) channel-major -> dv = g LeakyReLU'(v); sc2 dv as (points, 64) rows, the sums of dv and dv yhat
__global__ __launch_bounds__(256) void edge_bwd_pre_kernel(const float* __restrict__ g, const float* __restrict__ ext,
                                                           const float* __restrict__ cst, const double* __restrict__ st,
                                                           int N, int tiles_per_cloud, int ntiles, float* __restrict__ dv,
                                                           double* __restrict__ part) {
  __shared__ float tile[64][65];
  const int tid = threadIdx.x, rl = tid >> 5, c2 = tid & 31;
  double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  const double mu0 = st[kStMu2 + 2 * c2], mu1 = st[kStMu2 + 2 * c2 + 1];
  const double is0 = 1.0 / st[kStSig2 + 2 * c2], is1 = 1.0 / st[kStSig2 + 2 * c2 + 1];
  const int per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int t0 = blockIdx.x * per, t1 = min(t0 + per, ntiles);
  for (int t = t0; t < t1; ++t) {
    const int b = t / tiles_per_cloud, n0 = (t % tiles_per_cloud) * 64;
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += 256) {
      const int c = e >> 6, r = e & 63, n = n0 + r;
      tile[r][c] = n < N ? g[((long)b * kGC + c) * N + n] : 0.f;
    }
    __syncthreads();
    for (int r = rl; r < 64; r += 8) {
      const int n = n0 + r;
      if (n >= N) break;
      const long at = ((long)b * N + n) * kGC + 2 * c2;
      const float2 x = *reinterpret_cast<const float2*>(ext + at);
      const float v0 = fmaf(x.x, cst[kCstSc2 + 2 * c2], cst[kCstSh2 + 2 * c2]);
      const float v1 = fmaf(x.y, cst[kCstSc2 + 2 * c2 + 1], cst[kCstSh2 + 2 * c2 + 1]);
      const float d0 = tile[r][2 * c2] * (v0 > 0.f ? 1.f : 0.2f), d1 = tile[r][2 * c2 + 1] * (v1 > 0.f ? 1.f : 0.2f);
      // what the MLP sweep consumes is sc2 dv (the gradient that arrives at the extremal edge, BN2's scale applied)
      *reinterpret_cast<float2*>(dv + at) = make_float2(d0 * cst[kCstSc2 + 2 * c2], d1 * cst[kCstSc2 + 2 * c2 + 1]);
      acc[0][0] += (double)d0;
      acc[0][1] += (double)d1;
      acc[1][0] += (double)d0 * (((double)x.x - mu0) * is0);
      acc[1][1] += (double)d1 * (((double)x.y - mu1) * is1);
    }
  }
  chan_partial_store<2>(acc, part);
}

// ---- backward 2: BN2's dense correction dy += c0 + c1 y per edge (c1 = -sc2 m2 / sig2, c0 = -sc2 m1 - c1 mu2 with
// m1 = mean dv, m2 = mean dv yhat over the EDGES), d gamma2 = sum dv yhat, d beta2 = sum dv
__global__ __launch_bounds__(kFinThreads) void edge_bwd2_finalize_kernel(const double* __restrict__ part, int nparts, double E,
                                                                const float* __restrict__ gamma, float* __restrict__ cst,
                                                                const double* __restrict__ st, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, const double* __restrict__ pooled) {
  double sdv, sdvy;
  const int c = chan_totals_or_pooled(part, nparts, pooled, sdv, sdvy, E);
  if (threadIdx.x >= 8) return;
  const double m1 = sdv / E, m2 = sdvy / E;
  const double sig = st[kStSig2 + c], mu = st[kStMu2 + c], sc = (double)gamma[c] / sig;
  const double c1 = -sc * m2 / sig;
  cst[kCstC1 + c] = (float)c1;
  cst[kCstC0 + c] = (float)(-sc * m1 - c1 * mu);
  if (!pooled) {   // (pooled: this rank's sums went out with the fold)
    dgamma[c] = (float)sdvy;
    dbeta[c] = (float)sdv;
  }
}

// ---- backward 3: sums over the points of du_i (= sum_k du_ik) and of a_i du_i + b_i D_i (D_i = sum over the incoming
// edges of du): the edge sums of du and du u that BN1's backward needs
__global__ __launch_bounds__(256) void edge_bwd_stats_kernel(const float* __restrict__ a, const float* __restrict__ bp, long rs,
                                                             const float* __restrict__ dusum, const float* __restrict__ D,
                                                             long npoints, double* __restrict__ part) {
  const int rl = threadIdx.x >> 5, c2 = threadIdx.x & 31;
  double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  const long per = (npoints + gridDim.x - 1) / gridDim.x;
  const long p0 = (long)blockIdx.x * per, p1 = min(p0 + per, npoints);
  for (long p = p0 + rl; p < p1; p += 8) {
    const long at = p * kGC + 2 * c2, ar = p * rs + 2 * c2;
    const float2 av = *reinterpret_cast<const float2*>(a + ar), bv = *reinterpret_cast<const float2*>(bp + ar);
    const float2 uv = *reinterpret_cast<const float2*>(dusum + at), dvv = *reinterpret_cast<const float2*>(D + at);
    acc[0][0] += (double)uv.x;
    acc[0][1] += (double)uv.y;
    acc[1][0] += (double)av.x * (double)uv.x + (double)bv.x * (double)dvv.x;
    acc[1][1] += (double)av.y * (double)uv.y + (double)bv.y * (double)dvv.y;
  }
  chan_partial_store<2>(acc, part);
}

// ---- backward 4: m1' = mean du, m2' = mean du zhat over the edges; d gamma1 = sum du zhat, d beta1 = sum du
__global__ __launch_bounds__(kFinThreads) void edge_bwd1_finalize_kernel(const double* __restrict__ part, int nparts, double E,
                                                                float* __restrict__ cst, const double* __restrict__ st,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                const double* __restrict__ pooled) {
  double sdu, raw;
  const int c = chan_totals_or_pooled(part, nparts, pooled, sdu, raw, E);
  if (threadIdx.x >= 8) return;
  const double sduz = (raw - st[kStMu1 + c] * sdu) / st[kStSig1 + c];
  cst[kCstM1p + c] = (float)(sdu / E);
  cst[kCstM2p + c] = (float)(sduz / E);
  if (!pooled) {
    dgamma[c] = (float)sduz;
    dbeta[c] = (float)sdu;
  }
}

// ---- backward 5: the per-point gradients of the two projections
//   da = sc1 (du_i - K m1' - m2' Zs),  Zs = (K a + S - K mu1) / sig1       (the K outgoing edges of point i)
//   db = sc1 (D_i - deg m1' - m2' Zr), Zr = (R + deg (b - mu1)) / sig1     (its deg incoming edges; R = sum of their a)
__global__ __launch_bounds__(256) void edge_bwd_final_kernel(const float* __restrict__ a, const float* __restrict__ bp,
                                                             const float* __restrict__ S, const float* __restrict__ R,
                                                             const float* __restrict__ dusum, const float* __restrict__ D,
                                                             const int* __restrict__ indeg, const float* __restrict__ cst,
                                                             const double* __restrict__ st, long npoints, long rs,
                                                             float* __restrict__ da, float* __restrict__ db, long drs) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;  // one float2 of a row
  if (e >= npoints * 32) return;
  const long p = e >> 5;
  const int c = 2 * (int)(e & 31);
  const long at = p * kGC + c, ar = p * rs + c;
  const float deg = (float)indeg[p];
  const float2 av = *reinterpret_cast<const float2*>(a + ar), bv = *reinterpret_cast<const float2*>(bp + ar);
  const float2 sv = *reinterpret_cast<const float2*>(S + at), rv = *reinterpret_cast<const float2*>(R + at);
  const float2 uv = *reinterpret_cast<const float2*>(dusum + at), dvv = *reinterpret_cast<const float2*>(D + at);
  float oa[2], ob[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const float mu = (float)st[kStMu1 + c + u], sig = (float)st[kStSig1 + c + u];
    const float sc = cst[kCstSc1 + c + u], m1 = cst[kCstM1p + c + u], m2 = cst[kCstM2p + c + u];
    const float aa = u ? av.y : av.x, bb = u ? bv.y : bv.x, ss = u ? sv.y : sv.x, rr = u ? rv.y : rv.x;
    const float du = u ? uv.y : uv.x, dd = u ? dvv.y : dvv.x;
    const float zs = (kGK * aa + ss - kGK * mu) / sig;
    const float zr = (rr + deg * (bb - mu)) / sig;
    oa[u] = sc * (du - kGK * m1 - m2 * zs);
    ob[u] = sc * (dd - deg * m1 - m2 * zr);
  }
  *reinterpret_cast<float2*>(da + p * drs + c) = make_float2(oa[0], oa[1]);
  *reinterpret_cast<float2*>(db + p * drs + c) = make_float2(ob[0], ob[1]);
}

// out[e] = sum over the nparts blocks (each n floats): 64 x 64 dW2 from the per-wave partials.  Workgroup = 16 elements
// x 16 groups of partials (p = g, g + 16, ...: eight loads in flight), the 16 group sums added in index order:
// deterministic.  (One thread per element walking all 2048 partials took 100 us: a dependent chain of strided loads.)
__global__ __launch_bounds__(256) void edge_sum_parts_kernel(const float* __restrict__ part, int nparts, int n,
                                                             float* __restrict__ out) {
  __shared__ float red[16][17];
  const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  float s = 0.f;
  if (e < n) {
#pragma unroll 8
    for (int p = g; p < nparts; p += 16) s += part[(long)p * n + e];
  }
  red[g][el] = s;
  __syncthreads();
  if (g == 0 && e < n) {
    float t = red[0][el];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += red[k][el];
    out[e] = t;
  }
}

}  // namespace samble

using namespace samble;

extern "C" size_t samble_edge_glue_part_bytes(void) { return (size_t)kGParts * 2 * kGC * sizeof(double); }
extern "C" size_t samble_edge_glue_cst_bytes(void) { return (size_t)kCstWords * sizeof(float); }
extern "C" size_t samble_edge_glue_st_bytes(void) { return (size_t)kStWords * sizeof(double); }

extern "C" size_t samble_edge_glue_pool_bytes(void) { return (size_t)kPoolWords * sizeof(double); }

// phase 0: one rank, everything.  phase 1 ("sums"): up to the fold of this rank's totals into `pooled`.  phase 2 ("apply"):
// from the finalize kernel on, the totals read from `pooled` (all-reduced by the caller in between).
#define FOLD(partials, nparts, E, mode, st, dg, db) \
  hipLaunchKernelGGL(edge_fold_totals_kernel, dim3(kFinWgs), dim3(kFinThreads), 0, s, partials, nparts, E, pooled, mode, st, dg, db)

// forward, before the MLP sweep: S, Q, BN1 constants (cst: sc1, sh1; st: mu1, sig1), running statistics (optional)
extern "C" int samble_launch_edge_pre(const float* a, const float* b, long rs, const int* nn, int B, int N, const float* gamma1,
                                      const float* beta1, float eps, float* rmean, float* rvar, float momentum, long long* nbt, float* S,
                                      float* Q, float* ap, float* bp, float* cst, double* st, double* part, int phase, double* pooled,
                                      hipStream_t s) {
  const long np = (long)B * N;
  Timed timed(kT_edge_sums, s);
  if (phase != 2) hipLaunchKernelGGL(edge_sums_stats_kernel, dim3(kGParts), dim3(256), 0, s, a, b, rs, nn, N, np, S, Q, part);
  if (phase == 1) {
    FOLD(part, kGParts, (double)np * kGK, 0, (const double*)nullptr, (float*)nullptr, (float*)nullptr);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(edge_bn1_finalize_kernel, dim3(kFinWgs), dim3(kFinThreads), 0, s, part, kGParts, (double)np * kGK, gamma1, beta1, eps,
                     cst, st, rmean, rvar, momentum, nbt, phase == 2 ? (const double*)pooled : (const double*)nullptr);
  hipLaunchKernelGGL(edge_fold_kernel, dim3((unsigned)((np * 16 + 255) / 256)), dim3(256), 0, s, a, b, rs, cst, np * 16, ap, bp);
  return (int)hipGetLastError();
}

// forward, after the MLP sweep (its per-wave sums in `mlp_part`, nwaves of them): BN2 constants, the selected extremum and
// its edge, the activated output channel-major
extern "C" int samble_launch_edge_post(const float* ymax, const float* ymin, const unsigned char* kmax,
                                       const unsigned char* kmin, const double* mlp_part, int nwaves, int B, int N,
                                       const float* gamma2, const float* beta2, float eps, float* rmean, float* rvar,
                                       float momentum, long long* nbt, float* cst, double* st, float* ext, unsigned char* kext,
                                       float* out, int phase, double* pooled, hipStream_t s) {
  const long np = (long)B * N;
  if (phase == 1) {
    FOLD(mlp_part, nwaves, (double)np * kGK, 0, (const double*)nullptr, (float*)nullptr, (float*)nullptr);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(edge_bn2_finalize_kernel, dim3(kFinWgs), dim3(kFinThreads), 0, s, mlp_part, nwaves, (double)np * kGK, gamma2, beta2,
                     eps, cst, st, rmean, rvar, momentum, nbt, phase == 2 ? (const double*)pooled : (const double*)nullptr);
  hipLaunchKernelGGL(edge_out_kernel, dim3((N + 63) / 64, B), dim3(256), 0, s, ymax, ymin, kmax, kmin, gamma2, cst, N, ext,
                     kext, out);
  return (int)hipGetLastError();
}

// backward, before the MLP sweep: dv rows, c0 / c1 of BN2's dense correction, d gamma2, d beta2
extern "C" int samble_launch_edge_bwd_pre(const float* g, const float* ext, int B, int N, const float* gamma2, float* cst,
                                          const double* st, float* dv, float* dgamma2, float* dbeta2, double* part,
                                          int phase, double* pooled, hipStream_t s) {
  const long np = (long)B * N;
  const int tpc = (N + 63) / 64, ntiles = tpc * B;
  const int grid = ntiles < kGParts ? ntiles : kGParts;
  if (phase != 2) hipLaunchKernelGGL(edge_bwd_pre_kernel, dim3(grid), dim3(256), 0, s, g, ext, cst, st, N, tpc, ntiles, dv, part);
  if (phase == 1) {
    FOLD(part, grid, (double)np * kGK, 2, st, dgamma2, dbeta2);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(edge_bwd2_finalize_kernel, dim3(kFinWgs), dim3(kFinThreads), 0, s, part, grid, (double)np * kGK, gamma2, cst, st,
                     dgamma2, dbeta2, phase == 2 ? (const double*)pooled : (const double*)nullptr);
  return (int)hipGetLastError();
}

// backward, after the MLP sweep and the reverse-neighbour sums D (of du) and R (of a): d gamma1, d beta1, da, db; dW2
extern "C" int samble_launch_edge_bwd_post(const float* a, const float* b, long rs, const float* S, const float* R,
                                           const float* dusum, const float* D, const int* indeg, int B, int N, float* cst,
                                           const double* st, const float* dw2part, int nwaves, float* da, float* db,
                                           long drs, float* dgamma1, float* dbeta1, float* dW2, double* part, int phase,
                                           double* pooled, hipStream_t s) {
  const long np = (long)B * N;
  if (phase != 2) hipLaunchKernelGGL(edge_bwd_stats_kernel, dim3(kGParts), dim3(256), 0, s, a, b, rs, dusum, D, np, part);
  if (phase == 1) {
    FOLD(part, kGParts, (double)np * kGK, 1, st, dgamma1, dbeta1);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(edge_bwd1_finalize_kernel, dim3(kFinWgs), dim3(kFinThreads), 0, s, part, kGParts, (double)np * kGK, cst, st, dgamma1,
                     dbeta1, phase == 2 ? (const double*)pooled : (const double*)nullptr);
  hipLaunchKernelGGL(edge_bwd_final_kernel, dim3((unsigned)((np * 32 + 255) / 256)), dim3(256), 0, s, a, b, S, R, dusum, D,
                     indeg, cst, st, np, rs, da, db, drs);
  hipLaunchKernelGGL(edge_sum_parts_kernel, dim3(kGC * kGC / 16), dim3(256), 0, s, dw2part, nwaves, kGC * kGC, dW2);
  return (int)hipGetLastError();
}
#undef FOLD
