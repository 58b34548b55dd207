// Neighbor-to-point attention (reference models/attention.py:165-250, scalar_dot / asm dot).
//
// The reference gathers K neighbours per point into a (B,C,N,K) tensor (1 GB at B=32, N=2048) and
// runs two 1x1 Conv2d over it (2 x 68.7 GFLOP).  A 1x1 conv is linear, so
//     Wk (x_j - x_i) = (Wk x)_j - (Wk x)_i
// and the projection is done ONCE per point (proj_fwd_kernel, 0.2 GFLOP/cloud); this kernel then
// gathers the K projected neighbour rows of each point, runs the per-head 1 x K softmax attention
// online, and writes the (B,C,N) result.  Half-wave = one point: lane c owns channels 4c..4c+3,
// head = c / 8 (H = 4 heads of D = 32 -> 8 lanes per head, logits reduced with 3 shuffles).
// Bound: L2 / Infinity-Cache gather of N*K rows of 2 x 512 B (workgroups of a cloud share an XCD).
#include "samble_dev.h"

namespace samble {

// sum over the lanes of one head: hl = lanes per head = 32 / heads (8 for the 4 heads of N2P, 16 for two heads, 32 for
// the single head of DownSampleLocal); wave-uniform
// Round 6: the butterfly on DPP lane moves (the vector ALU's own cross-lane path, fused into the add) -- `__shfl_xor` is a
// `ds_bpermute_b32`: an LDS-crossbar instruction plus its address arithmetic and a wait, three per sum, and the gather kernels
// are half vector issue.  quad_perm swaps give lane ^ 1 and lane ^ 2; after them a quad's four lanes hold the same value, so
// the octet's other quad is reached by row_half_mirror (lane 7 - i) and the row's other octet by row_mirror (15 - i): the
// same partial sums in the same order as the xor butterfly, bit for bit.
template <int CTRL>
__device__ __forceinline__ float head_dpp(float x) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float head_sum(float v, int hl) {
  v += head_dpp<0xB1>(v);                   // quad_perm [1,0,3,2]: lane ^ 1
  v += head_dpp<0x4E>(v);                   // quad_perm [2,3,0,1]: lane ^ 2
  v += head_dpp<0x141>(v);                  // row_half_mirror: the octet's other quad
  if (hl > 8) v += head_dpp<0x140>(v);      // row_mirror: 2 heads of 16 lanes each
  if (hl > 16) v += __shfl_xor(v, 16, 64);  // 1 head: all 32 lanes of the half-wave (DownSampleLocal)
  return v;
}

__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
}

// grid (ceil(N/32), B), 256 threads = 8 half-waves x 4 points each
__global__ __launch_bounds__(256) void n2p_attn_fwd_kernel(const float* __restrict__ qkv, long bs, long rs,
                                                           const int* __restrict__ nn, int N, int KN, int diff,
                                                           float scale, float* __restrict__ out, int heads,
                                                           float* __restrict__ att, int B, const float* __restrict__ res) {
  __shared__ float tile[128 * 33];
  __shared__ float lgs[8][64];  // att output: the logits of a half-wave's point (this lane's head), K <= 64
  const int hl = 32 / heads;
  const int tid = threadIdx.x, hw = tid >> 5, c = tid & 31;
  // persistent: XCD x (workgroups x, x + 8, ...) walks the 32-point chunks of the clouds x, x + 8, ... cloud after
  // cloud, so that its L2 holds the [Q|K|V] rows of as few clouds as the grid size allows (placement only)
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3, cpc = (N + 31) / 32;
  for (int v = blockIdx.x >> 3;; v += per_xcd) {
  const int b = (v / cpc) * 8 + xcd, chunk = v % cpc;
  if (b >= B) break;  // uniform over the workgroup
  const float* base = qkv + (long)b * bs;
  for (int pp = 0; pp < 4; ++pp) {
    const int lp = hw * 4 + pp;
    const int i = chunk * 32 + lp;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (i < N) {  // uniform per half-wave
      const float* row = base + (long)i * rs + 4 * c;
      const f32x4 q = *reinterpret_cast<const f32x4*>(row);
      f32x4 kc = {0.f, 0.f, 0.f, 0.f}, vc = {0.f, 0.f, 0.f, 0.f};
      float qkc = 0.f;
      if (diff) {
        kc = *reinterpret_cast<const f32x4*>(row + 128);
        vc = *reinterpret_cast<const f32x4*>(row + 256);
        qkc = head_sum(dot4(q, kc), hl);
      }
      const int* ni = nn + ((long)b * N + i) * KN;
      float m = kNegInf, l = 0.f;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < KN; k0 += 8) {  // eight neighbours = sixteen row loads in flight (four: 7 % slower)
        f32x4 kv[8], vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = (k0 + u < KN) ? ni[k0 + u] : ni[0];
          const float* jr = base + (long)j * rs + 4 * c;
          kv[u] = *reinterpret_cast<const f32x4*>(jr + 128);
          vv[u] = *reinterpret_cast<const f32x4*>(jr + 256);
        }
        // online softmax with ONE rescale per chunk of eight neighbours (round 6): the chunk's logits first, their maximum
        // joins the running one, the running sums are rescaled once, then eight plain accumulations -- one exponential and
        // five multiplies less per neighbour than a rescale at every neighbour (the kernel is half vector issue)
        float lg8[8];
        float mn = m;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float s = head_sum(dot4(q, kv[u]), hl);
          lg8[u] = (k0 + u < KN) ? (s - qkc) * scale : kNegInf;
          if (att && heads == 1 && k0 + u < KN) lgs[hw][k0 + u] = lg8[u];
          mn = fmaxf(mn, lg8[u]);
        }
        const float al = __expf(m - mn);   // (first chunk: exp(-inf) = 0 on zero sums)
        l *= al;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] *= al;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float p = __expf(lg8[u] - mn);   // (masked slots: exp(-inf) = 0)
          l += p;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = fmaf(p, vv[u][e], acc[e]);
        }
        m = mn;
      }
      const float inv = 1.f / l;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = acc[e] * inv - vc[e];
      if (att && heads == 1) {  // (B,1,N,K) probabilities (reference attention_map of DownSampleLocal)
        for (int k = c; k < KN; k += 32) att[((long)b * N + i) * KN + k] = __expf(lgs[hw][k] - m) * inv;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[(4 * c + e) * 33 + lp] = o[e];
  }
  __syncthreads();
  float* ob = out + (long)b * 128 * N;
  const float* rb = res ? res + (long)b * 128 * N : nullptr;   // the layer's residual x + attention(x), added on the way out
  for (int e = tid; e < 128 * 32; e += 256) {
    const int d = e >> 5, p = e & 31;
    if (chunk * 32 + p < N) {
      const long at = (long)d * N + chunk * 32 + p;
      ob[at] = rb ? rb[at] + tile[d * 33 + p] : tile[d * 33 + p];
    }
  }
  __syncthreads();  // the tile is rewritten by the next chunk
  }
}

}  // namespace samble

using namespace samble;

#ifndef SAMBLE_INV_BLOCKS
#define SAMBLE_INV_BLOCKS 8  // target-row blocks per cloud of the LDS marking (A/B: 4 -> 8: -10 %, 16: as 4)
#endif
#ifndef SAMBLE_N2P_FWD_GRID
#define SAMBLE_N2P_FWD_GRID 2048  // workgroups of the persistent forward (a multiple of 8)
#endif

extern "C" int samble_launch_n2p_fwd(const float* qkv, long bs, long rs, const int* nn, int B, int N, int KN, int diff,
                                     float scale, float* out, int heads, float* att, const float* residual, hipStream_t s) {
  if ((heads != 1 && heads != 2 && heads != 4) || (att && (heads != 1 || KN > 64))) return -22;
  Timed timed(kT_n2p_fwd, s);
  hipLaunchKernelGGL(n2p_attn_fwd_kernel, dim3(SAMBLE_N2P_FWD_GRID), dim3(256), 0, s, qkv, bs, rs, nn, N, KN, diff, scale,
                     out, heads, att, B, residual);
  return (int)hipGetLastError();
}

// ================================================================================================
// Backward of the gather-attention.  With k_ij = Kp[j] - d*Kp[i], v_ij = Vp[j] - d*Vp[i] (d = diff),
// a_ij = softmax_j(scale q_i.k_ij) per head, out_i = sum_j a_ij v_ij and g_i = dL/dout_i:
//     da_ij = g_i . Vp[j],  delta_i = sum_j a_ij da_ij,  dl_ij = scale a_ij (da_ij - delta_i)
//     dQ[i]  = sum_j dl_ij k_ij
//     dKp[j] += dl_ij q_i      dKp[i] -= d (sum_j dl_ij) q_i
//     dVp[j] += a_ij g_i       dVp[i] -= d g_i
// Pass 1 (n2p_bwd_point): half-wave per point like the forward; writes dQ[i], the two "self" rows
//   and the per-pair coefficients a_ij, dl_ij (B,N,K,4 heads).
// Pass 2 (n2p_bwd_scatter): the scatter-add over neighbours as a GATHER with a fixed order: a
//   workgroup owns 64 target rows j (accumulators in LDS) and walks the cloud's rows i in order, 128
//   at a time; hits (nn[i][k] in the block) get their slot in an ordered list from a block-wide
//   prefix sum (thread order = (i,k) order), and are then applied one after the other.  No atomics,
//   run-to-run identical.
// ================================================================================================
namespace samble {

// (B,C,N) -> (B,N,C), C = 128
__global__ __launch_bounds__(256) void transpose_cn_kernel(const float* __restrict__ in, int N, float* __restrict__ out) {
  __shared__ float tile[128 * 33];
  int chunk, b;
  xcd_assign(chunk, b);
  const int tid = threadIdx.x, n0 = chunk * 32;
  const float* ib = in + (long)b * 128 * N;
  for (int e = tid; e < 128 * 32; e += 256) {
    const int c = e >> 5, p = e & 31;
    tile[c * 33 + p] = (n0 + p < N) ? ib[(long)c * N + n0 + p] : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < 32 * 32; e += 256) {
    const int p = e >> 5, c4 = (e & 31) * 4;
    if (n0 + p < N) {
      f32x4 v = {tile[(c4 + 0) * 33 + p], tile[(c4 + 1) * 33 + p], tile[(c4 + 2) * 33 + p], tile[(c4 + 3) * 33 + p]};
      *reinterpret_cast<f32x4*>(out + ((long)b * N + n0 + p) * 128 + c4) = v;
    }
  }
}

template <bool FULL>  // FULL: KN == 32, no slot of the 32 is masked (the shipped configs)
__global__ __launch_bounds__(256) void n2p_bwd_point_kernel(const float* __restrict__ qkv, long bs, long rs,
                                                            const int* __restrict__ nn,
                                                            const float* __restrict__ gt,  // (B,N,128)
                                                            int N, int KN, int diff, float scale,
                                                            float* __restrict__ dqkv, long dbs, long drs,
                                                            float* __restrict__ A, float* __restrict__ DL, int heads,
                                                            int B) {
  const int tid = threadIdx.x, hw = tid >> 5, c = tid & 31;
  const int hl = 32 / heads;
  // a half-wave per point; XCD x (workgroups x, x + 8, ...) walks the clouds x, x + 8, ... one after the other: its L2
  // then holds ONE cloud's [Q|K|V] rows (3 MB at N = 2048) while every point gathers 32 of them, instead of a slice of
  // every cloud in flight (the same placement as n2p_bwd_gather_kernel below)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
#pragma unroll 1
  for (long u = (long)slot * 8 + hw;; u += (long)per_xcd * 8) {
    const int b = (int)(u / N) * 8 + xcd;
    if (b >= B) break;  // uniform per half-wave
    const int i = (int)(u % N);
    const float* base = qkv + (long)b * bs;
    const float* row = base + (long)i * rs + 4 * c;
    const f32x4 q = *reinterpret_cast<const f32x4*>(row);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gt + ((long)b * N + i) * 128 + 4 * c);
    f32x4 kc = {0.f, 0.f, 0.f, 0.f};
    float qkc = 0.f;
    if (diff) {
      kc = *reinterpret_cast<const f32x4*>(row + 128);
      qkc = head_sum(dot4(q, kc), hl);
    }
    const int* ni = nn + ((long)b * N + i) * KN;
    // ONE sweep over the neighbours (their K and V rows are gathered once, as in the forward): the logits and da stay in
    // registers (one value per neighbour: this lane's head), and dQ comes from three running sums under the
    // online-softmax rescaling -- with p_k = exp(logit_k - m):  l = sum p_k,  D = sum p_k da_k,
    // S2 = sum p_k K_j,  S1 = sum p_k da_k K_j  =>  delta = D / l  and
    //     dQ = sum_k dl_k (K_j - k_c) = scale (S1 - delta S2) / l        (sum_k dl_k = 0: the k_c term drops out)
    // (round 2 re-read every K row in a second loop over the neighbours: three row gathers per pair, 160 registers.)
    float lg[32], da[32];
    float m = kNegInf, l = 0.f, D = 0.f;
    f32x4 S1 = {0.f, 0.f, 0.f, 0.f}, S2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k0 = 0; k0 < 32; k0 += 8) {
      f32x4 kv[8], vv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = (FULL || k0 + u < KN) ? ni[k0 + u] : ni[0];
        const float* jr = base + (long)j * rs + 4 * c;
        kv[u] = *reinterpret_cast<const f32x4*>(jr + 128);
        vv[u] = *reinterpret_cast<const f32x4*>(jr + 256);
      }
      // (round 6) one rescale of the running sums per chunk of eight neighbours, as in the forward
      float mn = m;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u;
        lg[k] = (FULL || k < KN) ? (head_sum(dot4(q, kv[u]), hl) - qkc) * scale : kNegInf;
        da[k] = head_sum(dot4(g, vv[u]), hl);
        mn = fmaxf(mn, lg[k]);
      }
      {
        const float al = __expf(m - mn);
        l *= al;
        D *= al;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          S2[e] *= al;
          S1[e] *= al;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u;
        const float p = __expf(lg[k] - mn), pd = p * da[k];   // (masked slots: exp(-inf) = 0)
        l += p;
        D += pd;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          S2[e] = fmaf(p, kv[u][e], S2[e]);
          S1[e] = fmaf(pd, kv[u][e], S1[e]);
        }
      }
      m = mn;
      // the running sums are DUE here (else their updates sink below the loop and all 32 K rows stay in registers)
      asm volatile("" : "+v"(S1), "+v"(S2), "+v"(l), "+v"(D), "+v"(m));
    }
    const float inv = 1.f / l;
    float delta = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      lg[k] = __expf(lg[k] - m) * inv;  // a_ij (masked slots: exp(-inf) = 0)
      delta = fmaf(lg[k], da[k], delta);
    }
    float sdl = 0.f;
    // (one lane-dependent base per array + constant offsets: the compiler otherwise keeps 64 precomputed 64-bit
    // addresses alive across the loop over the points)
    float* Ai = A + ((long)b * N + i) * KN * 4 + c / hl;
    float* Di = DL + ((long)b * N + i) * KN * 4 + c / hl;
    asm volatile("" : "+v"(Ai), "+v"(Di));
    const bool writer = (c & (hl - 1)) == 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      if (FULL || k < KN) {
        const float dl = lg[k] * (da[k] - delta) * scale;
        sdl += dl;
        if (writer) {
          Ai[k * 4] = lg[k];
          Di[k * 4] = dl;
        }
      }
    }
    f32x4 dq;
    {
      const float dsum = D * inv;  // = delta up to rounding; the same normalisation as S1 / l, S2 / l
#pragma unroll
      for (int e = 0; e < 4; ++e) dq[e] = scale * inv * (S1[e] - dsum * S2[e]);
    }
    (void)kc;
    float* drow = dqkv + (long)b * dbs + (long)i * drs + 4 * c;
    *reinterpret_cast<f32x4*>(drow) = dq;
    f32x4 sk = {0.f, 0.f, 0.f, 0.f}, sv = {0.f, 0.f, 0.f, 0.f};
    if (diff) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sk[e] = -sdl * q[e];
        sv[e] = -g[e];
      }
    }
    *reinterpret_cast<f32x4*>(drow + 128) = sk;
    *reinterpret_cast<f32x4*>(drow + 256) = sv;
  }
}

constexpr int kScatRows = 64;    // target rows per workgroup
constexpr int kScatChunk = 128;  // source rows scanned per chunk
constexpr int kScatHits = 1024;  // capacity of the per-chunk hit list (longer lists are consumed in rounds)

__global__ __launch_bounds__(256, 2) void n2p_bwd_scatter_kernel(const float* __restrict__ qkv, long bs, long rs,
                                                                 const int* __restrict__ nn,
                                                                 const float* __restrict__ gt,
                                                                 const float* __restrict__ A,
                                                                 const float* __restrict__ DL, int N, int KN,
                                                                 float* __restrict__ dqkv, long dbs, long drs,
                                                                 int heads) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* acc = smem;                                              // [2][64][128]
  int* hits = reinterpret_cast<int*>(smem + 2 * kScatRows * 128);  // packed (i_local << 16) | (k << 8) | j_local
  int* scan = hits + kScatHits;                                   // 256
  __shared__ int nhits_s;
  int blk, b;
  xcd_assign(blk, b);
  const int tid = threadIdx.x;
  const int c = tid & 127, which = tid >> 7, head = c / (128 / heads);
  const int j0 = blk * kScatRows;
  for (int e = tid; e < 2 * kScatRows * 128; e += 256) acc[e] = 0.f;
  const float* coef = which ? A : DL;
  const float* valbase = which ? gt + (long)b * N * 128 + c : qkv + (long)b * bs + c;
  const long valrs = which ? 128 : rs;
  const int per = (kScatChunk * KN + 255) / 256;  // idx entries per thread per chunk, in (i,k) order
  for (int i0 = 0; i0 < N; i0 += kScatChunk) {
    const int rows = min(kScatChunk, N - i0);
    const int total = rows * KN;
    int first = tid * per;
    // ---- ordered hit list: count, block prefix sum, fill
    int my = 0;
    for (int e = first; e < min(first + per, total); ++e) {
      const int j = nn[((long)b * N + i0) * KN + e];
      my += (j >= j0 && j < j0 + kScatRows) ? 1 : 0;
    }
    __syncthreads();  // previous chunk's list fully consumed
    scan[tid] = my;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const int a = (tid >= o) ? scan[tid - o] : 0;
      __syncthreads();
      scan[tid] += a;
      __syncthreads();
    }
    const int slot0 = scan[tid] - my;
    if (tid == 255) nhits_s = scan[255];
    __syncthreads();
    const int nh_total = nhits_s;
    // index-local clouds can put thousands of hits into one chunk: consume the list in rounds of kScatHits
    for (int hb = 0; hb < nh_total; hb += kScatHits) {
    if (hb > 0) __syncthreads();  // previous round's list fully consumed
    int slot = slot0;
    for (int e = first; e < min(first + per, total); ++e) {
      const int j = nn[((long)b * N + i0) * KN + e];
      if (j >= j0 && j < j0 + kScatRows) {
        if (slot >= hb && slot < hb + kScatHits) hits[slot - hb] = ((e / KN) << 16) | ((e % KN) << 8) | (j - j0);
        ++slot;
      }
    }
    __syncthreads();
    const int nh = min(nh_total - hb, kScatHits);
    // ---- apply in list order.  Loads of 4 hits are issued together; the accumulators live in
    // registers for the run of hits (the compiler must not reorder LDS read-modify-writes of
    // possibly equal rows, so each update is an explicit read / fma / write in order)
    for (int e0 = 0; e0 < nh; e0 += 4) {
      float cf[4], vl[4];
      int jl[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int hcode = hits[min(e0 + u, nh - 1)];
        const int il = hcode >> 16, k = (hcode >> 8) & 255;
        jl[u] = hcode & 255;
        const long i = i0 + il;
        cf[u] = (e0 + u < nh) ? coef[(((long)b * N + i) * KN + k) * 4 + head] : 0.f;
        vl[u] = valbase[i * valrs];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        volatile float* a = acc + (which * kScatRows + jl[u]) * 128 + c;
        const float cur = *a;
        *a = fmaf(cf[u], vl[u], cur);
      }
    }
    }
  }
  __syncthreads();
  // add to the self rows written by pass 1
  for (int e = tid; e < kScatRows * 128; e += 256) {
    const int jr = e >> 7, cc = e & 127;
    if (j0 + jr < N) {
      float* drow = dqkv + (long)b * dbs + (long)(j0 + jr) * drs;
      drow[128 + cc] += acc[jr * 128 + cc];
      drow[256 + cc] += acc[(kScatRows + jr) * 128 + cc];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Pass 2 as a segment gather over INVERSE neighbour lists.  `order` lists the edge ids e = i*K + k
// (per cloud-global point index) grouped by target j = nn[e], inside a group in ascending e (a stable
// sort of the neighbour table: built once per kNN by the host), `offs` the group boundaries.  A
// half-wave owns one target and sums its incoming edges in list order: no scan of the whole table,
// no atomics, run-to-run identical, and the work is the forward's gather volume.
//   dK[j] += sum_e DL[e][head] q[i(e)],   dV[j] += sum_e A[e][head] g[i(e)]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void n2p_bwd_gather_kernel(const float* __restrict__ qkv, long bs, long rs,
                                                             const float* __restrict__ gt, const float* __restrict__ A,
                                                             const float* __restrict__ DL,
                                                             const int* __restrict__ order,
                                                             const int* __restrict__ offs, int N, int KN, long ntargets,
                                                             float* __restrict__ dqkv, long dbs, long drs, int heads) {
  const int hw = threadIdx.x >> 5, c = threadIdx.x & 31;  // lane = channels 4c .. 4c+3
  const int head = c / (32 / heads);
#ifndef SAMBLE_N2P_GATHER_BATCH
#define SAMBLE_N2P_GATHER_BATCH 4
#endif
  constexpr int GB = SAMBLE_N2P_GATHER_BATCH;
  // Workgroups are dealt round-robin over the 8 XCDs (a private 4 MB L2 each): XCD x walks the clouds x, x + 8, ... one
  // after the other, so that what its L2 holds at any time is ONE cloud's q / g rows and coefficients (about 4 MB at
  // N = 2048) instead of a slice of every cloud in flight.  Placement only: every target is visited exactly once.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const long nclouds = ntargets / N;
  for (long u = (long)slot * 8 + hw;; u += (long)per_xcd * 8) {
    const long cloud = (u / N) * 8 + xcd;
    if (cloud >= nclouds) break;
    const int j = (int)(u % N);
    const long t = cloud * N + j;
    f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
    const int e0 = offs[t], e1 = offs[t + 1];
    int s = e0;
    for (; s + GB <= e1; s += GB) {  // GB edges' rows and coefficients in flight; the sums in list order as before
      long e[GB];
      float dl[GB], aa[GB];
      f32x4 qv[GB], gv[GB];
#pragma unroll
      for (int w = 0; w < GB; ++w) e[w] = order[s + w];   // global edge id: (cloud*N + i)*KN + k
#pragma unroll
      for (int w = 0; w < GB; ++w) {
        const long pi = e[w] / KN;          // cloud*N + i
        dl[w] = DL[e[w] * 4 + head];
        aa[w] = A[e[w] * 4 + head];
        qv[w] = *reinterpret_cast<const f32x4*>(qkv + cloud * bs + (pi - cloud * N) * rs + 4 * c);
        gv[w] = *reinterpret_cast<const f32x4*>(gt + pi * 128 + 4 * c);
      }
#pragma unroll
      for (int w = 0; w < GB; ++w)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          ak[u] = fmaf(dl[w], qv[w][u], ak[u]);
          av[u] = fmaf(aa[w], gv[w][u], av[u]);
        }
    }
    for (; s < e1; ++s) {
      const long e = order[s];
      const long pi = e / KN;
      const long i = pi - cloud * N;
      const float dl = DL[e * 4 + head], aa = A[e * 4 + head];
      const f32x4 qv = *reinterpret_cast<const f32x4*>(qkv + cloud * bs + i * rs + 4 * c);
      const f32x4 gv = *reinterpret_cast<const f32x4*>(gt + pi * 128 + 4 * c);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ak[u] = fmaf(dl, qv[u], ak[u]);
        av[u] = fmaf(aa, gv[u], av[u]);
      }
    }
    float* drow = dqkv + cloud * dbs + (long)j * drs + 4 * c;
    f32x4 k0 = *reinterpret_cast<f32x4*>(drow + 128), v0 = *reinterpret_cast<f32x4*>(drow + 256);
    k0 += ak;
    v0 += av;
    *reinterpret_cast<f32x4*>(drow + 128) = k0;  // the self rows of pass 1 + the gathered share
    *reinterpret_cast<f32x4*>(drow + 256) = v0;
  }
}

// out[t][0:C] = sum over the incoming edges e of target t of src[per_edge ? e : e / K][0:C]  (C = 64),
// in list order: the deterministic replacement of index_add_ for EdgeConv's backward
__global__ __launch_bounds__(256) void seg_sum_rows64_kernel(const float* __restrict__ src, long src_rs, const int* __restrict__ order,
                                                             const int* __restrict__ offs, int KN, int per_edge,
                                                             long ntargets, float* __restrict__ out) {
  const int hw = threadIdx.x >> 5, c = threadIdx.x & 31;  // lane = channels 2c, 2c+1
  for (long t = (long)blockIdx.x * 8 + hw; t < ntargets; t += (long)gridDim.x * 8) {
    float s0 = 0.f, s1 = 0.f;
    const int e0 = offs[t], e1 = offs[t + 1];
    int s = e0;
    for (; s + 4 <= e1; s += 4) {  // four rows in flight
      float2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long e = order[s + u];
        v[u] = *reinterpret_cast<const float2*>(src + (per_edge ? e : e / KN) * src_rs + 2 * c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s0 += v[u].x;
        s1 += v[u].y;
      }
    }
    for (; s < e1; ++s) {
      const long e = order[s];
      const float2 v = *reinterpret_cast<const float2*>(src + (per_edge ? e : e / KN) * src_rs + 2 * c);
      s0 += v.x;
      s1 += v.y;
    }
    *reinterpret_cast<float2*>(out + t * 64 + 2 * c) = make_float2(s0, s1);
  }
}

// both sums of EdgeConv's backward in ONE pass over the lists: outE[t] = sum of srcE[e] (per-edge rows), outP[t] = sum of
// srcP[e / K] (the sources' per-point rows), each in list order -- bit for bit the two single sums, which walked the same
// lists twice (2 x ~120 us per layer at B = 32, N = 2048)
__global__ __launch_bounds__(256) void seg_sum_rows64_pair_kernel(const float* __restrict__ srcE, long e_rs,
                                                                  const float* __restrict__ srcP, long p_rs,
                                                                  const int* __restrict__ order, const int* __restrict__ offs,
                                                                  int KN, long ntargets, float* __restrict__ outE,
                                                                  float* __restrict__ outP) {
  const int hw = threadIdx.x >> 5, c = threadIdx.x & 31;  // lane = channels 2c, 2c+1
  for (long t = (long)blockIdx.x * 8 + hw; t < ntargets; t += (long)gridDim.x * 8) {
    float s0 = 0.f, s1 = 0.f, r0 = 0.f, r1 = 0.f;
    const int e0 = offs[t], e1 = offs[t + 1];
    int s = e0;
#ifndef SAMBLE_SEGSUM_BATCH
#define SAMBLE_SEGSUM_BATCH 4
#endif
    constexpr int SB = SAMBLE_SEGSUM_BATCH;
    for (; s + SB <= e1; s += SB) {  // SB edges = 2 SB rows in flight
      float2 v[SB], w[SB];
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const long e = order[s + u];
        v[u] = *reinterpret_cast<const float2*>(srcE + e * e_rs + 2 * c);
        w[u] = *reinterpret_cast<const float2*>(srcP + (e / KN) * p_rs + 2 * c);
      }
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        s0 += v[u].x;
        s1 += v[u].y;
        r0 += w[u].x;
        r1 += w[u].y;
      }
    }
    for (; s < e1; ++s) {
      const long e = order[s];
      const float2 v = *reinterpret_cast<const float2*>(srcE + e * e_rs + 2 * c);
      const float2 w = *reinterpret_cast<const float2*>(srcP + (e / KN) * p_rs + 2 * c);
      s0 += v.x;
      s1 += v.y;
      r0 += w.x;
      r1 += w.y;
    }
    *reinterpret_cast<float2*>(outE + t * 64 + 2 * c) = make_float2(s0, s1);
    *reinterpret_cast<float2*>(outP + t * 64 + 2 * c) = make_float2(r0, r1);
  }
}

// ------------------------------------------------------------------------------------------------
// Inverse neighbour lists without a sort.  A query lists a target at most once, so the incoming edges of target t
// come from distinct queries and "ascending edge id" is "ascending query": the position of edge (i -> t) in t's
// group is the number of queries i' < i that list t.  Three launches over a bit matrix (B, N targets, ceil(N/32)
// words over the queries):
//   mark     one thread per edge: bits[b][t][i >> 5] |= 1 << (i & 31)                        (atomic OR)
//   count    one workgroup per cloud: per-word prefix popcounts of every target row (uint16), the row totals =
//            in-degrees, their exclusive scan + b N K = the group boundaries
//   place    one thread per edge: order[offset[t] + prefix[t][i >> 5] + popc(bits below i)] = e
// Exact and run-to-run identical (the atomics only set bits).  Replaces a 2 M-key radix sort + bincount + cumsum
// (~450 us per table at B=32, N=2048, K=32) by ~100 us.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void inv_mark_kernel(const int* __restrict__ nn, int N, int K, int W, long nedges,
                                                       unsigned* __restrict__ bits) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= nedges) return;
  const long q = e / K;           // b * N + i
  const int i = (int)(q % N);
  const long b = q / N;
  const int t = nn[e];
  if (t < 0 || t >= N) return;
  atomicOr(&bits[(b * N + t) * W + (i >> 5)], 1u << (i & 31));
}

// mark + count in one launch for clouds of up to 4 096 points: a workgroup owns a block of R target rows of one cloud, keeps
// their membership words in LDS (R W words <= 128 KB), reads the cloud's whole neighbour table once (coalesced, out of
// L2 for the cloud's other blocks) and sets the bits of the edges that point into its block with LDS atomics; the rows
// then leave as whole lines together with their prefix popcounts and in-degrees.  2 M global atomic ORs into a 16 MB
// matrix (memory-side transactions: 62 us per table at B=32, N=2048) + the memset + the row scan become ~10 us.
__global__ __launch_bounds__(1024) void inv_mark_scan_kernel(const int* __restrict__ nn, int N, int K, int W, int R,
                                                             unsigned* __restrict__ bits, unsigned short* __restrict__ pre,
                                                             int* __restrict__ total) {
  extern __shared__ unsigned lbits[];
  int blk, b;
  xcd_assign(blk, b);  // a cloud's blocks on one XCD: its neighbour table is read from HBM once
  const int t0 = blk * R, rows = min(R, N - t0), tid = threadIdx.x;
  for (int w = tid; w < rows * W; w += 1024) lbits[w] = 0u;
  __syncthreads();
  const int* tab = nn + (long)b * N * K;
  const int nedges = N * K;
  auto mark = [&](int e, int tgt) {
    const int t = tgt - t0;
    if (t >= 0 && t < rows) {
      const int i = e / K;
      atomicOr(&lbits[t * W + (i >> 5)], 1u << (i & 31));
    }
  };
  if ((nedges & 3) == 0) {  // (then every cloud's table starts on a 16-byte boundary) eight 16-byte loads in flight per thread
    const int4* tab4 = reinterpret_cast<const int4*>(tab);
    const int n4 = nedges >> 2;
    for (int base = tid; base < n4; base += 8 * 1024) {
      int4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base + 1024 * u < n4 ? tab4[base + 1024 * u] : int4{-1, -1, -1, -1};
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e0 = 4 * (base + 1024 * u);
        mark(e0, v[u].x);
        mark(e0 + 1, v[u].y);
        mark(e0 + 2, v[u].z);
        mark(e0 + 3, v[u].w);
      }
    }
  } else {
    for (int e = tid; e < nedges; e += 1024) mark(e, tab[e]);
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  for (int row = wave; row < rows; row += 16) {
    const unsigned* r = lbits + row * W;
    const long grow = (long)b * N + t0 + row;
    int carry = 0;
    for (int w0 = 0; w0 < W; w0 += 64) {
      const int w = w0 + lane;
      const unsigned word = w < W ? r[w] : 0u;
      const int c = __popc(word);
      int inc = c;  // inclusive scan over the wave
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
      }
      if (w < W) {
        bits[grow * W + w] = word;
        pre[grow * W + w] = (unsigned short)(carry + inc - c);
      }
      carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) total[grow] = carry;
  }
}

// per target row (one wave each): exclusive prefix popcounts of its W membership words (`pre`) and its in-degree.
// Lane = word: coalesced 256-byte reads (round 2: one thread walked a row's words one by one, 32 workgroups in all:
// 125-190 us per table against ~10 for this kernel and the scan below)
__global__ __launch_bounds__(256) void inv_rowscan_kernel(const unsigned* __restrict__ bits, long nrows, int W,
                                                          unsigned short* __restrict__ pre, int* __restrict__ total) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= nrows) return;
  const unsigned* r = bits + row * W;
  unsigned short* p = pre + row * W;
  int carry = 0;
  for (int w0 = 0; w0 < W; w0 += 64) {
    const int w = w0 + lane;
    const int c = w < W ? __popc(r[w]) : 0;
    int inc = c;  // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(inc, o, 64);
      if (lane >= o) inc += v;
    }
    if (w < W) p[w] = (unsigned short)(carry + inc - c);
    carry += __shfl(inc, 63, 64);
  }
  if (lane == 0) total[row] = carry;
}

// per cloud: group boundaries = exclusive scan of the targets' in-degrees, seeded with b * N * K
__global__ __launch_bounds__(1024) void inv_offsets_kernel(const int* __restrict__ total, int N, int K,
                                                           int* __restrict__ offsets) {
  __shared__ int part[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int per = (N + 1023) / 1024;
  const int t0 = min(tid * per, N), t1 = min(t0 + per, N);
  int mine = 0;
  for (int t = t0; t < t1; ++t) mine += total[(long)b * N + t];
  part[tid] = mine;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // inclusive scan of the thread totals
    const int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = (int)((long)b * N * K) + part[tid] - mine;
  for (int t = t0; t < t1; ++t) {
    offsets[(long)b * N + t] = run;
    run += total[(long)b * N + t];
  }
  if (b == gridDim.x - 1 && tid == 1023) offsets[(long)gridDim.x * N] = run;
}

__global__ __launch_bounds__(256) void inv_place_kernel(const int* __restrict__ nn, int N, int K, int W, long nedges,
                                                        const unsigned* __restrict__ bits,
                                                        const unsigned short* __restrict__ pre,
                                                        const int* __restrict__ offsets, int* __restrict__ order) {
  // XCD x (workgroups x, x + 8, ...) takes the clouds x, x + 8, ... one after the other: the bit rows and prefix counts
  // an edge looks up are its cloud's (0.75 MB at N = 2048) and stay in that XCD's L2
  const long epc = (long)N * K, nclouds = nedges / epc;
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  for (long u = (long)(blockIdx.x >> 3) * 256 + threadIdx.x;; u += (long)per_xcd * 256) {
    const long b = (u / epc) * 8 + xcd;
    if (b >= nclouds) break;
    const long e = b * epc + u % epc;
    const int i = (int)((e / K) % N);
    const int t = nn[e];
    if (t < 0 || t >= N) continue;
    const long r = (b * N + t) * W + (i >> 5);
    const int rank = (int)pre[r] + __popc(bits[r] & ((1u << (i & 31)) - 1u));
    order[offsets[b * N + t] + rank] = (int)e;
  }
}

}  // namespace samble

extern "C" size_t samble_inverse_neighbors_ws_bytes(int B, int N) {
  const size_t W = (size_t)(N + 31) / 32;
  return (size_t)B * N * W * 4 + (size_t)B * N * W * 2 + (size_t)B * N * 4 + 768;
}

extern "C" int samble_launch_inverse_neighbors(const int* nn, int B, int N, int K, int* order, int* offsets, int* indeg,
                                               void* ws, hipStream_t s) {
  const int W = (N + 31) / 32;
  const long nedges = (long)B * N * K;
  unsigned* bits = reinterpret_cast<unsigned*>(ws);
  unsigned short* pre = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(ws) + (((size_t)B * N * W * 4 + 255) & ~(size_t)255));
  // (the bit matrix is cleared below, where the global-atomic marking still runs)
  hipError_t e = hipSuccess;
  // PRECONDITION (include/samble.h): every row of nn holds K DISTINCT indices in [0, N), as samble_knn_f32 writes
  // them.  A table that breaks it (duplicates collapse into one bit, out-of-range entries are skipped) places
  // fewer than N*K edges per cloud, and the slots between a cloud's last placed edge and the next cloud's first
  // offset stay unwritten: zero them, so that whatever a consumer gathers through such a slot is edge 0 -- a wrong
  // gradient for a malformed table, never an out-of-bounds read
  e = hipMemsetAsync(order, 0, (size_t)nedges * sizeof(int), s);
  if (e != hipSuccess) return (int)e;
  const unsigned blocks = (unsigned)((nedges + 255) / 256);
  // (in-degrees: the caller's array, or the tail of the workspace)
  int* tot = indeg ? indeg : reinterpret_cast<int*>(reinterpret_cast<char*>(pre) + (((size_t)B * N * W * 2 + 255) & ~(size_t)255));
  // target rows per workgroup of the LDS marking: what 128 KB hold, at least SAMBLE_INV_BLOCKS workgroups per cloud
  const int R = N <= 4096 ? min(((N + SAMBLE_INV_BLOCKS - 1) / SAMBLE_INV_BLOCKS + 15) & ~15, (128 * 1024 / 4) / W) : 0;
  if (R == 0) {
    e = hipMemsetAsync(bits, 0, (size_t)B * N * W * 4, s);
    if (e != hipSuccess) return (int)e;
  }
  samble::Timed timed(samble::kT_inv_nn, s);
  if (R > 0) {
    const int lds = R * W * 4;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(samble::inv_mark_scan_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(samble::inv_mark_scan_kernel, dim3((N + R - 1) / R, B), dim3(1024), lds, s, nn, N, K, W, R, bits, pre,
                       tot);
  } else {
    hipLaunchKernelGGL(samble::inv_mark_kernel, dim3(blocks), dim3(256), 0, s, nn, N, K, W, nedges, bits);
    hipLaunchKernelGGL(samble::inv_rowscan_kernel, dim3((unsigned)(((long)B * N + 3) / 4)), dim3(256), 0, s, bits, (long)B * N,
                       W, pre, tot);
  }
  hipLaunchKernelGGL(samble::inv_offsets_kernel, dim3(B), dim3(1024), 0, s, tot, N, K, offsets);
  {  // one cloud per XCD and pass: 8 x ceil(N K / 256) workgroups (at most 4096)
    const long per = ((long)N * K + 255) / 256;
    const unsigned place_blocks = (unsigned)(8 * (per < 512 ? per : 512));
    hipLaunchKernelGGL(samble::inv_place_kernel, dim3(place_blocks), dim3(256), 0, s, nn, N, K, W, nedges, bits, pre, offsets,
                       order);
  }
  return (int)hipGetLastError();
}

extern "C" int samble_launch_seg_sum_rows64(const float* src, long src_rs, const int* order, const int* offs, int KN, int per_edge,
                                            long ntargets, float* out, hipStream_t s) {
  samble::Timed timed(samble::kT_seg_sum, s);
  hipLaunchKernelGGL(samble::seg_sum_rows64_kernel, dim3(2048), dim3(256), 0, s, src, src_rs, order, offs, KN, per_edge, ntargets,
                     out);
  return (int)hipGetLastError();
}

extern "C" int samble_launch_seg_sum_rows64_pair(const float* srcE, long e_rs, const float* srcP, long p_rs, const int* order,
                                                 const int* offs, int KN, long ntargets, float* outE, float* outP,
                                                 hipStream_t s) {
  samble::Timed timed(samble::kT_seg_sum, s);
  hipLaunchKernelGGL(samble::seg_sum_rows64_pair_kernel, dim3(2048), dim3(256), 0, s, srcE, e_rs, srcP, p_rs, order, offs, KN,
                     ntargets, outE, outP);
  return (int)hipGetLastError();
}

extern "C" size_t samble_n2p_bwd_ws_floats(int B, int N, int KN) {
  return (size_t)B * N * 128 + 2 * (size_t)B * N * KN * 4 + 64;
}

extern "C" int samble_launch_n2p_bwd(const float* qkv, long bs, long rs, const int* nn, const float* g, int B, int N,
                                     int KN, int diff, float scale, float* dqkv, long dbs, long drs, float* ws,
                                     int heads, const int* order, const int* offs, hipStream_t s) {
  using namespace samble;
  if (heads != 1 && heads != 2 && heads != 4) return -22;
  float* gt = ws;
  float* A = gt + (size_t)B * N * 128;
  float* DL = A + (size_t)B * N * KN * 4;
  const size_t lds = (size_t)(2 * kScatRows * 128) * 4 + (size_t)(kScatHits + 256) * 4;
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(n2p_bwd_scatter_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  Timed timed(kT_n2p_bwd, s);  // (transpose + per-point kernel + gather / scatter of the neighbours' shares)
  hipLaunchKernelGGL(transpose_cn_kernel, dim3((N + 31) / 32, B), dim3(256), 0, s, g, N, gt);
  hipLaunchKernelGGL(KN == 32 ? n2p_bwd_point_kernel<true> : n2p_bwd_point_kernel<false>, dim3(2048), dim3(256), 0, s, qkv, bs,
                     rs, nn, gt, N, KN, diff, scale, dqkv, dbs, drs, A, DL, heads, B);
  if (order && offs)
    hipLaunchKernelGGL(n2p_bwd_gather_kernel, dim3(2048), dim3(256), 0, s, qkv, bs, rs, gt, A, DL, order, offs, N, KN,
                       (long)B * N, dqkv, dbs, drs, heads);
  else
    hipLaunchKernelGGL(n2p_bwd_scatter_kernel, dim3((N + kScatRows - 1) / kScatRows, B), dim3(256), lds, s, qkv, bs, rs,
                       nn, gt, A, DL, N, KN, dqkv, dbs, drs, heads);
  return (int)hipGetLastError();
}
