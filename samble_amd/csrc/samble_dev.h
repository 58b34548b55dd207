// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the SAMBLE sampler path.
//
// All matrix work uses v_mfma_f32_32x32x2_f32 (fp32 in / fp32 accumulate, exact fp32):
//   A operand: lane l holds A[row = l & 31][k = l >> 5]
//   B operand: lane l holds B[k = l >> 5][col = l & 31]
//   C/D:       lane l, register r (0..15) holds D[row = crow(r, l >> 5)][col = l & 31],
//              crow(r, h) = (r & 3) + 8 * (r >> 2) + 4 * h
// Two idioms used everywhere (validated against a lane-level emulation before being written):
//   * k-permutation: MFMA step kk (0..63) of a 128-deep contraction consumes channel
//     kperm(kk, h) = 64 * h + kk in lane half h, for BOTH operands, so a lane's operand
//     registers are 64 consecutive floats of its row (16-byte global loads).
//   * accumulator-as-operand: with the reduced index on the accumulator's ROW (register)
//     axis, register t of a 32x32 tile is directly the B operand of step t of the next
//     product (its k pair is rows crow(t,0), crow(t,1)); the other operand is read from LDS
//     at those rows.  No transpose, no LDS round trip for P / dS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// measurement hook (abi.hip): HIP events around a launch when its kernel id is selected
// (samble_timing_select / samble_timing_read, ids = SAMBLE_T_* of include/samble.h)
extern "C" void samble_time_begin(int id, hipStream_t s);
extern "C" void samble_time_end(int id, hipStream_t s);

namespace samble {

enum TimedKernel {
  kT_attn_stats = 1, kT_attn_rows = 2, kT_bwd_dv = 3, kT_knn = 4, kT_attn_fwd = 5, kT_bwd_dq = 6, kT_bwd_dk = 7,
  kT_proj_fwd = 8, kT_proj_dx = 9, kT_proj_dw = 10, kT_tri_split = 11, kT_knn_prep = 12, kT_sparse_score = 13,
  kT_quantiles = 14, kT_bin_assign = 15, kT_alloc_counts = 16, kT_bin_select = 17, kT_bwd_prep = 18,
  kT_gather = 19, kT_knn_seed = 20, kT_select_chain = 21, kT_bwd_rows_f32 = 22, kT_nn_prepare = 23,
  kT_edge_fwd = 24, kT_edge_bwd = 25, kT_n2p_fwd = 26, kT_n2p_bwd = 27, kT_inv_nn = 28, kT_seg_sum = 29,
  kT_edge_sums = 30, kT_knn_small = 31, kT_lin_fwd = 32, kT_lin_dx = 33, kT_lin_dw = 34, kT_lin_amax = 35,
  kT_lin_amax_bwd = 36, kT_bn_fwd = 37, kT_lin_chain = 38, kT_bn_bwd = 39,
};
// the three 128 x 128 projection weights [Wq; Wk; Wv] where they live (one (384, 128) block, or three tensors)
struct ProjW {
  const float* q;
  const float* k;
  const float* v;
  __host__ __device__ const float* row(int o) const { return (o < 128 ? q : o < 256 ? k : v) + (long)(o & 127) * 128; }
};
inline ProjW proj_w(const float* W, const float* Wk, const float* Wv) {
  return Wk ? ProjW{W, Wk, Wv} : ProjW{W, W + 128 * 128, W + 2 * 128 * 128};
}

struct Timed {  // brackets the launches made during its lifetime
  int id;
  hipStream_t s;
  Timed(int i, hipStream_t st) : id(i), s(st) { samble_time_begin(i, st); }
  ~Timed() { samble_time_end(id, s); }
};

// ---- cross-lane moves that stay on the vector ALU (round 6) ---------------------------------------------------------------
// `__shfl_xor(x, o)` compiles to `ds_bpermute_b32`: an LDS-crossbar instruction with its address arithmetic and a wait.  Inside
// a row of 16 lanes DPP does the same move fused into the consumer (quad_perm for lane ^ 1 and lane ^ 2; once a quad / an octet
// holds one value, row_half_mirror / row_mirror reach the other quad / octet); across rows gfx950 has v_permlane16_swap and
// v_permlane32_swap: with both operands = x they return (rows 0 0 2 2, rows 1 1 3 3) and (rows 0 1 0 1, rows 2 3 2 3) of x, so
// combining the two results lane by lane is the xor-16 / xor-32 butterfly step.  max / min / + of the same values in the same
// pairing as the xor butterfly: identical results.
template <int CTRL>
__device__ __forceinline__ unsigned lane_dpp_u(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
}
template <class Op>
__device__ __forceinline__ float wave_butterfly64(float x, Op op) {   // every lane: op over all 64 lanes
  x = op(x, __uint_as_float(lane_dpp_u<0xB1>(__float_as_uint(x))));     // quad_perm [1,0,3,2]
  x = op(x, __uint_as_float(lane_dpp_u<0x4E>(__float_as_uint(x))));     // quad_perm [2,3,0,1]
  x = op(x, __uint_as_float(lane_dpp_u<0x141>(__float_as_uint(x))));    // row_half_mirror
  x = op(x, __uint_as_float(lane_dpp_u<0x140>(__float_as_uint(x))));    // row_mirror
  const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = op(__uint_as_float(p[0]), __uint_as_float(p[1]));
  const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return op(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float wave_max64(float x) { return wave_butterfly64(x, [](float a, float b) { return fmaxf(a, b); }); }
// op(x, value of lane ^ 32)
__device__ __forceinline__ float xor32_max(float x) {
  const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ int xor32_min(int x) {
  const auto q = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
  return min((int)q[0], (int)q[1]);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
constexpr int kTile = 32;           // MFMA tile edge
// Row stride (floats) of a [row][128] LDS tile that is read by rows (lane = row, 16 bytes = 4 k-steps
// per ds_read_b128): 132 = 33 x 16 B keeps rows 16-byte aligned, and 33 being odd makes both the
// 16-byte row reads (16-lane groups) and the 16-byte staging writes bank-conflict free (checked
// against the bank rules of MI355X_MICROARCH.md).  Column-wise reads (lane = channel) are
// conflict free for any stride.
constexpr int kLdsPad = 132;
constexpr float kNegInf = -__builtin_huge_valf();

__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// Load this lane's 64 operand floats of row `row_ptr` (128 contiguous floats): channels
// 64*h .. 64*h+63, as sixteen 16-byte loads.
__device__ __forceinline__ void load_row_half(const float* __restrict__ row_ptr, int h, float (&dst)[64]) {
  const f32x4* p = reinterpret_cast<const f32x4*>(row_ptr + 64 * h);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    f32x4 v = p[i];
    dst[4 * i + 0] = v[0];
    dst[4 * i + 1] = v[1];
    dst[4 * i + 2] = v[2];
    dst[4 * i + 3] = v[3];
  }
}

// Cooperative copy of a [32][128] fp32 tile (rows `row0..row0+31` of a point-major matrix
// with `row_stride` floats per row) into LDS with row stride `lds_stride`.  256 threads,
// 16-byte global loads (a row is 512 contiguous bytes) and 16-byte LDS stores (lds_stride must be
// a multiple of 4).  Rows >= n_rows are filled with zeros.  Split in two so the global loads can be issued a phase early.
template <int NTHREADS = 256>
struct TileRegsT {
  static constexpr int kPer = 1024 / NTHREADS;  // float4 per thread: 32 rows x 32 float4 per tile
  f32x4 v[kPer];
};
using TileRegs = TileRegsT<256>;

template <int NTHREADS>
__device__ __forceinline__ void tile_load_issue(TileRegsT<NTHREADS>& t, const float* __restrict__ base,
                                                long row_stride, int row0, int n_rows, int tid) {
#pragma unroll
  for (int i = 0; i < TileRegsT<NTHREADS>::kPer; ++i) {
    int e = tid + NTHREADS * i;   // float4 index in the tile: 32 rows x 32 float4
    int r = e >> 5, c4 = e & 31;
    int row = row0 + r;
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    t.v[i] = (row < n_rows) ? *reinterpret_cast<const f32x4*>(base + (long)row * row_stride + 4 * c4) : z;
  }
}

template <int NTHREADS>
__device__ __forceinline__ void tile_store_lds(const TileRegsT<NTHREADS>& t, float* __restrict__ lds, int lds_stride,
                                               int tid) {
#pragma unroll
  for (int i = 0; i < TileRegsT<NTHREADS>::kPer; ++i) {
    int e = tid + NTHREADS * i;
    int r = e >> 5, c4 = e & 31;
    *reinterpret_cast<f32x4*>(lds + r * lds_stride + 4 * c4) = t.v[i];
  }
}

// acc(32x32) += Ltile(rows from LDS, read row-wise) x Rreg(64 register-resident floats)^T
//   D[row][col] += sum_c L[row][c] * R[col][c]; lane (x, h) supplies L[x][kperm] and R[x][kperm].
__device__ __forceinline__ f32x16 mma_rows_x_regs(const float* __restrict__ lds_tile, int lds_stride, int lane_lo,
                                                  int h, const float (&reg)[64], f32x16 acc) {
  const f32x4* lp = reinterpret_cast<const f32x4*>(lds_tile + lane_lo * lds_stride + 64 * h);
#pragma unroll
  for (int q4 = 0; q4 < 16; ++q4) {
    const f32x4 a = lp[q4];  // one ds_read_b128 feeds four MFMA steps
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = mfma32(a[e], reg[4 * q4 + e], acc);
  }
  return acc;
}

// Same contraction with the register operand on the A side (rows) and LDS on the B side.
__device__ __forceinline__ f32x16 mma_regs_x_rows(const float (&reg)[64], const float* __restrict__ lds_tile,
                                                  int lds_stride, int lane_lo, int h, f32x16 acc) {
  const f32x4* lp = reinterpret_cast<const f32x4*>(lds_tile + lane_lo * lds_stride + 64 * h);
#pragma unroll
  for (int q4 = 0; q4 < 16; ++q4) {
    const f32x4 b = lp[q4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = mfma32(reg[4 * q4 + e], b[e], acc);
  }
  return acc;
}

// out[dt](32 x 32) += T(:, 32dt..)^T x P  with P = a 32x32 accumulator whose ROW axis is the
// reduced index:  out[dt][d][col] += sum_row T[row][32dt + d] * P[row][col].
// T is an LDS [32][stride] tile; lane (d, h) reads T[crow(t,h)][32dt + d].
__device__ __forceinline__ void mma_tileT_x_acc(const float* __restrict__ lds_tile, int lds_stride, int lane_lo, int h,
                                                const f32x16& p, f32x16 (&out)[4]) {
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float* row = lds_tile + crow(t, h) * lds_stride + lane_lo;
    float b = p[t];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) out[dt] = mfma32(row[32 * dt], b, out[dt]);
  }
}

// One step t of mma_tileT_x_acc (rows crow(t,0), crow(t,1) of the tile): lets the caller interleave
// the VALU that produces register t of P / dS with the MFMAs that consume it.
__device__ __forceinline__ void mma_tileT_step(const float* __restrict__ lds_tile, int lds_stride, int lane_lo, int h,
                                               int t, float b, f32x16 (&out)[4]) {
  const float* row = lds_tile + crow(t, h) * lds_stride + lane_lo;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) out[dt] = mfma32(row[32 * dt], b, out[dt]);
}

// XCD-aware (chunk, cloud) assignment for a grid (chunks, clouds): workgroups are dealt round-robin
// over the 8 XCDs, each with a private 4 MB L2.  Give every XCD its own clouds so that the K/V (or
// key-set) tiles all of a cloud's workgroups stream are re-read from that XCD's L2 rather than
// across the fabric.  Placement only changes speed: any bijection is correct, and grids whose
// cloud count is not a multiple of 8 keep the identity mapping.
__device__ __forceinline__ void xcd_assign(int& chunk, int& cloud) {
  chunk = blockIdx.x;
  cloud = blockIdx.y;
  if ((gridDim.y & 7) == 0) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, slot = lin >> 3, per = gridDim.y >> 3;
    cloud = xcd * per + slot % per;
    chunk = slot / per;
  }
}

// Which cloud's slice of the two M-row maps (P, dS) cloud b uses: its own.  -DSAMBLE_MAP_ALIAS (timing-only scratch
// builds, tools/experiments/map_alias.md; results are WRONG): B = 32 clouds share 8 slices, each one by four clouds that
// xcd_assign puts on four different XCDs -- the maps' working set (2 x 67 MB) then lives in the memory-side cache, an
// upper bound of what a backward that keeps the maps on-die could gain.
// (modes 3 / 4: as 2, but only the P map's / only the dS map's accesses of the two query-stationary kernels: `which` = 0 P, 1 dS)
__device__ __forceinline__ long map_cloud(int b, int which = -1) {
#if defined(SAMBLE_MAP_ALIAS) && SAMBLE_MAP_ALIAS == 2
  return 0;  // every cloud on ONE slice: 2 x 8.4 MB, the strongest form of the bound
#elif defined(SAMBLE_MAP_ALIAS) && (SAMBLE_MAP_ALIAS == 3 || SAMBLE_MAP_ALIAS == 4)
  return which == SAMBLE_MAP_ALIAS - 3 ? 0 : (long)b;
#elif defined(SAMBLE_MAP_ALIAS)
  return (long)(((b & 3) << 1) | ((b >> 4) & 1));
#else
  return (long)b;
#endif
}

// value of lane ^ 32 (`ds_bpermute_b32`).  Its callers are the attention kernels' per-tile statistics; v_permlane32_swap in its
// place was measured at +-0 there (round 6: three alternating runs per build on one box, the step's kernels within their
// run-to-run spread), so the form that needs no lane test stays.  The gather kernels' head sums, where the cross-lane traffic
// is a third of the instructions, take the vector-ALU forms above.
__device__ __forceinline__ float wave_xor32(float v) { return __shfl_xor(v, 32, 64); }

// order-preserving float -> uint32 (larger float -> larger uint); NaN sorts above +inf
__device__ __forceinline__ uint32_t ordered_bits(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

}  // namespace samble
