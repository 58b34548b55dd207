// extern "C" boundary of libsamble_hip.so (include/samble.h): argument checks, workspace carving,
// error reporting.  Never throws, allocates, frees or synchronises.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

#include "../../include/samble.h"

#define SAMBLE_API extern "C" __attribute__((visibility("default")))

// kernel launchers (one per .hip file)
extern "C" {
size_t samble_knn_ws_floats(int B, int C, int Nq, int Nk, int K, int variant);
int samble_launch_knn(const float*, long, int, const float*, long, int, int, int, int, int, int*, float*, float*, hipStream_t);
int samble_launch_attn_fwd(const float*, long, long, const float*, long, long, const float*, long, long, int, int, int,
                           float, float*, float*, float*, int, float*, hipStream_t);
int samble_launch_attn_colsum(const float*, long, long, const float*, long, long, const float*, int, int, float, float*,
                              hipStream_t);
int samble_launch_stat_score(const float*, int, int, float*, float*, hipStream_t);
size_t samble_score_ws_bytes(int B, int N);
int samble_launch_sparse_score(const float*, long, long, const float*, long, long, const float*, const int*, int, int,
                               int, float, int, float*, float*, int*, void*, hipStream_t);
int samble_launch_zscore(const float*, int, int, float*, hipStream_t);
int samble_launch_batch_quantiles(const float*, long, int, float*, void*, hipStream_t);
size_t samble_quantiles_ws_bytes(void);
int samble_launch_bin_assign(const float*, const float*, int, const float*, const float*, int, int, int, int,
                             unsigned char*, int*, float*, float*, hipStream_t);
int samble_launch_alloc_counts(const float*, const int*, int, int, int, int*, hipStream_t);
int samble_launch_bin_select(const float*, const float*, const unsigned char*, const int*, const float*, unsigned long long,
                             unsigned long long, int, int, int, int, int, int, float, long long*, hipStream_t);
int samble_launch_exp1_noise(unsigned long long, unsigned long long, long, float*, hipStream_t);
int samble_launch_attn_heads_fwd(const float*, long, long, const float*, long, long, const float*, long, long, const float*,
                                 float, float, int, int, int, int, float*, long, long, float*, hipStream_t);
int samble_launch_attn_heads_bwd(const float*, long, long, const float*, long, long, const float*, long, long, const float*,
                                 float, float, int, int, int, int, const float*, long, long, const float*, const float*, long,
                                 long, float*, float*, long, long, float*, long, long, float*, long, long, float*, hipStream_t);
int samble_launch_gather_rows(const float*, long, long, const long long*, int, int, float*, hipStream_t);
int samble_launch_blend_boundaries(const float*, float*, float*, int, float, float, int, hipStream_t);
int samble_edge_waves(void);
int samble_launch_edge_gather_sums(const float*, const int*, int, int, float*, float*, hipStream_t);
int samble_launch_edge_mlp_fwd(const float*, const float*, const int*, const float*, int, int, float*, float*,
                               unsigned char*, unsigned char*, double*, hipStream_t);
int samble_launch_edge_mlp_bwd(const float*, const float*, const int*, const float*, const unsigned char*, const float*,
                               const float*, int, int, float*, float*, float*, hipStream_t);
int samble_launch_group_gather(const float*, const int*, int, int, int, int, int, float*, hipStream_t);
int samble_launch_fps(const float*, const long long*, int, int, int, long long*, hipStream_t);
int samble_launch_gather_points(const float*, int, int, int, const long long*, int, float*, hipStream_t);
size_t samble_proj_tri_image_bytes();
int samble_launch_proj_fwd(const float*, long, int, int, const float*, int, const float*, const float*, const float*, float*,
                           long, long, float*, void*, void* const*, int, void*, hipStream_t);
size_t samble_proj_bwd_ws_floats(int B, int N);
int samble_launch_proj_bwd(const float*, long, long, const float*, long, int, int, const float*, int, const float*,
                           const float*, const float*, float*, long, float*, float*, float*, void*, const void*, const float*,
                           hipStream_t);
size_t samble_n2p_bwd_ws_floats(int B, int N, int KN);
int samble_launch_n2p_bwd(const float*, long, long, const int*, const float*, int, int, int, int, float, float*, long,
                          long, float*, int, const int*, const int*, hipStream_t);
int samble_launch_seg_sum_rows64(const float*, long, const int*, const int*, int, int, long, float*, hipStream_t);
int samble_launch_seg_sum_rows64_pair(const float*, long, const float*, long, const int*, const int*, int, long, float*, float*,
                                      hipStream_t);
size_t samble_inverse_neighbors_ws_bytes(int B, int N);
int samble_launch_inverse_neighbors(const int*, int, int, int, int*, int*, int*, void*, hipStream_t);
int samble_launch_n2p_fwd(const float*, long, long, const int*, int, int, int, int, float, float*, int, float*,
                          const float*, hipStream_t);
int samble_launch_attn_bwd(const float*, long, long, const float*, long, long, const float*, long, long, const float*,
                           const float*, const float*, int, const float*, const long long*, const float*, int, int, int,
                           int, float, float*, float*, float*, float*, float*, float*, float*, long, long, float*, long,
                           long, float*, long, long, int, float*, float*, const void*, const void*, void*, int, int, hipStream_t);
int samble_attn_bwd_prep_clears_dq(const float*, const float*, const void*, const void*, const void*);
int samble_attn_map_ld(int N, int nt);
size_t samble_tri_image_size(int, int, int);
size_t samble_bwd_tri_dsmap_bytes(int, int, int);
int samble_launch_tri_split(const float*, long, long, int, int, void*, void*, hipStream_t);
int samble_launch_tri_split_qkv(const float*, long, long, int, int, int, void*, void*, void*, void*, void*, hipStream_t);
int samble_launch_k_to_duo(void*, int, int, int, hipStream_t);
int samble_launch_attn_rows_tri(const float*, int, const float*, const void*, const long long*, int, int, int, int, float*,
                                hipStream_t);
int samble_launch_attn_stats_nl_tri(const void*, const void*, int, int, int, float, const unsigned*, int, float*, float*,
                                    float*, const int*, int, void*, size_t, hipStream_t);
int samble_launch_attn_rows_rc_tri(const void*, const void*, const void*, const float*, const long long*, int, int, int,
                                   int, float, float*, float*, int, hipStream_t);
int samble_launch_nn_prepare(const int*, int, int, int, int*, unsigned*, void*, size_t, hipStream_t);
int samble_launch_attn_stats_tri(const void*, const void*, int, int, int, float, float*, int, float*, float*, const float*,
                                 const float*, hipStream_t);
int samble_launch_attn_stats(const float*, long, long, const float*, long, long, int, int, int, float, float*, int,
                             float*, float*, const float*, const float*, hipStream_t);
int samble_launch_attn_rows(const float*, int, const float*, const float*, long, long, const long long*, int, int, int,
                            int, float*, hipStream_t);
int samble_launch_sparse_score_map(const float*, int, const float*, const int*, int, int, int, int, float*, float*, int*,
                                   void*, hipStream_t);
size_t samble_attn_bwd_slab_floats(int B, int N, int M);
size_t samble_chain_ws_bytes(void);
int samble_chain_supported(int B, int N, int nb);
int samble_launch_sparse_score_map_acc(const float*, int, const float*, const int*, int, int, int, int, void*, size_t,
                                       hipStream_t);
int samble_launch_score_quantiles(const void*, const int*, const float*, int, int, int, int, float*, float*, int*, void*,
                                  float*, unsigned int, int*, hipStream_t);
int samble_launch_bin_plan(const float*, const float*, int, const float*, const float*, float*, float*, int, float, float, int, int, int,
                           int, int, unsigned char*, int*, float*, float*, int*, void*, unsigned int, int*, hipStream_t);
int samble_launch_select_chain(const void*, const int*, const float*, int, int, int, int, float*, float*, int*, void*, float*,
                               const float*, int, float*, float*, int, float, float, int, int, unsigned char*, int*, float*,
                               float*, int*, unsigned int, int*, hipStream_t);
size_t samble_chain_flag_offset(void);
size_t samble_edge_glue_part_bytes(void);
size_t samble_edge_glue_cst_bytes(void);
size_t samble_edge_glue_st_bytes(void);
size_t samble_edge_glue_pool_bytes(void);
int samble_launch_edge_pre(const float*, const float*, long, const int*, int, int, const float*, const float*, float, float*, float*,
                           float, long long*, float*, float*, float*, float*, float*, double*, double*, int, double*, hipStream_t);
int samble_launch_edge_post(const float*, const float*, const unsigned char*, const unsigned char*, const double*, int, int, int,
                            const float*, const float*, float, float*, float*, float, long long*, float*, double*, float*, unsigned char*,
                            float*, int, double*, hipStream_t);
int samble_launch_edge_bwd_pre(const float*, const float*, int, int, const float*, float*, const double*, float*, float*, float*,
                               double*, int, double*, hipStream_t);
int samble_launch_edge_bwd_post(const float*, const float*, long, const float*, const float*, const float*, const float*, const int*,
                                int, int, float*, const double*, const float*, int, float*, float*, long, float*, float*, float*,
                                double*, int, double*, hipStream_t);
int samble_launch_interp_fwd(const float*, int, int, int, const int*, const float*, int, int, float*, float*, hipStream_t);
int samble_launch_interp_bwd(const float*, int, int, int, const float*, const int*, const int*, int, int, float*, void*, hipStream_t);
size_t samble_interp_bwd_ws_bytes(int, int, int, int);
size_t samble_linear_image_bytes_impl(int O);
int samble_launch_linear_images(const float*, int, void*, void*, int, hipStream_t);
int samble_linear_is_duo(void);
int samble_launch_linear_chain(const float*, long, int, int, const void*, const void*, int, int, float*, long, long, void*, float*,
                               long, const float*, hipStream_t);
int samble_launch_linear_images_pair(const float*, int, void*, void*, const float*, int, void*, void*, hipStream_t);
int samble_launch_linear_fwd(const float*, long, int, int, int, const void*, int, int, const float*, float*, long, long, hipStream_t);
size_t samble_linear_amax_ws_bytes(int, int, int);
int samble_launch_linear_amax(const float*, long, int, int, const void*, int, float*, int*, void*, hipStream_t);
int samble_launch_linear_dx(const float*, long, long, const void*, int, int, int, int, float*, long, const float*, hipStream_t);
size_t samble_linear_dw_ws_bytes(int, int, int);
int samble_launch_linear_dw(const float*, long, long, const float*, long, int, int, int, int, float*, int, void*, hipStream_t, int);
int samble_launch_linear_fwd_cm(const float*, long, int, int, int, const void*, int, int, float*, long, hipStream_t);
size_t samble_amax_bwd_ws_bytes(int, int, int);
int samble_launch_amax_bwd(const float*, long, int, int, const int*, const float*, const float*, int, float*, long, float*,
                           void*, hipStream_t);
}

namespace {
thread_local char g_err[256] = "";

__global__ void zero_rows_kernel(float* p, long bs, long rs, int N, int quads_per_row, long total) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c4 = (int)(e % quads_per_row);
  const long row = e / quads_per_row;
  const long b = row / N, n = row % N;
  float4 z = {0.f, 0.f, 0.f, 0.f};
  *reinterpret_cast<float4*>(p + b * bs + n * rs + 4 * c4) = z;
}

int fail(int code, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s", what);
  return code;
}
int done(int hip_code, const char* where) {
  if (hip_code == 0) return SAMBLE_OK;
  snprintf(g_err, sizeof(g_err), "%s: HIP error %d (%s)", where, hip_code, hipGetErrorString((hipError_t)hip_code));
  return hip_code < 0 ? hip_code : SAMBLE_E_HIP_BASE - hip_code;
}
float inv_sqrt_d(int D) { return (float)(1.0 / sqrt((double)D)); }
}  // namespace

// ---- measurement hook (the only entry points that are not part of the path): HIP events around the
// library's own launches of the SELECTED kernels, on the stream each is launched on.  Kernel ids are
// listed in include/samble.h (SAMBLE_T_*).  Up to kTimeSlots launches per kernel are kept (round robin).
// Process-wide and not thread-safe by design: a benchmark selects, runs its steps, then reads.
constexpr int kTimeSlots = 32;
constexpr int kTimeIds = 48;
static unsigned long long g_time_mask = 0;
static hipEvent_t g_time_ev[kTimeIds][kTimeSlots][2];
static bool g_time_have[kTimeIds];
static int g_time_count[kTimeIds];
extern "C" void samble_time_begin(int id, hipStream_t s) {
  if ((g_time_mask >> id) & 1ull) (void)hipEventRecord(g_time_ev[id][g_time_count[id] % kTimeSlots][0], s);
}
extern "C" void samble_time_end(int id, hipStream_t s) {
  if ((g_time_mask >> id) & 1ull) {
    (void)hipEventRecord(g_time_ev[id][g_time_count[id] % kTimeSlots][1], s);
    ++g_time_count[id];
  }
}
SAMBLE_API int samble_timing_select(uint64_t kernel_mask) {
  g_time_mask = 0;
  for (int id = 1; id < kTimeIds; ++id) {
    g_time_count[id] = 0;
    if (!((kernel_mask >> id) & 1ull) || g_time_have[id]) continue;
    for (int i = 0; i < kTimeSlots; ++i)
      for (int k = 0; k < 2; ++k)
        if (hipEventCreate(&g_time_ev[id][i][k]) != hipSuccess)
          return fail(SAMBLE_E_INVALID, "samble_timing_select: cannot create events");
    g_time_have[id] = true;
  }
  g_time_mask = kernel_mask & ~1ull & ((1ull << kTimeIds) - 1);
  return SAMBLE_OK;
}
SAMBLE_API int samble_timing_read(int id, float* mean_ms, float* median_ms, int* launches) {
  if (id < 1 || id >= kTimeIds) return fail(SAMBLE_E_INVALID, "samble_timing_read: unknown kernel id");
  const int seen = g_time_count[id];
  const int n = seen < kTimeSlots ? seen : kTimeSlots;
  if (launches) *launches = seen;
  if (n == 0) return fail(SAMBLE_E_INVALID, "samble_timing_read: no launch of that kernel was recorded");
  float ms[kTimeSlots];
  double total = 0.0;
  for (int i = 0; i < n; ++i) {
    if (hipEventSynchronize(g_time_ev[id][i][1]) != hipSuccess ||
        hipEventElapsedTime(&ms[i], g_time_ev[id][i][0], g_time_ev[id][i][1]) != hipSuccess)
      return fail(SAMBLE_E_INVALID, "samble_timing_read: event query failed");
    total += ms[i];
  }
  for (int i = 1; i < n; ++i)  // insertion sort of <= 32 samples
    for (int j = i; j > 0 && ms[j] < ms[j - 1]; --j) {
      const float t = ms[j];
      ms[j] = ms[j - 1];
      ms[j - 1] = t;
    }
  if (mean_ms) *mean_ms = (float)(total / n);
  if (median_ms) *median_ms = (n & 1) ? ms[n / 2] : 0.5f * (ms[n / 2 - 1] + ms[n / 2]);
  return SAMBLE_OK;
}

SAMBLE_API const char* samble_version(void) { return "samble-hip 0.3 (gfx950)"; }
SAMBLE_API int samble_abi_version(void) { return SAMBLE_ABI_VERSION; }
SAMBLE_API const char* samble_last_error(void) { return g_err; }

SAMBLE_API size_t samble_knn_workspace_bytes(int B, int C, int Nq, int Nk, int K, int variant) {
  return samble_knn_ws_floats(B, C, Nq, Nk, K, variant) * sizeof(float);
}

SAMBLE_API int samble_knn_f32(const float* xq, int64_t q_bs, int Nq, const float* xk, int64_t k_bs, int Nk, int B, int C,
                              int K, int variant, int32_t* idx_out, float* dist_out, void* ws, size_t ws_bytes,
                              void* stream) {
  if (!xq || !xk || !idx_out || !ws) return fail(SAMBLE_E_INVALID, "samble_knn_f32: null pointer");
  if (B <= 0 || C <= 0 || Nq <= 0 || Nk <= 0 || K <= 0 || K > Nk)
    return fail(SAMBLE_E_INVALID, "samble_knn_f32: need B,C,Nq,Nk > 0 and 0 < K <= Nk");
  if (variant < 0 || variant > (SAMBLE_KNN_FP32_MFMA | SAMBLE_KNN_TWO_KERNEL))
    return fail(SAMBLE_E_INVALID, "samble_knn_f32: unknown variant bits");
  if (ws_bytes < samble_knn_workspace_bytes(B, C, Nq, Nk, K, variant))
    return fail(SAMBLE_E_WORKSPACE, "samble_knn_f32: workspace too small (samble_knn_workspace_bytes)");
  int rc = samble_launch_knn(xq, q_bs, Nq, xk, k_bs, Nk, B, C, K, variant, idx_out, dist_out, (float*)ws,
                             (hipStream_t)stream);
  if (rc == -22) return fail(SAMBLE_E_INVALID, "samble_knn_f32: K must be one of 1,3,8,16,20,32,40,64");
  return done(rc, "samble_knn_f32");
}

SAMBLE_API int samble_attn_fwd_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs,
                                   int64_t k_rs, const float* V, int64_t v_bs, int64_t v_rs, int B, int N, int nt, int D,
                                   float* O, float* lse, float* tok, float* row_std, void* stream) {
  if (!Q || !K || !V || !O || !lse) return fail(SAMBLE_E_INVALID, "samble_attn_fwd_f32: null pointer");
  if (D != 128) return fail(SAMBLE_E_INVALID, "samble_attn_fwd_f32: D must be 128");
  if (B <= 0 || N <= 0 || nt < 0 || nt > 8 || (nt > 0 && !tok))
    return fail(SAMBLE_E_INVALID, "samble_attn_fwd_f32: bad B/N/nt");
  if ((q_rs & 3) || (k_rs & 3) || (v_rs & 3) || (q_bs & 3) || (k_bs & 3) || (v_bs & 3))
    return fail(SAMBLE_E_INVALID, "samble_attn_fwd_f32: strides must be multiples of 4 elements (16-byte rows)");
  return done(samble_launch_attn_fwd(Q, q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs, B, N, N + nt, inv_sqrt_d(D), O, lse,
                                     tok, nt, row_std, (hipStream_t)stream),
              "samble_attn_fwd_f32");
}

SAMBLE_API size_t samble_score_workspace_bytes(int B, int N) { return samble_score_ws_bytes(B, N); }

SAMBLE_API int samble_sparse_score_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs,
                                       int64_t k_rs, const float* lse, const int32_t* nn, int B, int N, int KN, int D,
                                       int mode, float* score, float* z, int32_t* indeg_out, void* ws, size_t ws_bytes,
                                       void* stream) {
  if (!Q || !K || !lse || !nn || !score || !z || !ws)
    return fail(SAMBLE_E_INVALID, "samble_sparse_score_f32: null pointer");
  if (D != 128) return fail(SAMBLE_E_INVALID, "samble_sparse_score_f32: D must be 128");
  if (mode < 0 || mode > SAMBLE_SCORE_SPARSE_ROW_STD)
    return fail(SAMBLE_E_INVALID, "samble_sparse_score_f32: unknown score mode");
  if (ws_bytes < samble_score_ws_bytes(B, N))
    return fail(SAMBLE_E_WORKSPACE, "samble_sparse_score_f32: workspace too small");
  return done(samble_launch_sparse_score(Q, q_bs, q_rs, K, k_bs, k_rs, lse, nn, B, N, KN, inv_sqrt_d(D), mode, score, z,
                                         indeg_out, ws, (hipStream_t)stream),
              "samble_sparse_score_f32");
}

SAMBLE_API int samble_zscore_f32(const float* score, int B, int N, float* z, void* stream) {
  if (!score || !z || B <= 0 || N <= 0) return fail(SAMBLE_E_INVALID, "samble_zscore_f32: bad argument");
  return done(samble_launch_zscore(score, B, N, z, (hipStream_t)stream), "samble_zscore_f32");
}

SAMBLE_API size_t samble_quantiles_workspace_bytes(void) { return samble_quantiles_ws_bytes(); }

SAMBLE_API int samble_batch_quantiles_f32(const float* z, int64_t n, int nb, float* out, void* ws, size_t ws_bytes,
                                          void* stream) {
  if (ws && ws_bytes < samble_quantiles_ws_bytes())
    return fail(SAMBLE_E_WORKSPACE, "samble_batch_quantiles_f32: workspace too small");
  if (!z || !out || n <= 0) return fail(SAMBLE_E_INVALID, "samble_batch_quantiles_f32: bad argument");
  if (nb < 2 || nb > 8) return fail(SAMBLE_E_INVALID, "samble_batch_quantiles_f32: need 2 <= num_bins <= 8");
  if (n >= (1ll << 31)) return fail(SAMBLE_E_INVALID, "samble_batch_quantiles_f32: n must be < 2^31");
  return done(samble_launch_batch_quantiles(z, n, nb, out, ws, (hipStream_t)stream), "samble_batch_quantiles_f32");
}

SAMBLE_API int samble_blend_boundaries_f32(const float* quantiles, float* upper, float* lower, int nb, float momentum,
                                           float one_minus_momentum, int first, void* stream) {
  if (!quantiles || !upper || !lower) return fail(SAMBLE_E_INVALID, "samble_blend_boundaries_f32: null pointer");
  if (nb < 2 || nb > 8) return fail(SAMBLE_E_INVALID, "samble_blend_boundaries_f32: need 2 <= num_bins <= 8");
  return done(samble_launch_blend_boundaries(quantiles, upper, lower, nb, momentum, one_minus_momentum, first,
                                             (hipStream_t)stream),
              "samble_blend_boundaries_f32");
}

SAMBLE_API int samble_bin_assign_f32(const float* z, const float* tok, int nt, const float* upper, const float* lower,
                                     int B, int N, int nb, int relu_first, uint8_t* member, int32_t* cap, float* w_pre,
                                     float* w, void* stream) {
  if (!z || !tok || !upper || !lower || !member || !cap || !w_pre || !w)
    return fail(SAMBLE_E_INVALID, "samble_bin_assign_f32: null pointer");
  if (nb < 1 || nb > 8 || (nt != 1 && nt != nb))
    return fail(SAMBLE_E_INVALID, "samble_bin_assign_f32: need 1 <= num_bins <= 8 and nt in {1, num_bins}");
  return done(samble_launch_bin_assign(z, tok, nt, upper, lower, B, N, nb, relu_first, member, cap, w_pre, w,
                                       (hipStream_t)stream),
              "samble_bin_assign_f32");
}

SAMBLE_API int samble_alloc_counts_f32(const float* w, const int32_t* cap, int B, int nb, int M, int32_t* counts,
                                       void* stream) {
  if (!w || !cap || !counts) return fail(SAMBLE_E_INVALID, "samble_alloc_counts_f32: null pointer");
  if (B < 1 || B > 1024 || nb < 1 || nb > 8 || M < 0)
    return fail(SAMBLE_E_INVALID, "samble_alloc_counts_f32: need 1 <= B <= 1024, 1 <= num_bins <= 8");
  return done(samble_launch_alloc_counts(w, cap, B, nb, M, counts, (hipStream_t)stream), "samble_alloc_counts_f32");
}

SAMBLE_API int samble_bin_select_f32(const float* score, const float* z, const uint8_t* member, const int32_t* counts,
                                     const float* noise, int B, int N, int nb, int M, int sample_mode, int temp_mode,
                                     float temp, int64_t* idx_out, void* stream) {
  if (!score || !z || !member || !counts || !idx_out)
    return fail(SAMBLE_E_INVALID, "samble_bin_select_f32: null pointer");
  if (sample_mode < 0 || sample_mode > SAMBLE_SAMPLE_BOTTOM_RAW)
    return fail(SAMBLE_E_INVALID, "Please check the setting of bin sample mode. It must be topk, uniform or random!");
  if ((sample_mode == SAMBLE_SAMPLE_UNIFORM || sample_mode == SAMBLE_SAMPLE_RANDOM) && !noise)
    return fail(SAMBLE_E_INVALID, "samble_bin_select_f32: uniform/random need the Exp(1) noise tensor");
  if (nb < 1 || nb > 8 || N > 16384) return fail(SAMBLE_E_INVALID, "samble_bin_select_f32: need num_bins <= 8, N <= 16384");
  if (B <= 0 || N <= 0 || M < 1 || M > N) return fail(SAMBLE_E_INVALID, "samble_bin_select_f32: need B, N > 0 and 1 <= M <= N");
  return done(samble_launch_bin_select(score, z, member, counts, noise, 0, 0, B, N, nb, M, sample_mode, temp_mode, temp,
                                       (long long*)idx_out, (hipStream_t)stream),
              "samble_bin_select_f32");
}

SAMBLE_API int samble_bin_select_seeded_f32(const float* score, const float* z, const uint8_t* member,
                                            const int32_t* counts, uint64_t seed, uint64_t offset, int B, int N, int nb,
                                            int M, int sample_mode, int temp_mode, float temp, int64_t* idx_out,
                                            void* stream) {
  if (!score || !z || !member || !counts || !idx_out)
    return fail(SAMBLE_E_INVALID, "samble_bin_select_seeded_f32: null pointer");
  if (sample_mode < 0 || sample_mode > SAMBLE_SAMPLE_BOTTOM_RAW)
    return fail(SAMBLE_E_INVALID, "Please check the setting of bin sample mode. It must be topk, uniform or random!");
  if (offset & 3) return fail(SAMBLE_E_INVALID, "samble_bin_select_seeded_f32: the Philox offset is a multiple of 4");
  if (nb < 1 || nb > 8 || N > 16384)
    return fail(SAMBLE_E_INVALID, "samble_bin_select_seeded_f32: need num_bins <= 8, N <= 16384");
  if (B <= 0 || N <= 0 || M < 1 || M > N)
    return fail(SAMBLE_E_INVALID, "samble_bin_select_seeded_f32: need B, N > 0 and 1 <= M <= N");
  return done(samble_launch_bin_select(score, z, member, counts, nullptr, seed, offset, B, N, nb, M, sample_mode, temp_mode,
                                       temp, (long long*)idx_out, (hipStream_t)stream),
              "samble_bin_select_seeded_f32");
}

SAMBLE_API int samble_exp1_noise_f32(uint64_t seed, uint64_t offset, int rows, int N, float* noise, void* stream) {
  if (!noise || rows <= 0 || N <= 0) return fail(SAMBLE_E_INVALID, "samble_exp1_noise_f32: null pointer or empty shape");
  if (offset & 3) return fail(SAMBLE_E_INVALID, "samble_exp1_noise_f32: the Philox offset is a multiple of 4");
  return done(samble_launch_exp1_noise(seed, offset, (long)rows * N, noise, (hipStream_t)stream), "samble_exp1_noise_f32");
}

SAMBLE_API int samble_gather_rows_f32(const float* O, int64_t o_bs, int64_t o_rs, const int64_t* idx, int B, int M, int D,
                                      float* x_ds, void* stream) {
  if (!O || !idx || !x_ds) return fail(SAMBLE_E_INVALID, "samble_gather_rows_f32: null pointer");
  if (D != 128) return fail(SAMBLE_E_INVALID, "samble_gather_rows_f32: D must be 128");
  return done(samble_launch_gather_rows(O, o_bs, o_rs, (const long long*)idx, B, M, x_ds, (hipStream_t)stream),
              "samble_gather_rows_f32");
}

SAMBLE_API int samble_gather_points_f32(const float* pcd, int B, int C, int N, const int64_t* idx, int M, float* out,
                                        void* stream) {
  if (!pcd || !idx || !out) return fail(SAMBLE_E_INVALID, "samble_gather_points_f32: null pointer");
  return done(samble_launch_gather_points(pcd, B, C, N, (const long long*)idx, M, out, (hipStream_t)stream),
              "samble_gather_points_f32");
}

SAMBLE_API int samble_edge_partial_count(void) { return samble_edge_waves(); }

SAMBLE_API int samble_edge_gather_sums_f32(const float* bp, const int32_t* nn, int B, int N, int K, int C, float* S,
                                           float* Q, void* stream) {
  if (!bp || !nn || !S || !Q) return fail(SAMBLE_E_INVALID, "samble_edge_gather_sums_f32: null pointer");
  if (K != 32 || C != 64 || B <= 0 || N <= 0)
    return fail(SAMBLE_E_INVALID, "samble_edge_gather_sums_f32: built for K = 32 neighbours, 64 channels");
  return done(samble_launch_edge_gather_sums(bp, nn, B, N, S, Q, (hipStream_t)stream), "samble_edge_gather_sums_f32");
}

SAMBLE_API int samble_edge_mlp_fwd_f32(const float* ap, const float* bp, const int32_t* nn, const float* W2, int B, int N,
                                       int K, int C, float* ymax, float* ymin, uint8_t* kmax, uint8_t* kmin,
                                       double* partials, void* stream) {
  if (!ap || !bp || !nn || !W2 || !ymax || !ymin || !kmax || !kmin || !partials)
    return fail(SAMBLE_E_INVALID, "samble_edge_mlp_fwd_f32: null pointer");
  if (K != 32 || C != 64 || B <= 0 || N <= 0)
    return fail(SAMBLE_E_INVALID, "samble_edge_mlp_fwd_f32: built for K = 32 neighbours, 64 channels");
  return done(samble_launch_edge_mlp_fwd(ap, bp, nn, W2, B, N, ymax, ymin, kmax, kmin, partials, (hipStream_t)stream),
              "samble_edge_mlp_fwd_f32");
}

SAMBLE_API int samble_edge_mlp_bwd_f32(const float* ap, const float* bp, const int32_t* nn, const float* W2,
                                       const uint8_t* kext, const float* sdv, const float* c0c1, int B, int N, int K, int C,
                                       float* du, float* dusum, float* dw2_partials, void* stream) {
  if (!ap || !bp || !nn || !W2 || !kext || !sdv || !c0c1 || !du || !dw2_partials)
    return fail(SAMBLE_E_INVALID, "samble_edge_mlp_bwd_f32: null pointer");
  if (K != 32 || C != 64 || B <= 0 || N <= 0)
    return fail(SAMBLE_E_INVALID, "samble_edge_mlp_bwd_f32: built for K = 32 neighbours, 64 channels");
  return done(samble_launch_edge_mlp_bwd(ap, bp, nn, W2, kext, sdv, c0c1, B, N, du, dusum, dw2_partials, (hipStream_t)stream),
              "samble_edge_mlp_bwd_f32");
}

SAMBLE_API size_t samble_inverse_neighbors_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  return samble_inverse_neighbors_ws_bytes(B, N);
}

SAMBLE_API int samble_inverse_neighbors(const int32_t* nn, int B, int N, int KN, int32_t* inv_order, int32_t* inv_offsets,
                                        int32_t* indegree, void* ws, size_t ws_bytes, void* stream) {
  if (!nn || !inv_order || !inv_offsets || !ws) return fail(SAMBLE_E_INVALID, "samble_inverse_neighbors: null pointer");
  if (B <= 0 || N <= 0 || N > 65535 || KN <= 0 || (long long)B * N * KN > 2147483647LL)
    return fail(SAMBLE_E_INVALID, "samble_inverse_neighbors: need 0 < N < 65536 and B N KN < 2^31");
  if (ws_bytes < samble_inverse_neighbors_ws_bytes(B, N))
    return fail(SAMBLE_E_WORKSPACE, "samble_inverse_neighbors: workspace too small");
  return done(samble_launch_inverse_neighbors(nn, B, N, KN, inv_order, inv_offsets, indegree, ws, (hipStream_t)stream),
              "samble_inverse_neighbors");
}

SAMBLE_API int samble_segment_sum_rows_f32(const float* src, int64_t src_row_stride, const int32_t* inv_order,
                                           const int32_t* inv_offsets, int K, int C, int per_edge, int64_t n_targets,
                                           float* out, void* stream) {
  if (!src || !inv_order || !inv_offsets || !out)
    return fail(SAMBLE_E_INVALID, "samble_segment_sum_rows_f32: null pointer");
  if (C != 64 || K <= 0 || n_targets <= 0) return fail(SAMBLE_E_INVALID, "samble_segment_sum_rows_f32: built for 64 channels");
  if (src_row_stride < C || (src_row_stride & 1) || ((uintptr_t)src & 7))
    return fail(SAMBLE_E_INVALID, "samble_segment_sum_rows_f32: src rows must be 8-byte aligned, row stride >= C");
  return done(samble_launch_seg_sum_rows64(src, (long)src_row_stride, inv_order, inv_offsets, K, per_edge, (long)n_targets, out,
                                           (hipStream_t)stream),
              "samble_segment_sum_rows_f32");
}

SAMBLE_API int samble_segment_sum_rows_pair_f32(const float* src_edge, int64_t edge_row_stride, const float* src_point,
                                                int64_t point_row_stride, const int32_t* inv_order, const int32_t* inv_offsets,
                                                int K, int C, int64_t n_targets, float* out_edge, float* out_point, void* stream) {
  if (!src_edge || !src_point || !inv_order || !inv_offsets || !out_edge || !out_point)
    return fail(SAMBLE_E_INVALID, "samble_segment_sum_rows_pair_f32: null pointer");
  if (C != 64 || K <= 0 || n_targets <= 0) return fail(SAMBLE_E_INVALID, "samble_segment_sum_rows_pair_f32: built for 64 channels");
  if (edge_row_stride < C || (edge_row_stride & 1) || ((uintptr_t)src_edge & 7) || point_row_stride < C || (point_row_stride & 1) ||
      ((uintptr_t)src_point & 7))
    return fail(SAMBLE_E_INVALID, "samble_segment_sum_rows_pair_f32: src rows must be 8-byte aligned, row stride >= C");
  return done(samble_launch_seg_sum_rows64_pair(src_edge, (long)edge_row_stride, src_point, (long)point_row_stride, inv_order,
                                                inv_offsets, K, (long)n_targets, out_edge, out_point, (hipStream_t)stream),
              "samble_segment_sum_rows_pair_f32");
}

SAMBLE_API int samble_group_gather_f32(const float* x, const int32_t* nn, int B, int C, int N, int K, int mode, float* out,
                                       void* stream) {
  if (!x || !nn || !out) return fail(SAMBLE_E_INVALID, "samble_group_gather_f32: null pointer");
  if (B <= 0 || C <= 0 || C > 65535 || N <= 0 || K <= 0 || mode < 0 || mode > 3)
    return fail(SAMBLE_E_INVALID, "samble_group_gather_f32: bad sizes or mode");
  return done(samble_launch_group_gather(x, nn, B, C, N, K, mode, out, (hipStream_t)stream), "samble_group_gather_f32");
}

SAMBLE_API int samble_fps_f32(const float* xyz, const int64_t* start, int B, int N, int npoint, int64_t* out,
                              void* stream) {
  if (!xyz || !start || !out) return fail(SAMBLE_E_INVALID, "samble_fps_f32: null pointer");
  if (B <= 0 || N <= 0 || N > 32768 || npoint <= 0 || npoint > N)
    return fail(SAMBLE_E_INVALID, "samble_fps_f32: need 1 <= npoint <= N <= 32768");
  return done(samble_launch_fps(xyz, (const long long*)start, B, N, npoint, (long long*)out, (hipStream_t)stream),
              "samble_fps_f32");
}

SAMBLE_API size_t samble_attn_bwd_workspace_bytes(int B, int N, int M, int D) {
  const size_t tok_part = (size_t)B * ((M + 31) / 32) * (2 * 8 * 128 + 8);  // + 8: dS column sums per token (l2)
  return ((size_t)B * M * D * 2 + (size_t)B * M * 2 + tok_part + samble_attn_bwd_slab_floats(B, N, M) + 64) *
         sizeof(float);
}

static int attn_bwd_common(const char* who, const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs,
                           int64_t k_rs, const float* V, int64_t v_bs, int64_t v_rs, const float* O, const float* Oc,
                           const float* smap, int ld, const float* lse, const int64_t* idx, const float* g, int B, int N,
                           int nt, int M, int D, float* dQ, int64_t dq_bs, int64_t dq_rs, float* dK, int64_t dk_bs,
                           int64_t dk_rs, float* dV, int64_t dv_bs, int64_t dv_rs, void* ws, size_t ws_bytes,
                           void* stream, int variant = 0, int l2 = 0, float* cs = nullptr,
                           const void* k_tr_image = nullptr, const void* v_rm_image = nullptr) {
  char msg[160];
  const int dq_token_rows = (variant & SAMBLE_BWD_DQ_TOKEN_ROWS) ? 2 : 0;
  variant &= ~SAMBLE_BWD_DQ_TOKEN_ROWS;
  if (!Q || !K || !V || (!O && !Oc) || !lse || !idx || !g || !dQ || !dK || !dV || !ws) {
    snprintf(msg, sizeof msg, "%s: null pointer", who);
    return fail(SAMBLE_E_INVALID, msg);
  }
  if (variant < 0 || variant > 2 || (variant >= 2 && !(k_tr_image && v_rm_image && smap))) {
    snprintf(msg, sizeof msg, "%s: unknown variant", who);
    return fail(SAMBLE_E_INVALID, msg);
  }
  if (D != 128 || nt < 0 || nt > 8 || B <= 0 || N <= 0 || M <= 0) {
    snprintf(msg, sizeof msg, "%s: need D == 128, 0 <= nt <= 8, positive B/N/M", who);
    return fail(SAMBLE_E_INVALID, msg);
  }
  const bool tri = k_tr_image && v_rm_image;
  const size_t base_bytes = samble_attn_bwd_workspace_bytes(B, N, M, D);
  if (ws_bytes < base_bytes + (tri ? 3 * samble_tri_image_size(B, M, 0) + samble_bwd_tri_dsmap_bytes(B, N, M) : 0)) {
    snprintf(msg, sizeof msg, "%s: workspace too small", who);
    return fail(SAMBLE_E_WORKSPACE, msg);
  }
  if ((dq_rs & 3) || (dk_rs & 3) || (dv_rs & 3) || (q_rs & 3) || (k_rs & 3) || (v_rs & 3)) {
    snprintf(msg, sizeof msg, "%s: row strides must be multiples of 4 elements", who);
    return fail(SAMBLE_E_INVALID, msg);
  }
  if (smap && (ld < samble_attn_map_ld(N, nt) || M > 14000)) {
    snprintf(msg, sizeof msg, "%s: map row stride below samble_attn_map_ld(N, nt), or M > 14000", who);
    return fail(SAMBLE_E_INVALID, msg);
  }
  hipStream_t s = (hipStream_t)stream;
  // rows of dQ that were not sampled carry no gradient: cleared by the split-bf16 preparation kernel on its way, else
  // by one strided zero-fill launch
  const int prep_clears = samble_attn_bwd_prep_clears_dq(smap, Oc, k_tr_image, v_rm_image, tri ? (char*)ws + base_bytes : nullptr);
  if (!prep_clears) {
    const long quads = (long)B * N * (D / 4);
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, dQ, (long)dq_bs,
                       (long)dq_rs, N, D / 4, quads);
  }
  float* Qs = (float*)ws;
  float* dOb = Qs + (size_t)B * M * D;
  float* lse_s = dOb + (size_t)B * M * D;
  float* delta = lse_s + (size_t)B * M;
  float* tok_part = delta + (size_t)B * M;
  float* cs_part = tok_part + (size_t)B * ((M + 31) / 32) * 2 * 8 * 128;
  float* slab = cs_part + (size_t)B * ((M + 31) / 32) * 8;
  return done(samble_launch_attn_bwd(Q, q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs, O, Oc, smap, ld, lse,
                                     (const long long*)idx, g, B, N, nt, M, inv_sqrt_d(D), Qs, dOb, lse_s, delta, tok_part,
                                     slab, dQ, dq_bs, dq_rs, dK, dk_bs, dk_rs, dV, dv_bs, dv_rs, l2, cs, cs_part,
                                     k_tr_image, v_rm_image, tri ? (char*)ws + base_bytes : nullptr, variant,
                                     (prep_clears ? 1 : 0) | dq_token_rows, s),
              who);
}

SAMBLE_API int samble_attn_bwd_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs,
                                   int64_t k_rs, const float* V, int64_t v_bs, int64_t v_rs, const float* O,
                                   const float* lse, const int64_t* idx, const float* g, int B, int N, int nt, int M,
                                   int D, float* dQ, int64_t dq_bs, int64_t dq_rs, float* dK, int64_t dk_bs,
                                   int64_t dk_rs, float* dV, int64_t dv_bs, int64_t dv_rs, int variant, void* ws,
                                   size_t ws_bytes, void* stream) {
  return attn_bwd_common("samble_attn_bwd_f32", Q, q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs, O, nullptr, nullptr, 0, lse,
                         idx, g, B, N, nt, M, D, dQ, dq_bs, dq_rs, dK, dk_bs, dk_rs, dV, dv_bs, dv_rs, ws, ws_bytes,
                         stream, variant);
}

SAMBLE_API int samble_attn_rows_bwd_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs,
                                        int64_t k_rs, const float* V, int64_t v_bs, int64_t v_rs, const float* smap,
                                        int ld, const float* lse, const float* x_ds, const int64_t* idx, const float* g,
                                        int B, int N, int nt, int M, int D, float* dQ, int64_t dq_bs, int64_t dq_rs,
                                        float* dK, int64_t dk_bs, int64_t dk_rs, float* dV, int64_t dv_bs, int64_t dv_rs,
                                        float* ds_colsum, void* ws, size_t ws_bytes, void* stream) {
  if (!smap || !x_ds) return fail(SAMBLE_E_INVALID, "samble_attn_rows_bwd_f32: null pointer");
  return attn_bwd_common("samble_attn_rows_bwd_f32", Q, q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs, nullptr, x_ds, smap, ld,
                         lse, idx, g, B, N, nt, M, D, dQ, dq_bs, dq_rs, dK, dk_bs, dk_rs, dV, dv_bs, dv_rs, ws, ws_bytes,
                         stream, 0, ds_colsum ? 1 : 0, ds_colsum);
}

SAMBLE_API size_t samble_attn_rows_bwd_tri_workspace_bytes(int B, int N, int M, int D) {
  if (B <= 0 || N <= 0 || M <= 0) return 0;
  return samble_attn_bwd_workspace_bytes(B, N, M, D) + 3 * samble_tri_image_size(B, M, 0) + samble_bwd_tri_dsmap_bytes(B, N, M);
}

SAMBLE_API int samble_attn_rows_bwd_tri_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs,
                                            int64_t k_rs, const float* V, int64_t v_bs, int64_t v_rs,
                                            const void* k_tr_image, const void* v_rm_image, const float* smap, int ld,
                                            const float* lse, const float* x_ds, const int64_t* idx, const float* g, int B,
                                            int N, int nt, int M, int D, float* dQ, int64_t dq_bs, int64_t dq_rs, float* dK,
                                            int64_t dk_bs, int64_t dk_rs, float* dV, int64_t dv_bs, int64_t dv_rs,
                                            float* ds_colsum, int variant, void* ws, size_t ws_bytes, void* stream) {
  if (!smap || !x_ds || !k_tr_image || !v_rm_image)
    return fail(SAMBLE_E_INVALID, "samble_attn_rows_bwd_tri_f32: null pointer");
  return attn_bwd_common("samble_attn_rows_bwd_tri_f32", Q, q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs, nullptr, x_ds, smap,
                         ld, lse, idx, g, B, N, nt, M, D, dQ, dq_bs, dq_rs, dK, dk_bs, dk_rs, dV, dv_bs, dv_rs, ws,
                         ws_bytes, stream, variant, ds_colsum ? 1 : 0, ds_colsum, k_tr_image, v_rm_image);
}

/* ---- multi-head global attention (Point2PointAttention) ---- */
static int heads_args_ok(const char* who, const void* Q, const void* K, const void* V, int B, int N, int H, int D,
                         int64_t s0, int64_t s1, int64_t s2, int64_t s3, int64_t s4, int64_t s5) {
  char msg[160];
  if (!Q || !K || !V) {
    snprintf(msg, sizeof msg, "%s: null pointer", who);
    return fail(SAMBLE_E_INVALID, msg);
  }
  if (B <= 0 || N <= 0 || H <= 0 || D < 4 || D > 128 || (D & 3)) {
    snprintf(msg, sizeof msg, "%s: need positive B/N/H and a head depth D in 4..128, a multiple of 4", who);
    return fail(SAMBLE_E_INVALID, msg);
  }
  if (((s0 | s1 | s2 | s3 | s4 | s5) & 3) || ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) |
                                               reinterpret_cast<uintptr_t>(V)) & 15)) {
    snprintf(msg, sizeof msg, "%s: rows move in 16-byte pieces: strides multiples of 4 elements, 16-byte aligned bases", who);
    return fail(SAMBLE_E_INVALID, msg);
  }
  return 0;
}

SAMBLE_API int samble_attn_heads_fwd_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs,
                                         int64_t k_rs, const float* V, int64_t v_bs, int64_t v_rs, const float* key_bias,
                                         float qk_mul, int B, int N, int H, int D, float* O, int64_t o_bs, int64_t o_rs,
                                         float* lse, void* stream) {
  if (int rc = heads_args_ok("samble_attn_heads_fwd_f32", Q, K, V, B, N, H, D, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs)) return rc;
  if (!O || !lse || ((o_bs | o_rs) & 3) || (reinterpret_cast<uintptr_t>(O) & 15))
    return fail(SAMBLE_E_INVALID, "samble_attn_heads_fwd_f32: O / lse missing, or O not in 16-byte pieces");
  return done(samble_launch_attn_heads_fwd(Q, q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs, key_bias, qk_mul, inv_sqrt_d(D), B, N,
                                           H, D, O, o_bs, o_rs, lse, (hipStream_t)stream),
              "samble_attn_heads_fwd_f32");
}

SAMBLE_API size_t samble_attn_heads_bwd_workspace_bytes(int B, int N, int H) {
  if (B <= 0 || N <= 0 || H <= 0) return 0;
  return (size_t)B * H * N * sizeof(float);
}

SAMBLE_API int samble_attn_heads_bwd_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs,
                                         int64_t k_rs, const float* V, int64_t v_bs, int64_t v_rs, const float* key_bias,
                                         float qk_mul, int B, int N, int H, int D, const float* O, int64_t o_bs,
                                         int64_t o_rs, const float* lse, const float* dO, int64_t g_bs, int64_t g_rs,
                                         float* dQ, int64_t dq_bs, int64_t dq_rs, float* dK, int64_t dk_bs, int64_t dk_rs,
                                         float* dV, int64_t dv_bs, int64_t dv_rs, float* bias_grad, void* ws,
                                         size_t ws_bytes, void* stream) {
  if (int rc = heads_args_ok("samble_attn_heads_bwd_f32", Q, K, V, B, N, H, D, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs)) return rc;
  if (!O || !lse || !dO || !dQ || !dK || !dV || !ws) return fail(SAMBLE_E_INVALID, "samble_attn_heads_bwd_f32: null pointer");
  if (((o_bs | o_rs | g_bs | g_rs | dq_bs | dq_rs | dk_bs | dk_rs | dv_bs | dv_rs) & 3) ||
      ((reinterpret_cast<uintptr_t>(O) | reinterpret_cast<uintptr_t>(dO) | reinterpret_cast<uintptr_t>(dQ) |
        reinterpret_cast<uintptr_t>(dK) | reinterpret_cast<uintptr_t>(dV)) & 15))
    return fail(SAMBLE_E_INVALID, "samble_attn_heads_bwd_f32: rows move in 16-byte pieces (strides, bases)");
  if (key_bias && !bias_grad) return fail(SAMBLE_E_INVALID, "samble_attn_heads_bwd_f32: key_bias without bias_grad");
  if (ws_bytes < samble_attn_heads_bwd_workspace_bytes(B, N, H))
    return fail(SAMBLE_E_WORKSPACE, "samble_attn_heads_bwd_f32: workspace too small");
  return done(samble_launch_attn_heads_bwd(Q, q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs, key_bias, qk_mul, inv_sqrt_d(D), B, N,
                                           H, D, O, o_bs, o_rs, lse, dO, g_bs, g_rs, (float*)ws, dQ, dq_bs, dq_rs, dK, dk_bs,
                                           dk_rs, dV, dv_bs, dv_rs, bias_grad, (hipStream_t)stream),
              "samble_attn_heads_bwd_f32");
}

SAMBLE_API int samble_attn_map_row_stride(int N, int nt) { return samble_attn_map_ld(N, nt); }

SAMBLE_API int samble_attn_stats_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs,
                                     int64_t k_rs, int B, int N, int nt, int D, float* smap, int ld, float* lse,
                                     float* tok, const float* q_sqnorm, const float* k_sqnorm, void* stream) {
  if ((q_sqnorm == nullptr) != (k_sqnorm == nullptr))
    return fail(SAMBLE_E_INVALID, "samble_attn_stats_f32: l2 scoring needs both squared-norm arrays");
  if (!Q || !K || !smap || !lse) return fail(SAMBLE_E_INVALID, "samble_attn_stats_f32: null pointer");
  if (D != 128) return fail(SAMBLE_E_INVALID, "samble_attn_stats_f32: D must be 128");
  if (B <= 0 || N <= 0 || nt < 0 || nt > 8 || (nt > 0 && !tok))
    return fail(SAMBLE_E_INVALID, "samble_attn_stats_f32: bad B/N/nt");
  if ((q_rs & 3) || (k_rs & 3) || (q_bs & 3) || (k_bs & 3))
    return fail(SAMBLE_E_INVALID, "samble_attn_stats_f32: strides must be multiples of 4 elements (16-byte rows)");
  if (ld < samble_attn_map_ld(N, nt) || (ld & 3) || (long)N * ld >= (1l << 31))
    return fail(SAMBLE_E_INVALID, "samble_attn_stats_f32: map row stride must be >= samble_attn_map_row_stride(N, nt), "
                                  "a multiple of 4, and N * ld < 2^31");
  return done(samble_launch_attn_stats(Q, q_bs, q_rs, K, k_bs, k_rs, B, N, nt, inv_sqrt_d(D), smap, ld, lse, tok,
                                       q_sqnorm, k_sqnorm, (hipStream_t)stream),
              "samble_attn_stats_f32");
}

SAMBLE_API size_t samble_tri_image_bytes(int B, int rows, int transposed) {
  return (B > 0 && rows > 0) ? samble_tri_image_size(B, rows, transposed) : 0;
}

SAMBLE_API int samble_tri_split_f32(const float* src, int64_t bs, int64_t rs, int B, int rows, int D, void* rm_image,
                                    void* tr_image, void* stream) {
  if (!src || (!rm_image && !tr_image)) return fail(SAMBLE_E_INVALID, "samble_tri_split_f32: null pointer");
  if (D != 128) return fail(SAMBLE_E_INVALID, "samble_tri_split_f32: D must be 128");
  if (B <= 0 || rows <= 0 || (rs & 3) || (bs & 3))
    return fail(SAMBLE_E_INVALID, "samble_tri_split_f32: bad sizes (strides must be multiples of 4 elements)");
  return done(samble_launch_tri_split(src, bs, rs, B, rows, rm_image, tr_image, (hipStream_t)stream),
              "samble_tri_split_f32");
}

SAMBLE_API int samble_tri_split_qkv_f32(const float* qkv, int64_t bs, int64_t rs, int B, int N, int nt, int D, void* q_image,
                                        void* k_image, void* v_tr_image, void* k_tr_image, void* v_rm_image,
                                        void* stream) {
  if (!qkv || !q_image || !k_image || !v_tr_image) return fail(SAMBLE_E_INVALID, "samble_tri_split_qkv_f32: null pointer");
  if (D != 128) return fail(SAMBLE_E_INVALID, "samble_tri_split_qkv_f32: D must be 128");
  if (B <= 0 || N <= 0 || nt < 0 || nt > 8 || (rs & 3) || (bs & 3))
    return fail(SAMBLE_E_INVALID, "samble_tri_split_qkv_f32: bad sizes (strides must be multiples of 4 elements)");
  return done(samble_launch_tri_split_qkv(qkv, bs, rs, B, N, nt, q_image, k_image, v_tr_image, k_tr_image, v_rm_image,
                                          (hipStream_t)stream),
              "samble_tri_split_qkv_f32");
}

SAMBLE_API int samble_tri_k_logit_form(void* k_image, int B, int rows, void* stream) {
  if (!k_image) return fail(SAMBLE_E_INVALID, "samble_tri_k_logit_form: null pointer");
  if (B <= 0 || rows <= 0) return fail(SAMBLE_E_INVALID, "samble_tri_k_logit_form: bad sizes");
  return done(samble_launch_k_to_duo(k_image, B, rows, 0, (hipStream_t)stream), "samble_tri_k_logit_form");
}

SAMBLE_API int samble_attn_stats_tri_f32(const void* q_image, const void* k_image, int B, int N, int nt, int D,
                                         float* smap, int ld, float* lse, float* tok, const float* q_sqnorm,
                                         const float* k_sqnorm, void* stream) {
  if ((q_sqnorm == nullptr) != (k_sqnorm == nullptr))
    return fail(SAMBLE_E_INVALID, "samble_attn_stats_tri_f32: l2 scoring needs both squared-norm arrays");
  if (!q_image || !k_image || !smap || !lse) return fail(SAMBLE_E_INVALID, "samble_attn_stats_tri_f32: null pointer");
  if (D != 128) return fail(SAMBLE_E_INVALID, "samble_attn_stats_tri_f32: D must be 128");
  if (B <= 0 || N <= 0 || nt < 0 || nt > 8 || (nt > 0 && !tok))
    return fail(SAMBLE_E_INVALID, "samble_attn_stats_tri_f32: bad B/N/nt");
  if (ld < samble_attn_map_ld(N, nt) || (ld & 3) || (long)N * ld >= (1l << 31))
    return fail(SAMBLE_E_INVALID, "samble_attn_stats_tri_f32: map row stride must be >= "
                                  "samble_attn_map_row_stride(N, nt), a multiple of 4, and N * ld < 2^31");
  return done(samble_launch_attn_stats_tri(q_image, k_image, B, N, nt, inv_sqrt_d(D), smap, ld, lse, tok, q_sqnorm,
                                           k_sqnorm, (hipStream_t)stream),
              "samble_attn_stats_tri_f32");
}

SAMBLE_API int samble_attn_rows_fwd_tri_f32(const float* smap, int ld, const float* lse, const void* v_tr_image,
                                            const int64_t* idx, int B, int N, int nt, int M, int D, float* x_ds,
                                            void* stream) {
  if (!smap || !lse || !v_tr_image || !idx || !x_ds)
    return fail(SAMBLE_E_INVALID, "samble_attn_rows_fwd_tri_f32: null pointer");
  if (D != 128) return fail(SAMBLE_E_INVALID, "samble_attn_rows_fwd_tri_f32: D must be 128");
  if (B <= 0 || N <= 0 || M <= 0 || nt < 0 || nt > 8)
    return fail(SAMBLE_E_INVALID, "samble_attn_rows_fwd_tri_f32: bad sizes");
  if ((ld & 3) || ld < samble_attn_map_ld(N, nt)) return fail(SAMBLE_E_INVALID, "samble_attn_rows_fwd_tri_f32: bad strides");
  return done(samble_launch_attn_rows_tri(smap, ld, lse, v_tr_image, (const long long*)idx, B, N, nt, M, x_ds,
                                          (hipStream_t)stream),
              "samble_attn_rows_fwd_tri_f32");
}

SAMBLE_API int samble_attn_rows_fwd_f32(const float* smap, int ld, const float* lse, const float* V, int64_t v_bs,
                                        int64_t v_rs, const int64_t* idx, int B, int N, int nt, int M, int D, float* x_ds,
                                        void* stream) {
  if (!smap || !lse || !V || !idx || !x_ds) return fail(SAMBLE_E_INVALID, "samble_attn_rows_fwd_f32: null pointer");
  if (D != 128) return fail(SAMBLE_E_INVALID, "samble_attn_rows_fwd_f32: D must be 128");
  if (B <= 0 || N <= 0 || M <= 0 || nt < 0 || nt > 8) return fail(SAMBLE_E_INVALID, "samble_attn_rows_fwd_f32: bad sizes");
  if ((v_rs & 3) || (v_bs & 3) || (ld & 3) || ld < samble_attn_map_ld(N, nt))
    return fail(SAMBLE_E_INVALID, "samble_attn_rows_fwd_f32: bad strides");
  return done(samble_launch_attn_rows(smap, ld, lse, V, v_bs, v_rs, (const long long*)idx, B, N, nt, M, x_ds,
                                      (hipStream_t)stream),
              "samble_attn_rows_fwd_f32");
}

SAMBLE_API size_t samble_nn_masks_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  return (size_t)B * ((N + 31) / 32) * N * sizeof(uint32_t);
}

SAMBLE_API int samble_nn_prepare(const int32_t* nn, int B, int N, int KN, int32_t* nn_sorted, uint32_t* masks,
                                 void* clear, size_t clear_bytes, void* stream) {
  if (!nn || !nn_sorted || !masks) return fail(SAMBLE_E_INVALID, "samble_nn_prepare: null pointer");
  if (B <= 0 || N <= 0 || (KN != 16 && KN != 32)) return fail(SAMBLE_E_INVALID, "samble_nn_prepare: need KN in {16, 32}");
  if (!clear) clear_bytes = 0;
  return done(samble_launch_nn_prepare(nn, B, N, KN, nn_sorted, masks, clear, clear_bytes, (hipStream_t)stream),
              "samble_nn_prepare");
}

SAMBLE_API int samble_attn_stats_nl_tri_f32(const void* q_image, const void* k_image, int B, int N, int nt, int D,
                                            const uint32_t* masks, int KN, float* nl, float* lse, float* tok,
                                            const int32_t* nn_sorted, int score_mode, void* score_ws,
                                            size_t score_ws_bytes, int score_ws_cleared, void* stream) {
  if (!q_image || !k_image || !masks || !lse || (nt > 0 && !tok) || (!nl && !score_ws))
    return fail(SAMBLE_E_INVALID, "samble_attn_stats_nl_tri_f32: null pointer");
  if (D != 128) return fail(SAMBLE_E_INVALID, "samble_attn_stats_nl_tri_f32: D must be 128");
  if (B <= 0 || N <= 0 || nt < 0 || nt > 8 || KN < 1 || KN > 32)
    return fail(SAMBLE_E_INVALID, "samble_attn_stats_nl_tri_f32: bad B/N/nt, or KN outside 1..32");
  if (score_ws) {
    if (!nn_sorted) return fail(SAMBLE_E_INVALID, "samble_attn_stats_nl_tri_f32: score accumulation needs nn_sorted");
    if (score_mode < 0 || score_mode > SAMBLE_SCORE_SPARSE_ROW_STD)
      return fail(SAMBLE_E_INVALID, "samble_attn_stats_nl_tri_f32: unknown score mode");
    if (N > 8192)
      return fail(SAMBLE_E_INVALID, "samble_attn_stats_nl_tri_f32: score accumulation needs N <= 8192 (LDS accumulators); "
                                    "take nl and run samble_sparse_score_map_f32 on it");
    if (score_ws_bytes < samble_score_ws_bytes(B, N))
      return fail(SAMBLE_E_WORKSPACE, "samble_attn_stats_nl_tri_f32: score workspace too small");
  }
  return done(samble_launch_attn_stats_nl_tri(q_image, k_image, B, N, nt, inv_sqrt_d(D), masks, KN, nl, lse, tok,
                                              score_ws ? nn_sorted : nullptr, score_mode, score_ws,
                                              score_ws_cleared ? 0 : score_ws_bytes, (hipStream_t)stream),
              "samble_attn_stats_nl_tri_f32");
}

SAMBLE_API int samble_attn_rows_fwd_recompute_tri_f32(const void* q_image, const void* k_image, const void* v_tr_image,
                                                      const float* lse, const int64_t* idx, int B, int N, int nt, int M,
                                                      int D, float* x_ds, float* pmap, int ld, void* stream) {
  if (!q_image || !k_image || !v_tr_image || !lse || !idx || !x_ds)
    return fail(SAMBLE_E_INVALID, "samble_attn_rows_fwd_recompute_tri_f32: null pointer");
  if (D != 128) return fail(SAMBLE_E_INVALID, "samble_attn_rows_fwd_recompute_tri_f32: D must be 128");
  if (B <= 0 || N <= 0 || M <= 0 || nt < 0 || nt > 8)
    return fail(SAMBLE_E_INVALID, "samble_attn_rows_fwd_recompute_tri_f32: bad sizes");
  if (pmap && ((ld & 3) || ld < samble_attn_map_ld(N, nt)))
    return fail(SAMBLE_E_INVALID, "samble_attn_rows_fwd_recompute_tri_f32: bad P map row stride");
  return done(samble_launch_attn_rows_rc_tri(q_image, k_image, v_tr_image, lse, (const long long*)idx, B, N, nt, M,
                                             inv_sqrt_d(D), x_ds, pmap, ld, (hipStream_t)stream),
              "samble_attn_rows_fwd_recompute_tri_f32");
}

SAMBLE_API int samble_sparse_score_map_f32(const float* smap, int ld, const float* lse, const int32_t* nn, int B, int N,
                                           int KN, int mode, float* score, float* z, int32_t* indeg_out, void* ws,
                                           size_t ws_bytes, void* stream) {
  if ((smap && (!lse || !nn)) || !score || !z || !ws)
    return fail(SAMBLE_E_INVALID, "samble_sparse_score_map_f32: null pointer");
  if (mode < 0 || mode > SAMBLE_SCORE_SPARSE_ROW_STD)
    return fail(SAMBLE_E_INVALID, "samble_sparse_score_map_f32: unknown score mode");
  if (ws_bytes < samble_score_ws_bytes(B, N))
    return fail(SAMBLE_E_WORKSPACE, "samble_sparse_score_map_f32: workspace too small");
  return done(samble_launch_sparse_score_map(smap, ld, lse, nn, B, N, KN, mode, score, z, indeg_out, ws,
                                             (hipStream_t)stream),
              "samble_sparse_score_map_f32");
}

// ---- the fused select chain (chain.hip): score + z + batch quantiles in one launch, boundaries + bins + counts in another
SAMBLE_API int samble_select_chain_supported(int B, int N, int nb) { return samble_chain_supported(B, N, nb); }

static size_t chain_score_bytes(int B, int N) { return (samble_score_ws_bytes(B, N) + 255) & ~(size_t)255; }

SAMBLE_API size_t samble_select_chain_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  return chain_score_bytes(B, N) + samble_chain_ws_bytes();
}

SAMBLE_API int samble_sparse_score_map_quantiles_f32(const float* smap, int ld, const float* lse, const int32_t* nn, int B,
                                                     int N, int KN, int mode, int nb, float* score, float* z,
                                                     int32_t* indeg_out, float* quantiles_out, void* ws, size_t ws_bytes,
                                                     unsigned int spin_budget, int32_t* host_status, void* stream) {
  if ((smap && (!lse || !nn)) || !score || !z || !ws)
    return fail(SAMBLE_E_INVALID, "samble_sparse_score_map_quantiles_f32: null pointer");
  if (mode < 0 || mode > SAMBLE_SCORE_SPARSE_ROW_STD)
    return fail(SAMBLE_E_INVALID, "samble_sparse_score_map_quantiles_f32: unknown score mode");
  if ((size_t)N * 12 > 150 * 1024) return fail(SAMBLE_E_INVALID, "samble_sparse_score_map_quantiles_f32: N too large for LDS");
  if (!samble_chain_supported(B, N, nb))
    return fail(SAMBLE_E_INVALID, "samble_sparse_score_map_quantiles_f32: shape not taken by the fused chain "
                                  "(samble_select_chain_supported)");
  if (ws_bytes < samble_select_chain_workspace_bytes(B, N))
    return fail(SAMBLE_E_WORKSPACE, "samble_sparse_score_map_quantiles_f32: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  if (smap) {  // NULL: samble_attn_stats_nl_tri_f32 accumulated into (and zeroed all of) this workspace already
    int rc = samble_launch_sparse_score_map_acc(smap, ld, lse, nn, B, N, KN, mode, ws, samble_select_chain_workspace_bytes(B, N), s);
    if (rc) return done(rc, "samble_sparse_score_map_quantiles_f32");
  }
  char* w8 = (char*)ws;
  const void* colacc = w8;
  const int* indeg = (const int*)(w8 + (size_t)B * N * 8);
  const float* rowstat = (const float*)(w8 + (size_t)B * N * 12);
  return done(samble_launch_score_quantiles(colacc, indeg, rowstat, B, N, mode, nb, score, z, indeg_out,
                                            w8 + chain_score_bytes(B, N), quantiles_out, spin_budget, host_status, s),
              "samble_sparse_score_map_quantiles_f32");
}

SAMBLE_API int samble_bin_plan_f32(const float* z, const float* tok, int nt, const float* quantiles,
                                   const float* quantile_divisor, float* upper, float* lower, int first, float momentum, float one_minus_momentum, int B, int N, int nb,
                                   int relu_first, int M, uint8_t* member, int32_t* cap, float* w_pre, float* w,
                                   int32_t* counts, void* ws, size_t ws_bytes, unsigned int spin_budget, int32_t* host_status, void* stream) {
  if (!z || !tok || !upper || !lower || !member || !cap || !w_pre || !w || !counts || !ws)
    return fail(SAMBLE_E_INVALID, "samble_bin_plan_f32: null pointer");
  if (nb < 1 || nb > 8 || (nt != 1 && nt != nb))
    return fail(SAMBLE_E_INVALID, "samble_bin_plan_f32: need 1 <= num_bins <= 8 and nt in {1, num_bins}");
  if (M < 1 || M > N) return fail(SAMBLE_E_INVALID, "samble_bin_plan_f32: need 1 <= M <= N");
  if (!samble_chain_supported(B, N, nb < 2 ? 2 : nb))
    return fail(SAMBLE_E_INVALID, "samble_bin_plan_f32: shape not taken by the fused chain (samble_select_chain_supported)");
  if (ws_bytes < samble_select_chain_workspace_bytes(B, N))
    return fail(SAMBLE_E_WORKSPACE, "samble_bin_plan_f32: workspace too small");
  return done(samble_launch_bin_plan(z, tok, nt, quantiles, quantiles ? quantile_divisor : nullptr, upper, lower, first, momentum, one_minus_momentum, B, N, nb,
                                     relu_first, M, member, cap, w_pre, w, counts, (char*)ws + chain_score_bytes(B, N),
                                     spin_budget, host_status, (hipStream_t)stream),
              "samble_bin_plan_f32");
}

SAMBLE_API int samble_select_chain_f32(const float* smap, int ld, const float* lse, const int32_t* nn, int KN, int mode,
                                       const float* tok, int nt, int want_quantiles, float* quantiles_out, float* upper,
                                       float* lower, int first, float momentum, float one_minus_momentum, int B, int N,
                                       int nb, int relu_first, int M, float* score, float* z, int32_t* indeg_out,
                                       uint8_t* member, int32_t* cap, float* w_pre, float* w, int32_t* counts, void* ws,
                                       size_t ws_bytes, unsigned int spin_budget, int32_t* host_status, void* stream) {
  if ((smap && (!lse || !nn)) || !score || !z || !ws || !tok || !upper || !lower || !member || !cap || !w_pre || !w ||
      !counts || (want_quantiles && !quantiles_out))
    return fail(SAMBLE_E_INVALID, "samble_select_chain_f32: null pointer");
  if (mode < 0 || mode > SAMBLE_SCORE_SPARSE_ROW_STD) return fail(SAMBLE_E_INVALID, "samble_select_chain_f32: unknown score mode");
  if (nb < 2 || nb > 8 || (nt != 1 && nt != nb))
    return fail(SAMBLE_E_INVALID, "samble_select_chain_f32: need 2 <= num_bins <= 8 and nt in {1, num_bins}");
  if ((size_t)N * 12 > 150 * 1024) return fail(SAMBLE_E_INVALID, "samble_select_chain_f32: N too large for LDS");
  if (M < 1 || M > N) return fail(SAMBLE_E_INVALID, "samble_select_chain_f32: need 1 <= M <= N");
  if (!samble_chain_supported(B, N, nb))
    return fail(SAMBLE_E_INVALID, "samble_select_chain_f32: shape not taken by the fused chain (samble_select_chain_supported)");
  if (ws_bytes < samble_select_chain_workspace_bytes(B, N))
    return fail(SAMBLE_E_WORKSPACE, "samble_select_chain_f32: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  if (smap) {
    int rc = samble_launch_sparse_score_map_acc(smap, ld, lse, nn, B, N, KN, mode, ws, samble_select_chain_workspace_bytes(B, N), s);
    if (rc) return done(rc, "samble_select_chain_f32");
  }
  char* w8 = (char*)ws;
  return done(samble_launch_select_chain(w8, (const int*)(w8 + (size_t)B * N * 8), (const float*)(w8 + (size_t)B * N * 12), B,
                                         N, mode, nb, score, z, indeg_out, w8 + chain_score_bytes(B, N),
                                         want_quantiles ? quantiles_out : nullptr, tok, nt, upper, lower, first, momentum,
                                         one_minus_momentum, relu_first, M, member, cap, w_pre, w, counts, spin_budget, host_status, s),
              "samble_select_chain_f32");
}

SAMBLE_API int samble_select_chain_status_async(const void* ws, int B, int N, int32_t* host_flag, void* stream) {
  if (!ws || !host_flag || B <= 0 || N <= 0) return fail(SAMBLE_E_INVALID, "samble_select_chain_status_async: bad argument");
  const char* word = (const char*)ws + chain_score_bytes(B, N) + samble_chain_flag_offset();
  return done((int)hipMemcpyAsync(host_flag, word, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream),
              "samble_select_chain_status_async");
}

/* ---- the closed forms around the fused EdgeConv's two MLP sweeps (csrc/edge_glue.hip) ------------------------------ */
SAMBLE_API size_t samble_edge_glue_partials_bytes(void) { return samble_edge_glue_part_bytes(); }
SAMBLE_API size_t samble_edge_glue_constants_bytes(void) { return samble_edge_glue_cst_bytes(); }
SAMBLE_API size_t samble_edge_glue_statistics_bytes(void) { return samble_edge_glue_st_bytes(); }
SAMBLE_API size_t samble_edge_glue_pooled_bytes(void) { return samble_edge_glue_pool_bytes(); }
static int edge_phase_ok(int phase, const double* pooled) { return phase == 0 || ((phase == 1 || phase == 2) && pooled); }

static int edge_shape_ok(int B, int N, int K, int C) { return B > 0 && N > 0 && K == 32 && C == 64; }

SAMBLE_API int samble_edge_bn1_f32(const float* a, const float* b, int64_t ab_row_stride, const int32_t* nn, int B, int N, int K, int C,
                                   const float* gamma1, const float* beta1, float eps, float* running_mean,
                                   float* running_var, float momentum, int64_t* num_batches_tracked, float* S, float* Q,
                                   float* ap, float* bp,
                                   float* constants, double* statistics, double* partials, int phase, double* pooled,
                                   void* stream) {
  if (!a || !b || !nn || !gamma1 || !beta1 || !S || !Q || !ap || !bp || !constants || !statistics || !partials)
    return fail(SAMBLE_E_INVALID, "samble_edge_bn1_f32: null pointer");
  if (!edge_phase_ok(phase, pooled)) return fail(SAMBLE_E_INVALID, "samble_edge_bn1_f32: phase 0, or 1 / 2 with a pooled block");
  if (!edge_shape_ok(B, N, K, C)) return fail(SAMBLE_E_INVALID, "samble_edge_bn1_f32: built for K = 32 neighbours, 64 channels");
  if (ab_row_stride < C || (ab_row_stride & 3) || ((uintptr_t)a & 15) || ((uintptr_t)b & 15))
    return fail(SAMBLE_E_INVALID, "samble_edge_bn1_f32: a / b rows must be 16-byte aligned, row stride >= C");
  if ((running_mean == nullptr) != (running_var == nullptr)) return fail(SAMBLE_E_INVALID, "samble_edge_bn1_f32: running statistics come as a pair");
  return done(samble_launch_edge_pre(a, b, (long)ab_row_stride, nn, B, N, gamma1, beta1, eps, running_mean, running_var, momentum,
                                     (long long*)num_batches_tracked, S, Q, ap, bp,
                                     constants, statistics, partials, phase, pooled, (hipStream_t)stream),
              "samble_edge_bn1_f32");
}

SAMBLE_API int samble_edge_bn2_out_f32(const float* ymax, const float* ymin, const uint8_t* kmax, const uint8_t* kmin,
                                       const double* mlp_partials, int n_partials, int B, int N, int C, const float* gamma2,
                                       const float* beta2, float eps, float* running_mean, float* running_var,
                                       float momentum, int64_t* num_batches_tracked, float* constants, double* statistics,
                                       float* ext, uint8_t* kext, float* out, int phase, double* pooled, void* stream) {
  if (!ymax || !ymin || !kmax || !kmin || !mlp_partials || !gamma2 || !beta2 || !constants || !statistics || !ext || !kext || !out)
    return fail(SAMBLE_E_INVALID, "samble_edge_bn2_out_f32: null pointer");
  if (!edge_phase_ok(phase, pooled)) return fail(SAMBLE_E_INVALID, "samble_edge_bn2_out_f32: phase 0, or 1 / 2 with a pooled block");
  if (!edge_shape_ok(B, N, 32, C) || n_partials <= 0) return fail(SAMBLE_E_INVALID, "samble_edge_bn2_out_f32: built for 64 channels");
  if ((running_mean == nullptr) != (running_var == nullptr)) return fail(SAMBLE_E_INVALID, "samble_edge_bn2_out_f32: running statistics come as a pair");
  return done(samble_launch_edge_post(ymax, ymin, kmax, kmin, mlp_partials, n_partials, B, N, gamma2, beta2, eps, running_mean,
                                      running_var, momentum, (long long*)num_batches_tracked, constants, statistics, ext, kext,
                                      out, phase, pooled, (hipStream_t)stream),
              "samble_edge_bn2_out_f32");
}

SAMBLE_API int samble_edge_bwd_pre_f32(const float* g, const float* ext, int B, int N, int C, const float* gamma2,
                                       float* constants, const double* statistics, float* sdv, float* dgamma2, float* dbeta2,
                                       double* partials, int phase, double* pooled, void* stream) {
  if (!g || !ext || !gamma2 || !constants || !statistics || !sdv || !dgamma2 || !dbeta2 || !partials)
    return fail(SAMBLE_E_INVALID, "samble_edge_bwd_pre_f32: null pointer");
  if (!edge_phase_ok(phase, pooled)) return fail(SAMBLE_E_INVALID, "samble_edge_bwd_pre_f32: phase 0, or 1 / 2 with a pooled block");
  if (!edge_shape_ok(B, N, 32, C)) return fail(SAMBLE_E_INVALID, "samble_edge_bwd_pre_f32: built for 64 channels");
  return done(samble_launch_edge_bwd_pre(g, ext, B, N, gamma2, constants, statistics, sdv, dgamma2, dbeta2, partials,
                                         phase, pooled, (hipStream_t)stream),
              "samble_edge_bwd_pre_f32");
}

SAMBLE_API int samble_edge_bwd_post_f32(const float* a, const float* b, int64_t ab_row_stride, const float* S, const float* R,
                                        const float* dusum, const float* D, const int32_t* indeg, int B, int N, int K, int C,
                                        float* constants, const double* statistics, const float* dw2_partials,
                                        int n_partials, float* da, float* db, int64_t dab_row_stride, float* dgamma1,
                                        float* dbeta1, float* dW2, double* partials, int phase, double* pooled, void* stream) {
  if (!a || !b || !S || !R || !dusum || !D || !indeg || !constants || !statistics || !dw2_partials || !da || !db || !dgamma1 ||
      !dbeta1 || !dW2 || !partials)
    return fail(SAMBLE_E_INVALID, "samble_edge_bwd_post_f32: null pointer");
  if (!edge_phase_ok(phase, pooled)) return fail(SAMBLE_E_INVALID, "samble_edge_bwd_post_f32: phase 0, or 1 / 2 with a pooled block");
  if (!edge_shape_ok(B, N, K, C) || n_partials <= 0) return fail(SAMBLE_E_INVALID, "samble_edge_bwd_post_f32: built for K = 32 neighbours, 64 channels");
  if (ab_row_stride < C || (ab_row_stride & 1) || dab_row_stride < C || (dab_row_stride & 1) || ((uintptr_t)a & 7) ||
      ((uintptr_t)b & 7) || ((uintptr_t)da & 7) || ((uintptr_t)db & 7))
    return fail(SAMBLE_E_INVALID, "samble_edge_bwd_post_f32: rows must be 8-byte aligned, row strides >= C");
  return done(samble_launch_edge_bwd_post(a, b, (long)ab_row_stride, S, R, dusum, D, indeg, B, N, constants, statistics,
                                          dw2_partials, n_partials, da, db, (long)dab_row_stride, dgamma1, dbeta1, dW2,
                                          partials, phase, pooled, (hipStream_t)stream),
              "samble_edge_bwd_post_f32");
}

/* ---- inverse-distance interpolation (csrc/interp.hip) --------------------------------------------------------------- */
SAMBLE_API int samble_interp_blend_fwd_f32(const float* feat, int B, int C, int M, const int32_t* idx, const float* dist,
                                           int N, int K, float* weights, float* out, void* stream) {
  if (!feat || !idx || !dist || !weights || !out) return fail(SAMBLE_E_INVALID, "samble_interp_blend_fwd_f32: null pointer");
  if (B <= 0 || C <= 0 || M <= 0 || N <= 0 || K < 1 || K > 8) return fail(SAMBLE_E_INVALID, "samble_interp_blend_fwd_f32: need 1 <= K <= 8");
  return done(samble_launch_interp_fwd(feat, B, C, M, idx, dist, N, K, weights, out, (hipStream_t)stream), "samble_interp_blend_fwd_f32");
}

SAMBLE_API size_t samble_interp_blend_bwd_workspace_bytes(int B, int C, int N, int M) {
  return (B > 0 && C > 0 && N > 0 && M > 0) ? samble_interp_bwd_ws_bytes(B, C, N, M) : 0;
}

SAMBLE_API int samble_interp_blend_bwd_f32(const float* g, int B, int C, int N, const float* weights, const int32_t* inv_order,
                                           const int32_t* inv_offsets, int K, int M, float* dfeat, void* ws, size_t ws_bytes,
                                           void* stream) {
  if (!g || !weights || !inv_order || !inv_offsets || !dfeat || !ws) return fail(SAMBLE_E_INVALID, "samble_interp_blend_bwd_f32: null pointer");
  if (B <= 0 || C <= 0 || M <= 0 || N <= 0 || M > N || K < 1 || K > 8)
    return fail(SAMBLE_E_INVALID, "samble_interp_blend_bwd_f32: need 1 <= K <= 8 and M <= N");
  if (ws_bytes < samble_interp_bwd_ws_bytes(B, C, N, M)) return fail(SAMBLE_E_WORKSPACE, "samble_interp_blend_bwd_f32: workspace too small");
  return done(samble_launch_interp_bwd(g, B, C, N, weights, inv_order, inv_offsets, K, M, dfeat, ws, (hipStream_t)stream),
              "samble_interp_blend_bwd_f32");
}

/* ---- BatchNorm1d in training mode, forward and backward (csrc/batchnorm.hip) ------------------------------------------ */
extern "C" {
size_t samble_bn_train_ws_bytes(int, int);
int samble_launch_bn_train_fwd(const float*, int, int, int, const float*, const float*, float, float, float*, float*, float*,
                               float*, float*, float, void*, hipStream_t);
int samble_launch_bn_train_stats(const float*, int, int, int, double*, void*, hipStream_t);
int samble_launch_bn_train_apply(const float*, int, int, int, const double*, const float*, const float*, float, float, float*,
                                 float*, float*, float*, float*, float, hipStream_t);
int samble_launch_bn_train_bwd(const float*, const float*, int, int, int, const float*, const float*, const float*, float*,
                               float*, float*, const float*, float, void*, hipStream_t);
int samble_launch_bn_train_bwd_sums(const float*, const float*, int, int, int, const float*, const float*, double*, float*,
                                    float*, const float*, const float*, float, void*, hipStream_t);
int samble_launch_bn_train_bwd_apply(const float*, const float*, int, int, int, const float*, const float*, const float*,
                                     const double*, const double*, float*, const float*, float, hipStream_t);
}
SAMBLE_API size_t samble_bn_train_workspace_bytes(int B, int C) { return (B > 0 && C > 0) ? samble_bn_train_ws_bytes(B, C) : 0; }

static int bn_shape_ok(int B, int C, int N) { return B > 0 && B <= 65535 && C > 0 && C <= 65535 && N > 0; }

SAMBLE_API int samble_bn_train_fwd_f32(const float* x, int B, int C, int N, const float* gamma, const float* beta, float eps,
                                       float momentum, float* running_mean, float* running_var, float* out, float* save_mean,
                                       float* save_invstd, float act_slope, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !out || !save_mean || !save_invstd || !ws) return fail(SAMBLE_E_INVALID, "samble_bn_train_fwd_f32: null pointer");
  if (!bn_shape_ok(B, C, N) || (long)B * N < 2)
    return fail(SAMBLE_E_INVALID, "samble_bn_train_fwd_f32: needs more than one value per channel, B <= 65535");
  if (ws_bytes < samble_bn_train_ws_bytes(B, C)) return fail(SAMBLE_E_WORKSPACE, "samble_bn_train_fwd_f32: workspace too small");
  return done(samble_launch_bn_train_fwd(x, B, C, N, gamma, beta, eps, momentum, running_mean, running_var, out, save_mean,
                                         save_invstd, act_slope, ws, (hipStream_t)stream),
              "samble_bn_train_fwd_f32");
}

SAMBLE_API int samble_bn_train_stats_f32(const float* x, int B, int C, int N, double* pooled, void* ws, size_t ws_bytes,
                                         void* stream) {
  if (!x || !pooled || !ws) return fail(SAMBLE_E_INVALID, "samble_bn_train_stats_f32: null pointer");
  if (!bn_shape_ok(B, C, N)) return fail(SAMBLE_E_INVALID, "samble_bn_train_stats_f32: bad shape (B, C <= 65535)");
  if (ws_bytes < samble_bn_train_ws_bytes(B, C)) return fail(SAMBLE_E_WORKSPACE, "samble_bn_train_stats_f32: workspace too small");
  return done(samble_launch_bn_train_stats(x, B, C, N, pooled, ws, (hipStream_t)stream), "samble_bn_train_stats_f32");
}

SAMBLE_API int samble_bn_train_apply_f32(const float* x, int B, int C, int N, const double* pooled, const float* gamma,
                                         const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                         float* out, float* save_mean, float* save_invstd, float act_slope, void* stream) {
  if (!x || !pooled || !out || !save_mean || !save_invstd) return fail(SAMBLE_E_INVALID, "samble_bn_train_apply_f32: null pointer");
  if (!bn_shape_ok(B, C, N)) return fail(SAMBLE_E_INVALID, "samble_bn_train_apply_f32: bad shape (B, C <= 65535)");
  return done(samble_launch_bn_train_apply(x, B, C, N, pooled, gamma, beta, eps, momentum, running_mean, running_var, out,
                                           save_mean, save_invstd, act_slope, (hipStream_t)stream),
              "samble_bn_train_apply_f32");
}

SAMBLE_API int samble_bn_train_bwd_f32(const float* x, const float* dy, int B, int C, int N, const float* save_mean,
                                       const float* save_invstd, const float* gamma, float* dx, float* dgamma, float* dbeta,
                                       const float* beta, float act_slope, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !dy || !save_mean || !save_invstd || !dx || !ws) return fail(SAMBLE_E_INVALID, "samble_bn_train_bwd_f32: null pointer");
  if (!bn_shape_ok(B, C, N)) return fail(SAMBLE_E_INVALID, "samble_bn_train_bwd_f32: bad shape (B, C <= 65535)");
  if (ws_bytes < samble_bn_train_ws_bytes(B, C)) return fail(SAMBLE_E_WORKSPACE, "samble_bn_train_bwd_f32: workspace too small");
  return done(samble_launch_bn_train_bwd(x, dy, B, C, N, save_mean, save_invstd, gamma, dx, dgamma, dbeta, beta, act_slope, ws,
                                         (hipStream_t)stream),
              "samble_bn_train_bwd_f32");
}

SAMBLE_API int samble_bn_train_bwd_sums_f32(const float* x, const float* dy, int B, int C, int N, const float* save_mean,
                                            const float* save_invstd, double* pooled, float* dgamma, float* dbeta,
                                            const float* gamma, const float* beta, float act_slope, void* ws,
                                            size_t ws_bytes, void* stream) {
  if (!x || !dy || !save_mean || !save_invstd || !pooled || !ws)
    return fail(SAMBLE_E_INVALID, "samble_bn_train_bwd_sums_f32: null pointer");
  if (!bn_shape_ok(B, C, N)) return fail(SAMBLE_E_INVALID, "samble_bn_train_bwd_sums_f32: bad shape (B, C <= 65535)");
  if (ws_bytes < samble_bn_train_ws_bytes(B, C)) return fail(SAMBLE_E_WORKSPACE, "samble_bn_train_bwd_sums_f32: workspace too small");
  return done(samble_launch_bn_train_bwd_sums(x, dy, B, C, N, save_mean, save_invstd, pooled, dgamma, dbeta, gamma, beta, act_slope,
                                              ws, (hipStream_t)stream),
              "samble_bn_train_bwd_sums_f32");
}

SAMBLE_API int samble_bn_train_bwd_apply_f32(const float* x, const float* dy, int B, int C, int N, const float* save_mean,
                                             const float* save_invstd, const float* gamma, const double* pooled,
                                             const double* count, float* dx, const float* beta, float act_slope, void* stream) {
  if (!x || !dy || !save_mean || !save_invstd || !pooled || !count || !dx)
    return fail(SAMBLE_E_INVALID, "samble_bn_train_bwd_apply_f32: null pointer");
  if (!bn_shape_ok(B, C, N)) return fail(SAMBLE_E_INVALID, "samble_bn_train_bwd_apply_f32: bad shape (B, C <= 65535)");
  return done(samble_launch_bn_train_bwd_apply(x, dy, B, C, N, save_mean, save_invstd, gamma, pooled, count, dx, beta, act_slope,
                                               (hipStream_t)stream),
              "samble_bn_train_bwd_apply_f32");
}

/* ---- 1x1 convolutions over 128 input channels (csrc/linear.hip) ------------------------------------------------- */
static int lin_shape_ok(int B, int N, int O) { return B > 0 && N > 0 && O >= 32 && O <= 4096 && (O & 31) == 0; }

SAMBLE_API size_t samble_linear_image_bytes(int O) { return O > 0 ? samble_linear_image_bytes_impl(O) : 0; }

SAMBLE_API int samble_linear_weight_images_f32(const float* W, int O, int C, void* rm_image, void* tr_image, void* stream) {
  if (!W || (!rm_image && !tr_image)) return fail(SAMBLE_E_INVALID, "samble_linear_weight_images_f32: null pointer");
  if (C != 128 || !lin_shape_ok(1, 1, O)) return fail(SAMBLE_E_INVALID, "samble_linear_weight_images_f32: W must be (O, 128) (narrower layers: zero columns), O a multiple of 32");
  return done(samble_launch_linear_images(W, O, rm_image, tr_image, 0, (hipStream_t)stream), "samble_linear_weight_images_f32");
}

/* 1: csrc/linear.hip was built on two fp16 planes (the default) -- the transposed-weight entry below exists in that build */
SAMBLE_API int samble_linear_two_plane_build(void) { return samble_linear_is_duo(); }

SAMBLE_API int samble_linear_weight_images_t_f32(const float* Wt, int O, int C, void* rm_image, void* tr_image, void* stream) {
  if (!Wt || (!rm_image && !tr_image)) return fail(SAMBLE_E_INVALID, "samble_linear_weight_images_t_f32: null pointer");
  if (C != 128 || !lin_shape_ok(1, 1, O)) return fail(SAMBLE_E_INVALID, "samble_linear_weight_images_t_f32: Wt must be (128, O), O a multiple of 32");
  if (!samble_linear_is_duo()) return fail(SAMBLE_E_INVALID, "samble_linear_weight_images_t_f32: needs the two-plane build of csrc/linear.hip");
  return done(samble_launch_linear_images(Wt, O, rm_image, tr_image, 1, (hipStream_t)stream), "samble_linear_weight_images_t_f32");
}

/* the two weights of a feed-forward layer in one launch: W1 (O1, 128) row-major, W2t (128, O2) row-major (images of its
   transpose), each with its row image and / or transposed image */
SAMBLE_API int samble_linear_weight_images_pair_f32(const float* W1, int O1, void* rm1, void* tr1, const float* W2t, int O2,
                                                    void* rm2, void* tr2, void* stream) {
  if (!W1 || !W2t || (!rm1 && !tr1) || (!rm2 && !tr2)) return fail(SAMBLE_E_INVALID, "samble_linear_weight_images_pair_f32: null pointer");
  if (!lin_shape_ok(1, 1, O1) || !lin_shape_ok(1, 1, O2)) return fail(SAMBLE_E_INVALID, "samble_linear_weight_images_pair_f32: O1, O2 multiples of 32");
  if (!samble_linear_is_duo()) return fail(SAMBLE_E_INVALID, "samble_linear_weight_images_pair_f32: needs the two-plane build of csrc/linear.hip");
  return done(samble_launch_linear_images_pair(W1, O1, rm1, tr1, W2t, O2, rm2, tr2, (hipStream_t)stream),
              "samble_linear_weight_images_pair_f32");
}

SAMBLE_API int samble_linear_fwd_tri_f32(const float* x, int64_t x_bs, int B, int C, int N, const void* w_rm_image, int O,
                                         int epilogue, const float* ref, float* out, int64_t o_bs, int64_t o_rs,
                                         void* stream) {
  if (!x || !w_rm_image || !out) return fail(SAMBLE_E_INVALID, "samble_linear_fwd_tri_f32: null pointer");
  if (C < 1 || C > 128 || !lin_shape_ok(B, N, O)) return fail(SAMBLE_E_INVALID, "samble_linear_fwd_tri_f32: 1 <= C <= 128, O a multiple of 32");
  const bool bits = epilogue == SAMBLE_LIN_LEAKY_BITS || epilogue == SAMBLE_LIN_LEAKY_MASK_BITS;
  if (!bits && (epilogue < SAMBLE_LIN_PLAIN || epilogue > SAMBLE_LIN_LEAKY_MASK))
    return fail(SAMBLE_E_INVALID, "samble_linear_fwd_tri_f32: unknown epilogue");
  if ((bits || epilogue == SAMBLE_LIN_LEAKY_MASK) && !ref)
    return fail(SAMBLE_E_INVALID, "samble_linear_fwd_tri_f32: the mask / sign-bit epilogues need ref");
  if (bits && ((uintptr_t)ref & 1)) return fail(SAMBLE_E_INVALID, "samble_linear_fwd_tri_f32: sign words must be 2-byte aligned");
  if ((o_rs & 3) || (o_bs & 3) || o_rs < O) return fail(SAMBLE_E_INVALID, "samble_linear_fwd_tri_f32: output strides must be multiples of 4");
  return done(samble_launch_linear_fwd(x, x_bs, B, C, N, w_rm_image, O, epilogue, ref, out, o_bs, o_rs, (hipStream_t)stream),
              "samble_linear_fwd_tri_f32");
}

/* the feed-forward layer's two convolutions in one sweep (csrc/linear.hip lin_chain): see include/samble.h */
SAMBLE_API int samble_linear_chain_f32(const float* x, int64_t x_bs, int B, int N, const void* wa_rm_image, const void* wb_tr_image,
                                       int H, int epilogue, float* mid, int64_t mid_bs, int64_t mid_rs, void* sign_words,
                                       float* out, int64_t out_bs, const float* residual, void* stream) {
  if (!x || !wa_rm_image || !wb_tr_image || !sign_words || !out) return fail(SAMBLE_E_INVALID, "samble_linear_chain_f32: null pointer");
  if (!lin_shape_ok(B, N, H)) return fail(SAMBLE_E_INVALID, "samble_linear_chain_f32: H a multiple of 32, at most 4096");
  if (epilogue != SAMBLE_LIN_LEAKY_BITS && epilogue != SAMBLE_LIN_LEAKY_MASK_BITS)
    return fail(SAMBLE_E_INVALID, "samble_linear_chain_f32: epilogue must be SAMBLE_LIN_LEAKY_BITS or SAMBLE_LIN_LEAKY_MASK_BITS");
  if (mid && ((mid_rs & 3) || (mid_bs & 3) || mid_rs < H || ((uintptr_t)mid & 15)))
    return fail(SAMBLE_E_INVALID, "samble_linear_chain_f32: mid rows must be 16-byte aligned");
  if ((uintptr_t)sign_words & 1) return fail(SAMBLE_E_INVALID, "samble_linear_chain_f32: sign words must be 2-byte aligned");
  if (!samble_linear_is_duo()) return fail(SAMBLE_E_INVALID, "samble_linear_chain_f32: needs the two-plane build of csrc/linear.hip");
  return done(samble_launch_linear_chain(x, x_bs, B, N, wa_rm_image, wb_tr_image, H, epilogue, mid, mid_bs, mid_rs, sign_words, out,
                                         out_bs, residual, (hipStream_t)stream),
              "samble_linear_chain_f32");
}

SAMBLE_API size_t samble_linear_sign_bytes(int B, int N, int O) {
  return lin_shape_ok(B, N, O) ? (size_t)B * (O / 32) * 2 * (size_t)N * sizeof(uint16_t) : 0;
}

SAMBLE_API size_t samble_linear_amax_workspace_bytes(int B, int N, int O) {
  return lin_shape_ok(B, N, O) ? samble_linear_amax_ws_bytes(B, N, O) : 0;
}

SAMBLE_API int samble_linear_amax_fwd_tri_f32(const float* x, int64_t x_bs, int B, int C, int N, const void* w_rm_image, int O,
                                              float* y, int32_t* arg, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !w_rm_image || !y || !arg || !ws) return fail(SAMBLE_E_INVALID, "samble_linear_amax_fwd_tri_f32: null pointer");
  if (C != 128 || !lin_shape_ok(B, N, O)) return fail(SAMBLE_E_INVALID, "samble_linear_amax_fwd_tri_f32: C must be 128, O a multiple of 32");
  if (ws_bytes < samble_linear_amax_ws_bytes(B, N, O)) return fail(SAMBLE_E_WORKSPACE, "samble_linear_amax_fwd_tri_f32: workspace too small");
  return done(samble_launch_linear_amax(x, x_bs, B, N, w_rm_image, O, y, arg, ws, (hipStream_t)stream),
              "samble_linear_amax_fwd_tri_f32");
}

SAMBLE_API int samble_linear_dx_tri_f32(const float* g, int64_t g_bs, int64_t g_rs, const void* w_tr_image, int O, int B, int C,
                                        int N, float* dx, int64_t dx_bs, const float* residual, void* stream) {
  if (!g || !w_tr_image || !dx) return fail(SAMBLE_E_INVALID, "samble_linear_dx_tri_f32: null pointer");
  if (C < 1 || C > 128 || !lin_shape_ok(B, N, O)) return fail(SAMBLE_E_INVALID, "samble_linear_dx_tri_f32: 1 <= C <= 128, O a multiple of 32");
  if ((g_rs & 3) || (g_bs & 3) || ((uintptr_t)g & 15)) return fail(SAMBLE_E_INVALID, "samble_linear_dx_tri_f32: g rows must be 16-byte aligned");
  return done(samble_launch_linear_dx(g, g_bs, g_rs, w_tr_image, O, B, C, N, dx, dx_bs, residual, (hipStream_t)stream), "samble_linear_dx_tri_f32");
}

SAMBLE_API size_t samble_linear_dw_workspace_bytes(int B, int N, int O) {
  return lin_shape_ok(B, N, O) ? samble_linear_dw_ws_bytes(B, N, O) : 0;
}

SAMBLE_API int samble_linear_dw_tri_f32(const float* g, int64_t g_bs, int64_t g_rs, const float* x, int64_t x_bs, int B, int C,
                                        int N, int O, float* dW, void* ws, size_t ws_bytes, void* stream) {
  if (!g || !x || !dW || !ws) return fail(SAMBLE_E_INVALID, "samble_linear_dw_tri_f32: null pointer");
  if (C < 1 || C > 128 || !lin_shape_ok(B, N, O) || (O & 127))
    return fail(SAMBLE_E_INVALID, "samble_linear_dw_tri_f32: 1 <= C <= 128, O a multiple of 128");
  if ((g_rs & 3) || (g_bs & 3) || ((uintptr_t)g & 15)) return fail(SAMBLE_E_INVALID, "samble_linear_dw_tri_f32: g rows must be 16-byte aligned");
  if (ws_bytes < samble_linear_dw_ws_bytes(B, N, O)) return fail(SAMBLE_E_WORKSPACE, "samble_linear_dw_tri_f32: workspace too small");
  return done(samble_launch_linear_dw(g, g_bs, g_rs, x, x_bs, B, C, N, O, dW, 0, ws, (hipStream_t)stream, 0), "samble_linear_dw_tri_f32");
}

/* the same sum, written transposed: dWt (128, O) row-major (C = 128 only) */
SAMBLE_API int samble_linear_dw_t_tri_f32(const float* g, int64_t g_bs, int64_t g_rs, const float* x, int64_t x_bs, int B, int C,
                                          int N, int O, float* dWt, void* ws, size_t ws_bytes, void* stream) {
  if (!g || !x || !dWt || !ws) return fail(SAMBLE_E_INVALID, "samble_linear_dw_t_tri_f32: null pointer");
  if (C != 128 || !lin_shape_ok(B, N, O) || (O & 127))
    return fail(SAMBLE_E_INVALID, "samble_linear_dw_t_tri_f32: C must be 128, O a multiple of 128");
  if ((g_rs & 3) || (g_bs & 3) || ((uintptr_t)g & 15)) return fail(SAMBLE_E_INVALID, "samble_linear_dw_t_tri_f32: g rows must be 16-byte aligned");
  if (ws_bytes < samble_linear_dw_ws_bytes(B, N, O)) return fail(SAMBLE_E_WORKSPACE, "samble_linear_dw_t_tri_f32: workspace too small");
  return done(samble_launch_linear_dw(g, g_bs, g_rs, x, x_bs, B, C, N, O, dWt, 1, ws, (hipStream_t)stream, 0), "samble_linear_dw_t_tri_f32");
}

/* channel-major in, channel-major out: out (B, O, N) = W x [+ out when accumulate != 0], and the weight gradient of that
   layer from a channel-major output gradient g (B, O, N): dW (O, 128) = sum over clouds and points of g x^T */
SAMBLE_API int samble_linear_fwd_cm_f32(const float* x, int64_t x_bs, int B, int C, int N, const void* w_rm_image, int O,
                                        int accumulate, float* out, int64_t o_bs, void* stream) {
  if (!x || !w_rm_image || !out) return fail(SAMBLE_E_INVALID, "samble_linear_fwd_cm_f32: null pointer");
  if (C < 1 || C > 128 || !lin_shape_ok(B, N, O)) return fail(SAMBLE_E_INVALID, "samble_linear_fwd_cm_f32: 1 <= C <= 128, O a multiple of 32");
  if (o_bs < (int64_t)O * N) return fail(SAMBLE_E_INVALID, "samble_linear_fwd_cm_f32: output clouds overlap");
  return done(samble_launch_linear_fwd_cm(x, x_bs, B, C, N, w_rm_image, O, accumulate, out, o_bs, (hipStream_t)stream),
              "samble_linear_fwd_cm_f32");
}

SAMBLE_API int samble_linear_dw_cm_f32(const float* g, int64_t g_bs, const float* x, int64_t x_bs, int B, int C, int N, int O,
                                       float* dW, void* ws, size_t ws_bytes, void* stream) {
  if (!g || !x || !dW || !ws) return fail(SAMBLE_E_INVALID, "samble_linear_dw_cm_f32: null pointer");
  if (C < 1 || C > 128 || !lin_shape_ok(B, N, O) || (O & 127))
    return fail(SAMBLE_E_INVALID, "samble_linear_dw_cm_f32: 1 <= C <= 128, O a multiple of 128");
  if (ws_bytes < samble_linear_dw_ws_bytes(B, N, O)) return fail(SAMBLE_E_WORKSPACE, "samble_linear_dw_cm_f32: workspace too small");
  return done(samble_launch_linear_dw(g, g_bs, N, x, x_bs, B, C, N, O, dW, 0, ws, (hipStream_t)stream, 1), "samble_linear_dw_cm_f32");
}

SAMBLE_API size_t samble_amax_bwd_workspace_bytes(int B, int N, int O) {
  return (B > 0 && N > 0 && O > 0) ? samble_amax_bwd_ws_bytes(B, N, O) : 0;
}

SAMBLE_API int samble_amax_bwd_f32(const float* x, int64_t x_bs, int B, int C, int N, const int32_t* arg, const float* gy,
                                   const float* W, int O, float* dx_inout, int64_t dx_bs, float* dW, void* ws,
                                   size_t ws_bytes, void* stream) {
  if (!x || !arg || !gy || !W || !dx_inout || !dW || !ws) return fail(SAMBLE_E_INVALID, "samble_amax_bwd_f32: null pointer");
  if (C != 128 || B <= 0 || N <= 0 || O <= 0 || (O & 3) || N > 32767 || O > 8192)
    return fail(SAMBLE_E_INVALID, "samble_amax_bwd_f32: C must be 128, N <= 32767, O <= 8192 and a multiple of 4");
  if (((size_t)N + 4 + (size_t)O) * 4 > 150 * 1024) return fail(SAMBLE_E_INVALID, "samble_amax_bwd_f32: N + O too large for LDS");
  if (ws_bytes < samble_amax_bwd_ws_bytes(B, N, O)) return fail(SAMBLE_E_WORKSPACE, "samble_amax_bwd_f32: workspace too small");
  return done(samble_launch_amax_bwd(x, x_bs, B, N, arg, gy, W, O, dx_inout, dx_bs, dW, ws, (hipStream_t)stream),
              "samble_amax_bwd_f32");
}

SAMBLE_API size_t samble_proj_workspace_bytes(int B, int N) {
  const size_t fwd = 8 * 384 * sizeof(float);
  const size_t bwd = samble_proj_bwd_ws_floats(B, N) * sizeof(float);
  return fwd > bwd ? fwd : bwd;
}

SAMBLE_API int samble_proj_fwd_f32(const float* x, int64_t x_bs, int B, int C, int N, const float* tokens, int nt,
                                   const float* W, float* qkv, int64_t o_bs, int64_t o_rs, void* ws, size_t ws_bytes,
                                   void* stream) {
  if (!x || !W || !qkv || !ws || (nt > 0 && !tokens)) return fail(SAMBLE_E_INVALID, "samble_proj_fwd_f32: null pointer");
  if (C != 128) return fail(SAMBLE_E_INVALID, "samble_proj_fwd_f32: C = D must be 128");
  if (nt < 0 || nt > 8 || B <= 0 || N <= 0) return fail(SAMBLE_E_INVALID, "samble_proj_fwd_f32: bad B/N/nt");
  if ((o_rs & 3) || (o_bs & 3)) return fail(SAMBLE_E_INVALID, "samble_proj_fwd_f32: output strides must be multiples of 4");
  if (ws_bytes < 8 * 384 * sizeof(float)) return fail(SAMBLE_E_WORKSPACE, "samble_proj_fwd_f32: workspace too small");
  return done(samble_launch_proj_fwd(x, x_bs, B, N, tokens, nt, W, nullptr, nullptr, qkv, o_bs, o_rs, (float*)ws, nullptr, nullptr, 0, nullptr,
                                     (hipStream_t)stream),
              "samble_proj_fwd_f32");
}

/* the same projection on the bf16 matrix cores with split fp32 operands: the workspace also holds W's image */
SAMBLE_API size_t samble_proj_w_image_bytes(void) { return samble_proj_tri_image_bytes(); }

SAMBLE_API size_t samble_proj_fwd_tri_workspace_bytes(void) { return 8 * 384 * sizeof(float) + 256 + samble_proj_tri_image_bytes(); }

SAMBLE_API int samble_proj_fwd_tri_f32(const float* x, int64_t x_bs, int B, int C, int N, const float* tokens, int nt,
                                       const float* W, float* qkv, int64_t o_bs, int64_t o_rs, void* ws, size_t ws_bytes,
                                       void* stream) {
  if (!x || !W || !qkv || !ws || (nt > 0 && !tokens)) return fail(SAMBLE_E_INVALID, "samble_proj_fwd_tri_f32: null pointer");
  if (C != 128) return fail(SAMBLE_E_INVALID, "samble_proj_fwd_tri_f32: C = D must be 128");
  if (nt < 0 || nt > 8 || B <= 0 || N <= 0) return fail(SAMBLE_E_INVALID, "samble_proj_fwd_tri_f32: bad B/N/nt");
  if ((o_rs & 3) || (o_bs & 3))
    return fail(SAMBLE_E_INVALID, "samble_proj_fwd_tri_f32: output strides must be multiples of 4");
  if (ws_bytes < samble_proj_fwd_tri_workspace_bytes())
    return fail(SAMBLE_E_WORKSPACE, "samble_proj_fwd_tri_f32: workspace too small");
  char* wimg = (char*)ws + 8 * 384 * sizeof(float) + 256 - ((8 * 384 * sizeof(float)) & 255);
  return done(samble_launch_proj_fwd(x, x_bs, B, N, tokens, nt, W, nullptr, nullptr, qkv, o_bs, o_rs, (float*)ws, wimg, nullptr, 0, nullptr,
                                     (hipStream_t)stream),
              "samble_proj_fwd_tri_f32");
}

SAMBLE_API int samble_proj_fwd_split_tri_f32(const float* x, int64_t x_bs, int B, int C, int N, const float* tokens, int nt,
                                             const float* W, const float* Wk, const float* Wv, float* qkv, int64_t o_bs,
                                             int64_t o_rs, void* q_image,
                                             void* k_image, void* v_tr_image, void* k_tr_image, void* v_rm_image, int rows,
                                             void* w_tr_image, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !W || !qkv || !ws || (nt > 0 && !tokens) || !q_image || !k_image || !v_tr_image || (!k_tr_image != !v_rm_image))
    return fail(SAMBLE_E_INVALID, "samble_proj_fwd_split_tri_f32: null pointer (the two backward images come as a pair)");
  if (!Wk != !Wv) return fail(SAMBLE_E_INVALID, "samble_proj_fwd_split_tri_f32: Wk and Wv come as a pair");
  if (C != 128) return fail(SAMBLE_E_INVALID, "samble_proj_fwd_split_tri_f32: C = D must be 128");
  if (rows != SAMBLE_PROJ_ROWS_ALL && rows != SAMBLE_PROJ_ROWS_Q_ONLY)
    return fail(SAMBLE_E_INVALID, "samble_proj_fwd_split_tri_f32: rows is SAMBLE_PROJ_ROWS_ALL or SAMBLE_PROJ_ROWS_Q_ONLY");
  if (nt < 0 || nt > 8 || B <= 0 || N <= 0) return fail(SAMBLE_E_INVALID, "samble_proj_fwd_split_tri_f32: bad B/N/nt");
  if ((o_rs & 3) || (o_bs & 3))
    return fail(SAMBLE_E_INVALID, "samble_proj_fwd_split_tri_f32: output strides must be multiples of 4");
  if (ws_bytes < samble_proj_fwd_tri_workspace_bytes())
    return fail(SAMBLE_E_WORKSPACE, "samble_proj_fwd_split_tri_f32: workspace too small");
  char* wimg = (char*)ws + 8 * 384 * sizeof(float) + 256 - ((8 * 384 * sizeof(float)) & 255);
  void* const images[5] = {q_image, k_image, v_tr_image, k_tr_image, v_rm_image};
  return done(samble_launch_proj_fwd(x, x_bs, B, N, tokens, nt, W, Wk, Wv, qkv, o_bs, o_rs, (float*)ws, wimg, images,
                                     rows == SAMBLE_PROJ_ROWS_Q_ONLY, w_tr_image, (hipStream_t)stream),
              "samble_proj_fwd_split_tri_f32");
}

SAMBLE_API int samble_proj_bwd_f32(const float* dqkv, int64_t g_bs, int64_t g_rs, const float* x, int64_t x_bs, int B,
                                   int C, int N, const float* tokens, int nt, const float* W, float* dx, int64_t dx_bs,
                                   float* dW, float* dtokens, void* ws, size_t ws_bytes, void* stream) {
  if (!dqkv || !x || !W || !ws) return fail(SAMBLE_E_INVALID, "samble_proj_bwd_f32: null pointer");
  if (C != 128) return fail(SAMBLE_E_INVALID, "samble_proj_bwd_f32: C = D must be 128");
  if (nt < 0 || nt > 8) return fail(SAMBLE_E_INVALID, "samble_proj_bwd_f32: need 0 <= nt <= 8");
  if (dW && nt > 0 && (!dtokens || !tokens)) return fail(SAMBLE_E_INVALID, "samble_proj_bwd_f32: dtokens/tokens missing");
  if ((g_rs & 3) || (g_bs & 3)) return fail(SAMBLE_E_INVALID, "samble_proj_bwd_f32: strides must be multiples of 4");
  if (ws_bytes < samble_proj_bwd_ws_floats(B, N) * sizeof(float))
    return fail(SAMBLE_E_WORKSPACE, "samble_proj_bwd_f32: workspace too small");
  return done(samble_launch_proj_bwd(dqkv, g_bs, g_rs, x, x_bs, B, N, tokens, nt, W, nullptr, nullptr, dx, dx_bs, dW, dtokens, (float*)ws, nullptr,
                                     nullptr, nullptr, (hipStream_t)stream),
              "samble_proj_bwd_f32");
}

/* the same backward with dx on the bf16 matrix cores (split fp32 operands); the workspace also holds W's transposed image */
SAMBLE_API size_t samble_proj_bwd_tri_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  return ((samble_proj_bwd_ws_floats(B, N) * sizeof(float) + 255) & ~(size_t)255) + samble_proj_tri_image_bytes();
}

SAMBLE_API int samble_proj_bwd_tri_f32(const float* dqkv, int64_t g_bs, int64_t g_rs, const float* x, int64_t x_bs, int B,
                                       int C, int N, const float* tokens, int nt, const float* W, const float* Wk,
                                       const float* Wv, const void* w_tr_image, float* dx, int64_t dx_bs, float* dW,
                                       float* dtokens, const float* dx_residual, void* ws, size_t ws_bytes, void* stream) {
  if (!dqkv || !x || !W || !ws) return fail(SAMBLE_E_INVALID, "samble_proj_bwd_tri_f32: null pointer");
  if (!Wk != !Wv || (Wk && !w_tr_image))
    return fail(SAMBLE_E_INVALID, "samble_proj_bwd_tri_f32: Wk and Wv come as a pair, and only together with w_tr_image");
  if (C != 128) return fail(SAMBLE_E_INVALID, "samble_proj_bwd_tri_f32: C = D must be 128");
  if (nt < 0 || nt > 8 || B <= 0 || N <= 0) return fail(SAMBLE_E_INVALID, "samble_proj_bwd_tri_f32: bad B/N/nt");
  if (dW && nt > 0 && (!dtokens || !tokens)) return fail(SAMBLE_E_INVALID, "samble_proj_bwd_tri_f32: dtokens/tokens missing");
  if ((g_rs & 3) || (g_bs & 3)) return fail(SAMBLE_E_INVALID, "samble_proj_bwd_tri_f32: strides must be multiples of 4");
  if (ws_bytes < samble_proj_bwd_tri_workspace_bytes(B, N))
    return fail(SAMBLE_E_WORKSPACE, "samble_proj_bwd_tri_f32: workspace too small");
  char* wtr = (char*)ws + ((samble_proj_bwd_ws_floats(B, N) * sizeof(float) + 255) & ~(size_t)255);
  return done(samble_launch_proj_bwd(dqkv, g_bs, g_rs, x, x_bs, B, N, tokens, nt, W, Wk, Wv, dx, dx_bs, dW, dtokens, (float*)ws, wtr,
                                     w_tr_image, dx ? dx_residual : nullptr, (hipStream_t)stream),
              "samble_proj_bwd_tri_f32");
}

SAMBLE_API int samble_n2p_attn_fwd_f32(const float* qkv, int64_t bs, int64_t rs, const int32_t* nn, int B, int N, int KN,
                                       int C, int heads, int diff, float* out, float* att, const float* residual,
                                       void* stream) {
  if (!qkv || !nn || !out) return fail(SAMBLE_E_INVALID, "samble_n2p_attn_fwd_f32: null pointer");
  if (C != 128 || (heads != 4 && heads != 2 && heads != 1))
    return fail(SAMBLE_E_INVALID, "samble_n2p_attn_fwd_f32: built for C = 128 with 4, 2 or 1 head(s)");
  if (att && (heads != 1 || KN > 64))
    return fail(SAMBLE_E_INVALID, "samble_n2p_attn_fwd_f32: the probability output needs heads == 1 and K <= 64");
  if ((rs & 3) || (bs & 3) || rs < 3 * C) return fail(SAMBLE_E_INVALID, "samble_n2p_attn_fwd_f32: bad strides");
  if (B <= 0 || N <= 0 || KN <= 0) return fail(SAMBLE_E_INVALID, "samble_n2p_attn_fwd_f32: bad sizes");
  return done(samble_launch_n2p_fwd(qkv, bs, rs, nn, B, N, KN, diff, (float)(1.0 / sqrt((double)(C / heads))), out,
                                    heads, att, residual, (hipStream_t)stream),
              "samble_n2p_attn_fwd_f32");
}

SAMBLE_API int samble_attn_colsum_f32(const float* Q, int64_t q_bs, int64_t q_rs, const float* K, int64_t k_bs,
                                      int64_t k_rs, const float* lse, int B, int N, int D, float* colsum, void* stream) {
  if (!Q || !K || !lse || !colsum) return fail(SAMBLE_E_INVALID, "samble_attn_colsum_f32: null pointer");
  if (D != 128) return fail(SAMBLE_E_INVALID, "samble_attn_colsum_f32: D must be 128");
  if ((q_rs & 3) || (k_rs & 3) || (q_bs & 3) || (k_bs & 3))
    return fail(SAMBLE_E_INVALID, "samble_attn_colsum_f32: strides must be multiples of 4 elements");
  return done(samble_launch_attn_colsum(Q, q_bs, q_rs, K, k_bs, k_rs, lse, B, N, inv_sqrt_d(D), colsum,
                                        (hipStream_t)stream),
              "samble_attn_colsum_f32");
}

SAMBLE_API int samble_stat_score_f32(const float* stat, int B, int N, float* score, float* z, void* stream) {
  if (!stat || !score || !z || B <= 0 || N <= 0) return fail(SAMBLE_E_INVALID, "samble_stat_score_f32: bad argument");
  return done(samble_launch_stat_score(stat, B, N, score, z, (hipStream_t)stream), "samble_stat_score_f32");
}

SAMBLE_API size_t samble_n2p_attn_bwd_workspace_bytes(int B, int N, int KN) {
  return samble_n2p_bwd_ws_floats(B, N, KN) * sizeof(float);
}

SAMBLE_API int samble_n2p_attn_bwd_f32(const float* qkv, int64_t bs, int64_t rs, const int32_t* nn, const float* g,
                                       int B, int N, int KN, int C, int heads, int diff, float* dqkv, int64_t dbs,
                                       int64_t drs, const int32_t* inv_order, const int32_t* inv_offsets, void* ws,
                                       size_t ws_bytes, void* stream) {
  if ((inv_order == nullptr) != (inv_offsets == nullptr))
    return fail(SAMBLE_E_INVALID, "samble_n2p_attn_bwd_f32: inverse lists need both arrays");
  if (!qkv || !nn || !g || !dqkv || !ws) return fail(SAMBLE_E_INVALID, "samble_n2p_attn_bwd_f32: null pointer");
  if (C != 128 || (heads != 4 && heads != 2 && heads != 1))
    return fail(SAMBLE_E_INVALID, "samble_n2p_attn_bwd_f32: built for C = 128 with 4, 2 or 1 head(s)");
  if (KN < 1 || KN > 32) return fail(SAMBLE_E_INVALID, "samble_n2p_attn_bwd_f32: need 1 <= K <= 32");
  if ((rs & 3) || (bs & 3) || (drs & 3) || (dbs & 3) || rs < 3 * C || drs < 3 * C)
    return fail(SAMBLE_E_INVALID, "samble_n2p_attn_bwd_f32: bad strides");
  if (ws_bytes < samble_n2p_attn_bwd_workspace_bytes(B, N, KN))
    return fail(SAMBLE_E_WORKSPACE, "samble_n2p_attn_bwd_f32: workspace too small");
  return done(samble_launch_n2p_bwd(qkv, bs, rs, nn, g, B, N, KN, diff, (float)(1.0 / sqrt((double)(C / heads))), dqkv,
                                    dbs, drs, (float*)ws, heads, inv_order, inv_offsets, (hipStream_t)stream),
              "samble_n2p_attn_bwd_f32");
}
