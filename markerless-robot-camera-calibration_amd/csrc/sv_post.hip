// K3/K8/K9: voxel->point slice, fused slice+argmax, per-batch global pooling.  HBM-bound row gathers / reductions.
//
// Replaces SparseTensor.slice(field) + utils/output.py:67-73 (row max -> label, sigmoid(conf))
// (app/inference_engine.py:417-419) and ME.MinkowskiGlobalMaxPooling / GlobalAvgPooling
// (model/robotnet.py:43,70; model/robotnet_encode.py:41,101).
#include "sv_common.h"

namespace sv {

__global__ __launch_bounds__(256) void batch_offsets_kernel(const uint64_t* __restrict__ keys, int64_t V, int B,
                                                             int32_t* __restrict__ batch_start) {
  int b = blockIdx.x * 256 + threadIdx.x;
  if (b > B) return;
  // first row whose batch index >= b (keys sorted, batch in the top bits)
  int64_t lo = 0, hi = V;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if ((int)(keys[mid] >> 54) < b)
      lo = mid + 1;
    else
      hi = mid;
  }
  batch_start[b] = (int32_t)lo;
}

// one workgroup per (batch, 64-channel slab); 4 waves stride the rows, lanes = channels (coalesced 256 B rows);
// per-wave partials are combined in fixed wave order -> deterministic.
__global__ __launch_bounds__(256) void global_pool_kernel(const float* __restrict__ F, int64_t ld, int C,
                                                           const int32_t* __restrict__ batch_start, int mode,
                                                           float* __restrict__ out) {
  __shared__ float part[4][64];
  const int b = blockIdx.x;
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const int s = batch_start[b], e = batch_start[b + 1];
  float acc = (mode == SV_POOL_MAX) ? -INFINITY : 0.0f;
  if (c < C) {
    for (int r = s + w; r < e; r += 4) {
      float v = F[(int64_t)r * ld + c];
      acc = (mode == SV_POOL_MAX) ? fmaxf(acc, v) : acc + v;
    }
  }
  part[w][threadIdx.x & 63] = acc;
  __syncthreads();
  if (w == 0 && c < C) {
    float a = part[0][threadIdx.x];
    for (int i = 1; i < 4; ++i) a = (mode == SV_POOL_MAX) ? fmaxf(a, part[i][threadIdx.x]) : a + part[i][threadIdx.x];
    if (mode == SV_POOL_AVG) a = (e > s) ? a / (float)(e - s) : 0.0f;
    if (mode == SV_POOL_MAX && e <= s) a = 0.0f;
    out[(int64_t)b * C + c] = a;
  }
}

__global__ __launch_bounds__(256) void slice_rows_kernel(const float* __restrict__ F, int64_t ld, int C,
                                                          const int64_t* __restrict__ inverse, int64_t N,
                                                          float* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= N * C) return;
  int64_t i = t / C;
  int c = (int)(t - i * C);
  out[t] = F[inverse[i] * ld + c];
}

__global__ __launch_bounds__(256) void slice_argmax_kernel(const float* __restrict__ F, int64_t ld, int C,
                                                            const int64_t* __restrict__ inverse, int64_t N,
                                                            int64_t* __restrict__ label, float* __restrict__ conf) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float* row = F + inverse[i] * ld;
  float best = row[0];
  int bi = 0;
  for (int c = 1; c < C; ++c) {
    float v = row[c];
    if (v > best) {
      best = v;
      bi = c;
    }
  }
  label[i] = bi;
  if (conf) conf[i] = 1.0f / (1.0f + expf(-best));
}

__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ in, int64_t in_ld, int C, int64_t V,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift,
                                                          const float* __restrict__ residual, int64_t res_ld, int act,
                                                          float slope, float* __restrict__ out, int64_t out_ld) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= V * C) return;
  int64_t v = t / C;
  int c = (int)(t - v * C);
  float y = in[v * in_ld + c];
  if (scale)
    y = __builtin_fmaf(y, scale[c], shift ? shift[c] : 0.0f);
  else if (shift)
    y = y + shift[c];
  if (residual) y = y + residual[v * res_ld + c];
  if (act == SV_ACT_RELU)
    y = y < 0.f ? 0.f : y;  // NaN stays NaN, as torch.relu
  else if (act == SV_ACT_LEAKY_RELU)
    y = y > 0.f ? y : y * slope;
  out[v * out_ld + c] = y;
}

// ---- A11 preprocessing (utils/preprocess.py:8-56) on device rows -----------------------------------------------------
// column statistics of x[N][C] (C <= 4): per column min, max (exact), sum (float64), and - when `sub` is given - the
// maximum over rows of the float32 norm sqrt(((x0-s0)^2 + (x1-s1)^2) + (x2-s2)^2), numpy's np.linalg.norm order.
// Stage 1: every workgroup reduces a contiguous slab of rows to one record; stage 2: one workgroup reduces the records
// in slab order (deterministic: no atomics).
constexpr int STAT_REC = 16;  // doubles per record: min[4], max[4], sum[4], normmax, pad
// numpy's min() / max() propagate NaN (fminf / fmaxf drop it): a NaN colour or point must steer the data-dependent
// branches of utils/preprocess.py:20-37 on the device exactly as on the host
__device__ __forceinline__ float nan_min(float a, float b) { return (a != a) ? a : ((b != b) ? b : fminf(a, b)); }
__device__ __forceinline__ float nan_max(float a, float b) { return (a != a) ? a : ((b != b) ? b : fmaxf(a, b)); }
__device__ __forceinline__ double nan_min(double a, double b) { return (a != a) ? a : ((b != b) ? b : fmin(a, b)); }
__device__ __forceinline__ double nan_max(double a, double b) { return (a != a) ? a : ((b != b) ? b : fmax(a, b)); }
__global__ __launch_bounds__(256) void col_stats_partial_kernel(const float* __restrict__ x, int64_t ld, int64_t N, int C,
                                                                 const float* __restrict__ sub, int64_t rows_per_block,
                                                                 double* __restrict__ rec) {
  __shared__ double sh[256 / 64][STAT_REC];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(N, r0 + rows_per_block);
  float mn[4], mx[4];
  double sm[4];
  float nmax = 0.0f;
  for (int c = 0; c < 4; ++c) {
    mn[c] = INFINITY;
    mx[c] = -INFINITY;
    sm[c] = 0.0;
  }
  for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
      v[c] = x[r * ld + c];
      mn[c] = nan_min(mn[c], v[c]);
      mx[c] = nan_max(mx[c], v[c]);
      sm[c] += (double)v[c];
    }
    if (sub) {
      const float a = v[0] - sub[0], b = v[1] - sub[1], d = (C > 2) ? v[2] - sub[2] : 0.0f;
      nmax = nan_max(nmax, sqrtf((a * a + b * b) + d * d));
    }
  }
  // wave reduction by shuffles, then across the four waves through LDS
  for (int off = 32; off >= 1; off >>= 1) {
    for (int c = 0; c < 4; ++c) {
      mn[c] = nan_min(mn[c], __shfl_down(mn[c], off));
      mx[c] = nan_max(mx[c], __shfl_down(mx[c], off));
      sm[c] += __shfl_down(sm[c], off);
    }
    nmax = nan_max(nmax, __shfl_down(nmax, off));
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) {
    for (int c = 0; c < 4; ++c) {
      sh[wid][c] = mn[c];
      sh[wid][4 + c] = mx[c];
      sh[wid][8 + c] = sm[c];
    }
    sh[wid][12] = nmax;
  }
  __syncthreads();
  if (threadIdx.x < STAT_REC) {
    const int j = threadIdx.x;
    double a = sh[0][j];
    for (int w = 1; w < 4; ++w)
      a = (j < 4) ? nan_min(a, sh[w][j]) : ((j < 8 || j == 12) ? nan_max(a, sh[w][j]) : a + sh[w][j]);
    rec[(int64_t)blockIdx.x * STAT_REC + j] = (j > 12) ? 0.0 : a;
  }
}

__global__ __launch_bounds__(64) void col_stats_final_kernel(const double* __restrict__ rec, int nrec, int C,
                                                              float* __restrict__ mn, float* __restrict__ mx,
                                                              double* __restrict__ sum, float* __restrict__ normmax) {
  const int j = threadIdx.x;
  if (j >= 13) return;
  double a = rec[j];
  for (int b = 1; b < nrec; ++b) {
    const double v = rec[(int64_t)b * STAT_REC + j];
    a = (j < 4) ? nan_min(a, v) : ((j < 8 || j == 12) ? nan_max(a, v) : a + v);
  }
  if (j < 4 && j < C) mn[j] = (float)a;
  if (j >= 4 && j < 8 && j - 4 < C) mx[j - 4] = (float)a;
  if (j >= 8 && j < 12 && j - 8 < C) sum[j - 8] = a;
  if (j == 12 && normmax) *normmax = (float)a;
}

// out = (x - sub[c]) / div[c] + add[c]; each of the three is applied only when its pointer is given (IEEE f32 operations
// in that order, so `points - offset` and `rgb / 255` round exactly as numpy's do)
__global__ __launch_bounds__(256) void center_scale_kernel(const float* __restrict__ x, int64_t ld, int64_t N, int C,
                                                            const float* __restrict__ sub, const float* __restrict__ div,
                                                            const float* __restrict__ add, float* __restrict__ out,
                                                            int64_t out_ld) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= N * C) return;
  const int64_t r = t / C;
  const int c = (int)(t - r * C);
  float v = x[r * ld + c];
  if (sub) v = v - sub[c];
  if (div) v = v / div[c];
  if (add) v = v + add[c];
  out[r * out_ld + c] = v;
}

// ---- key-point selection (utils/output.py:81-87): softmax over the classes of every point, then per class the highest
//      probability over all points and the point that has it.  One thread per point: the row's softmax lives in
//      registers; per class a wave reduction over packed (probability bits << 32 | ~index) keys - probabilities are
//      positive floats, so their bit patterns order like the values, and among equal probabilities the LOWEST point index
//      wins - and one 64-bit atomicMax per wave and class (max is associative and commutative: deterministic).
template <int CMAX>
__global__ __launch_bounds__(256) void kp_softmax_max_kernel(const float* __restrict__ logits, int64_t ld, int C,
                                                              int64_t N, unsigned long long* __restrict__ best) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float p[CMAX];
  const bool ok = r < N;
  if (ok) {
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      p[c] = c < C ? logits[r * ld + c] : -INFINITY;
      m = fmaxf(m, p[c]);
    }
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      p[c] = c < C ? expf(p[c] - m) : 0.0f;
      sum += p[c];
    }
#pragma unroll
    for (int c = 0; c < CMAX; ++c) p[c] = p[c] / sum;
  }
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    if (c >= C) break;
    unsigned long long key = 0ull;
    if (ok && p[c] == p[c])  // a NaN row never wins
      key = ((unsigned long long)__float_as_uint(p[c]) << 32) | (unsigned long long)(0xffffffffu - (unsigned)r);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned long long o = __shfl_xor(key, off, 64);
      key = o > key ? o : key;
    }
    if ((threadIdx.x & 63) == 0 && key) atomicMax(&best[c], key);
  }
}

__global__ void kp_finalize_kernel(const unsigned long long* __restrict__ best, int C, float conf_th,
                                   float* __restrict__ prob, int64_t* __restrict__ idx, int32_t* __restrict__ selected) {
  const int c = threadIdx.x;
  if (c >= C) return;
  const unsigned long long key = best[c];
  const float pr = key ? __uint_as_float((unsigned)(key >> 32)) : 0.0f;
  prob[c] = pr;
  idx[c] = key ? (int64_t)(0xffffffffu - (unsigned)(key & 0xffffffffull)) : -1;
  selected[c] = pr > conf_th ? 1 : 0;
}

// ---- top-k of one column (utils/output.py:45-64 get_pred_center: `out[:, 1].sort(descending=True)[1][:8]` - a full sort
//      of every point's vote to take eight).  Packed keys (order-preserving value bits << 32 | ~index): the largest key is
//      the largest value at the lowest index, keys are distinct, so the j-th selection is "the largest key below the
//      (j-1)-th" - no masking, no sort.  Stage 1: every block of 256 threads selects the k largest of its chunk; stage 2: one
//      block selects the k largest of the blocks' candidates.  NaN orders above +inf (torch.sort places NaN first).
__device__ __forceinline__ unsigned long long topk_key(float v, unsigned idx) {
  unsigned u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - idx);
}

constexpr int TOPK_CHUNK = 8192;  // values per stage-1 block
template <bool KEYS_IN>
__global__ __launch_bounds__(256) void topk_select_kernel(const float* __restrict__ x, int64_t ld,
                                                           const unsigned long long* __restrict__ keys_in, int64_t n, int k,
                                                           unsigned long long* __restrict__ keys_out) {
  __shared__ unsigned long long wave_best[4];
  __shared__ unsigned long long chosen;
  const int64_t base = (int64_t)blockIdx.x * TOPK_CHUNK;
  const int64_t end = KEYS_IN ? n : (base + TOPK_CHUNK < n ? base + TOPK_CHUNK : n);
  unsigned long long prev = ~0ull;
  for (int j = 0; j < k; ++j) {
    unsigned long long best = 0ull;  // key 0 = nothing left (a real key has index bits != all-ones or value bits != 0)
    for (int64_t i = (KEYS_IN ? 0 : base) + threadIdx.x; i < end; i += 256) {
      const unsigned long long key = KEYS_IN ? keys_in[i] : topk_key(x[i * ld], (unsigned)i);
      if (key < prev && key > best) best = key;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned long long o = __shfl_xor(best, off, 64);
      best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) wave_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long b = wave_best[0];
      for (int w = 1; w < 4; ++w) b = wave_best[w] > b ? wave_best[w] : b;
      chosen = b;
      keys_out[(int64_t)blockIdx.x * k + j] = b;
    }
    __syncthreads();
    prev = chosen ? chosen : 0ull;
    if (prev == 0ull) {  // fewer than k values: the remaining slots stay 0 (idx -1)
      if (threadIdx.x == 0)
        for (int jj = j + 1; jj < k; ++jj) keys_out[(int64_t)blockIdx.x * k + jj] = 0ull;
      break;
    }
  }
}

__global__ void topk_finalize_kernel(const unsigned long long* __restrict__ keys, int k, int64_t* __restrict__ idx) {
  const int j = threadIdx.x;
  if (j < k) idx[j] = keys[j] ? (int64_t)(0xffffffffu - (unsigned)(keys[j] & 0xffffffffull)) : -1;
}

}  // namespace sv

using namespace sv;

extern "C" {

size_t sv_topk_workspace_bytes(int64_t N, int k) {
  const int64_t blocks = N <= 0 ? 1 : (N + TOPK_CHUNK - 1) / TOPK_CHUNK;
  return (size_t)(blocks + 1) * (size_t)k * sizeof(unsigned long long) + 256;
}

int sv_topk_indices(const float* x, int64_t ld, int64_t N, int k, void* workspace, size_t workspace_bytes, int64_t* idx,
                    sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(k >= 1 && k <= 64 && ld >= 1 && N >= 0 && N < 0xffffffffll, "bad shape (1 <= k <= 64)");
  SV_CHECK_ARG(idx && workspace && (x || N == 0), "null pointer");
  if (workspace_bytes < sv_topk_workspace_bytes(N, k)) {
    set_error("sv_topk_indices: workspace too small");
    return SV_ERR_WORKSPACE;
  }
  const int64_t blocks = N <= 0 ? 1 : (N + TOPK_CHUNK - 1) / TOPK_CHUNK;
  unsigned long long* cand = (unsigned long long*)workspace;
  unsigned long long* fin = cand + blocks * k;
  hipLaunchKernelGGL(topk_select_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, x, ld, nullptr, N, k, cand);
  hipLaunchKernelGGL(topk_select_kernel<true>, dim3(1), dim3(256), 0, stream, nullptr, 0, cand, blocks * k, k, fin);
  hipLaunchKernelGGL(topk_finalize_kernel, dim3(1), dim3(64), 0, stream, fin, k, idx);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

size_t sv_col_stats_workspace_bytes(int64_t N) {
  const int64_t blocks = N <= 0 ? 1 : (N + 4095) / 4096;
  return (size_t)(blocks < 1024 ? blocks : 1024) * STAT_REC * sizeof(double) + 256;
}

int sv_col_stats(const float* x, int64_t ld, int64_t N, int C, const float* sub, void* workspace, size_t workspace_bytes,
                 float* col_min, float* col_max, double* col_sum, float* max_row_norm, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(N >= 1 && C >= 1 && C <= 4 && ld >= C, "bad shape (1 <= C <= 4, N >= 1)");
  SV_CHECK_ARG(x && workspace && col_min && col_max && col_sum, "null pointer");
  SV_CHECK_ARG(!max_row_norm || (sub && C >= 2), "max_row_norm needs sub and at least two columns");
  int64_t blocks = (N + 4095) / 4096;
  if (blocks > 1024) blocks = 1024;
  const int64_t rows_per_block = (N + blocks - 1) / blocks;
  if (workspace_bytes < (size_t)blocks * STAT_REC * sizeof(double)) {
    set_error("sv_col_stats: workspace too small");
    return SV_ERR_WORKSPACE;
  }
  double* rec = (double*)workspace;
  hipLaunchKernelGGL(col_stats_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, ld, N, C,
                     max_row_norm ? sub : nullptr, rows_per_block, rec);
  hipLaunchKernelGGL(col_stats_final_kernel, dim3(1), dim3(64), 0, stream, rec, (int)blocks, C, col_min, col_max, col_sum,
                     max_row_norm);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_center_scale(const float* x, int64_t ld, int64_t N, int C, const float* sub, const float* div, const float* add,
                    float* out, int64_t out_ld, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(N >= 0 && C >= 1 && ld >= C && out_ld >= C, "bad shape");
  if (N == 0) return SV_OK;
  SV_CHECK_ARG(x && out, "null pointer");
  hipLaunchKernelGGL(center_scale_kernel, dim3((unsigned)((N * C + 255) / 256)), dim3(256), 0, stream, x, ld, N, C, sub,
                     div, add, out, out_ld);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_batch_offsets(const uint64_t* keys, int64_t V, int B, int32_t* batch_start, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(B >= 1 && B <= SV_MAX_BATCH && V >= 0, "bad shape");
  SV_CHECK_ARG(batch_start && (keys || V == 0), "null pointer");
  hipLaunchKernelGGL(batch_offsets_kernel, dim3((unsigned)((B + 1 + 255) / 256)), dim3(256), 0, stream, keys, V, B,
                     batch_start);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_global_pool(const float* F, int64_t ld, int C, const int32_t* batch_start, int B, int mode, float* out,
                   sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(B >= 1 && C >= 1 && ld >= C, "bad shape");
  SV_CHECK_ARG(mode == SV_POOL_MAX || mode == SV_POOL_AVG, "bad mode");
  SV_CHECK_ARG(F && batch_start && out, "null pointer");
  hipLaunchKernelGGL(global_pool_kernel, dim3((unsigned)B, (unsigned)((C + 63) / 64)), dim3(256), 0, stream, F, ld, C,
                     batch_start, mode, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_slice_rows(const float* F, int64_t ld, int C, const int64_t* inverse, int64_t N, float* out,
                  sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(C >= 1 && ld >= C && N >= 0, "bad shape");
  if (N == 0) return SV_OK;
  SV_CHECK_ARG(F && inverse && out, "null pointer");
  int64_t total = N * C;
  hipLaunchKernelGGL(slice_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, F, ld, C, inverse,
                     N, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_slice_argmax(const float* F, int64_t ld, int C, const int64_t* inverse, int64_t N, int64_t* label, float* conf,
                    sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(C >= 1 && ld >= C && N >= 0, "bad shape");
  if (N == 0) return SV_OK;
  SV_CHECK_ARG(F && inverse && label, "null pointer");
  hipLaunchKernelGGL(slice_argmax_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, F, ld, C, inverse, N,
                     label, conf);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_affine_act(const float* in, int64_t in_ld, int C, int64_t V, const float* scale, const float* shift,
                  const float* residual, int64_t res_ld, int act, float slope, float* out, int64_t out_ld,
                  sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(C >= 1 && V >= 0 && in_ld >= C && out_ld >= C, "bad shape");
  SV_CHECK_ARG(act >= SV_ACT_NONE && act <= SV_ACT_LEAKY_RELU, "bad activation");
  if (V == 0) return SV_OK;
  SV_CHECK_ARG(in && out, "null pointer");
  SV_CHECK_ARG(!residual || res_ld >= C, "residual stride too small");
  int64_t total = V * C;
  hipLaunchKernelGGL(affine_act_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in, in_ld, C, V,
                     scale, shift, residual, res_ld, act, slope, out, out_ld);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_key_point_predictions(const float* logits, int64_t ld, int C, int64_t N, float conf_th, void* workspace,
                             size_t workspace_bytes, float* prob, int64_t* idx, int32_t* selected, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(C >= 1 && C <= 32 && ld >= C && N >= 0 && N < 0xffffffffll, "bad shape (1 <= C <= 32)");
  SV_CHECK_ARG(prob && idx && selected && workspace, "null pointer");
  SV_CHECK_ARG(workspace_bytes >= (size_t)C * sizeof(unsigned long long), "workspace too small (8 C bytes)");
  unsigned long long* best = (unsigned long long*)workspace;
  SV_HIP(hipMemsetAsync(best, 0, (size_t)C * sizeof(unsigned long long), stream));
  if (N > 0) {
    SV_CHECK_ARG(logits, "null pointer");
    const dim3 grid((unsigned)((N + 255) / 256));
    if (C <= 8)
      hipLaunchKernelGGL(kp_softmax_max_kernel<8>, grid, dim3(256), 0, stream, logits, ld, C, N, best);
    else
      hipLaunchKernelGGL(kp_softmax_max_kernel<32>, grid, dim3(256), 0, stream, logits, ld, C, N, best);
    SV_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(kp_finalize_kernel, dim3(1), dim3(32), 0, stream, best, C, conf_th, prob, idx, selected);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_key_point_predictions_batched(const float* logits, int64_t ld, int C, const int64_t* seg_start_host, int G, float conf_th,
                                     void* workspace, size_t workspace_bytes, float* prob, int64_t* idx, int32_t* selected,
                                     sv_stream_t stream_) {
  SV_CHECK_ARG(G >= 0 && seg_start_host, "bad segment table");
  SV_CHECK_ARG(workspace_bytes >= (size_t)G * (size_t)C * sizeof(unsigned long long), "workspace too small (8 C G bytes)");
  for (int g = 0; g < G; ++g) {
    const int64_t s = seg_start_host[g], e = seg_start_host[g + 1];
    SV_CHECK_ARG(e >= s && s >= 0, "segment starts must ascend");
    const int rc = sv_key_point_predictions(logits ? logits + s * ld : nullptr, ld, C, e - s, conf_th,
                                            (char*)workspace + (size_t)g * C * sizeof(unsigned long long),
                                            (size_t)C * sizeof(unsigned long long), prob + (size_t)g * C, idx + (size_t)g * C,
                                            selected + (size_t)g * C, stream_);
    if (rc) return rc;
  }
  return SV_OK;
}

}  // extern "C"
