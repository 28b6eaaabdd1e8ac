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

}  // namespace sv

using namespace sv;

extern "C" {

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

}  // extern "C"
