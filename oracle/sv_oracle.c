/*
 * sv_oracle.c — CPU ORACLE (test infrastructure, not product code).
 *
 * Plain-C restatement of the arithmetic MinkowskiEngine 0.5.4 performs for the reference on the hot path
 * (third-party, un-vendored: requirements.txt:101; call sites model/backbone/minkunet.py:125-187,
 * model/backbone/resnet.py:95-127, model/robotnet_segmentation.py:55-64, app/inference_engine.py:405-417).
 * PARITY UNPINNED for this part: the reference ships no golden vectors, no checkpoints and ME cannot be installed
 * here, so the definitions below are this build's documented restatement of ME's public semantics.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Accumulation order is part of the definition (it is what makes labels bit-exact, SURVEY.md §7 "hard parts"):
 *   acc[o][n] = one fmaf chain over kernel offsets k ascending, then input channels c ascending,
 *               skipping offsets whose neighbour is absent.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ACT_NONE 0
#define ACT_RELU 1
#define ACT_LEAKY 2

static inline float apply_act(float y, int act, float slope) {
  if (act == ACT_RELU) return y < 0.f ? 0.f : y; /* NaN propagates (torch.relu semantics) */
  if (act == ACT_LEAKY) return y > 0.f ? y : y * slope;
  return y;
}

/* Sparse convolution / transposed convolution / linear with fused BN(eval)-affine, residual and activation.
 * nbr: int32[K][ld] neighbour table in the canonical row order of the OUTPUT map (-1 = absent); NULL = identity
 * (kernel_size 1).  W: float[K][Cin][Cout].   ME.MinkowskiConvolution semantics: out[o] = sum_k in[nbr_k(o)] W[k]. */
int or_conv_fwd(const float* in, int64_t in_ld, int Cin, const float* W, int K, int Cout, const int32_t* nbr,
                int64_t ld, int64_t V_out, const float* scale, const float* shift, const float* residual,
                int64_t res_ld, int act, float slope, float* out, int64_t out_ld, int nthreads) {
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
    float* acc = (float*)aligned_alloc(64, ((size_t)Cout * sizeof(float) + 63) / 64 * 64);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
    for (int64_t o = 0; o < V_out; ++o) {
      for (int n = 0; n < Cout; ++n) acc[n] = 0.0f;
      for (int k = 0; k < K; ++k) {
        int64_t src = nbr ? (int64_t)nbr[(int64_t)k * ld + o] : o;
        if (src < 0) continue;
        const float* a = in + src * in_ld;
        const float* w = W + (size_t)k * Cin * Cout;
        for (int c = 0; c < Cin; ++c) {
          const float av = a[c];
          const float* wr = w + (size_t)c * Cout;
          for (int n = 0; n < Cout; ++n) acc[n] = fmaf(av, wr[n], acc[n]);
        }
      }
      float* dst = out + o * out_ld;
      for (int n = 0; n < Cout; ++n) {
        float y = acc[n];
        if (scale)
          y = fmaf(y, scale[n], shift ? shift[n] : 0.0f);
        else if (shift)
          y = y + shift[n];
        if (residual) y = y + residual[o * res_ld + n];
        dst[n] = apply_act(y, act, slope);
      }
    }
    free(acc);
  }
  return 0;
}

/* ME.MinkowskiBatchNorm(eval) / ReLU / LeakyReLU outside a conv: out = act(fmaf(x, scale, shift) + residual) */
int or_affine_act(const float* in, int64_t in_ld, int C, int64_t V, const float* scale, const float* shift,
                  const float* residual, int64_t res_ld, int act, float slope, float* out, int64_t out_ld) {
  for (int64_t v = 0; v < V; ++v)
    for (int c = 0; c < C; ++c) {
      float y = in[v * in_ld + c];
      if (scale)
        y = fmaf(y, scale[c], shift ? shift[c] : 0.0f);
      else if (shift)
        y = y + shift[c];
      if (residual) y = y + residual[v * res_ld + c];
      out[v * out_ld + c] = apply_act(y, act, slope);
    }
  return 0;
}

/* UNWEIGHTED_AVERAGE quantisation (app/inference_engine.py:411): per-voxel mean of point features, summed
 * sequentially in ascending original point index; mode 1 = first point (sparse_quantize representative). */
int or_voxel_reduce(const float* feats, int C, const int32_t* order, const int32_t* seg_start, int64_t V, int mode,
                    float* out) {
  for (int64_t v = 0; v < V; ++v) {
    int s = seg_start[v], e = seg_start[v + 1];
    for (int c = 0; c < C; ++c) {
      float acc = feats[(int64_t)order[s] * C + c];
      if (mode == 0) {
        for (int j = s + 1; j < e; ++j) acc += feats[(int64_t)order[j] * C + c];
        acc = acc / (float)(e - s);
      }
      out[v * C + c] = acc;
    }
  }
  return 0;
}

/* ME.MinkowskiGlobalMaxPooling / GlobalAvgPooling over rows [bs[b], bs[b+1]) (model/robotnet.py:43, robotnet_encode.py:41).
 * MAX is order-free.  AVG: a plain sequential float32 sum in row order divided by the row count - the oracle states the
 * operation, not the device kernel's reduction tree; a float32 mean depends on the summation order in its last bits, so
 * comparisons with the GPU's pooled values carry a tolerance (1e-4, the north_star's on pose floats; tests/test_gpu_model.py)
 * and the order-free float64 check lives in tests/test_gpu_dense_grid.py. */
int or_global_pool(const float* F, int64_t ld, int C, const int32_t* batch_start, int B, int mode, float* out) {
  for (int b = 0; b < B; ++b) {
    int s = batch_start[b], e = batch_start[b + 1];
    for (int c = 0; c < C; ++c) {
      float a = mode == 0 ? -INFINITY : 0.0f;
      for (int r = s; r < e; ++r) {
        float v = F[(int64_t)r * ld + c];
        a = mode == 0 ? fmaxf(a, v) : a + v;
      }
      if (mode == 1) a = e > s ? a / (float)(e - s) : 0.0f;
      if (mode == 0 && e <= s) a = 0.0f;
      out[(int64_t)b * C + c] = a;
    }
  }
  return 0;
}

/* utils/output.py:67-73 after SparseTensor.slice: label = first row maximum, conf = sigmoid(max) */
int or_slice_argmax(const float* F, int64_t ld, int C, const int64_t* inverse, int64_t N, int64_t* label,
                    float* conf) {
  for (int64_t i = 0; i < N; ++i) {
    const float* row = F + inverse[i] * ld;
    float best = row[0];
    int bi = 0;
    for (int c = 1; c < C; ++c)
      if (row[c] > best) {
        best = row[c];
        bi = c;
      }
    label[i] = bi;
    if (conf) conf[i] = 1.0f / (1.0f + expf(-best));
  }
  return 0;
}

int or_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
