/* Sanitizer driver for the CPU oracle (test infrastructure; `make -C oracle asan` builds sv_oracle.c together with this
 * file under -fsanitize=address,undefined and runs it).  Every or_* entry point is exercised on exactly-sized heap
 * buffers - so any out-of-bounds access, misaligned read or signed overflow inside the oracle trips the sanitizer - with
 * the shapes the parity tests use: ragged strides, absent neighbours, empty batches, K = 1 / 8 / 27, Cin = 3. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int or_conv_fwd(const float* in, int64_t in_ld, int Cin, const float* W, int K, int Cout, const int32_t* nbr, int64_t ld,
                int64_t V_out, const float* scale, const float* shift, const float* residual, int64_t res_ld, int act,
                float slope, float* out, int64_t out_ld, int nthreads);
int or_affine_act(const float* in, int64_t in_ld, int C, int64_t V, const float* scale, const float* shift,
                  const float* residual, int64_t res_ld, int act, float slope, float* out, int64_t out_ld);
int or_voxel_reduce(const float* feats, int C, const int32_t* order, const int32_t* seg_start, int64_t V, int mode,
                    float* out);
int or_global_pool(const float* F, int64_t ld, int C, const int32_t* batch_start, int B, int mode, float* out);
int or_slice_argmax(const float* F, int64_t ld, int C, const int64_t* inverse, int64_t N, int64_t* label, float* conf);

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd(void) {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 32);
}
static float frand(void) { return (float)(rnd() % 20001) / 10000.0f - 1.0f; }
static float* fbuf(size_t n) {
  float* p = (float*)malloc((n ? n : 1) * sizeof(float));
  for (size_t i = 0; i < n; ++i) p[i] = frand();
  return p;
}

int main(void) {
  double checksum = 0.0;
  /* conv: K in {1, 8, 27}, odd channel counts, row strides wider than the channel count, absent neighbours */
  const int Ks[3] = {1, 8, 27}, Cins[3] = {3, 32, 37}, Couts[3] = {32, 7, 64};
  for (int t = 0; t < 3; ++t) {
    const int K = Ks[t], Cin = Cins[t], Cout = Couts[t];
    const int64_t V_in = 301, V_out = K == 1 ? 301 : 257, in_ld = Cin + 5, out_ld = Cout + 3, res_ld = Cout + 1, ld = V_out + 2;
    float *in = fbuf((V_in - 1) * in_ld + Cin), *W = fbuf((size_t)K * Cin * Cout), *sc = fbuf(Cout), *sh = fbuf(Cout);
    float *res = fbuf((V_out - 1) * res_ld + Cout), *out = fbuf((V_out - 1) * out_ld + Cout);
    int32_t* nbr = NULL;
    if (K > 1) {
      nbr = (int32_t*)malloc(((size_t)(K - 1) * ld + V_out) * sizeof(int32_t));
      for (int k = 0; k < K; ++k)
        for (int64_t o = 0; o < V_out; ++o) nbr[(size_t)k * ld + o] = (rnd() % 3 == 0) ? -1 : (int32_t)(rnd() % V_in);
    }
    for (int act = 0; act < 3; ++act)
      for (int nth = 1; nth <= 3; nth += 2) {
        or_conv_fwd(in, in_ld, Cin, W, K, Cout, nbr, ld, V_out, act ? sc : NULL, act == 1 ? NULL : sh, act == 2 ? res : NULL,
                    res_ld, act, 0.01f, out, out_ld, nth);
        for (int64_t o = 0; o < V_out; ++o)
          for (int n = 0; n < Cout; ++n) checksum += out[o * out_ld + n];
      }
    or_affine_act(out, out_ld, Cout, V_out, sc, sh, res, res_ld, 1, 0.01f, out, out_ld);
    free(in); free(W); free(sc); free(sh); free(res); free(out); free(nbr);
  }
  /* voxel reduce: ragged segments incl. single-point voxels */
  {
    const int64_t V = 100;
    const int C = 3;
    int32_t* seg = (int32_t*)malloc((V + 1) * sizeof(int32_t));
    seg[0] = 0;
    for (int64_t v = 0; v < V; ++v) seg[v + 1] = seg[v] + 1 + (int32_t)(rnd() % 4);
    const int N = seg[V];
    int32_t* order = (int32_t*)malloc(N * sizeof(int32_t));
    for (int i = 0; i < N; ++i) order[i] = N - 1 - i;
    float *feats = fbuf((size_t)N * C), *out = fbuf((size_t)V * C);
    for (int mode = 0; mode < 2; ++mode) {
      or_voxel_reduce(feats, C, order, seg, V, mode, out);
      for (int64_t i = 0; i < V * C; ++i) checksum += out[i];
    }
    free(seg); free(order); free(feats); free(out);
  }
  /* global pool with an empty batch; slice + argmax */
  {
    const int C = 70, B = 4;
    const int32_t bs[5] = {0, 17, 17, 40, 64};
    const int64_t ld = C + 2;
    float *F = fbuf(63 * ld + C), *out = fbuf((size_t)B * C);
    for (int mode = 0; mode < 2; ++mode) {
      or_global_pool(F, ld, C, bs, B, mode, out);
      for (int i = 0; i < B * C; ++i) checksum += out[i];
    }
    const int64_t N = 500;
    int64_t *inv = (int64_t*)malloc(N * sizeof(int64_t)), *label = (int64_t*)malloc(N * sizeof(int64_t));
    float* conf = fbuf(N);
    for (int64_t i = 0; i < N; ++i) inv[i] = rnd() % 64;
    or_slice_argmax(F, ld, C, inv, N, label, conf);
    or_slice_argmax(F, ld, C, inv, N, label, NULL);
    for (int64_t i = 0; i < N; ++i) checksum += (double)label[i] + conf[i];
    free(F); free(out); free(inv); free(label); free(conf);
  }
  if (!isfinite(checksum)) {
    fprintf(stderr, "oracle sanitizer driver: non-finite checksum\n");
    return 1;
  }
  printf("oracle sanitizer driver OK (checksum %.6f)\n", checksum);
  return 0;
}
