// N3: point-to-point ICP (replaces utils/icp.py:13-83, i.e. open3d.pipelines.registration.registration_icp with
// TransformationEstimationPointToPoint, max_correspondence_distance = 0.1, at most 30 iterations, relative fitness /
// rmse tolerance 1e-6; call sites app/inference_engine.py:358-362).
//
// Per iteration: (1) nearest target point of every transformed source point — brute force, the target cloud staged
// through LDS in tiles, one thread per source point (S, T <= ~20k: 10^8 distance evaluations, latency-trivial on 256 CUs;
// no k-d tree); (2) one workgroup reduces the correspondences within the distance threshold to centroids + the 3x3
// cross-covariance in fp64 and solves the Kabsch problem with the same one-sided Jacobi SVD as sv_kabsch_batched.
// The iteration loop runs on the device side of the stream: a `state` record carries the current transform, the
// previous fitness / rmse and a converged flag that turns the remaining launches into no-ops (no host read-backs).
#include "sv_common.h"
#include "sv_dense_math.h"

namespace sv {

struct IcpState {
  double T[16];       // current source -> target transform (row-major 4x4)
  double fitness;     // inlier fraction of the last evaluation
  double rmse;        // inlier rmse of the last evaluation
  int iterations;     // updates applied
  int converged;
};

constexpr int NN_TILE = 1024;

__global__ __launch_bounds__(256) void icp_nn_kernel(const float* __restrict__ src, int S, const float* __restrict__ tgt,
                                                      int T, const IcpState* __restrict__ st, float max_d2,
                                                      int32_t* __restrict__ nn, float* __restrict__ d2out) {
  __shared__ float tile[NN_TILE * 3];
  if (st->converged) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (i < S) {
    const double x = src[i * 3], y = src[i * 3 + 1], z = src[i * 3 + 2];
    px = (float)(st->T[0] * x + st->T[1] * y + st->T[2] * z + st->T[3]);
    py = (float)(st->T[4] * x + st->T[5] * y + st->T[6] * z + st->T[7]);
    pz = (float)(st->T[8] * x + st->T[9] * y + st->T[10] * z + st->T[11]);
  }
  float best = INFINITY;
  int bi = -1;
  for (int base = 0; base < T; base += NN_TILE) {
    const int n = min(NN_TILE, T - base);
    __syncthreads();
    for (int e = threadIdx.x; e < n * 3; e += 256) tile[e] = tgt[(int64_t)base * 3 + e];
    __syncthreads();
    if (i < S) {
      for (int j = 0; j < n; ++j) {
        const float dx = tile[j * 3] - px, dy = tile[j * 3 + 1] - py, dz = tile[j * 3 + 2] - pz;
        const float d = dx * dx + dy * dy + dz * dz;
        if (d < best) {  // first minimum wins (ascending target index)
          best = d;
          bi = base + j;
        }
      }
    }
  }
  if (i < S) {
    const bool ok = bi >= 0 && best <= max_d2;
    nn[i] = ok ? bi : -1;
    d2out[i] = best;
  }
}

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double t = 0;
  for (int k = 0; k < 16; ++k) t += red[k];  // fixed order -> reproducible
  return t;
}

__global__ __launch_bounds__(1024) void icp_update_kernel(const float* __restrict__ src, int S,
                                                           const float* __restrict__ tgt,
                                                           const int32_t* __restrict__ nn,
                                                           const float* __restrict__ d2, IcpState* __restrict__ st,
                                                           double rel_fitness, double rel_rmse, int last) {
  __shared__ double red[16];
  if (st->converged) return;
  double Tm[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) Tm[k] = st->T[k];
  double acc[16];  // n, sp(3), sq(3), spq(9) -> 16 values
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.0;
  double err = 0.0;
  for (int i = threadIdx.x; i < S; i += 1024) {
    const int j = nn[i];
    if (j < 0) continue;
    const double x = src[i * 3], y = src[i * 3 + 1], z = src[i * 3 + 2];
    const double p[3] = {Tm[0] * x + Tm[1] * y + Tm[2] * z + Tm[3], Tm[4] * x + Tm[5] * y + Tm[6] * z + Tm[7],
                         Tm[8] * x + Tm[9] * y + Tm[10] * z + Tm[11]};
    const double q[3] = {tgt[j * 3], tgt[j * 3 + 1], tgt[j * 3 + 2]};
    acc[0] += 1.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      acc[1 + a] += p[a];
      acc[4 + a] += q[a];
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[7 + a * 3 + b] += p[a] * q[b];
    }
    err += (double)d2[i];
  }
  double tot[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) tot[k] = block_sum(acc[k], red);
  const double e2 = block_sum(err, red);
  if (threadIdx.x != 0) return;
  const double n = tot[0];
  const double fitness = n / (double)S;
  const double rmse = n > 0 ? sqrt(e2 / n) : 0.0;
  // open3d: stop when both the fitness and the rmse moved by less than the tolerances since the previous evaluation
  const bool stop = (st->iterations > 0 || st->fitness >= 0) &&
                    fabs(st->fitness - fitness) < rel_fitness && fabs(st->rmse - rmse) < rel_rmse;
  st->fitness = fitness;
  st->rmse = rmse;
  if (stop || n < 3 || last) {
    st->converged = 1;
    return;
  }
  double cp[3], cq[3], H[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    cp[a] = tot[1 + a] / n;
    cq[a] = tot[4 + a] / n;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) H[a][b] = tot[7 + a * 3 + b] - n * cp[a] * cq[b];
  double R[3][3], t[3];
  kabsch_from_covariance(H, cp, cq, R, t);
  // T <- [R t] * T
  double Tn[16];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c)
      Tn[r * 4 + c] = R[r][0] * Tm[0 * 4 + c] + R[r][1] * Tm[1 * 4 + c] + R[r][2] * Tm[2 * 4 + c] + (c == 3 ? t[r] : 0.0);
  Tn[12] = Tn[13] = Tn[14] = 0.0;
  Tn[15] = 1.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) st->T[k] = Tn[k];
  st->iterations += 1;
}

__global__ void icp_init_kernel(IcpState* st, const double* init_T) {
  if (threadIdx.x < 16) st->T[threadIdx.x] = init_T ? init_T[threadIdx.x] : ((threadIdx.x % 5 == 0) ? 1.0 : 0.0);
  if (threadIdx.x == 0) {
    st->fitness = -1.0;
    st->rmse = 0.0;
    st->iterations = 0;
    st->converged = 0;
  }
}

__global__ void icp_finish_kernel(const IcpState* st, double* out_T, double* out_stats) {
  if (threadIdx.x < 16) out_T[threadIdx.x] = st->T[threadIdx.x];
  if (threadIdx.x == 0 && out_stats) {
    out_stats[0] = st->fitness;
    out_stats[1] = st->rmse;
    out_stats[2] = (double)st->iterations;
  }
}

}  // namespace sv

using namespace sv;

extern "C" {

size_t sv_icp_workspace_bytes(int64_t S) { return align_up(sizeof(IcpState), 256) + align_up((size_t)S * 4, 256) * 2 + 1024; }

int sv_icp_point2point(const float* src, int64_t S, const float* tgt, int64_t T, const double* init_T,
                       double max_distance, int max_iterations, double rel_fitness, double rel_rmse, void* workspace,
                       size_t workspace_bytes, double* out_T, double* out_stats, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(S >= 3 && T >= 1 && S < (1 << 24) && T < (1 << 24), "need at least 3 source points and 1 target point");
  SV_CHECK_ARG(max_iterations >= 0 && max_distance > 0, "bad parameters");
  SV_CHECK_ARG(src && tgt && out_T && workspace, "null pointer");
  Workspace ws(workspace, workspace_bytes);
  IcpState* st = ws.take<IcpState>(1);
  int32_t* nn = ws.take<int32_t>(S);
  float* d2 = ws.take<float>(S);
  if (!ws.ok) {
    set_error("sv_icp_point2point: workspace too small");
    return SV_ERR_WORKSPACE;
  }
  hipLaunchKernelGGL(icp_init_kernel, dim3(1), dim3(64), 0, stream, st, init_T);
  const float max_d2 = (float)(max_distance * max_distance);
  const unsigned nb = (unsigned)((S + 255) / 256);
  // evaluation 0 .. max_iterations: each update is followed by a re-evaluation, the last one only evaluates
  for (int it = 0; it <= max_iterations; ++it) {
    hipLaunchKernelGGL(icp_nn_kernel, dim3(nb), dim3(256), 0, stream, src, (int)S, tgt, (int)T, st, max_d2, nn, d2);
    hipLaunchKernelGGL(icp_update_kernel, dim3(1), dim3(1024), 0, stream, src, (int)S, tgt, nn, d2, st, rel_fitness,
                       rel_rmse, it == max_iterations ? 1 : 0);
  }
  hipLaunchKernelGGL(icp_finish_kernel, dim3(1), dim3(64), 0, stream, st, out_T, out_stats);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

}  // extern "C"
