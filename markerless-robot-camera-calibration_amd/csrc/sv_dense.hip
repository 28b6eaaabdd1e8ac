// K10/K11/K12: batched small dense solves in float64, one wavefront (64 lanes) per problem.
//
//  sv_kabsch_batched      <- utils/transformation.py:178-222 get_rigid_transform_3D (np.linalg.svd at :211)
//                            + :80-84 get_q_from_matrix (scipy Rotation.from_matrix().as_quat(), reordered wxyz)
//  sv_quat_avg_batched    <- utils/calibration.py:69-95 compute_quaternions_weighted_average (np.linalg.eig at :89)
//  sv_add_metric_batched  <- utils/metrics.py:139-150 compute_ADD_np
//
// The lanes of a wave split the point sums (centroids, the 3x3 cross-covariance H, the 4x4 quaternion moment matrix,
// the ADD distances); after the xor-butterfly every lane holds the full sums and runs the (tiny, wave-uniform) Jacobi
// iteration redundantly; lane 0 stores.  Latency-bound by construction: report problems/s, not a roofline.
#include "sv_common.h"
#include "sv_dense_math.h"

namespace sv {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// scipy Rotation.from_matrix -> as_quat (x,y,z,w), returned as (w,x,y,z)
__device__ __forceinline__ void quat_from_matrix(const double m[3][3], double q[4]) {
  const double tr = m[0][0] + m[1][1] + m[2][2];
  double d[4] = {m[0][0], m[1][1], m[2][2], tr};
  int choice = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (d[i] > d[choice]) choice = i;
  double x, y, z, w;
  if (choice == 3) {
    x = m[2][1] - m[1][2];
    y = m[0][2] - m[2][0];
    z = m[1][0] - m[0][1];
    w = 1.0 + tr;
  } else if (choice == 0) {
    x = 1.0 - tr + 2.0 * m[0][0];
    y = m[1][0] + m[0][1];
    z = m[2][0] + m[0][2];
    w = m[2][1] - m[1][2];
  } else if (choice == 1) {
    y = 1.0 - tr + 2.0 * m[1][1];
    z = m[2][1] + m[1][2];
    x = m[0][1] + m[1][0];
    w = m[0][2] - m[2][0];
  } else {
    z = 1.0 - tr + 2.0 * m[2][2];
    x = m[0][2] + m[2][0];
    y = m[1][2] + m[2][1];
    w = m[1][0] - m[0][1];
  }
  const double n = sqrt(x * x + y * y + z * z + w * w);
  q[0] = w / n;
  q[1] = x / n;
  q[2] = y / n;
  q[3] = z / n;
}

__global__ __launch_bounds__(256) void kabsch_kernel(const double* __restrict__ ref, const double* __restrict__ tgt,
                                                      const int32_t* __restrict__ Kp, int Kmax, int B,
                                                      double* __restrict__ Rout, double* __restrict__ tout,
                                                      double* __restrict__ qout) {
  const int prob = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (prob >= B) return;  // whole wave exits together
  const int lane = threadIdx.x & 63;
  const int K = Kp ? Kp[prob] : Kmax;
  const double* A = ref + (int64_t)prob * Kmax * 3;
  const double* Bm = tgt + (int64_t)prob * Kmax * 3;
  double sa[3] = {0, 0, 0}, sb[3] = {0, 0, 0};
  for (int i = lane; i < K; i += 64) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      sa[d] += A[i * 3 + d];
      sb[d] += Bm[i * 3 + d];
    }
  }
  double cA[3], cB[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    cA[d] = wave_sum(sa[d]) / (double)K;
    cB[d] = wave_sum(sb[d]) / (double)K;
  }
  double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int i = lane; i < K; i += 64) {
    double a[3], b[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      a[d] = A[i * 3 + d] - cA[d];
      b[d] = Bm[i * 3 + d] - cB[d];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) H[r][c] += a[r] * b[c];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) H[r][c] = wave_sum(H[r][c]);

  double R[3][3], t[3];
  kabsch_from_covariance(H, cA, cB, R, t);
  double q[4];
  quat_from_matrix(R, q);
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Rout[(int64_t)prob * 9 + r * 3 + c] = R[r][c];
#pragma unroll
    for (int r = 0; r < 3; ++r) tout[(int64_t)prob * 3 + r] = t[r];
    if (qout) {
#pragma unroll
      for (int r = 0; r < 4; ++r) qout[(int64_t)prob * 4 + r] = q[r];
    }
  }
}

// cyclic Jacobi for a symmetric 4x4: A <- J^T A J, E <- E J
__device__ __forceinline__ void jacobi_eig4(double A[4][4], double E[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) E[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 16; ++sweep) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        const double apq = A[p][q];
        if (fabs(apq) > 1e-300 && fabs(apq) > 1e-18 * (fabs(A[p][p]) + fabs(A[q][q]))) {
          const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
          const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(1.0 + theta * theta));
          const double c = 1.0 / sqrt(1.0 + t * t);
          const double s = t * c;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const double akp = A[k][p], akq = A[k][q];
            A[k][p] = c * akp - s * akq;
            A[k][q] = s * akp + c * akq;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const double apk = A[p][k], aqk = A[q][k];
            A[p][k] = c * apk - s * aqk;
            A[q][k] = s * apk + c * aqk;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const double ekp = E[k][p], ekq = E[k][q];
            E[k][p] = c * ekp - s * ekq;
            E[k][q] = s * ekp + c * ekq;
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void quat_avg_kernel(const double* __restrict__ Q, const double* __restrict__ w,
                                                        const int32_t* __restrict__ Mp, int Mmax, int B,
                                                        double* __restrict__ out) {
  const int prob = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (prob >= B) return;
  const int lane = threadIdx.x & 63;
  const int M = Mp ? Mp[prob] : Mmax;
  const double* Qp = Q + (int64_t)prob * Mmax * 4;
  const double* wp = w ? w + (int64_t)prob * Mmax : nullptr;
  double A[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) A[i][j] = 0.0;
  double wsum = 0.0;
  for (int m = lane; m < M; m += 64) {
    const double wm = wp ? wp[m] : 1.0;
    wsum += wm;
    double q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = Qp[m * 4 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) A[i][j] += wm * q[i] * q[j];
  }
  wsum = wave_sum(wsum);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) A[i][j] = wave_sum(A[i][j]) / wsum;
  double E[4][4];
  jacobi_eig4(A, E);
  int best = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (A[i][i] > A[best][best]) best = i;
  double v[4];
  double n = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // select column `best` without dynamic register indexing
    v[i] = best == 0 ? E[i][0] : best == 1 ? E[i][1] : best == 2 ? E[i][2] : E[i][3];
    n += v[i] * v[i];
  }
  n = sqrt(n);
  int big = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (fabs(v[i]) > fabs(v[big])) big = i;
  const double vb = big == 0 ? v[0] : big == 1 ? v[1] : big == 2 ? v[2] : v[3];
  const double sgn = vb < 0 ? -1.0 : 1.0;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[(int64_t)prob * 4 + i] = sgn * v[i] / n;
  }
}

// utils/transformation.py:16-60 get_quaternion_rotation_matrix(switch_w=False): q = (w,x,y,z), not normalised
__device__ __forceinline__ void rot_from_quat(const double* q, double R[3][3]) {
  const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
  R[0][0] = 2 * (q0 * q0 + q1 * q1) - 1;
  R[0][1] = 2 * (q1 * q2 - q0 * q3);
  R[0][2] = 2 * (q1 * q3 + q0 * q2);
  R[1][0] = 2 * (q1 * q2 + q0 * q3);
  R[1][1] = 2 * (q0 * q0 + q2 * q2) - 1;
  R[1][2] = 2 * (q2 * q3 - q0 * q1);
  R[2][0] = 2 * (q1 * q3 - q0 * q2);
  R[2][1] = 2 * (q2 * q3 + q0 * q1);
  R[2][2] = 2 * (q0 * q0 + q3 * q3) - 1;
}

__global__ __launch_bounds__(256) void add_metric_kernel(const double* __restrict__ points,
                                                          const int32_t* __restrict__ Pp, int Pmax,
                                                          const double* __restrict__ gt, const double* __restrict__ pr,
                                                          int B, double* __restrict__ out) {
  const int prob = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (prob >= B) return;
  const int lane = threadIdx.x & 63;
  const int P = Pp ? Pp[prob] : Pmax;
  const double* pts = points + (int64_t)prob * Pmax * 3;
  const double* g = gt + (int64_t)prob * 7;
  const double* r = pr + (int64_t)prob * 7;
  double Rg[3][3], Rp[3][3];
  rot_from_quat(g + 3, Rg);
  rot_from_quat(r + 3, Rp);
  double acc = 0.0;
  for (int i = lane; i < P; i += 64) {
    const double x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    double d2 = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double pg = Rg[a][0] * x + Rg[a][1] * y + Rg[a][2] * z + g[a];
      const double pp = Rp[a][0] * x + Rp[a][1] * y + Rp[a][2] * z + r[a];
      d2 += (pg - pp) * (pg - pp);
    }
    acc += sqrt(d2);
  }
  acc = wave_sum(acc);
  if (lane == 0) out[prob] = P > 0 ? acc / (double)P : 0.0;
}

}  // namespace sv

using namespace sv;

extern "C" {

int sv_kabsch_batched(const double* ref, const double* tgt, const int32_t* K, int Kmax, int B, double* R, double* t,
                      double* q_wxyz, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(B >= 0 && Kmax >= 1, "bad shape");
  if (B == 0) return SV_OK;
  SV_CHECK_ARG(ref && tgt && R && t, "null pointer");
  hipLaunchKernelGGL(kabsch_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, ref, tgt, K, Kmax, B, R, t,
                     q_wxyz);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_quat_avg_batched(const double* Q, const double* w, const int32_t* M, int Mmax, int B, double* out,
                        sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(B >= 0 && Mmax >= 1, "bad shape");
  if (B == 0) return SV_OK;
  SV_CHECK_ARG(Q && out, "null pointer");
  hipLaunchKernelGGL(quat_avg_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, Q, w, M, Mmax, B, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_add_metric_batched(const double* points, const int32_t* P, int Pmax, const double* gt_pose,
                          const double* pred_pose, int B, double* add_out, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(B >= 0 && Pmax >= 1, "bad shape");
  if (B == 0) return SV_OK;
  SV_CHECK_ARG(points && gt_pose && pred_pose && add_out, "null pointer");
  hipLaunchKernelGGL(add_metric_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, points, P, Pmax, gt_pose,
                     pred_pose, B, add_out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

}  // extern "C"
