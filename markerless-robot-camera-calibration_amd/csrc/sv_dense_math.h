// Shared fp64 device math for the small dense solves (sv_dense.hip, sv_icp.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace sv {

// One-sided (Hestenes) Jacobi SVD of a 3x3 matrix: on exit G = H*V has orthogonal columns.
__device__ __forceinline__ void jacobi_svd3(double G[3][3], double V[3][3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = (pq == 2) ? 1 : 0;
      const int q = (pq == 0) ? 1 : 2;
      double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        alpha += G[i][p] * G[i][p];
        beta += G[i][q] * G[i][q];
        gamma += G[i][p] * G[i][q];
      }
      if (fabs(gamma) > 1e-300 && fabs(gamma) > 1e-17 * sqrt(alpha * beta)) {
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t);
        const double s = c * t;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double gp = G[i][p], gq = G[i][q];
          G[i][p] = c * gp - s * gq;
          G[i][q] = s * gp + c * gq;
          const double vp = V[i][p], vq = V[i][q];
          V[i][p] = c * vp - s * vq;
          V[i][q] = s * vp + c * vq;
        }
      }
    }
  }
}

__device__ __forceinline__ void swap_cols(double M[3][3], int a, int b) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double t = M[i][a];
    M[i][a] = M[i][b];
    M[i][b] = t;
  }
}

__device__ __forceinline__ double det3(const double M[3][3]) {
  return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
         M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}


// Kabsch from the 3x3 cross-covariance H = sum (a - cA)(b - cB)^T: R = V diag(1,1,det V) U^T with H = U S V^T
// (utils/transformation.py:203-220: R = Vt.T @ U.T, third row of Vt negated when det R < 0), t = cB - R cA.
__device__ __forceinline__ void kabsch_from_covariance(const double H[3][3], const double cA[3], const double cB[3],
                                                        double R[3][3], double t[3]) {
  double G[3][3], V[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) G[r][c] = H[r][c];
  jacobi_svd3(G, V);
  double sg[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) sg[j] = sqrt(G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j]);
#define SV_CSWAP(a, b)          \
  if (sg[a] < sg[b]) {          \
    double t_ = sg[a];          \
    sg[a] = sg[b];              \
    sg[b] = t_;                 \
    swap_cols(G, a, b);         \
    swap_cols(V, a, b);         \
  }
  SV_CSWAP(0, 1)
  SV_CSWAP(1, 2)
  SV_CSWAP(0, 1)
#undef SV_CSWAP
  double U[3][3];
  const double tiny = 1e-300;
  const double s0 = sg[0] > tiny ? sg[0] : 1.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) U[i][0] = sg[0] > tiny ? G[i][0] / s0 : (i == 0 ? 1.0 : 0.0);
  double g1[3], dot = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) dot += G[i][1] * U[i][0];
  double n1 = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    g1[i] = G[i][1] - dot * U[i][0];
    n1 += g1[i] * g1[i];
  }
  n1 = sqrt(n1);
  if (n1 > 1e-14 * s0 && n1 > tiny) {
#pragma unroll
    for (int i = 0; i < 3; ++i) U[i][1] = g1[i] / n1;
  } else {  // rank-1 H (collinear points): any unit vector orthogonal to U0
    int m = 0;
    if (fabs(U[1][0]) < fabs(U[m][0])) m = 1;
    if (fabs(U[2][0]) < fabs(U[m][0])) m = 2;
    double e[3] = {0, 0, 0};
    e[m] = 1.0;
    const double d2 = U[m][0];
    double nn = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      g1[i] = e[i] - d2 * U[i][0];
      nn += g1[i] * g1[i];
    }
    nn = sqrt(nn);
#pragma unroll
    for (int i = 0; i < 3; ++i) U[i][1] = g1[i] / nn;
  }
  U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
  U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
  U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
  const double dV = det3(V) < 0 ? -1.0 : 1.0;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) R[r][c] = V[r][0] * U[c][0] + V[r][1] * U[c][1] + dV * V[r][2] * U[c][2];
#pragma unroll
  for (int r = 0; r < 3; ++r) t[r] = -(R[r][0] * cA[0] + R[r][1] * cA[1] + R[r][2] * cA[2]) + cB[r];
}

}  // namespace sv
