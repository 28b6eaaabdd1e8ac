// A8: PointNet++ sampling / grouping for gfx950.
//
//  sv_fps        <- model/pointnet2_utils.py:65-86 farthest_point_sample (torch, sequential python loop of npoint
//                   kernel launches) and utils/data.py:13-34 get_farthest_point_sample_idx (numpy)
//  sv_ball_query <- model/pointnet2_utils.py:89-109 query_ball_point (full [B,S,N] distance matrix + sort)
//
// FPS: one workgroup per cloud, S sequential argmax rounds (first maximum wins, like numpy/torch argmax on CPU).
// Up to 16k points the cloud and the running min-distances live in registers and a round is a DPP argmax + one
// barrier (fps_reg_kernel); larger clouds keep the distances in LDS and re-read the points (fps_kernel).
// Ball query: one wavefront per query centre scans the cloud in index order, 64 points per step, and compacts the
// hits with ballot/popcount — no distance matrix, no sort.
#include "sv_common.h"

namespace sv {

constexpr int FPS_THREADS = 1024;

__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(const float* __restrict__ xyz, int N, int S,
                                                           const int64_t* __restrict__ start,
                                                           int64_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float dist[];  // [N]
  __shared__ float red_v[FPS_THREADS / 64];
  __shared__ int red_i[FPS_THREADS / 64];
  __shared__ int cur_s;
  const int b = blockIdx.x;
  const float* P = xyz + (int64_t)b * N * 3;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < N; i += FPS_THREADS) dist[i] = 1e10f;
  if (tid == 0) {
    int64_t s0 = start ? start[b] : 0;
    cur_s = (int)(s0 < 0 ? 0 : (s0 >= N ? N - 1 : s0));
  }
  __syncthreads();
  for (int it = 0; it < S; ++it) {
    const int cur = cur_s;
    if (tid == 0) out[(int64_t)b * S + it] = cur;
    const float cx = P[cur * 3], cy = P[cur * 3 + 1], cz = P[cur * 3 + 2];
    float best = -1.0f;
    int bi = 0x7fffffff;
    for (int i = tid; i < N; i += FPS_THREADS) {
      const float dx = P[i * 3] - cx, dy = P[i * 3 + 1] - cy, dz = P[i * 3 + 2] - cz;
      // (dx*dx + dy*dy) + dz*dz with separate roundings (torch.sum((xyz - c) ** 2, -1) in float32)
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      float m = dist[i];
      if (d < m) {
        m = d;
        dist[i] = d;
      }
      if (m > best) {  // strictly greater: the lowest index wins within a thread (i ascending)
        best = m;
        bi = i;
      }
    }
    // wave argmax, ties -> lowest index
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const float ov = __shfl_xor(best, d);
      const int oi = __shfl_xor(bi, d);
      if (ov > best || (ov == best && oi < bi)) {
        best = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      red_v[wid] = best;
      red_i[wid] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      float v = red_v[0];
      int ix = red_i[0];
      for (int w = 1; w < FPS_THREADS / 64; ++w) {
        if (red_v[w] > v || (red_v[w] == v && red_i[w] < ix)) {
          v = red_v[w];
          ix = red_i[w];
        }
      }
      cur_s = ix;
    }
    __syncthreads();
  }
}

// max of a 64-bit key over one row of 16 lanes (result in lane 15 of the row) and over the wave (result in lane 63):
// DPP row shifts / row broadcasts instead of ds_bpermute shuffles (a dozen dependent LDS-crossbar round trips per
// iteration were most of an FPS iteration).  Lanes without a source read the identity 0.
__device__ __forceinline__ unsigned long long dpp_max_step(unsigned long long k, unsigned long long o) {
  return o > k ? o : k;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_fetch(unsigned long long k) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)k, CTRL, ROW_MASK, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(k >> 32), CTRL, ROW_MASK, 0xf, false);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long row_max16(unsigned long long k) {
  k = dpp_max_step(k, dpp_fetch<0x111, 0xf>(k));  // row_shr:1
  k = dpp_max_step(k, dpp_fetch<0x112, 0xf>(k));  // row_shr:2
  k = dpp_max_step(k, dpp_fetch<0x114, 0xf>(k));  // row_shr:4
  k = dpp_max_step(k, dpp_fetch<0x118, 0xf>(k));  // row_shr:8
  return k;
}
__device__ __forceinline__ unsigned long long wave_max64(unsigned long long k) {
  k = row_max16(k);
  k = dpp_max_step(k, dpp_fetch<0x142, 0xa>(k));  // row_bcast:15 -> rows 1, 3
  k = dpp_max_step(k, dpp_fetch<0x143, 0xc>(k));  // row_bcast:31 -> rows 2, 3
  return k;  // lane 63
}
__device__ __forceinline__ unsigned long long read_lane64(unsigned long long k, int lane) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}

// Register-resident variant for clouds of at most PPT * 1024 points: coordinates and running min-distances stay in
// VGPRs (thread t owns points t, t + 1024, ...), an iteration touches no memory except the 16-entry cross-wave argmax
// exchange (double-buffered in LDS, so ONE barrier per iteration; every wave reduces the 16 entries redundantly and
// reads the winner's coordinates from an LDS copy of the cloud).
// Same arithmetic and tie rule as fps_kernel.
template <int PPT, bool LDS_XYZ>
__global__ __launch_bounds__(FPS_THREADS) void fps_reg_kernel(const float* __restrict__ xyz, int N, int S,
                                                               const int64_t* __restrict__ start,
                                                               int64_t* __restrict__ out) {
  constexpr int NW = FPS_THREADS / 64;
  static_assert(NW == 16, "the cross-wave reduction is one 16-lane row");
  __shared__ unsigned long long red_k[2][NW];
  extern __shared__ __attribute__((aligned(16))) float xyz_s[];  // [3 N] when LDS_XYZ
  const int b = blockIdx.x;
  const float* P = xyz + (int64_t)b * N * 3;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (LDS_XYZ) {
    for (int i = tid; i < 3 * N; i += FPS_THREADS) xyz_s[i] = P[i];
    __syncthreads();
  }
  float px[PPT], py[PPT], pz[PPT], dm[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int i = tid + j * FPS_THREADS;
    const bool ok = i < N;
    px[j] = ok ? P[i * 3] : 0.0f;
    py[j] = ok ? P[i * 3 + 1] : 0.0f;
    pz[j] = ok ? P[i * 3 + 2] : 0.0f;
    dm[j] = ok ? 1e10f : -2.0f;  // padding never wins the argmax (real distances are >= 0 > -1)
  }
  int64_t s0 = start ? start[b] : 0;
  int cur = (int)(s0 < 0 ? 0 : (s0 >= N ? N - 1 : s0));
  float cx = P[cur * 3], cy = P[cur * 3 + 1], cz = P[cur * 3 + 2];
  for (int it = 0; it < S; ++it) {
    if (tid == 0) out[(int64_t)b * S + it] = cur;
    float best = -1.0f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const float dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      if (d < dm[j]) dm[j] = d;
      if (dm[j] > best) {  // strictly greater: the lowest index wins within a thread (indices ascend with j)
        best = dm[j];
        bi = tid + j * FPS_THREADS;
      }
    }
    // argmax as the maximum of the key (distance bits, ~index): distances are >= 0, so their bit patterns order like
    // the floats, and among equal distances the lowest index has the largest key; padding (best < 0) maps to key 0
    unsigned long long key = best < 0.0f ? 0ull
                                         : (((unsigned long long)__float_as_uint(best)) << 32) | (unsigned)(~bi);
    key = read_lane64(wave_max64(key), 63);
    const int buf = it & 1;
    if (lane == 0) red_k[buf][wid] = key;
    __syncthreads();
    // every wave reduces the NW partial results itself (one row of 16 lanes)
    unsigned long long k2 = lane < NW ? red_k[buf][lane] : 0ull;
    k2 = read_lane64(row_max16(k2), 15);
    const int ix = (int)~(unsigned)k2;
    cur = __builtin_amdgcn_readfirstlane(ix);
    // coordinates of the winner: from the LDS copy of the cloud when it fits, else from global memory (cached)
    if (LDS_XYZ) {
      cx = xyz_s[cur * 3];
      cy = xyz_s[cur * 3 + 1];
      cz = xyz_s[cur * 3 + 2];
    } else {
      cx = P[cur * 3];
      cy = P[cur * 3 + 1];
      cz = P[cur * 3 + 2];
    }
  }
}

// 4 waves per block, one query centre per wave
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ xyz,
                                                          const float* __restrict__ new_xyz, int B, int N, int S,
                                                          float r2, int nsample, int64_t* __restrict__ out) {
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= (int64_t)B * S) return;
  const int lane = threadIdx.x & 63;
  const int b = (int)(q / S);
  const float* P = xyz + (int64_t)b * N * 3;
  const float qx = new_xyz[q * 3], qy = new_xyz[q * 3 + 1], qz = new_xyz[q * 3 + 2];
  const float qq = __fadd_rn(__fadd_rn(__fmul_rn(qx, qx), __fmul_rn(qy, qy)), __fmul_rn(qz, qz));
  int64_t* dst = out + q * nsample;
  int count = 0;
  int first = N;
  for (int base = 0; base < N && count < nsample; base += 64) {
    const int i = base + lane;
    bool hit = false;
    if (i < N) {
      const float x = P[i * 3], y = P[i * 3 + 1], z = P[i * 3 + 2];
      const float dot = __fadd_rn(__fadd_rn(__fmul_rn(qx, x), __fmul_rn(qy, y)), __fmul_rn(qz, z));
      const float pp = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
      // reference: dist = -2 * (q . p); dist += |q|^2; dist += |p|^2; keep when NOT (dist > r^2)
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(-2.0f, dot), qq), pp);
      hit = !(d > r2);
    }
    const unsigned long long m = __ballot(hit);
    if (m) {
      if (first == N) first = base + (int)__builtin_ctzll(m);
      const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
      if (hit && pos < nsample) dst[pos] = i;
      count += __popcll(m);
    }
  }
  if (count > nsample) count = nsample;
  for (int j = count + lane; j < nsample; j += 64) dst[j] = first;
}

// ---- 3-nearest-neighbour inverse-distance interpolation (PointNetFeaturePropagation.forward,
//      model/pointnet2_utils.py:298-305): a workgroup serves 64 query points.  Phase 1: one thread per query walks the S
//      source points (staged through LDS, 256 at a time) with the reference's expanded float32 distance
//      (-2 q.p + |q|^2) + |p|^2 and keeps the three smallest (ascending, first index wins a tie - the order of the
//      reference's full sort); weights 1 / (d + 1e-8) normalised.  Phase 2: all 256 threads write out[q][c], channel fastest.
__global__ __launch_bounds__(256) void three_nn_interpolate_kernel(const float* __restrict__ xyz1,
                                                                    const float* __restrict__ xyz2,
                                                                    const float* __restrict__ points2, int N, int S, int C,
                                                                    float* __restrict__ out) {
  __shared__ float src[256 * 3];
  __shared__ int nn_idx[64 * 3];
  __shared__ float nn_w[64 * 3];
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * 64;
  const float* x1 = xyz1 + (int64_t)b * N * 3;
  const float* x2 = xyz2 + (int64_t)b * S * 3;
  const int q = q0 + threadIdx.x;
  const bool active = threadIdx.x < 64 && q < N;
  float qx = 0.f, qy = 0.f, qz = 0.f, qq = 0.f;
  if (active) {
    qx = x1[q * 3 + 0];
    qy = x1[q * 3 + 1];
    qz = x1[q * 3 + 2];
    qq = (qx * qx + qy * qy) + qz * qz;
  }
  float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
  int i0 = 0, i1 = 0, i2 = 0;
  for (int s0 = 0; s0 < S; s0 += 256) {
    const int cnt = min(256, S - s0);
    __syncthreads();
    for (int e = threadIdx.x; e < cnt * 3; e += 256) src[e] = x2[(int64_t)s0 * 3 + e];
    __syncthreads();
    if (active) {
      for (int j = 0; j < cnt; ++j) {
        const float px = src[j * 3], py = src[j * 3 + 1], pz = src[j * 3 + 2];
        const float dot = (qx * px + qy * py) + qz * pz;
        const float pp = (px * px + py * py) + pz * pz;
        const float d = (-2.0f * dot + qq) + pp;
        const int i = s0 + j;
        if (d < d2) {
          if (d < d1) {
            d2 = d1; i2 = i1;
            if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = i; }
            else { d1 = d; i1 = i; }
          } else { d2 = d; i2 = i; }
        }
      }
    }
  }
  if (threadIdx.x < 64) {
    float w0 = 1.0f / (d0 + 1e-8f), w1 = 1.0f / (d1 + 1e-8f), w2 = 1.0f / (d2 + 1e-8f);
    const float ws = (w0 + w1) + w2;
    nn_idx[threadIdx.x * 3 + 0] = i0; nn_idx[threadIdx.x * 3 + 1] = i1; nn_idx[threadIdx.x * 3 + 2] = i2;
    nn_w[threadIdx.x * 3 + 0] = w0 / ws; nn_w[threadIdx.x * 3 + 1] = w1 / ws; nn_w[threadIdx.x * 3 + 2] = w2 / ws;
  }
  __syncthreads();
  const float* p2 = points2 + (int64_t)b * S * C;
  float* o = out + ((int64_t)b * N + q0) * C;
  const int nq = min(64, N - q0);
  for (int e = threadIdx.x; e < nq * C; e += 256) {
    const int ql = e / C, c = e - ql * C;
    const float v = (p2[(int64_t)nn_idx[ql * 3] * C + c] * nn_w[ql * 3] + p2[(int64_t)nn_idx[ql * 3 + 1] * C + c] * nn_w[ql * 3 + 1]) +
                    p2[(int64_t)nn_idx[ql * 3 + 2] * C + c] * nn_w[ql * 3 + 2];
    o[e] = v;
  }
}

}  // namespace sv

using namespace sv;

extern "C" {

int sv_three_nn_interpolate(const float* xyz1, const float* xyz2, const float* points2, int B, int N, int S, int C,
                            float* out, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(B >= 0 && N >= 1 && S >= 3 && C >= 1, "bad shape (S >= 3 source points)");
  if (B == 0) return SV_OK;
  SV_CHECK_ARG(xyz1 && xyz2 && points2 && out, "null pointer");
  hipLaunchKernelGGL(three_nn_interpolate_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)B), dim3(256), 0, stream, xyz1,
                     xyz2, points2, N, S, C, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_fps(const float* xyz, int B, int N, int S, const int64_t* start, int64_t* out, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(B >= 0 && N >= 1 && S >= 1, "bad shape");
  SV_CHECK_ARG((size_t)N * sizeof(float) <= 150 * 1024, "N too large for the LDS-resident distance array (38400)");
  if (B == 0) return SV_OK;
  SV_CHECK_ARG(xyz && out, "null pointer");
  if (N <= 16 * FPS_THREADS) {  // register-resident cloud
    static bool reg_attr_set = false;
    if (!reg_attr_set) {
      SV_HIP(hipFuncSetAttribute((const void*)fps_reg_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 150 * 1024));
      SV_HIP(hipFuncSetAttribute((const void*)fps_reg_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 150 * 1024));
      SV_HIP(hipFuncSetAttribute((const void*)fps_reg_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 150 * 1024));
      reg_attr_set = true;
    }
    const size_t xyz_bytes = (size_t)N * 3 * sizeof(float);
    const bool lds_xyz = xyz_bytes <= 150 * 1024;
    const dim3 g((unsigned)B), t(FPS_THREADS);
    if (N <= 4 * FPS_THREADS)
      hipLaunchKernelGGL((fps_reg_kernel<4, true>), g, t, xyz_bytes, stream, xyz, N, S, start, out);
    else if (N <= 8 * FPS_THREADS)
      hipLaunchKernelGGL((fps_reg_kernel<8, true>), g, t, xyz_bytes, stream, xyz, N, S, start, out);
    else if (lds_xyz)
      hipLaunchKernelGGL((fps_reg_kernel<16, true>), g, t, xyz_bytes, stream, xyz, N, S, start, out);
    else
      hipLaunchKernelGGL((fps_reg_kernel<16, false>), g, t, 0, stream, xyz, N, S, start, out);
    SV_LAUNCH_CHECK();
    return SV_OK;
  }
  static bool attr_set = false;
  if (!attr_set) {
    SV_HIP(hipFuncSetAttribute((const void*)fps_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(fps_kernel, dim3((unsigned)B), dim3(FPS_THREADS), (size_t)N * sizeof(float), stream, xyz, N, S,
                     start, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_ball_query(const float* xyz, const float* new_xyz, int B, int N, int S, double radius, int nsample, int64_t* out,
                  sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(B >= 0 && N >= 1 && S >= 1 && nsample >= 1, "bad shape");
  if (B == 0) return SV_OK;
  SV_CHECK_ARG(xyz && new_xyz && out, "null pointer");
  const float r2 = (float)(radius * radius);
  const int64_t nq = (int64_t)B * S;
  hipLaunchKernelGGL(ball_query_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, stream, xyz, new_xyz, B, N, S, r2,
                     nsample, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

}  // extern "C"
