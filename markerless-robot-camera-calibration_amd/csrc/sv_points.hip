// A8: PointNet++ sampling / grouping for gfx950.
//
//  sv_fps        <- model/pointnet2_utils.py:65-86 farthest_point_sample (torch, sequential python loop of npoint
//                   kernel launches) and utils/data.py:13-34 get_farthest_point_sample_idx (numpy)
//  sv_ball_query <- model/pointnet2_utils.py:89-109 query_ball_point (full [B,S,N] distance matrix + sort)
//
// FPS: one workgroup per cloud; the running min-distance lives in LDS, each iteration is one pass over the cloud
// plus a wave-shuffle + LDS argmax (first maximum wins, like numpy/torch argmax on CPU).
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

}  // namespace sv

using namespace sv;

extern "C" {

int sv_fps(const float* xyz, int B, int N, int S, const int64_t* start, int64_t* out, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(B >= 0 && N >= 1 && S >= 1, "bad shape");
  SV_CHECK_ARG((size_t)N * sizeof(float) <= 150 * 1024, "N too large for the LDS-resident distance array (38400)");
  if (B == 0) return SV_OK;
  SV_CHECK_ARG(xyz && out, "null pointer");
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
