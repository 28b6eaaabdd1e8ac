// Largest single-linkage cluster of the end-effector points on the device.
//
// Reference: utils/output.py:13-28 `ClusterUtil.get_largest_cluster` = sklearn AgglomerativeClustering(linkage="single",
// distance_threshold=0.06) + the most frequent label; call site app/inference_engine.py:422-433.  Single linkage cut at a
// distance threshold is exactly the connected components of the graph "distance < threshold", so no linkage tree is
// built: every pair (i, j < i) closer than the threshold is united in a lock-free union-find forest
// (hook the LARGER root under the smaller with a compare-and-swap, path halving on the way up), one workgroup per
// 256 x 256 block of the lower triangle with the j points staged through LDS.  parent[x] <= x always holds, so the
// forest is acyclic under any interleaving and the root a component ends up with is its smallest member: the result is
// deterministic although the order of the unions is not.  O(n^2 / 2) distance tests in float64
// (sqrt((dx*dx + dy*dy) + dz*dz) < dist, the host formula, no fma): 4 096 points 8.4 M pairs, 30 000 points 450 M -
// the reference's O(n^2) linkage takes seconds on the host for the same sets.
#include "sv_common.h"

namespace sv {
namespace {

constexpr int CT = 256;  // points per tile = threads per workgroup

// agent scope: the forest is shared by workgroups on all XCDs, whose L2s are not coherent for plain accesses - a find()
// that kept re-reading a stale "x is a root" while its compare-and-swap fails would never end
__device__ __forceinline__ int uf_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void uf_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int uf_find(int* parent, int x) {
  int p = uf_load(&parent[x]);
  while (p != x) {
    const int gp = uf_load(&parent[p]);
    if (gp != p) uf_store(&parent[x], gp);  // path halving: gp is an ancestor of x, gp <= p
    x = p;
    p = gp;
  }
  return x;
}

// unite the components of a and b, return the root of the result
__device__ __forceinline__ int uf_unite(int* parent, int a, int b) {
  for (;;) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return a;
    if (a < b) {
      const int t = a;
      a = b;
      b = t;
    }
    // a > b, both were roots a moment ago: hook a under b unless somebody hooked a first
    if (atomicCAS(&parent[a], a, b) == a) return b;
  }
}

template <typename T>
__device__ __forceinline__ void load_point(const T* xyz, int64_t ld, const int32_t* idx, int g, int n, double& x, double& y,
                                           double& z) {
  if (g < n) {
    const T* p = xyz + (int64_t)(idx ? idx[g] : g) * ld;
    x = (double)p[0];
    y = (double)p[1];
    z = (double)p[2];
  } else {
    x = y = z = 1e300;  // past the end: farther than any threshold from everything (the sum overflows to +inf)
  }
}

__global__ void iota_kernel(int* parent, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) parent[i] = i;
}

template <typename T>
__global__ __launch_bounds__(CT) void cluster_union_kernel(const T* xyz, int64_t ld, const int32_t* idx, int n, double dist,
                                                           int* parent) {
  // linear workgroup index -> (ti, tj) of the lower triangle, tj <= ti
  const long long b = blockIdx.x;
  int ti = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
  while ((long long)ti * (ti + 1) / 2 > b) --ti;
  while ((long long)(ti + 1) * (ti + 2) / 2 <= b) ++ti;
  const int tj = (int)(b - (long long)ti * (ti + 1) / 2);
  __shared__ double sx[CT], sy[CT], sz[CT];
  const int tid = threadIdx.x;
  load_point(xyz, ld, idx, tj * CT + tid, n, sx[tid], sy[tid], sz[tid]);
  double xi, yi, zi;
  const int ig = ti * CT + tid;
  load_point(xyz, ld, idx, ig, n, xi, yi, zi);
  __syncthreads();
  if (ig >= n) return;
  const int jend = (ti == tj) ? tid : CT;  // same tile: only j < i
  int ri = -1;                             // root of i as last seen (found lazily)
  for (int j = 0; j < jend; ++j) {
    const double dx = xi - sx[j], dy = yi - sy[j], dz = zi - sz[j];
    const double d = sqrt(dx * dx + dy * dy + dz * dz);
    if (d < dist) {
      const int jg = tj * CT + j;
      if (ri < 0) ri = uf_find(parent, ig);
      if (uf_load(&parent[jg]) != ri) ri = uf_unite(parent, ri, jg);
    }
  }
}

// root[i] = smallest member of i's component; counts[root] += 1
__global__ void cluster_flatten_kernel(int* parent, int n, int* root, int* counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int x = i, p = uf_load(&parent[x]);
  while (p != x) {
    x = p;
    p = uf_load(&parent[x]);
  }
  root[i] = x;
  atomicAdd(&counts[x], 1);
}

// best[0] = root of the largest component (ties: the smallest root), best[1] = its size.  One workgroup.
__global__ __launch_bounds__(1024) void cluster_argmax_kernel(const int* counts, int n, int* best) {
  __shared__ int s_cnt[1024], s_root[1024];
  int bc = -1, br = 0x7fffffff;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const int c = counts[i];
    if (c > bc) {  // i ascending per thread: the first maximum stays
      bc = c;
      br = i;
    }
  }
  s_cnt[threadIdx.x] = bc;
  s_root[threadIdx.x] = br;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const int oc = s_cnt[threadIdx.x + s], orr = s_root[threadIdx.x + s];
      if (oc > s_cnt[threadIdx.x] || (oc == s_cnt[threadIdx.x] && orr < s_root[threadIdx.x])) {
        s_cnt[threadIdx.x] = oc;
        s_root[threadIdx.x] = orr;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    best[0] = s_root[0];
    best[1] = s_cnt[0];
  }
}

// Ordered compaction: out = ascending i with v[i] == value (value read from value_dev[0] when given).  One workgroup
// walks the array in 1024-element strips (ballot + popcount ranks inside a wave, an LDS scan over the 16 waves).
template <typename V>
__global__ __launch_bounds__(1024) void select_equal_kernel(const V* v, int64_t n, long long value, const int* value_dev,
                                                            int64_t* out, int64_t* count) {
  __shared__ int wave_tot[16];
  __shared__ long long base_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (value_dev) value = value_dev[0];
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int64_t i0 = 0; i0 < n; i0 += 1024) {
    const int64_t i = i0 + tid;
    const bool hit = i < n && (long long)v[i] == value;
    const unsigned long long m = __ballot(hit);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[w] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int t = wave_tot[k];
      if (k < w) before += t;
      total += t;
    }
    const long long base = base_s;
    if (hit) out[base + before + rank] = i;
    __syncthreads();
    if (tid == 0) base_s = base + total;
    __syncthreads();
  }
  if (tid == 0) count[0] = base_s;
}

}  // namespace
}  // namespace sv

using namespace sv;

extern "C" size_t sv_cluster_workspace_bytes(int64_t n) {
  return 3 * align_up((size_t)(n > 0 ? n : 1) * sizeof(int), 256) + 256;  // parent, counts, best
}

extern "C" int sv_single_linkage_roots(const void* xyz, int elem_bytes, int64_t ld, const int32_t* idx, int64_t n, double dist,
                                       void* workspace, size_t workspace_bytes, int32_t* root, int32_t* best,
                                       sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(n >= 0 && n < (1ll << 31) - CT, "n out of range");
  SV_CHECK_ARG(elem_bytes == 4 || elem_bytes == 8, "xyz must be float32 or float64");
  SV_CHECK_ARG(ld >= 3, "row stride below 3");
  SV_CHECK_ARG(dist >= 0.0, "negative distance threshold");
  if (n == 0) return SV_OK;
  SV_CHECK_ARG(xyz && root && workspace, "null pointer");
  Workspace ws(workspace, workspace_bytes);
  int* parent = ws.take<int>((size_t)n);
  int* counts = ws.take<int>((size_t)n);
  int* best_ws = ws.take<int>(2);
  SV_CHECK_ARG(ws.ok, "workspace too small (sv_cluster_workspace_bytes)");
  const int ni = (int)n;
  hipLaunchKernelGGL(iota_kernel, dim3((ni + 255) / 256), dim3(256), 0, stream, parent, ni);
  SV_HIP(hipMemsetAsync(counts, 0, (size_t)n * sizeof(int), stream));
  const long long T = (ni + CT - 1) / CT, blocks = T * (T + 1) / 2;
  SV_CHECK_ARG(blocks < (1ll << 31), "too many points for one launch");
  if (elem_bytes == 4)
    hipLaunchKernelGGL(cluster_union_kernel<float>, dim3((unsigned)blocks), dim3(CT), 0, stream, (const float*)xyz, ld, idx, ni,
                       dist, parent);
  else
    hipLaunchKernelGGL(cluster_union_kernel<double>, dim3((unsigned)blocks), dim3(CT), 0, stream, (const double*)xyz, ld, idx, ni,
                       dist, parent);
  hipLaunchKernelGGL(cluster_flatten_kernel, dim3((ni + 255) / 256), dim3(256), 0, stream, parent, ni, root, counts);
  if (best) {
    hipLaunchKernelGGL(cluster_argmax_kernel, dim3(1), dim3(1024), 0, stream, counts, ni, best_ws);
    SV_HIP(hipMemcpyAsync(best, best_ws, 2 * sizeof(int), hipMemcpyDeviceToDevice, stream));
  }
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_select_equal(const void* v, int elem_bytes, int64_t n, int64_t value, const int32_t* value_dev, int64_t* out,
                               int64_t* count, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(n >= 0, "negative length");
  SV_CHECK_ARG(elem_bytes == 4 || elem_bytes == 8, "elements must be int32 or int64");
  SV_CHECK_ARG(count, "null pointer");
  SV_CHECK_ARG(n == 0 || (v && out), "null pointer");
  if (elem_bytes == 4)
    hipLaunchKernelGGL(select_equal_kernel<int32_t>, dim3(1), dim3(1024), 0, stream, (const int32_t*)v, n, (long long)value,
                       (const int*)value_dev, out, count);
  else
    hipLaunchKernelGGL(select_equal_kernel<int64_t>, dim3(1), dim3(1024), 0, stream, (const int64_t*)v, n, (long long)value,
                       (const int*)value_dev, out, count);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
