// Voxelisation, coordinate maps, kernel maps and conv plans for gfx950.
//
// Replaces the MinkowskiEngine coordinate manager the reference reaches through
// ME.TensorField(...).sparse() (app/inference_engine.py:405-415) and every ME.MinkowskiConvolution call
// (model/backbone/minkunet.py:125-187).  All of it is HBM/latency-bound integer work: one thread per
// point/voxel, coalesced row-id tables, wave ballot + popcount prefix sums for compaction, no float atomics.
#include <cstring>


#include "sv_common.h"

#include <stdarg.h>

namespace sv {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ------------------------------------------------------------------------------------------------
// order-preserving unique of sorted keys:  rank[j] = (number of segment heads in [0..j]) - 1
// ------------------------------------------------------------------------------------------------
constexpr int UB = 256;  // threads per block, one key per thread

__device__ __forceinline__ bool is_head(const uint64_t* __restrict__ keys, int64_t j, int64_t n, uint64_t keep_mask) {
  if (j >= n) return false;
  if (j == 0) return true;
  return (keys[j] & keep_mask) != (keys[j - 1] & keep_mask);
}

__global__ __launch_bounds__(UB) void unique_count_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                                           uint64_t keep_mask, int32_t* __restrict__ block_counts) {
  __shared__ int wave_cnt[UB / 64];
  int64_t j = (int64_t)blockIdx.x * UB + threadIdx.x;
  bool h = is_head(keys, j, n, keep_mask);
  unsigned long long b = __ballot(h);
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wid] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < UB / 64; ++w) s += wave_cnt[w];
    block_counts[blockIdx.x] = s;
  }
}

// single block: exclusive scan of block_counts in place, total to *total.  256 threads, so that the workgroup fits into
// the slot of any finishing convolution workgroup (a 1024-thread workgroup needs a whole drained CU)
constexpr int SCAN_T = 256;
__global__ __launch_bounds__(SCAN_T) void unique_scan_kernel(int32_t* __restrict__ block_counts, int nblocks,
                                                              int32_t* __restrict__ total) {
  __shared__ int wave_sum[SCAN_T / 64];
  __shared__ int carry_s;
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += SCAN_T) {
    int i = base + threadIdx.x;
    int v = (i < nblocks) ? block_counts[i] : 0;
    // inclusive scan within the wave
    int x = v;
    for (int d = 1; d < 64; d <<= 1) {
      int y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (lane == 63) wave_sum[wid] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wid; ++w) woff += wave_sum[w];
    int carry = carry_s;
    if (i < nblocks) block_counts[i] = carry + woff + x - v;
    __syncthreads();
    if (threadIdx.x == SCAN_T - 1) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry_s;
}

__global__ __launch_bounds__(UB) void unique_rank_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                                          uint64_t keep_mask, const int32_t* __restrict__ block_offsets,
                                                          int32_t* __restrict__ rank) {
  __shared__ int wave_cnt[UB / 64];
  int64_t j = (int64_t)blockIdx.x * UB + threadIdx.x;
  bool h = is_head(keys, j, n, keep_mask);
  unsigned long long b = __ballot(h);
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wid] = __popcll(b);
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wid; ++w) woff += wave_cnt[w];
  // heads at or before this lane
  unsigned long long le = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
  int incl = __popcll(b & le);
  if (j < n) rank[j] = block_offsets[blockIdx.x] + woff + incl - 1;
}

size_t unique_sorted_blocks(int64_t n) { return (size_t)((n + UB - 1) / UB); }

int launch_unique_sorted(const uint64_t* sorted_keys, int64_t n, uint64_t keep_mask, int32_t* rank,
                         int32_t* block_counts, int32_t* total, hipStream_t stream) {
  int nb = (int)unique_sorted_blocks(n);
  hipLaunchKernelGGL(unique_count_kernel, dim3(nb), dim3(UB), 0, stream, sorted_keys, n, keep_mask, block_counts);
  hipLaunchKernelGGL(unique_scan_kernel, dim3(1), dim3(SCAN_T), 0, stream, block_counts, nb, total);
  hipLaunchKernelGGL(unique_rank_kernel, dim3(nb), dim3(UB), 0, stream, sorted_keys, n, keep_mask, block_counts, rank);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------
// K1 voxelise
// ------------------------------------------------------------------------------------------------
template <bool IS_INT>
__global__ __launch_bounds__(256) void quantize_kernel(const void* __restrict__ coords4, int64_t n,
                                                        uint64_t* __restrict__ keys, int32_t* __restrict__ idx,
                                                        int32_t* __restrict__ counters) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int b, x, y, z;
  bool ok = true;
  if (IS_INT) {
    const int4 c = ((const int4*)coords4)[i];
    b = c.x; x = c.y; y = c.z; z = c.w;
  } else {
    const float4 c = ((const float4*)coords4)[i];
    const float lim = (float)SV_COORD_BIAS;
    ok = (c.x >= 0.0f) && (c.x < (float)SV_MAX_BATCH) && (fabsf(c.y) < lim) && (fabsf(c.z) < lim) &&
         (fabsf(c.w) < lim);  // also false for NaN
    b = ok ? (int)c.x : 0;
    x = ok ? (int)floorf(c.y) : 0;
    y = ok ? (int)floorf(c.z) : 0;
    z = ok ? (int)floorf(c.w) : 0;
  }
  ok = ok && coord_in_range(b, x, y, z);
  if (!ok) {
    atomicAdd(&counters[1], 1);
    b = x = y = z = 0;
  }
  keys[i] = make_key(b, x, y, z);
  idx[i] = (int32_t)i;
}

__global__ __launch_bounds__(256) void voxel_scatter_kernel(const uint64_t* __restrict__ skeys,
                                                             const int32_t* __restrict__ sidx,
                                                             const int32_t* __restrict__ rank, int64_t n,
                                                             uint64_t* __restrict__ keys_out,
                                                             int32_t* __restrict__ vcoords,
                                                             int64_t* __restrict__ inverse, int32_t* __restrict__ order,
                                                             int32_t* __restrict__ seg_start,
                                                             const int32_t* __restrict__ total) {
  int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  int v = rank[j];
  uint64_t k = skeys[j];
  int p = sidx[j];
  order[j] = p;
  inverse[p] = v;
  bool head = (j == 0) || (rank[j - 1] != v);
  if (head) {
    keys_out[v] = k;
    seg_start[v] = (int32_t)j;
    int b, x, y, z;
    decode_key(k, b, x, y, z);
    ((int4*)vcoords)[v] = make_int4(b, x, y, z);
  }
  if (j == n - 1) seg_start[*total] = (int32_t)n;
}

__global__ __launch_bounds__(256) void voxel_reduce_kernel(const float* __restrict__ feats, int C,
                                                            const int32_t* __restrict__ order,
                                                            const int32_t* __restrict__ seg_start, int64_t V, int mode,
                                                            float* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= V * C) return;
  int64_t v = t / C;
  int c = (int)(t - v * C);
  int s = seg_start[v], e = seg_start[v + 1];
  float acc = feats[(int64_t)order[s] * C + c];
  if (mode == SV_REDUCE_MEAN) {
    for (int j = s + 1; j < e; ++j) acc += feats[(int64_t)order[j] * C + c];
    acc = acc / (float)(e - s);
  }
  out[t] = acc;
}

// ------------------------------------------------------------------------------------------------
// hash table
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hash_insert_kernel(const uint64_t* __restrict__ keys, int64_t V,
                                                           uint64_t* __restrict__ tkeys, int32_t* __restrict__ tvals,
                                                           uint64_t cap_mask) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= V) return;
  uint64_t key = keys[i];
  uint64_t slot = hash64(key) & cap_mask;
  for (;;) {
    unsigned long long prev = atomicCAS((unsigned long long*)&tkeys[slot], (unsigned long long)SV_EMPTY_KEY,
                                        (unsigned long long)key);
    if (prev == SV_EMPTY_KEY || prev == key) {
      tvals[slot] = (int32_t)i;
      return;
    }
    slot = (slot + 1) & cap_mask;
  }
}

// ------------------------------------------------------------------------------------------------
// stride map
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stride_scatter_kernel(const uint64_t* __restrict__ keys_in,
                                                              const int32_t* __restrict__ rank, int64_t n,
                                                              uint64_t keep_mask, uint64_t* __restrict__ keys_out,
                                                              int32_t* __restrict__ vcoords_out,
                                                              int32_t* __restrict__ child_start,
                                                              const int32_t* __restrict__ total) {
  int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  int v = rank[j];
  bool head = (j == 0) || (rank[j - 1] != v);
  if (head) {
    uint64_t k = keys_in[j] & keep_mask;
    keys_out[v] = k;
    child_start[v] = (int32_t)j;
    int b, x, y, z;
    decode_key(k, b, x, y, z);
    ((int4*)vcoords_out)[v] = make_int4(b, x, y, z);
  }
  if (j == n - 1) child_start[*total] = (int32_t)n;
}

// ------------------------------------------------------------------------------------------------
// kernel maps
// ------------------------------------------------------------------------------------------------
// One THREAD per (voxel, offset) probe: blockIdx.y = offset k, so a workgroup resolves offset k for 256 consecutive voxels
// (coalesced coordinate reads and nbr[k][.] writes).  The probe of a neighbour is a dependent chain (hash -> slot -> maybe
// the next slot) that cannot be pipelined inside a thread; with one probe per thread the 27 V chains overlap through
// occupancy instead (2.4 M threads at the 2 cm room level), where the thread-per-voxel form of round 1 walked its 27
// chains one after the other on ~1.3 waves per SIMD (64 us alone on the GPU, ~320 us beside the convolutions).
__global__ __launch_bounds__(256) void kmap_k3_probe_kernel(const int32_t* __restrict__ vcoords, int64_t V, int step,
                                                             const uint64_t* __restrict__ tkeys,
                                                             const int32_t* __restrict__ tvals, uint64_t cap_mask,
                                                             int32_t* __restrict__ nbr, int64_t ld) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= V) return;
  const int k = blockIdx.y;
  const int dx = k % 3 - 1, dy = (k / 3) % 3 - 1, dz = k / 9 - 1;
  int r;
  if (k == 13) {
    r = (int)o;
  } else {
    const int4 c = ((const int4*)vcoords)[o];
    const int x = c.y + dx * step, y = c.z + dy * step, z = c.w + dz * step;
    r = coord_in_range(c.x, x, y, z) ? hash_find(tkeys, tvals, cap_mask, make_key(c.x, x, y, z)) : -1;
  }
  nbr[(int64_t)k * ld + o] = r;
}

// mask[o] = bit k set iff nbr[k][o] >= 0 (27 coalesced reads per voxel)
__global__ __launch_bounds__(256) void kmap_mask_kernel(const int32_t* __restrict__ nbr, int64_t ld, int K, int64_t V,
                                                         uint32_t* __restrict__ mask) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= V) return;
  uint32_t m = 0;
  for (int k = 0; k < K; ++k) m |= (nbr[(int64_t)k * ld + o] >= 0) ? (1u << k) : 0u;
  mask[o] = m;
}

__global__ __launch_bounds__(256) void kmap_down_kernel(const uint64_t* __restrict__ keys_fine,
                                                         const int32_t* __restrict__ parent, int64_t V_fine, int shift,
                                                         int32_t* __restrict__ nbr, int64_t ld,
                                                         uint32_t* __restrict__ mask) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= V_fine) return;
  int p = parent[i];
  int k = (int)((keys_fine[i] >> shift) & 7ull);
  nbr[(int64_t)k * ld + p] = (int32_t)i;
  atomicOr(&mask[p], 1u << k);
}

__global__ __launch_bounds__(256) void kmap_up_kernel(const uint64_t* __restrict__ keys_fine,
                                                       const int32_t* __restrict__ parent, int64_t V_fine, int shift,
                                                       int32_t* __restrict__ nbr, int64_t ld,
                                                       uint32_t* __restrict__ mask) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= V_fine) return;
  int p = parent[i];
  int cid = (int)((keys_fine[i] >> shift) & 7ull);
#pragma unroll
  for (int k = 0; k < 8; ++k) nbr[(int64_t)k * ld + i] = (k == cid) ? p : -1;
  mask[i] = 1u << cid;
}

// ------------------------------------------------------------------------------------------------
// conv plan
// ------------------------------------------------------------------------------------------------
// Sort key of a row's 27-bit neighbour mask: the RAREST offsets (the 8 corners, then the 12 edges, then the 6 faces, then
// the centre) become the most significant key bits, so rows that share their rare neighbours end up in the same 16-row
// MFMA sub-tile.  Measured on the Cfg-2 pyramid (tools/tile_experiment.py): useful row-slots 0.824 -> 0.854 at level 0,
// 0.912 -> 0.926 at level 1 versus sorting by the raw mask.  Row order never changes results, only skipped work.
struct KeyBits {
  unsigned char pos[27];
};
static KeyBits make_keybits() {
  KeyBits kb;
  int rank = 0;
  for (int cls = 3; cls >= 0; --cls)
    for (int k = 0; k < 27; ++k) {
      int dx = k % 3 - 1, dy = (k / 3) % 3 - 1, dz = k / 9 - 1;
      if (abs(dx) + abs(dy) + abs(dz) == cls) kb.pos[k] = (unsigned char)(26 - rank++);
    }
  return kb;
}

__global__ __launch_bounds__(256) void iota_key_kernel(const uint32_t* __restrict__ mask, int K, KeyBits kb,
                                                        int32_t* __restrict__ iota, uint32_t* __restrict__ key,
                                                        int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  iota[i] = (int32_t)i;
  uint32_t m = mask[i];
  if (K == 27) {
    uint32_t k2 = 0;
#pragma unroll
    for (int k = 0; k < 27; ++k) k2 |= ((m >> k) & 1u) << kb.pos[k];
    m = k2;
  }
  // Sort by the rank of the mask in reflected-Gray order rather than by its binary value: neighbours in the sorted order
  // then differ in fewer offsets, so a 16-row sub-tile's union of offsets is smaller (slot efficiency 0.854 -> 0.872 on
  // the 2 cm room level; tools/tile_experiment.py).
  m ^= m >> 1;
  m ^= m >> 2;
  m ^= m >> 4;
  m ^= m >> 8;
  m ^= m >> 16;
  key[i] = m;
}

// block = 128 threads = one tile of rows, blockIdx.y = offset k
__global__ __launch_bounds__(128) void plan_gather_kernel(const int32_t* __restrict__ nbr, int64_t ld,
                                                           const int32_t* __restrict__ perm_sorted, int64_t V,
                                                           int64_t Vpad, int K, int32_t nbr_base,
                                                           int32_t* __restrict__ perm,
                                                           int32_t* __restrict__ nbr_s,
                                                           uint32_t* __restrict__ submask) {
  int tile = blockIdx.x, k = blockIdx.y;
  int64_t r = (int64_t)tile * SV_TILE_ROWS + threadIdx.x;
  int o = (r < V) ? perm_sorted[r] : -1;
  if (k == 0) perm[r] = o;
  int n = (o >= 0) ? nbr[(int64_t)k * ld + o] : -1;
  if (n >= 0) n -= nbr_base;  // plans of a row range (sv_plan_build: nbr_base): indices relative to the range's first input row
  nbr_s[(int64_t)k * Vpad + r] = n;
  unsigned long long b = __ballot(n >= 0);
  int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    uint32_t bits = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if ((b >> (16 * g)) & 0xffffull) bits |= 1u << g;
    if (bits) atomicOr(&submask[(int64_t)tile * K + k], bits << (4 * wid));
  }
}

// work of a plan tile = active (offset, sub-tile) slots; sort key ascending = work descending
__global__ __launch_bounds__(256) void tile_cost_kernel(const uint32_t* __restrict__ submask, int K, int64_t tiles,
                                                         uint32_t* __restrict__ key, int32_t* __restrict__ idx) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= tiles) return;
  uint32_t cost = 0;
  for (int k = 0; k < K; ++k) cost += __popc(submask[t * K + k]);
  key[t] = 255u - min(cost, 255u);  // cost <= 27 * 8 = 216: one 8-bit pass
  (void)idx;
}

}  // namespace sv

using namespace sv;

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

const char* sv_last_error(void) { return sv::g_err; }
int sv_abi_version(void) { return SV_ABI_VERSION; }

static size_t sort_pairs_u64_temp_bytes(int64_t n) { return radix_sort_temp_bytes(n, sizeof(uint64_t)); }
static size_t sort_pairs_u32_temp_bytes(int64_t n) { return radix_sort_temp_bytes(n, sizeof(uint32_t)); }

size_t sv_voxelize_workspace_bytes(int64_t N) {
  if (N <= 0) return 256;
  size_t n = (size_t)N;
  size_t total = 0;
  total += align_up(n * 8, 256) * 2;  // keys in / sorted
  total += align_up(n * 4, 256) * 3;  // idx in / sorted, rank
  total += align_up(unique_sorted_blocks(N) * 4 + 4, 256);
  total += align_up(sort_pairs_u64_temp_bytes(N), 256);
  return total + 4096;
}

int sv_voxelize(const void* coords4, int coords_are_int, int64_t N, void* workspace, size_t workspace_bytes,
                uint64_t* keys, int32_t* vcoords, int64_t* inverse, int32_t* order, int32_t* seg_start,
                int32_t* counters, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(N >= 0 && N < (1ll << 31) - 1024, "N out of range");
  SV_CHECK_ARG(counters != nullptr, "counters is null");
  SV_HIP(hipMemsetAsync(counters, 0, 4 * sizeof(int32_t), stream));
  if (N == 0) return SV_OK;
  SV_CHECK_ARG(coords4 && keys && vcoords && inverse && order && seg_start && workspace, "null pointer");
  Workspace ws(workspace, workspace_bytes);
  uint64_t* k_in = ws.take<uint64_t>(N);
  uint64_t* k_sorted = ws.take<uint64_t>(N);
  int32_t* i_in = ws.take<int32_t>(N);
  int32_t* i_sorted = ws.take<int32_t>(N);
  int32_t* rank = ws.take<int32_t>(N);
  int32_t* blocks = ws.take<int32_t>(unique_sorted_blocks(N) + 1);
  size_t sort_bytes = sort_pairs_u64_temp_bytes(N);
  char* sort_tmp = ws.take<char>(sort_bytes);
  if (!ws.ok) {
    set_error("sv_voxelize: workspace too small (%zu given)", workspace_bytes);
    return SV_ERR_WORKSPACE;
  }
  int nb = (int)((N + 255) / 256);
  if (coords_are_int)
    hipLaunchKernelGGL(quantize_kernel<true>, dim3(nb), dim3(256), 0, stream, coords4, N, k_in, i_in, counters);
  else
    hipLaunchKernelGGL(quantize_kernel<false>, dim3(nb), dim3(256), 0, stream, coords4, N, k_in, i_in, counters);
  SV_LAUNCH_CHECK();
  // stable sort by the whole 64-bit key (batch | Morton); values = point indices (identity, generated by the sort)
  int rc = radix_sort_pairs<uint64_t>(k_in, nullptr, k_sorted, i_sorted, N, 0, 64, sort_tmp, sort_bytes, stream);
  if (rc) return rc;
  rc = launch_unique_sorted(k_sorted, N, ~0ull, rank, blocks, &counters[0], stream);
  if (rc) return rc;
  hipLaunchKernelGGL(voxel_scatter_kernel, dim3(nb), dim3(256), 0, stream, k_sorted, i_sorted, rank, N, keys, vcoords,
                     inverse, order, seg_start, &counters[0]);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_voxel_reduce(const float* feats, int C, const int32_t* order, const int32_t* seg_start, int64_t V, int mode,
                    float* out, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(C > 0 && V >= 0, "bad shape");
  SV_CHECK_ARG(mode == SV_REDUCE_MEAN || mode == SV_REDUCE_FIRST, "bad mode");
  if (V == 0) return SV_OK;
  SV_CHECK_ARG(feats && order && seg_start && out, "null pointer");
  int64_t total = V * C;
  hipLaunchKernelGGL(voxel_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, feats, C, order,
                     seg_start, V, mode, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_hash_build(const uint64_t* keys, int64_t V, uint64_t* table_keys, int32_t* table_vals, int64_t capacity,
                  sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(V >= 0 && capacity >= 2 && (capacity & (capacity - 1)) == 0, "capacity must be a power of two");
  SV_CHECK_ARG(capacity >= 2 * V, "capacity must be >= 2*V");
  SV_CHECK_ARG(table_keys && table_vals, "null pointer");
  SV_HIP(hipMemsetAsync(table_keys, 0xff, (size_t)capacity * sizeof(uint64_t), stream));
  if (V == 0) return SV_OK;
  SV_CHECK_ARG(keys, "null pointer");
  hipLaunchKernelGGL(hash_insert_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, stream, keys, V, table_keys,
                     table_vals, (uint64_t)(capacity - 1));
  SV_LAUNCH_CHECK();
  return SV_OK;
}

size_t sv_stride_map_workspace_bytes(int64_t V_in) {
  if (V_in <= 0) return 256;
  return align_up((size_t)V_in * 4, 256) + align_up(unique_sorted_blocks(V_in) * 4 + 4, 256) + 4096;
}

int sv_stride_map(const uint64_t* keys_in, int64_t V_in, int level, void* workspace, size_t workspace_bytes,
                  uint64_t* keys_out, int32_t* vcoords_out, int32_t* parent, int32_t* child_start, int32_t* counters,
                  sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(level >= 0 && level < SV_COORD_BITS - 1, "level out of range");
  SV_CHECK_ARG(counters != nullptr, "counters is null");
  SV_HIP(hipMemsetAsync(counters, 0, 4 * sizeof(int32_t), stream));
  if (V_in == 0) return SV_OK;
  SV_CHECK_ARG(keys_in && keys_out && vcoords_out && parent && child_start && workspace, "null pointer");
  Workspace ws(workspace, workspace_bytes);
  int32_t* blocks = ws.take<int32_t>(unique_sorted_blocks(V_in) + 1);
  if (!ws.ok) {
    set_error("sv_stride_map: workspace too small");
    return SV_ERR_WORKSPACE;
  }
  uint64_t keep = ~((1ull << (3 * (level + 1))) - 1ull);
  int rc = launch_unique_sorted(keys_in, V_in, keep, parent, blocks, &counters[0], stream);
  if (rc) return rc;
  hipLaunchKernelGGL(stride_scatter_kernel, dim3((unsigned)((V_in + 255) / 256)), dim3(256), 0, stream, keys_in, parent,
                     V_in, keep, keys_out, vcoords_out, child_start, &counters[0]);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_kernel_map_k3(const int32_t* vcoords, int64_t V, int tensor_stride, int dilation, const uint64_t* table_keys,
                     const int32_t* table_vals, int64_t capacity, int32_t* nbr, int64_t ld, uint32_t* mask,
                     sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(V >= 0 && ld >= V && tensor_stride > 0 && dilation > 0, "bad shape");
  SV_CHECK_ARG(capacity >= 2 && (capacity & (capacity - 1)) == 0, "capacity must be a power of two");
  if (V == 0) return SV_OK;
  SV_CHECK_ARG(vcoords && table_keys && table_vals && nbr && mask, "null pointer");
  hipLaunchKernelGGL(kmap_k3_probe_kernel, dim3((unsigned)((V + 255) / 256), 27), dim3(256), 0, stream, vcoords, V,
                     tensor_stride * dilation, table_keys, table_vals, (uint64_t)(capacity - 1), nbr, ld);
  hipLaunchKernelGGL(kmap_mask_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, stream, nbr, ld, 27, V, mask);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_kernel_map_down(const uint64_t* keys_fine, const int32_t* parent, int64_t V_fine, int level, int64_t V_coarse,
                       int32_t* nbr, int64_t ld, uint32_t* mask, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(V_fine >= 0 && V_coarse >= 0 && ld >= V_coarse && level >= 0, "bad shape");
  if (V_coarse == 0) return SV_OK;
  SV_CHECK_ARG(keys_fine && parent && nbr && mask, "null pointer");
  SV_HIP(hipMemsetAsync(nbr, 0xff, (size_t)8 * ld * sizeof(int32_t), stream));
  SV_HIP(hipMemsetAsync(mask, 0, (size_t)V_coarse * sizeof(uint32_t), stream));
  hipLaunchKernelGGL(kmap_down_kernel, dim3((unsigned)((V_fine + 255) / 256)), dim3(256), 0, stream, keys_fine, parent,
                     V_fine, 3 * level, nbr, ld, mask);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int sv_kernel_map_up(const uint64_t* keys_fine, const int32_t* parent, int64_t V_fine, int level, int32_t* nbr,
                     int64_t ld, uint32_t* mask, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(V_fine >= 0 && ld >= V_fine && level >= 0, "bad shape");
  if (V_fine == 0) return SV_OK;
  SV_CHECK_ARG(keys_fine && parent && nbr && mask, "null pointer");
  hipLaunchKernelGGL(kmap_up_kernel, dim3((unsigned)((V_fine + 255) / 256)), dim3(256), 0, stream, keys_fine, parent,
                     V_fine, 3 * level, nbr, ld, mask);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

size_t sv_plan_workspace_bytes(int64_t V) {
  if (V <= 0) return 256;
  return align_up((size_t)V * 4, 256) * 4 + align_up(sort_pairs_u32_temp_bytes(V), 256) * 2 + 8192 +
         align_up((size_t)(V / 128 + 2) * 4, 256) * 3;
}

int sv_plan_build(const int32_t* nbr, int64_t ld, const uint32_t* mask, int K, int64_t V, int64_t nbr_base,
                  void* workspace, size_t workspace_bytes, int32_t* perm, int32_t* nbr_s, uint32_t* submask,
                  int32_t* tile_order, int64_t Vpad, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(K >= 1 && K <= 32, "K must be in 1..32");
  SV_CHECK_ARG(V >= 0 && ld >= V, "bad shape");
  SV_CHECK_ARG(nbr_base >= 0 && nbr_base < (1ll << 31), "nbr_base out of range");
  SV_CHECK_ARG(Vpad % SV_TILE_ROWS == 0 && Vpad >= V, "Vpad must be a multiple of 128 >= V");
  if (V == 0) return SV_OK;
  SV_CHECK_ARG(nbr && mask && perm && nbr_s && submask && workspace, "null pointer");
  Workspace ws(workspace, workspace_bytes);
  int32_t* iota = ws.take<int32_t>(V);
  int32_t* sorted_rows = ws.take<int32_t>(V);
  uint32_t* sorted_mask = ws.take<uint32_t>(V);
  uint32_t* key = ws.take<uint32_t>(V);
  size_t sort_bytes = sort_pairs_u32_temp_bytes(V);
  char* sort_tmp = ws.take<char>(sort_bytes);
  const int64_t ntiles = Vpad / SV_TILE_ROWS;
  uint32_t* tkey = ws.take<uint32_t>(ntiles);
  uint32_t* tkey_sorted = ws.take<uint32_t>(ntiles);
  int32_t* tidx = ws.take<int32_t>(ntiles);
  size_t tsort_bytes = sort_pairs_u32_temp_bytes(ntiles);
  char* tsort_tmp = ws.take<char>(tsort_bytes);
  if (!ws.ok) {
    set_error("sv_plan_build: workspace too small");
    return SV_ERR_WORKSPACE;
  }
  static const KeyBits keybits = make_keybits();
  hipLaunchKernelGGL(iota_key_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, stream, mask, K, keybits, iota,
                     key, V);
  SV_LAUNCH_CHECK();
  int rc = radix_sort_pairs<uint32_t>(key, nullptr, sorted_mask, sorted_rows, V, 0, K, sort_tmp, sort_bytes, stream);
  if (rc) return rc;
  int64_t tiles = Vpad / SV_TILE_ROWS;
  SV_HIP(hipMemsetAsync(submask, 0, (size_t)tiles * K * sizeof(uint32_t), stream));
  hipLaunchKernelGGL(plan_gather_kernel, dim3((unsigned)tiles, (unsigned)K), dim3(128), 0, stream, nbr, ld, sorted_rows,
                     V, Vpad, K, (int32_t)nbr_base, perm, nbr_s, submask);
  SV_LAUNCH_CHECK();
  if (tile_order) {
    hipLaunchKernelGGL(tile_cost_kernel, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, stream, submask, K,
                       ntiles, tkey, tidx);
    SV_LAUNCH_CHECK();
    rc = radix_sort_pairs<uint32_t>(tkey, nullptr, tkey_sorted, tile_order, ntiles, 0, 8, tsort_tmp, tsort_bytes, stream);
    if (rc) return rc;
  }
  return SV_OK;
}

}  // extern "C"
