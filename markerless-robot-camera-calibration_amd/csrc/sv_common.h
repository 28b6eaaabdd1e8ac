// Internal helpers shared by the HIP translation units of libsvhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "sv_hip.h"

namespace sv {

void set_error(const char* fmt, ...);

#define SV_CHECK_ARG(cond, msg)                                  \
  do {                                                           \
    if (!(cond)) {                                               \
      sv::set_error("%s: %s", __func__, msg);                    \
      return SV_ERR_INVALID;                                     \
    }                                                            \
  } while (0)

#define SV_HIP(expr)                                                                   \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess) {                                                           \
      sv::set_error("%s: %s -> %s", __func__, #expr, hipGetErrorString(e__));          \
      return SV_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

#define SV_LAUNCH_CHECK() SV_HIP(hipGetLastError())

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's workspace.
struct Workspace {
  char* base;
  size_t size;
  size_t off = 0;
  bool ok = true;
  Workspace(void* p, size_t n) : base((char*)p), size(n) {}
  template <typename T>
  T* take(size_t count) {
    size_t o = align_up(off, 256);
    size_t bytes = count * sizeof(T);
    if (o + bytes > size) {
      ok = false;
      return nullptr;
    }
    off = o + bytes;
    return (T*)(base + o);
  }
};

// ---- key packing -------------------------------------------------------------------------------
// spread the low 18 bits of v so that bit j lands on bit 3j
__host__ __device__ __forceinline__ uint64_t part1by2(uint64_t v) {
  v &= 0x3ffffull;  // 18 bits
  v = (v | (v << 32)) & 0x001f00000000ffffull;
  v = (v | (v << 16)) & 0x001f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
__host__ __device__ __forceinline__ uint64_t compact1by2(uint64_t v) {
  v &= 0x1249249249249249ull;
  v = (v ^ (v >> 2)) & 0x10c30c30c30c30c3ull;
  v = (v ^ (v >> 4)) & 0x100f00f00f00f00full;
  v = (v ^ (v >> 8)) & 0x001f0000ff0000ffull;
  v = (v ^ (v >> 16)) & 0x001f00000000ffffull;
  v = (v ^ (v >> 32)) & 0x3ffffull;
  return v;
}
__host__ __device__ __forceinline__ bool coord_in_range(int b, int x, int y, int z) {
  return (unsigned)b < SV_MAX_BATCH && (unsigned)(x + SV_COORD_BIAS) < (1u << SV_COORD_BITS) &&
         (unsigned)(y + SV_COORD_BIAS) < (1u << SV_COORD_BITS) && (unsigned)(z + SV_COORD_BIAS) < (1u << SV_COORD_BITS);
}
__host__ __device__ __forceinline__ uint64_t make_key(int b, int x, int y, int z) {
  return ((uint64_t)b << 54) | part1by2((uint64_t)(x + SV_COORD_BIAS)) | (part1by2((uint64_t)(y + SV_COORD_BIAS)) << 1) |
         (part1by2((uint64_t)(z + SV_COORD_BIAS)) << 2);
}
__host__ __device__ __forceinline__ void decode_key(uint64_t key, int& b, int& x, int& y, int& z) {
  b = (int)(key >> 54);
  x = (int)compact1by2(key) - SV_COORD_BIAS;
  y = (int)compact1by2(key >> 1) - SV_COORD_BIAS;
  z = (int)compact1by2(key >> 2) - SV_COORD_BIAS;
}
#define SV_EMPTY_KEY 0xffffffffffffffffull

__host__ __device__ __forceinline__ uint64_t hash64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}

__device__ __forceinline__ int hash_find(const uint64_t* __restrict__ tkeys, const int32_t* __restrict__ tvals,
                                         uint64_t cap_mask, uint64_t key) {
  uint64_t slot = hash64(key) & cap_mask;
  // the table is at most half full, so a probe sequence always terminates at an empty slot
  for (;;) {
    uint64_t k = tkeys[slot];
    if (k == key) return tvals[slot];
    if (k == SV_EMPTY_KEY) return -1;
    slot = (slot + 1) & cap_mask;
  }
}

// Order-preserving stream compaction of head flags (ballot + popcount prefix, no atomics):
//  compact_count  : per-block number of set flags
//  compact_offsets: exclusive scan of the block counts (single block) and total
// are in sv_coords.hip; used by voxelise and stride maps.
int launch_unique_sorted(const uint64_t* sorted_keys, int64_t n, uint64_t clear_mask, int32_t* rank /*[n]*/,
                         int32_t* block_counts, int32_t* total /*device int32*/, hipStream_t stream);
size_t unique_sorted_blocks(int64_t n);

// Stable LSD radix sort of (key, int32 value) pairs by key bits [begin_bit, end_bit) (sv_sort.hip).  vals_in == nullptr
// sorts the identity permutation.  Outputs must not alias the inputs; temp holds radix_sort_temp_bytes(n, sizeof(KeyT)).
size_t radix_sort_temp_bytes(int64_t n, size_t key_bytes);
template <typename KeyT>
int radix_sort_pairs(const KeyT* keys_in, const int32_t* vals_in, KeyT* keys_out, int32_t* vals_out, int64_t n,
                     int begin_bit, int end_bit, void* temp, size_t temp_bytes, hipStream_t stream);

}  // namespace sv
