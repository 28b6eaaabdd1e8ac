// Frame-level composites of the coordinate work: ONE host call per phase instead of ~200 (gfx950).
//
// What MinkowskiEngine's coordinate manager does implicitly behind ME.TensorField(...).sparse() and every
// ME.MinkowskiConvolution of a MinkUNet (app/inference_engine.py:405-415, model/backbone/minkunet.py:125-183) is, per
// frame: voxelise, the stride-2 maps of the pyramid, a hash table + 27-offset kernel map + conv plan per level, the
// stride-2 down / transposed up maps + plans between neighbouring levels, and the offset-range plans of the wide decoder
// layers.  Driven call by call from Python that is ~200 kernel launches behind ~120 ctypes calls and ~60 tensor
// allocations: 2.2 ms of host time for 1.2 ms of kernels, with five size read-backs in between.  For the reference's
// consumer - one InferenceEngine.predict(data) per frame, app/main.py:432-456 - that host time is frame latency.
//   sv_frame_maps : voxelise + the stride-2 maps; the level sizes are read back here (the only host synchronisations of a
//                   frame's coordinate work: they wait for this phase's own small kernels and nothing else)
//   sv_frame_plans: everything whose size follows from those level sizes - no synchronisation at all
// Both carve their outputs from caller-supplied arenas and describe them in a host-side layout table; the kernels are
// the ones behind the piecewise entry points (sv_voxelize ... sv_plan_build), so every array has the same content.
#include "sv_common.h"

namespace sv {

__global__ __launch_bounds__(256) void mask_range_kernel(const uint32_t* __restrict__ mask, int64_t V, int k0, int nk,
                                                          uint32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < V) out[i] = (mask[i] >> k0) & ((1u << nk) - 1u);
}

static inline int64_t next_pow2_i64(int64_t x) {
  int64_t p = 1;
  while (p < x) p <<= 1;
  return p;
}
static inline int64_t round_up_i64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// bytes of one plan's arrays (perm, nbr_s, submask, tile_order) inside an arena, alignment padding included
static size_t plan_bytes(int64_t V, int K) {
  const int64_t Vpad = round_up_i64(V > 1 ? V : 1, SV_TILE_ROWS);
  const int64_t tiles = Vpad / SV_TILE_ROWS;
  return align_up((size_t)Vpad * 4, 256) + align_up((size_t)K * Vpad * 4, 256) + align_up((size_t)tiles * K * 4, 256) +
         align_up((size_t)tiles * 4, 256);
}

struct PlanOut {
  int32_t* perm;
  int32_t* nbr_s;
  uint32_t* submask;
  int32_t* tile_order;
  int64_t Vpad;
};
static bool take_plan(Workspace& A, int64_t V, int K, PlanOut& o) {
  o.Vpad = round_up_i64(V > 1 ? V : 1, SV_TILE_ROWS);
  const int64_t tiles = o.Vpad / SV_TILE_ROWS;
  o.perm = A.take<int32_t>(o.Vpad);
  o.nbr_s = A.take<int32_t>((size_t)K * o.Vpad);
  o.submask = A.take<uint32_t>((size_t)tiles * K);
  o.tile_order = A.take<int32_t>(tiles);
  return A.ok;
}

}  // namespace sv

using namespace sv;

extern "C" {

size_t sv_frame_maps_arena_bytes(int64_t N, int levels) {
  const size_t n = (size_t)(N > 1 ? N : 1);
  size_t total = align_up((size_t)(levels + 2) * 16, 256);
  total += align_up(n * 8, 256) + align_up(n * 16, 256) + align_up(n * 8, 256) + align_up(n * 4, 256) + align_up(n * 4 + 4, 256);
  total += (size_t)(levels > 0 ? levels : 0) * (align_up(n * 8, 256) + align_up(n * 16, 256) + align_up(n * 4, 256) + align_up(n * 4 + 4, 256));
  return total + 4096;
}

size_t sv_frame_maps_scratch_bytes(int64_t N) {
  const size_t a = sv_voxelize_workspace_bytes(N), b = sv_stride_map_workspace_bytes(N);
  return a > b ? a : b;
}

int sv_frame_maps(const void* coords4, int coords_are_int, int64_t N, int levels, void* arena, size_t arena_bytes, void* scratch,
                  size_t scratch_bytes, int32_t* counters_host, int64_t* layout, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(N >= 1 && N < (1ll << 31) - 1024, "N out of range (this entry point needs at least one point)");
  SV_CHECK_ARG(levels >= 0 && levels <= SV_FRAME_MAX_LEVELS, "levels out of range");
  SV_CHECK_ARG(coords4 && arena && scratch && counters_host && layout, "null pointer");
  Workspace A(arena, arena_bytes);
  int32_t* counters = A.take<int32_t>((size_t)4 * (levels + 2));
  uint64_t* keys = A.take<uint64_t>(N);
  int32_t* coords = A.take<int32_t>((size_t)4 * N);
  int64_t* inverse = A.take<int64_t>(N);
  int32_t* order = A.take<int32_t>(N);
  int32_t* seg_start = A.take<int32_t>((size_t)N + 1);
  if (!A.ok) {
    set_error("sv_frame_maps: arena too small (%zu given, sv_frame_maps_arena_bytes says %zu)", arena_bytes,
              sv_frame_maps_arena_bytes(N, levels));
    return SV_ERR_WORKSPACE;
  }
  int rc = sv_voxelize(coords4, coords_are_int, N, scratch, scratch_bytes, keys, (int32_t*)coords, inverse, order, seg_start,
                       counters, stream_);
  if (rc) return rc;
  SV_HIP(hipMemcpyAsync(counters_host, counters, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  SV_HIP(hipStreamSynchronize(stream));
  int64_t* L = layout;
  const char* base = (const char*)arena;
  L[1] = N;
  L[2] = levels;
  L[3] = (const char*)inverse - base;
  L[4] = (const char*)order - base;
  L[5] = (const char*)seg_start - base;
  L[6] = counters_host[1];  // points outside the key range
  L[7] = 0;
  if (counters_host[1] != 0) {
    set_error("sv_frame_maps: %d points have coordinates outside the key range (|coord| < 2^17 voxels, 0 <= batch < 1024)",
              counters_host[1]);
    return SV_ERR_RANGE;
  }
  int64_t V = counters_host[0];
  int64_t* rec = L + 8;
  rec[0] = V;
  rec[1] = (const char*)keys - base;
  rec[2] = (const char*)coords - base;
  rec[3] = rec[4] = -1;
  rec[5] = 0;
  const uint64_t* kin = keys;
  for (int l = 1; l <= levels; ++l) {
    const size_t vin = (size_t)(V > 1 ? V : 1);
    uint64_t* kout = A.take<uint64_t>(vin);
    int32_t* cout = A.take<int32_t>(4 * vin);
    int32_t* parent = A.take<int32_t>(vin);
    int32_t* child_start = A.take<int32_t>(vin + 1);
    if (!A.ok) {
      set_error("sv_frame_maps: arena too small at level %d", l);
      return SV_ERR_WORKSPACE;
    }
    int32_t* cnt = counters + 4 * l;
    rc = sv_stride_map(kin, V, l - 1, scratch, scratch_bytes, kout, cout, parent, child_start, cnt, stream_);
    if (rc) return rc;
    SV_HIP(hipMemcpyAsync(counters_host + 4 * l, cnt, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    SV_HIP(hipStreamSynchronize(stream));
    rec[3] = (const char*)parent - base;  // of the finer level's rows
    rec[4] = (const char*)child_start - base;
    V = V > 0 ? counters_host[4 * l] : 0;
    rec += 6;
    rec[0] = V;
    rec[1] = (const char*)kout - base;
    rec[2] = (const char*)cout - base;
    rec[3] = rec[4] = -1;
    rec[5] = 0;
    kin = kout;
  }
  L[0] = (int64_t)A.off;
  return SV_OK;
}

// ---- phase 2 --------------------------------------------------------------------------------------------------------
static int n_cuts(const int32_t* cuts) {
  int n = 0;
  while (n < SV_FRAME_MAX_CUTS && cuts[n] > 0) ++n;
  return n;
}

size_t sv_frame_plans_arena_bytes(const int64_t* V, int levels, int flags, const int32_t* split_cuts) {
  size_t total = 4096;
  for (int l = 0; l <= levels; ++l) {
    const int64_t v = V[l] > 1 ? V[l] : 1;
    if (flags & SV_FRAME_K3) {
      const int64_t cap = next_pow2_i64(2 * v > 2 ? 2 * v : 2);
      total += align_up((size_t)cap * 8, 256) + align_up((size_t)cap * 4, 256);
      total += align_up((size_t)27 * v * 4, 256) + align_up((size_t)v * 4, 256) + plan_bytes(V[l], 27);
    }
    if (l < levels && (flags & SV_FRAME_DOWN)) {
      const int64_t vc = V[l + 1] > 1 ? V[l + 1] : 1;
      total += align_up((size_t)8 * vc * 4, 256) + align_up((size_t)vc * 4, 256) + plan_bytes(V[l + 1], 8);
    }
    if (l < levels && (flags & SV_FRAME_UP)) total += align_up((size_t)8 * v * 4, 256) + align_up((size_t)v * 4, 256) + plan_bytes(V[l], 8);
    if ((flags & SV_FRAME_SPLIT) && split_cuts) {
      const int32_t* cuts = split_cuts + (size_t)l * SV_FRAME_MAX_CUTS;
      const int nc = n_cuts(cuts);
      if (nc > 0) {
        int k0 = 0;
        for (int i = 0; i <= nc; ++i) {
          const int k1 = i < nc ? cuts[i] : 27;
          total += align_up((size_t)v * 4, 256) + plan_bytes(V[l], k1 - k0);
          k0 = k1;
        }
      }
    }
  }
  return total;
}

size_t sv_frame_plans_scratch_bytes(const int64_t* V, int levels) {
  int64_t vmax = 1;
  for (int l = 0; l <= levels; ++l) vmax = V[l] > vmax ? V[l] : vmax;
  return sv_plan_workspace_bytes(vmax);
}

int sv_frame_plans(const void* const* keys, const void* const* coords, const void* const* parent, const int64_t* V, int levels,
                   int flags, const int32_t* split_cuts, const void* const* k3_nbr, const void* const* k3_mask, void* arena,
                   size_t arena_bytes, void* scratch, size_t scratch_bytes, int64_t* layout, int max_records, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(levels >= 0 && levels <= SV_FRAME_MAX_LEVELS, "levels out of range");
  SV_CHECK_ARG(keys && coords && V && arena && scratch && layout, "null pointer");
  SV_CHECK_ARG(!(flags & (SV_FRAME_DOWN | SV_FRAME_UP)) || parent, "down / up maps need the parent arrays");
  SV_CHECK_ARG(!(flags & SV_FRAME_SPLIT) || split_cuts, "offset-range plans need split_cuts");
  SV_CHECK_ARG((flags & SV_FRAME_K3) || !(flags & SV_FRAME_SPLIT) || (k3_nbr && k3_mask),
               "offset-range plans without SV_FRAME_K3 in the same call need the 27-offset maps (k3_nbr, k3_mask)");
  Workspace A(arena, arena_bytes);
  const char* base = (const char*)arena;
  int nrec = 0;
  bool table_full = false;
  int64_t* out = layout + SV_FRAME_RECORD;  // record 0 is the header
  auto emit = [&](int64_t kind, int64_t level, int64_t k0, int64_t k1, const void* nbr, const void* mask, const PlanOut& pl,
                  int64_t Vout, int64_t K, int64_t ld, const void* tk, const void* tv, int64_t cap) -> bool {
    if (nrec >= max_records) {
      table_full = true;
      return false;
    }
    int64_t* r = out + (size_t)nrec * SV_FRAME_RECORD;
    auto off = [&](const void* p) -> int64_t { return p ? (int64_t)((const char*)p - base) : -1; };
    r[0] = kind; r[1] = level; r[2] = k0; r[3] = k1;
    r[4] = off(nbr); r[5] = off(mask); r[6] = off(pl.perm); r[7] = off(pl.nbr_s); r[8] = off(pl.submask); r[9] = off(pl.tile_order);
    r[10] = Vout; r[11] = pl.Vpad; r[12] = K; r[13] = ld; r[14] = off(tk); r[15] = cap;
    (void)tv;
    ++nrec;
    return true;
  };
  const PlanOut no_plan = {nullptr, nullptr, nullptr, nullptr, 0};
  for (int l = 0; l <= levels; ++l) {
    const int64_t v = V[l];
    const int64_t v1 = v > 1 ? v : 1;
    const int32_t* nbr27 = k3_nbr ? (const int32_t*)k3_nbr[l] : nullptr;
    const uint32_t* mask27 = k3_mask ? (const uint32_t*)k3_mask[l] : nullptr;
    if (flags & SV_FRAME_K3) {
      const int64_t cap = next_pow2_i64(2 * v1 > 2 ? 2 * v1 : 2);
      uint64_t* tk = A.take<uint64_t>(cap);
      int32_t* tv = A.take<int32_t>(cap);
      int32_t* nbr = A.take<int32_t>((size_t)27 * v1);
      uint32_t* mask = A.take<uint32_t>(v1);
      PlanOut pl;
      if (!take_plan(A, v, 27, pl)) break;
      int rc = sv_hash_build((const uint64_t*)keys[l], v, tk, tv, cap, stream_);
      if (rc) return rc;
      rc = sv_kernel_map_k3((const int32_t*)coords[l], v, 1 << l, 1, tk, tv, cap, nbr, v1, mask, stream_);
      if (rc) return rc;
      rc = sv_plan_build(nbr, v1, mask, 27, v, 0, scratch, scratch_bytes, pl.perm, pl.nbr_s, pl.submask, pl.tile_order, pl.Vpad, stream_);
      if (rc) return rc;
      // hash record: slot 4 = the table's values, slot 14 = its keys, slot 15 = capacity
      if (!emit(SV_FRAME_REC_HASH, l, 0, 0, tv, nullptr, no_plan, v, 0, 0, tk, nullptr, cap)) break;
      if (!emit(SV_FRAME_REC_K3, l, 0, 27, nbr, mask, pl, v, 27, v1, nullptr, nullptr, 0)) break;
      nbr27 = nbr;
      mask27 = mask;
    }
    if (l < levels && (flags & SV_FRAME_DOWN)) {
      const int64_t vc = V[l + 1], vc1 = vc > 1 ? vc : 1;
      int32_t* nbr = A.take<int32_t>((size_t)8 * vc1);
      uint32_t* mask = A.take<uint32_t>(vc1);
      PlanOut pl;
      if (!take_plan(A, vc, 8, pl)) break;
      int rc = sv_kernel_map_down((const uint64_t*)keys[l], (const int32_t*)parent[l], v, l, vc, nbr, vc1, mask, stream_);
      if (rc) return rc;
      rc = sv_plan_build(nbr, vc1, mask, 8, vc, 0, scratch, scratch_bytes, pl.perm, pl.nbr_s, pl.submask, pl.tile_order, pl.Vpad, stream_);
      if (rc) return rc;
      if (!emit(SV_FRAME_REC_DOWN, l, 0, 8, nbr, mask, pl, vc, 8, vc1, nullptr, nullptr, 0)) break;
    }
    if (l < levels && (flags & SV_FRAME_UP)) {
      int32_t* nbr = A.take<int32_t>((size_t)8 * v1);
      uint32_t* mask = A.take<uint32_t>(v1);
      PlanOut pl;
      if (!take_plan(A, v, 8, pl)) break;
      int rc = sv_kernel_map_up((const uint64_t*)keys[l], (const int32_t*)parent[l], v, l, nbr, v1, mask, stream_);
      if (rc) return rc;
      rc = sv_plan_build(nbr, v1, mask, 8, v, 0, scratch, scratch_bytes, pl.perm, pl.nbr_s, pl.submask, pl.tile_order, pl.Vpad, stream_);
      if (rc) return rc;
      if (!emit(SV_FRAME_REC_UP, l, 0, 8, nbr, mask, pl, v, 8, v1, nullptr, nullptr, 0)) break;
    }
    if (flags & SV_FRAME_SPLIT) {
      const int32_t* cuts = split_cuts + (size_t)l * SV_FRAME_MAX_CUTS;
      const int nc = n_cuts(cuts);
      if (nc > 0) {
        SV_CHECK_ARG(nbr27 && mask27, "offset-range plans: the level's 27-offset map is missing");
        int k0 = 0;
        bool full = false;
        for (int i = 0; i <= nc && !full; ++i) {
          const int k1 = i < nc ? cuts[i] : 27;
          SV_CHECK_ARG(k1 > k0 && k1 <= 27, "split points must ascend inside (0, 27)");
          uint32_t* sub = A.take<uint32_t>(v1);
          PlanOut pl;
          if (!take_plan(A, v, k1 - k0, pl)) { full = true; break; }
          if (v > 0) {
            hipLaunchKernelGGL(mask_range_kernel, dim3((unsigned)((v + 255) / 256)), dim3(256), 0, stream, mask27, v, k0, k1 - k0, sub);
            SV_LAUNCH_CHECK();
          }
          int rc = sv_plan_build(nbr27 + (size_t)k0 * v1, v1, sub, k1 - k0, v, 0, scratch, scratch_bytes, pl.perm, pl.nbr_s, pl.submask,
                                 pl.tile_order, pl.Vpad, stream_);
          if (rc) return rc;
          // the raw rows of a range are rows k0 .. k1 of the level's 27-offset table (possibly in another arena: offset -1)
          const bool same = (const char*)nbr27 >= base && (const char*)nbr27 < base + arena_bytes;
          if (!emit(SV_FRAME_REC_SPLIT, l, k0, k1, same ? (const void*)(nbr27 + (size_t)k0 * v1) : nullptr, sub, pl, v, k1 - k0, v1,
                    nullptr, nullptr, 0)) { full = true; break; }
          k0 = k1;
        }
        if (full) break;
      }
    }
  }
  if (!A.ok) {
    set_error("sv_frame_plans: arena too small (%zu given, sv_frame_plans_arena_bytes says %zu)", arena_bytes,
              sv_frame_plans_arena_bytes(V, levels, flags, split_cuts));
    return SV_ERR_WORKSPACE;
  }
  if (table_full) {
    set_error("sv_frame_plans: layout table too small (%d records)", max_records);
    return SV_ERR_WORKSPACE;
  }
  layout[0] = (int64_t)A.off;
  layout[1] = nrec;
  return SV_OK;
}

}  // extern "C"
