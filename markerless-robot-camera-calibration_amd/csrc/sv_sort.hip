// Stable LSD radix sort of (key, int32 value) pairs, hand-written for gfx950 - the sort behind voxelisation (64-bit
// batch|Morton keys, ME.TensorField.sparse(): app/inference_engine.py:405-415) and behind the conv plans (27-bit
// Gray-ranked neighbour masks, 8-bit tile costs).  Replaces the rocPRIM calls of round 1.
//
// One pass = one digit of `bits` <= 9 bits, least significant first.  Ranks inside a tile come from wavefront ballots:
// a wave walks its contiguous segment of the tile 64 keys at a time; lanes holding the same digit find each other with
// `bits` ballots (match-any), the popcount of the match mask below a lane is its rank inside the round, and a per-wave
// per-digit counter in LDS carries the count from round to round.  No atomics, so the order inside a digit is the input
// order (stable), which is what makes LSD passes compose and what makes `order` = "point indices sorted by (key,
// original index)" exactly what the oracle's stable argsort gives.
//
//   n <= SMALL_MAX : ONE launch of one 256-thread workgroup runs every pass, tile by tile (ping-pong through global
//                    memory, the workgroup barrier orders the passes) - the small pyramid levels and the tile orders.
//   otherwise      : per pass   radix_hist (per-tile digit histogram, digit-major matrix)
//                               radix_scan (one workgroup per digit: exclusive scan over the tiles, digit total)
//                               radix_scatter (digit bases from the totals, ranks as above, scatter)
#include "sv_common.h"

namespace sv {

constexpr int RS_MAX_BITS = 9;
constexpr int RS_MAX_BINS = 1 << RS_MAX_BITS;
constexpr int RS_THREADS = 256, RS_ITEMS = 8, RS_TILE = RS_THREADS * RS_ITEMS;  // multi-workgroup path
constexpr int64_t RS_SMALL_MAX = 4 * RS_TILE;  // up to 8192 pairs go through the one-launch single-workgroup sort

// Ranks of one tile.  Wave w owns keys [tile_base + w*64*ITEMS, +64*ITEMS), visited in ITEMS rounds of 64 consecutive keys.
// On return: rank[r] = number of keys with the same digit that precede key r inside the wave's segment, and (after the
// caller's barrier) wcnt[w * bins + d] = number of keys with digit d in wave w's segment.
template <typename KeyT, int ITEMS>
__device__ __forceinline__ void rank_tile(const KeyT* __restrict__ kin, int64_t tile_base, int64_t n, int shift, int bits,
                                          uint32_t* wcnt, KeyT (&key)[ITEMS], uint32_t (&rank)[ITEMS],
                                          uint32_t (&dig)[ITEMS]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int bins = 1 << bits;
  const uint32_t dmask = (uint32_t)bins - 1u;
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int64_t i = tile_base + (int64_t)w * (64 * ITEMS) + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? kin[i] : (KeyT)0;
    const uint32_t d = (uint32_t)(key[r] >> shift) & dmask;
    dig[r] = valid ? d : 0xffffffffu;
    unsigned long long m = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const uint32_t before = (uint32_t)__popcll(m & below);
    uint32_t prev = 0;
    if (valid) prev = wcnt[w * bins + d];  // every lane of the match group reads the same counter ...
    if (valid && before == 0) wcnt[w * bins + d] = prev + (uint32_t)__popcll(m);  // ... and its first lane advances it
    rank[r] = prev + before;
    __builtin_amdgcn_wave_barrier();  // the next round's counter reads stay behind this round's update
  }
}

// wcnt[w][d] (counts) -> exclusive prefix over the waves, in place; returns nothing, totals[d] (LDS) = digit count
template <int NW>
__device__ __forceinline__ void prefix_over_waves(uint32_t* wcnt, uint32_t* totals, int bins) {
  for (int d = threadIdx.x; d < bins; d += blockDim.x) {
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const uint32_t c = wcnt[w * bins + d];
      wcnt[w * bins + d] = run;
      run += c;
    }
    totals[d] = run;
  }
}

// exclusive scan of vals[0..bins) in LDS (bins <= RS_MAX_BINS), called by the whole workgroup
__device__ __forceinline__ void exclusive_scan_bins(uint32_t* vals, int bins) {
  // bins <= 512: one wave scans 8 consecutive values per lane
  __syncthreads();
  if (threadIdx.x < 64) {
    const int per = (bins + 63) / 64;
    const int lane = threadIdx.x;
    uint32_t loc[RS_MAX_BINS / 64];
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < RS_MAX_BINS / 64; ++j) {
      const int d = lane * per + j;
      loc[j] = (j < per && d < bins) ? vals[d] : 0u;
      s += loc[j];
    }
    uint32_t x = s;
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
      const uint32_t y = __shfl_up(x, dlt);
      if (lane >= dlt) x += y;
    }
    uint32_t run = x - s;
#pragma unroll
    for (int j = 0; j < RS_MAX_BINS / 64; ++j) {
      const int d = lane * per + j;
      if (j < per && d < bins) {
        vals[d] = run;
        run += loc[j];
      }
    }
  }
  __syncthreads();
}

// ---- multi-workgroup path ---------------------------------------------------------------------------------------
template <typename KeyT>
__global__ __launch_bounds__(RS_THREADS) void radix_hist_kernel(const KeyT* __restrict__ kin, int64_t n, int shift,
                                                                int bits, int ntiles, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[RS_MAX_BINS];
  const int bins = 1 << bits;
  for (int d = threadIdx.x; d < bins; d += RS_THREADS) h[d] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
  const uint32_t dmask = (uint32_t)bins - 1u;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    const int64_t i = base + r * RS_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&h[(uint32_t)(kin[i] >> shift) & dmask], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < bins; d += RS_THREADS) hist[(int64_t)d * ntiles + blockIdx.x] = h[d];
}

// one workgroup per digit value: exclusive scan of its row of the histogram matrix over the tiles, total to totals[d]
__global__ __launch_bounds__(256) void radix_scan_kernel(uint32_t* __restrict__ hist, int ntiles,
                                                         uint32_t* __restrict__ totals) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry_s;
  uint32_t* row = hist + (int64_t)blockIdx.x * ntiles;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < ntiles; base += 256) {
    const int i = base + threadIdx.x;
    const uint32_t v = (i < ntiles) ? row[i] : 0u;
    uint32_t x = v;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += wsum[w];
    const uint32_t carry = carry_s;
    if (i < ntiles) row[i] = carry + woff + x - v;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

template <typename KeyT>
__global__ __launch_bounds__(RS_THREADS) void radix_scatter_kernel(const KeyT* __restrict__ kin,
                                                                   const int32_t* __restrict__ vin, int64_t n, int shift,
                                                                   int bits, int ntiles, const uint32_t* __restrict__ hist,
                                                                   const uint32_t* __restrict__ totals,
                                                                   KeyT* __restrict__ kout, int32_t* __restrict__ vout) {
  constexpr int NW = RS_THREADS / 64;
  __shared__ uint32_t wcnt[NW * RS_MAX_BINS];
  __shared__ uint32_t gofs[RS_MAX_BINS];
  __shared__ uint32_t tot[RS_MAX_BINS];
  const int bins = 1 << bits;
  for (int i = threadIdx.x; i < NW * bins; i += RS_THREADS) wcnt[i] = 0;
  for (int d = threadIdx.x; d < bins; d += RS_THREADS) gofs[d] = totals[d];
  exclusive_scan_bins(gofs, bins);  // digit bases; its barriers also publish the zeroed counters
  for (int d = threadIdx.x; d < bins; d += RS_THREADS) gofs[d] += hist[(int64_t)d * ntiles + blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
  KeyT key[RS_ITEMS];
  uint32_t rank[RS_ITEMS], dig[RS_ITEMS];
  rank_tile<KeyT, RS_ITEMS>(kin, base, n, shift, bits, wcnt, key, rank, dig);
  __syncthreads();
  prefix_over_waves<NW>(wcnt, tot, bins);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    if (dig[r] == 0xffffffffu) continue;
    const int64_t i = base + (int64_t)w * (64 * RS_ITEMS) + r * 64 + lane;
    const uint32_t pos = gofs[dig[r]] + wcnt[w * bins + dig[r]] + rank[r];
    kout[pos] = key[r];
    vout[pos] = vin ? vin[i] : (int32_t)i;
  }
}

// ---- single-workgroup path: every pass in one launch ----------------------------------------------------------------
// 256 threads (four waves): the workgroup fits into the slot any finishing convolution workgroup leaves behind.  (A
// 1024-thread form needs a whole drained CU; beside the chip-filling convolutions of the neighbour frame it waited for
// one ~350 us per call, profiles/r02: the prep stream's sorts are latency, not throughput.)
template <typename KeyT>
__global__ __launch_bounds__(RS_THREADS) void radix_sort_small_kernel(const KeyT* __restrict__ keys_in,
                                                                      const int32_t* __restrict__ vals_in, int n,
                                                                      int begin_bit, int end_bit, KeyT* __restrict__ kbuf0,
                                                                      int32_t* __restrict__ vbuf0, KeyT* __restrict__ kbuf1,
                                                                      int32_t* __restrict__ vbuf1, int first_dst) {
  constexpr int NW = RS_THREADS / 64;
  __shared__ uint32_t wcnt[NW * RS_MAX_BINS];
  __shared__ uint32_t base[RS_MAX_BINS];  // running start of every digit's output range
  __shared__ uint32_t tcnt[RS_MAX_BINS];  // digit counts of the current tile
  const KeyT* kin = keys_in;
  const int32_t* vin = vals_in;  // null = iota
  int dst = first_dst;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int shift = begin_bit; shift < end_bit;) {
    const int left = end_bit - shift;
    const int passes_left = (left + RS_MAX_BITS - 1) / RS_MAX_BITS;
    const int bits = (left + passes_left - 1) / passes_left;  // equal-ish digits, at most RS_MAX_BITS bits
    const int bins = 1 << bits;
    const uint32_t dmask = (uint32_t)bins - 1u;
    KeyT* kout = dst ? kbuf1 : kbuf0;
    int32_t* vout = dst ? vbuf1 : vbuf0;
    // digit histogram of the whole input -> digit bases
    for (int d = threadIdx.x; d < bins; d += RS_THREADS) base[d] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += RS_THREADS) atomicAdd(&base[(uint32_t)(kin[i] >> shift) & dmask], 1u);
    exclusive_scan_bins(base, bins);
    // tiles in input order: stable ranks inside the tile, then advance the digit bases by the tile's counts
    for (int tile = 0; tile < n; tile += RS_TILE) {
      for (int i = threadIdx.x; i < NW * bins; i += RS_THREADS) wcnt[i] = 0;
      __syncthreads();
      KeyT key[RS_ITEMS];
      uint32_t rank[RS_ITEMS], dig[RS_ITEMS];
      rank_tile<KeyT, RS_ITEMS>(kin, tile, n, shift, bits, wcnt, key, rank, dig);
      int32_t val[RS_ITEMS];
#pragma unroll
      for (int r = 0; r < RS_ITEMS; ++r) {
        const int i = tile + w * (64 * RS_ITEMS) + r * 64 + lane;
        val[r] = (i < n) ? (vin ? vin[i] : i) : 0;
      }
      __syncthreads();
      prefix_over_waves<NW>(wcnt, tcnt, bins);
      __syncthreads();
#pragma unroll
      for (int r = 0; r < RS_ITEMS; ++r) {
        if (dig[r] == 0xffffffffu) continue;
        const uint32_t pos = base[dig[r]] + wcnt[w * bins + dig[r]] + rank[r];
        kout[pos] = key[r];
        vout[pos] = val[r];
      }
      __syncthreads();
      for (int d = threadIdx.x; d < bins; d += RS_THREADS) base[d] += tcnt[d];
      __syncthreads();  // also the workgroup-scope release/acquire that lets the next pass read what this one wrote
    }
    kin = kout;
    vin = vout;
    dst ^= 1;
    shift += bits;
  }
}

static int plan_passes(int begin_bit, int end_bit) { return (end_bit - begin_bit + RS_MAX_BITS - 1) / RS_MAX_BITS; }

size_t radix_sort_temp_bytes(int64_t n, size_t key_bytes) {
  if (n <= 0) return 256;
  const size_t ntiles = (size_t)((n + RS_TILE - 1) / RS_TILE);
  return align_up((size_t)n * key_bytes, 256) + align_up((size_t)n * 4, 256) +     // ping-pong partner of the outputs
         align_up(ntiles * RS_MAX_BINS * 4, 256) + align_up(RS_MAX_BINS * 4, 256) + 1024;
}

// Sorts n pairs by key bits [begin_bit, end_bit), stable.  vals_in == nullptr means values 0..n-1.  The result lands in
// keys_out / vals_out (distinct from the inputs); `temp` holds radix_sort_temp_bytes(n, sizeof(KeyT)).
template <typename KeyT>
int radix_sort_pairs(const KeyT* keys_in, const int32_t* vals_in, KeyT* keys_out, int32_t* vals_out, int64_t n,
                     int begin_bit, int end_bit, void* temp, size_t temp_bytes, hipStream_t stream) {
  if (n <= 0) return SV_OK;
  if (temp_bytes < radix_sort_temp_bytes(n, sizeof(KeyT))) {
    set_error("radix_sort_pairs: temporary storage too small");
    return SV_ERR_WORKSPACE;
  }
  Workspace ws(temp, temp_bytes);
  KeyT* ktmp = ws.take<KeyT>((size_t)n);
  int32_t* vtmp = ws.take<int32_t>((size_t)n);
  const int ntiles = (int)((n + RS_TILE - 1) / RS_TILE);
  uint32_t* hist = ws.take<uint32_t>((size_t)ntiles * RS_MAX_BINS);
  uint32_t* totals = ws.take<uint32_t>(RS_MAX_BINS);
  if (end_bit <= begin_bit) end_bit = begin_bit + 1;  // a 0-bit key: one pass over a constant digit = a stable copy
  const int passes = plan_passes(begin_bit, end_bit);
  // the last pass must write keys_out / vals_out: buffer 1 = out, buffer 0 = temp, alternate backwards from the end
  int dst = (passes % 2 == 1) ? 1 : 0;
  if (n <= RS_SMALL_MAX) {
    hipLaunchKernelGGL(radix_sort_small_kernel<KeyT>, dim3(1), dim3(RS_THREADS), 0, stream, keys_in, vals_in, (int)n,
                       begin_bit, end_bit, ktmp, vtmp, keys_out, vals_out, dst);
    SV_LAUNCH_CHECK();
    return SV_OK;
  }
  const KeyT* kin = keys_in;
  const int32_t* vin = vals_in;
  for (int shift = begin_bit; shift < end_bit;) {
    const int left = end_bit - shift;
    const int passes_left = (left + RS_MAX_BITS - 1) / RS_MAX_BITS;
    const int bits = (left + passes_left - 1) / passes_left;
    KeyT* kout = dst ? keys_out : ktmp;
    int32_t* vout = dst ? vals_out : vtmp;
    hipLaunchKernelGGL(radix_hist_kernel<KeyT>, dim3(ntiles), dim3(RS_THREADS), 0, stream, kin, n, shift, bits, ntiles,
                       hist);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(1 << bits), dim3(256), 0, stream, hist, ntiles, totals);
    hipLaunchKernelGGL(radix_scatter_kernel<KeyT>, dim3(ntiles), dim3(RS_THREADS), 0, stream, kin, vin, n, shift, bits,
                       ntiles, hist, totals, kout, vout);
    SV_LAUNCH_CHECK();
    kin = kout;
    vin = vout;
    dst ^= 1;
    shift += bits;
  }
  return SV_OK;
}

template int radix_sort_pairs<uint64_t>(const uint64_t*, const int32_t*, uint64_t*, int32_t*, int64_t, int, int, void*,
                                        size_t, hipStream_t);
template int radix_sort_pairs<uint32_t>(const uint32_t*, const int32_t*, uint32_t*, int32_t*, int64_t, int, int, void*,
                                        size_t, hipStream_t);

}  // namespace sv
