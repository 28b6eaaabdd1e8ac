// K6/K7: output-stationary sparse convolution on fp32 MFMA (v_mfma_f32_16x16x4_f32) with fused epilogue.
//
// Replaces ME.MinkowskiConvolution / MinkowskiConvolutionTranspose / MinkowskiLinear followed by
// MinkowskiBatchNorm(eval) (+ residual) (+ ReLU / LeakyReLU): model/backbone/minkunet.py:125-187,
// model/backbone/resnet.py:95-127 (BasicBlock via ME), model/robotnet_segmentation.py:55-64.
//
// Work decomposition (details at ConvCfg / conv_fwd_kernel below, measurements in DESIGN.md 4.1)
//   * a workgroup (4 waves) owns one tile of TM output rows (in the plan's mask-sorted order) x TN output channels;
//   * it walks pipeline steps = (kernel offset k ASCENDING, input-channel chunk ascending); thin layers fuse several
//     offsets into one step;
//   * per step the gathered input rows (A, TM x KC) go global -> registers -> LDS (double buffered, one barrier per
//     step); the weights (B, KC x TN) go straight from L2/HBM to registers;
//   * each wave multiplies with v_mfma_f32_16x16x4_f32; a 16-row sub-tile is skipped for an offset when none of its rows
//     has a neighbour there (plan submask), which is what makes the mask-sorted row order pay.
// Numerics: every output element is ONE f32 fma chain over (k ascending, c ascending) — the MFMA is a k-ordered
// fmaf chain (MI355X guide §3) — so results are bitwise reproducible and match oracle/sv_oracle.c exactly.
// A missing neighbour contributes fma(0, w, acc) = acc.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <algorithm>
#include <vector>

#include "sv_common.h"

namespace sv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvParams {
  const float* in;
  int64_t in_ld;
  int Cin;
  const float* W;
  int K;
  int Cout;
  const int32_t* perm;
  const int32_t* nbr_s;
  const uint32_t* submask;
  const int32_t* tile_order;  // plan tiles (128 rows), longest first; NULL = reverse plan order
  int64_t V_out;
  int64_t Vpad;
  const float* scale;
  const float* shift;
  const float* residual;
  int64_t res_ld;
  // accumulator hand-over between the passes of a layer whose kernel offsets are split into ascending ranges (each range
  // with its own plan / row order): acc_init[o][n] = the raw fma chain over the EARLIER offsets of output element (o, n); it
  // is the matrix op's C operand at the start of this launch's chain, so the chain over all offsets is the one chain it
  // always was.  NULL = the chain starts at 0.
  const float* acc_init;
  int64_t acc_ld;
  uint32_t acc_bytes;
  int act;
  float slope;
  float* out;
  int64_t out_ld;
  int vec_a;  // in_ld % 4 == 0 && Cin % 4 == 0 && base aligned -> float4 gathers
  // FAST instances address `in` and `W` through buffer descriptors with 32-bit byte offsets (see conv_tile_body):
  uint32_t in_bytes, w_bytes, out_bytes, res_bytes;  // extents of in, W, out, residual
  int buf_ok;  // all of them below BUF_LIMIT (else the guarded generic form with 64-bit addresses runs)
  int ntiles;
  int ny;
  unsigned long long* trace;  // SV_CONV_TRACE experiments: per-workgroup {start, end, hw id, steps}; null otherwise
  int main_blocks;            // dual-body launches: workgroups [0, main_blocks) run the main tile shape over the plan tiles
  int main_tiles128;          //   tile_order[0, main_tiles128), the rest the tail shape over tile_order[main_tiles128, ..)
};

constexpr int PLAN_TILE = SV_TILE_ROWS;
// name of the kernel instance the last sv_conv_fwd call of this thread launched, "name|fast=F,ring=R,full=U" (the names
// mrcc_amd/profiling.py predicts; sv_conv_last_instance(): tests and the bench's per-kernel records read it back)
static thread_local char g_last_instance[128] = "";
// per-thread override of the dispatch thresholds (sv_conv_set_dispatch): < 0 = the library default / environment
static thread_local double g_want_scale_override = -1.0;
static thread_local double g_tail_override = -1.0;
static void note_instance(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_instance, sizeof(g_last_instance), fmt, ap);
  va_end(ap);
}
// Buffer addressing of the FAST instances.  Measured with tools/mfma_probe.py on gfx950: a `global_load` with a 64-bit
// VGPR address costs the SIMD's matrix pipe ~45 cycles of issue per instruction (one per 12 matrix ops: 0.98 -> 0.86 of
// the peak issue rate), and every VALU instruction in the loop (address arithmetic, selects) its own execution time;
// `buffer_load` with a 32-bit VGPR offset and an SGPR offset costs nothing measurable (0.97).  Out-of-range offsets
// return 0 without a memory access, which is how absent neighbours read as zero rows: no select, no branch.
constexpr uint32_t BUF_ABSENT = 0x80000000u;  // byte offset of an absent neighbour's row: beyond every extent
constexpr uint32_t BUF_LIMIT = 0x7fff0000u;   // extents stay below BUF_ABSENT minus the largest column offset
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x3_t __attribute__((ext_vector_type(3)));
typedef int i32x2_t __attribute__((ext_vector_type(2)));
template <int N>
__device__ __forceinline__ auto buffer_load_floats(__amdgpu_buffer_rsrc_t rsrc, uint32_t voffset, uint32_t soffset) {
  typedef float vec_t __attribute__((ext_vector_type(N)));
  if constexpr (N == 1) {
    vec_t r;
    r[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voffset, soffset, 0));
    return r;
  } else if constexpr (N == 2) {
    return __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voffset, soffset, 0));
  } else if constexpr (N == 3) {
    return __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b96(rsrc, voffset, soffset, 0));
  } else {
    static_assert(N == 4, "1..4 floats per load");
    return __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0));
  }
}
#ifndef SV_CONV_WANT_SCALE_DEFAULT
#define SV_CONV_WANT_SCALE_DEFAULT 0.3  // see select_and_launch
#endif
#ifndef SV_CONV_TAIL_DEFAULT
#define SV_CONV_TAIL_DEFAULT 0.15  // share of the plan tiles (the cheapest) that chip-filling launches run as half-height tiles
#endif  // plans (perm / nbr_s / submask) are laid out in 128-row tiles
// input channels per pipeline step: short tiles (used on small pyramid levels, where a launch is bound by the latency
// of a tile's sequential step chain) take wider chunks, i.e. fewer barriers / gather round trips per tile
constexpr int chunk_for(int tm, int waves_n) {
  if (waves_n == 1) return tm <= 64 ? 128 : 64;  // tall-narrow tiles (small levels): few, fat steps
  if (waves_n == 2) return tm <= 32 ? 128 : (tm <= 64 ? 64 : 32);
#ifdef SV_EXP_KC16
  return tm <= 16 ? 128 : (tm <= 32 ? 32 : 16);  // experiment: 16-channel steps on the 64-row tiles, 32 on the 32-row ones (fewer B / gather registers: five waves per SIMD)
#else
  return tm <= 16 ? 128 : (tm <= 32 ? 64 : 32);
#endif
}

// Workgroup = 4 waves.  Tile = TM_ output rows (mask-sorted plan order) x TN output channels.
//   WAVES_N waves split the columns (NT 16-wide MFMA column tiles each), WAVES_M = 4 / WAVES_N split the rows.
//   Column tiles are INTERLEAVED: MFMA column j of tile n is output channel n0 + wn*NT*16 + NT*j + n, so a lane's B
//   operands for its NT tiles are NT consecutive floats of a weight row (one dwordxNT load straight from L2/HBM into
//   registers — the weight slab is not shared between waves, an LDS round trip would be pure overhead) and the
//   epilogue stores NT consecutive floats per lane.
// Thin layers (Cin = 3 / 16 / 32 / 64) FUSE several kernel offsets into one pipeline step (CPO = channels per offset,
// compile-time, = Cin): the A tile row of a step is the concatenation of GK neighbours' Cin channels, which is a
// contiguous run of GK * Cin rows of W[K][Cin][Cout] - the same ascending (k, c) accumulation chain in 27 * Cin / 128
// steps instead of 27.  CPO = 0 is the one-offset-per-step form used by the wide layers.
constexpr int fused_offsets(int cpo) { return cpo == 0 ? 1 : (cpo < 8 ? 27 : (128 / cpo > 0 ? 128 / cpo : 1)); }
constexpr int lds_row_stride(int kc) {
  // smallest stride >= kc + 2 for which the 16x16x4 operand read (lane = row li, k-offset lq: word li * SA + lq)
  // touches 64 distinct LDS banks: SA mod 64 = 34, or SA = 4 * odd
  for (int sa = kc + 2;; ++sa)
    if (sa % 64 == 34 || (sa % 4 == 0 && (sa / 4) % 2 == 1)) return sa;
}

template <int TM_, int WAVES_N, int NT, int CPO = 0>
struct ConvCfg {
  static constexpr int WAVES_M = 4 / WAVES_N;
  static constexpr int MR = TM_ / WAVES_M / 16;  // 16-row sub-tiles per wave
  static constexpr int TN = WAVES_N * NT * 16;   // output channels per workgroup
  static constexpr int GK = fused_offsets(CPO);  // kernel offsets per pipeline step
  static constexpr int KC = CPO ? (GK * CPO + 3) / 4 * 4 : chunk_for(TM_, WAVES_N);  // A columns per step
  static constexpr int SA = lds_row_stride(KC);  // A row stride in floats
  // A-operand prefetch distance in k-steps: enough matrix work (MR * NT ops of 32 cycles) to cover ~250 cycles
  // (big tiles, MR * NT >= 6: one k-step, their waves hide the rest behind each other)
  static constexpr int PFD_RAW = MR * NT >= 6 ? 1 : (8 + MR * NT - 1) / (MR * NT);
  static constexpr int PFD = PFD_RAW > KC / 4 ? KC / 4 : PFD_RAW;
  // the FULL form (see the kernel) pays where a k-step is one or two matrix ops and the per-k-step bookkeeping of
  // the general form dominates; on the 64-row tile it costs the 129th VGPR (3 instead of 4 waves per SIMD) and on
  // 16x128 tiles it measured slower (profiles/r01_conv_full_form.txt)
#ifdef SV_EXP_FULL64
  static constexpr bool USE_FULL = (MR * NT == 1) || (TM_ == 32 && WAVES_N == 4 && NT == 3) || (TM_ == 16 && NT == 3) || (TM_ == 64 && WAVES_N == 4);
#else
  static constexpr bool USE_FULL = (MR * NT == 1) || (TM_ == 32 && WAVES_N == 4 && NT == 3) || (TM_ == 16 && NT == 3);
#endif
  static constexpr int F4_PER_ROW = KC / 4;      // float4 per gathered row and step
  static constexpr int ROWS_PER_PASS = 256 / F4_PER_ROW;
  static constexpr int A_F4 = (TM_ + ROWS_PER_PASS - 1) / ROWS_PER_PASS;  // float4 gathers per thread and step
  static_assert(MR >= 1, "tile too small for the wave layout");
  static constexpr size_t lds_bytes(int K) { return (size_t)(2 * TM_ * SA + K * TM_) * sizeof(float); }
};

// ---- epilogue of the buffer-addressed kernels: BN(eval) / bias -> residual -> activation -> store by `perm`.
//      acc[s][n][reg]: C/D map of the matrix op, column = lane & 15 of column tile n (output channel col0 + n), row =
//      rows0 + 16 s + 4 lq + reg of the plan.
// the 4 consecutive plan rows a lane stores per sub-tile (one int4 of `perm`); the single-wave kernels request them at
// their very start so that the epilogue does not begin with a dependent round trip
template <int MR>
__device__ __forceinline__ void load_perm_rows(const ConvParams& p, const int64_t rows0, const int lq, int (&o)[MR][4]) {
#pragma unroll
  for (int s = 0; s < MR; ++s) {
    const int64_t r = rows0 + s * 16 + lq * 4;
    if (p.perm) {
      const int4 o4 = *(const int4*)(p.perm + r);
      o[s][0] = o4.x;
      o[s][1] = o4.y;
      o[s][2] = o4.z;
      o[s][3] = o4.w;
    } else {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) o[s][reg] = (r + reg < p.V_out) ? (int)(r + reg) : -1;
    }
  }
}

template <int MR, int NT, bool PRELOADED = false>
__device__ __forceinline__ void epilogue_buffered(const ConvParams& p, const f32x4 (&acc)[MR][NT], const int64_t rows0,
                                                  const int lq, const int col0, const int (*o_pre)[4] = nullptr) {
  // branch-free: the 4 consecutive output rows a lane holds per sub-tile come from ONE int4 load of `perm`, all MR of
  // them requested up front; a sub-tile's residual rows are requested together; rows past V_out (perm < 0) get a
  // byte offset beyond the extents, so their residual loads return zeros and their stores are dropped by the
  // descriptor's range check.  (With a branch per row and per column the epilogue was a chain of dependent round
  // trips - 31 us of a 770 us workgroup on the 64-row tile; this form: dense layers +4-5 %, the level-0 launch and
  // the frame rate +1.2 %.)
  typedef float yvec_t __attribute__((ext_vector_type(NT)));
  const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_res =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, p.residual ? (int)p.res_bytes : 0, 0x00020000);
  int o[MR][4];
  if constexpr (PRELOADED) {
#pragma unroll
    for (int s = 0; s < MR; ++s)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) o[s][reg] = o_pre[s][reg];
  } else {
    load_perm_rows<MR>(p, rows0, lq, o);
  }
  // The arithmetic is unconditional: absent BN / bias / residual become operands that change no bit of any value
  // (fmaf(x, 1, -0) == x and x + (-0) == x for every x, signed zeros and NaN included; bias alone: fmaf(x, 1, b) is
  // the one rounding of x + b), and the activation is chosen ONCE, outside the unrolled element loops.  (With the
  // three run-time switches tested per element the 48 elements of a lane were ~150 scalar branches: 22 us from the
  // end of the loop to the last store of a 64-row workgroup, 7 us of a 16-row one - per-phase stamps of a trace build.)
  float scf[NT], shf[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    scf[n] = p.scale ? p.scale[col0 + n] : 1.0f;
    shf[n] = p.shift ? p.shift[col0 + n] : (p.scale ? 0.0f : -0.0f);
  }
  const float slope = p.slope;
  auto finish = [&](auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
    for (int s = 0; s < MR; ++s) {
      yvec_t res[4];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
#pragma unroll
        for (int n = 0; n < NT; ++n) res[reg][n] = -0.0f;
      if (p.residual) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          res[reg] = buffer_load_floats<NT>(
              rsrc_res, o[s][reg] >= 0 ? (uint32_t)o[s][reg] * (uint32_t)(p.res_ld * 4) + (uint32_t)col0 * 4u : BUF_ABSENT,
              0u);
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        yvec_t y;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          float v = __builtin_fmaf(acc[s][n][reg], scf[n], shf[n]) + res[reg][n];
          if constexpr (ACT == SV_ACT_RELU)
            v = v < 0.f ? 0.f : v;  // NaN stays NaN, as torch.relu
          else if constexpr (ACT == SV_ACT_LEAKY_RELU)
            v = v > 0.f ? v : v * slope;
          y[n] = v;
        }
        const uint32_t off =
            o[s][reg] >= 0 ? (uint32_t)o[s][reg] * (uint32_t)(p.out_ld * 4) + (uint32_t)col0 * 4u : BUF_ABSENT;
        if constexpr (NT == 1)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, y[0]), rsrc_out, off, 0, 0);
        else if constexpr (NT == 2)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2_t, y), rsrc_out, off, 0, 0);
        else if constexpr (NT == 3)
          __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(i32x3_t, y), rsrc_out, off, 0, 0);
        else
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, y), rsrc_out, off, 0, 0);
      }
    }
  };
  if (p.act == SV_ACT_RELU)
    finish(std::integral_constant<int, SV_ACT_RELU>{});
  else if (p.act == SV_ACT_LEAKY_RELU)
    finish(std::integral_constant<int, SV_ACT_LEAKY_RELU>{});
  else
    finish(std::integral_constant<int, SV_ACT_NONE>{});
}

// One pipeline step = (kernel offset k, input-channel chunk c0), visited in ascending (k, c0) order (CPO > 0: GK
// consecutive offsets x all Cin channels per step).
//   A (gathered rows, TM_ x KC): global -> registers one step ahead -> LDS (double buffered), one barrier per step.
//   B (weights, KC x TN): global -> registers, re-loaded for the next step right after their last use.
// FAST: float4 gathers, Cout a multiple of TN -> every load is unconditional (out-of-range lanes read a safe address and
//   are zeroed afterwards), so the loop body is straight-line and hipcc can emit COUNTED vmcnt waits; the generic
//   variant keeps per-lane guards (odd channel counts such as Cin = 3 or Cout = 3).
// RING: A operands are read PFD k-steps ahead of the matrix ops through a register ring (launches of few workgroups).
// FULL: no partial chunk in the layer -> compile-time trip count and plain weight addressing in the matrix loop.
// conv_tile_body: one workgroup's tile.  bidx = index of the workgroup inside its body's range of the grid, tile_base =
// first entry of tile_order (plan tiles of 128 rows, longest first) that range starts at.
template <int TM_, int WAVES_N, int NT, bool FAST, int CPO = 0, bool RING = false, bool FULL = false>
__device__ __forceinline__ void conv_tile_body(const ConvParams& p, const int bidx, const int tile_base) {
  static_assert(FAST || !FULL, "FULL is a refinement of the FAST form");
  using Cfg = ConvCfg<TM_, WAVES_N, NT, CPO>;
  constexpr int MR = Cfg::MR, TN = Cfg::TN, A_F4 = Cfg::A_F4, KC = Cfg::KC, SA = Cfg::SA, GK = Cfg::GK;
  constexpr int PFD = RING ? Cfg::PFD : 1;  // RING: launches too small to fill the chip (see launch_conv)
  constexpr int ROWS_PER_PASS = Cfg::ROWS_PER_PASS;
  constexpr int SUBS = TM_ / 16;  // sub-tiles per tile
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                             // [2][TM_][SA]
  int* idx_s = (int*)(lds + 2 * TM_ * SA);     // [K][TM_] gathered input row of every (offset, tile row)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  unsigned long long trace_t0 = 0, trace_t_loop = 0, trace_t_epi = 0;
  int trace_steps = 0;
  if (p.trace) trace_t0 = wall_clock64();
  // 1-D grid, longest tiles first (list scheduling): plan tiles are visited in the plan's tile_order (sorted by active
  // (offset, sub-tile) slots, descending); without one, in reverse plan order (rows are sorted by neighbour key, so
  // the tiles with the most neighbour offsets sit at the end).
  const int ny = (p.Cout + TN - 1) / TN;
  constexpr int SUB_PER_PLAN_TILE = PLAN_TILE / TM_;
  const int t_lin = bidx / ny;
  const int t128 = tile_base + t_lin / SUB_PER_PLAN_TILE;
  const int p128 = p.tile_order ? p.tile_order[t128] : ((int)(p.Vpad / PLAN_TILE) - 1 - t128);
  const int tile = p128 * SUB_PER_PLAN_TILE + (t_lin % SUB_PER_PLAN_TILE);
  const int n0 = (bidx % ny) * TN;
  const int64_t row0 = (int64_t)tile * TM_;
  const int li = lane & 15, lq = lane >> 4;
  const int K = p.K, Cin = p.Cin, Cout = p.Cout;
  const int col0 = n0 + wn * NT * 16 + NT * li;  // this lane's first output channel
  const bool cols_full = col0 + NT <= Cout;

  f32x4 acc[MR][NT];
#pragma unroll
  for (int s = 0; s < MR; ++s)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[s][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // timing-only ablations (results WRONG; tools/build_variant.sh <name> -DSV_ABL=<bits>): 1 = no weight reloads in the
  // loop, 2 = no gathers / LDS stores in the loop, 4 = no per-step barrier, 8 = A operands from a register, not LDS,
  // 16 = no epilogue (one guarded store keeps the accumulators alive), 32 = no global reads in the prologue (table computed,
  // no acc_init) - 16 + 32 bound what hiding a tile's fixed costs behind its neighbours' steps could gain (round 4: 2.6 %)
#ifndef SV_ABL
#define SV_ABL 0
#endif
  constexpr int ABL = SV_ABL;

  // ---- continue an earlier pass's chains: C operands from acc_init, addressed through `perm` like the epilogue's rows
  if (p.acc_init && !(ABL & 32)) {
    if constexpr (FAST) {
      const __amdgpu_buffer_rsrc_t rsrc_acc = __builtin_amdgcn_make_buffer_rsrc((void*)p.acc_init, 0, (int)p.acc_bytes, 0x00020000);
      int o_acc[MR][4];
      load_perm_rows<MR>(p, row0 + wm * MR * 16, lane >> 4, o_acc);
#pragma unroll
      for (int s = 0; s < MR; ++s)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const auto v = buffer_load_floats<NT>(
              rsrc_acc, o_acc[s][reg] >= 0 ? (uint32_t)o_acc[s][reg] * (uint32_t)(p.acc_ld * 4) + (uint32_t)col0 * 4u : BUF_ABSENT, 0u);
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[s][n][reg] = v[n];
        }
    } else {
#pragma unroll
      for (int s = 0; s < MR; ++s)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int64_t r = row0 + wm * MR * 16 + s * 16 + (lane >> 4) * 4 + reg;
          const int64_t o = p.perm ? (int64_t)p.perm[r] : (r < p.V_out ? r : -1);
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[s][n][reg] = (o >= 0 && col0 + n < Cout) ? p.acc_init[o * p.acc_ld + col0 + n] : 0.0f;
        }
    }
  }
  // ---- stage the tile's neighbour table (or the identity for dense rows) in LDS
  uint32_t dense_mask = (1u << SUBS) - 1u;
  if (p.submask == nullptr) {
    const int64_t rem = p.V_out - row0;
    const int nsub = rem >= TM_ ? SUBS : (int)((rem + 15) / 16);
    dense_mask = (1u << nsub) - 1u;
  }
  if constexpr (FAST) {
    // all of the thread's table entries are requested before the first is used (a rolled loop waits for each load in
    // turn); the table holds each row's BYTE offset in `in` (absent: beyond the buffer's extent -> zeros)
    constexpr int STAGE_IT = (32 * TM_ + 255) / 256;  // K <= 32
    int n_st[STAGE_IT];
#pragma unroll
    for (int it = 0; it < STAGE_IT; ++it) {
      const int e = tid + 256 * it;
      const int k = e / TM_, r = e % TM_;
      n_st[it] = -1;
      if (e < K * TM_) {
        if (ABL & 32)
          n_st[it] = (int)((row0 + r + 97 * k) % p.V_out);
        else if (p.nbr_s)
          n_st[it] = p.nbr_s[(int64_t)k * p.Vpad + row0 + r];
        else
          n_st[it] = (row0 + r < p.V_out) ? (int)(row0 + r) : -1;
      }
    }
#pragma unroll
    for (int it = 0; it < STAGE_IT; ++it) {
      const int e = tid + 256 * it;
      if (e < K * TM_) idx_s[e] = (int)(n_st[it] >= 0 ? (uint32_t)n_st[it] * (uint32_t)(p.in_ld * 4) : BUF_ABSENT);
    }
  } else {
    for (int e = tid; e < K * TM_; e += 256) {
      const int k = e / TM_, r = e - k * TM_;
      int n;
      if (p.nbr_s)
        n = p.nbr_s[(int64_t)k * p.Vpad + row0 + r];
      else
        n = (row0 + r < p.V_out) ? (int)(row0 + r) : -1;
      idx_s[e] = n;
    }
  }
  const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)p.w_bytes, 0x00020000);

  // ---- step iterator over (active offset, chunk).  The plan's submask words cover 128 rows = 8 sub-tiles; lane k of
  //      every wave keeps the tile's word for offset k in a register and a ballot gives the active-offset set, so
  //      stepping to the next active offset is a bit scan + v_readlane (no memory access between steps).
  const int64_t sm_word = row0 / PLAN_TILE;
  const int sm_shift = (int)((row0 % PLAN_TILE) / 16);
  uint32_t my_sm = 0;
  if (lane < K) my_sm = p.submask ? ((p.submask[sm_word * K + lane] >> sm_shift) & ((1u << SUBS) - 1u)) : dense_mask;
  const uint32_t amask = (uint32_t)__ballot(my_sm != 0);
  // fused-offset steps: a step starts at offset k0 = multiple of GK, its sub-tile mask is the union over its offsets
  auto fused_mask = [&](int k0) -> uint32_t {
    uint32_t m = 0;
#pragma unroll
    for (int g = 0; g < GK; ++g) m |= (uint32_t)__builtin_amdgcn_readlane((int)my_sm, min(k0 + g, 31));
    return m;  // lanes >= K hold 0
  };
  int k_n, c_n = 0;
  uint32_t sm_n = 0;
  bool have_n;
  if (CPO) {
    k_n = 0;
    while (k_n < K && (sm_n = fused_mask(k_n)) == 0) k_n += GK;
    have_n = k_n < K;
  } else {
    k_n = amask ? __builtin_ctz(amask) : K;
    sm_n = amask ? (uint32_t)__builtin_amdgcn_readlane((int)my_sm, k_n) : 0u;
    have_n = amask != 0;
  }
  auto advance = [&]() {
    if (CPO) {
      k_n += GK;
      while (k_n < K && (sm_n = fused_mask(k_n)) == 0) k_n += GK;
      have_n = k_n < K;
      return;
    }
    c_n += KC;
    if (c_n >= Cin) {
      c_n = 0;
      const uint32_t rest = amask & ~((2u << k_n) - 1u);
      have_n = rest != 0;
      k_n = have_n ? __builtin_ctz(rest) : K;
      sm_n = (uint32_t)__builtin_amdgcn_readlane((int)my_sm, have_n ? k_n : 0);
    }
  };

  // gathered rows in flight: one register set per step of gather depth.  With GDEPTH = 2 the rows of step s + 2 are
  // requested at the start of step s and written to LDS at the end of step s + 1 (loop unrolled by two so that the
  // register sets keep static names).  Measured and NOT used (-DSV_GATHER_DEPTH=2 rebuilds it): the wide layers lose a
  // wave per SIMD (181 VGPRs: 88 vs 98 TFLOP/s at level 0) and the thin fused-offset layers do not gain (32->32 at level 1:
  // 36 vs 33 us; 64->64 at level 2: 31 vs 30 us) - with cache-resident gathers the wide layers run no faster either, i.e.
  // gather latency is not what a step waits for.
#ifndef SV_GATHER_DEPTH
#define SV_GATHER_DEPTH 1
#endif
  constexpr int GDEPTH = SV_GATHER_DEPTH;
  float4 ra0[A_F4], ra1[A_F4];  // ra1 is dead (optimised away) at GDEPTH 1
  const int a_cc = (tid % Cfg::F4_PER_ROW) * 4;
  const int a_r = tid / Cfg::F4_PER_ROW;

  // FAST: this thread's column inside a chunk in bytes; in a partial last chunk, threads past Cin use minus the chunk's
  // start instead (voffset + soffset = the row's first columns)
  const uint32_t col_bytes = (uint32_t)a_cc * 4u;
  const int cin_rem = Cin % KC;
  const uint32_t col_last = (cin_rem == 0 || a_cc < cin_rem) ? col_bytes : 0u - (uint32_t)(Cin - cin_rem) * 4u;
  auto load_a = [&](float4 (&ra)[A_F4], int k, int c0, uint32_t sm) {
#pragma unroll
    for (int j = 0; j < A_F4; ++j) {
      const int r = a_r + ROWS_PER_PASS * j;
      if (FAST) {
        // one ds_read (the row's byte offset), one add (this thread's column) and a buffer_load whose SGPR offset is the
        // step's channel chunk: rows past the tile (clamped), sub-tiles without a neighbour at this offset (all their
        // rows are absent) and absent neighbours need no test - the descriptor's range check returns zeros for them
        const int rr = (A_F4 * ROWS_PER_PASS > TM_) ? min(r, TM_ - 1) : r;
        if (CPO) {
          // fused offsets: column a_cc of the step belongs to offset k + a_cc / CPO, channel a_cc % CPO
          const int kk = k + a_cc / (CPO ? CPO : 1);
          uint32_t off = (uint32_t)idx_s[min(kk, K - 1) * TM_ + rr];
          if (kk >= K) off = BUF_ABSENT;
          const f32x4 v = buffer_load_floats<4>(rsrc_in, off + (uint32_t)(a_cc % (CPO ? CPO : 1)) * 4u, 0u);
          ra[j] = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#ifdef SV_EXP_ASKIP
          // experiment: a wave's j-th gather covers rows of ONE sub-tile ((a_r >> 4) + ROWS_PER_PASS / 16 * j, wave-uniform
          // when ROWS_PER_PASS is a multiple of 16); skip it - and its LDS store - when that sub-tile is inactive at this step
          if (ROWS_PER_PASS % 16 == 0 && TM_ >= 32 &&
              !((sm >> __builtin_amdgcn_readfirstlane((a_r >> 4) + (ROWS_PER_PASS / 16) * j)) & 1u))
            continue;
#endif
          const uint32_t off = (uint32_t)idx_s[k * TM_ + rr];
          // the last chunk of a layer whose Cin is not a multiple of KC: columns past Cin are never multiplied; their
          // lanes re-read the row's first columns (col_last) so that no load reaches past a row
          const uint32_t cb = (!FULL && c0 + KC > Cin) ? col_last : col_bytes;
          const f32x4 v = buffer_load_floats<4>(rsrc_in, off + cb, (uint32_t)c0 * 4u);
          ra[j] = make_float4(v[0], v[1], v[2], v[3]);
        }
        (void)sm;
      } else if (CPO) {
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < TM_ && ((sm >> (r >> 4)) & 1u)) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = a_cc + q, kk = k + i / (CPO ? CPO : 1), c = i % (CPO ? CPO : 1);
            if (i < GK * CPO && kk < K) {
              const int n = idx_s[kk * TM_ + r];
              if (n >= 0) e[q] = p.in[(int64_t)n * p.in_ld + c];
            }
          }
        }
        ra[j] = make_float4(e[0], e[1], e[2], e[3]);
      } else {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < TM_ && ((sm >> (r >> 4)) & 1u)) {
          const int n = idx_s[k * TM_ + r];
          const int c = c0 + a_cc;
          if (n >= 0) {
            const float* src = p.in + (int64_t)n * p.in_ld + c;
            if (p.vec_a) {
              if (c < Cin) v = *(const float4*)src;
            } else {
              if (c + 0 < Cin) v.x = src[0];
              if (c + 1 < Cin) v.y = src[1];
              if (c + 2 < Cin) v.z = src[2];
              if (c + 3 < Cin) v.w = src[3];
            }
          }
        }
        ra[j] = v;
      }
    }
  };
  auto store_a = [&](float4 (&ra)[A_F4], float* dstbuf, uint32_t sm) {
#pragma unroll
    for (int j = 0; j < A_F4; ++j) {
      const int r = a_r + ROWS_PER_PASS * j;
      // FAST: rows of sub-tiles that are inactive at this offset arrive as zeros and are stored like the others (their
      // matrix ops are skipped anyway) - no per-row mask arithmetic in the loop
#ifdef SV_EXP_ASKIP
      if (FAST && !CPO && ROWS_PER_PASS % 16 == 0 && TM_ >= 32 &&
          !((sm >> __builtin_amdgcn_readfirstlane((a_r >> 4) + (ROWS_PER_PASS / 16) * j)) & 1u))
        continue;
#endif
      if (r < TM_ && (FAST || ((sm >> (r >> 4)) & 1u))) {
        float2* dst = (float2*)(dstbuf + r * SA + a_cc);
        dst[0] = make_float2(ra[j].x, ra[j].y);
        dst[1] = make_float2(ra[j].z, ra[j].w);
      }
    }
  };

  // B operands of one k-step: rows c0 + 4 ks + lq of W[k], NT consecutive output channels starting at col0.  Each
  // k-step's NT values are ONE register tuple (the destination of one global_load_dwordxNT): kept as a vector value the
  // loop carries it as a tuple, and the reload lands in the registers the next step's matrix ops read (as NT separate
  // floats the tuple was copied element by element at the loop back-edge, behind an s_waitcnt vmcnt(0))
  typedef float bvec_t __attribute__((ext_vector_type(NT)));
  typedef float bvec_load_t __attribute__((ext_vector_type(NT), aligned(4)));
  bvec_t b[KC / 4];
  // FAST: wstep = first weight row of the step (wave-uniform -> scalar registers), b_off = this lane's constant offset;
  // per k-step only a scalar add remains.  K-steps past the channel tail re-read row 0 of the step (never multiplied).
  const int b_off = lq * Cout + col0;
  const uint32_t b_off_bytes = (uint32_t)b_off * 4u;
  auto step_weights = [&](int k, int c0) -> const float* { return p.W + ((int64_t)k * Cin + c0) * Cout; };
  // valid A columns of the step starting at (k, c0): the rest of the chunk is zero padding
  auto cols_of = [&](int k, int c0) { return CPO ? min(GK, K - k) * CPO : min(KC, Cin - c0); };
  auto ksteps_of = [&](int k, int c0) { return (cols_of(k, c0) + 3) >> 2; };
  auto load_b = [&](const float* wstep, int k, int c0, int ksteps_valid, int ks, bvec_t& dst) {
    if (FAST) {
      // SGPR offset = the k-step's first weight row, VGPR offset = this lane's constant (row lq, column col0)
      const uint32_t row = (uint32_t)(k * Cin + c0 + (ks < ksteps_valid ? 4 * ks : 0));
      dst = buffer_load_floats<NT>(rsrc_w, b_off_bytes, row * (uint32_t)Cout * 4u);
      (void)wstep;
    } else {
      // W row of A column i of the step: (k * Cin + c0 + i) - fused offsets are consecutive row blocks of W
      const int i = 4 * ks + lq;
      const float* src = p.W + ((int64_t)k * Cin + c0 + i) * Cout + col0;
      const bool row_ok = i < cols_of(k, c0);
#pragma unroll
      for (int n = 0; n < NT; ++n) dst[n] = (row_ok && col0 + n < Cout) ? src[n] : 0.0f;
    }
  };

  __syncthreads();  // idx_s visible
  if (have_n) {
    // ---- prologue: operands of step 0 (cur).  x = the step after cur; with GDEPTH = 2 its gathers are already in flight
    //      and the iterator (n) stands one step further
    int k_c = k_n, c_c = c_n;
    uint32_t sm_c = sm_n;
    load_a(ra0, k_c, c_c, sm_c);
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) load_b(step_weights(k_c, c_c), k_c, c_c, ksteps_of(k_c, c_c), ks, b[ks]);
    store_a(ra0, As, sm_c);
    advance();
    int k_x = k_n, c_x = c_n;
    uint32_t sm_x = sm_n;
    bool have_x = have_n;
    if (GDEPTH == 2) {
      if (FAST || have_x) load_a(ra1, have_x ? k_x : 0, have_x ? c_x : 0, have_x ? sm_x : 0u);
      if (have_n) advance();
    }
    __syncthreads();
    int buf = 0;
    if (p.trace) trace_t_loop = wall_clock64();

    // one pipeline step; r_issue receives the gathers requested at its start, r_store holds the rows of step x
    auto step = [&](float4 (&r_issue)[A_F4], float4 (&r_store)[A_F4]) -> bool {
      // ---- gathers go out first: they land during matrix work and are written to the other LDS buffer at the end of
      //      this step (GDEPTH 1) or of the next one (GDEPTH 2).  (Issued here rather than after the barrier so that
      //      a conservative wait on the loop back-edge never waits for a gather that was just issued.)
      if (ABL & 2) {
      } else if (GDEPTH == 1) {
        if (FAST || have_x) load_a(r_issue, have_x ? k_x : 0, have_x ? c_x : 0, have_x ? sm_x : 0u);
      } else {
        const bool have2 = have_x && have_n;
        if (FAST || have2) load_a(r_issue, have2 ? k_n : 0, have2 ? c_n : 0, have2 ? sm_n : 0u);
      }
      // ---- MFMA over the current step; A operand reads run one k-step ahead of the matrix ops; each k-step's B
      //      registers are refilled for step x as soon as the matrix ops that read them are issued
      const float* a_base = As + buf * (TM_ * SA) + (wm * MR * 16 + li) * SA + lq;
      const uint32_t smw = (sm_c >> (wm * MR)) & ((1u << MR) - 1u);
      const int ksteps = ksteps_of(k_c, c_c);
      const int kb = have_x ? k_x : 0;
      const int cb = have_x ? c_x : 0;
      const float* wnext = step_weights(kb, cb);
      const int ksteps_next = ksteps_of(kb, cb);
      // A operands run PFD k-steps ahead of the matrix ops (register ring when PFD > 1): one k-step of a small tile is
      // only MR * NT * 32 cycles of matrix work, less than an LDS round trip, and on the small pyramid levels a wave
      // has no co-resident wave to hide that latency behind.
      // FULL (every step uses all KC columns: Cin % KC == 0, chosen at launch): compile-time trip count and plain
      // weight-row addressing, i.e. no per-k-step compare / select / branch in the hot loop.
      {
        auto reload_b = [&](int ks) {
          if (ABL & 1) return;
          if (FULL) {
            b[ks] = buffer_load_floats<NT>(rsrc_w, b_off_bytes, (uint32_t)(kb * Cin + cb + 4 * ks) * (uint32_t)Cout * 4u);
          } else if (FAST || have_x) {
            load_b(wnext, kb, cb, ksteps_next, ks, b[ks]);
          }
        };
        auto mfma_row = [&](int ks, const float (&a)[MR]) {
#pragma unroll
          for (int s = 0; s < MR; ++s) {
            // a tile with ONE sub-tile per wave row group is only visited for offsets where that sub-tile is active
            // (steps with an empty submask are skipped), so the test is compile-time true there
            if ((MR == 1 && Cfg::WAVES_M == 1) || ((smw >> s) & 1u)) {
#pragma unroll
              for (int n = 0; n < NT; ++n)
                acc[s][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[ks][n], acc[s][n], 0, 0, 0);
            }
          }
        };
        if constexpr (PFD == 1) {
          float a_cur[MR];
#pragma unroll
          for (int s = 0; s < MR; ++s) a_cur[s] = a_base[s * 16 * SA];
#pragma unroll
          for (int ks = 0; ks < KC / 4; ++ks) {
            if (FULL || ks < ksteps) {
              float a_nx[MR];
              if (ks + 1 < KC / 4) {
#pragma unroll
                for (int s = 0; s < MR; ++s) a_nx[s] = (ABL & 8) ? a_cur[s] : a_base[s * 16 * SA + (ks + 1) * 4];
                if (FULL) __builtin_amdgcn_sched_barrier(0);  // operand reads go out first
              }
              mfma_row(ks, a_cur);
              if (ks + 1 < KC / 4) {
#pragma unroll
                for (int s = 0; s < MR; ++s) a_cur[s] = a_nx[s];
              }
            }
            reload_b(ks);
            // straight-line code (FULL): keep the written interleaving of operand reads, matrix ops and weight reloads -
            // left alone, the scheduler batches the reloads behind all matrix ops and their latency is exposed
            if (FULL) __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          float a_ring[PFD + 1][MR];
#pragma unroll
          for (int d = 0; d < PFD; ++d)
#pragma unroll
            for (int s = 0; s < MR; ++s) a_ring[d][s] = a_base[s * 16 * SA + d * 4];
#pragma unroll
          for (int ks = 0; ks < KC / 4; ++ks) {
            if (FULL || ks < ksteps) {
              if (ks + PFD < KC / 4) {
#pragma unroll
                for (int s = 0; s < MR; ++s)
                  a_ring[(ks + PFD) % (PFD + 1)][s] = a_base[s * 16 * SA + (ks + PFD) * 4];
                if (FULL) __builtin_amdgcn_sched_barrier(0);
              }
              mfma_row(ks, a_ring[ks % (PFD + 1)]);
            }
            reload_b(ks);
            if (FULL) __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      ++trace_steps;
      if (!have_x) return false;
      // ---- hand over to step x
      if (!(ABL & 2)) store_a(r_store, As + (buf ^ 1) * (TM_ * SA), sm_x);
      k_c = k_x;
      c_c = c_x;
      sm_c = sm_x;
      if (GDEPTH == 1) advance();
      k_x = k_n;
      c_x = c_n;
      sm_x = sm_n;
      have_x = have_n;
      if (GDEPTH == 2 && have_n) advance();
      if (!(ABL & 4)) __syncthreads();
      buf ^= 1;
      return true;
    };
    if constexpr (GDEPTH == 1) {
      while (step(ra0, ra0)) {
      }
    } else {
      for (;;) {
        if (!step(ra0, ra1)) break;
        if (!step(ra1, ra0)) break;
      }
    }
  }

  if (p.trace) trace_t_epi = wall_clock64();
  // ---- epilogue: BN(eval)/bias -> residual -> activation -> store.  C/D map: MFMA col = lane&15, row = (lane>>4)*4+reg
  float sc[NT], sh[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const bool ok = col0 + n < Cout;
    sc[n] = (p.scale && ok) ? p.scale[col0 + n] : 1.0f;
    sh[n] = (p.shift && ok) ? p.shift[col0 + n] : 0.0f;
  }
  if constexpr (FAST && (ABL & 16) != 0) {
    float t = 0.f;
#pragma unroll
    for (int s = 0; s < MR; ++s)
#pragma unroll
      for (int n = 0; n < NT; ++n) t += acc[s][n][0] + acc[s][n][1] + acc[s][n][2] + acc[s][n][3];
    if (t == 12345.678f) p.out[0] = t;
  } else if constexpr (FAST) {
    epilogue_buffered<MR, NT>(p, acc, row0 + wm * MR * 16, lq, col0);
  } else
#pragma unroll
  for (int s = 0; s < MR; ++s) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int64_t r = row0 + wm * MR * 16 + s * 16 + lq * 4 + reg;
      int64_t o;
      if (p.perm)
        o = p.perm[r];
      else
        o = (r < p.V_out) ? r : -1;
      if (o < 0) continue;
      float y[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        float v = acc[s][n][reg];
        if (p.scale)
          v = __builtin_fmaf(v, sc[n], sh[n]);
        else if (p.shift)
          v = v + sh[n];
        y[n] = v;
      }
      if (p.residual) {
        const float* res = p.residual + o * p.res_ld + col0;
#pragma unroll
        for (int n = 0; n < NT; ++n)
          if (col0 + n < Cout) y[n] = y[n] + res[n];
      }
      float* dst = p.out + o * p.out_ld + col0;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        float v = y[n];
        if (p.act == SV_ACT_RELU)
          v = v < 0.f ? 0.f : v;  // NaN stays NaN, as torch.relu
        else if (p.act == SV_ACT_LEAKY_RELU)
          v = v > 0.f ? v : v * p.slope;
        if (col0 + n < Cout) dst[n] = v;
      }
    }
  }
  if (p.trace && tid == 0) {  // 100 MHz wall clock; HW_REG_HW_ID (CU / SE) and HW_REG_XCC_ID place the workgroup
    unsigned long long* t = p.trace + (size_t)blockIdx.x * 4;
    t[0] = trace_t0;
    t[1] = wall_clock64();
    t[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
           (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    // steps | prologue (start -> first step) and epilogue-start offsets in 100 MHz ticks / 16 (experiments)
    const unsigned long long pro = trace_t_loop ? ((trace_t_loop - trace_t0) >> 4) & 0xffffull : 0ull;
    const unsigned long long epi = ((t[1] - trace_t_epi) >> 4) & 0xffffull;
    t[3] = (unsigned long long)(unsigned)trace_steps | (pro << 32) | (epi << 48);
  }
}

template <int TM_, int WAVES_N, int NT, bool FAST, int CPO = 0, bool RING = false, bool FULL = false>
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvParams p) {
  conv_tile_body<TM_, WAVES_N, NT, FAST, CPO, RING, FULL>(p, (int)blockIdx.x, 0);
}

// Dual-body launch for chip-filling layers: the hardware hands out workgroups in blockIdx order, so the grid is the plan
// tiles in longest-first order as TM_-row tiles followed by the CHEAPEST plan tiles as TAIL_TM-row tiles.  A launch ends
// with every CU's residency decaying from 4 workgroups to 0 over about the duration of its last (cheapest) tiles; with
// half-height tiles at the end that decay is half as long (tools/wg_trace.py: ~22 % of a level-0 launch, i.e. ~11 % of
// its slot-time idle).  Everything else about a tile - offsets visited, the (k, c) order of every output element's fma
// chain, hence every result bit - is independent of the tile height.
template <int TM_, int TAIL_TM, int WAVES_N, int NT, bool FAST>
__global__ __launch_bounds__(256) void conv_fwd_dual_kernel(ConvParams p) {
  if ((int)blockIdx.x < p.main_blocks)
    conv_tile_body<TM_, WAVES_N, NT, FAST>(p, (int)blockIdx.x, 0);
  else
    conv_tile_body<TAIL_TM, WAVES_N, NT, FAST>(p, (int)blockIdx.x - p.main_blocks, p.main_tiles128);
}

// ---- narrow-output dense layer (K = 1, identity rows, Cout <= 4: the last Linear of the classification heads,
//      model/robotnet_segmentation.py:43-48).  1.5 flop per byte: an HBM stream, not a matrix problem - on the MFMA
//      tiles 13 of 16 output columns would be padding and the step chain (gather -> LDS -> barrier) is latency-bound.
//      Here a workgroup streams ROWS (64) rows: [ROWS x 32 channels] stages are read coalesced (8 lanes per 128-byte row
//      segment), double-buffered through LDS (row stride 33 words: conflict-free), and thread r walks row r with the
//      same ascending-channel fmaf chain as everywhere else (weights are wave-uniform -> scalar loads).
template <int C, int ROWS>
__global__ __launch_bounds__(256) void linear_narrow_kernel(ConvParams p) {
  constexpr int COLS = 32, SW = COLS + 1, LPT = ROWS / 32;  // LPT float4 loads per thread and stage
  __shared__ float tile[2][ROWS * SW];
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int Cin = p.Cin;
  const int lr = tid >> 3, lc = (tid & 7) * 4;  // load mapping: row lr + 32 i, channels lc .. lc + 3
  float4 ra[LPT];
  auto load = [&](int c0) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int64_t r = row0 + lr + 32 * i;
      const int c = c0 + lc;
      const bool ok = r < p.V_out && c < Cin;
      const float4 v = *(const float4*)(p.in + (ok ? r * p.in_ld + c : 0));
      ra[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store = [&](float* dst) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      float* d = dst + (lr + 32 * i) * SW + lc;
      d[0] = ra[i].x;
      d[1] = ra[i].y;
      d[2] = ra[i].z;
      d[3] = ra[i].w;
    }
  };
  float acc[C];
#pragma unroll
  for (int j = 0; j < C; ++j) acc[j] = 0.0f;
  load(0);
  store(tile[0]);
  __syncthreads();
  int buf = 0;
  for (int c0 = 0; c0 < Cin; c0 += COLS) {
    const bool more = c0 + COLS < Cin;
    if (more) load(c0 + COLS);
    if (tid < ROWS) {
      const float* x = tile[buf] + tid * SW;
      const float* w = p.W + (int64_t)c0 * p.Cout;
      const int nc = min(COLS, Cin - c0);  // channels beyond Cin hold zeros, but their weights would be out of bounds
#pragma unroll 8
      for (int c = 0; c < nc; ++c) {
        const float xv = x[c];
#pragma unroll
        for (int j = 0; j < C; ++j) acc[j] = __builtin_fmaf(xv, w[c * p.Cout + j], acc[j]);
      }
    }
    if (more) store(tile[buf ^ 1]);
    __syncthreads();
    buf ^= 1;
  }
  const int64_t r = row0 + tid;
  if (tid >= ROWS || r >= p.V_out) return;
#pragma unroll
  for (int j = 0; j < C; ++j) {
    float v = acc[j];
    if (p.scale)
      v = __builtin_fmaf(v, p.scale[j], p.shift ? p.shift[j] : 0.0f);
    else if (p.shift)
      v = v + p.shift[j];
    if (p.residual) v = v + p.residual[r * p.res_ld + j];
    if (p.act == SV_ACT_RELU)
      v = v < 0.f ? 0.f : v;  // NaN stays NaN, as torch.relu
    else if (p.act == SV_ACT_LEAKY_RELU)
      v = v > 0.f ? v : v * p.slope;
    p.out[r * p.out_ld + j] = v;
  }
}

// ---- thin gather-bound layers (Cin = 32 -> Cout = 32: block1's four convs, conv1p1s2, conv2p2s2;
//      model/backbone/minkunet.py:59-71).  14 flop per gathered byte: the layer is its gather, and a tile's life in the
//      LDS-staged kernel above is a chain of dependent round trips (neighbour table -> barrier -> gather -> LDS -> barrier,
//      then one gather per step).  Here ONE WAVE owns a 16-row sub-tile and never synchronises with anybody:
//        * the sub-tile's active offsets are compacted into a list (ballot over the plan's submask words);
//        * per active offset every lane gathers the CIN channels of ITS row straight into registers - lane (row li,
//          group lq) loads float4s at channels 16 j + 4 lq .. + 3 - with D offsets in flight (register ring);
//        * the matrix op wants lane group lq to hold channel 4 m + lq for op m: a 4 x 4 transpose between the four lane
//          groups and four registers, done with two v_permlane32_swap + two v_permlane16_swap per float4 (gfx950), so
//          the accumulation chain keeps its (offset ascending, channel ascending) order and every bit of the result;
//        * weights (wave-uniform per offset, 4 KB) come from L1/L2 one offset ahead.
//      No LDS traffic for the features, no barrier, D gathers in flight per wave from its first microsecond on.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void transpose4x4_lanegroups(float& r0, float& r1, float& r2, float& r3) {
  // in: lane group g (16 lanes) holds X[g][i] in r_i; out: lane group g holds X[i][g] in r_i
  u32x2_t a = __builtin_amdgcn_permlane32_swap(__float_as_uint(r0), __float_as_uint(r2), false, false);
  u32x2_t b = __builtin_amdgcn_permlane32_swap(__float_as_uint(r1), __float_as_uint(r3), false, false);
  u32x2_t c = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
  u32x2_t d = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
  r0 = __uint_as_float(c.x);
  r1 = __uint_as_float(c.y);
  r2 = __uint_as_float(d.x);
  r3 = __uint_as_float(d.y);
}

template <int CIN, int COUT, int MR, int D, int CSPLIT = 1>
__global__ __launch_bounds__(256) void conv_thin_kernel(ConvParams p) {
  // CSPLIT > 1: blockIdx.y selects a slice of COUT / CSPLIT output channels (the gathers are repeated per slice - they hit
  // the L2 - but a wave's matrix work and weight registers shrink by CSPLIT and the launch has CSPLIT times the waves: on
  // the small pyramid levels a launch is a few waves per SIMD, bound by one wave's chain of offsets)
  constexpr int NT = COUT / 16 / CSPLIT;  // MFMA column tiles per wave; interleaved: column li of tile n = channel NT * li + n
  const int cbase = (int)blockIdx.y * (COUT / CSPLIT);
  constexpr int KS = CIN / 4;    // k-steps per offset
  constexpr int G4 = CIN / 16;   // float4 gathers per lane, row and offset
  // MR: 16-row sub-tiles per wave (they share every weight register: half the weight traffic per row at MR = 2);
  // D: offsets in flight per wave (gather ring)
  static_assert(CIN % 16 == 0 && COUT % 16 == 0 && NT >= 1 && NT <= 4 && (MR == 1 || MR == 2), "shape");
  typedef float bvec_t __attribute__((ext_vector_type(NT)));
  __shared__ int idx_s[4][32 * 16 * MR];
  __shared__ int klist_s[4][32];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int K = p.K;
  constexpr int ROWS = 16 * MR;
  // (giving every XCD a contiguous eighth of the plan, for L2 locality of the gathers, measured no different: 24 vs 25 us
  //  at level 1, 53 vs 49 us at level 0)
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wid) * ROWS;  // Vpad is a multiple of 128: every wave has its rows
  const int64_t t128 = row0 / PLAN_TILE;
  const int sub = (int)((row0 % PLAN_TILE) / 16);
  // ---- active offsets of this wave's sub-tile(s), compacted; neighbour rows of its rows for every offset
  int o_pre[MR][4];
  load_perm_rows<MR>(p, row0, lq, o_pre);
  const bool active = lane < K && ((p.submask[t128 * K + lane] >> sub) & ((1u << MR) - 1u));
  const unsigned long long amask = __ballot(active);
  const int nact = __popcll(amask);
  if (active) klist_s[wid][__popcll(amask & ((1ull << lane) - 1ull))] = lane;
  // the wave's neighbour table: all entries requested before the first is stored (a rolled loop is a chain of K / 4
  // dependent round trips - a third of this kernel's life at level 1), kept as BYTE offsets of the rows in `in`
  // (absent: beyond the extent, the gather then returns zeros)
  {
    constexpr int ST = 32 * ROWS / 64;  // K <= 32
    int n_st[ST];
#pragma unroll
    for (int it = 0; it < ST; ++it) {
      const int e = lane + 64 * it;
      n_st[it] = e < K * ROWS ? p.nbr_s[(int64_t)(e / ROWS) * p.Vpad + row0 + (e % ROWS)] : -1;
    }
#pragma unroll
    for (int it = 0; it < ST; ++it) {
      const int e = lane + 64 * it;
      if (e < K * ROWS) idx_s[wid][e] = (int)(n_st[it] >= 0 ? (uint32_t)n_st[it] * (uint32_t)(p.in_ld * 4) : BUF_ABSENT);
    }
  }
  __builtin_amdgcn_wave_barrier();  // same-wave LDS traffic is ordered; keep the compiler from moving reads above
  const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)p.w_bytes, 0x00020000);

  f32x4 acc[MR][NT];
#pragma unroll
  for (int s = 0; s < MR; ++s)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[s][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 g[D][MR][G4];
  bvec_t b[2][KS];
  // Every load below is unconditional (slots past the end of the list re-read the last offset's weights and read zeros
  // through the range check, their matrix ops then add fma(0, w, acc) = acc): straight-line code, so hipcc can count its
  // s_waitcnt vmcnt instead of draining the ring at every control-flow merge.
  const int last = nact > 0 ? nact - 1 : 0;
  auto offset_of = [&](int j) { return __builtin_amdgcn_readfirstlane(klist_s[wid][min(j, last)]); };
  auto issue = [&](int j, float4 (&dst)[MR][G4]) {
    const int k = offset_of(j);
#pragma unroll
    for (int s = 0; s < MR; ++s) {
      // an absent neighbour (and a ring slot past the end of the list) reads beyond the buffer's extent: zeros from the
      // range check, no select behind the load (with `ok ? value : 0` hipcc sinks the load under the condition)
      const uint32_t off = j < nact ? (uint32_t)idx_s[wid][k * ROWS + s * 16 + li] : BUF_ABSENT;
#pragma unroll
      for (int jj = 0; jj < G4; ++jj) {
        const f32x4 v = buffer_load_floats<4>(rsrc_in, off + 16u * (uint32_t)lq, 64u * (uint32_t)jj);
        dst[s][jj] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  };
  auto load_w = [&](int j, bvec_t (&dst)[KS]) {
    const uint32_t wk = (uint32_t)offset_of(j) * (uint32_t)(CIN * COUT * 4);  // wave-uniform: the SGPR offset
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      dst[ks] = buffer_load_floats<NT>(rsrc_w, (uint32_t)(lq * COUT + cbase + NT * li) * 4u, wk + (uint32_t)(4 * ks * COUT * 4));
  };
  auto compute = [&](float4 (&a)[MR][G4], bvec_t (&w)[KS]) {
#pragma unroll
    for (int jj = 0; jj < G4; ++jj) {
      float am[MR][4];
#pragma unroll
      for (int s = 0; s < MR; ++s) {
        am[s][0] = a[s][jj].x; am[s][1] = a[s][jj].y; am[s][2] = a[s][jj].z; am[s][3] = a[s][jj].w;
        transpose4x4_lanegroups(am[s][0], am[s][1], am[s][2], am[s][3]);  // [m]: channel 16 jj + 4 m + lq of row li
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int s = 0; s < MR; ++s)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[s][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(am[s][m], w[4 * jj + m][n], acc[s][n], 0, 0, 0);
    }
  };
  if (nact > 0) {
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, g[d]);
    load_w(0, b[0]);
    constexpr int U = (D % 2 == 0) ? D : 2 * D;  // unroll so that ring slot and weight buffer indices are static
    for (int j0 = 0; j0 < nact; j0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = j0 + u;
        load_w(j + 1, b[(u + 1) & 1]);
        compute(g[u % D], b[u & 1]);
        issue(j + D, g[u % D]);
      }
    }
  }
  // ---- epilogue (shared with conv_tile_body): C/D map: MFMA col = lane & 15, row = (lane >> 4) * 4 + reg
  epilogue_buffered<MR, NT, true>(p, acc, row0, lq, cbase + NT * li, o_pre);
}

// ---- first layer on the matrix pipe (conv0: 3 -> 32, every voxel of the frame; model/backbone/minkunet.py:55-57).
//      The 27 x 3 = 81 (offset, channel) products of an output element are ONE ascending chain, so the layer is a
//      [V x 81] x [81 x 32] product whose A rows are gathered: 21 k-steps of 4 (the last three columns read zeros).  One
//      wave owns a 16-row sub-tile and never synchronises with anybody: lane (row li, group lq) fetches element
//      e = 4 ks + lq of its row - channel e % 3 of the neighbour at offset e / 3 - for all 21 k-steps at once (21 dword
//      gathers in flight per lane behind ONE round trip for the wave's neighbour table), the 81 x 32 weights arrive in the
//      matrix-op layout straight from L2 meanwhile (requested before the table), then 42 matrix ops.  Against the
//      thread-per-voxel VALU kernel: 5.4 instead of 1.3 waves per SIMD and 1 344 instead of 2 592 issue cycles per 16 rows.
//      Same chain order (k ascending, c ascending), absent neighbours contribute fma(0, w, acc) = acc.
template <int COUT>
__global__ __launch_bounds__(256) void conv_first_mfma_kernel(ConvParams p) {
  constexpr int CIN = 3, KMAX = 27, E = KMAX * CIN, KS = (E + 3) / 4, NT = COUT / 16;
  constexpr int SX = 85;  // LDS row stride of the gathered [16 rows][81] block: 85 = 21 mod 32 spreads the rows over the banks
  constexpr int PAIRS = KMAX * 16, IT = (PAIRS + 63) / 64;
  typedef float bvec_t __attribute__((ext_vector_type(NT)));
  __shared__ float xs[4][16 * SX];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int K = p.K;
  const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)p.w_bytes, 0x00020000);
  // weights first: they depend on nothing.  Row e of the flat [K * 3][COUT] weight block; rows past the extent (e >= 3 K)
  // read zeros through the range check
  bvec_t b[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    b[ks] = buffer_load_floats<NT>(rsrc_w, (uint32_t)(lq * COUT + NT * li) * 4u, (uint32_t)(4 * ks * COUT * 4));
  // A workgroup of this kernel lives ~1.7 us, and the chip starts only ~126 of them per microsecond (PMC: 0.8 waves per SIMD
  // on average at 5 542 workgroups): a launch of one workgroup per 64 rows is bound by the dispatcher, not by its memory
  // chain.  So the grid is capped and every wave walks several sub-tiles (64-row groups blockIdx.x, + gridDim.x, ...) with
  // its weights loaded once; the LDS block is the wave's own and same-wave LDS traffic is ordered.
  const int ngroups = (int)(p.Vpad / 64);
  for (int grp = (int)blockIdx.x; grp < ngroups; grp += (int)gridDim.x) {
  const int64_t row0 = ((int64_t)grp * 4 + wid) * 16;  // Vpad is a multiple of 128: every wave has its rows
  int o_pre[1][4];
  load_perm_rows<1>(p, row0, lq, o_pre);
  // the wave's 27 x 16 (offset, row) pairs, one per lane and pass: consecutive lanes = consecutive rows of one offset, so
  // the index reads are coalesced, and each pair is ONE 12-byte gather (a third of the requests of per-element gathers:
  // at four frames per tensor the per-element form was bound by the texture addresser, 0.32 of the HBM peak)
  int n_st[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int pr = lane + 64 * it;
    n_st[it] = (pr < K * 16) ? p.nbr_s[(int64_t)(pr >> 4) * p.Vpad + row0 + (pr & 15)] : -1;
  }
  typedef float f32x3 __attribute__((ext_vector_type(3)));
  f32x3 g[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it)
    g[it] = buffer_load_floats<3>(rsrc_in, n_st[it] >= 0 ? (uint32_t)n_st[it] * (uint32_t)(p.in_ld * 4) : BUF_ABSENT, 0u);
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int pr = lane + 64 * it;
    if (pr < PAIRS) {
      float* d = &xs[wid][(pr & 15) * SX + 3 * (pr >> 4)];
      d[0] = g[it][0];
      d[1] = g[it][1];
      d[2] = g[it][2];
    }
  }
  __builtin_amdgcn_wave_barrier();  // same-wave LDS traffic is ordered; keep the compiler from moving reads above
  float a[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int e = 4 * ks + lq;
    a[ks] = e < E ? xs[wid][li * SX + e] : 0.0f;
  }
  f32x4 acc[1][NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[0][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], b[ks][n], acc[0][n], 0, 0, 0);
  epilogue_buffered<1, NT, true>(p, acc, row0, lq, NT * li, o_pre);
  }
}

// ---- the thin layers with the layer's WHOLE weight tensor resident in LDS (round 4; conv_thin_kernel above is kept behind
//      SV_THIN_VARIANT for A/B).  What the counters said about conv_thin_kernel at 88k voxels (profiles/r04_pmc_thin_*.txt):
//      matrix pipe 29 % busy, waves 66 % of their life stalled at issue, and the CU's vector-memory pipe (TCP) handling
//      17.4 M cache accesses per launch = 68 k cycles per CU of a 110 k-cycle launch - two thirds of them WEIGHT loads: every
//      wave fetched the 4 KB weight block of every offset it visited through L1 (251 MB per launch for a 110 KB tensor),
//      and a wave alone on its SIMD waited an L2 round trip per offset for them (one offset of look-ahead is shorter).
//      27 x 32 x 32 floats are 110 KB: they fit the CU's 160 KB LDS beside the waves' neighbour tables.  So: ONE workgroup
//      of 16 waves per CU stages the weights once (coalesced float4, ~1 us), then every wave walks 16-row sub-tiles
//      (longest plan tiles first, wave w of workgroup b takes sub-tiles b + G w, b + G (w + 16), ...) exactly as
//      conv_thin_kernel does - compacted offset list, D gathers in flight, lane-group transposes, the (k, c) chain order
//      and therefore every result bit - with its B operands read from LDS (one conflict-free ds_read_b64 per k-step) and
//      only the row gathers left on the vector-memory path.
template <int CIN, int COUT, int D, int WAVES, bool PREFETCH>
__global__ __launch_bounds__(WAVES * 64) void conv_thin_lds_kernel(ConvParams p, int n_sub) {
  constexpr int NT = COUT / 16, KS = CIN / 4, G4 = CIN / 16;
  constexpr int ST = 32 * 16 / 64;  // table entries per lane (K <= 32)
  static_assert(CIN % 16 == 0 && COUT % 16 == 0 && NT >= 1 && NT <= 4, "shape");
  typedef float bvec_t __attribute__((ext_vector_type(NT)));
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int q_head;  // the workgroup's tile queue: next unassigned position of its tile list
  const int K = p.K;
  float* w_s = lds;                                     // [K][CIN][COUT]
  int* idx_all = (int*)(lds + (size_t)K * CIN * COUT);  // [WAVES][32 * 16] neighbour rows of a wave's sub-tile (byte offsets)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  // SV_THIN_TRACE experiments: per-wave cycle stamps {start, weights staged, first tile: table staged / loop done / stored,
  // end, tiles, offsets of the first tile}
  unsigned long long tr[6] = {0, 0, 0, 0, 0, 0}, tr_it[3] = {0, 0, 0};
  int tr_tiles = 0, tr_nact = 0;
  if (p.trace) tr[0] = __builtin_amdgcn_s_memtime();
  if (tid == 0) q_head = WAVES;  // positions 0 .. WAVES-1 are the waves' first tiles
  // The workgroup's tile list: position i = sub-tile blockIdx.x + gridDim.x * i in LONGEST-FIRST order (plan tile
  // tile_order[t / 8], its sub-tile t % 8): every workgroup gets the same mix of long and short tiles; inside the workgroup
  // the waves PULL positions from q_head (an LDS atomic: no global counter to reset), so a CU's sixteen waves finish within
  // one tile of each other.
  struct Tile {
    int t128, sub;   // wave-uniform (scalar registers)
    int perm[1][4];
    uint32_t sm;     // lane k: the tile's submask word of offset k
    int n_st[ST];    // neighbour rows (lane + 64 it: offset e / 16, row e % 16)
  };
  auto fetch = [&](int pos, Tile& T) -> bool {  // requests a tile's table; nothing waits here
    const int t = (int)blockIdx.x + (int)gridDim.x * pos;
    if (t >= n_sub) return false;
    T.t128 = __builtin_amdgcn_readfirstlane(p.tile_order ? p.tile_order[t >> 3] : (t >> 3));
    T.sub = t & 7;
    const int64_t row0 = (int64_t)T.t128 * PLAN_TILE + T.sub * 16;
    load_perm_rows<1>(p, row0, lq, T.perm);
    T.sm = lane < K ? p.submask[(int64_t)T.t128 * K + lane] : 0u;
#pragma unroll
    for (int it = 0; it < ST; ++it) {
      const int e = lane + 64 * it;
      T.n_st[it] = e < K * 16 ? p.nbr_s[(int64_t)(e >> 4) * p.Vpad + row0 + (e & 15)] : -1;
    }
    return true;
  };
  Tile cur;
  bool have = false;
  if (PREFETCH) have = fetch(wid, cur);  // on its way while the weights are staged
  // ---- the layer's weights, once per workgroup
  {
    const float4* src = (const float4*)p.W;
    float4* dst = (float4*)w_s;
    const int n4 = K * CIN * COUT / 4;
    for (int i = tid; i < n4; i += WAVES * 64) dst[i] = src[i];
  }
  __syncthreads();
  if (p.trace) tr[1] = __builtin_amdgcn_s_memtime();
  if (!PREFETCH) have = fetch(wid, cur);
  int* idx_s = idx_all + wid * (32 * 16);
  const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const float* w_lane = w_s + lq * COUT + NT * li;
  while (have) {
    // ---- this tile's table into LDS as byte offsets (absent: beyond the extent -> the gather returns zeros)
#pragma unroll
    for (int it = 0; it < ST; ++it) {
      const int e = lane + 64 * it;
      if (e < K * 16) idx_s[e] = (int)(cur.n_st[it] >= 0 ? (uint32_t)cur.n_st[it] * (uint32_t)(p.in_ld * 4) : BUF_ABSENT);
    }
    // active offsets of the sub-tile: a scalar bit mask walked with bit scans (no list in memory; two cursors: the gathers
    // run D offsets ahead of the matrix ops, the weight reads one)
    const uint32_t amask = (uint32_t)__ballot((cur.sm >> cur.sub) & 1u);
    const int nact = __builtin_popcount(amask);
    const int64_t row0 = (int64_t)cur.t128 * PLAN_TILE + cur.sub * 16;
    int o_pre[1][4] = {{cur.perm[0][0], cur.perm[0][1], cur.perm[0][2], cur.perm[0][3]}};
    __builtin_amdgcn_wave_barrier();
    if (p.trace && tr_tiles == 0) {
      tr[2] = __builtin_amdgcn_s_memtime();
      tr_nact = nact;
    }
    f32x4 acc[1][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[0][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 g[D][G4];
    bvec_t b[2][KS];
    uint32_t g_rest = amask, w_rest = amask;
    const int k_last = amask ? 31 - __builtin_clz(amask) : 0;
    auto issue = [&](float4 (&dst)[G4]) {  // gathers of the next offset of the gather cursor (past the end: zeros, no access)
      const bool any = g_rest != 0u;
      const int k = any ? __builtin_ctz(g_rest) : k_last;
      g_rest &= g_rest - 1u;
      const uint32_t off = any ? (uint32_t)idx_s[k * 16 + li] : BUF_ABSENT;
#pragma unroll
      for (int jj = 0; jj < G4; ++jj) {
        const f32x4 v = buffer_load_floats<4>(rsrc_in, off + 16u * (uint32_t)lq, 64u * (uint32_t)jj);
        dst[jj] = make_float4(v[0], v[1], v[2], v[3]);
      }
    };
    auto load_w = [&](bvec_t (&dst)[KS]) {  // B operands of the weight cursor's next offset, from LDS
      const int k = w_rest ? __builtin_ctz(w_rest) : k_last;
      w_rest &= w_rest - 1u;
      const float* wk = w_lane + k * (CIN * COUT);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) dst[ks] = *(const bvec_t*)(wk + 4 * ks * COUT);
    };
    auto compute = [&](float4 (&a)[G4], bvec_t (&w)[KS]) {
#pragma unroll
      for (int jj = 0; jj < G4; ++jj) {
        float am[4] = {a[jj].x, a[jj].y, a[jj].z, a[jj].w};
#if !defined(SV_THIN_ABL) || !(SV_THIN_ABL & 4)
        transpose4x4_lanegroups(am[0], am[1], am[2], am[3]);  // [m]: channel 16 jj + 4 m + lq of row li
#endif
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(am[m], w[4 * jj + m][n], acc[0][n], 0, 0, 0);
      }
    };
    // the first gathers go out BEFORE the next tile's table is requested: vmcnt counts in order, so the first matrix ops
    // wait for their own rows only
#pragma unroll
    for (int d = 0; d < D; ++d) issue(g[d]);
    load_w(b[0]);
    // ---- pull the next tile and request its table: it arrives while this tile is multiplied
    Tile nxt;
    int pos = 0;
    bool have_nxt = false;
    if (PREFETCH) {
      if (lane == 0) pos = __hip_atomic_fetch_add(&q_head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      pos = __builtin_amdgcn_readfirstlane(pos);
      have_nxt = fetch(pos, nxt);
    }
    if (nact > 0) {
      constexpr int U = (D % 2 == 0) ? D : 2 * D;
      for (int j0 = 0; j0 < nact; j0 += U) {
        if (p.trace && tr_tiles == 0 && j0 < 3 * U) tr_it[j0 / U] = __builtin_amdgcn_s_memtime() + (unsigned long long)(acc[0][0][0] == 12345.678f);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#ifndef SV_THIN_ABL  // timing-only ablations (results WRONG): 1 = no gathers in the loop, 2 = no weight reads, 4 = no transposes
#define SV_THIN_ABL 0
#endif
          if (!(SV_THIN_ABL & 2)) load_w(b[(u + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);  // the weight reads of the NEXT offset go out before this offset's matrix ops
          compute(g[u % D], b[(SV_THIN_ABL & 2) ? 0 : (u & 1)]);
          if (!(SV_THIN_ABL & 1)) issue(g[u % D]);
        }
      }
    }
    if (p.trace && tr_tiles == 0) {
      // the accumulators are the loop's last results: reading one orders the stamp behind the matrix ops
      tr[3] = __builtin_amdgcn_s_memtime() + (unsigned long long)(acc[0][0][0] == 12345.678f);
    }
    epilogue_buffered<1, NT, true>(p, acc, row0, lq, NT * li, o_pre);
    __builtin_amdgcn_wave_barrier();  // the next sub-tile's table overwrites this one's
    if (p.trace && tr_tiles == 0) tr[4] = __builtin_amdgcn_s_memtime();
    ++tr_tiles;
    if (!PREFETCH) {
      if (lane == 0) pos = __hip_atomic_fetch_add(&q_head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      pos = __builtin_amdgcn_readfirstlane(pos);
      have_nxt = fetch(pos, nxt);
    }
    have = have_nxt;
    if (have_nxt) cur = nxt;
  }
  if (p.trace && lane == 0) {
    tr[5] = __builtin_amdgcn_s_memtime();
    unsigned long long* o = p.trace + ((size_t)blockIdx.x * WAVES + wid) * 12;
    for (int i = 0; i < 6; ++i) o[i] = tr[i];
    o[6] = (unsigned long long)tr_tiles;
    o[7] = (unsigned long long)tr_nact;
    for (int i = 0; i < 3; ++i) o[8 + i] = tr_it[i];
  }
}

template <int D, int WAVES, bool PREFETCH>
static int launch_conv_thin_lds_t(const ConvParams& p, hipStream_t stream) {
  static int n_cu = 0;
  const size_t lds = ((size_t)p.K * 32 * 32 + (size_t)WAVES * (32 * 16)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    int dev = 0;
    hipDeviceProp_t prop;
    SV_HIP(hipGetDevice(&dev));
    SV_HIP(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    SV_HIP(hipFuncSetAttribute((const void*)conv_thin_lds_kernel<32, 32, D, WAVES, PREFETCH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(((size_t)27 * 32 * 32 + (size_t)WAVES * (32 * 16)) * sizeof(float))));
    attr_set = true;
  }
  const int n_sub = (int)(p.Vpad / 16);
  const int grid = n_sub < n_cu ? n_sub : n_cu;
  static const bool trace = getenv("SV_THIN_TRACE") != nullptr;  // experiments only: phase stamps of one launch to stderr
  ConvParams q = p;
  q.trace = nullptr;
  if (trace) {
    SV_HIP(hipMalloc((void**)&q.trace, (size_t)grid * WAVES * 12 * sizeof(unsigned long long)));
    SV_HIP(hipMemsetAsync(q.trace, 0, (size_t)grid * WAVES * 12 * sizeof(unsigned long long), stream));
  }
  hipLaunchKernelGGL((conv_thin_lds_kernel<32, 32, D, WAVES, PREFETCH>), dim3((unsigned)grid), dim3(WAVES * 64), lds, stream, q, n_sub);
  note_instance("conv_thin_lds_kernel<32, 32>|fast=1,ring=0,full=1");
  SV_LAUNCH_CHECK();
  if (trace) {
    std::vector<unsigned long long> h((size_t)grid * WAVES * 12);
    SV_HIP(hipStreamSynchronize(stream));
    SV_HIP(hipMemcpy(h.data(), q.trace, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    SV_HIP(hipFree(q.trace));
    std::vector<double> stage, table, loop, store, life, per_off, it01, it12, pre;
    for (size_t w = 0; w < (size_t)grid * WAVES; ++w) {
      const unsigned long long* o = &h[w * 12];
      if (!o[0]) continue;
      if (o[8] && o[9] && o[10]) {
        pre.push_back((double)(o[8] - o[2]));
        it01.push_back((double)(o[9] - o[8]));
        it12.push_back((double)(o[10] - o[9]));
      }
      stage.push_back((double)(o[1] - o[0]));
      life.push_back((double)(o[5] - o[0]));
      if (o[6]) {
        table.push_back((double)(o[2] - o[1]));
        loop.push_back((double)(o[3] - o[2]));
        store.push_back((double)(o[4] - o[3]));
        if (o[7]) per_off.push_back((double)(o[3] - o[2]) / (double)o[7]);
      }
    }
    auto med = [](std::vector<double>& v) {
      if (v.empty()) return 0.0;
      std::sort(v.begin(), v.end());
      return v[v.size() / 2];
    };
    fprintf(stderr, "[thin trace] K=%d n_sub=%d grid=%d | median cycles: weights staged %.0f, first tile: table %.0f, loop %.0f "
            "(%.0f per offset; table staged -> loop %.0f, first 4 offsets %.0f, next 4 %.0f), store %.0f, wave life %.0f\n", p.K, n_sub,
            grid, med(stage), med(table), med(loop), med(per_off), med(pre), med(it01), med(it12), med(store), med(life));
  }
  return SV_OK;
}

static int launch_conv_thin_lds(const ConvParams& p, hipStream_t stream, int variant) {
  // <gathers in flight, waves per workgroup, next tile's table requested during the current tile>: measured at 88k / 26k voxels
  // (tools/hbm_layers_microbench.py, profiles/r04_thin_variants.txt): <2,16,no> 33.9 / 19.3 us (126 VGPRs, no spills),
  // <2,16,yes> 36.9 / 18.7, <4,16,no> 38.5 / 20.2 (14 registers spilled at the 128-VGPR limit of a 16-wave workgroup),
  // <4,8,no> 39.4 / 21.2, <4,12,yes> 39.9 / 19.5, <1,16,no> 36.8 / 21.9, <6,16,no> 43.3 / 23.1
  switch (variant) {  // experiments: SV_THIN_VARIANT=10, 20
    case 10: return launch_conv_thin_lds_t<4, 16, false>(p, stream);
    case 20: return launch_conv_thin_lds_t<2, 16, true>(p, stream);
    default: return launch_conv_thin_lds_t<2, 16, false>(p, stream);
  }
}

static int launch_conv_thin(const ConvParams& p, hipStream_t stream) {
  // one 16-row sub-tile per wave, four offsets in flight: 24 us for block1's 32->32 at level 1 (26.5k voxels) against 35 us
  // on the LDS-staged fused-offset tile, 49 against 84 us at 88k voxels (2.65 TB/s on algorithmic gather-bytes).  Two
  // sub-tiles per wave (shared weight registers) 27-28 / 48 us, three or six offsets in flight 26 / 51-54 us.
  static const int variant = getenv("SV_THIN_VARIANT") ? atoi(getenv("SV_THIN_VARIANT")) : -1;  // experiments only
  // Round 4: the layer's weights resident in LDS, one 16-wave workgroup per CU (conv_thin_lds_kernel) - for a GPU that holds
  // ONE frame (the caller said so: sv_conv_set_dispatch(want_scale >= 1), the per-frame InferenceEngine.predict path and
  // every measurement of a layer alone).  Its workgroup needs a whole CU - 145 KB of LDS, sixteen wave slots at 126 VGPRs -
  // and beside the convolutions of other frames a CU only drains completely when a launch ends: inside the three-stream
  // pipeline its launches wait 0.2-3.6 ms for CUs (profiles/r04_bench_kernel_by_grid.txt, r04_cfg5_kernel_by_grid.txt) and
  // predict_stream loses 4 % (profiles/r04_ab_thin_in_pipeline.txt); there the four-wave workgroups of conv_thin_kernel,
  // which fit the slot any finishing convolution workgroup frees, stay the choice.  Same bits either way.
  const bool alone = g_want_scale_override >= 1.0;
  if ((variant < 0 ? alone : variant >= 10) && p.K <= 27 && (((uintptr_t)p.W) & 15) == 0) return launch_conv_thin_lds(p, stream, variant);
  const dim3 g1((unsigned)(p.Vpad / 64)), g2((unsigned)(p.Vpad / 64), 2);
  switch (variant) {
    case 1: hipLaunchKernelGGL((conv_thin_kernel<32, 32, 1, 4, 2>), g2, dim3(256), 0, stream, p); break;
    case 2: hipLaunchKernelGGL((conv_thin_kernel<32, 32, 1, 6, 2>), g2, dim3(256), 0, stream, p); break;
    case 3: hipLaunchKernelGGL((conv_thin_kernel<32, 32, 1, 8, 2>), g2, dim3(256), 0, stream, p); break;
    case 4: hipLaunchKernelGGL((conv_thin_kernel<32, 32, 1, 8, 1>), g1, dim3(256), 0, stream, p); break;
    default: hipLaunchKernelGGL((conv_thin_kernel<32, 32, 1, 4>), g1, dim3(256), 0, stream, p); break;
  }
  note_instance("conv_thin_kernel<32, 32>|fast=1,ring=0,full=1");
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ---- first layer of the networks (conv0: Cin = 3 colour channels -> 32, 3x3x3, every voxel of the frame;
//      model/backbone/minkunet.py:55-57).  5.8 flop per byte of gather traffic: the layer is its gather.  One THREAD
//      per output voxel (in the plan's mask-sorted order, so a wavefront's rows share most neighbour offsets and an
//      offset nobody has is skipped for the whole wave): neighbour indices are read coalesced from the plan, the
//      3-float input rows come from the (cache-resident, 1 MB) feature table, the 27 x 3 x 32 weights are wave-uniform
//      scalar loads (SGPR operands of the fma), and the 32 accumulators per voxel are plain VALU fmaf chains in the same
//      (offset ascending, channel ascending) order as the matrix path.
template <int CIN, int COUT, int SPLIT>
__global__ __launch_bounds__(256) void conv_first_layer_kernel(ConvParams p) {
  constexpr int CT = COUT / SPLIT;  // output channels per thread; blockIdx.y selects the slice
  constexpr int KMAX = 27;
  const int tid = threadIdx.x;
  const int K = p.K;
  const int j0 = (int)blockIdx.y * CT;
  const float* __restrict__ w_s = p.W + j0;  // wave-uniform indices below -> scalar loads, weights as SGPR operands
  const int64_t r = (int64_t)blockIdx.x * 256 + tid;  // plan position
  const bool in_range = r < p.Vpad;
  const int64_t o = in_range ? (int64_t)p.perm[r] : -1;
  float acc[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) acc[j] = 0.0f;
  int n[KMAX];  // all neighbour indices of the voxel are requested up front (coalesced across the wavefront)
#pragma unroll
  for (int k = 0; k < KMAX; ++k) n[k] = (o >= 0 && k < K) ? p.nbr_s[(int64_t)k * p.Vpad + r] : -1;
  constexpr int G = 9;  // input rows in flight per thread
#pragma unroll
  for (int k0 = 0; k0 < KMAX; k0 += G) {
    float x[G][CIN];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float* src = p.in + (n[k0 + g] >= 0 ? (int64_t)n[k0 + g] * p.in_ld : 0);
#pragma unroll
      for (int c = 0; c < CIN; ++c) x[g][c] = src[c];
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const bool has = n[k0 + g] >= 0;
      if (__ballot(has) == 0ull) continue;  // no voxel of this wavefront has a neighbour at this offset
      const float* w = w_s + (k0 + g) * CIN * COUT;
      if (has) {
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
#pragma unroll
          for (int j = 0; j < CT; ++j) acc[j] = __builtin_fmaf(x[g][c], w[c * COUT + j], acc[j]);
        }
      }
    }
  }
  if (o < 0) return;
  float* dst = p.out + o * p.out_ld + j0;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    float v = acc[j];
    if (p.scale)
      v = __builtin_fmaf(v, p.scale[j0 + j], p.shift ? p.shift[j0 + j] : 0.0f);
    else if (p.shift)
      v = v + p.shift[j0 + j];
    if (p.residual) v = v + p.residual[o * p.res_ld + j0 + j];
    if (p.act == SV_ACT_RELU)
      v = v < 0.f ? 0.f : v;  // NaN stays NaN, as torch.relu
    else if (p.act == SV_ACT_LEAKY_RELU)
      v = v > 0.f ? v : v * p.slope;
    acc[j] = v;
  }
  if ((p.out_ld & 3) == 0 && (((uintptr_t)p.out) & 15) == 0) {
#pragma unroll
    for (int j = 0; j < CT; j += 4) *(float4*)(dst + j) = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
  } else {
#pragma unroll
    for (int j = 0; j < CT; ++j) dst[j] = acc[j];
  }
}

static int launch_conv_first_mfma(const ConvParams& p, hipStream_t stream) {
  static const int cap = getenv("SV_CONV0_GRID") ? atoi(getenv("SV_CONV0_GRID")) : 768;  // workgroups (see the kernel)
  const unsigned groups = (unsigned)(p.Vpad / 64);
  hipLaunchKernelGGL((conv_first_mfma_kernel<32>), dim3(cap > 0 && groups > (unsigned)cap ? (unsigned)cap : groups), dim3(256), 0,
                     stream, p);
  SV_LAUNCH_CHECK();
  note_instance("conv_first_mfma_kernel<3, 32>|fast=1,ring=0,full=1");
  return SV_OK;
}

static int launch_conv_first_layer(const ConvParams& p, hipStream_t stream) {
  // SPLIT = 1: one thread computes all 32 channels of its voxel (two / four threads per voxel measured 18 / 26 us
  // against 16 us: the gathers are repeated per slice)
  dim3 grid((unsigned)((p.Vpad + 255) / 256), 1);
  hipLaunchKernelGGL((conv_first_layer_kernel<3, 32, 1>), grid, dim3(256), 0, stream, p);
  note_instance("conv_first_layer_kernel<3, 32>|fast=0,ring=0,full=0");
  SV_LAUNCH_CHECK();
  return SV_OK;
}

template <int ROWS>
static int launch_linear_narrow_rows(const ConvParams& p, hipStream_t stream) {
  dim3 grid((unsigned)((p.V_out + ROWS - 1) / ROWS));
  switch (p.Cout) {
    case 1: hipLaunchKernelGGL((linear_narrow_kernel<1, ROWS>), grid, dim3(256), 0, stream, p); break;
    case 2: hipLaunchKernelGGL((linear_narrow_kernel<2, ROWS>), grid, dim3(256), 0, stream, p); break;
    case 3: hipLaunchKernelGGL((linear_narrow_kernel<3, ROWS>), grid, dim3(256), 0, stream, p); break;
    default: hipLaunchKernelGGL((linear_narrow_kernel<4, ROWS>), grid, dim3(256), 0, stream, p); break;
  }
  note_instance("linear_narrow_kernel<%d>|fast=0,ring=0,full=0", p.Cout < 4 ? p.Cout : 4);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
static int launch_linear_narrow(const ConvParams& p, hipStream_t stream) {
  static const int rows_env = getenv("SV_NARROW_ROWS") ? atoi(getenv("SV_NARROW_ROWS")) : 0;  // experiments only
  const int rows = rows_env ? rows_env : 64;  // 256 / 128 / 64 / 32 rows: 3.2 / 3.6 / 3.9 / 2.8 TB/s on 88k x 1024 -> 3
  if (rows == 256) return launch_linear_narrow_rows<256>(p, stream);
  if (rows == 128) return launch_linear_narrow_rows<128>(p, stream);
  if (rows == 32) return launch_linear_narrow_rows<32>(p, stream);
  return launch_linear_narrow_rows<64>(p, stream);
}

template <int TM_, int WAVES_N, int NT, int CPO = 0>
static int launch_conv(const ConvParams& p, hipStream_t stream) {
  using Cfg = ConvCfg<TM_, WAVES_N, NT, CPO>;
  ConvParams q = p;
  q.ntiles = (int)(p.Vpad / TM_);
  q.ny = (p.Cout + Cfg::TN - 1) / Cfg::TN;
  dim3 grid((unsigned)(q.ntiles * q.ny));
  const bool fast = p.vec_a && p.buf_ok && (p.Cout % Cfg::TN == 0);
  // SV_CONV_TRACE=<file> (experiments only): trace the launch per workgroup, synchronise, append to <file>
  // (tools/wg_trace.py reads it: residency over time, per-CU tail, time per step)
  static const char* trace_path = getenv("SV_CONV_TRACE");
  q.trace = nullptr;
  if (trace_path) SV_HIP(hipMalloc((void**)&q.trace, (size_t)grid.x * 4 * sizeof(unsigned long long)));
  // launches of at most ~4 workgroups per CU are bound by the latency of one wave's step chain, not by the matrix
  // pipe: their waves read the A operands several k-steps ahead (register ring); fuller launches hide that latency
  // behind co-resident waves and are a few % faster with the plain one-step-ahead read
  const bool ring = Cfg::PFD > 1 && grid.x <= 1024;
  constexpr bool F = (CPO % 4 == 0);  // odd channel counts (Cin = 3) only exist in the guarded form
  const size_t lds = Cfg::lds_bytes(p.K);
  // FULL: no partial channel chunk anywhere in the layer
  const bool full = Cfg::USE_FULL && (CPO ? (p.K % Cfg::GK == 0) : (p.Cin % Cfg::KC == 0));
  if (F && fast && full) {
    if (ring)
      hipLaunchKernelGGL((conv_fwd_kernel<TM_, WAVES_N, NT, F, CPO, (Cfg::PFD > 1), (F && Cfg::USE_FULL)>), grid, dim3(256), lds, stream, q);
    else
      hipLaunchKernelGGL((conv_fwd_kernel<TM_, WAVES_N, NT, F, CPO, false, (F && Cfg::USE_FULL)>), grid, dim3(256), lds, stream, q);
  } else if (F && fast) {
    if (ring)
      hipLaunchKernelGGL((conv_fwd_kernel<TM_, WAVES_N, NT, F, CPO, (Cfg::PFD > 1)>), grid, dim3(256), lds, stream, q);
    else
      hipLaunchKernelGGL((conv_fwd_kernel<TM_, WAVES_N, NT, F, CPO, false>), grid, dim3(256), lds, stream, q);
  } else {
    if (ring)
      hipLaunchKernelGGL((conv_fwd_kernel<TM_, WAVES_N, NT, false, CPO, (Cfg::PFD > 1)>), grid, dim3(256), lds, stream,
                         q);
    else
      hipLaunchKernelGGL((conv_fwd_kernel<TM_, WAVES_N, NT, false, CPO, false>), grid, dim3(256), lds, stream, q);
  }
  SV_LAUNCH_CHECK();
  {
    const bool f = F && fast;
    if (CPO)
      note_instance("conv_fwd_kernel<%d, %d, %d, fused %d>|fast=%d,ring=%d,full=%d", TM_, WAVES_N, NT, CPO, (int)f, (int)ring,
                    (int)(f && full));
    else
      note_instance("conv_fwd_kernel<%d, %d, %d>|fast=%d,ring=%d,full=%d", TM_, WAVES_N, NT, (int)f, (int)ring, (int)(f && full));
  }
  if (q.trace) {
    std::vector<unsigned long long> host((size_t)grid.x * 4);
    SV_HIP(hipStreamSynchronize(stream));
    SV_HIP(hipMemcpy(host.data(), q.trace, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    SV_HIP(hipFree(q.trace));
    if (FILE* f = fopen(trace_path, "ab")) {
      const long long hdr[8] = {0x5356545243ll, (long long)grid.x, TM_, WAVES_N, NT, q.ny, p.K, p.Cin};
      fwrite(hdr, sizeof(hdr), 1, f);
      fwrite(host.data(), sizeof(unsigned long long), host.size(), f);
      fclose(f);
    }
  }
  return SV_OK;
}

// dual-body launch (conv_fwd_dual_kernel): TM_-row tiles for the expensive plan tiles, TAIL_TM-row tiles for the cheapest
// `tail_fraction` of them.  Returns SV_ERR_INVALID without launching when the layer does not qualify.
template <int TM_, int TAIL_TM, int WAVES_N, int NT>
static int launch_conv_dual(const ConvParams& p, hipStream_t stream, double tail_fraction) {
  using Main = ConvCfg<TM_, WAVES_N, NT, 0>;
  using Tail = ConvCfg<TAIL_TM, WAVES_N, NT, 0>;
  const bool fast = p.vec_a && p.buf_ok && (p.Cout % Main::TN == 0);
  const int n128 = (int)(p.Vpad / PLAN_TILE);
  const int tail128 = (int)(n128 * tail_fraction);
  if (!fast || !p.tile_order || tail128 < 1 || tail128 >= n128) return SV_ERR_INVALID;
  ConvParams q = p;
  q.ny = p.Cout / Main::TN;
  q.main_tiles128 = n128 - tail128;
  q.main_blocks = q.main_tiles128 * (PLAN_TILE / TM_) * q.ny;
  q.ntiles = (int)(p.Vpad / TM_);
  const unsigned grid = (unsigned)(q.main_blocks + tail128 * (PLAN_TILE / TAIL_TM) * q.ny);
  const size_t lds = Main::lds_bytes(p.K) > Tail::lds_bytes(p.K) ? Main::lds_bytes(p.K) : Tail::lds_bytes(p.K);
  static const char* trace_path = getenv("SV_CONV_TRACE");
  q.trace = nullptr;
  if (trace_path) SV_HIP(hipMalloc((void**)&q.trace, (size_t)grid * 4 * sizeof(unsigned long long)));
  hipLaunchKernelGGL((conv_fwd_dual_kernel<TM_, TAIL_TM, WAVES_N, NT, true>), dim3(grid), dim3(256), lds, stream, q);
  SV_LAUNCH_CHECK();
  note_instance("conv_fwd_dual_kernel<%d, %d, %d, %d>|fast=1,ring=0,full=0", TM_, TAIL_TM, WAVES_N, NT);
  if (q.trace) {
    std::vector<unsigned long long> host((size_t)grid * 4);
    SV_HIP(hipStreamSynchronize(stream));
    SV_HIP(hipMemcpy(host.data(), q.trace, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    SV_HIP(hipFree(q.trace));
    if (FILE* f = fopen(trace_path, "ab")) {
      const long long hdr[8] = {0x5356545243ll, (long long)grid, TM_, WAVES_N, NT, q.ny, p.K, p.Cin};
      fwrite(hdr, sizeof(hdr), 1, f);
      fwrite(host.data(), sizeof(unsigned long long), host.size(), f);
      fclose(f);
    }
  }
  return SV_OK;
}

// ---- instance selection ------------------------------------------------------------------------------------------
// A tile is processed sequentially (K * Cin / KC steps) and streams K*Cin*TN weights + its gathered rows through one CU,
// so the choice trades per-CU cache bandwidth (tall tiles, few column slices) against parallelism / tail (many tiles).
// Candidates are listed from least to most traffic; the first one that puts `want` workgroups on the chip wins,
// otherwise the one with the most workgroups.
struct Candidate {
  int tm, wn, nt;
  int64_t want;  // chosen when it yields at least this many workgroups (measured on the Cfg-2 pyramid, profiles/)
  int cpo;       // fused-offset form for this Cin (0 = one offset per step)
};

template <int TM_, int WAVES_N, int NT, int CPO = 0>
static bool try_launch(const Candidate& c, const ConvParams& p, hipStream_t stream, int& rc) {
  if (c.tm == TM_ && c.wn == WAVES_N && c.nt == NT && c.cpo == CPO) {
    rc = launch_conv<TM_, WAVES_N, NT, CPO>(p, stream);
    return true;
  }
  return false;
}

static int launch_candidate(const Candidate& c, const ConvParams& p, hipStream_t stream) {
  int rc = SV_ERR_INVALID;
  if (try_launch<128, 4, 3>(c, p, stream, rc) || try_launch<128, 2, 3>(c, p, stream, rc) ||
      try_launch<128, 1, 3>(c, p, stream, rc) || try_launch<64, 4, 3>(c, p, stream, rc) ||
      try_launch<64, 2, 3>(c, p, stream, rc) || try_launch<64, 1, 3>(c, p, stream, rc) ||
      try_launch<32, 4, 3>(c, p, stream, rc) || try_launch<32, 2, 3>(c, p, stream, rc) ||
      try_launch<16, 4, 3>(c, p, stream, rc) ||
      try_launch<128, 4, 2>(c, p, stream, rc) || try_launch<128, 2, 2>(c, p, stream, rc) ||
      try_launch<128, 1, 2>(c, p, stream, rc) || try_launch<64, 4, 2>(c, p, stream, rc) ||
      try_launch<64, 2, 2>(c, p, stream, rc) || try_launch<64, 1, 2>(c, p, stream, rc) ||
      try_launch<32, 4, 2>(c, p, stream, rc) || try_launch<32, 2, 2>(c, p, stream, rc) ||
      try_launch<16, 4, 2>(c, p, stream, rc) ||
      try_launch<64, 4, 1>(c, p, stream, rc) || try_launch<32, 4, 1>(c, p, stream, rc) ||
      try_launch<16, 4, 1>(c, p, stream, rc) || try_launch<128, 2, 1>(c, p, stream, rc) ||
      try_launch<64, 2, 1>(c, p, stream, rc) || try_launch<32, 2, 1>(c, p, stream, rc) ||
      try_launch<128, 1, 1>(c, p, stream, rc) || try_launch<64, 1, 1>(c, p, stream, rc) ||
      // fused-offset forms of the thin layers
      try_launch<64, 2, 1, 3>(c, p, stream, rc) || try_launch<32, 2, 1, 3>(c, p, stream, rc) ||
      try_launch<64, 2, 1, 32>(c, p, stream, rc) || try_launch<32, 2, 1, 32>(c, p, stream, rc) ||
      try_launch<32, 4, 1, 32>(c, p, stream, rc) || try_launch<16, 4, 1, 32>(c, p, stream, rc) ||
      try_launch<32, 4, 1, 64>(c, p, stream, rc) || try_launch<16, 4, 1, 64>(c, p, stream, rc))
    return rc;
  set_error("sv_conv_fwd: no kernel instance <%d,%d,%d> cpo %d", c.tm, c.wn, c.nt, c.cpo);
  return SV_ERR_INVALID;
}

static int64_t candidate_wgs(const Candidate& c, const ConvParams& p) {
  const int tn = c.wn * c.nt * 16;
  return (p.Vpad / c.tm) * ((p.Cout + tn - 1) / tn);
}

static int select_and_launch(const ConvParams& p, hipStream_t stream) {
  // The last entry of a list serves the small pyramid levels, where a layer is a few hundred short workgroups and its
  // duration is the serial (offset, chunk) chain of one of them: 16-row tiles with 128-channel chunks are fastest
  // there (tools/sweep_small.sh, profiles/r01_conv_small_level_sweep.txt).
  static const Candidate wide3[] = {{64, 4, 3, 2500}, {32, 4, 3, 1500}, {32, 2, 3, 0}};
  // Cout % 128 == 0 (384): settled-clock sweeps of every level, profiles/r02_conv_instance_sweep.txt.  Below the
  // chip-filling size the 16-row tile in its FULL form (Cin a multiple of its 128-channel step: 384, 512) wins at every
  // level (level 1: 96.4 TFLOP/s against 90.1 for 64x128, level 3: 41 against 34 for 16x128); with a partial last chunk
  // (Cin 416 / 448, the first conv after a concatenation) it only wins once 64x128 tiles no longer fill the chip
  // ({32,4,3} between them: inside the frame pipeline level 2 - 432 workgroups of 32 x 192 - is 0.8 % of a frame faster on
  // 32-row tiles although they are 25 % slower than the 16-row tiles alone: half the weight traffic beside the chip-filling
  // launches of the other frames; 64.3 against 63.8 frames/s, three alternating runs, tools/ab_env.sh)
  static const Candidate wide3_128[] = {{64, 4, 3, 2500}, {64, 4, 2, 1200}, {32, 4, 3, 1200}, {16, 4, 3, 0}};
  static const Candidate wide3_128_full[] = {{64, 4, 3, 2500}, {32, 4, 3, 1200}, {16, 4, 3, 0}};
  static const Candidate wide2[] = {{128, 4, 2, 1300}, {64, 4, 2, 1500}, {32, 4, 2, 1500}, {32, 2, 2, 0}};
  static const Candidate wide2_64[] = {{128, 4, 2, 1300}, {64, 4, 2, 1500}, {32, 4, 2, 1500}, {16, 4, 1, 0}};  // % 64
  static const Candidate c64[] = {{64, 4, 1, 1500}, {32, 4, 1, 1500}, {16, 4, 1, 0}};
  static const Candidate c32[] = {{128, 2, 1, 1500}, {64, 2, 1, 1500}, {32, 2, 1, 0}};
  static const Candidate c16[] = {{128, 1, 1, 600}, {64, 1, 1, 0}};
  const Candidate* list;
  int n;
  const int Cout = p.Cout;
  if (Cout > 128 && (Cout % 192 == 0 || Cout % 96 == 0 || Cout > 2048)) {
    if (Cout % 128 != 0) { list = wide3; n = 3; }
    else if (p.Cin % 128 == 0 && p.vec_a) { list = wide3_128_full; n = 3; }
    else { list = wide3_128; n = 4; }
  }
  else if (Cout > 64) { list = Cout % 64 == 0 ? wide2_64 : wide2; n = 4; }
  else if (Cout > 32) { list = c64; n = 3; }
  else if (Cout > 16) { list = c32; n = 3; }
  else { list = c16; n = 2; }
  // thin layers: fused-offset forms (the input row is at most 256 B, so one offset per step would be all overhead)
  static const Candidate f3[] = {{64, 2, 1, 3000, 3}, {32, 2, 1, 0, 3}};
  static const Candidate f32_32[] = {{64, 2, 1, 3000, 32}, {32, 2, 1, 0, 32}};
  static const Candidate f32_64[] = {{32, 4, 1, 1500, 32}, {16, 4, 1, 0, 32}};
  static const Candidate f64_64[] = {{32, 4, 1, 1500, 64}, {16, 4, 1, 0, 64}};
  static const bool no_fused = getenv("SV_CONV_NO_FUSED") != nullptr;  // experiments only
  if (p.K > 1 && !no_fused) {
    if (p.Cin == 3 && Cout > 16 && Cout <= 32) { list = f3; n = 2; }
    else if (p.Cin == 32 && p.vec_a && Cout == 32) { list = f32_32; n = 2; }
    else if (p.Cin == 32 && p.vec_a && Cout == 64) { list = f32_64; n = 2; }
    else if (p.Cin == 64 && p.vec_a && (Cout == 64 || Cout == 128)) { list = f64_64; n = 2; }
  }
  if (const char* f = getenv("SV_CONV_FORCE_RANGE")) {  // "vmin:vmax:cout:tm,wn,nt": one level's layers only (experiments)
    long long vmin = 0, vmax = 0;
    int cout = 0;
    Candidate c = {0, 0, 0, 0, 0};
    if (sscanf(f, "%lld:%lld:%d:%d,%d,%d", &vmin, &vmax, &cout, &c.tm, &c.wn, &c.nt) == 6 && p.Vpad >= vmin && p.Vpad <= vmax &&
        p.Cout == cout && p.K > 1)
      return launch_candidate(c, p, stream);
  }
  if (const char* f = getenv("SV_CONV_FORCE_DENSE")) {  // "vmin:cout:tm,wn,nt": dense (K = 1) layers of one width (experiments)
    long long vmin = 0;
    int cout = 0;
    Candidate c = {0, 0, 0, 0, 0};
    if (sscanf(f, "%lld:%d:%d,%d,%d", &vmin, &cout, &c.tm, &c.wn, &c.nt) == 5 && p.Vpad >= vmin && p.Cout == cout && p.K == 1)
      return launch_candidate(c, p, stream);
  }
  if (const char* f = getenv("SV_CONV_FORCE")) {  // "tm,wn,nt[,cpo]": experiments only
    Candidate c = {0, 0, 0, 0, 0};
    if (sscanf(f, "%d,%d,%d,%d", &c.tm, &c.wn, &c.nt, &c.cpo) >= 3) return launch_candidate(c, p, stream);
  }
  // The `want` figures of the lists were measured one launch at a time (profiles/*_conv_instance_sweep.txt).  Inside the
  // two-stream frame pipeline the taller tile wins earlier: its weight traffic per flop is lower (level 1, 384 -> 384:
  // 64-row tiles read W once per 64 rows, 16-row tiles once per 16 - alone on the GPU both run at ~107 TFLOP/s, with
  // perfect weight locality the 16-row tile would gain 7 %), and the launch tail that costs the short grid alone is
  // filled by the neighbour frame's kernels.  Scaling every threshold by 0.3 moves level 1 (830 workgroups of 64 x 192)
  // onto the dual-body launch and leaves levels 2-4 where they were: frames/s 60.9 -> 62.0 (0.15: 62.1, 0.08: 61.1,
  // 0.03: 60.7; three alternating runs each).
  static const double want_scale =
      getenv("SV_CONV_WANT_SCALE") ? atof(getenv("SV_CONV_WANT_SCALE")) : SV_CONV_WANT_SCALE_DEFAULT;
  // chip-filling 384-wide layers: 64-row tiles, and 32-row tiles for the cheapest plan tiles at the end of the grid
  static const double tail_fraction = getenv("SV_CONV_TAIL") ? atof(getenv("SV_CONV_TAIL")) : SV_CONV_TAIL_DEFAULT;
  const double ws = g_want_scale_override >= 0.0 ? g_want_scale_override : want_scale;
  const double tf = g_tail_override >= 0.0 ? g_tail_override : tail_fraction;
  for (int i = 0; i < n; ++i)
    if ((double)candidate_wgs(list[i], p) >= ws * (double)list[i].want) {
      const Candidate& c = list[i];
      if (c.tm == 64 && c.wn == 4 && c.nt == 3 && c.cpo == 0 && tf > 0.0 &&
          launch_conv_dual<64, 32, 4, 3>(p, stream, tf) == SV_OK)
        return SV_OK;
      return launch_candidate(c, p, stream);
    }
  return launch_candidate(list[n - 1], p, stream);
}

}  // namespace sv

using namespace sv;

extern "C" const char* sv_conv_last_instance(void) { return sv::g_last_instance; }

extern "C" int sv_conv_set_dispatch(double want_scale, double tail_fraction) {
  SV_CHECK_ARG(tail_fraction < 1.0, "tail_fraction must be below 1");
  sv::g_want_scale_override = want_scale;
  sv::g_tail_override = tail_fraction;
  return SV_OK;
}

extern "C" int sv_conv_fwd(const float* in, int64_t V_in, int64_t in_ld, int Cin, const float* W, int K, int Cout,
                           const int32_t* perm, const int32_t* nbr_s, const uint32_t* submask,
                           const int32_t* tile_order, int64_t V_out, int64_t Vpad, const float* scale, const float* shift, const float* residual, int64_t res_ld,
                           int act, float slope, float* out, int64_t out_ld, sv_stream_t stream_) {
  return sv_conv_fwd_acc(in, V_in, in_ld, Cin, W, K, Cout, perm, nbr_s, submask, tile_order, V_out, Vpad, nullptr, 0, scale, shift,
                         residual, res_ld, act, slope, out, out_ld, stream_);
}

extern "C" int sv_conv_fwd_acc(const float* in, int64_t V_in, int64_t in_ld, int Cin, const float* W, int K, int Cout,
                               const int32_t* perm, const int32_t* nbr_s, const uint32_t* submask, const int32_t* tile_order,
                               int64_t V_out, int64_t Vpad, const float* acc_init, int64_t acc_ld, const float* scale,
                               const float* shift, const float* residual, int64_t res_ld, int act, float slope, float* out,
                               int64_t out_ld, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(!acc_init || acc_ld >= Cout, "acc_init stride too small");
  SV_CHECK_ARG(Cin > 0 && Cout > 0 && K >= 1 && K <= 32, "bad channel / kernel volume");
  SV_CHECK_ARG(V_out >= 0 && Vpad >= V_out && Vpad % PLAN_TILE == 0, "Vpad must be a multiple of 128 >= V_out");
  SV_CHECK_ARG(in_ld >= Cin && out_ld >= Cout, "row strides too small");
  SV_CHECK_ARG(act >= SV_ACT_NONE && act <= SV_ACT_LEAKY_RELU, "bad activation");
  if (V_out == 0) return SV_OK;
  SV_CHECK_ARG(in && W && out, "null pointer");
  SV_CHECK_ARG(V_in >= 1, "V_in = rows of `in` (every index of the plan is below it)");
  const bool has_plan = perm || nbr_s || submask;
  SV_CHECK_ARG(!has_plan || (perm && nbr_s && submask), "perm, nbr_s and submask must be given together");
  SV_CHECK_ARG(has_plan || K == 1, "K > 1 needs a plan");
  SV_CHECK_ARG(!residual || res_ld >= Cout, "residual stride too small");
  ConvParams p;
  p.in = in; p.in_ld = in_ld; p.Cin = Cin; p.W = W; p.K = K; p.Cout = Cout;
  p.perm = perm; p.nbr_s = nbr_s; p.submask = submask; p.tile_order = tile_order; p.V_out = V_out; p.Vpad = Vpad;
  p.scale = scale; p.shift = shift; p.residual = residual; p.res_ld = res_ld;
  p.acc_init = acc_init; p.acc_ld = acc_ld;
  p.act = act; p.slope = slope; p.out = out; p.out_ld = out_ld;
  p.vec_a = (in_ld % 4 == 0) && (Cin % 4 == 0) && (((uintptr_t)in & 15) == 0);
  // extents for the buffer descriptors of the FAST instances: `in` through the last channel of its last row
  const uint64_t in_bytes = ((uint64_t)(V_in - 1) * (uint64_t)in_ld + (uint64_t)Cin) * 4u;
  const uint64_t w_bytes = (uint64_t)K * (uint64_t)Cin * (uint64_t)Cout * 4u;
  const uint64_t out_bytes = ((uint64_t)(V_out - 1) * (uint64_t)out_ld + (uint64_t)Cout) * 4u;
  const uint64_t res_bytes = residual ? ((uint64_t)(V_out - 1) * (uint64_t)res_ld + (uint64_t)Cout) * 4u : 0u;
  const uint64_t acc_bytes = acc_init ? ((uint64_t)(V_out - 1) * (uint64_t)acc_ld + (uint64_t)Cout) * 4u : 0u;
  // (the buffer-addressed epilogue reads `perm` four entries at a time: a plan carved from a workspace at an odd offset
  //  takes the guarded form)
  p.buf_ok = in_bytes < BUF_LIMIT && w_bytes < BUF_LIMIT && out_bytes < BUF_LIMIT && res_bytes < BUF_LIMIT &&
             acc_bytes < BUF_LIMIT && (((uintptr_t)perm & 15) == 0);
  p.acc_bytes = p.buf_ok ? (uint32_t)acc_bytes : 0u;
  if (!p.buf_ok && !has_plan && K == 1 && w_bytes < BUF_LIMIT) {
    // dense rows (Linear / 1x1 conv) of a tensor beyond the 2 GB extent: every row range is a layer of its own, so the
    // launch is split into ranges that fit the buffer-addressed instances (64 Cfg-2 frames x 1024 channels = 23 GB)
    int64_t ld_max = in_ld > out_ld ? in_ld : out_ld;
    if (residual && res_ld > ld_max) ld_max = res_ld;
    const int64_t rows = (int64_t)((BUF_LIMIT - 4096u) / ((uint64_t)ld_max * 4u)) / PLAN_TILE * PLAN_TILE;
    if (rows >= PLAN_TILE && rows < V_out) {
      for (int64_t r0 = 0; r0 < V_out; r0 += rows) {
        const int64_t n = V_out - r0 < rows ? V_out - r0 : rows;
        const int rc = sv_conv_fwd_acc(in + r0 * in_ld, n, in_ld, Cin, W, K, Cout, nullptr, nullptr, nullptr, nullptr, n,
                                       (n + PLAN_TILE - 1) / PLAN_TILE * PLAN_TILE, acc_init ? acc_init + r0 * acc_ld : nullptr,
                                       acc_ld, scale, shift, residual ? residual + r0 * res_ld : nullptr, res_ld, act, slope,
                                       out + r0 * out_ld, out_ld, stream_);
        if (rc != SV_OK) return rc;
      }
      return SV_OK;
    }
  }
  p.in_bytes = p.buf_ok ? (uint32_t)in_bytes : 0u;
  p.w_bytes = p.buf_ok ? (uint32_t)w_bytes : 0u;
  p.out_bytes = p.buf_ok ? (uint32_t)out_bytes : 0u;
  p.res_bytes = p.buf_ok ? (uint32_t)res_bytes : 0u;
  p.ntiles = 0;
  p.ny = 0;
  p.trace = nullptr;
  p.main_blocks = 0;
  p.main_tiles128 = 0;
  static const bool no_first = getenv("SV_CONV_NO_FIRST") != nullptr;  // experiments only
  static const bool first_valu = getenv("SV_CONV_FIRST_VALU") != nullptr;  // experiments only: the thread-per-voxel kernel
  if (has_plan && K > 1 && K <= 27 && Cin == 3 && Cout == 32 && !no_first && !acc_init) {
    if (p.buf_ok && !first_valu) return launch_conv_first_mfma(p, stream);
    return launch_conv_first_layer(p, stream);
  }
  static const bool no_thin = getenv("SV_CONV_NO_THIN") != nullptr;  // experiments only
  if (has_plan && K > 1 && K <= 32 && Cin == 32 && Cout == 32 && p.vec_a && p.buf_ok && !no_thin && !acc_init) return launch_conv_thin(p, stream);
  static const bool no_narrow = getenv("SV_CONV_NO_NARROW") != nullptr;  // experiments only
  if (!has_plan && K == 1 && Cout <= 4 && p.vec_a && Cin >= 64 && !no_narrow && !acc_init) return launch_linear_narrow(p, stream);
  return select_and_launch(p, stream);
}
