// K6/K7: output-stationary sparse convolution on fp32 MFMA (v_mfma_f32_16x16x4_f32) with fused epilogue.
//
// Replaces ME.MinkowskiConvolution / MinkowskiConvolutionTranspose / MinkowskiLinear followed by
// MinkowskiBatchNorm(eval) (+ residual) (+ ReLU / LeakyReLU): model/backbone/minkunet.py:125-187,
// model/backbone/resnet.py:95-127 (BasicBlock via ME), model/robotnet_segmentation.py:55-64.
//
// Work decomposition
//   * a workgroup (4 waves) owns one tile of 128 output rows (in the plan's mask-sorted order) x TN output channels;
//   * it walks the kernel offsets k in ASCENDING order and, per offset, the input channels in ascending chunks of 32;
//   * per chunk the gathered input rows (A, 128 x 32) and the weight slab (B, 32 x TN) are staged in LDS;
//   * each wave multiplies with v_mfma_f32_16x16x4_f32; a 16-row sub-tile is skipped for an offset when none of its rows
//     has a neighbour there (plan submask), which is what makes the mask-sorted row order pay.
// Numerics: every output element is ONE f32 fma chain over (k ascending, c ascending) — the MFMA is a k-ordered
// fmaf chain (MI355X guide §3) — so results are bitwise reproducible and match oracle/sv_oracle.c exactly.
// A missing neighbour contributes fma(0, w, acc) = acc.
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
  int64_t V_out;
  int64_t Vpad;
  const float* scale;
  const float* shift;
  const float* residual;
  int64_t res_ld;
  int act;
  float slope;
  float* out;
  int64_t out_ld;
  int vec_a;  // in_ld % 4 == 0 && Cin % 4 == 0 && base aligned -> float4 gathers
  int vec_b;  // Cout % 4 == 0 && W aligned -> float4 weight loads
  int ntiles;
  int ny;
};

constexpr int TM = SV_TILE_ROWS;  // 128 rows per tile
constexpr int KC = 32;            // input channels per LDS chunk
constexpr int SA = KC + 2;        // A row stride in floats: conflict-free 16x16x4 operand reads

template <int WAVES_N, int NT>
struct ConvCfg {
  static constexpr int WAVES_M = 4 / WAVES_N;
  static constexpr int MR = (TM / WAVES_M) / 16;  // 16-row sub-tiles per wave
  static constexpr int TN = WAVES_N * NT * 16;    // output channels per workgroup
  static constexpr int SB = TN + 16;              // B row stride in floats (== 16 mod 32)
  static constexpr int B_F4 = (KC * TN / 4 + 255) / 256;  // float4 weight loads per thread and step
  static constexpr size_t lds_bytes(int K) { return (size_t)(TM * SA + KC * SB + K * TM) * sizeof(float); }
};

// One pipeline step = (kernel offset k, input-channel chunk c0).  A step's global loads are issued into registers
// BEFORE the previous step's MFMAs and written to LDS after them, so HBM/L2 latency hides under matrix work.
template <int WAVES_N, int NT>
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvParams p) {
  using Cfg = ConvCfg<WAVES_N, NT>;
  constexpr int MR = Cfg::MR, TN = Cfg::TN, SB = Cfg::SB, B_F4 = Cfg::B_F4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                              // [TM][SA]
  float* Bs = lds + TM * SA;                    // [KC][SB]
  int* idx_s = (int*)(lds + TM * SA + KC * SB);  // [K][TM] gathered input row of every (offset, tile row)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  // 1-D grid, heavy tiles first: rows are mask-sorted ascending, so the tiles with the most neighbour offsets sit at the
  // end of the plan; dispatching them first keeps the tail of the launch short (list scheduling, longest first).
  const int ny = p.ny;
  const int tile = p.ntiles - 1 - (int)(blockIdx.x / ny);
  const int n0 = (int)(blockIdx.x % ny) * TN;
  const int64_t row0 = (int64_t)tile * TM;
  const int li = lane & 15, lq = lane >> 4;
  const int K = p.K, Cin = p.Cin, Cout = p.Cout;

  f32x4 acc[MR][NT];
#pragma unroll
  for (int s = 0; s < MR; ++s)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[s][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- stage the tile's neighbour table (or the identity for dense rows) in LDS
  uint32_t dense_mask = 0xffu;
  if (p.submask == nullptr) {
    const int64_t rem = p.V_out - row0;
    const int nsub = rem >= TM ? 8 : (int)((rem + 15) / 16);
    dense_mask = (nsub >= 8) ? 0xffu : ((1u << nsub) - 1u);
  }
  for (int e = tid; e < K * TM; e += 256) {
    const int k = e / TM, r = e - k * TM;
    int n;
    if (p.nbr_s)
      n = p.nbr_s[(int64_t)k * p.Vpad + row0 + r];
    else
      n = (row0 + r < p.V_out) ? (int)(row0 + r) : -1;
    idx_s[e] = n;
  }

  // ---- step iterator over (active offset, chunk)
  auto submask_of = [&](int k) -> uint32_t { return p.submask ? p.submask[(int64_t)tile * K + k] : dense_mask; };
  int k_nxt = 0;
  uint32_t sm_nxt = 0;
  while (k_nxt < K && (sm_nxt = submask_of(k_nxt)) == 0) ++k_nxt;
  int c_nxt = 0;
  bool have_nxt = k_nxt < K;

  float4 ra[4];
  float4 rb[B_F4];
  const int a_cc = (tid & 7) * 4;
  const int a_r = tid >> 3;

  auto issue_loads = [&](int k, int c0, uint32_t sm) {
    // A: 128 gathered rows x 32 channels; thread -> rows a_r + 32j, channels c0 + a_cc .. +3
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = a_r + 32 * j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((sm >> (r >> 4)) & 1u) {
        const int n = idx_s[k * TM + r];
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
    // B: 32 channels x TN output channels of W[k]
    const float* Wk = p.W + (int64_t)k * Cin * Cout;
    constexpr int F4_PER_ROW = TN / 4;
#pragma unroll
    for (int j = 0; j < B_F4; ++j) {
      const int e = tid + 256 * j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < KC * F4_PER_ROW) {
        const int kk = e / F4_PER_ROW;
        const int c4 = (e - kk * F4_PER_ROW) * 4;
        const int c = c0 + kk;
        const int col = n0 + c4;
        if (c < Cin) {
          const float* src = Wk + (int64_t)c * Cout + col;
          if (p.vec_b) {
            if (col < Cout) v = *(const float4*)src;
          } else {
            if (col + 0 < Cout) v.x = src[0];
            if (col + 1 < Cout) v.y = src[1];
            if (col + 2 < Cout) v.z = src[2];
            if (col + 3 < Cout) v.w = src[3];
          }
        }
      }
      rb[j] = v;
    }
  };

  auto store_lds = [&](uint32_t sm) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = a_r + 32 * j;
      if ((sm >> (r >> 4)) & 1u) {
        float2* dst = (float2*)(As + r * SA + a_cc);
        dst[0] = make_float2(ra[j].x, ra[j].y);
        dst[1] = make_float2(ra[j].z, ra[j].w);
      }
    }
    constexpr int F4_PER_ROW = TN / 4;
#pragma unroll
    for (int j = 0; j < B_F4; ++j) {
      const int e = tid + 256 * j;
      if (e < KC * F4_PER_ROW) {
        const int kk = e / F4_PER_ROW;
        const int c4 = (e - kk * F4_PER_ROW) * 4;
        *(float4*)(Bs + kk * SB + c4) = rb[j];
      }
    }
  };

  __syncthreads();  // idx_s visible
  if (have_nxt) issue_loads(k_nxt, c_nxt, sm_nxt);

  const float* a_base = As + (wm * MR * 16 + li) * SA + lq;
  const float* b_base = Bs + lq * SB + wn * NT * 16 + li;

  while (have_nxt) {
    const int c_cur = c_nxt;
    const uint32_t sm_cur = sm_nxt;
    __syncthreads();  // every wave finished the previous step's LDS reads
    store_lds(sm_cur);
    __syncthreads();  // tiles visible
    // advance the iterator and put the next step's loads in flight
    c_nxt += KC;
    if (c_nxt >= Cin) {
      c_nxt = 0;
      ++k_nxt;
      while (k_nxt < K && (sm_nxt = submask_of(k_nxt)) == 0) ++k_nxt;
      have_nxt = k_nxt < K;
    }
    if (have_nxt) issue_loads(k_nxt, c_nxt, sm_nxt);

    // ---- MFMA over the current step; operand reads run one k-step ahead of the matrix ops
    const uint32_t smw = (sm_cur >> (wm * MR)) & ((1u << MR) - 1u);
    const int kmax = min(KC, Cin - c_cur);
    const int ksteps = (kmax + 3) >> 2;
    float a_cur[MR], b_cur[NT];
#pragma unroll
    for (int s = 0; s < MR; ++s) a_cur[s] = a_base[s * 16 * SA];
#pragma unroll
    for (int n = 0; n < NT; ++n) b_cur[n] = b_base[n * 16];
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
      if (ks < ksteps) {
        float a_nx[MR], b_nx[NT];
        if (ks + 1 < KC / 4) {
#pragma unroll
          for (int s = 0; s < MR; ++s) a_nx[s] = a_base[s * 16 * SA + (ks + 1) * 4];
#pragma unroll
          for (int n = 0; n < NT; ++n) b_nx[n] = b_base[(ks + 1) * 4 * SB + n * 16];
        }
#pragma unroll
        for (int s = 0; s < MR; ++s) {
          if ((smw >> s) & 1u) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[s][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s], b_cur[n], acc[s][n], 0, 0, 0);
          }
        }
        if (ks + 1 < KC / 4) {
#pragma unroll
          for (int s = 0; s < MR; ++s) a_cur[s] = a_nx[s];
#pragma unroll
          for (int n = 0; n < NT; ++n) b_cur[n] = b_nx[n];
        }
      }
    }
  }

  // ---- epilogue: BN(eval)/bias -> residual -> activation -> store (C/D map: col = lane&15, row = (lane>>4)*4 + reg)
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = n0 + wn * NT * 16 + n * 16 + li;
    const bool col_ok = col < Cout;
    const float sc = (p.scale && col_ok) ? p.scale[col] : 1.0f;
    const float sh = (p.shift && col_ok) ? p.shift[col] : 0.0f;
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
        if (o < 0 || !col_ok) continue;
        float y = acc[s][n][reg];
        if (p.scale)
          y = __builtin_fmaf(y, sc, sh);
        else if (p.shift)
          y = y + sh;
        if (p.residual) y = y + p.residual[o * p.res_ld + col];
        if (p.act == SV_ACT_RELU)
          y = y > 0.f ? y : 0.f;
        else if (p.act == SV_ACT_LEAKY_RELU)
          y = y > 0.f ? y : y * p.slope;
        p.out[o * p.out_ld + col] = y;
      }
    }
  }
}

template <int WAVES_N, int NT>
static int launch_conv(const ConvParams& p, hipStream_t stream) {
  using Cfg = ConvCfg<WAVES_N, NT>;
  ConvParams q = p;
  q.ntiles = (int)(p.Vpad / TM);
  q.ny = (p.Cout + Cfg::TN - 1) / Cfg::TN;
  dim3 grid((unsigned)(q.ntiles * q.ny));
  hipLaunchKernelGGL((conv_fwd_kernel<WAVES_N, NT>), grid, dim3(256), Cfg::lds_bytes(p.K), stream, q);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

}  // namespace sv

using namespace sv;

extern "C" int sv_conv_fwd(const float* in, int64_t in_ld, int Cin, const float* W, int K, int Cout,
                           const int32_t* perm, const int32_t* nbr_s, const uint32_t* submask, int64_t V_out,
                           int64_t Vpad, const float* scale, const float* shift, const float* residual, int64_t res_ld,
                           int act, float slope, float* out, int64_t out_ld, sv_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SV_CHECK_ARG(Cin > 0 && Cout > 0 && K >= 1 && K <= 32, "bad channel / kernel volume");
  SV_CHECK_ARG(V_out >= 0 && Vpad >= V_out && Vpad % TM == 0, "Vpad must be a multiple of 128 >= V_out");
  SV_CHECK_ARG(in_ld >= Cin && out_ld >= Cout, "row strides too small");
  SV_CHECK_ARG(act >= SV_ACT_NONE && act <= SV_ACT_LEAKY_RELU, "bad activation");
  if (V_out == 0) return SV_OK;
  SV_CHECK_ARG(in && W && out, "null pointer");
  const bool has_plan = perm || nbr_s || submask;
  SV_CHECK_ARG(!has_plan || (perm && nbr_s && submask), "perm, nbr_s and submask must be given together");
  SV_CHECK_ARG(has_plan || K == 1, "K > 1 needs a plan");
  SV_CHECK_ARG(!residual || res_ld >= Cout, "residual stride too small");
  ConvParams p;
  p.in = in; p.in_ld = in_ld; p.Cin = Cin; p.W = W; p.K = K; p.Cout = Cout;
  p.perm = perm; p.nbr_s = nbr_s; p.submask = submask; p.V_out = V_out; p.Vpad = Vpad;
  p.scale = scale; p.shift = shift; p.residual = residual; p.res_ld = res_ld;
  p.act = act; p.slope = slope; p.out = out; p.out_ld = out_ld;
  p.vec_a = (in_ld % 4 == 0) && (Cin % 4 == 0) && (((uintptr_t)in & 15) == 0);
  p.vec_b = (Cout % 4 == 0) && (((uintptr_t)W & 15) == 0);
  if (Cout > 128) {
    // 192-wide tiles waste least for 384; 128-wide for 256 / 1024 / 2048
    if (Cout % 192 == 0 || Cout > 2048) return launch_conv<4, 3>(p, stream);
    return launch_conv<4, 2>(p, stream);
  }
  if (Cout > 64) return launch_conv<4, 2>(p, stream);
  if (Cout > 32) return launch_conv<4, 1>(p, stream);
  if (Cout > 16) return launch_conv<2, 1>(p, stream);
  return launch_conv<1, 1>(p, stream);
}
