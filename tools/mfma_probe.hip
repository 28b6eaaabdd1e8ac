// What can the chip SUSTAIN on v_mfma_f32_16x16x4_f32?  A register-only loop (and one that feeds A from LDS the way
// conv_tile_body does: one ds_read_b32 per NT matrix ops), at 1 / 2 / 4 waves per SIMD, long enough for clocks and the
// power limit to settle.  Measurement tool only (tools/mfma_probe.py drives it); not part of libsvhip.so.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_pure_kernel(float* out, int iters) {
    f32x4 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1e-9f * threadIdx.x, b = 1.0f + 1e-9f * blockIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 123.456f) out[0] = s;
}

// A from LDS: per k-step one ds_read_b32 per 16-row sub-tile (MT of them), reused by NT column tiles.
template <int MT, int NT>
__global__ __launch_bounds__(256) void mfma_lds_kernel(float* out, int iters) {
    __shared__ float tile[64 * 33];
    for (int i = threadIdx.x; i < 64 * 33; i += 256) tile[i] = 1e-9f * i;
    __syncthreads();
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    float b[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) b[n] = 1.0f + 1e-9f * (blockIdx.x + n);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float a = tile[(m * 16 + (lane & 15)) * 33 + k * 4 + (lane >> 4)];
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc[m][n], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (s == 123.456f) out[0] = s;
}

extern "C" {
// variant 0: register-only, 12 accumulators; 1: A through LDS, 4x3 accumulators.  Returns matrix ops per wave per launch.
long long mfma_probe_launch(int variant, int blocks, int iters, float* out, hipStream_t stream) {
    if (variant == 0) {
        hipLaunchKernelGGL(mfma_pure_kernel<12>, dim3(blocks), dim3(256), 0, stream, out, iters);
        return 8LL * 12 * iters;
    }
    hipLaunchKernelGGL((mfma_lds_kernel<4, 3>), dim3(blocks), dim3(256), 0, stream, out, iters);
    return 8LL * 12 * iters;
}
}
