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

// What does an instruction of another class cost the matrix pipe?  The LDS-fed loop above plus, per k-step of 12 matrix
// ops (MODE bits): 1 = one global_load_dwordx3 from a cache-resident 48 KB block (the weight reload), 2 = two
// global_load_dwordx4 per 8 k-steps from random rows of a big buffer (the gathers), 4 = two 64-bit VALU adds per
// k-step, 16 = four ds_write_b64 + barrier per 8 k-steps.  Waits as in conv_tile_body: a k-step waits for the weight row
// requested 8 k-steps earlier (s_waitcnt vmcnt(7)), the end of a step for its two gathers.  Loaded values are never read.
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void mfma_mix_kernel(float* out, const float* wbuf, const float* big, long long big_rows,
                                                       int iters) {
    __shared__ float tile[2 * 64 * 34];
    for (int i = threadIdx.x; i < 2 * 64 * 34; i += 256) tile[i] = 1e-9f * i;
    __syncthreads();
    f32x4 acc[4][3];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    float b[3] = {1.0f, 1.0f + 1e-9f * blockIdx.x, 0.5f};
    const float* pb = wbuf + (lane >> 4) * 384 + (lane & 15) * 3 + (threadIdx.x >> 6) * 48;
    unsigned long long rnd = 0x9E3779B97F4A7C15ull * (blockIdx.x * 32 + (threadIdx.x >> 3) + 1);  // 8 lanes share a row
    unsigned long long vaddr = (unsigned long long)pb;
    int sacc = blockIdx.x;
    // destinations stay allocated until the s_waitcnt at the end of the step (a load must never land in a register the
    // compiler has meanwhile given to an address)
    f32x4 gv[2];
    f32x3 wv[8];
    // MODE & 32: the same loads in buffer form (descriptor + 32-bit offsets; gather offsets from an LDS table)
    __shared__ unsigned row_off[64 * 32];
    i32x4 srd_w, srd_big;
    srd_w[0] = (int)(unsigned)(unsigned long long)wbuf; srd_w[1] = (int)(unsigned)((unsigned long long)wbuf >> 32);
    srd_w[2] = 1 << 18; srd_w[3] = 0x00020000;
    srd_big[0] = (int)(unsigned)(unsigned long long)big; srd_big[1] = (int)(unsigned)((unsigned long long)big >> 32);
    srd_big[2] = (int)(big_rows * 1536); srd_big[3] = 0x00020000;
    if (MODE & 32) {
        for (int e = threadIdx.x; e < 64 * 32; e += 256) {
            rnd = rnd * 6364136223846793005ull + 1442695040888963407ull;
            row_off[e] = (unsigned)(((rnd >> 20) % (unsigned long long)big_rows) * 1536ull);
        }
        __syncthreads();
    }
    const unsigned wv_off = (unsigned)((lane >> 4) * 384 + (lane & 15) * 3 + (threadIdx.x >> 6) * 48) * 4u;
    for (int i = 0; i < iters; ++i) {
        if ((MODE & 34) == 34) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const unsigned vo = row_off[((i * 2 + g) & 63) * 32 + (threadIdx.x >> 3)] + (threadIdx.x & 7) * 16u;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(gv[g]) : "v"(vo), "s"(srd_big) : "memory");
            }
        } else if (MODE & 2) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                rnd = rnd * 6364136223846793005ull + 1442695040888963407ull;
                const float* src = big + ((rnd >> 20) % (unsigned long long)big_rows) * 384 + (threadIdx.x & 7) * 4;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gv[g]) : "v"(src) : "memory");
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE & 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float a = tile[(m * 16 + (lane & 15)) * 34 + k * 4 + (lane >> 4)];
#pragma unroll
                for (int n = 0; n < 3; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc[m][n], 0, 0, 0);
            }
            if ((MODE & 33) == 33) {
                asm volatile("buffer_load_dwordx3 %0, %1, %2, %3 offen" : "=v"(wv[k]) : "v"(wv_off), "s"(srd_w), "s"(k * 4 * 384 * 4) : "memory");
            } else if (MODE & 1) {
                asm volatile("global_load_dwordx3 %0, %1, off" : "=v"(wv[k]) : "v"(pb + k * 4 * 384) : "memory");
            }
            if (MODE & 4) {
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n\tv_lshl_add_u64 %0, %0, 0, %1" : "+v"(vaddr) : "v"(rnd));
            }
        }
        // the gathered rows are needed now (in-order return: they are older than this step's 8 weight reloads)
        if ((MODE & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (MODE & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE & 2) asm volatile("" ::"v"(gv[0]), "v"(gv[1]));
        if (MODE & 16) {
            float2* dst = (float2*)(tile + ((i & 1) ^ 1) * 64 * 34 + (threadIdx.x >> 3) * 34 + (threadIdx.x & 7) * 4);
            dst[0] = make_float2(b[0], b[1]);
            dst[1] = make_float2(b[1], b[2]);
            dst[32 * 17] = make_float2(b[0], b[1]);
            dst[32 * 17 + 1] = make_float2(b[1], b[2]);
            __syncthreads();
        }
        if (MODE & 1)
            asm volatile("" ::"v"(wv[0]), "v"(wv[1]), "v"(wv[2]), "v"(wv[3]), "v"(wv[4]), "v"(wv[5]), "v"(wv[6]), "v"(wv[7]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = (float)(vaddr & 1) + (float)(sacc & 1);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (s == 123.456f) out[0] = s;
}

// Which part of the weight reload costs matrix-pipe time?  One load per k-step of 12 matrix ops in different forms
// (LD): 0 = global_load_dwordx3 v, v[addr 64-bit], off + s_waitcnt vmcnt(7) per k-step; 1 = same load, no wait in the
// loop (drain per step); 2 = the waits without the loads; 3 = global_load_dwordx3 v, v_offset, s[base]; 4 = dword
// instead of dwordx3; 5 = buffer_load_dwordx3 (resource descriptor + 32-bit offset); 6 = ds_read_b96 from LDS.
template <int LD>
__global__ __launch_bounds__(256) void mfma_ld_kernel(float* out, const float* wbuf, int iters) {
    __shared__ float tile[64 * 34 + 8 * 4 * 200];
    for (int i = threadIdx.x; i < 64 * 34 + 8 * 4 * 200; i += 256) tile[i] = 1e-9f * i;
    __syncthreads();
    f32x4 acc[4][3];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    float b[3] = {1.0f, 1.0f + 1e-9f * blockIdx.x, 0.5f};
    const int off_f = (lane >> 4) * 384 + (lane & 15) * 3 + (threadIdx.x >> 6) * 48;
    const float* pb = wbuf + off_f;
    const unsigned voff = off_f * 4u;
    i32x4 srd;
    srd[0] = (int)(unsigned)(unsigned long long)wbuf;
    srd[1] = (int)(unsigned)((unsigned long long)wbuf >> 32);
    srd[2] = 1 << 18;      // bytes
    srd[3] = 0x00020000;   // raw buffer, dword format
    const float* lb = tile + 64 * 34 + (lane >> 4) * 200 + (lane & 15) * 3 + (threadIdx.x >> 6) * 48;
    f32x3 wv[8];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (LD == 0 || LD == 2 || LD >= 3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float a = tile[(m * 16 + (lane & 15)) * 34 + k * 4 + (lane >> 4)];
#pragma unroll
                for (int n = 0; n < 3; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc[m][n], 0, 0, 0);
            }
            if (LD == 0 || LD == 1)
                asm volatile("global_load_dwordx3 %0, %1, off" : "=v"(wv[k]) : "v"(pb + k * 4 * 384) : "memory");
            if (LD == 3)
                asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(wv[k]) : "v"(voff + k * 4 * 384 * 4), "s"(wbuf) : "memory");
            if (LD == 4) asm volatile("global_load_dword %0, %1, off" : "=v"(wv[k][0]) : "v"(pb + k * 4 * 384) : "memory");
            if (LD == 5)
                asm volatile("buffer_load_dwordx3 %0, %1, %2, 0 offen" : "=v"(wv[k]) : "v"(voff + k * 4 * 384 * 4), "s"(srd) : "memory");
            if (LD == 6) {
                wv[k][0] = lb[k * 4 * 200];
                wv[k][1] = lb[k * 4 * 200 + 1];
                wv[k][2] = lb[k * 4 * 200 + 2];
            }
        }
        if (LD == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(wv[0]), "v"(wv[1]), "v"(wv[2]), "v"(wv[3]), "v"(wv[4]), "v"(wv[5]), "v"(wv[6]), "v"(wv[7]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (s == 123.456f) out[0] = s;
}

// v_mfma_f32_4x4x1_16B_f32: 16 blocks of 4 x 4 x 1 per instruction (512 flop, 8 cycles: the same 64 flop / cycle / SIMD as
// 16x16x4).  With A broadcast over the column groups and B over the row groups, one instruction is a 16 x 16 output tile at
// K = 1 whose rows could be skipped in groups of FOUR (row-slot efficiency 0.872 -> 0.943 on the level-0 plan) - at four
// times the operand instructions per flop.  FEED 0: registers only (is the peak really the same?); 1: the four A operands
// of a k from LDS (one ds_read_b32 each, reused by three column tiles); 2: plus the k's three B operands as one
// buffer_load_dwordx3 from a cache-resident block - the per-k operand traffic a conv step in this form would have.
template <int FEED>
__global__ __launch_bounds__(256) void mfma_4x4_kernel(float* out, const float* wbuf, int iters) {
    __shared__ float tile[64 * 36];
    for (int i = threadIdx.x; i < 64 * 36; i += 256) tile[i] = 1e-9f * i;
    __syncthreads();
    f32x4 acc[4][3];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    const int blk = lane >> 2, i4 = lane & 3, rg = blk >> 2, cg = blk & 3;
    float a[4] = {1e-9f * lane, 2e-9f * lane, 3e-9f, 4e-9f};
    f32x3 b = {1.0f, 1.0f + 1e-9f * blockIdx.x, 0.5f};
    i32x4 srd;
    srd[0] = (int)(unsigned)(unsigned long long)wbuf;
    srd[1] = (int)(unsigned)((unsigned long long)wbuf >> 32);
    srd[2] = 1 << 18;
    srd[3] = 0x00020000;
    const unsigned voff = (unsigned)((cg * 4 + i4) * 3 + (threadIdx.x >> 6) * 48) * 4u;
    f32x3 wv[8];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (FEED >= 2) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            if (FEED >= 1) {
#pragma unroll
                for (int m = 0; m < 4; ++m) a[m] = tile[(m * 16 + rg * 4 + i4) * 36 + k];
            }
            const f32x3 bk = FEED >= 2 ? wv[k & 7] : b;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], bk[n], acc[m][n], 0, 0, 0);
            if (FEED >= 2)
                asm volatile("buffer_load_dwordx3 %0, %1, %2, 0 offen" : "=v"(wv[k & 7]) : "v"(voff + (unsigned)k * 384u * 4u), "s"(srd) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (FEED >= 2) s += wv[0][0] + wv[7][2];
    if (s == 123.456f) out[0] = s;
}

extern "C" {
// returns 4x4x1 matrix ops per wave per launch (512 flop each)
long long mfma_4x4_launch(int feed, int blocks, int iters, float* out, const float* wbuf, hipStream_t stream) {
    switch (feed) {
        case 0: hipLaunchKernelGGL(mfma_4x4_kernel<0>, dim3(blocks), dim3(256), 0, stream, out, wbuf, iters); break;
        case 1: hipLaunchKernelGGL(mfma_4x4_kernel<1>, dim3(blocks), dim3(256), 0, stream, out, wbuf, iters); break;
        case 2: hipLaunchKernelGGL(mfma_4x4_kernel<2>, dim3(blocks), dim3(256), 0, stream, out, wbuf, iters); break;
        default: return -1;
    }
    return 32LL * 12 * iters;
}
long long mfma_ld_launch(int ld, int blocks, int iters, float* out, const float* wbuf, hipStream_t stream) {
#define LDK(M) case M: hipLaunchKernelGGL(mfma_ld_kernel<M>, dim3(blocks), dim3(256), 0, stream, out, wbuf, iters); break;
    switch (ld) {
        LDK(0) LDK(1) LDK(2) LDK(3) LDK(4) LDK(5) LDK(6)
        default: return -1;
    }
#undef LDK
    return 8LL * 12 * iters;
}
long long mfma_mix_launch(int mode, int blocks, int iters, float* out, const float* wbuf, const float* big, long long big_rows,
                          hipStream_t stream) {
#define MIX(M) case M: hipLaunchKernelGGL(mfma_mix_kernel<M>, dim3(blocks), dim3(256), 0, stream, out, wbuf, big, big_rows, iters); break;
    switch (mode) {
        MIX(0) MIX(1) MIX(2) MIX(3) MIX(4) MIX(16) MIX(19) MIX(23) MIX(33) MIX(34) MIX(35) MIX(51)
        default: return -1;
    }
#undef MIX
    return 8LL * 12 * iters;
}

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
