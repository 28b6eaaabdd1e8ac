#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// Does a buffer_load ... lds (LDS-DMA) whose offset is out of range write ZEROS to its LDS slot, or nothing?
__global__ void k(const float* in, int nbytes, float* out) {
  __shared__ float buf[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) buf[i] = -7.0f;   // sentinel
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, nbytes, 0x00020000);
  unsigned off = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16u;   // odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)buf, 16, off, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = buf[i];
}
int main() {
  float *in, *out; std::vector<float> h(1024), o(256);
  for (int i = 0; i < 1024; ++i) h[i] = i + 1;
  hipMalloc(&in, 4096); hipMalloc(&out, 1024); hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, 4096, out);
  hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 6; ++l) printf("lane %d: %g %g %g %g\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
  return 0;
}
