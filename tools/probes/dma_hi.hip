// Probe: can `buffer_load_dwordx4 ... lds` target LDS addresses beyond 64 KiB on gfx950 (160 KiB LDS)?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(const uint32_t* in, uint32_t* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) uint32_t smem[38 * 1024];   // 152 KiB
  for (int i = threadIdx.x; i < 38 * 1024; i += 256) smem[i] = 0xDEAD0000u + (i >> 8);
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, nbytes, 0x00020000);
  const int base_words[3] = {1024, 20 * 1024, 36 * 1024};               // 4 KiB, 80 KiB, 144 KiB
  if (threadIdx.x < 64) {
    for (int j = 0; j < 3; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + base_words[j]), 16,
                                               (uint32_t)(threadIdx.x * 16 + j * 1024), 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int j = 0; j < 3; ++j) out[j * 256 + threadIdx.x] = smem[base_words[j] + threadIdx.x];
}
int main() {
  uint32_t h[768], *din, *dout;
  for (int i = 0; i < 768; ++i) h[i] = 1000 + i;
  hipMalloc(&din, 3072); hipMalloc(&dout, 3072);
  hipMemcpy(din, h, 3072, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, din, dout, 3072);
  hipError_t e = hipDeviceSynchronize();
  hipMemcpy(h, dout, 3072, hipMemcpyDeviceToHost);
  printf("status %d\n", (int)e);
  for (int j = 0; j < 3; ++j) printf("dest %d: words %u %u ... %u (expect %d %d ... %d)\n", j, h[j * 256], h[j * 256 + 1], h[j * 256 + 255], 1000 + j * 256, 1001 + j * 256, 1255 + j * 256);
  return 0;
}
