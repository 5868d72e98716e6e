// Probe: what does an out-of-range `buffer_load_dwordx4 ... lds` write to LDS on gfx950 - zeros, or nothing?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(const uint32_t* in, uint32_t* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) uint32_t smem[256 + 256];
  smem[threadIdx.x] = 0xDEADBEEFu; smem[256 + threadIdx.x] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, nbytes, 0x00020000);
  // lanes 0..31 in range, lanes 32..63 out of range (bit 31 set)
  uint32_t off = threadIdx.x < 32 ? threadIdx.x * 16 : 0x80000000u + threadIdx.x * 16;
  if (threadIdx.x < 64)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[threadIdx.x] = smem[threadIdx.x];
}
int main() {
  uint32_t h[256], *din, *dout;
  for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
  hipMalloc(&din, 1024); hipMalloc(&dout, 1024);
  hipMemcpy(din, h, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, din, dout, 512);
  hipMemcpy(h, dout, 1024, hipMemcpyDeviceToHost);
  printf("in-range lane 0 words: %u %u %u %u\n", h[0], h[1], h[2], h[3]);
  printf("in-range lane 31 words: %u %u %u %u\n", h[124], h[125], h[126], h[127]);
  printf("out-of-range lane 32 words: %08x %08x %08x %08x\n", h[128], h[129], h[130], h[131]);
  printf("out-of-range lane 63 words: %08x %08x %08x %08x\n", h[252], h[253], h[254], h[255]);
  return 0;
}
