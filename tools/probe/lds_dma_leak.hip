// Does an LDS-DMA write (global_load_lds) of one workgroup ever land in the LDS of ANOTHER workgroup on the same CU?
// victim: fills 8 KB of LDS with a pattern and checks it for ~200 us; aggressor: 3 workgroups per CU, 48 KB each, DMA
// writes over its whole allocation in a loop.  Build: hipcc --offload-arch=gfx950 -O3 lds_dma_leak.hip -o lds_dma_leak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int BYTES>
__global__ __launch_bounds__(256) void aggressor(const unsigned* __restrict__ src, int iters, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned buf[12288];  // 48 KB
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int it = 0; it < iters; ++it) {
    for (int c = 0; c < 12; ++c) {  // 12 x 4 waves x 1 KB = 48 KB
      if (BYTES == 16)
        __builtin_amdgcn_global_load_lds(src + (size_t)(((it * 12 + c) * 4 + w) * 64 + lane) * 4, (lds_ptr)(buf + (c * 4 + w) * 256), 16, 0, 0);
      else
        for (int q = 0; q < 4; ++q)
          __builtin_amdgcn_global_load_lds(src + (size_t)((((it * 12 + c) * 4 + w) * 4 + q) * 64 + lane), (lds_ptr)(buf + (c * 4 + w) * 256 + 64 * q), 4, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
  }
  if (buf[threadIdx.x] == 0x12345u) sink[0] = 1;
}

__global__ __launch_bounds__(256) void victim(int iters, unsigned* bad, const unsigned* __restrict__ vsrc, size_t vwords) {
  __shared__ unsigned buf[2048];  // 8 KB
  const unsigned pat = 0xC0FFEE00u + blockIdx.x;
  for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = pat;
  __syncthreads();
  unsigned nbad = 0, nbad_g = 0;
  size_t pos = ((size_t)blockIdx.x * 7919 + threadIdx.x * 4) % (vwords - 4096);
  for (int it = 0; it < iters; ++it) {
    for (int i = threadIdx.x; i < 2048; i += 256)
      if (buf[i] != pat) { ++nbad; buf[i] = pat; }
    // gather-like 16-byte loads of a buffer whose word k holds k * 2654435761
    const uint4 v = *reinterpret_cast<const uint4*>(vsrc + pos);
    const unsigned k = (unsigned)pos;
    if (v.x != k * 2654435761u || v.y != (k + 1) * 2654435761u || v.z != (k + 2) * 2654435761u || v.w != (k + 3) * 2654435761u) ++nbad_g;
    pos = (pos * 5 + 4 * 1237) % (vwords - 4096);
    pos &= ~(size_t)3;
  }
  if (nbad) atomicAdd(bad, nbad);
  if (nbad_g) atomicAdd(bad + 1, nbad_g);
}
__global__ void fillk(unsigned* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i * 2654435761u;
}

int main() {
  unsigned *src, *sink, *bad;
  const size_t words = 64u << 20;
  hipMalloc(&src, words * 4); hipMemset(src, 0x5A, words * 4);
  hipMalloc(&sink, 4); hipMalloc(&bad, 8);
  unsigned* vsrc; const size_t vwords = 8u << 20; hipMalloc(&vsrc, vwords * 4);
  hipLaunchKernelGGL(fillk, dim3(1024), dim3(256), 0, 0, vsrc, vwords);
  hipStream_t sa, sb;
  hipStreamCreate(&sa); hipStreamCreate(&sb);
  for (int mode = 0; mode < 3; ++mode) {
    hipMemset(bad, 0, 8);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 20; ++rep) {
      if (mode == 1) hipLaunchKernelGGL(aggressor<16>, dim3(768), dim3(256), 0, sa, src, 200, sink);
      if (mode == 2) hipLaunchKernelGGL(aggressor<4>, dim3(768), dim3(256), 0, sa, src, 100, sink);
      hipLaunchKernelGGL(victim, dim3(1024), dim3(256), 0, sb, 3000, bad, vsrc, vwords);
    }
    hipDeviceSynchronize();
    unsigned h[2] = {0, 0};
    hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
    printf("%s: corrupted victim LDS words %u, wrong victim global loads %u  (%s)\n", mode == 0 ? "victim alone" : mode == 1 ? "beside 16-byte LDS-DMA" : "beside 4-byte LDS-DMA", h[0], h[1],
           hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
