// What limits a chain of v_mfma_f32_32x32x2_f32 fed like the kNN / EdgeConv / GEMM kernels feed it?
// Per unit: 32 MFMAs; A operand from LDS (MODE bit 0) or registers; B operand re-loaded from global per unit
// (MODE bit 1, 32 dword loads, L2 resident) or kept in registers.  WAVES waves per workgroup, grid = 2 per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ g, float* __restrict__ out, int units, long ld) {
  __shared__ float A[32 * 65];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  for (int e = threadIdx.x; e < 32 * 65; e += 256) A[e] = g[e];
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bf[32], af[32];
  for (int s = 0; s < 32; ++s) { bf[s] = g[(2 * s + h) * ld + j]; af[s] = A[j * 65 + 2 * s + h]; }
  const float* ap = A + j * 65 + h;
  for (int u = 0; u < units; ++u) {
    float bn[32];
    if (MODE & 2) {
      const float* p = g + (long)((blockIdx.x * 4 + w + u) & 63) * 32 + j;
#pragma unroll
      for (int s = 0; s < 32; ++s) bn[s] = p[(2 * s + h) * ld];
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const float a = (MODE & 1) ? ap[2 * s] : af[s];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bf[s], acc, 0, 0, 0);
    }
    if (MODE & 2) {
#pragma unroll
      for (int s = 0; s < 32; ++s) bf[s] = bn[s];
    }
  }
  float t = 0.f;
  for (int r = 0; r < 16; ++r) t += acc[r];
  if (t == 123.f) out[threadIdx.x] = t;
}
template <int MODE> void run(const float* g, float* out, const char* what) {
  const int units = 512, grid = 512;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, g, out, units, 2048L);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, g, out, units, 2048L);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flops = (double)grid * 4 * units * 32 * 4096.0;
  printf("%-44s %7.1f us  %6.1f TFLOP/s\n", what, ms * 1e3, flops / (ms * 1e-3) / 1e12);
}
int main() {
  float *g, *out; hipMalloc(&g, 64 * 2048 * 4 + 65536); hipMalloc(&out, 4096);
  hipMemset(g, 0, 64 * 2048 * 4 + 65536);
  run<0>(g, out, "A regs, B regs (pure MFMA chain)");
  run<1>(g, out, "A from LDS, B regs");
  run<2>(g, out, "A regs, B re-loaded from global per unit");
  run<3>(g, out, "A from LDS, B from global (kNN-like)");
  return 0;
}
