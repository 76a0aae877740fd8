// Shared helpers for the gfx950 kernels of the R3DFSSeg hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define R3D_OK 0
#define R3D_ERR_ARG 1
#define R3D_ERR_LAUNCH 2
#define R3D_ERR_UNSUPPORTED 3

void r3d_set_error(const char* fmt, ...);

#define R3D_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      r3d_set_error(__VA_ARGS__);       \
      return R3D_ERR_ARG;               \
    }                                   \
  } while (0)

#define R3D_LAUNCH_CHECK(name)                                              \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      r3d_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
      return R3D_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define R3D_WAVE 64

static __device__ __forceinline__ int r3d_lane() { return threadIdx.x & 63; }

// v_mfma_f32_32x32x2_f32 accumulator row of register r for this lane
// (col = lane & 31): MI355X guide, "Fragment layout".
static __device__ __forceinline__ int r3d_acc_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

static __device__ __forceinline__ float r3d_readlane_f(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

static __device__ __forceinline__ float r3d_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
static __device__ __forceinline__ float r3d_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// keep ? v : 0 WITHOUT a select on the loaded value: hipcc (CodeGenPrepare) turns
// `cond ? load : 0` -- even with the load hoisted by hand -- into a branch around the load and
// then waits vmcnt(0) after every such load, serialising a whole staging pass.
static __device__ __forceinline__ float r3d_keep(float v, bool keep) {
  return __uint_as_float(__float_as_uint(v) & (keep ? 0xffffffffu : 0u));
}

static inline int r3d_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Device-to-device fill / copy as plain KERNELS.  The library's launch sequences are frozen into hipGraphs
// (episode_graph.py); hipMemsetAsync / hipMemcpyAsync would become memset / memcpy graph nodes, and on this
// stack (ROCm 7.2, measured with tools/fault_probe.py) a graph holding such nodes replayed wrongly once the
// device had been synchronised between replays: stale neighbour bitmaps and status words, then a GPU memory
// fault in r3d_graph_weights_kernel.  With kernel nodes only the replays are stable, so every byte the
// library clears or moves goes through these two kernels.  Sizes are in 32-bit words.
static __global__ void r3d_fill_words_kernel(unsigned* __restrict__ p, unsigned v, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
static __global__ void r3d_copy_words_kernel(unsigned* __restrict__ dst, const unsigned* __restrict__ src, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}
static inline void r3d_zero_words(void* p, long n_words, hipStream_t st) {
  const int g = (int)(n_words < 256L * 2048 ? (n_words + 255) / 256 : 2048);
  hipLaunchKernelGGL(r3d_fill_words_kernel, dim3(g > 0 ? g : 1), dim3(256), 0, st, (unsigned*)p, 0u, n_words);
}
static inline void r3d_copy_words(void* dst, const void* src, long n_words, hipStream_t st) {
  const int g = (int)(n_words < 256L * 2048 ? (n_words + 255) / 256 : 2048);
  hipLaunchKernelGGL(r3d_copy_words_kernel, dim3(g > 0 ? g : 1), dim3(256), 0, st, (unsigned*)dst, (const unsigned*)src,
                     n_words);
}
