// Shared pieces of the EdgeConv kernels (edgeconv.hip: inference, edgeconv_train.hip: training forward / backward).
//
// Unit of work = 4 points = 4K edge rows = RT row tiles of 16 (K = 4 RT, K % 4 == 0), one workgroup of 4 waves per
// unit, on v_mfma_f32_16x16x4_f32: wave w gathers the K edges of point w and owns the 16 output channels 16w.. of
// the 64x64 edge GEMM (its W2 fragments stay in 16 registers).  Every phase is balanced over the 4 SIMDs and the
// LDS footprint (two [4K][68] fp32 tiles, 44 KB at K = 20) leaves room for 3 workgroups per CU, whose gather / MFMA
// / store phases overlap.  (The first version used 8-point units on 32x32x2: 5 waves on 4 SIMDs, 58..102 KB of
// LDS, one or two workgroups per CU; the change gave 1.35x..1.6x on these kernels.)
//
// LDS rows have a stride of 68 words and the k index of every MFMA is assigned so that operand reads are
// conflict-free: lane group g = lane >> 4 takes k = 16g..16g+15 of a row-major operand, read as four b128.  The
// summation order over k therefore differs from the ascending chain -- no result of these kernels decides an index.
#pragma once
#include "common.h"

#define E2_PTS 4
#define E2_LD 68

static __device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// acc[t] (rows 16t.., this wave's 16 columns) = A[rows][0..63] * B; A row-major in LDS, B fragments in registers:
// Bf[s] = B[k = 16g + s][column n] with n = lane & 15, g = lane >> 4
template <int RT>
static __device__ __forceinline__ void e2_rowgemm(const float* __restrict__ A, const float (&Bf)[16], int n, int g,
                                                  f32x4 (&acc)[RT]) {
#pragma unroll
  for (int t = 0; t < RT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int qk = 0; qk < 4; ++qk) {  // four k per lane and step group: one b128 per row tile
    float4 av[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) av[t] = *(const float4*)(A + (16 * t + n) * E2_LD + 16 * g + 4 * qk);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int t = 0; t < RT; ++t) {  // RT independent accumulator chains back to back (40-cycle dependent latency)
        const float a = s == 0 ? av[t].x : s == 1 ? av[t].y : s == 2 ? av[t].z : av[t].w;
        acc[t] = mfma16(a, Bf[4 * qk + s], acc[t]);
      }
    }
  }
}

// workgroups of `kernel` (256 threads, `lds` bytes of dynamic LDS) the chip holds at once, capped: the grid of a
// persistent loop over the units.  Sets the dynamic-LDS attribute.  0 on failure.
template <typename KernelT>
static int e2_resident_blocks(KernelT kernel, size_t lds, int cap) {
  if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
      hipGetDeviceProperties(&prop, dev) != hipSuccess || per_cu <= 0)
    return 0;
  const int r = per_cu * prop.multiProcessorCount;
  return r < cap ? r : cap;
}
