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

// Sum over aligned groups of 8 lanes, every lane gets the total: three DPP steps inside the VALU (lane ^ 1, lane ^ 2,
// mirror of the 8-lane half row).  No LDS crossbar (ds_bpermute / ds_swizzle), so nothing another workgroup does to the
// CU's LDS pipeline can reach it.
static __device__ __forceinline__ float r3d_sum8_dpp(float v) {
  int x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true));  // row_half_mirror
  return v;
}

// Inclusive prefix sum over the 64 lanes in the VALU (DPP row shifts inside the rows of 16, then the row totals carried
// over by row_bcast:15 and row_bcast:31): 12 instructions, no LDS crossbar round trips (six ds_bpermute of __shfl_up are
// ~100 cycles each).
static __device__ __forceinline__ int r3d_wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); // row_bcast:31 into rows 2 and 3
  return v;
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

// ---------------------------------------------------------------------------------------------------------------
// fp32 products on the bf16 matrix core ("bf16 x 3").  An fp32 value is cut into three bf16 pieces by TRUNCATION,
//   x = hi + mid + lo  exactly  (24 significant bits = 8 + 8 + 8; x - hi and (x - hi) - mid are exact in fp32),
// and a product block a.b is summed from the six piece products of weight >= 2^-16,
//   hi.hi + hi.mid + mid.hi + mid.mid + hi.lo + lo.hi        (dropped: mid.lo, lo.mid, lo.lo <= 2^-23 |a||b|),
// each a v_mfma_f32_32x32x16_bf16 accumulating in fp32.  Six bf16 MFMAs cover K = 16 in 6 x 32 cycles where
// v_mfma_f32_32x32x2_f32 needs 8 x 64: 2.67x the fp32 matrix rate at fp32-level accuracy (relative error of a product
// 2^-22 against 2^-24).  Only kernels that decide NO index use it (kNN scores stay on the fp32 core, bit-exact).
// A fragment is 8 bf16 per lane and piece: r3d_bx3 holds the three u32x4 of one k-step.
typedef unsigned r3d_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 r3d_bf16x8 __attribute__((ext_vector_type(8)));
extern int g_r3d_matrix_arith;  // 0: fp32 MFMA everywhere; 1: bf16 x 3 in the kernels that have that form (error.hip)

// pieces of two values, packed as (x1 << 16 | x0) bf16 pairs
static __device__ __forceinline__ void r3d_bx3_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
  const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
  const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
  const float t0 = r0 - __uint_as_float(v0 & 0xffff0000u), t1 = r1 - __uint_as_float(v1 & 0xffff0000u);
  h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
  m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
  l = __builtin_amdgcn_perm(__float_as_uint(t1), __float_as_uint(t0), 0x07060302u);
}
struct r3d_bx3 {
  r3d_u32x4 h, m, l;
};
static __device__ __forceinline__ r3d_bx3 r3d_bx3_split8(const float* x) {
  r3d_bx3 f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned h, m, l;
    r3d_bx3_split2(x[2 * i], x[2 * i + 1], h, m, l);
    f.h[i] = h; f.m[i] = m; f.l[i] = l;
  }
  return f;
}
static __device__ __forceinline__ f32x16 r3d_mfma_bf16(r3d_u32x4 a, r3d_u32x4 b, f32x16 c) {
#if defined(ATT_ABL) && (ATT_ABL & 4)
  c[0] += __uint_as_float(a[0] ^ b[1]);  // probe build: no matrix core
  return c;
#endif
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(r3d_bf16x8, a), __builtin_bit_cast(r3d_bf16x8, b), c, 0,
                                                 0, 0);
}
// c += A B over one k-step of 16, both operands in pieces (smallest terms first)
static __device__ __forceinline__ f32x16 r3d_bx3_mma(const r3d_bx3& a, const r3d_bx3& b, f32x16 c) {
  c = r3d_mfma_bf16(a.l, b.h, c);
  c = r3d_mfma_bf16(a.h, b.l, c);
  c = r3d_mfma_bf16(a.m, b.m, c);
  c = r3d_mfma_bf16(a.m, b.h, c);
  c = r3d_mfma_bf16(a.h, b.m, c);
  c = r3d_mfma_bf16(a.h, b.h, c);
  return c;
}

extern int g_r3d_gemm_bx3;  // bits: 1 point-wise GEMM on bf16 x 3, 2 weight-gradient GEMM on bf16 x 3 (error.hip)

static inline int r3d_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// XCD-aware work order (cdna_hip_programming.md 5.5, T1).  Workgroups are dealt to the 8 XCDs round robin in launch
// order, so blocks b and b + 8 share an XCD (and its L2); the remap hands every such group a CONTIGUOUS range of the
// work items -- the tiles of one cloud, the chunks of one cloud -- so that what they re-read or gather (a cloud's K / V,
// its PQ rows) is fetched into ONE L2 instead of all eight.  Bijective for any item count; a speed hint only: results
// never depend on it (every item is processed exactly once whatever the placement).
static __device__ __forceinline__ int r3d_xcd_swizzle(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

// Row (or cloud) segments of a batch of episodes: segments of `a` and `b` units alternate, [a | b | a | b | ...]
// (a = support part, b = query part of one episode; b == 0: every segment has `a` units).  Segment 2 e + p is part p of
// episode e.  BatchNorm statistics, dropout seeds and every per-episode quantity of the training path are keyed by it.
struct r3d_segmap {
  long a, b;
  __host__ __device__ bool odd(int seg) const { return b > 0 && (seg & 1); }
  __host__ __device__ long seg_rows(int seg) const { return odd(seg) ? b : a; }
  __host__ __device__ long seg_row0(int seg) const {
    return b > 0 ? (long)(seg >> 1) * (a + b) + (odd(seg) ? a : 0) : (long)seg * a;
  }
  __host__ __device__ int seg_of_row(long r) const {
    if (b == 0) return (int)(r / a);
    const long p = r / (a + b);
    return (int)(2 * p + ((r - p * (a + b)) >= a ? 1 : 0));
  }
  // the same for totals below 2^31 (every entry point checks): 32-bit division, a fraction of the VALU work of the
  // 64-bit one, which matters in element-wise kernels that look a segment up per row
  __device__ int seg_of_row32(int r) const {
    const int A = (int)a, B = (int)b;
    if (B == 0) return r / A;
    const int p = r / (A + B);
    return 2 * p + ((r - p * (A + B)) >= A ? 1 : 0);
  }
  __host__ __device__ int n_seg(long total) const { return (int)(b > 0 ? 2 * (total / (a + b)) : total / a); }
  __host__ __device__ bool covers(long total) const {
    return a > 0 && b >= 0 && total > 0 && total < 0x7fffffffL && total % (a + b) == 0;
  }
};

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
// the same fill for every episode of a batch: blockIdx.y = episode, its words start stride_words further on
static __global__ void r3d_fill_words_ep_kernel(unsigned* __restrict__ p, unsigned v, long n, long stride_words) {
  p += (long)blockIdx.y * stride_words;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
static inline void r3d_fill_words_ep(void* p, unsigned v, long n_words, int n_ep, long stride_words, hipStream_t st) {
  const int g = (int)(n_words < 256L * 2048 ? (n_words + 255) / 256 : 2048);
  hipLaunchKernelGGL(r3d_fill_words_ep_kernel, dim3(g > 0 ? g : 1, n_ep), dim3(256), 0, st, (unsigned*)p, v, n_words, stride_words);
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
