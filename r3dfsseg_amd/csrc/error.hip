// Thread-local error string behind the C ABI (no C++ exceptions cross it).
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void r3d_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* r3d_last_error_string(void) { return g_err; }

extern "C" int r3d_abi_version(void) { return 4; }

// Arithmetic of the GEMM-shaped kernels that decide no index (attention; see common.h "bf16 x 3"): 0 = fp32 matrix
// core, 1 = fp32 values as three bf16 pieces on the bf16 matrix core (default: same accuracy against a float64
// reference, 1.6 - 1.8x the kernel speed).  Process-wide; set before the first launch.
int g_r3d_matrix_arith = 1;
extern "C" int r3d_set_matrix_arith(int mode) {
  R3D_REQUIRE(mode == 0 || mode == 1, "r3d_set_matrix_arith: mode %d (0 = fp32 MFMA, 1 = bf16 x 3)", mode);
  g_r3d_matrix_arith = mode;
  return R3D_OK;
}
extern "C" int r3d_get_matrix_arith(void) { return g_r3d_matrix_arith; }
// Which of the index-free GEMMs take the bf16 x 3 form when the mode above is 1 (bit 0: point-wise, bit 1: weight
// gradient; bit 2: the point-wise kernel with W cut once per call, gemm_bx3.hip).  Default: all.  An A/B knob for tests
// and tools/probe/gemm_arith_sweep.sh.
int g_r3d_gemm_bx3 = 1 | 2 | 4;
extern "C" int r3d_debug_set_gemm_bx3(int mask) {
  R3D_REQUIRE(mask >= 0 && mask < 8, "r3d_debug_set_gemm_bx3: mask %d", mask);
  g_r3d_gemm_bx3 = mask;
  return R3D_OK;
}

// Test utility: leave `pattern` in every byte of LDS the chip has (64 KB per workgroup, enough workgroups to visit every
// CU several times).  A kernel that reads LDS it has not written sees this instead of whatever ran before it: the tests
// use it to show that no result depends on stale LDS contents.
__global__ __launch_bounds__(256) void r3d_lds_poison_kernel(unsigned pattern, unsigned* __restrict__ sink) {
  __shared__ unsigned buf[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) buf[i] = pattern;
  __syncthreads();
  if (buf[(threadIdx.x * 61 + blockIdx.x) & 16383] != pattern) sink[0] = 1;  // keeps the stores alive
}
extern "C" int r3d_debug_poison_lds(unsigned pattern, unsigned* sink, void* stream) {
  R3D_REQUIRE(sink, "r3d_debug_poison_lds: null pointer");
  hipLaunchKernelGGL(r3d_lds_poison_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, pattern, sink);
  R3D_LAUNCH_CHECK("r3d_debug_poison_lds");
  return R3D_OK;
}
