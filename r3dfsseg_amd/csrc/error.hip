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

extern "C" int r3d_abi_version(void) { return 2; }

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
