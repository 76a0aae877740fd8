// Where the time of r3d_pointwise_gemm_bx3_kernel goes: the kernel built with parts removed (GB_ABL bits: 1 no output
// stores, 2 no three-piece cut, 4 no MFMA, 8 no global loads inside the K loop), timed at two layer shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGB_ABL=<bits> -I r3dfsseg_amd/csrc tools/probe/gemm_bx3_abl.hip -o /tmp/gb_abl_<bits>
#include "gemm_bx3.hip"
#include <stdarg.h>
#include <stdlib.h>
int g_r3d_matrix_arith = 1;
void r3d_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }

int main(int argc, char** argv) {
  const long M = argc > 1 ? atol(argv[1]) : 262144;
  const int shapes[][2] = {{192, 512}, {512, 256}, {256, 128}, {128, 64}};
  float *X, *W, *Out;
  hipMalloc(&X, M * 512 * 4); hipMalloc(&W, 512 * 512 * 4); hipMalloc(&Out, M * 512 * 4);
  hipMemset(X, 0x3c, M * 512 * 4); hipMemset(W, 0x3b, 512 * 512 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  printf("GB_ABL=%d:", GB_ABL);
  for (auto& s : shapes) {
    const int K = s[0], Co = s[1];
    for (int it = 0; it < 2; ++it) r3d_pointwise_bx3_launch(X, K, W, M, K, Co, nullptr, nullptr, 0, Out, Co, 0, nullptr, 0);
    hipEventRecord(a, 0);
    for (int it = 0; it < 10; ++it) r3d_pointwise_bx3_launch(X, K, W, M, K, Co, nullptr, nullptr, 0, Out, Co, 0, nullptr, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("  %d->%d %.1f us", K, Co, ms * 100);
  }
  printf("\n");
  return 0;
}
