// Do small kernels on different HIP streams co-execute on MI355X?  N launches of a ~10 us one-workgroup kernel per
// stream, on 1 / 2 / 4 / 8 streams, eager launches and hipGraph replays.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) a = __builtin_fmaf(a, b, 1e-7f);
  if (a == 123.f) out[threadIdx.x] = a;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  float* out; hipMalloc(&out, 4096);
  const int N = 2000, iters = 6000;
  std::vector<hipStream_t> st(8);
  for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[0], out, iters); hipDeviceSynchronize();
  for (int ns : {1, 2, 4, 8}) {
    double t0 = now();
    for (int i = 0; i < N; ++i) for (int s = 0; s < ns; ++s) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[s], out, iters);
    hipDeviceSynchronize();
    double t = now() - t0;
    printf("eager  streams=%d  %.2f us per kernel per stream (total %.1f ms)\n", ns, t / N * 1e6, t * 1e3);
  }
  // graphs: a chain of 200 kernels captured per stream
  std::vector<hipGraphExec_t> ge(8);
  for (int s = 0; s < 8; ++s) {
    hipGraph_t g; hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[s], out, iters);
    hipStreamEndCapture(st[s], &g); hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0);
  }
  for (int ns : {1, 2, 4, 8}) {
    for (int s = 0; s < ns; ++s) hipGraphLaunch(ge[s], st[s]);
    hipDeviceSynchronize();
    double t0 = now();
    for (int r = 0; r < 10; ++r) for (int s = 0; s < ns; ++s) hipGraphLaunch(ge[s], st[s]);
    hipDeviceSynchronize();
    double t = now() - t0;
    printf("graph  streams=%d  %.2f us per kernel per stream (total %.1f ms)\n", ns, t / 2000 * 1e6, t * 1e3);
  }
  return 0;
}
