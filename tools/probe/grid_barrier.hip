// Cost of a device-wide barrier inside a persistent kernel on MI355X, for the fence placements one can choose.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__device__ __forceinline__ void gbar(unsigned* bar, unsigned& phase) {
  if (MODE == 0) __threadfence();
  __syncthreads();
  ++phase;
  if (threadIdx.x == 0) {
    if (MODE == 1) __threadfence();
    if (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = phase * gridDim.x;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    if (MODE == 1) __threadfence();
    if (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (MODE == 0) __threadfence();
}
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* bar, float* data, int iters) {
  unsigned phase = 0;
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    data[(blockIdx.x * 256 + threadIdx.x)] = acc + i;      // a write every phase
    gbar<MODE>(bar, phase);
    acc += data[((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x];  // read the neighbour's write
  }
  if (acc == -1.f) data[0] = acc;
}
template <int MODE>
void run(int G, unsigned* bar, float* data) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 500;
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(bar, 0, 4);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(G), dim3(256), 0, 0, bar, data, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("mode %d (0 all-thread fences, 1 thread-0 __threadfence, 2 no fence, 3 thread-0 rel/acq)  G=%3d  %.2f us per barrier\n", MODE, G, ms * 1e3 / iters);
}
int main() {
  unsigned* bar; float* data;
  hipMalloc(&bar, 4); hipMalloc(&data, 1024 * 256 * 4);
  for (int G : {32, 64, 128, 256}) { run<0>(G, bar, data); run<1>(G, bar, data); run<3>(G, bar, data); run<2>(G, bar, data); }
  return 0;
}
