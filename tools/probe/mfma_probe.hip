// Hardware probe for the assumptions the bit-exact kNN design rests on (gfx950).
//   1. v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 operand + C/D lane maps.
//   2. The MFMA result is bitwise a k-ordered fp32 fmaf chain (k ascending).
//   3. A VALU __builtin_fmaf chain gives the same bits.
//   4. sqrtf and fp32 division are correctly rounded in default hipcc mode.
//   5. expf: report max ulp distance to the host libm (information only).
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__,   \
             __LINE__);                                                    \
      exit(2);                                                             \
    }                                                                      \
  } while (0)

__global__ void k_mfma32(const float* A, const float* B, float* C, int K) {
  int l = threadIdx.x;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 2) {
    int k = k0 + (l >> 5);
    float a = A[(l & 31) * K + k];
    float b = B[k * 32 + (l & 31)];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    C[row * 32 + (l & 31)] = acc[r];
  }
}

__global__ void k_mfma16(const float* A, const float* B, float* C, int K) {
  int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 4) {
    int k = k0 + (l >> 4);
    float a = A[(l & 15) * K + k];
    float b = B[k * 16 + (l & 15)];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r;
    C[row * 16 + (l & 15)] = acc[r];
  }
}

__global__ void k_valu(const float* A, const float* B, float* C, int M, int N,
                       int K) {
  int i = blockIdx.x, j = threadIdx.x;
  if (j >= N) return;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = __builtin_fmaf(A[i * K + k], B[k * N + j], acc);
  C[i * N + j] = acc;
}

__global__ void k_math(const float* x, const float* y, float* s, float* d,
                       float* e, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  s[i] = sqrtf(x[i]);
  d[i] = x[i] / y[i];
  e[i] = expf(-x[i]);
}

static int ulp_diff(float a, float b) {
  int ia, ib;
  memcpy(&ia, &a, 4);
  memcpy(&ib, &b, 4);
  return abs(ia - ib);
}

int main() {
  srand(1234);
  const int K = 64;
  int fails = 0;
  {  // 32x32x2
    std::vector<float> A(32 * K), B(K * 32), C(32 * 32), R(32 * 32), Rrev(32 * 32);
    for (auto& v : A) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto& v : B) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        float acc = 0.f, acc2 = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(A[i * K + k], B[k * 32 + j], acc);
        for (int k0 = 0; k0 < K; k0 += 2) {  // pair-reversed order
          acc2 = fmaf(A[i * K + k0 + 1], B[(k0 + 1) * 32 + j], acc2);
          acc2 = fmaf(A[i * K + k0], B[k0 * 32 + j], acc2);
        }
        R[i * 32 + j] = acc;
        Rrev[i * 32 + j] = acc2;
      }
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, A.size() * 4));
    CK(hipMalloc(&dB, B.size() * 4));
    CK(hipMalloc(&dC, C.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma32, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0, badrev = 0;
    for (int i = 0; i < 1024; ++i) {
      bad += memcmp(&C[i], &R[i], 4) != 0;
      badrev += memcmp(&C[i], &Rrev[i], 4) != 0;
    }
    printf("mfma_f32_32x32x2: mismatches vs k-ascending fmaf chain = %d / 1024 (pair-reversed: %d)\n", bad, badrev);
    fails += bad != 0;
    // VALU chain
    hipLaunchKernelGGL(k_valu, dim3(32), dim3(64), 0, 0, dA, dB, dC, 32, 32, K);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    bad = 0;
    for (int i = 0; i < 1024; ++i) bad += memcmp(&C[i], &R[i], 4) != 0;
    printf("VALU fmaf chain: mismatches vs host fmaf chain = %d / 1024\n", bad);
    fails += bad != 0;
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }
  {  // 16x16x4
    std::vector<float> A(16 * K), B(K * 16), C(256), R(256);
    for (auto& v : A) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto& v : B) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(A[i * K + k], B[k * 16 + j], acc);
        R[i * 16 + j] = acc;
      }
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, A.size() * 4));
    CK(hipMalloc(&dB, B.size() * 4));
    CK(hipMalloc(&dC, C.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma16, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += memcmp(&C[i], &R[i], 4) != 0;
    printf("mfma_f32_16x16x4: mismatches vs k-ascending fmaf chain = %d / 256\n", bad);
    fails += bad != 0;
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }
  {  // sqrt / div / exp
    const int n = 1 << 20;
    std::vector<float> x(n), y(n), s(n), d(n), e(n);
    for (int i = 0; i < n; ++i) {
      x[i] = (rand() / (float)RAND_MAX) * 400.f + 1e-6f;
      y[i] = (rand() / (float)RAND_MAX) * 7.f + 1e-3f;
    }
    float *dx, *dy, *ds, *dd, *de;
    CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dy, n * 4)); CK(hipMalloc(&ds, n * 4));
    CK(hipMalloc(&dd, n * 4)); CK(hipMalloc(&de, n * 4));
    CK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_math, dim3(n / 256), dim3(256), 0, 0, dx, dy, ds, dd, de, n);
    CK(hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(d.data(), dd, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(e.data(), de, n * 4, hipMemcpyDeviceToHost));
    int bs = 0, bd = 0, maxe = 0;
    for (int i = 0; i < n; ++i) {
      float rs = sqrtf(x[i]), rd = x[i] / y[i], re = expf(-x[i]);
      bs += memcmp(&rs, &s[i], 4) != 0;
      bd += memcmp(&rd, &d[i], 4) != 0;
      int u = ulp_diff(re, e[i]);
      if (u > maxe) maxe = u;
    }
    printf("sqrtf mismatches = %d / %d ; div mismatches = %d / %d ; expf max ulp diff = %d\n", bs, n, bd, n, maxe);
    fails += (bs != 0) + (bd != 0);
  }
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s, CUs=%d, clock=%d kHz, LDS/block=%zu\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
  printf(fails ? "PROBE: FAIL (%d)\n" : "PROBE: OK\n", fails);
  return fails ? 1 : 0;
}
