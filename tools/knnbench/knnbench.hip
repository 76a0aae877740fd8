// Micro-benchmark harness for knn.hip ablations (build-time -DABL_* switches).
// build (from this directory; the other translation units come from the library's objects):
//   hipcc <the flags of r3dfsseg_amd/build.py> [-DKNN_STAMPS] -c knnbench.hip -o kb.o && hipcc --offload-arch=gfx950 kb.o $(ls ../../r3dfsseg_amd/csrc/*.o | grep -v /knn.o) -o kb
#include "../../r3dfsseg_amd/csrc/knn.hip"
extern "C" const char* r3d_last_error_string(void);
#include <vector>
#include <cstdlib>
int main(int argc, char** argv) {
  // usage: kb [C=64] [k=20] [mode=0] [N=2048] [B=12] [bf: 0 none, 1 threshold pass on bf16, 2 + filter pass on bf16]
  int B = 12, N = 2048, C = argc > 1 ? atoi(argv[1]) : 64, k = argc > 2 ? atoi(argv[2]) : 20, mode = argc > 3 ? atoi(argv[3]) : 0;
  if (argc > 4) { B = 1; N = atoi(argv[4]); }
  if (argc > 5) B = atoi(argv[5]);
  const int bf = argc > 6 ? atoi(argv[6]) : 0;
  r3d_debug_set_knn_bf16_filter(bf >= 2 ? bf - 1 : 0);  // bf 2: filter in the k <= 32 configuration, 3: everywhere
  std::vector<float> h((size_t)B * N * C);
  srand(1);
  for (auto& v : h) v = rand() / (float)RAND_MAX - 0.5f;
  float *x, *nrm, *cm; int* idx; int* status; hipMalloc(&status, 4);
  hipMalloc(&cm, (size_t)B * C * (N + 64) * 4);
  float* bfws = nullptr;
  const long bfw = bf ? r3d_knn_bf_ws_words(B, N, C) : 0;
  if (bf) hipMalloc(&bfws, (size_t)bfw * 4);
  hipMalloc(&x, h.size() * 4); hipMalloc(&nrm, (size_t)r3d_knn_norm_ws_words(B, N) * 4); hipMalloc(&idx, (size_t)B * N * k * 4);
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) r3d_knn_topk_batched(x, C, nullptr, B, N, C, k, mode, nullptr, 0, nrm, cm, idx, nullptr, k > 32 ? status : nullptr, nullptr, 0, bfws, bfw, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int reps = 10;
  for (int it = 0; it < reps; ++it) r3d_knn_topk_batched(x, C, nullptr, B, N, C, k, mode, nullptr, 0, nrm, cm, idx, nullptr, k > 32 ? status : nullptr, nullptr, 0, bfws, bfw, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  int hs = 0; hipMemcpy(&hs, status, 4, hipMemcpyDeviceToHost);
  printf("B=%d N=%d C=%d k=%d mode=%d bf=%d: %.1f us per call (%s) status=%d\n", B, N, C, k, mode, bf, ms * 1000 / reps, r3d_last_error_string(), hs);
#ifdef KNN_STAMPS
  unsigned long long z[16] = {0}, d[16];
  hipMemcpyToSymbol(HIP_SYMBOL(g_knn_dbg), z, sizeof(z));
  r3d_knn_topk_batched(x, C, nullptr, B, N, C, k, mode, nullptr, 0, nrm, cm, idx, nullptr, k > 32 ? status : nullptr, nullptr, 0, bfws, bfw, 0);
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(d, HIP_SYMBOL(g_knn_dbg), sizeof(d));
  printf("  append stamps (ticks): stage %llu  passA %llu  tau %llu  passB %llu  lists %llu  exact %llu  rank %llu\n", d[9]-d[8], d[10]-d[9], d[11]-d[10], d[12]-d[11], d[14]-d[12], d[15]-d[14], d[13]-d[15]);
  if (getenv("KB_RAW")) { for (int i = 0; i < 16; ++i) printf(" d[%d]=%llu", i, d[i]); printf("\n"); }
  printf("  stamps: passA %llu  tau %llu  passB %llu  merge %llu\n", d[1]-d[0], d[2]-d[1], d[3]-d[2], d[4]-d[3]);
#endif
  return 0;
}
