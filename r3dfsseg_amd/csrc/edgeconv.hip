// Fused EdgeConv for gfx950: neighbour gather + edge MLP + max over the K neighbours,
// without materialising the (B, 2C, N, K) edge tensor.
//
// Replaces (reference): models/dgcnn.py:26-42 get_edge_feature, :45-61 conv2d
// ([Conv2d 1x1 -> BN -> LeakyReLU(0.2)] x 2) and the max over K at :118.
//
// Algebra: with W1 = [Wa | Wb] (64 x 2C), W1 [x_j - x_i ; x_i] = Wa x_j + (Wb - Wa) x_i.
// A point-wise GEMM (gemm.hip) produces per point PQ[p] = [P | Q], P = s1 * Wa x,
// Q = s1 * (Wb - Wa) x + t1 (BN1 folded in eval mode), so the first edge layer is
//   h1(i, j) = lrelu(P[j] + Q[i])
// and only the second 64x64 layer runs per edge, on the matrix core:
//   out[i] = max_j lrelu(s2 * (W2 h1(i, j)) + t2).
//
// Work unit: 4 points = 4K edges (K % 4 == 0), 4 waves (edge_tile.h): wave w gathers the K neighbour rows of point w
// (256-B coalesced rows, all K in flight) into LDS, the 64x64 layer runs on 16x16x4 MFMAs with wave w owning output
// channels 16w..16w+15 (W2 fragments in registers), the activated block goes to a second LDS tile and the max over
// each point's K rows is a column scan (point w, channel lane).  Workgroups walk units with a grid stride; 44 KB of
// LDS at K = 20 keep 3 workgroups per CU.
#include "edge_tile.h"

template <int RT>
__global__ __launch_bounds__(256) void r3d_edgeconv_kernel(
    const float* __restrict__ PQ, const int* __restrict__ idx, const float* __restrict__ W2,
    const float* __restrict__ s2, const float* __restrict__ t2, float* __restrict__ out, long ldo,
    int N, long total_points, int* __restrict__ argmax_out) {
  constexpr int K = 4 * RT, R = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* H = smem;              // [R][E2_LD] h1
  float* Z = smem + R * E2_LD;  // [R][E2_LD] activated second layer
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int c = 16 * w + n;
  float Bz[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) Bz[s] = W2[c * 64 + 16 * g + s];  // z2[e][c] = sum_k h1[e][k] W2[c][k]
  const float s2c = s2[c], t2c = t2[c];
  const long units = total_points / E2_PTS;
  for (long item = blockIdx.x; item < units; item += gridDim.x) {
    const long u = r3d_xcd_swizzle((int)item, (int)units);  // workgroups sharing an L2 walk the units of the same clouds
    const long pt0 = u * E2_PTS;        // first point of the unit (global row)
    const long cloud0 = (pt0 / N) * N;  // first row of its cloud
    {
      const int my_idx = min(max(idx[(pt0 + w) * K + min(lane, K - 1)], 0), N - 1);  // never gather outside the cloud, whatever the list holds
      const float q = PQ[(pt0 + w) * 128 + 64 + lane];
      float pv[K];
#pragma unroll
      for (int t = 0; t < K; ++t) pv[t] = PQ[(cloud0 + __builtin_amdgcn_readlane(my_idx, t)) * 128 + lane];
      float* hrow = H + (K * w) * E2_LD + lane;
#pragma unroll
      for (int t = 0; t < K; ++t) {
        const float h = pv[t] + q;
        hrow[t * E2_LD] = h > 0.f ? h : 0.2f * h;
      }
    }
    __syncthreads();  // H complete; every wave is done scanning the previous unit's Z
    f32x4 acc[RT];
    e2_rowgemm<RT>(H, Bz, n, g, acc);
#pragma unroll
    for (int t = 0; t < RT; ++t) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = s2c * acc[t][i] + t2c;  // BN2 + LeakyReLU
        Z[(16 * t + 4 * g + i) * E2_LD + c] = v > 0.f ? v : 0.2f * v;
      }
    }
    __syncthreads();  // Z complete; H free for the next unit's gather
    {
      // max over the K rows of point w, channel lane (first maximum wins, as torch.max)
      const float* zp = Z + (K * w) * E2_LD + lane;
      float m = zp[0];
      int am = 0;
#pragma unroll
      for (int t = 1; t < K; ++t) {
        const float v = zp[t * E2_LD];
        if (v > m) { m = v; am = t; }
      }
      out[(pt0 + w) * ldo + lane] = m;
      if (argmax_out) argmax_out[(pt0 + w) * 64 + lane] = am;
    }
  }
}

template <int RT>
static int edgeconv_launch_rt(const float* PQ, const int32_t* idx, const float* W2, const float* s2, const float* t2, float* out,
                              long ldo, int N, long total_points, int32_t* argmax_out, hipStream_t st) {
  const size_t lds = sizeof(float) * ((size_t)2 * 16 * RT * E2_LD);
  static int resident = 0;  // workgroups the chip holds at once
  if (!resident) {
    resident = e2_resident_blocks(r3d_edgeconv_kernel<RT>, lds, 1024);
    R3D_REQUIRE(resident > 0, "r3d_edgeconv_fwd: cannot reserve %zu B of LDS", lds);
  }
  const long units = total_points / E2_PTS;
  const int grid = (int)(units <= resident ? units : (resident & ~7));  // several units per workgroup: keep their XCD label
  hipLaunchKernelGGL(r3d_edgeconv_kernel<RT>, dim3(grid), dim3(256), lds, st, PQ, idx, W2, s2, t2, out, ldo, N, total_points,
                     argmax_out);
  return R3D_OK;
}

// PQ: (B*N, 128) point-major [P | Q]; idx: (B, N, K) int32 neighbour ids local to the
// cloud; W2: (64, 64) [out][in]; s2/t2: (64) folded BN2; out: (B*N, ldo) point-major
// (may be a column slice of a wider buffer); argmax_out: optional (B*N, 64) int32
// position (0..K-1) of the winning neighbour, kept for the backward pass.
extern "C" int r3d_edgeconv_fwd(const float* PQ, const int32_t* idx, const float* W2, const float* s2,
                                const float* t2, float* out, long ldo, int B, int N, int K,
                                int32_t* argmax_out, void* stream) {
  R3D_REQUIRE(PQ && idx && W2 && s2 && t2 && out, "r3d_edgeconv_fwd: null pointer");
  R3D_REQUIRE(B > 0 && N > 0 && ldo >= 64, "r3d_edgeconv_fwd: bad shape");
  R3D_REQUIRE(N % E2_PTS == 0, "r3d_edgeconv_fwd: N=%d must be a multiple of %d", N, E2_PTS);
  R3D_REQUIRE(K >= 4 && K <= 32 && K % 4 == 0, "r3d_edgeconv_fwd: K=%d unsupported (need K %% 4 == 0, 4..32)", K);
  hipStream_t st = (hipStream_t)stream;
  int rc = R3D_ERR_ARG;
#define E2_CASE(RT) \
  case RT: rc = edgeconv_launch_rt<RT>(PQ, idx, W2, s2, t2, out, ldo, N, (long)B * N, argmax_out, st); break
  switch (K / 4) {
    E2_CASE(1); E2_CASE(2); E2_CASE(3); E2_CASE(4); E2_CASE(5); E2_CASE(6); E2_CASE(7); E2_CASE(8);
  }
#undef E2_CASE
  if (rc) return rc;
  R3D_LAUNCH_CHECK("r3d_edgeconv_fwd");
  return R3D_OK;
}
