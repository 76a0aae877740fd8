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
// Work unit: 8 points = 8K edges (K % 4 == 0), one wave per 32 edges (K = 20: 5 waves).
// Each wave gathers the P rows of its own 32 edges into LDS (256-B coalesced rows),
// runs 2 x 32 MFMAs (32 edges x 64 outputs, W2 fragments read from LDS), writes
// the activated 32 x 64 block back over its LDS rows; the max over each point's K rows
// is then a plain LDS reduction.  Workgroups walk units with a grid stride.
#include "common.h"

#define EC_PTS 8
#define EC_LD 65

__global__ __launch_bounds__(512) void r3d_edgeconv_kernel(
    const float* __restrict__ PQ, const int* __restrict__ idx, const float* __restrict__ W2,
    const float* __restrict__ s2, const float* __restrict__ t2, float* __restrict__ out, long ldo,
    int N, int K, long total_points, int* __restrict__ argmax_out) {
  extern __shared__ __attribute__((aligned(16))) float H[];  // [8K][EC_LD]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nthreads = blockDim.x;
  // W2 in LDS behind the edge rows: its MFMA B fragments (B[k][j] = W2[j][k]) would cost 64 VGPRs in registers
  // and, with the 32-deep gather, one workgroup per CU; from LDS two fit and overlap gather and MFMA phases
  float* W2s = H + EC_PTS * K * EC_LD;  // [64][EC_LD]
  for (int o = tid; o < 64 * 64; o += nthreads) W2s[(o >> 6) * EC_LD + (o & 63)] = W2[o];
  __syncthreads();
  const float sc0 = s2[lane & 31], sh0 = t2[lane & 31];
  const float sc1 = s2[32 + (lane & 31)], sh1 = t2[32 + (lane & 31)];
  const long units = total_points / EC_PTS;
  for (long u = blockIdx.x; u < units; u += gridDim.x) {
    const long pt0 = u * EC_PTS;            // first point of the unit (global row)
    const long cloud0 = (pt0 / N) * N;      // first row of its cloud
    // ---- gather: this wave's 32 edges
    {
      const int e = 32 * w + (lane & 31);  // edge handled by this lane for the index load
      const int my_idx = idx[pt0 * K + e];  // (8 points x K) indices are contiguous
      float* hrow = H + (32 * w) * EC_LD;
      // all 32 neighbour rows of the wave in flight at once (the gather is L2-latency bound: 8 at a time took
      // four round trips per unit), then the adds / stores; the point's own Q row changes at most twice in 32 edges
      float pv[32];
#pragma unroll
      for (int t = 0; t < 32; ++t) pv[t] = PQ[(cloud0 + __builtin_amdgcn_readlane(my_idx, t)) * 128 + lane];
      const int p_first = (32 * w) / K, p_last = (32 * w + 31) / K;
      const float q0 = PQ[(pt0 + p_first) * 128 + 64 + lane];
      const float q1 = PQ[(pt0 + min(p_first + 1, p_last)) * 128 + 64 + lane];
      const float q2 = PQ[(pt0 + p_last) * 128 + 64 + lane];
      const int e1 = (p_first + 1) * K - 32 * w, e2 = (p_first + 2) * K - 32 * w;  // first edge of the 2nd / 3rd point
      if (K >= 16) {  // 32 consecutive edges touch at most 3 points
#pragma unroll
        for (int t = 0; t < 32; ++t) {
          const float q = t < e1 ? q0 : (t < e2 ? q1 : q2);
          float h = pv[t] + q;
          h = h > 0.f ? h : 0.2f * h;
          hrow[t * EC_LD + lane] = h;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 32; ++t) {
          float h = pv[t] + PQ[(pt0 + (32 * w + t) / K) * 128 + 64 + lane];
          h = h > 0.f ? h : 0.2f * h;
          hrow[t * EC_LD + lane] = h;
        }
      }
    }
    // ---- second layer on the matrix core (wave-local rows: no barrier needed)
    f32x16 a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    {
      const float* ap = H + (32 * w + (lane & 31)) * EC_LD + (lane >> 5);
      const float* bp0 = W2s + (lane & 31) * EC_LD + (lane >> 5);
      const float* bp1 = bp0 + 32 * EC_LD;
#pragma unroll 8
      for (int s = 0; s < 32; ++s) {
        const float a = ap[2 * s];
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp0[2 * s], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp1[2 * s], a1, 0, 0, 0);
      }
    }
    // ---- BN2 + LeakyReLU, back into this wave's LDS rows
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * w + r3d_acc_row(r, lane);
      float v0 = sc0 * a0[r] + sh0;
      v0 = v0 > 0.f ? v0 : 0.2f * v0;
      float v1 = sc1 * a1[r] + sh1;
      v1 = v1 > 0.f ? v1 : 0.2f * v1;
      H[row * EC_LD + (lane & 31)] = v0;
      H[row * EC_LD + 32 + (lane & 31)] = v1;
    }
    __syncthreads();
    // ---- max over the K rows of each point (first maximum wins, as torch.max)
    for (int o = tid; o < EC_PTS * 64; o += nthreads) {
      const int pt = o >> 6, ch = o & 63;
      const float* hp = H + (pt * K) * EC_LD + ch;
      float m = hp[0];
      int am = 0;
      for (int t = 1; t < K; ++t) {
        const float v = hp[t * EC_LD];
        if (v > m) { m = v; am = t; }
      }
      out[(pt0 + pt) * ldo + ch] = m;
      if (argmax_out) argmax_out[(pt0 + pt) * 64 + ch] = am;
    }
    __syncthreads();
  }
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
  R3D_REQUIRE(N % EC_PTS == 0, "r3d_edgeconv_fwd: N=%d must be a multiple of %d", N, EC_PTS);
  R3D_REQUIRE(K >= 4 && K <= 32 && K % 4 == 0, "r3d_edgeconv_fwd: K=%d unsupported (need K %% 4 == 0, 4..32)", K);
  const int waves = EC_PTS * K / 32;
  const size_t lds = sizeof(float) * ((size_t)EC_PTS * K * EC_LD + 64 * EC_LD);
  const long units = (long)B * N / EC_PTS;
  int grid = (int)(units < 1024 ? units : 1024);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)r3d_edgeconv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(r3d_edgeconv_kernel, dim3(grid), dim3(64 * waves), lds, (hipStream_t)stream, PQ, idx,
                     W2, s2, t2, out, ldo, N, K, (long)B * N, argmax_out);
  R3D_LAUNCH_CHECK("r3d_edgeconv_fwd");
  return R3D_OK;
}
