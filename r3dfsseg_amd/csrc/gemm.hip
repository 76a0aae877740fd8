// Point-wise (1x1) convolution as an fp32 matrix-core GEMM with fused per-channel
// affine (+ folded BatchNorm / bias) and activation.
//
// Replaces the [Conv1d/Conv2d 1x1 -> BatchNorm -> (Leaky)ReLU] stacks of the reference:
//   models/dgcnn.py:64-80,121-122  conv1d 192->512->256
//   models/mpti.py:18-40           BaseLearner 256->128->64
//   models/attention.py:39-41      q/k/v maps 256->64 (fused into one 256->192 GEMM)
//   models/dgcnn.py:53-57          first EdgeConv conv, rewritten per point:
//        W1 [x_j - x_i ; x_i] = Wa x_j + (Wb - Wa) x_i   (see edgeconv.hip)
//
//   Out[m][j] = act( scale[j] * sum_k X[m][k] * W[j][k] + shift[j] )
// X (M, ldx) and Out (M, ldo) are point-major (one point per row); W is (Co, K)
// row-major exactly as the reference stores conv weights.  v_mfma_f32_32x32x2_f32
// keeps exact fp32 products (north_star tolerance 1e-4 rules out bf16 inputs).
//
// Tile: 64 x 64 per workgroup, 2 x 2 waves of one 32x32 accumulator, K in steps of 32
// staged through LDS with an odd row stride (conflict-free ds_read_b32 operands).
#include "common.h"

#define G_BM 64
#define G_BN 64
#define G_BK 32
#define G_LD (G_BK + 1)

enum { R3D_ACT_NONE = 0, R3D_ACT_RELU = 1, R3D_ACT_LRELU02 = 2 };

__global__ __launch_bounds__(256) void r3d_pointwise_gemm_kernel(
    const float* __restrict__ X, long ldx, const float* __restrict__ W, int M, int K, int Co,
    const float* __restrict__ scale, const float* __restrict__ shift, int act,
    float* __restrict__ Out, long ldo, int accumulate, float* __restrict__ stats_part /* [tiles_m][2][Co] or NULL */) {
  __shared__ float Xs[2][G_BM * G_LD];
  __shared__ float Ws[2][G_BN * G_LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  // Tile order: a 1-D grid, XCD-aware (common.h: r3d_xcd_swizzle), the column tiles of one row tile next to each other.
  // The workgroups that share an L2 then work on the SAME 64 rows of X, which is fetched into that L2 once and re-read
  // there by the other column tiles; with the row tile as the fast grid axis every column tile of a row block ran ~12 000
  // workgroups later and on another XCD, and X (the large operand: M = 786 432 rows for 32 episodes) came from HBM once
  // per column tile (Co = 512: 8 times).
  const int tiles_n = (Co + G_BN - 1) / G_BN;
  const int tile = r3d_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
  const int tile_m = tile / tiles_n;
  const long m0 = (long)tile_m * G_BM;
  const int n0 = (tile - tile_m * tiles_n) * G_BN;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int srow = tid >> 5, scol = tid & 31;  // 8 rows x 32 cols per pass
  // all global loads of a K-step first (unconditional, clamped addresses; common.h: r3d_keep), the LDS writes
  // afterwards, and the NEXT K-step's loads in flight behind the current step's MFMAs (two LDS buffers, one
  // barrier per step)
  float xv[G_BM / 8], wv[G_BN / 8];
  // row pointers once per workgroup: rows beyond M / Co are clamped duplicates whose products only reach output
  // elements that are never stored, so the only mask left is the K tail (address arithmetic and masks in the load
  // loop cost as much issue time as the MFMAs they feed: tools/probe/mfma_feed.hip)
  const float* xrow[G_BM / 8];
  const float* wrow[G_BN / 8];
#pragma unroll
  for (int p = 0; p < G_BM / 8; ++p) {
    xrow[p] = X + min(m0 + srow + 8 * p, (long)M - 1) * ldx;
    wrow[p] = W + (long)min(n0 + srow + 8 * p, Co - 1) * K;
  }
  auto load_step = [&](int k0) {
    const int gk = k0 + scol;
    if (k0 + G_BK <= K) {  // uniform: full K step, no masks
#pragma unroll
      for (int p = 0; p < G_BM / 8; ++p) {
        xv[p] = xrow[p][gk];
        wv[p] = wrow[p][gk];
      }
    } else {
      const int gkc = min(gk, K - 1);
#pragma unroll
      for (int p = 0; p < G_BM / 8; ++p) {
        xv[p] = r3d_keep(xrow[p][gkc], gk < K);
        wv[p] = r3d_keep(wrow[p][gkc], gk < K);
      }
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int p = 0; p < G_BM / 8; ++p) {
      const int row = srow + 8 * p;
      Xs[buf][row * G_LD + scol] = xv[p];
      Ws[buf][row * G_LD + scol] = wv[p];
    }
  };
  load_step(0);
  store_step(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < K; k0 += G_BK, buf ^= 1) {
    const bool more = k0 + G_BK < K;
    if (more) load_step(k0 + G_BK);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch loads in front of the MFMAs (the scheduler sinks them to their use)
    const float* ap = Xs[buf] + (32 * wm + (lane & 31)) * G_LD + (lane >> 5);
    const float* bp = Ws[buf] + (32 * wn + (lane & 31)) * G_LD + (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < G_BK; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], acc, 0, 0, 0);
    if (more) store_step(buf ^ 1);
    __syncthreads();
  }
  const int j = n0 + 32 * wn + (lane & 31);
  const bool jok = j < Co;
  const float sc = (scale && jok) ? scale[j] : 1.f;
  const float sh = (shift && jok) ? shift[j] : 0.f;
  float s1 = 0.f, s2 = 0.f;  // column sums of the values written (training: batch statistics of z)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long m = m0 + 32 * wm + r3d_acc_row(r, lane);
    if (m >= M || !jok) continue;
    float v = sc * acc[r] + sh;
    if (act == R3D_ACT_RELU) v = fmaxf(v, 0.f);
    else if (act == R3D_ACT_LRELU02) v = v > 0.f ? v : 0.2f * v;
    Out[m * ldo + j] = accumulate ? Out[m * ldo + j] + v : v;
    s1 += v;
    s2 += v * v;
  }
  if (stats_part) {  // uniform over the workgroup
    // rows of one column sit in lanes l and l^32 of the two wm waves: fixed-order combine, no atomics
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    __syncthreads();  // the MFMA loop is done with Xs
    float* red = Xs[0];  // [wm][wn][2][32]
    if (lane < 32) {
      red[((wm * 2 + wn) * 2 + 0) * 32 + lane] = s1;
      red[((wm * 2 + wn) * 2 + 1) * 32 + lane] = s2;
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, cc = tid & 63, wn2 = cc >> 5, l2 = cc & 31;
      const int jj = n0 + cc;
      if (jj < Co)
        stats_part[((long)tile_m * 2 + which) * Co + jj] =
            red[((0 * 2 + wn2) * 2 + which) * 32 + l2] + red[((1 * 2 + wn2) * 2 + which) * 32 + l2];
    }
  }
}

static int pointwise_launch(const float* X, long ldx, const float* W, long M, int K, int Co, const float* scale,
                            const float* shift, int act, float* Out, long ldo, int accumulate, float* stats_part,
                            void* stream);
// gemm_bx3.hip: the same GEMM on the bf16 matrix core in three-piece arithmetic
bool r3d_pointwise_bx3_ok(const float* X, long ldx, const float* W, long M, int K, int Co);
int r3d_pointwise_bx3_launch(const float* X, long ldx, const float* W, long M, int K, int Co, const float* scale,
                             const float* shift, int act, float* Out, long ldo, int accumulate, float* stats_part,
                             hipStream_t st);

extern "C" int r3d_pointwise_conv(const float* X, long ldx, const float* W, long M, int K, int Co,
                                  const float* scale, const float* shift, int act, float* Out,
                                  long ldo, void* stream) {
  return pointwise_launch(X, ldx, W, M, K, Co, scale, shift, act, Out, ldo, 0, nullptr, stream);
}

// Training forward of a conv + BatchNorm layer: Out = X W^T (raw) AND the column sums the batch statistics need,
// sums_out[0..Co) = sum_m Out[m][c], sums_out[Co..2Co) = sum_m Out[m][c]^2, produced in the GEMM epilogue (one
// partial per 64-row tile, added in ascending tile order in fp64) instead of a second pass over Out.
// ws: r3d_pointwise_conv_stats_ws_words(M, Co) floats.
extern "C" int r3d_colreduce(const float* part, int chunks, int C, float* sums_out, void* stream);
extern "C" long r3d_pointwise_conv_stats_ws_words(long M, int Co) { return (long)r3d_cdiv(M, G_BM) * 2 * Co; }
extern "C" int r3d_pointwise_conv_stats(const float* X, long ldx, const float* W, long M, int K, int Co, float* Out,
                                        long ldo, float* sums_out, float* ws, void* stream) {
  R3D_REQUIRE(sums_out && ws, "r3d_pointwise_conv_stats: null pointer");
  int rc = pointwise_launch(X, ldx, W, M, K, Co, nullptr, nullptr, R3D_ACT_NONE, Out, ldo, 0, ws, stream);
  if (rc) return rc;
  return r3d_colreduce(ws, r3d_cdiv(M, G_BM), Co, sums_out, stream);
}

// The same over TWO row segments with separate batch statistics (the support clouds and the query clouds of an episode
// go through one GEMM launch, mpti.py:434,436 keep their BatchNorm statistics apart): rows [0, M_first) -> sums_a,
// rows [M_first, M) -> sums_b.  M_first must be a multiple of the 64-row tile.
extern "C" int r3d_pointwise_conv_stats2(const float* X, long ldx, const float* W, long M, int K, int Co, float* Out,
                                         long ldo, long M_first, float* sums_a, float* sums_b, float* ws, void* stream) {
  R3D_REQUIRE(sums_a && sums_b && ws, "r3d_pointwise_conv_stats2: null pointer");
  R3D_REQUIRE(M_first > 0 && M_first < M && M_first % G_BM == 0,
              "r3d_pointwise_conv_stats2: first segment of %ld rows (of %ld) must be a positive multiple of %d", M_first, M,
              G_BM);
  int rc = pointwise_launch(X, ldx, W, M, K, Co, nullptr, nullptr, R3D_ACT_NONE, Out, ldo, 0, ws, stream);
  if (rc) return rc;
  const int tiles_a = (int)(M_first / G_BM);
  rc = r3d_colreduce(ws, tiles_a, Co, sums_a, stream);
  if (rc) return rc;
  return r3d_colreduce(ws + (long)tiles_a * 2 * Co, r3d_cdiv(M, G_BM) - tiles_a, Co, sums_b, stream);
}

// ... and over the alternating segments of a batch of episodes (common.h: r3d_segmap; rows_a, rows_b multiples of the
// 64-row tile, rows_b == 0: equal segments): ONE GEMM launch over all rows, sums_out [seg][2][Co] reduced per segment
// from the tile partials.  A tile belongs to one segment and a segment's tiles are added relative to its first one, so
// its statistics are bit for bit those of the episode running alone.
extern "C" int r3d_colreduce_seg(const float* part, int count_a, int count_b, int n_seg, int C, float* sums_out, void* stream);
extern "C" int r3d_pointwise_conv_stats_seg(const float* X, long ldx, const float* W, long M, int K, int Co, float* Out,
                                            long ldo, long rows_a, long rows_b, float* sums_out, float* ws, void* stream) {
  R3D_REQUIRE(sums_out && ws, "r3d_pointwise_conv_stats_seg: null pointer");
  const r3d_segmap sm{rows_a, rows_b};
  R3D_REQUIRE(sm.covers(M) && (sm.n_seg(M) == 1 || (rows_a % G_BM == 0 && rows_b % G_BM == 0)),
              "r3d_pointwise_conv_stats_seg: %ld rows in segments of %ld + %ld rows (multiples of %d needed)", M, rows_a, rows_b,
              G_BM);
  int rc = pointwise_launch(X, ldx, W, M, K, Co, nullptr, nullptr, R3D_ACT_NONE, Out, ldo, 0, ws, stream);
  if (rc) return rc;
  return r3d_colreduce_seg(ws, r3d_cdiv(rows_a, G_BM), r3d_cdiv(rows_b, G_BM), sm.n_seg(M), Co, sums_out, stream);
}

// Out += act(scale * X W^T + shift): gradient accumulation into a (slice of a) wider buffer
extern "C" int r3d_pointwise_conv_acc(const float* X, long ldx, const float* W, long M, int K, int Co,
                                      const float* scale, const float* shift, int act, float* Out,
                                      long ldo, void* stream) {
  return pointwise_launch(X, ldx, W, M, K, Co, scale, shift, act, Out, ldo, 1, nullptr, stream);
}

static int pointwise_launch(const float* X, long ldx, const float* W, long M, int K, int Co, const float* scale,
                            const float* shift, int act, float* Out, long ldo, int accumulate, float* stats_part,
                            void* stream) {
  R3D_REQUIRE(X && W && Out, "r3d_pointwise_conv: null pointer");
  R3D_REQUIRE(M > 0 && K > 0 && Co > 0 && ldx >= K && ldo >= Co,
              "r3d_pointwise_conv: bad shape M=%ld K=%d Co=%d ldx=%ld ldo=%ld", M, K, Co, ldx, ldo);
  R3D_REQUIRE(act >= 0 && act <= 2, "r3d_pointwise_conv: unknown activation %d", act);
  if (r3d_pointwise_bx3_ok(X, ldx, W, M, K, Co)) {  // (a function of the layer's shape only: never of the batch)
    const int rc = r3d_pointwise_bx3_launch(X, ldx, W, M, K, Co, scale, shift, act, Out, ldo, accumulate, stats_part,
                                            (hipStream_t)stream);
    if (rc > 0) return rc;
    if (rc == 0) {
      R3D_LAUNCH_CHECK("r3d_pointwise_conv");
      return R3D_OK;
    }
  }
  const long tiles = (long)r3d_cdiv(M, G_BM) * r3d_cdiv(Co, G_BN);
  R3D_REQUIRE(tiles < 0x7fffffffL, "r3d_pointwise_conv: too many tiles");
  dim3 grid((unsigned)tiles);
  hipLaunchKernelGGL(r3d_pointwise_gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, ldx, W,
                     (int)M, K, Co, scale, shift, act, Out, ldo, accumulate, stats_part);
  R3D_LAUNCH_CHECK("r3d_pointwise_conv");
  return R3D_OK;
}

// ---- layout helpers -------------------------------------------------------
// (B, C, N) channel-major (the reference's tensor layout at the forward() boundary,
// models/mpti.py:433-436) -> (B*N, ld) point-major, via a 32x32 LDS tile transpose.
__global__ void r3d_cm_to_pm_kernel(const float* __restrict__ in, int C, int N, float* __restrict__ out,
                                    long ld) {
  __shared__ float t[32][33];
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, n = n0 + tx;
    t[i][tx] = (c < C && n < N) ? in[((long)b * C + c) * N + n] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + i, c = c0 + tx;
    if (n < N && c < C) out[((long)b * N + n) * ld + c] = t[tx][i];
  }
}

__global__ void r3d_pm_to_cm_kernel(const float* __restrict__ in, long ld, int C, int N,
                                    float* __restrict__ out, long pitch) {
  __shared__ float t[32][33];
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + i, c = c0 + tx;
    t[i][tx] = (c < C && n < N) ? in[((long)b * N + n) * ld + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, n = n0 + tx;
    if (c < C && n < N) out[((long)b * C + c) * pitch + n] = t[tx][i];
  }
}

extern "C" int r3d_cm_to_pm(const float* in, int B, int C, int N, float* out, long ld, void* stream) {
  R3D_REQUIRE(in && out && B > 0 && C > 0 && N > 0 && ld >= C, "r3d_cm_to_pm: bad arguments");
  dim3 grid(r3d_cdiv(N, 32), r3d_cdiv(C, 32), B);
  hipLaunchKernelGGL(r3d_cm_to_pm_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, C, N, out, ld);
  R3D_LAUNCH_CHECK("r3d_cm_to_pm");
  return R3D_OK;
}

extern "C" int r3d_pm_to_cm(const float* in, long ld, int B, int C, int N, float* out, void* stream) {
  R3D_REQUIRE(in && out && B > 0 && C > 0 && N > 0 && ld >= C, "r3d_pm_to_cm: bad arguments");
  dim3 grid(r3d_cdiv(N, 32), r3d_cdiv(C, 32), B);
  hipLaunchKernelGGL(r3d_pm_to_cm_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, ld, C, N, out, (long)N);
  R3D_LAUNCH_CHECK("r3d_pm_to_cm");
  return R3D_OK;
}

// same with a row pitch >= N between channels (out is (B, C, pitch))
extern "C" int r3d_pm_to_cm_pitched(const float* in, long ld, int B, int C, int N, float* out, long pitch,
                                    void* stream) {
  R3D_REQUIRE(in && out && B > 0 && C > 0 && N > 0 && ld >= C && pitch >= N, "r3d_pm_to_cm_pitched: bad arguments");
  dim3 grid(r3d_cdiv(N, 32), r3d_cdiv(C, 32), B);
  hipLaunchKernelGGL(r3d_pm_to_cm_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, ld, C, N, out, pitch);
  R3D_LAUNCH_CHECK("r3d_pm_to_cm_pitched");
  return R3D_OK;
}

// strided column-block copy: dst[m][0..C) = src[m][0..C)
// (a wave takes four rows, its lanes the columns: no 64-bit division per element)
__global__ __launch_bounds__(256) void r3d_copy_cols_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst,
                                                            long ldd, long M, int C) {
  const int lane = threadIdx.x & 63;
  const long r0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  for (int c = lane; c < C; c += 64) {
    float a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = src[(r0 + u < M ? r0 + u : M - 1) * lds_ + c];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (r0 + u < M) dst[(r0 + u) * ldd + c] = a[u];
  }
}

extern "C" int r3d_copy_cols(const float* src, long ld_src, float* dst, long ld_dst, long M, int C,
                             void* stream) {
  R3D_REQUIRE(src && dst && M > 0 && C > 0, "r3d_copy_cols: bad arguments");
  hipLaunchKernelGGL(r3d_copy_cols_kernel, dim3(r3d_cdiv(M, 16)), dim3(256), 0,
                     (hipStream_t)stream, src, ld_src, dst, ld_dst, M, C);
  R3D_LAUNCH_CHECK("r3d_copy_cols");
  return R3D_OK;
}
