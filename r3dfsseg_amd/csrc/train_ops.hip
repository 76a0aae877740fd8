// Training-mode building blocks, gfx950: BatchNorm with batch statistics (forward + backward),
// the weight-gradient GEMM (A^T B with the reduction over the point axis) and small helpers.
//
// Reference semantics: torch.nn.BatchNorm{1,2}d in train mode inside models/dgcnn.py:45-80 and
// models/mpti.py:31-39 -- biased batch variance for the normalisation, unbiased for the running
// estimate, momentum 0.1, eps 1e-5 -- followed by LeakyReLU(0.2) / ReLU / nothing.
//
// A conv+BN+act layer in training is:  z = X W^T (r3d_pointwise_conv, no affine) ->
// r3d_colstats(z) -> r3d_bn_fold (batch mean / invstd -> scale, shift; running stats update) ->
// r3d_affine_act (y = act(scale z + shift)).  Backward:  r3d_bn_bwd_stats (sum du, sum du zhat) ->
// r3d_bn_bwd_apply (dz) -> r3d_pointwise_conv(dz, W^T) for dX and r3d_gemm_tn(dz, X) for dW.
// All reductions run in a fixed order (partials per row chunk, chunks added ascending, in fp64).
#include "common.h"

#define TS_ROWS 512      // rows per partial
#define TS_MAXC 1024

// ---- per-column sums over rows: partial[chunk][which][c] ------------------------------------
// which 0: sum f(x); which 1: sum g(x)
// mode 0: (x, x*x)                                             -- forward statistics
// mode 1: (du, du * zhat), du = dy * act'(s z + t), zhat = (z - mean) * invstd   -- BN backward
__global__ __launch_bounds__(256) void r3d_colpartial_kernel(
    const float* __restrict__ X, long ldx, const float* __restrict__ DY, long lddy, long M, int C, int mode,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, int act, int rows_per_chunk, float* __restrict__ part /* [chunks][2][C] */) {
  __shared__ float sa[4][64], sb[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const long r0 = (long)blockIdx.y * rows_per_chunk;
  const long r1 = min(M, r0 + rows_per_chunk);
  float a = 0.f, b = 0.f;
  if (c < C) {
    float sc = 1.f, sh = 0.f, mu = 0.f, is = 1.f;
    if (mode == 1) { sc = scale[c]; sh = shift[c]; mu = mean[c]; is = invstd[c]; }
    // 8 rows in flight per step (unconditional loads on clamped rows, masked afterwards: common.h r3d_keep); the
    // row order of the sums is unchanged
    for (long rb = r0 + w; rb < r1; rb += 32) {
      float xv[8], gv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long r = min(rb + 4 * u, r1 - 1);
        xv[u] = X[r * ldx + c];
        gv[u] = mode == 1 ? DY[r * lddy + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (rb + 4 * u < r1) {
          const float x = xv[u];
          if (mode == 0) {
            a += x;
            b += x * x;
          } else {
            const float uu = sc * x + sh;
            float g = gv[u];
            if (act == 1) g = uu > 0.f ? g : 0.f;
            else if (act == 2) g = uu > 0.f ? g : 0.2f * g;
            a += g;
            b += g * ((x - mu) * is);
          }
        }
      }
    }
  }
  sa[w][lane] = a;
  sb[w][lane] = b;
  __syncthreads();
  if (w == 0 && c < C) {
    part[((long)blockIdx.y * 2 + 0) * C + c] = ((sa[0][lane] + sa[1][lane]) + sa[2][lane]) + sa[3][lane];
    part[((long)blockIdx.y * 2 + 1) * C + c] = ((sb[0][lane] + sb[1][lane]) + sb[2][lane]) + sb[3][lane];
  }
}

// 64 columns per workgroup; the chunk axis is split over the 4 waves (chunks q, q+4, ...), each adding in
// ascending order in fp64, and the four wave totals are combined in a fixed order: deterministic for any count.
__global__ __launch_bounds__(256) void r3d_colreduce_kernel(const float* __restrict__ part, int chunks, int C,
                                                            float* __restrict__ out /* [2][C] */) {
  __shared__ double sa[4][64], sb[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  double a = 0.0, b = 0.0;
  if (c < C) {
    int k = w;
    for (; k + 12 < chunks; k += 16) {  // 4 chunks (8 loads) in flight, added in the same ascending order
      float va[4], vb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        va[u] = part[((long)(k + 4 * u) * 2 + 0) * C + c];
        vb[u] = part[((long)(k + 4 * u) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { a += (double)va[u]; b += (double)vb[u]; }
    }
    for (; k < chunks; k += 4) {
      a += (double)part[((long)k * 2 + 0) * C + c];
      b += (double)part[((long)k * 2 + 1) * C + c];
    }
  }
  sa[w][lane] = a;
  sb[w][lane] = b;
  __syncthreads();
  if (w == 0 && c < C) {
    out[c] = (float)(((sa[0][lane] + sa[1][lane]) + sa[2][lane]) + sa[3][lane]);
    out[C + c] = (float)(((sb[0][lane] + sb[1][lane]) + sb[2][lane]) + sb[3][lane]);
  }
}

// ---- batch statistics -> affine, running statistics update ------------------------------------
// sums [2][C] (sum, sumsq over `count` elements per channel).  Writes mean, invstd, scale = gamma*invstd,
// shift = beta - mean*scale; running_mean/var updated in place (momentum 0.1, unbiased variance).
__global__ void r3d_bn_fold_kernel(const float* __restrict__ sums, double count, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ rec, const int* __restrict__ rec_index,
                                   long rec_stride) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double m = (double)sums[c] / count;
  double var = (double)sums[C + c] / count - m * m;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = (float)m;
  invstd[c] = is;
  const float sc = gamma[c] * is;
  scale[c] = sc;
  shift[c] = beta[c] - (float)m * sc;
  const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
  if (rec) {  // captured episodes: the batch statistics are recorded, r3d_bn_running_update applies them in episode order
    float* r = rec + (long)(rec_index ? *rec_index : 0) * rec_stride;
    r[c] = (float)m;
    r[C + c] = (float)unb;
  } else if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

// The running-statistics updates of n_records recorded batches, applied in record order exactly as r3d_bn_fold_kernel
// (and nn.BatchNorm) would have applied them one after the other; bias (optional): a conv bias in front of the
// BatchNorm shifts the batch mean it sees by exactly that bias.
__global__ void r3d_bn_running_update_kernel(const float* __restrict__ rec, int n_records, long rec_stride, int C, float momentum,
                                             const float* __restrict__ bias, float* __restrict__ running_mean,
                                             float* __restrict__ running_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float rm = running_mean[c], rv = running_var[c];
  const float mb = bias ? momentum * bias[c] : 0.f;
  for (int k0 = 0; k0 < n_records; k0 += 8) {  // 16 loads in flight; the update order stays the record order
    float mv[8], vv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long o = (long)min(k0 + u, n_records - 1) * rec_stride;
      mv[u] = rec[o + c];
      vv[u] = rec[o + C + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (k0 + u < n_records) {
        rm = (1.f - momentum) * rm + momentum * mv[u];
        if (bias) rm = rm + mb;
        rv = (1.f - momentum) * rv + momentum * vv[u];
      }
    }
  }
  running_mean[c] = rm;
  running_var[c] = rv;
}


// ---- y = act(scale * z + shift), elementwise over (M, C) ---------------------------------------
__global__ void r3d_affine_act_kernel(const float* __restrict__ Z, long ldz, long M, int C,
                                      const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                      float* __restrict__ Y, long ldy) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * C) return;
  const long r = i / C;
  const int c = (int)(i - r * C);
  float v = scale[c] * Z[r * ldz + c] + shift[c];
  if (act == 1) v = fmaxf(v, 0.f);
  else if (act == 2) v = v > 0.f ? v : 0.2f * v;
  Y[r * ldy + c] = v;
}

// ---- BN backward, apply: dz = scale * (du - sum_du / n - zhat * sum_du_zhat / n) ----------------
__global__ void r3d_bn_bwd_apply_kernel(const float* __restrict__ Z, long ldz, const float* __restrict__ DY, long lddy,
                                        long M, int C, const float* __restrict__ scale, const float* __restrict__ shift,
                                        const float* __restrict__ mean, const float* __restrict__ invstd, int act,
                                        const float* __restrict__ sums /* [2][C]: sum du, sum du*zhat */, double count,
                                        float* __restrict__ DZ, long lddz) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * C) return;
  const long r = i / C;
  const int c = (int)(i - r * C);
  const float z = Z[r * ldz + c];
  const float u = scale[c] * z + shift[c];
  float g = DY[r * lddy + c];
  if (act == 1) g = u > 0.f ? g : 0.f;
  else if (act == 2) g = u > 0.f ? g : 0.2f * g;
  const float zh = (z - mean[c]) * invstd[c];
  const float m1 = (float)((double)sums[c] / count), m2 = (float)((double)sums[C + c] / count);
  DZ[r * lddz + c] = scale[c] * (g - m1 - zh * m2);
}

// ---- C = A^T B over the row axis: out[i][j] = sum_m A[m][i] * B[m][j] ---------------------------
// (weight gradients: A = dz (M, Ca), B = X (M, Cb) -> dW (Ca, Cb)).  64 x 64 tile per workgroup on the
// fp32 matrix core, the M axis split in chunks of TN_ROWS with per-chunk partial tiles that a second
// kernel adds in ascending chunk order (deterministic, no float atomics).
#define TN_ROWS_MAX 1024
static int tn_rows(long M, int Ca, int Cb) {
  // enough workgroups to fill the chip even for 64 x 64 outputs, chunks of at least 128 rows; never a few more than
  // the 1024 the chip holds at once (4 per CU): 24 tiles x 43 chunks = 1032 workgroups ran as two rounds
  const long tiles = (long)((Ca + 63) / 64) * ((Cb + 63) / 64);
  long chunks = 1024 / tiles;
  if (chunks < 1) chunks = 1;
  long rows = (M + chunks - 1) / chunks;
  rows = ((rows + 31) / 32) * 32;
  if (rows < 128) rows = 128;
  if (rows > TN_ROWS_MAX) rows = TN_ROWS_MAX;
  return (int)rows;
}
__global__ __launch_bounds__(256) void r3d_gemm_tn_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B,
                                                          long ldb, long M, int Ca, int Cb, int TN_ROWS,
                                                          float* __restrict__ part /* [chunks][Ca][Cb] */) {
  __shared__ float As[2][32 * 65];  // [buffer][m][i]
  __shared__ float Bs[2][32 * 65];  // [buffer][m][j]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wi = w >> 1, wj = w & 1;
  const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
  const long m_beg = (long)blockIdx.z * TN_ROWS, m_end = min(M, m_beg + TN_ROWS);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int srow = tid >> 6, scol = tid & 63;  // 4 rows x 64 cols per pass
  float av[8], bv[8];
  // columns beyond Ca / Cb are clamped duplicates that only reach unstored outputs; rows beyond m_end must be
  // zero (they enter every sum), so only the last stage of a chunk is masked
  const float* acol = A + min(i0 + scol, Ca - 1);
  const float* bcol = B + min(j0 + scol, Cb - 1);
  auto load_stage = [&](long m0) {
    if (m0 + 32 <= m_end) {  // uniform: full stage, no masks
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const long m = m0 + srow + 4 * p;
        av[p] = acol[m * lda];
        bv[p] = bcol[m * ldb];
      }
    } else {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const long m = m0 + srow + 4 * p;
        const long mc = min(m, M - 1);
        av[p] = r3d_keep(acol[mc * lda], m < m_end);
        bv[p] = r3d_keep(bcol[mc * ldb], m < m_end);
      }
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      As[buf][(srow + 4 * p) * 65 + scol] = av[p];
      Bs[buf][(srow + 4 * p) * 65 + scol] = bv[p];
    }
  };
  // software pipeline: the next 32-row stage is in flight (global -> registers) behind the MFMAs of the current
  // one and lands in the other LDS buffer: one barrier per stage
  load_stage(m_beg);
  store_stage(0);
  __syncthreads();
  int buf = 0;
  for (long m0 = m_beg; m0 < m_end; m0 += 32, buf ^= 1) {
    const bool more = m0 + 32 < m_end;
    if (more) load_stage(m0 + 32);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch loads in front of the MFMAs (the scheduler sinks them to their use)
    // MFMA A operand: A^T[i][m] -> lane (i = lane&31, k = m): As[m][32*wi + i]
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const int m = 2 * s2 + (lane >> 5);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][m * 65 + 32 * wi + (lane & 31)], Bs[buf][m * 65 + 32 * wj + (lane & 31)],
                                                 acc, 0, 0, 0);
    }
    if (more) store_stage(buf ^ 1);
    __syncthreads();
  }
  const int j = j0 + 32 * wj + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + 32 * wi + r3d_acc_row(r, lane);
    if (i < Ca && j < Cb) part[((long)blockIdx.z * Ca + i) * Cb + j] = acc[r];
  }
}

__global__ void r3d_chunk_reduce_kernel(const float* __restrict__ part, int chunks, long n, float alpha,
                                        float* __restrict__ out, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  int k = 0;
  for (; k + 8 <= chunks; k += 8) {  // 8 loads in flight; summation order stays ascending
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(long)(k + u) * n + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < chunks; ++k) s += part[(long)k * n + i];
  out[i] = accumulate ? out[i] + alpha * s : alpha * s;
}

// ---- out (M, C) (+)= in (M, C) with row strides --------------------------------------------------
__global__ void r3d_add_cols_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst, long ldd, long M,
                                    int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * C) return;
  const long r = i / C;
  const int c = (int)(i - r * C);
  dst[r * ldd + c] += src[r * lds_ + c];
}

// ===========================================================================
// C ABI
// ===========================================================================
// rows per partial: ~1024 workgroups whatever the column count (512 rows per partial left 40 workgroups for a 64-column
// matrix, each walking 16 dependent round trips), at least 64 rows, a multiple of the 32 rows of one step
static int ts_rows(long M, int C) {
  const long groups = (C + 63) / 64;
  long chunks = 1024 / groups;
  if (chunks > M / 64) chunks = M / 64;
  if (chunks < 1) chunks = 1;
  long rows = (M + chunks - 1) / chunks;
  rows = ((rows + 31) / 32) * 32;
  if (rows > TS_ROWS) rows = TS_ROWS;
  return (int)rows;
}
extern "C" long r3d_colstats_ws_words(long M, int C) {
  const int rows = ts_rows(M, C);
  return ((M + rows - 1) / rows) * 2L * C + 16;
}

// sums_out [2][C]: mode 0 (sum x, sum x^2); mode 1 (sum du, sum du*zhat) -- see kernel comment
extern "C" int r3d_colstats(const float* X, long ldx, const float* DY, long lddy, long M, int C, int mode,
                            const float* scale, const float* shift, const float* mean, const float* invstd, int act,
                            float* sums_out, float* ws, void* stream) {
  R3D_REQUIRE(X && sums_out && ws && M > 0 && C > 0 && C <= TS_MAXC, "r3d_colstats: bad arguments");
  R3D_REQUIRE(mode == 0 || (DY && scale && shift && mean && invstd), "r3d_colstats: mode 1 needs dy and the BN vectors");
  const int rows = ts_rows(M, C);
  const int chunks = r3d_cdiv(M, rows);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(r3d_colpartial_kernel, dim3(r3d_cdiv(C, 64), chunks), dim3(256), 0, st, X, ldx, DY, lddy, M, C, mode,
                     scale, shift, mean, invstd, act, rows, ws);
  hipLaunchKernelGGL(r3d_colreduce_kernel, dim3(r3d_cdiv(C, 64)), dim3(256), 0, st, ws, chunks, C, sums_out);
  R3D_LAUNCH_CHECK("r3d_colstats");
  return R3D_OK;
}

// [chunks][2][C] partial column sums -> sums_out [2][C] (shared with the GEMM-epilogue statistics of gemm.hip)
extern "C" int r3d_colreduce(const float* part, int chunks, int C, float* sums_out, void* stream) {
  R3D_REQUIRE(part && sums_out && chunks > 0 && C > 0, "r3d_colreduce: bad arguments");
  hipLaunchKernelGGL(r3d_colreduce_kernel, dim3(r3d_cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, part, chunks, C,
                     sums_out);
  R3D_LAUNCH_CHECK("r3d_colreduce");
  return R3D_OK;
}

extern "C" int r3d_bn_fold(const float* sums, double count, int C, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                           float* scale, float* shift, float* rec, const int32_t* rec_index_dev, long rec_stride,
                           void* stream) {
  R3D_REQUIRE(sums && gamma && beta && mean && invstd && scale && shift && C > 0 && count > 0, "r3d_bn_fold: bad arguments");
  R3D_REQUIRE(!rec || rec_stride >= 2L * C, "r3d_bn_fold: a record holds 2 C floats");
  hipLaunchKernelGGL(r3d_bn_fold_kernel, dim3(r3d_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, count, C, gamma,
                     beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, rec, rec_index_dev, rec_stride);
  R3D_LAUNCH_CHECK("r3d_bn_fold");
  return R3D_OK;
}

extern "C" int r3d_bn_running_update(const float* rec, int n_records, long rec_stride, int C, float momentum, const float* bias,
                                     float* running_mean, float* running_var, void* stream) {
  R3D_REQUIRE(rec && running_mean && running_var && n_records > 0 && C > 0 && rec_stride >= 2L * C,
              "r3d_bn_running_update: bad arguments");
  hipLaunchKernelGGL(r3d_bn_running_update_kernel, dim3(r3d_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, rec, n_records,
                     rec_stride, C, momentum, bias, running_mean, running_var);
  R3D_LAUNCH_CHECK("r3d_bn_running_update");
  return R3D_OK;
}

extern "C" int r3d_affine_act(const float* Z, long ldz, long M, int C, const float* scale, const float* shift, int act,
                              float* Y, long ldy, void* stream) {
  R3D_REQUIRE(Z && Y && scale && shift && M > 0 && C > 0, "r3d_affine_act: bad arguments");
  hipLaunchKernelGGL(r3d_affine_act_kernel, dim3(r3d_cdiv(M * C, 256)), dim3(256), 0, (hipStream_t)stream, Z, ldz, M, C,
                     scale, shift, act, Y, ldy);
  R3D_LAUNCH_CHECK("r3d_affine_act");
  return R3D_OK;
}

extern "C" int r3d_bn_bwd_apply(const float* Z, long ldz, const float* DY, long lddy, long M, int C, const float* scale,
                                const float* shift, const float* mean, const float* invstd, int act, const float* sums,
                                double count, float* DZ, long lddz, void* stream) {
  R3D_REQUIRE(Z && DY && DZ && scale && shift && mean && invstd && sums, "r3d_bn_bwd_apply: null pointer");
  hipLaunchKernelGGL(r3d_bn_bwd_apply_kernel, dim3(r3d_cdiv(M * C, 256)), dim3(256), 0, (hipStream_t)stream, Z, ldz, DY,
                     lddy, M, C, scale, shift, mean, invstd, act, sums, count, DZ, lddz);
  R3D_LAUNCH_CHECK("r3d_bn_bwd_apply");
  return R3D_OK;
}

extern "C" long r3d_gemm_tn_ws_words(long M, int Ca, int Cb) {
  const int rows = tn_rows(M, Ca, Cb);
  return ((M + rows - 1) / rows) * (long)Ca * Cb + 16;
}

// out (Ca, Cb) = alpha * A^T B  (+ out if accumulate)
extern "C" int r3d_gemm_tn(const float* A, long lda, const float* B, long ldb, long M, int Ca, int Cb, float alpha,
                           float* out, int accumulate, float* ws, void* stream) {
  R3D_REQUIRE(A && B && out && ws && M > 0 && Ca > 0 && Cb > 0 && lda >= Ca && ldb >= Cb, "r3d_gemm_tn: bad arguments");
  const int rows = tn_rows(M, Ca, Cb);
  const int chunks = r3d_cdiv(M, rows);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(r3d_gemm_tn_kernel, dim3(r3d_cdiv(Ca, 64), r3d_cdiv(Cb, 64), chunks), dim3(256), 0, st, A, lda, B, ldb,
                     M, Ca, Cb, rows, ws);
  hipLaunchKernelGGL(r3d_chunk_reduce_kernel, dim3(r3d_cdiv((long)Ca * Cb, 256)), dim3(256), 0, st, ws, chunks,
                     (long)Ca * Cb, alpha, out, accumulate);
  R3D_LAUNCH_CHECK("r3d_gemm_tn");
  return R3D_OK;
}

extern "C" int r3d_add_cols(const float* src, long ld_src, float* dst, long ld_dst, long M, int C, void* stream) {
  R3D_REQUIRE(src && dst && M > 0 && C > 0, "r3d_add_cols: bad arguments");
  hipLaunchKernelGGL(r3d_add_cols_kernel, dim3(r3d_cdiv(M * C, 256)), dim3(256), 0, (hipStream_t)stream, src, ld_src, dst,
                     ld_dst, M, C);
  R3D_LAUNCH_CHECK("r3d_add_cols");
  return R3D_OK;
}
