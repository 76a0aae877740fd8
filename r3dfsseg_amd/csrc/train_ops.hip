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
//
// SEGMENTS.  A batch of E training episodes goes through ONE launch sequence, but BatchNorm keeps the statistics of
// every getFeatures call apart (models/mpti.py:434,436: the S support clouds, then the Q query clouds of an episode).
// Rows are laid out episode after episode, [support rows | query rows] each, so the segment of a row follows from two
// numbers (common.h: r3d_segmap): rows_a = S N, rows_b = Q N; segment 2 e + p is call p of episode e -- the order in
// which the reference updates the running statistics.  Every reduction below is per segment with a partition that
// depends on the segment's size alone, so a segment's statistics are bit for bit the same whether its episode runs
// alone or inside a batch.  The single-segment entry points of ABI version 2 are the case rows_b = 0, rows_a = M.
#include "common.h"

#define TS_ROWS 512      // rows per partial
#define TS_MAXC 1024

// ---- per-column sums over rows: partial[chunk][which][c] ------------------------------------
// which 0: sum f(x); which 1: sum g(x)
// mode 0: (x, x*x)                                             -- forward statistics
// mode 1: (du, du * zhat), du = dy * act'(s z + t), zhat = (z - mean) * invstd   -- BN backward
__global__ __launch_bounds__(256) void r3d_colpartial_kernel(
    const float* __restrict__ X, long ldx, const float* __restrict__ DY, long lddy, r3d_segmap sm, int C, int mode,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, long bn_stride, int act, int rows_per_chunk_a, int rows_per_chunk_b, int cmax,
    float* __restrict__ part /* [seg][cmax][2][C] */) {
  __shared__ float sa[4][64], sb[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int seg = blockIdx.z;
  const long srows = sm.seg_rows(seg);
  const int rows_per_chunk = sm.odd(seg) ? rows_per_chunk_b : rows_per_chunk_a;
  const long s0 = (long)blockIdx.y * rows_per_chunk;
  if (s0 >= srows) return;  // uniform: this segment has fewer chunks than the longest one
  const long base = sm.seg_row0(seg);
  const long r0 = base + s0;
  const long r1 = base + min(srows, s0 + rows_per_chunk);
  float a = 0.f, b = 0.f;
  if (c < C) {
    float sc = 1.f, sh = 0.f, mu = 0.f, is = 1.f;
    if (mode == 1) {
      const long o = (long)seg * bn_stride + c;
      sc = scale[o]; sh = shift[o]; mu = mean[o]; is = invstd[o];
    }
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
    const long slot = (long)seg * cmax + blockIdx.y;
    part[(slot * 2 + 0) * C + c] = ((sa[0][lane] + sa[1][lane]) + sa[2][lane]) + sa[3][lane];
    part[(slot * 2 + 1) * C + c] = ((sb[0][lane] + sb[1][lane]) + sb[2][lane]) + sb[3][lane];
  }
}

// The same sums for C a multiple of 256 and 16-byte aligned rows: 256 columns per workgroup, a lane owns FOUR consecutive
// columns (a wave reads 1 KB of a row per instruction instead of 256 bytes).  Which wave adds which rows of a column, in
// which order, and how the four wave totals meet is unchanged: the same bits.
__global__ __launch_bounds__(256) void r3d_colpartial_v4_kernel(
    const float* __restrict__ X, long ldx, const float* __restrict__ DY, long lddy, r3d_segmap sm, int C, int mode,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, long bn_stride, int act, int rows_per_chunk_a, int rows_per_chunk_b, int cmax,
    float* __restrict__ part /* [seg][cmax][2][C] */) {
  __shared__ float4 sa[4][64], sb[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + 4 * lane;
  const int seg = blockIdx.z;
  const long srows = sm.seg_rows(seg);
  const int rows_per_chunk = sm.odd(seg) ? rows_per_chunk_b : rows_per_chunk_a;
  const long s0 = (long)blockIdx.y * rows_per_chunk;
  if (s0 >= srows) return;  // uniform: this segment has fewer chunks than the longest one
  const long base = sm.seg_row0(seg);
  const long r0 = base + s0;
  const long r1 = base + min(srows, s0 + rows_per_chunk);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f), mu = sh, is = sc;
  if (mode == 1) {
    const long o = (long)seg * bn_stride + c;
    sc = *reinterpret_cast<const float4*>(scale + o); sh = *reinterpret_cast<const float4*>(shift + o);
    mu = *reinterpret_cast<const float4*>(mean + o); is = *reinterpret_cast<const float4*>(invstd + o);
  }
  auto acc = [&](float x, float g0, float sc_, float sh_, float mu_, float is_, float& a_, float& b_) {
    if (mode == 0) {
      a_ += x;
      b_ += x * x;
    } else {
      const float uu = sc_ * x + sh_;
      float g = g0;
      if (act == 1) g = uu > 0.f ? g : 0.f;
      else if (act == 2) g = uu > 0.f ? g : 0.2f * g;
      a_ += g;
      b_ += g * ((x - mu_) * is_);
    }
  };
  for (long rb = r0 + w; rb < r1; rb += 32) {
    float4 xv[8], gv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long r = min(rb + 4 * u, r1 - 1);
      xv[u] = *reinterpret_cast<const float4*>(X + r * ldx + c);
      gv[u] = mode == 1 ? *reinterpret_cast<const float4*>(DY + r * lddy + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (rb + 4 * u < r1) {
        acc(xv[u].x, gv[u].x, sc.x, sh.x, mu.x, is.x, a.x, b.x);
        acc(xv[u].y, gv[u].y, sc.y, sh.y, mu.y, is.y, a.y, b.y);
        acc(xv[u].z, gv[u].z, sc.z, sh.z, mu.z, is.z, a.z, b.z);
        acc(xv[u].w, gv[u].w, sc.w, sh.w, mu.w, is.w, a.w, b.w);
      }
    }
  }
  sa[w][lane] = a;
  sb[w][lane] = b;
  __syncthreads();
  if (w == 0) {
    const long slot = (long)seg * cmax + blockIdx.y;
    float4 ta, tb;
    ta.x = ((sa[0][lane].x + sa[1][lane].x) + sa[2][lane].x) + sa[3][lane].x; ta.y = ((sa[0][lane].y + sa[1][lane].y) + sa[2][lane].y) + sa[3][lane].y;
    ta.z = ((sa[0][lane].z + sa[1][lane].z) + sa[2][lane].z) + sa[3][lane].z; ta.w = ((sa[0][lane].w + sa[1][lane].w) + sa[2][lane].w) + sa[3][lane].w;
    tb.x = ((sb[0][lane].x + sb[1][lane].x) + sb[2][lane].x) + sb[3][lane].x; tb.y = ((sb[0][lane].y + sb[1][lane].y) + sb[2][lane].y) + sb[3][lane].y;
    tb.z = ((sb[0][lane].z + sb[1][lane].z) + sb[2][lane].z) + sb[3][lane].z; tb.w = ((sb[0][lane].w + sb[1][lane].w) + sb[2][lane].w) + sb[3][lane].w;
    *reinterpret_cast<float4*>(part + (slot * 2 + 0) * C + c) = ta;
    *reinterpret_cast<float4*>(part + (slot * 2 + 1) * C + c) = tb;
  }
}

// 64 columns per workgroup; the chunk axis is split over the 4 waves (chunks q, q+4, ...), each adding in
// ascending order in fp64, and the four wave totals are combined in a fixed order: deterministic for any count.
// blockIdx.y = segment.  Its chunks are rows [first, first + count) of `part`: cmax > 0: first = seg * cmax (the
// layout r3d_colpartial_kernel writes); cmax == 0: the chunks of all segments follow each other (one partial per
// 64-row tile of a GEMM, gemm.hip), count_a / count_b of them alternating.  Chunk indices are relative to the
// segment's first chunk, so a segment's sums do not depend on where it sits in the batch.
__global__ __launch_bounds__(256) void r3d_colreduce_kernel(const float* __restrict__ part, int count_a, int count_b,
                                                            int cmax, int C, float* __restrict__ out /* [seg][2][C] */) {
  __shared__ double sa[4][64], sb[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int seg = blockIdx.y;
  const bool odd = count_b > 0 && (seg & 1);
  const int chunks = odd ? count_b : count_a;
  const long first = cmax > 0 ? (long)seg * cmax
                              : (count_b > 0 ? (long)(seg >> 1) * (count_a + count_b) + (odd ? count_a : 0) : (long)seg * count_a);
  part += first * 2 * C;
  out += (long)seg * 2 * C;
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
// One thread per channel walks the segments IN ORDER (segment 2 e + p = call p of episode e): the running statistics
// see the batches one after the other exactly as the reference's one-episode-at-a-time schedule applies them.
__global__ void r3d_bn_fold_kernel(const float* __restrict__ sums /* [seg][2][C] */, int n_seg, double count_a,
                                   double count_b, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                   float* __restrict__ shift, long bn_stride, float* __restrict__ rec,
                                   const int* __restrict__ rec_index, long rec_stride) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float g = gamma[c], bt = beta[c];
  float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
  const long rec0 = rec_index ? *rec_index : 0;
  for (int s0 = 0; s0 < n_seg; s0 += 8) {  // 16 loads in flight; the update order stays the segment order
    float sv[8], qv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long o = (long)min(s0 + u, n_seg - 1) * 2 * C;
      sv[u] = sums[o + c];
      qv[u] = sums[o + C + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int seg = s0 + u;
      if (seg >= n_seg) break;
      const double count = (count_b > 0.0 && (seg & 1)) ? count_b : count_a;
      const double m = (double)sv[u] / count;
      double var = (double)qv[u] / count - m * m;
      if (var < 0.0) var = 0.0;
      const float is = (float)(1.0 / sqrt(var + (double)eps));
      const long o = (long)seg * bn_stride + c;
      mean[o] = (float)m;
      invstd[o] = is;
      const float sc = g * is;
      scale[o] = sc;
      shift[o] = bt - (float)m * sc;
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      if (rec) {  // captured episodes: the batch statistics are recorded, r3d_bn_running_update applies them in episode order
        float* r = rec + (rec0 + seg) * rec_stride;
        r[c] = (float)m;
        r[C + c] = (float)unb;
      } else if (running_mean) {
        rm = (1.f - momentum) * rm + momentum * (float)m;
        rv = (1.f - momentum) * rv + momentum * (float)unb;
      }
    }
  }
  if (!rec && running_mean) {
    running_mean[c] = rm;
    running_var[c] = rv;
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


// ---- y = act(scale * z + shift), elementwise over (M, C); scale / shift of the row's segment -------
// One workgroup = EW_ROWS rows, wave w takes rows w, w + 4, ... with the lanes along the columns: no division per
// element (a 64-bit i / C and a segment lookup per element are ~250 VALU instructions against 8 bytes of traffic: the
// kernel was instruction bound), the row's segment once per row, eight rows of loads in flight per wave.
#define EW_ROWS 32
__global__ __launch_bounds__(256) void r3d_affine_act_kernel(const float* __restrict__ Z, long ldz, int M, int C, r3d_segmap sm,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             long bn_stride, int act, float* __restrict__ Y, long ldy) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r0 = blockIdx.x * EW_ROWS + w;
  long zo[EW_ROWS / 4], yo[EW_ROWS / 4], bo[EW_ROWS / 4];
#pragma unroll
  for (int u = 0; u < EW_ROWS / 4; ++u) {
    const int r = min(r0 + 4 * u, M - 1);
    zo[u] = (long)r * ldz; yo[u] = (long)r * ldy;
    bo[u] = (long)sm.seg_of_row32(r) * bn_stride;
  }
  for (int c = lane; c < C; c += 64) {
    float zv[EW_ROWS / 4], sc[EW_ROWS / 4], sh[EW_ROWS / 4];
#pragma unroll
    for (int u = 0; u < EW_ROWS / 4; ++u) { zv[u] = Z[zo[u] + c]; sc[u] = scale[bo[u] + c]; sh[u] = shift[bo[u] + c]; }
#pragma unroll
    for (int u = 0; u < EW_ROWS / 4; ++u) {
      float v = sc[u] * zv[u] + sh[u];
      if (act == 1) v = fmaxf(v, 0.f);
      else if (act == 2) v = v > 0.f ? v : 0.2f * v;
      if (r0 + 4 * u < M) Y[yo[u] + c] = v;
    }
  }
}

// ---- BN backward, apply: dz = scale * (du - sum_du / n - zhat * sum_du_zhat / n), per segment ----
__global__ __launch_bounds__(256) void r3d_bn_bwd_apply_kernel(
    const float* __restrict__ Z, long ldz, const float* __restrict__ DY, long lddy, int M, int C, r3d_segmap sm,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, long bn_stride, int act, const float* __restrict__ sums /* [seg][2][C]: sum du, sum du*zhat */,
    double count_a, double count_b, float* __restrict__ DZ, long lddz) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r0 = blockIdx.x * EW_ROWS + w;
  constexpr int U = 4;  // rows in flight per trip (each brings two row loads and six vector loads)
  for (int rr = 0; rr < EW_ROWS / 4; rr += U) {
    long zo[U], go[U], oo[U], bo[U], so[U];
    double cnt[U];
    bool same = true;  // the usual case: the trip's rows lie in one segment -> one pair of fp64 divisions per column
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = min(r0 + 4 * (rr + u), M - 1);
      const int seg = sm.seg_of_row32(r);
      zo[u] = (long)r * ldz; go[u] = (long)r * lddy; oo[u] = (long)r * lddz;
      bo[u] = (long)seg * bn_stride; so[u] = (long)seg * 2 * C;
      cnt[u] = sm.odd(seg) ? count_b : count_a;
      same = same && so[u] == so[0];
    }
    for (int c = lane; c < C; c += 64) {
      float z[U], g[U], sc[U], sh[U], mu[U], is[U], m1[U], m2[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        z[u] = Z[zo[u] + c]; g[u] = DY[go[u] + c];
        sc[u] = scale[bo[u] + c]; sh[u] = shift[bo[u] + c]; mu[u] = mean[bo[u] + c]; is[u] = invstd[bo[u] + c];
      }
      if (same) {  // wave-uniform
        const float a1 = (float)((double)sums[so[0] + c] / cnt[0]), a2 = (float)((double)sums[so[0] + C + c] / cnt[0]);
#pragma unroll
        for (int u = 0; u < U; ++u) { m1[u] = a1; m2[u] = a2; }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          m1[u] = (float)((double)sums[so[u] + c] / cnt[u]);
          m2[u] = (float)((double)sums[so[u] + C + c] / cnt[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float uu = sc[u] * z[u] + sh[u];
        float gg = g[u];
        if (act == 1) gg = uu > 0.f ? gg : 0.f;
        else if (act == 2) gg = uu > 0.f ? gg : 0.2f * gg;
        const float zh = (z[u] - mu[u]) * is[u];
        if (r0 + 4 * (rr + u) < M) DZ[oo[u] + c] = sc[u] * (gg - m1[u] - zh * m2[u]);
      }
    }
  }
}

// The same for C a power of two (16 .. 1024) and 16-byte aligned rows: a lane owns FOUR consecutive columns (16-byte loads
// and stores: a wave moves 1 KB per instruction, 64 / (C / 4) rows of it when C < 256), its columns' six constants --
// scale, shift, mean, 1 / std and the two means, the latter the same fp64 quotients as above -- are formed ONCE per
// workgroup and segment instead of once per four rows, and the segment is looked up once per workgroup when its rows
// lie in one (otherwise per row).  Every element goes through the same expression: same bits.  (The kernel above moved
// 3.6 TB/s at workload S -- 26 loads and two fp64 divisions per lane for four elements --, the other element-wise
// passes 4.6-5.7.)
#define BA_ROWS 64
template <int NK /* column groups of 256 per lane: C / 256, at least 1 */>
__global__ __launch_bounds__(256) void r3d_bn_bwd_apply_v4_kernel(
    const float* __restrict__ Z, long ldz, const float* __restrict__ DY, long lddy, int M, int C, r3d_segmap sm,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, long bn_stride, int act, const float* __restrict__ sums, double count_a, double count_b,
    float* __restrict__ DZ, long lddz) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lpr = NK > 1 ? 64 : C >> 2;   // lanes of a row (C >= 256: the whole wave)
  const int rpw = 64 / lpr;               // rows per wave and instruction
  const int col0 = 4 * (lane & (lpr - 1));
  const int rsub = lane / lpr;            // (lpr is a power of two: a shift)
  const int rb0 = blockIdx.x * BA_ROWS;
  const int rb1 = min(rb0 + BA_ROWS, M) - 1;
  const int seg_first = sm.seg_of_row32(rb0), seg_last = sm.seg_of_row32(rb1);
  float4 sc[NK], sh[NK], mu[NK], is[NK], m1[NK], m2[NK];
  int cur = -1;
  auto load_consts = [&](int seg) {
    const long bo = (long)seg * bn_stride, so = (long)seg * 2 * C;
    const double cnt = sm.odd(seg) ? count_b : count_a;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      const int c = col0 + 256 * kk;
      sc[kk] = *reinterpret_cast<const float4*>(scale + bo + c); sh[kk] = *reinterpret_cast<const float4*>(shift + bo + c);
      mu[kk] = *reinterpret_cast<const float4*>(mean + bo + c); is[kk] = *reinterpret_cast<const float4*>(invstd + bo + c);
      const float4 s1 = *reinterpret_cast<const float4*>(sums + so + c), s2 = *reinterpret_cast<const float4*>(sums + so + C + c);
      m1[kk] = make_float4((float)((double)s1.x / cnt), (float)((double)s1.y / cnt), (float)((double)s1.z / cnt), (float)((double)s1.w / cnt));
      m2[kk] = make_float4((float)((double)s2.x / cnt), (float)((double)s2.y / cnt), (float)((double)s2.z / cnt), (float)((double)s2.w / cnt));
    }
    cur = seg;
  };
  load_consts(seg_first);
  const bool one_seg = seg_first == seg_last;  // (uniform over the workgroup)
  auto elem = [&](float z, float g, float sc_, float sh_, float mu_, float is_, float m1_, float m2_) {
    const float uu = sc_ * z + sh_;
    float gg = g;
    if (act == 1) gg = uu > 0.f ? gg : 0.f;
    else if (act == 2) gg = uu > 0.f ? gg : 0.2f * gg;
    const float zh = (z - mu_) * is_;
    return sc_ * (gg - m1_ - zh * m2_);
  };
  constexpr int U = NK > 1 ? 2 : 4;  // row groups in flight per trip
  const int step = 4 * rpw;          // rows the workgroup's four waves cover per row group
  for (int it = 0; it < BA_ROWS; it += U * step) {
    float4 z[U][NK], g[U][NK];
    int row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      row[u] = rb0 + it + u * step + rpw * w + rsub;
      const int r = min(row[u], M - 1);
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        z[u][kk] = *reinterpret_cast<const float4*>(Z + (long)r * ldz + col0 + 256 * kk);
        g[u][kk] = *reinterpret_cast<const float4*>(DY + (long)r * lddy + col0 + 256 * kk);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!one_seg) {
        const int seg = sm.seg_of_row32(min(row[u], M - 1));
        if (seg != cur) load_consts(seg);
      }
      if (row[u] < M && row[u] <= rb1) {
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          float4 o;
          o.x = elem(z[u][kk].x, g[u][kk].x, sc[kk].x, sh[kk].x, mu[kk].x, is[kk].x, m1[kk].x, m2[kk].x);
          o.y = elem(z[u][kk].y, g[u][kk].y, sc[kk].y, sh[kk].y, mu[kk].y, is[kk].y, m1[kk].y, m2[kk].y);
          o.z = elem(z[u][kk].z, g[u][kk].z, sc[kk].z, sh[kk].z, mu[kk].z, is[kk].z, m1[kk].z, m2[kk].z);
          o.w = elem(z[u][kk].w, g[u][kk].w, sc[kk].w, sh[kk].w, mu[kk].w, is[kk].w, m1[kk].w, m2[kk].w);
          *reinterpret_cast<float4*>(DZ + (long)row[u] * lddz + col0 + 256 * kk) = o;
        }
      }
    }
  }
}

// ---- C = A^T B over the row axis: out[i][j] = sum_m A[m][i] * B[m][j] ---------------------------
// (weight gradients: A = dz (M, Ca), B = X (M, Cb) -> dW (Ca, Cb)).  64 x 64 tile per workgroup on the
// fp32 matrix core, the M axis split in chunks of TN_ROWS with per-chunk partial tiles that a second
// kernel adds in ascending chunk order (deterministic, no float atomics).
#define TN_ROWS_MAX 4096  // (1024 until round 4: 768 partial tiles per weight gradient at 786 432 rows, whose ascending sum --
                          // r3d_chunk_reduce_kernel, one thread per output walking its chunks -- took 48 us per GEMM; 19 us now)
// gemm_bx3.hip: the same product on the bf16 matrix core in three-piece arithmetic (128 x 128 / 256 x 64 tiles, 2 per CU)
bool r3d_gemm_tn_bx3_ok(int Ca, int Cb);
int r3d_gemm_tn_bx3_tiles(int Ca, int Cb);
int r3d_gemm_tn_bx3_launch(const float* A, long lda, const float* B, long ldb, long M, int Ca, int Cb, int rows, int chunks,
                           float* part, hipStream_t st);
static int tn_rows(long M, int Ca, int Cb, bool bx3) {
  // enough workgroups to fill the chip even for 64 x 64 outputs, chunks of at least 128 rows; never a few more than
  // the 1024 (512) the chip holds at once (4 (2) per CU): 24 tiles x 43 chunks = 1032 workgroups ran as two rounds
  const long tiles = bx3 ? r3d_gemm_tn_bx3_tiles(Ca, Cb) : (long)((Ca + 63) / 64) * ((Cb + 63) / 64);
  long chunks = (bx3 ? 512 : 1024) / tiles;
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
  // 1-D grid in XCD-aware order (common.h: r3d_xcd_swizzle): the output tiles of one row chunk sit next to each other, so
  // the workgroups sharing an L2 read the same rows of A and B
  const int nti = (Ca + 63) / 64, ntj = (Cb + 63) / 64;
  const int tile = r3d_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
  const int chunk = tile / (nti * ntj), tij = tile - chunk * (nti * ntj);
  const int i0 = (tij % nti) * 64, j0 = (tij / nti) * 64;
  const long m_beg = (long)chunk * TN_ROWS, m_end = min(M, m_beg + TN_ROWS);
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
    if (i < Ca && j < Cb) part[((long)chunk * Ca + i) * Cb + j] = acc[r];
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
// (a wave takes four rows, its lanes the columns: no 64-bit division per element -- ~100 instructions for 8 bytes of traffic)
__global__ __launch_bounds__(256) void r3d_add_cols_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst,
                                                           long ldd, long M, int C) {
  const int lane = threadIdx.x & 63;
  const long r0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  for (int c = lane; c < C; c += 64) {
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long r = r0 + u < M ? r0 + u : M - 1;
      a[u] = src[r * lds_ + c]; b[u] = dst[r * ldd + c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (r0 + u < M) dst[(r0 + u) * ldd + c] = b[u] + a[u];
  }
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
static int ts_chunks(long rows, int C) { return rows > 0 ? r3d_cdiv(rows, ts_rows(rows, C)) : 0; }
extern "C" long r3d_colstats_seg_ws_words(long M, int C, long rows_a, long rows_b) {
  const r3d_segmap sm{rows_a, rows_b};
  if (!sm.covers(M)) return 0;
  const int ca = ts_chunks(rows_a, C), cb = ts_chunks(rows_b, C);
  return (long)sm.n_seg(M) * (ca > cb ? ca : cb) * 2L * C + 16;
}
extern "C" long r3d_colstats_ws_words(long M, int C) { return r3d_colstats_seg_ws_words(M, C, M, 0); }

// sums_out [seg][2][C]: mode 0 (sum x, sum x^2); mode 1 (sum du, sum du*zhat) -- see kernel comment.  The BatchNorm
// vectors of segment s are read at scale + s * bn_stride, ...
extern "C" int r3d_colstats_seg(const float* X, long ldx, const float* DY, long lddy, long M, int C, long rows_a,
                                long rows_b, int mode, const float* scale, const float* shift, const float* mean,
                                const float* invstd, long bn_stride, int act, float* sums_out, float* ws, void* stream) {
  R3D_REQUIRE(X && sums_out && ws && M > 0 && C > 0 && C <= TS_MAXC, "r3d_colstats: bad arguments");
  R3D_REQUIRE(mode == 0 || (DY && scale && shift && mean && invstd), "r3d_colstats: mode 1 needs dy and the BN vectors");
  const r3d_segmap sm{rows_a, rows_b};
  R3D_REQUIRE(sm.covers(M), "r3d_colstats: %ld rows are not whole segments of %ld + %ld rows", M, rows_a, rows_b);
  const int ra = ts_rows(rows_a, C), rb = rows_b > 0 ? ts_rows(rows_b, C) : ra;
  const int ca = ts_chunks(rows_a, C), cb = ts_chunks(rows_b, C);
  const int cmax = ca > cb ? ca : cb, n_seg = sm.n_seg(M);
  R3D_REQUIRE(n_seg <= 65535 && cmax <= 65535, "r3d_colstats: too many segments");
  hipStream_t st = (hipStream_t)stream;
  const bool v4 = C % 256 == 0 && ((ldx | (DY ? lddy : 0) | bn_stride) & 3) == 0 &&
                  (((uintptr_t)X | (uintptr_t)DY | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)mean | (uintptr_t)invstd |
                    (uintptr_t)ws) & 15) == 0;
  if (v4)
    hipLaunchKernelGGL(r3d_colpartial_v4_kernel, dim3(C / 256, cmax, n_seg), dim3(256), 0, st, X, ldx, DY, lddy, sm, C, mode,
                       scale, shift, mean, invstd, bn_stride, act, ra, rb, cmax, ws);
  else
  hipLaunchKernelGGL(r3d_colpartial_kernel, dim3(r3d_cdiv(C, 64), cmax, n_seg), dim3(256), 0, st, X, ldx, DY, lddy, sm, C, mode,
                     scale, shift, mean, invstd, bn_stride, act, ra, rb, cmax, ws);
  hipLaunchKernelGGL(r3d_colreduce_kernel, dim3(r3d_cdiv(C, 64), n_seg), dim3(256), 0, st, ws, ca, cb, cmax, C, sums_out);
  R3D_LAUNCH_CHECK("r3d_colstats");
  return R3D_OK;
}
extern "C" int r3d_colstats(const float* X, long ldx, const float* DY, long lddy, long M, int C, int mode,
                            const float* scale, const float* shift, const float* mean, const float* invstd, int act,
                            float* sums_out, float* ws, void* stream) {
  return r3d_colstats_seg(X, ldx, DY, lddy, M, C, M, 0, mode, scale, shift, mean, invstd, 0, act, sums_out, ws, stream);
}

// [chunks][2][C] partial column sums -> sums_out [seg][2][C] (shared with the GEMM-epilogue statistics of gemm.hip): the
// chunks of the segments follow each other, count_a / count_b of them alternating (count_b == 0: count_a each)
extern "C" int r3d_colreduce_seg(const float* part, int count_a, int count_b, int n_seg, int C, float* sums_out, void* stream) {
  R3D_REQUIRE(part && sums_out && count_a > 0 && count_b >= 0 && n_seg > 0 && n_seg <= 65535 && C > 0,
              "r3d_colreduce: bad arguments");
  hipLaunchKernelGGL(r3d_colreduce_kernel, dim3(r3d_cdiv(C, 64), n_seg), dim3(256), 0, (hipStream_t)stream, part, count_a, count_b,
                     0, C, sums_out);
  R3D_LAUNCH_CHECK("r3d_colreduce");
  return R3D_OK;
}
extern "C" int r3d_colreduce(const float* part, int chunks, int C, float* sums_out, void* stream) {
  return r3d_colreduce_seg(part, chunks, 0, 1, C, sums_out, stream);
}

// sums [seg][2][C] -> mean / invstd / scale / shift of segment s at (pointer + s * bn_stride).  Running statistics (or
// the records, rec + (*rec_index_dev + s) * rec_stride) are updated in segment order.
extern "C" int r3d_bn_fold_seg(const float* sums, int n_seg, double count_a, double count_b, int C, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                               float* mean, float* invstd, float* scale, float* shift, long bn_stride, float* rec,
                               const int32_t* rec_index_dev, long rec_stride, void* stream) {
  R3D_REQUIRE(sums && gamma && beta && mean && invstd && scale && shift && C > 0 && count_a > 0 && count_b >= 0 && n_seg > 0,
              "r3d_bn_fold: bad arguments");
  R3D_REQUIRE(n_seg == 1 || bn_stride >= C, "r3d_bn_fold: %d segments need a vector stride of at least C", n_seg);
  R3D_REQUIRE(!rec || rec_stride >= 2L * C, "r3d_bn_fold: a record holds 2 C floats");
  hipLaunchKernelGGL(r3d_bn_fold_kernel, dim3(r3d_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, sums, n_seg, count_a, count_b,
                     C, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, bn_stride, rec,
                     rec_index_dev, rec_stride);
  R3D_LAUNCH_CHECK("r3d_bn_fold");
  return R3D_OK;
}
extern "C" int r3d_bn_fold(const float* sums, double count, int C, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                           float* scale, float* shift, float* rec, const int32_t* rec_index_dev, long rec_stride,
                           void* stream) {
  return r3d_bn_fold_seg(sums, 1, count, 0.0, C, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale,
                         shift, 0, rec, rec_index_dev, rec_stride, stream);
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

extern "C" int r3d_affine_act_seg(const float* Z, long ldz, long M, int C, long rows_a, long rows_b, const float* scale,
                                  const float* shift, long bn_stride, int act, float* Y, long ldy, void* stream) {
  R3D_REQUIRE(Z && Y && scale && shift && M > 0 && C > 0, "r3d_affine_act: bad arguments");
  const r3d_segmap sm{rows_a, rows_b};
  R3D_REQUIRE(sm.covers(M), "r3d_affine_act: %ld rows are not whole segments of %ld + %ld rows", M, rows_a, rows_b);
  hipLaunchKernelGGL(r3d_affine_act_kernel, dim3(r3d_cdiv(M, EW_ROWS)), dim3(256), 0, (hipStream_t)stream, Z, ldz, (int)M, C, sm,
                     scale, shift, bn_stride, act, Y, ldy);
  R3D_LAUNCH_CHECK("r3d_affine_act");
  return R3D_OK;
}
extern "C" int r3d_affine_act(const float* Z, long ldz, long M, int C, const float* scale, const float* shift, int act,
                              float* Y, long ldy, void* stream) {
  return r3d_affine_act_seg(Z, ldz, M, C, M, 0, scale, shift, 0, act, Y, ldy, stream);
}

extern "C" int r3d_bn_bwd_apply_seg(const float* Z, long ldz, const float* DY, long lddy, long M, int C, long rows_a,
                                    long rows_b, const float* scale, const float* shift, const float* mean,
                                    const float* invstd, long bn_stride, int act, const float* sums, double count_a,
                                    double count_b, float* DZ, long lddz, void* stream) {
  R3D_REQUIRE(Z && DY && DZ && scale && shift && mean && invstd && sums, "r3d_bn_bwd_apply: null pointer");
  const r3d_segmap sm{rows_a, rows_b};
  R3D_REQUIRE(sm.covers(M) && count_a > 0, "r3d_bn_bwd_apply: %ld rows are not whole segments of %ld + %ld rows", M, rows_a,
              rows_b);
  const bool pow2 = C >= 16 && C <= 1024 && (C & (C - 1)) == 0;
  const bool al16 = ((ldz | lddy | lddz | bn_stride) & 3) == 0 &&
                    (((uintptr_t)Z | (uintptr_t)DY | (uintptr_t)DZ | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)mean |
                      (uintptr_t)invstd | (uintptr_t)sums) & 15) == 0;
  if (pow2 && al16) {
#define BA_GO(NK)                                                                                                             \
  hipLaunchKernelGGL(r3d_bn_bwd_apply_v4_kernel<NK>, dim3(r3d_cdiv(M, BA_ROWS)), dim3(256), 0, (hipStream_t)stream, Z, ldz, DY, \
                     lddy, (int)M, C, sm, scale, shift, mean, invstd, bn_stride, act, sums, count_a, count_b, DZ, lddz)
    if (C <= 256) BA_GO(1);
    else if (C == 512) BA_GO(2);
    else BA_GO(4);
#undef BA_GO
  } else
  hipLaunchKernelGGL(r3d_bn_bwd_apply_kernel, dim3(r3d_cdiv(M, EW_ROWS)), dim3(256), 0, (hipStream_t)stream, Z, ldz, DY,
                     lddy, (int)M, C, sm, scale, shift, mean, invstd, bn_stride, act, sums, count_a, count_b, DZ, lddz);
  R3D_LAUNCH_CHECK("r3d_bn_bwd_apply");
  return R3D_OK;
}
extern "C" int r3d_bn_bwd_apply(const float* Z, long ldz, const float* DY, long lddy, long M, int C, const float* scale,
                                const float* shift, const float* mean, const float* invstd, int act, const float* sums,
                                double count, float* DZ, long lddz, void* stream) {
  return r3d_bn_bwd_apply_seg(Z, ldz, DY, lddy, M, C, M, 0, scale, shift, mean, invstd, 0, act, sums, count, 0.0, DZ, lddz,
                              stream);
}

extern "C" long r3d_gemm_tn_ws_words(long M, int Ca, int Cb) {
  const int r0 = tn_rows(M, Ca, Cb, false), r1 = tn_rows(M, Ca, Cb, true);  // (whichever arithmetic is selected later)
  const int rows = r0 < r1 ? r0 : r1;
  return ((M + rows - 1) / rows) * (long)Ca * Cb + 16;
}

// out (Ca, Cb) = alpha * A^T B  (+ out if accumulate)
extern "C" int r3d_gemm_tn(const float* A, long lda, const float* B, long ldb, long M, int Ca, int Cb, float alpha,
                           float* out, int accumulate, float* ws, void* stream) {
  R3D_REQUIRE(A && B && out && ws && M > 0 && Ca > 0 && Cb > 0 && lda >= Ca && ldb >= Cb, "r3d_gemm_tn: bad arguments");
  const bool bx3 = r3d_gemm_tn_bx3_ok(Ca, Cb);
  const int rows = tn_rows(M, Ca, Cb, bx3);
  const int chunks = r3d_cdiv(M, rows);
  hipStream_t st = (hipStream_t)stream;
  if (bx3) {
    const int rc = r3d_gemm_tn_bx3_launch(A, lda, B, ldb, M, Ca, Cb, rows, chunks, ws, st);
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL(r3d_gemm_tn_kernel, dim3(r3d_cdiv(Ca, 64) * r3d_cdiv(Cb, 64) * chunks), dim3(256), 0, st, A, lda, B,
                       ldb, M, Ca, Cb, rows, ws);
  }
  hipLaunchKernelGGL(r3d_chunk_reduce_kernel, dim3(r3d_cdiv((long)Ca * Cb, 256)), dim3(256), 0, st, ws, chunks,
                     (long)Ca * Cb, alpha, out, accumulate);
  R3D_LAUNCH_CHECK("r3d_gemm_tn");
  return R3D_OK;
}

extern "C" int r3d_add_cols(const float* src, long ld_src, float* dst, long ld_dst, long M, int C, void* stream) {
  R3D_REQUIRE(src && dst && M > 0 && C > 0, "r3d_add_cols: bad arguments");
  hipLaunchKernelGGL(r3d_add_cols_kernel, dim3(r3d_cdiv(M, 16)), dim3(256), 0, (hipStream_t)stream, src, ld_src, dst,
                     ld_dst, M, C);
  R3D_LAUNCH_CHECK("r3d_add_cols");
  return R3D_OK;
}
