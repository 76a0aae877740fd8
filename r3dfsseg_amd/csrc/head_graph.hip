// Transductive label propagation on the k-NN graph of [prototypes ; query points], gfx950.
//
// Replaces (reference): models/mpti.py:717-756 calculateLocalConstrainedAffinity and
// :758-776 label_propagate.  The reference materialises a (n, 200, 192) neighbour tensor
// (675 MB @ n=4396), a dense n x n affinity (77 MB) and inverts a dense n x n matrix
// (170 GFLOP).  Here the graph stays sparse end to end:
//   1. neighbour lists (r3d_knn_topk, mode L2, k+1 columns, column 0 dropped)
//   2. symmetric pattern as an n x n BIT matrix (atomicOr; result order independent)
//   3. CSR rows enumerated from the bitmap in ascending column order (deterministic),
//      A_ij = [j in nbr(i)] w(i,j) + [i in nbr(j)] w(j,i),  w = exp(-0.5 (d/sigma)^2),
//      d = || x_i - x_j + 1e-6 ||_2 (torch-1.8 pairwise_distance)
//   4. S = D^-1/2 A D^-1/2, rounded as the reference's two diag matmuls
//   5. Z = (I - alpha S)^-1 Y by conjugate gradients (I - alpha S is SPD with spectrum in
//      [1-alpha, 1+alpha]); all n_way+1 right-hand sides share every SpMV.  The reference's
//      "+ eps" on every matrix element (2.2e-16) is below fp32 resolution of the diagonal
//      and perturbs Z by < 1e-9; it is dropped (tests/test_oracle_props.py shows the bound).
// Node count n lives in device memory (descriptor word HD_N_NODES); grids are sized by
// the capacity n_cap.
#include "common.h"

#define HG_NC 4                 // label columns carried (n_way + 1 <= 4), float4 per node
#define HG_ROWS_PER_BLOCK_MIN 4 // CG SpMV: rows per 256-thread block (1 per wave; more when n_cap / 4 > HG_MAX_PART)
#define HG_MAX_PART 2048        // max CG blocks

// ---------------------------------------------------------------------------
// 2. bitmaps: outb[i] = { j : j in nbr(i) } ; sym[i] = outb[i] | { j : i in nbr(j) }
// ---------------------------------------------------------------------------
__global__ void r3d_graph_bits_kernel(const int* __restrict__ nbr, int kp1, const int* __restrict__ n_dev,
                                      int n_cap, int words, unsigned* __restrict__ outb,
                                      unsigned* __restrict__ sym) {
  const int n = min(*n_dev, n_cap);
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = kp1 - 1;
  if (e >= (long)n * k) return;
  const int i = (int)(e / k);
  const int t = (int)(e - (long)i * k) + 1;  // column 0 is dropped (mpti.py:736)
  const int j = nbr[(long)i * kp1 + t];
  if (j < 0 || j >= n || j == i) return;      // diagonal is zeroed by the reference (mpti.py:755)
  atomicOr(&outb[(long)i * words + (j >> 5)], 1u << (j & 31));
  atomicOr(&sym[(long)i * words + (j >> 5)], 1u << (j & 31));
  atomicOr(&sym[(long)j * words + (i >> 5)], 1u << (i & 31));
}

// ---------------------------------------------------------------------------
// 3a. row lengths + exclusive scan (single workgroup; n_cap <= 32768)
// ---------------------------------------------------------------------------
__global__ void r3d_graph_rowlen_kernel(const unsigned* __restrict__ sym, int words,
                                        const int* __restrict__ n_dev, int n_cap, int* __restrict__ row_len) {
  const int n = min(*n_dev, n_cap);
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n_cap) return;
  int c = 0;
  if (row < n)
    for (int wd = lane; wd < words; wd += 64) c += __popc(sym[(long)row * words + wd]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (lane == 0) row_len[row] = c;
}

__global__ __launch_bounds__(1024) void r3d_scan_kernel(const int* __restrict__ in, int n, int* __restrict__ out) {
  // exclusive scan of in[0..n) -> out[0..n], out[n] = total.  One workgroup.
  __shared__ int wave_tot[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int t0 = 0; t0 < n; t0 += 1024) {
    const int i = t0 + tid;
    const int v = i < n ? in[i] : 0;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (lane == 63) wave_tot[w] = x;
    __syncthreads();
    int wbase = 0, tot = 0;
    for (int q = 0; q < 16; ++q) {
      const int tv = wave_tot[q];
      if (q < w) wbase += tv;
      tot += tv;
    }
    const int carry = carry_s;
    if (i < n) out[i] = carry + wbase + x - v;
    __syncthreads();
    if (tid == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (tid == 0) out[n] = carry_s;
}

// ---------------------------------------------------------------------------
// 3b. CSR columns, ascending, one wave per row
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void r3d_graph_cols_kernel(const unsigned* __restrict__ sym, int words,
                                                             const int* __restrict__ n_dev, int n_cap,
                                                             const int* __restrict__ row_ptr, int* __restrict__ col) {
  const int n = min(*n_dev, n_cap);
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  // lane owns a contiguous run of bitmap words so that columns come out ascending
  const int wpl = (words + 63) / 64;
  const int w0 = lane * wpl;
  int mycount = 0;
  for (int t = 0; t < wpl; ++t) {
    const int wd = w0 + t;
    if (wd < words) mycount += __popc(sym[(long)i * words + wd]);
  }
  int incl = mycount;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(incl, o);
    if (lane >= o) incl += y;
  }
  int wpos = row_ptr[i] + incl - mycount;
  for (int t = 0; t < wpl; ++t) {
    const int wd = w0 + t;
    if (wd >= words) break;
    unsigned bits = sym[(long)i * words + wd];
    while (bits) {
      const int b = __ffs((int)bits) - 1;
      bits &= bits - 1;
      col[wpos++] = wd * 32 + b;
    }
  }
}

// 3c. gaussian weights + row sums D.  One WORKGROUP per row i, its four waves take every fourth group of eight
//     entries, EIGHT LANES PER ENTRY: lane 8e + p reads float4 number p, p + 8, ... of neighbour e's row (a wave load
//     touches 8 rows x one 128-B line), the eight partial sums of an entry meet in a three-step butterfly.  A wave's
//     work is a chain of (column index -> neighbour rows) round trips, ~3 us each with the 3.4 MB node matrix
//     spilling out of a 4 MB L2: one wave per row walked 34 of them (250 us at S), a quarter of a row leaves 9 and
//     four times as many waves to overlap.  Weights are not index-deciding: the summation order differs from
//     oracle/r3d_oracle.c:orc_pair_dist by rounding only.
__global__ __launch_bounds__(256) void r3d_graph_weights_kernel(
    const float* __restrict__ nodes, long ldn, int D, const unsigned* __restrict__ outb, int words,
    const int* __restrict__ n_dev, int n_cap, const int* __restrict__ row_ptr, const int* __restrict__ col,
    float sigma, float* __restrict__ val, float* __restrict__ dinv, float* __restrict__ wdir /* [nnz][2] */) {
  __shared__ __attribute__((aligned(16))) float xs[256];
  __shared__ float wsum[4];
  const int n = min(*n_dev, n_cap);
  const int w = threadIdx.x >> 6;
  const int i = blockIdx.x;
  const int lane = threadIdx.x & 63;
  if (i >= n) return;  // uniform over the workgroup
  const float* xi = nodes + (long)i * ldn;
  xs[threadIdx.x] = (int)threadIdx.x < D ? xi[threadIdx.x] : 0.f;
  __syncthreads();
  const int beg = row_ptr[i], end = row_ptr[i + 1];
  const int D4 = D >> 2;
  const int part = lane & 7, slot = lane >> 3;
  float dsum = 0.f;
  for (int e0 = beg + 8 * w; e0 < end; e0 += 32) {
    const int e = e0 + slot;
    const bool ok = e < end;
    const int j = col[min(e, end - 1)];
    const float* xj = nodes + (long)j * ldn;
    float a = 0.f, b = 0.f;  // a: ||x_i - x_j + eps||^2, b: ||x_j - x_i + eps||^2 (partial: this lane's channels)
    auto step = [&](const float4 y, int c4) {
      const float4 x = *reinterpret_cast<const float4*>(&xs[4 * c4]);
      float d1, d2;
      d1 = (x.x - y.x) + 1e-6f; d2 = (y.x - x.x) + 1e-6f; a = __builtin_fmaf(d1, d1, a); b = __builtin_fmaf(d2, d2, b);
      d1 = (x.y - y.y) + 1e-6f; d2 = (y.y - x.y) + 1e-6f; a = __builtin_fmaf(d1, d1, a); b = __builtin_fmaf(d2, d2, b);
      d1 = (x.z - y.z) + 1e-6f; d2 = (y.z - x.z) + 1e-6f; a = __builtin_fmaf(d1, d1, a); b = __builtin_fmaf(d2, d2, b);
      d1 = (x.w - y.w) + 1e-6f; d2 = (y.w - x.w) + 1e-6f; a = __builtin_fmaf(d1, d1, a); b = __builtin_fmaf(d2, d2, b);
    };
    int c4 = part;
    for (; c4 + 40 < D4; c4 += 48) {  // six float4 per lane in flight (D = 192: exactly one trip)
      float4 y[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) y[u] = *reinterpret_cast<const float4*>(xj + 4 * (c4 + 8 * u));
#pragma unroll
      for (int u = 0; u < 6; ++u) step(y[u], c4 + 8 * u);
    }
    for (; c4 < D4; c4 += 8) step(*reinterpret_cast<const float4*>(xj + 4 * c4), c4);
    if (part == 0) {
      for (int c = 4 * D4; c < D; ++c) {
        const float y = xj[c], x = xs[c];
        const float d1 = (x - y) + 1e-6f, d2 = (y - x) + 1e-6f;
        a = __builtin_fmaf(d1, d1, a);
        b = __builtin_fmaf(d2, d2, b);
      }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      a += __shfl_xor(a, o);
      b += __shfl_xor(b, o);
    }
    const bool out_ij = (outb[(long)i * words + (j >> 5)] >> (j & 31)) & 1u;
    const bool out_ji = (outb[(long)j * words + (i >> 5)] >> (i & 31)) & 1u;
    float wij = 0.f, wji = 0.f;
    if (out_ij) { const float d = sqrtf(a) / sigma; wij = expf(-0.5f * (d * d)); }
    if (out_ji) { const float d = sqrtf(b) / sigma; wji = expf(-0.5f * (d * d)); }
    const float wgt = wij + wji;
    if (ok && part == 0) {
      val[e] = wgt;
      *reinterpret_cast<float2*>(wdir + 2L * e) = make_float2(wij, wji);
      dsum += wgt;
    }
  }
  dsum = r3d_wave_sum(dsum);
  if (lane == 0) wsum[w] = dsum;
  __syncthreads();
  if (threadIdx.x == 0)
    dinv[i] = sqrtf(1.0f / ((((wsum[0] + wsum[1]) + wsum[2]) + wsum[3]) + 2.220446049250313e-16f));  // mpti.py:768-770
}

// 4. S_ij = (dinv_i * A_ij) * dinv_j   (mpti.py:771-772: two diagonal matmuls)
__global__ void r3d_graph_normalize_kernel(const int* __restrict__ row_ptr, const int* __restrict__ col,
                                           const float* __restrict__ dinv, const int* __restrict__ n_dev,
                                           int n_cap, float* __restrict__ val) {
  const int n = min(*n_dev, n_cap);
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const float di = dinv[i];
  for (int e = row_ptr[i] + lane; e < row_ptr[i + 1]; e += 64) val[e] = (di * val[e]) * dinv[col[e]];
}

// ---------------------------------------------------------------------------
// 5. conjugate gradients on M = I - alpha S, HG_NC right-hand sides at once.
//    Two launches per iteration; scalar reductions are done redundantly by every block
//    from per-block partials written by the previous launch (no grid barrier, no host
//    sync, deterministic summation order).
//    Why not one persistent kernel with a grid barrier: measured on MI355X (tools/probe/grid_barrier.hip) a
//    device-wide barrier whose workgroups exchange data needs agent-scope release/acquire fences, i.e. an L2
//    write-back + invalidate per workgroup, and costs 3 us at 32 workgroups, 7 us at 128 and 13 us at 256
//    (43 us with a fence in every wave); a persistent CG with two such barriers per iteration ran 3x SLOWER
//    (100 us / iteration) than these two launches (17 us / iteration back to back).
// ---------------------------------------------------------------------------
#define HG_MAX_ITER 1022
struct CgState {            // device memory
  float rr_hist[2][HG_NC];  // rr of the last two iterations
  float bb[HG_NC];          // ||b||^2 (of the deflated right-hand side)
  float defl[HG_NC];        // <u,b> / (<u,u> (1 - alpha)): multiple of u = D^1/2 1 added back to the solution
  int done;                 // statistics only: set once every column converged
  int iters;                // statistics only: iterations actually performed
  // stop[it] != 0: iteration `it` must not run.  A launch only READS stop[it] and only
  // WRITES stop[it + 1], so no flag is read and written inside one launch.
  int stop[HG_MAX_ITER + 2];
};

static __device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// fixed-order block reduction of a float4, result broadcast to all threads
static __device__ __forceinline__ float4 block_sum4(float4 v, float4* sm /*[4]*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o);
    v.z += __shfl_xor(v.z, o); v.w += __shfl_xor(v.w, o);
  }
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  float4 r = sm[0];
  for (int q = 1; q < 4; ++q) { r.x += sm[q].x; r.y += sm[q].y; r.z += sm[q].z; r.w += sm[q].w; }
  return r;
}

static __device__ __forceinline__ float4 reduce_partials(const float4* part, int nblk, float4* sm) {
  float4 a = f4_zero();
  for (int q = threadIdx.x; q < nblk; q += blockDim.x) {
    const float4 p = part[q];
    a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
  }
  return block_sum4(a, sm);
}

// Deflation of the one eigenvector that is known in closed form: S u = u for u = D^1/2 1 (u_i = 1 / dinv_i), so
// (I - alpha S) u = (1 - alpha) u is the SMALLEST eigenvalue of the system (0.01 at alpha = 0.99) -- the mode CG
// otherwise spends its first iterations discovering.  The right-hand side is split b = b' + u <u,b>/<u,u>; CG solves
// for b' (orthogonal to u, and every Krylov vector stays so up to rounding), the other part is u <u,b>/(<u,u>(1-alpha)).
// Measured at S: 25 -> 20 iterations, same residual.  Exact for disconnected graphs too (u is still an eigenvector).
__global__ __launch_bounds__(256) void r3d_cg_defl_dots_kernel(const float4* __restrict__ Y, const float* __restrict__ dinv,
                                                               const int* __restrict__ n_dev, int n_cap,
                                                               float4* __restrict__ part /* [2][HG_MAX_PART/2] */) {
  __shared__ float4 sm[4];
  const int n = min(*n_dev, n_cap);
  const int i = blockIdx.x * 256 + threadIdx.x;
  float4 uy = f4_zero(), uu = f4_zero();
  if (i < n) {
    const float u = 1.f / dinv[i];
    const float4 b = Y[i];
    uy = make_float4(u * b.x, u * b.y, u * b.z, u * b.w);
    uu.x = u * u;
  }
  uy = block_sum4(uy, sm);
  uu = block_sum4(uu, sm);
  if (threadIdx.x == 0) { part[blockIdx.x] = uy; part[HG_MAX_PART / 2 + blockIdx.x] = uu; }
}

__global__ __launch_bounds__(256) void r3d_cg_init_kernel(const float4* __restrict__ Y, const float* __restrict__ dinv,
                                                          const int* __restrict__ n_dev, int n_cap, float alpha_lp,
                                                          const float4* __restrict__ part_defl, int nblk,
                                                          float4* __restrict__ x, float4* __restrict__ r,
                                                          float4* __restrict__ p, float4* __restrict__ part_rr,
                                                          CgState* __restrict__ st) {
  __shared__ float4 sm[4];
  const int n = min(*n_dev, n_cap);
  const int i = blockIdx.x * 256 + threadIdx.x;
  // <u,b> and <u,u>: every block reduces the same partials in the same order
  const float4 uy = reduce_partials(part_defl, nblk, sm);
  const float4 uu4 = reduce_partials(part_defl + HG_MAX_PART / 2, nblk, sm);
  const float inv_uu = uu4.x > 0.f ? 1.f / uu4.x : 0.f;
  const float4 c = make_float4(uy.x * inv_uu, uy.y * inv_uu, uy.z * inv_uu, uy.w * inv_uu);
  float4 b = f4_zero();
  if (i < n) {
    const float u = 1.f / dinv[i];
    const float4 y = Y[i];
    b = make_float4(y.x - u * c.x, y.y - u * c.y, y.z - u * c.z, y.w - u * c.w);
  }
  if (i < n_cap) { x[i] = f4_zero(); r[i] = b; p[i] = f4_zero(); }
  float4 sq = make_float4(b.x * b.x, b.y * b.y, b.z * b.z, b.w * b.w);
  sq = block_sum4(sq, sm);
  if (threadIdx.x == 0) part_rr[blockIdx.x] = sq;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->done = 0;
    st->iters = 0;
    const float k = 1.f / (1.f - alpha_lp);
    st->defl[0] = c.x * k; st->defl[1] = c.y * k; st->defl[2] = c.z * k; st->defl[3] = c.w * k;
  }
  if (blockIdx.x == 0)
    for (int q = threadIdx.x; q < HG_MAX_ITER + 2; q += 256) st->stop[q] = 0;
}

// x += u * defl: the closed-form component along the deflated eigenvector
__global__ void r3d_cg_defl_add_kernel(float4* __restrict__ x, const float* __restrict__ dinv, const int* __restrict__ n_dev,
                                       int n_cap, const CgState* __restrict__ st) {
  const int n = min(*n_dev, n_cap);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float u = 1.f / dinv[i];
  float4 v = x[i];
  v.x += u * st->defl[0]; v.y += u * st->defl[1]; v.z += u * st->defl[2]; v.w += u * st->defl[3];
  x[i] = v;
}

// A: p_new = r + beta p_old ; q = (I - alpha S) p_new ; partial <p_new, q>
// 5 waves per SIMD: the 4396 one-row waves of workload S must all be resident at once (108 registers -> 4 per SIMD ->
// two rounds of waves)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void r3d_cg_spmv_kernel(
    const int* __restrict__ row_ptr, const int* __restrict__ col, const float* __restrict__ val,
    const int* __restrict__ n_dev, int n_cap, float alpha_lp, int it, int nblk_rr, float tol2, int rows_per_block,
    const float4* __restrict__ r, const float4* __restrict__ p_old, float4* __restrict__ p_new,
    float4* __restrict__ q, const float4* __restrict__ part_rr, float4* __restrict__ part_pq,
    CgState* __restrict__ st) {
  __shared__ float4 sm[4];
  __shared__ float4 wsum[4];
  if (st->stop[it]) return;
  const int n = min(*n_dev, n_cap);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float4 rr = reduce_partials(part_rr, nblk_rr, sm);
  float4 beta = f4_zero();
  if (it == 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->bb[0] = rr.x; st->bb[1] = rr.y; st->bb[2] = rr.z; st->bb[3] = rr.w; }
  } else {
    const float* ro = st->rr_hist[(it - 1) & 1];
    beta.x = ro[0] > 0.f ? rr.x / ro[0] : 0.f;
    beta.y = ro[1] > 0.f ? rr.y / ro[1] : 0.f;
    beta.z = ro[2] > 0.f ? rr.z / ro[2] : 0.f;
    beta.w = ro[3] > 0.f ? rr.w / ro[3] : 0.f;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float* rn = st->rr_hist[it & 1];
    rn[0] = rr.x; rn[1] = rr.y; rn[2] = rr.z; rn[3] = rr.w;
  }
  float4 acc_pq = f4_zero();
  // few rows per wave: the SpMV is a chain of dependent loads per row (col -> gather), so its time is
  // rows-per-wave x that latency: 1 row per wave at n = 4.4k (was 8) took the CG iteration from 21 to 16 us
  const int row0 = blockIdx.x * rows_per_block + w * (rows_per_block / 4);
  for (int rr_i = 0; rr_i < rows_per_block / 4; ++rr_i) {
    const int i = row0 + rr_i;
    if (i >= n) break;
    float4 s = f4_zero();
    {
      const int rb = row_ptr[i], re = row_ptr[i + 1];
      // 6 entries per lane in flight: a row of the symmetrised 200-NN graph (~270 entries at S) is ONE trip of the
      // dependent chain column index -> gather (with 4 per lane the last 14 entries cost a second full round trip)
      for (int e0 = rb + lane; e0 < re; e0 += 384) {
        int jv[6];
        float av[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const int e = e0 + 64 * u;
          const int ec = min(e, re - 1);
          jv[u] = col[ec];
          av[u] = r3d_keep(val[ec], e < re);
        }
        float4 rj[6], pj[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) { rj[u] = r[jv[u]]; pj[u] = p_old[jv[u]]; }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          s.x += av[u] * (rj[u].x + beta.x * pj[u].x);
          s.y += av[u] * (rj[u].y + beta.y * pj[u].y);
          s.z += av[u] * (rj[u].z + beta.z * pj[u].z);
          s.w += av[u] * (rj[u].w + beta.w * pj[u].w);
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s.x += __shfl_xor(s.x, o); s.y += __shfl_xor(s.y, o);
      s.z += __shfl_xor(s.z, o); s.w += __shfl_xor(s.w, o);
    }
    if (lane == 0) {
      const float4 ri = r[i], pi = p_old[i];
      float4 pn = make_float4(ri.x + beta.x * pi.x, ri.y + beta.y * pi.y, ri.z + beta.z * pi.z, ri.w + beta.w * pi.w);
      float4 qi = make_float4(pn.x - alpha_lp * s.x, pn.y - alpha_lp * s.y, pn.z - alpha_lp * s.z, pn.w - alpha_lp * s.w);
      p_new[i] = pn;
      q[i] = qi;
      acc_pq.x += pn.x * qi.x; acc_pq.y += pn.y * qi.y; acc_pq.z += pn.z * qi.z; acc_pq.w += pn.w * qi.w;
    }
  }
  // fixed-order combine of the four waves' lane-0 partials
  __syncthreads();
  if (lane == 0) wsum[w] = acc_pq;
  __syncthreads();
  if (threadIdx.x == 0) {
    float4 t = wsum[0];
    for (int q2 = 1; q2 < 4; ++q2) { t.x += wsum[q2].x; t.y += wsum[q2].y; t.z += wsum[q2].z; t.w += wsum[q2].w; }
    part_pq[blockIdx.x] = t;
  }
}

// B: alpha = rr / <p,q> ; x += alpha p ; r -= alpha q ; partial rr ; convergence flag
__global__ __launch_bounds__(256) void r3d_cg_update_kernel(
    const int* __restrict__ n_dev, int n_cap, int it, int nblk_pq, float tol2, const float4* __restrict__ p,
    const float4* __restrict__ q, float4* __restrict__ x, float4* __restrict__ r,
    const float4* __restrict__ part_pq, float4* __restrict__ part_rr, CgState* __restrict__ st) {
  __shared__ float4 sm[4];
  if (st->stop[it]) {
    if (blockIdx.x == 0 && threadIdx.x == 0) st->stop[it + 1] = 1;
    return;
  }
  const int n = min(*n_dev, n_cap);
  const float4 pq = reduce_partials(part_pq, nblk_pq, sm);
  const float* rn = st->rr_hist[it & 1];
  const float* bb = st->bb;
  // converged already before this step? (all columns)  -> freeze the solution
  const bool conv = rn[0] <= tol2 * bb[0] && rn[1] <= tol2 * bb[1] && rn[2] <= tol2 * bb[2] && rn[3] <= tol2 * bb[3];
  if (conv) {
    // every block takes this branch together (same inputs, same arithmetic); the flag is
    // only read by LATER launches
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->done = 1; st->stop[it + 1] = 1; }
    return;
  }
  float4 al;
  al.x = pq.x > 0.f ? rn[0] / pq.x : 0.f;
  al.y = pq.y > 0.f ? rn[1] / pq.y : 0.f;
  al.z = pq.z > 0.f ? rn[2] / pq.z : 0.f;
  al.w = pq.w > 0.f ? rn[3] / pq.w : 0.f;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float4 sq = f4_zero();
  if (i < n) {
    const float4 pi = p[i], qi = q[i];
    float4 xi = x[i], ri = r[i];
    xi.x += al.x * pi.x; xi.y += al.y * pi.y; xi.z += al.z * pi.z; xi.w += al.w * pi.w;
    ri.x -= al.x * qi.x; ri.y -= al.y * qi.y; ri.z -= al.z * qi.z; ri.w -= al.w * qi.w;
    x[i] = xi; r[i] = ri;
    sq = make_float4(ri.x * ri.x, ri.y * ri.y, ri.z * ri.z, ri.w * ri.w);
  }
  sq = block_sum4(sq, sm);
  if (threadIdx.x == 0) part_rr[blockIdx.x] = sq;
  if (blockIdx.x == 0 && threadIdx.x == 0) st->iters = it + 1;
}

// ---------------------------------------------------------------------------
// 6. query logits (mpti.py:558-559) + cross entropy (mpti.py:778-781)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void r3d_logits_ce_kernel(const float4* __restrict__ Z, const int* __restrict__ desc_nproto,
                                                             int n_q, int N, int n_classes,
                                                             const long long* __restrict__ labels,
                                                             float* __restrict__ logits /* (n_q, n_classes, N) */,
                                                             float* __restrict__ loss_out, int* __restrict__ pred_out) {
  __shared__ float red[16];
  const int n_proto = *desc_nproto;
  float acc = 0.f;
  for (int e = threadIdx.x; e < n_q * N; e += blockDim.x) {
    const int qi = e / N, p = e - qi * N;
    const float4 z = Z[n_proto + e];
    const float zv[4] = {z.x, z.y, z.z, z.w};
    float mx = zv[0];
    int am = 0;
    for (int c = 1; c < n_classes; ++c) if (zv[c] > mx) { mx = zv[c]; am = c; }
    float se = 0.f;
    for (int c = 0; c < n_classes; ++c) {
      logits[((long)qi * n_classes + c) * N + p] = zv[c];
      se += expf(zv[c] - mx);
    }
    if (labels) {
      const int lab = (int)labels[e];
      acc += (mx + logf(se)) - zv[lab];
    }
    if (pred_out) pred_out[e] = am;
  }
  acc = r3d_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    if (loss_out) *loss_out = t / (float)(n_q * N);
  }
}

// ===========================================================================
// C ABI
// ===========================================================================
// scratch words for r3d_label_propagate: bitmaps, CSR, CG vectors
extern "C" long r3d_lp_ws_words(int n_cap, int kp1) {
  const long words = (n_cap + 31) / 32;
  const long nnz_cap = 2L * n_cap * (kp1 - 1);
  long t = 0;
  t += 2 * n_cap * words;          // outb, sym
  t += n_cap + 8;                  // row_len
  t += n_cap + 8;                  // row_ptr
  t += nnz_cap * 2;                // col, val
  t += nnz_cap * 2;                // wdir: directed gaussian weights (w_ij, w_ji) per entry, kept for the backward
  t += n_cap;                      // dinv
  t += 5L * n_cap * HG_NC;         // x(out is separate) r, p0, p1, q  (+1 spare)
  t += 2L * HG_MAX_PART * HG_NC;   // partials
  t += sizeof(CgState) / 4 + 8;    // CgState
  return t + 64;
}

// nodes (n_cap, ldn), nbr (n_cap, kp1) from r3d_knn_topk (mode L2), Y (n_cap, 4) one-hot
// rows for prototypes / zeros for queries.  Z (n_cap, 4) out.  n_dev: device int = n.
// stats_out (optional, device, 2 ints): {converged flag, iterations}.
struct LpWs {
  unsigned *outb, *sym;
  int *row_len, *row_ptr, *col;
  float *val, *dinv, *wdir;
  float4 *r, *p0, *p1, *q, *part_rr, *part_pq;
  CgState* cg;
  long words;
};

static LpWs lp_carve(int32_t* ws, int n_cap, int kp1) {
  LpWs L;
  L.words = (n_cap + 31) / 32;
  const long nnz_cap = 2L * n_cap * (kp1 - 1);
  int32_t* wp = ws;
  L.outb = (unsigned*)wp; wp += n_cap * L.words;
  L.sym = (unsigned*)wp; wp += n_cap * L.words;
  L.row_len = wp; wp += n_cap + 8;
  L.row_ptr = wp; wp += n_cap + 8;
  L.col = wp; wp += nnz_cap;
  L.val = (float*)wp; wp += nnz_cap;
  L.wdir = (float*)wp; wp += 2 * nnz_cap;
  L.dinv = (float*)wp; wp += n_cap;
  wp += (4 - ((wp - ws) & 3)) & 3;  // float4 alignment (ws itself must be 16-B aligned)
  L.r = (float4*)wp; wp += 4L * n_cap;
  L.p0 = (float4*)wp; wp += 4L * n_cap;
  L.p1 = (float4*)wp; wp += 4L * n_cap;
  L.q = (float4*)wp; wp += 4L * n_cap;
  L.part_rr = (float4*)wp; wp += 4L * HG_MAX_PART;
  L.part_pq = (float4*)wp; wp += 4L * HG_MAX_PART;
  L.cg = (CgState*)wp;
  return L;
}

// CG on the already built graph: X = (I - alpha S)^-1 RHS
static int lp_solve(const LpWs& L, const float* RHS, const int32_t* n_dev, int n_cap, float alpha, int max_iter, float tol,
                    float* X, int32_t* stats_out, hipStream_t st) {
  const int nblk_v = r3d_cdiv(n_cap, 256);
  int rpb = HG_ROWS_PER_BLOCK_MIN;
  while (r3d_cdiv(n_cap, rpb) > HG_MAX_PART) rpb += 4;
  const int nblk_s = r3d_cdiv(n_cap, rpb);
  R3D_REQUIRE(nblk_s <= HG_MAX_PART, "r3d_label_propagate: n_cap too large");
  float4* x = (float4*)X;
  R3D_REQUIRE(nblk_v <= HG_MAX_PART / 2, "r3d_label_propagate: n_cap too large");
  hipLaunchKernelGGL(r3d_cg_defl_dots_kernel, dim3(nblk_v), dim3(256), 0, st, (const float4*)RHS, L.dinv, n_dev, n_cap, L.part_pq);
  hipLaunchKernelGGL(r3d_cg_init_kernel, dim3(nblk_v), dim3(256), 0, st, (const float4*)RHS, L.dinv, n_dev, n_cap, alpha,
                     L.part_pq, nblk_v, x, L.r, L.p0, L.part_rr, L.cg);
  const float tol2 = tol * tol;
  for (int it = 0; it < max_iter; ++it) {
    float4* pold = (it & 1) ? L.p1 : L.p0;
    float4* pnew = (it & 1) ? L.p0 : L.p1;
    hipLaunchKernelGGL(r3d_cg_spmv_kernel, dim3(nblk_s), dim3(256), 0, st, L.row_ptr, L.col, L.val, n_dev, n_cap, alpha,
                       it, nblk_v, tol2, rpb, L.r, pold, pnew, L.q, L.part_rr, L.part_pq, L.cg);
    hipLaunchKernelGGL(r3d_cg_update_kernel, dim3(nblk_v), dim3(256), 0, st, n_dev, n_cap, it, nblk_s, tol2, pnew, L.q,
                       x, L.r, L.part_pq, L.part_rr, L.cg);
  }
  hipLaunchKernelGGL(r3d_cg_defl_add_kernel, dim3(nblk_v), dim3(256), 0, st, x, L.dinv, n_dev, n_cap, L.cg);
  if (stats_out) r3d_copy_words(stats_out, &L.cg->done, 2, st);
  return R3D_OK;
}

extern "C" int r3d_label_propagate(const float* nodes, long ldn, int D, const int32_t* nbr, int kp1,
                                   const float* Y, const int32_t* n_dev, int n_cap, float sigma,
                                   float alpha, int max_iter, float tol, float* Z, int32_t* ws,
                                   int32_t* stats_out, void* stream) {
  R3D_REQUIRE(nodes && nbr && Y && n_dev && Z && ws, "r3d_label_propagate: null pointer");
  R3D_REQUIRE(n_cap > 0 && n_cap <= 32768 && D > 0 && D <= 256 && kp1 >= 2,
              "r3d_label_propagate: unsupported n_cap=%d D=%d kp1=%d", n_cap, D, kp1);
  R3D_REQUIRE(max_iter > 0 && max_iter <= HG_MAX_ITER && sigma > 0.f, "r3d_label_propagate: bad solver parameters");
  R3D_REQUIRE(((uintptr_t)ws & 15) == 0 && ((uintptr_t)Y & 15) == 0 && ((uintptr_t)Z & 15) == 0,
              "r3d_label_propagate: ws, Y and Z must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const LpWs L = lp_carve(ws, n_cap, kp1);
  const long words = L.words;
  r3d_zero_words(L.outb, 2L * n_cap * words, st);
  const long edges = (long)n_cap * (kp1 - 1);
  hipLaunchKernelGGL(r3d_graph_bits_kernel, dim3(r3d_cdiv(edges, 256)), dim3(256), 0, st, nbr, kp1, n_dev, n_cap,
                     (int)words, L.outb, L.sym);
  hipLaunchKernelGGL(r3d_graph_rowlen_kernel, dim3(r3d_cdiv(n_cap, 4)), dim3(256), 0, st, L.sym, (int)words, n_dev,
                     n_cap, L.row_len);
  hipLaunchKernelGGL(r3d_scan_kernel, dim3(1), dim3(1024), 0, st, L.row_len, n_cap, L.row_ptr);
  hipLaunchKernelGGL(r3d_graph_cols_kernel, dim3(r3d_cdiv(n_cap, 4)), dim3(256), 0, st, L.sym, (int)words, n_dev,
                     n_cap, L.row_ptr, L.col);
  hipLaunchKernelGGL(r3d_graph_weights_kernel, dim3(n_cap), dim3(256), 0, st, nodes, ldn, D, L.outb,
                     (int)words, n_dev, n_cap, L.row_ptr, L.col, sigma, L.val, L.dinv, L.wdir);
  hipLaunchKernelGGL(r3d_graph_normalize_kernel, dim3(r3d_cdiv(n_cap, 4)), dim3(256), 0, st, L.row_ptr, L.col, L.dinv,
                     n_dev, n_cap, L.val);
  int rc = lp_solve(L, Y, n_dev, n_cap, alpha, max_iter, tol, Z, stats_out, st);
  if (rc) return rc;
  R3D_LAUNCH_CHECK("r3d_label_propagate");
  return R3D_OK;
}

// ---------------------------------------------------------------------------
// backward of the head's graph part (training): reference autograd through mpti.py:739-776
//   G = dL/dZ ; lambda = (I - alpha S)^-1 G (S symmetric) ; dL/dS_ij = alpha <lambda_i, Z_j>
//   S = dinv_i A_ij dinv_j, dinv = (D + eps)^-1/2, D_i = sum_j A_ij, A = W + W^T,
//   w_ij = exp(-0.5 ||x_i - x_j + eps||^2 / sigma^2) for j in nbr(i)
// Both passes walk the symmetric CSR rows (gather only, no atomics, fixed order).
// ---------------------------------------------------------------------------
static __device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// pass 1: dD_i = -1/2 dinv_i^3 * sum_j (dS_ij + dS_ji) A_ij dinv_j      (A_ij dinv_j = S_ij / dinv_i)
__global__ __launch_bounds__(256) void r3d_lp_bwd_dd_kernel(const int* __restrict__ row_ptr, const int* __restrict__ col,
                                                            const float* __restrict__ val, const float* __restrict__ dinv,
                                                            const int* __restrict__ n_dev, int n_cap, float alpha,
                                                            const float4* __restrict__ lam, const float4* __restrict__ Z,
                                                            float* __restrict__ dD) {
  const int n = min(*n_dev, n_cap);
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const float4 li = lam[i], zi = Z[i];
  const float di = dinv[i];
  float g = 0.f;
  for (int e = row_ptr[i] + lane; e < row_ptr[i + 1]; e += 64) {
    const int j = col[e];
    const float ds = alpha * (dot4(li, Z[j]) + dot4(lam[j], zi));
    g += ds * (val[e] / di);
  }
  g = r3d_wave_sum(g);
  if (lane == 0) dD[i] = -0.5f * di * di * di * g;
}

// pass 2: dx_i = sum over the row of c_ij (x_i - x_j + eps) + c_ji (x_i - x_j - eps),
//   c_ij = -(dA_ij + dA_ji) w_ij / sigma^2 with the directed weights w_ij, w_ji the forward pass left in wdir
//   (no distance is recomputed here).  One wave per row; lane-per-entry for the coefficients, then the wave
//   walks the entries and every lane accumulates its channels of the weighted neighbour sum (coalesced rows).
__global__ __launch_bounds__(256) void r3d_lp_bwd_dx_kernel(
    const float* __restrict__ nodes, long ldn, int D, const float* __restrict__ wdir,
    const int* __restrict__ row_ptr, const int* __restrict__ col, const float* __restrict__ dinv,
    const int* __restrict__ n_dev, int n_cap, float sigma, float alpha, const float4* __restrict__ lam,
    const float4* __restrict__ Z, const float* __restrict__ dD, float* __restrict__ dnodes, long ldd) {
  // one WORKGROUP per row, its four waves take every fourth chunk of 64 entries (a wave's work is a chain of
  // column-index -> neighbour-row round trips; one wave per row walked five chunks: 185 us at S)
  __shared__ float acc_s[4][256];
  __shared__ float uv_s[4][2];
  const int n = min(*n_dev, n_cap);
  const int w = threadIdx.x >> 6;
  const int i = blockIdx.x;
  const int lane = threadIdx.x & 63;
  if (i >= n) return;  // uniform over the workgroup
  const float* xi = nodes + (long)i * ldn;
  const float4 li = lam[i], zi = Z[i];
  const float di = dinv[i], ddi = dD[i];
  const int beg = row_ptr[i], end = row_ptr[i + 1];
  const float inv_s2 = 1.f / (sigma * sigma);
  const int c0 = min(lane, D - 1), c1 = min(lane + 64, D - 1), c2 = min(lane + 128, D - 1), c3 = min(lane + 192, D - 1);
  float U = 0.f, V = 0.f;                        // sum (c_ij + c_ji), sum (c_ij - c_ji)
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // sum_j u_j x_j[c], channels lane, lane+64, ...
  for (int e0 = beg + 64 * w; e0 < end; e0 += 256) {
    const int e = e0 + lane;
    const bool ok = e < end;
    const int ec = min(e, end - 1);
    const int j = col[ec];
    const float2 wd = *reinterpret_cast<const float2*>(wdir + 2L * ec);
    const float dj = dinv[j];
    const float T = alpha * (dot4(li, Z[j]) + dot4(lam[j], zi)) * di * dj + ddi + dD[j];  // dA_ij + dA_ji
    const float cij = ok ? -T * wd.x * inv_s2 : 0.f;
    const float cji = ok ? -T * wd.y * inv_s2 : 0.f;
    const float u = cij + cji;
    U += u;
    V += cij - cji;
    const int cnt = min(64, end - e0);
    int t = 0;
    for (; t + 8 <= cnt; t += 8) {  // 8 neighbour rows (up to 32 loads) in flight
      float x0[8], x1[8], x2[8], x3[8], ut[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        ut[q] = r3d_readlane_f(u, t + q);
        const float* xr = nodes + (long)__builtin_amdgcn_readlane(j, t + q) * ldn;
        x0[q] = xr[c0]; x1[q] = xr[c1]; x2[q] = xr[c2];
        x3[q] = D > 192 ? xr[c3] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        a0 = __builtin_fmaf(ut[q], x0[q], a0); a1 = __builtin_fmaf(ut[q], x1[q], a1);
        a2 = __builtin_fmaf(ut[q], x2[q], a2); a3 = __builtin_fmaf(ut[q], x3[q], a3);
      }
    }
    for (; t < cnt; ++t) {
      const float ut = r3d_readlane_f(u, t);
      const float* xr = nodes + (long)__builtin_amdgcn_readlane(j, t) * ldn;
      a0 = __builtin_fmaf(ut, xr[c0], a0); a1 = __builtin_fmaf(ut, xr[c1], a1);
      a2 = __builtin_fmaf(ut, xr[c2], a2); a3 = __builtin_fmaf(ut, D > 192 ? xr[c3] : 0.f, a3);
    }
  }
  U = r3d_wave_sum(U);
  V = r3d_wave_sum(V);
  acc_s[w][lane] = a0; acc_s[w][64 + lane] = a1; acc_s[w][128 + lane] = a2; acc_s[w][192 + lane] = a3;
  if (lane == 0) { uv_s[w][0] = U; uv_s[w][1] = V; }
  __syncthreads();
  const int c = threadIdx.x;
  if (c < D) {
    const float At = ((acc_s[0][c] + acc_s[1][c]) + acc_s[2][c]) + acc_s[3][c];
    const float Ut = ((uv_s[0][0] + uv_s[1][0]) + uv_s[2][0]) + uv_s[3][0];
    const float Vt = ((uv_s[0][1] + uv_s[1][1]) + uv_s[2][1]) + uv_s[3][1];
    dnodes[(long)i * ldd + c] = xi[c] * Ut - At + 1e-6f * Vt;
  }
}

// dL/dZ of the mean cross entropy over the query rows (mpti.py:778-781), scaled by *gscale
__global__ void r3d_ce_grad_kernel(const float4* __restrict__ Z, const int* __restrict__ n_proto_dev, int n_cap, int n_qpts,
                                   int n_classes, const long long* __restrict__ labels, const float* __restrict__ gscale,
                                   float4* __restrict__ G) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cap) return;
  const int n_proto = *n_proto_dev;
  const int q = i - n_proto;
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  if (q >= 0 && q < n_qpts) {
    const float4 z = Z[i];
    const float zv[4] = {z.x, z.y, z.z, z.w};
    float mx = zv[0];
    for (int c = 1; c < n_classes; ++c) mx = fmaxf(mx, zv[c]);
    float se = 0.f;
    for (int c = 0; c < n_classes; ++c) se += expf(zv[c] - mx);
    const int lab = (int)labels[q];
    const float sc = gscale[0] / (float)n_qpts;
    for (int c = 0; c < n_classes; ++c) g[c] = (expf(zv[c] - mx) / se - (c == lab ? 1.f : 0.f)) * sc;
  }
  G[i] = make_float4(g[0], g[1], g[2], g[3]);
}

// Backward through label propagation + affinity.  Requires ws exactly as r3d_label_propagate left it.
// G (n_cap,4) = dL/dZ (from r3d_ce_grad); lam scratch (n_cap,4); dnodes (n_cap, ldd) out.
extern "C" int r3d_label_propagate_bwd(const float* nodes, long ldn, int D, int kp1, const float* Z, const float* G,
                                       const int32_t* n_dev, int n_cap, float sigma, float alpha, int max_iter, float tol,
                                       float* lam, float* dnodes, long ldd, int32_t* ws, int32_t* stats_out, void* stream) {
  R3D_REQUIRE(nodes && Z && G && n_dev && lam && dnodes && ws, "r3d_label_propagate_bwd: null pointer");
  R3D_REQUIRE(n_cap > 0 && n_cap <= 32768 && D > 0 && D <= 256 && max_iter > 0 && max_iter <= HG_MAX_ITER,
              "r3d_label_propagate_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const LpWs L = lp_carve(ws, n_cap, kp1);
  int rc = lp_solve(L, G, n_dev, n_cap, alpha, max_iter, tol, lam, stats_out, st);
  if (rc) return rc;
  float* dD = (float*)L.q;  // the CG vectors are free again after the solve
  hipLaunchKernelGGL(r3d_lp_bwd_dd_kernel, dim3(r3d_cdiv(n_cap, 4)), dim3(256), 0, st, L.row_ptr, L.col, L.val, L.dinv, n_dev,
                     n_cap, alpha, (const float4*)lam, (const float4*)Z, dD);
  hipLaunchKernelGGL(r3d_lp_bwd_dx_kernel, dim3(n_cap), dim3(256), 0, st, nodes, ldn, D, L.wdir,
                     L.row_ptr, L.col, L.dinv, n_dev, n_cap, sigma, alpha, (const float4*)lam, (const float4*)Z, dD, dnodes,
                     ldd);
  R3D_LAUNCH_CHECK("r3d_label_propagate_bwd");
  return R3D_OK;
}

extern "C" int r3d_ce_grad(const float* Z, const int32_t* n_proto_dev, int n_cap, int n_query_pts, int n_classes,
                           const int64_t* labels, const float* gscale_dev, float* G, void* stream) {
  R3D_REQUIRE(Z && n_proto_dev && labels && gscale_dev && G, "r3d_ce_grad: null pointer");
  hipLaunchKernelGGL(r3d_ce_grad_kernel, dim3(r3d_cdiv(n_cap, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)Z,
                     n_proto_dev, n_cap, n_query_pts, n_classes, (const long long*)labels, gscale_dev, (float4*)G);
  R3D_LAUNCH_CHECK("r3d_ce_grad");
  return R3D_OK;
}

// logits (n_q, n_classes, N) fp32, loss (1) fp32, pred (n_q*N) int32 (argmax), labels int64
extern "C" int r3d_query_logits_ce(const float* Z, const int32_t* n_proto_dev, int n_q, int N, int n_classes,
                                   const int64_t* labels, float* logits, float* loss_out, int32_t* pred_out,
                                   void* stream) {
  R3D_REQUIRE(Z && n_proto_dev && logits, "r3d_query_logits_ce: null pointer");
  R3D_REQUIRE(n_q > 0 && N > 0 && n_classes >= 2 && n_classes <= HG_NC, "r3d_query_logits_ce: bad shape");
  hipLaunchKernelGGL(r3d_logits_ce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float4*)Z,
                     n_proto_dev, n_q, N, n_classes, (const long long*)labels, logits, loss_out, pred_out);
  R3D_LAUNCH_CHECK("r3d_query_logits_ce");
  return R3D_OK;
}

// ---------------------------------------------------------------------------
// Captured episodes (hipGraph): the CG loop of r3d_label_propagate / _bwd is frozen into the graph with its full
// launch budget; iterations after convergence are launches that return at once, yet each still costs ~2.5 us of
// queue time (141 of 200 pairs on average at workload S).  This call enables the CG kernel nodes of iterations
// < budget in the instantiated graph and disables the others (a disabled node is an empty node: no dispatch), so
// the owner can track the iteration counts it observes without re-capturing.  `graph` is the captured hipGraph_t
// the executable graph was instantiated from.  Returns the number of CG nodes found through *n_cg (optional).
// A solve that needs more than the enabled iterations reports "not converged" through stats_out as before.
// ---------------------------------------------------------------------------
extern "C" int r3d_graph_set_lp_budget(void* graph, void* graph_exec, int budget, int* n_cg) {
  R3D_REQUIRE(graph && graph_exec && budget > 0, "r3d_graph_set_lp_budget: bad arguments");
  hipGraph_t g = (hipGraph_t)graph;
  hipGraphExec_t ge = (hipGraphExec_t)graph_exec;
  size_t n = 0;
  R3D_REQUIRE(hipGraphGetNodes(g, nullptr, &n) == hipSuccess, "r3d_graph_set_lp_budget: hipGraphGetNodes failed");
  hipGraphNode_t* nodes = (hipGraphNode_t*)malloc(sizeof(hipGraphNode_t) * (n ? n : 1));
  R3D_REQUIRE(nodes, "r3d_graph_set_lp_budget: out of memory");
  int found = 0, rc = R3D_OK;
  if (hipGraphGetNodes(g, nodes, &n) != hipSuccess) rc = R3D_ERR_LAUNCH;
  for (size_t i = 0; rc == R3D_OK && i < n; ++i) {
    hipGraphNodeType ty;
    if (hipGraphNodeGetType(nodes[i], &ty) != hipSuccess) { rc = R3D_ERR_LAUNCH; break; }
    if (ty != hipGraphNodeTypeKernel) continue;
    hipKernelNodeParams kp;
    if (hipGraphKernelNodeGetParams(nodes[i], &kp) != hipSuccess) { rc = R3D_ERR_LAUNCH; break; }
    int it;
    if (kp.func == (void*)r3d_cg_spmv_kernel) it = *(const int*)kp.kernelParams[6];
    else if (kp.func == (void*)r3d_cg_update_kernel) it = *(const int*)kp.kernelParams[2];
    else continue;
    ++found;
    if (hipGraphNodeSetEnabled(ge, nodes[i], it < budget ? 1u : 0u) != hipSuccess) { rc = R3D_ERR_LAUNCH; break; }
  }
  free(nodes);
  if (rc != R3D_OK) {
    r3d_set_error("r3d_graph_set_lp_budget: HIP graph call failed: %s", hipGetErrorString(hipGetLastError()));
    return rc;
  }
  if (n_cg) *n_cg = found;
  return R3D_OK;
}

