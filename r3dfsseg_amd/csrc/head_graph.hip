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
typedef unsigned short hg_col_t; // CSR column ids: n_cap <= 32768 (checked at the entry points), half the bytes of int
#define HG_ROWS_PER_BLOCK_MIN 4 // CG SpMV: rows per 256-thread block (1 per wave; more when n_cap / 4 > HG_MAX_PART)
#define HG_MAX_PART 2048        // max CG blocks

// Batches of episodes (round 3): E label-propagation systems go through every launch of this file together --
// blockIdx.y (or .z) is the system, its arrays sit `stride` further on than system 0's (HgEp: node / label / neighbour
// ROWS, descriptor and status WORDS, and ONE stride in int32 words for everything carved out of the scratch).  The CG
// iteration is then two launches for all E systems, each with its own convergence flag (a converged system's
// workgroups return at once), and the SpMV streams E matrices: genuinely HBM bound instead of one 7 MB matrix that
// sits in L2 while 18 workgroups wait on each other.
struct HgEp {
  long nodes, nbr, y, z, g, lam, dn;  // rows of nodes / nbr / Y / Z / G / lambda / dnodes between systems
  long desc;                          // int32 words between the systems' (n_nodes, n_proto) descriptor words
  long ws;                            // int32 words between the systems' scratch (a multiple of 4: float4 arrays inside)
  long stats;                         // int32 words between the systems' {converged, iterations} pairs
  long labels, logits, loss, pred;    // query labels (int64), logits (floats), loss (floats), predictions (int32) per system
};
#define HG_WS(p) p = (decltype(p))((const char*)(p) + (long)ep * st.ws * 4)
#define HG_AT(p, stride) p += (long)ep * (stride)

// ---------------------------------------------------------------------------
// 2. bitmaps: outb[i] = { j : j in nbr(i) } ; sym[i] = outb[i] | { j : i in nbr(j) }
//    No global atomics.  r3d_graph_bits_kernel: one wave per row i builds outb[i] in LDS (its 200 neighbours, LDS
//    atomics) and stores it whole.  r3d_graph_transpose_kernel: the in-edges are the TRANSPOSE of that bit matrix: a
//    wave holds one word of 64 consecutive rows, `ballot` of bit b over the lanes IS column 32w + b of those rows (64
//    bits of the transposed row), tiles of 1024 rows x 256 columns meet in LDS and leave as 128-byte row pieces.
//    The union with the out-edges is taken by r3d_graph_rowlen_kernel, which reads every word anyway.
//    (First version: three device-scope atomicOr per edge, 84 M per 32-system step, 1.95 ms; one atomic per edge with
//    row-owned out-edges: 1.13 ms.)  Neither bitmap needs a zero fill: every word up to n_cap rows is stored.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void r3d_graph_bits_kernel(const int* __restrict__ nbr, int kp1, const int* __restrict__ n_dev,
                                                             int n_cap, int words, unsigned* __restrict__ outb, HgEp st) {
  __shared__ unsigned bm[4][1024];  // n_cap <= 32768 (checked at the entry points)
  const int ep = blockIdx.y;
  nbr += (long)ep * st.nbr * kp1; HG_AT(n_dev, st.desc); HG_WS(outb);
  const int n = min(*n_dev, n_cap);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + w;
  unsigned* b = bm[w];
  for (int wd = lane; wd < words; wd += 64) b[wd] = 0u;
  __syncthreads();
  if (i < n) {
    const int k = kp1 - 1;
    for (int t = lane; t < k; t += 64) {
      const int j = nbr[(long)i * kp1 + 1 + t];    // column 0 is dropped (mpti.py:736)
      if (j < 0 || j >= n || j == i) continue;     // diagonal is zeroed by the reference (mpti.py:755)
      atomicOr(&b[j >> 5], 1u << (j & 31));
    }
  }
  __syncthreads();
  if (i < n_cap)
    for (int wd = lane; wd < words; wd += 64) outb[(long)i * words + wd] = b[wd];
}

#define BT_ROWS 1024  // source rows per workgroup = 32 words of every transposed row
#define BT_CW 8       // source column words per workgroup = 256 transposed rows
__global__ __launch_bounds__(256) void r3d_graph_transpose_kernel(const unsigned* __restrict__ outb, int words, int n_cap,
                                                                  unsigned* __restrict__ inb /* inb[c] bit r = outb[r] bit c */,
                                                                  HgEp st) {
  __shared__ unsigned tile[BT_CW * 32][BT_ROWS / 32 + 1];
  const int ep = blockIdx.z;
  HG_WS(outb); HG_WS(inb);
  const int c0w = blockIdx.x * BT_CW, r0 = blockIdx.y * BT_ROWS;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int sb = w; sb < BT_ROWS / 64; sb += 4) {  // sub-blocks of 64 source rows, one lane per row
    const int r = r0 + sb * 64 + lane;
    unsigned v[BT_CW];
#pragma unroll
    for (int cw = 0; cw < BT_CW; ++cw) v[cw] = (r < n_cap && c0w + cw < words) ? outb[(long)r * words + c0w + cw] : 0u;
#pragma unroll
    for (int cw = 0; cw < BT_CW; ++cw) {
#pragma unroll
      for (int b = 0; b < 32; ++b) {
        const unsigned long long m = __ballot((v[cw] >> b) & 1u);
        if (lane == b) {
          tile[cw * 32 + b][2 * sb] = (unsigned)m;
          tile[cw * 32 + b][2 * sb + 1] = (unsigned)(m >> 32);
        }
      }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < BT_CW * 32 * (BT_ROWS / 32); e += 256) {
    const int c = e >> 5, wd = e & 31;
    const int col = c0w * 32 + c, ow = r0 / 32 + wd;
    if (col < n_cap && ow < words) inb[(long)col * words + ow] = tile[c][wd];
  }
}

// ---------------------------------------------------------------------------
// 3a. row lengths + exclusive scan (single workgroup; n_cap <= 32768)
// ---------------------------------------------------------------------------
__global__ void r3d_graph_rowlen_kernel(unsigned* __restrict__ sym /* in: in-edges; out: in | out edges */,
                                        const unsigned* __restrict__ outb, int words,
                                        const int* __restrict__ n_dev, int n_cap, int* __restrict__ row_len, HgEp st) {
  const int ep = blockIdx.y;
  HG_WS(sym); HG_WS(outb); HG_AT(n_dev, st.desc); HG_WS(row_len);
  const int n = min(*n_dev, n_cap);
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n_cap) return;
  int c = 0;
  if (row < n)
    for (int wd = lane; wd < words; wd += 64) {
      const unsigned v = sym[(long)row * words + wd] | outb[(long)row * words + wd];
      sym[(long)row * words + wd] = v;
      c += __popc(v);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (lane == 0) row_len[row] = c;
}

__global__ __launch_bounds__(1024) void r3d_scan_kernel(const int* __restrict__ in, int n, int* __restrict__ out, HgEp st) {
  // exclusive scan of in[0..n) -> out[0..n], out[n] = total.  One workgroup per system.
  const int ep = blockIdx.x;
  HG_WS(in); HG_WS(out);
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
                                                             const int* __restrict__ row_ptr, hg_col_t* __restrict__ col,
                                                             HgEp st) {
  const int ep = blockIdx.y;
  HG_WS(sym); HG_AT(n_dev, st.desc); HG_WS(row_ptr); HG_WS(col);
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
      col[wpos++] = (hg_col_t)(wd * 32 + b);
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
    const int* __restrict__ n_dev, int n_cap, const int* __restrict__ row_ptr, const hg_col_t* __restrict__ col,
    float sigma, float* __restrict__ val, float* __restrict__ dinv, float* __restrict__ wdir /* [nnz][2] */, HgEp st) {
  __shared__ __attribute__((aligned(16))) float xs[256];
  __shared__ float wsum[4];
  const int ep = blockIdx.y;
  nodes += (long)ep * st.nodes * ldn; HG_WS(outb); HG_AT(n_dev, st.desc); HG_WS(row_ptr); HG_WS(col); HG_WS(val); HG_WS(dinv);
  HG_WS(wdir);
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
    a = r3d_sum8_dpp(a);  // the eight lanes of an entry
    b = r3d_sum8_dpp(b);
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
__global__ void r3d_graph_normalize_kernel(const int* __restrict__ row_ptr, const hg_col_t* __restrict__ col,
                                           const float* __restrict__ dinv, const int* __restrict__ n_dev,
                                           int n_cap, float* __restrict__ val, HgEp st) {
  const int ep = blockIdx.y;
  HG_WS(row_ptr); HG_WS(col); HG_WS(dinv); HG_AT(n_dev, st.desc); HG_WS(val);
  const int n = min(*n_dev, n_cap);
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const float di = dinv[i];
  for (int e = row_ptr[i] + lane; e < row_ptr[i + 1]; e += 64) val[e] = (di * val[e]) * dinv[col[e]];
}

// ---------------------------------------------------------------------------
// 5. Two-level conjugate gradients on M = I - alpha S, HG_NC right-hand sides at once.
//
//    Spectrum: [1 - alpha, 1 + alpha] = [0.01, 1.99].  The slow modes are known from measurement (dumped systems,
//    profiles/r02_experiments.md): D^1/2 1 (eigenvalue of S exactly 1), then ONE MODE PER FEATURE CLUSTER once the
//    encoder has trained (0.9999, 0.9996 at workload S after 150 steps), then smooth modes inside the clusters
//    (0.98, 0.97, 0.96 ...).  Plain CG needs 48-67 iterations on such a system, 20-25 at initialisation.
//    Coarse space: W = D^1/2 [indicator of HG_M = 64 aggregates]; aggregate = nearest of 64 SEEDS in feature space,
//    the seeds being an even subsample of the prototypes (they are FPS-spread cluster means, so they cover the
//    feature clusters; the graph is the kNN graph of that same space).  span(W) contains D^1/2 1 exactly and the
//    cluster modes up to their boundary error.  The coarse operator E = W^T M W (64 x 64) is inverted once per
//    episode in fp64 and shared by the forward and the adjoint solve.
//    Preconditioner: A-DEF2 (Tang, Nabben, Vuik, Erlangga 2009), z = r + W E^-1 (W^T r - (M W)^T r), start vector
//    x0 = W E^-1 W^T b.  Unlike plain deflation it does not rely on r staying orthogonal to W, which fp32 does not
//    keep (measured: deflated CG stagnates / diverges on the trained system once W^T r has drifted to 1e-6 |b|).
//    Measured on the dumped systems (fp32, tol 1e-6): 20 -> 12 iterations at initialisation, 48 -> 15 (forward) and
//    49 -> 13 (adjoint) after 150 training steps.
//    q = M p by recurrence, q_new = M r + (M W) mu + beta q: the SpMV gathers ONE float4 per entry (r), not two.
//
//    Two launches per iteration, no grid barrier, no host sync, fixed summation orders:
//      S (1 row per wave) p = r + u mu[agg] + beta p ; q = M r + (M W) mu + beta q ; partial <p,q>
//      U (256 rows / wg)  alpha = rz / <p,q> ; x += alpha p ; r -= alpha q ; partial rr, W^T r, (M W)^T r ;
//                         the workgroup that delivers its partials LAST (ticket counter) reduces them: mu = E^-1 (t - t2),
//                         rz, beta, convergence test -- the step that was a one-workgroup launch of its own (R) at first:
//                         every launch on this dependent chain costs ~3 us of dispatch plus a ~2 us memory hop between
//                         XCDs (measured: R 7.5 us, U 7.5 us, S 6.4 us per iteration as three launches)
//    Why not one persistent kernel with a grid barrier: measured on MI355X (tools/probe/grid_barrier.hip) a
//    device-wide barrier whose workgroups exchange data needs agent-scope release/acquire fences, i.e. an L2
//    write-back + invalidate per workgroup, and costs 3 us at 32 workgroups, 7 us at 128 and 13 us at 256.
// ---------------------------------------------------------------------------
#define HG_MAX_ITER 1022
#define HG_M 64                 // aggregates = coarse dimensions (one per lane)
#define HG_UROWS 256            // rows per workgroup of the vector kernels (U, init): one per thread
#define HG_EBLOCKS 16           // workgroups (= partial matrices) of the E = W^T M W accumulation, two waves each
#define HG_PART (4 + 2 * HG_M * HG_NC)  // floats of one workgroup's partial: rr[4], t[HG_M][4], t2[HG_M][4]
static_assert(HG_M == 64 && HG_M * HG_NC == 256, "the CG kernels map (aggregate, column) onto 256 threads, aggregate = lane");
struct CgState {            // device memory
  int done;                 // set once every column has converged; later S / U launches return at once
  int iters;                // iterations performed when `done` was set (or so far)
  unsigned ticket;          // workgroups of the vector kernels that have delivered their partials (zeroed per solve)
  int pad_;
  float bb[HG_NC];          // ||b||^2
  float rz[HG_NC];          // <r, z> of the current iteration
  float beta[HG_NC];
  float mu[HG_M * HG_NC];   // E^-1 (W^T r - (MW)^T r) of the current iteration; c0 = E^-1 W^T b before the first
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
  for (int q0 = threadIdx.x; q0 < nblk; q0 += 8 * 256) {  // 8 loads in flight per trip (blockDim.x == 256), fixed add order
    float4 p[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = q0 + 256 * u;
      const bool ok = q < nblk;
      p[u] = part[min(q, nblk - 1)];
      p[u].x = r3d_keep(p[u].x, ok); p[u].y = r3d_keep(p[u].y, ok); p[u].z = r3d_keep(p[u].z, ok); p[u].w = r3d_keep(p[u].w, ok);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { a.x += p[u].x; a.y += p[u].y; a.z += p[u].z; a.w += p[u].w; }
  }
  return block_sum4(a, sm);
}

static __device__ __forceinline__ float f4_get(const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

// ---- 5a. aggregates: node -> nearest of HG_M seed prototypes (squared L2 in feature space, ties to the lower seed).
// One wave per node, LANE = SEED: the seeds sit in LDS with an odd row pitch (conflict-free column reads), the node's
// row is read through LDS as a broadcast.  Not index-deciding for the result (any partition gives a valid coarse
// space), so the summation order is free.
#define HG_SEED_PITCH 257
__global__ __launch_bounds__(256) void r3d_cg_aggregate_kernel(const float* __restrict__ nodes, long ldn, int D,
                                                               const int* __restrict__ n_dev, const int* __restrict__ n_proto_dev,
                                                               int n_cap, int rows_per_wave, int* __restrict__ agg, HgEp st) {
  extern __shared__ float smem[];           // seeds [HG_M][pitch] + 4 node rows [D]
  {
    const int ep = blockIdx.y;
    nodes += (long)ep * st.nodes * ldn; HG_AT(n_dev, st.desc); HG_AT(n_proto_dev, st.desc); HG_WS(agg);
  }
  const int pitch = D | 1;
  float* seeds = smem;
  float* xrow = smem + HG_M * pitch + (threadIdx.x >> 6) * D;
  const int n = min(*n_dev, n_cap);
  const int n_proto = max(1, min(*n_proto_dev, n));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // seed rows -> LDS: 16 seeds per pass, thread = (seed, 16-float4 stripe): up to 4 independent float4 loads per thread
  // per pass are in flight together (a one-load-per-trip loop is a chain of 48 memory round trips: measured 30 us)
  {
    const int D4 = D >> 2;  // D % 4 == 0 (checked by the entry point through ldn / float4 rows)
    for (int s0 = 0; s0 < HG_M; s0 += 16) {
      const int sd = s0 + (threadIdx.x >> 4), l16 = threadIdx.x & 15;
      const int src = (int)(((long)sd * (n_proto - 1)) / (HG_M - 1));
      const float4* row = reinterpret_cast<const float4*>(nodes + (long)src * ldn);
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = row[min(l16 + 16 * u, D4 - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c4 = l16 + 16 * u;
        if (c4 < D4) {
          float* dst = seeds + sd * pitch + 4 * c4;
          dst[0] = v[u].x; dst[1] = v[u].y; dst[2] = v[u].z; dst[3] = v[u].w;
        }
      }
    }
  }
  const int row0 = (blockIdx.x * 4 + w) * rows_per_wave;
  // the wave's node rows are requested together, in front of the seed staging barrier's wait (rows_per_wave <= 4, D <= 256)
  float xr[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int u = 0; u < 4; ++u)
      xr[k][u] = nodes[(long)min(row0 + min(k, rows_per_wave - 1), n - 1) * ldn + min(lane + 64 * u, D - 1)];
  __syncthreads();
  for (int k = 0; k < rows_per_wave; ++k) {
    const int i = row0 + k;
    if (i >= n) break;  // uniform over the wave
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (lane + 64 * u < D) xrow[lane + 64 * u] = k == 0 ? xr[0][u] : k == 1 ? xr[1][u] : k == 2 ? xr[2][u] : xr[3][u];
    float d2 = 0.f;
    const float* sp = seeds + lane * pitch;
    for (int c0 = 0; c0 < D; c0 += 16) {  // 16 + 16 LDS reads in flight, then the fma chain
      float xs_[16], ss_[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) { xs_[c] = xrow[min(c0 + c, D - 1)]; ss_[c] = sp[min(c0 + c, D - 1)]; }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float df = c0 + c < D ? xs_[c] - ss_[c] : 0.f;
        d2 = __builtin_fmaf(df, df, d2);
      }
    }
    // arg-min over the lanes on 64-bit keys (distance bits are order preserving for non-negative floats)
    unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(key, o);
      key = other < key ? other : key;
    }
    if (lane == 0) agg[i] = (int)(key & 63u);
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- 5b. MW = W - alpha S W, row by row.  (S W)[i][a] = sum over the row's entries of S_ij u_j [agg_j == a]: the
// wave writes (agg_j, S_ij u_j) pairs of a chunk of entries to LDS, then lane a walks the chunk and adds the pairs of
// its aggregate in entry order (deterministic; all lanes read the same LDS word = a broadcast).
__global__ __launch_bounds__(256) void r3d_cg_mw_kernel(const int* __restrict__ row_ptr, const hg_col_t* __restrict__ col,
                                                        const float* __restrict__ val, const float* __restrict__ dinv,
                                                        const int* __restrict__ agg, const int* __restrict__ n_dev, int n_cap,
                                                        float alpha_lp, int rows_per_wave, float* __restrict__ MW, HgEp st) {
  __shared__ float2 pairs[4][256];
  {
    const int ep = blockIdx.y;
    HG_WS(row_ptr); HG_WS(col); HG_WS(val); HG_WS(dinv); HG_WS(agg); HG_AT(n_dev, st.desc); HG_WS(MW);
  }
  const int n = min(*n_dev, n_cap);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + w) * rows_per_wave;
  for (int k = 0; k < rows_per_wave; ++k) {
    const int i = row0 + k;
    if (i >= n_cap) break;
    float acc = 0.f;
    if (i < n) {
      const int rb = row_ptr[i], re = row_ptr[i + 1];
      for (int e0 = rb; e0 < re; e0 += 256) {
        const int cnt = min(256, re - e0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int t = lane + 64 * u;
          if (t < cnt) {
            const int j = col[e0 + t];
            pairs[w][t] = make_float2(__int_as_float(agg[j]), val[e0 + t] / dinv[j]);
          }
        }
        __builtin_amdgcn_wave_barrier();
        for (int t0 = 0; t0 < cnt; t0 += 16) {  // 16 broadcast reads in flight; entries beyond cnt are stale pairs: masked
          float2 pr[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) pr[t] = pairs[w][t0 + t];
#pragma unroll
          for (int t = 0; t < 16; ++t) acc += (__float_as_int(pr[t].x) == lane && t0 + t < cnt) ? pr[t].y : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
      }
      acc = (agg[i] == lane ? 1.f / dinv[i] : 0.f) - alpha_lp * acc;
    }
    MW[(long)i * HG_M + lane] = acc;  // rows >= n are zero
  }
}

// ---- 5c. E = W^T (M W): per workgroup of HG_UROWS rows a partial HG_M x HG_M matrix (lane = column b, one LDS
// accumulator matrix per wave, row = aggregate of the node), then the single workgroup below adds the partials in
// fixed order, symmetrises, and inverts in fp64 (Gauss-Jordan without pivoting: E is SPD; an EMPTY aggregate --
// duplicate seeds, or fewer prototypes than seeds -- has a zero row and column and gets a unit diagonal, its
// coefficient is then always zero).
__global__ __launch_bounds__(128) void r3d_cg_epart_kernel(const float* __restrict__ MW, const float* __restrict__ dinv,
                                                           const int* __restrict__ agg, const int* __restrict__ n_dev, int n_cap,
                                                           float* __restrict__ Epart /* [HG_EBLOCKS][HG_M][HG_M] */, HgEp st) {
  __shared__ float Ew[2][HG_M][HG_M];  // one accumulator matrix per wave (lane = column: conflict free), 32 KB
  {
    const int ep = blockIdx.y;
    HG_WS(MW); HG_WS(dinv); HG_WS(agg); HG_AT(n_dev, st.desc); HG_WS(Epart);
  }
  const int n = min(*n_dev, n_cap);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int a = 0; a < HG_M; ++a) Ew[w][a][lane] = 0.f;
  // rows are dealt to the 2 * HG_EBLOCKS waves in a fixed pattern; 8 rows of loads in flight per trip
  const int stride = 2 * HG_EBLOCKS;
  for (int i0 = blockIdx.x * 2 + w; i0 < n; i0 += 8 * stride) {
    float m[8], u[8];
    int a[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = min(i0 + t * stride, n - 1);
      m[t] = MW[(long)i * HG_M + lane]; u[t] = 1.f / dinv[i]; a[t] = agg[i];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (i0 + t * stride < n) Ew[w][a[t]][lane] += m[t] * u[t];
  }
  __syncthreads();
  float* out = Epart + (long)blockIdx.x * HG_M * HG_M;
  for (int e = threadIdx.x; e < HG_M * HG_M; e += 128) {
    const int a = e >> 6, b = e & 63;
    out[e] = Ew[0][a][b] + Ew[1][a][b];
  }
}

// E^-1 in fp64 by Gauss-Jordan elimination on [E | I].  Thread (row = tid >> 4, column group = 4 * (tid & 15)) keeps its
// 4 + 4 entries in registers for the whole elimination; per step the owners of pivot row k and of column k publish
// them through double-buffered LDS, so a step costs ONE workgroup barrier.
__global__ __launch_bounds__(1024) void r3d_cg_einv_kernel(const float* __restrict__ Epart, float* __restrict__ Einv, HgEp st) {
  __shared__ double S0[HG_M][HG_M + 1];
  {
    const int ep = blockIdx.x;
    HG_WS(Epart); HG_WS(Einv);
  }
  __shared__ double prow[2][2 * HG_M];  // pivot row of [A | B], already scaled by 1 / pivot
  __shared__ double pcol[2][HG_M];      // column k of A
  const int tid = threadIdx.x;
  const int row = tid >> 4, cg = (tid & 15) * 4;
  // sum of the partial matrices (fixed order), symmetrised through LDS
  double a[4], b[4];
  {
    float v[HG_EBLOCKS][4];
#pragma unroll
    for (int q = 0; q < HG_EBLOCKS; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(Epart + (long)q * HG_M * HG_M + row * HG_M + cg);
      v[q][0] = t.x; v[q][1] = t.y; v[q][2] = t.z; v[q][3] = t.w;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < HG_EBLOCKS; ++q) s += (double)v[q][c];
      S0[row][cg + c] = s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double v = 0.5 * (S0[row][cg + c] + S0[cg + c][row]);
    if (row == cg + c && !(v > 0.0)) v = 1.0;  // empty aggregate: zero row and column, unit diagonal
    a[c] = v;
    b[c] = row == cg + c ? 1.0 : 0.0;
  }
  // owners of pivot row k (16 threads of one wave) and of column k (the threads whose column group holds k) publish
  auto publish = [&](int k, int buf) {
    const int kc = k & 3;
    const double sel = kc == 0 ? a[0] : kc == 1 ? a[1] : kc == 2 ? a[2] : a[3];
    if (row == k) {
      const double piv = 1.0 / __shfl(sel, (k * 16 + (k >> 2)) & 63, 64);  // A[k][k] sits in the row's thread of group k / 4
#pragma unroll
      for (int c = 0; c < 4; ++c) { prow[buf][cg + c] = a[c] * piv; prow[buf][HG_M + cg + c] = b[c] * piv; }
    }
    if (cg == (k & ~3)) pcol[buf][row] = sel;
  };
  publish(0, 0);
  __syncthreads();
  for (int k = 0; k < HG_M; ++k) {
    const int buf = k & 1;
    double pa[4], pb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { pa[c] = prow[buf][cg + c]; pb[c] = prow[buf][HG_M + cg + c]; }
    if (row == k) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { a[c] = pa[c]; b[c] = pb[c]; }
    } else {
      const double f = pcol[buf][row];
#pragma unroll
      for (int c = 0; c < 4; ++c) { a[c] -= f * pa[c]; b[c] -= f * pb[c]; }
    }
    if (k + 1 < HG_M) publish(k + 1, buf ^ 1);
    __syncthreads();
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; ++c) S0[row][cg + c] = b[c];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; ++c) Einv[row * HG_M + cg + c] = (float)(0.5 * (S0[row][cg + c] + S0[cg + c][row]));  // the
  // inverse is computed in fp64 and stored in fp32: it is a preconditioner (its rounding changes no solution), and the
  // vector kernels hold a column of it in registers
}

// ---- 5d. per-workgroup partials of a residual block: rr, t = W^T r, t2 = (M W)^T r.
// rs / us / ag: this workgroup's HG_UROWS = 256 rows of r, u = D^1/2, aggregate (LDS).  Wave w owns rows 64 w .. 64 w + 63,
// lane a owns aggregate a for all four columns; the wave's 64 values of column a of M W sit in REGISTERS, loaded by the
// caller before anything else in the kernel (64 independent loads in flight: one memory round trip instead of a walk
// -- a kernel of 18 workgroups on a dependent chain is pure latency: measured 8-15 us per launch for per-row / LDS-tile
// walks against ~3 us).  The four waves' partials meet in LDS and are added in wave order (deterministic).
static __device__ __forceinline__ void cg_store_sc1(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static __device__ __forceinline__ float cg_load_sc1(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

static __device__ __forceinline__ void cg_load_mw_column(const float* __restrict__ MW, int row0, int n_cap, float (&m)[64]) {
  const int a = threadIdx.x & 63, w = threadIdx.x >> 6;
  // rows < n_cap exist and rows in [n, n_cap) are zero (r3d_cg_mw_kernel); beyond n_cap the index is clamped
#pragma unroll
  for (int k = 0; k < 64; ++k) m[k] = MW[(long)min(row0 + 64 * w + k, n_cap - 1) * HG_M + a];
}

static __device__ __forceinline__ void cg_block_partials(const float4* rs, const float* us, const int* ag, int row0, int n,
                                                         const float (&m)[64], bool with_t2, float* __restrict__ part,
                                                         float4* sm, float4* wpart /* [2][4][HG_M] */) {
  const int a = threadIdx.x & 63, w = threadIdx.x >> 6;
  float4 t = f4_zero(), t2 = f4_zero();
  // rows >= n carry r = 0 and u = 0 in LDS (the callers zero them), so no row guard is needed
#pragma unroll 16
  for (int k = 0; k < 64; ++k) {
    const float4 rv = rs[64 * w + k];
    const float uw = ag[64 * w + k] == a ? us[64 * w + k] : 0.f;
    t.x = __builtin_fmaf(uw, rv.x, t.x); t.y = __builtin_fmaf(uw, rv.y, t.y);
    t.z = __builtin_fmaf(uw, rv.z, t.z); t.w = __builtin_fmaf(uw, rv.w, t.w);
    if (with_t2) {
      t2.x = __builtin_fmaf(m[k], rv.x, t2.x); t2.y = __builtin_fmaf(m[k], rv.y, t2.y);
      t2.z = __builtin_fmaf(m[k], rv.z, t2.z); t2.w = __builtin_fmaf(m[k], rv.w, t2.w);
    }
  }
  wpart[w * HG_M + a] = t;
  wpart[4 * HG_M + w * HG_M + a] = t2;
  float4 sq = f4_zero();
  if ((int)threadIdx.x < n - row0) {
    const float4 v = rs[threadIdx.x];
    sq = make_float4(v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w);
  }
  sq = block_sum4(sq, sm);  // its barriers also publish wpart
  // the partials are handed to another workgroup INSIDE this launch (cg_delivered_last): write-through (sc1) stores,
  // which need no release fence in front of the ticket
  if (threadIdx.x < 4) cg_store_sc1(part + threadIdx.x, f4_get(sq, threadIdx.x));
  {
    const int c = w;  // thread (a, c = w) adds the four waves' partials of entry [a][c]
    float st = 0.f, st2 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      st += f4_get(wpart[q * HG_M + a], c);
      st2 += f4_get(wpart[4 * HG_M + q * HG_M + a], c);
    }
    cg_store_sc1(part + 4 + a * HG_NC + c, st);
    cg_store_sc1(part + 4 + HG_M * HG_NC + a * HG_NC + c, st2);
  }
}

// The reduction step between U and S, run by the workgroup that delivered its partials last.
// phase 0: c0 = E^-1 W^T b, bb (partials of b itself); phase 1: mu, rz, beta, convergence test before iteration `it`.
// ev: column a = tid & 63 of E^-1 in registers (loaded by the caller at kernel start).
static __device__ __forceinline__ void cg_reduce_step(const float* __restrict__ part, int nblk, const float* __restrict__ Einv,
                                                      int phase, int it, float tol2, CgState* __restrict__ cg, double* d_s,
                                                      float* t_s, float* rr_s) {
  const int a = threadIdx.x & 63, c = threadIdx.x >> 6;
  // column a of E^-1 (symmetric: read as row-strided, coalesced over the lanes), requested together with the partials.
  // (Measured with phase stamps: prefetching it in EVERY workgroup at kernel start put 64 more loads in front of the
  // alpha-dependent work of all 18 workgroups -- the 63-deep vmcnt queue turns that into a third load round.)
  float ev[HG_M];
#pragma unroll
  for (int b = 0; b < HG_M; ++b) ev[b] = Einv[b * HG_M + a];
  const float rz_old = cg->rz[c];
  const float bbv = cg->bb[threadIdx.x & 3];
  float t = 0.f, t2 = 0.f, rrp = 0.f;
  const float* pa = part + 4 + a * HG_NC + c;
  for (int b0 = 0; b0 < nblk; b0 += 32) {  // 32 workgroups' partials in flight per trip (ONE trip at workload S: each
    // trip is a ~2 us round trip to the memory side); the add order stays fixed
    float tv[32], t2v[32], rv[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const long b = min(b0 + u, nblk - 1);
      const bool ok = b0 + u < nblk;
      // sc1 loads: the partials were stored write-through by other workgroups of THIS launch (cg_delivered_last)
      tv[u] = r3d_keep(cg_load_sc1(pa + b * HG_PART), ok);
      t2v[u] = r3d_keep(cg_load_sc1(pa + b * HG_PART + HG_M * HG_NC), ok);
      rv[u] = r3d_keep(cg_load_sc1(part + b * HG_PART + (threadIdx.x & 3)), ok);
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) { t += tv[u]; t2 += t2v[u]; rrp += rv[u]; }
  }
  if (threadIdx.x < HG_NC) rr_s[threadIdx.x] = rrp;
  d_s[a * HG_NC + c] = phase == 0 ? (double)t : (double)t - (double)t2;
  t_s[a * HG_NC + c] = t;
  __syncthreads();
  if (phase == 1) {
    const bool conv = rr_s[0] <= tol2 * __shfl(bbv, 0) && rr_s[1] <= tol2 * __shfl(bbv, 1) &&
                      rr_s[2] <= tol2 * __shfl(bbv, 2) && rr_s[3] <= tol2 * __shfl(bbv, 3);
    if (conv) {  // uniform over the workgroup
      if (threadIdx.x == 0) { cg->done = 1; cg->iters = it; }
      return;
    }
  }
  double m4[4] = {0.0, 0.0, 0.0, 0.0};  // four independent fp64 chains (a single 64-long chain is 64 dependent DFMAs)
#pragma unroll
  for (int b = 0; b < HG_M; b += 4) {
#pragma unroll
    for (int e = 0; e < 4; ++e) m4[e] += (double)ev[b + e] * d_s[(b + e) * HG_NC + c];
  }
  const float mu = (float)((m4[0] + m4[1]) + (m4[2] + m4[3]));
  float tm = t_s[a * HG_NC + c] * mu;
  tm = r3d_wave_sum(tm);  // one wave = one column c
  cg->mu[a * HG_NC + c] = mu;
  if (a == 0) {
    if (phase == 0) {
      cg->bb[c] = rr_s[c];
    } else {
      const float rz = rr_s[c] + tm;
      cg->beta[c] = (it > 0 && rz_old > 0.f) ? rz / rz_old : 0.f;
      cg->rz[c] = rz;
    }
  }
  if (threadIdx.x == 0) {
    if (phase == 0) { cg->done = 0; cg->iters = 0; }
    else cg->iters = it;
  }
}

// Was this workgroup the last of its launch to deliver its partials?  The in-launch hand-off of the CDNA guide
// (cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "visibility", counter form, first table row): the partials
// are stored write-through (sc1, cg_block_partials) -> EVERY storing wave drains its stores (s_waitcnt vmcnt(0)) ->
// workgroup barrier -> lane 0: relaxed agent-scope ticket add (no release fence: nothing dirty to write back); the
// workgroup whose add came last reads the partials with sc1 loads ONLY (cg_reduce_step), after a workgroup barrier its
// adding wave joins -- so no acquire (cache invalidate) is needed either.  One workgroup per CU (18 - 128 workgroups),
// hipMalloc'ed memory, 4-byte stores and loads: the measured-valid form.  (The XCDs' L2 caches are not coherent with each
// other; `__threadfence()` in every workgroup is the measured-slower form.)  Every launch that reaches this point adds
// exactly `nblk` tickets, so "last" is ticket % nblk == nblk - 1 without resetting the counter inside a solve.
static __device__ __forceinline__ bool cg_delivered_last(CgState* __restrict__ cg, int nblk, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(&cg->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = (t + 1u) % (unsigned)nblk == 0u;  // (the returned value is used: the add has returned before any load below)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler only: no load moves above the add
  }
  __syncthreads();
  return *flag != 0;
}

// mode 0: partials of b itself (rr = ||b||^2, t = W^T b).  mode 1: x0 = W c0, r0 = b - (M W) c0, p = q = 0, partials
// of r0.  c0 = cg->mu.
__global__ __launch_bounds__(256) void r3d_cg_init_kernel(const float4* __restrict__ B, const float* __restrict__ dinv,
                                                          const int* __restrict__ agg, const float* __restrict__ MW,
                                                          const int* __restrict__ n_dev, int n_cap, int mode,
                                                          float4* __restrict__ x, float4* __restrict__ r,
                                                          float4* __restrict__ p, float4* __restrict__ q,
                                                          float* __restrict__ part, const float* __restrict__ Einv, float tol2,
                                                          CgState* __restrict__ cg, HgEp st, long b_stride, long x_stride) {
  {
    const int ep = blockIdx.y;  // B, x: float4 rows (right-hand side and solution of the system)
    HG_AT(B, b_stride); HG_WS(dinv); HG_WS(agg); HG_WS(MW); HG_AT(n_dev, st.desc); HG_AT(x, x_stride); HG_WS(r); HG_WS(p); HG_WS(q);
    HG_WS(part); HG_WS(Einv); HG_WS(cg);
  }
  __shared__ float4 sm[4];
  __shared__ float4 rs[HG_UROWS];
  __shared__ float us[HG_UROWS];
  __shared__ int ag[HG_UROWS];
  __shared__ __attribute__((aligned(16))) float mu_s[HG_M * HG_NC];
  __shared__ float4 wpart[2 * 4 * HG_M];
  __shared__ double d_s[HG_M * HG_NC];
  __shared__ float t_s[HG_M * HG_NC];
  __shared__ float rr_s[HG_NC];
  __shared__ int last_s;
  __shared__ float tile_s[4 * 32 * 65];
  __shared__ float4 dot_s[HG_UROWS];
  const int row0 = blockIdx.x * HG_UROWS;
  float m[64];
  if (mode == 1) cg_load_mw_column(MW, row0, n_cap, m);
  const int n = min(*n_dev, n_cap);
  if (mode == 1) mu_s[threadIdx.x] = cg->mu[threadIdx.x];
  __syncthreads();
  if (mode == 1) {
    // dot_s[row] = (M W)[row] . c0: the wave's 64 rows x 64 columns sit in registers with the COLUMN on the lane; two
    // half tiles of 32 rows go through LDS (pitch 65: conflict free both ways), lane l then owns row l & 31 over the
    // column half l >> 5, and one shuffle joins the halves.  (A thread-per-row walk over global memory reads 64 cache
    // lines per load instruction: measured 27 us per launch.)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* tl = tile_s + w * (32 * 65);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int k = 0; k < 32; ++k) tl[k * 65 + lane] = m[32 * h + k];
      __builtin_amdgcn_wave_barrier();
      const int rk = lane & 31, a0 = 32 * (lane >> 5);
      float4 acc = f4_zero();
#pragma unroll 8
      for (int a = 0; a < 32; ++a) {
        const float mv = tl[rk * 65 + a0 + a];
        const float4 cv = *reinterpret_cast<const float4*>(&mu_s[(a0 + a) * HG_NC]);
        acc.x = __builtin_fmaf(mv, cv.x, acc.x); acc.y = __builtin_fmaf(mv, cv.y, acc.y);
        acc.z = __builtin_fmaf(mv, cv.z, acc.z); acc.w = __builtin_fmaf(mv, cv.w, acc.w);
      }
      acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32);
      acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
      if (lane < 32) dot_s[64 * w + 32 * h + lane] = acc;
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  {
    const int i = row0 + threadIdx.x;
    float4 rv = f4_zero();
    float u = 0.f;
    int a = 0;
    if (i < n) {
      u = 1.f / dinv[i];
      a = agg[i];
      rv = B[i];
      if (mode == 1) {
        const float4 s = dot_s[threadIdx.x];  // (M W)[i] . c0, computed by the wave below
        rv = make_float4(rv.x - s.x, rv.y - s.y, rv.z - s.z, rv.w - s.w);
        x[i] = make_float4(u * mu_s[a * HG_NC + 0], u * mu_s[a * HG_NC + 1], u * mu_s[a * HG_NC + 2], u * mu_s[a * HG_NC + 3]);
        r[i] = rv;
      }
    }
    if (mode == 1 && i < n_cap) {
      p[i] = f4_zero(); q[i] = f4_zero();
      if (i >= n) { x[i] = f4_zero(); r[i] = f4_zero(); }
    }
    rs[threadIdx.x] = rv; us[threadIdx.x] = u; ag[threadIdx.x] = a;
  }
  __syncthreads();
  cg_block_partials(rs, us, ag, row0, n, m, mode == 1, part + (long)blockIdx.x * HG_PART, sm, wpart);
  if (cg_delivered_last(cg, gridDim.x, &last_s)) cg_reduce_step(part, gridDim.x, Einv, mode, 0, tol2, cg, d_s, t_s, rr_s);
}

// S: p = r + u mu[agg] + beta p ; q = (r - alpha S r) + (M W) mu + beta q ; partial <p, q>.
// One matrix row per wave taken in ONE trip of the chain column index -> gather (6 entries per lane in flight).
__global__ __launch_bounds__(256) void r3d_cg_spmv_kernel(
    const int* __restrict__ row_ptr, const hg_col_t* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ dinv, const int* __restrict__ agg, const float* __restrict__ MW,
    const int* __restrict__ n_dev, int n_cap, float alpha_lp, int it, int rows_per_block,
    const float4* __restrict__ r, float4* __restrict__ p, float4* __restrict__ q,
    float4* __restrict__ part_pq, const CgState* __restrict__ cg, HgEp st) {
  __shared__ float4 mu_s[HG_M];
  __shared__ float4 wsum[4];
  {
    const int ep = blockIdx.y;
    HG_WS(cg);
    if (cg->done) return;  // this system has converged: the launch goes on for the others
    HG_WS(row_ptr); HG_WS(col); HG_WS(val); HG_WS(dinv); HG_WS(agg); HG_WS(MW); HG_AT(n_dev, st.desc); HG_WS(r); HG_WS(p);
    HG_WS(q); HG_WS(part_pq);
  }
  const int n = min(*n_dev, n_cap);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x < HG_M) mu_s[threadIdx.x] = reinterpret_cast<const float4*>(cg->mu)[threadIdx.x];
  const float4 beta = *reinterpret_cast<const float4*>(cg->beta);
  __syncthreads();
  const float4 mul = mu_s[lane];
  float4 acc_pq = f4_zero();
  const int row0 = blockIdx.x * rows_per_block + w * (rows_per_block / 4);
  for (int rr_i = 0; rr_i < rows_per_block / 4; ++rr_i) {
    const int i = row0 + rr_i;
    if (i >= n) break;
    float4 s = f4_zero();
    const int rb = row_ptr[i], re = row_ptr[i + 1];
    const float mw = MW[(long)i * HG_M + lane];
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
      float4 rj[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) rj[u] = r[jv[u]];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        s.x = __builtin_fmaf(av[u], rj[u].x, s.x); s.y = __builtin_fmaf(av[u], rj[u].y, s.y);
        s.z = __builtin_fmaf(av[u], rj[u].z, s.z); s.w = __builtin_fmaf(av[u], rj[u].w, s.w);
      }
    }
    // (M W) mu rides the same butterfly: per lane -alpha s + MW[i][lane] mu[lane]
    s.x = __builtin_fmaf(mw, mul.x, -alpha_lp * s.x); s.y = __builtin_fmaf(mw, mul.y, -alpha_lp * s.y);
    s.z = __builtin_fmaf(mw, mul.z, -alpha_lp * s.z); s.w = __builtin_fmaf(mw, mul.w, -alpha_lp * s.w);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s.x += __shfl_xor(s.x, o); s.y += __shfl_xor(s.y, o);
      s.z += __shfl_xor(s.z, o); s.w += __shfl_xor(s.w, o);
    }
    if (lane == 0) {
      const float4 ri = r[i], pi = p[i], qi = q[i];
      const float u = 1.f / dinv[i];
      const float4 m = mu_s[agg[i]];
      const float4 pn = make_float4((ri.x + u * m.x) + beta.x * pi.x, (ri.y + u * m.y) + beta.y * pi.y,
                                    (ri.z + u * m.z) + beta.z * pi.z, (ri.w + u * m.w) + beta.w * pi.w);
      const float4 qn = make_float4((ri.x + s.x) + beta.x * qi.x, (ri.y + s.y) + beta.y * qi.y,
                                    (ri.z + s.z) + beta.z * qi.z, (ri.w + s.w) + beta.w * qi.w);
      p[i] = pn;
      q[i] = qn;
      acc_pq.x += pn.x * qn.x; acc_pq.y += pn.y * qn.y; acc_pq.z += pn.z * qn.z; acc_pq.w += pn.w * qn.w;
    }
  }
  // fixed-order combine of the four waves' lane-0 partials
  if (lane == 0) wsum[w] = acc_pq;
  __syncthreads();
  if (threadIdx.x == 0) {
    float4 t = wsum[0];
    for (int q2 = 1; q2 < 4; ++q2) { t.x += wsum[q2].x; t.y += wsum[q2].y; t.z += wsum[q2].z; t.w += wsum[q2].w; }
    part_pq[blockIdx.x] = t;
  }
}

// S with the residual of the system in LDS (n x float4 <= SL_MAX_LDS).  Once a batch of systems streams its matrices
// from HBM, what is left in the way is the gather: every matrix entry fetches a 16-byte row of r through a 64-byte L2
// sector (4 x the traffic, ~14 TB/s of sector reads per iteration for 32 systems of workload S) -- the launch was bound
// by L2 -> CU bandwidth at 1.6 TB/s of algorithmic bytes.  Here a workgroup of 8 waves takes SL_ROWS consecutive rows of
// ONE system, stages that system's r in LDS once (70 KB at S: coalesced, 12 % of what the gathers read) and gathers
// from there; a wave walks 16 rows with the next row's column / value loads in flight behind the current row's
// gather, and the 16 row tails (p, q updates) run lane-parallel with coalesced loads and stores.  The arithmetic is
// that of r3d_cg_spmv_kernel bit for bit, INCLUDING the <p, q> partials (one per 4 rows, added in row order): which of
// the two kernels a solve runs on -- a matter of how many systems the launch holds -- changes no bit of its result.
//
// WHICH rows a wave takes is free (the whole r is in LDS), and it matters: the graph's hubs are the prototype nodes, the
// first ~200 rows of a system, and they grow with training (after 150 optimiser steps at S: mean row 273 entries,
// p99.9 2745; the 16 consecutive hub rows of one wave held 10 x the entries of an average wave and the iteration took
// 181 us instead of 107).  So the 4-row groups of the system are dealt cyclically to its workgroups (group c n_wg + b is
// the c-th of workgroup b) and the 128 rows of a workgroup cyclically to its 8 waves (local row 8 k + w is the k-th of
// wave w): consecutive hub rows land in different waves of different workgroups.  The four <p q> products of a group
// then sit in four waves; they meet in LDS and are added in row order as before.
#define SL_ROWS 128
#define SL_WAVES 8
#define SL_RPW (SL_ROWS / SL_WAVES)
#define SL_MAX_LDS (144 * 1024)
__global__ __launch_bounds__(64 * SL_WAVES) void r3d_cg_spmv_lds_kernel(
    const int* __restrict__ row_ptr, const hg_col_t* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ dinv, const int* __restrict__ agg, const float* __restrict__ MW,
    const int* __restrict__ n_dev, int n_cap, float alpha_lp, int it, int rows_per_block,
    const float4* __restrict__ r, float4* __restrict__ p, float4* __restrict__ q,
    float4* __restrict__ part_pq /* [ceil(n_cap / 4)] */, const CgState* __restrict__ cg, HgEp st) {
  extern __shared__ __attribute__((aligned(16))) float4 rs[];  // [n] the system's residual
  __shared__ float4 mu_s[HG_M];
  {
    const int ep = blockIdx.y;
    HG_WS(cg);
    if (cg->done) return;  // this system has converged: the launch goes on for the others
    HG_WS(row_ptr); HG_WS(col); HG_WS(val); HG_WS(dinv); HG_WS(agg); HG_WS(MW); HG_AT(n_dev, st.desc); HG_WS(r); HG_WS(p);
    HG_WS(q); HG_WS(part_pq);
  }
  __shared__ float4 pq_s[SL_ROWS];  // the <p q> products of the workgroup's rows, by local row
  const int n = min(*n_dev, n_cap);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n_wg = gridDim.x, bx = blockIdx.x;
  const int n_part = (n_cap + 3) / 4;  // partials of <p, q>: one per 4 rows, as r3d_cg_spmv_kernel with rows_per_block = 4
  // the k-th row of this wave: local row l = 8 k + w of the workgroup = member l & 3 of its group l >> 2
  auto row_of = [&](int k) {
    const int l = SL_WAVES * k + w;
    return 4 * ((l >> 2) * n_wg + bx) + (l & 3);
  };
  // (rows ascend with k, so the rows below n are a prefix of the wave's 16)
  int nrows = 0;
#pragma unroll
  for (int k = 0; k < SL_RPW; ++k) nrows += row_of(k) < n ? 1 : 0;  // uniform over the wave
  // this wave's row bounds: lane k holds those of its k-th row
  const int my_row = min(row_of(min(lane, SL_RPW - 1)), max(n - 1, 0));
  const int my_rb = row_ptr[my_row], my_re = row_ptr[my_row + 1];
  // stage r: 4 float4 per thread in flight
  for (int i0 = tid; i0 < n; i0 += 4 * 64 * SL_WAVES) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = r[min(i0 + u * 64 * SL_WAVES, n - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * 64 * SL_WAVES < n) rs[i0 + u * 64 * SL_WAVES] = v[u];
  }
  if (tid < HG_M) mu_s[tid] = reinterpret_cast<const float4*>(cg->mu)[tid];
  const float4 beta = *reinterpret_cast<const float4*>(cg->beta);
  __syncthreads();
  const float4 mul = mu_s[lane];
  int jv[6], jn[6];
  float av[6], an[6];
  float mw = 0.f, mwn = 0.f;
  auto issue = [&](int k, int (&jj)[6], float (&aa)[6], float& mm) {
    const int rb = __builtin_amdgcn_readlane(my_rb, k), re = __builtin_amdgcn_readlane(my_re, k);
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int e = rb + lane + 64 * u;
      const int ec = max(min(e, re - 1), 0);
      jj[u] = col[ec];
      aa[u] = r3d_keep(val[ec], e < re);
    }
    mm = MW[(long)row_of(k) * HG_M + lane];
  };
  // Rows go in two groups of 8.  A lane keeps its partial sums of the group's rows in registers and the 64 lanes'
  // partials meet in ONE transposed butterfly per group: the xor-32 step halves the rows a lane is responsible for (it
  // sends the other half to its partner), xor-16 and xor-8 halve again, and the last three steps run on the one row
  // left -- lanes 8 g .. 8 g + 7 end up with the total of row g.  Every row still sees the pairs (l, l ^ 32), (l, l ^ 16),
  // ... (l, l ^ 1) in this order, i.e. the additions of the plain butterfly of r3d_cg_spmv_kernel (x + y = y + x bit for
  // bit), at 5 cross-lane moves per row and column instead of 24: the moves go through the LDS crossbar, which the
  // gathers need (the first version of this kernel was bound by exactly that: 153 us per launch, as slow as gathering
  // from L2).
  float4 tot[2] = {f4_zero(), f4_zero()};  // lanes 8 g ..: row 8 h + g of this wave (h = group)
  if (nrows > 0) issue(0, jv, av, mw);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float4 acc[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int k = 8 * h + kk;
      acc[kk] = f4_zero();
      if (k < nrows) {  // uniform
        if (k + 1 < nrows) issue(k + 1, jn, an, mwn);
        const int rb = __builtin_amdgcn_readlane(my_rb, k), re = __builtin_amdgcn_readlane(my_re, k);
        float4 s = f4_zero();
        {
          float4 rj[6];
#pragma unroll
          for (int u = 0; u < 6; ++u) rj[u] = rs[jv[u]];
#pragma unroll
          for (int u = 0; u < 6; ++u) {
            s.x = __builtin_fmaf(av[u], rj[u].x, s.x); s.y = __builtin_fmaf(av[u], rj[u].y, s.y);
            s.z = __builtin_fmaf(av[u], rj[u].z, s.z); s.w = __builtin_fmaf(av[u], rj[u].w, s.w);
          }
        }
        // rows with more than 384 entries (uniform trip count).  (Prefetching the next 384 entries behind the current
        // gather made the whole kernel slower -- 141 -> 194 us per iteration in the trained state: the registers it takes
        // cost every row.)
        for (int e0 = rb + 384; e0 < re; e0 += 384) {
          int j2[6];
          float a2[6];
#pragma unroll
          for (int u = 0; u < 6; ++u) {
            const int e = e0 + lane + 64 * u;
            const int ec = min(e, re - 1);
            j2[u] = col[ec];
            a2[u] = r3d_keep(val[ec], e < re);
          }
#pragma unroll
          for (int u = 0; u < 6; ++u) {
            const float4 rj = rs[j2[u]];
            s.x = __builtin_fmaf(a2[u], rj.x, s.x); s.y = __builtin_fmaf(a2[u], rj.y, s.y);
            s.z = __builtin_fmaf(a2[u], rj.z, s.z); s.w = __builtin_fmaf(a2[u], rj.w, s.w);
          }
        }
        // (M W) mu rides the same butterfly: per lane -alpha s + MW[i][lane] mu[lane]
        s.x = __builtin_fmaf(mw, mul.x, -alpha_lp * s.x); s.y = __builtin_fmaf(mw, mul.y, -alpha_lp * s.y);
        s.z = __builtin_fmaf(mw, mul.z, -alpha_lp * s.z); s.w = __builtin_fmaf(mw, mul.w, -alpha_lp * s.w);
        acc[kk] = s;
#pragma unroll
        for (int u = 0; u < 6; ++u) { jv[u] = jn[u]; av[u] = an[u]; }
        mw = mwn;
      }
    }
    // transposed butterfly: 8 rows -> 4 -> 2 -> 1 row per lane, then the plain steps
    float4 a4[4], a2[2], a1;
    {
      const bool hi = (lane & 32) != 0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 keepv = hi ? acc[4 + t] : acc[t], sendv = hi ? acc[t] : acc[4 + t];
        a4[t].x = keepv.x + __shfl_xor(sendv.x, 32); a4[t].y = keepv.y + __shfl_xor(sendv.y, 32);
        a4[t].z = keepv.z + __shfl_xor(sendv.z, 32); a4[t].w = keepv.w + __shfl_xor(sendv.w, 32);
      }
    }
    {
      const bool hi = (lane & 16) != 0;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float4 keepv = hi ? a4[2 + t] : a4[t], sendv = hi ? a4[t] : a4[2 + t];
        a2[t].x = keepv.x + __shfl_xor(sendv.x, 16); a2[t].y = keepv.y + __shfl_xor(sendv.y, 16);
        a2[t].z = keepv.z + __shfl_xor(sendv.z, 16); a2[t].w = keepv.w + __shfl_xor(sendv.w, 16);
      }
    }
    {
      const bool hi = (lane & 8) != 0;
      const float4 keepv = hi ? a2[1] : a2[0], sendv = hi ? a2[0] : a2[1];
      a1.x = keepv.x + __shfl_xor(sendv.x, 8); a1.y = keepv.y + __shfl_xor(sendv.y, 8);
      a1.z = keepv.z + __shfl_xor(sendv.z, 8); a1.w = keepv.w + __shfl_xor(sendv.w, 8);
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
      a1.x += __shfl_xor(a1.x, o); a1.y += __shfl_xor(a1.y, o);
      a1.z += __shfl_xor(a1.z, o); a1.w += __shfl_xor(a1.w, o);
    }
    tot[h] = a1;
  }
  // the wave's row tails: lane 8 g + h takes row 8 h + g (16 lanes busy; rows of a group sit 8 lanes apart)
  float4 acc_pq = f4_zero();
  const int trow = 8 * (lane & 7) + (lane >> 3);  // (meaningful for lane & 7 < 2)
  if ((lane & 7) < 2 && trow < nrows) {
    const float4 keep = (lane & 1) ? tot[1] : tot[0];
    const int i = row_of(trow);
    const float4 ri = rs[i], pi = p[i], qi = q[i];
    const float u = 1.f / dinv[i];
    const float4 m = mu_s[agg[i]];
    const float4 pn = make_float4((ri.x + u * m.x) + beta.x * pi.x, (ri.y + u * m.y) + beta.y * pi.y,
                                  (ri.z + u * m.z) + beta.z * pi.z, (ri.w + u * m.w) + beta.w * pi.w);
    const float4 qn = make_float4((ri.x + keep.x) + beta.x * qi.x, (ri.y + keep.y) + beta.y * qi.y,
                                  (ri.z + keep.z) + beta.z * qi.z, (ri.w + keep.w) + beta.w * qi.w);
    p[i] = pn;
    q[i] = qn;
    acc_pq = make_float4(pn.x * qn.x, pn.y * qn.y, pn.z * qn.z, pn.w * qn.w);
  }
  // one partial per 4 rows, ((r0 + r1) + r2) + r3: the order in which r3d_cg_spmv_kernel adds its four waves.  The rows
  // of a group are the k-th rows of four neighbouring waves: through LDS (rows beyond n contribute the zero they hold).
  if ((lane & 7) < 2) pq_s[SL_WAVES * trow + w] = acc_pq;
  __syncthreads();
  if (tid < SL_ROWS / 4) {
    const float4 a = pq_s[4 * tid], b2 = pq_s[4 * tid + 1], c2 = pq_s[4 * tid + 2], d2 = pq_s[4 * tid + 3];
    const float4 t = make_float4(((a.x + b2.x) + c2.x) + d2.x, ((a.y + b2.y) + c2.y) + d2.y, ((a.z + b2.z) + c2.z) + d2.z,
                                 ((a.w + b2.w) + c2.w) + d2.w);
    const int gidx = tid * n_wg + bx;
    if (gidx < n_part) part_pq[gidx] = t;
  }
}

#ifdef CG_STAMPS  // phase stamps of the update kernel (tools/cg_stamps.py; never in the product build)
__device__ unsigned long long g_cg_dbg[32];
#define CSTAMP(i) do { if (it == 3 && threadIdx.x == 0 && blockIdx.x == 0) g_cg_dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define CSTAMP_LAST(i) do { if (it == 3 && threadIdx.x == 0) g_cg_dbg[16 + i] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int r3d_cg_debug_read(unsigned long long* out32) {
  return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_cg_dbg), sizeof(g_cg_dbg)) == hipSuccess ? 0 : 1;
}
#else
#define CSTAMP(i)
#define CSTAMP_LAST(i)
#endif

// U: alpha = rz / <p,q> ; x += alpha p ; r -= alpha q ; partials of the new residual.
// 18 workgroups at workload S: pure latency, so every load that does not depend on alpha is issued first.
__global__ __launch_bounds__(256) void r3d_cg_update_kernel(
    const int* __restrict__ n_dev, int n_cap, int it, int nblk_pq, const float* __restrict__ dinv, const int* __restrict__ agg,
    const float* __restrict__ MW, const float4* __restrict__ p, const float4* __restrict__ q, float4* __restrict__ x,
    float4* __restrict__ r, const float4* __restrict__ part_pq, float* __restrict__ part, const float* __restrict__ Einv,
    float tol2, CgState* __restrict__ cg, HgEp st, long x_stride) {
  __shared__ float4 sm[4];
  __shared__ float4 rs[HG_UROWS];
  __shared__ float us[HG_UROWS];
  __shared__ int ag[HG_UROWS];
  __shared__ float4 wpart[2 * 4 * HG_M];
  __shared__ double d_s[HG_M * HG_NC];
  __shared__ float t_s[HG_M * HG_NC];
  __shared__ float rr_s[HG_NC];
  __shared__ int last_s;
  {
    const int ep = blockIdx.y;
    HG_AT(n_dev, st.desc); HG_WS(dinv); HG_WS(agg); HG_WS(MW); HG_WS(p); HG_WS(q); HG_AT(x, x_stride); HG_WS(r); HG_WS(part_pq);
    HG_WS(part); HG_WS(Einv); HG_WS(cg);
  }
  const int row0 = blockIdx.x * HG_UROWS;
  const int i = row0 + threadIdx.x;
  const int ic = min(i, n_cap - 1);
  float m[64];
  CSTAMP(0);
  cg_load_mw_column(MW, row0, n_cap, m);
  const float4 pi = p[ic], qi = q[ic];
  float4 xi = x[ic], ri = r[ic];
  const float dv = dinv[ic];
  const int av = agg[ic];
  const int done = cg->done;
  const float4 rz = *reinterpret_cast<const float4*>(cg->rz);
  const int n = min(*n_dev, n_cap);
  const float4 pq = reduce_partials(part_pq, nblk_pq, sm);
  CSTAMP(1);
  if (done) return;  // uniform
  float4 al;
  al.x = pq.x > 0.f ? rz.x / pq.x : 0.f;
  al.y = pq.y > 0.f ? rz.y / pq.y : 0.f;
  al.z = pq.z > 0.f ? rz.z / pq.z : 0.f;
  al.w = pq.w > 0.f ? rz.w / pq.w : 0.f;
  float u = 0.f;
  int a = 0;
  if (i < n) {
    xi.x += al.x * pi.x; xi.y += al.y * pi.y; xi.z += al.z * pi.z; xi.w += al.w * pi.w;
    ri.x -= al.x * qi.x; ri.y -= al.y * qi.y; ri.z -= al.z * qi.z; ri.w -= al.w * qi.w;
    x[i] = xi; r[i] = ri;
    u = 1.f / dv;
    a = av;
  } else {
    ri = f4_zero();
  }
  rs[threadIdx.x] = ri; us[threadIdx.x] = u; ag[threadIdx.x] = a;
  __syncthreads();
  CSTAMP(2);
  cg_block_partials(rs, us, ag, row0, n, m, true, part + (long)blockIdx.x * HG_PART, sm, wpart);
  CSTAMP(3);
  // the convergence test / coefficients of iteration it + 1
  const bool last = cg_delivered_last(cg, gridDim.x, &last_s);
  CSTAMP(4);
  if (last) {
    CSTAMP_LAST(0);
    cg_reduce_step(part, gridDim.x, Einv, 1, it + 1, tol2, cg, d_s, t_s, rr_s);
    CSTAMP_LAST(1);
  }
}

// ---------------------------------------------------------------------------
// 6. query logits (mpti.py:558-559) + cross entropy (mpti.py:778-781)
// ---------------------------------------------------------------------------
// More than 4 classes (n_way > 3): Z comes as two planes of 4 columns, classes 4 .. 7 in Z2 (the label propagation is
// column-wise independent and is solved plane after plane).
__global__ __launch_bounds__(1024) void r3d_logits_ce_kernel(const float4* __restrict__ Z, const float4* __restrict__ Z2,
                                                             const int* __restrict__ desc_nproto, int n_q, int N, int n_classes,
                                                             const long long* __restrict__ labels,
                                                             float* __restrict__ logits /* (n_q, n_classes, N) */,
                                                             float* __restrict__ loss_out, int* __restrict__ pred_out, HgEp st) {
  __shared__ float red[16];
  {
    const int ep = blockIdx.x;
    HG_AT(Z, st.z); HG_AT(desc_nproto, st.desc); HG_AT(logits, st.logits);
    if (Z2) HG_AT(Z2, st.z);
    if (labels) HG_AT(labels, st.labels);
    if (loss_out) HG_AT(loss_out, st.loss);
    if (pred_out) HG_AT(pred_out, st.pred);
  }
  const int n_proto = *desc_nproto;
  float acc = 0.f;
  for (int e = threadIdx.x; e < n_q * N; e += blockDim.x) {
    const int qi = e / N, p = e - qi * N;
    const float4 z = Z[n_proto + e];
    const float4 z2 = Z2 ? Z2[n_proto + e] : f4_zero();
    const float zv[8] = {z.x, z.y, z.z, z.w, z2.x, z2.y, z2.z, z2.w};
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
// scratch words for r3d_label_propagate: bitmaps, CSR, coarse space, CG vectors
struct LpWs {
  unsigned *outb, *sym;
  int *row_len, *row_ptr, *agg;
  hg_col_t* col;
  float *val, *dinv, *wdir, *MW, *Epart, *part;
  float* Einv;
  float4 *r, *p, *q, *part_pq;
  CgState* cg;
  long words, total;
};

static inline long hg_vblocks(int n_cap) { return r3d_cdiv(n_cap, HG_UROWS); }

static LpWs lp_carve(int32_t* ws, int n_cap, int kp1) {
  LpWs L;
  L.words = (n_cap + 31) / 32;
  const long nnz_cap = 2L * n_cap * (kp1 - 1);
  int32_t* wp = ws;
  auto align4 = [&]() { wp += (4 - ((wp - ws) & 3)) & 3; };  // 16-byte alignment (ws itself must be 16-B aligned)
  L.outb = (unsigned*)wp; wp += n_cap * L.words;
  L.sym = (unsigned*)wp; wp += n_cap * L.words;
  L.row_len = wp; wp += n_cap + 8;
  L.row_ptr = wp; wp += n_cap + 8;
  L.col = (hg_col_t*)wp; wp += (nnz_cap + 1) / 2;
  L.val = (float*)wp; wp += nnz_cap;
  L.wdir = (float*)wp; wp += 2 * nnz_cap;
  L.dinv = (float*)wp; wp += n_cap;
  L.agg = wp; wp += n_cap;
  align4();
  L.MW = (float*)wp; wp += (long)n_cap * HG_M;
  L.Epart = (float*)wp; wp += (long)HG_EBLOCKS * HG_M * HG_M;
  align4();
  L.Einv = (float*)wp; wp += (long)HG_M * HG_M;
  L.part = (float*)wp; wp += hg_vblocks(n_cap) * HG_PART;
  align4();
  L.r = (float4*)wp; wp += 4L * n_cap;
  L.p = (float4*)wp; wp += 4L * n_cap;
  L.q = (float4*)wp; wp += 4L * n_cap;
  L.part_pq = (float4*)wp; wp += 4L * HG_MAX_PART;
  L.cg = (CgState*)wp; wp += (sizeof(CgState) + 3) / 4;
  L.total = (wp - ws) + 64;
  return L;
}

extern "C" long r3d_lp_ws_words(int n_cap, int kp1) { return lp_carve(nullptr, n_cap, kp1).total; }

// word offsets (int32 units from ws) of what tests / tools read back: row_ptr, col (uint16 entries!), val, dinv, agg, CgState
extern "C" int r3d_lp_ws_offsets(int n_cap, int kp1, long* out6) {
  R3D_REQUIRE(out6 && n_cap > 0 && kp1 >= 2, "r3d_lp_ws_offsets: bad arguments");
  const LpWs L = lp_carve(nullptr, n_cap, kp1);
  const int32_t* base = nullptr;
  out6[0] = (const int32_t*)L.row_ptr - base; out6[1] = (const int32_t*)L.col - base; out6[2] = (const int32_t*)L.val - base;
  out6[3] = (const int32_t*)L.dinv - base; out6[4] = (const int32_t*)L.agg - base; out6[5] = (const int32_t*)L.cg - base;
  return R3D_OK;
}

// ---------------------------------------------------------------------------
// Runtime guard of the multi-stream schedule (packed fp32 arithmetic beside bf16-MFMA-dense waves computed
// wrong graph weights on this chip; the library is built without those instructions and tests/test_gpu_concurrency.py
// gates it, but the cause is not established).  Recomputes the directed gaussian weights of the systems whose graph a
// preceding r3d_label_propagate(_batched) left in ws -- call it with nothing else on the chip -- into `scratch`
// (r3d_graph_weights_verify_words floats per system) and counts the entries whose BITS differ from the ones the solve
// used: *mismatch_out += count (device int).  ~2 ms for 32 systems of workload S.
// ---------------------------------------------------------------------------
__global__ void r3d_weights_compare_kernel(const int* __restrict__ row_ptr, const int* __restrict__ n_dev, int n_cap,
                                           const float* __restrict__ wdir, const float* __restrict__ wdir2, long scratch_stride,
                                           int* __restrict__ mismatch, HgEp st) {
  const int ep = blockIdx.y;
  HG_WS(row_ptr); HG_AT(n_dev, st.desc); HG_WS(wdir);
  wdir2 += (long)ep * scratch_stride;
  const int n = min(*n_dev, n_cap);
  const long nnz2 = 2L * row_ptr[n];
  int bad = 0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < nnz2; e += (long)gridDim.x * blockDim.x)
    bad += __float_as_uint(wdir[e]) != __float_as_uint(wdir2[e]) ? 1 : 0;
  bad = (int)r3d_wave_sum((float)bad);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(mismatch, bad);
}
extern "C" long r3d_graph_weights_verify_words(int n_cap, int kp1) {
  const long nnz_cap = 2L * n_cap * (kp1 - 1);
  return 2 * nnz_cap + nnz_cap + n_cap + 64;  // wdir | val | dinv
}
extern "C" int r3d_graph_weights_verify(int n_ep, const float* nodes, long ldn, int D, const int32_t* n_dev, long desc_stride,
                                        int n_cap, int kp1, float sigma, int32_t* ws, long ws_words, long ws_stride,
                                        float* scratch, int32_t* mismatch_out, void* stream) {
  R3D_REQUIRE(nodes && n_dev && ws && scratch && mismatch_out, "r3d_graph_weights_verify: null pointer");
  R3D_REQUIRE(n_cap > 0 && kp1 >= 2 && ws_words >= r3d_lp_ws_words(n_cap, kp1) && n_ep >= 1 &&
                  (n_ep == 1 || ((ws_stride & 3) == 0 && ws_stride >= ws_words)),
              "r3d_graph_weights_verify: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const LpWs L = lp_carve(ws, n_cap, kp1);
  const long nnz_cap = 2L * n_cap * (kp1 - 1), per = r3d_graph_weights_verify_words(n_cap, kp1);
  // the kernel addresses its outputs through the systems' scratch stride: run it system by system into the guard's own
  // arrays (outputs are plain pointers there)
  for (int e = 0; e < n_ep; ++e) {
    HgEp one{};
    float* base = scratch + (long)e * per;
    int32_t* wse = ws + (long)e * ws_stride;
    const LpWs Le = lp_carve(wse, n_cap, kp1);
    hipLaunchKernelGGL(r3d_graph_weights_kernel, dim3(n_cap, 1), dim3(256), 0, st, nodes + (long)e * n_cap * ldn, ldn, D, Le.outb,
                       (int)Le.words, n_dev + (long)e * desc_stride, n_cap, Le.row_ptr, Le.col, sigma, base + 2 * nnz_cap,
                       base + 3 * nnz_cap, base, one);
  }
  HgEp ep{};
  ep.desc = desc_stride; ep.ws = ws_stride;
  hipLaunchKernelGGL(r3d_weights_compare_kernel, dim3(256, n_ep), dim3(256), 0, st, L.row_ptr, n_dev, n_cap, L.wdir, scratch, per,
                     mismatch_out, ep);
  R3D_LAUNCH_CHECK("r3d_graph_weights_verify");
  return R3D_OK;
}

// {converged, iterations} of every system -> stats_out (stride st.stats words)
__global__ void r3d_cg_stats_kernel(const CgState* __restrict__ cg, int* __restrict__ stats_out, HgEp st) {
  const int ep = blockIdx.x;
  HG_WS(cg); HG_AT(stats_out, st.stats);
  if (threadIdx.x == 0) { stats_out[0] = cg->done; stats_out[1] = cg->iters; }
}

// coarse space of the graph r3d_label_propagate built: aggregates, M W, E^-1 (shared by forward and adjoint solve)
static int lp_coarse_space(const LpWs& L, const float* nodes, long ldn, int D, const int32_t* n_dev, const int32_t* n_proto_dev,
                           int n_cap, float alpha, int n_ep, const HgEp& ep, hipStream_t st) {
  const int agg_rpw = 4;  // 16 nodes per workgroup: the 64 seed rows are staged once per workgroup
  const size_t agg_lds = ((size_t)HG_M * (D | 1) + 4 * D) * sizeof(float);
  static bool lds_opt_in = false;
  if (!lds_opt_in) {  // 69 KB of dynamic LDS at D = 256 (64 KB is the default ceiling)
    hipFuncSetAttribute((const void*)r3d_cg_aggregate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    lds_opt_in = true;
  }
  hipLaunchKernelGGL(r3d_cg_aggregate_kernel, dim3(r3d_cdiv(n_cap, 4 * agg_rpw), n_ep), dim3(256), agg_lds, st, nodes, ldn, D,
                     n_dev, n_proto_dev, n_cap, agg_rpw, L.agg, ep);
  hipLaunchKernelGGL(r3d_cg_mw_kernel, dim3(r3d_cdiv(n_cap, 4), n_ep), dim3(256), 0, st, L.row_ptr, L.col, L.val, L.dinv, L.agg,
                     n_dev, n_cap, alpha, 1, L.MW, ep);
  hipLaunchKernelGGL(r3d_cg_epart_kernel, dim3(HG_EBLOCKS, n_ep), dim3(128), 0, st, L.MW, L.dinv, L.agg, n_dev, n_cap, L.Epart,
                     ep);
  hipLaunchKernelGGL(r3d_cg_einv_kernel, dim3(n_ep), dim3(1024), 0, st, L.Epart, L.Einv, ep);
  return R3D_OK;
}

// workgroups a launch must hold for the LDS-resident SpMV to be chosen (test / A-B utility: 0 = always, a huge value = never)
static int g_cg_lds_min_blocks = 256;
extern "C" int r3d_debug_set_cg_spmv_lds_min_blocks(int min_blocks) {
  const int old = g_cg_lds_min_blocks;
  g_cg_lds_min_blocks = min_blocks;
  return old;
}

// two-level CG on the already built graphs and coarse spaces: X = (I - alpha S)^-1 RHS for every system of the batch.
// rhs_stride / x_stride: float4 rows between the systems' right-hand sides / solutions.
static int lp_solve(const LpWs& L, const float* RHS, long rhs_stride, const int32_t* n_dev, int n_cap, float alpha, int max_iter,
                    float tol, float* X, long x_stride, int32_t* stats_out, int n_ep, const HgEp& ep, hipStream_t st) {
  const int nblk_v = (int)hg_vblocks(n_cap);
  int rpb = HG_ROWS_PER_BLOCK_MIN;
  while (r3d_cdiv(n_cap, rpb) > HG_MAX_PART) rpb += 4;
  const int nblk_s = r3d_cdiv(n_cap, rpb);
  R3D_REQUIRE(nblk_s <= HG_MAX_PART, "r3d_label_propagate: n_cap too large");
  float4* x = (float4*)X;
  const float tol2 = tol * tol;
  r3d_fill_words_ep(&L.cg->ticket, 0u, 1, n_ep, ep.ws, st);
  // partials of b -> (last workgroup) c0 = E^-1 W^T b ; x0 = W c0, r0 = b - (M W) c0 -> (last workgroup) mu, rz of iteration 0
  for (int mode = 0; mode < 2; ++mode)
    hipLaunchKernelGGL(r3d_cg_init_kernel, dim3(nblk_v, n_ep), dim3(256), 0, st, (const float4*)RHS, L.dinv, L.agg, L.MW, n_dev,
                       n_cap, mode, x, L.r, L.p, L.q, L.part, L.Einv, tol2, L.cg, ep, rhs_stride, x_stride);
  // the SpMV with the residual in LDS when it fits and the launch holds enough systems to fill the chip with 128-row
  // workgroups (a single system is 35 of them at workload S: there the one-row-per-wave kernel, 1099 workgroups, is the
  // faster one); same bits either way
  const size_t lds_r = (size_t)n_cap * sizeof(float4);
  const bool use_lds = lds_r <= SL_MAX_LDS && rpb == HG_ROWS_PER_BLOCK_MIN && (long)n_ep * r3d_cdiv(n_cap, SL_ROWS) >= g_cg_lds_min_blocks;
  if (use_lds) {
    static size_t attr = 0;
    if (lds_r > attr) {
      hipError_t e = hipFuncSetAttribute((const void*)r3d_cg_spmv_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r);
      R3D_REQUIRE(e == hipSuccess, "r3d_label_propagate: cannot reserve %zu B of LDS: %s", lds_r, hipGetErrorString(e));
      attr = lds_r;
    }
  }
  const int nblk_l = r3d_cdiv(n_cap, SL_ROWS);
  const int nblk_pq = nblk_s;  // (the LDS kernel writes the same 4-row partials)
  for (int it = 0; it < max_iter; ++it) {
    if (use_lds)
      hipLaunchKernelGGL(r3d_cg_spmv_lds_kernel, dim3(nblk_l, n_ep), dim3(64 * SL_WAVES), lds_r, st, L.row_ptr, L.col, L.val, L.dinv,
                         L.agg, L.MW, n_dev, n_cap, alpha, it, SL_ROWS, L.r, L.p, L.q, L.part_pq, L.cg, ep);
    else
      hipLaunchKernelGGL(r3d_cg_spmv_kernel, dim3(nblk_s, n_ep), dim3(256), 0, st, L.row_ptr, L.col, L.val, L.dinv, L.agg, L.MW,
                         n_dev, n_cap, alpha, it, rpb, L.r, L.p, L.q, L.part_pq, L.cg, ep);
    hipLaunchKernelGGL(r3d_cg_update_kernel, dim3(nblk_v, n_ep), dim3(256), 0, st, n_dev, n_cap, it, nblk_pq, L.dinv, L.agg, L.MW,
                       L.p, L.q, x, L.r, L.part_pq, L.part, L.Einv, tol2, L.cg, ep, x_stride);
  }
  if (stats_out) hipLaunchKernelGGL(r3d_cg_stats_kernel, dim3(n_ep), dim3(64), 0, st, L.cg, stats_out, ep);
  return R3D_OK;
}

static int label_propagate_impl(int n_ep, const HgEp& ep, const float* nodes, long ldn, int D, const int32_t* nbr, int kp1,
                                const float* Y, const int32_t* n_dev, const int32_t* n_proto_dev, int n_cap, float sigma,
                                float alpha, int max_iter, float tol, float* Z, int32_t* ws, long ws_words, int32_t* stats_out,
                                void* stream) {
  R3D_REQUIRE(nodes && nbr && Y && n_dev && n_proto_dev && Z && ws, "r3d_label_propagate: null pointer");
  R3D_REQUIRE(n_cap > 0 && kp1 >= 2 && ws_words >= r3d_lp_ws_words(n_cap, kp1),
              "r3d_label_propagate: workspace of %ld words, r3d_lp_ws_words(%d, %d) = %ld needed", ws_words, n_cap, kp1,
              n_cap > 0 && kp1 >= 2 ? r3d_lp_ws_words(n_cap, kp1) : -1L);
  R3D_REQUIRE((ldn & 3) == 0 && ((uintptr_t)nodes & 15) == 0,
              "r3d_label_propagate: node rows are read as float4: ldn must be a multiple of 4 and nodes 16-byte aligned");
  R3D_REQUIRE(n_cap > 0 && n_cap <= 32768 && D > 0 && D <= 256 && (D & 3) == 0 && kp1 >= 2,
              "r3d_label_propagate: unsupported n_cap=%d D=%d kp1=%d", n_cap, D, kp1);
  R3D_REQUIRE(max_iter > 0 && max_iter <= HG_MAX_ITER && sigma > 0.f, "r3d_label_propagate: bad solver parameters");
  R3D_REQUIRE(((uintptr_t)ws & 15) == 0 && ((uintptr_t)Y & 15) == 0 && ((uintptr_t)Z & 15) == 0,
              "r3d_label_propagate: ws, Y and Z must be 16-byte aligned");
  R3D_REQUIRE(n_ep >= 1 && n_ep <= 4096 && (n_ep == 1 || ((ep.ws & 3) == 0 && ep.ws >= ws_words)),
              "r3d_label_propagate: %d systems need a scratch stride that is a multiple of 4 words and >= the scratch size", n_ep);
  hipStream_t st = (hipStream_t)stream;
  const LpWs L = lp_carve(ws, n_cap, kp1);
  const long words = L.words;
  hipLaunchKernelGGL(r3d_graph_bits_kernel, dim3(r3d_cdiv(n_cap, 4), n_ep), dim3(256), 0, st, nbr, kp1, n_dev, n_cap,
                     (int)words, L.outb, ep);
  hipLaunchKernelGGL(r3d_graph_transpose_kernel, dim3(r3d_cdiv(words, BT_CW), r3d_cdiv(n_cap, BT_ROWS), n_ep), dim3(256), 0, st,
                     L.outb, (int)words, n_cap, L.sym, ep);
  hipLaunchKernelGGL(r3d_graph_rowlen_kernel, dim3(r3d_cdiv(n_cap, 4), n_ep), dim3(256), 0, st, L.sym, L.outb, (int)words,
                     n_dev, n_cap, L.row_len, ep);
  hipLaunchKernelGGL(r3d_scan_kernel, dim3(n_ep), dim3(1024), 0, st, L.row_len, n_cap, L.row_ptr, ep);
  hipLaunchKernelGGL(r3d_graph_cols_kernel, dim3(r3d_cdiv(n_cap, 4), n_ep), dim3(256), 0, st, L.sym, (int)words, n_dev,
                     n_cap, L.row_ptr, L.col, ep);
  hipLaunchKernelGGL(r3d_graph_weights_kernel, dim3(n_cap, n_ep), dim3(256), 0, st, nodes, ldn, D, L.outb,
                     (int)words, n_dev, n_cap, L.row_ptr, L.col, sigma, L.val, L.dinv, L.wdir, ep);
  hipLaunchKernelGGL(r3d_graph_normalize_kernel, dim3(r3d_cdiv(n_cap, 4), n_ep), dim3(256), 0, st, L.row_ptr, L.col, L.dinv,
                     n_dev, n_cap, L.val, ep);
  int rc = lp_coarse_space(L, nodes, ldn, D, n_dev, n_proto_dev, n_cap, alpha, n_ep, ep, st);
  if (rc) return rc;
  rc = lp_solve(L, Y, ep.y, n_dev, n_cap, alpha, max_iter, tol, Z, ep.z, stats_out, n_ep, ep, st);
  if (rc) return rc;
  R3D_LAUNCH_CHECK("r3d_label_propagate");
  return R3D_OK;
}

extern "C" int r3d_label_propagate(const float* nodes, long ldn, int D, const int32_t* nbr, int kp1,
                                   const float* Y, const int32_t* n_dev, const int32_t* n_proto_dev, int n_cap, float sigma,
                                   float alpha, int max_iter, float tol, float* Z, int32_t* ws, long ws_words,
                                   int32_t* stats_out, void* stream) {
  const HgEp one{};
  return label_propagate_impl(1, one, nodes, ldn, D, nbr, kp1, Y, n_dev, n_proto_dev, n_cap, sigma, alpha, max_iter, tol, Z, ws,
                              ws_words, stats_out, stream);
}

// n_ep systems at once.  System e: nodes / nbr / Y / Z rows [e * n_cap, (e + 1) * n_cap) of the batch arrays, its node and
// prototype counts at n_dev[e * desc_stride] / n_proto_dev[e * desc_stride], scratch ws + e * ws_stride (a multiple of 4
// words, >= r3d_lp_ws_words), {converged, iterations} at stats_out + e * stats_stride.  Every CG launch serves all systems;
// max_iter launches are issued, a system that converged earlier idles through the rest.
// The solve alone, on the graph and coarse space a preceding r3d_label_propagate(_batched) left in ws: further right-hand
// sides of the same systems (label columns 4..7 of an episode with more than 3 ways: Y and Z = plane 1 of the two-plane
// arrays).  The CG vectors in ws are overwritten; graph, weights and preconditioner -- what the backward needs -- are not.
extern "C" int r3d_label_propagate_solve_batched(int n_ep, const float* Y, const int32_t* n_dev, long desc_stride, int n_cap,
                                                 int kp1, float alpha, int max_iter, float tol, float* Z, int32_t* ws,
                                                 long ws_words, long ws_stride, int32_t* stats_out, long stats_stride,
                                                 void* stream) {
  R3D_REQUIRE(Y && n_dev && Z && ws, "r3d_label_propagate_solve: null pointer");
  R3D_REQUIRE(n_cap > 0 && kp1 >= 2 && ws_words >= r3d_lp_ws_words(n_cap, kp1), "r3d_label_propagate_solve: workspace too short");
  R3D_REQUIRE(max_iter > 0 && max_iter <= HG_MAX_ITER, "r3d_label_propagate_solve: bad solver parameters");
  R3D_REQUIRE(((uintptr_t)ws & 15) == 0 && ((uintptr_t)Y & 15) == 0 && ((uintptr_t)Z & 15) == 0,
              "r3d_label_propagate_solve: ws, Y and Z must be 16-byte aligned");
  R3D_REQUIRE(n_ep >= 1 && n_ep <= 4096 && (n_ep == 1 || ((ws_stride & 3) == 0 && ws_stride >= ws_words)),
              "r3d_label_propagate_solve: bad scratch stride");
  HgEp ep{};
  ep.nodes = ep.nbr = ep.y = ep.z = n_cap;
  ep.desc = desc_stride; ep.ws = ws_stride; ep.stats = stats_stride;
  const LpWs L = lp_carve(ws, n_cap, kp1);
  const int rc = lp_solve(L, Y, ep.y, n_dev, n_cap, alpha, max_iter, tol, Z, ep.z, stats_out, n_ep, ep, (hipStream_t)stream);
  if (rc) return rc;
  R3D_LAUNCH_CHECK("r3d_label_propagate_solve");
  return R3D_OK;
}

extern "C" int r3d_label_propagate_batched(int n_ep, const float* nodes, long ldn, int D, const int32_t* nbr, int kp1,
                                           const float* Y, const int32_t* n_dev, const int32_t* n_proto_dev, long desc_stride,
                                           int n_cap, float sigma, float alpha, int max_iter, float tol, float* Z, int32_t* ws,
                                           long ws_words, long ws_stride, int32_t* stats_out, long stats_stride, void* stream) {
  HgEp ep{};
  ep.nodes = ep.nbr = ep.y = ep.z = n_cap;
  ep.desc = desc_stride; ep.ws = ws_stride; ep.stats = stats_stride;
  return label_propagate_impl(n_ep, ep, nodes, ldn, D, nbr, kp1, Y, n_dev, n_proto_dev, n_cap, sigma, alpha, max_iter, tol, Z, ws,
                              ws_words, stats_out, stream);
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
__global__ __launch_bounds__(256) void r3d_lp_bwd_dd_kernel(const int* __restrict__ row_ptr, const hg_col_t* __restrict__ col,
                                                            const float* __restrict__ val, const float* __restrict__ dinv,
                                                            const int* __restrict__ n_dev, int n_cap, float alpha,
                                                            const float4* __restrict__ lam, const float4* __restrict__ Z,
                                                            float* __restrict__ dD, HgEp st) {
  const int ep = blockIdx.y;
  HG_WS(row_ptr); HG_WS(col); HG_WS(val); HG_WS(dinv); HG_AT(n_dev, st.desc); HG_AT(lam, st.lam); HG_AT(Z, st.z); HG_WS(dD);
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
    const int* __restrict__ row_ptr, const hg_col_t* __restrict__ col, const float* __restrict__ dinv,
    const int* __restrict__ n_dev, int n_cap, float sigma, float alpha, const float4* __restrict__ lam,
    const float4* __restrict__ Z, const float* __restrict__ dD, float* __restrict__ dnodes, long ldd, HgEp st) {
  {
    const int ep = blockIdx.y;
    nodes += (long)ep * st.nodes * ldn; HG_WS(wdir); HG_WS(row_ptr); HG_WS(col); HG_WS(dinv); HG_AT(n_dev, st.desc);
    HG_AT(lam, st.lam); HG_AT(Z, st.z); HG_WS(dD); dnodes += (long)ep * st.dn * ldd;
  }
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
                                   float4* __restrict__ G, const float4* __restrict__ Z2, float4* __restrict__ G2, HgEp st) {
  {
    const int ep = blockIdx.y;
    HG_AT(Z, st.z); HG_AT(n_proto_dev, st.desc); HG_AT(labels, st.labels); HG_AT(G, st.g);
    if (Z2) { HG_AT(Z2, st.z); HG_AT(G2, st.g); }
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cap) return;
  const int n_proto = *n_proto_dev;
  const int q = i - n_proto;
  float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (q >= 0 && q < n_qpts) {
    const float4 z = Z[i];
    const float4 z2 = Z2 ? Z2[i] : f4_zero();
    const float zv[8] = {z.x, z.y, z.z, z.w, z2.x, z2.y, z2.z, z2.w};
    float mx = zv[0];
    for (int c = 1; c < n_classes; ++c) mx = fmaxf(mx, zv[c]);
    float se = 0.f;
    for (int c = 0; c < n_classes; ++c) se += expf(zv[c] - mx);
    const int lab = (int)labels[q];
    const float sc = gscale[0] / (float)n_qpts;
    for (int c = 0; c < n_classes; ++c) g[c] = (expf(zv[c] - mx) / se - (c == lab ? 1.f : 0.f)) * sc;
  }
  G[i] = make_float4(g[0], g[1], g[2], g[3]);
  if (G2) G2[i] = make_float4(g[4], g[5], g[6], g[7]);
}

// Backward through label propagation + affinity.  Requires ws exactly as r3d_label_propagate left it.
// G (n_cap,4) = dL/dZ (from r3d_ce_grad); lam scratch (n_cap,4); dnodes (n_cap, ldd) out.
static int label_propagate_bwd_impl(int n_ep, const HgEp& ep, const float* nodes, long ldn, int D, int kp1, const float* Z,
                                    const float* G, const int32_t* n_dev, int n_cap, float sigma, float alpha, int max_iter,
                                    float tol, float* lam, float* dnodes, long ldd, int32_t* ws, long ws_words,
                                    int32_t* stats_out, void* stream) {
  R3D_REQUIRE(nodes && Z && G && n_dev && lam && dnodes && ws, "r3d_label_propagate_bwd: null pointer");
  R3D_REQUIRE(n_cap > 0 && kp1 >= 2 && ws_words >= r3d_lp_ws_words(n_cap, kp1),
              "r3d_label_propagate_bwd: workspace of %ld words is shorter than r3d_lp_ws_words(%d, %d)", ws_words, n_cap, kp1);
  R3D_REQUIRE((ldn & 3) == 0 && ((uintptr_t)nodes & 15) == 0 && ((uintptr_t)ws & 15) == 0 && ((uintptr_t)Z & 15) == 0 &&
                  ((uintptr_t)G & 15) == 0 && ((uintptr_t)lam & 15) == 0,
              "r3d_label_propagate_bwd: ldn must be a multiple of 4; nodes, ws, Z, G, lam 16-byte aligned");
  R3D_REQUIRE(n_cap > 0 && n_cap <= 32768 && D > 0 && D <= 256 && max_iter > 0 && max_iter <= HG_MAX_ITER,
              "r3d_label_propagate_bwd: bad arguments");
  R3D_REQUIRE(n_ep >= 1 && n_ep <= 4096 && (n_ep == 1 || ((ep.ws & 3) == 0 && ep.ws >= ws_words)),
              "r3d_label_propagate_bwd: %d systems need a scratch stride that is a multiple of 4 words and >= the scratch size", n_ep);
  hipStream_t st = (hipStream_t)stream;
  const LpWs L = lp_carve(ws, n_cap, kp1);
  int rc = lp_solve(L, G, ep.g, n_dev, n_cap, alpha, max_iter, tol, lam, ep.lam, stats_out, n_ep, ep, st);
  if (rc) return rc;
  float* dD = (float*)L.q;  // the CG vectors are free again after the solve
  hipLaunchKernelGGL(r3d_lp_bwd_dd_kernel, dim3(r3d_cdiv(n_cap, 4), n_ep), dim3(256), 0, st, L.row_ptr, L.col, L.val, L.dinv, n_dev,
                     n_cap, alpha, (const float4*)lam, (const float4*)Z, dD, ep);
  hipLaunchKernelGGL(r3d_lp_bwd_dx_kernel, dim3(n_cap, n_ep), dim3(256), 0, st, nodes, ldn, D, L.wdir,
                     L.row_ptr, L.col, L.dinv, n_dev, n_cap, sigma, alpha, (const float4*)lam, (const float4*)Z, dD, dnodes,
                     ldd, ep);
  R3D_LAUNCH_CHECK("r3d_label_propagate_bwd");
  return R3D_OK;
}
extern "C" int r3d_label_propagate_bwd(const float* nodes, long ldn, int D, int kp1, const float* Z, const float* G,
                                       const int32_t* n_dev, int n_cap, float sigma, float alpha, int max_iter, float tol,
                                       float* lam, float* dnodes, long ldd, int32_t* ws, long ws_words, int32_t* stats_out,
                                       void* stream) {
  const HgEp one{};
  return label_propagate_bwd_impl(1, one, nodes, ldn, D, kp1, Z, G, n_dev, n_cap, sigma, alpha, max_iter, tol, lam, dnodes, ldd,
                                  ws, ws_words, stats_out, stream);
}
// n_ep systems at once (layout as r3d_label_propagate_batched; G, lam, dnodes: n_cap rows per system)
extern "C" int r3d_label_propagate_bwd_batched(int n_ep, const float* nodes, long ldn, int D, int kp1, const float* Z,
                                               const float* G, const int32_t* n_dev, long desc_stride, int n_cap, float sigma,
                                               float alpha, int max_iter, float tol, float* lam, float* dnodes, long ldd,
                                               int32_t* ws, long ws_words, long ws_stride, int32_t* stats_out,
                                               long stats_stride, void* stream) {
  HgEp ep{};
  ep.nodes = ep.z = ep.g = ep.lam = ep.dn = n_cap;
  ep.desc = desc_stride; ep.ws = ws_stride; ep.stats = stats_stride;
  return label_propagate_bwd_impl(n_ep, ep, nodes, ldn, D, kp1, Z, G, n_dev, n_cap, sigma, alpha, max_iter, tol, lam, dnodes, ldd,
                                  ws, ws_words, stats_out, stream);
}

// G = dL/dZ of n_ep systems (n_ep == 1: the ABI-version-2 call).  labels: n_query_pts int64 per system; *gscale_dev scales
// every system alike (the step's loss is the SUM of the episodes' losses).
extern "C" int r3d_ce_grad_batched(int n_ep, const float* Z, const int32_t* n_proto_dev, long desc_stride, int n_cap,
                                   int n_query_pts, int n_classes, const int64_t* labels, const float* gscale_dev, float* G,
                                   void* stream) {
  R3D_REQUIRE(Z && n_proto_dev && labels && gscale_dev && G && n_ep >= 1 && n_ep <= 65535, "r3d_ce_grad: bad arguments");
  R3D_REQUIRE(n_classes >= 2 && n_classes <= 2 * HG_NC, "r3d_ce_grad: %d classes (2..%d)", n_classes, 2 * HG_NC);
  HgEp ep{};
  ep.z = ep.g = n_cap; ep.desc = desc_stride; ep.labels = n_query_pts;
  // more than 4 classes: Z and G are (2, n_ep * n_cap, 4), plane 1 = classes 4..7
  const long plane = (long)n_ep * n_cap;
  const float4* Z2 = n_classes > HG_NC ? (const float4*)Z + plane : nullptr;
  float4* G2 = n_classes > HG_NC ? (float4*)G + plane : nullptr;
  hipLaunchKernelGGL(r3d_ce_grad_kernel, dim3(r3d_cdiv(n_cap, 256), n_ep), dim3(256), 0, (hipStream_t)stream, (const float4*)Z,
                     n_proto_dev, n_cap, n_query_pts, n_classes, (const long long*)labels, gscale_dev, (float4*)G, Z2, G2, ep);
  R3D_LAUNCH_CHECK("r3d_ce_grad");
  return R3D_OK;
}
extern "C" int r3d_ce_grad(const float* Z, const int32_t* n_proto_dev, int n_cap, int n_query_pts, int n_classes,
                           const int64_t* labels, const float* gscale_dev, float* G, void* stream) {
  return r3d_ce_grad_batched(1, Z, n_proto_dev, 0, n_cap, n_query_pts, n_classes, labels, gscale_dev, G, stream);
}

// logits (n_q, n_classes, N) fp32, loss (1) fp32, pred (n_q*N) int32 (argmax), labels int64 -- per system; system e reads
// Z rows [e * z_ep_rows, ...) and writes logits / loss / pred number e of the batch arrays
extern "C" int r3d_query_logits_ce_batched(int n_ep, const float* Z, long z_ep_rows, const int32_t* n_proto_dev, long desc_stride,
                                           int n_q, int N, int n_classes, const int64_t* labels, float* logits, float* loss_out,
                                           int32_t* pred_out, void* stream) {
  R3D_REQUIRE(Z && n_proto_dev && logits, "r3d_query_logits_ce: null pointer");
  R3D_REQUIRE(n_q > 0 && N > 0 && n_classes >= 2 && n_classes <= 2 * HG_NC && n_ep >= 1, "r3d_query_logits_ce: bad shape");
  R3D_REQUIRE(n_classes <= HG_NC || z_ep_rows > 0, "r3d_query_logits_ce: more than %d classes need the batched form "
              "(Z as two planes (2, n_ep * z_ep_rows, 4))", HG_NC);
  HgEp ep{};
  ep.z = z_ep_rows; ep.desc = desc_stride; ep.labels = (long)n_q * N; ep.logits = (long)n_q * n_classes * N; ep.loss = 1;
  ep.pred = (long)n_q * N;
  const float4* Z2 = n_classes > HG_NC ? (const float4*)Z + (long)n_ep * z_ep_rows : nullptr;
  hipLaunchKernelGGL(r3d_logits_ce_kernel, dim3(n_ep), dim3(1024), 0, (hipStream_t)stream, (const float4*)Z, Z2,
                     n_proto_dev, n_q, N, n_classes, (const long long*)labels, logits, loss_out, pred_out, ep);
  R3D_LAUNCH_CHECK("r3d_query_logits_ce");
  return R3D_OK;
}
extern "C" int r3d_query_logits_ce(const float* Z, const int32_t* n_proto_dev, int n_q, int N, int n_classes,
                                   const int64_t* labels, float* logits, float* loss_out, int32_t* pred_out,
                                   void* stream) {
  return r3d_query_logits_ce_batched(1, Z, 0, n_proto_dev, 0, n_q, N, n_classes, labels, logits, loss_out, pred_out, stream);
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
    int it, on;
    if (kp.func == (void*)r3d_cg_spmv_kernel || kp.func == (void*)r3d_cg_spmv_lds_kernel) { it = *(const int*)kp.kernelParams[9]; on = it < budget; }
    else if (kp.func == (void*)r3d_cg_update_kernel) { it = *(const int*)kp.kernelParams[2]; on = it < budget; }
    else continue;
    ++found;
    if (hipGraphNodeSetEnabled(ge, nodes[i], on ? 1u : 0u) != hipSuccess) { rc = R3D_ERR_LAUNCH; break; }
  }
  free(nodes);
  if (rc != R3D_OK) {
    r3d_set_error("r3d_graph_set_lp_budget: HIP graph call failed: %s", hipGetErrorString(hipGetLastError()));
    return rc;
  }
  if (n_cg) *n_cg = found;
  return R3D_OK;
}

