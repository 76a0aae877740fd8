// EdgeConv in TRAINING mode (batch-statistics BatchNorm over all B*N*K edges) and its backward, gfx950.
//
// Reference: models/dgcnn.py:26-61,117-118 with nn.BatchNorm2d in train mode.  With W1 = [Wa | Wb]:
//   e1(i,t) = P[j_it] + Q[i],  P = Wa x, Q = (Wb - Wa) x          (PQ: one point-wise GEMM, raw)
//   h1 = lrelu(s1 e1 + t1)        s1 = g1 / sqrt(var1 + eps), t1 = b1 - mu1 s1   (stats over edges)
//   z2 = W2 h1 ;  a2 = lrelu(s2 z2 + t2) ;  out[i] = max_t a2(i,t)
// Forward:  r3d_edge_stats1 (sum e1, sum e1^2) -> fold -> r3d_edgeconv_train_fwd mode 1 (sum z2,
// sum z2^2) -> fold -> mode 0 (output, argmax, z2 at the argmax).
// Backward: the max routes dout to one edge per (point, channel), but BatchNorm's mean terms make
// dz2 DENSE over edges, so the backward re-runs the edge GEMM:
//   B1 (matrix core): recompute h1, z2; dz2 = s2 (dy2 - m1 - zhat2 m2); dW2 += dz2^T h1;
//       dh1 = dz2 W2; dy1 = dh1 lrelu'(u1) -> stored per edge; partial sums of dy1, dy1 ehat1
//       also per point: B[i] = sum_t dy1(i,t), EH[i] = sum_t ehat1(i,t)
//   R  (once per layer): the REVERSE neighbour list -- for every target j the edges (i,t) with idx[i][t] == j, in
//       ascending edge order -- by a counting sort per (cloud, target range) in LDS
//   B2 (gather only): de1 = s1 (dy1 - n1 - ehat1 n2), so
//       dQ[i] = s1 (B[i] - K n1 - n2 EH[i])
//       dP[j] = s1 (sum_in dy1 - c_j n1 - n2 is1 (c_j (P[j] - mu1) + sum_in Q[i]))   over the c_j incoming edges of j
//       one wave per point, plain stores, fixed summation order: deterministic.  (The first version scattered de1 with
//       float atomics on 256-B rows: 3/4 of that kernel's time, and the one non-deterministic sum of the training step.)
//
// Batches of episodes (round 3).  All B clouds of a batch of episodes go through ONE launch per pass; BatchNorm keeps
// the statistics of every getFeatures call apart (segments of clouds_a support clouds and clouds_b query clouds
// alternating, common.h: r3d_segmap; segment 2 e + p = call p of episode e).  The unit of the statistics is the CHUNK =
// EC_CHUNK consecutive points of one cloud: a workgroup takes whole chunks, writes one partial per chunk, and a segment's
// sums are its chunks added in ascending order in fp64 -- the same numbers whether the episode runs alone or inside a
// batch, and whatever the grid.  The BatchNorm vectors of segment s sit at (pointer + s * bn_stride).
#include <stdlib.h>
#include "edge_tile.h"

#define ET_LD 65
#define ET_MAXBLK 1024
#define EC_CHUNK 32  // points per statistics chunk (8 units of 4 points)

static __device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : 0.2f * v; }

// chunk -> its cloud and point range [p0, p1) (global rows).  32-bit and forced into scalar registers: the 64-bit
// division of a uniform value is VALU code whose results otherwise stay in VGPRs for the whole unit loop (B N K < 2^31
// is checked at the entry points, so rows and chunks fit an int).
struct EcGeom {
  int N, cpc;  // points per cloud, chunks per cloud
  __host__ __device__ static EcGeom make(int N) { return EcGeom{N, (N + EC_CHUNK - 1) / EC_CHUNK}; }
  __device__ void range(int chunk, int& cloud, int& p0, int& p1) const {
    cloud = __builtin_amdgcn_readfirstlane(chunk / cpc);
    p0 = __builtin_amdgcn_readfirstlane(cloud * N + (chunk - cloud * cpc) * EC_CHUNK);
    p1 = __builtin_amdgcn_readfirstlane(min(p0 + EC_CHUNK, (cloud + 1) * N));
  }
};
// segment of a cloud, as a scalar
static __device__ __forceinline__ int ec_seg(const r3d_segmap& cs, int cloud) {
  return __builtin_amdgcn_readfirstlane(cs.seg_of_row32(cloud));
}

// ---- BN1 statistics over edges: partial[chunk][2][64] ------------------------------------------
template <int RT>
__global__ __launch_bounds__(256) void r3d_edge_stats1_kernel(const float* __restrict__ PQ, const int* __restrict__ idx,
                                                              EcGeom gm, int n_chunks, float* __restrict__ part,
                                                              float* __restrict__ esum /* (points, 64) sum_t e1, or null */) {
  constexpr int K = 4 * RT;
  __shared__ float sa[4][64], sb[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int N = gm.N;
  for (int item = blockIdx.x; item < n_chunks; item += gridDim.x) {
    const int chunk = r3d_xcd_swizzle(item, n_chunks);  // workgroups sharing an L2 walk the chunks of the same clouds
    int cloud, p0, p1;
    gm.range(chunk, cloud, p0, p1);
    const long cloud0 = (long)cloud * N;
    float a = 0.f, b = 0.f;
    for (int pt_ = p0 + w; pt_ < p1; pt_ += 4) {
      const long pt = pt_;
      const float q = PQ[pt * 128 + 64 + lane];
      const int my_idx = min(max(idx[pt * K + min(lane, K - 1)], 0), N - 1);  // never gather outside the cloud, whatever the list holds
      float pv[K];  // all K neighbour rows in flight (four at a time was a chain of K/4 L2 round trips per point)
#pragma unroll
      for (int t = 0; t < K; ++t) pv[t] = PQ[(cloud0 + __builtin_amdgcn_readlane(my_idx, t)) * 128 + lane];
      float ps = 0.f;  // the point's own sum (the bf16 x 3 backward takes sum_t e1-hat from it)
#pragma unroll
      for (int t = 0; t < K; ++t) {
        const float e = pv[t] + q;
        a += e;
        b += e * e;
        ps += e;
      }
      if (esum) esum[pt * 64 + lane] = ps;
    }
    sa[w][lane] = a;
    sb[w][lane] = b;
    __syncthreads();
    if (w == 0) {
      part[((long)chunk * 2 + 0) * 64 + lane] = ((sa[0][lane] + sa[1][lane]) + sa[2][lane]) + sa[3][lane];
      part[((long)chunk * 2 + 1) * 64 + lane] = ((sb[0][lane] + sb[1][lane]) + sb[2][lane]) + sb[3][lane];
    }
    __syncthreads();
  }
}

static __device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// out[seg][i] = sum over the segment's partial rows part[row][i] in fp64.  64 consecutive outputs per workgroup
// (coalesced 256-B rows of `part`), the row axis split over 16 waves (rows w, w+16, ... RELATIVE to the segment's first
// row; 8 loads in flight), wave totals combined in a fixed order: deterministic, and independent of where the segment
// sits in the batch.  (One wave per output with lanes striding over the rows fetched a 128-B line per 4 useful bytes:
// 20 us for the 768 x 4224 partials of the backward pass.)  blockIdx.y = segment; its rows: count_a / count_b
// alternating (count_b == 0: count_a each).  Outputs i >= split go to out2[seg][i - split].
__global__ __launch_bounds__(1024) void r3d_part_reduce_kernel(const float* __restrict__ part, int count_a, int count_b, int n,
                                                               float* __restrict__ out, long out_stride, int split,
                                                               float* __restrict__ out2, long out2_stride) {
  __shared__ double sm[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const int seg = blockIdx.y;
  const bool odd = count_b > 0 && (seg & 1);
  const int nblk = odd ? count_b : count_a;
  part += (count_b > 0 ? (long)(seg >> 1) * (count_a + count_b) + (odd ? count_a : 0) : (long)seg * count_a) * n;
  double s = 0.0;
  if (i < n) {
    int k = w;
    for (; k + 7 * 16 < nblk; k += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long)(k + 16 * u) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; k < nblk; k += 16) s += (double)part[(long)k * n + i];
  }
  sm[w][lane] = s;
  __syncthreads();
  if (w == 0 && i < n) {
    double t = sm[0][lane];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += sm[q][lane];
    if (i < split) out[seg * out_stride + i] = (float)t;
    else out2[seg * out2_stride + i - split] = (float)t;
  }
}

// ---- backward pass B1 -------------------------------------------------------------------------
// 4-point units on 16x16x4 MFMA (edge_tile.h).  The dW2 edge contraction takes edge 16(s>>2) + 4g + (s&3) at step s
// (conflict-free b32 reads of dz2, one b128 of h1 feeding four column tiles).
// Partials: part_dw [block][64*64] (dW2 is summed over the whole batch: per workgroup), part_bn [chunk][2][64]
// (v = 0: sum dy1, 1: sum dy1*ehat1; per chunk, reduced per segment).
struct EcBn {  // BatchNorm vectors of both edge layers, segment s at + s * stride
  const float *s1, *t1, *mean1, *invstd1, *s2, *t2, *mean2, *invstd2;
  long stride;
};
#include "edgeconv_bwd_bx3.h"

template <int RT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RT <= 5 ? 3 : 2))) void r3d_edgeconv_bwd1_kernel(
    const float* __restrict__ PQ, const int* __restrict__ idx, EcBn bn, const float* __restrict__ W2,
    const float* __restrict__ bn2_sums /* [seg][2][64]: sum dy2, sum dy2 zhat2 */, const float* __restrict__ dout, long lddo,
    const int* __restrict__ argmax, EcGeom gm, r3d_segmap cs /* clouds */, int n_chunks,
    float* __restrict__ DY1 /* (total_points*K, 64) */, float* __restrict__ BE /* (total_points, 128): sum_t dy1 | sum_t ehat1 */,
    float* __restrict__ part_dw, float* __restrict__ part_bn) {
  constexpr int K = 4 * RT, R = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* H = smem;                        // [R][E2_LD] h1
  float* G = smem + R * E2_LD;            // [R][E2_LD] dz2, later dy1
  float* dsm = G + R * E2_LD;             // [4][64] dout of the unit
  int* asm_ = (int*)(dsm + E2_PTS * 64);  // [4][64] argmax of the unit
  float* bnc = (float*)(asm_ + E2_PTS * 64);  // [6][64] layer-2 BatchNorm values of the chunk's segment
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int c = 16 * w + n;  // this lane's column of the z2 / dh1 tiles
  const int N = gm.N;
  f32x4 dw[4];  // dW2 rows 16w + 4g + i, columns 4n + tj
#pragma unroll
  for (int tj = 0; tj < 4; ++tj) dw[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int item = blockIdx.x; item < n_chunks; item += gridDim.x) {
    const int chunk = r3d_xcd_swizzle(item, n_chunks);  // workgroups sharing an L2 walk the chunks of the same clouds
    int cloud, p0, p1;
    gm.range(chunk, cloud, p0, p1);
    const long cloud0 = (long)cloud * N;
    // the chunk's segment: its BatchNorm vectors (L1 / L2 hits after the first chunk of a segment)
    const int seg = ec_seg(cs, cloud);
    const long bo = (long)seg * bn.stride;
    const float sc1 = bn.s1[bo + lane], sh1 = bn.t1[bo + lane], mu1 = bn.mean1[bo + lane], is1 = bn.invstd1[bo + lane];
    // the six per-column values of layer 2 go through LDS: they are needed in one phase of a unit only, and as
    // registers held across the whole unit they are what tips the kernel over 168 (three workgroups per CU)
    if (tid < 64) {
      const double E = (double)cs.seg_rows(seg) * N * K;  // edges of the segment
      bnc[0 * 64 + tid] = bn.s2[bo + tid];
      bnc[1 * 64 + tid] = bn.t2[bo + tid];
      bnc[2 * 64 + tid] = bn.mean2[bo + tid];
      bnc[3 * 64 + tid] = bn.invstd2[bo + tid];
      bnc[4 * 64 + tid] = (float)((double)bn2_sums[(long)seg * 128 + tid] / E);
      bnc[5 * 64 + tid] = (float)((double)bn2_sums[(long)seg * 128 + 64 + tid] / E);
    }
    float sdy = 0.f, sdye = 0.f;
    for (int pt0_ = p0; pt0_ < p1; pt0_ += E2_PTS) {
      const long pt0 = pt0_;
      dsm[tid] = dout[(pt0 + w) * lddo + lane];
      asm_[tid] = argmax[(pt0 + w) * 64 + lane];
      // W2 fragments are re-read (L1 / L2 hits) in the phase that uses them: holding both sets for the whole loop
      // costs 32 registers and the third workgroup per CU.  The opaque zero keeps the loads inside the loop.
      int keep = 0;
      asm volatile("" : "+v"(keep));
      float Bf[16];
      {
        const float4* bz = (const float4*)(W2 + c * 64 + 16 * g + keep);  // z2[e][c] = sum_k h1[e][k] W2[c][k]
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const float4 v = bz[s4];
          Bf[4 * s4] = v.x; Bf[4 * s4 + 1] = v.y; Bf[4 * s4 + 2] = v.z; Bf[4 * s4 + 3] = v.w;
        }
      }
      float eh[K];  // ehat1 of the K edges of point w (this wave's rows), channel = lane
      {
        const int my_idx = min(max(idx[(pt0 + w) * K + min(lane, K - 1)], 0), N - 1);  // never gather outside the cloud, whatever the list holds
        const float q = PQ[(pt0 + w) * 128 + 64 + lane];
        float pv[K];
#pragma unroll
        for (int t = 0; t < K; ++t) pv[t] = PQ[(cloud0 + __builtin_amdgcn_readlane(my_idx, t)) * 128 + lane];
        float* hrow = H + (K * w) * E2_LD + lane;
#pragma unroll
        for (int t = 0; t < K; ++t) {
          const float e1 = pv[t] + q;
          eh[t] = (e1 - mu1) * is1;
          hrow[t * E2_LD] = lrelu(sc1 * e1 + sh1);
        }
      }
      __syncthreads();  // H, dsm, asm_ complete
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc[RT];
      e2_rowgemm<RT>(H, Bf, n, g, acc);
#pragma unroll
      for (int s = 0; s < 16; ++s) Bf[s] = W2[(16 * g + s) * 64 + c + keep];  // dh1[e][c] = sum_k dz2[e][k] W2[k][c]
      const float s2c = bnc[c], t2c = bnc[64 + c], mu2c = bnc[128 + c], is2c = bnc[192 + c], m1c = bnc[256 + c], m2c = bnc[320 + c];
#pragma unroll
      for (int t = 0; t < RT; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = 16 * t + 4 * g + i;
          const int pt = e / K, tt = e - pt * K;
          const float z = acc[t][i];
          const float uu = s2c * z + t2c;
          const float dy = (asm_[pt * 64 + c] == tt) ? dsm[pt * 64 + c] * (uu > 0.f ? 1.f : 0.2f) : 0.f;
          G[e * E2_LD + c] = s2c * (dy - m1c - ((z - mu2c) * is2c) * m2c);
        }
      }
      __syncthreads();  // dz2 complete
      __builtin_amdgcn_sched_barrier(0);
      e2_rowgemm<RT>(G, Bf, n, g, acc);
#pragma unroll
      for (int t = 0; t < RT; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] *= (H[(16 * t + 4 * g + i) * E2_LD + c] > 0.f) ? 1.f : 0.2f;  // u1 > 0 <=> h1 > 0
      }
      // dW2 += dz2^T h1 over the unit's edges
#pragma unroll 4
      for (int s = 0; s < 4 * RT; ++s) {
        const int e = 16 * (s >> 2) + 4 * g + (s & 3);
        const float a = G[e * E2_LD + 16 * w + n];
        const float4 b = *(const float4*)(H + e * E2_LD + 4 * n);
        dw[0] = mfma16(a, b.x, dw[0]);
        dw[1] = mfma16(a, b.y, dw[1]);
        dw[2] = mfma16(a, b.z, dw[2]);
        dw[3] = mfma16(a, b.w, dw[3]);
      }
      __syncthreads();  // every read of dz2 is done
#pragma unroll
      for (int t = 0; t < RT; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) G[(16 * t + 4 * g + i) * E2_LD + c] = acc[t][i];
      }
      __syncthreads();  // dy1 complete
      {
        float* drow = DY1 + ((pt0 + w) * K) * 64 + lane;
        const float* grow = G + (K * w) * E2_LD + lane;
        float bs = 0.f, es = 0.f;
#pragma unroll
        for (int t = 0; t < K; ++t) {
          const float v = grow[t * E2_LD];
          drow[t * 64] = v;
          bs += v;
          es += eh[t];
          sdye += v * eh[t];
        }
        sdy += bs;
        BE[(pt0 + w) * 128 + lane] = bs;
        BE[(pt0 + w) * 128 + 64 + lane] = es;
      }
      // the next unit writes H / dsm / asm_ (last read before the third barrier) before its first barrier and G after it
    }
    // the chunk's BatchNorm-1 partial: four waves (= four points per unit) in a fixed order.  dsm is free here: its last
    // read (dz2) lies before the third barrier of the chunk's last unit; the barrier behind the store keeps the next
    // chunk's first unit from overwriting it early.
    float (*ps)[2][64] = (float (*)[2][64])dsm;  // [4][2][64] aliases dsm | asm_
    ps[w][0][lane] = sdy;
    ps[w][1][lane] = sdye;
    __syncthreads();
    if (tid < 128) {
      const int v = tid >> 6, cc = tid & 63;
      part_bn[((long)chunk * 2 + v) * 64 + cc] = ((ps[0][v][cc] + ps[1][v][cc]) + ps[2][v][cc]) + ps[3][v][cc];
    }
    __syncthreads();
  }
  // thread coordinates again from an opaque copy: carried across the loop they cost a register the loop needs
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  const int w2 = tid2 >> 6, n2 = tid2 & 15, g2 = (tid2 >> 4) & 3;
  float* mypart = part_dw + (long)blockIdx.x * 4096;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *(float4*)(mypart + (16 * w2 + 4 * g2 + i) * 64 + 4 * n2) = float4{dw[0][i], dw[1][i], dw[2][i], dw[3][i]};
}

// ---- one-pass training forward on the same 4-point / 16x16x4 unit ---------------------------------------------
// statistics of z2 (one partial per chunk) and, per point and channel, max / min of the raw z2 over the K edges with
// their positions
template <int RT>
__global__ __launch_bounds__(256) void r3d_edgeconv_train_fwd2_kernel(
    const float* __restrict__ PQ, const int* __restrict__ idx, const float* __restrict__ s1, const float* __restrict__ t1,
    long bn_stride, const float* __restrict__ W2, EcGeom gm, r3d_segmap cs /* clouds */, int n_chunks,
    int* __restrict__ argmax_out, float* __restrict__ zmax_out, float* __restrict__ zmin_out, int* __restrict__ argmin_out,
    float* __restrict__ part /* [chunk][2][64] */) {
  constexpr int K = 4 * RT, R = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* H = smem;              // [R][E2_LD] h1
  float* Z = smem + R * E2_LD;  // [R][E2_LD] z2
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int c = 16 * w + n;
  const int N = gm.N;
  float Bz[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) Bz[s] = W2[c * 64 + 16 * g + s];  // z2[e][c] = sum_k h1[e][k] W2[c][k]
  for (int item = blockIdx.x; item < n_chunks; item += gridDim.x) {
    const int chunk = r3d_xcd_swizzle(item, n_chunks);  // workgroups sharing an L2 walk the chunks of the same clouds
    int cloud, p0, p1;
    gm.range(chunk, cloud, p0, p1);
    const long cloud0 = (long)cloud * N;
    const long bo = (long)ec_seg(cs, cloud) * bn_stride;
    const float sc1 = s1[bo + lane], sh1 = t1[bo + lane];
    float za = 0.f, zb = 0.f;
    for (int pt0_ = p0; pt0_ < p1; pt0_ += E2_PTS) {
      const long pt0 = pt0_;
      {
        const int my_idx = min(max(idx[(pt0 + w) * K + min(lane, K - 1)], 0), N - 1);  // never gather outside the cloud, whatever the list holds
        const float q = PQ[(pt0 + w) * 128 + 64 + lane];
        float pv[K];
#pragma unroll
        for (int t = 0; t < K; ++t) pv[t] = PQ[(cloud0 + __builtin_amdgcn_readlane(my_idx, t)) * 128 + lane];
        float* hrow = H + (K * w) * E2_LD + lane;
#pragma unroll
        for (int t = 0; t < K; ++t) hrow[t * E2_LD] = lrelu(sc1 * (pv[t] + q) + sh1);
      }
      __syncthreads();  // H complete; every wave is done scanning the previous unit's Z
      f32x4 acc[RT];
      e2_rowgemm<RT>(H, Bz, n, g, acc);
#pragma unroll
      for (int t = 0; t < RT; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float z = acc[t][i];
          za += z;
          zb += z * z;
          Z[(16 * t + 4 * g + i) * E2_LD + c] = z;
        }
      }
      __syncthreads();  // Z complete; H free for the next unit's gather
      {
        const float* zp = Z + (K * w) * E2_LD + lane;  // point w, channel lane
        float zx = zp[0], zn = zx;
        int ax = 0, an = 0;
#pragma unroll
        for (int t = 1; t < K; ++t) {
          const float z = zp[t * E2_LD];
          if (z > zx) { zx = z; ax = t; }
          if (z < zn) { zn = z; an = t; }
        }
        const long o = (pt0 + w) * 64 + lane;
        zmax_out[o] = zx;
        zmin_out[o] = zn;
        argmax_out[o] = ax;
        argmin_out[o] = an;
      }
    }
    // the four lane groups of a wave hold partial sums of the same 16 channels; wave w owns channels 16w..16w+15
    za += __shfl_xor(za, 16); zb += __shfl_xor(zb, 16);
    za += __shfl_xor(za, 32); zb += __shfl_xor(zb, 32);
    if (g == 0) {
      part[((long)chunk * 2 + 0) * 64 + c] = za;
      part[((long)chunk * 2 + 1) * 64 + c] = zb;
    }
  }
}

// ---- reverse neighbour list -----------------------------------------------------------------------
// rev_ptr[cloud * N + j] .. rev_ptr[cloud * N + j + 1]: positions in `rev` of the GLOBAL edge ids e = point * K + t whose
// neighbour is point j of that cloud, ascending in e.  One workgroup per (cloud, range of RV_RANGE targets): it reads
// the cloud's lists twice (count, fill), keeps its entries in LDS, and writes every target's segment out in rank
// order (rank = number of smaller edge ids of the segment: an out-of-place sort from LDS to global memory, so the LDS
// atomics that placed the entries in arbitrary order leave no trace).  If the range's entries exceed the LDS buffer it is
// processed in sub-ranges; a single target with more entries than the buffer (impossible for lists of distinct
// neighbours, possible for the garbage lists a non-finite feature cascade can leave) is written by an ordered
// compaction over the edges instead.  Neighbour ids are clamped to the cloud like everywhere else.
#define RV_RANGE 256
#define RV_CAP 12288
__global__ __launch_bounds__(1024) void r3d_edge_reverse_kernel(const int* __restrict__ idx, int N, int K, int n_clouds,
                                                                int* __restrict__ rev_ptr, int* __restrict__ rev) {
  __shared__ int cnt[RV_RANGE];
  __shared__ int off[RV_RANGE + 1];
  __shared__ int cur[RV_RANGE];
  __shared__ int buf[RV_CAP];
  __shared__ int wsum[16];
  __shared__ int below_s, b_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int cloud = blockIdx.y;
  const int r0 = blockIdx.x * RV_RANGE, nr = min(RV_RANGE, N - r0);
  const long E = (long)N * K, gbase = (long)cloud * E;
  const int* lst = idx + gbase;
  if (tid < RV_RANGE) cnt[tid] = 0;
  if (tid == 0) below_s = 0;
  __syncthreads();
  int my_below = 0;
  for (long e0 = tid; e0 < E; e0 += 8 * 1024) {  // 8 list entries in flight per thread and trip
    int jv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) jv[u] = lst[min(e0 + 1024 * u, E - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = min(max(jv[u], 0), N - 1) - r0;
      if (e0 + 1024 * u < E) {
        if (j < 0) ++my_below;
        else if (j < nr) atomicAdd(&cnt[j], 1);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) my_below += __shfl_xor(my_below, o);
  if (lane == 0) atomicAdd(&below_s, my_below);  // integer adds: order free
  __syncthreads();
  // exclusive scan of cnt[0 .. RV_RANGE) (8 waves of 64)
  if (tid < RV_RANGE) {
    const int v = tid < nr ? cnt[tid] : 0;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    cur[tid] = x - v;  // exclusive inside the wave
  }
  __syncthreads();
  if (tid < RV_RANGE) {
    int wb = 0;
    for (int q = 0; q < w; ++q) wb += wsum[q];
    off[tid] = cur[tid] + wb;
  }
  if (tid == 0) {
    int t = 0;
    for (int q = 0; q < RV_RANGE / 64; ++q) t += wsum[q];
    off[RV_RANGE] = t;
  }
  __syncthreads();
  const long seg0 = gbase + below_s;  // position in rev of the first entry of target r0
  if (tid < nr) rev_ptr[(long)cloud * N + r0 + tid] = (int)(seg0 + off[tid]);
  if (cloud == n_clouds - 1 && r0 + nr == N && tid == 0) rev_ptr[(long)n_clouds * N] = (int)((long)n_clouds * E);
  // sub-ranges [a, b) of targets whose entries fit the LDS buffer
  int a = 0;
  while (a < nr) {
    if (tid == 0) {
      int b = nr;  // the usual case: everything that is left fits
      if (off[nr] - off[a] > RV_CAP) {
        int lo = a, hi = nr;  // largest b with off[b] - off[a] <= RV_CAP (off is non-decreasing)
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (off[mid] - off[a] <= RV_CAP) lo = mid; else hi = mid - 1;
        }
        b = lo;
      }
      b_s = b;
    }
    __syncthreads();
    const int b = b_s;
    if (b == a) {
      // one target with more entries than the buffer: ordered compaction over the cloud's edges
      long outp = seg0 + off[a];
      for (long e0 = 0; e0 < E; e0 += 1024) {
        const long e = e0 + tid;
        const bool hit = e < E && min(max(lst[min(e, E - 1)], 0), N - 1) - r0 == a;
        const unsigned long long m = __ballot(hit);
        if (lane == 0) wsum[w] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
        for (int q = 0; q < 16; ++q) { before += q < w ? wsum[q] : 0; total += wsum[q]; }
        if (hit) rev[outp + before + __popcll(m & ((1ull << lane) - 1ull))] = (int)(gbase + e);
        outp += total;
        __syncthreads();
      }
      a += 1;
      continue;
    }
    if (tid < RV_RANGE) cur[tid] = (tid >= a && tid < b) ? off[tid] - off[a] : 0;
    __syncthreads();
    for (long e0 = tid; e0 < E; e0 += 8 * 1024) {
      int jv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) jv[u] = lst[min(e0 + 1024 * u, E - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = min(max(jv[u], 0), N - 1) - r0;
        if (e0 + 1024 * u < E && j >= a && j < b)
          buf[atomicAdd(&cur[j], 1)] = (int)(e0 + 1024 * u);  // arbitrary order inside a segment: ranked below
      }
    }
    __syncthreads();
    for (int j = a + w; j < b; j += 16) {
      const int s0 = off[j] - off[a], c = cnt[j];
      const long out0 = seg0 + off[j];
      for (int m0 = 0; m0 < c; m0 += 64) {
        const int mine = m0 + lane < c ? buf[s0 + m0 + lane] : 0x7fffffff;
        int rank = 0;
        for (int t0 = 0; t0 < c; t0 += 16) {  // 16 broadcast reads in flight (entries beyond c: clamped, masked)
          int ov[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) ov[t] = buf[s0 + min(t0 + t, c - 1)];
#pragma unroll
          for (int t = 0; t < 16; ++t) rank += (t0 + t < c && ov[t] < mine) ? 1 : 0;
        }
        if (m0 + lane < c) rev[out0 + rank] = (int)(gbase + mine);
      }
    }
    __syncthreads();
    a = b;
  }
}

// ---- the same lists without a sort (round 4): one workgroup of 8 waves per cloud ------------------------------------
// The cloud's E = N K edges are dealt to the waves as 8 CONTIGUOUS ranges, walked in ascending order 64 edges at a time.
//   count   hist[w][j] = edges of wave w's range that name target j (LDS atomics on the wave's own row);
//   scan    off[j] = exclusive scan over j of the totals; hist[w][j] becomes the position of wave w's first entry of j;
//   fill    a wave walks its range again: an edge's position is hist[w][j]++.  The 64 edges of a step belong to at most
//           64 / K + 2 source points and the K neighbours of ONE point are distinct, so the step is done source point by
//           source point: the lanes of one point increment DIFFERENT counters (no two of them race for a position), and
//           the points follow each other in order (LDS operations of a wave execute in order) -- every target's entries
//           come out in ascending edge order by construction, bit for bit the sorted lists of the kernel above, with no
//           ranking pass and the cloud's lists read twice instead of 2 x N / 256 times.  (Lists that repeat a neighbour
//           -- the garbage a non-finite feature cascade can leave, ids clamped to the cloud -- still get one position per
//           edge from the atomics; only the order among the repeats is then unspecified.)
// LDS: 8 N counters (64 KB at N = 2048, 128 KB at N = 4096); larger clouds take the kernel above.
#define RO_WAVES 8
__global__ __launch_bounds__(64 * RO_WAVES) void r3d_edge_reverse_ordered_kernel(const int* __restrict__ idx, int N, int K,
                                                                                 int n_clouds, int* __restrict__ rev_ptr,
                                                                                 int* __restrict__ rev) {
  extern __shared__ int ro_hist[];  // [RO_WAVES][N]
  __shared__ int ro_wsum[RO_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cloud = blockIdx.x;
  const int E = N * K;
  const long gbase = (long)cloud * E;
  const int* lst = idx + gbase;
  const int per = ((E + RO_WAVES - 1) / RO_WAVES + 63) & ~63;  // edges per wave, whole steps
  const int e_beg = min(w * per, E), e_end = min(e_beg + per, E);
  int* mine = ro_hist + w * N;
  for (int i = tid; i < RO_WAVES * N; i += 64 * RO_WAVES) ro_hist[i] = 0;
  __syncthreads();
  // ---- count (8 list entries in flight per lane)
  for (int e0 = e_beg + lane; e0 < e_end; e0 += 8 * 64) {
    int jv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) jv[u] = lst[min(e0 + 64 * u, E - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + 64 * u < e_end) atomicAdd(&mine[min(max(jv[u], 0), N - 1)], 1);
  }
  __syncthreads();
  // ---- scan: thread t owns targets [t * TPT, (t + 1) * TPT)
  const int TPT = (N + 64 * RO_WAVES - 1) / (64 * RO_WAVES);
  const int j0 = tid * TPT;
  int tot = 0;
  for (int j = j0; j < min(j0 + TPT, N); ++j)
#pragma unroll
    for (int q = 0; q < RO_WAVES; ++q) tot += ro_hist[q * N + j];
  int incl = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(incl, o);
    if (lane >= o) incl += y;
  }
  if (lane == 63) ro_wsum[w] = incl;
  __syncthreads();
  int run = incl - tot;  // exclusive over the threads in front of this one
  for (int q = 0; q < w; ++q) run += ro_wsum[q];
  for (int j = j0; j < min(j0 + TPT, N); ++j) {
    rev_ptr[(long)cloud * N + j] = (int)(gbase + run);
#pragma unroll
    for (int q = 0; q < RO_WAVES; ++q) {  // counts -> first positions, wave after wave
      const int c = ro_hist[q * N + j];
      ro_hist[q * N + j] = run;
      run += c;
    }
  }
  if (cloud == n_clouds - 1 && tid == 0) rev_ptr[(long)n_clouds * N] = (int)((long)n_clouds * E);
  __syncthreads();
  // ---- fill, source point by source point inside every step of 64 edges
  for (int e0 = e_beg; e0 < e_end; e0 += 64) {
    const int e = e0 + lane;
    const bool ok = e < e_end;
    const int j = min(max(lst[min(e, E - 1)], 0), N - 1);
    const int src = e / K;
    const int src0 = __builtin_amdgcn_readfirstlane(e0 / K);
    const int src1 = __builtin_amdgcn_readfirstlane(min(e0 + 63, e_end - 1) / K);
    int pos = 0;
    for (int g = src0; g <= src1; ++g)  // (uniform trip count; a lane takes part in exactly one trip)
      if (ok && src == g) pos = atomicAdd(&mine[j], 1);
    if (ok) rev[gbase + pos] = (int)(gbase + e);
  }
}

// ---- backward pass B2: dQ from the point sums of B1, dP by a gather over the incoming edges -----------------------
template <int RT>
__global__ __launch_bounds__(256) void r3d_edgeconv_bwd2_kernel(
    const float* __restrict__ PQ, const float* __restrict__ s1, const float* __restrict__ mean1,
    const float* __restrict__ invstd1, long bn_stride, const float* __restrict__ bn1_sums /* [seg][2][64] */,
    const float* __restrict__ DY1, const float* __restrict__ BE, const float* __restrict__ esum /* null: BE holds sum_t e1-hat */,
    const int* __restrict__ rev_ptr, const int* __restrict__ rev,
    int e_off, EcGeom gm, r3d_segmap cs /* clouds */, int n_chunks, float* __restrict__ dPQ /* (M,128), every entry written */) {
  // rev holds edge ids of the batch the reverse list was built for; this call's clouds start e_off edges into it
  constexpr int K = 4 * RT;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int item = blockIdx.x; item < n_chunks; item += gridDim.x) {
    const int chunk = r3d_xcd_swizzle(item, n_chunks);  // workgroups sharing an L2 walk the chunks of the same clouds
    int cloud, p0, p1;
    gm.range(chunk, cloud, p0, p1);
    const int seg = ec_seg(cs, cloud);
    const long bo = (long)seg * bn_stride;
    const double E = (double)cs.seg_rows(seg) * gm.N * K;
    const float sc1 = s1[bo + lane], mu1 = mean1[bo + lane], is1 = invstd1[bo + lane];
    const float n1 = (float)((double)bn1_sums[(long)seg * 128 + lane] / E), n2 = (float)((double)bn1_sums[(long)seg * 128 + 64 + lane] / E);
    for (int pt_ = p0 + w; pt_ < p1; pt_ += 4) {
      const long pt = pt_;
      const int rb = rev_ptr[pt], re = rev_ptr[pt + 1];
      const float pj = PQ[pt * 128 + lane];
      const float bs = BE[pt * 128 + lane];
      const float es = esum ? (esum[pt * 64 + lane] - (float)K * mu1) * is1 : BE[pt * 128 + 64 + lane];
      float asum = 0.f, gsum = 0.f;
      for (int e0 = rb; e0 < re; e0 += 16) {  // 16 incoming edges = 32 rows in flight per trip; adds in list order
        const int my_e = rev[min(e0 + min(lane, 15), re - 1)];
        float dv[16], qv[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int e = __builtin_amdgcn_readlane(my_e, t) - e_off;
          dv[t] = DY1[(long)e * 64 + lane];
          qv[t] = PQ[(long)(e / K) * 128 + 64 + lane];
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const bool ok = e0 + t < re;
          asum += r3d_keep(dv[t], ok);
          gsum += r3d_keep(qv[t], ok);
        }
      }
      const float cj = (float)(re - rb);
      dPQ[pt * 128 + lane] = sc1 * (asum - cj * n1 - n2 * (is1 * (cj * (pj - mu1) + gsum)));
      dPQ[pt * 128 + 64 + lane] = sc1 * (bs - (float)K * n1 - n2 * es);
    }
  }
}

// ===========================================================================
// C ABI
// ===========================================================================
static int et_check(const char* fn, int B, int N, int K, int clouds_a, int clouds_b) {
  if (B <= 0 || N <= 0 || N % E2_PTS != 0 || K < 4 || K > 32 || K % 4 != 0) {
    r3d_set_error("%s: unsupported shape B=%d N=%d K=%d (N %% 4 == 0, K %% 4 == 0, 4..32)", fn, B, N, K);
    return R3D_ERR_ARG;
  }
  const r3d_segmap cs{clouds_a, clouds_b};
  if (!cs.covers(B)) {
    r3d_set_error("%s: %d clouds are not whole segments of %d + %d clouds", fn, B, clouds_a, clouds_b);
    return R3D_ERR_ARG;
  }
  if ((long)B * N * K >= 0x7fffffffL) {
    r3d_set_error("%s: B*N*K = %ld edges exceed the 31-bit edge ids", fn, (long)B * N * K);
    return R3D_ERR_ARG;
  }
  return 0;
}
static int et_chunks(int B, int N) { return B * EcGeom::make(N).cpc; }
// grid of a persistent chunk loop: all chunks, or -- when workgroups take several -- a multiple of 8 below the cap, so that
// a workgroup's chunks keep its XCD label (r3d_xcd_swizzle)
static int et_grid8(int n_chunks, int cap) { return n_chunks <= cap ? n_chunks : (cap >= 8 ? cap & ~7 : cap); }

// floats of the scratch of the training-mode EdgeConv passes over B clouds of N points: one dW2 partial per workgroup
// and one statistics partial per chunk
extern "C" long r3d_edgeconv_train_ws_words(int B, int N) { return (long)ET_MAXBLK * 4096 + (long)et_chunks(B, N) * 128 + 64; }

// launch the per-segment reduction of [chunk][2][64] partials -> out [seg][2][64]
static void et_reduce_chunks(const float* part, int N, int clouds_a, int clouds_b, int n_seg, float* out, hipStream_t st) {
  const int cpc = EcGeom::make(N).cpc;
  hipLaunchKernelGGL(r3d_part_reduce_kernel, dim3(128 / 64, n_seg), dim3(1024), 0, st, part, clouds_a * cpc, clouds_b * cpc, 128,
                     out, 128L, 128, (float*)nullptr, 0L);
}

// sums_out [seg][2][64] = (sum e1, sum e1^2) over the edges of every segment of clouds (clouds_a, clouds_b alternating;
// clouds_b == 0: segments of clouds_a clouds)
// esum (optional, (B*N, 64)): sum_t e1 of every point -- what r3d_edgeconv_bwd's bf16 x 3 form takes sum_t e1-hat from
extern "C" int r3d_edge_stats1(const float* PQ, const int32_t* idx, int B, int N, int K, int clouds_a, int clouds_b,
                               float* sums_out, float* esum, float* ws, void* stream) {
  R3D_REQUIRE(PQ && idx && sums_out && ws, "r3d_edge_stats1: null pointer");
  int rc = et_check("r3d_edge_stats1", B, N, K, clouds_a, clouds_b);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const EcGeom gm = EcGeom::make(N);
  const int n_chunks = et_chunks(B, N);
  const int grid = et_grid8(n_chunks, 2048);
#define E2_CASE(RT) \
  case RT: hipLaunchKernelGGL(r3d_edge_stats1_kernel<RT>, dim3(grid), dim3(256), 0, st, PQ, idx, gm, n_chunks, ws, esum); break
  switch (K / 4) {
    E2_CASE(1); E2_CASE(2); E2_CASE(3); E2_CASE(4); E2_CASE(5); E2_CASE(6); E2_CASE(7); E2_CASE(8);
  }
#undef E2_CASE
  et_reduce_chunks(ws, N, clouds_a, clouds_b, r3d_segmap{clouds_a, clouds_b}.n_seg(B), sums_out, st);
  R3D_LAUNCH_CHECK("r3d_edge_stats1");
  return R3D_OK;
}

template <int RT>
static int fwd2_launch_rt(int n_chunks, hipStream_t st, const float* PQ, const int32_t* idx, const float* s1, const float* t1,
                          long bn_stride, const float* W2, EcGeom gm, r3d_segmap cs, int32_t* argmax, float* zmax, float* zmin,
                          int32_t* argmin, float* ws) {
  const size_t lds = sizeof(float) * ((size_t)2 * 16 * RT * E2_LD);
  static int resident = 0;
  if (!resident) {
    resident = e2_resident_blocks(r3d_edgeconv_train_fwd2_kernel<RT>, lds, ET_MAXBLK);
    R3D_REQUIRE(resident > 0, "r3d_edgeconv_train_fwd_minmax: cannot reserve %zu B of LDS", lds);
  }
  const int grid = et_grid8(n_chunks, resident);
  hipLaunchKernelGGL(r3d_edgeconv_train_fwd2_kernel<RT>, dim3(grid), dim3(256), lds, st, PQ, idx, s1, t1, bn_stride, W2, gm, cs,
                     n_chunks, argmax, zmax, zmin, argmin, ws);
  return R3D_OK;
}

// One-pass training forward (r3d_edgeconv_train_fwd2_kernel): sums_out [seg][2][64] = (sum z2, sum z2^2) over the edges of
// every segment; zmax / zmin / argmax / argmin (B*N, 64) per point and channel.  s1 / t1 of segment s at + s * bn_stride.
// Follow with r3d_bn_fold_seg and r3d_edge_select.
extern "C" int r3d_edgeconv_train_fwd_minmax(const float* PQ, const int32_t* idx, const float* s1, const float* t1,
                                             long bn_stride, const float* W2, int B, int N, int K, int clouds_a, int clouds_b,
                                             float* zmax, float* zmin, int32_t* argmax, int32_t* argmin, float* sums_out,
                                             float* ws, void* stream) {
  R3D_REQUIRE(PQ && idx && s1 && t1 && W2 && zmax && zmin && argmax && argmin && sums_out && ws,
              "r3d_edgeconv_train_fwd_minmax: null pointer");
  int rc = et_check("r3d_edgeconv_train_fwd_minmax", B, N, K, clouds_a, clouds_b);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const EcGeom gm = EcGeom::make(N);
  const r3d_segmap cs{clouds_a, clouds_b};
  const int n_chunks = et_chunks(B, N);
#define E2_CASE(RT)                                                                                                    \
  case RT:                                                                                                             \
    rc = fwd2_launch_rt<RT>(n_chunks, st, PQ, idx, s1, t1, bn_stride, W2, gm, cs, argmax, zmax, zmin, argmin, ws);      \
    break
  switch (K / 4) {
    E2_CASE(1); E2_CASE(2); E2_CASE(3); E2_CASE(4); E2_CASE(5); E2_CASE(6); E2_CASE(7); E2_CASE(8);
  }
#undef E2_CASE
  if (rc) return rc;
  et_reduce_chunks(ws, N, clouds_a, clouds_b, cs.n_seg(B), sums_out, st);
  R3D_LAUNCH_CHECK("r3d_edgeconv_train_fwd_minmax");
  return R3D_OK;
}

// out[m][c] = lrelu(s2[c] * z + t2[c]) with z = zmax (s2[c] >= 0) or zmin (s2[c] < 0); IN PLACE zmax[m][c] := z and
// argmax[m][c] := position of that edge (what the backward pass consumes).  Equal activations keep the
// lowest edge position, like the reference's max over the K axis (dgcnn.py:78).  s2 / t2 of the row's segment.
__global__ void r3d_edge_select_kernel(float* __restrict__ zmax, const float* __restrict__ zmin, int* __restrict__ argmax,
                                       const int* __restrict__ argmin, const float* __restrict__ s2,
                                       const float* __restrict__ t2, long bn_stride, r3d_segmap sm /* rows */, long M,
                                       float* __restrict__ out, long ldo) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * 64) return;
  const int c = (int)(i & 63);
  // (a wave is one row: its segment is looked up on the scalar unit -- the two integer divisions per ELEMENT were most of
  // this kernel's instructions: 170 us per call at 786 432 rows for 0.4 GB of traffic)
  const long m = (long)__builtin_amdgcn_readfirstlane((int)(i >> 6));
  const long bo = (long)__builtin_amdgcn_readfirstlane(sm.seg_of_row32((int)m)) * bn_stride + c;
  const float sc = s2[bo];
  const bool up = sc >= 0.f;
  const float z = up ? zmax[i] : zmin[i];
  if (!up) { zmax[i] = z; argmax[i] = argmin[i]; }
  out[m * ldo + c] = lrelu(sc * z + t2[bo]);
}
extern "C" int r3d_edge_select(float* zmax, const float* zmin, int32_t* argmax, const int32_t* argmin, const float* s2,
                               const float* t2, long bn_stride, long M, long rows_a, long rows_b, float* out, long ldo,
                               void* stream) {
  R3D_REQUIRE(zmax && zmin && argmax && argmin && s2 && t2 && out && M > 0 && ldo >= 64, "r3d_edge_select: bad arguments");
  const r3d_segmap sm{rows_a, rows_b};
  R3D_REQUIRE(sm.covers(M), "r3d_edge_select: %ld rows are not whole segments of %ld + %ld rows", M, rows_a, rows_b);
  hipLaunchKernelGGL(r3d_edge_select_kernel, dim3(r3d_cdiv(M * 64, 256)), dim3(256), 0, (hipStream_t)stream, zmax, zmin,
                     argmax, argmin, s2, t2, bn_stride, sm, M, out, ldo);
  R3D_LAUNCH_CHECK("r3d_edge_select");
  return R3D_OK;
}

template <int RT>
static int bwd1_launch_rt(int n_chunks, hipStream_t st, const float* PQ, const int32_t* idx, EcBn bn, const float* W2,
                          const float* bn2_sums, const float* dout, long lddo, const int32_t* argmax, EcGeom gm, r3d_segmap cs,
                          float* DY1, float* BE, float* part_dw, float* part_bn, int* grid_out) {
  const size_t lds = sizeof(float) * ((size_t)2 * 16 * RT * E2_LD + 2 * E2_PTS * 64 + 6 * 64);
  static int resident = 0;  // workgroups the chip holds at once (persistent loop over the chunks)
  if (!resident) {
    resident = e2_resident_blocks(r3d_edgeconv_bwd1_kernel<RT>, lds, ET_MAXBLK);
    R3D_REQUIRE(resident > 0, "r3d_edgeconv_bwd: cannot reserve %zu B of LDS", lds);
  }
  const int grid = et_grid8(n_chunks, resident);
  hipLaunchKernelGGL(r3d_edgeconv_bwd1_kernel<RT>, dim3(grid), dim3(256), lds, st, PQ, idx, bn, W2, bn2_sums, dout, lddo, argmax,
                     gm, cs, n_chunks, DY1, BE, part_dw, part_bn);
  *grid_out = grid;
  return R3D_OK;
}

template <int RT>
static int bwd1_bx3_launch_rt(int n_chunks, hipStream_t st, const float* PQ, const int32_t* idx, EcBn bn, const float* W2,
                              const float* bn2_sums, const float* dout, long lddo, const int32_t* argmax, const float* zwin,
                              EcGeom gm, r3d_segmap cs, float* DY1, float* BE, float* part_dw, float* part_bn, int* grid_out) {
  static int resident = 0;  // workgroups the chip holds at once (persistent loop over the chunks); two dW2 partials each
  if (!resident) {
    resident = e2_resident_blocks(r3d_edgeconv_bwd1_bx3_kernel<RT>, EB_LDS_BYTES, ET_MAXBLK / 2);
    R3D_REQUIRE(resident > 0, "r3d_edgeconv_bwd: cannot reserve %d B of LDS", (int)EB_LDS_BYTES);
  }
  const int grid = et_grid8(n_chunks, resident);
  hipLaunchKernelGGL(r3d_edgeconv_bwd1_bx3_kernel<RT>, dim3(grid), dim3(256), EB_LDS_BYTES, st, PQ, idx, bn, W2, bn2_sums, dout,
                     lddo, argmax, zwin, gm, cs, n_chunks, DY1, BE, part_dw, part_bn);
  *grid_out = grid;
  return R3D_OK;
}

// Reverse neighbour list of a layer's kNN lists (used by r3d_edgeconv_bwd): rev_ws = B*N + 1 offsets followed by B*N*K
// edge ids.  Deterministic; no reference counterpart (autograd's scatter-add does this implicitly, dgcnn.py:38).
extern "C" long r3d_edge_reverse_ws_words(int B, int N, int K) { return (long)B * N + 1 + (long)B * N * K + 16; }
extern "C" int r3d_edge_reverse(const int32_t* idx, int B, int N, int K, int32_t* rev_ws, long ws_words, void* stream) {
  R3D_REQUIRE(idx && rev_ws, "r3d_edge_reverse: null pointer");
  R3D_REQUIRE(B > 0 && B <= 65535 && N > 0 && K > 0 && (long)B * N * K < 0x7fffffffL, "r3d_edge_reverse: bad shape B=%d N=%d K=%d",
              B, N, K);
  R3D_REQUIRE(ws_words >= r3d_edge_reverse_ws_words(B, N, K), "r3d_edge_reverse: workspace of %ld words is shorter than "
              "r3d_edge_reverse_ws_words(%d, %d, %d)", ws_words, B, N, K);
  const size_t ro_lds = sizeof(int) * (size_t)RO_WAVES * N;
  static const bool ro_off = getenv("R3D_EDGE_REVERSE_SORT") != nullptr;  // A/B switch: the sorting kernel
  if (!ro_off && ro_lds <= 144 * 1024 && (long)N * K >= 64 * RO_WAVES) {
    static size_t attr = 0;
    if (ro_lds > attr) {
      R3D_REQUIRE(hipFuncSetAttribute((const void*)r3d_edge_reverse_ordered_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)ro_lds) == hipSuccess, "r3d_edge_reverse: cannot reserve %zu B of LDS", ro_lds);
      attr = ro_lds;
    }
    hipLaunchKernelGGL(r3d_edge_reverse_ordered_kernel, dim3(B), dim3(64 * RO_WAVES), ro_lds, (hipStream_t)stream, idx, N, K, B,
                       rev_ws, rev_ws + (long)B * N + 1);
  } else {
    hipLaunchKernelGGL(r3d_edge_reverse_kernel, dim3(r3d_cdiv(N, RV_RANGE), B), dim3(1024), 0, (hipStream_t)stream, idx, N, K, B,
                       rev_ws, rev_ws + (long)B * N + 1);
  }
  R3D_LAUNCH_CHECK("r3d_edge_reverse");
  return R3D_OK;
}

// Backward over all B clouds of a batch.  The BatchNorm vectors of both edge layers (s1 .. invstd2) hold one set per
// segment at + s * bn_stride; bn2_sums [seg][2][64] = (sum dy2, sum dy2*zhat2) of every segment (from the point-level
// winners, computed by the caller with r3d_colstats_seg mode 1 on zmax).  Outputs: dW2 (64,64) summed over the WHOLE batch,
// bn1_sums [seg][2][64] (sum dy1, sum dy1*ehat1), dPQ (B*N,128) (every entry written).  Scratch: DY1 B*N*K*64 floats, BE
// B*N*128 floats, ws r3d_edgeconv_train_ws_words(B, N); rev_ws from r3d_edge_reverse on the same idx.
// zwin (B*N, 64): z2 of every (point, channel)'s max-pool winner as r3d_edge_select left it in zmax; esum (B*N, 64): sum_t e1
// from r3d_edge_stats1.  With both given, N % 8 == 0 and r3d_set_matrix_arith(1) (the default) the edge GEMMs run on the
// bf16 matrix core in three-piece arithmetic (edgeconv_bwd_bx3.h); otherwise (either may be NULL) on the fp32 core.
extern "C" int r3d_edgeconv_bwd(const float* PQ, const int32_t* idx, const float* s1, const float* t1, const float* mean1,
                                const float* invstd1, const float* W2, const float* s2, const float* t2, const float* mean2,
                                const float* invstd2, long bn_stride, const float* bn2_sums, const float* dout, long lddo,
                                const int32_t* argmax, const float* zwin, const float* esum, int B, int N, int K, int clouds_a,
                                int clouds_b, float* DY1, float* BE, const int32_t* rev_ws, float* dW2, float* bn1_sums,
                                float* dPQ, float* ws, void* stream) {
  R3D_REQUIRE(PQ && idx && s1 && t1 && mean1 && invstd1 && W2 && s2 && t2 && mean2 && invstd2 && bn2_sums && dout &&
                  argmax && DY1 && BE && rev_ws && dW2 && bn1_sums && dPQ && ws,
              "r3d_edgeconv_bwd: null pointer");
  int rc = et_check("r3d_edgeconv_bwd", B, N, K, clouds_a, clouds_b);
  if (rc) return rc;
  const EcGeom gm = EcGeom::make(N);
  const r3d_segmap cs{clouds_a, clouds_b};
  const int n_chunks = et_chunks(B, N);
  const int n_seg = cs.n_seg(B);
  hipStream_t st = (hipStream_t)stream;
  float* part_dw = ws;
  float* part_bn = ws + (long)ET_MAXBLK * 4096;
  const EcBn bn{s1, t1, mean1, invstd1, s2, t2, mean2, invstd2, bn_stride};
  int grid = 0;
  const bool bx3 = g_r3d_matrix_arith == 1 && zwin && esum && N % EB_PTS == 0 && (lddo & 3) == 0 && ((uintptr_t)dout & 15) == 0;
#define E2_CASE(RT)                                                                                                        \
  case RT:                                                                                                                 \
    rc = bx3 ? bwd1_bx3_launch_rt<RT>(n_chunks, st, PQ, idx, bn, W2, bn2_sums, dout, lddo, argmax, zwin, gm, cs, DY1, BE,  \
                                      part_dw, part_bn, &grid)                                                             \
             : bwd1_launch_rt<RT>(n_chunks, st, PQ, idx, bn, W2, bn2_sums, dout, lddo, argmax, gm, cs, DY1, BE, part_dw,   \
                                  part_bn, &grid);                                                                         \
    break
  switch (K / 4) {
    E2_CASE(1); E2_CASE(2); E2_CASE(3); E2_CASE(4); E2_CASE(5); E2_CASE(6); E2_CASE(7); E2_CASE(8);
  }
#undef E2_CASE
  if (rc) return rc;
  // dW2 partials: one per workgroup (fp32 form) or one per pair of waves (bf16 x 3 form), added in ascending order
  hipLaunchKernelGGL(r3d_part_reduce_kernel, dim3(4096 / 64, 1), dim3(1024), 0, st, part_dw, bx3 ? 2 * grid : grid, 0, 4096, dW2,
                     0L, 4096, (float*)nullptr, 0L);
  et_reduce_chunks(part_bn, N, clouds_a, clouds_b, n_seg, bn1_sums, st);
  const int32_t* rev_ptr = rev_ws;
  const int32_t* rev = rev_ws + (long)B * N + 1;
  const int grid2 = et_grid8(n_chunks, 4096);
#define E2_CASE(RT)                                                                                                       \
  case RT:                                                                                                                \
    hipLaunchKernelGGL(r3d_edgeconv_bwd2_kernel<RT>, dim3(grid2), dim3(256), 0, st, PQ, s1, mean1, invstd1, bn_stride, bn1_sums, \
                       DY1, BE, bx3 ? esum : (const float*)nullptr, rev_ptr, rev, 0, gm, cs, n_chunks, dPQ);              \
    break
  switch (K / 4) {
    E2_CASE(1); E2_CASE(2); E2_CASE(3); E2_CASE(4); E2_CASE(5); E2_CASE(6); E2_CASE(7); E2_CASE(8);
  }
#undef E2_CASE
  R3D_LAUNCH_CHECK("r3d_edgeconv_bwd");
  return R3D_OK;
}
