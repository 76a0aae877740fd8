// EdgeConv backward pass B1 on the bf16 matrix core in three-piece arithmetic ("bf16 x 3", common.h), gfx950.
// Included by edgeconv_train.hip (same translation unit: EcGeom, EcBn, the launch helpers).
//
// Reference: the autograd backward of models/dgcnn.py:45-61,117-118 in train mode (see the header of edgeconv_train.hip
// for the algebra).  What changes against r3d_edgeconv_bwd1_kernel:
//   * all three edge GEMMs -- the recompute of z2, dh1 = dz2 W2, dW2 += dz2^T h1 -- run as six v_mfma_f32_32x32x16_bf16
//     per fp32 product block (2.67 x the fp32 matrix rate at fp32-level accuracy).  None of them decides an index: the one
//     thing the forward's z2 decided -- which edge wins the max-pool and on which side of the LeakyReLU kink the winner
//     sits -- is read from what the forward stored (argmax, and the winner's z2 itself: r3d_edge_select's zmax), so the
//     recomputed z2 only enters the BatchNorm mean terms, like every other fp32-accurate quantity of the backward;
//   * no LDS transposes between the GEMMs but one: z2^T = W2 h1^T and dh1^T = W2^T dz2^T are computed TRANSPOSED, so the
//     edge index sits on the lane and the channels in the accumulator registers for h1, z2, dz2, dh1 and dy1 alike --
//     the element-wise steps between the GEMMs are register-local -- and the operand images in LDS (32 edges x 64
//     channels per piece, swizzled 128-byte rows) serve row reads (ds_read_b128: contraction over channels) and
//     transposed reads (ds_read_b64_tr_b16: contraction over edges, dW2) alike;
//   * per-point sums of e1-hat come from the forward (r3d_edge_stats1 writes sum_t e1 per point).
//
// Work split.  A workgroup = 4 waves = 2 PAIRS; a pair owns a UNIT of 8 points (8 K edge rows = K / 4 tiles of 32 rows),
// wave h of the pair owns channel tile h (32 of the 64 channels) of every output: z2^T rows (c_out), dh1^T rows (c_in),
// dW2 rows (c_out).  Per tile and wave: 24 + 24 + 24 MFMAs.  A workgroup takes a statistics chunk (32 points = 4 units);
// the pairs only share the W2 image and the chunk's BatchNorm vectors.  Every sum has a fixed order that depends on the
// chunk alone: bit-reproducible, and the same bits whether an episode runs alone or inside a batch.
#pragma once

#define EB_PTS 8   // points per unit
#ifndef EB_PREFETCH
#define EB_PREFETCH 0  // 1: a step's rows are requested one step ahead (32 more live registers; measured: no gain, the
                       // second wave of the SIMD already covers the round trips -- the kernel is bound by vector issue)
#endif
#define EB_YS 36   // words per row of a wave's fp32 staging tile (32 channels + 4: conflict-free 16-byte writes)
#define EB_PIECE 2048  // bf16 per piece image (32 rows x 64 channels)
#define EB_IMG (3 * EB_PIECE)

typedef short eb_s16x4 __attribute__((ext_vector_type(4)));

// chunk c (8 channels) of image row r sits at 16-byte position c ^ eb_swz(r) of the row (attention.hip: ag_swz; row reads
// and transposed reads are both conflict-free on it)
static __device__ __forceinline__ int eb_swz(int r) { return 4 * ((r >> 1) & 1) | ((r >> 2) & 3); }
struct EbOffs {
  int row[4];     // row read, k-step st: image row lane & 31, channels 16 st + 8 half ..
  int col[2][2];  // transposed read, channel tile cc, second index j: image rows 4 half + 8 j + q (+ 16 sI)
};
static __device__ __forceinline__ EbOffs eb_make_offs(int lane) {
  EbOffs o;
  const int k = lane & 31, half = lane >> 5;
#pragma unroll
  for (int st = 0; st < 4; ++st) o.row[st] = k * 64 + 8 * ((2 * st + half) ^ eb_swz(k));
  const int g16 = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
#pragma unroll
  for (int cc = 0; cc < 2; ++cc)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 4 * half + 8 * j + q;  // + 16 sI: eb_swz is periodic in 16 rows
      o.col[cc][j] = r * 64 + 8 * ((4 * cc + 2 * g16 + (p >> 1)) ^ eb_swz(r)) + 4 * (p & 1);
    }
  return o;
}
// B[k = channel 16 st + 8 half + j][col = image row lane & 31] (or A with rows and columns swapped)
static __device__ __forceinline__ r3d_bx3 eb_row_frag(const unsigned short* img, const EbOffs& o, int st) {
  const unsigned short* p = img + o.row[st];
  r3d_bx3 a;
  a.h = *reinterpret_cast<const r3d_u32x4*>(p);
  a.m = *reinterpret_cast<const r3d_u32x4*>(p + EB_PIECE);
  a.l = *reinterpret_cast<const r3d_u32x4*>(p + 2 * EB_PIECE);
  return a;
}
// A[i = channel 32 cc + (lane & 31)][k = image rows 16 sI + 4 half + {0..3, 8..11}] (as B: [k][col = channel])
static __device__ __forceinline__ r3d_bx3 eb_col_frag2(const unsigned short* img, int off0, int off1, int sI) {
  const unsigned short* a0 = img + off0 + 16 * 64 * sI;
  const unsigned short* a1 = img + off1 + 16 * 64 * sI;
  r3d_bx3 f;
  r3d_u32x4* pieces[3] = {&f.h, &f.m, &f.l};
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) {
    const eb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) eb_s16x4*)(a0 + pc * EB_PIECE));
    const eb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) eb_s16x4*)(a1 + pc * EB_PIECE));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    (*pieces[pc])[0] = l2.x; (*pieces[pc])[1] = l2.y; (*pieces[pc])[2] = h2.x; (*pieces[pc])[3] = h2.y;
  }
  return f;
}
static __device__ __forceinline__ r3d_bx3 eb_col_frag(const unsigned short* img, const EbOffs& o, int sI, int cc) {
  return eb_col_frag2(img, o.col[cc][0], o.col[cc][1], sI);  // cc: a compile-time constant at every call
}
// B[k = channels 16 sI + 4 half + {0..3, 8..11} of a 32-channel tile][col = image row lane & 31]: the k order of an accumulator
// lane (and of eb_col_frag); off0 / off1: this lane's two 4-channel groups of its row
static __device__ __forceinline__ r3d_bx3 eb_acc_order_frag(const unsigned short* img, int off0, int off1) {
  r3d_bx3 f;
  r3d_u32x4* pieces[3] = {&f.h, &f.m, &f.l};
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) {
    const uint2 lo = *reinterpret_cast<const uint2*>(img + off0 + pc * EB_PIECE);
    const uint2 hi = *reinterpret_cast<const uint2*>(img + off1 + pc * EB_PIECE);
    (*pieces[pc])[0] = lo.x; (*pieces[pc])[1] = lo.y; (*pieces[pc])[2] = hi.x; (*pieces[pc])[3] = hi.y;
  }
  return f;
}
// four consecutive channels of one image row (a lane's share of one 8-channel chunk), all three pieces
static __device__ __forceinline__ void eb_store4(unsigned short* img, int off, float x0, float x1, float x2, float x3) {
  unsigned h0, m0, l0, h1, m1, l1;
  r3d_bx3_split2(x0, x1, h0, m0, l0);
  r3d_bx3_split2(x2, x3, h1, m1, l1);
  *reinterpret_cast<uint2*>(img + off) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(img + off + EB_PIECE) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(img + off + 2 * EB_PIECE) = make_uint2(l0, l1);
}

// LDS of a workgroup (bytes): W2 image 24576 | per pair: H image 12288, G image 12288 (the waves' fp32 staging tiles alias
// it), unit tables 2048 + 512 | chunk vectors 9 x 256
#define EB_LDS_PAIR (2 * EB_IMG * 2 + 2048 + 512)
#define EB_LDS_W (2 * EB_IMG * 2)
#define EB_LDS_BYTES (EB_LDS_W + 2 * EB_LDS_PAIR + 9 * 256)
enum { EB_S1 = 0, EB_T1, EB_MU1, EB_IS1, EB_MU2, EB_CC, EB_BB, EB_S2, EB_T2 };

#ifdef EB_STAMPS  // phase clocks of two waves (tools/probe/eb_stamps.py); never in the shipped library
__device__ unsigned long long g_eb_dbg[2][16];
extern "C" int r3d_edgeconv_bwd_debug_read(unsigned long long* out32) {
  return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_eb_dbg), sizeof(g_eb_dbg)) == hipSuccess ? 0 : 1;
}
#define EBSTAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); dbg_acc[i] += now_ - dbg_last; dbg_last = now_; } while (0)
#else
#define EBSTAMP(i)
#endif

template <int RT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void r3d_edgeconv_bwd1_bx3_kernel(
    const float* __restrict__ PQ, const int* __restrict__ idx, EcBn bn, const float* __restrict__ W2,
    const float* __restrict__ bn2_sums /* [seg][2][64]: sum dy2, sum dy2 zhat2 */, const float* __restrict__ dout, long lddo,
    const int* __restrict__ argmax, const float* __restrict__ zwin /* (points, 64): z2 of the max-pool winner */, EcGeom gm,
    r3d_segmap cs /* clouds */, int n_chunks, float* __restrict__ DY1 /* (points * K, 64) */,
    float* __restrict__ BE /* (points, 128): sum_t dy1 in the first 64 columns */, float* __restrict__ part_dw /* [2 grid][64 x 64] */,
    float* __restrict__ part_bn /* [chunk][2][64] */) {
  constexpr int K = 4 * RT;
  extern __shared__ __attribute__((aligned(16))) unsigned char eb_smem[];
  unsigned short* Wimg = reinterpret_cast<unsigned short*>(eb_smem);  // two images: rows = c_out (tile, row), channels = c_in
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pr = w >> 1, h = w & 1;        // pair, channel tile of this wave (scalars: the branches on them are uniform)
  const int e = lane & 31, half = lane >> 5;
  const int tid2 = tid & 127;              // thread of the pair
  unsigned char* pbase = eb_smem + EB_LDS_W + pr * EB_LDS_PAIR;
  unsigned short* Himg = reinterpret_cast<unsigned short*>(pbase);
  unsigned short* Gimg = Himg + EB_IMG;
  float* Ytile = reinterpret_cast<float*>(Gimg) + h * (32 * EB_YS);  // wave-private, aliases the G image
  float* G2 = reinterpret_cast<float*>(pbase + 2 * EB_IMG * 2);       // [8][64] s2 * dout * slope of the winner
  unsigned* AM = reinterpret_cast<unsigned*>(pbase + 2 * EB_IMG * 2 + 2048);  // [8][16] winner positions, 4 bytes per word
  float* cst = reinterpret_cast<float*>(eb_smem + EB_LDS_W + 2 * EB_LDS_PAIR);  // [9][64]
  float* red = G2;  // chunk-end exchange [2][64] per pair (the unit tables are free by then)
  const int N = gm.N;

  // ---- once per workgroup: W2 as two dual-use images (image row = c_out, channel = c_in).  Row reads give the A operand of
  // z2^T = W2 h1^T (k = c_in, natural order), transposed reads the A operand of dh1^T = W2^T dz2^T (k = c_out in the order
  // an accumulator lane holds its rows: 16 sI + 4 half + {0..3, 8..11}) -- no W2^T copy, no weight fragment in registers
  for (int it = tid; it < 64 * 8; it += 256) {
    const int co = it >> 3, c = it & 7;  // row c_out, chunk of 8 c_in
    const float4 a = *reinterpret_cast<const float4*>(W2 + co * 64 + 8 * c), b = *reinterpret_cast<const float4*>(W2 + co * 64 + 8 * c + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const r3d_bx3 f = r3d_bx3_split8(v);
    unsigned short* d = Wimg + (co >> 5) * EB_IMG + (co & 31) * 64 + 8 * (c ^ eb_swz(co & 31));
    *reinterpret_cast<r3d_u32x4*>(d) = f.h;
    *reinterpret_cast<r3d_u32x4*>(d + EB_PIECE) = f.m;
    *reinterpret_cast<r3d_u32x4*>(d + 2 * EB_PIECE) = f.l;
  }
  const EbOffs offs = eb_make_offs(lane);
  const int colh0 = h ? offs.col[1][0] : offs.col[0][0], colh1 = h ? offs.col[1][1] : offs.col[0][1];  // this wave's own tile
  // 4-channel groups of image row e: channels 32 ct + 8 q + 4 half ..; tile ct = h are the ones this lane writes
  int gofs[2][4];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int q = 0; q < 4; ++q) gofs[ct][q] = e * 64 + 8 * ((4 * ct + q) ^ eb_swz(e)) + 4 * half;
  int wofs[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) wofs[q] = h ? gofs[1][q] : gofs[0][q];
  const int cl0 = 32 * h + 4 * half;  // + 8 q + i: the channels this lane holds in every accumulator / gather register
  f32x16 dw[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dw[nt][r] = 0.f;
  const int r4 = lane >> 4, c2 = lane & 15;  // scan layout: row 4 g + r4, channels 32 h + 2 c2, + 1
#ifdef EB_STAMPS
  unsigned long long dbg_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_last = __builtin_amdgcn_s_memtime();
  const unsigned long long dbg_t0 = dbg_last, dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif

  for (int item = blockIdx.x; item < n_chunks; item += gridDim.x) {
    const int chunk = r3d_xcd_swizzle(item, n_chunks);
    int cloud, p0, p1;
    gm.range(chunk, cloud, p0, p1);
    const long cloud0 = (long)cloud * N;
    const int seg = ec_seg(cs, cloud);
    const long bo = (long)seg * bn.stride;
    __syncthreads();  // the previous chunk's last reads of cst / red are done (and the W2 image is complete)
    if (tid < 64) {
      const double E = (double)cs.seg_rows(seg) * N * K;  // edges of the segment
      const float s2 = bn.s2[bo + tid], is2 = bn.invstd2[bo + tid];
      const float m1 = (float)((double)bn2_sums[(long)seg * 128 + tid] / E);
      const float m2 = (float)((double)bn2_sums[(long)seg * 128 + 64 + tid] / E);
      cst[EB_S1 * 64 + tid] = bn.s1[bo + tid];
      cst[EB_T1 * 64 + tid] = bn.t1[bo + tid];
      cst[EB_MU1 * 64 + tid] = bn.mean1[bo + tid];
      cst[EB_IS1 * 64 + tid] = bn.invstd1[bo + tid];
      cst[EB_MU2 * 64 + tid] = bn.mean2[bo + tid];
      cst[EB_CC * 64 + tid] = (s2 * is2) * m2;  // dz2 = s2 dy - s2 m1 - (z - mu2) (s2 is2 m2)
      cst[EB_BB * 64 + tid] = s2 * m1;
      cst[EB_S2 * 64 + tid] = s2;
      cst[EB_T2 * 64 + tid] = bn.t2[bo + tid];
    }
    __syncthreads();
    f32x16 sye;  // sum over this lane's edges of dy1 * (e1 - mean1), channels as in the accumulators
#pragma unroll
    for (int r = 0; r < 16; ++r) sye[r] = 0.f;
    float sdy0 = 0.f, sdy1 = 0.f;  // sum of dy1 over the pair's points (scan layout; lanes r4 == 0 hold the totals)
    // The pair's two units as ONE loop over 2 RT tile steps, software-pipelined over the gathers: the rows a step cuts were
    // requested one step earlier, the neighbour index they depend on two steps earlier -- the L2 / HBM round trips of a
    // step's 9 dependent loads lie under the previous step's MFMAs.  Steps past the end (and absent units of a chunk's
    // tail: N % 32 != 0) request rows of the chunk's first unit again, which nobody uses.
    const int u_first = 2 * pr;
    auto step_unit0 = [&](int st_) {  // first point of step st_'s unit (clamped into the chunk)
      const int u = u_first + (st_ >= RT ? 1 : 0);
      const int up = p0 + EB_PTS * u;
      return up < p1 ? up : p0;
    };
    auto load_idx = [&](int st_) {
      st_ = min(st_, 2 * RT - 1);
      const int up = step_unit0(st_), tl = st_ >= RT ? st_ - RT : st_;
      return idx[(long)up * K + 32 * tl + e];
    };
    float4 pv[4], qv[4];
    auto load_rows = [&](int st_, int jraw) {
      st_ = min(st_, 2 * RT - 1);
      const int up = step_unit0(st_), tl = st_ >= RT ? st_ - RT : st_;
      const int row_ = 32 * tl + e;
      const int j = min(max(jraw, 0), N - 1);  // never gather outside the cloud
      const float* prow = PQ + (cloud0 + j) * 128 + cl0;
      const float* qrow = PQ + (long)(up + row_ / K) * 128 + 64 + cl0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pv[q] = *reinterpret_cast<const float4*>(prow + 8 * q);
        qv[q] = *reinterpret_cast<const float4*>(qrow + 8 * q);
      }
    };
    int j_next = load_idx(0);
    if (EB_PREFETCH) {
      load_rows(0, j_next);
      j_next = load_idx(1);
    }
    float bs0 = 0.f, bs1 = 0.f;  // running per-point sum of dy1 (scan layout)
    for (int step = 0; step < 2 * RT; ++step) {
      const int ui = step >= RT ? 1 : 0, tile = step - RT * ui;
      const int upt0 = p0 + EB_PTS * (u_first + ui);  // first point of the unit (global row)
      const bool valid = upt0 < p1;                   // (N % 8 == 0: a unit is whole or absent); wave-uniform
      if (tile == 0 && valid) {  // unit tables: 8 points x 64 channels, 4 channels per thread of the pair
        const int pt = tid2 >> 4, c4 = 4 * (tid2 & 15);
        const long prow = upt0 + pt;
        const float4 dv = *reinterpret_cast<const float4*>(dout + prow * lddo + c4);
        const int4 av = *reinterpret_cast<const int4*>(argmax + prow * 64 + c4);
        const float4 zv = *reinterpret_cast<const float4*>(zwin + prow * 64 + c4);
        const float4 s2v = *reinterpret_cast<const float4*>(cst + EB_S2 * 64 + c4);
        const float4 t2v = *reinterpret_cast<const float4*>(cst + EB_T2 * 64 + c4);
        float4 g;  // s2 * dy2 of the winner: dout times the LeakyReLU slope on the winner's side of the kink
        g.x = s2v.x * (dv.x * ((s2v.x * zv.x + t2v.x) > 0.f ? 1.f : 0.2f));
        g.y = s2v.y * (dv.y * ((s2v.y * zv.y + t2v.y) > 0.f ? 1.f : 0.2f));
        g.z = s2v.z * (dv.z * ((s2v.z * zv.z + t2v.z) > 0.f ? 1.f : 0.2f));
        g.w = s2v.w * (dv.w * ((s2v.w * zv.w + t2v.w) > 0.f ? 1.f : 0.2f));
        const float4 bbv = *reinterpret_cast<const float4*>(cst + EB_BB * 64 + c4);
        g.x -= bbv.x; g.y -= bbv.y; g.z -= bbv.z; g.w -= bbv.w;  // (the table holds s2 dy2 - s2 m1: one select per element later)
        *reinterpret_cast<float4*>(G2 + pt * 64 + c4) = g;
        AM[pt * 16 + (c4 >> 2)] = (unsigned)(av.x & 255) | ((unsigned)(av.y & 255) << 8) | ((unsigned)(av.z & 255) << 16) |
                                  ((unsigned)(av.w & 255) << 24);
      }
      {
        const int row = 32 * tile + e;          // edge row of the unit
        const int ptl = row / K, t = row - ptl * K;
        float eh[16], sl[16];  // e1 - mean1 and the LeakyReLU-1 slope of this lane's 16 (edge, channel) elements
        {
          if (!EB_PREFETCH) {
            load_rows(step, j_next);
            j_next = load_idx(step + 1);  // (the neighbour index of the next step: one round trip less on its path)
          }
          // ---- h1 = lrelu(s1 (P[j] + Q[i]) + t1), cut into the H image
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 s1v = *reinterpret_cast<const float4*>(cst + EB_S1 * 64 + cl0 + 8 * q);
            const float4 t1v = *reinterpret_cast<const float4*>(cst + EB_T1 * 64 + cl0 + 8 * q);
            const float4 muv = *reinterpret_cast<const float4*>(cst + EB_MU1 * 64 + cl0 + 8 * q);
            const float pa[4] = {pv[q].x, pv[q].y, pv[q].z, pv[q].w}, qa[4] = {qv[q].x, qv[q].y, qv[q].z, qv[q].w};
            const float s1a[4] = {s1v.x, s1v.y, s1v.z, s1v.w}, t1a[4] = {t1v.x, t1v.y, t1v.z, t1v.w};
            const float mua[4] = {muv.x, muv.y, muv.z, muv.w};
            float hv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float e1 = pa[i] + qa[i];
              eh[4 * q + i] = e1 - mua[i];  // e1-hat without its factor invstd1: applied once per chunk to the sum
              const float u = s1a[i] * e1 + t1a[i];
              sl[4 * q + i] = u > 0.f ? 1.f : 0.2f;
              hv[i] = u * sl[4 * q + i];  // == lrelu(u) bit for bit
            }
            eb_store4(Himg, wofs[q], hv[0], hv[1], hv[2], hv[3]);
          }
          if (EB_PREFETCH) {  // request the next step's rows (their index arrived meanwhile) and the index of the step after
            load_rows(step + 1, j_next);
            j_next = load_idx(step + 2);
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // (the requests stay here: the scheduler would sink them to their first use)
        EBSTAMP(0);
        __syncthreads();  // B1: the H image (and, on the unit's first tile, its tables) complete
        EBSTAMP(1);
        f32x16 acc;
        if (valid) {
          // ---- z2^T tile h = W2[tile h] h1^T
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int st = 0; st < 4; ++st)
            acc = r3d_bx3_mma(eb_row_frag(Wimg + h * EB_IMG, offs, st), eb_row_frag(Himg, offs, st), acc);
          EBSTAMP(2);
          // ---- dz2 = s2 dy2 - s2 m1 - (z2 - mu2) s2 is2 m2, dy2 non-zero on the winner edge only; cut into the G image
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned am = AM[ptl * 16 + 8 * h + 2 * q + half];
            const float4 gv = *reinterpret_cast<const float4*>(G2 + ptl * 64 + cl0 + 8 * q);
            const float4 m2v = *reinterpret_cast<const float4*>(cst + EB_MU2 * 64 + cl0 + 8 * q);
            const float4 ccv = *reinterpret_cast<const float4*>(cst + EB_CC * 64 + cl0 + 8 * q);
            const float4 bbv = *reinterpret_cast<const float4*>(cst + EB_BB * 64 + cl0 + 8 * q);
            const float ga[4] = {gv.x, gv.y, gv.z, gv.w}, m2a[4] = {m2v.x, m2v.y, m2v.z, m2v.w};
            const float cca[4] = {ccv.x, ccv.y, ccv.z, ccv.w}, nba[4] = {-bbv.x, -bbv.y, -bbv.z, -bbv.w};
            float dz[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float dyb = (int)((am >> (8 * i)) & 255u) == t ? ga[i] : nba[i];  // s2 dy2 - s2 m1
              dz[i] = dyb - (acc[4 * q + i] - m2a[i]) * cca[i];
            }
            eb_store4(Gimg, wofs[q], dz[0], dz[1], dz[2], dz[3]);
          }
        }
        EBSTAMP(3);
        __syncthreads();  // B2: the G image complete
        EBSTAMP(4);
        if (valid) {
          // ---- dh1^T tile h = W2^T[tile h] dz2^T: W2^T by transposed reads of the W2 images, dz2 in the matching k order
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int sI = 0; sI < 2; ++sI)
              acc = r3d_bx3_mma(eb_col_frag2(Wimg + ct * EB_IMG, colh0, colh1, sI),
                                eb_acc_order_frag(Gimg, gofs[ct][2 * sI], gofs[ct][2 * sI + 1]), acc);
          // ---- dW2 rows of tile h += dz2^T h1 over the tile's 32 edges
#pragma unroll
          for (int sI = 0; sI < 2; ++sI) {
            const r3d_bx3 a = eb_col_frag2(Gimg, colh0, colh1, sI);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) dw[nt] = r3d_bx3_mma(a, eb_col_frag(Himg, offs, sI, nt), dw[nt]);
          }
        }
        EBSTAMP(5);
        __syncthreads();  // B3: every read of the two images is done (the staging tiles alias G; the next step writes H)
        EBSTAMP(6);
        if (valid) {
          // ---- dy1 = dh1 lrelu'(u1): sums in registers, rows out through the wave's staging tile
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              y[i] = acc[4 * q + i] * sl[4 * q + i];
              sye[4 * q + i] = __builtin_fmaf(y[i], eh[4 * q + i], sye[4 * q + i]);
            }
            *reinterpret_cast<float4*>(Ytile + e * EB_YS + 8 * q + 4 * half) = make_float4(y[0], y[1], y[2], y[3]);
          }
          // (the wave reads back what it wrote itself: LDS operations of one wave execute in order)
          float* drow = DY1 + ((long)upt0 * K + 32 * tile + r4) * 64 + 32 * h + 2 * c2;
          int rows_in_point = (32 * tile) % K;  // rows of the current point in front of this tile (scalar)
          int pidx = upt0 + (32 * tile) / K;
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            const float2 v = *reinterpret_cast<const float2*>(Ytile + (4 * g + r4) * EB_YS + 2 * c2);
            *reinterpret_cast<float2*>(drow + (long)(4 * g) * 64) = v;
            bs0 += v.x;
            bs1 += v.y;
            rows_in_point += 4;
            if (rows_in_point == K) {  // (uniform) a point ends every K rows, K % 4 == 0
              float t0 = bs0 + __shfl_xor(bs0, 16), t1 = bs1 + __shfl_xor(bs1, 16);
              t0 += __shfl_xor(t0, 32);
              t1 += __shfl_xor(t1, 32);
              if (r4 == 0) *reinterpret_cast<float2*>(BE + (long)pidx * 128 + 32 * h + 2 * c2) = make_float2(t0, t1);
              sdy0 += t0;
              sdy1 += t1;
              bs0 = bs1 = 0.f;
              rows_in_point = 0;
              ++pidx;
            }
          }
        }
        EBSTAMP(7);
#ifdef EB_STAMPS
        dbg_acc[10] += 1;
#endif
      }
    }
    // ---- the chunk's BatchNorm-1 partial: pair 0 + pair 1, in that order
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = sye[r];
      v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
      sye[r] = v;
    }
    __syncthreads();  // both pairs are done with their unit tables (red aliases them)
    if (e == 0) {  // sum dy1 e1-hat = invstd1 * sum dy1 (e1 - mean1)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = cl0 + 8 * (r >> 2) + (r & 3);
        red[64 + c] = sye[r] * cst[EB_IS1 * 64 + c];
      }
    }
    if (r4 == 0) *reinterpret_cast<float2*>(red + 32 * h + 2 * c2) = make_float2(sdy0, sdy1);
    __syncthreads();
    if (tid < 128) {
      const float* r0 = reinterpret_cast<const float*>(eb_smem + EB_LDS_W + 2 * EB_IMG * 2);
      const float* r1 = reinterpret_cast<const float*>(eb_smem + EB_LDS_W + EB_LDS_PAIR + 2 * EB_IMG * 2);
      part_bn[(long)chunk * 128 + tid] = r0[tid] + r1[tid];
    }
  }
#ifdef EB_STAMPS
  if (blockIdx.x == 9 && lane == 0 && (w == 0 || w == 3)) {
    dbg_acc[8] = __builtin_amdgcn_s_memtime() - dbg_t0;
    dbg_acc[9] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    dbg_acc[11] = gridDim.x;
    for (int i = 0; i < 12; ++i) g_eb_dbg[w ? 1 : 0][i] = dbg_acc[i];
  }
#endif
  // dW2 partial of this wave's pair: rows 32 h + ..., all 64 columns
  float* mypart = part_dw + ((long)blockIdx.x * 2 + pr) * 4096;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) mypart[(32 * h + r3d_acc_row(r, lane)) * 64 + 32 * nt + e] = dw[nt][r];
}
