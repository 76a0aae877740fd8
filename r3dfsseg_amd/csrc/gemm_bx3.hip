// The GEMMs that decide no index -- the 1x1 convolutions (forward and input gradient) and the weight-gradient GEMM -- on
// the bf16 matrix core with fp32 accuracy ("bf16 x 3", common.h): every fp32 operand is cut by truncation into three bf16
// pieces (exactly), a product block is six v_mfma_f32_32x32x16_bf16 accumulating in fp32; 6 x 32 cycles cover K = 16
// where v_mfma_f32_32x32x2_f32 needs 8 x 64 (2.67 x the fp32 matrix rate, relative error of a product 2^-22).
//
// Replaces, when r3d_set_matrix_arith(1) (the default) and the shapes allow it, the fp32 kernels of gemm.hip
// (r3d_pointwise_gemm_kernel: models/dgcnn.py:64-80,121-122, models/mpti.py:18-40, models/attention.py:39-41) and of
// train_ops.hip (r3d_gemm_tn_kernel: the weight gradients autograd computes for those layers).  Index-deciding arithmetic
// -- kNN scores, the EdgeConv edge GEMM whose max-pool winners route the gradient, FPS -- stays on the fp32 core.
//
// Round 2 tried this arithmetic on these GEMMs and dropped it: the operands were cut by a separate pack pass through
// HBM (6 B per element written and read back).  Here the cut happens in registers between the global load and the MFMA
// (pointwise kernel: X never touches LDS) or the LDS store of the staging step (weight-gradient kernel).
#if defined(GB_ABL) && (GB_ABL & 4)  // probe builds (tools/probe/gemm_bx3_abl.hip): 1 no stores, 2 no cut, 4 no MFMA, 8 no loads
#define ATT_ABL 4
#endif
#include "common.h"
#include <mutex>
#include <type_traits>
#include <vector>
#ifndef GB_ABL
#define GB_ABL 0
#endif

#define GB_RS 40  // LDS row stride in bf16 units (64 B of data + 16 B: consecutive rows start 20 banks apart)

enum { GB_ACT_NONE = 0, GB_ACT_RELU = 1, GB_ACT_LRELU02 = 2 };

// one staged chunk: 8 consecutive k of a row, as three 16-byte piece vectors
static __device__ __forceinline__ void gb_store_chunk(unsigned short* __restrict__ planes, int rows, int row, int kc,
                                                      const float (&v)[8]) {
#if GB_ABL & 2
  r3d_bx3 f;
  for (int i = 0; i < 4; ++i) { f.h[i] = __float_as_uint(v[i]); f.m[i] = __float_as_uint(v[4 + i]); f.l[i] = f.h[i] ^ f.m[i]; }
#else
  const r3d_bx3 f = r3d_bx3_split8(v);
#endif
  unsigned short* p = planes + row * GB_RS + 8 * kc;
  *reinterpret_cast<r3d_u32x4*>(p) = f.h;
  *reinterpret_cast<r3d_u32x4*>(p + rows * GB_RS) = f.m;
  *reinterpret_cast<r3d_u32x4*>(p + 2 * rows * GB_RS) = f.l;
}
static __device__ __forceinline__ r3d_bx3 gb_load_frag(const unsigned short* __restrict__ planes, int rows, int row, int ks,
                                                       int g) {
  const unsigned short* p = planes + row * GB_RS + 16 * ks + 8 * g;
  r3d_bx3 f;
  f.h = *reinterpret_cast<const r3d_u32x4*>(p);
  f.m = *reinterpret_cast<const r3d_u32x4*>(p + rows * GB_RS);
  f.l = *reinterpret_cast<const r3d_u32x4*>(p + 2 * rows * GB_RS);
  return f;
}

// the 64 x 64 quarter of a wave over one staged K-step of 32: 2 k-steps x (2 x 2 tiles) x 6 MFMAs
static __device__ __forceinline__ void gb_wave_mma(const unsigned short* __restrict__ As, int a_rows, int a_row0,
                                                   const unsigned short* __restrict__ Bs, int b_rows, int b_row0, int lane,
                                                   f32x16 (&acc)[2][2]) {
  const int j = lane & 31, g = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    r3d_bx3 a[2], b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      a[t] = gb_load_frag(As, a_rows, a_row0 + 32 * t + j, ks, g);
      b[t] = gb_load_frag(Bs, b_rows, b_row0 + 32 * t + j, ks, g);
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = r3d_bx3_mma(a[tm], b[tn], acc[tm][tn]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Out[m][j] = act(scale[j] * sum_k X[m][k] W[j][k] + shift[j])   (+ Out if accumulate); K % 32 == 0, rows of X 16-byte
// aligned.  stats_part (optional): [ceil(M / 64)][2][Co] column sums (sum v, sum v^2) of every 64-row tile: the
// BatchNorm batch statistics from the epilogue, partitioned exactly like the fp32 kernel's.
//
// A workgroup = 4 waves x 64 rows, all on the same NT x 32 columns.  The MFMA A operand of a lane -- 8 consecutive k of
// ONE row of X -- is 32 contiguous bytes of global memory: every wave loads its own rows' operands straight into
// registers (two dwordx4 per lane and 32-row tile, one K-step of 16 ahead) and cuts them there into the three
// fragments; X never passes through LDS and no wave waits for another wave's share of it.  Only the W tile, which the
// four waves share, is staged: cut by the workgroup (8 elements per thread and K-step) into LDS in fragment order
// ([n-tile][piece][lane] blocks of 1 KB: a B operand is one lane-linear ds_read_b128), double buffered, ONE barrier per
// K-step.  Per K-step and wave: 132 VALU instructions of cutting for 48 MFMAs (NT = 4).
// ---------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void r3d_pointwise_gemm_bx3_kernel(
    const float* __restrict__ X, long ldx, const float* __restrict__ W, int M, int K, int Co,
    const float* __restrict__ scale, const float* __restrict__ shift, int act, float* __restrict__ Out, long ldo,
    int accumulate, float* __restrict__ stats_part) {
  constexpr int BM = 256, BN = 32 * NT;
  // one K-step = 32 k = two MFMA steps s.  The k owned by a lane's half g are the 16 CONSECUTIVE ones [16 g, 16 g + 16)
  // of the step (MFMA step s takes 16 g + 8 s + 0..7): a lane reads 64 contiguous bytes of its row per K-step, the two
  // halves of a row one whole 128-byte line.  (Any assignment of k to the MFMA's contraction slots is valid as long as
  // both operands use it: W is staged accordingly.)
  __shared__ __attribute__((aligned(16))) unsigned short Bs[2][2][NT][3][64 * 8];  // [buffer][s][n-tile][piece][lane]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, g = lane >> 5;
  // XCD-aware tile order, the column tiles of one row tile next to each other (gemm.hip)
  const int tiles_n = (Co + BN - 1) / BN;
  const int tile = r3d_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
  const int tile_m = tile / tiles_n;
  const long mw = (long)tile_m * BM + 64 * w;  // the wave's 64 rows
  const int n0 = (tile - tile_m * tiles_n) * BN;
  // Operands come through buffer loads: rows beyond M / Co read as zeros (the descriptor's range check), and -- the
  // reason for using them -- the compiler keeps these intrinsics where they are written, where it gathers plain loads
  // of neighbouring addresses in one place and then waits for all of them at once.
  const long m0 = (long)tile_m * BM;
  const long arows = min((long)BM, (long)M - m0);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(X + m0 * ldx), 0, (int)(((arows - 1) * ldx + K) * 4), 0x00020000);
  const int wrows = min(BN, Co - n0);
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void*)(W + (long)n0 * K), 0, wrows * K * 4, 0x00020000);
  int oa[2];  // byte offsets of the lane's operands inside the tile
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) oa[tm] = (int)(((64 * w + 32 * tm + j) * ldx + 16 * g) * 4);
  // W staging: BN rows x 4 chunks of 8 k per K-step, chunk c = 2 g + s; thread t cuts chunk (t & 3) of rows (t >> 2) + 64 u
  constexpr int WU = BN / 64;
  const int wc = tid & 3;
  int ow[WU];
  unsigned short* wdst[WU];
#pragma unroll
  for (int u = 0; u < WU; ++u) {
    const int wrow = (tid >> 2) + 64 * u;
    ow[u] = (wrow * K + 8 * wc) * 4;
    wdst[u] = &Bs[0][wc & 1][wrow >> 5][0][((wrow & 31) + 32 * (wc >> 1)) * 8];
  }
  constexpr int BUF = 2 * NT * 3 * 64 * 8;  // bf16 per buffer
  float araw[2][2][8], wraw[WU][8];        // araw[s][tm]
  auto ld8 = [&](const __amdgpu_buffer_rsrc_t r, int voff, int soff, float (&d)[8]) {
    const r3d_u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    const r3d_u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16, soff, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { d[i] = __uint_as_float(a[i]); d[4 + i] = __uint_as_float(b[i]); }
  };
  auto aload = [&](int k, int s) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) ld8(ra, oa[tm] + 32 * s, k * 4, araw[s][tm]);
  };
  auto asplit = [&](int s, r3d_bx3 (&fa)[2]) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#if GB_ABL & 2
      for (int i = 0; i < 4; ++i) { fa[tm].h[i] = __float_as_uint(araw[s][tm][i]); fa[tm].m[i] = __float_as_uint(araw[s][tm][4 + i]); fa[tm].l[i] = fa[tm].h[i] ^ fa[tm].m[i]; }
#else
      fa[tm] = r3d_bx3_split8(araw[s][tm]);
#endif
    }
  };
  auto wload = [&](int k) {
#pragma unroll
    for (int u = 0; u < WU; ++u) ld8(rw, ow[u], k * 4, wraw[u]);
  };
  auto wstore = [&](int buf) {
#pragma unroll
    for (int u = 0; u < WU; ++u) {
#if GB_ABL & 2
      r3d_bx3 f;
      for (int i = 0; i < 4; ++i) { f.h[i] = __float_as_uint(wraw[u][i]); f.m[i] = __float_as_uint(wraw[u][4 + i]); f.l[i] = f.h[i] ^ f.m[i]; }
#else
      const r3d_bx3 f = r3d_bx3_split8(wraw[u]);
#endif
      unsigned short* d = wdst[u] + buf * BUF;
      *reinterpret_cast<r3d_u32x4*>(d) = f.h;
      *reinterpret_cast<r3d_u32x4*>(d + 64 * 8) = f.m;
      *reinterpret_cast<r3d_u32x4*>(d + 2 * 64 * 8) = f.l;
    }
  };
  auto mma = [&](int buf, int s, const r3d_bx3 (&fa)[2], f32x16 (&acc)[2][NT]) {
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {  // (no branch in the K loop: the wait counts below must be exact, see the loop)
      r3d_bx3 fb;
      const unsigned short* bsrc = &Bs[buf][s][tn][0][lane * 8];
      fb.h = *reinterpret_cast<const r3d_u32x4*>(bsrc);
      fb.m = *reinterpret_cast<const r3d_u32x4*>(bsrc + 64 * 8);
      fb.l = *reinterpret_cast<const r3d_u32x4*>(bsrc + 2 * 64 * 8);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) acc[tm][tn] = r3d_bx3_mma(fa[tm], fb, acc[tm][tn]);
    }
  };
  f32x16 acc[2][NT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  wload(0);
  aload(0, 0);
  aload(0, 1);
  wstore(0);
  __syncthreads();
  // The loop body is free of branches on purpose: behind a conditional load the compiler can no longer count the
  // loads in flight and waits for ALL of them (s_waitcnt vmcnt(0)) before touching the previous step's registers --
  // which makes the prefetch synchronous.  The last step therefore requests its own operands once more (k clamped)
  // and stages them into the buffer nobody reads again.
  for (int k = 0, it = 0; k < K; k += 32, ++it) {
    const int cur = it & 1;
    const int kn = (GB_ABL & 8) ? 0 : min(k + 32, K - 32);
    // the next K-step's operands are requested as soon as their registers are free, one K-step of MFMAs ahead of their use
    r3d_bx3 fa0[2], fa1[2];
    asplit(0, fa0);
    wload(kn);
    aload(kn, 0);
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler would sink the loads to just in front of their use)
    mma(cur, 0, fa0, acc);
    asplit(1, fa1);
    aload(kn, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(cur, 1, fa1, acc);
    wstore(cur ^ 1);
    __syncthreads();
  }
  // Epilogue.  An accumulator register holds ONE column for the lane: stored as it lies, a wave instruction writes two
  // 128-byte pieces of two rows, and the chip took 190 of this kernel's 450 us (192 -> 512 channels) to absorb the
  // output that way.  So the wave's tile goes through LDS (the W buffers, free after the last barrier; wave-private
  // regions, no barrier) 8 rows at a time and leaves as 16 bytes per lane, whole rows of BN columns per instruction:
  // 290 us for the same layer (profiles/r03_experiments.md).
  constexpr int ER = BN + 4;  // floats per staged row
  float* es = reinterpret_cast<float*>(&Bs[0][0][0][0][0]) + w * 8 * ER;
  constexpr int LPR = BN / 4;          // lanes per row of the tile (16 bytes each)
  constexpr int RPI = 64 / LPR;        // rows per store instruction
  const int erow = lane / LPR, ecol = 4 * (lane % LPR);
  const bool cok = n0 + ecol < Co;     // (Co % 4 == 0: a lane's four columns are inside or outside together)
  // (column sums per 32-row half first, the halves added lower first: the association of the packed-W kernel below, whose
  // 64-row statistics tile is two waves -- which of the two kernels a launch takes must not change a bit of the result)
  float sc[NT], sh[NT], s1[2][NT], s2[2][NT];
#pragma unroll
  for (int tn = 0; tn < NT; ++tn) {
    const int jc = n0 + 32 * tn + j;
    sc[tn] = (scale && jc < Co) ? scale[jc] : 1.f;
    sh[tn] = (shift && jc < Co) ? shift[jc] : 0.f;
    s1[0][tn] = s2[0][tn] = s1[1][tn] = s2[1][tn] = 0.f;
  }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // registers 4q .. 4q+3: rows 32 tm + 8 q + 4 g + (0..3)
      const long mrow0 = mw + 32 * tm + 8 * q;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = sc[tn] * acc[tm][tn][4 * q + i] + sh[tn];
          if (act == GB_ACT_RELU) v = fmaxf(v, 0.f);
          else if (act == GB_ACT_LRELU02) v = fmaxf(v, 0.2f * v);
          es[(4 * g + i) * ER + 32 * tn + j] = v;
          {  // column sums of the values written (training: batch statistics of z); rows beyond M add zeros
            const float u = r3d_keep(v, mrow0 + 4 * g + i < M);
            s1[tm][tn] += u;
            s2[tm][tn] += u * u;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 8 / RPI; ++i) {
        const int row = erow + RPI * i;
        const long m = mrow0 + row;
        const float4 v = *reinterpret_cast<const float4*>(&es[row * ER + ecol]);
        if (m < M && cok && !(GB_ABL & 1)) {
          float4* dst = reinterpret_cast<float4*>(&Out[m * ldo + n0 + ecol]);
          if (accumulate) {
            const float4 o = *dst;
            *dst = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
          } else {
            *dst = v;
          }
        }
      }
    }
  }
  if (stats_part && mw < M) {  // uniform.  The rows of one column sit in lanes l and l ^ 32; the wave owns its 64-row tile
    const long t64 = mw >> 6;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      const int jc = n0 + 32 * tn + j;
      const float a1 = (s1[0][tn] + __shfl_xor(s1[0][tn], 32)) + (s1[1][tn] + __shfl_xor(s1[1][tn], 32));
      const float a2 = (s2[0][tn] + __shfl_xor(s2[0][tn], 32)) + (s2[1][tn] + __shfl_xor(s2[1][tn], 32));
      if (lane < 32 && jc < Co) {
        stats_part[(t64 * 2 + 0) * Co + jc] = a1;
        stats_part[(t64 * 2 + 1) * Co + jc] = a2;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same product with W cut ONCE per call (r3d_w_pack_bx3_kernel: the pieces in the order the main kernel's LDS image
// wants them, 24 KB per (128-column tile, K-step) -- L2-resident: W is at most 512 x 512) and 32-row wave tiles:
// 64 accumulator registers instead of 128, ~150 registers in all -> THREE workgroups per CU where the kernel above
// fits two, and per K-step and wave 88 VALU instructions of cutting (X only) for 48 MFMAs instead of 264 for 96.
// A workgroup = 4 waves x 32 rows on the same NT x 32 columns; the W tile of the next K-step is copied global -> LDS
// behind the current step's MFMAs (two halves of 3 x 16 bytes per thread), one barrier per K-step.
// ---------------------------------------------------------------------------------------------------------------
__global__ void r3d_w_pack_bx3_kernel(const float* __restrict__ W, int K, int Co, int BN /* 128 or 64 */,
                                      unsigned short* __restrict__ out) {
  const int NT = BN / 32, ksteps = K / 32;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (column n of the padded matrix, chunk of 8 k)
  const int Cop = (Co + BN - 1) / BN * BN;
  if (i >= (long)Cop * (K / 8)) return;
  const int n = (int)(i / (K / 8)), c8 = (int)(i - (long)n * (K / 8));
  const int ks = c8 >> 2, c = c8 & 3;  // chunk c of K-step ks: memory k = 32 ks + 8 c + 0..7 = 16 g + 8 s + i with c = 2 g + s
  float v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = r3d_keep(W[(long)min(n, Co - 1) * K + 32 * ks + 8 * c + u], n < Co);
  const r3d_bx3 f = r3d_bx3_split8(v);
  const int ct = n / BN, tn = (n % BN) >> 5, lane = (n & 31) + 32 * (c >> 1), sx = c & 1;
  // [ct][ks][s][tn][piece][lane][8]
  unsigned short* d = out + ((((long)(ct * ksteps + ks) * 2 + sx) * NT + tn) * 3) * 512 + lane * 8;
  *reinterpret_cast<r3d_u32x4*>(d) = f.h;
  *reinterpret_cast<r3d_u32x4*>(d + 512) = f.m;
  *reinterpret_cast<r3d_u32x4*>(d + 1024) = f.l;
}

template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void r3d_pointwise_gemm_bx3p_kernel(
    const float* __restrict__ X, long ldx, const unsigned short* __restrict__ Wp, int M, int K, int Co,
    const float* __restrict__ scale, const float* __restrict__ shift, int act, float* __restrict__ Out, long ldo,
    int accumulate, float* __restrict__ stats_part) {
  constexpr int BM = 128, BN = 32 * NT;
  constexpr int TILE = 2 * NT * 3 * 512;  // bf16 of one (column tile, K-step) image
  constexpr int CPT = TILE / 8 / 256;     // 16-byte chunks per thread and K-step: 6 (NT = 4) or 3 (NT = 2)
  __shared__ __attribute__((aligned(16))) unsigned short Bs[2][TILE];
  __shared__ float st_s[2][2][BN];        // [64-row tile of the workgroup][sum, sum of squares][column] of its upper wave
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, g = lane >> 5;
  const int tiles_n = (Co + BN - 1) / BN, ksteps = K / 32;
  const int tile = r3d_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
  const int tile_m = tile / tiles_n, ct = tile - tile_m * tiles_n;
  const long m0 = (long)tile_m * BM, mw = m0 + 32 * w;  // the wave's 32 rows
  const int n0 = ct * BN;
  const long arows = min((long)BM, (long)M - m0);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(X + m0 * ldx), 0, (int)(((arows - 1) * ldx + K) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(Wp + (long)ct * ksteps * TILE), 0, ksteps * TILE * 2, 0x00020000);
  const int oa = (int)(((32 * w + j) * ldx + 16 * g) * 4);
  float araw[16];
  auto aload = [&](int k) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const r3d_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, oa + 16 * q, k * 4, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) araw[4 * q + i] = __uint_as_float(v[i]);
    }
  };
  // W image of K-step ks: chunk (tid + 256 u) of the 24 / 12 KB, copied as it lies
  constexpr int H0 = (CPT + 1) / 2;  // chunks [0, H0) travel in the first half of a step, [H0, CPT) in the second
  r3d_u32x4 wst[CPT];
  auto wload = [&](int ks, int half) {
#pragma unroll
    for (int u = 0; u < CPT; ++u)
      if ((u < H0) == (half == 0)) wst[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, (tid + 256 * u) * 16, ks * TILE * 2, 0);
  };
  auto wstore = [&](int buf, int half) {
#pragma unroll
    for (int u = 0; u < CPT; ++u)
      if ((u < H0) == (half == 0)) *reinterpret_cast<r3d_u32x4*>(&Bs[buf][(tid + 256 * u) * 8]) = wst[u];
  };
  f32x16 acc[NT];
#pragma unroll
  for (int b = 0; b < NT; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  auto mma = [&](int buf, int sx, const r3d_bx3& fa) {
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      r3d_bx3 fb;
      const unsigned short* bsrc = &Bs[buf][((sx * NT + tn) * 3) * 512 + lane * 8];
      fb.h = *reinterpret_cast<const r3d_u32x4*>(bsrc);
      fb.m = *reinterpret_cast<const r3d_u32x4*>(bsrc + 512);
      fb.l = *reinterpret_cast<const r3d_u32x4*>(bsrc + 1024);
      acc[tn] = r3d_bx3_mma(fa, fb, acc[tn]);
    }
  };
  wload(0, 0);
  wload(0, 1);
  aload(0);
  wstore(0, 0);
  wstore(0, 1);
  __syncthreads();
  // branch-free loop body (exact wait counts, see the kernel above): the last step requests its own operands again
  for (int ks = 0; ks < ksteps; ++ks) {
    const int cur = ks & 1, kn = min(ks + 1, ksteps - 1);
    const r3d_bx3 fa0 = r3d_bx3_split8(&araw[0]), fa1 = r3d_bx3_split8(&araw[8]);
    aload(32 * kn);
    wload(kn, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma(cur, 0, fa0);
    wstore(cur ^ 1, 0);
    wload(kn, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(cur, 1, fa1);
    wstore(cur ^ 1, 1);
    __syncthreads();
  }
  // Epilogue: the wave's 32 rows x BN columns through LDS (the W buffers, free after the last barrier), whole rows out.
  // The SIMD's vector ISSUE is what bounds this kernel (SQ counters: 4.2 VALU + 1.8 LDS instructions per MFMA with a
  // one-size-fits-all epilogue -- ~37 issue cycles against the MFMA's 32), and the epilogue is paid per output element
  // whatever K is: so it comes in variants chosen by wave-uniform branches -- PLAIN (no scale / shift / activation:
  // raw z of a training layer, input gradients), STATS (column sums wanted), FULL (no row or column of the tile
  // beyond M / Co: no bounds arithmetic per element).
  constexpr int ER = BN + 4;
  float* es = reinterpret_cast<float*>(&Bs[0][0]) + w * 8 * ER;
  constexpr int LPR = BN / 4, RPI = 64 / LPR;
  const int erow = lane / LPR, ecol = 4 * (lane % LPR);
  float s1[NT], s2[NT];
#pragma unroll
  for (int tn = 0; tn < NT; ++tn) s1[tn] = s2[tn] = 0.f;
  auto epilogue = [&](auto PLAIN_, auto STATS_, auto FULL_) {
    constexpr bool PLAIN = decltype(PLAIN_)::value, STATS = decltype(STATS_)::value, FULL = decltype(FULL_)::value;
    const bool cok = FULL || n0 + ecol < Co;
    float sc[NT], sh[NT];
    if (!PLAIN) {
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const int jc = n0 + 32 * tn + j;
        sc[tn] = (scale && (FULL || jc < Co)) ? scale[jc] : 1.f;
        sh[tn] = (shift && (FULL || jc < Co)) ? shift[jc] : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // registers 4q .. 4q+3: rows 8 q + 4 g + (0..3)
      const int rleft = FULL ? 8 : (int)min((long)8, (long)M - (mw + 8 * q));  // valid rows of this group of 8 (uniform)
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = acc[tn][4 * q + i];
          if (!PLAIN) {
            v = sc[tn] * v + sh[tn];
            if (act == GB_ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == GB_ACT_LRELU02) v = fmaxf(v, 0.2f * v);
          }
          es[(4 * g + i) * ER + 32 * tn + j] = v;
          if (STATS) {
            const float u = FULL ? v : r3d_keep(v, 4 * g + i < rleft);  // (rows beyond M hold zeros anyway; PLAIN keeps them 0)
            s1[tn] += u;
            s2[tn] += u * u;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 8 / RPI; ++i) {
        const int row = erow + RPI * i;
        const float4 v = *reinterpret_cast<const float4*>(&es[row * ER + ecol]);
        if ((FULL || row < rleft) && cok) {
          float4* dst = reinterpret_cast<float4*>(&Out[(mw + 8 * q + row) * ldo + n0 + ecol]);
          if (accumulate) {
            const float4 o = *dst;
            *dst = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
          } else {
            *dst = v;
          }
        }
      }
    }
  };
  {
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    const bool plain = !scale && !shift && act == GB_ACT_NONE, full = mw + 32 <= M && n0 + BN <= Co, stats = stats_part != nullptr;
    if (plain) {
      if (stats) { if (full) epilogue(T_{}, T_{}, T_{}); else epilogue(T_{}, T_{}, F_{}); }
      else       { if (full) epilogue(T_{}, F_{}, T_{}); else epilogue(T_{}, F_{}, F_{}); }
    } else {
      if (stats) epilogue(F_{}, T_{}, F_{});  // (not a combination the model launches: one generic variant)
      else       { if (full) epilogue(F_{}, F_{}, T_{}); else epilogue(F_{}, F_{}, F_{}); }
    }
  }
  if (stats_part) {  // uniform.  A 64-row tile = two waves: the lower one adds the upper one's sums (fixed order)
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      const float a1 = s1[tn] + __shfl_xor(s1[tn], 32), a2 = s2[tn] + __shfl_xor(s2[tn], 32);
      if ((w & 1) && lane < 32) { st_s[w >> 1][0][32 * tn + j] = a1; st_s[w >> 1][1][32 * tn + j] = a2; }
      s1[tn] = a1;
      s2[tn] = a2;
    }
    __syncthreads();
    if (!(w & 1) && lane < 32 && mw < M) {
      const long t64 = mw >> 6;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const int jc = n0 + 32 * tn + j;
        if (jc < Co) {
          stats_part[(t64 * 2 + 0) * Co + jc] = s1[tn] + st_s[w >> 1][0][32 * tn + j];
          stats_part[(t64 * 2 + 1) * Co + jc] = s2[tn] + st_s[w >> 1][1][32 * tn + j];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// part[chunk][i][j] = sum over the chunk's rows m of A[m][i] * B[m][j]   (weight gradients: A = dz (M, Ca), B = X (M, Cb)).
// The contraction runs over the ROW axis of both operands: a staged chunk is 8 consecutive rows of one column (8 loads,
// the lanes along the columns: coalesced), everything behind the staging is the kernel above.  The M axis is split in
// chunks of tn_rows rows whose partial tiles a second kernel adds in ascending order (train_ops.hip).
// ---------------------------------------------------------------------------------------------------------------
template <int WM, int WN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void r3d_gemm_tn_bx3_kernel(
    const float* __restrict__ A, long lda, const float* __restrict__ B, long ldb, long M, int Ca, int Cb, int tn_rows,
    float* __restrict__ part /* [chunks][Ca][Cb] */) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int XU = BM / 64, WU = BN / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned short gb_smem[];
  unsigned short* As = gb_smem;
  unsigned short* Bs = gb_smem + 3 * BM * GB_RS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w / WN, wn = w - wm * WN;
  const int nti = (Ca + BM - 1) / BM, ntj = (Cb + BN - 1) / BN;
  const int tile = r3d_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
  const int chunk = tile / (nti * ntj), tij = tile - chunk * (nti * ntj);
  const int i0 = (tij % nti) * BM, j0 = (tij / nti) * BN;
  const long m_beg = (long)chunk * tn_rows, m_end = min(M, m_beg + tn_rows);
  // staging map: chunk c = tid + 256 u -> column c % BM (consecutive threads: consecutive columns), row group c / BM (8 rows).
  // Buffer loads over the chunk's rows: rows beyond m_end read as zeros (they enter every sum), and the loop body stays
  // free of branches, so the compiler counts the loads in flight exactly (see the pointwise kernel).
  const int crows = (int)(m_end - m_beg);
  const __amdgpu_buffer_rsrc_t ra =
      __builtin_amdgcn_make_buffer_rsrc((void*)(A + m_beg * lda), 0, (int)(((long)(crows - 1) * lda + Ca) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb =
      __builtin_amdgcn_make_buffer_rsrc((void*)(B + m_beg * ldb), 0, (int)(((long)(crows - 1) * ldb + Cb) * 4), 0x00020000);
  const int lda4 = (int)lda * 4, ldb4 = (int)ldb * 4;
  int ao[XU], bo[WU], amc[XU], bmc[WU];
#pragma unroll
  for (int u = 0; u < XU; ++u) {
    const int c = tid + 256 * u;
    amc[u] = c / BM;
    ao[u] = 8 * amc[u] * lda4 + 4 * min(i0 + c % BM, Ca - 1);
  }
#pragma unroll
  for (int u = 0; u < WU; ++u) {
    const int c = tid + 256 * u;
    bmc[u] = c / BN;
    bo[u] = 8 * bmc[u] * ldb4 + 4 * min(j0 + c % BN, Cb - 1);
  }
  float ar[XU][8], br[WU][8];
  auto gload = [&](int mk) {  // mk: row of the chunk (all of the offset in the vector part: that is what is range-checked)
#pragma unroll
    for (int u = 0; u < XU; ++u)
#pragma unroll
      for (int t = 0; t < 8; ++t) ar[u][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, ao[u] + (mk + t) * lda4, 0, 0));
#pragma unroll
    for (int u = 0; u < WU; ++u)
#pragma unroll
      for (int t = 0; t < 8; ++t) br[u][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, bo[u] + (mk + t) * ldb4, 0, 0));
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  gload(0);
  for (int mk = 0; mk < crows; mk += 32) {
#pragma unroll
    for (int u = 0; u < XU; ++u) { const int c = tid + 256 * u; gb_store_chunk(As, BM, c % BM, amc[u], ar[u]); }
#pragma unroll
    for (int u = 0; u < WU; ++u) { const int c = tid + 256 * u; gb_store_chunk(Bs, BN, c % BN, bmc[u], br[u]); }
    __syncthreads();
    gload(mk + 32);  // (behind the last step: all rows out of range, zeros nobody uses)
    __builtin_amdgcn_sched_barrier(0);
    gb_wave_mma(As, BM, 64 * wm, Bs, BN, 64 * wn, lane, acc);
    __syncthreads();
  }
  float* out = part + (long)chunk * Ca * Cb;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int j = j0 + 64 * wn + 32 * tn + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + 64 * wm + 32 * tm + r3d_acc_row(r, lane);
        if (i < Ca && j < Cb) out[(long)i * Cb + j] = acc[tm][tn][r];
      }
    }
}

// ===========================================================================
// launchers (called from gemm.hip / train_ops.hip when the bf16 x 3 arithmetic is selected and the shape allows it)
// ===========================================================================
// three piece planes of BM + BN rows: 61440 B for 128 x 128, 76800 B for 256 x 64 (two workgroups per CU either way)
static const size_t GB_LDS_22 = sizeof(unsigned short) * 3 * (128 + 128) * GB_RS;
static const size_t GB_LDS_41 = sizeof(unsigned short) * 3 * (256 + 64) * GB_RS;

template <typename KernelT>
static int gb_lds_attr(KernelT k, size_t bytes) {
  return hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : 1;
}

bool r3d_pointwise_bx3_ok(const float* X, long ldx, const float* W, long M, int K, int Co) {
  return g_r3d_matrix_arith == 1 && (g_r3d_gemm_bx3 & 1) && K >= 32 && K % 32 == 0 && (ldx & 3) == 0 &&
         ((uintptr_t)X & 15) == 0 && ((uintptr_t)W & 15) == 0 && Co >= 32 && Co % 4 == 0 && M >= 64;
}

// Scratch for the packed W of r3d_pointwise_gemm_bx3p_kernel: one buffer per stream (launches of one stream are
// ordered), grown on demand.  NEVER for a stream that is capturing: a captured launch would bake this pointer into a
// graph that may replay on any stream beside other graphs captured on the same one (episode_graph.py captures its slots
// one after the other and replays them concurrently: two slots sharing the buffer computed with each other's weights
// pieces, caught by tests/test_gpu_parity_full.py) -- captured launch sequences take the kernel that cuts W itself,
// which gives the same bits.
// r3d_set_wpack_in_capture(1): the owner of a capture promises that the graph it captures is the ONLY user of its stream's
// scratch while it replays (one graph per stream, replayed on that stream or never beside another replay of itself:
// batched.BatchGraph) -- then a captured launch may use the stream's scratch too, provided it exists already (the
// capture's eager warm-up passes on the same stream allocate it; nothing is allocated while capturing).
static int g_wpack_in_capture = 0;
extern "C" int r3d_set_wpack_in_capture(int on) {
  const int old = g_wpack_in_capture;
  g_wpack_in_capture = on ? 1 : 0;
  return old;
}
struct WPackBuf { int dev; hipStream_t st; void* p; size_t cap; };
static unsigned short* wpack_scratch(hipStream_t st, size_t bytes) {
  static std::mutex mu;
  static std::vector<WPackBuf> pool;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) return nullptr;
  const bool capturing = cs != hipStreamCaptureStatusNone;
  if (capturing && !g_wpack_in_capture) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  for (auto& b : pool)
    if (b.dev == dev && b.st == st) {  // (the null stream is one handle on every device: the device is part of the key)
      if (b.cap >= bytes) return (unsigned short*)b.p;
      if (capturing) return nullptr;
      void* p = nullptr;
      if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
      b.p = p; b.cap = bytes;  // (the old buffer stays allocated: launches already queued on the stream still read it)
      return (unsigned short*)p;
    }
  if (capturing) return nullptr;
  void* p = nullptr;
  const size_t cap = bytes < (4u << 20) ? (4u << 20) : bytes;  // (512 x 512 pieces = 1.5 MB: one size fits the model)
  if (hipMalloc(&p, cap) != hipSuccess) return nullptr;
  pool.push_back(WPackBuf{dev, st, p, cap});
  return (unsigned short*)p;
}

int r3d_pointwise_bx3_launch(const float* X, long ldx, const float* W, long M, int K, int Co, const float* scale,
                             const float* shift, int act, float* Out, long ldo, int accumulate, float* stats_part,
                             hipStream_t st) {
  if ((ldo & 3) != 0 || ((uintptr_t)Out & 15) != 0) return -1;  // (the caller takes the fp32 kernel)
  if (!(GB_ABL & 128) && (g_r3d_gemm_bx3 & 4) && M >= 256) {  // W cut once per call, 32-row wave tiles, 3 workgroups per CU
    const int bnp = Co > 64 ? 128 : 64;
    const int Cop = r3d_cdiv(Co, bnp) * bnp;
    unsigned short* wp = wpack_scratch(st, (size_t)Cop * K * 6);
    if (wp) {
      hipLaunchKernelGGL(r3d_w_pack_bx3_kernel, dim3(r3d_cdiv((long)Cop * (K / 8), 256)), dim3(256), 0, st, W, K, Co, bnp, wp);
      const long tiles = (long)r3d_cdiv(M, 128) * (Cop / bnp);
      R3D_REQUIRE(tiles < 0x7fffffffL, "r3d_pointwise_conv: too many tiles");
      if (bnp == 128)
        hipLaunchKernelGGL((r3d_pointwise_gemm_bx3p_kernel<4>), dim3((unsigned)tiles), dim3(256), 0, st, X, ldx, wp, (int)M, K, Co,
                           scale, shift, act, Out, ldo, accumulate, stats_part);
      else
        hipLaunchKernelGGL((r3d_pointwise_gemm_bx3p_kernel<2>), dim3((unsigned)tiles), dim3(256), 0, st, X, ldx, wp, (int)M, K, Co,
                           scale, shift, act, Out, ldo, accumulate, stats_part);
      return R3D_OK;
    }
  }
  const int bn = (Co > 64 && !(GB_ABL & 64)) ? 128 : 64;  // (probe bit 64: 64-column tiles for every layer)
  const long tiles = (long)r3d_cdiv(M, 256) * r3d_cdiv(Co, bn);
  R3D_REQUIRE(tiles < 0x7fffffffL, "r3d_pointwise_conv: too many tiles");
  if (bn == 128)
    hipLaunchKernelGGL((r3d_pointwise_gemm_bx3_kernel<4>), dim3((unsigned)tiles), dim3(256), 0, st, X, ldx, W, (int)M, K, Co, scale,
                       shift, act, Out, ldo, accumulate, stats_part);
  else
    hipLaunchKernelGGL((r3d_pointwise_gemm_bx3_kernel<2>), dim3((unsigned)tiles), dim3(256), 0, st, X, ldx, W, (int)M, K, Co, scale,
                       shift, act, Out, ldo, accumulate, stats_part);
  return R3D_OK;
}

bool r3d_gemm_tn_bx3_ok(int Ca, int Cb) { return g_r3d_matrix_arith == 1 && (g_r3d_gemm_bx3 & 2) && Ca >= 32 && Cb >= 32; }

// tiles of the (Ca, Cb) output for the chunk count of train_ops.hip's tn_rows()
int r3d_gemm_tn_bx3_tiles(int Ca, int Cb) {
  return Cb > 64 ? r3d_cdiv(Ca, 128) * r3d_cdiv(Cb, 128) : r3d_cdiv(Ca, 256) * r3d_cdiv(Cb, 64);
}

template <int WM, int WN>
static int gemm_tn_bx3_go(size_t lds, int blocks, hipStream_t st, const float* A, long lda, const float* B, long ldb, long M,
                          int Ca, int Cb, int rows, float* part) {
  static bool attr = false;
  if (!attr) {
    R3D_REQUIRE(gb_lds_attr(r3d_gemm_tn_bx3_kernel<WM, WN>, lds) == 0, "r3d_gemm_tn: cannot reserve %zu B of LDS", lds);
    attr = true;
  }
  hipLaunchKernelGGL((r3d_gemm_tn_bx3_kernel<WM, WN>), dim3(blocks), dim3(256), lds, st, A, lda, B, ldb, M, Ca, Cb, rows, part);
  return R3D_OK;
}

int r3d_gemm_tn_bx3_launch(const float* A, long lda, const float* B, long ldb, long M, int Ca, int Cb, int rows, int chunks,
                           float* part, hipStream_t st) {
  const int blocks = r3d_gemm_tn_bx3_tiles(Ca, Cb) * chunks;
  if (Cb > 64) return gemm_tn_bx3_go<2, 2>(GB_LDS_22, blocks, st, A, lda, B, ldb, M, Ca, Cb, rows, part);
  return gemm_tn_bx3_go<4, 1>(GB_LDS_41, blocks, st, A, lda, B, ldb, M, Ca, Cb, rows, part);
}
