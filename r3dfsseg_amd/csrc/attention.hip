// Point self-attention (single head, d = 64) for gfx950, flash style: the N x N
// attention matrix never leaves registers.
//
// Replaces (reference): models/attention.py:43-46
//   attn = softmax((q^T / sqrt(64)) k, dim=-1) ; y = attn v^T
// q/k/v come from one fused 256->192 point-wise GEMM (gemm.hip), q already scaled by
// 1/8 (exact), stored point-major in one (B*N, ld) buffer at column offsets 0/64/128.
//
// fp32 matrix core throughout (north_star asks for fp32 features within 1e-4).
// Per wave: 32 query rows.  For each tile of 32 keys
//   S^T (keys x queries) = K_tile Q^T          32 x v_mfma_f32_32x32x2_f32
//   online softmax over the key axis: the 16 accumulator registers of a lane are 16
//     keys of ONE query (col = lane & 31), so row max / row sum are in-lane plus one
//     exchange with lane ^ 32
//   O^T (channels x queries) += V_tile^T P^T   2 x 16 MFMAs; P^T is consumed straight
//     from the S^T accumulator registers (accumulator row r of lane half h is key
//     r3d_acc_row(r): both halves feed one MFMA k-pair, no data movement).
// 4 waves share the K/V tiles through double-buffered LDS.
#include <stdlib.h>
#include "common.h"
#include <type_traits>

#define AT_LD 65
#define AT_PROW 68  // floats per row of a forward partial: 64 o + m + l, padded to 16-byte rows

// Dropout keep decision of attention weight (row = b*N + query, key): a stateless integer hash, so the
// backward kernels regenerate the same mask (reference: nn.Dropout(0.1) on the attention matrix,
// attention.py:45; the reference's torch RNG stream itself is not reproducible elsewhere).
static __device__ __forceinline__ bool attn_keep(unsigned seed, unsigned row, unsigned key, unsigned thresh) {
  unsigned x = row * 0x9E3779B1u ^ key * 0x85EBCA77u ^ seed;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x >= thresh;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void r3d_attention_fwd_kernel(
    const float* __restrict__ qkv, long ld, int N, float* __restrict__ out, long ldo,
    float* __restrict__ lse_out, float p_drop, unsigned seed, const unsigned* __restrict__ seed_dev, int seed_group,
    int tiles_per_split, float* __restrict__ part /* split > 1: [split][M][AT_PROW] = unnormalised o | m | l */) {
  // (plain block order.  An XCD-aware order -- common.h: r3d_xcd_swizzle, the workgroups sharing an L2 on the same clouds --
  // was measured and lost 3-8 % on these kernels: an XCD then holds 4 clouds' K / V pieces at a time, 6 MB against its 4 MB
  // L2, while in plain order the operands come out of the Infinity Cache: profiles/r03_experiments.md)
  const int bid_x = blockIdx.x, bid_y = blockIdx.y, bid_z = blockIdx.z;
  if (seed_dev) seed += *seed_dev;  // per-replay seed of a captured hipGraph lives in device memory
  const unsigned thresh = p_drop > 0.f ? (unsigned)(p_drop * 4294967296.0) : 0u;
  const float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  __shared__ float Ks[2][32 * AT_LD];
  __shared__ float Vs[2][32 * AT_LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = bid_y;
  const long base = (long)b * N;
  unsigned hbase = (unsigned)base;  // row id the dropout hash sees
  if (seed_group > 0) {             // batch of episodes: every group of seed_group clouds is one episode with its own seed
    const int ep_ = b / seed_group;
    seed += 2u * (unsigned)ep_;
    hbase = (unsigned)((b - ep_ * seed_group) * N);
  }
  const int q_row = bid_x * 128 + 32 * w + (lane & 31);
  const bool q_ok = q_row < N;
  // Q^T fragments: B[k = ch][j = query]
  float bq[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    const float v = qkv[(base + min(q_row, N - 1)) * ld + 2 * s + (lane >> 5)];
    bq[s] = r3d_keep(v, q_ok);
  }
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  // staging map: 32 rows x 16 float4 per operand = 512 float4, 2 per thread
  const int st_row = tid >> 3, st_c4 = (tid & 7) * 2;  // rows 0..31, float4 cols {0..15}
  float4 kreg[2], vreg[2];
  auto load_tile = [&](int key0) {
    const int kr = key0 + st_row;
    const bool ok = kr < N;
    const int krc = min(kr, N - 1);  // unconditional loads, masked afterwards
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* p = qkv + (base + krc) * ld + 4 * (st_c4 + i);
      const float4 kk = *reinterpret_cast<const float4*>(p + 64);
      const float4 vv = *reinterpret_cast<const float4*>(p + 128);
      kreg[i] = make_float4(r3d_keep(kk.x, ok), r3d_keep(kk.y, ok), r3d_keep(kk.z, ok), r3d_keep(kk.w, ok));
      vreg[i] = make_float4(r3d_keep(vv.x, ok), r3d_keep(vv.y, ok), r3d_keep(vv.z, ok), r3d_keep(vv.w, ok));
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float* kd = &Ks[buf][st_row * AT_LD + 4 * (st_c4 + i)];
      float* vd = &Vs[buf][st_row * AT_LD + 4 * (st_c4 + i)];
      kd[0] = kreg[i].x; kd[1] = kreg[i].y; kd[2] = kreg[i].z; kd[3] = kreg[i].w;
      vd[0] = vreg[i].x; vd[1] = vreg[i].y; vd[2] = vreg[i].z; vd[3] = vreg[i].w;
    }
  };

  // key tiles [t_beg, t_end) of this workgroup (bid_z = split of the key axis: more workgroups for small grids)
  const int t_beg = bid_z * tiles_per_split;
  const int ntiles = min((N + 31) / 32, t_beg + tiles_per_split);
  load_tile(32 * t_beg);
  store_tile(t_beg & 1);
  __syncthreads();
  for (int t = t_beg; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) load_tile(32 * (t + 1));
    // S^T = K Q^T
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    {
      const float* ap = &Ks[buf][(lane & 31) * AT_LD + (lane >> 5)];
#pragma unroll
      for (int st = 0; st < 32; ++st)
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * st], bq[st], s, 0, 0, 0);
    }
    // mask keys beyond N (last tile only)
    if (32 * (t + 1) > N) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (32 * t + r3d_acc_row(r, lane) >= N) s[r] = -INFINITY;
    }
    float mt = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __expf(m_run - m_new);  // exp(-inf) = 0 on the first tile
    float lt = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __expf(s[r] - m_new);
      lt += s[r];
    }
    lt += __shfl_xor(lt, 32);
    l_run = l_run * alpha + lt;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    if (thresh) {  // dropout acts on the normalised weights: numerator only, the row sum l stays undropped
#pragma unroll
      for (int r = 0; r < 16; ++r)
        s[r] = attn_keep(seed, hbase + (unsigned)(q_row), (unsigned)(32 * t + r3d_acc_row(r, lane)), thresh) ? s[r] * keep_scale : 0.f;
    }
    // O^T += V^T P^T
    {
      const float* vp = &Vs[buf][(lane & 31)];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = r3d_acc_row(r, lane);
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[key * AT_LD], s[r], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[key * AT_LD + 32], s[r], o1, 0, 0, 0);
      }
    }
    if (t + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();
  }
  if (!q_ok) return;
  if (part) {  // partial (o, m, l) of this key range; r3d_attention_combine_kernel merges the splits
    float* prow = part + ((long)bid_z * gridDim.y * N + base + q_row) * AT_PROW;
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // registers 4g .. 4g+3 are 4 consecutive channels: 16-byte stores
      const int ch = 8 * g + 4 * (lane >> 5);
      *reinterpret_cast<float4*>(prow + ch) = make_float4(o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]);
      *reinterpret_cast<float4*>(prow + 32 + ch) = make_float4(o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]);
    }
    if (lane < 32) { prow[64] = m_run; prow[65] = l_run; }
    return;
  }
  const float inv = 1.f / l_run;
  float* orow = out + (base + q_row) * ldo;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ch = r3d_acc_row(r, lane);
    orow[ch] = o0[r] * inv;
    orow[32 + ch] = o1[r] * inv;
  }
  if (lse_out && lane < 32) lse_out[base + q_row] = m_run + __logf(l_run);
}

// ---------------------------------------------------------------------------------------------------------------
// The same attention on the bf16 matrix core, fp32 values as three bf16 pieces (common.h "bf16 x 3"): 48 bf16 MFMAs per
// 32-key tile and wave of the forward (1536 cycles) in place of 64 fp32 MFMAs (4096 cycles).
// The operands arrive already cut: r3d_bx3_pack_kernel turns 64-column blocks of a point-major fp32 matrix into rows of
// [piece 0..2][64] bf16 (384 B per point), so the cut is paid once per element, not once per workgroup that streams it,
// and staging is a copy (by the LDS-DMA, below).
#define AB_ROW 192  // bf16 per point of a packed operand: 3 pieces x 64 channels

// dst[blk][row][piece][c] = piece of src[row][64 blk + c], c < 64, blk < nblk.  One thread: 8 channels of a row.
__global__ void r3d_bx3_pack_kernel(const float* __restrict__ src, long ld, int nblk, long M,
                                    unsigned short* __restrict__ dst) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * 8 * nblk) return;
  const int c8 = (int)(i % (8 * nblk));  // consecutive threads walk a row: coalesced reads
  const long row = i / (8 * nblk);
  const int blk = c8 >> 3, c0 = (c8 & 7) * 8;
  const float* p = src + row * ld + 64 * blk + c0;
  const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
  const float x[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
  const r3d_bx3 f = r3d_bx3_split8(x);
  unsigned short* d = dst + ((long)blk * M + row) * AB_ROW + c0;
  *reinterpret_cast<r3d_u32x4*>(d) = f.h;
  *reinterpret_cast<r3d_u32x4*>(d + 64) = f.m;
  *reinterpret_cast<r3d_u32x4*>(d + 128) = f.l;
}
static void bx3_pack(const float* src, long ld, int nblk, long M, unsigned short* dst, hipStream_t st) {
  hipLaunchKernelGGL(r3d_bx3_pack_kernel, dim3(r3d_cdiv(M * 8 * nblk, 256)), dim3(256), 0, st, src, ld, nblk, M, dst);
}
static __device__ __forceinline__ r3d_u32x4 ab_mask(r3d_u32x4 v, bool ok) {
  const unsigned m = ok ? 0xffffffffu : 0u;
  v[0] &= m; v[1] &= m; v[2] &= m; v[3] &= m;
  return v;
}
// B[k = channel 16 st + 8 half + e][j = row] fragments of one packed row (the lane's own query / key), 4 k-steps
static __device__ __forceinline__ void ab_load_row_frags(const unsigned short* __restrict__ row, int half, bool ok,
                                                         r3d_bx3 (&f)[4]) {
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const unsigned short* p = row + 16 * st + 8 * half;
    f[st].h = ab_mask(*reinterpret_cast<const r3d_u32x4*>(p), ok);
    f[st].m = ab_mask(*reinterpret_cast<const r3d_u32x4*>(p + 64), ok);
    f[st].l = ab_mask(*reinterpret_cast<const r3d_u32x4*>(p + 128), ok);
  }
}
// the three pieces of a fragment: 16 bytes each, `plane` bf16 apart
static __device__ __forceinline__ r3d_bx3 ab_frag(const unsigned short* p, int plane) {
  r3d_bx3 a;
  a.h = *reinterpret_cast<const r3d_u32x4*>(p);
  a.m = *reinterpret_cast<const r3d_u32x4*>(p + plane);
  a.l = *reinterpret_cast<const r3d_u32x4*>(p + 2 * plane);
  return a;
}
static __device__ __forceinline__ r3d_bx3 ab_split_acc(const f32x16& s, int sI) {
  const float px[8] = {s[8 * sI], s[8 * sI + 1], s[8 * sI + 2], s[8 * sI + 3],
                       s[8 * sI + 4], s[8 * sI + 5], s[8 * sI + 6], s[8 * sI + 7]};
  return r3d_bx3_split8(px);
}

// ---- LDS images filled by the LDS-DMA (global_load_lds, 16 B per lane, no VGPR and no ds_write on the way)
// A 32-row tile of one piece is 4 KB of unpadded 128-B rows; one DMA instruction fills 8 rows (lane L -> row L / 8,
// 16-B position L % 8) and the lane's SOURCE address chooses which chunk lands there, so the image is swizzled for free:
//   chunk c (8 channels) of row r sits at position c ^ g(r),  g(r) = 4 ((r >> 1) & 1) | ((r >> 2) & 3).
// One image serves both kinds of read, conflict-free under the LDS bank rules (64 banks; ds_read_b128 in four 16-lane
// groups, ds_read_b64_tr_b16 per 32-lane half):
//   row read  (A[i = image row][k = channel]):  ds_read_b128, lane = row; inside a 16-lane group the 8 even and the 8 odd
//             rows have 8 distinct g;
//   col read  (A[i = channel][k = image row]):  ds_read_b64_tr_b16 takes 4 rows x 16 channels per 16 lanes and hands
//             them over transposed; rows q and q + 2 of an aligned group of 4 differ in bit 2 of g, so the two rows of
//             equal bank parity use different halves of the 128-B row.
#define AG_PIECE 2048  // bf16 per piece image (32 rows x 64)
#define AG_TILE (3 * AG_PIECE)
typedef __attribute__((address_space(3))) void* ag_lds_ptr;
typedef short ag_s16x4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ int ag_swz(int r) { return 4 * ((r >> 1) & 1) | ((r >> 2) & 3); }
// wave w of 4 fills row group w (rows 8w .. 8w+7) of the three piece images of tile rows [row0, row0 + 32) of X.
// ag_dma_off: the lane's byte offset inside a full tile (fixed for the kernel); the tile base is wave-uniform, so the
// DMA takes the scalar-base + 32-bit-offset address form and a full tile costs no vector arithmetic at all.
static __device__ __forceinline__ unsigned ag_dma_off(int w, int lane) {
  const int r = 8 * w + (lane >> 3), pos = lane & 7;
  return (unsigned)(r * AB_ROW + 8 * (pos ^ ag_swz(r))) * 2u;
}
#ifndef ATT_ABL
#define ATT_ABL 0  // probe builds (tools/probe/att_ablate.sh): 1 no LDS-DMA, 2 plain instead of transposed LDS reads, 4 no MFMA
#endif
static __device__ __forceinline__ void ag_dma_tile(const unsigned short* __restrict__ X, long base, int row0, int N,
                                                   unsigned short* img, int w, int lane, unsigned off) {
  if (ATT_ABL & 1) return;
  const char* tile = reinterpret_cast<const char*>(X + (base + row0) * AB_ROW);
  if (row0 + 32 <= N) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
      __builtin_amdgcn_global_load_lds(tile + off + 128 * p, (ag_lds_ptr)(img + p * AG_PIECE + 512 * w), 16, 0, 0);
  } else {  // rows past N: any finite row of the cloud (the kernels mask what such rows produce)
    const int r = 8 * w + (lane >> 3);
    const char* src = tile + off - (long)(r - min(r, N - 1 - row0)) * (AB_ROW * 2);
#pragma unroll
    for (int p = 0; p < 3; ++p)
      __builtin_amdgcn_global_load_lds(src + 128 * p, (ag_lds_ptr)(img + p * AG_PIECE + 512 * w), 16, 0, 0);
  }
}
// element offsets of this lane's fragments inside an image (fixed for the kernel)
struct ag_offs {
  int row[4];     // row read, k-step st: row lane & 31, channels 16 st + 8 half ..
  int col[2][2];  // col read, channel block cc, second index j: image rows 4 half + 8 j + q (+ 16 sI)
};
static __device__ __forceinline__ ag_offs ag_make_offs(int lane) {
  ag_offs o;
  const int k = lane & 31, half = lane >> 5;
#pragma unroll
  for (int st = 0; st < 4; ++st) o.row[st] = k * 64 + 8 * ((2 * st + half) ^ ag_swz(k));
  const int g16 = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
#pragma unroll
  for (int cc = 0; cc < 2; ++cc)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 4 * half + 8 * j + q;  // + 16 sI: ag_swz is periodic in 16 rows
      o.col[cc][j] = r * 64 + 8 * ((4 * cc + 2 * g16 + (p >> 1)) ^ ag_swz(r)) + 4 * (p & 1);
    }
  return o;
}
static __device__ __forceinline__ r3d_bx3 ag_row_frag(const unsigned short* img, const ag_offs& o, int st) {
  return ab_frag(img + o.row[st], AG_PIECE);
}
// A[i = channel 32 cc + (lane & 31)][k = image rows 16 sI + 4 half + {0..3, 8..11}]: the rows an accumulator lane half
// holds in registers 8 sI .. 8 sI + 7
static __device__ __forceinline__ r3d_bx3 ag_col_frag(const unsigned short* img, const ag_offs& o, int sI, int cc) {
  const unsigned short* a0 = img + o.col[cc][0] + 16 * 64 * sI;
  const unsigned short* a1 = img + o.col[cc][1] + 16 * 64 * sI;
  r3d_bx3 f;
  r3d_u32x4* pieces[3] = {&f.h, &f.m, &f.l};
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) {
#if ATT_ABL & 2
    const ag_s16x4 lo = *reinterpret_cast<const ag_s16x4*>(a0 + pc * AG_PIECE);
    const ag_s16x4 hi = *reinterpret_cast<const ag_s16x4*>(a1 + pc * AG_PIECE);
#else
    const ag_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ag_s16x4*)(a0 + pc * AG_PIECE));
    const ag_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ag_s16x4*)(a1 + pc * AG_PIECE));
#endif
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    (*pieces[pc])[0] = l2.x; (*pieces[pc])[1] = l2.y; (*pieces[pc])[2] = h2.x; (*pieces[pc])[3] = h2.y;
  }
  return f;
}

// Software pipeline of one wave: the S^T MFMAs of tile t + 1 are issued beside the softmax of tile t (independent
// accumulators), the P^T V MFMAs of tile t beside the second half of the P cut, so the VALU work sits in the issue slots
// the matrix core leaves free (an MFMA holds the issue port for 8 of its 32 cycles).  K is staged two tiles ahead, V one,
// by the LDS-DMA; one barrier per tile.  DROP: dropout compiled in (training with p > 0).
#ifdef ATT_STAMPS  // phase clocks of one wave (tools/probe/att_stamps.py); never in the shipped library
__device__ unsigned long long g_att_dbg[16];
extern "C" int r3d_attention_debug_read(unsigned long long* out16) {
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_att_dbg), sizeof(g_att_dbg)) == hipSuccess ? 0 : 1;
}
#define ASTAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); dbg_acc[i] += now_ - dbg_last; dbg_last = now_; } while (0)
#else
#define ASTAMP(i)
#endif
template <bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void r3d_attention_fwd_bx3_kernel(
    const unsigned short* __restrict__ Qp, const unsigned short* __restrict__ Kp, const unsigned short* __restrict__ Vp,
    int N, float* __restrict__ out, long ldo, float* __restrict__ lse_out, float p_drop, unsigned seed,
    const unsigned* __restrict__ seed_dev, int seed_group, int tiles_per_split, float* __restrict__ part) {
  // (plain block order.  An XCD-aware order -- common.h: r3d_xcd_swizzle, the workgroups sharing an L2 on the same clouds --
  // was measured and lost 3-8 % on these kernels: an XCD then holds 4 clouds' K / V pieces at a time, 6 MB against its 4 MB
  // L2, while in plain order the operands come out of the Infinity Cache: profiles/r03_experiments.md)
  const int bid_x = blockIdx.x, bid_y = blockIdx.y, bid_z = blockIdx.z;
  if (seed_dev) seed += *seed_dev;
  const unsigned thresh = DROP ? (unsigned)(p_drop * 4294967296.0) : 0u;
  const float keep_scale = DROP ? 1.f / (1.f - p_drop) : 1.f;
  __shared__ __attribute__((aligned(16))) unsigned short Ks[2][AG_TILE];
  __shared__ __attribute__((aligned(16))) unsigned short Vs[2][AG_TILE];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5;
  const int b = bid_y;
  const long base = (long)b * N;
  unsigned hbase = (unsigned)base;  // row id the dropout hash sees
  if (seed_group > 0) {             // batch of episodes: every group of seed_group clouds is one episode with its own seed
    const int ep_ = b / seed_group;
    seed += 2u * (unsigned)ep_;
    hbase = (unsigned)((b - ep_ * seed_group) * N);
  }
  const int q_row = bid_x * 128 + 32 * w + (lane & 31);
  const bool q_ok = q_row < N;
  r3d_bx3 bq[4];
  ab_load_row_frags(Qp + (base + min(q_row, N - 1)) * AB_ROW, half, q_ok, bq);
  f32x16 o0, o1, s_a, s_b;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; s_a[r] = 0.f; }
  const float LOG2E = 1.4426950408889634f;
  float m_run = -INFINITY, l_run = 0.f;  // m_run in the log2 domain: max of s * log2(e)
  const int t_beg = bid_z * tiles_per_split;
  const int ntiles = min((N + 31) / 32, t_beg + tiles_per_split);
  const ag_offs offs = ag_make_offs(lane);
  auto s_tile = [&](int t, f32x16& s) {  // S^T = K Q^T of tile t
#pragma unroll
    for (int st = 0; st < 4; ++st) s = r3d_bx3_mma(ag_row_frag(Ks[t & 1], offs, st), bq[st], s);
  };
  const unsigned koff = ag_dma_off(w, lane), voff = koff;
  // prologue: K(t_beg), K(t_beg + 1), V(t_beg) into LDS; S of the first tile
  ag_dma_tile(Kp, base, 32 * t_beg, N, Ks[t_beg & 1], w, lane, koff);
  ag_dma_tile(Vp, base, 32 * t_beg, N, Vs[t_beg & 1], w, lane, voff);
  if (t_beg + 1 < ntiles) ag_dma_tile(Kp, base, 32 * (t_beg + 1), N, Ks[(t_beg + 1) & 1], w, lane, koff);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  s_tile(t_beg, s_a);
  __syncthreads();  // the first step's DMA reuses the buffer of K(t_beg)
#ifdef ATT_STAMPS
  unsigned long long dbg_acc[6] = {0, 0, 0, 0, 0, 0}, dbg_last = __builtin_amdgcn_s_memtime();
  const unsigned long long dbg_t0 = dbg_last, dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  // s_cur: S of tile t (computed one step ago); s_next receives S of tile t + 1
  auto step = [&](int t, f32x16& s_cur, f32x16& s_next, auto HAS1, auto HAS2, auto MASK) {
    constexpr bool has1 = decltype(HAS1)::value, has2 = decltype(HAS2)::value, mask = decltype(MASK)::value;
    ASTAMP(5);
    // K(t) was consumed one step ago (S of tile t ran beside the softmax of t - 1), V(t - 1) as well
    if (has2) ag_dma_tile(Kp, base, 32 * (t + 2), N, Ks[t & 1], w, lane, koff);
    if (has1) ag_dma_tile(Vp, base, 32 * (t + 1), N, Vs[(t + 1) & 1], w, lane, voff);
    ASTAMP(0);
    // ---- block A: S of the next tile beside the softmax of this one
#pragma unroll
    for (int r = 0; r < 16; ++r) s_next[r] = 0.f;
    if (has1) s_tile(t + 1, s_next);
    f32x16 s = s_cur;
    if (mask) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (32 * t + r3d_acc_row(r, lane) >= N) s[r] = -INFINITY;
    }
    float mt = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mt = fmaxf(fmaxf(mt, s[r]), s[r + 1]);
    mt = fmaxf(mt, s[15]);
    mt = fmaxf(mt, __shfl_xor(mt, 32)) * LOG2E;
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // raw v_exp_f32: arguments <= 0, underflow to 0 is right
    float lt = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], LOG2E, -m_new));
      lt += s[r];
    }
    lt += __shfl_xor(lt, 32);
    l_run = l_run * alpha + lt;
    if (DROP) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        s[r] = attn_keep(seed, hbase + (unsigned)(q_row), (unsigned)(32 * t + r3d_acc_row(r, lane)), thresh) ? s[r] * keep_scale : 0.f;
    }
    const r3d_bx3 pf0 = ab_split_acc(s, 0);
    ASTAMP(1);
    if (__any(m_new != m_run)) {  // wave-uniform: the running maximum rarely moves after the first tiles
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    }
    m_run = m_new;
    // ---- block B: O^T += V^T P^T (accumulator registers 8 sI .. 8 sI + 7 are the B fragment of k-step sI) beside the
    // second half of the cut
    const unsigned short* vimg = Vs[t & 1];
    o0 = r3d_bx3_mma(ag_col_frag(vimg, offs, 0, 0), pf0, o0);
    o1 = r3d_bx3_mma(ag_col_frag(vimg, offs, 0, 1), pf0, o1);
    const r3d_bx3 pf1 = ab_split_acc(s, 1);
    o0 = r3d_bx3_mma(ag_col_frag(vimg, offs, 1, 0), pf1, o0);
    o1 = r3d_bx3_mma(ag_col_frag(vimg, offs, 1, 1), pf1, o1);
    ASTAMP(2);
    __builtin_amdgcn_s_waitcnt(0);  // this wave's DMA has landed
    ASTAMP(3);
    __syncthreads();
    ASTAMP(4);
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
  int t = t_beg;
  for (; t + 3 < ntiles; t += 2) {  // two steps per trip: the S accumulators swap roles instead of being copied
    step(t, s_a, s_b, T_{}, T_{}, F_{});
    step(t + 1, s_b, s_a, T_{}, T_{}, F_{});
  }
  if (t + 2 < ntiles) { step(t, s_a, s_b, T_{}, T_{}, F_{}); ++t; s_a = s_b; }
  if (t + 1 < ntiles) { step(t, s_a, s_b, T_{}, F_{}, F_{}); ++t; s_a = s_b; }
  if (32 * (t + 1) > N) step(t, s_a, s_b, F_{}, F_{}, T_{});
  else step(t, s_a, s_b, F_{}, F_{}, F_{});

#ifdef ATT_STAMPS
  if (tid == 0 && bid_x == 1 && bid_y == 1 && bid_z == 0) {
    for (int i = 0; i < 6; ++i) g_att_dbg[i] = dbg_acc[i];
    g_att_dbg[6] = __builtin_amdgcn_s_memtime() - dbg_t0;
    g_att_dbg[7] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    g_att_dbg[8] = ntiles - t_beg;
  }
#endif
  if (!q_ok) return;
  const float LN2 = 0.6931471805599453f;
  if (part) {
    float* prow = part + ((long)bid_z * gridDim.y * N + base + q_row) * AT_PROW;
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // registers 4g .. 4g+3 are 4 consecutive channels: 16-byte stores
      const int ch = 8 * g + 4 * (lane >> 5);
      *reinterpret_cast<float4*>(prow + ch) = make_float4(o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]);
      *reinterpret_cast<float4*>(prow + 32 + ch) = make_float4(o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]);
    }
    if (lane < 32) { prow[64] = m_run * LN2; prow[65] = l_run; }
    return;
  }
  const float inv = 1.f / l_run;
  float* orow = out + (base + q_row) * ldo;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ch = r3d_acc_row(r, lane);
    orow[ch] = o0[r] * inv;
    orow[32 + ch] = o1[r] * inv;
  }
  if (lse_out && lane < 32) lse_out[base + q_row] = m_run * LN2 + __logf(l_run);
}

// merge the key splits of the forward: m = max m_z, l = sum l_z e^(m_z - m), o = sum o_z e^(m_z - m) / l
__global__ void r3d_attention_combine_kernel(const float* __restrict__ part, int nsplit, long M, float* __restrict__ out,
                                             long ldo, float* __restrict__ lse_out) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  float m = -INFINITY;
  for (int z = 0; z < nsplit; ++z) m = fmaxf(m, part[((long)z * M + row) * AT_PROW + 64]);
  float l = 0.f, o = 0.f;
  for (int z = 0; z < nsplit; ++z) {
    const float* pr = part + ((long)z * M + row) * AT_PROW;
    const float sc = __expf(pr[64] - m);
    l += pr[65] * sc;
    o += pr[lane] * sc;
  }
  out[row * ldo + lane] = o / l;
  if (lse_out && lane == 0) lse_out[row] = m + __logf(l);
}

// qkv: (B*N, ld) with q (pre-scaled by 1/sqrt(64)) | k | v at columns 0 | 64 | 128;
// out: (B*N, ldo) point-major, 64 columns written; lse_out optional (B*N) log-sum-exp
// per query (saved for the backward pass).
// Streamed-axis split: one workgroup per 128 rows leaves small grids (B * N / 128 workgroups, 32 for the two query
// clouds) with most CUs idle while every workgroup walks all N keys; the streamed axis is cut into `split` ranges
// (blockIdx.z) so that ~512 workgroups exist, and the partial results are merged by a small kernel in a fixed
// order (deterministic).  ws == NULL keeps the unsplit launch.
// The split is chosen per kernel against the workgroups the chip holds at once (`slots`): a grid slightly above a
// multiple of the slots pays a whole extra round of workgroups -- 640 workgroups on 512 slots ran at 62 %.  Cost model:
// rounds x (key tiles per workgroup + 2 tiles' worth of fixed work) + a merge term.
static int attention_split(int B, int N, int slots) {
  const int T = B * r3d_cdiv(N, 128), ntiles = r3d_cdiv(N, 32);
  // Default since round 3: NO split.  The split has to be the same whether an episode runs alone or in a batch (a
  // cloud's partials, their merge order and hence its output bits must not depend on the batch), and a 32-episode step has
  // 6144 query tiles to fill the chip with: split like ONE episode (4 ways forward, 2 ways backward at S) it paid 1.6 ms
  // per step for the merge kernels plus the extra prologues -- 349 -> 359 episodes/s without, against 7.80 -> 8.04 ms
  // for a single episode per step (profiles/r03_experiments.md).  R3D_ATT_SPLIT=0 restores the cost model below (chosen
  // per episode), R3D_ATT_SPLIT=n forces n.
  static const int forced = getenv("R3D_ATT_SPLIT") ? atoi(getenv("R3D_ATT_SPLIT")) : 1;
  if (forced > 0) return forced < ntiles ? forced : ntiles;
  int best = 1;
  float best_cost = 1e30f;
  for (int s = 1; s <= 16 && s <= ntiles; ++s) {
    const int tps = r3d_cdiv(ntiles, s), nz = r3d_cdiv(ntiles, tps);
    if (nz != s) continue;  // same launch as a smaller s
    const int rounds = r3d_cdiv(T * nz, slots);
    const float cost = (float)rounds * (float)(tps + 2) + 0.25f * (float)nz;
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}
enum { ATT_FWD = 0, ATT_BWD_KV = 1, ATT_BWD_Q = 2, ATT_FWD_BX3 = 3, ATT_BWD_KV_BX3 = 4, ATT_BWD_Q_BX3 = 5, ATT_N = 6 };
static int attention_slots(int which);  // defined below the kernels
// forward: split * M * AT_PROW; backward: M (row dots) + split * M * 128 (dK | dV partials, reused for dQ)
// Bs: the number of clouds the key-axis split is chosen for.  A batch of episodes is split like ONE episode (Bs = clouds
// per episode), so a cloud's partials, their merge order and hence its output bits are the same whether its episode
// runs alone or inside a batch.  (Moot with the default of no split, see attention_split.)
static long attention_part_words(int B, int N, int Bs) {
  int smax = 1;
  for (int w = 0; w < ATT_N; ++w) {
    const int sp = attention_split(Bs, N, attention_slots(w));
    smax = sp > smax ? sp : smax;
  }
  return (((long)B * N * (1 + 128L * smax) + 63) / 64) * 64;
}
extern "C" long r3d_attention_ws_words_ep(int B, int N, int group) {
  // ... followed by the packed bf16 x 3 operands (q | k | v | dO: 96 words per point each)
  return attention_part_words(B, N, group > 0 ? group : B) + 4L * B * N * (AB_ROW / 2) + 64;
}
extern "C" long r3d_attention_ws_words(int B, int N) { return r3d_attention_ws_words_ep(B, N, 0); }

static int attention_launch(const float* qkv, long ld, int B, int N, float* out, long ldo, float* lse_out, float p_drop,
                            unsigned seed, const unsigned* seed_dev, int seed_group, float* ws, void* stream) {
  R3D_REQUIRE(seed_group >= 0 && (seed_group == 0 || B % seed_group == 0), "r3d_attention_fwd: %d clouds in groups of %d", B,
              seed_group);
  R3D_REQUIRE(qkv && out, "r3d_attention_fwd: null pointer");
  R3D_REQUIRE(B > 0 && N > 0 && ld >= 192 && ld % 4 == 0 && ldo >= 64,
              "r3d_attention_fwd: bad shape B=%d N=%d ld=%ld ldo=%ld", B, N, ld, ldo);
  R3D_REQUIRE(((uintptr_t)qkv & 15) == 0, "r3d_attention_fwd: qkv must be 16-byte aligned");
  R3D_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "r3d_attention_fwd: dropout probability %f out of range", p_drop);
  const bool bx3 = g_r3d_matrix_arith == 1 && ws;  // the packed operands live in the workspace
  const int Bs = seed_group > 0 ? seed_group : B;
  const int split = ws ? attention_split(Bs, N, attention_slots(bx3 ? ATT_FWD_BX3 : ATT_FWD)) : 1;
  const int ntiles = r3d_cdiv(N, 32);
  const int tps = r3d_cdiv(ntiles, split);
  const int nz = r3d_cdiv(ntiles, tps);  // no empty split
  dim3 grid(r3d_cdiv(N, 128), B, nz);
  if (bx3) {
    const long M = (long)B * N;
    unsigned short* pk = reinterpret_cast<unsigned short*>(ws + attention_part_words(B, N, Bs));
    bx3_pack(qkv, ld, 3, M, pk, (hipStream_t)stream);  // q | k | v; a backward given the same workspace finds them there
    if (p_drop > 0.f)
      hipLaunchKernelGGL(r3d_attention_fwd_bx3_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, pk, pk + M * AB_ROW,
                         pk + 2 * M * AB_ROW, N, out, ldo, lse_out, p_drop, seed, seed_dev, seed_group, tps, nz > 1 ? ws : nullptr);
    else
      hipLaunchKernelGGL(r3d_attention_fwd_bx3_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, pk, pk + M * AB_ROW,
                         pk + 2 * M * AB_ROW, N, out, ldo, lse_out, p_drop, seed, seed_dev, seed_group, tps, nz > 1 ? ws : nullptr);
  } else
    hipLaunchKernelGGL(r3d_attention_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, ld, N, out,
                       ldo, lse_out, p_drop, seed, seed_dev, seed_group, tps, nz > 1 ? ws : nullptr);
  if (nz > 1) {
    const long M = (long)B * N;
    hipLaunchKernelGGL(r3d_attention_combine_kernel, dim3(r3d_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, ws, nz, M,
                       out, ldo, lse_out);
  }
  R3D_LAUNCH_CHECK("r3d_attention_fwd");
  return R3D_OK;
}

extern "C" int r3d_attention_fwd(const float* qkv, long ld, int B, int N, float* out, long ldo, float* lse_out,
                                 float* ws /* opt: r3d_attention_ws_words(B, N) floats enable the key split */,
                                 void* stream) {
  return attention_launch(qkv, ld, B, N, out, ldo, lse_out, 0.f, 0u, nullptr, 0, ws, stream);
}

// training forward: dropout p_drop on the attention weights with the stateless mask of attn_keep.
// Effective seed = seed + *seed_dev (seed_dev may be NULL): a captured hipGraph bumps the device word per replay.
extern "C" int r3d_attention_fwd_train(const float* qkv, long ld, int B, int N, float* out, long ldo, float* lse_out,
                                       float p_drop, unsigned seed, const unsigned* seed_dev, float* ws, void* stream) {
  R3D_REQUIRE(lse_out, "r3d_attention_fwd_train: lse_out is required (saved for the backward pass)");
  return attention_launch(qkv, ld, B, N, out, ldo, lse_out, p_drop, seed, seed_dev, 0, ws, stream);
}
// the same over a batch of episodes: clouds [e * seed_group, (e + 1) * seed_group) are episode e, whose dropout mask is
// the one a call on those clouds alone would draw with seed + 2 e (the eager schedule advances its seed by 2 per episode)
extern "C" int r3d_attention_fwd_train_ep(const float* qkv, long ld, int B, int N, float* out, long ldo, float* lse_out,
                                          float p_drop, unsigned seed, const unsigned* seed_dev, int seed_group, float* ws,
                                          void* stream) {
  R3D_REQUIRE(lse_out, "r3d_attention_fwd_train: lse_out is required (saved for the backward pass)");
  return attention_launch(qkv, ld, B, N, out, ldo, lse_out, p_drop, seed, seed_dev, seed_group, ws, stream);
}

// ---------------------------------------------------------------------------
// backward (flash style, P recomputed from the saved log-sum-exp)
//   D_i = <dO_i, O_i> ; dP~ = dO V^T ; dS = P (dP~ mask/(1-p) - D) ; dV = P~^T dO ; dK = dS^T Q' ; dQ' = dS K
// Kernel 1: one wave owns 32 keys, streams the query tiles, accumulates dK^T / dV^T in registers.
// Kernel 2: one wave owns 32 queries, streams the key tiles, accumulates dQ'^T.
// Both use the accumulator-as-operand trick of the forward kernel (sum over the accumulator ROW index).
// ---------------------------------------------------------------------------
__global__ void r3d_attention_rowdot_kernel(const float* __restrict__ dO, long lddo, const float* __restrict__ O, long ldo,
                                            long M, float* __restrict__ Dv) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  float v = dO[row * lddo + lane] * O[row * ldo + lane];
  v = r3d_wave_sum(v);
  if (lane == 0) Dv[row] = v;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void r3d_attention_bwd_kv_kernel(
    const float* __restrict__ qkv, long ld, int N, const float* __restrict__ dO, long lddo, const float* __restrict__ lse,
    const float* __restrict__ Dv, float* __restrict__ dqkv, long ldd, float p_drop, unsigned seed,
    const unsigned* __restrict__ seed_dev, int seed_group, int tiles_per_split, float* __restrict__ part /* [split][M][128] or NULL */) {
  // (plain block order.  An XCD-aware order -- common.h: r3d_xcd_swizzle, the workgroups sharing an L2 on the same clouds --
  // was measured and lost 3-8 % on these kernels: an XCD then holds 4 clouds' K / V pieces at a time, 6 MB against its 4 MB
  // L2, while in plain order the operands come out of the Infinity Cache: profiles/r03_experiments.md)
  const int bid_x = blockIdx.x, bid_y = blockIdx.y, bid_z = blockIdx.z;
  if (seed_dev) seed += *seed_dev;
  __shared__ float Qs[2][32 * AT_LD];
  __shared__ float Gs[2][32 * AT_LD];  // dO tile
  __shared__ float Ls[2][32], Ds[2][32];
  const unsigned thresh = p_drop > 0.f ? (unsigned)(p_drop * 4294967296.0) : 0u;
  const float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int b = bid_y;
  const long base = (long)b * N;
  unsigned hbase = (unsigned)base;  // row id the dropout hash sees
  if (seed_group > 0) {             // batch of episodes: every group of seed_group clouds is one episode with its own seed
    const int ep_ = b / seed_group;
    seed += 2u * (unsigned)ep_;
    hbase = (unsigned)((b - ep_ * seed_group) * N);
  }
  const int key = bid_x * 128 + 32 * w + j;  // this lane's key column
  const bool key_ok = key < N;
  float bk[32], bv[32];  // B[k = ch][j = key] fragments of K and V
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    const float* p = qkv + (base + min(key, N - 1)) * ld + 2 * s + h;
    bk[s] = r3d_keep(p[64], key_ok);
    bv[s] = r3d_keep(p[128], key_ok);
  }
  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
  const int st_row = tid >> 3, st_c4 = (tid & 7) * 2;
  float4 qreg[2], greg[2];
  float lreg = 0.f, dreg = 0.f;
  auto load_tile = [&](int q0) {
    const int qr = q0 + st_row;
    const bool ok = qr < N;
    const int qc = min(qr, N - 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 qq = *reinterpret_cast<const float4*>(qkv + (base + qc) * ld + 4 * (st_c4 + i));
      const float4 gg = *reinterpret_cast<const float4*>(dO + (base + qc) * lddo + 4 * (st_c4 + i));
      qreg[i] = make_float4(r3d_keep(qq.x, ok), r3d_keep(qq.y, ok), r3d_keep(qq.z, ok), r3d_keep(qq.w, ok));
      greg[i] = make_float4(r3d_keep(gg.x, ok), r3d_keep(gg.y, ok), r3d_keep(gg.z, ok), r3d_keep(gg.w, ok));
    }
    if (tid < 32) {
      const int q2 = min(q0 + tid, N - 1);
      lreg = lse[base + q2];
      dreg = Dv[base + q2];
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float* qd = &Qs[buf][st_row * AT_LD + 4 * (st_c4 + i)];
      float* gd = &Gs[buf][st_row * AT_LD + 4 * (st_c4 + i)];
      qd[0] = qreg[i].x; qd[1] = qreg[i].y; qd[2] = qreg[i].z; qd[3] = qreg[i].w;
      gd[0] = greg[i].x; gd[1] = greg[i].y; gd[2] = greg[i].z; gd[3] = greg[i].w;
    }
    if (tid < 32) { Ls[buf][tid] = lreg; Ds[buf][tid] = dreg; }
  };
  const int t_beg = bid_z * tiles_per_split;
  const int ntiles = min((N + 31) / 32, t_beg + tiles_per_split);
  load_tile(32 * t_beg);
  store_tile(t_beg & 1);
  __syncthreads();
  for (int t = t_beg; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) load_tile(32 * (t + 1));
    // S[query][key] = Q' K^T ; dP~[query][key] = dO V^T   (rows = queries of the tile, column = this lane's key)
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    {
      const float* qp = &Qs[buf][j * AT_LD + h];
      const float* gp = &Gs[buf][j * AT_LD + h];
#pragma unroll
      for (int st = 0; st < 32; ++st) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(qp[2 * st], bk[st], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(gp[2 * st], bv[st], dp, 0, 0, 0);
      }
    }
    f32x16 pt;  // P~ (dropped, scaled)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ql = r3d_acc_row(r, lane);
      const int q = 32 * t + ql;
      float p = (q < N && key_ok) ? __expf(s[r] - Ls[buf][ql]) : 0.f;
      float keep = 1.f;
      if (thresh) keep = attn_keep(seed, hbase + (unsigned)(q), (unsigned)key, thresh) ? keep_scale : 0.f;
      pt[r] = p * keep;
      s[r] = p * (dp[r] * keep - Ds[buf][ql]);  // dS
    }
    // dV^T[c][key] += sum_q dO[q][c] P~[q][key] ;  dK^T[c][key] += sum_q Q'[q][c] dS[q][key]
    {
      const float* gp = &Gs[buf][j];
      const float* qp = &Qs[buf][j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = r3d_acc_row(r, lane);
        dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(gp[ql * AT_LD], pt[r], dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(gp[ql * AT_LD + 32], pt[r], dv1, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qp[ql * AT_LD], s[r], dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qp[ql * AT_LD + 32], s[r], dk1, 0, 0, 0);
      }
    }
    if (t + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();
  }
  if (!key_ok) return;
  float* drow = part ? part + ((long)bid_z * gridDim.y * N + base + key) * 128 - 64 : dqkv + (base + key) * ldd;
  if (part) {  // 16-byte rows: registers 4g .. 4g+3 are 4 consecutive channels
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 8 * g + 4 * h;
      *reinterpret_cast<float4*>(drow + 64 + c) = make_float4(dk0[4 * g], dk0[4 * g + 1], dk0[4 * g + 2], dk0[4 * g + 3]);
      *reinterpret_cast<float4*>(drow + 96 + c) = make_float4(dk1[4 * g], dk1[4 * g + 1], dk1[4 * g + 2], dk1[4 * g + 3]);
      *reinterpret_cast<float4*>(drow + 128 + c) = make_float4(dv0[4 * g], dv0[4 * g + 1], dv0[4 * g + 2], dv0[4 * g + 3]);
      *reinterpret_cast<float4*>(drow + 160 + c) = make_float4(dv1[4 * g], dv1[4 * g + 1], dv1[4 * g + 2], dv1[4 * g + 3]);
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = r3d_acc_row(r, lane);
    drow[64 + c] = dk0[r];
    drow[64 + 32 + c] = dk1[r];
    drow[128 + c] = dv0[r];
    drow[128 + 32 + c] = dv1[r];
  }
}

// out[row][col0 + c] = scale * sum_z part[z][row][c], z ascending (fixed order), c < width
__global__ void r3d_attention_sum_kernel(const float* __restrict__ part, int nsplit, long M, int width, float scale,
                                         float* __restrict__ out, long ldo, int col0) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * width) return;
  const long row = i / width;
  const int c = (int)(i - row * width);
  float s = 0.f;
  for (int z = 0; z < nsplit; ++z) s += part[((long)z * M + row) * width + c];
  out[row * ldo + col0 + c] = s * scale;
}

__global__ __launch_bounds__(256) void r3d_attention_bwd_q_kernel(
    const float* __restrict__ qkv, long ld, int N, const float* __restrict__ dO, long lddo, const float* __restrict__ lse,
    const float* __restrict__ Dv, float* __restrict__ dqkv, long ldd, float p_drop, unsigned seed,
    const unsigned* __restrict__ seed_dev, int seed_group, float q_scale, int tiles_per_split,
    float* __restrict__ part /* [split][M][64] unscaled, or NULL */) {
  // (plain block order.  An XCD-aware order -- common.h: r3d_xcd_swizzle, the workgroups sharing an L2 on the same clouds --
  // was measured and lost 3-8 % on these kernels: an XCD then holds 4 clouds' K / V pieces at a time, 6 MB against its 4 MB
  // L2, while in plain order the operands come out of the Infinity Cache: profiles/r03_experiments.md)
  const int bid_x = blockIdx.x, bid_y = blockIdx.y, bid_z = blockIdx.z;
  if (seed_dev) seed += *seed_dev;
  __shared__ float Ks[2][32 * AT_LD];
  __shared__ float Vs[2][32 * AT_LD];
  const unsigned thresh = p_drop > 0.f ? (unsigned)(p_drop * 4294967296.0) : 0u;
  const float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int b = bid_y;
  const long base = (long)b * N;
  unsigned hbase = (unsigned)base;  // row id the dropout hash sees
  if (seed_group > 0) {             // batch of episodes: every group of seed_group clouds is one episode with its own seed
    const int ep_ = b / seed_group;
    seed += 2u * (unsigned)ep_;
    hbase = (unsigned)((b - ep_ * seed_group) * N);
  }
  const int q_row = bid_x * 128 + 32 * w + j;
  const bool q_ok = q_row < N;
  float bq[32], bg[32];  // B[k = ch][j = query] fragments of Q' and dO
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    bq[s] = r3d_keep(qkv[(base + min(q_row, N - 1)) * ld + 2 * s + h], q_ok);
    bg[s] = r3d_keep(dO[(base + min(q_row, N - 1)) * lddo + 2 * s + h], q_ok);
  }
  const float my_lse = lse[base + min(q_row, N - 1)];
  const float my_D = Dv[base + min(q_row, N - 1)];
  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  const int st_row = tid >> 3, st_c4 = (tid & 7) * 2;
  float4 kreg[2], vreg[2];
  auto load_tile = [&](int key0) {
    const int kr = key0 + st_row;
    const bool ok = kr < N;
    const int krc = min(kr, N - 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* p = qkv + (base + krc) * ld + 4 * (st_c4 + i);
      const float4 kk = *reinterpret_cast<const float4*>(p + 64);
      const float4 vv = *reinterpret_cast<const float4*>(p + 128);
      kreg[i] = make_float4(r3d_keep(kk.x, ok), r3d_keep(kk.y, ok), r3d_keep(kk.z, ok), r3d_keep(kk.w, ok));
      vreg[i] = make_float4(r3d_keep(vv.x, ok), r3d_keep(vv.y, ok), r3d_keep(vv.z, ok), r3d_keep(vv.w, ok));
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float* kd = &Ks[buf][st_row * AT_LD + 4 * (st_c4 + i)];
      float* vd = &Vs[buf][st_row * AT_LD + 4 * (st_c4 + i)];
      kd[0] = kreg[i].x; kd[1] = kreg[i].y; kd[2] = kreg[i].z; kd[3] = kreg[i].w;
      vd[0] = vreg[i].x; vd[1] = vreg[i].y; vd[2] = vreg[i].z; vd[3] = vreg[i].w;
    }
  };
  const int t_beg = bid_z * tiles_per_split;
  const int ntiles = min((N + 31) / 32, t_beg + tiles_per_split);
  load_tile(32 * t_beg);
  store_tile(t_beg & 1);
  __syncthreads();
  for (int t = t_beg; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) load_tile(32 * (t + 1));
    // S^T[key][query] = K Q'^T ; dP~^T[key][query] = V dO^T
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    {
      const float* kp = &Ks[buf][j * AT_LD + h];
      const float* vp = &Vs[buf][j * AT_LD + h];
#pragma unroll
      for (int st = 0; st < 32; ++st) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * st], bq[st], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[2 * st], bg[st], dp, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 * t + r3d_acc_row(r, lane);
      const float p = (key < N && q_ok) ? __expf(s[r] - my_lse) : 0.f;
      float keep = 1.f;
      if (thresh) keep = attn_keep(seed, hbase + (unsigned)(q_row), (unsigned)key, thresh) ? keep_scale : 0.f;
      s[r] = p * (dp[r] * keep - my_D);  // dS^T
    }
    // dQ'^T[c][query] += sum_key K[key][c] dS^T[key][query]
    {
      const float* kp = &Ks[buf][j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kl = r3d_acc_row(r, lane);
        dq0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[kl * AT_LD], s[r], dq0, 0, 0, 0);
        dq1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[kl * AT_LD + 32], s[r], dq1, 0, 0, 0);
      }
    }
    if (t + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();
  }
  if (!q_ok) return;
  float* drow = part ? part + ((long)bid_z * gridDim.y * N + base + q_row) * 64 : dqkv + (base + q_row) * ldd;
  const float osc = part ? 1.f : q_scale;  // partials stay unscaled; r3d_attention_sum_kernel applies q_scale
  if (part) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 8 * g + 4 * h;
      *reinterpret_cast<float4*>(drow + c) = make_float4(dq0[4 * g], dq0[4 * g + 1], dq0[4 * g + 2], dq0[4 * g + 3]);
      *reinterpret_cast<float4*>(drow + 32 + c) = make_float4(dq1[4 * g], dq1[4 * g + 1], dq1[4 * g + 2], dq1[4 * g + 3]);
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = r3d_acc_row(r, lane);
    drow[c] = dq0[r] * osc;  // gradient w.r.t. the UNscaled q map output (q' = q * q_scale)
    drow[32 + c] = dq1[r] * osc;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The two backward kernels on the bf16 matrix core (common.h "bf16 x 3"): operands packed by r3d_bx3_pack_kernel,
// streamed tiles staged by the LDS-DMA into dual-use images (row reads for the products that sum over channels, column
// reads for the products that sum over the tile's rows), stationary operands as B fragments in registers, P~ / dS cut
// into pieces straight from the accumulator registers.  96 (kv) and 72 (q) bf16 MFMAs per 32-row tile and wave against
// 128 and 96 fp32 MFMAs of twice the cycles.
template <bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void r3d_attention_bwd_kv_bx3_kernel(
    const unsigned short* __restrict__ Qp, const unsigned short* __restrict__ Kp, const unsigned short* __restrict__ Vp,
    const unsigned short* __restrict__ Gp /* dO */, int N, const float* __restrict__ lse, const float* __restrict__ Dv,
    float* __restrict__ dqkv, long ldd, float p_drop, unsigned seed, const unsigned* __restrict__ seed_dev, int seed_group,
    int tiles_per_split, float* __restrict__ part) {
  // (plain block order.  An XCD-aware order -- common.h: r3d_xcd_swizzle, the workgroups sharing an L2 on the same clouds --
  // was measured and lost 3-8 % on these kernels: an XCD then holds 4 clouds' K / V pieces at a time, 6 MB against its 4 MB
  // L2, while in plain order the operands come out of the Infinity Cache: profiles/r03_experiments.md)
  const int bid_x = blockIdx.x, bid_y = blockIdx.y, bid_z = blockIdx.z;
  if (seed_dev) seed += *seed_dev;
  __shared__ __attribute__((aligned(16))) unsigned short Qs[2][AG_TILE];
  __shared__ __attribute__((aligned(16))) unsigned short Gs[2][AG_TILE];
  __shared__ float Ls[2][32], Ds[2][32];  // lse * log2(e) and D of the tile's queries
  const unsigned thresh = DROP ? (unsigned)(p_drop * 4294967296.0) : 0u;
  const float keep_scale = DROP ? 1.f / (1.f - p_drop) : 1.f;
  const float LOG2E = 1.4426950408889634f;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;
  const int b = bid_y;
  const long base = (long)b * N;
  unsigned hbase = (unsigned)base;  // row id the dropout hash sees
  if (seed_group > 0) {             // batch of episodes: every group of seed_group clouds is one episode with its own seed
    const int ep_ = b / seed_group;
    seed += 2u * (unsigned)ep_;
    hbase = (unsigned)((b - ep_ * seed_group) * N);
  }
  const int key = bid_x * 128 + 32 * w + j;  // this lane's key column
  const bool key_ok = key < N;
  r3d_bx3 bk[4], bv[4];  // B[k = ch][j = key] fragments of K and V
  ab_load_row_frags(Kp + (base + min(key, N - 1)) * AB_ROW, h, key_ok, bk);
  ab_load_row_frags(Vp + (base + min(key, N - 1)) * AB_ROW, h, key_ok, bv);
  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
  const ag_offs offs = ag_make_offs(lane);
  const unsigned doff = ag_dma_off(w, lane);
  const int t_beg = bid_z * tiles_per_split;
  const int ntiles = min((N + 31) / 32, t_beg + tiles_per_split);
  float lreg = 0.f, dreg = 0.f;
  auto load_rows = [&](int q0) {
    if (tid < 32) {
      const int q2 = min(q0 + tid, N - 1);
      lreg = lse[base + q2] * LOG2E;
      dreg = Dv[base + q2];
    }
  };
  ag_dma_tile(Qp, base, 32 * t_beg, N, Qs[t_beg & 1], w, lane, doff);
  ag_dma_tile(Gp, base, 32 * t_beg, N, Gs[t_beg & 1], w, lane, doff);
  load_rows(32 * t_beg);
  if (tid < 32) { Ls[t_beg & 1][tid] = lreg; Ds[t_beg & 1][tid] = dreg; }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int t = t_beg; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) {
      ag_dma_tile(Qp, base, 32 * (t + 1), N, Qs[buf ^ 1], w, lane, doff);
      ag_dma_tile(Gp, base, 32 * (t + 1), N, Gs[buf ^ 1], w, lane, doff);
      load_rows(32 * (t + 1));
    }
    // S[query][key] = Q' K^T ; dP~[query][key] = dO V^T   (rows = queries of the tile, column = this lane's key)
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      s = r3d_bx3_mma(ag_row_frag(Qs[buf], offs, st), bk[st], s);
      dp = r3d_bx3_mma(ag_row_frag(Gs[buf], offs, st), bv[st], dp);
    }
    // P~ (dropped, scaled) stays in s, dS goes to dp
    const bool tail = 32 * (t + 1) > N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ql = r3d_acc_row(r, lane);
      float p = __builtin_amdgcn_exp2f(fmaf(s[r], LOG2E, -Ls[buf][ql]));
      p = r3d_keep(p, key_ok && !(tail && 32 * t + ql >= N));
      float keep = 1.f;
      if (DROP) keep = attn_keep(seed, hbase + (unsigned)(32 * t + ql), (unsigned)key, thresh) ? keep_scale : 0.f;
      s[r] = p * keep;
      dp[r] = p * (dp[r] * keep - Ds[buf][ql]);
    }
    // dV^T[c][key] += sum_q dO[q][c] P~[q][key] ;  dK^T[c][key] += sum_q Q'[q][c] dS[q][key]
#pragma unroll
    for (int sI = 0; sI < 2; ++sI) {
      const r3d_bx3 pf = ab_split_acc(s, sI);
      dv0 = r3d_bx3_mma(ag_col_frag(Gs[buf], offs, sI, 0), pf, dv0);
      dv1 = r3d_bx3_mma(ag_col_frag(Gs[buf], offs, sI, 1), pf, dv1);
      const r3d_bx3 df = ab_split_acc(dp, sI);
      dk0 = r3d_bx3_mma(ag_col_frag(Qs[buf], offs, sI, 0), df, dk0);
      dk1 = r3d_bx3_mma(ag_col_frag(Qs[buf], offs, sI, 1), df, dk1);
    }
    if (t + 1 < ntiles && tid < 32) { Ls[buf ^ 1][tid] = lreg; Ds[buf ^ 1][tid] = dreg; }
    __builtin_amdgcn_s_waitcnt(0);  // this wave's DMA has landed
    __syncthreads();
  }
  if (!key_ok) return;
  float* drow = part ? part + ((long)bid_z * gridDim.y * N + base + key) * 128 - 64 : dqkv + (base + key) * ldd;
  if (part) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 8 * g + 4 * h;
      *reinterpret_cast<float4*>(drow + 64 + c) = make_float4(dk0[4 * g], dk0[4 * g + 1], dk0[4 * g + 2], dk0[4 * g + 3]);
      *reinterpret_cast<float4*>(drow + 96 + c) = make_float4(dk1[4 * g], dk1[4 * g + 1], dk1[4 * g + 2], dk1[4 * g + 3]);
      *reinterpret_cast<float4*>(drow + 128 + c) = make_float4(dv0[4 * g], dv0[4 * g + 1], dv0[4 * g + 2], dv0[4 * g + 3]);
      *reinterpret_cast<float4*>(drow + 160 + c) = make_float4(dv1[4 * g], dv1[4 * g + 1], dv1[4 * g + 2], dv1[4 * g + 3]);
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = r3d_acc_row(r, lane);
    drow[64 + c] = dk0[r];
    drow[64 + 32 + c] = dk1[r];
    drow[128 + c] = dv0[r];
    drow[128 + 32 + c] = dv1[r];
  }
}

template <bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void r3d_attention_bwd_q_bx3_kernel(
    const unsigned short* __restrict__ Qp, const unsigned short* __restrict__ Kp, const unsigned short* __restrict__ Vp,
    const unsigned short* __restrict__ Gp /* dO */, int N, const float* __restrict__ lse, const float* __restrict__ Dv,
    float* __restrict__ dqkv, long ldd, float p_drop, unsigned seed, const unsigned* __restrict__ seed_dev, int seed_group, float q_scale,
    int tiles_per_split, float* __restrict__ part) {
  // (plain block order.  An XCD-aware order -- common.h: r3d_xcd_swizzle, the workgroups sharing an L2 on the same clouds --
  // was measured and lost 3-8 % on these kernels: an XCD then holds 4 clouds' K / V pieces at a time, 6 MB against its 4 MB
  // L2, while in plain order the operands come out of the Infinity Cache: profiles/r03_experiments.md)
  const int bid_x = blockIdx.x, bid_y = blockIdx.y, bid_z = blockIdx.z;
  if (seed_dev) seed += *seed_dev;
  __shared__ __attribute__((aligned(16))) unsigned short Ks[2][AG_TILE];
  __shared__ __attribute__((aligned(16))) unsigned short Vs[2][AG_TILE];
  const unsigned thresh = DROP ? (unsigned)(p_drop * 4294967296.0) : 0u;
  const float keep_scale = DROP ? 1.f / (1.f - p_drop) : 1.f;
  const float LOG2E = 1.4426950408889634f;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;
  const int b = bid_y;
  const long base = (long)b * N;
  unsigned hbase = (unsigned)base;  // row id the dropout hash sees
  if (seed_group > 0) {             // batch of episodes: every group of seed_group clouds is one episode with its own seed
    const int ep_ = b / seed_group;
    seed += 2u * (unsigned)ep_;
    hbase = (unsigned)((b - ep_ * seed_group) * N);
  }
  const int q_row = bid_x * 128 + 32 * w + j;
  const bool q_ok = q_row < N;
  r3d_bx3 bq[4], bg[4];  // B[k = ch][j = query] fragments of Q' and dO
  ab_load_row_frags(Qp + (base + min(q_row, N - 1)) * AB_ROW, h, q_ok, bq);
  ab_load_row_frags(Gp + (base + min(q_row, N - 1)) * AB_ROW, h, q_ok, bg);
  const float my_lse2 = lse[base + min(q_row, N - 1)] * LOG2E;
  const float my_D = Dv[base + min(q_row, N - 1)];
  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  const ag_offs offs = ag_make_offs(lane);
  const unsigned doff = ag_dma_off(w, lane);
  const int t_beg = bid_z * tiles_per_split;
  const int ntiles = min((N + 31) / 32, t_beg + tiles_per_split);
  ag_dma_tile(Kp, base, 32 * t_beg, N, Ks[t_beg & 1], w, lane, doff);
  ag_dma_tile(Vp, base, 32 * t_beg, N, Vs[t_beg & 1], w, lane, doff);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int t = t_beg; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) {
      ag_dma_tile(Kp, base, 32 * (t + 1), N, Ks[buf ^ 1], w, lane, doff);
      ag_dma_tile(Vp, base, 32 * (t + 1), N, Vs[buf ^ 1], w, lane, doff);
    }
    // S^T[key][query] = K Q'^T ; dP~^T[key][query] = V dO^T
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      s = r3d_bx3_mma(ag_row_frag(Ks[buf], offs, st), bq[st], s);
      dp = r3d_bx3_mma(ag_row_frag(Vs[buf], offs, st), bg[st], dp);
    }
    const bool tail = 32 * (t + 1) > N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kl = r3d_acc_row(r, lane);
      float p = __builtin_amdgcn_exp2f(fmaf(s[r], LOG2E, -my_lse2));
      p = r3d_keep(p, q_ok && !(tail && 32 * t + kl >= N));
      float keep = 1.f;
      if (DROP) keep = attn_keep(seed, hbase + (unsigned)(q_row), (unsigned)(32 * t + kl), thresh) ? keep_scale : 0.f;
      s[r] = p * (dp[r] * keep - my_D);  // dS^T
    }
    // dQ'^T[c][query] += sum_key K[key][c] dS^T[key][query]
#pragma unroll
    for (int sI = 0; sI < 2; ++sI) {
      const r3d_bx3 df = ab_split_acc(s, sI);
      dq0 = r3d_bx3_mma(ag_col_frag(Ks[buf], offs, sI, 0), df, dq0);
      dq1 = r3d_bx3_mma(ag_col_frag(Ks[buf], offs, sI, 1), df, dq1);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
  }
  if (!q_ok) return;
  float* drow = part ? part + ((long)bid_z * gridDim.y * N + base + q_row) * 64 : dqkv + (base + q_row) * ldd;
  if (part) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 8 * g + 4 * h;
      *reinterpret_cast<float4*>(drow + c) = make_float4(dq0[4 * g], dq0[4 * g + 1], dq0[4 * g + 2], dq0[4 * g + 3]);
      *reinterpret_cast<float4*>(drow + 32 + c) = make_float4(dq1[4 * g], dq1[4 * g + 1], dq1[4 * g + 2], dq1[4 * g + 3]);
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = r3d_acc_row(r, lane);
    drow[c] = dq0[r] * q_scale;  // gradient w.r.t. the UNscaled q map output (q' = q * q_scale)
    drow[32 + c] = dq1[r] * q_scale;
  }
}

// dqkv (B*N, ldd >= 192): gradients of the q | k | v GEMM outputs (before the 1/sqrt(d) scale of q).
// O: forward output (B*N, ldo); lse: saved log-sum-exp; ws: r3d_attention_ws_words(B, N) floats (row dots + the
// partial dK | dV / dQ of the streamed-axis split).
// ws_holds_packed_qkv: ws is the workspace the forward of the SAME qkv ran with (r3d_attention_fwd_train) and nothing
// has written to it since: its packed q | k | v pieces are reused instead of cut again (bf16 x 3 arithmetic only).
extern "C" int r3d_attention_bwd_ep(const float* qkv, long ld, int B, int N, const float* O, long ldo, const float* dO,
                                    long lddo, const float* lse, float p_drop, unsigned seed, const unsigned* seed_dev,
                                    int seed_group, float q_scale, float* dqkv, long ldd, float* ws, int ws_holds_packed_qkv,
                                    void* stream) {
  R3D_REQUIRE(qkv && O && dO && lse && dqkv && ws, "r3d_attention_bwd: null pointer");
  R3D_REQUIRE(seed_group >= 0 && (seed_group == 0 || B % seed_group == 0), "r3d_attention_bwd: %d clouds in groups of %d", B,
              seed_group);
  R3D_REQUIRE(B > 0 && N > 0 && ld >= 192 && ld % 4 == 0 && lddo % 4 == 0 && ldd >= 192 && ldo >= 64,
              "r3d_attention_bwd: bad shape");
  R3D_REQUIRE((((uintptr_t)qkv | (uintptr_t)dO) & 15) == 0, "r3d_attention_bwd: qkv and dO must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const long M = (long)B * N;
  hipLaunchKernelGGL(r3d_attention_rowdot_kernel, dim3(r3d_cdiv(M, 4)), dim3(256), 0, st, dO, lddo, O, ldo, M, ws);
  const int ntiles = r3d_cdiv(N, 32);
  const int Bs = seed_group > 0 ? seed_group : B;
  if (g_r3d_matrix_arith == 1) {
    unsigned short* pk = reinterpret_cast<unsigned short*>(ws + attention_part_words(B, N, Bs));
    const unsigned short *Qp = pk, *Kp = pk + M * AB_ROW, *Vp = pk + 2 * M * AB_ROW, *Gp = pk + 3 * M * AB_ROW;
    if (!ws_holds_packed_qkv) bx3_pack(qkv, ld, 3, M, pk, st);
    bx3_pack(dO, lddo, 1, M, pk + 3 * M * AB_ROW, st);
    {
      const int tps = r3d_cdiv(ntiles, attention_split(Bs, N, attention_slots(ATT_BWD_KV_BX3)));
      const int nz = r3d_cdiv(ntiles, tps);
      float* part = nz > 1 ? ws + M : nullptr;
      dim3 grid(r3d_cdiv(N, 128), B, nz);
      if (p_drop > 0.f)
        hipLaunchKernelGGL(r3d_attention_bwd_kv_bx3_kernel<true>, grid, dim3(256), 0, st, Qp, Kp, Vp, Gp, N, lse, ws, dqkv, ldd,
                           p_drop, seed, seed_dev, seed_group, tps, part);
      else
        hipLaunchKernelGGL(r3d_attention_bwd_kv_bx3_kernel<false>, grid, dim3(256), 0, st, Qp, Kp, Vp, Gp, N, lse, ws, dqkv, ldd,
                           p_drop, seed, seed_dev, seed_group, tps, part);
      if (part)
        hipLaunchKernelGGL(r3d_attention_sum_kernel, dim3(r3d_cdiv(M * 128, 256)), dim3(256), 0, st, part, nz, M, 128, 1.f, dqkv,
                           ldd, 64);
    }
    {
      const int tps = r3d_cdiv(ntiles, attention_split(Bs, N, attention_slots(ATT_BWD_Q_BX3)));
      const int nz = r3d_cdiv(ntiles, tps);
      float* part = nz > 1 ? ws + M : nullptr;
      dim3 grid(r3d_cdiv(N, 128), B, nz);
      if (p_drop > 0.f)
        hipLaunchKernelGGL(r3d_attention_bwd_q_bx3_kernel<true>, grid, dim3(256), 0, st, Qp, Kp, Vp, Gp, N, lse, ws, dqkv, ldd,
                           p_drop, seed, seed_dev, seed_group, q_scale, tps, part);
      else
        hipLaunchKernelGGL(r3d_attention_bwd_q_bx3_kernel<false>, grid, dim3(256), 0, st, Qp, Kp, Vp, Gp, N, lse, ws, dqkv, ldd,
                           p_drop, seed, seed_dev, seed_group, q_scale, tps, part);
      if (part)
        hipLaunchKernelGGL(r3d_attention_sum_kernel, dim3(r3d_cdiv(M * 64, 256)), dim3(256), 0, st, part, nz, M, 64, q_scale, dqkv,
                           ldd, 0);
    }
    R3D_LAUNCH_CHECK("r3d_attention_bwd");
    return R3D_OK;
  }
  {
    const int tps = r3d_cdiv(ntiles, attention_split(Bs, N, attention_slots(ATT_BWD_KV)));
    const int nz = r3d_cdiv(ntiles, tps);
    float* part = nz > 1 ? ws + M : nullptr;
    hipLaunchKernelGGL(r3d_attention_bwd_kv_kernel, dim3(r3d_cdiv(N, 128), B, nz), dim3(256), 0, st, qkv, ld, N, dO, lddo, lse, ws,
                       dqkv, ldd, p_drop, seed, seed_dev, seed_group, tps, part);
    if (part)
      hipLaunchKernelGGL(r3d_attention_sum_kernel, dim3(r3d_cdiv(M * 128, 256)), dim3(256), 0, st, part, nz, M, 128, 1.f, dqkv,
                         ldd, 64);
  }
  {
    const int tps = r3d_cdiv(ntiles, attention_split(Bs, N, attention_slots(ATT_BWD_Q)));
    const int nz = r3d_cdiv(ntiles, tps);
    float* part = nz > 1 ? ws + M : nullptr;
    hipLaunchKernelGGL(r3d_attention_bwd_q_kernel, dim3(r3d_cdiv(N, 128), B, nz), dim3(256), 0, st, qkv, ld, N, dO, lddo, lse, ws,
                       dqkv, ldd, p_drop, seed, seed_dev, seed_group, q_scale, tps, part);
    if (part)
      hipLaunchKernelGGL(r3d_attention_sum_kernel, dim3(r3d_cdiv(M * 64, 256)), dim3(256), 0, st, part, nz, M, 64, q_scale, dqkv,
                         ldd, 0);
  }
  R3D_LAUNCH_CHECK("r3d_attention_bwd");
  return R3D_OK;
}
extern "C" int r3d_attention_bwd_ws(const float* qkv, long ld, int B, int N, const float* O, long ldo, const float* dO,
                                    long lddo, const float* lse, float p_drop, unsigned seed, const unsigned* seed_dev,
                                    float q_scale, float* dqkv, long ldd, float* ws, int ws_holds_packed_qkv, void* stream) {
  return r3d_attention_bwd_ep(qkv, ld, B, N, O, ldo, dO, lddo, lse, p_drop, seed, seed_dev, 0, q_scale, dqkv, ldd, ws,
                              ws_holds_packed_qkv, stream);
}
extern "C" int r3d_attention_bwd(const float* qkv, long ld, int B, int N, const float* O, long ldo, const float* dO,
                                 long lddo, const float* lse, float p_drop, unsigned seed, const unsigned* seed_dev,
                                 float q_scale, float* dqkv, long ldd, float* ws, void* stream) {
  return r3d_attention_bwd_ws(qkv, ld, B, N, O, ldo, dO, lddo, lse, p_drop, seed, seed_dev, q_scale, dqkv, ldd, ws, 0, stream);
}


// workgroups (256 threads) of each attention kernel the chip holds at once; the defaults stand in when there is no
// device to ask (host-only sizing calls)
static int attention_slots(int which) {
  static int cache[ATT_N] = {0, 0, 0, 0, 0, 0};
  if (cache[which]) return cache[which];
  const int fallback[ATT_N] = {512, 256, 512, 768, 512, 512};
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  hipError_t e = hipErrorUnknown;
  if (which == ATT_FWD) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, r3d_attention_fwd_kernel, 256, 0);
  else if (which == ATT_FWD_BX3) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, r3d_attention_fwd_bx3_kernel<false>, 256, 0);
  else if (which == ATT_BWD_KV_BX3) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, r3d_attention_bwd_kv_bx3_kernel<false>, 256, 0);
  else if (which == ATT_BWD_Q_BX3) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, r3d_attention_bwd_q_bx3_kernel<false>, 256, 0);
  else if (which == ATT_BWD_KV) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, r3d_attention_bwd_kv_kernel, 256, 0);
  else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, r3d_attention_bwd_q_kernel, 256, 0);
  if (e == hipSuccess && per_cu > 0 && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
    cache[which] = per_cu * prop.multiProcessorCount;
  else {
    (void)hipGetLastError();
    cache[which] = fallback[which];
  }
  return cache[which];
}
