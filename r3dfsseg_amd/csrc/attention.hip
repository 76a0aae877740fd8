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
#include "common.h"

#define AT_LD 65

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
    float* __restrict__ lse_out, float p_drop, unsigned seed, const unsigned* __restrict__ seed_dev,
    int tiles_per_split, float* __restrict__ part /* split > 1: [split][M][66] = unnormalised o | m | l */) {
  if (seed_dev) seed += *seed_dev;  // per-replay seed of a captured hipGraph lives in device memory
  const unsigned thresh = p_drop > 0.f ? (unsigned)(p_drop * 4294967296.0) : 0u;
  const float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  __shared__ float Ks[2][32 * AT_LD];
  __shared__ float Vs[2][32 * AT_LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.y;
  const long base = (long)b * N;
  const int q_row = blockIdx.x * 128 + 32 * w + (lane & 31);
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

  // key tiles [t_beg, t_end) of this workgroup (blockIdx.z = split of the key axis: more workgroups for small grids)
  const int t_beg = blockIdx.z * tiles_per_split;
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
        s[r] = attn_keep(seed, (unsigned)(base + q_row), (unsigned)(32 * t + r3d_acc_row(r, lane)), thresh) ? s[r] * keep_scale : 0.f;
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
    float* prow = part + ((long)blockIdx.z * gridDim.y * N + base + q_row) * 66;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ch = r3d_acc_row(r, lane);
      prow[ch] = o0[r];
      prow[32 + ch] = o1[r];
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

// merge the key splits of the forward: m = max m_z, l = sum l_z e^(m_z - m), o = sum o_z e^(m_z - m) / l
__global__ void r3d_attention_combine_kernel(const float* __restrict__ part, int nsplit, long M, float* __restrict__ out,
                                             long ldo, float* __restrict__ lse_out) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  float m = -INFINITY;
  for (int z = 0; z < nsplit; ++z) m = fmaxf(m, part[((long)z * M + row) * 66 + 64]);
  float l = 0.f, o = 0.f;
  for (int z = 0; z < nsplit; ++z) {
    const float* pr = part + ((long)z * M + row) * 66;
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
enum { ATT_FWD = 0, ATT_BWD_KV = 1, ATT_BWD_Q = 2 };
static int attention_slots(int which);  // defined below the kernels
extern "C" long r3d_attention_ws_words(int B, int N) {
  // forward: split * M * 66; backward: M (row dots) + split * M * 128 (dK | dV partials, reused for dQ)
  int smax = 1;
  for (int w = 0; w < 3; ++w) {
    const int sp = attention_split(B, N, attention_slots(w));
    smax = sp > smax ? sp : smax;
  }
  return (long)B * N * (1 + 128L * smax) + 64;
}

static int attention_launch(const float* qkv, long ld, int B, int N, float* out, long ldo, float* lse_out, float p_drop,
                            unsigned seed, const unsigned* seed_dev, float* ws, void* stream) {
  R3D_REQUIRE(qkv && out, "r3d_attention_fwd: null pointer");
  R3D_REQUIRE(B > 0 && N > 0 && ld >= 192 && ld % 4 == 0 && ldo >= 64,
              "r3d_attention_fwd: bad shape B=%d N=%d ld=%ld ldo=%ld", B, N, ld, ldo);
  R3D_REQUIRE(((uintptr_t)qkv & 15) == 0, "r3d_attention_fwd: qkv must be 16-byte aligned");
  R3D_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "r3d_attention_fwd: dropout probability %f out of range", p_drop);
  const int split = ws ? attention_split(B, N, attention_slots(ATT_FWD)) : 1;
  const int ntiles = r3d_cdiv(N, 32);
  const int tps = r3d_cdiv(ntiles, split);
  const int nz = r3d_cdiv(ntiles, tps);  // no empty split
  dim3 grid(r3d_cdiv(N, 128), B, nz);
  hipLaunchKernelGGL(r3d_attention_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, ld, N, out,
                     ldo, lse_out, p_drop, seed, seed_dev, tps, nz > 1 ? ws : nullptr);
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
  return attention_launch(qkv, ld, B, N, out, ldo, lse_out, 0.f, 0u, nullptr, ws, stream);
}

// training forward: dropout p_drop on the attention weights with the stateless mask of attn_keep.
// Effective seed = seed + *seed_dev (seed_dev may be NULL): a captured hipGraph bumps the device word per replay.
extern "C" int r3d_attention_fwd_train(const float* qkv, long ld, int B, int N, float* out, long ldo, float* lse_out,
                                       float p_drop, unsigned seed, const unsigned* seed_dev, float* ws, void* stream) {
  R3D_REQUIRE(lse_out, "r3d_attention_fwd_train: lse_out is required (saved for the backward pass)");
  return attention_launch(qkv, ld, B, N, out, ldo, lse_out, p_drop, seed, seed_dev, ws, stream);
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
    const unsigned* __restrict__ seed_dev, int tiles_per_split, float* __restrict__ part /* [split][M][128] or NULL */) {
  if (seed_dev) seed += *seed_dev;
  __shared__ float Qs[2][32 * AT_LD];
  __shared__ float Gs[2][32 * AT_LD];  // dO tile
  __shared__ float Ls[2][32], Ds[2][32];
  const unsigned thresh = p_drop > 0.f ? (unsigned)(p_drop * 4294967296.0) : 0u;
  const float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int b = blockIdx.y;
  const long base = (long)b * N;
  const int key = blockIdx.x * 128 + 32 * w + j;  // this lane's key column
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
  const int t_beg = blockIdx.z * tiles_per_split;
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
      if (thresh) keep = attn_keep(seed, (unsigned)(base + q), (unsigned)key, thresh) ? keep_scale : 0.f;
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
  float* drow = part ? part + ((long)blockIdx.z * gridDim.y * N + base + key) * 128 - 64 : dqkv + (base + key) * ldd;
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
    const unsigned* __restrict__ seed_dev, float q_scale, int tiles_per_split,
    float* __restrict__ part /* [split][M][64] unscaled, or NULL */) {
  if (seed_dev) seed += *seed_dev;
  __shared__ float Ks[2][32 * AT_LD];
  __shared__ float Vs[2][32 * AT_LD];
  const unsigned thresh = p_drop > 0.f ? (unsigned)(p_drop * 4294967296.0) : 0u;
  const float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int b = blockIdx.y;
  const long base = (long)b * N;
  const int q_row = blockIdx.x * 128 + 32 * w + j;
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
  const int t_beg = blockIdx.z * tiles_per_split;
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
      if (thresh) keep = attn_keep(seed, (unsigned)(base + q_row), (unsigned)key, thresh) ? keep_scale : 0.f;
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
  float* drow = part ? part + ((long)blockIdx.z * gridDim.y * N + base + q_row) * 64 : dqkv + (base + q_row) * ldd;
  const float osc = part ? 1.f : q_scale;  // partials stay unscaled; r3d_attention_sum_kernel applies q_scale
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = r3d_acc_row(r, lane);
    drow[c] = dq0[r] * osc;  // gradient w.r.t. the UNscaled q map output (q' = q * q_scale)
    drow[32 + c] = dq1[r] * osc;
  }
}

// dqkv (B*N, ldd >= 192): gradients of the q | k | v GEMM outputs (before the 1/sqrt(d) scale of q).
// O: forward output (B*N, ldo); lse: saved log-sum-exp; ws: r3d_attention_ws_words(B, N) floats (row dots + the
// partial dK | dV / dQ of the streamed-axis split).
extern "C" int r3d_attention_bwd(const float* qkv, long ld, int B, int N, const float* O, long ldo, const float* dO,
                                 long lddo, const float* lse, float p_drop, unsigned seed, const unsigned* seed_dev,
                                 float q_scale, float* dqkv, long ldd, float* ws, void* stream) {
  R3D_REQUIRE(qkv && O && dO && lse && dqkv && ws, "r3d_attention_bwd: null pointer");
  R3D_REQUIRE(B > 0 && N > 0 && ld >= 192 && ld % 4 == 0 && lddo % 4 == 0 && ldd >= 192 && ldo >= 64,
              "r3d_attention_bwd: bad shape");
  R3D_REQUIRE((((uintptr_t)qkv | (uintptr_t)dO) & 15) == 0, "r3d_attention_bwd: qkv and dO must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const long M = (long)B * N;
  hipLaunchKernelGGL(r3d_attention_rowdot_kernel, dim3(r3d_cdiv(M, 4)), dim3(256), 0, st, dO, lddo, O, ldo, M, ws);
  const int ntiles = r3d_cdiv(N, 32);
  {
    const int tps = r3d_cdiv(ntiles, attention_split(B, N, attention_slots(ATT_BWD_KV)));
    const int nz = r3d_cdiv(ntiles, tps);
    float* part = nz > 1 ? ws + M : nullptr;
    hipLaunchKernelGGL(r3d_attention_bwd_kv_kernel, dim3(r3d_cdiv(N, 128), B, nz), dim3(256), 0, st, qkv, ld, N, dO, lddo, lse, ws,
                       dqkv, ldd, p_drop, seed, seed_dev, tps, part);
    if (part)
      hipLaunchKernelGGL(r3d_attention_sum_kernel, dim3(r3d_cdiv(M * 128, 256)), dim3(256), 0, st, part, nz, M, 128, 1.f, dqkv,
                         ldd, 64);
  }
  {
    const int tps = r3d_cdiv(ntiles, attention_split(B, N, attention_slots(ATT_BWD_Q)));
    const int nz = r3d_cdiv(ntiles, tps);
    float* part = nz > 1 ? ws + M : nullptr;
    hipLaunchKernelGGL(r3d_attention_bwd_q_kernel, dim3(r3d_cdiv(N, 128), B, nz), dim3(256), 0, st, qkv, ld, N, dO, lddo, lse, ws,
                       dqkv, ldd, p_drop, seed, seed_dev, q_scale, tps, part);
    if (part)
      hipLaunchKernelGGL(r3d_attention_sum_kernel, dim3(r3d_cdiv(M * 64, 256)), dim3(256), 0, st, part, nz, M, 64, q_scale, dqkv,
                         ldd, 0);
  }
  R3D_LAUNCH_CHECK("r3d_attention_bwd");
  return R3D_OK;
}


// workgroups (256 threads) of each attention kernel the chip holds at once; the defaults stand in when there is no
// device to ask (host-only sizing calls)
static int attention_slots(int which) {
  static int cache[3] = {0, 0, 0};
  if (cache[which]) return cache[which];
  const int fallback[3] = {512, 256, 512};
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  hipError_t e = hipErrorUnknown;
  if (which == ATT_FWD) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, r3d_attention_fwd_kernel, 256, 0);
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
