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

__global__ __launch_bounds__(256) void r3d_attention_fwd_kernel(
    const float* __restrict__ qkv, long ld, int N, float* __restrict__ out, long ldo,
    float* __restrict__ lse_out) {
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

  const int ntiles = (N + 31) / 32;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
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

// qkv: (B*N, ld) with q (pre-scaled by 1/sqrt(64)) | k | v at columns 0 | 64 | 128;
// out: (B*N, ldo) point-major, 64 columns written; lse_out optional (B*N) log-sum-exp
// per query (saved for the backward pass).
extern "C" int r3d_attention_fwd(const float* qkv, long ld, int B, int N, float* out, long ldo,
                                 float* lse_out, void* stream) {
  R3D_REQUIRE(qkv && out, "r3d_attention_fwd: null pointer");
  R3D_REQUIRE(B > 0 && N > 0 && ld >= 192 && ld % 4 == 0 && ldo >= 64,
              "r3d_attention_fwd: bad shape B=%d N=%d ld=%ld ldo=%ld", B, N, ld, ldo);
  R3D_REQUIRE(((uintptr_t)qkv & 15) == 0, "r3d_attention_fwd: qkv must be 16-byte aligned");
  dim3 grid(r3d_cdiv(N, 128), B);
  hipLaunchKernelGGL(r3d_attention_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, ld, N, out,
                     ldo, lse_out);
  R3D_LAUNCH_CHECK("r3d_attention_fwd");
  return R3D_OK;
}
