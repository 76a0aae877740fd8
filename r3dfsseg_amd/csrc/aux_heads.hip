// Small device-side heads around the main path, gfx950:
//   * clean-shot detection of eval_noise.py's eval=True path   (reference models/mpti.py:87-223, 316-371)
//   * ProtoNet head: masked average pooling + cosine / euclidean similarity (models/protonet.py:295-349)
//   * mIoU accumulator                                          (eval_noise.py:23-72)
// All are HBM / latency bound reductions over (S*N, D) point-major features.
#include "common.h"

#define AH_MAXSHOT 32   // n_way * k_shot
#define AH_MAXK 8       // shots per way in clean-shot detection (<= 4 seeds each)
#define AH_BOXES 5      // box 0: scale (1,1,1); boxes 1..4: scale (2,2,1), x outer / y inner
#define AH_DMAX 256

// ---------------------------------------------------------------------------
// clean-shot detection, step 1: per (shot, box) feature sums over the foreground points inside
// the box.  Box bounds replicate grid_sampling (mpti.py:316-371) in fp32, inclusive on both sides.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void r3d_clean_boxsum_kernel(
    const float* __restrict__ feat /* (S*N, ldf) */, long ldf, int D, const float* __restrict__ support_x /* (S, Cin, N) */,
    int Cin, const int* __restrict__ support_y /* (S, N) */, int N, float* __restrict__ box_sum /* (S, 5, 256) */,
    int* __restrict__ box_cnt /* (S, 5) */, long feat_ep_rows, long ws_stride) {
  {
    const long ep = blockIdx.z, S_ = gridDim.x;  // batch of episodes: rows / scratch of episode ep
    feat += ep * feat_ep_rows * ldf; support_x += ep * S_ * Cin * N; support_y += ep * S_ * N;
    box_sum += ep * ws_stride; box_cnt += ep * ws_stride;
  }
  __shared__ float red[4][6];
  __shared__ float bb[6];  // x_min, x_max, y_min, y_max, z_min, z_max
  __shared__ float psum[4][AH_DMAX];
  __shared__ int pcnt[4];
  const int shot = blockIdx.x, box = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int* my = support_y + (long)shot * N;
  const float* px = support_x + (long)shot * Cin * N;
  const float* py = px + N;
  const float* pz = px + 2 * N;
  // bounding box of the foreground points
  float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
  for (int p = tid; p < N; p += 256) {
    if (my[p] == 1) {
      const float x = px[p], y = py[p], z = pz[p];
      xmin = fminf(xmin, x); xmax = fmaxf(xmax, x);
      ymin = fminf(ymin, y); ymax = fmaxf(ymax, y);
      zmin = fminf(zmin, z); zmax = fmaxf(zmax, z);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    xmin = fminf(xmin, __shfl_xor(xmin, o)); xmax = fmaxf(xmax, __shfl_xor(xmax, o));
    ymin = fminf(ymin, __shfl_xor(ymin, o)); ymax = fmaxf(ymax, __shfl_xor(ymax, o));
    zmin = fminf(zmin, __shfl_xor(zmin, o)); zmax = fmaxf(zmax, __shfl_xor(zmax, o));
  }
  if (lane == 0) { red[w][0] = xmin; red[w][1] = xmax; red[w][2] = ymin; red[w][3] = ymax; red[w][4] = zmin; red[w][5] = zmax; }
  __syncthreads();
  if (tid < 6) {
    const float a = red[0][tid], b = red[1][tid], c = red[2][tid], d = red[3][tid];
    bb[tid] = (tid & 1) ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : fminf(fminf(a, b), fminf(c, d));
  }
  __syncthreads();
  // n_z = 1 at both scales: [z_min + 0*d_z, z_min + 0*d_z + d_z] (may exclude z_max by one rounding, as the reference)
  const float dz = (bb[5] - bb[4]) / 1.f;
  const float zlo = bb[4] + 0.f * dz, zhi = zlo + dz;
  float xlo, xhi, ylo, yhi;
  if (box == 0) {  // n_x = n_y = 1: d = (max - min) / 1 ; [min + 0*d, min + 0*d + d]
    const float dx = (bb[1] - bb[0]) / 1.f, dy = (bb[3] - bb[2]) / 1.f;
    xlo = bb[0] + 0.f * dx; xhi = xlo + dx;
    ylo = bb[2] + 0.f * dy; yhi = ylo + dy;
  } else {
    const int ix = (box - 1) >> 1, iy = (box - 1) & 1;
    const float dx = (bb[1] - bb[0]) / 2.f, dy = (bb[3] - bb[2]) / 2.f;
    xlo = bb[0] + (float)ix * dx; xhi = xlo + dx;
    ylo = bb[2] + (float)iy * dy; yhi = ylo + dy;
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int cnt = 0;
  for (int base = 64 * w; base < N; base += 256) {
    const int p = base + lane;
    bool f = false;
    if (p < N && my[p] == 1) {
      const float x = px[p], y = py[p], z = pz[p];
      f = x >= xlo && x <= xhi && y >= ylo && y <= yhi && z >= zlo && z <= zhi;
    }
    unsigned long long mm = __ballot(f);
    cnt += __popcll(mm);
    while (mm) {
      const int src = __ffsll((long long)mm) - 1;
      mm &= mm - 1;
      const float* fr = feat + ((long)shot * N + base + src) * ldf;
      if (lane < D) a0 += fr[lane];
      if (lane + 64 < D) a1 += fr[lane + 64];
      if (lane + 128 < D) a2 += fr[lane + 128];
      if (lane + 192 < D) a3 += fr[lane + 192];
    }
  }
  psum[w][lane] = a0; psum[w][lane + 64] = a1; psum[w][lane + 128] = a2; psum[w][lane + 192] = a3;
  if (lane == 0) pcnt[w] = cnt;
  __syncthreads();
  box_sum[((long)shot * AH_BOXES + box) * AH_DMAX + tid] = ((psum[0][tid] + psum[1][tid]) + psum[2][tid]) + psum[3][tid];
  if (tid == 0) box_cnt[shot * AH_BOXES + box] = pcnt[0] + pcnt[1] + pcnt[2] + pcnt[3];
}

// step 2: per way: seeds (non-empty boxes, shot order, x outer / y inner) -> L2 normalise -> cosine
// map (zero diagonal; cubed at scale (1,1,1)) -> row sums > mean -> per-shot majority -> average of
// the two scales < 0.5 drops the shot; a way that loses every foreground point keeps all its shots.
__global__ __launch_bounds__(256) void r3d_clean_decide_kernel(const float* __restrict__ box_sum,
                                                               const int* __restrict__ box_cnt, int n_way, int k_shot,
                                                               int D, int* __restrict__ shot_keep /* (n_way*k_shot) */,
                                                               float* __restrict__ dbg_cos_sum /* opt (n_way, 2, 4*k_shot) */,
                                                               long ws_stride) {
  {
    const long ep = blockIdx.y;
    box_sum += ep * ws_stride; box_cnt += ep * ws_stride; shot_keep += ep * n_way * k_shot;
    if (dbg_cos_sum) dbg_cos_sum += ep * n_way * 2 * 4 * k_shot;
  }
  __shared__ float seed[4 * AH_MAXK][AH_DMAX + 1];  // <= 4 seeds per shot
  __shared__ float rowsum[4 * AH_MAXK];
  __shared__ int seed_shot[4 * AH_MAXK];
  __shared__ float flag[2][AH_MAXK];
  __shared__ int nseed_s;
  const int way = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int scale = 0; scale < 2; ++scale) {
    __syncthreads();
    if (tid == 0) {
      int ns = 0;
      for (int k = 0; k < k_shot; ++k) {
        const int shot = way * k_shot + k;
        const int b0 = scale == 0 ? 0 : 1, b1 = scale == 0 ? 1 : 5;
        for (int b = b0; b < b1; ++b)
          if (box_cnt[shot * AH_BOXES + b] > 0) { seed_shot[ns] = (k << 8) | b; ++ns; }
      }
      nseed_s = ns;
    }
    __syncthreads();
    const int ns = nseed_s;
    // seed means, then F.normalize (x / max(||x||, 1e-12))
    for (int s = w; s < ns; s += 4) {
      const int k = seed_shot[s] >> 8, b = seed_shot[s] & 255;
      const int shot = way * k_shot + k;
      const float cnt = (float)box_cnt[shot * AH_BOXES + b];
      float sq = 0.f;
      for (int c = lane; c < D; c += 64) {
        const float v = box_sum[((long)shot * AH_BOXES + b) * AH_DMAX + c] / cnt;
        seed[s][c] = v;
        sq += v * v;
      }
      sq = r3d_wave_sum(sq);
      const float nrm = fmaxf(sqrtf(sq), 1e-12f);
      for (int c = lane; c < D; c += 64) seed[s][c] = seed[s][c] / nrm;
    }
    __syncthreads();
    // row sums of the (masked, optionally cubed) cosine map
    for (int i = w; i < ns; i += 4) {
      float rs = 0.f;
      for (int j = 0; j < ns; ++j) {
        float d = 0.f;
        for (int c = lane; c < D; c += 64) d += seed[i][c] * seed[j][c];
        d = r3d_wave_sum(d);
        if (j == i) d = 0.f;
        if (scale == 0) d = d * d * d;
        rs += d;
      }
      if (lane == 0) {
        rowsum[i] = rs;
        if (dbg_cos_sum) dbg_cos_sum[(way * 2 + scale) * 4 * k_shot + i] = rs;
      }
    }
    __syncthreads();
    if (tid == 0) {
      float mean = 0.f;
      for (int i = 0; i < ns; ++i) mean += rowsum[i];
      mean /= (float)ns;
      for (int k = 0; k < k_shot; ++k) {
        int tot = 0, pos = 0;
        for (int i = 0; i < ns; ++i)
          if ((seed_shot[i] >> 8) == k) { ++tot; pos += rowsum[i] > mean ? 1 : 0; }
        flag[scale][k] = (tot > 0 && ((float)pos / (float)tot) > 0.5f) ? 1.f : 0.f;
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    bool any = false;
    for (int k = 0; k < k_shot; ++k) {
      const float total = (flag[0][k] + flag[1][k]) / 2.f;
      const int keep = total < 0.5f ? 0 : 1;
      shot_keep[way * k_shot + k] = keep;
      any = any || keep;
    }
    if (!any)  // every foreground point would be dropped: reset to all ones (mpti.py:216-219)
      for (int k = 0; k < k_shot; ++k) shot_keep[way * k_shot + k] = 1;
  }
}

// ---------------------------------------------------------------------------
// ProtoNet head (models/protonet.py:295-349)
// ---------------------------------------------------------------------------
// masked sums of one support cloud: fg (mask == 1) and bg (mask == 0)
__global__ __launch_bounds__(256) void r3d_proto_pool_kernel(const float* __restrict__ feat, long ldf, int D,
                                                             const int* __restrict__ support_y, int N,
                                                             float* __restrict__ pooled /* (S, 2, 256): fg, bg */) {
  const int shot = blockIdx.x, tid = threadIdx.x;
  const int* my = support_y + (long)shot * N;
  float fg = 0.f, bg = 0.f;
  int nfg = 0;
  if (tid < D) {
    for (int p = 0; p < N; ++p) {  // rows are read coalesced across the D threads
      const float v = feat[((long)shot * N + p) * ldf + tid];
      const int m = my[p];
      fg += v * (float)m;
      bg += v * (float)(m == 0);
      nfg += m;
    }
    // getMaskedFeatures: sum(feat * mask) / (mask.sum() + 1e-5)
    pooled[((long)shot * 2 + 0) * AH_DMAX + tid] = fg / ((float)nfg + 1e-5f);
    pooled[((long)shot * 2 + 1) * AH_DMAX + tid] = bg / ((float)(N - nfg) + 1e-5f);
  }
}

// prototypes (getPrototype) + per-point similarity (calculateSimilarity) -> Z rows (n_q*N, 4)
__global__ __launch_bounds__(256) void r3d_proto_sim_kernel(const float* __restrict__ pooled, int n_way, int k_shot,
                                                            const float* __restrict__ qfeat, long ldq, int D, int n_pts,
                                                            int method /*0 cosine, 1 euclidean*/, float scaler,
                                                            float4* __restrict__ Zq, float4* __restrict__ Zq2 /* classes 4..7 */) {
  __shared__ float proto[8][AH_DMAX];
  __shared__ float pnorm[8];
  const int tid = threadIdx.x;
  const int n_classes = n_way + 1;
  if (tid < D) {
    float bgp = 0.f;
    for (int s = 0; s < n_way * k_shot; ++s) bgp += pooled[((long)s * 2 + 1) * AH_DMAX + tid];
    proto[0][tid] = bgp / (float)(n_way * k_shot);
    for (int wy = 0; wy < n_way; ++wy) {
      float f = 0.f;
      for (int k = 0; k < k_shot; ++k) f += pooled[((long)(wy * k_shot + k) * 2 + 0) * AH_DMAX + tid];
      proto[wy + 1][tid] = f / (float)k_shot;
    }
  }
  __syncthreads();
  if (tid < D)
    for (int k = n_classes; k < 8; ++k) proto[k][tid] = 0.f;
  if (tid < 8) {
    float s = 0.f;
    if (tid < n_classes) for (int c = 0; c < D; ++c) s += proto[tid][c] * proto[tid][c];
    pnorm[tid] = sqrtf(s);
  }
  __syncthreads();
  const int lane = tid & 63, w = tid >> 6;
  for (int p = blockIdx.x * 4 + w; p < n_pts; p += gridDim.x * 4) {  // one wave per query point
    const float* q = qfeat + (long)p * ldq;
    float dot[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, qq = 0.f, dd[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = lane; c < D; c += 64) {
      const float v = q[c];
      qq += v * v;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        dot[k] += v * proto[k][c];
        const float df = (v - proto[k][c]) + 1e-6f;  // pairwise_distance eps (torch 1.8 semantics)
        dd[k] += df * df;
      }
    }
    qq = r3d_wave_sum(qq);
    float out[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float d = r3d_wave_sum(dot[k]);
      const float e = r3d_wave_sum(dd[k]);
      if (method == 0) out[k] = d / fmaxf(sqrtf(qq) * pnorm[k], 1e-8f) * scaler;
      else { const float dist = sqrtf(e); out[k] = -(dist * dist); }
      if (k >= n_classes) out[k] = 0.f;
    }
    if (lane == 0) Zq[p] = make_float4(out[0], out[1], out[2], out[3]);
    if (lane == 0 && Zq2) Zq2[p] = make_float4(out[4], out[5], out[6], out[7]);
  }
}

// ---------------------------------------------------------------------------
// mIoU accumulator (eval_noise.py:23-72): hist[0]=GT, hist[1]=predicted, hist[2]=true positive,
// per test class index (0 = background).  lut[l] = index of episode label l in test_classes (+1).
// ---------------------------------------------------------------------------
__global__ void r3d_miou_accumulate_kernel(const int* __restrict__ pred, const long long* __restrict__ gt, long n,
                                           const int* __restrict__ lut, int n_lut, int n_classes,
                                           unsigned long long* __restrict__ hist) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = (int)gt[i], p = pred[i];
  const int gi = lut[min(max(g, 0), n_lut - 1)], pi = lut[min(max(p, 0), n_lut - 1)];
  atomicAdd(&hist[gi], 1ull);
  atomicAdd(&hist[n_classes + pi], 1ull);
  if (g == p) atomicAdd(&hist[2 * n_classes + gi], 1ull);
}

// ===========================================================================
// C ABI
// ===========================================================================
// ws: 5*256*S floats + 5*S ints.  shot_keep (n_way*k_shot) int32 out.  dbg optional (n_way,2,4*k_shot).
extern "C" long r3d_clean_ws_words(int n_way, int k_shot) { return (long)n_way * k_shot * (AH_BOXES * AH_DMAX + AH_BOXES) + 64; }

// n_ep episodes at once: episode e's support features start feat_ep_rows rows after episode e - 1's, its coordinates /
// masks are entry e of (n_ep, S, Cin, N) / (n_ep, S, N) arrays, its scratch ws + e * ws_stride, its flags shot_keep + e * S
extern "C" int r3d_clean_shot_detect_batched(int n_ep, const float* feat, long ldf, long feat_ep_rows, int D,
                                             const float* support_x, int Cin, const int32_t* support_y, int n_way, int k_shot,
                                             int N, int32_t* shot_keep, float* dbg_cos_sum, int32_t* ws, long ws_stride,
                                             void* stream) {
  R3D_REQUIRE(feat && support_x && support_y && shot_keep && ws, "r3d_clean_shot_detect: null pointer");
  R3D_REQUIRE(n_way >= 1 && n_way * k_shot <= AH_MAXSHOT && k_shot <= AH_MAXK && D >= 1 && D <= AH_DMAX && Cin >= 3 && N >= 1,
              "r3d_clean_shot_detect: unsupported shape n_way=%d k_shot=%d D=%d", n_way, k_shot, D);
  R3D_REQUIRE(n_ep >= 1 && n_ep <= 65535 && (n_ep == 1 || ws_stride >= r3d_clean_ws_words(n_way, k_shot)),
              "r3d_clean_shot_detect: %d episodes, scratch stride %ld", n_ep, ws_stride);
  const int S = n_way * k_shot;
  float* box_sum = (float*)ws;
  int* box_cnt = ws + (long)S * AH_BOXES * AH_DMAX;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(r3d_clean_boxsum_kernel, dim3(S, AH_BOXES, n_ep), dim3(256), 0, st, feat, ldf, D, support_x, Cin,
                     support_y, N, box_sum, box_cnt, feat_ep_rows, ws_stride);
  hipLaunchKernelGGL(r3d_clean_decide_kernel, dim3(n_way, n_ep), dim3(256), 0, st, box_sum, box_cnt, n_way, k_shot, D,
                     shot_keep, dbg_cos_sum, ws_stride);
  R3D_LAUNCH_CHECK("r3d_clean_shot_detect");
  return R3D_OK;
}
extern "C" int r3d_clean_shot_detect(const float* feat, long ldf, int D, const float* support_x, int Cin,
                                     const int32_t* support_y, int n_way, int k_shot, int N, int32_t* shot_keep,
                                     float* dbg_cos_sum, int32_t* ws, void* stream) {
  return r3d_clean_shot_detect_batched(1, feat, ldf, 0, D, support_x, Cin, support_y, n_way, k_shot, N, shot_keep, dbg_cos_sum, ws,
                                       0, stream);
}

// Z (n_q*N, 4) fp32 similarity rows (feed r3d_query_logits_ce with n_proto = 0); more than 3 ways: two planes
// (2, n_q*N, 4), classes 4..7 in plane 1 (r3d_query_logits_ce_batched with z_ep_rows = n_q*N).  ws: S*2*256 floats.
extern "C" int r3d_protonet_head(const float* sfeat, long ldf, const float* qfeat, long ldq, int D,
                                 const int32_t* support_y, int n_way, int k_shot, int N, int n_query_pts, int method,
                                 float scaler, float* Z, float* ws, void* stream) {
  R3D_REQUIRE(sfeat && qfeat && support_y && Z && ws, "r3d_protonet_head: null pointer");
  R3D_REQUIRE(n_way >= 1 && n_way <= 7 && D >= 1 && D <= AH_DMAX, "r3d_protonet_head: unsupported shape");
  if (method != 0 && method != 1) {
    // the reference raises NotImplementedError for anything but cosine / euclidean (protonet.py:347)
    r3d_set_error("Error! Distance computation method (%d) is unknown!", method);
    return R3D_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(r3d_proto_pool_kernel, dim3(n_way * k_shot), dim3(256), 0, st, sfeat, ldf, D, support_y, N, ws);
  hipLaunchKernelGGL(r3d_proto_sim_kernel, dim3(256), dim3(256), 0, st, ws, n_way, k_shot, qfeat, ldq, D, n_query_pts,
                     method, scaler, (float4*)Z, n_way > 3 ? (float4*)Z + n_query_pts : nullptr);
  R3D_LAUNCH_CHECK("r3d_protonet_head");
  return R3D_OK;
}

extern "C" int r3d_miou_accumulate(const int32_t* pred, const int64_t* gt, long n, const int32_t* lut, int n_lut,
                                   int n_classes, uint64_t* hist, void* stream) {
  R3D_REQUIRE(pred && gt && lut && hist && n > 0 && n_lut > 0 && n_classes > 0, "r3d_miou_accumulate: bad arguments");
  hipLaunchKernelGGL(r3d_miou_accumulate_kernel, dim3(r3d_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pred,
                     (const long long*)gt, n, lut, n_lut, n_classes, (unsigned long long*)hist);
  R3D_LAUNCH_CHECK("r3d_miou_accumulate");
  return R3D_OK;
}
