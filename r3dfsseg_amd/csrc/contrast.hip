// Per-way supervised contrastive loss of the training path (reference models/mpti.py:226-313) and the
// training-only debug metrics (mpti.py:515-568), gfx950.  Everything here is tiny (<= 10 shots,
// <= 4 prototypes per shot, <= 48 vectors per way): one workgroup per shot / per way, latency bound.
//
//   per shot: foreground points -> FPS(k=4, start 0) -> sorted unique seeds -> nearest-seed assignment
//             -> cluster means (getMutiplePrototypes, mpti.py:597-634; same fmaf-chain distances as
//             head_proto.hip so indices match oracle/r3d_oracle.c bit for bit)
//   per way:  vectors of its shots (+ 2 shots of the next way labelled -1 when the support set is clean)
//             -> proj (Linear 192->128) -> L2 normalise -> SupCon (temperature 0.1), and its gradient
#include "common.h"

#define CT_K 4
#define CT_MAXV 48     // vectors per way: (k_shot + 2) * 4, k_shot <= 10
#define CT_PD 128      // projection width
#define CT_DMAX 256
#define CT_NMAX 4096

// Batches of episodes (round 3): blockIdx.y = episode; its arrays sit a stride further on (CtEp, elements of each
// array; feature strides in ROWS).  The parameter gradients of the projection are summed over the batch in a fixed
// order (episode, then way); everything else stays per episode.
struct CtEp {
  long feat, sy, flag, ws, loss;           // feature rows, support_y words, support_flag words, scratch floats, loss floats
  long pred, qy, z, desc, pws, assign, gsy, out;  // train metrics: pred / labels / Z rows / descriptor / proto scratch / assign / gt masks / out4
};

// ---------------------------------------------------------------------------
// A. per-shot prototypes
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void r3d_contrast_protos_kernel(
    const float* __restrict__ feat /* (S*N, ldf) */, long ldf, int D, const int* __restrict__ support_y, int N,
    float* __restrict__ protos /* (S, 4, 256) */, int* __restrict__ cnt /* (S, 4) */, int* __restrict__ m_out /* (S) */,
    int* __restrict__ assign_out /* (S, N): cluster of the point or -1 */, CtEp st) {
  {
    const long ep = blockIdx.y;
    feat += ep * st.feat * ldf; support_y += ep * st.sy; protos += ep * st.ws; cnt += ep * st.ws; m_out += ep * st.ws;
    assign_out += ep * st.ws;
  }
  __shared__ int fg[CT_NMAX];            // compacted foreground point ids
  __shared__ float seedf[CT_K][CT_DMAX];
  __shared__ float red_v[4];
  __shared__ int red_p[4];
  __shared__ int wave_tot[4];
  __shared__ int nfg_s, seeds_s[CT_K], m_s;
  __shared__ float csum[CT_K][CT_DMAX];
  __shared__ signed char asg[CT_NMAX];   // cluster of list position pos (block-local copy: no global round trip)
  __shared__ int cnt_s[CT_K];
  const int shot = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int* my = support_y + (long)shot * N;
  const float* fb = feat + (long)shot * N * ldf;
  // --- stable compaction of the foreground points
  if (tid == 0) nfg_s = 0;
  __syncthreads();
  for (int p0 = 0; p0 < N; p0 += 256) {
    const int p = p0 + tid;
    const bool f = p < N && my[p] == 1;
    const unsigned long long mk = __ballot(f);
    if (lane == 0) wave_tot[w] = __popcll(mk);
    __syncthreads();
    int base = nfg_s;
    for (int q = 0; q < w; ++q) base += wave_tot[q];
    if (f) fg[base + __popcll(mk & ((1ull << lane) - 1ull))] = p;
    __syncthreads();
    if (tid == 0) nfg_s += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
  }
  const int nfg = nfg_s;
  for (int p = tid; p < N; p += 256) assign_out[(long)shot * N + p] = -1;
  if (nfg == 0) {
    if (tid == 0) m_out[shot] = 0;
    return;
  }
  int m;
  if (nfg <= CT_K) {  // identity (mpti.py:631-634)
    m = nfg;
    if (tid < CT_K) seeds_s[tid] = tid;
  } else {
    // --- FPS: positions pos = tid + 256*i (i < 16) owned by this thread, min-distances in registers
    float mind[CT_NMAX / 256];
#pragma unroll
    for (int i = 0; i < CT_NMAX / 256; ++i) mind[i] = INFINITY;
    int sel[CT_K];
    sel[0] = 0;
    for (int round = 1; round < CT_K; ++round) {
      const int sp = fg[sel[round - 1]];
      for (int c = tid; c < D; c += 256) seedf[0][c] = fb[(long)sp * ldf + c];
      __syncthreads();
      float bv = -INFINITY;
      int bp = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < CT_NMAX / 256; ++i) {
        const int pos = tid + 256 * i;
        if (pos < nfg) {
          const float* xr = fb + (long)fg[pos] * ldf;
          float acc = 0.f;
          int c = 0;
          for (; c + 8 <= D; c += 8) {  // 8 channel loads in flight, chain stays channel-ascending
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = xr[c + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const float df = v[u] - seedf[0][c + u]; acc = __builtin_fmaf(df, df, acc); }
          }
          for (; c < D; ++c) { const float df = xr[c] - seedf[0][c]; acc = __builtin_fmaf(df, df, acc); }
          mind[i] = acc < mind[i] ? acc : mind[i];
          if (mind[i] > bv || (mind[i] == bv && pos < bp)) { bv = mind[i]; bp = pos; }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(bv, o);
        const int p2 = __shfl_xor(bp, o);
        if (v2 > bv || (v2 == bv && p2 < bp)) { bv = v2; bp = p2; }
      }
      if (lane == 0) { red_v[w] = bv; red_p[w] = bp; }
      __syncthreads();
      for (int q = 0; q < 4; ++q)
        if (red_v[q] > bv || (red_v[q] == bv && red_p[q] < bp)) { bv = red_v[q]; bp = red_p[q]; }
      sel[round] = bp;
      __syncthreads();
    }
    if (tid == 0) {  // sorted unique (torch .unique(), mpti.py:613)
      int s[CT_K];
      for (int i = 0; i < CT_K; ++i) s[i] = sel[i];
      for (int i = 1; i < CT_K; ++i) { int v = s[i], j = i - 1; while (j >= 0 && s[j] > v) { s[j + 1] = s[j]; --j; } s[j + 1] = v; }
      int mm = 0;
      for (int i = 0; i < CT_K; ++i) if (i == 0 || s[i] != s[i - 1]) seeds_s[mm++] = s[i];
      m_s = mm;
    }
    __syncthreads();
    m = m_s;
  }
  __syncthreads();
  // --- seed features
  for (int e = tid; e < m * D; e += 256) {
    const int s = e / D, c = e - s * D;
    seedf[s][c] = fb[(long)fg[seeds_s[s]] * ldf + c];
    csum[s][c] = 0.f;
  }
  __syncthreads();
  // --- nearest-seed assignment (mpti.py:618-622), kept in LDS by overwriting nothing: stored to global
  for (int pos = tid; pos < nfg; pos += 256) {
    const float* xr = fb + (long)fg[pos] * ldf;
    int best = 0;
    if (nfg > CT_K) {
      float bestd = INFINITY;
      float acc[CT_K] = {0.f, 0.f, 0.f, 0.f};  // one pass over the point's channels feeds all (<= 4) seeds
      int c = 0;
      for (; c + 8 <= D; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xr[c + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
          for (int s = 0; s < CT_K; ++s) { const float df = (v[u] - seedf[s][c + u]) + 1e-6f; acc[s] = __builtin_fmaf(df, df, acc[s]); }
        }
      }
      for (; c < D; ++c) {
        const float v = xr[c];
#pragma unroll
        for (int s = 0; s < CT_K; ++s) { const float df = (v - seedf[s][c]) + 1e-6f; acc[s] = __builtin_fmaf(df, df, acc[s]); }
      }
#pragma unroll
      for (int s = 0; s < CT_K; ++s) {
        if (s < m) {
          const float d = sqrtf(acc[s]);
          if (d < bestd) { bestd = d; best = s; }
        }
      }
    } else {
      best = pos;
    }
    assign_out[(long)shot * N + fg[pos]] = best;
    asg[pos] = (signed char)best;
  }
  __syncthreads();
  // --- cluster sums: thread = channel, points in list order, 8 row loads in flight
  if (tid < D) {
    float acc[CT_K] = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = 0; p0 < nfg; p0 += 8) {
      float xv[8];
      int av[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int pos = min(p0 + u, nfg - 1);
        const int gp = fg[pos];
        xv[u] = fb[(long)gp * ldf + tid];
        av[u] = (p0 + u < nfg) ? (int)asg[pos] : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int s = 0; s < CT_K; ++s) acc[s] += (av[u] == s) ? xv[u] : 0.f;
      }
    }
    for (int s = 0; s < CT_K; ++s) csum[s][tid] = acc[s];
  }
  if (tid < CT_K) {
    int c = 0;
    for (int pos = 0; pos < nfg; ++pos) c += ((int)asg[pos] == tid) ? 1 : 0;
    cnt_s[tid] = tid < m ? c : 0;
    cnt[shot * CT_K + tid] = cnt_s[tid];
  }
  __syncthreads();
  for (int e = tid; e < CT_K * CT_DMAX; e += 256) {
    const int s = e / CT_DMAX, c = e - s * CT_DMAX;
    float v = 0.f;
    if (s < m && c < D) v = (nfg > CT_K) ? csum[s][c] / (float)cnt_s[s] : csum[s][c];
    protos[((long)shot * CT_K + s) * CT_DMAX + c] = v;
  }
  if (tid == 0) m_out[shot] = m;
}

// ---------------------------------------------------------------------------
// B. per-way loss and gradients (unscaled: d loss_way / d .)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void r3d_contrast_loss_kernel(
    const float* __restrict__ protos, const int* __restrict__ m_arr, const int* __restrict__ support_flag, int n_way,
    int k_shot, int D, const float* __restrict__ W /* (128, D) */, const float* __restrict__ bias, float temp,
    float* __restrict__ loss_way /* (n_way) */, float* __restrict__ dW_way /* (n_way,128,256) */,
    float* __restrict__ db_way /* (n_way,128) */, float* __restrict__ dp_way /* (n_way, CT_MAXV, 256) */,
    int* __restrict__ vec_src /* (n_way, CT_MAXV): shot*4 + proto, or -1 */, CtEp st) {
  {
    const long ep = blockIdx.y;
    protos += ep * st.ws; m_arr += ep * st.ws; support_flag += ep * st.flag; loss_way += ep * st.ws; dW_way += ep * st.ws;
    db_way += ep * st.ws; dp_way += ep * st.ws; vec_src += ep * st.ws;
  }
  extern __shared__ __attribute__((aligned(16))) float ct_smem[];
  float (*P)[CT_DMAX + 1] = reinterpret_cast<float (*)[CT_DMAX + 1]>(ct_smem);
  float (*Y)[CT_PD + 1] = reinterpret_cast<float (*)[CT_PD + 1]>(ct_smem + CT_MAXV * (CT_DMAX + 1));            // y, then f
  float (*DF)[CT_PD + 1] = reinterpret_cast<float (*)[CT_PD + 1]>(ct_smem + CT_MAXV * (CT_DMAX + 1 + CT_PD + 1));  // df, then dy
  float (*Zm)[CT_MAXV + 1] =
      reinterpret_cast<float (*)[CT_MAXV + 1]>(ct_smem + CT_MAXV * (CT_DMAX + 1 + 2 * (CT_PD + 1)));
  __shared__ float lab[CT_MAXV], nrm[CT_MAXV], lv[CT_MAXV];
  __shared__ int src[CT_MAXV];
  __shared__ int K_s;
  const int way = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    const int ele = support_flag[0];
    int sum0 = 0;
    for (int k = 0; k < k_shot; ++k) sum0 += support_flag[k];
    const bool clean = ele * k_shot == sum0;  // mpti.py:238-244
    int K = 0;
    for (int k = 0; k < k_shot; ++k) {
      const int shot = way * k_shot + k;
      for (int s = 0; s < m_arr[shot] && K < CT_MAXV; ++s) { src[K] = shot * CT_K + s; lab[K] = (float)support_flag[shot]; ++K; }
    }
    if (clean) {
      const int other = way < n_way - 1 ? way + 1 : 0;
      for (int k = 0; k < 2 && k < k_shot; ++k) {
        const int shot = other * k_shot + k;
        for (int s = 0; s < m_arr[shot] && K < CT_MAXV; ++s) { src[K] = shot * CT_K + s; lab[K] = -1.f; ++K; }
      }
    }
    K_s = K;
  }
  __syncthreads();
  const int K = K_s;
  for (int e = tid; e < CT_MAXV; e += 256) vec_src[way * CT_MAXV + e] = e < K ? src[e] : -1;
  for (int e = tid; e < K * D; e += 256) {
    const int v = e / D, c = e - v * D;
    P[v][c] = protos[(long)src[v] * CT_DMAX + c];
  }
  __syncthreads();
  // y = W p + b
  for (int e = tid; e < K * CT_PD; e += 256) {
    const int v = e / CT_PD, o = e - v * CT_PD;
    float acc = bias[o];
    for (int c = 0; c < D; ++c) acc += W[(long)o * D + c] * P[v][c];
    Y[v][o] = acc;
  }
  __syncthreads();
  if (tid < K) {
    float s = 0.f;
    for (int o = 0; o < CT_PD; ++o) s += Y[tid][o] * Y[tid][o];
    nrm[tid] = fmaxf(sqrtf(s), 1e-12f);
  }
  __syncthreads();
  for (int e = tid; e < K * CT_PD; e += 256) { const int v = e / CT_PD, o = e - v * CT_PD; Y[v][o] = Y[v][o] / nrm[v]; }
  __syncthreads();
  // logits
  for (int e = tid; e < K * K; e += 256) {
    const int v = e / K, t = e - v * K;
    float d = 0.f;
    for (int o = 0; o < CT_PD; ++o) d += Y[v][o] * Y[t][o];
    Zm[v][t] = d / temp;
  }
  __syncthreads();
  // per-vector loss, then dz in place: dz_vt = (1/K) (q_vt - [t in P_v]/|P_v|), t != v
  if (tid < K) {
    const int v = tid;
    float se = 0.f, npos = 0.f, spos = 0.f;
    for (int t = 0; t < K; ++t) {
      if (t == v) continue;
      se += expf(Zm[v][t]);
      if (lab[t] == lab[v]) { npos += 1.f; spos += Zm[v][t]; }
    }
    const float lse = logf(se);
    lv[v] = -(spos - npos * lse) / npos;  // -(sum_pos (z - log sum exp)) / |P_v|
    for (int t = 0; t < K; ++t) {
      if (t == v) { Zm[v][t] = 0.f; continue; }
      const float q = expf(Zm[v][t]) / se;
      Zm[v][t] = (q - ((lab[t] == lab[v]) ? 1.f / npos : 0.f)) / (float)K;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int v = 0; v < K; ++v) s += lv[v];
    loss_way[way] = s / (float)K;
  }
  // df_v = (1/temp) sum_t (dz_vt + dz_tv) f_t
  for (int e = tid; e < K * CT_PD; e += 256) {
    const int v = e / CT_PD, o = e - v * CT_PD;
    float acc = 0.f;
    for (int t = 0; t < K; ++t) acc += (Zm[v][t] + Zm[t][v]) * Y[t][o];
    DF[v][o] = acc / temp;
  }
  __syncthreads();
  // dy = (df - f <f, df>) / ||y||
  if (tid < K) {
    float s = 0.f;
    for (int o = 0; o < CT_PD; ++o) s += Y[tid][o] * DF[tid][o];
    lv[tid] = s;
  }
  __syncthreads();
  for (int e = tid; e < K * CT_PD; e += 256) {
    const int v = e / CT_PD, o = e - v * CT_PD;
    DF[v][o] = (DF[v][o] - Y[v][o] * lv[v]) / nrm[v];
  }
  __syncthreads();
  // dW = dy^T P, db = sum dy, dp = W^T dy
  for (int e = tid; e < CT_PD * D; e += 256) {
    const int o = e / D, c = e - o * D;
    float acc = 0.f;
    for (int v = 0; v < K; ++v) acc += DF[v][o] * P[v][c];
    dW_way[((long)way * CT_PD + o) * CT_DMAX + c] = acc;
  }
  for (int o = tid; o < CT_PD; o += 256) {
    float acc = 0.f;
    for (int v = 0; v < K; ++v) acc += DF[v][o];
    db_way[way * CT_PD + o] = acc;
  }
  for (int e = tid; e < K * D; e += 256) {
    const int v = e / D, c = e - v * D;
    float acc = 0.f;
    for (int o = 0; o < CT_PD; ++o) acc += W[(long)o * D + c] * DF[v][o];
    dp_way[((long)way * CT_MAXV + v) * CT_DMAX + c] = acc;
  }
}

// loss = mean over ways (mpti.py:311)
__global__ void r3d_contrast_mean_kernel(const float* __restrict__ loss_way, int n_way, float* __restrict__ loss, CtEp st) {
  loss_way += (long)blockIdx.x * st.ws; loss += (long)blockIdx.x * st.loss;
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < n_way; ++w) s += loss_way[w];
    *loss = s / (float)n_way;
  }
}

// ---------------------------------------------------------------------------
// C. backward: parameter gradients (sum over ways) and point gradients (through the cluster means)
// ---------------------------------------------------------------------------
__global__ void r3d_contrast_param_grad_kernel(const float* __restrict__ dW_way, const float* __restrict__ db_way, int n_way,
                                               int D, const float* __restrict__ gscale, float* __restrict__ dW,
                                               float* __restrict__ db, int n_ep, CtEp st) {
  // the projection's gradient of the whole batch: episodes in order, ways in order (one episode: the sum over its ways)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float sc = gscale[0] / (float)n_way;
  if (i < CT_PD * D) {
    const int o = i / D, c = i - o * D;
    float tot = 0.f;
    for (int e = 0; e < n_ep; ++e) {
      float s = 0.f;
      for (int w = 0; w < n_way; ++w) s += dW_way[(long)e * st.ws + ((long)w * CT_PD + o) * CT_DMAX + c];
      tot = e == 0 ? s * sc : tot + s * sc;
    }
    dW[i] = tot;
  }
  if (i < CT_PD) {
    float tot = 0.f;
    for (int e = 0; e < n_ep; ++e) {
      float s = 0.f;
      for (int w = 0; w < n_way; ++w) s += db_way[(long)e * st.ws + w * CT_PD + i];
      tot = e == 0 ? s * sc : tot + s * sc;
    }
    db[i] = tot;
  }
}

__global__ __launch_bounds__(256) void r3d_contrast_point_grad_kernel(
    const float* __restrict__ dp_way, const int* __restrict__ vec_src, const int* __restrict__ cnt,
    const int* __restrict__ assign, int n_way, int N, int D, const float* __restrict__ gscale,
    float* __restrict__ dfeat /* (S*N, ldd), zero-initialised */, long ldd, CtEp st) {
  {
    const long ep = blockIdx.y;
    dp_way += ep * st.ws; vec_src += ep * st.ws; cnt += ep * st.ws; assign += ep * st.ws; dfeat += ep * st.feat * ldd;
  }
  __shared__ float dproto[CT_K][CT_DMAX];
  const int shot = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // total gradient of this shot's prototypes: every use (own way, negative of another way), fixed order
  for (int e = tid; e < CT_K * CT_DMAX; e += 256) {
    const int s = e / CT_DMAX, c = e - s * CT_DMAX;
    float acc = 0.f;
    for (int wy = 0; wy < n_way; ++wy)
      for (int v = 0; v < CT_MAXV; ++v)
        if (vec_src[wy * CT_MAXV + v] == shot * CT_K + s) acc += dp_way[((long)wy * CT_MAXV + v) * CT_DMAX + c];
    const int cn = cnt[shot * CT_K + s];
    dproto[s][c] = cn > 0 ? acc * (gscale[0] / (float)n_way) / (float)cn : 0.f;
  }
  __syncthreads();
  for (int p = w; p < N; p += 4) {  // one wave per point row
    const int a = assign[(long)shot * N + p];
    if (a < 0) continue;
    float* dr = dfeat + ((long)shot * N + p) * ldd;
    for (int c = lane; c < D; c += 64) dr[c] = dproto[a][c];
  }
}

// ---------------------------------------------------------------------------
// D. training-only debug metrics (mpti.py:515-568): out[0] query_acc_LP, [1] query_acc_original,
//    [2] clean_ratio_LP_avg, [3] clean_ratio_original_avg
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void r3d_train_metrics_kernel(
    const int* __restrict__ pred, const long long* __restrict__ query_y, const long long* __restrict__ gt_query_y, int n_qpts,
    const float4* __restrict__ Z, const float4* __restrict__ Z2 /* classes 4..7 (n_way > 3), or null */,
    const int* __restrict__ desc, const int* __restrict__ comp, const int* __restrict__ assign,
    const int* __restrict__ gt_support_y, int n_way, int k_shot, int N, float* __restrict__ out, CtEp st) {
  {
    const long ep = blockIdx.x;
    pred += ep * st.pred; query_y += ep * st.qy; gt_query_y += ep * st.qy; Z += ep * st.z; desc += ep * st.desc;
    if (Z2) Z2 += ep * st.z;
    comp += ep * st.pws; assign += ep * st.assign; gt_support_y += ep * st.gsy; out += ep * st.out;
  }
  __shared__ int acc[4];
  const int tid = threadIdx.x;
  float lp_sum = 0.f, orig_sum = 0.f;
  if (tid < 4) acc[tid] = 0;
  __syncthreads();
  int a = 0, b = 0;
  for (int i = tid; i < n_qpts; i += 256) {
    a += ((long long)pred[i] == gt_query_y[i]) ? 1 : 0;
    b += (query_y[i] == gt_query_y[i]) ? 1 : 0;
  }
  atomicAdd(&acc[0], a);
  atomicAdd(&acc[1], b);
  __syncthreads();
  const long SN = (long)n_way * k_shot * N;
  for (int wy = 0; wy < n_way; ++wy) {
    const int seg = 1 + wy;
    const int count = desc[seg];                 // HD_SEG_COUNT
    const int poff = desc[16 + seg];             // HD_SEG_POFF
    const long off = SN + (long)wy * k_shot * N;  // SegGeom::off(seg)
    if (tid == 0) { acc[2] = 0; acc[3] = 0; }
    __syncthreads();
    int c_lp = 0, c_or = 0;
    for (int pos = tid; pos < count; pos += 256) {
      const int gp = comp[off + pos];
      const int zr = poff + assign[off + pos];
      const float4 z = Z[zr];
      const float4 z2 = Z2 ? Z2[zr] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float zv[8] = {z.x, z.y, z.z, z.w, z2.x, z2.y, z2.z, z2.w};
      int am = 0;
      for (int c = 1; c <= n_way; ++c) if (zv[c] > zv[am]) am = c;
      const int point_pred = (am == wy + 1) ? 1 : 0;
      const int gt = gt_support_y[gp];
      c_lp += (point_pred == gt) ? 1 : 0;
      c_or += (1 == gt) ? 1 : 0;
    }
    atomicAdd(&acc[2], c_lp);
    atomicAdd(&acc[3], c_or);
    __syncthreads();
    if (tid == 0) { lp_sum += (float)acc[2] / (float)count; orig_sum += (float)acc[3] / (float)count; }
    __syncthreads();
  }
  if (tid == 0) {
    out[0] = (float)acc[0] / (float)n_qpts;
    out[1] = (float)acc[1] / (float)n_qpts;
    out[2] = lp_sum / (float)n_way;
    out[3] = orig_sum / (float)n_way;
  }
}

// ===========================================================================
// C ABI
// ===========================================================================
// ws layout (floats): protos S*4*256 | loss_way 8 | dW_way n_way*128*256 | db_way n_way*128 | dp_way n_way*48*256 |
//                     ints: cnt S*4 | m S | vec_src n_way*48 | assign S*N
extern "C" long r3d_contrast_ws_words(int n_way, int k_shot, int N) {
  const long S = (long)n_way * k_shot;
  return S * CT_K * CT_DMAX + 8 + (long)n_way * CT_PD * CT_DMAX + n_way * CT_PD + (long)n_way * CT_MAXV * CT_DMAX +
         S * CT_K + S + n_way * CT_MAXV + S * N + 64;
}

struct CtWs { float *protos, *loss_way, *dW_way, *db_way, *dp_way; int *cnt, *m, *vec_src, *assign; };
static CtWs ct_carve(float* ws, int n_way, int k_shot, int N) {
  const long S = (long)n_way * k_shot;
  CtWs c;
  float* p = ws;
  c.protos = p; p += S * CT_K * CT_DMAX;
  c.loss_way = p; p += 8;
  c.dW_way = p; p += (long)n_way * CT_PD * CT_DMAX;
  c.db_way = p; p += n_way * CT_PD;
  c.dp_way = p; p += (long)n_way * CT_MAXV * CT_DMAX;
  int* q = (int*)p;
  c.cnt = q; q += S * CT_K;
  c.m = q; q += S;
  c.vec_src = q; q += n_way * CT_MAXV;
  c.assign = q;
  return c;
}

// loss_out: device float per episode.  ws keeps everything the backward needs.
static int contrast_fwd_impl(int n_ep, const CtEp& ep, const float* feat, long ldf, int D, const int32_t* support_y,
                             const int32_t* support_flag, int n_way, int k_shot, int N, const float* W, const float* bias,
                             float temp, float* loss_out, float* ws, long ws_words, void* stream) {
  R3D_REQUIRE(feat && support_y && support_flag && W && bias && loss_out && ws, "r3d_contrast_fwd: null pointer");
  R3D_REQUIRE(ws_words >= r3d_contrast_ws_words(n_way, k_shot, N), "r3d_contrast_fwd: workspace of %ld words is shorter than "
              "r3d_contrast_ws_words(%d, %d, %d)", ws_words, n_way, k_shot, N);
  R3D_REQUIRE(n_way >= 1 && n_way <= 7 && (k_shot + 2) * CT_K <= CT_MAXV && D <= CT_DMAX && N <= CT_NMAX,
              "r3d_contrast_fwd: unsupported shape n_way=%d k_shot=%d D=%d N=%d", n_way, k_shot, D, N);
  R3D_REQUIRE(n_ep >= 1 && n_ep <= 65535 && (n_ep == 1 || ep.ws >= ws_words), "r3d_contrast_fwd: %d episodes, scratch stride %ld",
              n_ep, ep.ws);
  const CtWs c = ct_carve(ws, n_way, k_shot, N);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(r3d_contrast_protos_kernel, dim3(n_way * k_shot, n_ep), dim3(256), 0, st, feat, ldf, D, support_y, N, c.protos,
                     c.cnt, c.m, c.assign, ep);
  const size_t lds = sizeof(float) * CT_MAXV * (CT_DMAX + 1 + 2 * (CT_PD + 1) + CT_MAXV + 1);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)r3d_contrast_loss_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    R3D_REQUIRE(e == hipSuccess, "r3d_contrast_fwd: cannot reserve %zu B of LDS", lds);
    attr = true;
  }
  hipLaunchKernelGGL(r3d_contrast_loss_kernel, dim3(n_way, n_ep), dim3(256), lds, st, c.protos, c.m, support_flag, n_way, k_shot, D,
                     W, bias, temp, c.loss_way, c.dW_way, c.db_way, c.dp_way, c.vec_src, ep);
  hipLaunchKernelGGL(r3d_contrast_mean_kernel, dim3(n_ep), dim3(64), 0, st, c.loss_way, n_way, loss_out, ep);
  R3D_LAUNCH_CHECK("r3d_contrast_fwd");
  return R3D_OK;
}
extern "C" int r3d_contrast_fwd(const float* feat, long ldf, int D, const int32_t* support_y, const int32_t* support_flag,
                                int n_way, int k_shot, int N, const float* W, const float* bias, float temp,
                                float* loss_out, float* ws, long ws_words, void* stream) {
  const CtEp one{};
  return contrast_fwd_impl(1, one, feat, ldf, D, support_y, support_flag, n_way, k_shot, N, W, bias, temp, loss_out, ws, ws_words,
                           stream);
}
// n_ep episodes: episode e reads feature rows from feat + e * feat_ep_rows * ldf, masks / flags e of (n_ep, S, N) / (n_ep, S)
// arrays, scratch ws + e * ws_stride floats, and writes loss_out[e]
extern "C" int r3d_contrast_fwd_batched(int n_ep, const float* feat, long ldf, long feat_ep_rows, int D, const int32_t* support_y,
                                        const int32_t* support_flag, int n_way, int k_shot, int N, const float* W,
                                        const float* bias, float temp, float* loss_out, float* ws, long ws_words, long ws_stride,
                                        void* stream) {
  CtEp ep{};
  ep.feat = feat_ep_rows; ep.sy = (long)n_way * k_shot * N; ep.flag = (long)n_way * k_shot; ep.ws = ws_stride; ep.loss = 1;
  return contrast_fwd_impl(n_ep, ep, feat, ldf, D, support_y, support_flag, n_way, k_shot, N, W, bias, temp, loss_out, ws, ws_words,
                           stream);
}

// dfeat (S*N, ldd) zero-initialised by the caller; dW (128, D), db (128) summed over the batch; gscale: device float (dL/dloss)
static int contrast_bwd_impl(int n_ep, const CtEp& ep, int D, int n_way, int k_shot, int N, const float* gscale_dev, float* dfeat,
                             long ldd, float* dW, float* db, float* ws, void* stream) {
  R3D_REQUIRE(gscale_dev && dfeat && dW && db && ws && n_ep >= 1 && n_ep <= 65535, "r3d_contrast_bwd: bad arguments");
  const CtWs c = ct_carve(ws, n_way, k_shot, N);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(r3d_contrast_param_grad_kernel, dim3(r3d_cdiv(CT_PD * D, 256)), dim3(256), 0, st, c.dW_way, c.db_way, n_way,
                     D, gscale_dev, dW, db, n_ep, ep);
  hipLaunchKernelGGL(r3d_contrast_point_grad_kernel, dim3(n_way * k_shot, n_ep), dim3(256), 0, st, c.dp_way, c.vec_src, c.cnt,
                     c.assign, n_way, N, D, gscale_dev, dfeat, ldd, ep);
  R3D_LAUNCH_CHECK("r3d_contrast_bwd");
  return R3D_OK;
}
extern "C" int r3d_contrast_bwd(int D, int n_way, int k_shot, int N, const float* gscale_dev, float* dfeat, long ldd, float* dW,
                                float* db, float* ws, void* stream) {
  const CtEp one{};
  return contrast_bwd_impl(1, one, D, n_way, k_shot, N, gscale_dev, dfeat, ldd, dW, db, ws, stream);
}
extern "C" int r3d_contrast_bwd_batched(int n_ep, int D, int n_way, int k_shot, int N, const float* gscale_dev, float* dfeat,
                                        long ldd, long dfeat_ep_rows, float* dW, float* db, float* ws, long ws_stride,
                                        void* stream) {
  CtEp ep{};
  ep.feat = dfeat_ep_rows; ep.ws = ws_stride;
  return contrast_bwd_impl(n_ep, ep, D, n_way, k_shot, N, gscale_dev, dfeat, ldd, dW, db, ws, stream);
}

// out4 per episode (n_ep, 4).  Strides: pred / labels n_query_pts per episode, Z rows z_ep_rows, desc / proto scratch /
// assign words, gt masks S*N per episode.
extern "C" int r3d_train_metrics_batched(int n_ep, const int32_t* pred, const int64_t* query_y, const int64_t* gt_query_y,
                                         int n_query_pts, const float* Z, long z_ep_rows, const int32_t* desc, long desc_stride,
                                         const int32_t* proto_ws, long pws_stride, const int32_t* assign, long assign_stride,
                                         const int32_t* gt_support_y, int n_way, int k_shot, int N, float* out4, void* stream) {
  R3D_REQUIRE(pred && query_y && gt_query_y && Z && desc && proto_ws && assign && gt_support_y && out4 && n_ep >= 1,
              "r3d_train_metrics: null pointer");
  CtEp ep{};
  ep.pred = n_query_pts; ep.qy = n_query_pts; ep.z = z_ep_rows; ep.desc = desc_stride; ep.pws = pws_stride;
  ep.assign = assign_stride; ep.gsy = (long)n_way * k_shot * N; ep.out = 4;
  R3D_REQUIRE(n_way >= 1 && n_way <= 7 && (n_way <= 3 || z_ep_rows > 0), "r3d_train_metrics: n_way = %d (more than 3 ways: Z as two "
              "planes (2, n_ep * z_ep_rows, 4), batched form)", n_way);
  const float4* Z2 = n_way > 3 ? (const float4*)Z + (long)n_ep * z_ep_rows : nullptr;
  hipLaunchKernelGGL(r3d_train_metrics_kernel, dim3(n_ep), dim3(256), 0, (hipStream_t)stream, pred, (const long long*)query_y,
                     (const long long*)gt_query_y, n_query_pts, (const float4*)Z, Z2, desc, proto_ws, assign, gt_support_y, n_way,
                     k_shot, N, out4, ep);
  R3D_LAUNCH_CHECK("r3d_train_metrics");
  return R3D_OK;
}
extern "C" int r3d_train_metrics(const int32_t* pred, const int64_t* query_y, const int64_t* gt_query_y, int n_query_pts,
                                 const float* Z, const int32_t* desc, const int32_t* proto_ws /* comp at offset 0 */,
                                 const int32_t* assign, const int32_t* gt_support_y, int n_way, int k_shot, int N,
                                 float* out4, void* stream) {
  return r3d_train_metrics_batched(1, pred, query_y, gt_query_y, n_query_pts, Z, 0, desc, 0, proto_ws, 0, assign, 0, gt_support_y,
                                   n_way, k_shot, N, out4, stream);
}
