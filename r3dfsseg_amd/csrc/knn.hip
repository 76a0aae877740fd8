// Fused k-nearest-neighbour search for gfx950: all-pairs score on the fp32 matrix
// core + wave-ballot top-k, never materialising the N x N matrix.
//
// Replaces (reference, /root/reference):
//   models/dgcnn.py:17-23   knn(x, k)               -> mode R3D_SCORE_DGCNN
//   models/mpti.py:731-736  faiss IndexFlatL2.search -> mode R3D_SCORE_L2
//
// Bit-exactness contract (oracle/r3d_oracle.c): the inner product is a
// channel-ascending fp32 fmaf chain.  v_mfma_f32_32x32x2_f32 is bitwise that chain
// (verified on hardware by tools/probe/mfma_probe.hip), so the scores, and with the
// tie rule "lower index first" the indices, equal the oracle's bit for bit.
//
// Layout: points are rows of a point-major matrix (row stride ldx floats); batch b
// owns rows [b*N, b*N+N).  One workgroup = 32 query rows of one batch; it walks the
// candidates in chunks of 128 (one 32x32 MFMA tile per wave), channels in slabs of
// 64 staged through LDS, writes the 32x128 score tile to LDS and lets each wave keep
// the sorted top-(64*R) list of 8 rows in registers (list entry t lives in register
// t/64 of lane t%64; insertion = ballot + popcount + lane shift).
#include <stdlib.h>
#include "common.h"

#define KNN_Q 32
#define KNN_CH 128
#define KNN_SLAB 64
#define KNN_ROWS_PER_WAVE 8

enum { R3D_SCORE_DGCNN = 0, R3D_SCORE_L2 = 1 };

// ---- squared norms: channel-ascending fmaf chain (== diagonal of the MFMA product)
__global__ void r3d_sqnorm_kernel(const float* __restrict__ x, long ldx, int rows, int C,
                                  float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const float* p = x + (long)i * ldx;
  float acc = 0.f;
  for (int c = 0; c < C; ++c) acc = __builtin_fmaf(p[c], p[c], acc);
  out[i] = acc;
}

template <int R>
struct TopList {
  float v[R];
  int id[R];
};

template <int R>
static __device__ __forceinline__ void list_init(TopList<R>& L) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    L.v[r] = -INFINITY;
    L.id[r] = -1;
  }
}

// entry at list position pos (wave-uniform)
template <int R>
static __device__ __forceinline__ float list_value_at(const TopList<R>& L, int pos) {
  float out = -INFINITY;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float t = r3d_readlane_f(L.v[r], pos & 63);
    if ((pos >> 6) == r) out = t;
  }
  return out;
}

// Insert (val, id) keeping the list sorted by descending value; among equal values
// earlier insertions (lower candidate index) stay ahead.  val/id are wave-uniform.
template <int R>
static __device__ __forceinline__ void list_insert(TopList<R>& L, float val, int id, int lane) {
  int p = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) p += __popcll(__ballot(L.v[r] >= val));
  float prev_last_v = 0.f;
  int prev_last_i = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float upv = __shfl_up(L.v[r], 1);
    int upi = __shfl_up(L.id[r], 1);
    float lastv = r3d_readlane_f(L.v[r], 63);
    int lasti = __builtin_amdgcn_readlane(L.id[r], 63);
    if (lane == 0) {
      upv = prev_last_v;
      upi = prev_last_i;
    }
    int t = 64 * r + lane;
    if (t == p) {
      L.v[r] = val;
      L.id[r] = id;
    } else if (t > p) {
      L.v[r] = upv;
      L.id[r] = upi;
    }
    prev_last_v = lastv;
    prev_last_i = lasti;
  }
}

template <int R>
__global__ __launch_bounds__(256) void r3d_knn_topk_kernel(
    const float* __restrict__ x, long ldx, int N, int C, int k, int mode,
    const int* __restrict__ n_dev, int n_dev_stride, const float* __restrict__ nrm, int* __restrict__ idx_out,
    float* __restrict__ score_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * KNN_Q;
  const int n = n_dev ? min(n_dev[(long)b * n_dev_stride], N) : N;  // valid rows of this batch
  if (q0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int Cp = (C + 1) & ~1;
  const int As = Cp + 1;
  const int Bs = KNN_SLAB + 1;
  const int Ds = KNN_CH + 1;
  float* Aq = smem;                  // [KNN_Q][As]
  float* Bc = Aq + KNN_Q * As;       // [KNN_CH][Bs]
  float* Dt = Bc + KNN_CH * Bs;      // [KNN_Q][Ds]
  float* nq = Dt + KNN_Q * Ds;       // [KNN_Q]
  const float* xb = x + (long)b * N * ldx;
  const float* nb = nrm + (long)b * N;

  // stage the 32 query rows (zero beyond n and in the odd-C pad column)
  for (int r = w; r < KNN_Q; r += 4) {
    const int row = q0 + r;
    for (int c = lane; c < Cp; c += 64) {
      const float v = xb[(long)min(row, n - 1) * ldx + min(c, C - 1)];
      Aq[r * As + c] = r3d_keep(v, row < n && c < C);
    }
  }
  if (tid < KNN_Q) nq[tid] = (q0 + tid < n) ? nb[q0 + tid] : 0.f;

  TopList<R> L[KNN_ROWS_PER_WAVE];
  float thr[KNN_ROWS_PER_WAVE];
#pragma unroll
  for (int i = 0; i < KNN_ROWS_PER_WAVE; ++i) {
    list_init<R>(L[i]);
    thr[i] = -INFINITY;
  }

  for (int c0 = 0; c0 < n; c0 += KNN_CH) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s0 = 0; s0 < Cp; s0 += KNN_SLAB) {
      const int sw = min(KNN_SLAB, Cp - s0);
      __syncthreads();  // Bc (and, first time round, Dt of the previous chunk) free
      for (int r = w; r < KNN_CH; r += 4) {
        const int row = c0 + r;
        for (int c = lane; c < sw; c += 64) {
          const float v = xb[(long)min(row, n - 1) * ldx + min(s0 + c, C - 1)];
          Bc[r * Bs + c] = r3d_keep(v, row < n && s0 + c < C);
        }
      }
      __syncthreads();
      const float* ap = Aq + (lane & 31) * As + s0 + (lane >> 5);
      const float* bp = Bc + (32 * w + (lane & 31)) * Bs + (lane >> 5);
      for (int kk = 0; kk < sw; kk += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], acc, 0, 0, 0);
    }
    // scores -> LDS tile.  acc[r]: query row r3d_acc_row(r), candidate column lane&31.
    {
      const int cj = c0 + 32 * w + (lane & 31);
      const bool valid = cj < n;
      const float nj = r3d_keep(nb[min(cj, n - 1)], valid);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r3d_acc_row(r, lane);
        const float ni = nq[row];
        float s;
        if (mode == R3D_SCORE_DGCNN) {
          // dgcnn.py:18-20: pd[i][j] = -xx[j] - (-2 dot) - xx[i]
          const float inner = -2.f * acc[r];
          const float t = (-nj) - inner;
          s = t - ni;
        } else {
          // faiss exhaustive_L2sqr_blas: dis = |x|^2 + |y|^2 - 2<x,y>, clamped at 0
          float dis = (ni + nj) - 2.f * acc[r];
          if (dis < 0.f) dis = 0.f;
          s = -dis;
        }
        Dt[row * Ds + 32 * w + (lane & 31)] = valid ? s : -INFINITY;
      }
    }
    __syncthreads();
    // selection: wave w owns rows 8w .. 8w+7 of the tile
#pragma unroll
    for (int i = 0; i < KNN_ROWS_PER_WAVE; ++i) {
      const int row = KNN_ROWS_PER_WAVE * w + i;
#pragma unroll
      for (int g = 0; g < KNN_CH / 64; ++g) {
        const float v = Dt[row * Ds + 64 * g + lane];
        unsigned long long m = __ballot(v > thr[i]);
        while (m) {
          const int src = __ffsll((long long)m) - 1;
          m &= m - 1;
          const float cv = r3d_readlane_f(v, src);
          if (cv > thr[i]) {
            list_insert<R>(L[i], cv, c0 + 64 * g + src, lane);
            thr[i] = list_value_at<R>(L[i], k - 1);
          }
        }
      }
    }
  }
  // emit: list position t -> column t
#pragma unroll
  for (int i = 0; i < KNN_ROWS_PER_WAVE; ++i) {
    const int row = q0 + KNN_ROWS_PER_WAVE * w + i;
    if (row >= n) continue;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int t = 64 * r + lane;
      if (t < k) {
        idx_out[((long)b * N + row) * k + t] = L[i].id[r];
        if (score_out) score_out[((long)b * N + row) * k + t] = L[i].v[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// k <= 32, C <= 64 (the DGCNN encoder's kNN): exact two-pass selection, streamed.
//
// Operands come from a CHANNEL-MAJOR copy xT (B, C, N): for a fixed channel the 32 lanes of
// a half-wave read 32 consecutive points = one 128-B line, which is exactly the MFMA operand
// layout (lane -> point, lane half -> channel of the k-pair).  So fragments go global -> VGPR
// with no LDS staging and NO barrier in the main loops; every wave streams independently with
// the next sub-tile's 32 loads in flight behind the current MFMA chain.
//
// One workgroup = 32 query rows; its 4 waves split the 32-candidate sub-tiles round-robin.
// Scores never leave the MFMA accumulator: register r of lane half h holds, for query row
// r3d_acc_row(r), the 32 candidates of the sub-tile, one per lane of the half.
//   pass A  every (wave, lane) pair is a GROUP of candidates of a query; keep the group
//           maximum (one v_max per score).  The k-th largest of the 128 group maxima is a
//           lower bound tau of the k-th best score (128 distinct real candidates), and a
//           tight one (expected rank ~ k + 2).
//   pass B  recompute the scores (matrix-core time is cheap, insertion latency is not) and
//           insert only candidates with score >= tau into the wave's sorted list, which
//           lives in lv[r]/li[r] across the SAME 32 lanes: one instruction serves two
//           queries (one per half), an insertion is ballot + popcount + DPP wave_shr:1.
//   merge   the 4 partial lists of a query meet in LDS; rank by counting, emit rank < k.
// Exactness never depends on tau (the lists keep the best 32 whatever passes).
// ---------------------------------------------------------------------------
static __device__ __forceinline__ float dpp_wave_shr1_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
static __device__ __forceinline__ int dpp_wave_shr1_i(int v) {
  return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
}
static __device__ __forceinline__ unsigned f2key(float v) {  // order-preserving float -> uint
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static __device__ __forceinline__ float key2f(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
static __device__ __forceinline__ bool entry_better(float v1, int i1, float v2, int i2) {
  return v1 > v2 || (v1 == v2 && i1 < i2);
}

// squared norms from the channel-major copy (same channel-ascending fmaf chain)
__global__ void r3d_sqnorm_cm_kernel(const float* __restrict__ xT, long ldT, int C, int N, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float* p = xT + (long)b * C * ldT + i;
  float acc = 0.f;
  int c = 0;
  for (; c + 8 <= C; c += 8) {  // 8 channel loads in flight; the chain stays channel-ascending
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(long)(c + u) * ldT];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_fmaf(v[u], v[u], acc);
  }
  for (; c < C; ++c) acc = __builtin_fmaf(p[(long)c * ldT], p[(long)c * ldT], acc);
  out[(long)b * N + i] = acc;
}

#ifdef KNN_STAMPS
__device__ unsigned long long g_knn_dbg[16];
#define KSTAMP(i) do { if (blockIdx.x == 7 && blockIdx.y == (gridDim.y > 3 ? 3 : 0) && blockIdx.z == 0 && threadIdx.x == 0) g_knn_dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define KSTAMP(i)
#endif

template <int KS>
__global__ __launch_bounds__(256) void r3d_knn_small_kernel(
    const float* __restrict__ xT, long ldT, int N, int C, int k, int mode, const int* __restrict__ n_dev, int n_dev_stride,
    const float* __restrict__ nrm, int* __restrict__ idx_out, float* __restrict__ score_out,
    const int* __restrict__ tile_flags /* optional: only 32-row tiles with a non-zero flag are computed */) {
  if (tile_flags && !tile_flags[(long)blockIdx.y * gridDim.x + blockIdx.x]) return;
  __shared__ float smem[32 * 129 > 2 * 32 * 128 ? 32 * 129 : 2 * 32 * 128];
  __shared__ float tau_s[32];
  const int b = blockIdx.y;
  const int n = n_dev ? min(n_dev[(long)b * n_dev_stride], N) : N;
  const int q0 = blockIdx.x * 32;
  if (q0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const float* xb = xT + (long)b * C * ldT;
  const float* nb = nrm + (long)b * N;

  float a[KS];
  {
    // loads are UNCONDITIONAL on clamped addresses and masked afterwards: a load inside a
    // conditional arm makes hipcc branch around every one of them and wait vmcnt(0) each time
    const int qrow = q0 + j;
    const int qc = min(qrow, n - 1);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c = 2 * s + h;
      const float v = xb[(long)min(c, C - 1) * ldT + qc];
      a[s] = r3d_keep(v, qrow < n && c < C);
    }
  }
  float nq[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = q0 + r3d_acc_row(r, lane);
    nq[r] = r3d_keep(nb[min(row, n - 1)], row < n);
  }
  const int nsub = (n + 31) / 32;

  auto bload = [&](int st, float (&bf)[KS], float& nj) {
    const int cand = 32 * st + j;
    const bool ok = cand < n;
    const int cc = min(cand, n - 1);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c = 2 * s + h;
      const float v = xb[(long)min(c, C - 1) * ldT + cc];
      bf[s] = r3d_keep(v, ok && c < C);
    }
    nj = r3d_keep(nb[cc], ok);
  };
  auto scores = [&](int st, const float (&bf)[KS], float nj, f32x16& sc) {
    const bool valid = 32 * st + j < n;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bf[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v;
      if (mode == R3D_SCORE_DGCNN) {
        const float inner = -2.f * acc[r];
        const float tt = (-nj) - inner;
        v = tt - nq[r];
      } else {
        float dis = (nq[r] + nj) - 2.f * acc[r];
        if (dis < 0.f) dis = 0.f;
        v = -dis;
      }
      sc[r] = valid ? v : -INFINITY;
    }
  };

  float bfA[KS], bfB[KS], njA, njB;
  KSTAMP(0);
  // ------------------------------------------------------------------ pass A: group maxima
  float gm[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) gm[r] = -INFINITY;
  {
    int st = w;
    if (st < nsub) bload(st, bfA, njA);
    while (st < nsub) {
      const int st2 = st + 4;
      if (st2 < nsub) bload(st2, bfB, njB);
      {
        f32x16 sc;
        scores(st, bfA, njA, sc);
#pragma unroll
        for (int r = 0; r < 16; ++r) gm[r] = fmaxf(gm[r], sc[r]);
      }
      st = st2;
      if (st >= nsub) break;
      const int st3 = st + 4;
      if (st3 < nsub) bload(st3, bfA, njA);
      {
        f32x16 sc;
        scores(st, bfB, njB, sc);
#pragma unroll
        for (int r = 0; r < 16; ++r) gm[r] = fmaxf(gm[r], sc[r]);
      }
      st = st3;
    }
  }
  KSTAMP(1);
  // tau[q] = k-th largest of the 128 group maxima of query q
  {
    float* gmax = smem;  // [32][129]
#pragma unroll
    for (int r = 0; r < 16; ++r) gmax[r3d_acc_row(r, lane) * 129 + 32 * w + j] = gm[r];
    __syncthreads();
    for (int qq = 0; qq < 8; ++qq) {
      const int q = 8 * w + qq;
      const unsigned k0 = f2key(gmax[q * 129 + lane]);
      const unsigned k1 = f2key(gmax[q * 129 + 64 + lane]);
      unsigned res = 0;
      for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = res | (1u << bit);
        const int cnt = __popcll(__ballot(k0 >= cand)) + __popcll(__ballot(k1 >= cand));
        if (cnt >= k) res = cand;
      }
      if (lane == 0) tau_s[q] = key2f(res);
    }
    __syncthreads();
  }
  float tauq[16], lv[16], thr[16];
  int li[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    tauq[r] = tau_s[r3d_acc_row(r, lane)];
    lv[r] = -INFINITY;
    li[r] = -1;
    thr[r] = -INFINITY;
  }
  const int klo = k - 1, khi = 32 + k - 1;
  KSTAMP(2);

  // ------------------------------------------------------------------ pass B: select
  auto select = [&](int st, const f32x16& sc) {
    const int cbase = 32 * st;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float s_r = sc[r];
      unsigned long long m = __ballot(s_r >= tauq[r] && s_r > thr[r]);
      while (m) {
        unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
        const int slo = lo ? __ffs((int)lo) - 1 : 0;
        const int shi = hi ? __ffs((int)hi) - 1 : 0;
        const float cvlo = lo ? r3d_readlane_f(s_r, slo) : -INFINITY;
        const float cvhi = hi ? r3d_readlane_f(s_r, 32 + shi) : -INFINITY;
        lo &= lo - 1;
        hi &= hi - 1;
        m = ((unsigned long long)hi << 32) | lo;
        const float cv = h ? cvhi : cvlo;
        const int cidx = cbase + (h ? shi : slo);
        const bool act = cv > thr[r];  // uniform within a half; false for the -inf filler
        const unsigned long long ge = __ballot(lv[r] >= cv);
        const int p = h ? __popc((unsigned)(ge >> 32)) : __popc((unsigned)ge);
        const float upv = dpp_wave_shr1_f(lv[r]);
        const int upi = dpp_wave_shr1_i(li[r]);
        if (act) {
          if (j == p) { lv[r] = cv; li[r] = cidx; }
          else if (j > p) { lv[r] = upv; li[r] = upi; }
        }
        const float tlo = r3d_readlane_f(lv[r], klo);
        const float thi = r3d_readlane_f(lv[r], khi);
        thr[r] = h ? thi : tlo;
      }
    }
  };
  {
    int st = w;
    if (st < nsub) bload(st, bfA, njA);
    while (st < nsub) {
      const int st2 = st + 4;
      if (st2 < nsub) bload(st2, bfB, njB);
      {
        f32x16 sc;
        scores(st, bfA, njA, sc);
        select(st, sc);
      }
      st = st2;
      if (st >= nsub) break;
      const int st3 = st + 4;
      if (st3 < nsub) bload(st3, bfA, njA);
      {
        f32x16 sc;
        scores(st, bfB, njB, sc);
        select(st, sc);
      }
      st = st3;
    }
  }
  KSTAMP(3);

  // ------------------------------------------------------------------ merge the 4 partial lists
  __syncthreads();                         // gmax (aliased) fully consumed by every wave
  float* mv = smem;                        // [32 queries][4 waves][32]
  int* mi = (int*)(smem + 32 * 128);       // same shape
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int q = r3d_acc_row(r, lane);
    mv[(q * 4 + w) * 32 + j] = lv[r];
    mi[(q * 4 + w) * 32 + j] = li[r];
  }
  __syncthreads();
  for (int qq = 0; qq < 8; ++qq) {
    const int q = 8 * w + qq;
    const int row = q0 + q;
    if (row >= n) break;
    // lane holds entries `lane` and `64 + lane` of the query's 4 x 32 slots (list = slot / 32)
    const float v0 = mv[q * 128 + lane], v1 = mv[q * 128 + 64 + lane];
    const int i0 = mi[q * 128 + lane], i1 = mi[q * 128 + 64 + lane];
    int rank0 = 0, rank1 = 0;
#pragma unroll
    for (int l4 = 0; l4 < 4; ++l4) {
      const float lvv = (l4 < 2) ? v0 : v1;
      const int lii = (l4 < 2) ? i0 : i1;
      const unsigned long long real = __ballot(lvv != -INFINITY);
      const int cnt = __popc((unsigned)(real >> (32 * (l4 & 1))));  // sorted list: real entries first
      for (int t = 0; t < cnt; ++t) {
        const float ov = r3d_readlane_f(lvv, 32 * (l4 & 1) + t);
        const int oi = __builtin_amdgcn_readlane(lii, 32 * (l4 & 1) + t);
        rank0 += entry_better(ov, oi, v0, i0) ? 1 : 0;
        rank1 += entry_better(ov, oi, v1, i1) ? 1 : 0;
      }
    }
    if (v0 != -INFINITY && rank0 < k) {
      idx_out[((long)b * N + row) * k + rank0] = i0;
      if (score_out) score_out[((long)b * N + row) * k + rank0] = v0;
    }
    if (v1 != -INFINITY && rank1 < k) {
      idx_out[((long)b * N + row) * k + rank1] = i1;
      if (score_out) score_out[((long)b * N + row) * k + rank1] = v1;
    }
  }
  KSTAMP(4);
}

// Bitonic sort of 64 R keys of 64 bits across one wave, DESCENDING; element e = 64 i + lane sits in register i of lane
// `lane`.  Stages with a partner distance below 64 exchange across lanes (knn_lane_xor, twice per register), the others
// between registers of the same lane; every loop is unrolled, the directions are compile-time or lane constants.
// Used to rank the 201-NN's survivors: sorting 256 keys takes 36 stages of ~6 instructions per register, where counting
// every key against every other one takes M x (2 v_readlane + 2 R instructions) -- 3.2 k instructions for 230 survivors.
// v of lane (lane ^ j), j a power of two below 64, inside the VALU: DPP quad permutes (1, 2), row shifts under bank masks
// (4), a row rotation (8), and gfx950's row / half swaps (16, 32).  No LDS crossbar: a ds_bpermute is a ~200-cycle round
// trip, and a sorting network is a chain of them (36 dependent stages: 20 k cycles per 256 keys with __shfl_xor, measured).
static __device__ __forceinline__ unsigned knn_lane_xor(unsigned v, int j, int lane) {
  const int x = (int)v;
  switch (j) {
    case 1: return (unsigned)__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    case 2: return (unsigned)__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    case 4: {
      const int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);            // row_shl:4 into banks 0, 2 (lane + 4)
      return (unsigned)__builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);         // row_shr:4 into banks 1, 3 (lane - 4)
    }
    case 8: return (unsigned)__builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false);   // row_ror:8
    case 16: {
      const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // {rows 0 0 2 2, rows 1 1 3 3}
      return (lane & 16) ? r[0] : r[1];
    }
    default: {
      const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // {rows 0 1 0 1, rows 2 3 2 3}
      return (lane & 32) ? r[0] : r[1];
    }
  }
}

template <int R>
static __device__ __forceinline__ void knn_bitonic_desc(unsigned (&khi)[R], unsigned (&klo)[R], int lane) {
  static_assert(R == 1 || R == 2 || R == 4 || R == 8, "a power of two");
#pragma unroll
  for (int k = 2; k <= 64 * R; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64) {
        const int dj = j >> 6;
#pragma unroll
        for (int i = 0; i < R; ++i) {
          if (i & dj) continue;
          const bool desc = ((64 * i) & k) == 0;  // (k >= 128 here: the block's direction is a property of the register)
          const unsigned long long a = ((unsigned long long)khi[i] << 32) | klo[i];
          const unsigned long long b = ((unsigned long long)khi[i + dj] << 32) | klo[i + dj];
          const bool swap = desc ? a < b : a > b;
          const unsigned ah = khi[i], al = klo[i];
          khi[i] = swap ? khi[i + dj] : ah; klo[i] = swap ? klo[i + dj] : al;
          khi[i + dj] = swap ? ah : khi[i + dj]; klo[i + dj] = swap ? al : klo[i + dj];
        }
      } else {
        const bool lower = (lane & j) == 0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const bool desc = (((64 * i) | lane) & k) == 0;
          const unsigned oh = knn_lane_xor(khi[i], j, lane), ol = knn_lane_xor(klo[i], j, lane);
          const unsigned long long a = ((unsigned long long)khi[i] << 32) | klo[i];
          const unsigned long long b = ((unsigned long long)oh << 32) | ol;
          const bool take = (lower == desc) ? b > a : b < a;  // this element keeps the larger (smaller) of the pair
          khi[i] = take ? oh : khi[i];
          klo[i] = take ? ol : klo[i];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// 32 < k <= 256 (the head's 201-NN over graph nodes, C = 192): exact two-pass selection,
// streamed like the small kernel, with the survivors of pass B APPENDED to a per-query LDS
// buffer (one LDS atomic per survivor, no ordering work) and ranked once at the end.
//
// One workgroup = 32 query rows x 8 waves; wave w takes sub-tiles w, w+8, ...  A group of
// candidates of a query is (wave, lane) = candidate index mod 256; pass A keeps the TOP-2
// of every group (512 distinct real candidates per query): their k-th largest value is a
// lower bound tau of the k-th best score, expected rank ~ 1.07 k.  Pass B appends every
// candidate with score >= tau; the buffer (KB_CAP entries) is then ranked by counting
// (score desc, index asc) and ranks < k are emitted, i.e. the output is sorted.  If more
// than KB_CAP candidates reach tau (pathological index/score correlation, or > KB_CAP - k
// exact ties at tau) the overflow bit of *status is set and the caller must fall back to the
// insertion kernel above -- never silently wrong.
// Channels run in chunks of 64 (32 MFMA k-pairs): fragments of the next chunk are in flight
// behind the current chunk's MFMA chain; query fragments sit in LDS (conflict-free stride).
// ---------------------------------------------------------------------------
// Two configurations of the same kernel:
//   big <8 waves, 384 slots, top-2 per group>: 32 < k <= 256 (the head's 201-NN); overflow -> *status bit 0.
//   mid <4 waves, 128 slots, top-1 per group>: k <= 32 on 17..256 channels (DGCNN layers 2 and 3): half the
//       threads and a third of the LDS, so two workgroups share a CU and overlap their phases; a 32-row tile whose
//       buffer overflowed (> 128 candidates at or above tau: duplicated points tie exactly) sets tile_flags[tile]
//       and is recomputed by the exact insertion kernel launched right behind (it returns at once elsewhere).
template <int KB_WAVES, int KB_CAP, int KB_TOP, int KCH /* k-pairs per channel chunk: 32 (64 channels) or 8 (16) */,
          int KB_SAMPLE /* pass A visits every KB_SAMPLE-th sub-tile of a wave: tau from a sample is still a lower bound */,
          bool FULLC /* C is a multiple of the chunk width: no channel clamping / masking in the fragment loads */,
          int SMODE /* score form (R3D_SCORE_*), a compile-time constant: the score arithmetic is VALU work that
                       competes with the MFMAs for issue slots, 8 instructions per element with both forms computed */,
          bool BFA /* pass A on the bf16 matrix core (KCH == 32, FULLC, nsplit == 1; xpk = the packed pieces).  Pass A only
                      ESTIMATES the threshold tau: any LOWER bound of every true score will do.  Two bf16 pieces per
                      coordinate (hi + lo = the top 16 bits), three v_mfma_f32_32x32x16_bf16 per 16 channels (hi hi, hi lo,
                      lo hi) = 96 cycles where the fp32 form takes 8 x 64.  The pieces are cut from the points MINUS
                      their set's mean (a distance does not see a common offset, a 16-bit inner product does: with the
                      eval-mode features of a random-weight model, norms 1000 x the neighbour distances, the uncentred
                      bound let every other tile overflow its buffer).  With x', y' the centred points the score is
                      2 x'.y' - |x'|^2 - |y'|^2 in real arithmetic; the dropped piece products are below 3 * 2^-16
                      |x'||y'|, the bf16 form's accumulation below C * 2^-24 |x'||y'|, the exact fp32 form's own roundings
                      (it works on the uncentred points) below (C + 3) 2^-24 (|x|^2 + |y|^2) in the worst case: the
                      bound subtracts 2^-12 (|x'|^2 + |y'|^2) + (C + 8) 2^-24 (|x|^2 + |y|^2).  Pass B -- every score
                      that can be emitted -- stays on the fp32 core. */,
          bool BFB = false /* (needs BFA) pass B on the bf16 matrix core as well, as a FILTER: the same pieces give an UPPER
                      bound of every score (the error terms of the lower bound, with the other sign); a candidate is
                      appended iff its upper bound reaches tau -- every true top-k member is (its exact score is at least the
                      k-th best >= tau) together with the few whose bound straddles tau -- and only the appended ones
                      (~k + 10 per query instead of all N) get the EXACT score: the channel-ascending fp32 fmaf chain from
                      0 over the point-major rows, bit for bit what v_mfma_f32_32x32x2_f32 accumulates (the diagonal
                      of that product is how r3d_sqnorm_kernel computes the norms), then the same three roundings of the
                      score form.  Indices and scores are those of the all-pairs fp32 pass; 12 bf16 MFMAs of 32 cycles
                      per 64 channels and 32 x 32 scores replace 32 fp32 MFMAs of 64. */>
__global__ __launch_bounds__(64 * KB_WAVES) __attribute__((amdgpu_waves_per_eu(KB_WAVES == 4 && (FULLC || KCH < 32) ? 3 : 2)))
void r3d_knn_append_kernel(
    const float* __restrict__ xT, long ldT, int N, int C, int k, int mode, const int* __restrict__ n_dev, int n_dev_stride,
    const float* __restrict__ nrm, int* __restrict__ idx_out, float* __restrict__ score_out,
    int* __restrict__ status, int* __restrict__ tile_flags, int nsplit, int* __restrict__ idx_tmp,
    float* __restrict__ sc_tmp, const unsigned short* __restrict__ xpk /* [B][2 pieces][nch 8 chunks][N][8] bf16, BFA only */,
    const float* __restrict__ cnorm /* [B N] squared norms of the CENTRED points the pieces were cut from, BFA only */,
    const float* __restrict__ xpm /* point-major rows (16-byte aligned, ldx % 4 == 0), BFB only */, long ldx) {
  static_assert(!BFA || (KCH == 32 && FULLC), "the bf16 threshold pass takes whole 64-channel chunks");
  static_assert(!BFB || BFA, "the bf16 filter pass shares the threshold pass's pieces");
  // nsplit > 1 (gridDim.z): the CANDIDATE axis is dealt to nsplit workgroups per query tile (sub-tile s goes to
  // workgroup s % nsplit); each selects its own top-k -- its tau is a lower bound of the k-th best score of ITS
  // candidates, which is <= the k-th best of all of them -- into idx_tmp / sc_tmp, and r3d_knn_merge_kernel merges the
  // sorted lists.  Used while 2 x tiles <= 256 workgroups (one workgroup per CU: 123 KB of LDS).
  constexpr int KB_GROUPS = KB_WAVES * 32;
  const int z = blockIdx.z;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int cnt_s[32];
  __shared__ float tau_s[32];
  // XCD-aware work order (common.h: r3d_xcd_swizzle): the workgroups that share an L2 take the query tiles of the SAME
  // point sets -- every tile of a set streams that set's whole candidate matrix, so one L2 then holds one or two sets
  // instead of every set in flight.  (Measured: 2 % on the k <= 32 configuration; the passes are not bound by it.)
  const int lin = r3d_xcd_swizzle((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int b = lin / (int)gridDim.x, tile_x = lin - b * (int)gridDim.x;
  const int n = n_dev ? min(n_dev[(long)b * n_dev_stride], N) : N;
  const int q0 = tile_x * 32;
  if (q0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int nch = (C + 2 * KCH - 1) / (2 * KCH);  // channel chunks of 2 KCH
  // LDS row stride of the fp32 query rows: odd for the fp32 passes (lane = row reads a column); a multiple of 4 where only
  // the exact scores of the bf16 filter's survivors read them (BFB: 16-byte row segments, 8 distinct rows per wave)
  const int Cs = nch * 2 * KCH + (BFB ? 4 : 1);
  float* Aq = smem;                       // [32][Cs]
  float* region = smem + 32 * Cs;         // pass A: gmax [32][2*KB_GROUPS + 1]; pass B: buffers
  const float* xb = xT + (long)b * C * ldT;
  const float* nb = nrm + (long)b * N;

  for (int e = tid; e < 32 * nch * 2 * KCH; e += 64 * KB_WAVES) {
    const int c = e >> 5, jj = e & 31;
    const float v = xb[(long)min(c, C - 1) * ldT + min(q0 + jj, n - 1)];
    Aq[jj * Cs + c] = r3d_keep(v, c < C && q0 + jj < n);
  }
  if (tid < 32) cnt_s[tid] = 0;
  // bf16 threshold pass: the query rows' pieces sit behind the group maxima (the survivor buffers of pass B, which are
  // larger, take the whole region afterwards), [row][hi nch 64 | lo nch 64] bf16, rows 16 bytes apart in bank phase
  const int Cp = nch * 64;
  const int QRS = 2 * Cp + 8;
  unsigned short* Aqb = reinterpret_cast<unsigned short*>(region + ((32 * (KB_TOP * KB_GROUPS + 1) + 3) & ~3));
  const unsigned short* xpb = BFA ? xpk + (long)b * N * 2 * Cp : nullptr;
  if (BFA) {
    // the packed pieces are CHUNK-major: chunk c8 (8 channels of one piece; hi chunks first, then lo) of all N points lies
    // contiguous, 16 bytes per point -- the MFMA operand of a lane is one such 16-byte unit, and the 32 lanes of a half-wave
    // (32 consecutive candidates, one chunk) read 512 contiguous bytes = 4 cache lines per instruction.  (Point-major
    // pieces, a row of 2 Cp bf16 per point, made every lane touch its own line: 32 line look-ups per instruction, and
    // the bf16 passes ran at the L1's look-up rate -- 5 500 cycles per 32 x 32 sub-tile for 384 cycles of MFMA.)
    const int cpr = 2 * Cp / 8;  // 16-byte chunks of a point
    for (int e = tid; e < 32 * cpr; e += 64 * KB_WAVES) {
      const int c8 = e >> 5, jj = e & 31;
      *reinterpret_cast<r3d_u32x4*>(Aqb + jj * QRS + 8 * c8) =
          *reinterpret_cast<const r3d_u32x4*>(xpb + ((long)c8 * N + min(q0 + jj, n - 1)) * 8);
    }
  }
  KSTAMP(8);
  float nq[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = q0 + r3d_acc_row(r, lane);
    nq[r] = r3d_keep(nb[min(row, n - 1)], row < n);
  }
  // bf16 threshold pass: score >= 2 acc - kq[row] - kj[candidate], k = |x'|^2 (1 + 2^-12) + |x|^2 (C + 8) 2^-24 (see BFA
  // above: the second term is the WORST case of the exact form's C sequential fp32 accumulations, not the typical
  // sqrt(C) -- with 2^-20 in its place one row in a few thousand kept k - 1 survivors)
  const float* cnb = BFA ? cnorm + (long)b * N : nullptr;
  const float slack = (float)(C + 8) * 0x1p-24f;
  auto bound_k = [&](int i) { return cnb[i] * (1.f + 0x1p-12f) + nb[i] * slack; };
  float* kq_s = reinterpret_cast<float*>(Aqb + 32 * QRS);  // [32] behind the query pieces (+ [32] for the upper bound)
  if (BFA && tid < 32) kq_s[tid] = bound_k(min(q0 + tid, n - 1));
  // upper bound of the scores (BFB): score <= 2 acc - ku[row] - ku[candidate], ku = |x'|^2 (1 - 2^-12) - |x|^2 (C + 8) 2^-24
  auto bound_ku = [&](int i) { return cnb[i] * (1.f - 0x1p-12f) - nb[i] * slack; };
  if (BFB && tid < 32) kq_s[32 + tid] = bound_ku(min(q0 + tid, n - 1));
  __syncthreads();

  const int nsub = ((n + 31) / 32 - z + nsplit - 1) / nsplit;  // sub-tiles of this workgroup: z, z + nsplit, ...
  const int my_sub = (nsub - w + KB_WAVES - 1) / KB_WAVES;  // sub-tiles of this wave (w < nsub assumed below)
  const int T = (w < nsub) ? my_sub * nch : 0;              // (sub-tile, chunk) units

  int stride = KB_SAMPLE;  // sub-tile stride of the running pass (pass A: the sample; pass B: 1)
  auto bload = [&](int t, float (&bf)[KCH]) {
    const int st = (w + KB_WAVES * stride * (t / nch)) * nsplit + z, ch = (t % nch);
    const int cand = 32 * st + j;
    const bool ok = cand < n;
    const int cc = min(cand, n - 1);
    if (FULLC) {
      // address arithmetic per load costs as much issue time as the MFMA it feeds: one base pointer, constant
      // strides, no masks (a clamped candidate is a real one and is discarded at score time: `valid`)
      const float* p = xb + (long)(2 * KCH * ch + h) * ldT + cc;
      const long st2 = 2 * ldT;
#pragma unroll
      for (int s = 0; s < KCH; ++s) bf[s] = p[s * st2];
    } else {
#pragma unroll
      for (int s = 0; s < KCH; ++s) {
        const int c = 2 * KCH * ch + 2 * s + h;
        bf[s] = r3d_keep(xb[(long)min(c, C - 1) * ldT + cc], ok && c < C);
      }
    }
  };
  f32x16 acc;
  auto mma = [&](int t, const float (&bf)[KCH]) {
    const int ch = (t % nch);
    if (ch == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
    const float* ap = Aq + j * Cs + 2 * KCH * ch + h;
    // all A fragments of the chunk into registers first: read-two / wait / two MFMAs / read-two ... (what hipcc
    // emits for `mfma(ap[2 * s], ...)`) exposes the LDS latency between every pair of MFMAs
    float av[KCH];
#pragma unroll
    for (int s = 0; s < KCH; ++s) av[s] = ap[2 * s];
    __builtin_amdgcn_sched_barrier(0);  // keep the machine scheduler from sinking the reads back next to their uses
#pragma unroll
    for (int s = 0; s < KCH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bf[s], acc, 0, 0, 0);
  };
  // lower bounds of the scores from the bf16 pass (acc = hi hi + hi lo + lo hi of the inner products)
  // (the bound terms of this lane's 16 query rows sit in registers for the length of a pass: kb[r]; a score bound is
  // 2 acc - (kb[row] + k[candidate]) -- one add, one fma and the clamp of the L2 form per element.  The fma's single
  // rounding against the mul + sub of the exact form is far inside the bounds' (C + 8) 2^-24 slack.)
  float kb[16];
  auto load_kb = [&](int which) {
#pragma unroll
    for (int r = 0; r < 16; ++r) kb[r] = kq_s[32 * which + r3d_acc_row(r, lane)];
  };
  auto scores_lb = [&](int st, f32x16& sc) {
    const int cand = 32 * st + j;
    const float kj = bound_k(min(cand, n - 1));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = __builtin_fmaf(2.f, acc[r], -(kb[r] + kj));  // <= the score: -(an upper bound of the squared distance)
      sc[r] = SMODE == R3D_SCORE_DGCNN ? v : fminf(v, 0.f);
    }
    if (32 * st + 32 > n) {  // uniform: only the last sub-tile has candidates beyond n
      const bool valid = cand < n;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = valid ? sc[r] : -INFINITY;
    }
  };
  auto scores_ub = [&](int st, f32x16& sc) {
    const int cand = 32 * st + j;
    const float kj = bound_ku(min(cand, n - 1));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = __builtin_fmaf(2.f, acc[r], -(kb[r] + kj));  // >= the score the exact form computes
      sc[r] = SMODE == R3D_SCORE_DGCNN ? v : fminf(v, 0.f);
    }
    if (32 * st + 32 > n) {
      const bool valid = cand < n;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = valid ? sc[r] : -INFINITY;
    }
  };
  auto scores = [&](int st, f32x16& sc) {
    const int cand = 32 * st + j;
    const bool valid = cand < n;
    const float nj = r3d_keep(nb[min(cand, n - 1)], valid);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v;
      if (SMODE == R3D_SCORE_DGCNN) {
        const float inner = -2.f * acc[r];
        v = ((-nj) - inner) - nq[r];
      } else {
        const float dis = (nq[r] + nj) - 2.f * acc[r];
        v = -fmaxf(dis, 0.f);
      }
      sc[r] = v;
    }
    if (32 * st + 32 > n) {  // uniform: only the last sub-tile has candidates beyond n
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = valid ? sc[r] : -INFINITY;
    }
  };

  float bfA[KCH], bfB[KCH];
  // ------------------------------------------------------------------ pass A: top-2 per group
  float g1[16], g2[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { g1[r] = -INFINITY; g2[r] = -INFINITY; }
  auto finishA_sub = [&](int i) {  // the wave's i-th sub-tile of pass A is complete in acc
    f32x16 sc;
    if (BFA) scores_lb((w + KB_WAVES * stride * i) * nsplit + z, sc);
    else scores((w + KB_WAVES * stride * i) * nsplit + z, sc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float x = sc[r];
      if (KB_TOP == 2) {
        const float lo = fminf(g1[r], x);
        g2[r] = fmaxf(g2[r], lo);
      }
      g1[r] = fmaxf(g1[r], x);
    }
  };
  auto finishA = [&](int t) {
    if ((t % nch) == nch - 1) finishA_sub(t / nch);
  };
  KSTAMP(9);
  const int TA = (w < nsub) ? ((my_sub + KB_SAMPLE - 1) / KB_SAMPLE) * nch : 0;  // sampled units of pass A
  // one-unit prefetch: the next fragment's loads are issued in front of the current MFMA chain and handed over by
  // register copies afterwards (tools/probe/mfma_feed.hip sustains 110 TFLOP/s this way).  The two-buffer form
  // unrolled by hand compiled to chains that waited, through the single in-order vmcnt counter, on the loads
  // issued right in front of them.
  // candidate pieces: 8 consecutive channels of a piece are 16 contiguous bytes of the packed row
  r3d_u32x4 pfA[8], pfB[8];  // [k-step of 16 channels][hi, lo]
  // (buffer loads: one resource for the set's pieces, the lane's part of the address is its candidate and half alone,
  // the chunk is a scalar offset -- no 64-bit vector address arithmetic per load)
  const __amdgpu_buffer_rsrc_t rpk = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(BFA ? xpb : (const unsigned short*)xT), 0, BFA ? (int)((long)N * 2 * Cp * 2) : 0, 0x00020000);
  const int chunk_b = N * 16;  // bytes of one chunk of all points
  auto pload = [&](int i, int ch, r3d_u32x4 (&pf)[8]) {  // chunk ch of this wave's i-th sub-tile at the running stride
    const int st = (w + KB_WAVES * stride * i) * nsplit + z;
    const int cc = min(32 * st + j, n - 1);
    const int voff = (h * N + cc) * 16;                  // chunk 8 ch + 2 s4 + h of the hi piece ...
    const int soff = 8 * ch * chunk_b, lo = (Cp / 8) * chunk_b;  // ... and of the lo piece, Cp / 8 chunks further on
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      pf[2 * s4] = __builtin_amdgcn_raw_buffer_load_b128(rpk, voff, soff + 2 * s4 * chunk_b, 0);
      pf[2 * s4 + 1] = __builtin_amdgcn_raw_buffer_load_b128(rpk, voff, soff + lo + 2 * s4 * chunk_b, 0);
    }
  };
  auto pmma = [&](int ch, const r3d_u32x4 (&pf)[8]) {
    if (ch == 0) {  // (uniform)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
    const unsigned short* ap = Aqb + j * QRS + 64 * ch + 8 * h;
#pragma unroll
    for (int half = 0; half < 2; ++half) {  // (two k-steps' query pieces at a time: 16 registers, not 32)
      r3d_u32x4 ah[2], al[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        ah[u] = *reinterpret_cast<const r3d_u32x4*>(ap + 16 * (2 * half + u));
        al[u] = *reinterpret_cast<const r3d_u32x4*>(ap + Cp + 16 * (2 * half + u));
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int s4 = 2 * half + u;
        acc = r3d_mfma_bf16(al[u], pf[2 * s4], acc);
        acc = r3d_mfma_bf16(ah[u], pf[2 * s4 + 1], acc);
        acc = r3d_mfma_bf16(ah[u], pf[2 * s4], acc);
      }
    }
  };
  // One bf16 pass over n_sub sub-tiles of this wave: (sub-tile, chunk) units walked by two scalar counters (no division
  // per unit), the two fragment buffers taking turns (no register copies), the next unit's 8 loads issued in front of the
  // current unit's 12 MFMAs; the last unit requests its own fragments again, so that no load sits behind a branch (the
  // compiler would wait for ALL loads in flight there).  finish(i): the i-th sub-tile's 32 x 32 bounds are in acc.
  auto bf_pass = [&](int n_sub, auto&& finish) {
    const int U = n_sub * nch;
    if (U <= 0) return;
    int i_c = 0, ch_c = 0, i_n = 0, ch_n = 0;
    auto advance = [&](int& i, int& ch) {
      if (++ch == nch) { ch = 0; ++i; }
    };
    pload(0, 0, pfA);
    advance(i_n, ch_n);
    int u = 0;
    for (; u + 1 < U; u += 2) {
      pload(i_n, ch_n, pfB);  // unit u + 1
      __builtin_amdgcn_sched_barrier(0);
      pmma(ch_c, pfA);
      if (ch_c == nch - 1) finish(i_c);
      i_c = i_n; ch_c = ch_n;
      advance(i_n, ch_n);
      const bool more = u + 2 < U;
      pload(more ? i_n : i_c, more ? ch_n : ch_c, pfA);  // unit u + 2 (or this one again)
      __builtin_amdgcn_sched_barrier(0);
      pmma(ch_c, pfB);
      if (ch_c == nch - 1) finish(i_c);
      i_c = i_n; ch_c = ch_n;
      advance(i_n, ch_n);
    }
    if (u < U) {
      pmma(ch_c, pfA);
      if (ch_c == nch - 1) finish(i_c);
    }
  };
  if (BFA) {
    load_kb(0);
    bf_pass((w < nsub) ? (my_sub + KB_SAMPLE - 1) / KB_SAMPLE : 0, finishA_sub);
  } else {
    if (TA > 0) bload(0, bfB);
    for (int t = 0; t < TA; ++t) {  // (copies at the top of the trip: see pass B)
#pragma unroll
      for (int s = 0; s < KCH; ++s) bfA[s] = bfB[s];
      if (t + 1 < TA) bload(t + 1, bfB);
      __builtin_amdgcn_sched_barrier(0);
      mma(t, bfA);
      finishA(t);
    }
  }
  KSTAMP(10);
  stride = 1;
  {
    constexpr int GS = KB_TOP * KB_GROUPS + 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = r3d_acc_row(r, lane);
      region[q * GS + KB_TOP * (32 * w + j)] = g1[r];
      if (KB_TOP == 2) region[q * GS + 2 * (32 * w + j) + 1] = g2[r];
    }
    __syncthreads();
    for (int qq = 0; qq < 32 / KB_WAVES; ++qq) {
      const int q = (32 / KB_WAVES) * w + qq;
      unsigned key[KB_TOP * KB_GROUPS / 64];
#pragma unroll
      for (int i = 0; i < KB_TOP * KB_GROUPS / 64; ++i) key[i] = f2key(region[q * GS + 64 * i + lane]) >> 16;
      unsigned res = 0;  // largest 16-bit key prefix with at least k retained values at or above it
      for (int bit = 15; bit >= 0; --bit) {
        const unsigned cand = res | (1u << bit);
        int c = 0;
#pragma unroll
        for (int i = 0; i < KB_TOP * KB_GROUPS / 64; ++i) c += __popcll(__ballot(key[i] >= cand));
        if (c >= k) res = cand;
      }
      if (lane == 0) tau_s[q] = key2f(res << 16);  // truncation rounds DOWN in key order: still a lower bound
    }
    __syncthreads();
  }
  KSTAMP(11);
  // (rows beyond n are never emitted: nothing is appended for them -- their pass-B scores, from zeroed fragments, have
  // nothing to do with a threshold taken from the clamped row's bf16 pieces, and would overflow the buffer for nothing)
  float tauq[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) tauq[r] = q0 + r3d_acc_row(r, lane) < n ? tau_s[r3d_acc_row(r, lane)] : INFINITY;
  __syncthreads();  // gmax region is reused as the survivor buffers from here on

  // ------------------------------------------------------------------ pass B: append survivors
  // (BFB: the filter pass appends indices only, at the start of the region -- 32 KB_CAP words lie below the query pieces,
  // which that pass still reads; the exact scores are written behind them once the pass is over)
  static_assert(!BFB || 32 * KB_CAP <= 32 * (KB_TOP * KB_GROUPS + 1), "the index buffer must end in front of the query pieces");
  float* bufv = region + (BFB ? 32 * KB_CAP : 0);          // [32][KB_CAP]
  int* bufi = (int*)(region + (BFB ? 0 : 32 * KB_CAP));    // [32][KB_CAP]
  // BMP (the filter pass of the k <= 32 configuration): survivors are recorded as ONE BIT each -- the ballot of a score
  // register's comparison is the 32-candidate mask of two queries -- in a bitmap [sub-tile][query] that sits in the index
  // buffer's own place (32 KB_CAP words: up to KB_CAP sub-tiles, N <= 32 KB_CAP; the launcher sees to it) and is turned
  // into the index lists per query once the pass is over.  No LDS atomic: an appending ds_add_rtn costs ~70 cycles per
  // INSTRUCTION whatever the number of survivors among its lanes, 16 per sub-tile and wave = 18 k of the pass's 59 k
  // cycles per tile (profiles/r04_experiments.md section 6); and the lists come out in candidate order.
  constexpr bool BMP = BFB && KB_WAVES == 4;
  unsigned* bm = reinterpret_cast<unsigned*>(region);  // [sub-tile ls][(q + ls) & 31]: rows written, columns read without bank conflicts
  bool overflow = false;
  auto finishB = [&](int t) {
    if ((t % nch) != nch - 1) return;
    const int st = (w + KB_WAVES * (t / nch)) * nsplit + z;
    f32x16 sc;
    scores(st, sc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (sc[r] >= tauq[r]) {
        const int q = r3d_acc_row(r, lane);
        const int slot = atomicAdd(&cnt_s[q], 1);
        if (slot < KB_CAP) {
          bufv[q * KB_CAP + slot] = sc[r];
          bufi[q * KB_CAP + slot] = 32 * st + j;
        } else {
          overflow = true;
        }
      }
    }
  };
  // the bf16 filter (BFB): append the INDEX of every candidate whose score's upper bound reaches tau
  auto finishBF = [&](int i) {  // the wave's i-th sub-tile of the filter pass is complete in acc
    const int st = (w + KB_WAVES * i) * nsplit + z;
    f32x16 sc;
    scores_ub(st, sc);
    if (BMP) {
      // lane q < 32 collects the mask of query q: registers r of the two lane halves are queries row(r, 0), row(r, 1)
      // (v_writelane_b32 with a constant lane: this clang has no builtin for it.  A v_writelane that reads an SGPR as DATA
      // right behind the VALU instruction that wrote it gets the old value on gfx950 -- measured: every tile came out with
      // wrong masks until wait states separated the two; the compiler's hazard recogniser does not look into inline
      // assembly -- so all 16 comparisons are issued first, then the 32 lane writes: 16+ instructions lie between a
      // comparison and the first read of its mask; five wait states in front of each statement -- the distance that was
      // measured to be enough -- cover a mask the register allocator might reload right there.  A wrong mask cannot go
      // unnoticed in the tests: indices and score bits are compared with the oracle's.)
      unsigned long long m[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) m[r] = __ballot(sc[r] >= tauq[r]);
      __builtin_amdgcn_sched_barrier(0);
      int word = 0;
#define KB_LO(r) "s"((int)(unsigned)(m[r] & 0xffffffffull))
#define KB_HI(r) "s"((int)(unsigned)(m[r] >> 32))
      // (two statements: an asm statement takes at most 30 operands; register r's two masks go to lanes row(r, 0), row(r, 1))
      asm volatile(
                   "s_nop 4\n\t"
                   "v_writelane_b32 %0, %1, 0\n\t"
                   "v_writelane_b32 %0, %2, 4\n\t"
                   "v_writelane_b32 %0, %3, 1\n\t"
                   "v_writelane_b32 %0, %4, 5\n\t"
                   "v_writelane_b32 %0, %5, 2\n\t"
                   "v_writelane_b32 %0, %6, 6\n\t"
                   "v_writelane_b32 %0, %7, 3\n\t"
                   "v_writelane_b32 %0, %8, 7\n\t"
                   "v_writelane_b32 %0, %9, 8\n\t"
                   "v_writelane_b32 %0, %10, 12\n\t"
                   "v_writelane_b32 %0, %11, 9\n\t"
                   "v_writelane_b32 %0, %12, 13\n\t"
                   "v_writelane_b32 %0, %13, 10\n\t"
                   "v_writelane_b32 %0, %14, 14\n\t"
                   "v_writelane_b32 %0, %15, 11\n\t"
                   "v_writelane_b32 %0, %16, 15"
                   : "+v"(word)
                   : KB_LO(0), KB_HI(0), KB_LO(1), KB_HI(1), KB_LO(2), KB_HI(2), KB_LO(3), KB_HI(3), KB_LO(4), KB_HI(4), KB_LO(5), KB_HI(5), KB_LO(6), KB_HI(6), KB_LO(7), KB_HI(7));
      asm volatile(
                   "s_nop 4\n\t"  // (as above: should the register allocator ever reload a mask right in front of this statement)
                   "v_writelane_b32 %0, %1, 16\n\t"
                   "v_writelane_b32 %0, %2, 20\n\t"
                   "v_writelane_b32 %0, %3, 17\n\t"
                   "v_writelane_b32 %0, %4, 21\n\t"
                   "v_writelane_b32 %0, %5, 18\n\t"
                   "v_writelane_b32 %0, %6, 22\n\t"
                   "v_writelane_b32 %0, %7, 19\n\t"
                   "v_writelane_b32 %0, %8, 23\n\t"
                   "v_writelane_b32 %0, %9, 24\n\t"
                   "v_writelane_b32 %0, %10, 28\n\t"
                   "v_writelane_b32 %0, %11, 25\n\t"
                   "v_writelane_b32 %0, %12, 29\n\t"
                   "v_writelane_b32 %0, %13, 26\n\t"
                   "v_writelane_b32 %0, %14, 30\n\t"
                   "v_writelane_b32 %0, %15, 27\n\t"
                   "v_writelane_b32 %0, %16, 31"
                   : "+v"(word)
                   : KB_LO(8), KB_HI(8), KB_LO(9), KB_HI(9), KB_LO(10), KB_HI(10), KB_LO(11), KB_HI(11), KB_LO(12), KB_HI(12), KB_LO(13), KB_HI(13), KB_LO(14), KB_HI(14), KB_LO(15), KB_HI(15));
#undef KB_LO
#undef KB_HI
      __builtin_amdgcn_sched_barrier(0);
      const int ls = w + KB_WAVES * i;  // (nsplit == 1 in this configuration: st == ls)
      if (lane < 32) bm[ls * 32 + ((lane + ls) & 31)] = (unsigned)word;
      return;
    }
    // (one returning LDS atomic per survivor.  Reserving a half-wave's slots with ONE atomic per register -- all 16 issued
    // back to back, the survivors ranked by ballot -- was measured and lost: 68.8 k against 59.2 k cycles per tile for this
    // pass; most registers have no survivor, and the branch around the atomic is cheaper than the ranking arithmetic.)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (sc[r] >= tauq[r]) {
        const int q = r3d_acc_row(r, lane);
        const int slot = atomicAdd(&cnt_s[q], 1);
        if (slot < KB_CAP) bufi[q * KB_CAP + slot] = 32 * st + j;
        else overflow = true;
      }
    }
  };
  // one-unit prefetch: the next fragment's loads are issued in front of the current MFMA chain and handed over by
  // register copies afterwards (tools/probe/mfma_feed.hip sustains 110 TFLOP/s this way).  The two-buffer form
  // unrolled by hand compiled to chains that waited, through the single in-order vmcnt counter, on the loads
  // issued right in front of them.
  if (BFB) {
    load_kb(1);
    bf_pass((w < nsub) ? my_sub : 0, finishBF);
  } else {
    // (the hand-over copies stand at the TOP of the trip: with the first fragment loaded straight into bfA in front of
    // the loop, the compiler's wait-count pass took bfA for load results at the loop header and made every MFMA of a
    // chain wait for one more of the loads issued right in front of it -- a memory round trip at the start of every
    // chain, 78 k of pass B's 358 k cycles per tile in the 201-NN)
    if (T > 0) bload(0, bfB);
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int s = 0; s < KCH; ++s) bfA[s] = bfB[s];
      if (t + 1 < T) bload(t + 1, bfB);
      __builtin_amdgcn_sched_barrier(0);
      mma(t, bfA);
      finishB(t);
    }
  }
  KSTAMP(12);
  if (BMP) {
    __syncthreads();  // every sub-tile's masks are in the bitmap
    constexpr int QW = 32 / KB_WAVES, NCH = (KB_CAP + 63) / 64;  // (nsub <= KB_CAP sub-tiles: NCH words per lane and query)
    unsigned wq[QW][NCH];
#pragma unroll
    for (int i = 0; i < QW; ++i)
#pragma unroll
      for (int cch = 0; cch < NCH; ++cch) {
        const int ls = 64 * cch + lane;
        wq[i][cch] = ls < nsub ? bm[ls * 32 + ((QW * w + i + ls) & 31)] : 0u;
      }
    __syncthreads();  // the index lists take the bitmap's place
#pragma unroll
    for (int i = 0; i < QW; ++i) {
      const int q = QW * w + i;
      int total = 0;
#pragma unroll
      for (int cch = 0; cch < NCH; ++cch) {
        if (64 * cch >= nsub) break;  // (uniform)
        unsigned word = wq[i][cch];
        const int c = __popc(word);
        const int incl = r3d_wave_incl_scan(c);
        int off = total + incl - c;
        const int cand0 = 32 * ((64 * cch + lane) * nsplit + z);
        while (word) {  // candidates in ascending order
          const int bit = __ffs((int)word) - 1;
          if (off < KB_CAP) bufi[q * KB_CAP + off] = cand0 + bit;
          ++off;
          word &= word - 1;
        }
        total += __builtin_amdgcn_readlane(incl, 63);
      }
      if (lane == 0) cnt_s[q] = total;
      overflow = overflow || total > KB_CAP;
    }
  }
  if (__any(overflow) && lane == 0) {
    if (status) atomicOr(status, 1);
    if (tile_flags) tile_flags[(long)b * gridDim.x + tile_x] = 1;
  }
  __syncthreads();
  if (BFA && tid < 32 && q0 + tid < n && cnt_s[tid] < min(k, n)) {
    // fewer than k survivors: the bound was not one (non-finite features).  Same exit as an overflow: flagged (bit 1 tells
    // the two apart for diagnosis; callers test the word against 0), redone exactly.
    if (status) atomicOr(status, 2);
    if (tile_flags) tile_flags[(long)b * gridDim.x + tile_x] = 1;
  }
  KSTAMP(14);

  // ------------------------------------------------------------------ exact scores of the survivors (BFB)
  // The survivors of this wave's queries, dealt to the lanes as one flat list: a lane evaluates its survivor's score as
  // the channel-ascending fmaf chain from 0 (== the fp32 MFMA's accumulation) over the point-major row, then the score
  // form's own roundings exactly as `scores` above.  Rows are read 32 channels (8 x 16 bytes) at a time with the next
  // request in flight behind the current chain.
  if (BFB) {
    constexpr int QW = 32 / KB_WAVES;
    int off[QW + 1];
    off[0] = 0;
#pragma unroll
    for (int i = 0; i < QW; ++i) off[i + 1] = off[i] + __builtin_amdgcn_readfirstlane(min(cnt_s[QW * w + i], KB_CAP));
    const int total = off[QW];
    const int nseg = C / 32;                      // (C % 64 == 0)
    const int T2 = ((total + 63) / 64) * nseg;    // (round, segment) units of this wave
    int q_cur = 0, slot_cur = 0;
    bool ok_cur = false;
    const float4* xr_cur = nullptr;
    auto locate = [&](int t, int& q, int& slot, bool& ok, const float4*& xr) {
      const int f = 64 * (t / nseg) + lane;
      ok = f < total;
      int ql = 0;
#pragma unroll
      for (int i = 1; i < QW; ++i) ql += f >= off[i] ? 1 : 0;
      ql = ok ? ql : 0;
      q = QW * w + ql;
      slot = ok ? f - off[ql] : 0;
      const int cand = min(max(bufi[q * KB_CAP + slot], 0), n - 1);
      xr = reinterpret_cast<const float4*>(xpm + ((long)b * N + cand) * ldx) + 8 * (t % nseg);
    };
    float4 xa[8], xb[8];
    int qn = 0, slotn = 0;
    bool okn = false;
    const float4* xrn = nullptr;
    if (T2 > 0) {
      locate(0, q_cur, slot_cur, ok_cur, xr_cur);
#pragma unroll
      for (int u = 0; u < 8; ++u) xa[u] = xr_cur[u];
    }
    float a = 0.f;
    for (int t = 0; t < T2; ++t) {
      const int tn = min(t + 1, T2 - 1);  // (the last unit requests its own rows again: no branch around the loads)
      locate(tn, qn, slotn, okn, xrn);
#pragma unroll
      for (int u = 0; u < 8; ++u) xb[u] = xrn[u];
      __builtin_amdgcn_sched_barrier(0);
      const int sg = t % nseg;
      if (sg == 0) a = 0.f;
      const float4* aq = reinterpret_cast<const float4*>(Aq + q_cur * Cs + 32 * sg);  // (Cs % 4 == 0 here)
      float4 qa[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) qa[u] = aq[u];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a = __builtin_fmaf(qa[u].x, xa[u].x, a);
        a = __builtin_fmaf(qa[u].y, xa[u].y, a);
        a = __builtin_fmaf(qa[u].z, xa[u].z, a);
        a = __builtin_fmaf(qa[u].w, xa[u].w, a);
      }
      if (sg == nseg - 1 && ok_cur) {
        const int cand = min(max(bufi[q_cur * KB_CAP + slot_cur], 0), n - 1);
        const float nj = nb[cand], nqq = nb[min(q0 + q_cur, n - 1)];
        float v;
        if (SMODE == R3D_SCORE_DGCNN) {
          const float inner = -2.f * a;
          v = ((-nj) - inner) - nqq;
        } else {
          const float dis = (nqq + nj) - 2.f * a;
          v = -fmaxf(dis, 0.f);
        }
        bufv[q_cur * KB_CAP + slot_cur] = v;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) xa[u] = xb[u];
      q_cur = qn; slot_cur = slotn; ok_cur = okn;
    }
  }
  KSTAMP(15);
  // ------------------------------------------------------------------ rank the survivors
  for (int qq = 0; qq < 32 / KB_WAVES; ++qq) {
    const int q = (32 / KB_WAVES) * w + qq;
    const int row = q0 + q;
    if (row >= n) break;
    const int M = __builtin_amdgcn_readfirstlane(min(cnt_s[q], KB_CAP));
    float mv[KB_CAP / 64];
    int mi[KB_CAP / 64], rank[KB_CAP / 64];
    // one 64-bit key per survivor, larger = better: order-preserving score bits above the complemented index
    // (equal scores: the lower index wins, as entry_better)
    unsigned khi[KB_CAP / 64], klo[KB_CAP / 64];
    unsigned long long key[KB_CAP / 64];
#pragma unroll
    for (int i = 0; i < KB_CAP / 64; ++i) {
      const int e = 64 * i + lane;
      mi[i] = e < M ? bufi[q * KB_CAP + e] : 0x7fffffff;
      mv[i] = e < M ? bufv[q * KB_CAP + e] : -INFINITY;
      khi[i] = f2key(mv[i] + 0.0f);  // -0 -> +0: equal scores must have equal keys
      klo[i] = 0x7fffffffu - (unsigned)mi[i];
      key[i] = ((unsigned long long)khi[i] << 32) | klo[i];
      rank[i] = 0;
    }
    // M > 64 (the 201-NN: ~230 survivors): SORT the keys (knn_bitonic_desc) with the buffer slot in the low 9 bits
    // behind a 22-bit index, so that a sorted key still finds its score bits (-0 and +0 share a key) and its index
    bool sorted = false;
    if constexpr (KB_CAP >= 256) {
    if (M > 64 && N <= (1 << 22)) {  // (uniform)
      constexpr int RMAX = KB_CAP > 256 ? 8 : 4;  // registers of the larger network: 64 RMAX >= KB_CAP
      static_assert(64 * RMAX >= KB_CAP && KB_CAP <= 512, "the slot field of a sort key has 9 bits");
      unsigned sh[RMAX], sl[RMAX];
#pragma unroll
      for (int i = 0; i < RMAX; ++i) {
        const int e = 64 * i + lane;
        const bool in = i < KB_CAP / 64 && e < M;
        sh[i] = in ? khi[i < KB_CAP / 64 ? i : 0] : 0u;
        sl[i] = in ? ((0x3fffffu - (unsigned)mi[i < KB_CAP / 64 ? i : 0]) << 9) | (unsigned)e : 0u;
      }
      if (M <= 256) {  // (uniform) the usual case: four registers, 36 stages
        unsigned h4[4] = {sh[0], sh[1], sh[2], sh[3]}, l4[4] = {sl[0], sl[1], sl[2], sl[3]};
        knn_bitonic_desc<4>(h4, l4, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) sl[i] = l4[i];
      } else {
        knn_bitonic_desc<RMAX>(sh, sl, lane);
      }
#pragma unroll
      for (int i = 0; i < KB_CAP / 64; ++i) {
        const int e = 64 * i + lane;
        const int slot = (int)(sl[i] & 511u);
        const bool out = e < M;
        mi[i] = out ? bufi[q * KB_CAP + slot] : 0x7fffffff;
        mv[i] = out ? bufv[q * KB_CAP + slot] : -INFINITY;
        rank[i] = e;  // the sorted position
      }
      sorted = true;
    }
    }
    // every survivor against every survivor: the opponent's key comes out of the registers the entries sit in
    // (two v_readlane with a uniform lane index), one 64-bit compare and one add per own entry
    if (sorted) {
    } else if (M <= 64) {  // uniform; the usual case of the k <= 32 configuration: one register of entries, half the compares
#pragma unroll 4
      for (int t = 0; t < M; ++t) {
        const unsigned long long ok = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)khi[0], t) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)klo[0], t);
        rank[0] += ok > key[0] ? 1 : 0;
      }
    } else {
#pragma unroll
      for (int c = 0; c < KB_CAP / 64; ++c) {
        const int cnt = min(max(M - 64 * c, 0), 64);  // uniform
#pragma unroll 4
        for (int t = 0; t < cnt; ++t) {
          const unsigned long long ok = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)khi[c], t) << 32) |
                                        (unsigned)__builtin_amdgcn_readlane((int)klo[c], t);
#pragma unroll
          for (int i = 0; i < KB_CAP / 64; ++i) rank[i] += ok > key[i] ? 1 : 0;
        }
      }
    }
    if (nsplit == 1) {
#pragma unroll
      for (int i = 0; i < KB_CAP / 64; ++i) {
        if (64 * i + lane < M && rank[i] < k) {
          idx_out[((long)b * N + row) * k + rank[i]] = mi[i];
          if (score_out) score_out[((long)b * N + row) * k + rank[i]] = mv[i];
        }
      }
    } else {
      const long o = (((long)z * gridDim.y + b) * N + row) * k;
#pragma unroll
      for (int i = 0; i < KB_CAP / 64; ++i) {
        if (64 * i + lane < M && rank[i] < k) {
          idx_tmp[o + rank[i]] = mi[i];
          sc_tmp[o + rank[i]] = mv[i];
        }
      }
      for (int t = min(M, k) + lane; t < k; t += 64) {  // fewer than k survivors (a part with < k candidates)
        idx_tmp[o + t] = 0x7fffffff;
        sc_tmp[o + t] = -INFINITY;
      }
    }
  }
  KSTAMP(13);
}

// Merge of the nsplit sorted top-k lists of a row (r3d_knn_append_kernel with nsplit > 1): an entry's final rank is
// its own rank plus, for every other list, the number of that list's entries that beat it (binary search; the lists
// hold disjoint candidates, so keys never tie across lists).  One wave per row, lists in LDS.  Keys as in the rank
// phase: (order-preserving score bits, complemented index), larger = better.
__global__ __launch_bounds__(256) void r3d_knn_merge_kernel(const int* __restrict__ idx_tmp, const float* __restrict__ sc_tmp,
                                                            int nsplit, long rows, int k, const int* __restrict__ n_dev, int n_dev_stride, int N,
                                                            int* __restrict__ idx_out, float* __restrict__ score_out) {
  __shared__ unsigned long long keys[4][2][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + w;
  if (row >= rows) return;
  if (n_dev && (int)(row % N) >= min(n_dev[(row / N) * n_dev_stride], N)) return;  // rows beyond n are not queried
  // nsplit == 2 (the only configuration launched)
  for (int p = 0; p < 2; ++p)
    for (int t = lane; t < k; t += 64) {
      const long o = ((long)p * rows + row) * k + t;
      const float v = sc_tmp[o];
      const int id = idx_tmp[o];
      keys[w][p][t] = id == 0x7fffffff ? 0ull
                                       : (((unsigned long long)f2key(v + 0.0f) << 32) | (0x7fffffffu - (unsigned)id));
    }
  __builtin_amdgcn_wave_barrier();
  for (int p = 0; p < 2; ++p)
    for (int t = lane; t < k; t += 64) {
      const unsigned long long mine = keys[w][p][t];
      if (mine == 0ull) continue;
      const unsigned long long* other = keys[w][1 - p];
      int lo = 0, hi = k;  // number of entries of the other list with a larger key (lists are sorted descending)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (other[mid] > mine) lo = mid + 1; else hi = mid;
      }
      const int rank = t + lo;
      if (rank < k) {
        const long o = ((long)p * rows + row) * k + t;
        idx_out[row * k + rank] = idx_tmp[o];
        if (score_out) score_out[row * k + rank] = sc_tmp[o];
      }
    }
}

static size_t knn_append_lds_bytes(int C, int waves, int cap, int top, int kch = 32, bool bfa = false) {
  const int nch = (C + 2 * kch - 1) / (2 * kch);
  const size_t a = 32 * (size_t)(nch * 2 * kch + 4);  // (row stride + 1 for the fp32 passes, + 4 for the filter form)
  size_t g = 32 * (size_t)(top * waves * 32 + 1);
  if (bfa) g = ((g + 3) & ~(size_t)3) + 32 * (size_t)(2 * nch * 64 + 8) / 2 + 64;  // + the query rows' bf16 pieces, bound terms
  const size_t bsz = 2 * 32 * (size_t)cap;
  return sizeof(float) * (a + (g > bsz ? g : bsz));
}
static size_t knn_big_lds_bytes(int C, bool bfa = false) { return knn_append_lds_bytes(C, 8, 384, 2, 32, bfa); }

// bf16 pieces of the points for the bf16 passes of r3d_knn_append_kernel<..., BFA = true>, per set [piece][chunk of 8
// channels][point][8] (chunk-major: see the kernel's staging loop for why),
// Cp = C rounded up to 64 (zeros), cut from the point MINUS its set's mean: hi = the top 16 bits of the value, lo = the top
// 16 bits of what is left.  Any mean will do (the bound only needs pieces, centred norms and inner products to belong to
// the same shifted points); it is the plain fp32 mean over the set's valid rows.
__global__ __launch_bounds__(1024) void r3d_knn_mean_kernel(const float* __restrict__ x, long ldx, int N, int C,
                                                            const int* __restrict__ n_dev, int n_dev_stride, int Cp,
                                                            float* __restrict__ mean /* [B][Cp] */) {
  __shared__ float part[16][64];
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  const int n = n_dev ? min(n_dev[(long)b * n_dev_stride], N) : N;
  float s = 0.f;
  if (c < C) {
    const float* p = x + (long)b * N * ldx + c;
    int r = ph;
    for (; r + 7 * 16 < n; r += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(long)(r + 16 * u) * ldx];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < n; r += 16) s += p[(long)r * ldx];
  }
  part[ph][threadIdx.x & 63] = s;
  __syncthreads();
  if (ph == 0 && c < Cp) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += part[q][threadIdx.x];
    mean[(long)b * Cp + c] = c < C && n > 0 ? t / (float)n : 0.f;
  }
}
__global__ void r3d_knn_pack_bf_kernel(const float* __restrict__ x, long ldx, long rows, int N, int C, int Cp,
                                       const float* __restrict__ mean, unsigned short* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cpr = Cp / 8;
  if (i >= rows * cpr) return;
  // consecutive threads: the consecutive 32-byte chunks of one point's row, then the next point -- a wave reads eight
  // whole rows (16 cache lines per load instruction; chunk-fastest over points it touched 64) and stores eight runs of
  // 128 contiguous bytes per piece plane
  const long m = i / cpr;
  const int c8 = (int)(i - m * cpr);
  const long set = m / N;
  const int pt = (int)(m - set * N);
  const int c0 = 8 * c8;
  const float* mu = mean + set * Cp + c0;
  float v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = r3d_keep(x[m * ldx + min(c0 + u, C - 1)] - mu[u], c0 + u < C);
  const r3d_bx3 f = r3d_bx3_split8(v);
  unsigned short* o = out + set * N * 2 * Cp;
  *reinterpret_cast<r3d_u32x4*>(o + ((long)c8 * N + pt) * 8) = f.h;
  *reinterpret_cast<r3d_u32x4*>(o + ((long)(cpr + c8) * N + pt) * 8) = f.m;
}
// squared norms of the centred points (one wave per row; fixed order)
__global__ __launch_bounds__(256) void r3d_knn_cnorm_kernel(const float* __restrict__ x, long ldx, long rows, int N, int C, int Cp,
                                                            const float* __restrict__ mean, float* __restrict__ cnorm) {
  // 16 lanes per point, four points per wave: a lane adds the squares of its 16-byte pieces, the 16 partial sums meet in
  // four DPP steps (ldx % 4 == 0 and 16-byte aligned rows: the filter pass asks the same of them; otherwise one float per
  // lane and trip).  The sum's order only moves the bound it feeds by roundings that the bound's slack covers.
  const int lane = threadIdx.x & 63, sub = lane & 15;
  const long m = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const long mc = m < rows ? m : rows - 1;
  const float* mu = mean + (mc / N) * Cp;
  float s = 0.f;
  if ((ldx & 3) == 0 && (C & 3) == 0 && ((uintptr_t)x & 15) == 0) {
    for (int c = 4 * sub; c < C; c += 64) {
      const float4 v = *reinterpret_cast<const float4*>(x + mc * ldx + c), u = *reinterpret_cast<const float4*>(mu + c);
      const float d0 = v.x - u.x, d1 = v.y - u.y, d2 = v.z - u.z, d3 = v.w - u.w;
      s = __builtin_fmaf(d0, d0, s); s = __builtin_fmaf(d1, d1, s); s = __builtin_fmaf(d2, d2, s); s = __builtin_fmaf(d3, d3, s);
    }
  } else {
    for (int c = sub; c < C; c += 16) {
      const float d = x[mc * ldx + c] - mu[c];
      s = __builtin_fmaf(d, d, s);
    }
  }
  s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0x141, 0xF, 0xF, true));  // row_half_mirror
  s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0x140, 0xF, 0xF, true));  // row_mirror
  if (sub == 0 && m < rows) cnorm[m] = s;
}
// A/B switch (tests, tools): 0 = the threshold pass stays on the fp32 core even when bf_ws is given.  Same results.
static int g_knn_bf16_threshold = getenv("R3D_KNN_FP32_THRESHOLD") ? 0 : 1;
// ... and: 0 = pass B stays the all-pairs fp32 pass (no bf16 filter + exact scores of the survivors).  Same results.
static int g_knn_bf16_filter = getenv("R3D_KNN_FP32_PASS_B") ? 0 : 1;
// (2: also in the k > 32 configuration, where it loses -- 230 survivors per query, each a row gather the L1 serves one line
// look-up at a time: measured 6.9 ms against 5.6 ms per 32 graphs of 4 396 nodes; kept for tests and tools/knnbench)
extern "C" int r3d_debug_set_knn_bf16_filter(int on) {
  const int old = g_knn_bf16_filter;
  g_knn_bf16_filter = on < 0 ? 0 : on > 2 ? 2 : on;
  return old;
}
extern "C" int r3d_debug_set_knn_bf16_threshold(int on) {
  const int old = g_knn_bf16_threshold;
  g_knn_bf16_threshold = on ? 1 : 0;
  return old;
}
// floats of bf_ws (r3d_knn_topk_batched): the packed pieces, 4 bytes per point and (padded) channel
// (pieces B N Cp | means B Cp | centred squared norms B N)
extern "C" long r3d_knn_bf_ws_words(int B, int N, int C) {
  const long Cp = ((long)C + 63) / 64 * 64;
  return (long)B * N * Cp + (long)B * Cp + (long)B * N + 64;
}
#ifndef KM_WAVES  // mid configuration (overridable for tools/knnbench sweeps)
#define KM_WAVES 4
#define KM_CAP 128
#define KM_TOP 1
#endif
#ifndef KM_SAMPLE
#define KM_SAMPLE 2  // pass A on half of the candidates: ~2k survivors in pass B (40 of 128 slots at k = 20); measured
                     // 277 us per call against 306 (no sampling), 324 (1/4, 192 slots) and 841 (1/8: repairs)
#endif

static size_t knn_lds_bytes(int C) {
  const int Cp = (C + 1) & ~1;
  return sizeof(float) * ((size_t)KNN_Q * (Cp + 1) + (size_t)KNN_CH * (KNN_SLAB + 1) +
                          (size_t)KNN_Q * (KNN_CH + 1) + KNN_Q);
}

// launch one instance of the append-and-rank kernel (raising its dynamic-LDS limit once per instance)
template <int WAVES, int CAP, int TOP, int KCH, int SAMPLE, bool FULLC, int SMODE, bool BFA = false, bool BFB = false>
static int knn_append_launch_mode(dim3 grid, size_t lds, hipStream_t st, const float* xT, long ldT, int N, int C, int k,
                                  const int* n_valid_dev, int n_valid_stride, const float* nrm, int* idx_out, float* score_out, int* status,
                                  int* tile_flags, int nsplit = 1, int* idx_tmp = nullptr, float* sc_tmp = nullptr,
                                  const unsigned short* xpk = nullptr, const float* cnorm = nullptr, const float* xpm = nullptr,
                                  long ldx = 0) {
  static size_t attr = 0;
  if (lds > attr) {
    hipError_t e = hipFuncSetAttribute((const void*)r3d_knn_append_kernel<WAVES, CAP, TOP, KCH, SAMPLE, FULLC, SMODE, BFA, BFB>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    R3D_REQUIRE(e == hipSuccess, "r3d_knn_topk: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
    attr = lds;
  }
  hipLaunchKernelGGL((r3d_knn_append_kernel<WAVES, CAP, TOP, KCH, SAMPLE, FULLC, SMODE, BFA, BFB>), grid, dim3(64 * WAVES), lds, st,
                     xT, ldT, N, C, k, SMODE, n_valid_dev, n_valid_stride, nrm, idx_out, score_out, status, tile_flags, nsplit,
                     idx_tmp, sc_tmp, xpk, cnorm, xpm, ldx);
  return R3D_OK;
}
// the same with the threshold pass on the bf16 matrix core (xpk: r3d_knn_pack_bf_kernel's output)
// xpm != nullptr: the filter pass on the bf16 core as well, exact scores for the survivors alone (BFB)
template <int WAVES, int CAP, int TOP, int SAMPLE>
static int knn_append_launch_bfa(dim3 grid, size_t lds, hipStream_t st, const float* xT, long ldT, int N, int C, int k, int mode,
                                 const int* n_valid_dev, int n_valid_stride, const float* nrm, int* idx_out, float* score_out,
                                 int* status, int* tile_flags, const unsigned short* xpk, const float* cnorm,
                                 const float* xpm = nullptr, long ldx = 0) {
#define KB_GO(SM, FB)                                                                                                            \
  knn_append_launch_mode<WAVES, CAP, TOP, 32, SAMPLE, true, SM, true, FB>(grid, lds, st, xT, ldT, N, C, k, n_valid_dev, n_valid_stride, \
                                                                          nrm, idx_out, score_out, status, tile_flags, 1, nullptr,   \
                                                                          nullptr, xpk, cnorm, xpm, ldx)
  if (xpm) return mode == R3D_SCORE_DGCNN ? KB_GO(R3D_SCORE_DGCNN, true) : KB_GO(R3D_SCORE_L2, true);
  return mode == R3D_SCORE_DGCNN ? KB_GO(R3D_SCORE_DGCNN, false) : KB_GO(R3D_SCORE_L2, false);
#undef KB_GO
}
template <int WAVES, int CAP, int TOP, int KCH, int SAMPLE, bool FULLC>
static int knn_append_launch(dim3 grid, size_t lds, hipStream_t st, const float* xT, long ldT, int N, int C, int k, int mode,
                             const int* n_valid_dev, int n_valid_stride, const float* nrm, int* idx_out, float* score_out, int* status,
                             int* tile_flags, int nsplit = 1, int* idx_tmp = nullptr, float* sc_tmp = nullptr) {
  return mode == R3D_SCORE_DGCNN
             ? knn_append_launch_mode<WAVES, CAP, TOP, KCH, SAMPLE, FULLC, R3D_SCORE_DGCNN>(grid, lds, st, xT, ldT, N, C, k, n_valid_dev, n_valid_stride,
                                                                                            nrm, idx_out, score_out, status, tile_flags,
                                                                                            nsplit, idx_tmp, sc_tmp)
             : knn_append_launch_mode<WAVES, CAP, TOP, KCH, SAMPLE, FULLC, R3D_SCORE_L2>(grid, lds, st, xT, ldT, N, C, k, n_valid_dev, n_valid_stride, nrm,
                                                                                         idx_out, score_out, status, tile_flags, nsplit,
                                                                                         idx_tmp, sc_tmp);
}

extern "C" int r3d_sqnorm(const float* x, long ldx, long rows, int C, float* out, void* stream) {
  R3D_REQUIRE(x && out && rows > 0 && C > 0 && ldx >= C, "r3d_sqnorm: bad arguments");
  hipLaunchKernelGGL(r3d_sqnorm_kernel, dim3(r3d_cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     ldx, (int)rows, C, out);
  R3D_LAUNCH_CHECK("r3d_sqnorm");
  return R3D_OK;
}

// Forward declarations of the layout kernel living in gemm.hip.
extern "C" int r3d_pm_to_cm_pitched(const float* in, long ld, int B, int C, int N, float* out, long pitch, void* stream);
// row pitch of the internal channel-major copies: N + 32 floats, so that consecutive channels of
// one point do not sit a power-of-two stride apart (same L2 channel for every load of a chain)
extern "C" long r3d_cm_pitch(int N) { return (long)((N + 31) / 32) * 32 + 32; }
// floats of norm_ws: the squared norms (B*N) + one overflow flag per 32-row tile (k <= 32 path) + padding
extern "C" long r3d_knn_norm_ws_words(int B, int N) { return (long)B * N + (long)B * ((N + 31) / 32) + 64; }

// x: (B*N, ldx) point-major fp32; norm_ws: r3d_knn_norm_ws_words(B, N) fp32 scratch; idx_out: (B, N, k) int32;
// score_out: optional (B, N, k) fp32; n_valid_dev: optional device int, rows >= *n are
// neither queried nor offered as candidates (used by the head where the node count
// is data dependent and stays on the device).
// x_cm: optional (B, C, N) channel-major copy of x (the reference's own layout); when NULL and
// a streamed kernel applies the copy is made into cm_ws (B*C*r3d_cm_pitch(N) floats).
// status: optional device int.  With k > 32 a non-NULL status selects the append-and-rank
// kernel; bit 0 set afterwards = its survivor buffer overflowed and the result is unusable
// (re-run with status == NULL for the insertion kernel).
// floats of split_ws for r3d_knn_topk_split (two partial top-k lists per row: indices and scores)
extern "C" long r3d_knn_split_ws_words(int B, int N, int k) { return 4L * B * N * k + 64; }

static int knn_topk_impl(const float* x, long ldx, const float* x_cm, int B, int N, int C, int k, int mode,
                         const int* n_valid_dev, int n_valid_stride, float* norm_ws, float* cm_ws, int32_t* idx_out,
                         float* score_out, int32_t* status, float* split_ws, long split_ws_words, float* bf_ws,
                         long bf_ws_words, void* stream);

extern "C" int r3d_knn_topk(const float* x, long ldx, const float* x_cm, int B, int N, int C, int k, int mode,
                            const int* n_valid_dev, float* norm_ws, float* cm_ws, int32_t* idx_out,
                            float* score_out, int32_t* status, void* stream) {
  return knn_topk_impl(x, ldx, x_cm, B, N, C, k, mode, n_valid_dev, 0, norm_ws, cm_ws, idx_out, score_out, status, nullptr, 0,
                       nullptr, 0, stream);
}

// r3d_knn_topk with an optional scratch for the large-k streamed kernel: when the query tiles alone cannot fill the
// chip even when doubled (2 B ceil(N / 32) <= 256 workgroups) and split_ws holds r3d_knn_split_ws_words(B, N, k) floats, the candidate
// axis is split over two workgroups per tile and the two sorted lists are merged by a second small kernel.
extern "C" int r3d_knn_topk_split(const float* x, long ldx, const float* x_cm, int B, int N, int C, int k, int mode,
                                  const int* n_valid_dev, float* norm_ws, float* cm_ws, int32_t* idx_out,
                                  float* score_out, int32_t* status, float* split_ws, long split_ws_words, void* stream) {
  return knn_topk_impl(x, ldx, x_cm, B, N, C, k, mode, n_valid_dev, 0, norm_ws, cm_ws, idx_out, score_out, status, split_ws,
                       split_ws_words, nullptr, 0, stream);
}

// The same over a batch of B point sets with their OWN valid counts: set b has n_valid_dev[b * n_valid_stride] rows (the
// graph nodes of B episodes' label-propagation systems, each at its capacity N).  status: ONE word for the batch (bit 0:
// some set's survivor buffer overflowed).
// bf_ws (optional, r3d_knn_bf_ws_words(B, N, C) floats, needs x): lets the streamed kernels run their THRESHOLD pass on
// the bf16 matrix core (a lower bound of every score; the pass that emits neighbours and scores stays fp32, results are
// bit-identical with and without it).
extern "C" int r3d_knn_topk_batched(const float* x, long ldx, const float* x_cm, int B, int N, int C, int k, int mode,
                                    const int* n_valid_dev, int n_valid_stride, float* norm_ws, float* cm_ws,
                                    int32_t* idx_out, float* score_out, int32_t* status, float* split_ws,
                                    long split_ws_words, float* bf_ws, long bf_ws_words, void* stream) {
  R3D_REQUIRE(n_valid_stride >= 0, "r3d_knn_topk_batched: negative stride");
  return knn_topk_impl(x, ldx, x_cm, B, N, C, k, mode, n_valid_dev, n_valid_stride, norm_ws, cm_ws, idx_out, score_out, status,
                       split_ws, split_ws_words, bf_ws, bf_ws_words, stream);
}

static int knn_topk_impl(const float* x, long ldx, const float* x_cm, int B, int N, int C, int k, int mode,
                         const int* n_valid_dev, int n_valid_stride, float* norm_ws, float* cm_ws, int32_t* idx_out,
                         float* score_out, int32_t* status, float* split_ws, long split_ws_words, float* bf_ws,
                         long bf_ws_words, void* stream) {
  R3D_REQUIRE((x || x_cm) && norm_ws && idx_out, "r3d_knn_topk: null pointer");
  // the threshold pass on the bf16 core: whole 64-channel chunks, the packed pieces from the point-major matrix
  const bool bfa = g_knn_bf16_threshold && bf_ws && x && C % 64 == 0 && bf_ws_words >= r3d_knn_bf_ws_words(B, N, C) &&
                   ((uintptr_t)bf_ws & 15) == 0;
  // the filter pass on it as well: exact scores come from the point-major rows, read as 16-byte vectors
  const float* xpm = bfa && g_knn_bf16_filter && (ldx & 3) == 0 && ((uintptr_t)x & 15) == 0 ? x : nullptr;
  float* bf_mean = bf_ws ? bf_ws + (long)B * N * C : nullptr;  // (C == Cp here)
  float* bf_cnorm = bf_ws ? bf_mean + (long)B * C : nullptr;
  auto pack_bf = [&]() {
    hipStream_t s_ = (hipStream_t)stream;
    hipLaunchKernelGGL(r3d_knn_mean_kernel, dim3(C / 64, B), dim3(1024), 0, s_, x, ldx, N, C, n_valid_dev, n_valid_stride, C, bf_mean);
    const long chunks = (long)B * N * (C / 8);
    hipLaunchKernelGGL(r3d_knn_pack_bf_kernel, dim3(r3d_cdiv(chunks, 256)), dim3(256), 0, s_, x, ldx, (long)B * N, N, C, C, bf_mean,
                       (unsigned short*)bf_ws);
    hipLaunchKernelGGL(r3d_knn_cnorm_kernel, dim3(r3d_cdiv((long)B * N, 16)), dim3(256), 0, s_, x, ldx, (long)B * N, N, C, C, bf_mean,
                       bf_cnorm);
  };
  R3D_REQUIRE(B > 0 && N > 0 && C > 0 && (!x || ldx >= C), "r3d_knn_topk: bad shape B=%d N=%d C=%d ldx=%ld", B, N, C, ldx);
  R3D_REQUIRE(k > 0 && k <= N && k <= 256, "r3d_knn_topk: unsupported k=%d (need 1..min(N,256))", k);
  R3D_REQUIRE(mode == R3D_SCORE_DGCNN || mode == R3D_SCORE_L2, "r3d_knn_topk: unknown mode %d", mode);
  hipStream_t st = (hipStream_t)stream;
  if (k <= 32 && C <= 64) {
    const float* xT = x_cm;
    long ldT = N;
    if (!xT) {
      R3D_REQUIRE(cm_ws && x, "r3d_knn_topk: need x_cm or (x and cm_ws)");
      ldT = r3d_cm_pitch(N);
      int rc = r3d_pm_to_cm_pitched(x, ldx, B, C, N, cm_ws, ldT, stream);
      if (rc) return rc;
      xT = cm_ws;
    }
    hipLaunchKernelGGL(r3d_sqnorm_cm_kernel, dim3(r3d_cdiv(N, 256), B), dim3(256), 0, st, xT, ldT, C, N, norm_ws);
    dim3 g2(r3d_cdiv(N, 32), B);
    static const bool two_pass_only = getenv("R3D_KNN_TWO_PASS") != nullptr;  // A/B switch for tools/knnbench
    int* tile_flags = (int*)(norm_ws + (long)B * N);  // r3d_knn_norm_ws_words reserves B * ceil(N/32) words here
    if (two_pass_only) {
      if (C <= 16)
        hipLaunchKernelGGL(r3d_knn_small_kernel<8>, g2, dim3(256), 0, st, xT, ldT, N, C, k, mode, n_valid_dev, n_valid_stride, norm_ws,
                           idx_out, score_out, (const int*)nullptr);
      else
        hipLaunchKernelGGL(r3d_knn_small_kernel<32>, g2, dim3(256), 0, st, xT, ldT, N, C, k, mode, n_valid_dev, n_valid_stride, norm_ws,
                           idx_out, score_out, (const int*)nullptr);
    } else {
      // append-and-rank (mid configuration); tiles whose survivor buffer overflowed are redone by the exact
      // sorted-insertion kernel in the same stream -- no host round trip, never a wrong result
      r3d_zero_words(tile_flags, (long)B * g2.x, st);
      // fewer row tiles than CUs (the 2 query clouds of a training episode: 128 tiles): 8 waves per tile instead of 4,
      // each wave's chain of sub-tiles is half as long
      const bool few = (long)g2.x * B <= 256;
      const size_t lds = knn_append_lds_bytes(C, few ? 8 : KM_WAVES, KM_CAP, KM_TOP, C <= 16 ? 8 : 32);
      int rc;
#define KM_LAUNCH(WAVES, KCH, FULLC)                                                                                       \
  knn_append_launch<WAVES, KM_CAP, KM_TOP, KCH, KM_SAMPLE, FULLC>(g2, lds, st, xT, ldT, N, C, k, mode, n_valid_dev, n_valid_stride, norm_ws, \
                                                                  idx_out, score_out, nullptr, tile_flags)
      if (C <= 16) {
        rc = few ? KM_LAUNCH(8, 8, false) : KM_LAUNCH(KM_WAVES, 8, false);
        if (rc) return rc;
        hipLaunchKernelGGL(r3d_knn_small_kernel<8>, g2, dim3(256), 0, st, xT, ldT, N, C, k, mode, n_valid_dev, n_valid_stride, norm_ws,
                           idx_out, score_out, (const int*)tile_flags);
      } else {
        if (C % 64 == 0 && bfa && !few) {
          pack_bf();
          // (with the filter on the bf16 core the threshold pass visits EVERY sub-tile: a tighter threshold means fewer
          // survivors, and their exact scores cost more than the half pass saved -- 3.30 against 3.63 ms per 384 clouds)
          // (the filter's survivor bitmap, 32 words per sub-tile, sits in the index buffer's 32 KM_CAP words)
          const float* xpm_f = (r3d_cdiv(N, 32) <= KM_CAP) ? xpm : nullptr;
          rc = xpm_f ? knn_append_launch_bfa<KM_WAVES, KM_CAP, KM_TOP, 1>(
                         g2, knn_append_lds_bytes(C, KM_WAVES, KM_CAP, KM_TOP, 32, true), st, xT, ldT, N, C, k, mode, n_valid_dev,
                         n_valid_stride, norm_ws, idx_out, score_out, nullptr, tile_flags, (const unsigned short*)bf_ws, bf_cnorm, xpm_f, ldx)
                   : knn_append_launch_bfa<KM_WAVES, KM_CAP, KM_TOP, KM_SAMPLE>(
                         g2, knn_append_lds_bytes(C, KM_WAVES, KM_CAP, KM_TOP, 32, true), st, xT, ldT, N, C, k, mode, n_valid_dev,
                         n_valid_stride, norm_ws, idx_out, score_out, nullptr, tile_flags, (const unsigned short*)bf_ws, bf_cnorm, nullptr, 0);
        } else if (C % 64 == 0) rc = few ? KM_LAUNCH(8, 32, true) : KM_LAUNCH(KM_WAVES, 32, true);
        else rc = few ? KM_LAUNCH(8, 32, false) : KM_LAUNCH(KM_WAVES, 32, false);
        if (rc) return rc;
        hipLaunchKernelGGL(r3d_knn_small_kernel<32>, g2, dim3(256), 0, st, xT, ldT, N, C, k, mode, n_valid_dev, n_valid_stride, norm_ws,
                           idx_out, score_out, (const int*)tile_flags);
      }
#undef KM_LAUNCH
    }
    R3D_LAUNCH_CHECK("r3d_knn_topk(small)");
    return R3D_OK;
  }
  if (status && k <= 256 && knn_big_lds_bytes(C) <= 160 * 1024) {
    // fast path for large k; *status bit 0 reports survivor-buffer overflow (caller re-runs with
    // status == NULL, which selects the insertion kernel below)
    const float* xT = x_cm;
    long ldT = N;
    if (!xT) {
      R3D_REQUIRE(cm_ws && x, "r3d_knn_topk: need x_cm or (x and cm_ws)");
      ldT = r3d_cm_pitch(N);
      int rc = r3d_pm_to_cm_pitched(x, ldx, B, C, N, cm_ws, ldT, stream);
      if (rc) return rc;
      xT = cm_ws;
    }
    hipLaunchKernelGGL(r3d_sqnorm_cm_kernel, dim3(r3d_cdiv(N, 256), B), dim3(256), 0, st, xT, ldT, C, N, norm_ws);
    r3d_zero_words(status, 1, st);
    {
      const int tiles = r3d_cdiv(N, 32);
      // two workgroups per query tile when BOTH halves of every tile still fit the chip in one round (the kernel's
      // 123 KB of LDS admit one workgroup per CU: at workload S, 138 tiles -> 276 workgroups ran as two rounds, 489 us
      // against 412 us unsplit) and every half still holds >= 2 k candidates
      const bool split = split_ws && 2L * tiles * B <= 256 && N >= 4 * k && split_ws_words >= r3d_knn_split_ws_words(B, N, k);
      const int nsplit = split ? 2 : 1;
      const dim3 gb(tiles, B, nsplit);
      int* idx_tmp = (int*)split_ws;
      float* sc_tmp = split_ws ? split_ws + 2L * B * N * k : nullptr;
      const bool bfa_big = bfa && nsplit == 1 && knn_big_lds_bytes(C, true) <= 160 * 1024;
      if (bfa_big) pack_bf();
      const int rc = bfa_big
                         ? knn_append_launch_bfa<8, 384, 2, 1>(gb, knn_big_lds_bytes(C, true), st, xT, ldT, N, C, k, mode, n_valid_dev,
                                                               n_valid_stride, norm_ws, idx_out, score_out, status, nullptr,
                                                               (const unsigned short*)bf_ws, bf_cnorm, g_knn_bf16_filter > 1 ? xpm : nullptr, ldx)
                     : C % 64 == 0
                         ? knn_append_launch<8, 384, 2, 32, 1, true>(gb, knn_big_lds_bytes(C), st, xT, ldT, N, C, k, mode, n_valid_dev, n_valid_stride,
                                                                     norm_ws, idx_out, score_out, status, nullptr, nsplit, idx_tmp, sc_tmp)
                         : knn_append_launch<8, 384, 2, 32, 1, false>(gb, knn_big_lds_bytes(C), st, xT, ldT, N, C, k, mode, n_valid_dev, n_valid_stride,
                                                                      norm_ws, idx_out, score_out, status, nullptr, nsplit, idx_tmp, sc_tmp);
      if (rc) return rc;
      if (split)
        hipLaunchKernelGGL(r3d_knn_merge_kernel, dim3(r3d_cdiv((long)B * N, 4)), dim3(256), 0, st, idx_tmp, sc_tmp, 2,
                           (long)B * N, k, n_valid_dev, n_valid_stride, N, idx_out, score_out);
    }
    R3D_LAUNCH_CHECK("r3d_knn_topk(big)");
    return R3D_OK;
  }
  R3D_REQUIRE(x, "r3d_knn_topk: the insertion kernel needs the point-major matrix");
  const size_t lds = knn_lds_bytes(C);
  R3D_REQUIRE(lds <= 160 * 1024, "r3d_knn_topk: C=%d needs %zu B of LDS (> 160 KiB)", C, lds);
  int rc = r3d_sqnorm(x, ldx, (long)B * N, C, norm_ws, stream);
  if (rc) return rc;
  dim3 grid(r3d_cdiv(N, KNN_Q), B), block(256);
#define KNN_LAUNCH(RR)                                                                           \
  do {                                                                                           \
    static bool attr_set = false;                                                                \
    if (!attr_set) {                                                                             \
      hipFuncSetAttribute((const void*)r3d_knn_topk_kernel<RR>,                                  \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);               \
      attr_set = true;                                                                           \
    }                                                                                            \
    hipLaunchKernelGGL(r3d_knn_topk_kernel<RR>, grid, block, lds, st, x, ldx, N, C, k, mode,     \
                       n_valid_dev, n_valid_stride, norm_ws, idx_out, score_out);                \
  } while (0)
  if (k <= 64) KNN_LAUNCH(1);
  else if (k <= 128) KNN_LAUNCH(2);
  else KNN_LAUNCH(4);
#undef KNN_LAUNCH
  R3D_LAUNCH_CHECK("r3d_knn_topk");
  return R3D_OK;
}
