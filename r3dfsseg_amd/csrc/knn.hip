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
    const int* __restrict__ n_dev, const float* __restrict__ nrm, int* __restrict__ idx_out,
    float* __restrict__ score_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * KNN_Q;
  const int n = n_dev ? min(*n_dev, N) : N;  // valid rows of this batch
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
    for (int c = lane; c < Cp; c += 64)
      Aq[r * As + c] = (row < n && c < C) ? xb[(long)row * ldx + c] : 0.f;
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
        for (int c = lane; c < sw; c += 64)
          Bc[r * Bs + c] = (row < n && s0 + c < C) ? xb[(long)row * ldx + s0 + c] : 0.f;
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
      const float nj = valid ? nb[cj] : 0.f;
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
// k <= 32, C <= 64 (the DGCNN encoder's kNN): register-resident selection.
//
// One wave owns 32 query rows and walks ALL candidates; 4 waves (128 queries) share the
// candidate tiles staged in LDS.  Scores never leave the MFMA accumulator: register r of
// lane half h holds, for query row r3d_acc_row(r), the 32 candidates of the tile (one per
// lane of the half).  The sorted top-32 list of that query lives in register lv[r]/li[r]
// across the SAME 32 lanes, so one instruction serves two queries (one per half) and an
// insertion is ballot + popcount + DPP wave_shr:1.  No score tile in LDS, one barrier per
// 64-candidate tile.
// ---------------------------------------------------------------------------
static __device__ __forceinline__ float dpp_wave_shr1_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
static __device__ __forceinline__ int dpp_wave_shr1_i(int v) {
  return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
}

template <int CPAD>
__global__ __launch_bounds__(256) void r3d_knn_small_kernel(
    const float* __restrict__ x, long ldx, int N, int C, int k, int mode, const int* __restrict__ n_dev,
    const float* __restrict__ nrm, int* __restrict__ idx_out, float* __restrict__ score_out) {
  constexpr int KS = CPAD / 2;
  constexpr int LD = CPAD + 1;
  constexpr int TILE = 64;
  constexpr int PER = TILE * CPAD / 256;
  __shared__ float Bc[2][TILE * LD];
  const int b = blockIdx.y;
  const int n = n_dev ? min(*n_dev, N) : N;
  if ((int)blockIdx.x * 128 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int qw0 = blockIdx.x * 128 + 32 * w;
  const float* xb = x + (long)b * N * ldx;
  const float* nb = nrm + (long)b * N;

  float a[KS];
  {
    const int qrow = qw0 + j;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c = 2 * s + h;
      a[s] = (qrow < n && c < C) ? xb[(long)qrow * ldx + c] : 0.f;
    }
  }
  float nq[16], lv[16], thr[16];
  int li[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = qw0 + r3d_acc_row(r, lane);
    nq[r] = row < n ? nb[row] : 0.f;
    lv[r] = -INFINITY;
    li[r] = -1;
    thr[r] = -INFINITY;
  }
  float pre[PER];
  auto gload = [&](int c0) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i;
      const int row = e / CPAD, c = e % CPAD;
      const int grow = c0 + row;
      pre[i] = (grow < n && c < C) ? xb[(long)grow * ldx + c] : 0.f;
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i;
      Bc[buf][(e / CPAD) * LD + (e % CPAD)] = pre[i];
    }
  };
  const int ntiles = (n + TILE - 1) / TILE;
  const int klo = k - 1, khi = 32 + k - 1;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) gload(TILE * (t + 1));
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int cbase = TILE * t + 32 * sub;
      if (cbase >= n) break;
      const int cand = cbase + j;
      const bool valid = cand < n;
      const float nj = valid ? nb[cand] : 0.f;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* bp = &Bc[buf][(32 * sub + j) * LD + h];
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bp[2 * s], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sc;
        if (mode == R3D_SCORE_DGCNN) {
          const float inner = -2.f * acc[r];
          const float tt = (-nj) - inner;
          sc = tt - nq[r];
        } else {
          float dis = (nq[r] + nj) - 2.f * acc[r];
          if (dis < 0.f) dis = 0.f;
          sc = -dis;
        }
        if (!valid) sc = -INFINITY;
        unsigned long long m = __ballot(sc > thr[r]);
        while (m) {
          unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
          const int slo = lo ? __ffs((int)lo) - 1 : 0;
          const int shi = hi ? __ffs((int)hi) - 1 : 0;
          const float cvlo = lo ? r3d_readlane_f(sc, slo) : -INFINITY;
          const float cvhi = hi ? r3d_readlane_f(sc, 32 + shi) : -INFINITY;
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
    }
    if (t + 1 < ntiles) sstore(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = qw0 + r3d_acc_row(r, lane);
    if (row < n && j < k) {
      idx_out[((long)b * N + row) * k + j] = li[r];
      if (score_out) score_out[((long)b * N + row) * k + j] = lv[r];
    }
  }
}

static size_t knn_lds_bytes(int C) {
  const int Cp = (C + 1) & ~1;
  return sizeof(float) * ((size_t)KNN_Q * (Cp + 1) + (size_t)KNN_CH * (KNN_SLAB + 1) +
                          (size_t)KNN_Q * (KNN_CH + 1) + KNN_Q);
}

extern "C" int r3d_sqnorm(const float* x, long ldx, long rows, int C, float* out, void* stream) {
  R3D_REQUIRE(x && out && rows > 0 && C > 0 && ldx >= C, "r3d_sqnorm: bad arguments");
  hipLaunchKernelGGL(r3d_sqnorm_kernel, dim3(r3d_cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     ldx, (int)rows, C, out);
  R3D_LAUNCH_CHECK("r3d_sqnorm");
  return R3D_OK;
}

// x: (B*N, ldx) point-major fp32; norm_ws: (B*N) fp32 scratch; idx_out: (B, N, k) int32;
// score_out: optional (B, N, k) fp32; n_valid_dev: optional device int, rows >= *n are
// neither queried nor offered as candidates (used by the head where the node count
// is data dependent and stays on the device).
extern "C" int r3d_knn_topk(const float* x, long ldx, int B, int N, int C, int k, int mode,
                            const int* n_valid_dev, float* norm_ws, int32_t* idx_out,
                            float* score_out, void* stream) {
  R3D_REQUIRE(x && norm_ws && idx_out, "r3d_knn_topk: null pointer");
  R3D_REQUIRE(B > 0 && N > 0 && C > 0 && ldx >= C, "r3d_knn_topk: bad shape B=%d N=%d C=%d ldx=%ld", B, N, C, ldx);
  R3D_REQUIRE(k > 0 && k <= N && k <= 256, "r3d_knn_topk: unsupported k=%d (need 1..min(N,256))", k);
  R3D_REQUIRE(mode == R3D_SCORE_DGCNN || mode == R3D_SCORE_L2, "r3d_knn_topk: unknown mode %d", mode);
  const size_t lds = knn_lds_bytes(C);
  R3D_REQUIRE(lds <= 160 * 1024, "r3d_knn_topk: C=%d needs %zu B of LDS (> 160 KiB)", C, lds);
  int rc = r3d_sqnorm(x, ldx, (long)B * N, C, norm_ws, stream);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (k <= 32 && C <= 64) {
    dim3 g2(r3d_cdiv(N, 128), B);
    if (C <= 16)
      hipLaunchKernelGGL(r3d_knn_small_kernel<16>, g2, dim3(256), 0, st, x, ldx, N, C, k, mode, n_valid_dev, norm_ws,
                         idx_out, score_out);
    else
      hipLaunchKernelGGL(r3d_knn_small_kernel<64>, g2, dim3(256), 0, st, x, ldx, N, C, k, mode, n_valid_dev, norm_ws,
                         idx_out, score_out);
    R3D_LAUNCH_CHECK("r3d_knn_topk(small)");
    return R3D_OK;
  }
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
                       n_valid_dev, norm_ws, idx_out, score_out);                                \
  } while (0)
  if (k <= 64) KNN_LAUNCH(1);
  else if (k <= 128) KNN_LAUNCH(2);
  else KNN_LAUNCH(4);
#undef KNN_LAUNCH
  R3D_LAUNCH_CHECK("r3d_knn_topk");
  return R3D_OK;
}
