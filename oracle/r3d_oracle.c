/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the index-producing ops of the
 * R3DFSSeg hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path never does.
 *
 * Every distance below is defined as ONE fixed fp32 accumulation order: a
 * channel-ascending fmaf chain, acc = fmaf(a_c, b_c, acc), acc0 = 0.  That is
 * bitwise what gfx950's v_mfma_f32_32x32x2_f32 computes (k-ordered fmaf chain)
 * and what a VALU __builtin_fmaf loop computes, so the HIP kernels can match
 * these indices bit for bit.  The reference's own order (an MKL/cuBLAS GEMM,
 * faiss' sgemm, torch_cluster's reduction) is not reproducible on any other
 * machine; see DESIGN.md "parity pinning".
 *
 * Build: gcc -O2 -mfma -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 * -ffp-contract=off: nothing but the explicit fmaf calls may fuse.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- a1: DGCNN kNN, reference models/dgcnn.py:17-23 ----------------------
 * inner = -2 * x^T x ; xx = sum_c x_c^2 ; pd[i][j] = -xx[j] - inner[i][j] - xx[i]
 * idx = topk(pd, k) (largest first).  Ties: lowest index first.
 * x: (B, C, N) channel-major fp32.  idx_out: (B, N, k) int32, dist_out (optional)
 * the k selected pd values.                                                   */
typedef struct { float v; int32_t i; } vi_t;

static int cmp_desc(const void* a, const void* b) {
  const vi_t* p = (const vi_t*)a; const vi_t* q = (const vi_t*)b;
  if (p->v > q->v) return -1;
  if (p->v < q->v) return 1;
  return (p->i > q->i) - (p->i < q->i);
}
static int cmp_asc(const void* a, const void* b) {
  const vi_t* p = (const vi_t*)a; const vi_t* q = (const vi_t*)b;
  if (p->v < q->v) return -1;
  if (p->v > q->v) return 1;
  return (p->i > q->i) - (p->i < q->i);
}

int orc_knn_topk(const float* x, int B, int C, int N, int k, int32_t* idx_out,
                 float* dist_out) {
  if (k > N) return 1;
  float* xx = (float*)malloc(sizeof(float) * N);
  for (int b = 0; b < B; ++b) {
    const float* xb = x + (size_t)b * C * N;
    for (int j = 0; j < N; ++j) {
      float acc = 0.f;
      for (int c = 0; c < C; ++c) acc = fmaf(xb[(size_t)c * N + j], xb[(size_t)c * N + j], acc);
      xx[j] = acc;
    }
    /* rows are independent: the OpenMP split changes nothing in any result */
#pragma omp parallel
    {
      vi_t* row = (vi_t*)malloc(sizeof(vi_t) * N);
      float* xi = (float*)malloc(sizeof(float) * C);
#pragma omp for schedule(dynamic, 16)
      for (int i = 0; i < N; ++i) {
        for (int c = 0; c < C; ++c) xi[c] = xb[(size_t)c * N + i];
        for (int j = 0; j < N; ++j) {
          float dot = 0.f;
          for (int c = 0; c < C; ++c) dot = fmaf(xi[c], xb[(size_t)c * N + j], dot);
          float inner = -2.f * dot;
          float t = (-xx[j]) - inner;
          row[j].v = t - xx[i];
          row[j].i = j;
        }
        qsort(row, N, sizeof(vi_t), cmp_desc);
        for (int t = 0; t < k; ++t) {
          idx_out[((size_t)b * N + i) * k + t] = row[t].i;
          if (dist_out) dist_out[((size_t)b * N + i) * k + t] = row[t].v;
        }
      }
      free(row); free(xi);
    }
  }
  free(xx);
  return 0;
}

/* Gap between the k-th and (k+1)-th pd value per row (marks near-tie rows when
 * comparing against the reference's GEMM-ordered ranking).                    */
int orc_knn_gap(const float* x, int B, int C, int N, int k, float* gap_out) {
  if (k >= N) return 1;
  float* xx = (float*)malloc(sizeof(float) * N);
  vi_t* row = (vi_t*)malloc(sizeof(vi_t) * N);
  for (int b = 0; b < B; ++b) {
    const float* xb = x + (size_t)b * C * N;
    for (int j = 0; j < N; ++j) {
      float acc = 0.f;
      for (int c = 0; c < C; ++c) acc = fmaf(xb[(size_t)c * N + j], xb[(size_t)c * N + j], acc);
      xx[j] = acc;
    }
    for (int i = 0; i < N; ++i) {
      for (int j = 0; j < N; ++j) {
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot = fmaf(xb[(size_t)c * N + i], xb[(size_t)c * N + j], dot);
        row[j].v = ((-xx[j]) - (-2.f * dot)) - xx[i];
        row[j].i = j;
      }
      qsort(row, N, sizeof(vi_t), cmp_desc);
      gap_out[(size_t)b * N + i] = row[k - 1].v - row[k].v;
    }
  }
  free(xx); free(row);
  return 0;
}

/* ---- a11 (search half): exact k-NN in R^d, reference models/mpti.py:731-736
 * faiss.IndexFlatL2(d).add(X).search(X, k): faiss is not vendored (version
 * unpinned).  Restated from its published batched path (exhaustive_L2sqr_blas):
 * dis = ||x||^2 + ||y||^2 - 2 <x,y>, negative values clamped to 0, results
 * ascending.  Ties (faiss: heap order, unspecified): lowest index first.
 * X: (n, d) row-major.  idx_out (n, k) int32, dist_out optional.             */
int orc_knn_l2(const float* X, int n, int d, int k, int32_t* idx_out, float* dist_out) {
  if (k > n) return 1;
  float* nrm = (float*)malloc(sizeof(float) * n);
  for (int i = 0; i < n; ++i) {
    float acc = 0.f;
    for (int c = 0; c < d; ++c) acc = fmaf(X[(size_t)i * d + c], X[(size_t)i * d + c], acc);
    nrm[i] = acc;
  }
#pragma omp parallel
  {
    vi_t* prow = (vi_t*)malloc(sizeof(vi_t) * n);
#pragma omp for schedule(dynamic, 16)
    for (int i = 0; i < n; ++i) {
      const float* xi = X + (size_t)i * d;
      for (int j = 0; j < n; ++j) {
        const float* xj = X + (size_t)j * d;
        float ip = 0.f;
        for (int c = 0; c < d; ++c) ip = fmaf(xi[c], xj[c], ip);
        float dis = (nrm[i] + nrm[j]) - 2.f * ip;
        if (dis < 0.f) dis = 0.f;
        prow[j].v = dis;
        prow[j].i = j;
      }
      qsort(prow, n, sizeof(vi_t), cmp_asc);
      for (int t = 0; t < k; ++t) {
        idx_out[(size_t)i * k + t] = prow[t].i;
        if (dist_out) dist_out[(size_t)i * k + t] = prow[t].v;
      }
    }
    free(prow);
  }
  free(nrm);
  return 0;
}

/* ---- a9 (sampling half): farthest point sampling, reference
 * models/mpti.py:613 torch_cluster.fps(feat, None, ratio=k/n, random_start=False)
 * torch_cluster is not vendored (version unpinned).  Restated from its published
 * algorithm: start at index 0; dist[i] = min(dist[i], ||x_i - x_last||^2);
 * next = argmax dist.  Here: squared distance = fmaf chain over (x_c - s_c),
 * argmax ties -> lowest index, exactly k samples (caller guarantees k < n).
 * feat: (n, d) row-major.  out: k indices in selection order.                 */
int orc_fps(const float* feat, int n, int d, int k, int32_t* out) {
  if (k > n || n <= 0) return 1;
  float* mind = (float*)malloc(sizeof(float) * n);
  for (int i = 0; i < n; ++i) mind[i] = INFINITY;
  int last = 0;
  out[0] = 0;
  for (int t = 1; t < k; ++t) {
    const float* s = feat + (size_t)last * d;
    int best = 0; float bestv = -1.f;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
      const float* xi = feat + (size_t)i * d;
      float acc = 0.f;
      for (int c = 0; c < d; ++c) { float df = xi[c] - s[c]; acc = fmaf(df, df, acc); }
      if (acc < mind[i]) mind[i] = acc;
    }
    for (int i = 0; i < n; ++i)
      if (mind[i] > bestv) { bestv = mind[i]; best = i; }
    out[t] = best;
    last = best;
  }
  free(mind);
  return 0;
}

/* ---- a9 (assignment half): reference models/mpti.py:618-622
 * distances = F.pairwise_distance(feat[...,None], seeds.T[None,...], p=2)
 * In the reference's pinned environment (pytorch 1.8, README.md:14-15)
 * pairwise_distance is norm(x1 - x2 + 1e-6, p, dim=1): the reduction runs over
 * the feature axis and eps is added to every difference.  assignments =
 * argmin over seeds (first minimum).
 * dist = sqrtf( fmaf chain of ((x_c - s_c) + 1e-6f)^2 ).                      */
int orc_assign(const float* feat, int n, int d, const float* seeds, int m, int32_t* assign) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    const float* xi = feat + (size_t)i * d;
    int best = 0; float bestv = INFINITY;
    for (int s = 0; s < m; ++s) {
      const float* sv = seeds + (size_t)s * d;
      float acc = 0.f;
      for (int c = 0; c < d; ++c) { float df = (xi[c] - sv[c]) + 1e-6f; acc = fmaf(df, df, acc); }
      float dist = sqrtf(acc);
      if (dist < bestv) { bestv = dist; best = s; }
    }
    assign[i] = best;
  }
  return 0;
}

/* ---- a11 (weight half): reference models/mpti.py:745-746
 * dist = pairwise_distance(node_feat[:,:,None], knn_feat^T) (same 1.8 semantics)
 * w = exp(-0.5 * (dist / sigma)^2).  Returns dist only (exp is taken in the
 * python restatement so its libm is the one torch would use).                 */
int orc_pair_dist(const float* X, int n, int d, const int32_t* nbr, int k, float* dist_out) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    const float* xi = X + (size_t)i * d;
    for (int t = 0; t < k; ++t) {
      const float* xj = X + (size_t)nbr[(size_t)i * k + t] * d;
      float acc = 0.f;
      for (int c = 0; c < d; ++c) { float df = (xi[c] - xj[c]) + 1e-6f; acc = fmaf(df, df, acc); }
      dist_out[(size_t)i * k + t] = sqrtf(acc);
    }
  }
  return 0;
}
