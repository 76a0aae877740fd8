// Multi-prototype extraction of the transductive head, gfx950.
//
// Replaces (reference): models/mpti.py:636-715 getForeground/BackgroundPrototypes and
// :597-634 getMutiplePrototypes (torch_cluster.fps + nearest-seed argmin + python loop
// of masked means).  Everything stays on the device: the data-dependent counts
// (points per class, prototypes per class, graph nodes) live in a small descriptor in
// device memory and kernels are launched on capacity-sized grids.
//
// Segments: 0 = background (all support clouds, mask == 0), 1 + w = foreground of way w
// (mask == 1, optionally restricted to shots kept by clean-shot detection).  Compacted
// point lists sit at fixed capacity offsets: seg 0 at 0 (capacity S*N), seg 1+w at
// S*N + w*k_shot*N (capacity k_shot*N).  A "position" is an index inside a segment's
// list; list order is the reference's torch.nonzero order, so ties resolve the same way.
//
// Bit-exactness contract with oracle/r3d_oracle.c: FPS and assignment distances are
// channel-ascending fmaf chains evaluated by ONE thread per point.
#include "common.h"

#define HP_MAXSEG 8
#define HP_BLOCK 256
#define HP_MAXK 128
#define HP_DP 256      // row pitch (floats) of the compacted point-major copy

// int32 words of the device-side descriptor
enum {
  HD_SEG_COUNT = 0,                     // [HP_MAXSEG] points per segment
  HD_SEG_M = HP_MAXSEG,                 // [HP_MAXSEG] prototypes per segment
  HD_SEG_POFF = 2 * HP_MAXSEG,          // [HP_MAXSEG] first node row of the segment's prototypes
  HD_N_PROTO = 3 * HP_MAXSEG,
  HD_N_NODES = 3 * HP_MAXSEG + 1,
  HD_FPS_TIMEOUT = 3 * HP_MAXSEG + 2,   // != 0: a workgroup of the one-launch FPS gave up waiting for its peers
  HD_WORDS = 3 * HP_MAXSEG + 8
};

// Batches of episodes (round 3): every kernel below takes the episode from a grid dimension and shifts its per-episode
// pointers by the strides of HpEp (elements of each array between consecutive episodes; all capacity sized, so episode e
// of a batch lives at base + e * stride).  One episode = strides unused = the ABI-version-2 entry points.
struct HpEp {
  long sy, keep, feat, qfeat;                   // support_y, shot_keep, support feature ROWS, query feature ROWS
  long nodes, labels, desc, assign, ccount, ws;  // node rows, label rows, descriptor words, assign words, counts, scratch words
};
#define HP_SHIFT(p, stride) (p) += (long)ep * (stride)

struct SegGeom {
  int n_way, k_shot, N;
  __host__ __device__ int nseg() const { return n_way + 1; }
  __host__ __device__ long cap(int s) const { return s == 0 ? (long)n_way * k_shot * N : (long)k_shot * N; }
  __host__ __device__ long off(int s) const {
    return s == 0 ? 0 : (long)n_way * k_shot * N + (long)(s - 1) * k_shot * N;
  }
  __host__ __device__ long total_cap() const { return 2L * n_way * k_shot * N; }
  __host__ __device__ int blocks(int s) const { return (int)((cap(s) + HP_BLOCK - 1) / HP_BLOCK); }
  __host__ __device__ int block0(int s) const {
    int b = 0;
    for (int i = 0; i < s; ++i) b += blocks(i);
    return b;
  }
  __host__ __device__ int total_blocks() const { return block0(nseg()); }
  // segment of a block index
  __device__ int seg_of_block(int blk, int* first) const {
    int b = 0;
    for (int s = 0; s < nseg(); ++s) {
      int nb = blocks(s);
      if (blk < b + nb) { *first = b; return s; }
      b += nb;
    }
    *first = b;
    return -1;
  }
};

// ---------------------------------------------------------------------------
// 1. mask compaction (stable): one workgroup per segment
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void r3d_head_compact_kernel(const int* __restrict__ support_y,
                                                                const int* __restrict__ shot_keep,
                                                                SegGeom g, int* __restrict__ comp,
                                                                int* __restrict__ desc, HpEp st) {
  __shared__ int wave_tot[16];
  __shared__ int base_s;
  const int seg = blockIdx.x, ep = blockIdx.y;
  HP_SHIFT(support_y, st.sy); HP_SHIFT(comp, st.ws); HP_SHIFT(desc, st.desc);
  if (shot_keep) HP_SHIFT(shot_keep, st.keep);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long first = seg == 0 ? 0 : (long)(seg - 1) * g.k_shot * g.N;
  const long count = g.cap(seg);
  int* dst = comp + g.off(seg);
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (long t0 = 0; t0 < count; t0 += 1024) {
    const long e = t0 + tid;
    bool f = false;
    long gp = 0;
    if (e < count) {
      gp = first + e;
      const int y = support_y[gp];
      if (seg == 0) f = (y == 0);
      else {
        f = (y == 1);
        if (f && shot_keep) f = shot_keep[gp / g.N] != 0;
      }
    }
    const unsigned long long m = __ballot(f);
    const int pre = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[w] = __popcll(m);
    __syncthreads();
    int wbase = 0, tot = 0;
    for (int i = 0; i < 16; ++i) {
      const int v = wave_tot[i];
      if (i < w) wbase += v;
      tot += v;
    }
    const int base = base_s;
    if (f) dst[base + wbase + pre] = (int)gp;
    __syncthreads();
    if (tid == 0) base_s = base + tot;
    __syncthreads();
  }
  if (tid == 0) desc[HD_SEG_COUNT + seg] = base_s;
  if (tid == 0 && seg == 0) desc[HD_FPS_TIMEOUT] = 0;
}

// ---------------------------------------------------------------------------
// 1b. compacted channel-major copy of the listed points: featC[c][off(seg) + pos].  FPS and the
//     assignment then address their point by list position alone (no index indirection in the
//     per-round dependency chain) and read perfectly coalesced.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(HP_BLOCK) void r3d_head_gather_kernel(const float* __restrict__ feat /* (S*N, ldf) */,
                                                                   long ldf, int D, SegGeom g,
                                                                   const int* __restrict__ comp,
                                                                   const int* __restrict__ desc,
                                                                   float* __restrict__ featC, long pitch,
                                                                   float* __restrict__ featP /* [pos][HP_DP] */, HpEp st) {
  __shared__ float t[64][65];
  const int ep = blockIdx.y;
  feat += (long)ep * st.feat * ldf; HP_SHIFT(comp, st.ws); HP_SHIFT(desc, st.desc); HP_SHIFT(featC, st.ws);
  HP_SHIFT(featP, st.ws);
  int blk0;
  const int seg = g.seg_of_block(blockIdx.x, &blk0);
  const int count = desc[HD_SEG_COUNT + seg];
  const int bis = blockIdx.x - blk0;
  if ((long)bis * HP_BLOCK >= count) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // 256 list positions per block, 64 at a time: rows in (coalesced 64-float pieces), columns out
  for (int p0 = 0; p0 < HP_BLOCK; p0 += 64) {
    const int posb = bis * HP_BLOCK + p0;
    if (posb >= count) break;
    // the 16 list entries of this wave once per 64 positions, then per channel block every row load issued before the
    // first LDS store (a loop of one dependent load per trip is a chain of round trips on this small grid)
    int gp[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) gp[i] = comp[g.off(seg) + min(posb + w + 4 * i, count - 1)];
    for (int c0 = 0; c0 < D; c0 += 64) {
      __syncthreads();
      const int c = min(c0 + lane, D - 1);
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = feat[(long)gp[i] * ldf + c];
#pragma unroll
      for (int i = 0; i < 16; ++i) t[w + 4 * i][lane] = v[i];
      __syncthreads();
      for (int r = w; r < 64; r += 4) {
        const int c = c0 + r;
        const int pos = posb + lane;
        if (c < D && pos < count) featC[(long)c * pitch + g.off(seg) + pos] = t[lane][r];
      }
      // ... and the same rows compacted but still point-major: the FPS seed of a round is ONE row, read by every
      // workgroup of its segment (6 cache lines here against one line per channel from the channel-major copy)
      for (int r = w; r < 64; r += 4) {
        const int pos = posb + r;
        if (c0 + lane < D && pos < count) featP[(g.off(seg) + pos) * HP_DP + c0 + lane] = t[r][lane];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// 2. farthest point sampling, one launch per round, all segments at once
// ---------------------------------------------------------------------------
struct Cand { float v; int pos; };

static __device__ __forceinline__ void cand_better(float& v, int& p, float v2, int p2) {
  if (v2 > v || (v2 == v && p2 < p)) { v = v2; p = p2; }
}

// Samples torch_cluster.fps draws from n points at ratio = k / n (models/mpti.py:612-613; n > k): the published
// implementation (csrc/cpu/fps_cpu.cpp, csrc/cuda/fps_cuda.cu) takes ceil(float32(n) * float32(ratio)), ratio = the
// Python double k / n rounded to float32 -- k or k + 1 depending on how the two roundings fall (k = 100: 101 samples
// for 5.8 % of the n <= 20480).  Restated with the same roundings; tests/golden/head_*.npz hold both cases as the
// reference's own forward produced them (oracle/gen_golden_head.py).
static __device__ __forceinline__ int hp_fps_count(int n, int k) {
  const float ratio = __double2float_rn(__ddiv_rn((double)k, (double)n));
  const int m = (int)ceilf(__fmul_rn((float)n, ratio));
  return m < n ? m : n;
}

// `rounds` = k + 1: every sampled segment runs the extra round; r3d_fps_finalize_kernel keeps hp_fps_count of them.
template <int DP>  // feature dimension rounded up to a multiple of 64 (registers hold the whole point)
__global__ __launch_bounds__(HP_BLOCK) void r3d_fps_round_kernel(
    const float* __restrict__ featC /* (D, pitch) compacted channel-major */, long pitch, int D, SegGeom g,
    const int* __restrict__ desc, int k, int rounds, int round,
    float* __restrict__ mind, const Cand* __restrict__ cand_prev, Cand* __restrict__ cand_next,
    int* __restrict__ sel /* [nseg][HP_MAXK] */, HpEp st) {
  __shared__ float seedf[DP];
  {
    const int ep = blockIdx.y;  // Cand = 2 words
    HP_SHIFT(featC, st.ws); HP_SHIFT(desc, st.desc); HP_SHIFT(mind, st.ws); HP_SHIFT(sel, st.ws);
    cand_prev = (const Cand*)((const int*)cand_prev + (long)ep * st.ws);
    cand_next = (Cand*)((int*)cand_next + (long)ep * st.ws);
  }
  __shared__ float red_v[4];
  __shared__ int red_p[4];
  __shared__ int seed_pos_s;
  int blk0;
  const int seg = g.seg_of_block(blockIdx.x, &blk0);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int count = desc[HD_SEG_COUNT + seg];
  if (count <= k) return;  // identity case (mpti.py:631-634): no sampling
  const int bis = blockIdx.x - blk0;  // block index inside the segment
  if ((long)bis * HP_BLOCK >= count) return;
  const int nblk = (count + HP_BLOCK - 1) / HP_BLOCK;
  // --- this thread's point: ALL its channel loads are issued first (they do not depend on the
  //     seed), so their latency hides behind the seed election below
  const int pos = bis * HP_BLOCK + tid;
  const bool have = pos < count && round < rounds - 1;
  float xv[DP];
  float md_old = INFINITY;
  {
    const float* fp = featC + g.off(seg) + min(pos, count - 1);
#pragma unroll
    for (int c = 0; c < DP; ++c) xv[c] = fp[(long)min(c, D - 1) * pitch];
    if (round > 0) md_old = mind[g.off(seg) + min(pos, count - 1)];
  }
  // --- seed of this round
  if (round == 0) {
    if (tid == 0) seed_pos_s = 0;
  } else {
    float v = -INFINITY;
    int p = 0x7fffffff;
    for (int i = tid; i < nblk; i += HP_BLOCK) {
      const Cand c = cand_prev[blk0 + i];
      cand_better(v, p, c.v, c.pos);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(v, o);
      const int p2 = __shfl_xor(p, o);
      cand_better(v, p, v2, p2);
    }
    if (lane == 0) { red_v[w] = v; red_p[w] = p; }
    __syncthreads();
    if (tid == 0) {
      for (int i = 1; i < 4; ++i) cand_better(v, p, red_v[i], red_p[i]);
      seed_pos_s = p;
    }
  }
  __syncthreads();
  const int seed_pos = seed_pos_s;
  if (bis == 0 && tid == 0) sel[seg * HP_MAXK + round] = seed_pos;
  if (round == rounds - 1) return;
  // --- stage the seed's feature vector
  for (int c = tid; c < D; c += HP_BLOCK) seedf[c] = featC[(long)c * pitch + g.off(seg) + seed_pos];
  __syncthreads();
  // --- distance update (channel-ascending fmaf chain) + block argmax
  float v = -INFINITY;
  int p = 0x7fffffff;
  {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < DP; ++c) {
      if (c < D) {
        const float df = xv[c] - seedf[c];
        acc = __builtin_fmaf(df, df, acc);
      }
    }
    const float md = acc < md_old ? acc : md_old;
    if (have) {
      mind[g.off(seg) + pos] = md;
      v = md;
      p = pos;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(v, o);
    const int p2 = __shfl_xor(p, o);
    cand_better(v, p, v2, p2);
  }
  __syncthreads();  // red_* reuse
  if (lane == 0) { red_v[w] = v; red_p[w] = p; }
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < 4; ++i) cand_better(v, p, red_v[i], red_p[i]);
    Cand c; c.v = v; c.pos = p;
    cand_next[blockIdx.x] = c;
  }
}

// ---------------------------------------------------------------------------
// 2b. the same sampling as ONE launch: every workgroup keeps its 256 points (all channels) in registers for
//     all k rounds instead of re-reading them every round (the per-round launches above fetch the whole
//     compacted feature matrix from the memory side each round: 100 x 16 MB per episode on the S workload).
//     Election of a round, per segment: every workgroup publishes its best candidate as ONE 64-bit word -- key =
//     distance bits (non-negative floats order like their bits) above the complemented position, so the largest key is
//     the largest distance and, among equal distances, the lowest position: exactly the scan order of the oracle; 0 =
//     not yet written (the words are zeroed before the launch) -- with a relaxed agent-scope atomic store.  The
//     segment's FIRST workgroup (the leader) reads the words of all its workgroups, takes the maximum and publishes it
//     as the round's result word; the other workgroups wait for that ONE word (one lane polling).  The words are the
//     only data exchanged and the accesses are performed at the memory side, so no fence / cache write-back is needed
//     (cf. profiles/r01_experiments.md: a fenced grid barrier costs 3-13 us).  Measured with tools/probe/fps_exchange.hip
//     (us per round, exchange only, 1 segment of 66 workgroups -> 6 segments of 66 resident together): every workgroup
//     reads all words (round 2) 2.4 -> 3.2; atomic maximum into one word + arrival counter 2.5 -> 6.6 (same-address
//     atomics serialise at the memory side); leader 2.0 -> 2.1.
//     The seed's feature row comes from the compacted point-major copy (6 cache lines, not 192).
//     Co-residency: all workgroups of a grid that hold points must be resident together (one workgroup waits
//     for its peers).  A workgroup needs ~200 VGPRs, i.e. 2 workgroups fit a CU, 512 on the chip; the caller
//     groups the episodes of a batch into launches that stay below that (fps_group).  A waiter that sees
//     nothing for ~2 s sets desc[HD_FPS_TIMEOUT] and proceeds, so a mis-sized launch ends with an error flag
//     instead of a hung GPU.
// ---------------------------------------------------------------------------
#define FPS_SPIN_LIMIT (1 << 22)
// FPS_SCALAR_SEED 1: the seed row read by scalar loads (s_load_dwordx16, SGPR operands) instead of staged in LDS: no
// staging barrier, but twelve dependent scalar round trips per round -- measured 6.09 against 4.94 ms for the whole
// r3d_head_prototypes_batched call on 32 episodes (tools/fps_group_bench.py).  Off.
#ifndef FPS_SCALAR_SEED
#define FPS_SCALAR_SEED 0
#endif
#ifdef FPS_STAMPS  // residency probe of the persistent kernel (tools/fps_stamps.py; never in the product build)
__device__ unsigned long long g_fps_dbg[8192 * 4];
extern "C" int r3d_fps_debug_read(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fps_dbg), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 1;
}
#define FSTAMP_BEGIN()                                                                                       \
  unsigned long long fs_t0 = 0;                                                                              \
  unsigned fs_hw = 0, fs_xcc = 0;                                                                            \
  if (threadIdx.x == 0) {                                                                                    \
    fs_t0 = __builtin_amdgcn_s_memtime();                                                                    \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(fs_hw));                                      \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(fs_xcc));                                    \
  }
#define FSTAMP_END(active)                                                                                   \
  if (threadIdx.x == 0) {                                                                                    \
    const long fs_i = ((long)blockIdx.y * gridDim.x + blockIdx.x) & 8191;                                    \
    g_fps_dbg[4 * fs_i + 0] = fs_t0;                                                                         \
    g_fps_dbg[4 * fs_i + 1] = __builtin_amdgcn_s_memtime();                                                  \
    g_fps_dbg[4 * fs_i + 2] = ((unsigned long long)fs_xcc << 32) | fs_hw;                                    \
    g_fps_dbg[4 * fs_i + 3] = (active);                                                                      \
  }
#else
#define FSTAMP_BEGIN()
#define FSTAMP_END(active)
#endif
// FULL: D == DP.  With a run-time D < DP the unrolled channel loop carries a uniform `c < D` test per channel, which the
// compiler turns into 192 precomputed lane masks parked in a VGPR: two v_readlane, a wait state and a v_cndmask per
// channel beside the sub and the fma -- 3.5 times the VALU work of the round (5 of its 8 us with two workgroups per CU).
// LC: the point's LAST LC channels live in LDS ([channel][thread]: conflict-free), the others in registers.  Measured
// for DP = 192, LC = 52 (168 registers: three workgroups per CU, 9 episodes of workload S per launch, four launches for
// a batch of 32 instead of six): 0.83 ms per launch against 0.47 -- a round is not only latency, the CU's waves share
// the LDS pipe for the seed row's broadcast reads (48 ds_read_b128 per wave and round) -- so LC = 0 is what runs
// (tools/fps_group_bench.py).  The distance is the same channel-ascending chain either way.
template <int DP, bool FULL, int LC>
__global__ __launch_bounds__(HP_BLOCK) __attribute__((amdgpu_waves_per_eu(LC ? 3 : 1))) void r3d_fps_persistent_kernel(
    const float* __restrict__ featC, long pitch, const float* __restrict__ featP, int D, SegGeom g, int* __restrict__ desc,
    int k, int rounds /* k + 1 */, unsigned long long* __restrict__ xch /* [rounds][tb_dense] candidates, then [nseg][HP_MAXK] results */, int tb_dense,
    int* __restrict__ sel, HpEp st, int ep0) {
  __shared__ float seedf[DP];
  __shared__ float red_v[4];
  __shared__ int red_p[4];
  __shared__ int seed_pos_s;
  __shared__ float xl[LC ? LC : 1][HP_BLOCK];
  constexpr int DR = DP - LC;  // channels in registers
  {
    const int ep = ep0 + blockIdx.y;  // the launch holds episodes ep0 .. ep0 + gridDim.y - 1 (co-resident together)
    HP_SHIFT(featC, st.ws); HP_SHIFT(featP, st.ws); HP_SHIFT(desc, st.desc); HP_SHIFT(sel, st.ws);
    xch = (unsigned long long*)((int*)xch + (long)ep * st.ws);  // st.ws is a multiple of 4 words: 8-byte alignment is kept
  }
  FSTAMP_BEGIN();
  // DENSE block -> (segment, block in segment) map: the grid has just the ceil(S N / 256) + nseg workgroups an episode
  // can need, and the ones that hold points are the first of them.  Workgroups go to the XCDs round robin in launch
  // order, so a dense range spreads evenly; the capacity-sized grid of the other kernels (blocks(seg) per segment, most
  // of them exiting at once) left the XCDs up to 71 : 55 unbalanced, a segment's last workgroups then waited for
  // another segment's to END, and a launch of six episodes took 2.6 times one episode's time (tools/fps_stamps.py).
  int seg = -1, bis = 0, count = 0, blk0 = 0;
  {
    int first = 0;
    for (int s = 0; s < g.nseg(); ++s) {
      const int cnt = desc[HD_SEG_COUNT + s];
      const int nb = cnt > k ? (cnt + HP_BLOCK - 1) / HP_BLOCK : 0;  // count <= k: identity case (mpti.py:631-634), no sampling
      if (seg < 0 && (int)blockIdx.x < first + nb) { seg = s; bis = blockIdx.x - first; count = cnt; blk0 = first; }
      first += nb;
    }
  }
  if (seg < 0) { FSTAMP_END(0); return; }
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned nblk = (unsigned)((count + HP_BLOCK - 1) / HP_BLOCK);
  const int pos = bis * HP_BLOCK + tid;
  const bool have = pos < count;
  float xv[DR];
  {
    const float* fp = featC + g.off(seg) + min(pos, count - 1);
#pragma unroll
    for (int c = 0; c < DR; ++c) xv[c] = fp[(long)min(c, D - 1) * pitch];
#pragma unroll
    for (int c = 0; c < LC; ++c) xl[c][tid] = fp[(long)min(DR + c, D - 1) * pitch];  // (read back by this thread alone)
  }
  const float* rows = featP + g.off(seg) * HP_DP;
  unsigned long long* res = xch + (long)rounds * tb_dense + (long)seg * HP_MAXK;  // the segment's result word of every round
  float md = INFINITY;
  int seed_pos = 0;
  for (int round = 0; round < rounds; ++round) {
    if (bis == 0 && tid == 0) sel[seg * HP_MAXK + round] = seed_pos;
    if (round == rounds - 1) break;
#if FPS_SCALAR_SEED
    __syncthreads();  // red_* of the previous round are no longer read
    const float* __restrict__ srow = rows + (long)__builtin_amdgcn_readfirstlane(seed_pos) * HP_DP;  // (uniform: scalar loads)
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < DR; ++c) {
      if (FULL || c < D) {
        const float df = xv[c] - srow[c];
        acc = __builtin_fmaf(df, df, acc);
      }
    }
#else
    __syncthreads();  // seedf / red_* of the previous round are no longer read
    for (int c = tid; c < D; c += HP_BLOCK) seedf[c] = rows[(long)seed_pos * HP_DP + c];
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < DR; ++c) {
      if (FULL || c < D) {
        const float df = xv[c] - seedf[c];
        acc = __builtin_fmaf(df, df, acc);
      }
    }
#endif
#pragma unroll
    for (int c = 0; c < LC; ++c) {
      if (FULL || DR + c < D) {
        const float df = xl[c][tid] - seedf[DR + c];
        acc = __builtin_fmaf(df, df, acc);
      }
    }
    md = acc < md ? acc : md;
    float v = have ? md : -INFINITY;
    int p = have ? pos : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(v, o);
      const int p2 = __shfl_xor(p, o);
      cand_better(v, p, v2, p2);
    }
    if (lane == 0) { red_v[w] = v; red_p[w] = p; }
    __syncthreads();
    unsigned long long* cw = xch + (long)round * tb_dense + blk0;  // the candidate words of the segment's workgroups
    if (tid == 0) {
      for (int i = 1; i < 4; ++i) cand_better(v, p, red_v[i], red_p[i]);
      // (a workgroup that holds points has a candidate with v >= 0; anything else must not win: key 1, never 0)
      const unsigned long long key =
          (v >= 0.f && (unsigned)p < (unsigned)count) ? (((unsigned long long)__float_as_uint(v) << 32) | (0xffffffffu - (unsigned)p)) : 1ull;
      __hip_atomic_store(&cw[bis], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned long long win = 0ull;
    if (bis == 0) {  // leader: the maximum of the segment's candidate words -> the round's result word
      for (unsigned i = tid; i < nblk; i += HP_BLOCK) {
        unsigned long long word = __hip_atomic_load(&cw[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (word == 0ull && spins < FPS_SPIN_LIMIT) {
          __builtin_amdgcn_s_sleep(1);
          word = __hip_atomic_load(&cw[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ++spins;
        }
        if (word == 0ull) desc[HD_FPS_TIMEOUT] = 1;
        win = word > win ? word : win;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long y = __shfl_xor(win, o);
        win = y > win ? y : win;
      }
      __syncthreads();  // red_* reuse
      if (lane == 0) { red_v[w] = __uint_as_float((unsigned)(win >> 32)); red_p[w] = (int)(unsigned)(win & 0xffffffffull); }
      __syncthreads();
      if (tid == 0) {
        for (int i = 1; i < 4; ++i) {
          const unsigned long long y = ((unsigned long long)__float_as_uint(red_v[i]) << 32) | (unsigned)red_p[i];
          win = y > win ? y : win;
        }
        __hip_atomic_store(&res[round], win | 1ull << 63, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // bit 63: written (distances are >= 0: the sign bit is free)
      }
    } else if (tid == 0) {
      win = __hip_atomic_load(&res[round], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (win == 0ull && spins < FPS_SPIN_LIMIT) {
        __builtin_amdgcn_s_sleep(1);
        win = __hip_atomic_load(&res[round], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++spins;
      }
      if (win == 0ull) desc[HD_FPS_TIMEOUT] = 1;
    }
    if (tid == 0) {
      const unsigned wp = 0xffffffffu - (unsigned)(win & 0xffffffffull);
      seed_pos_s = (win != 0ull && wp < (unsigned)count) ? (int)wp : 0;  // stays a valid position even after a timeout
    }
    __syncthreads();
    seed_pos = seed_pos_s;
  }
  FSTAMP_END(1);
}

// ---------------------------------------------------------------------------
// 3. finalize: sort + unique the sampled positions (torch .unique(), mpti.py:613),
//    node-row offsets (background first, then ways: mpti.py:493)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(HP_MAXK) void r3d_fps_finalize_kernel(SegGeom g, int k, int n_query_pts,
                                                                   const int* __restrict__ sel,
                                                                   int* __restrict__ seeds /*[nseg][HP_MAXK]*/,
                                                                   int* __restrict__ desc, HpEp st) {
  const int ep = blockIdx.x;
  HP_SHIFT(sel, st.ws); HP_SHIFT(seeds, st.ws); HP_SHIFT(desc, st.desc);
  __shared__ int a[HP_MAXK];
  __shared__ int m_s[HP_MAXSEG];
  const int tid = threadIdx.x;
  for (int seg = 0; seg < g.nseg(); ++seg) {
    const int count = desc[HD_SEG_COUNT + seg];
    int m;
    if (count <= k) {  // identity: every point is its own prototype
      m = count;
      if (tid < HP_MAXK) seeds[seg * HP_MAXK + tid] = tid < count ? tid : -1;
    } else {
      const int kseg = hp_fps_count(count, k);  // k or k + 1 samples (torch_cluster's float-rounded count)
      a[tid] = tid < kseg ? sel[seg * HP_MAXK + tid] : 0x7fffffff;
      __syncthreads();
      for (int sz = 2; sz <= HP_MAXK; sz <<= 1)
        for (int st = sz >> 1; st > 0; st >>= 1) {
          const int j = tid ^ st;
          if (j > tid) {
            const bool up = (tid & sz) == 0;
            const int x = a[tid], y = a[j];
            if ((x > y) == up) { a[tid] = y; a[j] = x; }
          }
          __syncthreads();
        }
      // unique on the sorted list
      const bool keep = tid < kseg && (tid == 0 || a[tid] != a[tid - 1]);
      __syncthreads();
      // rank = number of kept entries before tid (k <= 128: serial count is fine)
      __shared__ int keepf[HP_MAXK];
      keepf[tid] = keep ? 1 : 0;
      __syncthreads();
      int rank = 0;
      for (int i = 0; i < tid; ++i) rank += keepf[i];
      int total = 0;
      for (int i = 0; i < HP_MAXK; ++i) total += keepf[i];
      m = total;
      seeds[seg * HP_MAXK + tid] = -1;
      __syncthreads();
      if (keep) seeds[seg * HP_MAXK + rank] = a[tid];
    }
    if (tid == 0) m_s[seg] = m;
    __syncthreads();
  }
  if (tid == 0) {
    int off = 0;
    for (int seg = 0; seg < g.nseg(); ++seg) {
      desc[HD_SEG_M + seg] = m_s[seg];
      desc[HD_SEG_POFF + seg] = off;
      off += m_s[seg];
    }
    desc[HD_N_PROTO] = off;
    desc[HD_N_NODES] = off + n_query_pts;
  }
}

// ---------------------------------------------------------------------------
// 4. nearest-seed assignment (mpti.py:618-622): dist = sqrt(chain(((x - s) + 1e-6)^2))
// ---------------------------------------------------------------------------
// One workgroup = 256 points x ONE tile of 16 seeds (blockIdx.y): the seed axis is spread over the grid (160
// workgroups of one wave per SIMD took 170 us at S) and the tiles meet in a 64-bit minimum per point: distance bits
// above the seed position, so that the smallest distance wins and, among equal distances, the first seed -- what
// the ascending scan with `<` gave.  Distances are non-negative: their bit patterns order like the values.
#define AS_TILE 16
__global__ __launch_bounds__(HP_BLOCK) void r3d_assign_kernel(const float* __restrict__ featC, long pitch, int D,
                                                              SegGeom g, const int* __restrict__ desc,
                                                              const int* __restrict__ seeds,
                                                              unsigned long long* __restrict__ best_packed, HpEp st) {
  __shared__ float sf[256 * AS_TILE];  // [c][AS_TILE], D <= 256
  {
    const int ep = blockIdx.z;
    HP_SHIFT(featC, st.ws); HP_SHIFT(desc, st.desc); HP_SHIFT(seeds, st.ws);
    best_packed = (unsigned long long*)((int*)best_packed + (long)ep * st.ws);
  }
  int blk0;
  const int seg = g.seg_of_block(blockIdx.x, &blk0);
  const int tid = threadIdx.x;
  const int count = desc[HD_SEG_COUNT + seg];
  const int bis = blockIdx.x - blk0;
  if ((long)bis * HP_BLOCK >= count) return;
  const int m = desc[HD_SEG_M + seg];
  const int s0 = blockIdx.y * AS_TILE;
  if (s0 >= m) return;
  const int pos = bis * HP_BLOCK + tid;
  const bool ok = pos < count;
  const float* fp = featC + g.off(seg) + min(pos, count - 1);
  for (int e = tid; e < D * AS_TILE; e += HP_BLOCK) {
    const int c = e / AS_TILE, s = e - c * AS_TILE;
    const int sp = seeds[seg * HP_MAXK + min(s0 + s, m - 1)];
    sf[c * AS_TILE + s] = r3d_keep(featC[(long)c * pitch + g.off(seg) + sp], s0 + s < m);
  }
  __syncthreads();
  if (!ok) return;
  float acc[AS_TILE];
#pragma unroll
  for (int s = 0; s < AS_TILE; ++s) acc[s] = 0.f;
  int c = 0;
  for (; c + 8 <= D; c += 8) {  // 8 channel loads in flight; chains stay channel-ascending
    float xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[u] = fp[(long)(c + u) * pitch];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int s = 0; s < AS_TILE; ++s) {
        const float df = (xv[u] - sf[(c + u) * AS_TILE + s]) + 1e-6f;
        acc[s] = __builtin_fmaf(df, df, acc[s]);
      }
    }
  }
  for (; c < D; ++c) {
    const float xv = fp[(long)c * pitch];
#pragma unroll
    for (int s = 0; s < AS_TILE; ++s) {
      const float df = (xv - sf[c * AS_TILE + s]) + 1e-6f;
      acc[s] = __builtin_fmaf(df, df, acc[s]);
    }
  }
  float best = INFINITY;
  int besti = 0;
#pragma unroll
  for (int s = 0; s < AS_TILE; ++s) {
    if (s0 + s < m) {
      const float d = sqrtf(acc[s]);
      if (d < best) { best = d; besti = s0 + s; }
    }
  }
  atomicMin(&best_packed[g.off(seg) + pos], ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)besti);
}
__global__ void r3d_assign_unpack_kernel(const unsigned long long* __restrict__ best_packed, long n, int* __restrict__ assign,
                                         HpEp st) {
  const int ep = blockIdx.y;
  best_packed = (const unsigned long long*)((const int*)best_packed + (long)ep * st.ws);
  HP_SHIFT(assign, st.assign);
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) assign[i] = (int)(unsigned)(best_packed[i] & 0xffffffffull);
}

// ---------------------------------------------------------------------------
// 5. cluster means -> prototype rows of the node matrix (mpti.py:625-629)
//    5a  partial sums per (segment, prototype, chunk of CM_CHUNK list positions)
//    5b  chunks added in ascending order: a fixed summation order, whatever the cluster sizes
//        (FPS clusters are very uneven: one workgroup per cluster would serialise on the
//        largest one)
// ---------------------------------------------------------------------------
#define CM_CHUNK 2048
__global__ __launch_bounds__(HP_BLOCK) void r3d_cluster_partial_kernel(
    const float* __restrict__ feat /* (S*N, ldf) point-major */, long ldf, int D, SegGeom g,
    const int* __restrict__ comp, const int* __restrict__ desc, const int* __restrict__ assign, int max_chunks,
    float* __restrict__ part /* [nseg][HP_MAXK][max_chunks][256] */, int* __restrict__ part_cnt, HpEp st) {
  __shared__ float psum[4][256];
  __shared__ int cnt_s[4];
  const int ep = blockIdx.z / g.nseg();
  const int seg = blockIdx.z - ep * g.nseg(), s = blockIdx.x, chunk = blockIdx.y;
  feat += (long)ep * st.feat * ldf; HP_SHIFT(comp, st.ws); HP_SHIFT(desc, st.desc); HP_SHIFT(assign, st.assign);
  HP_SHIFT(part, st.ws); HP_SHIFT(part_cnt, st.ws);
  const int m = desc[HD_SEG_M + seg];
  const int count = desc[HD_SEG_COUNT + seg];
  if (s >= m || (long)chunk * CM_CHUNK >= count) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int cnt = 0;
  const int* cp = comp + g.off(seg);
  const int* ap = assign + g.off(seg);
  const int end = min(count, (chunk + 1) * CM_CHUNK);
  for (int base = chunk * CM_CHUNK + 64 * w; base < end; base += 256) {
    const int pos = base + lane;
    const bool f = pos < end && ap[min(pos, end - 1)] == s;
    const int gp = cp[min(pos, end - 1)];
    unsigned long long mm = __ballot(f);
    cnt += __popcll(mm);
    // the wave walks the matches in ascending position (fixed summation order), 8 feature rows in flight per
    // step: one row at a time was a chain of L2 round trips, 200 us for a cluster holding most of a chunk
    const int c0 = min(lane, D - 1), c1 = min(lane + 64, D - 1), c2 = min(lane + 128, D - 1), c3 = min(lane + 192, D - 1);
    while (mm) {
      float v0[8], v1[8], v2[8], v3[8];
      bool on[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        on[u] = mm != 0ull;  // wave-uniform
        const int src = on[u] ? __ffsll((long long)mm) - 1 : 0;
        mm &= mm - 1;        // 0 stays 0
        const float* fr = feat + (long)__builtin_amdgcn_readlane(gp, src) * ldf;
        v0[u] = fr[c0]; v1[u] = fr[c1]; v2[u] = fr[c2]; v3[u] = fr[c3];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (on[u]) {
          if (lane < D) a0 += v0[u];
          if (lane + 64 < D) a1 += v1[u];
          if (lane + 128 < D) a2 += v2[u];
          if (lane + 192 < D) a3 += v3[u];
        }
      }
    }
  }
  psum[w][lane] = a0; psum[w][lane + 64] = a1; psum[w][lane + 128] = a2; psum[w][lane + 192] = a3;
  if (lane == 0) cnt_s[w] = cnt;
  __syncthreads();
  const long slot = ((long)(seg * HP_MAXK + s) * max_chunks + chunk);
  part[slot * 256 + tid] = ((psum[0][tid] + psum[1][tid]) + psum[2][tid]) + psum[3][tid];
  if (tid == 0) part_cnt[slot] = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3];
}

__global__ __launch_bounds__(HP_BLOCK) void r3d_cluster_reduce_kernel(
    int D, const int* __restrict__ desc, int max_chunks, const float* __restrict__ part,
    const int* __restrict__ part_cnt, float* __restrict__ nodes /* (n_cap, ldn) */, long ldn,
    float* __restrict__ node_labels /* (n_cap, 4) */, float* __restrict__ node_labels2 /* classes 4..7, or null */,
    int* __restrict__ cluster_count, HpEp st) {
  const int seg = blockIdx.y, s = blockIdx.x, ep = blockIdx.z;
  HP_SHIFT(desc, st.desc); HP_SHIFT(part, st.ws); HP_SHIFT(part_cnt, st.ws);
  nodes += (long)ep * st.nodes * ldn; node_labels += (long)ep * st.labels * 4;
  if (node_labels2) node_labels2 += (long)ep * st.labels * 4;
  if (cluster_count) HP_SHIFT(cluster_count, st.ccount);
  const int m = desc[HD_SEG_M + seg];
  if (s >= m) return;
  const int count = desc[HD_SEG_COUNT + seg];
  const int nchunks = (count + CM_CHUNK - 1) / CM_CHUNK;
  const int tid = threadIdx.x;
  float sum = 0.f;
  int total = 0;
  for (int c = 0; c < nchunks; ++c) {
    const long slot = ((long)(seg * HP_MAXK + s) * max_chunks + c);
    sum += part[slot * 256 + tid];
    total += part_cnt[slot];
  }
  const int row = desc[HD_SEG_POFF + seg] + s;
  if (tid < D) nodes[(long)row * ldn + tid] = sum / (float)total;  // 0/0 = NaN for an empty cluster, as torch
  if (tid < 4) node_labels[(long)row * 4 + tid] = (tid == seg) ? 1.f : 0.f;
  if (tid < 4 && node_labels2) node_labels2[(long)row * 4 + tid] = (tid + 4 == seg) ? 1.f : 0.f;
  if (tid == 0 && cluster_count) cluster_count[row] = total;
}

// ---------------------------------------------------------------------------
// 6. query rows of the node matrix (mpti.py:508: node_feat = cat(prototypes, query_feat))
// ---------------------------------------------------------------------------
__global__ void r3d_nodes_append_query_kernel(const float* __restrict__ qfeat, long ldq, int D, int nq_pts,
                                              const int* __restrict__ desc, float* __restrict__ nodes,
                                              long ldn, float* __restrict__ node_labels,
                                              float* __restrict__ node_labels2, HpEp st) {
  const int ep = blockIdx.y;
  qfeat += (long)ep * st.qfeat * ldq; HP_SHIFT(desc, st.desc);
  nodes += (long)ep * st.nodes * ldn; node_labels += (long)ep * st.labels * 4;
  if (node_labels2) node_labels2 += (long)ep * st.labels * 4;
  const int n_proto = desc[HD_N_PROTO];
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nq_pts * D) return;
  const long r = (long)((unsigned)i / (unsigned)D);  // (n_query_pts D < 2^31, checked by the entry point: a 32-bit division)
  const int c = (int)(i - r * D);
  nodes[(n_proto + r) * ldn + c] = qfeat[r * ldq + c];
  if (c < 4) node_labels[(n_proto + r) * 4 + c] = 0.f;
  if (c < 4 && node_labels2) node_labels2[(n_proto + r) * 4 + c] = 0.f;
}

// ===========================================================================
// C ABI
// ===========================================================================
static int check_geom(const char* fn, int n_way, int k_shot, int N, int D) {
  if (n_way < 1 || n_way + 1 > HP_MAXSEG || k_shot < 1 || N < 1 || D < 1 || D > 256) {
    r3d_set_error("%s: unsupported geometry n_way=%d k_shot=%d N=%d D=%d (n_way<=7, D<=256)", fn, n_way,
                  k_shot, N, D);
    return R3D_ERR_ARG;
  }
  if ((long)n_way * k_shot * N > 65536L * 4) {
    r3d_set_error("%s: too many support points", fn);
    return R3D_ERR_ARG;
  }
  return 0;
}
static int check_query_rows(const char* fn, int n_query_pts, int D) {  // (the query-row kernels index elements in 32 bits)
  if (n_query_pts < 0 || (long)n_query_pts * D >= 0x7fffffffL) {
    r3d_set_error("%s: %d query points x %d channels do not fit 31 bits", fn, n_query_pts, D);
    return R3D_ERR_ARG;
  }
  return 0;
}

extern "C" int r3d_head_desc_words(void) { return HD_WORDS; }
extern "C" int r3d_head_max_k(void) { return HP_MAXK; }

// out[n] = the seeds a segment of n points gets at k (n in [0, n_max)): the device's evaluation of hp_fps_count, so that
// a test can hold it to numpy's float32 arithmetic for EVERY point count (tests/test_gpu_head.py)
__global__ void r3d_fps_count_table_kernel(int k, int n_max, int* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < n_max) out[n] = n > k ? hp_fps_count(n, k) : n;
}
extern "C" int r3d_fps_sample_count_table(int k, int n_max, int32_t* out, void* stream) {
  R3D_REQUIRE(out && k >= 1 && n_max >= 1, "r3d_fps_sample_count_table: bad arguments");
  hipLaunchKernelGGL(r3d_fps_count_table_kernel, dim3(r3d_cdiv(n_max, 256)), dim3(256), 0, (hipStream_t)stream, k, n_max, out);
  R3D_LAUNCH_CHECK("r3d_fps_sample_count_table");
  return R3D_OK;
}

// Scratch layout (4-byte words) of r3d_head_prototypes.
struct HpWs {
  long comp, mind, assign, cand, sel, seeds, part, part_cnt, featC, xch, best, featP, total;
  long pitch;
  int max_chunks;
};
static HpWs hp_carve(const SegGeom& g) {
  HpWs L;
  const long cap = g.total_cap();
  L.max_chunks = (int)((g.cap(0) + CM_CHUNK - 1) / CM_CHUNK);
  L.pitch = cap + 32;
  long o = 0;
  L.comp = o; o += cap;
  L.mind = o; o += cap;
  L.assign = o; o += cap;
  L.cand = o; o += 4L * g.total_blocks();                          // 2 x blocks x (value, position)
  L.sel = o; o += (long)HP_MAXSEG * HP_MAXK;
  L.seeds = o; o += (long)HP_MAXSEG * HP_MAXK + 64;
  L.part = o; o += (long)g.nseg() * HP_MAXK * L.max_chunks * 256;  // cluster partial sums
  L.part_cnt = o; o += (long)g.nseg() * HP_MAXK * L.max_chunks;
  L.featC = o; o += 256L * (cap + 64);                             // compacted channel-major copy (D <= 256 rows)
  o = (o + 3) & ~3L;
  // one-launch FPS: a candidate word per (round, workgroup) and a result word per (segment, round), 64 bit each
  L.xch = o; o += 2L * HP_MAXK * ((g.cap(0) + HP_BLOCK - 1) / HP_BLOCK + g.nseg()) + 2L * HP_MAXSEG * HP_MAXK;
  L.best = o; o += 2L * cap + 2;                                   // 64-bit (distance, seed) minimum per point
  o = (o + 3) & ~3L;
  L.featP = o; o += cap * HP_DP;                                   // compacted point-major copy
  L.total = o + 8;
  return L;
}
extern "C" long r3d_head_proto_ws_words(int n_way, int k_shot, int N) { return hp_carve(SegGeom{n_way, k_shot, N}).total; }

// Builds prototypes into node rows [0, n_proto) and appends the query rows, for n_ep episodes at once.
//   support_y : (n_way*k_shot, N) int32 {0,1}
//   shot_keep : optional (n_way*k_shot) int32, 0 drops a shot's foreground (clean-shot detection)
//   feat      : (S*N, ldf) point-major support features;  qfeat: (n_q*N, ldq) point-major query features
//   nodes     : (n_cap, ldn) out, n_cap = (n_way+1)*(k+1) + n_q*N;  node_labels: (n_cap, 4) one-hot Y; n_way > 3: two such
//               planes, (2, n_ep * n_cap, 4), classes 0..3 and 4..7
//   desc      : device descriptor (r3d_head_desc_words int32);  ws: scratch words
// Every pointer addresses episode 0; episode e sits ep->... elements further on (HpEp; feature strides in ROWS).
// fps_group: episodes whose farthest-point samplings share ONE persistent launch (flags & R3D_HEAD_FPS_ONE_LAUNCH): all
// their workgroups that hold points must be co-resident, so the caller sizes it to the chip (~500 workgroup slots at
// D <= 192, 250 above; an episode needs ceil(S N / 256) + n_way + 1); the batch takes ceil(n_ep / fps_group) such launches.
// Workgroups of the persistent FPS kernel the chip holds at once (occupancy of the instantiation x CUs; asked once per
// shape class) -> episodes per launch: what the caller asked for, clamped to that.
static int fps_slots_clamp(int D, int tb_dense, int fps_group) {
  static int slots[4] = {0, 0, 0, 0};
  const int cls = D <= 64 ? 0 : D <= 128 ? 1 : D <= 192 ? 2 : 3;
  if (!slots[cls]) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    hipError_t e = cls == 0   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, r3d_fps_persistent_kernel<64, true, 0>, HP_BLOCK, 0)
                   : cls == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, r3d_fps_persistent_kernel<128, true, 0>, HP_BLOCK, 0)
                   : cls == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, r3d_fps_persistent_kernel<192, true, 0>, HP_BLOCK, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, r3d_fps_persistent_kernel<256, true, 0>, HP_BLOCK, 0);
    if (e == hipSuccess && per_cu > 0 && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      slots[cls] = per_cu * prop.multiProcessorCount;
    else {
      (void)hipGetLastError();
      slots[cls] = 256;  // one workgroup per CU: always resident
    }
  }
  const int fit = slots[cls] / tb_dense;
  return fps_group < 1 ? 1 : (fit < 1 ? 1 : (fps_group < fit ? fps_group : fit));
}

static int head_prototypes_impl(int n_ep, const HpEp& ep, int fps_group, const int32_t* support_y, const int32_t* shot_keep,
                                const float* feat, long ldf, const float* qfeat, long ldq, int n_way, int k_shot, int N, int D,
                                int n_query_pts, int k, float* nodes, long ldn, float* node_labels, int32_t* desc,
                                int32_t* assign_out, int32_t* cluster_count, int32_t* ws, long ws_words, int flags,
                                void* stream) {
  R3D_REQUIRE(support_y && feat && qfeat && nodes && node_labels && desc && ws, "r3d_head_prototypes: null pointer");
  int rc = check_geom("r3d_head_prototypes", n_way, k_shot, N, D);
  if (rc) return rc;
  rc = check_query_rows("r3d_head_prototypes", n_query_pts, D);
  if (rc) return rc;
  R3D_REQUIRE(ws_words >= r3d_head_proto_ws_words(n_way, k_shot, N),
              "r3d_head_prototypes: workspace of %ld words, r3d_head_proto_ws_words = %ld needed", ws_words,
              r3d_head_proto_ws_words(n_way, k_shot, N));
  R3D_REQUIRE(k >= 1 && k < HP_MAXK, "r3d_head_prototypes: k=%d unsupported (1..%d)", k, HP_MAXK - 1);
  const int kr = k + 1;  // sampling rounds run and prototype slots per segment (hp_fps_count: k or k + 1 samples)
  R3D_REQUIRE(n_ep >= 1 && n_ep <= 4096 && fps_group >= 1, "r3d_head_prototypes: %d episodes, %d per FPS launch", n_ep, fps_group);
  R3D_REQUIRE(((uintptr_t)ws & 15) == 0, "r3d_head_prototypes: ws must be 16-byte aligned");
  R3D_REQUIRE(n_ep == 1 || (assign_out && (ep.ws & 3) == 0 && ep.ws >= ws_words),
              "r3d_head_prototypes: a batch needs assign_out and a scratch stride that is a multiple of 4 words >= the scratch size");
  hipStream_t st = (hipStream_t)stream;
  SegGeom g{n_way, k_shot, N};
  const long cap = g.total_cap();
  const HpWs L = hp_carve(g);
  int* comp = ws + L.comp;
  float* mind = (float*)(ws + L.mind);
  int* assign = assign_out ? assign_out : ws + L.assign;
  HpEp e2 = ep;
  if (!assign_out) e2.assign = ep.ws;  // (single episode only)
  // more than 4 classes: the one-hot labels are two planes of 4 columns, plane 1 (classes 4..7) behind the n_ep episodes'
  // rows of plane 0 (the label propagation solves the planes one after the other: head_graph.hip)
  const long label_rows = ep.labels ? ep.labels : (long)g.nseg() * kr + n_query_pts;
  float* node_labels2 = n_way > 3 ? node_labels + (long)n_ep * label_rows * 4 : nullptr;
  Cand* cand0 = (Cand*)(ws + L.cand);
  Cand* cand1 = cand0 + g.total_blocks();
  int* sel = ws + L.sel;
  int* seeds = ws + L.seeds;
  const int max_chunks = L.max_chunks;
  float* part = (float*)(ws + L.part);
  int* part_cnt = ws + L.part_cnt;
  float* featC = (float*)(ws + L.featC);
  const long pitch = L.pitch;
  unsigned long long* xch = (unsigned long long*)(ws + L.xch);
  unsigned long long* best_packed = (unsigned long long*)(ws + L.best);
  float* featP = (float*)(ws + L.featP);
  const int tb = g.total_blocks();
  hipLaunchKernelGGL(r3d_head_compact_kernel, dim3(g.nseg(), n_ep), dim3(1024), 0, st, support_y, shot_keep, g, comp, desc, e2);
  hipLaunchKernelGGL(r3d_head_gather_kernel, dim3(tb, n_ep), dim3(HP_BLOCK), 0, st, feat, ldf, D, g, comp, desc, featC, pitch,
                     featP, e2);
  if (flags & 1 /* R3D_HEAD_FPS_ONE_LAUNCH */) {
    const int tb_dense = (int)((g.cap(0) + HP_BLOCK - 1) / HP_BLOCK) + g.nseg();  // the segments' counts add up to cap(0)
    r3d_fill_words_ep(xch, 0u, 2L * kr * tb_dense + 2L * HP_MAXSEG * HP_MAXK, n_ep, e2.ws, st);
    // the launch's workgroups wait for each other: never more of them than the chip holds at once, whatever the caller asks
    const int grp = fps_slots_clamp(D, tb_dense, fps_group);
    for (int e0 = 0; e0 < n_ep; e0 += grp) {
      const int ne = n_ep - e0 < grp ? n_ep - e0 : grp;
#define FPS_ONE(DPAD, LC)                                                                                               \
      do {                                                                                                              \
        if (D == DPAD)                                                                                                  \
          hipLaunchKernelGGL((r3d_fps_persistent_kernel<DPAD, true, LC>), dim3(tb_dense, ne), dim3(HP_BLOCK), 0, st, featC, pitch, \
                             featP, D, g, desc, k, kr, xch, tb_dense, sel, e2, e0);                                         \
        else                                                                                                            \
          hipLaunchKernelGGL((r3d_fps_persistent_kernel<DPAD, false, LC>), dim3(tb_dense, ne), dim3(HP_BLOCK), 0, st, featC, pitch, \
                             featP, D, g, desc, k, kr, xch, tb_dense, sel, e2, e0);                                         \
      } while (0)
      if (D <= 64) FPS_ONE(64, 0);
      else if (D <= 128) FPS_ONE(128, 0);
      else if (D <= 192) FPS_ONE(192, 0);
      else FPS_ONE(256, 0);
#undef FPS_ONE
    }
  } else {
    for (int t = 0; t < kr; ++t) {
  #define FPS_LAUNCH(DPAD)                                                                                          \
      hipLaunchKernelGGL(r3d_fps_round_kernel<DPAD>, dim3(tb, n_ep), dim3(HP_BLOCK), 0, st, featC, pitch, D, g,      \
                         desc, k, kr, t, mind, (t & 1) ? cand0 : cand1, (t & 1) ? cand1 : cand0, sel, e2)
      if (D <= 64) FPS_LAUNCH(64);
      else if (D <= 128) FPS_LAUNCH(128);
      else if (D <= 192) FPS_LAUNCH(192);
      else FPS_LAUNCH(256);
  #undef FPS_LAUNCH
    }
  }
  hipLaunchKernelGGL(r3d_fps_finalize_kernel, dim3(n_ep), dim3(HP_MAXK), 0, st, g, k, n_query_pts, sel, seeds, desc, e2);
  r3d_fill_words_ep(best_packed, 0xffffffffu, 2 * cap, n_ep, e2.ws, st);
  hipLaunchKernelGGL(r3d_assign_kernel, dim3(tb, r3d_cdiv(kr, AS_TILE), n_ep), dim3(HP_BLOCK), 0, st, featC, pitch, D, g, desc,
                     seeds, best_packed, e2);
  hipLaunchKernelGGL(r3d_assign_unpack_kernel, dim3(r3d_cdiv(cap, 256), n_ep), dim3(256), 0, st, best_packed, cap, assign, e2);
  hipLaunchKernelGGL(r3d_cluster_partial_kernel, dim3(kr, max_chunks, g.nseg() * n_ep), dim3(HP_BLOCK), 0, st, feat, ldf, D, g,
                     comp, desc, assign, max_chunks, part, part_cnt, e2);
  hipLaunchKernelGGL(r3d_cluster_reduce_kernel, dim3(kr, g.nseg(), n_ep), dim3(HP_BLOCK), 0, st, D, desc, max_chunks, part,
                     part_cnt, nodes, ldn, node_labels, node_labels2, cluster_count, e2);
  hipLaunchKernelGGL(r3d_nodes_append_query_kernel, dim3(r3d_cdiv((long)n_query_pts * D, 256), n_ep), dim3(256), 0, st, qfeat,
                     ldq, D, n_query_pts, desc, nodes, ldn, node_labels, node_labels2, e2);
  R3D_LAUNCH_CHECK("r3d_head_prototypes");
  return R3D_OK;
}

extern "C" int r3d_head_prototypes(const int32_t* support_y, const int32_t* shot_keep, const float* feat,
                                   long ldf, const float* featT, const float* qfeat, long ldq, int n_way,
                                   int k_shot, int N, int D, int n_query_pts, int k, float* nodes, long ldn,
                                   float* node_labels, int32_t* desc, int32_t* assign_out,
                                   int32_t* cluster_count, int32_t* ws, long ws_words, int flags, void* stream) {
  (void)featT;  // (ABI version 2 took a channel-major copy; the compacted copy is built here)
  const HpEp one{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  return head_prototypes_impl(1, one, 1, support_y, shot_keep, feat, ldf, qfeat, ldq, n_way, k_shot, N, D, n_query_pts, k, nodes,
                              ldn, node_labels, desc, assign_out, cluster_count, ws, ws_words, flags, stream);
}

// n_ep episodes in one launch sequence.  Per-episode strides: support_y / shot_keep in int32 words, feat / qfeat in ROWS
// (episode e's support rows start feat_ep_rows rows after episode e - 1's), nodes / node_labels in rows, desc / assign /
// cluster_count / ws in int32 words (ws stride even and >= r3d_head_proto_ws_words).
extern "C" int r3d_head_prototypes_batched(int n_ep, int fps_group, const int32_t* support_y, long sy_stride,
                                           const int32_t* shot_keep, long keep_stride, const float* feat, long ldf,
                                           long feat_ep_rows, const float* qfeat, long ldq, long qfeat_ep_rows, int n_way,
                                           int k_shot, int N, int D, int n_query_pts, int k, float* nodes, long ldn,
                                           long nodes_ep_rows, float* node_labels, int32_t* desc, long desc_stride,
                                           int32_t* assign_out, long assign_stride, int32_t* cluster_count,
                                           long ccount_stride, int32_t* ws, long ws_words, long ws_stride, int flags,
                                           void* stream) {
  const HpEp ep{sy_stride, keep_stride, feat_ep_rows, qfeat_ep_rows, nodes_ep_rows, nodes_ep_rows, desc_stride, assign_stride,
                ccount_stride, ws_stride};
  return head_prototypes_impl(n_ep, ep, fps_group, support_y, shot_keep, feat, ldf, qfeat, ldq, n_way, k_shot, N, D,
                              n_query_pts, k, nodes, ldn, node_labels, desc, assign_out, cluster_count, ws, ws_words, flags,
                              stream);
}

// ---------------------------------------------------------------------------
// backward of the cluster means (training): every listed support point receives
// dproto[cluster] / |cluster| ; query rows copy through.  One wave per list position.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(HP_BLOCK) void r3d_proto_bwd_kernel(const float* __restrict__ dnodes, long ldd, int D, SegGeom g,
                                                                 const int* __restrict__ comp, const int* __restrict__ desc,
                                                                 const int* __restrict__ assign,
                                                                 const int* __restrict__ cluster_count,
                                                                 float* __restrict__ dsfeat, long lds_, HpEp st) {
  const int ep = blockIdx.y;
  dnodes += (long)ep * st.nodes * ldd; HP_SHIFT(comp, st.ws); HP_SHIFT(desc, st.desc); HP_SHIFT(assign, st.assign);
  HP_SHIFT(cluster_count, st.ccount); dsfeat += (long)ep * st.feat * lds_;
  int blk0;
  const int seg = g.seg_of_block(blockIdx.x, &blk0);
  const int count = desc[HD_SEG_COUNT + seg];
  const int bis = blockIdx.x - blk0;
  if ((long)bis * HP_BLOCK >= count) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int poff = desc[HD_SEG_POFF + seg];
  for (int pos = bis * HP_BLOCK + w; pos < min(count, (bis + 1) * HP_BLOCK); pos += 4) {
    const int gp = comp[g.off(seg) + pos];
    const int row = poff + assign[g.off(seg) + pos];
    const float inv = 1.f / (float)cluster_count[row];
    for (int c = lane; c < D; c += 64) dsfeat[(long)gp * lds_ + c] = dnodes[(long)row * ldd + c] * inv;
  }
}

__global__ void r3d_query_bwd_kernel(const float* __restrict__ dnodes, long ldd, int D, int nq_pts,
                                     const int* __restrict__ desc, float* __restrict__ dqfeat, long ldq, HpEp st) {
  const int ep = blockIdx.y;
  dnodes += (long)ep * st.nodes * ldd; HP_SHIFT(desc, st.desc); dqfeat += (long)ep * st.qfeat * ldq;
  const int n_proto = desc[HD_N_PROTO];
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nq_pts * D) return;
  const long r = (long)((unsigned)i / (unsigned)D);  // (n_query_pts D < 2^31, checked by the entry point: a 32-bit division)
  const int c = (int)(i - r * D);
  dqfeat[r * ldq + c] = dnodes[(n_proto + r) * ldd + c];
}

// dsfeat (S*N, lds) must be zero-initialised by the caller (points in no list keep a zero gradient)
static int head_prototypes_bwd_impl(int n_ep, const HpEp& ep, const float* dnodes, long ldd, int n_way, int k_shot, int N, int D,
                                    int n_query_pts, const int32_t* desc, const int32_t* assign, const int32_t* cluster_count,
                                    const int32_t* ws, float* dsfeat, long lds_, float* dqfeat, long ldq, void* stream) {
  R3D_REQUIRE(dnodes && desc && assign && cluster_count && ws && dsfeat && dqfeat, "r3d_head_prototypes_bwd: null pointer");
  int rc = check_geom("r3d_head_prototypes_bwd", n_way, k_shot, N, D);
  if (rc) return rc;
  rc = check_query_rows("r3d_head_prototypes_bwd", n_query_pts, D);
  if (rc) return rc;
  R3D_REQUIRE(n_ep >= 1 && n_ep <= 4096, "r3d_head_prototypes_bwd: %d episodes", n_ep);
  SegGeom g{n_way, k_shot, N};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(r3d_proto_bwd_kernel, dim3(g.total_blocks(), n_ep), dim3(HP_BLOCK), 0, st, dnodes, ldd, D, g, ws /* comp */,
                     desc, assign, cluster_count, dsfeat, lds_, ep);
  hipLaunchKernelGGL(r3d_query_bwd_kernel, dim3(r3d_cdiv((long)n_query_pts * D, 256), n_ep), dim3(256), 0, st, dnodes, ldd, D,
                     n_query_pts, desc, dqfeat, ldq, ep);
  R3D_LAUNCH_CHECK("r3d_head_prototypes_bwd");
  return R3D_OK;
}
extern "C" int r3d_head_prototypes_bwd(const float* dnodes, long ldd, int n_way, int k_shot, int N, int D, int n_query_pts,
                                       const int32_t* desc, const int32_t* assign, const int32_t* cluster_count, const int32_t* ws,
                                       float* dsfeat, long lds_, float* dqfeat, long ldq, void* stream) {
  const HpEp one{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  return head_prototypes_bwd_impl(1, one, dnodes, ldd, n_way, k_shot, N, D, n_query_pts, desc, assign, cluster_count, ws, dsfeat,
                                  lds_, dqfeat, ldq, stream);
}
// strides as r3d_head_prototypes_batched (dsfeat / dqfeat: rows of the gradient buffers between episodes)
extern "C" int r3d_head_prototypes_bwd_batched(int n_ep, const float* dnodes, long ldd, long nodes_ep_rows, int n_way, int k_shot,
                                               int N, int D, int n_query_pts, const int32_t* desc, long desc_stride,
                                               const int32_t* assign, long assign_stride, const int32_t* cluster_count,
                                               long ccount_stride, const int32_t* ws, long ws_stride, float* dsfeat, long lds_,
                                               long dsfeat_ep_rows, float* dqfeat, long ldq, long dqfeat_ep_rows, void* stream) {
  const HpEp ep{0, 0, dsfeat_ep_rows, dqfeat_ep_rows, nodes_ep_rows, nodes_ep_rows, desc_stride, assign_stride, ccount_stride,
                ws_stride};
  return head_prototypes_bwd_impl(n_ep, ep, dnodes, ldd, n_way, k_shot, N, D, n_query_pts, desc, assign, cluster_count, ws,
                                  dsfeat, lds_, dqfeat, ldq, stream);
}

// Word offsets of the scratch sub-arrays inside ws (for tests that inspect the
// intermediate index results): comp, mind, assign, cand, sel, seeds.
extern "C" int r3d_head_proto_ws_offsets(int n_way, int k_shot, int N, long* out6) {
  const HpWs L = hp_carve(SegGeom{n_way, k_shot, N});
  out6[0] = L.comp; out6[1] = L.mind; out6[2] = L.assign; out6[3] = L.cand; out6[4] = L.sel; out6[5] = L.seeds;
  return R3D_OK;
}
