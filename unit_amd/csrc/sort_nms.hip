// sort_nms.hip -- stable descending sort of RPN objectness (K10) and NMS (K11), latency-bound integer work.
#include "common.h"
#include <type_traits>

// ---------------------------------------------------------------------------------------------------
// K10  logits.sort(descending=True) (find_top_rpn_proposals via rpn.py:48).
// One 1024-thread workgroup per image: LSD radix sort, 4-bit digits, 8 passes, stable (ties keep ascending
// original index, like torch's stable CPU sort).  keys are read from a strided matrix: key(i) =
// src[b*bstride + (i / A) * ld + col0 + (i % A)]  (NHWC RPN head output) -- pass A=1, ld=1 for a flat vector.
// workspace: 2 * B * n * (4+4) bytes (ping-pong keys + idx).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned desc_key(float f) {
  unsigned u = __float_as_uint(f);
  if (u == 0x80000000u) u = 0u;   // -0.0 == +0.0 for the comparison sort the reference uses
  unsigned asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending-order key
  return ~asc;                                                 // descending
}
__device__ __forceinline__ float key_to_float(unsigned k) {
  unsigned asc = ~k;
  unsigned u = (asc & 0x80000000u) ? (asc & 0x7fffffffu) : ~asc;
  return __uint_as_float(u);
}

#define SORT_THREADS 1024
__global__ void __launch_bounds__(SORT_THREADS) radix_sort_kernel(const float* __restrict__ src, long bstride, int ld, int A, int col0,
                                                          int n, unsigned* __restrict__ kbuf, int* __restrict__ ibuf,
                                                          float* __restrict__ out_keys, int* __restrict__ out_idx) {
  __shared__ unsigned short hist[16 * SORT_THREADS];  // [digit][thread]
  __shared__ int wsum[17];
  int b = blockIdx.x;
  unsigned* k0 = kbuf + (size_t)b * 2 * n; unsigned* k1 = k0 + n;
  int* i0 = ibuf + (size_t)b * 2 * n; int* i1 = i0 + n;
  int tid = threadIdx.x;
  int chunk = (n + SORT_THREADS - 1) / SORT_THREADS;
  int s = tid * chunk, e = min(n, s + chunk);
  for (int i = tid; i < n; i += SORT_THREADS) {
    int pix = i / A, a = i - pix * A;
    k0[i] = desc_key(src[(size_t)b * bstride + (size_t)pix * ld + col0 + a]);
    i0[i] = i;
  }
  __syncthreads();
  for (int pass = 0; pass < 8; ++pass) {
    int shift = pass * 4;
    unsigned short cnt[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) cnt[d] = 0;
    for (int i = s; i < e; ++i) {
      unsigned d = (k0[i] >> shift) & 15u;
#pragma unroll
      for (int q = 0; q < 16; ++q) cnt[q] += (d == (unsigned)q) ? 1 : 0;
    }
#pragma unroll
    for (int d = 0; d < 16; ++d) hist[d * SORT_THREADS + tid] = cnt[d];
    __syncthreads();
    // exclusive scan over the 16*1024 counters in (digit, thread) order; thread t owns entries [16t, 16t+16)
    int loc[16]; int sum = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { loc[j] = sum; sum += hist[tid * 16 + j]; }
    int lane = tid & 63, wid = tid >> 6;
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int w = 0; w < SORT_THREADS / 64; ++w) { int t = wsum[w]; wsum[w] = run; run += t; } }
    __syncthreads();
    int base = wsum[wid] + inc - sum;
    // overwrite hist with exclusive prefixes (as int would overflow u16: keep in registers, re-layout through LDS ints)
    __syncthreads();
    int* pref = reinterpret_cast<int*>(hist);  // 16*1024 u16 = 32 KB = 8192 ints: not enough for 16384 ints -> two halves
    // scatter positions for this thread's digits are needed by thread tid for digit d: entry index d*1024+tid, owned by
    // thread (d*1024+tid)/16. Exchange through LDS in two halves of 8 digits.
    int mypos[16];
    for (int half = 0; half < 2; ++half) {
      // owner thread t holds entries [16t,16t+16) i.e. digits d = (16t)/1024 .. ; entries of digits [8*half, 8*half+8) are
      // owned by threads [512*half, 512*half+512).
      if ((tid >> 9) == half) {
#pragma unroll
        for (int j = 0; j < 16; ++j) pref[(tid & 511) * 16 + j] = base + loc[j];
      }
      __syncthreads();
#pragma unroll
      for (int dd = 0; dd < 8; ++dd) mypos[half * 8 + dd] = pref[dd * SORT_THREADS + tid];
      __syncthreads();
    }
    for (int i = s; i < e; ++i) {
      unsigned k = k0[i]; unsigned d = (k >> shift) & 15u;
      int p = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) if (d == (unsigned)q) { p = mypos[q]; mypos[q] = p + 1; }
      k1[p] = k; i1[p] = i0[i];
    }
    __syncthreads();
    unsigned* tk = k0; k0 = k1; k1 = tk;
    int* ti = i0; i0 = i1; i1 = ti;
  }
  for (int i = tid; i < n; i += SORT_THREADS) {
    int id = i0[i];
    int pix = id / A, a = id - pix * A;
    out_keys[(size_t)b * n + i] = src[(size_t)b * bstride + (size_t)pix * ld + col0 + a];   // original bits (keeps -0.0)
    out_idx[(size_t)b * n + i] = id;
  }
}

// Same algorithm with each thread's contiguous chunk (<= ITEMS keys) held in REGISTERS for a whole pass: the chunk is read
// once with ITEMS independent loads in flight (the loop form above serialises ~2*chunk dependent L2 round trips per pass).
template <int ITEMS>
__global__ void __launch_bounds__(SORT_THREADS) radix_sort_reg_kernel(const float* __restrict__ src, long bstride, int ld, int A, int col0,
                                                              int n, unsigned* __restrict__ kbuf, int* __restrict__ ibuf,
                                                              float* __restrict__ out_keys, int* __restrict__ out_idx) {
  __shared__ unsigned short hist[16 * SORT_THREADS];
  __shared__ int wsum[17];
  int b = blockIdx.x;
  unsigned* k0 = kbuf + (size_t)b * 2 * n; unsigned* k1 = k0 + n;
  int* i0 = ibuf + (size_t)b * 2 * n; int* i1 = i0 + n;
  int tid = threadIdx.x;
  int chunk = (n + SORT_THREADS - 1) / SORT_THREADS;      // <= ITEMS
  int s = tid * chunk;
  int cnt_mine = max(0, min(n - s, chunk));
  unsigned keys[ITEMS]; int idxs[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    keys[j] = 0u; idxs[j] = 0;
    if (j < cnt_mine) {
      int i = s + j;
      int pix = i / A, a = i - pix * A;
      keys[j] = desc_key(src[(size_t)b * bstride + (size_t)pix * ld + col0 + a]);
      idxs[j] = i;
    }
  }
  for (int pass = 0; pass < 8; ++pass) {
    int shift = pass * 4;
    if (pass > 0) {
#pragma unroll
      for (int j = 0; j < ITEMS; ++j)
        if (j < cnt_mine) { keys[j] = k0[s + j]; idxs[j] = i0[s + j]; }
    }
    unsigned short cnt[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) cnt[d] = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      unsigned d = (keys[j] >> shift) & 15u;
#pragma unroll
      for (int q = 0; q < 16; ++q) cnt[q] += (j < cnt_mine && d == (unsigned)q) ? 1 : 0;
    }
#pragma unroll
    for (int d = 0; d < 16; ++d) hist[d * SORT_THREADS + tid] = cnt[d];
    __syncthreads();
    int loc[16]; int sum = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { loc[j] = sum; sum += hist[tid * 16 + j]; }
    int lane = tid & 63, wid = tid >> 6;
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int w = 0; w < SORT_THREADS / 64; ++w) { int t = wsum[w]; wsum[w] = run; run += t; } }
    __syncthreads();
    int base = wsum[wid] + inc - sum;
    __syncthreads();
    int* pref = reinterpret_cast<int*>(hist);
    int mypos[16];
    for (int half = 0; half < 2; ++half) {
      if ((tid >> 9) == half) {
#pragma unroll
        for (int j = 0; j < 16; ++j) pref[(tid & 511) * 16 + j] = base + loc[j];
      }
      __syncthreads();
#pragma unroll
      for (int dd = 0; dd < 8; ++dd) mypos[half * 8 + dd] = pref[dd * SORT_THREADS + tid];
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      if (j < cnt_mine) {
        unsigned d = (keys[j] >> shift) & 15u;
        int p = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) if (d == (unsigned)q) { p = mypos[q]; mypos[q] = p + 1; }
        k1[p] = keys[j]; i1[p] = idxs[j];
      }
    }
    __syncthreads();
    unsigned* tk = k0; k0 = k1; k1 = tk;
    int* ti = i0; i0 = i1; i1 = ti;
  }
  for (int i = tid; i < n; i += SORT_THREADS) {
    int id = i0[i];
    int pix = id / A, a = id - pix * A;
    out_keys[(size_t)b * n + i] = src[(size_t)b * bstride + (size_t)pix * ld + col0 + a];
    out_idx[(size_t)b * n + i] = id;
  }
}

// [B][n] candidates / sort ping-pong buffers, candidate counts, and the [B][8192] (start, end) group table of the top-k path
extern "C" size_t unit_sort_workspace_bytes(int B, int n) { return (size_t)B * 2 * n * 8 + (size_t)B * 8192 * 8 + 256; }

extern "C" int unit_sort_desc_stable(const float* src, long batch_stride, int ld, int A, int col0, int B, int n,
                                     float* out_keys, int* out_idx, void* workspace, size_t workspace_bytes, void* stream) {
  UNIT_CHECK_ARG(n <= 1024 * SORT_THREADS, "sort: n too large for the single-workgroup sorter (<= 1M keys)");
  if (workspace_bytes < unit_sort_workspace_bytes(B, n)) { unit_set_error("sort: workspace too small"); return UNIT_ERR_WORKSPACE; }
  if (B == 0 || n == 0) return UNIT_OK;
  unsigned* kbuf = (unsigned*)workspace;
  int* ibuf = (int*)((char*)workspace + (size_t)B * 2 * n * 4);
  int chunk = (n + SORT_THREADS - 1) / SORT_THREADS;
  hipStream_t st = (hipStream_t)stream;
  if (chunk <= 8) radix_sort_reg_kernel<8><<<B, SORT_THREADS, 0, st>>>(src, batch_stride, ld, A, col0, n, kbuf, ibuf, out_keys, out_idx);
  else if (chunk <= 40) radix_sort_reg_kernel<40><<<B, SORT_THREADS, 0, st>>>(src, batch_stride, ld, A, col0, n, kbuf, ibuf, out_keys, out_idx);
  else if (chunk <= 64) radix_sort_reg_kernel<64><<<B, SORT_THREADS, 0, st>>>(src, batch_stride, ld, A, col0, n, kbuf, ibuf, out_keys, out_idx);
  else radix_sort_kernel<<<B, SORT_THREADS, 0, st>>>(src, batch_stride, ld, A, col0, n, kbuf, ibuf, out_keys, out_idx);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// K10b  top-k form of the same sort: find_top_rpn_proposals only consumes the first pre_nms_topk entries of the sorted
// row (12 000 of 35 910 per image in training).  A single-workgroup radix sort leaves 252 of the 256 CUs idle for ~1 ms,
// so the top-k path is chip-wide instead:
//  (1) topk_select_kernel (one workgroup per image): 8192-bin histogram of the top 13 key bits -> the bin T holding the
//      topk-th best key -> every key in a bin <= T becomes a candidate, packed as (key << 32 | index) and written GROUPED BY
//      BIN (a counting sort on the 13 bits: position = prefix of the bin + an LDS cursor; order inside a group arbitrary);
//      the (start, end) of every group goes to a table. Nc >= min(topk, n) candidates; non-candidates sort after them.
//  (2) group_rank_kernel (64 candidates per workgroup): rank(i) = start of its group + #{j in the group : cand_j < cand_i} on the
//      unique 64-bit composites (ties broken by ascending original index = the stable order); writes out[rank]. The groups
//      hold a few hundred candidates (16 bins per octave of score), so this is ~1/50 of the all-pairs count it replaces
//      (rank_sort_kernel, kept for reference: 85 us for 4 x 12 000). Entries beyond Nc get index -1 / score -inf.
// ---------------------------------------------------------------------------------------------------
#define SEL_BINS 8192
__global__ void __launch_bounds__(1024) topk_select_kernel(const float* __restrict__ src, long bstride, int ld, int A, int col0, int n,
                                                           int topk, float min_excl, unsigned long long* __restrict__ cand,
                                                           int* __restrict__ cand_count, int2* __restrict__ groups) {
  __builtin_amdgcn_s_setprio(2);   // proposal chain = critical path of the step; the other streams' kernels are throughput work
  __shared__ int hist[SEL_BINS];
  __shared__ int wsum[16];
  __shared__ int s_T, s_cnt;
  int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* sb = src + (size_t)b * bstride + col0;
  for (int i = tid; i < SEL_BINS; i += 1024) hist[i] = 0;
  if (tid == 0) { s_T = SEL_BINS - 1; s_cnt = 0; }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    int pix = i / A, a = i - pix * A;
    float v = sb[(size_t)pix * ld + a];
    if (v > min_excl) atomicAdd(&hist[desc_key(v) >> 19], 1);       // keys <= min_excl never become candidates
  }
  __syncthreads();
  int loc[8], sum = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { loc[j] = sum; sum += hist[tid * 8 + j]; }
  int inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  int wbase = 0;
  for (int w = 0; w < wid; ++w) wbase += wsum[w];
  int excl = wbase + inc - sum;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int lo = excl + loc[j], hi = lo + hist[tid * 8 + j];
    if (lo < topk && hi >= topk) s_T = tid * 8 + j;      // exactly one bin satisfies this when n >= topk
  }
  __syncthreads();
  unsigned T = (unsigned)s_T;
  // group table + cursors: hist[bin] becomes the write cursor of the bin (its exclusive prefix)
  int2* gb = groups + (size_t)b * SEL_BINS;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int bin = tid * 8 + j;
    int lo = excl + loc[j], hi = lo + hist[bin];
    gb[bin] = make_int2(lo, hi);
    if ((unsigned)bin == T) s_cnt = hi;                   // candidates = every key of the bins <= T
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) hist[tid * 8 + j] = excl + loc[j];
  __syncthreads();
  unsigned long long* cb = cand + (size_t)b * n;
  for (int i = tid; i < n; i += 1024) {
    int pix = i / A, a = i - pix * A;
    float v = sb[(size_t)pix * ld + a];
    unsigned key = desc_key(v);
    if ((key >> 19) <= T && v > min_excl) cb[atomicAdd(&hist[key >> 19], 1)] = ((unsigned long long)key << 32) | (unsigned)i;
  }
  if (tid == 0) cand_count[b] = s_cnt;
}

__global__ void __launch_bounds__(256) group_rank_kernel(const unsigned long long* __restrict__ cand, const int* __restrict__ cand_count,
                                                         const int2* __restrict__ groups, int n, const float* __restrict__ src, long bstride,
                                                         int ld, int A, int col0, float* __restrict__ out_keys, int* __restrict__ out_idx) {
  // 64 consecutive (bin-grouped) candidates per workgroup; they are compared with the candidates of THEIR groups only: everything
  // before the first group is smaller, everything behind the last one larger. Tiles of 1024 composites through LDS, the four
  // waves take a quarter of each tile (rank_sort_kernel's loop over a sub-range).
  __builtin_amdgcn_s_setprio(2);   // proposal chain = critical path of the step
  __shared__ unsigned long long tile[1024];
  __shared__ int part[4][64];
  int b = blockIdx.y;
  int nc = cand_count[b];
  int i0 = blockIdx.x * 64;
  // entries that are no candidates (NaN scores, scores <= min_exclusive; with finite scores: those past the bin of the topk-th) get
  // index -1 / score -inf instead of staying unwritten: a consumer that reads `topk` entries (unit_rpn_decode_select skips index < 0)
  // must never see stale memory when a diverged model leaves fewer than topk valid scores
  if (threadIdx.x < 64 && i0 + (int)threadIdx.x >= nc && i0 + (int)threadIdx.x < n) {
    out_keys[(size_t)b * n + i0 + threadIdx.x] = -__builtin_inff();
    out_idx[(size_t)b * n + i0 + threadIdx.x] = -1;
  }
  if (i0 >= nc) return;
  int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const unsigned long long* c = cand + (size_t)b * n;
  const int2* gb = groups + (size_t)b * SEL_BINS;
  unsigned long long mine = (i0 + lane < nc) ? c[i0 + lane] : ~0ull;
  int rs = gb[(unsigned)(c[i0] >> 51)].x;                                  // bin = key >> 19 = composite >> 51
  int re = gb[(unsigned)(c[min(i0 + 63, nc - 1)] >> 51)].y;
  int cnt = 0;
  for (int j0 = rs; j0 < re; j0 += 1024) {
    __syncthreads();
#pragma unroll
    for (int t = tid; t < 1024; t += 256) tile[t] = (j0 + t < re) ? c[j0 + t] : ~0ull;   // ~0 is never < anything
    __syncthreads();
    const unsigned long long* tq = tile + wid * 256;
#pragma unroll 16
    for (int j = 0; j < 256; ++j) cnt += (tq[j] < mine) ? 1 : 0;
  }
  part[wid][lane] = cnt;
  __syncthreads();
  if (wid == 0 && i0 + lane < nc) {
    int r = rs + part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
    int id = (int)(unsigned)(mine & 0xffffffffull);
    int pix = id / A, a = id - pix * A;
    out_keys[(size_t)b * n + r] = src[(size_t)b * bstride + (size_t)pix * ld + col0 + a];   // original bits (keeps -0.0)
    out_idx[(size_t)b * n + r] = id;
  }
}

__global__ void __launch_bounds__(256) rank_sort_kernel(const unsigned long long* __restrict__ cand, const int* __restrict__ cand_count,
                                                        int n, const float* __restrict__ src, long bstride, int ld, int A, int col0,
                                                        float* __restrict__ out_keys, int* __restrict__ out_idx) {
  __builtin_amdgcn_s_setprio(2);   // proposal chain = critical path of the step; the other streams' kernels are throughput work
  __shared__ unsigned long long tile[1024];
  __shared__ int part[4][64];
  int b = blockIdx.y;
  int nc = cand_count[b];
  int i0 = blockIdx.x * 64;
  if (threadIdx.x < 64 && i0 + (int)threadIdx.x >= nc && i0 + (int)threadIdx.x < n) {      // (group_rank_kernel)
    out_keys[(size_t)b * n + i0 + threadIdx.x] = -__builtin_inff();
    out_idx[(size_t)b * n + i0 + threadIdx.x] = -1;
  }
  if (i0 >= nc) return;
  int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const unsigned long long* c = cand + (size_t)b * n;
  unsigned long long mine = (i0 + lane < nc) ? c[i0 + lane] : ~0ull;
  int cnt = 0;
  for (int j0 = 0; j0 < nc; j0 += 1024) {
    __syncthreads();
#pragma unroll
    for (int t = tid; t < 1024; t += 256) tile[t] = (j0 + t < nc) ? c[j0 + t] : ~0ull;   // ~0 is never < anything
    __syncthreads();
    const unsigned long long* tq = tile + wid * 256;
#pragma unroll 16
    for (int j = 0; j < 256; ++j) cnt += (tq[j] < mine) ? 1 : 0;
  }
  part[wid][lane] = cnt;
  __syncthreads();
  if (wid == 0 && i0 + lane < nc) {
    int r = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
    int id = (int)(unsigned)(mine & 0xffffffffull);
    int pix = id / A, a = id - pix * A;
    out_keys[(size_t)b * n + r] = src[(size_t)b * bstride + (size_t)pix * ld + col0 + a];   // original bits (keeps -0.0)
    out_idx[(size_t)b * n + r] = id;
  }
}

extern "C" int unit_sort_desc_stable_topk(const float* src, long batch_stride, int ld, int A, int col0, int B, int n, int topk,
                                          float min_exclusive, float* out_keys, int* out_idx, void* workspace, size_t workspace_bytes,
                                          void* stream) {
  if (workspace_bytes < unit_sort_workspace_bytes(B, n)) { unit_set_error("sort: workspace too small"); return UNIT_ERR_WORKSPACE; }
  UNIT_CHECK_ARG(topk > 0, "sort_topk: topk must be positive");
  if (B == 0 || n == 0) return UNIT_OK;
  unsigned long long* cand = (unsigned long long*)workspace;
  int* cand_count = (int*)((char*)workspace + (size_t)B * n * 8);
  int2* groups = (int2*)((char*)workspace + (((size_t)B * 2 * n * 8 + 255) & ~(size_t)255));
  hipStream_t st = (hipStream_t)stream;
  topk_select_kernel<<<B, 1024, 0, st>>>(src, batch_stride, ld, A, col0, n, topk, min_exclusive, cand, cand_count, groups);
  UNIT_LAUNCH_CHECK();
  // UNIT_RANK_ALLPAIRS=1: the all-pairs rank count this replaced (A/B, tools/nms_bench.py)
  static int allpairs = -1;
  if (allpairs < 0) { const char* e = getenv("UNIT_RANK_ALLPAIRS"); allpairs = e ? atoi(e) : 0; }
  if (allpairs) rank_sort_kernel<<<dim3((n + 63) / 64, B), 256, 0, st>>>(cand, cand_count, n, src, batch_stride, ld, A, col0, out_keys, out_idx);
  else group_rank_kernel<<<dim3((n + 63) / 64, B), 256, 0, st>>>(cand, cand_count, groups, n, src, batch_stride, ld, A, col0, out_keys, out_idx);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// K11  NMS (torchvision.ops.nms semantics): boxes sorted by descending score; box j is suppressed
// iff an already-kept i<j has IoU(i,j) > thresh (strict); IoU = inter / (area_i + area_j - inter), no +1.
//  (1) nms_mask_kernel: 64x64 tiles of the upper triangle -> bitmask[i][j/64]
//  (2) nms_scan_kernel: one 256-thread workgroup per image; thread t owns word t of the "removed" bitmap;
//      wave 0 resolves each 64-box chunk serially from the diagonal words, all threads then OR in the rows of the
//      boxes kept in that chunk.  Stops after max_keep boxes.  Emits kept boxes/scores (gathered) + indices.
// ---------------------------------------------------------------------------------------------------
// IoU > thresh without the IEEE division for all but the borderline pairs: with u = area_a + area_b - inter > 0 and
// p = RN(thresh * u), inter > p (1 + 2^-20) implies inter / u > thresh (1 + 2^-21) >= the float after thresh, so the rounded
// quotient exceeds thresh; inter < p (1 - 2^-20) implies the true (hence the rounded) quotient is <= thresh. Only pairs inside
// that 2^-19-wide band (and degenerate unions) take the division -- the result is the reference's bit for every pair.
__device__ __forceinline__ float vmax(float x, float y) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float vmin(float x, float y) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ bool nms_suppress_fast(const f32x4 a, float areaa, const f32x4 b, float areab, float thresh, float t_hi,
                                                  float t_lo) {
  // t_hi = RN(thresh (1 + 2^-20)), t_lo = RN(thresh (1 - 2^-20)): one rounding more than in the bound above, still inside it
  // (v_max / v_min through asm: fmaxf / fminf on values loaded from memory cost a canonicalising v_max x, x each in IEEE mode)
  float xx1 = vmax(a[0], b[0]), yy1 = vmax(a[1], b[1]);
  float xx2 = vmin(a[2], b[2]), yy2 = vmin(a[3], b[3]);
  float w = vmax(0.0f, xx2 - xx1), h = vmax(0.0f, yy2 - yy1);
  float inter = w * h;
  float u = areaa + areab - inter;
  bool sure_yes = inter > t_hi * u;
  bool sure_no = inter < t_lo * u;
  if (__builtin_expect(u > 1e-30f && (sure_yes || sure_no), 1)) return sure_yes;
  return inter / u > thresh;
}

// 256 threads = four waves, wave w owns column block 4 * blockIdx.x + w of row block blockIdx.y
__global__ void __launch_bounds__(256) nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ count, int cap, int nw,
                                                       float thresh, unsigned long long* __restrict__ mask) {
  __builtin_amdgcn_s_setprio(2);   // proposal chain = critical path of the step; the other streams' kernels are throughput work
  int b = blockIdx.z;
  int n = count ? min(count[b], cap) : cap;
  int rb = blockIdx.y, cb0 = blockIdx.x * 4;
  if (cb0 + 3 < rb) return;
  if (rb * 64 >= n || cb0 * 64 >= n) return;
  __shared__ f32x4 cbox[256];
  __shared__ float carea[256];
  const float* bx = boxes + (size_t)b * cap * 4;
  int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int cj = cb0 * 64 + tid;
  f32x4 cbx = cj < n ? *reinterpret_cast<const f32x4*>(bx + 4 * (size_t)cj) : f32x4{0.f, 0.f, 0.f, 0.f};
  cbox[tid] = cbx;
  carea[tid] = (cbx[2] - cbx[0]) * (cbx[3] - cbx[1]);
  __syncthreads();
  int cb = cb0 + wv;
  int i = rb * 64 + lane;
  if (cb < rb || cb >= nw || cb * 64 >= n || i >= n) return;
  f32x4 me = *reinterpret_cast<const f32x4*>(bx + 4 * (size_t)i);
  float areame = (me[2] - me[0]) * (me[3] - me[1]);
  // all 64 columns unconditionally (columns >= n are zero boxes: never suppressed)
  const float t_hi = thresh * 1.00000095367431640625f, t_lo = thresh * 0.99999904632568359375f;
  unsigned lo = 0, hi = 0;
  const f32x4* cbw = cbox + wv * 64;
  const float* caw = carea + wv * 64;
#pragma unroll 8
  for (int j = 0; j < 32; ++j)
    if (nms_suppress_fast(me, areame, cbw[j], caw[j], thresh, t_hi, t_lo)) lo |= 1u << j;
#pragma unroll 8
  for (int j = 32; j < 64; ++j)
    if (nms_suppress_fast(me, areame, cbw[j], caw[j], thresh, t_hi, t_lo)) hi |= 1u << (j - 32);
  if (rb == cb) {       // the diagonal block is stored SYMMETRIC (IoU is, bit for bit): bit j of row i for every j != i, so that
    unsigned long long keepm = ~(1ull << lane);    // lane i of the scan has its predecessors (bits j < i) without a transpose
    lo &= (unsigned)keepm; hi &= (unsigned)(keepm >> 32);
  }
  mask[((size_t)b * cap + i) * nw + cb] = ((unsigned long long)hi << 32) | lo;
}

#define NMS_SCAN_THREADS 1024
__global__ void __launch_bounds__(NMS_SCAN_THREADS) nms_scan_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                            const int* __restrict__ count, int cap, int nw,
                                                            const unsigned long long* __restrict__ mask, int max_keep,
                                                            int* __restrict__ keep_idx, int* __restrict__ keep_count,
                                                            float* __restrict__ out_boxes, float* __restrict__ out_scores) {
  __shared__ unsigned long long s_cur, s_kept;
  __shared__ int s_nkept;
  int b = blockIdx.x;
  int n = count ? min(count[b], cap) : cap;
  const unsigned long long* mk = mask + (size_t)b * cap * nw;
  int tid = threadIdx.x;
  unsigned long long removed = 0;  // word `tid` of the removed bitmap
  int nkept = 0;
  if (tid == 0) s_nkept = 0;
  int nchunks = (n + 63) / 64;
  // wave 0 prefetches the next chunk's diagonal words while the row ORs of the current chunk are in flight
  unsigned long long diag_next = (tid < 64 && tid < n) ? mk[(size_t)tid * nw] : 0ull;
  for (int c = 0; c < nchunks; ++c) {
    if (tid == c) s_cur = removed;
    __syncthreads();
    if (tid < 64) {
      int i = c * 64 + tid;
      unsigned long long diag = diag_next;
      int inext = i + 64;
      diag_next = (inext < n) ? mk[(size_t)inext * nw + c + 1] : 0ull;
      // the 64-step dependency chain runs on the SCALAR unit: every value is wave-uniform (v_readlane with a uniform lane
      // index, s_or / s_bitcmp), instead of 64 dependent ds_bpermute round trips
      unsigned dlo_v = (unsigned)diag, dhi_v = (unsigned)(diag >> 32);
      unsigned long long cur0 = s_cur;
      // (the builtins return int: cast to unsigned before widening, or the low word sign-extends over the high one)
      unsigned long long cur = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(cur0 >> 32)) << 32) |
                               (unsigned)__builtin_amdgcn_readfirstlane((unsigned)cur0);
      unsigned long long kept = 0;
      int cnt = __builtin_amdgcn_readfirstlane(s_nkept);
      int lim = min(64, n - c * 64);
      // visit only the boxes that survive: the next alive bit is kept and its diagonal word (bits above it only) kills others
      unsigned long long alive = ~cur & (lim >= 64 ? ~0ull : ((1ull << lim) - 1ull));
      while (alive != 0ull && cnt < max_keep) {
        int j = __builtin_ctzll(alive);
        unsigned long long dj = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(dhi_v, j) << 32) |
                                (unsigned)__builtin_amdgcn_readlane(dlo_v, j);
        kept |= 1ull << j;
        alive &= ~(dj | (1ull << j));
        cnt++;
      }
      if (tid == 0) { s_kept = kept; s_nkept = cnt; }
      // emit this chunk's kept boxes in order
      if ((kept >> tid) & 1ull) {
        int rank = nkept + __popcll(kept & ((1ull << tid) - 1ull));
        size_t o = (size_t)b * max_keep + rank;
        keep_idx[o] = i;
        if (out_boxes) *reinterpret_cast<f32x4*>(out_boxes + 4 * o) = *reinterpret_cast<const f32x4*>(boxes + ((size_t)b * cap + i) * 4);
        if (out_scores) out_scores[o] = scores[(size_t)b * cap + i];
      }
    }
    __syncthreads();
    unsigned long long kept = s_kept;
    nkept = s_nkept;
    if (nkept >= max_keep) break;
    if (tid < nw && tid > c) {
      // OR in the mask rows of this chunk's kept boxes: 16 independent loads in flight per batch (latency-bound loop)
      const unsigned long long* base = mk + (size_t)(c * 64) * nw + tid;
      while (kept) {
        unsigned long long v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          v[u] = 0ull;
          if (kept) {
            int j = __ffsll((long long)kept) - 1;
            kept &= kept - 1;
            v[u] = base[(size_t)j * nw];
          }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) removed |= v[u];
      }
    }
  }
  if (tid == 0) keep_count[b] = nkept;
}

// nms_scan_pf_kernel: the scan for up to 16 384 candidates (nw <= 256 words, one thread per word) with the mask rows
// PREFETCHED one chunk ahead. In nms_scan_kernel every chunk pays a dependent HBM / Infinity-Cache round trip (~2-3 us)
// for the rows of the boxes it just kept, on a single CU, with the rest of the chip idle on the critical path of the step
// (0.8 ms for 4 x 12 000 RPN candidates). Here, while chunk c is being resolved, every thread already loads its word of the
// rows of chunk c+1's CANDIDATES: the boxes of chunk c+1 not yet removed by chunks < c (a superset of what chunk c+1 can
// keep, because chunk c can only remove more), first PF = 32 of them, into registers. After chunk c+1 is resolved the kept
// boxes' rows are already there; kept boxes beyond the 32 prefetched candidates (rare) take the old dependent-load path.
// Same greedy result, bit for bit.
#define NMS_PF 32
#define NMS_PF_MAXKEEP 4096
__device__ __forceinline__ unsigned long long nms_uniform64(unsigned long long v) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((unsigned)v);
}

__global__ void __launch_bounds__(256) nms_scan_pf_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                          const int* __restrict__ count, int cap, int nw,
                                                          const unsigned long long* __restrict__ mask, int max_keep,
                                                          int* __restrict__ keep_idx, int* __restrict__ keep_count,
                                                          float* __restrict__ out_boxes, float* __restrict__ out_scores) {
  // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the prefetch loads that must stay in flight
#define NMS_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
  __shared__ unsigned long long s_cur, s_kept, s_next;
  __shared__ int s_nkept;
  __shared__ int s_keep[NMS_PF_MAXKEEP];                   // kept indices: boxes / scores are gathered after the scan (a load
                                                          // inside it would wait, in order, for the prefetches behind it)
  const int b = blockIdx.x;
  const int n = count ? min(count[b], cap) : cap;
  const unsigned long long* mk = mask + (size_t)b * cap * nw;
  const int tid = threadIdx.x;
  unsigned long long removed = 0;  // word `tid` of the removed bitmap
  int nkept = 0;
  if (tid == 0) s_nkept = 0;
  const int nchunks = (n + 63) / 64;
  // diagonal word (word ch of row ch*64 + lane) of the chunk to resolve next: loaded by EVERY wave, unconditionally and one
  // chunk ahead like the prefetches (a load under the wave-0 branch would be merged back with a register copy, i.e. waited for)
  unsigned long long diag_next = mk[(size_t)min(tid & 63, cap - 1) * nw];

  // candidate rows of chunk `ch` (uniform word `cand`): thread t > ch loads word t of the first NMS_PF of them
  auto prefetch = [&](int ch, unsigned long long cand, unsigned long long (&buf)[NMS_PF]) {
    unsigned long long w = cand;
    const int tcl = min(tid, nw - 1);                     // loads are UNCONDITIONAL (clamped, value unused where it does not apply):
    const int rowmax = cap - 1;                            // exact in-order vmcnt counts let them stay in flight across the chunk
#pragma unroll
    for (int u = 0; u < NMS_PF; ++u) {
      bool have = w != 0ull;
      int j = have ? __builtin_ctzll(w) : 0;
      w &= w - 1ull;                                       // 0 stays 0
      buf[u] = mk[(size_t)min(ch * 64 + j, rowmax) * nw + tcl];
    }
  };
  auto valid_bits = [&](int ch) -> unsigned long long {
    int lim = n - ch * 64;
    return lim >= 64 ? ~0ull : (lim <= 0 ? 0ull : ((1ull << lim) - 1ull));
  };

  // one chunk: CUR / candc = rows prefetched for chunk c and the candidate word they belong to; NXT / candn are filled for c+1.
  // returns true when max_keep boxes have been kept
  auto step = [&](int c, unsigned long long (&CUR)[NMS_PF], unsigned long long candc, unsigned long long (&NXT)[NMS_PF],
                  unsigned long long& candn) -> bool {
    if (tid == c) s_cur = removed;
    if (tid == c + 1) s_next = removed;                    // still without the rows of chunk c: a superset of chunk c+1's survivors
    NMS_LDS_BARRIER();
    candn = (c + 1 < nchunks) ? (~nms_uniform64(s_next) & valid_bits(c + 1)) : 0ull;
    const unsigned long long diag_cur = diag_next;
    diag_next = mk[(size_t)min((c + 1) * 64 + (tid & 63), cap - 1) * nw + min(c + 1, nw - 1)];
    prefetch(c + 1, candn, NXT);
    if (tid < 64) {
      int i = c * 64 + tid;
      unsigned long long diag = i < n ? diag_cur : 0ull;
      // the 64-step dependency chain runs on the SCALAR unit (v_readlane with a uniform lane index, s_or / s_bitcmp)
      unsigned dlo_v = (unsigned)diag, dhi_v = (unsigned)(diag >> 32);
      unsigned long long cur = nms_uniform64(s_cur);
      unsigned long long kept = 0;
      int cnt = __builtin_amdgcn_readfirstlane(s_nkept);
      int lim = min(64, n - c * 64);
      // visit only the boxes that survive: the next alive bit is kept and its diagonal word (bits above it only) kills others
      unsigned long long alive = ~cur & (lim >= 64 ? ~0ull : ((1ull << lim) - 1ull));
      while (alive != 0ull && cnt < max_keep) {
        int j = __builtin_ctzll(alive);
        unsigned long long dj = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(dhi_v, j) << 32) |
                                (unsigned)__builtin_amdgcn_readlane(dlo_v, j);
        kept |= 1ull << j;
        alive &= ~(dj | (1ull << j));
        cnt++;
      }
      if (tid == 0) { s_kept = kept; s_nkept = cnt; }
      if ((kept >> tid) & 1ull) {
        int rank = nkept + __popcll(kept & ((1ull << tid) - 1ull));
        keep_idx[(size_t)b * max_keep + rank] = i;
        s_keep[rank] = i;
      }
    }
    NMS_LDS_BARRIER();
    unsigned long long kept = nms_uniform64(s_kept);
    nkept = s_nkept;
    if (nkept >= max_keep) return true;
    if (tid < nw && tid > c) {
      unsigned long long w = candc;
#pragma unroll
      for (int u = 0; u < NMS_PF; ++u) {
        if (w != 0ull) {
          int j = __builtin_ctzll(w);
          w &= w - 1ull;
          if ((kept >> j) & 1ull) { removed |= CUR[u]; kept &= ~(1ull << j); }
        }
      }
      // kept boxes that were not among the prefetched candidates
      const unsigned long long* base = mk + (size_t)(c * 64) * nw + tid;
      // (one batch = one memory round trip on the chunk's critical path: 32 rows cover 64 kept boxes with the prefetched ones)
      while (kept) {
        unsigned long long v[NMS_PF];
#pragma unroll
        for (int u = 0; u < NMS_PF; ++u) {
          v[u] = 0ull;
          if (kept) {                                      // wave-uniform
            int j = __ffsll((long long)kept) - 1;
            kept &= kept - 1;
            v[u] = base[(size_t)j * nw];
          }
        }
#pragma unroll
        for (int u = 0; u < NMS_PF; ++u) removed |= v[u];
      }
    }
    return false;
  };

  unsigned long long bufa[NMS_PF], bufb[NMS_PF];
  unsigned long long canda = nchunks > 0 ? valid_bits(0) : 0ull, candb = 0ull;
  prefetch(0, canda, bufa);
  for (int c = 0; c < nchunks; c += 2) {
    if (step(c, bufa, canda, bufb, candb)) break;
    if (c + 1 >= nchunks) break;
    if (step(c + 1, bufb, candb, bufa, canda)) break;
  }
  if (tid == 0) keep_count[b] = nkept;
  NMS_LDS_BARRIER();
  for (int r = tid; r < nkept; r += 256) {
    int i = s_keep[r];
    size_t o = (size_t)b * max_keep + r;
    if (out_boxes) *reinterpret_cast<f32x4*>(out_boxes + 4 * o) = *reinterpret_cast<const f32x4*>(boxes + ((size_t)b * cap + i) * 4);
    if (out_scores) out_scores[o] = scores[(size_t)b * cap + i];
  }
#undef NMS_LDS_BARRIER
}

// ---------------------------------------------------------------------------------------------------
// nms_scan_dq_kernel: the scan with the serial chain DECOUPLED from the bulk of the row ORs (nw <= 256, max_keep <= 4096).
// Measured on nms_scan_pf_kernel (clock64 per section, 4 x 12 000 candidates): a 64-box chunk cost 5.7 us = 1 us issuing the
// candidate prefetch + 2.5 us for the scalar resolve chain (one step per kept box) + 2.2 us for a memory round trip after it
// (rows of kept boxes that were not prefetched), with every wave of the workgroup in lock-step. Here:
//   wave 0 (resolver) owns the chain and nothing else. Lane i of chunk c has the diagonal word of row 64 c + i (symmetric, from
//     nms_mask_kernel): the greedy choice inside the chunk is the fixed point of "kept if no kept predecessor overlaps, removed
//     if one does", found by iterating over the undecided set with two ballots per round (rounds = overlap chain depth, a
//     handful) instead of one scalar step per kept box. It also ORs the rows of the boxes it keeps into the next D = 8 words of
//     the removed bitmap itself: lane (k, g) holds word c+1+k of rows 8g .. 8g+7 (one 64-B line per row) -> select by the kept
//     bits, LDS atomic OR into s_urg[c+1+k]. It issues no global load: a FEEDER wave (the last one) runs ahead of it, fetches
//     the diagonal + lookahead words of F = 4 chunks per memory round trip and hands them over through an LDS ring of 8 chunks.
//   waves 1 .. 2 NB (bulk, lane = word t of the bitmap, two waves per 64 words taking alternate 64-entry blocks of the list)
//     follow the published kept list at their own pace: up to 64 row loads in flight per lane, rows of chunk r applied to words t >= r + D + 1 only (the resolver covers r+1 .. r+D), result and
//     progress published in LDS. The resolver needs word c complete through chunk c-D-1 -- D chunks of slack -- and spins
//     on the owner's progress counter if it is not (no barrier inside the scan; LDS ops of a wave execute in order, so a
//     reader that sees a counter sees the data written before it).
// Same keep set, order and max_keep cut as the serial scan (the fixed point is unique).
// ---------------------------------------------------------------------------------------------------
// LDS mailbox accesses of nms_scan_dq_kernel through asm: a `volatile` access makes hipcc drain vmcnt as well (the row loads
// in flight), and ordering is all that is needed -- LDS ops of a wave execute in issue order
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
__device__ __forceinline__ int lds_ld32(const void* p) {
  int r; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(lds_addr(p)) : "memory"); return r;
}
__device__ __forceinline__ unsigned long long lds_ld64(const void* p) {
  unsigned long long r; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(lds_addr(p)) : "memory"); return r;
}
__device__ __forceinline__ i32x4 lds_ld128(const void* p) {
  i32x4 r; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(lds_addr(p)) : "memory"); return r;
}
__device__ __forceinline__ void lds_st32(void* p, int v) { asm volatile("ds_write_b32 %0, %1" :: "v"(lds_addr(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_st64(void* p, unsigned long long v) { asm volatile("ds_write_b64 %0, %1" :: "v"(lds_addr(p)), "v"(v) : "memory"); }

#define NMS_DQ_D 8
#ifndef NMS_DQ_RING
#define NMS_DQ_RING 12
#endif
#ifndef NMS_DQ_F
#define NMS_DQ_F 6
#endif
#ifndef NMS_DQ_BATCH
#define NMS_DQ_BATCH 64
#endif
__global__ void __launch_bounds__(640) nms_scan_dq_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                          const int* __restrict__ count, int cap, int nw,
                                                          const unsigned long long* __restrict__ mask, int max_keep,
                                                          int* __restrict__ keep_idx, int* __restrict__ keep_count,
                                                          float* __restrict__ out_boxes, float* __restrict__ out_scores) {
  __shared__ int s_keep[NMS_PF_MAXKEEP];
  __shared__ int s_cend[260];                               // s_cend[c + 1] = boxes kept through chunk c
  __shared__ unsigned long long s_bulk[2][256];             // word t: rows applied by the two bulk lanes that own it
  __shared__ unsigned long long s_urg[256 + NMS_DQ_D + 8];  // word t: rows applied by the resolver (chunks t-D .. t-1)
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_ring_raw[];
  unsigned long long(*s_ring)[9][64] = reinterpret_cast<unsigned long long(*)[9][64]>(s_ring_raw);   // [RING] per chunk: [0][i] diagonal word of row i; [1+j][lane (k, g)] word
                                                            // c+1+k of row 8g+j
  __shared__ __attribute__((aligned(8))) int s_done[4][2];  // bulk wave (k, sub): every kept-list entry of ITS blocks below this index is applied
  __shared__ __attribute__((aligned(16))) int s_pub[4];     // [0] kept-list length, [1] chunks resolved, [2] stop, [3] chunks fed
#define NMS_CBAR() asm volatile("" ::: "memory")
  const int b = blockIdx.x;
  const int n = count ? min(count[b], cap) : cap;
  const unsigned long long* mk = mask + (size_t)b * cap * nw;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nbulk = (nw + 63) / 64;
  const int nchunks = (n + 63) / 64;
  for (int i = tid; i < 512; i += blockDim.x) (&s_bulk[0][0])[i] = 0ull;
  for (int i = tid; i < 256 + NMS_DQ_D + 8; i += blockDim.x) s_urg[i] = 0ull;
  if (tid < 4) { s_done[tid][0] = 0; s_done[tid][1] = 64; s_pub[tid] = 0; }     // sub 1 owns nothing below entry 64
  if (tid == 0) s_cend[0] = 0;
  __syncthreads();
  // the scan is a latency chain on the step's critical path, co-resident with throughput kernels of the other streams: its waves
  // issue first on their SIMDs
  __builtin_amdgcn_s_setprio(3);

  if (wid == 0) {
    // ------------------------------------------------------------------ resolver
    const int l = lane, g = l >> 3;
    const unsigned long long lowmask = (1ull << l) - 1ull;
    int nkept = 0, fed = 0, done = 0;
    for (int c = 0; c < nchunks; ++c) {
      const int need = c > NMS_DQ_D ? s_cend[c - NMS_DQ_D] : 0;        // kept through chunk c - D - 1
      // (the counters only grow: poll again only when the cached value does not already satisfy the chunk)
      while (fed <= c) { fed = __builtin_amdgcn_readfirstlane(lds_ld32(&s_pub[3])); if (fed <= c) __builtin_amdgcn_s_sleep(1); }
      if ((c & 63) == 0) done = 0;                                     // next bulk wave's counter
      while (done < need) {
        const unsigned long long dd = lds_ld64(&s_done[c >> 6][0]);
        done = __builtin_amdgcn_readfirstlane(min((int)(unsigned)dd, (int)(unsigned)(dd >> 32)));
        if (done < need) __builtin_amdgcn_s_sleep(1);
      }
      NMS_CBAR();
      const unsigned long long cur = nms_uniform64(s_bulk[0][c] | s_bulk[1][c] | s_urg[c]);
      const unsigned long long(*rg)[64] = s_ring[c % NMS_DQ_RING];
      const int lim = n - c * 64;
      unsigned long long U = ~cur & (lim >= 64 ? ~0ull : ((1ull << lim) - 1ull));
      const unsigned long long pred = rg[0][l] & lowmask;
      unsigned long long uw[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) uw[i] = rg[1 + i][l];
      unsigned long long K = 0ull;
      while (U != 0ull) {
        const bool in_u = (U >> l) & 1ull;
        const bool rem = in_u && (pred & K) != 0ull;
        const bool kp = in_u && !rem && (pred & U) == 0ull;
        const unsigned long long kb = __ballot(kp), rb = __ballot(rem);
        K |= kb;
        U &= ~(kb | rb);
      }
      {   // max_keep cut: the first `room` kept boxes of the chunk
        const int room = max_keep - nkept;
        const int rank = __popcll(K & lowmask);
        K = __ballot(((K >> l) & 1ull) && rank < room);
        if ((K >> l) & 1ull) s_keep[nkept + rank] = c * 64 + l;
      }
      nkept += __popcll(K);
      if (l == 0) s_cend[c + 1] = nkept;
      {   // rows of the kept boxes -> words c+1 .. c+D
        unsigned long long acc = 0ull;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc |= ((K >> (g * 8 + i)) & 1ull) ? uw[i] : 0ull;
        if (acc != 0ull) {          // eight lanes per word: LDS atomic OR (no return) instead of three 64-bit xor-shuffles
          unsigned a = lds_addr(&s_urg[c + 1 + (l & 7)]);
          asm volatile("ds_or_b64 %0, %1" :: "v"(a), "v"(acc) : "memory");
        }
      }
      NMS_CBAR();
      if (l == 0) { lds_st32(&s_pub[0], nkept); lds_st32(&s_pub[1], c + 1); }
      if (nkept >= max_keep) break;
    }
    NMS_CBAR();
    if (l == 0) lds_st32(&s_pub[2], 1);
  } else if (wid == 1 + 2 * nbulk) {
    // ------------------------------------------------------------------ feeder: diagonal + lookahead words, F chunks per round trip
    const int l = lane, k = l & 7, g = l >> 3;
    for (int c0 = 0; c0 < nchunks; c0 += NMS_DQ_F) {
      bool stop = false;
      while (true) {
        const i32x4 pub = lds_ld128(s_pub);
        stop = __builtin_amdgcn_readfirstlane(pub[2]) != 0;
        if (stop || c0 + NMS_DQ_F <= __builtin_amdgcn_readfirstlane(pub[1]) + NMS_DQ_RING) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (stop) break;
      unsigned long long d[NMS_DQ_F], u[NMS_DQ_F][8];
#pragma unroll
      for (int f = 0; f < NMS_DQ_F; ++f) {
        const int cc = c0 + f;
        d[f] = mk[(size_t)min(cc * 64 + l, cap - 1) * nw + min(cc, nw - 1)];
        const int wq = min(cc + 1 + k, nw - 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) u[f][i] = mk[(size_t)min(cc * 64 + g * 8 + i, cap - 1) * nw + wq];
      }
#pragma unroll
      for (int f = 0; f < NMS_DQ_F; ++f) {
        unsigned long long(*rg)[64] = s_ring[(c0 + f) % NMS_DQ_RING];
        rg[0][l] = d[f];
#pragma unroll
        for (int i = 0; i < 8; ++i) rg[1 + i][l] = u[f][i];
      }
      NMS_CBAR();
      if (l == 0) lds_st32(&s_pub[3], c0 + NMS_DQ_F);
    }
  } else {
    // ------------------------------------------------------------------ bulk: lane = word t of the removed bitmap
    // two waves per 64 words: wave (kb, sub) applies the kept-list blocks j = sub, sub + 2, ... (64 entries each) -- 2 x 64 row
    // loads in flight per word when the resolver runs ahead (the scan is then bound by exactly that: latency x loads in flight)
    const int kb = (wid - 1) >> 1, sub = (wid - 1) & 1, t = kb * 64 + lane, tcl = min(t, nw - 1);
    const int mylim = t < nw ? t - NMS_DQ_D - 1 : -1;            // rows of chunks <= mylim are this lane's
    const int last_word = kb * 64 + 63;
    unsigned long long removed = 0ull;
    int base = sub * 64, off = 0;                                // current block of this wave, entries of it already applied
    while (true) {
      const i32x4 pub = lds_ld128(s_pub);
      const int stop = __builtin_amdgcn_readfirstlane(pub[2]);
      const int prog = __builtin_amdgcn_readfirstlane(pub[1]);
      const int avail = __builtin_amdgcn_readfirstlane(pub[0]);
      if (stop || prog > last_word) break;                       // nobody will read this wave's words any more
      const int pos = base + off;
      const int end = min(avail, base + 64);
      if (end <= pos) { __builtin_amdgcn_s_sleep(2); continue; }
      // all row loads of a batch are issued before the first use (unconditional, clamped to the batch: a select next to its
      // load makes hipcc wait for every load in turn); a short batch when the resolver is only a few boxes ahead
      auto batch = [&](auto BT) {
        constexpr int B = decltype(BT)::value;
        const int cnt = min(end - pos, B);
        unsigned long long v[B];
        int ch[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
          const int idx = __builtin_amdgcn_readfirstlane(s_keep[pos + min(u, cnt - 1)]);
          ch[u] = u < cnt ? (idx >> 6) : 0x7fffffff;
          v[u] = mk[(size_t)idx * nw + tcl];
        }
#pragma unroll
        for (int u = 0; u < B; ++u) removed |= ch[u] <= mylim ? v[u] : 0ull;
        off += cnt;
      };
      if (end - pos > 16) batch(std::integral_constant<int, NMS_DQ_BATCH>{});
      else batch(std::integral_constant<int, 16>{});
      if (off == 64) { base += 128; off = 0; }                   // the block in between is the other wave's
      lds_st64(&s_bulk[sub][t], removed);
      if (lane == 0) lds_st32(&s_done[kb][sub], base + off);
    }
  }
  __syncthreads();
  const int nkept = s_pub[0];
  if (tid == 0) keep_count[b] = nkept;
  for (int r = tid; r < nkept; r += blockDim.x) {
    int i = s_keep[r];
    size_t o = (size_t)b * max_keep + r;
    keep_idx[o] = i;
    if (out_boxes) *reinterpret_cast<f32x4*>(out_boxes + 4 * o) = *reinterpret_cast<const f32x4*>(boxes + ((size_t)b * cap + i) * 4);
    if (out_scores) out_scores[o] = scores[(size_t)b * cap + i];
  }
  for (int r = nkept + tid; r < max_keep; r += blockDim.x) {        // tail: index -1, zero box / score (no host-side fills)
    size_t o = (size_t)b * max_keep + r;
    keep_idx[o] = -1;
    if (out_boxes) *reinterpret_cast<f32x4*>(out_boxes + 4 * o) = f32x4{0.f, 0.f, 0.f, 0.f};
    if (out_scores) out_scores[o] = 0.f;
  }
#undef NMS_CBAR
}

extern "C" size_t unit_nms_workspace_bytes(int B, int cap) { return (size_t)B * cap * ((cap + 63) / 64) * 8; }

extern "C" int unit_nms(const float* boxes_sorted, const float* scores_sorted, const int* count, int B, int cap,
                        float thresh, int max_keep, int* keep_idx, int* keep_count, float* out_boxes, float* out_scores,
                        void* workspace, size_t workspace_bytes, void* stream) {
  int nw = (cap + 63) / 64;
  UNIT_CHECK_ARG(nw <= NMS_SCAN_THREADS, "nms: more than 65536 candidates per image");
  int scan_threads = nw <= 256 ? 256 : ((nw + 63) / 64) * 64;
  if (workspace_bytes < unit_nms_workspace_bytes(B, cap)) { unit_set_error("nms: workspace too small"); return UNIT_ERR_WORKSPACE; }
  if (B == 0) return UNIT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (cap > 0) {
    nms_mask_kernel<<<dim3((nw + 3) / 4, nw, B), 256, 0, st>>>(boxes_sorted, count, cap, nw, thresh, (unsigned long long*)workspace);
    UNIT_LAUNCH_CHECK();
  }
  // UNIT_NMS_SCAN: 2 (default) decoupled resolver / bulk waves, 1 lock-step scan with prefetch, 0 plain scan
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("UNIT_NMS_SCAN"); mode = e ? atoi(e) : 2; }
  if (nw <= 256 && cap > 0 && max_keep <= NMS_PF_MAXKEEP && mode == 2) {
    constexpr int ring_bytes = NMS_DQ_RING * 9 * 64 * 8;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)nms_scan_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ring_bytes); attr_set = true; }
    nms_scan_dq_kernel<<<B, 64 * (2 + 2 * ((nw + 63) / 64)), ring_bytes, st>>>(boxes_sorted, scores_sorted, count, cap, nw, (const unsigned long long*)workspace,
                                                                          max_keep, keep_idx, keep_count, out_boxes, out_scores);
    UNIT_LAUNCH_CHECK();
    return UNIT_OK;
  }
  // the other scans write the kept prefix only: index -1, zero box / score behind it
  (void)hipMemsetAsync(keep_idx, 0xFF, (size_t)B * max_keep * sizeof(int), st);
  if (out_boxes) (void)hipMemsetAsync(out_boxes, 0, (size_t)B * max_keep * 4 * sizeof(float), st);
  if (out_scores) (void)hipMemsetAsync(out_scores, 0, (size_t)B * max_keep * sizeof(float), st);
  if (nw <= 256 && cap > 0 && max_keep <= NMS_PF_MAXKEEP && mode == 1)
    nms_scan_pf_kernel<<<B, 256, 0, st>>>(boxes_sorted, scores_sorted, count, cap, nw, (const unsigned long long*)workspace,
                                          max_keep, keep_idx, keep_count, out_boxes, out_scores);
  else
    nms_scan_kernel<<<B, scan_threads, 0, st>>>(boxes_sorted, scores_sorted, count, cap, nw, (const unsigned long long*)workspace,
                                                max_keep, keep_idx, keep_count, out_boxes, out_scores);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
