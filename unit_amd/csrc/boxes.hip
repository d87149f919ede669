// boxes.hip -- anchors, pairwise-IoU + Matcher, explicit-permutation sub-sampling, Box2BoxTransform.
// All index-producing kernels are bit-exact restatements (compile with -ffp-contract=off: no FMA fusion).
#include "common.h"

// ---------------------------------------------------------------------------------------------------
// K5  DefaultAnchorGenerator grid (reached from modeling/proposal_generator/rpn.py:22):
// anchors[(y*W + x)*A + a] = cell[a] + (x*stride+off, y*stride+off, x*stride+off, y*stride+off)
// ---------------------------------------------------------------------------------------------------
__global__ void anchor_grid_kernel(float* __restrict__ out, int H, int W, int A, float stride, float offset,
                                   const float* __restrict__ cell) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * W * A) return;
  int a = idx % A; int p = idx / A; int x = p % W, y = p / W;
  float sx = offset * stride + (float)x * stride, sy = offset * stride + (float)y * stride;
  f32x4 c = *reinterpret_cast<const f32x4*>(cell + 4 * a);
  f32x4 o = {sx + c[0], sy + c[1], sx + c[2], sy + c[3]};
  *reinterpret_cast<f32x4*>(out + 4 * (size_t)idx) = o;
}
extern "C" int unit_anchor_grid(float* out, int H, int W, int A, float stride, float offset, const float* cell_dev,
                                void* stream) {
  int n = H * W * A;
  if (n == 0) return UNIT_OK;
  anchor_grid_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(out, H, W, A, stride, offset, cell_dev);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// K6+K7  pairwise_iou fused with Matcher (/root/reference/modeling/matcher.py:54-120).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float iou1(const f32x4 a, const f32x4 b) {
  float area1 = (a[2] - a[0]) * (a[3] - a[1]);
  float area2 = (b[2] - b[0]) * (b[3] - b[1]);
  float w = fminf(a[2], b[2]) - fmaxf(a[0], b[0]);
  float h = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
  w = w < 0.f ? 0.f : w;
  h = h < 0.f ? 0.f : h;
  float inter = w * h;
  return inter > 0.f ? inter / (area1 + area2 - inter) : 0.0f;
}

#define MATCH_MAX_GT 256
struct MatchCfg { float thr[4]; int lab[5]; int nthr; };

__global__ void iou_match_kernel(const float* __restrict__ gt, const int* __restrict__ gt_count, int Mcap,
                                 const float* __restrict__ boxes, long box_bstride, const int* __restrict__ box_count,
                                 int Ncap, MatchCfg cfg, int64_t* __restrict__ midx, int8_t* __restrict__ mlab,
                                 float* __restrict__ mval, unsigned* __restrict__ rowmax) {
  __shared__ f32x4 sgt[MATCH_MAX_GT];
  __shared__ unsigned srow[MATCH_MAX_GT];
  int b = blockIdx.y;
  int M = gt_count ? gt_count[b] : Mcap;
  int Nb = box_count ? box_count[b] : Ncap;
  for (int m = threadIdx.x; m < M; m += blockDim.x) {
    sgt[m] = *reinterpret_cast<const f32x4*>(gt + ((size_t)b * Mcap + m) * 4);
    srow[m] = 0u;
  }
  __syncthreads();
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = n < Nb;
  f32x4 bx = {0, 0, 0, 0};
  if (valid) bx = *reinterpret_cast<const f32x4*>(boxes + (size_t)b * box_bstride + (size_t)n * 4);
  float best = -1.f; int bi = 0;
  for (int m = 0; m < M; ++m) {
    float v = valid ? iou1(sgt[m], bx) : 0.f;
    if (v > best) { best = v; bi = m; }          // first occurrence wins ties (torch.max(dim=0))
    float wm = wave_reduce_max(v);
    if ((threadIdx.x & 63) == 0 && wm > 0.f) atomicMax(&srow[m], __float_as_uint(wm));
  }
  __syncthreads();
  for (int m = threadIdx.x; m < M; m += blockDim.x)
    if (srow[m]) atomicMax(&rowmax[(size_t)b * Mcap + m], srow[m]);
  if (n >= Ncap) return;
  size_t o = (size_t)b * Ncap + n;
  if (!valid) { midx[o] = 0; mlab[o] = -1; if (mval) mval[o] = 0.f; return; }
  if (M == 0) { midx[o] = 0; mlab[o] = (int8_t)cfg.lab[0]; if (mval) mval[o] = 0.f; return; }  // matcher.py:68-82
  int8_t lab = 1;
  float low = -INFINITY;
  for (int l = 0; l <= cfg.nthr; ++l) {
    float high = l < cfg.nthr ? cfg.thr[l] : INFINITY;
    if (best >= low && best < high) lab = (int8_t)cfg.lab[l];
    low = high;
  }
  midx[o] = bi; mlab[o] = lab; if (mval) mval[o] = best;
}

// matcher.py:100-120 set_low_quality_matches_: exact fp32 equality with each gt's row max, ties included.
__global__ void low_quality_kernel(const float* __restrict__ gt, const int* __restrict__ gt_count, int Mcap,
                                   const float* __restrict__ boxes, long box_bstride, const int* __restrict__ box_count,
                                   int Ncap, int8_t* __restrict__ mlab, const unsigned* __restrict__ rowmax) {
  __shared__ f32x4 sgt[MATCH_MAX_GT];
  __shared__ float srow[MATCH_MAX_GT];
  int b = blockIdx.y;
  int M = gt_count ? gt_count[b] : Mcap;
  int Nb = box_count ? box_count[b] : Ncap;
  for (int m = threadIdx.x; m < M; m += blockDim.x) {
    sgt[m] = *reinterpret_cast<const f32x4*>(gt + ((size_t)b * Mcap + m) * 4);
    srow[m] = __uint_as_float(rowmax[(size_t)b * Mcap + m]);
  }
  __syncthreads();
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= Nb) return;
  f32x4 bx = *reinterpret_cast<const f32x4*>(boxes + (size_t)b * box_bstride + (size_t)n * 4);
  bool hit = false;
  for (int m = 0; m < M; ++m) hit |= (iou1(sgt[m], bx) == srow[m]);
  if (hit) mlab[(size_t)b * Ncap + n] = 1;
}

extern "C" size_t unit_iou_match_workspace_bytes(int B, int Mcap) { return sizeof(unsigned) * (size_t)B * (Mcap > 0 ? Mcap : 1); }

extern "C" int unit_iou_match(const float* gt, const int* gt_count, int B, int Mcap, const float* boxes,
                              long box_batch_stride, const int* box_count, int Ncap, const float* thresholds,
                              const int* labels, int n_thresh, int allow_low_quality, int64_t* match_idx,
                              int8_t* match_label, float* match_val, void* workspace, size_t workspace_bytes,
                              void* stream) {
  UNIT_CHECK_ARG(Mcap <= MATCH_MAX_GT, "iou_match: Mcap > 256");
  UNIT_CHECK_ARG(n_thresh >= 1 && n_thresh <= 4, "iou_match: 1..4 thresholds");
  if (workspace_bytes < unit_iou_match_workspace_bytes(B, Mcap)) { unit_set_error("iou_match: workspace too small"); return UNIT_ERR_WORKSPACE; }
  if (B == 0 || Ncap == 0) return UNIT_OK;
  hipStream_t st = (hipStream_t)stream;
  MatchCfg cfg;
  for (int i = 0; i < n_thresh; ++i) cfg.thr[i] = thresholds[i];
  for (int i = 0; i <= n_thresh; ++i) cfg.lab[i] = labels[i];
  cfg.nthr = n_thresh;
  (void)hipMemsetAsync(workspace, 0, unit_iou_match_workspace_bytes(B, Mcap), st);
  dim3 grid(cdiv(Ncap, 256), B);
  iou_match_kernel<<<grid, 256, 0, st>>>(gt, gt_count, Mcap, boxes, box_batch_stride, box_count, Ncap, cfg, match_idx,
                                         match_label, match_val, (unsigned*)workspace);
  UNIT_LAUNCH_CHECK();
  if (allow_low_quality && Mcap > 0) {
    low_quality_kernel<<<grid, 256, 0, st>>>(gt, gt_count, Mcap, boxes, box_batch_stride, box_count, Ncap, match_label,
                                             (const unsigned*)workspace);
    UNIT_LAUNCH_CHECK();
  }
  return UNIT_OK;
}

// plain pairwise IoU matrix (weak_detector_fast_rcnn.py:327 style call; also a test hook)
__global__ void pairwise_iou_kernel(const float* __restrict__ b1, int M, const float* __restrict__ b2, int Nb, float* __restrict__ out) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  int m = blockIdx.y;
  if (n >= Nb) return;
  out[(size_t)m * Nb + n] = iou1(*reinterpret_cast<const f32x4*>(b1 + 4 * (size_t)m), *reinterpret_cast<const f32x4*>(b2 + 4 * (size_t)n));
}
extern "C" int unit_pairwise_iou(const float* b1, int M, const float* b2, int Nb, float* out, void* stream) {
  if (M == 0 || Nb == 0) return UNIT_OK;
  pairwise_iou_kernel<<<dim3(cdiv(Nb, 256), M), 256, 0, (hipStream_t)stream>>>(b1, M, b2, Nb, out);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// K8  subsample_labels with the explicit-permutation contract (detectron2.modeling.sampling via rpn.py:41,
// roi_heads.py:563): candidates are visited in the caller's permutation order (entries >= count
// are skipped); the first num_pos positives and first num_neg negatives are taken.
// One 1024-thread workgroup per image; order-preserving compaction by a block-wide scan.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int2 block_excl_scan2(int2 v, int2* total, int2* lds /* 16 + 1 entries */) {
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int2 inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int ax = __shfl_up(inc.x, o, 64), ay = __shfl_up(inc.y, o, 64);
    if (lane >= o) { inc.x += ax; inc.y += ay; }
  }
  if (lane == 63) lds[wid] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int2 run = {0, 0};
    for (int w = 0; w < nw; ++w) { int2 t = lds[w]; lds[w] = run; run.x += t.x; run.y += t.y; }
    lds[16] = run;
  }
  __syncthreads();
  int2 base = lds[wid];
  *total = lds[16];
  int2 ex = {base.x + inc.x - v.x, base.y + inc.y - v.y};
  __syncthreads();
  return ex;
}

template <typename LT>
__global__ void subsample_kernel(const LT* __restrict__ labels, const int* __restrict__ count, int Ncap,
                                 const int* __restrict__ perm, int Pcap, int num_samples, int max_pos, int bg_label,
                                 int8_t* __restrict__ out_labels, int* __restrict__ sampled_idx, int* __restrict__ out_counts) {
  __shared__ int2 lds[17];
  int b = blockIdx.x;
  int n = count ? count[b] : Ncap;
  const LT* lab = labels + (size_t)b * Ncap;
  const int* pm = perm + (size_t)b * Pcap;
  if (out_labels)
    for (int i = threadIdx.x; i < Ncap; i += blockDim.x) out_labels[(size_t)b * Ncap + i] = -1;
  if (sampled_idx)
    for (int i = threadIdx.x; i < num_samples; i += blockDim.x) sampled_idx[(size_t)b * num_samples + i] = -1;
  __syncthreads();
  int chunk = (Pcap + blockDim.x - 1) / blockDim.x;
  int i0 = threadIdx.x * chunk, i1 = min(Pcap, i0 + chunk);
  int2 c = {0, 0};
  for (int i = i0; i < i1; ++i) {
    int id = pm[i];
    if (id < n) {
      int l = (int)lab[id];
      if (l == bg_label) c.y++; else if (l != -1) c.x++;
    }
  }
  int2 tot;
  int2 ex = block_excl_scan2(c, &tot, lds);
  int num_pos = min(tot.x, max_pos);
  int num_neg = min(tot.y, num_samples - num_pos);
  for (int i = i0; i < i1; ++i) {
    int id = pm[i];
    if (id < n) {
      int l = (int)lab[id];
      if (l == bg_label) {
        if (ex.y < num_neg) {
          if (out_labels) out_labels[(size_t)b * Ncap + id] = 0;
          if (sampled_idx) sampled_idx[(size_t)b * num_samples + num_pos + ex.y] = id;
        }
        ex.y++;
      } else if (l != -1) {
        if (ex.x < num_pos) {
          if (out_labels) out_labels[(size_t)b * Ncap + id] = 1;
          if (sampled_idx) sampled_idx[(size_t)b * num_samples + ex.x] = id;
        }
        ex.x++;
      }
    }
  }
  if (threadIdx.x == 0) { out_counts[2 * b] = num_pos; out_counts[2 * b + 1] = num_neg; }
}

extern "C" int unit_subsample_labels(const void* labels, int labels_are_int64, const int* count, int B, int Ncap,
                                     const int* perm, int Pcap, int num_samples, int max_pos, int bg_label,
                                     int8_t* out_labels, int* sampled_idx, int* out_counts, void* stream) {
  if (B == 0) return UNIT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (labels_are_int64)
    subsample_kernel<int64_t><<<B, 1024, 0, st>>>((const int64_t*)labels, count, Ncap, perm, Pcap, num_samples, max_pos, bg_label, out_labels, sampled_idx, out_counts);
  else
    subsample_kernel<int8_t><<<B, 1024, 0, st>>>((const int8_t*)labels, count, Ncap, perm, Pcap, num_samples, max_pos, bg_label, out_labels, sampled_idx, out_counts);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// K9  Box2BoxTransform.get_deltas / apply_deltas (rpn.py:70, fast_rcnn.py:71, :455-468)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 box_encode1(f32x4 s, f32x4 t, f32x4 w) {
  float sw = s[2] - s[0], sh = s[3] - s[1];
  float scx = s[0] + 0.5f * sw, scy = s[1] + 0.5f * sh;
  float tw = t[2] - t[0], th = t[3] - t[1];
  float tcx = t[0] + 0.5f * tw, tcy = t[1] + 0.5f * th;
  f32x4 d = {w[0] * (tcx - scx) / sw, w[1] * (tcy - scy) / sh, w[2] * logf(tw / sw), w[3] * logf(th / sh)};
  return d;
}
__device__ __forceinline__ f32x4 box_decode1(f32x4 d, f32x4 b, f32x4 wt, float clampv) {
  float w = b[2] - b[0], h = b[3] - b[1];
  float cx = b[0] + 0.5f * w, cy = b[1] + 0.5f * h;
  float dx = d[0] / wt[0], dy = d[1] / wt[1];
  float dw = fminf(d[2] / wt[2], clampv), dh = fminf(d[3] / wt[3], clampv);
  float pcx = dx * w + cx, pcy = dy * h + cy;
  float pw = expf(dw) * w, ph = expf(dh) * h;
  f32x4 o = {pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph};
  return o;
}

__global__ void box_encode_kernel(const float* __restrict__ src, const float* __restrict__ tgt, f32x4 w, float* __restrict__ out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  *reinterpret_cast<f32x4*>(out + 4 * (size_t)i) =
      box_encode1(*reinterpret_cast<const f32x4*>(src + 4 * (size_t)i), *reinterpret_cast<const f32x4*>(tgt + 4 * (size_t)i), w);
}
extern "C" int unit_box_encode(const float* src, const float* tgt, const float* weights4, float* out, int n, void* stream) {
  if (n == 0) return UNIT_OK;
  f32x4 w = {weights4[0], weights4[1], weights4[2], weights4[3]};
  box_encode_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(src, tgt, w, out, n);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// deltas [n][ld] with K boxes per row starting at column col0 ; boxes [n][4] ; out [n][K][4]
__global__ void box_decode_kernel(const float* __restrict__ deltas, int ld, int col0, int K, const float* __restrict__ boxes,
                                  f32x4 w, float clampv, float* __restrict__ out, int n) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * K) return;
  int i = idx / K, k = idx - i * K;
  const float* dp = deltas + (size_t)i * ld + col0 + 4 * k;
  f32x4 d = {dp[0], dp[1], dp[2], dp[3]};
  *reinterpret_cast<f32x4*>(out + 4 * (size_t)idx) = box_decode1(d, *reinterpret_cast<const f32x4*>(boxes + 4 * (size_t)i), w, clampv);
}
extern "C" int unit_box_decode(const float* deltas, int ld, int col0, int K, const float* boxes, const float* weights4,
                               float scale_clamp, float* out, int n, void* stream) {
  if (n == 0 || K == 0) return UNIT_OK;
  f32x4 w = {weights4[0], weights4[1], weights4[2], weights4[3]};
  box_decode_kernel<<<cdiv(n * K, 256), 256, 0, (hipStream_t)stream>>>(deltas, ld, col0, K, boxes, w, scale_clamp, out, n);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// a6 (part)  find_top_rpn_proposals decode/clip/filter (reached from rpn.py:48):
// for the first `topk` entries of the score-sorted anchor list: apply_deltas, clip to the un-padded image,
// drop non-finite / empty boxes, and compact preserving order. One 1024-thread workgroup per image.
// head [B][HW][ld]: logits at col a, deltas at delta_col0 + 4a .. ; sorted_idx [B][Ncap] (anchor ids).
// ---------------------------------------------------------------------------------------------------
__global__ void rpn_decode_select_kernel(const float* __restrict__ head, long head_bstride, int ld, int A, int delta_col0,
                                         const float* __restrict__ anchors, const int* __restrict__ sorted_idx,
                                         const float* __restrict__ sorted_logit, int Ncap, int topk,
                                         const float* __restrict__ image_hw, float clampv, float min_size,
                                         float* __restrict__ cand_boxes, float* __restrict__ cand_scores,
                                         int* __restrict__ cand_count) {
  __shared__ int2 lds[17];
  int b = blockIdx.x;
  float imh = image_hw[2 * b], imw = image_hw[2 * b + 1];
  int chunk = (topk + blockDim.x - 1) / blockDim.x;
  int i0 = threadIdx.x * chunk, i1 = min(topk, i0 + chunk);
  const f32x4 wt = {1.f, 1.f, 1.f, 1.f};
  int2 c = {0, 0};
  for (int pass = 0; pass < 2; ++pass) {
    int2 ex = {0, 0}, tot;
    if (pass == 1) ex = block_excl_scan2(c, &tot, lds);
    for (int i = i0; i < i1; ++i) {
      int aid = sorted_idx[(size_t)b * Ncap + i];
      float sc = sorted_logit[(size_t)b * Ncap + i];
      int pix = aid / A, a = aid - pix * A;
      const float* dp = head + (size_t)b * head_bstride + (size_t)pix * ld + delta_col0 + 4 * a;
      f32x4 d = {dp[0], dp[1], dp[2], dp[3]};
      f32x4 bx = box_decode1(d, *reinterpret_cast<const f32x4*>(anchors + 4 * (size_t)aid), wt, clampv);
      bool fin = isfinite(bx[0]) && isfinite(bx[1]) && isfinite(bx[2]) && isfinite(bx[3]) && isfinite(sc);
      bx[0] = fminf(fmaxf(bx[0], 0.f), imw); bx[1] = fminf(fmaxf(bx[1], 0.f), imh);
      bx[2] = fminf(fmaxf(bx[2], 0.f), imw); bx[3] = fminf(fmaxf(bx[3], 0.f), imh);
      bool keep = fin && (bx[2] - bx[0] > min_size) && (bx[3] - bx[1] > min_size);
      if (pass == 0) { c.x += keep ? 1 : 0; }
      else if (keep) {
        size_t o = (size_t)b * topk + ex.x;
        *reinterpret_cast<f32x4*>(cand_boxes + 4 * o) = bx;
        cand_scores[o] = sc;
        ex.x++;
      }
    }
    if (pass == 1 && threadIdx.x == 0) cand_count[b] = tot.x;
  }
}

// Same result for topk <= 16 * 1024: thread t decodes entries t, t+1024, ... (all gathers of a thread independent and in
// flight together, one decode per entry), keeps are counted per (row of 1024, wave) with ballots, one LDS scan gives every
// (row, wave) its ordered output offset.
#define DEC_ROWS 16
__global__ void __launch_bounds__(1024) rpn_decode_select_rows_kernel(const float* __restrict__ head, long head_bstride, int ld, int A,
                                                                      int delta_col0, const float* __restrict__ anchors,
                                                                      const int* __restrict__ sorted_idx,
                                                                      const float* __restrict__ sorted_logit, int Ncap, int topk,
                                                                      const float* __restrict__ image_hw, float clampv, float min_size,
                                                                      float* __restrict__ cand_boxes, float* __restrict__ cand_scores,
                                                                      int* __restrict__ cand_count) {
  __builtin_amdgcn_s_setprio(2);   // proposal chain = critical path of the step
  __shared__ int cnt[DEC_ROWS * 16 + 1];
  int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float imh = image_hw[2 * b], imw = image_hw[2 * b + 1];
  const f32x4 wt = {1.f, 1.f, 1.f, 1.f};
  int aid[DEC_ROWS]; float sc[DEC_ROWS]; f32x4 bx[DEC_ROWS]; unsigned long long bal[DEC_ROWS];
#pragma unroll
  for (int k = 0; k < DEC_ROWS; ++k) {
    int i = k * 1024 + tid;
    aid[k] = i < topk ? sorted_idx[(size_t)b * Ncap + i] : -1;
    sc[k] = i < topk ? sorted_logit[(size_t)b * Ncap + i] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < DEC_ROWS; ++k) {
    bool keep = false;
    if (aid[k] >= 0) {
      int pix = aid[k] / A, a = aid[k] - pix * A;
      const float* dp = head + (size_t)b * head_bstride + (size_t)pix * ld + delta_col0 + 4 * a;
      f32x4 d = {dp[0], dp[1], dp[2], dp[3]};
      f32x4 v = box_decode1(d, *reinterpret_cast<const f32x4*>(anchors + 4 * (size_t)aid[k]), wt, clampv);
      bool fin = isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]) && isfinite(v[3]) && isfinite(sc[k]);
      v[0] = fminf(fmaxf(v[0], 0.f), imw); v[1] = fminf(fmaxf(v[1], 0.f), imh);
      v[2] = fminf(fmaxf(v[2], 0.f), imw); v[3] = fminf(fmaxf(v[3], 0.f), imh);
      keep = fin && (v[2] - v[0] > min_size) && (v[3] - v[1] > min_size);
      bx[k] = v;
    }
    bal[k] = __ballot(keep);
    if (lane == 0) cnt[k * 16 + wid] = __popcll(bal[k]);
  }
  __syncthreads();
  // exclusive scan of the DEC_ROWS*16 = 256 (row, wave) counts by waves 0-3
  __shared__ int wtot[4];
  int v = 0, inc = 0;
  if (tid < DEC_ROWS * 16) {
    v = cnt[tid]; inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wtot[wid] = inc;
  }
  __syncthreads();
  if (tid < DEC_ROWS * 16) {
    int base = 0;
    for (int w = 0; w < wid; ++w) base += wtot[w];
    cnt[tid] = base + inc - v;
    if (tid == DEC_ROWS * 16 - 1) cand_count[b] = base + inc;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < DEC_ROWS; ++k) {
    if ((bal[k] >> lane) & 1ull) {
      size_t o = (size_t)b * topk + cnt[k * 16 + wid] + __popcll(bal[k] & ((1ull << lane) - 1ull));
      *reinterpret_cast<f32x4*>(cand_boxes + 4 * o) = bx[k];
      cand_scores[o] = sc[k];
    }
  }
}

extern "C" int unit_rpn_decode_select(const float* head, long head_batch_stride, int ld, int A, int delta_col0,
                                      const float* anchors, const int* sorted_idx, const float* sorted_logit, int B,
                                      int Ncap, int topk, const float* image_hw_dev, float scale_clamp, float min_size,
                                      float* cand_boxes, float* cand_scores, int* cand_count, void* stream) {
  UNIT_CHECK_ARG(topk <= Ncap, "rpn_decode_select: topk > Ncap");
  if (B == 0) return UNIT_OK;
  if (topk <= DEC_ROWS * 1024) {
    rpn_decode_select_rows_kernel<<<B, 1024, 0, (hipStream_t)stream>>>(head, head_batch_stride, ld, A, delta_col0, anchors, sorted_idx,
                                                                    sorted_logit, Ncap, topk, image_hw_dev, scale_clamp, min_size,
                                                                    cand_boxes, cand_scores, cand_count);
    UNIT_LAUNCH_CHECK();
    return UNIT_OK;
  }
  rpn_decode_select_kernel<<<B, 1024, 0, (hipStream_t)stream>>>(head, head_batch_stride, ld, A, delta_col0, anchors, sorted_idx,
                                                             sorted_logit, Ncap, topk, image_hw_dev, scale_clamp, min_size,
                                                             cand_boxes, cand_scores, cand_count);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// Matcher on a precomputed M x N quality matrix -- the exact call shape of /root/reference/modeling/matcher.py:54
// (`Matcher.__call__(match_quality_matrix)`), used by the plugin-surface `unit_amd.modeling.Matcher`.
// ---------------------------------------------------------------------------------------------------
__global__ void matrix_rowmax_kernel(const float* __restrict__ q, int M, int N, float* __restrict__ rowmax) {
  __shared__ float lds[16];
  int m = blockIdx.x;
  float v = -INFINITY;
  for (int n = threadIdx.x; n < N; n += blockDim.x) v = fmaxf(v, q[(size_t)m * N + n]);
  v = wave_reduce_max(v);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) { float mm = -INFINITY; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) mm = fmaxf(mm, lds[w]); rowmax[m] = mm; }
}
__global__ void matrix_match_kernel(const float* __restrict__ q, int M, int N, MatchCfg cfg, int allow_lq, const float* __restrict__ rowmax,
                                    int64_t* __restrict__ midx, int8_t* __restrict__ mlab, float* __restrict__ mval) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  if (M == 0) { midx[n] = 0; mlab[n] = (int8_t)cfg.lab[0]; mval[n] = 0.f; return; }
  float best = q[n]; int bi = 0; bool hit = (best == rowmax[0]);
  for (int m = 1; m < M; ++m) {
    float v = q[(size_t)m * N + n];
    if (v > best) { best = v; bi = m; }
    hit |= (v == rowmax[m]);
  }
  int8_t lab = 1; float low = -INFINITY;
  for (int l = 0; l <= cfg.nthr; ++l) {
    float high = l < cfg.nthr ? cfg.thr[l] : INFINITY;
    if (best >= low && best < high) lab = (int8_t)cfg.lab[l];
    low = high;
  }
  if (allow_lq && hit) lab = 1;
  midx[n] = bi; mlab[n] = lab; mval[n] = best;
}
extern "C" int unit_match_matrix(const float* q, int M, int N, const float* thresholds, const int* labels, int n_thresh,
                                 int allow_low_quality, int64_t* match_idx, int8_t* match_label, float* match_val,
                                 float* rowmax_ws, void* stream) {
  UNIT_CHECK_ARG(n_thresh >= 1 && n_thresh <= 4, "match_matrix: 1..4 thresholds");
  if (N == 0) return UNIT_OK;
  hipStream_t st = (hipStream_t)stream;
  MatchCfg cfg;
  for (int i = 0; i < n_thresh; ++i) cfg.thr[i] = thresholds[i];
  for (int i = 0; i <= n_thresh; ++i) cfg.lab[i] = labels[i];
  cfg.nthr = n_thresh;
  if (M > 0) { matrix_rowmax_kernel<<<M, 256, 0, st>>>(q, M, N, rowmax_ws); UNIT_LAUNCH_CHECK(); }
  matrix_match_kernel<<<cdiv(N, 256), 256, 0, st>>>(q, M, N, cfg, allow_low_quality, rowmax_ws, match_idx, match_label, match_val);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
