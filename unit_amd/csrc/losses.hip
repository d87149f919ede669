// losses.hip -- fused forward+backward loss kernels and the small index plumbing of the RoI stage.
// Every kernel emits the loss value AND d(loss)/d(input) (already scaled by the normaliser and `gscale`),
// in the layout/dtype the Linear / 1x1-conv dgrad+wgrad kernels consume, so no autograd graph is needed.
#include "common.h"

template <typename T> __device__ __forceinline__ void st(T* p, float v) { *p = (T)v; }

__device__ __forceinline__ float block_sum(float v, float* lds /* >= 17 floats */) {
  v = wave_reduce_sum(v);
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) lds[wid] = v;
  __syncthreads();
  if (threadIdx.x == 0) { float s = 0.f; for (int w = 0; w < nw; ++w) s += lds[w]; lds[16] = s; }
  __syncthreads();
  return lds[16];
}

// Sum of one non-negative partial per workgroup, bit-reproducible, in ONE atomic per workgroup and no fence: the partial goes into the low
// 56 bits of a 64-bit accumulator as fixed point (2^-24 units: integer addition is associative, so the arrival order does not matter), the
// arrival count into the top 8 bits. The workgroup whose atomicAdd returns count == nblk - 1 holds the complete sum in (returned value +
// its own addend); it leaves the accumulator zero for the next launch (the caller zeroes it once). nblk <= 255.
// A partial that is NaN, infinite or >= 2^23 (a diverged model) is entered as 2^47 units: 255 of them still do not carry into the count, the
// ticket keeps working, the accumulator is handed back zero, and a total >= 2^47 units is reported as NaN -- the loss the single-workgroup form
// would have produced and TrainerNoMeta.loss_dict()'s anomaly check looks for. (History: 2^-36 units overflowed into the count at 20 000 rows
// x loss 60; a NaN partial converted to garbage, broke the ticket and left the accumulator dirty for every later launch.)
__device__ __forceinline__ bool packed_sum_finish(unsigned long long* acc, float partial, int nblk, float* total) {
  const unsigned long long POISON = 1ull << 47;
  bool ok = partial >= 0.f && partial < 8388608.f;          // false for NaN
  unsigned long long q = (ok ? (unsigned long long)((double)partial * 16777216.0) : POISON) + (1ull << 56);
  unsigned long long old = atomicAdd(acc, q);
  if ((int)(old >> 56) != nblk - 1) return false;
  unsigned long long sum = (old + q) & ((1ull << 56) - 1);
  *total = sum >= POISON ? __builtin_nanf("") : (float)((double)sum * (1.0 / 16777216.0));
  *acc = 0ull;
  return true;
}

// ---------------------------------------------------------------------------------------------------
// a5  WSRPN.losses  modeling/proposal_generator/rpn.py:55-101
//   loss_rpn_cls = sum_{label>=0} BCEwithlogits(logit, label) / (256*N) ; loss_rpn_loc = sum_{label==1} |d - t| / (256*N)
//   t = get_deltas(anchor, gt[matched_idx]) weights (1,1,1,1).  head [B][HW][ld] fp32: logit col a, delta col dcol0+4a+j.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 encode1(f32x4 s, f32x4 t, f32x4 w) {
  float sw = s[2] - s[0], sh = s[3] - s[1];
  float scx = s[0] + 0.5f * sw, scy = s[1] + 0.5f * sh;
  float tw = t[2] - t[0], th = t[3] - t[1];
  float tcx = t[0] + 0.5f * tw, tcy = t[1] + 0.5f * th;
  f32x4 d = {w[0] * (tcx - scx) / sw, w[1] * (tcy - scy) / sh, w[2] * logf(tw / sw), w[3] * logf(th / sh)};
  return d;
}

template <typename TD>
__global__ void rpn_loss_kernel(const float* __restrict__ head, int ld, int A, int dcol0, const int8_t* __restrict__ labels,
                                const int64_t* __restrict__ midx, const float* __restrict__ gt, int Mcap,
                                const float* __restrict__ anchors, int Ncap, float inv_norm, float gscale, float w_cls, float w_loc,
                                float* __restrict__ loss2, TD* __restrict__ dhead, float* __restrict__ scratch) {
  __shared__ float lds[17];
  __shared__ int s_last;
  int b = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float lc = 0.f, ll = 0.f;
  if (i < Ncap) {
    int pix = i / A, a = i - pix * A;
    size_t row = ((size_t)b * (Ncap / A) + pix) * ld;
    int lab = labels[(size_t)b * Ncap + i];
    float dl = 0.f; f32x4 dd = {0, 0, 0, 0};
    if (lab >= 0) {
      float x = head[row + a], y = (float)lab;
      // binary_cross_entropy_with_logits: max(x,0) - x*y + log1p(exp(-|x|))
      lc = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
      float sg = 1.f / (1.f + expf(-x));
      dl = (sg - y) * inv_norm * gscale * w_cls;
    }
    if (lab == 1) {
      f32x4 an = *reinterpret_cast<const f32x4*>(anchors + 4 * (size_t)i);
      f32x4 g = *reinterpret_cast<const f32x4*>(gt + ((size_t)b * Mcap + midx[(size_t)b * Ncap + i]) * 4);
      const f32x4 w1 = {1.f, 1.f, 1.f, 1.f};
      f32x4 t = encode1(an, g, w1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float df = head[row + dcol0 + 4 * a + j] - t[j];
        ll += fabsf(df);
        dd[j] = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * inv_norm * gscale * w_loc;
      }
    }
    st(dhead + row + a, dl);
#pragma unroll
    for (int j = 0; j < 4; ++j) st(dhead + row + dcol0 + 4 * a + j, dd[j]);
  }
  float sc = block_sum(lc, lds);
  float sl = block_sum(ll, lds);
  // Bit-reproducible sums: every workgroup parks its two partials in its own scratch slot, the last one to arrive adds all slots in
  // a fixed order. Hand-off through device-scope atomics only (a slot is zero and gets exactly one atomicAdd, the reader fetches it
  // with atomicAdd(.., 0)): they execute at the memory side, so no assumption about the per-XCD L2s is needed.
  const int nblk = gridDim.x * gridDim.y, blk = blockIdx.y * gridDim.x + blockIdx.x;
  unsigned* ticket = reinterpret_cast<unsigned*>(scratch + 2 * nblk);
  if (threadIdx.x == 0) {
    atomicAdd(scratch + 2 * blk, sc);
    atomicAdd(scratch + 2 * blk + 1, sl);
    __threadfence();
    s_last = (atomicAdd(ticket, 1u) == (unsigned)(nblk - 1)) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  float tc = 0.f, tl = 0.f;
  for (int j = threadIdx.x; j < nblk; j += blockDim.x) {        // fixed assignment of slots to threads, fixed order per thread
    tc += atomicAdd(scratch + 2 * j, 0.f);
    tl += atomicAdd(scratch + 2 * j + 1, 0.f);
  }
  tc = block_sum(tc, lds);
  tl = block_sum(tl, lds);
  if (threadIdx.x == 0) { loss2[0] = tc * inv_norm * w_cls; loss2[1] = tl * inv_norm * w_loc; }
}

// two partial sums per workgroup + the arrival counter
extern "C" size_t unit_rpn_loss_scratch_bytes(int B, int Ncap) { return ((size_t)2 * cdiv(Ncap, 256) * (B > 0 ? B : 1) + 1) * sizeof(float); }

extern "C" int unit_rpn_loss_w(const float* head, int ld, int A, int dcol0, const int8_t* labels, const int64_t* match_idx,
                               const float* gt_boxes, int Mcap, const float* anchors, int B, int Ncap, float normalizer,
                               float gscale, float w_cls, float w_loc, float* loss2, void* dhead, int dhead_dtype, float* scratch,
                               size_t scratch_bytes, void* stream);
extern "C" int unit_rpn_loss(const float* head, int ld, int A, int dcol0, const int8_t* labels, const int64_t* match_idx,
                             const float* gt_boxes, int Mcap, const float* anchors, int B, int Ncap, float normalizer,
                             float gscale, float* loss2, void* dhead, int dhead_dtype, float* scratch, size_t scratch_bytes,
                             void* stream) {
  return unit_rpn_loss_w(head, ld, A, dcol0, labels, match_idx, gt_boxes, Mcap, anchors, B, Ncap, normalizer, gscale, 1.0f, 1.0f, loss2, dhead,
                         dhead_dtype, scratch, scratch_bytes, stream);
}

// w_cls / w_loc: the `loss_weight` dictionary of Detectron2's RPN (rpn.py:100 `losses = {k: v * self.loss_weight.get(k, 1.0)}`;
// MODEL.RPN.LOSS_WEIGHT and LOSS_WEIGHT * BBOX_REG_LOSS_WEIGHT): both the loss values and their gradients carry them
extern "C" int unit_rpn_loss_w(const float* head, int ld, int A, int dcol0, const int8_t* labels, const int64_t* match_idx,
                               const float* gt_boxes, int Mcap, const float* anchors, int B, int Ncap, float normalizer,
                               float gscale, float w_cls, float w_loc, float* loss2, void* dhead, int dhead_dtype, float* scratch,
                               size_t scratch_bytes, void* stream) {
  UNIT_CHECK_ARG(Ncap % A == 0, "rpn_loss: Ncap % A != 0");
  hipStream_t s = (hipStream_t)stream;
  size_t need = unit_rpn_loss_scratch_bytes(B, Ncap);
  if (scratch == nullptr || scratch_bytes < need) { unit_set_error("rpn_loss: scratch too small"); return UNIT_ERR_WORKSPACE; }
  (void)hipMemsetAsync(loss2, 0, 2 * sizeof(float), s);
  (void)hipMemsetAsync(scratch, 0, need, s);
  size_t esz = dhead_dtype == UNIT_BF16 ? 2 : 4;
  (void)hipMemsetAsync(dhead, 0, (size_t)B * (Ncap / A) * ld * esz, s);   // pad columns stay zero
  if (B == 0 || Ncap == 0) return UNIT_OK;
  dim3 grid(cdiv(Ncap, 256), B);
  float inv = 1.0f / normalizer;
  if (dhead_dtype == UNIT_BF16)
    rpn_loss_kernel<bf16_t><<<grid, 256, 0, s>>>(head, ld, A, dcol0, labels, match_idx, gt_boxes, Mcap, anchors, Ncap, inv, gscale, w_cls, w_loc, loss2, (bf16_t*)dhead, scratch);
  else
    rpn_loss_kernel<float><<<grid, 256, 0, s>>>(head, ld, A, dcol0, labels, match_idx, gt_boxes, Mcap, anchors, Ncap, inv, gscale, w_cls, w_loc, loss2, (float*)dhead, scratch);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// a7 (tail)  ROIHeads.label_and_sample_proposals plumbing (roi_heads.py:563)
// ---------------------------------------------------------------------------------------------------
// cat[b] = proposals[b][0:pc] ++ gt[b][0:gc]  (add_ground_truth_to_proposals), count_out = pc + gc
__global__ void append_gt_kernel(const float* __restrict__ props, const int* __restrict__ pcount, int Pcap, const float* __restrict__ gt,
                                 const int* __restrict__ gcount, int Mcap, float* __restrict__ cat, int* __restrict__ ccount) {
  int b = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int pc = min(pcount[b], Pcap), gc = gcount[b];
  int cap = Pcap + Mcap;
  if (i >= cap) return;
  f32x4 v = {0, 0, 0, 0};
  if (i < pc) v = *reinterpret_cast<const f32x4*>(props + ((size_t)b * Pcap + i) * 4);
  else if (i < pc + gc) v = *reinterpret_cast<const f32x4*>(gt + ((size_t)b * Mcap + (i - pc)) * 4);
  *reinterpret_cast<f32x4*>(cat + ((size_t)b * cap + i) * 4) = v;
  if (i == 0) ccount[b] = pc + gc;
}
extern "C" int unit_append_gt(const float* props, const int* pcount, int Pcap, const float* gt, const int* gcount, int Mcap, int B,
                              float* cat, int* ccount, void* stream) {
  if (B == 0) return UNIT_OK;
  append_gt_kernel<<<dim3(cdiv(Pcap + Mcap, 256), B), 256, 0, (hipStream_t)stream>>>(props, pcount, Pcap, gt, gcount, Mcap, cat, ccount);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// cls[i] = gt_classes[idx[i]] ; label==0 -> K ; label==-1 -> -1 ; no gt -> K      (int64 for the sub-sampler)
__global__ void roi_classes_kernel(const int64_t* __restrict__ midx, const int8_t* __restrict__ mlab, const int* __restrict__ count,
                                   const int64_t* __restrict__ gt_classes, const int* __restrict__ gcount, int Mcap, int Ncap, int K,
                                   int64_t* __restrict__ cls) {
  int b = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Ncap) return;
  size_t o = (size_t)b * Ncap + i;
  if (i >= count[b]) { cls[o] = -1; return; }
  int64_t c;
  if (gcount[b] > 0) {
    c = gt_classes[(size_t)b * Mcap + midx[o]];
    int l = mlab[o];
    if (l == 0) c = K; else if (l == -1) c = -1;
  } else c = K;
  cls[o] = c;
}
extern "C" int unit_roi_classes(const int64_t* match_idx, const int8_t* match_label, const int* count, const int64_t* gt_classes,
                                const int* gcount, int Mcap, int B, int Ncap, int K, int64_t* cls, void* stream) {
  if (B == 0 || Ncap == 0) return UNIT_OK;
  roi_classes_kernel<<<dim3(cdiv(Ncap, 256), B), 256, 0, (hipStream_t)stream>>>(match_idx, match_label, count, gt_classes, gcount, Mcap, Ncap, K, cls);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// gather the sampled RoIs: rois[b*S+i] = (b, cat[sidx]) ; roi_cls ; roi_gt = gt[midx[sidx]] ; empty slots: zeros / -1
__global__ void gather_rois_kernel(const float* __restrict__ cat, int Ncap, const int* __restrict__ sidx, int S,
                                   const int64_t* __restrict__ cls, const int64_t* __restrict__ midx, const float* __restrict__ gt,
                                   const int* __restrict__ gcount, int Mcap, float* __restrict__ rois, int* __restrict__ roi_cls,
                                   float* __restrict__ roi_gt) {
  int b = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S) return;
  size_t o = (size_t)b * S + i;
  int id = sidx[o];
  f32x4 bx = {0, 0, 0, 0}, g = {0, 0, 0, 0};
  int c = -1;
  if (id >= 0) {
    bx = *reinterpret_cast<const f32x4*>(cat + ((size_t)b * Ncap + id) * 4);
    c = (int)cls[(size_t)b * Ncap + id];
    if (gcount[b] > 0) g = *reinterpret_cast<const f32x4*>(gt + ((size_t)b * Mcap + midx[(size_t)b * Ncap + id]) * 4);
  }
  rois[o * 5 + 0] = (float)b; rois[o * 5 + 1] = bx[0]; rois[o * 5 + 2] = bx[1]; rois[o * 5 + 3] = bx[2]; rois[o * 5 + 4] = bx[3];
  roi_cls[o] = c;
  *reinterpret_cast<f32x4*>(roi_gt + o * 4) = g;
}
extern "C" int unit_gather_rois(const float* cat, int Ncap, const int* sampled_idx, int S, const int64_t* cls, const int64_t* match_idx,
                                const float* gt, const int* gcount, int Mcap, int B, float* rois, int* roi_cls, float* roi_gt,
                                void* stream) {
  if (B == 0 || S == 0) return UNIT_OK;
  gather_rois_kernel<<<dim3(cdiv(S, 256), B), 256, 0, (hipStream_t)stream>>>(cat, Ncap, sampled_idx, S, cls, match_idx, gt, gcount, Mcap, rois, roi_cls, roi_gt);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// weak proposals: first `S` RPN outputs per image (roi_heads.py:566-572) -> rois (b, box), valid flag via cls (0 / -1)
__global__ void first_k_rois_kernel(const float* __restrict__ props, const int* __restrict__ pcount, int Pcap, int S, int b0,
                                    float* __restrict__ rois, int* __restrict__ valid) {
  int b = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S) return;
  size_t o = (size_t)b * S + i;
  bool ok = i < min(pcount[b], Pcap);
  f32x4 bx = {0, 0, 0, 0};
  if (ok) bx = *reinterpret_cast<const f32x4*>(props + ((size_t)b * Pcap + i) * 4);
  rois[o * 5 + 0] = (float)(b + b0); rois[o * 5 + 1] = bx[0]; rois[o * 5 + 2] = bx[1]; rois[o * 5 + 3] = bx[2]; rois[o * 5 + 4] = bx[3];
  valid[o] = ok ? 0 : -1;
}
extern "C" int unit_first_k_rois(const float* props, const int* pcount, int Pcap, int S, int B, int batch_index_offset,
                                 float* rois, int* valid, void* stream) {
  if (B == 0 || S == 0) return UNIT_OK;
  first_k_rois_kernel<<<dim3(cdiv(S, 256), B), 256, 0, (hipStream_t)stream>>>(props, pcount, Pcap, S, batch_index_offset, rois, valid);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// a10  SupervisedDetectorOutputsBase.forward score assembly  modeling/roi_heads/fast_rcnn.py:425-428
//   scores = delta_scores + mean_k(oicr_k(x_sup_weak)) ; training: scores[:, novel] = -inf   (novel_mask[c] != 0)
// ---------------------------------------------------------------------------------------------------
__global__ void sup_scores_kernel(const float* __restrict__ delta, int ldd, int dcol0, const float* __restrict__ weak, int ldw,
                                  int wcol0, int n_oicr, int ncls, const unsigned char* __restrict__ novel_mask,
                                  const float* __restrict__ extra, int lde, int ecol0, float* __restrict__ out, int ldo, int R) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * ncls) return;
  int r = idx / ncls, c = idx - r * ncls;
  float v = delta[(size_t)r * ldd + dcol0 + c];
  if (weak) {
    // torch.mean(torch.stack(x_weak,0),0): sum in order then / n
    float s = 0.f;
    for (int k = 0; k < n_oicr; ++k) s += weak[(size_t)r * ldw + wcol0 + k * ncls + c];
    v = v + s / (float)n_oicr;
  }
  if (extra) v = v + extra[(size_t)r * lde + ecol0 + c];
  if (novel_mask && c < ncls - 1 && novel_mask[c]) v = -INFINITY;
  out[(size_t)r * ldo + c] = v;
}
extern "C" int unit_sup_scores(const float* delta, int ldd, int dcol0, const float* weak, int ldw, int wcol0, int n_oicr,
                               int ncls, const unsigned char* novel_mask_dev, const float* extra, int lde, int ecol0,
                               float* out, int ldo, int R, void* stream) {
  if (R == 0) return UNIT_OK;
  sup_scores_kernel<<<cdiv(R * ncls, 256), 256, 0, (hipStream_t)stream>>>(delta, ldd, dcol0, weak, ldw, wcol0, n_oicr, ncls, novel_mask_dev,
                                                                       extra, lde, ecol0, out, ldo, R);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// a11/a12  softmax cross-entropy (mean over valid rows, optional per-row weights, -inf-safe)
//   d2 FastRCNNOutputs.softmax_cross_entropy_loss (fast_rcnn.py:438-445) and
//   WeakDetectorOutputsBase.weighted_softmax_with_loss (weak_detector_fast_rcnn.py:262-268).
//   rows with label < 0 are not part of the batch (empty RoI slots).  Single workgroup (R is ~1-2 k rows).
// ---------------------------------------------------------------------------------------------------
template <typename TD>
__global__ void softmax_ce_kernel(const float* __restrict__ logits, int ld, int col0, int ncls, const int* __restrict__ labels,
                                  const float* __restrict__ weights, int R, float gscale, float* __restrict__ loss,
                                  TD* __restrict__ dy, int ldd, int dcol0, unsigned long long* __restrict__ pacc) {
  __shared__ float lds[17];
  float cnt = 0.f;
  for (int r = threadIdx.x; r < R; r += blockDim.x) cnt += labels[r] >= 0 ? 1.f : 0.f;          // every workgroup counts all rows: same total
  float total = block_sum(cnt, lds);
  float inv = total > 0.f ? 1.f / total : 0.f;
  float acc = 0.f;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
    int lab = labels[r];
    const float* x = logits + (size_t)r * ld + col0;
    if (lab < 0) {
      if (dy) for (int c = 0; c < ncls; ++c) st(dy + (size_t)r * ldd + dcol0 + c, 0.f);
      continue;
    }
    if (ncls <= 32) {
      // the row in registers: one read of the logits instead of four dependent passes over L2 (same arithmetic, same order)
      float xv[32];
#pragma unroll
      for (int c = 0; c < 32; ++c) xv[c] = c < ncls ? x[c] : -INFINITY;
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < 32; ++c) if (c < ncls) mx = fmaxf(mx, xv[c]);
      float se = 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c) if (c < ncls) se += expf(xv[c] - mx);
      float lse = logf(se) + mx;
      float w = weights ? weights[r] : 1.f;
      float xl = 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c) if (c == lab) xl = xv[c];
      acc += (lse - xl) * w;
      if (dy) {
        float g = w * inv * gscale;
#pragma unroll
        for (int c = 0; c < 32; ++c)
          if (c < ncls) st(dy + (size_t)r * ldd + dcol0 + c, (expf(xv[c] - lse) - (c == lab ? 1.f : 0.f)) * g);
      }
      continue;
    }
    float mx = -INFINITY;
    for (int c = 0; c < ncls; ++c) mx = fmaxf(mx, x[c]);
    float se = 0.f;
    for (int c = 0; c < ncls; ++c) se += expf(x[c] - mx);
    float lse = logf(se) + mx;
    float w = weights ? weights[r] : 1.f;
    acc += (lse - x[lab]) * w;
    if (dy) {
      float g = w * inv * gscale;
      for (int c = 0; c < ncls; ++c) {
        float pr = expf(x[c] - lse);   // exp(-inf) = 0 for the novel columns
        st(dy + (size_t)r * ldd + dcol0 + c, (pr - (c == lab ? 1.f : 0.f)) * g);
      }
    }
  }
  float s = block_sum(acc, lds);
  if (threadIdx.x == 0) {
    if (gridDim.x == 1) *loss = s * inv;
    else { float t; if (packed_sum_finish(pacc, s, gridDim.x, &t)) *loss = t * inv; }
  }
}
// rows spread over one-wave workgroups when the caller supplies an accumulator (8 zero bytes, handed back zero): the single 1024-thread
// workgroup spent 25-50 us of VALU time on ONE CU (1-4 k rows x 21 classes, two expf per element) on the step's critical path
static int loss_grid(int R, const void* pacc, int* threads) {
  if (pacc == nullptr || R <= 256) { *threads = 1024; return 1; }
  *threads = 64;
  int g = (R + 63) / 64;
  return g > 240 ? 240 : g;
}
extern "C" int unit_softmax_ce(const float* logits, int ld, int col0, int ncls, const int* labels, const float* weights, int R,
                               float gscale, float* loss, void* dy, int dy_dtype, int ldd, int dcol0, unsigned long long* acc, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  int th, g = loss_grid(R, acc, &th);
  if (dy_dtype == UNIT_BF16)
    softmax_ce_kernel<bf16_t><<<g, th, 0, s>>>(logits, ld, col0, ncls, labels, weights, R, gscale, loss, (bf16_t*)dy, ldd, dcol0, acc);
  else
    softmax_ce_kernel<float><<<g, th, 0, s>>>(logits, ld, col0, ncls, labels, weights, R, gscale, loss, (float*)dy, ldd, dcol0, acc);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// a11  box regression loss (d2 FastRCNNOutputs.box_reg_loss; arithmetic documented in-tree fast_rcnn.py:37-101):
//   sum_{fg rows} | bbox[r, 4c:4c+4] - get_deltas(prop, gt; w) |  /  (#rows)       smooth-L1 beta=0 == L1
// ---------------------------------------------------------------------------------------------------
template <typename TD>
__global__ void box_reg_loss_kernel(const float* __restrict__ bbox, int ld, int col0, int K, const int* __restrict__ labels,
                                    const float* __restrict__ rois5, const float* __restrict__ gtb, f32x4 w, int R, float gscale,
                                    float* __restrict__ loss, TD* __restrict__ dy, int ldd, int dcol0, unsigned long long* __restrict__ pacc) {
  __shared__ float lds[17];
  float cnt = 0.f;
  for (int r = threadIdx.x; r < R; r += blockDim.x) cnt += labels[r] >= 0 ? 1.f : 0.f;
  float total = block_sum(cnt, lds);
  float inv = total > 0.f ? 1.f / total : 0.f;
  float acc = 0.f;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
    int lab = labels[r];
    if (dy) for (int c = 0; c < 4 * K; ++c) st(dy + (size_t)r * ldd + dcol0 + c, 0.f);
    if (lab < 0 || lab >= K) continue;
    f32x4 pb = {rois5[(size_t)r * 5 + 1], rois5[(size_t)r * 5 + 2], rois5[(size_t)r * 5 + 3], rois5[(size_t)r * 5 + 4]};
    f32x4 g = *reinterpret_cast<const f32x4*>(gtb + (size_t)r * 4);
    f32x4 t = encode1(pb, g, w);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float df = bbox[(size_t)r * ld + col0 + 4 * lab + j] - t[j];
      acc += fabsf(df);
      if (dy) st(dy + (size_t)r * ldd + dcol0 + 4 * lab + j, (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * inv * gscale);
    }
  }
  float s = block_sum(acc, lds);
  if (threadIdx.x == 0) {
    if (gridDim.x == 1) *loss = s * inv;
    else { float t; if (packed_sum_finish(pacc, s, gridDim.x, &t)) *loss = t * inv; }
  }
}
extern "C" int unit_box_reg_loss(const float* bbox, int ld, int col0, int K, const int* labels, const float* rois5, const float* gt_boxes,
                                 const float* weights4, int R, float gscale, float* loss, void* dy, int dy_dtype, int ldd, int dcol0,
                                 unsigned long long* acc, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  f32x4 w = {weights4[0], weights4[1], weights4[2], weights4[3]};
  int th, g = loss_grid(R, acc, &th);
  if (dy_dtype == UNIT_BF16)
    box_reg_loss_kernel<bf16_t><<<g, th, 0, s>>>(bbox, ld, col0, K, labels, rois5, gt_boxes, w, R, gscale, loss, (bf16_t*)dy, ldd, dcol0, acc);
  else
    box_reg_loss_kernel<float><<<g, th, 0, s>>>(bbox, ld, col0, K, labels, rois5, gt_boxes, w, R, gscale, loss, (float*)dy, ldd, dcol0, acc);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// a12  WSDDN / MIL  modeling/roi_heads/weak_detector_fast_rcnn.py:202-214,257-260
//   per image: x_r = softmax(cls/Tc, dim=-1) * softmax(det/Td, dim=0) ; p_c = clamp(sum_r x_r, 1e-6, 1-1e-6)
//   loss_im_cls = mil_multiplier * mean_{img,c} BCE(p_c, y_c).   One workgroup per image, one thread per RoI row.
// streams [Rtot][ld] fp32: cls cols [ccol0, +K), det cols [dcol0, +K) (raw Linear outputs, temperatures applied here).
// ---------------------------------------------------------------------------------------------------
#define MIL_MAXK 96
template <typename TD>
__global__ void wsddn_mil_kernel(const float* __restrict__ streams, int ld, int ccol0, int dcol0, int K, const int* __restrict__ valid,
                                 int S, const unsigned char* __restrict__ multihot, float tc, float td, float mult, float gscale,
                                 int B, float* __restrict__ loss, float* __restrict__ xr_out, TD* __restrict__ dy, int ldd,
                                 int dyc0, int dyd0) {
  __shared__ float colmax[MIL_MAXK], colsum[MIL_MAXK], pcls[MIL_MAXK], dp[MIL_MAXK], coldot[MIL_MAXK];
  __shared__ float lds[17];
  int b = blockIdx.x;
  int tid = threadIdx.x;
  // All K column reductions of a pass are batched: each wave reduces its rows for every column, lane 0 parks the K
  // partials in LDS, ONE barrier, threads c < K combine the waves in wave order (3 barriers per pass instead of 3 per column).
  __shared__ float part[16][MIL_MAXK];
  const int lane = tid & 63, wid = tid >> 6, nwv = (blockDim.x + 63) >> 6;
  // pass 1: column max of det logits over valid rows
  for (int c = 0; c < K; ++c) {
    float m = -INFINITY;
    for (int r = tid; r < S; r += blockDim.x) {
      size_t row = (size_t)b * S + r;
      if (valid[row] >= 0) m = fmaxf(m, streams[row * ld + dcol0 + c] / td);
    }
    m = wave_reduce_max(m);
    if (lane == 0) part[wid][c] = m;
  }
  __syncthreads();
  if (tid < K) { float mm = -INFINITY; for (int w = 0; w < nwv; ++w) mm = fmaxf(mm, part[w][tid]); colmax[tid] = mm; }
  __syncthreads();
  // pass 2: column sum of exp
  for (int c = 0; c < K; ++c) {
    float sacc = 0.f;
    for (int r = tid; r < S; r += blockDim.x) {
      size_t row = (size_t)b * S + r;
      if (valid[row] >= 0) sacc += expf(streams[row * ld + dcol0 + c] / td - colmax[c]);
    }
    sacc = wave_reduce_sum(sacc);
    if (lane == 0) part[wid][c] = sacc;
  }
  __syncthreads();
  if (tid < K) { float t = 0.f; for (int w = 0; w < nwv; ++w) t += part[w][tid]; colsum[tid] = t; }
  __syncthreads();
  // pass 3: x_r (row softmax evaluated once per row) ...
  for (int r = tid; r < S; r += blockDim.x) {
    size_t row = (size_t)b * S + r;
    if (valid[row] >= 0) {
      const float* cs = streams + row * ld + ccol0;
      float mx = -INFINITY;
      for (int k = 0; k < K; ++k) mx = fmaxf(mx, cs[k] / tc);
      float se = 0.f;
      for (int k = 0; k < K; ++k) se += expf(cs[k] / tc - mx);
      for (int c = 0; c < K; ++c) {
        float s1 = expf(cs[c] / tc - mx) / se;
        float s2 = expf(streams[row * ld + dcol0 + c] / td - colmax[c]) / colsum[c];
        xr_out[row * K + c] = s1 * s2;
      }
    } else {
      for (int c = 0; c < K; ++c) xr_out[row * K + c] = 0.f;
    }
  }
  // ... and the class vector p_c = sum_r x_rc (every thread re-reads only the rows it wrote itself)
  for (int c = 0; c < K; ++c) {
    float sacc = 0.f;
    for (int r = tid; r < S; r += blockDim.x) sacc += xr_out[((size_t)b * S + r) * K + c];
    sacc = wave_reduce_sum(sacc);
    if (lane == 0) part[wid][c] = sacc;
  }
  __syncthreads();
  if (tid < K) { float t = 0.f; for (int w = 0; w < nwv; ++w) t += part[w][tid]; pcls[tid] = t; }
  __syncthreads();
  // BCE on clamped class vector ; d loss / d p
  float l = 0.f;
  for (int c = tid; c < K; c += blockDim.x) {
    float p = pcls[c];
    float pc = fminf(fmaxf(p, 1e-6f), 1.f - 1e-6f);
    float y = multihot[(size_t)b * K + c] ? 1.f : 0.f;
    // F.binary_cross_entropy clamps log terms at -100
    float lp = fmaxf(logf(pc), -100.f), l1p = fmaxf(logf(1.f - pc), -100.f);
    l += -(y * lp + (1.f - y) * l1p);
    float g = (p > 1e-6f && p < 1.f - 1e-6f) ? (-(y / pc) + (1.f - y) / (1.f - pc)) : 0.f;
    dp[c] = g * mult * gscale / (float)(B * K);
  }
  float lt = block_sum(l, lds);
  if (tid == 0) atomicAdd(loss, lt * mult / (float)(B * K));
  if (!dy) return;
  __syncthreads();
  // backward: x = s1*s2 ; dcs = s1 * (dx*s2 - sum_c dx*s2*s1) / tc ; dds = s2 * (dx*s1 - coldot_c) / td , coldot_c = sum_r dx*s1*s2
  if (tid < K) coldot[tid] = dp[tid] * pcls[tid];      // sum_r dp_c * x_rc  (x_rc = 0 on invalid rows)
  __syncthreads();
  for (int r = tid; r < S; r += blockDim.x) {
    size_t row = (size_t)b * S + r;
    if (valid[row] < 0) {
      for (int c = 0; c < K; ++c) { st(dy + row * ldd + dyc0 + c, 0.f); st(dy + row * ldd + dyd0 + c, 0.f); }
      continue;
    }
    const float* cs = streams + row * ld + ccol0;
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, cs[k] / tc);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(cs[k] / tc - mx);
    float rowdot = 0.f;
    for (int c = 0; c < K; ++c) rowdot += dp[c] * xr_out[row * K + c];   // sum_c dx * s2 * s1
    for (int c = 0; c < K; ++c) {
      float s1 = expf(cs[c] / tc - mx) / se;
      float s2 = expf(streams[row * ld + dcol0 + c] / td - colmax[c]) / colsum[c];
      float x = xr_out[row * K + c];
      float dcs = (dp[c] * x - s1 * rowdot) / tc;
      float dds = (dp[c] * x - s2 * coldot[c]) / td;
      st(dy + row * ldd + dyc0 + c, dcs);
      st(dy + row * ldd + dyd0 + c, dds);
    }
  }
}
// K <= KR (VOC: 20): the same arithmetic in the same order with the row's 2K logits and its K products held in REGISTERS -- one
// global read of the row instead of a dozen dependent ones (each pass of the kernel above re-reads its inputs from L2 and waits
// for them: 105-112 us on the critical path of the step for 2 x 512 x 20 values).
template <typename TD, int KR>
__global__ void __launch_bounds__(512) wsddn_mil_reg_kernel(const float* __restrict__ streams, int ld, int ccol0, int dcol0, int K,
                                                            const int* __restrict__ valid, int S, const unsigned char* __restrict__ multihot,
                                                            float tc, float td, float mult, float gscale, int B, float* __restrict__ loss,
                                                            float* __restrict__ xr_out, TD* __restrict__ dy, int ldd, int dyc0, int dyd0) {
  __shared__ float colmax[KR], colsum[KR], pcls[KR], dp[KR], coldot[KR];
  __shared__ float lds[17];
  __shared__ float part[8][KR];
  const int b = blockIdx.x, tid = threadIdx.x;          // one thread per RoI row (S <= 512)
  const int lane = tid & 63, wid = tid >> 6, nwv = (blockDim.x + 63) >> 6;
  const size_t row = (size_t)b * S + tid;
  const bool have = tid < S && valid[row] >= 0;
  float cs[KR], ds[KR], xr[KR];
#pragma unroll
  for (int c = 0; c < KR; ++c) {
    cs[c] = (have && c < K) ? streams[row * ld + ccol0 + c] : 0.f;
    ds[c] = (have && c < K) ? streams[row * ld + dcol0 + c] : 0.f;
  }
  // pass 1: column max of det logits over valid rows
#pragma unroll
  for (int c = 0; c < KR; ++c) {
    float m = (have && c < K) ? ds[c] / td : -INFINITY;
    m = wave_reduce_max(m);
    if (lane == 0) part[wid][c] = m;
  }
  __syncthreads();
  if (tid < K) { float mm = -INFINITY; for (int w = 0; w < nwv; ++w) mm = fmaxf(mm, part[w][tid]); colmax[tid] = mm; }
  __syncthreads();
  // pass 2: column sum of exp
#pragma unroll
  for (int c = 0; c < KR; ++c) {
    float sacc = (have && c < K) ? expf(ds[c] / td - colmax[c]) : 0.f;
    sacc = wave_reduce_sum(sacc);
    if (lane == 0) part[wid][c] = sacc;
  }
  __syncthreads();
  if (tid < K) { float t = 0.f; for (int w = 0; w < nwv; ++w) t += part[w][tid]; colsum[tid] = t; }
  __syncthreads();
  // pass 3: x_r
  float mx = -INFINITY, se = 0.f;
  if (have) {
#pragma unroll
    for (int k = 0; k < KR; ++k) if (k < K) mx = fmaxf(mx, cs[k] / tc);
#pragma unroll
    for (int k = 0; k < KR; ++k) if (k < K) se += expf(cs[k] / tc - mx);
  }
#pragma unroll
  for (int c = 0; c < KR; ++c) {
    float x = 0.f;
    if (have && c < K) {
      float s1 = expf(cs[c] / tc - mx) / se;
      float s2 = expf(ds[c] / td - colmax[c]) / colsum[c];
      x = s1 * s2;
    }
    xr[c] = x;
    if (tid < S && c < K) xr_out[row * K + c] = x;
  }
  // class vector p_c = sum_r x_rc
#pragma unroll
  for (int c = 0; c < KR; ++c) {
    float sacc = wave_reduce_sum(xr[c]);
    if (lane == 0) part[wid][c] = sacc;
  }
  __syncthreads();
  if (tid < K) { float t = 0.f; for (int w = 0; w < nwv; ++w) t += part[w][tid]; pcls[tid] = t; }
  __syncthreads();
  float l = 0.f;
  for (int c = tid; c < K; c += blockDim.x) {
    float p = pcls[c];
    float pc = fminf(fmaxf(p, 1e-6f), 1.f - 1e-6f);
    float y = multihot[(size_t)b * K + c] ? 1.f : 0.f;
    float lp = fmaxf(logf(pc), -100.f), l1p = fmaxf(logf(1.f - pc), -100.f);
    l += -(y * lp + (1.f - y) * l1p);
    float g = (p > 1e-6f && p < 1.f - 1e-6f) ? (-(y / pc) + (1.f - y) / (1.f - pc)) : 0.f;
    dp[c] = g * mult * gscale / (float)(B * K);
  }
  float lt = block_sum(l, lds);
  if (tid == 0) atomicAdd(loss, lt * mult / (float)(B * K));
  if (!dy) return;
  __syncthreads();
  if (tid < K) coldot[tid] = dp[tid] * pcls[tid];
  __syncthreads();
  if (tid >= S) return;
  if (!have) {
    for (int c = 0; c < K; ++c) { st(dy + row * ldd + dyc0 + c, 0.f); st(dy + row * ldd + dyd0 + c, 0.f); }
    return;
  }
  float rowdot = 0.f;
#pragma unroll
  for (int c = 0; c < KR; ++c) if (c < K) rowdot += dp[c] * xr[c];
#pragma unroll
  for (int c = 0; c < KR; ++c) {
    if (c < K) {
      float s1 = expf(cs[c] / tc - mx) / se;
      float s2 = expf(ds[c] / td - colmax[c]) / colsum[c];
      float x = xr[c];
      st(dy + row * ldd + dyc0 + c, (dp[c] * x - s1 * rowdot) / tc);
      st(dy + row * ldd + dyd0 + c, (dp[c] * x - s2 * coldot[c]) / td);
    }
  }
}

extern "C" int unit_wsddn_mil(const float* streams, int ld, int ccol0, int dcol0, int K, const int* valid, int S, int B,
                              const unsigned char* multihot, float cls_temp, float det_temp, float mil_multiplier, float gscale,
                              float* loss, float* xr_out, void* dy, int dy_dtype, int ldd, int dyc0, int dyd0, void* stream) {
  UNIT_CHECK_ARG(K <= MIL_MAXK, "wsddn: K > 96");
  UNIT_CHECK_ARG(xr_out != nullptr, "wsddn: xr_out required");
  hipStream_t s = (hipStream_t)stream;
  (void)hipMemsetAsync(loss, 0, sizeof(float), s);
  if (B == 0) return UNIT_OK;
  if (K <= 32 && S <= 512) {     // register-resident rows (VOC)
    if (dy_dtype == UNIT_BF16)
      wsddn_mil_reg_kernel<bf16_t, 32><<<B, 512, 0, s>>>(streams, ld, ccol0, dcol0, K, valid, S, multihot, cls_temp, det_temp, mil_multiplier, gscale, B, loss, xr_out, (bf16_t*)dy, ldd, dyc0, dyd0);
    else
      wsddn_mil_reg_kernel<float, 32><<<B, 512, 0, s>>>(streams, ld, ccol0, dcol0, K, valid, S, multihot, cls_temp, det_temp, mil_multiplier, gscale, B, loss, xr_out, (float*)dy, ldd, dyc0, dyd0);
    UNIT_LAUNCH_CHECK();
    return UNIT_OK;
  }
  if (dy_dtype == UNIT_BF16)
    wsddn_mil_kernel<bf16_t><<<B, 512, 0, s>>>(streams, ld, ccol0, dcol0, K, valid, S, multihot, cls_temp, det_temp, mil_multiplier, gscale, B, loss, xr_out, (bf16_t*)dy, ldd, dyc0, dyd0);
  else
    wsddn_mil_kernel<float><<<B, 512, 0, s>>>(streams, ld, ccol0, dcol0, K, valid, S, multihot, cls_temp, det_temp, mil_multiplier, gscale, B, loss, xr_out, (float*)dy, ldd, dyc0, dyd0);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// a12  OICR pseudo-GT targets  weak_detector_fast_rcnn.py:353-408 (get_proposal_clusters + label_and_sample_proposals
//   with the UniT Matcher([0.5],[0,1]) + compute_loss_inputs): sync-free, one workgroup per image.
//   probs: mode 0 -> given [Rtot][K] (detached MIL x_r) ; mode 1 -> softmax of logits cols [col0, col0+K+1).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float iou_box(const f32x4 a, const f32x4 b) {
  float area1 = (a[2] - a[0]) * (a[3] - a[1]);
  float area2 = (b[2] - b[0]) * (b[3] - b[1]);
  float w = fminf(a[2], b[2]) - fmaxf(a[0], b[0]);
  float h = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
  w = w < 0.f ? 0.f : w; h = h < 0.f ? 0.f : h;
  float inter = w * h;
  return inter > 0.f ? inter / (area1 + area2 - inter) : 0.0f;
}

__global__ void oicr_targets_kernel(const float* __restrict__ src, int ld, int col0, int mode, int K, const float* __restrict__ rois5,
                                    const int* __restrict__ valid, int S, const unsigned char* __restrict__ multihot,
                                    float fg_thresh, float bg_thresh, int* __restrict__ labels, float* __restrict__ weights) {
  __shared__ float s_val[16]; __shared__ int s_idx[16];
  __shared__ f32x4 gbox[MIL_MAXK]; __shared__ float gscore[MIL_MAXK]; __shared__ int gcls[MIL_MAXK];
  __shared__ int s_zero_row, s_ngt;
  int b = blockIdx.x, tid = threadIdx.x;
  int ncol = mode == 0 ? K : K + 1;
  // each thread owns row r = tid (S <= blockDim.x)
  size_t row = (size_t)b * S + tid;
  bool have = tid < S && valid[row] >= 0;
  // probabilities are evaluated on demand (no per-thread array -> no scratch)
  float mx = 0.f, se = 1.f;
  const float* xrow = src + row * ld + col0;
  if (have && mode == 1) {
    mx = -INFINITY;
    for (int c = 0; c < ncol; ++c) mx = fmaxf(mx, xrow[c]);
    se = 0.f;
    for (int c = 0; c < ncol; ++c) se += expf(xrow[c] - mx);
  }
  if (tid == 0) s_ngt = 0;
  __syncthreads();
  bool zeroed = false;
  for (int c = 0; c < K; ++c) {
    if (!multihot[(size_t)b * K + c]) continue;     // wave-uniform (same for the whole block)
    float v = have ? (zeroed ? 0.f : (mode == 0 ? xrow[c] : expf(xrow[c] - mx) / se)) : -INFINITY;
    int idx = tid;
    // block argmax, first (lowest) index on ties -- torch.max(dim=0)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ov = __shfl_xor(v, o, 64); int oi = __shfl_xor(idx, o, 64);
      if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    if ((tid & 63) == 0) { s_val[tid >> 6] = v; s_idx[tid >> 6] = idx; }
    __syncthreads();
    if (tid == 0) {
      float bv = s_val[0]; int bi = s_idx[0];
      for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
        if (s_val[w] > bv || (s_val[w] == bv && s_idx[w] < bi)) { bv = s_val[w]; bi = s_idx[w]; }
      int g = s_ngt;
      gscore[g] = bv; gcls[g] = c; s_zero_row = bi;
      size_t rr = (size_t)b * S + bi;
      gbox[g] = f32x4{rois5[rr * 5 + 1], rois5[rr * 5 + 2], rois5[rr * 5 + 3], rois5[rr * 5 + 4]};
      s_ngt = g + 1;
    }
    __syncthreads();
    if (tid == s_zero_row) zeroed = true;            // cls_prob[max_index, :] = 0 (:364)
    __syncthreads();
  }
  if (tid >= S) return;
  if (!have) { labels[row] = -1; weights[row] = 0.f; return; }
  int ng = s_ngt;
  if (ng == 0) { labels[row] = K; weights[row] = 0.f; return; }
  f32x4 me = {rois5[row * 5 + 1], rois5[row * 5 + 2], rois5[row * 5 + 3], rois5[row * 5 + 4]};
  float best = -1.f; int bi = 0;
  for (int g = 0; g < ng; ++g) { float v = iou_box(gbox[g], me); if (v > best) { best = v; bi = g; } }
  int lab = (best >= fg_thresh) ? gcls[bi] : K;     // Matcher([fg],[0,1]): < thr -> 0 -> bg ; >= thr -> 1 -> class
  float w = gscore[bi];
  if (best < bg_thresh) w = 0.f;                      // :392-396
  labels[row] = lab; weights[row] = w;
}
extern "C" int unit_oicr_targets(const float* src, int ld, int col0, int mode, int K, const float* rois5, const int* valid, int S,
                                 int B, const unsigned char* multihot, float fg_thresh, float bg_thresh, int* labels, float* weights,
                                 void* stream) {
  UNIT_CHECK_ARG(K < MIL_MAXK && S <= 1024, "oicr_targets: K >= 96 or S > 1024");
  if (B == 0) return UNIT_OK;
  int threads = ((S + 63) / 64) * 64;
  oicr_targets_kernel<<<B, threads, 0, (hipStream_t)stream>>>(src, ld, col0, mode, K, rois5, valid, S, multihot, fg_thresh, bg_thresh, labels, weights);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// total = sum_i losses[i]  (one launch, keeps the step free of framework arithmetic)
__global__ void sum_losses_kernel(const float* __restrict__ losses, int n, float* __restrict__ out) {
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += losses[i];
  *out = s;
}
extern "C" int unit_sum_losses(const float* losses, int n, float* out, void* stream) {
  sum_losses_kernel<<<1, 1, 0, (hipStream_t)stream>>>(losses, n, out);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
