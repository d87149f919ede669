// mask.hip -- Mask-RCNN head pieces for the C4-segm configuration:
//   MaskRCNNConvUpsampleHeadWithSimilarity.forward  /root/reference/modeling/roi_heads/mask_head.py:16-37
//   (detectron2 MaskRCNNConvUpsampleHead: ConvTranspose2d(2048,256,k2,s2) -> ReLU -> Conv2d(256,K,1); mask_rcnn_loss /
//    mask_rcnn_inference; BitMasks.crop_and_resize).
// The 2x2 stride-2 transposed conv has no overlapping taps: it is ONE 1x1 GEMM with 4*Cout output columns
// ([q = dy*2+dx][oc]) evaluated by the conv kernel; the output pixel (2y+dy, 2x+dx) lives at [roi, y, x, q, oc].  The 1x1
// predictor and the loss work directly on that layout, so no pixel shuffle is ever materialised.
#include "common.h"

// ConvTranspose2d weight fp32 [Cin][Cout][2][2] -> forward GEMM weight [4*Cout][Cin] and dgrad weight [Cin][4*Cout]
template <typename T>
__global__ void deconv_prep_kernel(const float* __restrict__ w, int Cin, int Cout, T* __restrict__ wf, T* __restrict__ wd) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)Cin * Cout * 4;
  if (idx >= total) return;
  int q = idx % 4; long t = idx / 4; int oc = t % Cout; int ic = t / Cout;
  float v = w[idx];                                   // w[ic][oc][dy][dx], q = dy*2+dx
  wf[((size_t)q * Cout + oc) * Cin + ic] = (T)v;
  wd[(size_t)ic * (4 * Cout) + q * Cout + oc] = (T)v;
}
extern "C" int unit_deconv2x2_weight_prep(const float* w, int Cin, int Cout, void* w_fwd, void* w_dgrad, int dtype, void* stream) {
  long total = (long)Cin * Cout * 4;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UNIT_BF16) deconv_prep_kernel<bf16_t><<<cdiv(total, 256), 256, 0, st>>>(w, Cin, Cout, (bf16_t*)w_fwd, (bf16_t*)w_dgrad);
  else deconv_prep_kernel<float><<<cdiv(total, 256), 256, 0, st>>>(w, Cin, Cout, (float*)w_fwd, (float*)w_dgrad);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
// gradient of the GEMM weight [4*Cout][Cin] -> ConvTranspose2d layout [Cin][Cout][2][2]; bias grad: sum over the 4 taps
__global__ void deconv_unpack_kernel(const float* __restrict__ dwp, const float* __restrict__ dbp, int Cin, int Cout, float* __restrict__ dw,
                                     float* __restrict__ db) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)Cin * Cout * 4;
  if (idx < total) {
    int q = idx % 4; long t = idx / 4; int oc = t % Cout; int ic = t / Cout;
    dw[idx] = dwp[((size_t)q * Cout + oc) * Cin + ic];
  }
  if (db && idx < Cout) db[idx] = dbp[idx] + dbp[Cout + idx] + dbp[2 * Cout + idx] + dbp[3 * Cout + idx];
}
extern "C" int unit_deconv2x2_grad_unpack(const float* dw_gemm, const float* db_gemm, int Cin, int Cout, float* dw, float* db, void* stream) {
  long total = (long)Cin * Cout * 4;
  deconv_unpack_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(dw_gemm, db_gemm, Cin, Cout, dw, db);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// BitMasks.crop_and_resize: ROIAlign((M,M), 1.0, 0, aligned=True) on the float bitmask, then >= 0.5.
// gt_masks u8 [B][Mcap][Hm][Wm]; slot s uses mask (image = rois5[s][0], instance = gt_index[s]); cls < 0 slots -> 0.
__global__ void mask_targets_kernel(const unsigned char* __restrict__ masks, int Mcap, int Hm, int Wm, const float* __restrict__ rois5,
                                    const int* __restrict__ gt_index, const int* __restrict__ cls, int K, int M, unsigned char* __restrict__ out) {
  int s = blockIdx.x;
  int c = cls[s];
  for (int bin = threadIdx.x; bin < M * M; bin += blockDim.x) {
    unsigned char v = 0;
    if (c >= 0 && c < K) {
      int ph = bin / M, pw = bin - ph * M;
      const float* roi = rois5 + 5 * (size_t)s;
      int b = (int)roi[0];
      const unsigned char* mk = masks + ((size_t)b * Mcap + gt_index[s]) * Hm * Wm;
      float sw = roi[1] - 0.5f, sh = roi[2] - 0.5f, ew = roi[3] - 0.5f, eh = roi[4] - 0.5f;
      float rw = ew - sw, rh = eh - sh;
      float bh = rh / (float)M, bw = rw / (float)M;
      int gh = (int)ceilf(rh / (float)M), gw = (int)ceilf(rw / (float)M);
      float count = (float)(gh * gw > 1 ? gh * gw : 1);
      float acc = 0.f;
      for (int iy = 0; iy < gh; ++iy) {
        float y = sh + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
          float x = sw + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
          if (y < -1.0f || y > (float)Hm || x < -1.0f || x > (float)Wm) continue;
          float yy = y <= 0.f ? 0.f : y, xx = x <= 0.f ? 0.f : x;
          int yl = (int)yy, xl = (int)xx, yh, xh;
          if (yl >= Hm - 1) { yh = yl = Hm - 1; yy = (float)yl; } else yh = yl + 1;
          if (xl >= Wm - 1) { xh = xl = Wm - 1; xx = (float)xl; } else xh = xl + 1;
          float ly = yy - (float)yl, lx = xx - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
          float val = hy * hx * (float)mk[yl * Wm + xl] + hy * lx * (float)mk[yl * Wm + xh] + ly * hx * (float)mk[yh * Wm + xl] +
                      ly * lx * (float)mk[yh * Wm + xh];
          acc += val;
        }
      }
      v = (acc / count) >= 0.5f ? 1 : 0;
    }
    out[(size_t)s * M * M + bin] = v;
  }
}
extern "C" int unit_mask_targets(const unsigned char* gt_masks, int Mcap, int Hm, int Wm, const float* rois5, const int* gt_index,
                                 const int* cls, int K, int S, int M, unsigned char* out, void* stream) {
  if (S == 0) return UNIT_OK;
  mask_targets_kernel<<<S, 256, 0, (hipStream_t)stream>>>(gt_masks, Mcap, Hm, Wm, rois5, gt_index, cls, K, M, out);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// mask_rcnn_loss: mean over (#fg slots x M x M) of BCE-with-logits on the gt-class channel. logits [S][P][P][4][K] fp32
// (P = M/2); pixel (Y,X) -> [Y/2][X/2][(Y&1)*2 + (X&1)]. Emits d(loss)/d(logits) in the same layout (dtype TD).
template <typename TD>
__global__ void mask_loss_kernel(const float* __restrict__ logits, int K, int ldk, const int* __restrict__ cls, const unsigned char* __restrict__ tgt,
                                 int S, int M, float gscale, float* __restrict__ loss, TD* __restrict__ dlogits) {
  __shared__ float lds[18];
  int P = M / 2;
  float cnt = 0.f;
  for (int s = threadIdx.x; s < S; s += blockDim.x) cnt += (cls[s] >= 0 && cls[s] < K) ? 1.f : 0.f;
  cnt = wave_reduce_sum(cnt);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) { float t = 0.f; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += lds[w]; lds[16] = t; }
  __syncthreads();
  float nfg = lds[16];
  float inv = nfg > 0.f ? 1.f / (nfg * (float)(M * M)) : 0.f;
  float acc = 0.f;
  long total = (long)S * M * M;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int s = i / (M * M); int r = i - (long)s * M * M; int Y = r / M, X = r - Y * M;
    int c = cls[s];
    if (c < 0 || c >= K) continue;
    size_t o = ((((size_t)s * P + (Y >> 1)) * P + (X >> 1)) * 4 + ((Y & 1) * 2 + (X & 1))) * ldk + c;
    float x = logits[o], y = (float)tgt[i];
    acc += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
    if (dlogits) dlogits[o] = (TD)((1.f / (1.f + expf(-x)) - y) * inv * gscale);
  }
  acc = wave_reduce_sum(acc);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { float t = 0.f; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += lds[w]; atomicAdd(loss, t * inv); }
}
extern "C" int unit_mask_bce_loss(const float* logits, int K, int ldk, const int* cls, const unsigned char* targets, int S, int M,
                                  float gscale, float* loss, void* dlogits, int d_dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(loss, 0, sizeof(float), st);
  if (S == 0) return UNIT_OK;
  size_t esz = d_dtype == UNIT_BF16 ? 2 : 4;
  if (dlogits) (void)hipMemsetAsync(dlogits, 0, (size_t)S * M * M * ldk * esz, st);
  int blocks = min(1024, cdiv((long)S * M * M, 256));
  if (d_dtype == UNIT_BF16) mask_loss_kernel<bf16_t><<<blocks, 256, 0, st>>>(logits, K, ldk, cls, targets, S, M, gscale, loss, (bf16_t*)dlogits);
  else mask_loss_kernel<float><<<blocks, 256, 0, st>>>(logits, K, ldk, cls, targets, S, M, gscale, loss, (float*)dlogits);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// MaskRCNNConvUpsampleHeadWithFineTune in TRAINING (mask_head.py:74-93 with similarity['seg'][fg], roi_heads.py:888-906):
//   logit[s][c] = transfer(predictor)[c] + predictor_delta[c],  transfer = row[c] (c base) | sum_b sim[row_s][j][b] row[base_b] (c novel)
// loss = mean BCE on the gt-class channel; emits d(loss)/d(logits) for BOTH column groups (the transferred gradient lands on the
// base columns of `predictor`, whose weights are frozen but whose input is trainable) and ADDS d(loss)/d(sim) into
// dsim[row_s][j][:] (each foreground RoI owns its row of dsim: plain adds, fixed order -> reproducible). One workgroup per fg slot.
template <typename TD>
__global__ void __launch_bounds__(256) mask_loss_ft_kernel(const float* __restrict__ logits, int K, int ldk, int delta_col0,
                                                           const int* __restrict__ cls, const unsigned char* __restrict__ tgt,
                                                           const float* __restrict__ sim, const int* __restrict__ rows,
                                                           const int* __restrict__ base, int n_base, int n_novel,
                                                           const int8_t* __restrict__ role, const int* __restrict__ slot, int S, int M,
                                                           float gscale, float* __restrict__ loss, TD* __restrict__ dlogits,
                                                           float* __restrict__ dsim) {
  __shared__ float lds[8];
  __shared__ float s_nfg;
  int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int P = M / 2, MM = M * M;
  float cnt = 0.f;
  for (int q = tid; q < S; q += 256) cnt += (cls[q] >= 0 && cls[q] < K) ? 1.f : 0.f;
  cnt = wave_reduce_sum(cnt);
  if (lane == 0) lds[wid] = cnt;
  __syncthreads();
  if (tid == 0) s_nfg = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  int c = cls[s];
  if (c < 0 || c >= K) return;
  float inv = 1.f / (s_nfg * (float)MM);
  int rl = sim ? role[c] : 1;
  const float* sm = (sim && rl == 2) ? sim + ((size_t)rows[s] * n_novel + slot[c]) * n_base : nullptr;
  int i = tid;
  bool act = i < MM;
  float g = 0.f, acc = 0.f;
  const float* row = nullptr;
  size_t ro = 0;
  if (act) {
    int Y = i / M, X = i - Y * M;
    ro = ((((size_t)s * P + (Y >> 1)) * P + (X >> 1)) * 4 + ((Y & 1) * 2 + (X & 1))) * ldk;
    row = logits + ro;
    float x;
    if (sm) { x = 0.f; for (int b = 0; b < n_base; ++b) x += sm[b] * row[base[b]]; }
    else x = (rl == 0) ? 0.f : row[c];
    if (delta_col0 >= 0) x = x + row[delta_col0 + c];
    float y = (float)tgt[(size_t)s * MM + i];
    acc = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
    g = (1.f / (1.f + expf(-x)) - y) * inv * gscale;
    if (dlogits) {
      TD* d = dlogits + ro;
      if (delta_col0 >= 0) d[delta_col0 + c] = (TD)g;
      if (sm) { for (int b = 0; b < n_base; ++b) d[base[b]] = (TD)(sm[b] * g); }
      else if (rl != 0) d[c] = (TD)g;
    }
  }
  // loss
  acc = wave_reduce_sum(acc);
  __syncthreads();
  if (lane == 0) lds[wid] = acc;
  __syncthreads();
  if (tid == 0) atomicAdd(loss, (lds[0] + lds[1] + lds[2] + lds[3]) * inv);
  // d sim[row_s][j][b] += sum_pixels g * row[base_b]
  if (sm && dsim) {
    float* ds = dsim + ((size_t)rows[s] * n_novel + slot[c]) * n_base;
    for (int b = 0; b < n_base; ++b) {
      float v = act ? g * row[base[b]] : 0.f;
      v = wave_reduce_sum(v);
      __syncthreads();
      if (lane == 0) lds[wid] = v;
      __syncthreads();
      if (tid == 0) ds[b] += lds[0] + lds[1] + lds[2] + lds[3];
    }
  }
}
extern "C" int unit_mask_bce_loss_ft(const float* logits, int K, int ldk, int delta_col0, const int* cls, const unsigned char* targets,
                                     const float* sim, const int* sim_rows, const int* base_dev, int n_base, int n_novel,
                                     const int8_t* role_dev, const int* slot_dev, int S, int M, float gscale, float* loss, void* dlogits,
                                     int d_dtype, float* dsim, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(loss, 0, sizeof(float), st);
  if (S == 0) return UNIT_OK;
  UNIT_CHECK_ARG(M * M <= 256, "mask_bce_loss_ft: mask side > 16");
  size_t esz = d_dtype == UNIT_BF16 ? 2 : 4;
  if (dlogits) (void)hipMemsetAsync(dlogits, 0, (size_t)S * M * M * ldk * esz, st);
  if (d_dtype == UNIT_BF16)
    mask_loss_ft_kernel<bf16_t><<<S, 256, 0, st>>>(logits, K, ldk, delta_col0, cls, targets, sim, sim_rows, base_dev, n_base, n_novel, role_dev,
                                                 slot_dev, S, M, gscale, loss, (bf16_t*)dlogits, dsim);
  else
    mask_loss_ft_kernel<float><<<S, 256, 0, st>>>(logits, K, ldk, delta_col0, cls, targets, sim, sim_rows, base_dev, n_base, n_novel, role_dev,
                                                slot_dev, S, M, gscale, loss, (float*)dlogits, dsim);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// mask_rcnn_inference (+ the base->novel mask transfer of mask_head.py:18-31 for the predicted class):
// prob[s][Y][X] = sigmoid(logit of pred class)  where for a novel class j: logit = sum_b sim[s][j][b] * logit[base_b];
// delta_col0 >= 0: + logits[delta_col0 + c], the `predictor_delta` columns of MaskRCNNConvUpsampleHeadWithFineTune (mask_head.py:91)
__global__ void mask_probs_kernel(const float* __restrict__ logits, int K, int ldk, int delta_col0, const int* __restrict__ cls,
                                  const float* __restrict__ sim, const int* __restrict__ base, int n_base, int n_novel,
                                  const int8_t* __restrict__ role, const int* __restrict__ slot, int S, int M, float* __restrict__ out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)S * M * M;
  if (i >= total) return;
  int P = M / 2;
  int s = i / (M * M); int r = i - (long)s * M * M; int Y = r / M, X = r - Y * M;
  int c = cls[s];
  float v = 0.f;
  if (c >= 0 && c < K) {
    const float* row = logits + ((((size_t)s * P + (Y >> 1)) * P + (X >> 1)) * 4 + ((Y & 1) * 2 + (X & 1))) * ldk;
    float x;
    if (sim && role[c] == 2) {
      const float* sm = sim + ((size_t)s * n_novel + slot[c]) * n_base;
      x = 0.f;
      for (int b = 0; b < n_base; ++b) x += sm[b] * row[base[b]];
    } else if (sim && role[c] == 0) x = 0.f;
    else x = row[c];
    if (delta_col0 >= 0) x = x + row[delta_col0 + c];
    v = 1.f / (1.f + expf(-x));
  }
  out[i] = v;
}
extern "C" int unit_mask_probs(const float* logits, int K, int ldk, int delta_col0, const int* cls, const float* sim, const int* base_dev,
                               int n_base, int n_novel, const int8_t* role_dev, const int* slot_dev, int S, int M, float* out, void* stream) {
  if (S == 0) return UNIT_OK;
  mask_probs_kernel<<<cdiv((long)S * M * M, 256), 256, 0, (hipStream_t)stream>>>(logits, K, ldk, delta_col0, cls, sim, base_dev, n_base, n_novel,
                                                                                role_dev, slot_dev, S, M, out);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// index of the matched GT instance of every sampled RoI slot (gt_masks[matched_idx[sampled]] in label_and_sample_proposals)
__global__ void gather_match_index_kernel(const int* __restrict__ sidx, int S, const int64_t* __restrict__ midx, int Ncap, int* __restrict__ out) {
  int b = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S) return;
  int id = sidx[(size_t)b * S + i];
  out[(size_t)b * S + i] = id >= 0 ? (int)midx[(size_t)b * Ncap + id] : 0;
}
extern "C" int unit_gather_match_index(const int* sampled_idx, int S, const int64_t* match_idx, int Ncap, int B, int* out, void* stream) {
  if (B == 0 || S == 0) return UNIT_OK;
  gather_match_index_kernel<<<dim3(cdiv(S, 256), B), 256, 0, (hipStream_t)stream>>>(sampled_idx, S, match_idx, Ncap, out);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// a16 / postprocess: paste_masks_in_image of detector_postprocess (reference call site modeling/meta_arch/rcnn.py:423;
// Detectron2 v0.3 layers/mask_ops.py `_do_paste_mask`, GPU branch: skip_empty = False, whole image):
//   gx = ((x + 0.5) - x0) / (x1 - x0) * 2 - 1, gy likewise; F.grid_sample(mask[None], grid, align_corners=False) (bilinear, zero
//   padding): ix = ((gx + 1) * M - 1) / 2; out = (value >= threshold). One thread per output pixel; same operation order as ATen.
// probs [S][M][M] fp32, boxes [S][4] (already in output-image coordinates), valid [S] (0 = skip, row left zero), out uint8 [S][H][W].
// ---------------------------------------------------------------------------------------------------
__global__ void paste_masks_kernel(const float* __restrict__ probs, const float* __restrict__ boxes, const unsigned char* __restrict__ valid,
                                   int M, int H, int W, float thr, unsigned char* __restrict__ out) {
  int s = blockIdx.z, y = blockIdx.y;
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= W) return;
  size_t o = ((size_t)s * H + y) * W + x;
  if (valid && !valid[s]) { out[o] = 0; return; }
  float x0 = boxes[4 * s], y0 = boxes[4 * s + 1], x1 = boxes[4 * s + 2], y1 = boxes[4 * s + 3];
  float gx = ((float)x + 0.5f - x0) / (x1 - x0) * 2.f - 1.f;
  float gy = ((float)y + 0.5f - y0) / (y1 - y0) * 2.f - 1.f;
  float ix = ((gx + 1.f) * (float)M - 1.f) / 2.f, iy = ((gy + 1.f) * (float)M - 1.f) / 2.f;
  float fx = floorf(ix), fy = floorf(iy);
  int ix_nw = (int)fx, iy_nw = (int)fy;
  float nw = (fx + 1.f - ix) * (fy + 1.f - iy), ne = (ix - fx) * (fy + 1.f - iy);
  float sw = (fx + 1.f - ix) * (iy - fy), se = (ix - fx) * (iy - fy);
  const float* m = probs + (size_t)s * M * M;
  auto at = [&](int yy, int xx) { return (yy >= 0 && yy < M && xx >= 0 && xx < M) ? m[yy * M + xx] : 0.f; };
  float v = 0.f;
  // ATen grid_sampler_2d accumulates the in-bounds taps in the order nw, ne, sw, se
  if (iy_nw >= 0 && iy_nw < M && ix_nw >= 0 && ix_nw < M) v += at(iy_nw, ix_nw) * nw;
  if (iy_nw >= 0 && iy_nw < M && ix_nw + 1 >= 0 && ix_nw + 1 < M) v += at(iy_nw, ix_nw + 1) * ne;
  if (iy_nw + 1 >= 0 && iy_nw + 1 < M && ix_nw >= 0 && ix_nw < M) v += at(iy_nw + 1, ix_nw) * sw;
  if (iy_nw + 1 >= 0 && iy_nw + 1 < M && ix_nw + 1 >= 0 && ix_nw + 1 < M) v += at(iy_nw + 1, ix_nw + 1) * se;
  out[o] = v >= thr ? 1 : 0;
}
extern "C" int unit_paste_masks(const float* probs, const float* boxes, const unsigned char* valid, int S, int M, int H, int W,
                                float threshold, unsigned char* out, void* stream) {
  if (S == 0 || H == 0 || W == 0) return UNIT_OK;
  UNIT_CHECK_ARG(S <= 65535 && H <= 65535, "paste_masks: S, H <= 65535");
  paste_masks_kernel<<<dim3(cdiv(W, 256), H, S), 256, 0, (hipStream_t)stream>>>(probs, boxes, valid, M, H, W, threshold, out);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Mask targets from POLYGON ground truth: PolygonMasks.crop_and_resize (mask_head.py:34 -> Detectron2 mask_rcnn_loss ->
// rasterize_polygons_within_box -> pycocotools frPyObjects / merge / decode; the reference's COCO-segm yaml leaves INPUT.MASK_FORMAT at
// "polygon"). The arithmetic is pycocotools' (common/maskApi.c rleFrPoly, restated in oracle/oracle_c.c): vertices in box coordinates x
// M / side (fp64), snapped to a 5x finer integer grid with C truncation, every edge walked as a dense integer line; wherever the walk
// changes column, a "y-boundary" point (column, first row at or below the crossing) toggles everything behind it in COLUMN-MAJOR order.
// The reference sorts the points and run-length encodes; a pixel's value is simply the parity of the number of boundary points at or
// before its column-major position, which is what this kernel counts: one workgroup per foreground slot, per polygon a crossing-count
// table in LDS (waves take edges, lanes take steps of the walk), a prefix parity over the M*M positions, OR over the polygons (rleMerge).
// poly_xy [V][2] fp64 image coordinates; poly_start [P + 1]; inst_start [I + 1] = polygon range of flat instance i; image_inst0 [B] = flat
// index of an image's first instance; slot s -> instance image_inst0[rois5[s][0]] + gt_index[s]. cls outside [0, K): zeros.
__device__ __forceinline__ int ctrunc(double v) { return (int)v; }          // C conversion: toward zero (the reference's (int) casts)

__global__ void __launch_bounds__(256) mask_targets_polygon_kernel(const double* __restrict__ xy, const int* __restrict__ poly_start,
                                                                   const int* __restrict__ inst_start, const int* __restrict__ image_inst0,
                                                                   const float* __restrict__ rois5, const int* __restrict__ gt_index,
                                                                   const int* __restrict__ cls, int K, int M, unsigned char* __restrict__ out) {
  __shared__ int cnt[28 * 28 + 1];          // crossings per column-major position 0 .. M*M (position M*M = the end sentinel's)
  __shared__ unsigned char acc[28 * 28];    // union over the polygons, column-major
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int c = cls[s];
  const int MM = M * M;
  if (c < 0 || c >= K) {
    for (int i = tid; i < MM; i += 256) out[(size_t)s * MM + i] = 0;
    return;
  }
  const float* roi = rois5 + 5 * (size_t)s;
  const float bx0 = roi[1], by0 = roi[2];
  const float wf = roi[3] - roi[1], hf = roi[4] - roi[2];          // float32 sides, as box[2] - box[0] on the float32 tensor row
  const double rw = (double)M / (wf >= 0.1f ? (double)wf : 0.1), rh = (double)M / (hf >= 0.1f ? (double)hf : 0.1);
  const int inst = image_inst0[(int)roi[0]] + gt_index[s];
  const int p0 = inst_start[inst], p1 = inst_start[inst + 1];
  for (int i = tid; i < MM; i += 256) acc[i] = 0;
  for (int p = p0; p < p1; ++p) {
    const int v0 = poly_start[p], k = poly_start[p + 1] - v0;
    for (int i = tid; i <= MM; i += 256) cnt[i] = 0;
    __syncthreads();
    for (int e = wid; e < k; e += 4) {          // edge e: vertex e -> vertex (e + 1) % k
      const int e1 = e + 1 == k ? 0 : e + 1;
      const int xa = ctrunc(5.0 * ((xy[2 * (size_t)(v0 + e)] - (double)bx0) * rw) + .5), ya = ctrunc(5.0 * ((xy[2 * (size_t)(v0 + e) + 1] - (double)by0) * rh) + .5);
      const int xb = ctrunc(5.0 * ((xy[2 * (size_t)(v0 + e1)] - (double)bx0) * rw) + .5), yb = ctrunc(5.0 * ((xy[2 * (size_t)(v0 + e1) + 1] - (double)by0) * rh) + .5);
      int xs = xa, xe = xb, ys = ya, ye = yb;
      const int dx = abs(xe - xs), dy = abs(ys - ye);
      const bool flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
      if (flip) { int t = xs; xs = xe; xe = t; t = ys; ys = ye; ye = t; }
      const bool xmajor = dx >= dy;
      const int n = xmajor ? dx : dy;          // steps 0 .. n of the walk, from the ORIGINAL first vertex to the second
      const double sl = n == 0 ? 0.0 : (xmajor ? (double)(ye - ys) / dx : (double)(xe - xs) / dy);
      auto point = [&](int d, int& u, int& v) {
        const int t = flip ? n - d : d;
        if (xmajor) { u = t + xs; v = ctrunc(ys + sl * t + .5); } else { v = t + ys; u = ctrunc(xs + sl * t + .5); }
      };
      for (int d = 1 + lane; d <= n; d += 64) {          // (the junction between two edges repeats a vertex: same column, no crossing)
        int u0, w0, u1, w1;
        point(d - 1, u0, w0);
        point(d, u1, w1);
        if (u1 == u0) continue;
        double xd = (double)(u1 < u0 ? u1 : u1 - 1);
        xd = (xd + .5) / 5.0 - .5;
        if (floor(xd) != xd || xd < 0 || xd > M - 1) continue;
        double yd = (double)(w1 < w0 ? w1 : w0);
        yd = (yd + .5) / 5.0 - .5;
        if (yd < 0) yd = 0; else if (yd > M) yd = M;
        yd = ceil(yd);
        atomicAdd(&cnt[(int)xd * M + (int)yd], 1);
      }
    }
    __syncthreads();
    // pixel i (column-major) = parity of the crossings at positions <= i: wave 0 scans the M*M positions, 64 per round
    if (wid == 0) {
      int carry = 0;
      for (int base = 0; base < MM; base += 64) {
        const int i = base + lane;
        int v = i < MM ? cnt[i] : 0;
        for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
        if (i < MM && ((carry + v) & 1)) acc[i] = 1;
        carry += __shfl(v, 63, 64);
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < MM; i += 256) {          // column-major -> row-major [y][x]
    const int yy = i / M, xx = i - yy * M;
    out[(size_t)s * MM + i] = acc[xx * M + yy];
  }
}

extern "C" int unit_mask_targets_polygon(const double* poly_xy, const int* poly_start, const int* inst_start, const int* image_inst0,
                                         const float* rois5, const int* gt_index, const int* cls, int K, int S, int M, unsigned char* out,
                                         void* stream) {
  UNIT_CHECK_ARG(M >= 1 && M <= 28, "unit_mask_targets_polygon: mask side 1 .. 28");
  if (S == 0) return UNIT_OK;
  mask_targets_polygon_kernel<<<S, 256, 0, (hipStream_t)stream>>>(poly_xy, poly_start, inst_start, image_inst0, rois5, gt_index, cls, K, M, out);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
