/*
 * oracle_c.c -- TEST INFRASTRUCTURE ONLY (never linked/loaded by the product path).
 *
 * Plain-C restatement of the integer/index-exact operators of the Faster-R-CNN-C4 hot path that
 * ubc-vision/UniT executes through Detectron2 v0.3 / torchvision (both un-vendored, absent from
 * /root/reference => "parity unpinned" at that boundary; see DESIGN.md and SURVEY.md section 8c).
 * Each function cites the reference call site that reaches it and the published algorithm it restates.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC (see oracle/build_oracle.py).
 * -ffp-contract=off matters: every float expression below must round exactly like the scalar CPU/CUDA
 * reference (no FMA fusion) so that thresholds / ties compare bit-identically with the HIP kernels.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * RoIAlign (ROIAlignV2: aligned=True, sampling_ratio=0 -> adaptive grid, AVERAGE of bilinear taps)
 * reached from modeling/roi_heads/roi_heads.py:499,511,708 via detectron2 ROIPooler -> _C.roi_align_forward
 * Layout here is NCHW fp32 like the reference. out[R][C][PH][PW].
 * ---------------------------------------------------------------------------------------------- */
static inline void bilinear_setup(float y, float x, int H, int W, float* w1, float* w2, float* w3, float* w4,
                                  int* yl, int* xl, int* yh, int* xh, int* valid) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) { *valid = 0; return; }
  *valid = 1;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  float ly = y - (float)y_low, lx = x - (float)x_low;
  float hy = 1.0f - ly, hx = 1.0f - lx;
  *w1 = hy * hx; *w2 = hy * lx; *w3 = ly * hx; *w4 = ly * lx;
  *yl = y_low; *xl = x_low; *yh = y_high; *xh = x_high;
}

void oracle_roi_align_forward(const float* feat, int N, int C, int H, int W, const float* rois, int R,
                              int PH, int PW, float spatial_scale, int sampling_ratio, int aligned,
                              float* out) {
  (void)N;
#pragma omp parallel for schedule(dynamic, 1)
  for (int r = 0; r < R; ++r) {
    const float* roi = rois + 5 * r;
    int b = (int)roi[0];
    float offset = aligned ? 0.5f : 0.0f;
    float sw = roi[1] * spatial_scale - offset, sh = roi[2] * spatial_scale - offset;
    float ew = roi[3] * spatial_scale - offset, eh = roi[4] * spatial_scale - offset;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
    float bh = rh / (float)PH, bw = rw / (float)PW;
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
    float count = (float)(gh * gw > 1 ? gh * gw : 1);
    for (int c = 0; c < C; ++c) {
      const float* f = feat + ((size_t)b * C + c) * H * W;
      for (int ph = 0; ph < PH; ++ph)
        for (int pw = 0; pw < PW; ++pw) {
          float acc = 0.0f;
          for (int iy = 0; iy < gh; ++iy) {
            float y = sh + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
              float x = sw + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
              float w1, w2, w3, w4; int yl, xl, yh, xh, valid;
              bilinear_setup(y, x, H, W, &w1, &w2, &w3, &w4, &yl, &xl, &yh, &xh, &valid);
              if (!valid) continue;
              float val = w1 * f[yl * W + xl] + w2 * f[yl * W + xh] + w3 * f[yh * W + xl] + w4 * f[yh * W + xh];
              acc += val;
            }
          }
          out[(((size_t)r * C + c) * PH + ph) * PW + pw] = acc / count;
        }
    }
  }
}

/* backward: the reference uses atomicAdd (summation order undefined); the oracle accumulates in double
 * and rounds once, tests compare with a tolerance. dfeat must be zeroed by caller. */
void oracle_roi_align_backward(const float* gout, int N, int C, int H, int W, const float* rois, int R,
                               int PH, int PW, float spatial_scale, int sampling_ratio, int aligned,
                               double* dfeat) {
  (void)N;
  for (int r = 0; r < R; ++r) {
    const float* roi = rois + 5 * r;
    int b = (int)roi[0];
    float offset = aligned ? 0.5f : 0.0f;
    float sw = roi[1] * spatial_scale - offset, sh = roi[2] * spatial_scale - offset;
    float ew = roi[3] * spatial_scale - offset, eh = roi[4] * spatial_scale - offset;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
    float bh = rh / (float)PH, bw = rw / (float)PW;
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
    float count = (float)(gh * gw > 1 ? gh * gw : 1);
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
      double* d = dfeat + ((size_t)b * C + c) * H * W;
      for (int ph = 0; ph < PH; ++ph)
        for (int pw = 0; pw < PW; ++pw) {
          float g = gout[(((size_t)r * C + c) * PH + ph) * PW + pw];
          for (int iy = 0; iy < gh; ++iy) {
            float y = sh + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
              float x = sw + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
              float w1, w2, w3, w4; int yl, xl, yh, xh, valid;
              bilinear_setup(y, x, H, W, &w1, &w2, &w3, &w4, &yl, &xl, &yh, &xh, &valid);
              if (!valid) continue;
              d[yl * W + xl] += (double)(g * w1 / count);
              d[yl * W + xh] += (double)(g * w2 / count);
              d[yh * W + xl] += (double)(g * w3 / count);
              d[yh * W + xh] += (double)(g * w4 / count);
            }
          }
        }
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * pairwise IoU + Matcher in one pass.
 * IoU: detectron2.structures.pairwise_iou reached from rpn.py:41 / roi_heads.py:563 /
 *      weak_detector_fast_rcnn.py:327.   Matcher: /root/reference/modeling/matcher.py:54-120.
 * gt[M][4], boxes[Nb][4]; thresholds/labels as in Matcher.__init__ (thresholds WITHOUT the +-inf ends).
 * outputs: match_idx int64[Nb], match_label int8[Nb], match_val float[Nb].
 * ---------------------------------------------------------------------------------------------- */
static inline float iou1(const float* a, const float* b) {
  float area1 = (a[2] - a[0]) * (a[3] - a[1]);
  float area2 = (b[2] - b[0]) * (b[3] - b[1]);
  float w = fminf(a[2], b[2]) - fmaxf(a[0], b[0]);
  float h = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
  if (w < 0) w = 0;
  if (h < 0) h = 0;
  float inter = w * h;
  return inter > 0 ? inter / (area1 + area2 - inter) : 0.0f;
}

void oracle_pairwise_iou(const float* b1, int M, const float* b2, int Nb, float* out) {
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < Nb; ++n) out[(size_t)m * Nb + n] = iou1(b1 + 4 * m, b2 + 4 * n);
}

void oracle_iou_match(const float* gt, int M, const float* boxes, int Nb, const float* thresholds,
                      const int* labels, int n_thresh, int allow_low_quality, int64_t* match_idx,
                      int8_t* match_label, float* match_val) {
  if (M == 0) { /* matcher.py:68-82 */
    for (int n = 0; n < Nb; ++n) { match_idx[n] = 0; match_label[n] = (int8_t)labels[0]; match_val[n] = 0.0f; }
    return;
  }
  float* rowmax = (float*)malloc(sizeof(float) * M);
  for (int m = 0; m < M; ++m) rowmax[m] = -1.0f;
  for (int n = 0; n < Nb; ++n) {
    float best = -1.0f; int bi = 0;
    for (int m = 0; m < M; ++m) { /* torch.max(dim=0): first occurrence on ties */
      float v = iou1(gt + 4 * m, boxes + 4 * n);
      if (v > best) { best = v; bi = m; }
      if (v > rowmax[m]) rowmax[m] = v;
    }
    match_idx[n] = bi; match_val[n] = best;
    int8_t lab = 1; /* matcher.py:88-92 : levels (-inf,t0),[t0,t1),...,[t_last,inf) */
    float low = -INFINITY;
    for (int l = 0; l <= n_thresh; ++l) {
      float high = (l < n_thresh) ? thresholds[l] : INFINITY;
      if (best >= low && best < high) lab = (int8_t)labels[l];
      low = high;
    }
    match_label[n] = lab;
  }
  if (allow_low_quality) { /* matcher.py:100-120: exact fp32 equality with each gt's row max, ties included */
    for (int n = 0; n < Nb; ++n)
      for (int m = 0; m < M; ++m)
        if (iou1(gt + 4 * m, boxes + 4 * n) == rowmax[m]) { match_label[n] = 1; break; }
  }
  free(rowmax);
}

/* ------------------------------------------------------------------------------------------------
 * NMS (torchvision.ops.nms semantics) reached from rpn.py:48 (find_top_rpn_proposals) and
 * fast_rcnn.py:461 (fast_rcnn_inference). boxes must already be sorted by descending score.
 * keep[] receives indices (into the sorted order); returns count.  Suppress when IoU > thresh (strict).
 * ---------------------------------------------------------------------------------------------- */
int oracle_nms_sorted(const float* boxes, int n, float thresh, int64_t* keep) {
  uint8_t* dead = (uint8_t*)calloc(n > 0 ? n : 1, 1);
  int nk = 0;
  for (int i = 0; i < n; ++i) {
    if (dead[i]) continue;
    keep[nk++] = i;
    const float* a = boxes + 4 * i;
    float areaa = (a[2] - a[0]) * (a[3] - a[1]);
    for (int j = i + 1; j < n; ++j) {
      if (dead[j]) continue;
      const float* b = boxes + 4 * j;
      float xx1 = fmaxf(a[0], b[0]), yy1 = fmaxf(a[1], b[1]);
      float xx2 = fminf(a[2], b[2]), yy2 = fminf(a[3], b[3]);
      float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
      float inter = w * h;
      float areab = (b[2] - b[0]) * (b[3] - b[1]);
      float ovr = inter / (areaa + areab - inter);
      if (ovr > thresh) dead[j] = 1;
    }
  }
  free(dead);
  return nk;
}

/* ----------------------------------------------------------------------------------------------
 * a16 mask targets from POLYGON ground truth. TEST INFRASTRUCTURE ONLY.
 * /root/reference/modeling/roi_heads/mask_head.py:34 `mask_rcnn_loss(x, instances)` [d2-ext] calls
 * `instances.gt_masks.crop_and_resize(proposal_boxes, M)`; the reference's COCO-segm yaml sets no INPUT.MASK_FORMAT, so gt_masks are
 * Detectron2 PolygonMasks (data/dataset_mapper.py:104-106 `annotations_to_instances(..., mask_format=self.mask_format)`), whose
 * crop_and_resize is `rasterize_polygons_within_box(polygons, box, M)` per instance (detectron2/structures/masks.py):
 *     w, h = box[2] - box[0], box[3] - box[1]                      (float32, the box is a float32 tensor row)
 *     p[0::2] -= box[0]; p[1::2] -= box[1]                          (float64 polygons)
 *     p[0::2] *= M / max(w, 0.1); p[1::2] *= M / max(h, 0.1)        (quotient in float64 of the float32 side: numpy 1.x scalar promotion)
 *     mask = decode(merge(frPyObjects(polygons, M, M)))             (pycocotools: un-vendored, absent from this image -- "parity unpinned")
 * pycocotools 2.0.x common/maskApi.c, restated from the published algorithm:
 *   rleFrPoly: vertices to a 5x finer integer grid (x = (int)(5 * x + .5), C truncation), every edge walked as a dense integer line
 *   (major axis one step at a time, minor axis rounded half up), the points where the walk changes column give "y-boundary" points
 *   (column, first row at or below the crossing, on the original grid), their column-major positions are sorted, and consecutive
 *   differences are the run lengths of an RLE that starts with a run of zeros;   rleMerge(intersect = 0): union;   rleDecode.
 * This function follows that literally (point lists, qsort, run lengths, decode); the HIP kernel counts crossings per position instead.
 * xy: all vertices (x, y) of the instance's polygons, poly_start[n_poly + 1] = vertex ranges. out: M x M bytes, row-major [y][x].
 * ---------------------------------------------------------------------------------------------- */
static int oracle_uint_cmp(const void* a, const void* b) {
  unsigned c = *(const unsigned*)a, d = *(const unsigned*)b;
  return c > d ? 1 : c < d ? -1 : 0;
}

static void oracle_rle_fr_poly(const double* xy, int k, int h, int w, uint8_t* mask /* column-major h x w, OR-ed into */) {
  const double scale = 5;
  int* x = (int*)malloc(sizeof(int) * (k + 1));
  int* y = (int*)malloc(sizeof(int) * (k + 1));
  long m = 0;
  for (int j = 0; j < k; ++j) x[j] = (int)(scale * xy[j * 2 + 0] + .5);
  x[k] = x[0];
  for (int j = 0; j < k; ++j) y[j] = (int)(scale * xy[j * 2 + 1] + .5);
  y[k] = y[0];
  for (int j = 0; j < k; ++j) {
    int ax = abs(x[j] - x[j + 1]), ay = abs(y[j] - y[j + 1]);
    m += (ax > ay ? ax : ay) + 1;
  }
  int* u = (int*)malloc(sizeof(int) * (m > 0 ? m : 1));
  int* v = (int*)malloc(sizeof(int) * (m > 0 ? m : 1));
  m = 0;
  for (int j = 0; j < k; ++j) {
    int xs = x[j], xe = x[j + 1], ys = y[j], ye = y[j + 1], dx, dy, t, d;
    int flip;
    double s;
    dx = abs(xe - xs);
    dy = abs(ys - ye);
    flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
    if (flip) { t = xs; xs = xe; xe = t; t = ys; ys = ye; ye = t; }
    s = dx >= dy ? (double)(ye - ys) / dx : (double)(xe - xs) / dy;
    if (dx >= dy) for (d = 0; d <= dx; d++) {
      t = flip ? dx - d : d; u[m] = t + xs; v[m] = (int)(ys + s * t + .5); m++;
    } else for (d = 0; d <= dy; d++) {
      t = flip ? dy - d : d; v[m] = t + ys; u[m] = (int)(xs + s * t + .5); m++;
    }
  }
  free(x); free(y);
  long kk = m;
  unsigned* a = (unsigned*)malloc(sizeof(unsigned) * (kk + 1));
  m = 0;
  for (long j = 1; j < kk; ++j) if (u[j] != u[j - 1]) {
    double xd = (double)(u[j] < u[j - 1] ? u[j] : u[j] - 1);
    xd = (xd + .5) / scale - .5;
    if (floor(xd) != xd || xd < 0 || xd > w - 1) continue;
    double yd = (double)(v[j] < v[j - 1] ? v[j] : v[j - 1]);
    yd = (yd + .5) / scale - .5;
    if (yd < 0) yd = 0; else if (yd > h) yd = h;
    yd = ceil(yd);
    a[m++] = (unsigned)((int)xd * h + (int)yd);
  }
  free(u); free(v);
  a[m++] = (unsigned)(h * w);
  qsort(a, m, sizeof(unsigned), oracle_uint_cmp);
  unsigned p = 0;
  for (long j = 0; j < m; ++j) { unsigned t = a[j]; a[j] -= p; p = t; }
  unsigned* b = (unsigned*)malloc(sizeof(unsigned) * m);
  long j = 0, nb = 0;
  b[nb++] = a[j++];
  while (j < m) if (a[j] > 0) b[nb++] = a[j++]; else { j++; if (j < m) b[nb - 1] += a[j++]; }
  /* rleDecode: runs alternate 0 / 1 starting with 0, column-major */
  long pos = 0;
  uint8_t val = 0;
  for (long r = 0; r < nb; ++r) {
    for (unsigned c = 0; c < b[r] && pos < (long)h * w; ++c) { if (val) mask[pos] = 1; pos++; }
    val = !val;
  }
  free(a); free(b);
}

void oracle_rasterize_polygons_within_box(const double* xy, const int* poly_start, int n_poly, const float* box, int M, uint8_t* out) {
  float w = box[2] - box[0], h = box[3] - box[1];
  double ratio_w = (double)M / (double)(w > 0.1f ? w : 0.1f), ratio_h = (double)M / (double)(h > 0.1f ? h : 0.1f);
  if (!(w >= 0.1f)) ratio_w = (double)M / 0.1;          /* max(w, 0.1) picks the Python float 0.1 (a double) when the side is smaller */
  if (!(h >= 0.1f)) ratio_h = (double)M / 0.1;
  uint8_t* cm = (uint8_t*)calloc((size_t)M * M, 1);
  for (int p = 0; p < n_poly; ++p) {
    int k = poly_start[p + 1] - poly_start[p];
    double* q = (double*)malloc(sizeof(double) * 2 * (k > 0 ? k : 1));
    for (int i = 0; i < k; ++i) {
      q[2 * i] = (xy[2 * (poly_start[p] + i)] - (double)box[0]) * ratio_w;
      q[2 * i + 1] = (xy[2 * (poly_start[p] + i) + 1] - (double)box[1]) * ratio_h;
    }
    if (k > 0) oracle_rle_fr_poly(q, k, M, M, cm);
    free(q);
  }
  for (int yy = 0; yy < M; ++yy)
    for (int xx = 0; xx < M; ++xx) out[yy * M + xx] = cm[xx * M + yy];
  free(cm);
}
