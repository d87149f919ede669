// roi_align.hip -- ROIAlignV2 (aligned=True, adaptive sampling grid, AVERAGE) forward / backward, NHWC.
// Reference: detectron2 ROIPooler -> _C.roi_align_forward/backward reached from
// /root/reference/modeling/roi_heads/roi_heads.py:499,511,708.
//
// HBM-bound: per RoI bin every lane owns 8 consecutive channels (16 B bf16 / 32 B fp32), the four bilinear taps of a
// sample are four fully-coalesced row reads of the (L2/MALL-resident) res4 map, the bin result is one coalesced write.
// `bin_step` / `out_size`: out_size = PH = PW bins are produced at pooled positions (ph*bin_step, pw*bin_step) of a
// `pooled_size` grid -- bin_step=2,out_size=7,pooled_size=14 is the "strided" mode that only materialises the bins a
// stride-2 1x1 conv (Res5 conv1 + shortcut, stride_in_1x1) ever reads; bin_step=1 is the reference-identical full mode.
#include "common.h"
#include <stdlib.h>

// Workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MB L2). With bin b on XCD b % 8 every XCD touched every
// RoI: the 19.6 MB of res4 maps were fetched 13x past the L2s (264 MB per forward launch, rocprofv3 FETCH_SIZE). A contiguous run
// of work items per XCD keeps a RoI's 49 bins (forward) / a band of image rows (gather backward) behind ONE L2.
__device__ __forceinline__ long xcd_contiguous(long bid, long n) {
  long q = n / 8, r = n % 8, xcd = bid % 8, loc = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

struct RoiGeom {
  float sw, sh, bw, bh, count;
  int gh, gw, b;
};

__device__ __forceinline__ RoiGeom roi_geom(const float* __restrict__ roi, float scale, int pooled, int sampling_ratio, bool aligned) {
  RoiGeom g;
  g.b = (int)roi[0];
  float offset = aligned ? 0.5f : 0.0f;
  g.sw = roi[1] * scale - offset; g.sh = roi[2] * scale - offset;
  float ew = roi[3] * scale - offset, eh = roi[4] * scale - offset;
  float rw = ew - g.sw, rh = eh - g.sh;
  if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
  g.bh = rh / (float)pooled; g.bw = rw / (float)pooled;
  g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)pooled);
  g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pooled);
  g.count = (float)(g.gh * g.gw > 1 ? g.gh * g.gw : 1);
  return g;
}

struct Taps { float w1, w2, w3, w4; int yl, xl, yh, xh; bool valid; };

__device__ __forceinline__ Taps bilinear_taps(float y, float x, int H, int W) {
  Taps t;
  t.valid = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
  float ly = y - (float)yl, lx = x - (float)xl;
  float hy = 1.0f - ly, hx = 1.0f - lx;
  t.w1 = hy * hx; t.w2 = hy * lx; t.w3 = ly * hx; t.w4 = ly * lx;
  t.yl = yl; t.xl = xl; t.yh = yh; t.xh = xh;
  return t;
}

// grid: (R * out*out) blocks ; block: C/8 lanes (rounded up to 64)
// SR > 0: sampling_ratio == SR known at compile time -- the SR x SR sample loop is unrolled and all 4 SR^2 tap loads of the bin are issued
// before the first is used (one memory round trip per workgroup instead of SR^2 dependent ones); same arithmetic in the same order.
template <typename T, int SR>
__global__ void roi_align_fwd_kernel(const T* __restrict__ feat, int H, int W, int C, const float* __restrict__ rois,
                                     const int* __restrict__ roi_count, int pooled, int out_size, int bin_step, float scale,
                                     int sampling_ratio, int aligned, T* __restrict__ out) {
  int bin = (int)xcd_contiguous(blockIdx.x, gridDim.x);
  int r = bin / (out_size * out_size);
  int pp = bin - r * out_size * out_size;
  int oph = pp / out_size, opw = pp - oph * out_size;
  int c0 = threadIdx.x * 8;
  if (c0 >= C) return;
  T* o = out + ((size_t)bin) * C + c0;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (roi_count && r >= *roi_count) { Vec8<T>::store(o, acc); return; }
  RoiGeom g = roi_geom(rois + 5 * (size_t)r, scale, pooled, sampling_ratio, aligned != 0);
  int ph = oph * bin_step, pw = opw * bin_step;
  const T* f = feat + (size_t)g.b * H * W * C + c0;
  if constexpr (SR > 0) {
    Taps t[SR * SR];
    float v[SR * SR][4][8];
#pragma unroll
    for (int iy = 0; iy < SR; ++iy) {
      float y = g.sh + (float)ph * g.bh + ((float)iy + 0.5f) * g.bh / (float)SR;
#pragma unroll
      for (int ix = 0; ix < SR; ++ix) {
        float x = g.sw + (float)pw * g.bw + ((float)ix + 0.5f) * g.bw / (float)SR;
        Taps& q = t[iy * SR + ix];
        q = bilinear_taps(y, x, H, W);            // the taps of an invalid sample are clamped in-range addresses: loaded, not used
        Vec8<T>::load(f + ((size_t)q.yl * W + q.xl) * C, v[iy * SR + ix][0]);
        Vec8<T>::load(f + ((size_t)q.yl * W + q.xh) * C, v[iy * SR + ix][1]);
        Vec8<T>::load(f + ((size_t)q.yh * W + q.xl) * C, v[iy * SR + ix][2]);
        Vec8<T>::load(f + ((size_t)q.yh * W + q.xh) * C, v[iy * SR + ix][3]);
      }
    }
#pragma unroll
    for (int k = 0; k < SR * SR; ++k) {
      if (!t[k].valid) continue;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float val = t[k].w1 * v[k][0][i] + t[k].w2 * v[k][1][i] + t[k].w3 * v[k][2][i] + t[k].w4 * v[k][3][i];
        acc[i] += val;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = acc[i] / g.count;
    Vec8<T>::store(o, acc);
    return;
  }
  for (int iy = 0; iy < g.gh; ++iy) {
    float y = g.sh + (float)ph * g.bh + ((float)iy + 0.5f) * g.bh / (float)g.gh;
    for (int ix = 0; ix < g.gw; ++ix) {
      float x = g.sw + (float)pw * g.bw + ((float)ix + 0.5f) * g.bw / (float)g.gw;
      Taps t = bilinear_taps(y, x, H, W);
      if (!t.valid) continue;
      float v1[8], v2[8], v3[8], v4[8];
      Vec8<T>::load(f + ((size_t)t.yl * W + t.xl) * C, v1);
      Vec8<T>::load(f + ((size_t)t.yl * W + t.xh) * C, v2);
      Vec8<T>::load(f + ((size_t)t.yh * W + t.xl) * C, v3);
      Vec8<T>::load(f + ((size_t)t.yh * W + t.xh) * C, v4);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float val = t.w1 * v1[i] + t.w2 * v2[i] + t.w3 * v3[i] + t.w4 * v4[i];
        acc[i] += val;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = acc[i] / g.count;
  Vec8<T>::store(o, acc);
}

// bf16, one workgroup per (RoI, row of output bins): the row's bins are walked by the same lanes -- 7x fewer workgroups to dispatch (100 352
// two-wave workgroups of ~1 us were launch-rate-bound: 226 us for 2048 RoIs even with all tap loads of a bin in flight, 249 us before), the y
// taps of the row computed once, neighbouring bins' taps re-read through the same CU's L1.
// Sampling grids up to 2 x 2 per bin (sampling_ratio 1 or 2, or the adaptive ceil(roi / pooled) of sampling_ratio 0 for RoIs up to 2 x pooled
// feature pixels: nearly all of them): the samples of a bin mostly fall into the same one or two pixel rows / columns. The bilinear weights
// are separable (w = wy * wx, valid = valid_y && valid_x), so the bin is
//     (1 / count) * sum_k wr[k] * sum_m wc[m] * F[row_k][col_m]        over the up to 4 tap rows x 4 tap columns,
// and equal rows (columns) are merged by adding their weights: typically 4-9 loads of 16 B per lane instead of 16, zero-weight taps skipped,
// all of them in flight before the first is used. Mathematically the reference's sum in another association: bf16 outputs only (the fp32
// parity mode runs roi_align_fwd_kernel, the reference's order). Larger grids: the reference's per-sample loop, row form.
__device__ __forceinline__ void taps_1d(float v, int n, bool& ok, int& lo, int& hi, float& wlo, float& whi) {
  ok = !(v < -1.0f || v > (float)n);
  if (v <= 0.f) v = 0.f;
  lo = (int)v;
  if (lo >= n - 1) { hi = lo = n - 1; v = (float)lo; } else { hi = lo + 1; }
  whi = v - (float)lo; wlo = 1.0f - whi;
}

template <typename T>
__global__ void __launch_bounds__(256) roi_align_fwd_row_kernel(const T* __restrict__ feat, int H, int W, int C, const float* __restrict__ rois,
                                                                const int* __restrict__ roi_count, int pooled, int out_size, int bin_step,
                                                                float scale, int sampling_ratio, int aligned, T* __restrict__ out) {
  int row = (int)xcd_contiguous(blockIdx.x, gridDim.x);
  int r = row / out_size, oph = row - r * out_size;
  int c0 = threadIdx.x * 8;
  if (c0 >= C) return;
  T* o = out + ((size_t)row * out_size) * C + c0;
  if (roi_count && r >= *roi_count) {
    float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int opw = 0; opw < out_size; ++opw) Vec8<T>::store(o + (size_t)opw * C, z);
    return;
  }
  RoiGeom g = roi_geom(rois + 5 * (size_t)r, scale, pooled, sampling_ratio, aligned != 0);
  int ph = oph * bin_step;
  const T* f = feat + (size_t)g.b * H * W * C + c0;
  if (g.gh <= 2 && g.gw <= 2) {
    int rr[4]; float wr[4];
#pragma unroll
    for (int iy = 0; iy < 2; ++iy) {
      float y = g.sh + (float)ph * g.bh + ((float)iy + 0.5f) * g.bh / (float)g.gh;
      bool ok; float wl, wh;
      taps_1d(y, H, ok, rr[2 * iy], rr[2 * iy + 1], wl, wh);
      ok = ok && iy < g.gh;
      wr[2 * iy] = ok ? wl : 0.f; wr[2 * iy + 1] = ok ? wh : 0.f;
    }
#pragma unroll
    for (int k = 1; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < k; ++j)
        if (wr[k] != 0.f && rr[k] == rr[j]) { wr[j] += wr[k]; wr[k] = 0.f; }
    for (int opw = 0; opw < out_size; ++opw) {
      int pw = opw * bin_step;
      int cc[4]; float wc[4];
#pragma unroll
      for (int ix = 0; ix < 2; ++ix) {
        float x = g.sw + (float)pw * g.bw + ((float)ix + 0.5f) * g.bw / (float)g.gw;
        bool ok; float wl, wh;
        taps_1d(x, W, ok, cc[2 * ix], cc[2 * ix + 1], wl, wh);
        ok = ok && ix < g.gw;
        wc[2 * ix] = ok ? wl : 0.f; wc[2 * ix + 1] = ok ? wh : 0.f;
      }
#pragma unroll
      for (int k = 1; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < k; ++j)
          if (wc[k] != 0.f && cc[k] == cc[j]) { wc[j] += wc[k]; wc[k] = 0.f; }
      bf16x8 raw[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (wr[k] != 0.f && wc[m] != 0.f) raw[k][m] = *reinterpret_cast<const bf16x8*>(f + ((size_t)rr[k] * W + cc[m]) * C);
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (wr[k] == 0.f) continue;
        float ra[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          if (wc[m] == 0.f) continue;
#pragma unroll
          for (int i = 0; i < 8; ++i) ra[i] += wc[m] * (float)raw[k][m][i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += wr[k] * ra[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = acc[i] / g.count;
      Vec8<T>::store(o + (size_t)opw * C, acc);
    }
    return;
  }
  for (int opw = 0; opw < out_size; ++opw) {          // large RoIs (more than 2 samples per bin and axis): the reference's loop
    int pw = opw * bin_step;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int iy = 0; iy < g.gh; ++iy) {
      float y = g.sh + (float)ph * g.bh + ((float)iy + 0.5f) * g.bh / (float)g.gh;
      for (int ix = 0; ix < g.gw; ++ix) {
        float x = g.sw + (float)pw * g.bw + ((float)ix + 0.5f) * g.bw / (float)g.gw;
        Taps t = bilinear_taps(y, x, H, W);
        if (!t.valid) continue;
        float v1[8], v2[8], v3[8], v4[8];
        Vec8<T>::load(f + ((size_t)t.yl * W + t.xl) * C, v1);
        Vec8<T>::load(f + ((size_t)t.yl * W + t.xh) * C, v2);
        Vec8<T>::load(f + ((size_t)t.yh * W + t.xl) * C, v3);
        Vec8<T>::load(f + ((size_t)t.yh * W + t.xh) * C, v4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float val = t.w1 * v1[i] + t.w2 * v2[i] + t.w3 * v3[i] + t.w4 * v4[i];
          acc[i] += val;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = acc[i] / g.count;
    Vec8<T>::store(o + (size_t)opw * C, acc);
  }
}

extern "C" int unit_roi_align_fwd(const void* feat_nhwc, int dtype, int N, int H, int W, int C, const float* rois,
                                  const int* roi_count_dev, int R, int pooled_size, int out_size, int bin_step,
                                  float spatial_scale, int sampling_ratio, int aligned, void* out, void* stream) {
  (void)N;
  UNIT_CHECK_ARG(C % 8 == 0 && C / 8 <= 1024, "roi_align: C % 8 != 0 or C > 8192");
  UNIT_CHECK_ARG((out_size - 1) * bin_step < pooled_size, "roi_align: out_size/bin_step exceed pooled_size");
  if (R == 0) return UNIT_OK;
  int threads = ((C / 8 + 63) / 64) * 64;
  long blocks = (long)R * out_size * out_size;
  hipStream_t st = (hipStream_t)stream;
  static int row_form = -1;
  if (row_form < 0) { const char* e = getenv("UNIT_ROI_ROW_FORM"); row_form = e ? atoi(e) : 1; }
  if (row_form && dtype == UNIT_BF16 && threads <= 256) {
    roi_align_fwd_row_kernel<bf16_t><<<(long)R * out_size, threads, 0, st>>>((const bf16_t*)feat_nhwc, H, W, C, rois, roi_count_dev, pooled_size, out_size,
                                                                           bin_step, spatial_scale, sampling_ratio, aligned, (bf16_t*)out);
    UNIT_LAUNCH_CHECK();
    return UNIT_OK;
  }
  if (dtype == UNIT_BF16 && sampling_ratio == 2)
    roi_align_fwd_kernel<bf16_t, 2><<<blocks, threads, 0, st>>>((const bf16_t*)feat_nhwc, H, W, C, rois, roi_count_dev, pooled_size, out_size, bin_step,
                                                              spatial_scale, sampling_ratio, aligned, (bf16_t*)out);
  else if (dtype == UNIT_BF16)
    roi_align_fwd_kernel<bf16_t, 0><<<blocks, threads, 0, st>>>((const bf16_t*)feat_nhwc, H, W, C, rois, roi_count_dev, pooled_size, out_size, bin_step,
                                                              spatial_scale, sampling_ratio, aligned, (bf16_t*)out);
  else
    roi_align_fwd_kernel<float, 0><<<blocks, threads, 0, st>>>((const float*)feat_nhwc, H, W, C, rois, roi_count_dev, pooled_size, out_size, bin_step,
                                                             spatial_scale, sampling_ratio, aligned, (float*)out);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// backward: dfeat32 (fp32 NHWC accumulator, caller-zeroed) += g/count * w at the four taps (atomic fp32 adds, shaped as
// contiguous channel runs per wave-instruction; same non-deterministic summation order as the reference's atomicAdd).
template <typename T>
__global__ void roi_align_bwd_kernel(const T* __restrict__ gout, int H, int W, int C, const float* __restrict__ rois,
                                     const int* __restrict__ roi_count, int pooled, int out_size, int bin_step, float scale,
                                     int sampling_ratio, int aligned, float* __restrict__ dfeat) {
  // lane l owns channels l, l+256, ... so that every atomic wave-instruction covers 256 contiguous bytes
  int bin = blockIdx.x;
  int r = bin / (out_size * out_size);
  int pp = bin - r * out_size * out_size;
  int oph = pp / out_size, opw = pp - oph * out_size;
  if (roi_count && r >= *roi_count) return;
  RoiGeom g = roi_geom(rois + 5 * (size_t)r, scale, pooled, sampling_ratio, aligned != 0);
  int ph = oph * bin_step, pw = opw * bin_step;
  float* d = dfeat + (size_t)g.b * H * W * C;
  const T* go = gout + (size_t)bin * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float gg = (float)go[c];
    for (int iy = 0; iy < g.gh; ++iy) {
      float y = g.sh + (float)ph * g.bh + ((float)iy + 0.5f) * g.bh / (float)g.gh;
      for (int ix = 0; ix < g.gw; ++ix) {
        float x = g.sw + (float)pw * g.bw + ((float)ix + 0.5f) * g.bw / (float)g.gw;
        Taps t = bilinear_taps(y, x, H, W);
        if (!t.valid) continue;
        atomicAdd(d + ((size_t)t.yl * W + t.xl) * C + c, gg * t.w1 / g.count);
        atomicAdd(d + ((size_t)t.yl * W + t.xh) * C + c, gg * t.w2 / g.count);
        atomicAdd(d + ((size_t)t.yh * W + t.xl) * C + c, gg * t.w3 / g.count);
        atomicAdd(d + ((size_t)t.yh * W + t.xh) * C + c, gg * t.w4 / g.count);
      }
    }
  }
}

extern "C" int unit_roi_align_bwd(const void* gout, int dtype, int N, int H, int W, int C, const float* rois,
                                  const int* roi_count_dev, int R, int pooled_size, int out_size, int bin_step,
                                  float spatial_scale, int sampling_ratio, int aligned, float* dfeat_f32, void* stream) {
  (void)N;
  if (R == 0) return UNIT_OK;
  int threads = C >= 256 ? 256 : ((C + 63) / 64) * 64;
  long blocks = (long)R * out_size * out_size;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UNIT_BF16)
    roi_align_bwd_kernel<bf16_t><<<blocks, threads, 0, st>>>((const bf16_t*)gout, H, W, C, rois, roi_count_dev, pooled_size, out_size, bin_step,
                                                           spatial_scale, sampling_ratio, aligned, dfeat_f32);
  else
    roi_align_bwd_kernel<float><<<blocks, threads, 0, st>>>((const float*)gout, H, W, C, rois, roi_count_dev, pooled_size, out_size, bin_step,
                                                          spatial_scale, sampling_ratio, aligned, dfeat_f32);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// =====================================================================================================================
// Deterministic RoIAlign backward in GATHER form (no atomics): one workgroup per feature-map pixel, lanes over channels.
//   dfeat[n,y,x,c] = sum_r  (1/count_r) * sum_ph Wy_r[ph](y) * sum_pw Wx_r[pw](x) * g[r,ph,pw,c]
// where Wy_r[ph](y) = sum_iy (1-D bilinear weight of sample (ph,iy) on row y) -- the 2-D bilinear weights of ROIAlignV2
// factor into a row term and a column term, so the per-sample scatter of the reference's atomicAdd kernel becomes a
// per-pixel sum over the (few) RoI bins that touch the pixel. Summation order is fixed -> bit-reproducible gradients.
// Optionally fuses the consumer: out = cast((acc + addend) * (mask_ref > 0)) (the RPN-branch gradient and the ReLU mask
// of the res4 output), which removes the fp32 accumulator round trip.
// =====================================================================================================================
struct RoiG { float sw, sh, bw, bh, inv_count; int gh, gw, b, y0, y1, x0, x1; };

__global__ void roi_geom_kernel(const float* __restrict__ rois, const int* __restrict__ roi_count, int R, int H, int W, int pooled,
                                int out_size, int bin_step, float scale, int sampling_ratio, int aligned, RoiG* __restrict__ tab) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  RoiG t;
  if (roi_count && r >= *roi_count) { t.b = -1; t.y0 = 1; t.y1 = 0; t.x0 = 1; t.x1 = 0; t.gh = t.gw = 0; t.sw = t.sh = t.bw = t.bh = t.inv_count = 0.f; tab[r] = t; return; }
  RoiGeom g = roi_geom(rois + 5 * (size_t)r, scale, pooled, sampling_ratio, aligned != 0);
  t.sw = g.sw; t.sh = g.sh; t.bw = g.bw; t.bh = g.bh; t.inv_count = 1.0f / g.count; t.gh = g.gh; t.gw = g.gw; t.b = g.b;
  int plast = (out_size - 1) * bin_step;
  if (g.gh <= 0 || g.gw <= 0) { t.y0 = 1; t.y1 = 0; t.x0 = 1; t.x1 = 0; }
  else {
    float yf = g.sh + 0.5f * g.bh / (float)g.gh, yl = g.sh + (float)plast * g.bh + ((float)g.gh - 0.5f) * g.bh / (float)g.gh;
    float xf = g.sw + 0.5f * g.bw / (float)g.gw, xl = g.sw + (float)plast * g.bw + ((float)g.gw - 0.5f) * g.bw / (float)g.gw;
    t.y0 = max(0, (int)floorf(fminf(yf, yl)) - 1); t.y1 = min(H - 1, (int)floorf(fmaxf(yf, yl)) + 2);
    t.x0 = max(0, (int)floorf(fminf(xf, xl)) - 1); t.x1 = min(W - 1, (int)floorf(fmaxf(xf, xl)) + 2);
  }
  tab[r] = t;
}

// 1-D weight of sample coordinate s on pixel index p (same clamping rules as bilinear_taps)
__device__ __forceinline__ float tap1d(float s, int p, int L) {
  if (s < -1.0f || s > (float)L) return 0.f;
  if (s <= 0.f) s = 0.f;
  int lo = (int)s, hi;
  if (lo >= L - 1) { hi = lo = L - 1; s = (float)lo; } else hi = lo + 1;
  float l = s - (float)lo, h = 1.0f - l;
  return (lo == p ? h : 0.f) + (hi == p ? l : 0.f);
}

template <typename T, typename TOUT>
__global__ void roi_align_bwd_gather_kernel(const T* __restrict__ gout, int H, int W, int C, const RoiG* __restrict__ tab, int R,
                                            int rois_per_image, int image_offset, int out_size, int bin_step,
                                            const T* __restrict__ addend, int addend_images, const T* __restrict__ mask_ref,
                                            TOUT* __restrict__ dfeat) {
  extern __shared__ float wlds[];   // per wave: 2 * 16 floats
  int pix = (int)xcd_contiguous(blockIdx.x, gridDim.x);
  int n = pix / (H * W); int rem = pix - n * H * W; int py = rem / W, px = rem - py * W;
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* wy_s = wlds + wave * 32; float* wx_s = wy_s + 16;
  int c0 = threadIdx.x * 8;
  bool cvalid = c0 < C;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int r_begin = 0, r_end = R;
  if (rois_per_image > 0) { r_begin = n * rois_per_image; r_end = min(R, r_begin + rois_per_image); }
  // 64 RoIs are tested per iteration (one per lane, bounding box of their sample taps); only the hits are visited
  for (int rb = r_begin; rb < r_end; rb += 64) {
    bool hit = false;
    if (rb + lane < r_end) {
      const RoiG& q = tab[rb + lane];
      hit = q.b == n + image_offset && py >= q.y0 && py <= q.y1 && px >= q.x0 && px <= q.x1;
    }
    for (unsigned long long hm = __ballot(hit); hm; hm &= hm - 1) {
      int r = rb + __ffsll((long long)hm) - 1;
      RoiG t = tab[r];                                   // wave-uniform
      // lanes 0..out-1: row weights of bin (lane*step); lanes 32..32+out-1: column weights
      float wv = 0.f;
      int o = lane & 31;
      if (o < out_size) {
        bool isx = lane >= 32;
        int p = o * bin_step;
        float start = isx ? t.sw : t.sh, bsz = isx ? t.bw : t.bh;
        int gn = isx ? t.gw : t.gh; int L = isx ? W : H; int pp = isx ? px : py;
        for (int i = 0; i < gn; ++i) {
          float s = start + (float)p * bsz + ((float)i + 0.5f) * bsz / (float)gn;
          wv += tap1d(s, pp, L);
        }
      }
      unsigned long long nz = __ballot(wv != 0.f);
      unsigned ymask = (unsigned)(nz & 0xFFFFFFFFull), xmask = (unsigned)(nz >> 32);
      if (ymask == 0u || xmask == 0u) continue;
      if (o < out_size) { if (lane < 32) wy_s[o] = wv; else wx_s[o] = wv; }
      __builtin_amdgcn_wave_barrier();
      if (cvalid) {
        const T* gr = gout + (size_t)r * out_size * out_size * C + c0;
        for (unsigned ym = ymask; ym; ym &= ym - 1) {
          int oy = __ffs(ym) - 1;
          float wy = wy_s[oy] * t.inv_count;
          for (unsigned xm = xmask; xm; xm &= xm - 1) {
            int ox = __ffs(xm) - 1;
            float w = wy * wx_s[ox];
            float v[8];
            Vec8<T>::load(gr + ((size_t)oy * out_size + ox) * C, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += w * v[j];
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (!cvalid) return;
  size_t o = (size_t)pix * C + c0;
  if (addend && n < addend_images) {
    float a[8]; Vec8<T>::load(addend + o, a);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += a[j];
  }
  if (mask_ref) {
    float m[8]; Vec8<T>::load(mask_ref + o, m);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = m[j] > 0.f ? acc[j] : 0.f;
  }
  Vec8<TOUT>::store(dfeat + o, acc);
}

extern "C" size_t unit_roi_align_bwd_gather_workspace_bytes(int R) { return sizeof(RoiG) * (size_t)(R > 0 ? R : 1); }

// dfeat (out_dtype: 0 fp32 / 1 bf16) [N,H,W,C] is fully written (no pre-zeroing needed).
// rois_per_image > 0 asserts that the RoIs of local image i occupy slots [i*S, (i+1)*S) (fixed-slot layout of the step).
// image_offset: dfeat/addend/mask_ref point at image `image_offset` of the batch; RoI batch indices are global.
extern "C" int unit_roi_align_bwd_gather(const void* gout, int dtype, int N, int H, int W, int C, const float* rois,
                                         const int* roi_count_dev, int R, int rois_per_image, int image_offset, int pooled_size, int out_size,
                                         int bin_step, float spatial_scale, int sampling_ratio, int aligned, const void* addend,
                                         int addend_images, const void* mask_ref, void* dfeat, int out_dtype, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  UNIT_CHECK_ARG(C % 8 == 0 && C / 8 <= 1024 && out_size <= 16, "roi_align_bwd_gather: C % 8, C <= 8192, out_size <= 16");
  UNIT_CHECK_ARG(out_dtype == UNIT_F32 || out_dtype == dtype, "roi_align_bwd_gather: out dtype must be fp32 or the input dtype");
  if (workspace_bytes < unit_roi_align_bwd_gather_workspace_bytes(R)) { unit_set_error("roi_align_bwd_gather: workspace too small"); return UNIT_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  RoiG* tab = (RoiG*)workspace;
  if (R > 0) {
    roi_geom_kernel<<<cdiv(R, 256), 256, 0, st>>>(rois, roi_count_dev, R, H, W, pooled_size, out_size, bin_step, spatial_scale,
                                                sampling_ratio, aligned, tab);
    UNIT_LAUNCH_CHECK();
  }
  int threads = ((C / 8 + 63) / 64) * 64;
  size_t lds = (size_t)(threads / 64) * 32 * sizeof(float);
  long blocks = (long)N * H * W;
  if (blocks == 0) return UNIT_OK;
#define LAUNCH_G(T, TOUT) roi_align_bwd_gather_kernel<T, TOUT><<<blocks, threads, lds, st>>>((const T*)gout, H, W, C, tab, R, rois_per_image, \
    image_offset, out_size, bin_step, (const T*)addend, addend_images, (const T*)mask_ref, (TOUT*)dfeat)
  if (dtype == UNIT_BF16 && out_dtype == UNIT_BF16) LAUNCH_G(bf16_t, bf16_t);
  else if (dtype == UNIT_BF16) LAUNCH_G(bf16_t, float);
  else LAUNCH_G(float, float);
#undef LAUNCH_G
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
