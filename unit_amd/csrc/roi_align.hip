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
template <typename T>
__global__ void roi_align_fwd_kernel(const T* __restrict__ feat, int H, int W, int C, const float* __restrict__ rois,
                                     const int* __restrict__ roi_count, int pooled, int out_size, int bin_step, float scale,
                                     int sampling_ratio, int aligned, T* __restrict__ out) {
  int bin = blockIdx.x;
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
  if (dtype == UNIT_BF16)
    roi_align_fwd_kernel<bf16_t><<<blocks, threads, 0, st>>>((const bf16_t*)feat_nhwc, H, W, C, rois, roi_count_dev, pooled_size, out_size, bin_step,
                                                           spatial_scale, sampling_ratio, aligned, (bf16_t*)out);
  else
    roi_align_fwd_kernel<float><<<blocks, threads, 0, st>>>((const float*)feat_nhwc, H, W, C, rois, roi_count_dev, pooled_size, out_size, bin_step,
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
