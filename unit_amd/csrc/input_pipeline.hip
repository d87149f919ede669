// input_pipeline.hip -- SURVEY section 8(f) row 4: the training input pipeline of the reference on the device.
//   reference: data/dataset_mapper.py:13-31, data/build.py:476-497 (Detectron2 DatasetMapper: ResizeShortestEdge + RandomFlip,
//   uint8 HWC image -> Pillow BILINEAR resize -> float32 CHW), then modeling/meta_arch/rcnn.py:257-266 (normalise, pad, batch).
// Pillow's 8-bit resampler (src/libImaging/Resample.c) is an antialiased separable filter in 22-bit fixed point with a uint8
// intermediate between the horizontal and the vertical pass; the coefficient tables (bounds + integer taps per output
// column / row) are computed on the host in double precision exactly as Pillow does (unit_amd/data_pipeline.py) and the two
// passes below are pure integer arithmetic: results are bit-identical to Pillow. HBM-bound byte work, one thread per output
// pixel (all channels), coalesced along x.
#include "common.h"

#define RESIZE_PRECISION_BITS 22

__global__ void resize_u8_h_kernel(const unsigned char* __restrict__ src, int H, int W, int C, const int* __restrict__ bounds,
                                   const int* __restrict__ kk, int ksize, int OW, unsigned char* __restrict__ dst) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)H * OW) return;
  int y = (int)(idx / OW), xx = (int)(idx - (long)y * OW);
  int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
  const int* k = kk + (size_t)xx * ksize;
  const unsigned char* row = src + ((size_t)y * W + xmin) * C;
  for (int c = 0; c < C; ++c) {
    int ss = 1 << (RESIZE_PRECISION_BITS - 1);
    for (int x = 0; x < xmax; ++x) ss += (int)row[(size_t)x * C + c] * k[x];
    int v = ss >> RESIZE_PRECISION_BITS;
    dst[((size_t)y * OW + xx) * C + c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

__global__ void resize_u8_v_kernel(const unsigned char* __restrict__ src, int H, int W, int C, const int* __restrict__ bounds,
                                   const int* __restrict__ kk, int ksize, int OH, unsigned char* __restrict__ dst) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)OH * W) return;
  int yy = (int)(idx / W), x = (int)(idx - (long)yy * W);
  int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
  const int* k = kk + (size_t)yy * ksize;
  for (int c = 0; c < C; ++c) {
    int ss = 1 << (RESIZE_PRECISION_BITS - 1);
    for (int y = 0; y < ymax; ++y) ss += (int)src[((size_t)(ymin + y) * W + x) * C + c] * k[y];
    int v = ss >> RESIZE_PRECISION_BITS;
    dst[((size_t)yy * W + x) * C + c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

// one resampling pass of a uint8 [H][W][C] image. axis 1: -> [H][out][C] ; axis 0: -> [out][W][C].
// bounds int32 [out][2] = (first source index, number of taps), kk int32 [out][ksize] (22-bit fixed point), both on the device.
extern "C" int unit_resize_u8_pass(const unsigned char* src, int H, int W, int C, int axis, const int* bounds, const int* kk, int ksize,
                                   int out_size, unsigned char* dst, void* stream) {
  UNIT_CHECK_ARG(C >= 1 && C <= 4 && (axis == 0 || axis == 1) && ksize >= 1, "resize_u8_pass: bad arguments");
  if (H == 0 || W == 0 || out_size == 0) return UNIT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (axis == 1) resize_u8_h_kernel<<<cdiv((long)H * out_size, 256), 256, 0, st>>>(src, H, W, C, bounds, kk, ksize, out_size, dst);
  else resize_u8_v_kernel<<<cdiv((long)out_size * W, 256), 256, 0, st>>>(src, H, W, C, bounds, kk, ksize, out_size, dst);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// uint8 HWC image (already resized) -> one slot of the model's input batch: optional horizontal flip, `.astype(float32)`,
// (x[/prescale] - mean) / std (rcnn.py:257-266), zero padding to [Hmax][Wmax], channel padding to Cpad, NHWC.
template <typename T>
__global__ void preprocess_u8_kernel(const unsigned char* __restrict__ img, int C, int H, int W, int hflip, f32x4 mean, f32x4 stdv,
                                     float prescale, T* __restrict__ out, int Hmax, int Wmax, int Cpad) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Hmax * Wmax) return;
  int y = idx / Wmax, x = idx - y * Wmax;
  T* o = out + (size_t)idx * Cpad;
  bool in = (y < H) && (x < W);
  int sx = hflip ? W - 1 - x : x;
  for (int c = 0; c < Cpad; ++c) {
    float v = 0.f;
    if (in && c < C) {
      float p = (float)img[((size_t)y * W + sx) * C + c];
      if (prescale != 1.0f) p = p / prescale;
      v = (p - mean[c]) / stdv[c];
    }
    o[c] = (T)v;
  }
}

extern "C" int unit_preprocess_u8(const unsigned char* img_hwc, int C, int H, int W, int hflip, const float* mean3, const float* std3,
                                  float prescale, void* out_nhwc, int out_dtype, int Hmax, int Wmax, int Cpad, void* stream) {
  UNIT_CHECK_ARG(C <= 4 && Cpad >= C && H <= Hmax && W <= Wmax, "preprocess_u8: bad shape");
  f32x4 m = {0, 0, 0, 0}, s = {1, 1, 1, 1};
  for (int c = 0; c < C; ++c) { m[c] = mean3[c]; s[c] = std3[c]; }
  int n = Hmax * Wmax;
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == UNIT_BF16)
    preprocess_u8_kernel<bf16_t><<<cdiv(n, 256), 256, 0, st>>>(img_hwc, C, H, W, hflip, m, s, prescale, (bf16_t*)out_nhwc, Hmax, Wmax, Cpad);
  else if (out_dtype == UNIT_F32)
    preprocess_u8_kernel<float><<<cdiv(n, 256), 256, 0, st>>>(img_hwc, C, H, W, hflip, m, s, prescale, (float*)out_nhwc, Hmax, Wmax, Cpad);
  else { unit_set_error("preprocess_u8: unsupported out dtype"); return UNIT_ERR_UNSUPPORTED; }
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
