// elementwise.hip -- HBM-bound helper kernels of the Faster-R-CNN-C4 hot path (NHWC, 16 B / lane).
//   preprocess (a1), maxpool 3x3 s2 (a2 stem), global avg-pool fwd/bwd (a9), weight prep (FrozenBN fold +
//   bf16 cast + dgrad re-layout), bias grad, SGD momentum (K18), small fills/casts.
#include "common.h"

static thread_local char g_err[512] = "";
extern "C" void unit_set_error(const char* msg) { strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1); }
extern "C" const char* unit_last_error(void) { return g_err; }
extern "C" int unit_version(void) { return 100; }

// ---------------------------------------------------------------------------------------------------
// a1  preprocess_image: modeling/meta_arch/rcnn.py:257-266 -- (x[*prescale] - mean) / std, zero pad to
// (Hmax,Wmax), CHW fp32 -> one image slot of an NHWC batch with channels padded to Cpad (pad channels = 0).
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void preprocess_kernel(const float* __restrict__ img, int C, int H, int W, f32x4 mean, f32x4 stdv,
                                  float prescale, T* __restrict__ out, int Hmax, int Wmax, int Cpad) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Hmax * Wmax) return;
  int y = idx / Wmax, x = idx - y * Wmax;
  T* o = out + (size_t)idx * Cpad;
  bool in = (y < H) && (x < W);
  if (Cpad == 8) {          // the model's stem input: one 16-B (bf16) / two 16-B (fp32) stores per pixel instead of eight scalar ones
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float t = 0.f;
      if (in && c < C) {
        float p = img[((size_t)c * H + y) * W + x];
        if (prescale != 1.0f) p = p / prescale;
        t = (p - mean[c & 3]) / stdv[c & 3];
      }
      v[c] = t;
    }
    Vec8<T>::store(o, v);
    return;
  }
  for (int c = 0; c < Cpad; ++c) {
    float v = 0.f;
    if (in && c < C) {
      float p = img[((size_t)c * H + y) * W + x];
      if (prescale != 1.0f) p = p / prescale;  // normalize_images: x/255.0 (rcnn.py:262-263); prescale = divisor
      v = (p - mean[c]) / stdv[c];
    }
    o[c] = (T)v;
  }
}

extern "C" int unit_preprocess_image(const float* img_chw, int C, int H, int W, const float* mean3,
                                     const float* std3, float prescale, void* out_nhwc, int out_dtype,
                                     int Hmax, int Wmax, int Cpad, void* stream) {
  UNIT_CHECK_ARG(C <= 4 && Cpad >= C && H <= Hmax && W <= Wmax, "preprocess: bad shape");
  f32x4 m = {0, 0, 0, 0}, s = {1, 1, 1, 1};
  for (int c = 0; c < C; ++c) { m[c] = mean3[c]; s[c] = std3[c]; }
  int n = Hmax * Wmax;
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == UNIT_BF16)
    preprocess_kernel<bf16_t><<<cdiv(n, 256), 256, 0, st>>>(img_chw, C, H, W, m, s, prescale, (bf16_t*)out_nhwc, Hmax, Wmax, Cpad);
  else
    preprocess_kernel<float><<<cdiv(n, 256), 256, 0, st>>>(img_chw, C, H, W, m, s, prescale, (float*)out_nhwc, Hmax, Wmax, Cpad);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// a2  stem max_pool2d(k3,s2,p1) NHWC, 8 channels per lane (no backward: FREEZE_AT=2 freezes stem+res2)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool3x3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C,
                                    int OH, int OW) {
  int c8n = C / 8;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)N * OH * OW * c8n;
  if (idx >= total) return;
  int c8 = idx % c8n; long p = idx / c8n;
  int ow = p % OW; p /= OW;
  int oh = p % OH; int n = p / OH;
  float m[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) m[i] = -INFINITY;
  for (int dy = 0; dy < 3; ++dy) {
    int iy = oh * 2 - 1 + dy;
    if (iy < 0 || iy >= H) continue;
    for (int dx = 0; dx < 3; ++dx) {
      int ix = ow * 2 - 1 + dx;
      if (ix < 0 || ix >= W) continue;
      float v[8];
      Vec8<T>::load(x + (((size_t)n * H + iy) * W + ix) * C + c8 * 8, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], v[i]);
    }
  }
  Vec8<T>::store(y + (((size_t)n * OH + oh) * OW + ow) * C + c8 * 8, m);
}

extern "C" int unit_maxpool3x3s2_fwd(const void* x, void* y, int dtype, int N, int H, int W, int C, void* stream) {
  UNIT_CHECK_ARG(C % 8 == 0, "maxpool: C % 8 != 0");
  int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  long total = (long)N * OH * OW * (C / 8);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UNIT_BF16)
    maxpool3x3s2_kernel<bf16_t><<<cdiv(total, 256), 256, 0, st>>>((const bf16_t*)x, (bf16_t*)y, N, H, W, C, OH, OW);
  else
    maxpool3x3s2_kernel<float><<<cdiv(total, 256), 256, 0, st>>>((const float*)x, (float*)y, N, H, W, C, OH, OW);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// a9  Res5BoxHead x.mean(dim=[2,3]) : modeling/roi_heads/box_head.py:80.  [R][P][C] -> [R][C] (fp32 accumulate)
// backward fused with the ReLU mask of the last res5 block: g[r,p,c] = (out[r,p,c] > 0) ? dfeat[r,c]/P : 0
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int R, int P, int C) {
  int c8n = C / 8;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)R * c8n) return;
  int c8 = idx % c8n; int r = idx / c8n;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int p = 0; p < P; ++p) {
    float v[8];
    Vec8<T>::load(x + ((size_t)r * P + p) * C + c8 * 8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += v[i];
  }
  float inv = (float)P;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = acc[i] / inv;
  Vec8<T>::store(y + (size_t)r * C + c8 * 8, acc);
}

template <typename T>
__global__ void avgpool_bwd_mask_kernel(const T* __restrict__ dfeat, const T* __restrict__ out, T* __restrict__ g,
                                        int R, int P, int C) {
  int c8n = C / 8;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)R * P * c8n) return;
  int c8 = idx % c8n; long rp = idx / c8n; int r = rp / P;
  float d[8], o[8];
  Vec8<T>::load(dfeat + (size_t)r * C + c8 * 8, d);
  Vec8<T>::load(out + (size_t)rp * C + c8 * 8, o);
  float inv = (float)P;
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = o[i] > 0.f ? d[i] / inv : 0.f;
  Vec8<T>::store(g + (size_t)rp * C + c8 * 8, d);
}

extern "C" int unit_global_avgpool_fwd(const void* x, void* y, int dtype, int R, int P, int C, void* stream) {
  UNIT_CHECK_ARG(C % 8 == 0, "avgpool: C % 8 != 0");
  long total = (long)R * (C / 8);
  hipStream_t st = (hipStream_t)stream;
  if (total == 0) return UNIT_OK;
  if (dtype == UNIT_BF16) avgpool_fwd_kernel<bf16_t><<<cdiv(total, 256), 256, 0, st>>>((const bf16_t*)x, (bf16_t*)y, R, P, C);
  else avgpool_fwd_kernel<float><<<cdiv(total, 256), 256, 0, st>>>((const float*)x, (float*)y, R, P, C);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

extern "C" int unit_global_avgpool_bwd_relu(const void* dfeat, const void* out, void* g, int dtype, int R, int P,
                                            int C, void* stream) {
  UNIT_CHECK_ARG(C % 8 == 0, "avgpool: C % 8 != 0");
  long total = (long)R * P * (C / 8);
  hipStream_t st = (hipStream_t)stream;
  if (total == 0) return UNIT_OK;
  if (dtype == UNIT_BF16) avgpool_bwd_mask_kernel<bf16_t><<<cdiv(total, 256), 256, 0, st>>>((const bf16_t*)dfeat, (const bf16_t*)out, (bf16_t*)g, R, P, C);
  else avgpool_bwd_mask_kernel<float><<<cdiv(total, 256), 256, 0, st>>>((const float*)dfeat, (const float*)out, (float*)g, R, P, C);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// Weight prep: master weights fp32 [K][R][S][C] (= channels_last view of the reference's [K,C,R,S] tensor),
// FrozenBN scale folded (scale[k] = bn.weight * rsqrt(var+eps)) ->
//   w_fwd  [K][R][S][Cp]            (Cp >= C, zero padded)  -- forward implicit-GEMM B operand
//   w_dgrad[C][R][S][K]  with (r,s) flipped                -- dgrad = forward conv of dy with this tensor
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void weight_prep_kernel(const float* __restrict__ w, const float* __restrict__ scale, int K, int R, int S,
                                   int C, int Cp, T* __restrict__ wf, T* __restrict__ wd) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)K * R * S * Cp;
  if (idx >= total) return;
  int c = idx % Cp; long t = idx / Cp;
  int s = t % S; t /= S;
  int r = t % R; int k = t / R;
  float v = 0.f;
  if (c < C) {
    v = w[(((size_t)k * R + r) * S + s) * C + c];
    if (scale) v = v * scale[k];
  }
  if (wf) wf[idx] = (T)v;
  if (wd && c < C) wd[(((size_t)c * R + (R - 1 - r)) * S + (S - 1 - s)) * K + k] = (T)v;
}

extern "C" int unit_weight_prep(const float* w_krsc, const float* scale_k, int K, int R, int S, int C, int Cp,
                                void* w_fwd, void* w_dgrad, int dtype, void* stream) {
  UNIT_CHECK_ARG(Cp >= C, "weight_prep: Cp < C");
  long total = (long)K * R * S * Cp;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UNIT_BF16)
    weight_prep_kernel<bf16_t><<<cdiv(total, 256), 256, 0, st>>>(w_krsc, scale_k, K, R, S, C, Cp, (bf16_t*)w_fwd, (bf16_t*)w_dgrad);
  else
    weight_prep_kernel<float><<<cdiv(total, 256), 256, 0, st>>>(w_krsc, scale_k, K, R, S, C, Cp, (float*)w_fwd, (float*)w_dgrad);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// FrozenBN -> (scale, shift):  detectron2 FrozenBatchNorm2d eps=1e-5
__global__ void frozen_bn_fold_kernel(const float* w, const float* b, const float* rm, const float* rv, float eps,
                                      float* scale, float* shift, int C) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C) return;
  float s = w[i] * (1.0f / sqrtf(rv[i] + eps));
  scale[i] = s;
  shift[i] = b[i] - rm[i] * s;
}
extern "C" int unit_frozen_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float eps,
                                   float* scale, float* shift, int C, void* stream) {
  frozen_bn_fold_kernel<<<cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(w, b, rm, rv, eps, scale, shift, C);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// bias grad: db[k] (+)= sum_m dy[m][k]   (dy [M][ld], columns [0,K))
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void bias_grad_kernel(const T* __restrict__ dy, int M, int K, int ld, float* __restrict__ db, int rows_per_block) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  int m0 = blockIdx.y * rows_per_block;
  if (k >= K) return;
  int m1 = min(M, m0 + rows_per_block);
  float acc = 0.f;
  for (int m = m0; m < m1; ++m) acc += (float)dy[(size_t)m * ld + k];
  atomicAdd(db + k, acc);
}
// bf16, ld % 8 == 0: 16-byte loads (8 columns per thread), 8 row lanes per workgroup reduced through LDS
__global__ void __launch_bounds__(256) bias_grad_vec_kernel(const bf16_t* __restrict__ dy, int M, int K, int ld, float* __restrict__ db,
                                                            int rows_per_block) {
  __shared__ float red[8][257];
  int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  int c0 = (blockIdx.x * 32 + cg) * 8;
  int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < K) {
    for (int m = m0 + rl; m < m1; m += 8) {
      bf16x8 v = *reinterpret_cast<const bf16x8*>(dy + (size_t)m * ld + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][cg * 8 + j] = acc[j];
  __syncthreads();
  int c = blockIdx.x * 256 + threadIdx.x;
  if (c < K) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r][threadIdx.x];
    atomicAdd(db + c, s);
  }
}
// Tall inputs (the RPN conv bias: 9 576 rows x 1024 columns had FOUR workgroups walking all rows, 75-190 us): row blocks of
// BG_ROWS rows write their column sums to scratch [row block][K] (bias_grad_vec_kernel with a partial pointer), then one thread
// per column adds the row blocks in index order: a fixed summation order -> still bit-reproducible, 76 + 4 workgroups, ~10 us.
#define BG_ROWS 512
__global__ void __launch_bounds__(256) bias_grad_partial_kernel(const bf16_t* __restrict__ dy, int M, int K, int ld, float* __restrict__ part) {
  __shared__ float red[8][257];
  int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  int c0 = (blockIdx.x * 32 + cg) * 8;
  int m0 = blockIdx.y * BG_ROWS, m1 = min(M, m0 + BG_ROWS);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < K) {
    for (int m = m0 + rl; m < m1; m += 8) {
      bf16x8 v = *reinterpret_cast<const bf16x8*>(dy + (size_t)m * ld + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][cg * 8 + j] = acc[j];
  __syncthreads();
  int c = blockIdx.x * 256 + threadIdx.x;
  if (c < K) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r][threadIdx.x];
    part[(size_t)blockIdx.y * K + c] = s;
  }
}
__global__ void bias_grad_combine_kernel(const float* __restrict__ part, int nblk, int K, float* __restrict__ db, int accumulate) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= K) return;
  float s = accumulate ? db[c] : 0.f;
  for (int b = 0; b < nblk; ++b) s += part[(size_t)b * K + c];
  db[c] = s;
}
extern "C" size_t unit_bias_grad_scratch_bytes(int M, int K) { return (size_t)cdiv(M, BG_ROWS) * (size_t)K * sizeof(float); }
extern "C" int unit_bias_grad(const void* dy, int dtype, int M, int K, int ld, float* db, int accumulate, float* scratch,
                              size_t scratch_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UNIT_BF16 && ld % 8 == 0 && ((uintptr_t)dy % 16 == 0) && M > 2 * BG_ROWS && scratch &&
      scratch_bytes >= unit_bias_grad_scratch_bytes(M, K)) {
    int nblk = cdiv(M, BG_ROWS);
    bias_grad_partial_kernel<<<dim3(cdiv(K, 256), nblk), 256, 0, st>>>((const bf16_t*)dy, M, K, ld, scratch);
    UNIT_LAUNCH_CHECK();
    bias_grad_combine_kernel<<<cdiv(K, 256), 256, 0, st>>>(scratch, nblk, K, db, accumulate);
    UNIT_LAUNCH_CHECK();
    return UNIT_OK;
  }
  if (!accumulate) hipMemsetAsync(db, 0, sizeof(float) * K, st);
  if (M == 0) return UNIT_OK;
  if (dtype == UNIT_BF16 && ld % 8 == 0 && ((uintptr_t)dy % 16 == 0)) {
    // up to 16 384 rows: one workgroup per 256 columns walks all rows, so each db[c] has a single adder and the sum is
    // bit-reproducible; beyond that (and without scratch) the row blocks add atomically
    int rows = M <= 16384 ? M : 128;
    bias_grad_vec_kernel<<<dim3(cdiv(K, 256), cdiv(M, rows)), 256, 0, st>>>((const bf16_t*)dy, M, K, ld, db, rows);
    UNIT_LAUNCH_CHECK();
    return UNIT_OK;
  }
  int rpb = 256;
  dim3 grid(cdiv(K, 64), cdiv(M, rpb));
  if (dtype == UNIT_BF16) bias_grad_kernel<bf16_t><<<grid, 64, 0, st>>>((const bf16_t*)dy, M, K, ld, db, rpb);
  else bias_grad_kernel<float><<<grid, 64, 0, st>>>((const float*)dy, M, K, ld, db, rpb);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// K18  SGD momentum on flat fp32 buffers (torch.optim.SGD semantics: solver/build.py:110-112):
//   g = grad + wd*p ; buf = momentum*buf + g ; p -= lr*buf        (first step: buf = g, via buf init 0 + momentum*0)
// ---------------------------------------------------------------------------------------------------
// lr_dev (optional): the step's learning rate lives in device memory and `lr` is only the parameter group's multiplier -- a captured
// hipGraph of the step then follows the warm-up / multi-step schedule without being re-captured (the host writes one float per step)
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n,
                           float lr, float momentum, float wd, float grad_scale, int first, const float* __restrict__ lr_dev) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  if (lr_dev) lr = lr * *lr_dev;
  if (i + 4 <= n) {
    f32x4 pv = *reinterpret_cast<f32x4*>(p + i), gv = *reinterpret_cast<const f32x4*>(g + i);
    f32x4 bv = *reinterpret_cast<f32x4*>(buf + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float d = gv[j] * grad_scale + wd * pv[j];
      float b = first ? d : momentum * bv[j] + d;
      bv[j] = b; pv[j] = pv[j] - lr * b;
    }
    *reinterpret_cast<f32x4*>(p + i) = pv; *reinterpret_cast<f32x4*>(buf + i) = bv;
  } else {
    for (long j = i; j < n; ++j) {
      float d = g[j] * grad_scale + wd * p[j];
      float b = first ? d : momentum * buf[j] + d;
      buf[j] = b; p[j] = p[j] - lr * b;
    }
  }
}
__global__ void sgd_scalar_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n,
                                  float lr, float momentum, float wd, float grad_scale, int first, const float* __restrict__ lr_dev) {
  long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  if (lr_dev) lr = lr * *lr_dev;
  float d = g[j] * grad_scale + wd * p[j];
  float b = first ? d : momentum * buf[j] + d;
  buf[j] = b; p[j] = p[j] - lr * b;
}
extern "C" int unit_sgd_momentum(float* p, const float* g, float* buf, long n, float lr, float momentum, float wd,
                                 float grad_scale, int first_step, const float* lr_dev, void* stream) {
  if (n == 0) return UNIT_OK;
  UNIT_CHECK_ARG(((uintptr_t)p % 4 == 0) && ((uintptr_t)g % 4 == 0) && ((uintptr_t)buf % 4 == 0), "sgd: 4B alignment");
  if (((uintptr_t)p % 16) || ((uintptr_t)g % 16) || ((uintptr_t)buf % 16)) {
    // a hyper-parameter segment that starts inside a packed fused head (flat.py segments()): small, one element per thread
    sgd_scalar_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(p, g, buf, n, lr, momentum, wd, grad_scale, first_step, lr_dev);
    UNIT_LAUNCH_CHECK();
    return UNIT_OK;
  }
  sgd_kernel<<<cdiv(cdiv(n, 4), 256), 256, 0, (hipStream_t)stream>>>(p, g, buf, n, lr, momentum, wd, grad_scale, first_step, lr_dev);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// small helpers: y = cast(a32 [+ b]) ; used to merge the fp32 RoIAlign-backward accumulator with the RPN dgrad
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void add_cast_kernel(const float* __restrict__ a32, const T* __restrict__ b, const T* __restrict__ mask, T* __restrict__ y, long n8) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float a[8], bb[8];
  Vec8<float>::load(a32 + i * 8, a);
  if (b) {
    Vec8<T>::load(b + i * 8, bb);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += bb[j];
  }
  if (mask) {
    Vec8<T>::load(mask + i * 8, bb);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = bb[j] > 0.f ? a[j] : 0.f;
  }
  Vec8<T>::store(y + i * 8, a);
}
// y = cast((a32 [+ b]) [* (mask_ref > 0)])
extern "C" int unit_add_cast(const float* a32, const void* b, const void* mask_ref, void* y, int dtype, long n, void* stream) {
  UNIT_CHECK_ARG(n % 8 == 0, "add_cast: n % 8 != 0");
  if (n == 0) return UNIT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UNIT_BF16) add_cast_kernel<bf16_t><<<cdiv(n / 8, 256), 256, 0, st>>>(a32, (const bf16_t*)b, (const bf16_t*)mask_ref, (bf16_t*)y, n / 8);
  else add_cast_kernel<float><<<cdiv(n / 8, 256), 256, 0, st>>>(a32, (const float*)b, (const float*)mask_ref, (float*)y, n / 8);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// unit_fill_zero: the step's own memset (loss slots, the full-resolution target of a strided dgrad scatter, bitmaps): 16 bytes per lane,
// byte tail by the last lanes. hipMemsetAsync would launch the runtime's fillBuffer kernel -- a stock op the hot path does not use.
__global__ void __launch_bounds__(256) fill_zero_kernel(unsigned char* __restrict__ p, size_t nbytes) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (i + 16 <= nbytes) *reinterpret_cast<u32x4*>(p + i) = u32x4{0u, 0u, 0u, 0u};
  else for (; i < nbytes; ++i) p[i] = 0;
}
extern "C" int unit_fill_zero(void* p, size_t nbytes, void* stream) {
  if (nbytes == 0) return UNIT_OK;
  UNIT_CHECK_ARG(((uintptr_t)p & 15) == 0, "fill_zero: 16-byte aligned pointer");
  fill_zero_kernel<<<cdiv(cdiv(nbytes, 16), 256), 256, 0, (hipStream_t)stream>>>((unsigned char*)p, nbytes);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// unit_debug_spin: ONE workgroup of one wave that spins for `cycles` shader clocks (s_memtime). Diagnostic: two of these on two HIP streams take
// one spin time when the streams sit on different hardware queues and two when they share one (tools/queue_probe.py; the runtime maps the
// process's streams onto GPU_MAX_HW_QUEUES = 4 queues, and which of the step's streams share a queue decides 10 % of its time).
__global__ void debug_spin_kernel(long long cycles, int* sink) {
  long long t0 = (long long)__builtin_amdgcn_s_memtime();
  int n = 0;
  while ((long long)__builtin_amdgcn_s_memtime() - t0 < cycles) ++n;
  if (n == -1) *sink = n;
}
extern "C" int unit_debug_spin(long long cycles, void* sink, void* stream) {
  debug_spin_kernel<<<1, 64, 0, (hipStream_t)stream>>>(cycles, (int*)sink);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// y[i] = cast(x[i])  between fp32 and bf16 (either direction)
template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ x, TO* __restrict__ y, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (TO)(float)x[i];
}
extern "C" int unit_cast(const void* x, int in_dtype, void* y, int out_dtype, long n, void* stream) {
  if (n == 0) return UNIT_OK;
  hipStream_t st = (hipStream_t)stream;
  int g = cdiv(n, 256);
  if (in_dtype == UNIT_F32 && out_dtype == UNIT_BF16) cast_kernel<float, bf16_t><<<g, 256, 0, st>>>((const float*)x, (bf16_t*)y, n);
  else if (in_dtype == UNIT_BF16 && out_dtype == UNIT_F32) cast_kernel<bf16_t, float><<<g, 256, 0, st>>>((const bf16_t*)x, (float*)y, n);
  else if (in_dtype == UNIT_F32 && out_dtype == UNIT_F32) cast_kernel<float, float><<<g, 256, 0, st>>>((const float*)x, (float*)y, n);
  else cast_kernel<bf16_t, bf16_t><<<g, 256, 0, st>>>((const bf16_t*)x, (bf16_t*)y, n);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// NCHW fp32 <-> NHWC (T) layout conversion for the plugin boundary (reference tensors are NCHW fp32).
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int C, int H, int W, int Cp) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)N * H * W * Cp;
  if (idx >= total) return;
  int c = idx % Cp; long p = idx / Cp;
  int w = p % W; p /= W; int h = p % H; int n = p / H;
  y[idx] = (T)(c < C ? x[(((size_t)n * C + c) * H + h) * W + w] : 0.f);
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W, int Cp) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)N * C * H * W;
  if (idx >= total) return;
  int w = idx % W; long p = idx / W;
  int h = p % H; p /= H; int c = p % C; int n = p / C;
  y[idx] = (float)x[(((size_t)n * H + h) * W + w) * Cp + c];
}
extern "C" int unit_nchw_to_nhwc(const float* x, void* y, int dtype, int N, int C, int H, int W, int Cp, void* stream) {
  long total = (long)N * H * W * Cp;
  if (total == 0) return UNIT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UNIT_BF16) nchw_to_nhwc_kernel<bf16_t><<<cdiv(total, 256), 256, 0, st>>>(x, (bf16_t*)y, N, C, H, W, Cp);
  else nchw_to_nhwc_kernel<float><<<cdiv(total, 256), 256, 0, st>>>(x, (float*)y, N, C, H, W, Cp);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
extern "C" int unit_nhwc_to_nchw(const void* x, int dtype, float* y, int N, int C, int H, int W, int Cp, void* stream) {
  long total = (long)N * C * H * W;
  if (total == 0) return UNIT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UNIT_BF16) nhwc_to_nchw_kernel<bf16_t><<<cdiv(total, 256), 256, 0, st>>>((const bf16_t*)x, y, N, C, H, W, Cp);
  else nhwc_to_nchw_kernel<float><<<cdiv(total, 256), 256, 0, st>>>((const float*)x, y, N, C, H, W, Cp);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// Sampling permutations without torch.randperm (d2 `subsample_labels` draws torch.randperm on the device: rocprim sorts + host
// bookkeeping, and not capturable in a hipGraph). keys[b][i] = a counter-based hash of (seed, *counter, stream, b, i) laid out as
// a POSITIVE FINITE float (31 random bits: positive floats order like their bit patterns); a stable descending sort of the keys
// (unit_sort_desc_stable) is then a uniformly random permutation up to ~n^2 / 2^32 tied pairs. The step counter lives on the
// device and is bumped by its own tiny launch, so that a captured graph draws fresh permutations on every replay.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {     // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void perm_keys_kernel(unsigned long long seed, const long long* __restrict__ counter, int stream_id, int n, float* __restrict__ keys) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int b = blockIdx.y;
  if (i >= n) return;
  unsigned long long c = counter ? (unsigned long long)*counter : 0ull;
  unsigned long long h = mix64(mix64(seed ^ (c * 0xD1342543DE82EF95ull)) ^ (((unsigned long long)(unsigned)stream_id << 40) | ((unsigned long long)(unsigned)b << 32) | (unsigned)i));
  unsigned bits = (unsigned)(h >> 33);                          // 31 bits
  if ((bits & 0x7F800000u) == 0x7F800000u) bits &= 0x7F7FFFFFu; // never inf / nan
  keys[(size_t)b * n + i] = __uint_as_float(bits);
}
extern "C" int unit_perm_keys(unsigned long long seed, const long long* counter_dev, int stream_id, int B, int n, float* keys, void* stream) {
  if (B == 0 || n == 0) return UNIT_OK;
  perm_keys_kernel<<<dim3(cdiv(n, 256), B), 256, 0, (hipStream_t)stream>>>(seed, counter_dev, stream_id, n, keys);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
__global__ void counter_bump_kernel(long long* c, long long d) { *c += d; }
extern "C" int unit_counter_bump(long long* counter_dev, long long delta, void* stream) {
  counter_bump_kernel<<<1, 1, 0, (hipStream_t)stream>>>(counter_dev, delta);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
