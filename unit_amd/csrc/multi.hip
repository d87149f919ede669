// multi.hip -- multi-tensor end-of-step kernels: ONE launch over all trainable conv weights instead of one per layer.
//   unit_multi_wgrad_reduce : grads[off+i] = scale[k] * sum_s partial_s[i]      (ordered split-M slab reduction + FrozenBN fold)
//   unit_multi_weight_prep  : bf16/fp32 forward copy [K][R][S][C] (scale folded) + flipped/transposed dgrad copy [C][R][S][K]
// The per-layer launches they replace (~110 wgrad_reduce + ~110 weight_prep of 5-10 us each) were launch-bound.
#include "common.h"

struct UnitTensorDesc {
  const float* partial;
  const float* scale;
  void* wf;
  void* wd;
  long offset;
  int splits, K, R, S, C, block0;
};

#define MT_ELEMS_PER_BLOCK 1024

__device__ __forceinline__ int find_tensor(const UnitTensorDesc* __restrict__ d, int n, int b) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (d[mid].block0 <= b) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ void multi_reduce_kernel(const UnitTensorDesc* __restrict__ descs, int n, float* __restrict__ grads) {
  int t = find_tensor(descs, n, blockIdx.x);
  UnitTensorDesc d = descs[t];
  long KK = (long)d.K * d.R * d.S * d.C;
  long i = (long)(blockIdx.x - d.block0) * MT_ELEMS_PER_BLOCK + threadIdx.x * 4;
  if (i >= KK || d.partial == nullptr) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(d.partial + i);
  for (int sp = 1; sp < d.splits; ++sp) s += *reinterpret_cast<const f32x4*>(d.partial + (size_t)sp * KK + i);
  if (d.scale) s *= d.scale[i / ((long)d.R * d.S * d.C)];
  *reinterpret_cast<f32x4*>(grads + d.offset + i) = s;
}

template <typename T>
__global__ void multi_prep_kernel(const UnitTensorDesc* __restrict__ descs, int n, const float* __restrict__ params) {
  int t = find_tensor(descs, n, blockIdx.x);
  UnitTensorDesc d = descs[t];
  int RSC = d.R * d.S * d.C;
  long KK = (long)d.K * RSC;
  long i0 = (long)(blockIdx.x - d.block0) * MT_ELEMS_PER_BLOCK + threadIdx.x * 4;
  if (i0 >= KK) return;
  f32x4 p = *reinterpret_cast<const f32x4*>(params + d.offset + i0);
  int k = i0 / RSC; int rem = i0 - (long)k * RSC;          // the 4 elements share k, r, s (C % 4 == 0)
  int c = rem % d.C; int rs = rem / d.C; int s = rs % d.S; int r = rs / d.S;
  float sc = d.scale ? d.scale[k] : 1.0f;
  T* wf = (T*)d.wf; T* wd = (T*)d.wd;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = p[j] * sc;
    wf[i0 + j] = (T)v;
    if (wd) wd[(((size_t)(c + j) * d.R + (d.R - 1 - r)) * d.S + (d.S - 1 - s)) * d.K + k] = (T)v;
  }
}

extern "C" size_t unit_tensor_desc_bytes(void) { return sizeof(UnitTensorDesc); }

extern "C" int unit_multi_wgrad_reduce(const void* descs_dev, int n, int total_blocks, float* grads_flat, void* stream) {
  if (n == 0 || total_blocks == 0) return UNIT_OK;
  multi_reduce_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>((const UnitTensorDesc*)descs_dev, n, grads_flat);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

extern "C" int unit_multi_weight_prep(const void* descs_dev, int n, int total_blocks, const float* params_flat, int dtype, void* stream) {
  if (n == 0 || total_blocks == 0) return UNIT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UNIT_BF16) multi_prep_kernel<bf16_t><<<total_blocks, 256, 0, st>>>((const UnitTensorDesc*)descs_dev, n, params_flat);
  else multi_prep_kernel<float><<<total_blocks, 256, 0, st>>>((const UnitTensorDesc*)descs_dev, n, params_flat);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
