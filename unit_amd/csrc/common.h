// common.h -- shared helpers for the libunit_hip.so kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define UNIT_OK 0
#define UNIT_ERR_ARG -1
#define UNIT_ERR_LAUNCH -2
#define UNIT_ERR_WORKSPACE -3
#define UNIT_ERR_UNSUPPORTED -4

#define UNIT_F32 0
#define UNIT_BF16 1

extern "C" void unit_set_error(const char* msg);

#define UNIT_CHECK_ARG(cond, msg)                                   \
  do {                                                              \
    if (!(cond)) {                                                  \
      unit_set_error(msg);                                          \
      return UNIT_ERR_ARG;                                          \
    }                                                               \
  } while (0)

#define UNIT_LAUNCH_CHECK()                                         \
  do {                                                              \
    hipError_t e__ = hipGetLastError();                             \
    if (e__ != hipSuccess) {                                        \
      unit_set_error(hipGetErrorString(e__));                       \
      return UNIT_ERR_LAUNCH;                                       \
    }                                                               \
  } while (0)

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// 8-element vector of activations in registers as floats, loaded/stored as 16 B (bf16) or 32 B (f32).
template <typename T> struct Vec8;
template <> struct Vec8<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4*>(p) = a; *reinterpret_cast<f32x4*>(p + 4) = b;
  }
};
template <> struct Vec8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = a;
  }
};

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
