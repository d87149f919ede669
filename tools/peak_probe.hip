// peak_probe.hip -- the two rooflines of this box, measured (SURVEY section 8d asks for measured peaks next to the vendor figures):
//   (1) dense bf16 MFMA rate: every SIMD of every CU issues independent v_mfma_f32_16x16x32_bf16 (and 32x32x16) back to back from registers;
//   (2) HBM streaming: read-only, write-only and copy over buffers far larger than the 256 MB Infinity Cache, 16 B per lane.
//   build + run on the GPU box:  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/peak_probe tools/peak_probe.hip && /tmp/peak_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int SHAPE>   // 0: 16x16x32, 1: 32x32x16
__global__ void __launch_bounds__(256) mfma_kernel(int iters, float* sink) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
  if (SHAPE == 0) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int t = 0; t < iters; ++t) {
      // inline asm with VGPR operands: left to itself hipcc parks these accumulators in AGPRs and copies them in and out around
      // every MFMA of this loop (826 / 1290 TFLOP/s "measured" that way)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.678f) sink[0] = s;
  } else {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int t = 0; t < iters; ++t) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    if (s == 12345.678f) sink[0] = s;
  }
}

// MODE 0: read, 1: write, 2: copy ; every lane moves 16 B per step, grid-stride
template <int MODE>
__global__ void __launch_bounds__(256) stream_kernel(const i32x4* __restrict__ src, i32x4* __restrict__ dst, size_t n16, int* sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  i32x4 acc = {0, 0, 0, 0};
  for (; i < n16; i += stride) {
    if (MODE == 0) { i32x4 v = src[i]; acc[0] ^= v[0]; acc[1] ^= v[3]; }
    else if (MODE == 1) dst[i] = i32x4{(int)i, 1, 2, 3};
    else dst[i] = src[i];
  }
  if (MODE == 0 && acc[0] == 0x7fffffff && acc[1] == 0x12345) sink[0] = 1;
}

static float time_ms(hipEvent_t e0, hipEvent_t e1) { float ms; CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms; }

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device: %s, %d CUs, clockRate %.0f MHz, memoryClockRate %.0f MHz, bus %d bit\n", prop.gcnArchName, cus, prop.clockRate / 1e3, prop.memoryClockRate / 1e3,
         prop.memoryBusWidth);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float* sinkf; int* sinki; CHECK(hipMalloc(&sinkf, 64)); CHECK(hipMalloc(&sinki, 64));
  // ---- MFMA
  for (int shape = 0; shape < 2; ++shape)
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
      int iters = 20000, blocks = cus * waves_per_simd;            // 256 threads = 4 waves = one per SIMD
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        if (shape == 0) mfma_kernel<0><<<blocks, 256>>>(iters, sinkf); else mfma_kernel<1><<<blocks, 256>>>(iters, sinkf);
        CHECK(hipEventRecord(e1));
        float ms = time_ms(e0, e1);
        double flop = (double)blocks * 4 * iters * (shape == 0 ? 8 * 2.0 * 16 * 16 * 32 : 4 * 2.0 * 32 * 32 * 16);
        if (rep) printf("MFMA %s, %d wave(s) per SIMD: %.1f TFLOP/s (%.1f ms)\n", shape == 0 ? "16x16x32 bf16" : "32x32x16 bf16", waves_per_simd, flop / ms / 1e9, ms);
      }
    }
  // ---- HBM streams: 4 GiB buffers
  size_t bytes = (size_t)4 << 30, n16 = bytes / 16;
  i32x4 *a, *b; CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
  CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 2, bytes));
  for (int mode = 0; mode < 3; ++mode)
    for (int wg_per_cu = 4; wg_per_cu <= 16; wg_per_cu *= 2) {
      int blocks = cus * wg_per_cu;
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        if (mode == 0) stream_kernel<0><<<blocks, 256>>>(a, b, n16, sinki);
        else if (mode == 1) stream_kernel<1><<<blocks, 256>>>(a, b, n16, sinki);
        else stream_kernel<2><<<blocks, 256>>>(a, b, n16, sinki);
        CHECK(hipEventRecord(e1));
        float ms = time_ms(e0, e1);
        if (rep && ms < best) best = ms;
      }
      double moved = mode == 2 ? 2.0 * bytes : (double)bytes;
      printf("HBM %s, %2d workgroups per CU: %.2f TB/s (%.2f ms for %.1f GB moved)\n", mode == 0 ? "read " : mode == 1 ? "write" : "copy ", wg_per_cu, moved / best / 1e9, best,
             moved / 1e9);
    }
  return 0;
}
