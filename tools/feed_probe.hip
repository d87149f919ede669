// feed_probe.hip -- how fast can one CU stage L2-resident operand bytes into LDS?  (decides whether the feed-bound 4-wave conv
// kernels should stage one operand through registers instead of LDS-DMA)
//   build (here):  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o unit_amd/_build/feed_probe tools/feed_probe.hip
//   run (GPU box): unit_amd/_build/feed_probe
// One 512-thread workgroup per CU (128 KB of LDS), every iteration stages a 64 KB tile (512 rows x 128 B, rows 2 KB apart in a
// pool that stays L2-resident) with two tiles in flight:
//   mode 0: LDS-DMA (buffer_load_dwordx4 ... lds), 8 pieces per thread and tile
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2: half of each tile by LDS-DMA, half through registers
//   mode 3: as 1 without the ds_write (global_load only: the L2 -> register path alone)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((address_space(3))) void lds_void;

template <int MODE>
__global__ void __launch_bounds__(512) feed_kernel(const char* __restrict__ pool, size_t pool_bytes, int iters, int* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pool), 0, (int)pool_bytes, 0x00020000);
  // tile t of this workgroup: rows (blockIdx*7 + t*13 + r) mod nrows, r = 0..511, row pitch 2048 B, 128 B taken per row
  const int nrows = (int)(pool_bytes / 2048);
  i32x4 acc = {0, 0, 0, 0};
  i32x4 ra[8], rb[8];
  auto src_off = [&](int t, int i) -> unsigned {
    int piece = i * 8 + wid;                       // 64 pieces of 8 rows
    int row = (blockIdx.x * 7 + t * 13 + piece * 8 + (lane >> 3)) % nrows;
    return (unsigned)row * 2048u + (unsigned)(lane & 7) * 16u;
  };
  auto issue = [&](int t, int buf, i32x4 (&regs)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      bool dma = MODE == 0 || (MODE == 2 && i < 4);
      unsigned off = src_off(t, i);
      if (dma) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + buf * 65536 + (i * 8 + wid) * 1024), 16, off, 0, 0, 0);
      else regs[i] = *reinterpret_cast<const i32x4*>(pool + off);
    }
  };
  auto land = [&](int buf, i32x4 (&regs)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      bool dma = MODE == 0 || (MODE == 2 && i < 4);
      if (!dma) {
        if (MODE == 3) { acc[0] ^= regs[i][0]; acc[1] ^= regs[i][3]; }
        else *reinterpret_cast<i32x4*>(smem + buf * 65536 + (i * 8 + wid) * 1024 + lane * 16) = regs[i];
      }
    }
  };
  issue(0, 0, ra);
  for (int t = 0; t < iters; t += 2) {
    issue(t + 1, 1, rb);
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    land(0, ra);                                   // register loads: the compiler's own counted vmcnt
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    acc[2] ^= *reinterpret_cast<const int*>(smem + ((tid * 67 + t) & 16383) * 4);                // keep the tile live
    __builtin_amdgcn_s_barrier();
    issue(t + 2, 0, ra);
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    land(1, rb);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    acc[3] ^= *reinterpret_cast<const int*>(smem + 65536 + ((tid * 67 + t) & 16383) * 4);
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  land(0, ra);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) sink[0] = 1;
}

template <int MODE>
static void run(const char* name, const char* pool, size_t pool_bytes, int* sink, int nblocks) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)feed_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  feed_kernel<MODE><<<nblocks, 512, 131072>>>(pool, pool_bytes, 200, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  feed_kernel<MODE><<<nblocks, 512, 131072>>>(pool, pool_bytes, iters, sink);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  double bytes = (double)nblocks * (iters + 1) * 65536.0;
  printf("%-44s %8.3f ms  %7.2f TB/s chip  %6.1f GB/s per CU  (%5.1f B/clk/CU at 2.4 GHz)\n", name, ms, bytes / ms / 1e9, bytes / ms / 1e6 / nblocks,
         bytes / ms / 1e6 / nblocks / 2.4);
}

int main() {
  size_t pool_bytes = 16u << 20;                   // 16 MB: 2 MB per XCD's worth of rows, L2-resident after the first pass
  char* pool; int* sink;
  hipMalloc(&pool, pool_bytes); hipMalloc(&sink, 64);
  hipMemset(pool, 1, pool_bytes);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  int ncu = pr.multiProcessorCount;
  printf("device %s, %d CUs\n", pr.name, ncu);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("mode 0: LDS-DMA", pool, pool_bytes, sink, ncu);
    run<1>("mode 1: global_load_dwordx4 + ds_write_b128", pool, pool_bytes, sink, ncu);
    run<2>("mode 2: half LDS-DMA, half registers", pool, pool_bytes, sink, ncu);
    run<3>("mode 3: global_load_dwordx4 only", pool, pool_bytes, sink, ncu);
  }
  return 0;
}
