// feed_matrix.hip -- what sets the LDS-DMA operand feed of a CU inside a conv / weight-gradient loop?
// The kernels stage operands with `buffer_load_dwordx4 ... lds` pieces (1 KB per wave instruction). Measured in round 3:
//   pure staging loop (tools/feed_probe.hip)                         41-43 B/clk/CU
//   conv loop (conv_igemm256p8: 8 rows x 128 B per piece)            ~27 B/clk/CU
//   weight-gradient loop (conv_wgrad256p8: 4 rows x 256 B per piece) ~17.5 B/clk/CU
// This probe is ONE loop with the structure of those kernels -- 512 threads, 128 KB of LDS as 8 slots of 16 KB, a "phase" = {2 pieces per
// wave staged into the next slot ; barrier ; NM MFMAs (16x16x32 bf16) with NR LDS fragment reads issued between them ; barrier}, a counted
// vmcnt every 4 phases, optionally the two wave groups half a phase apart -- and every ingredient is a knob:
//   RPP    rows per piece: 8 (8 x 128 B, conv k-slices), 4 (4 x 256 B, weight gradient), 2, 1 (one 1 KB row)
//   TRAV   0: k-major -- the SAME rows at the next 128 / 256 B column every phase (1x1 conv walking its contraction);
//          1: m-major -- the same columns of the NEXT rows every phase (weight gradient walking the pixels)
//   pitch  bytes between source rows
//   pool   bytes of source the workgroups of the chip share (2 MB: every XCD's L2 holds it; 64 MB: Infinity Cache; 1 GB: HBM)
//   NM     MFMAs per phase and wave (0 or 16), NR / RK: LDS reads per phase and wave, RK 0 = ds_read_b128, 1 = ds_read_b64_tr_b16
//   DEPTH  pieces per wave that may stay in flight across the wait (vmcnt immediate)
//   STAG   waves 4-7 one barrier behind waves 0-3
//   NV     VALU instructions (dependent v_fma chains, 8 independent chains) + one fp32 LDS round trip (4 ds_write_b128 + 2 ds_read_b128) that the
//          STAGING section carries besides its two pieces: the epilogue sub-step an X-stationary kernel would put there
// Output: bytes staged per CU and cycle (in-kernel s_memtime of wave 0, median over workgroups) and the chip-wide rate by wall clock.
//   build:  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o unit_amd/_build/feed_matrix tools/feed_matrix.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((address_space(3))) void lds_void;

struct Args {
  const char* pool; unsigned pool_bytes; unsigned pitch; int iters; int depth; unsigned long long* cycles; int* sink;
};

template <int RPP, int TRAV, int NM, int NR, int RK, int STAG, int NV = 0>
__global__ void __launch_bounds__(512, 2) feed_kernel(Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.pool), 0, (int)p.pool_bytes, 0x00020000);
  constexpr int SEG = 1024 / RPP;                    // bytes a piece takes from one source row
  constexpr int LPR = SEG / 16;                      // lanes per row
  const unsigned nrows = p.pool_bytes / p.pitch;
  const unsigned cols = p.pitch / SEG;               // column positions of a row
  // this wave's two pieces of a 16 KB slot: pieces wid and 8 + wid ; lane -> row (lane / LPR), 16-B chunk lane % LPR
  const unsigned lrow = lane / LPR, lch = lane % LPR;
  // a workgroup starts at its own rows (different workgroups read different rows, as different output tiles do)
  unsigned row0 = (blockIdx.x * 1031u) % nrows, col = 0;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  bf16x8 fa, fb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)(lane + i); fb[i] = (__bf16)(float)(wid - i); }
  i32x4 rd[NR > 0 ? NR : 1];
  float vsink = 0.f;
  auto stage = [&](int slot) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned piece = j * 8 + wid;                                    // 16 pieces per slot
      unsigned row = (row0 + piece * RPP + lrow) % nrows;
      unsigned off = row * p.pitch + col * SEG + lch * 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + slot * 16384 + piece * 1024), 16, off, 0, 0, 0);
    }
    if (TRAV == 0) { if (++col == cols) { col = 0; row0 = (row0 + 16 * RPP) % nrows; } }
    else row0 = (row0 + 16 * RPP) % nrows;
    if (NV > 0) {          // epilogue sub-step stand-in: scratch round trip + NV VALU on eight independent chains
      char* scr = smem + 8 * 16384 - 8 * 4096 + wid * 4096;          // top of slot 7 (this probe never validates data)
      f32x4 v = acc[0];
#pragma unroll
      for (int a = 0; a < 4; ++a) *reinterpret_cast<f32x4*>(scr + (lane & 15) * 272 + (a * 16 + (lane >> 4) * 4) * 4) = v;
      f32x4 r0 = *reinterpret_cast<const f32x4*>(scr + (lane >> 3) * 272 + (lane & 7) * 32);
      f32x4 r1 = *reinterpret_cast<const f32x4*>(scr + (lane >> 3) * 272 + (lane & 7) * 32 + 16);
      float e[8] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
#pragma unroll
      for (int i = 0; i < NV; ++i) e[i & 7] = __builtin_fmaf(e[i & 7], 1.0009765625f, 0.5f);
      vsink += e[0] + e[1] + e[2] + e[3] + e[4] + e[5] + e[6] + e[7];
    }
  };
  auto wait_depth = [&]() {
    switch (p.depth) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
  };
  auto compute = [&](int slot) {
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(smem + slot * 16384);
    // NR fragment reads between NM MFMAs (issued by asm so that nothing is folded away; conflict-free addresses)
    constexpr int PER = NM > 0 ? (NR + NM - 1) / NM : NR;
    int r = 0;
#pragma unroll
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) {
      if (NM > 0) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i & 3], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        if (r < NR) {
          unsigned a = base + ((r * 1024 + lane * 16) & 16383);
          if (RK == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(rd[r]) : "v"(a));
          else { i32x2 t; asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(t) : "v"(base + ((r * 512 + lane * 8) & 16383))); rd[r][0] = t[0]; rd[r][1] = t[1]; }
          ++r;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < NR; ++k) asm volatile("" :: "v"(rd[k]));
  };
  // prologue: 4 slots in flight
  for (int s = 0; s < 4; ++s) stage(s);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (STAG && grp == 1) __builtin_amdgcn_s_barrier();
  for (int it = 0; it < p.iters; ++it) {
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      stage((it * 4 + ph + 4) & 7);
      if (ph == 3) wait_depth();
      __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
      compute((it * 4 + ph) & 7);
      __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (STAG && grp == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) p.cycles[blockIdx.x] = t1 - t0;
  float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + vsink;
  if (s == 12345.678f) p.sink[0] = 1;
}

struct Pool { char* p; size_t bytes; };

template <int RPP, int TRAV, int NM, int NR, int RK, int STAG, int NV = 0>
static void run(const char* name, Pool pool, unsigned pitch, int depth, int ncu, unsigned long long* cyc_dev, int* sink) {
  const int iters = 400;
  auto k = feed_kernel<RPP, TRAV, NM, NR, RK, STAG, NV>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  Args a{pool.p, (unsigned)pool.bytes, pitch, 50, depth, cyc_dev, sink};
  k<<<ncu, 512, 131072>>>(a);
  hipDeviceSynchronize();
  a.iters = iters;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<<<ncu, 512, 131072>>>(a);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> cyc(ncu);
  hipMemcpy(cyc.data(), cyc_dev, ncu * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(cyc.begin(), cyc.end());
  double bytes = (double)iters * 4 * 16384.0;
  printf("%-26s rows/piece %d %s pitch %5u pool %5zu MB depth %d | MFMA/phase %2d reads/phase %2d %-4s %s VALU+LDS in staging %3d | %6.1f B/clk/CU  (cycles/phase %6.0f)  chip %6.2f TB/s\n",
         name, RPP, TRAV ? "m-major" : "k-major", pitch, pool.bytes >> 20, depth, NM, NR, NR ? (RK ? "tr64" : "b128") : "-", STAG ? "stag" : "lock", NV,
         bytes / (double)cyc[ncu / 2], (double)cyc[ncu / 2] / (iters * 4.0), bytes * ncu / ms / 1e9);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  int ncu = pr.multiProcessorCount;
  printf("device %s, %d CUs\n", pr.name, ncu);
  unsigned long long* cyc; int* sink;
  hipMalloc(&cyc, ncu * 8); hipMalloc(&sink, 64);
  Pool small{nullptr, 2u << 20}, mall{nullptr, 64u << 20}, big{nullptr, 1u << 30};
  hipMalloc(&small.p, small.bytes); hipMalloc(&mall.p, mall.bytes); hipMalloc(&big.p, big.bytes);
  hipMemset(small.p, 1, small.bytes); hipMemset(mall.p, 1, mall.bytes); hipMemset(big.p, 1, big.bytes);
  for (Pool pool : {small, mall, big}) {
    printf("---- pool %zu MB\n", pool.bytes >> 20);
    // (A) staging alone, by piece shape, traversal, pitch, depth
    run<8, 0, 0, 0, 0, 0>("A feed only", pool, 1024, 8, ncu, cyc, sink);
    run<8, 0, 0, 0, 0, 0>("A feed only", pool, 4096, 8, ncu, cyc, sink);
    run<8, 0, 0, 0, 0, 0>("A feed only", pool, 1024, 4, ncu, cyc, sink);
    run<8, 0, 0, 0, 0, 0>("A feed only", pool, 1024, 0, ncu, cyc, sink);
    run<4, 1, 0, 0, 0, 0>("A feed only", pool, 1024, 8, ncu, cyc, sink);
    run<4, 1, 0, 0, 0, 0>("A feed only", pool, 4096, 8, ncu, cyc, sink);
    run<4, 1, 0, 0, 0, 0>("A feed only", pool, 1024, 4, ncu, cyc, sink);
    run<2, 1, 0, 0, 0, 0>("A feed only", pool, 1024, 8, ncu, cyc, sink);
    run<1, 1, 0, 0, 0, 0>("A feed only", pool, 1024, 8, ncu, cyc, sink);
    run<1, 1, 0, 0, 0, 0>("A feed only", pool, 4096, 8, ncu, cyc, sink);
    // (B) + 16 MFMAs per phase
    run<8, 0, 16, 0, 0, 0>("B + MFMA", pool, 1024, 4, ncu, cyc, sink);
    run<8, 0, 16, 0, 0, 1>("B + MFMA", pool, 1024, 4, ncu, cyc, sink);
    run<4, 1, 16, 0, 0, 0>("B + MFMA", pool, 1024, 4, ncu, cyc, sink);
    run<4, 1, 16, 0, 0, 1>("B + MFMA", pool, 1024, 4, ncu, cyc, sink);
    // (C) + fragment reads (conv: 6 ds_read_b128 per phase and wave; weight gradient: 12 ds_read_b64_tr_b16)
    run<8, 0, 0, 6, 0, 0>("C + reads (no MFMA)", pool, 1024, 4, ncu, cyc, sink);
    run<4, 1, 0, 12, 1, 0>("C + reads (no MFMA)", pool, 1024, 4, ncu, cyc, sink);
    run<8, 0, 16, 6, 0, 0>("C conv-like", pool, 1024, 4, ncu, cyc, sink);
    run<8, 0, 16, 6, 0, 1>("C conv-like", pool, 1024, 4, ncu, cyc, sink);
    run<8, 0, 16, 6, 0, 1>("C conv-like", pool, 4096, 4, ncu, cyc, sink);
    run<4, 1, 16, 12, 1, 0>("C wgrad-like", pool, 1024, 4, ncu, cyc, sink);
    run<4, 1, 16, 12, 1, 1>("C wgrad-like", pool, 1024, 4, ncu, cyc, sink);
    run<4, 1, 16, 12, 1, 1>("C wgrad-like", pool, 4096, 4, ncu, cyc, sink);
    run<8, 1, 16, 12, 1, 1>("C wgrad, 8-row pieces", pool, 1024, 4, ncu, cyc, sink);
    run<4, 1, 16, 12, 0, 1>("C wgrad, b128 reads", pool, 1024, 4, ncu, cyc, sink);
    run<4, 1, 16, 6, 1, 1>("C wgrad, half reads", pool, 1024, 4, ncu, cyc, sink);
    run<4, 0, 16, 12, 1, 1>("C wgrad, k-major", pool, 1024, 4, ncu, cyc, sink);
    // (D) the X-stationary question: a conv-like staggered phase whose staging section also carries an epilogue sub-step
    run<8, 0, 16, 8, 0, 1, 1>("D conv + epilogue step", pool, 1024, 4, ncu, cyc, sink);
    run<8, 0, 16, 8, 0, 1, 25>("D conv + epilogue step", pool, 1024, 4, ncu, cyc, sink);
    run<8, 0, 16, 8, 0, 1, 50>("D conv + epilogue step", pool, 1024, 4, ncu, cyc, sink);
    run<8, 0, 16, 8, 0, 1, 100>("D conv + epilogue step", pool, 1024, 4, ncu, cyc, sink);
  }
  return 0;
}
