// conv_igemm128.hip -- mid-size LDS-DMA implicit-GEMM convolution (forward / dgrad) for the backbone layers
// (res2-res4 on 2-4 images: M = 9 576 .. 150 000 output pixels, 64..1024 channels), where a 256x256 tile leaves most of
// the 256 CUs without a workgroup and the register-staged conv_igemm.hip kernel is latency-bound at one wave per SIMD.
//
// Same math, operand layout, swizzle and epilogue as conv_igemm256.hip, scaled to 4 waves (2 x 2) per workgroup:
//   BM x BN x 64 tile, BM, BN in {128, 64}; each wave (BM/2) x (BN/2); operands HBM/L2 -> LDS by LDS-DMA
//   (`buffer_load_dwordx4 ... lds`, no staging VGPRs), NS LDS stages of (BM+BN)*128 B (2 x 32 KB at 128x128, 3 x 24 KB at
//   64x128 / 128x64), so that two workgroups share a CU (2 waves per SIMD: one wave's fragment reads hide under the
//   other's MFMAs) and the DMA of a k-tile has NS-1 iterations to land.
//   All 16 fragment reads of a k-tile are issued up front; the MFMAs start on counted lgkmcnt waits as they land.
// Requires bf16 operands and C % 64 == 0.
#include "common.h"
#include "conv_epilogue.h"

#ifndef UNIT_DBGMID
#define UNIT_DBGMID 0
#endif

#include "conv_igemm128.h"
#include "conv_pair.h"

__device__ __forceinline__ int swz128(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename TO> struct Out4;
template <> struct Out4<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) { f32x4 a = *reinterpret_cast<const f32x4*>(p); v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { f32x4 a = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(p) = a; }
};
template <> struct Out4<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
    bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)a[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
    bf16x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x4*>(p) = a;
  }
};

typedef __attribute__((address_space(3))) void lds_void_t;

// X3: bf16x3 operands (conv_epilogue.h SplitK) -- split x, three k segments per 64-channel block, split output planes.
template <typename TO, int BM, int BN, int NS, bool X3 = false, bool PAIR = false>
__global__ void __launch_bounds__(256, 2) conv_igemm_dma_kernel(ConvDmaArgs p) {
  static_assert(!X3 || sizeof(TO) == 2, "bf16x3 operands: split bf16 output");
  constexpr int BK = 64;
  constexpr int BUF = (BM + BN) * 128;           // bytes per stage
  constexpr int WMT = BM / 2, WNT = BN / 2;      // wave tile: pixels x channels
  constexpr int FB = WMT / 16, FA = WNT / 16;    // MFMA tiles per wave: pixels (B operand), channels (A operand)
  constexpr int XI = BM / 32, WI = BN / 32;      // LDS-DMA instructions per wave and operand (8 rows each)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int bid = blockIdx.x;
  if constexpr (PAIR) pair_enter(p, bid);
  int nwg = p.tiles_m * p.tiles_n;
  {
    int q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
  int m0 = tile_m * BM, n0 = tile_n * BN;

  const bf16_t* __restrict__ X = (const bf16_t*)p.x;
  const bf16_t* __restrict__ Wt = (const bf16_t*)p.w;
  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(X), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Wt), 0, (int)p.w_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;

  int tid = threadIdx.x, lane = tid & 63;
  int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wm = wid >> 1, wn = wid & 1;
  int lrow = lane >> 3, lc = lane & 7;

  // staging: wave `wid`, instruction i covers tile rows R0 = (i*4 + wid)*8 .. +8 ; lane -> row R0 + lrow, LDS chunk lc
  // (lane-linear image), source chunk lc ^ f(row)
  // (arrays sized 4 = max(XI, WI): hipcc's host pass silently drops the kernel instantiation when these are sized by XI / WI)
  int x_ih0[4], x_iw0[4]; unsigned x_base[4]; bool x_ok[4]; int x_q[4];
  unsigned w_off[4]; bool w_ok[4];
  static_assert(XI <= 4 && WI <= 4, "staging tables hold 4 rows groups per wave");
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    int row = (i * 4 + wid) * 8 + lrow;
    x_q[i] = lc ^ ((row >> 1) & 7);
    int m = m0 + row;
    x_ok[i] = m < p.M;
    int mm = x_ok[i] ? m : 0;
    int ow = mm % p.OW; int t = mm / p.OW; int oh = t % p.OH; int n = t / p.OH;
    x_ih0[i] = oh * p.stride - p.pad; x_iw0[i] = ow * p.stride - p.pad;
    x_base[i] = (unsigned)n * (unsigned)(p.H * p.W * (X3 ? p.sk.x_pitch : p.C));
  }
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    int row = (i * 4 + wid) * 8 + lrow;
    int q = lc ^ ((row >> 1) & 7);
    int nn = n0 + row;
    w_ok[i] = nn < p.K;
    w_off[i] = ((unsigned)(w_ok[i] ? nn : 0) * (unsigned)p.Kgemm + (unsigned)q * 8u) * 2u;
  }

  auto stage = [&](int kt, int buf) {
    // k-tile order: channel block outermost, the R*S taps innermost (input rows stay L2/L1-resident across the taps)
    int RS = p.R * p.S;
    int cb = kt / RS; int rs = kt - cb * RS;  // wave-uniform: C % 64 == 0 -> the whole k-tile sits inside one (r,s)
    int ch0 = cb * BK; int k0 = rs * p.C + ch0; int r = rs / p.S; int s = rs - r * p.S;
    int xpitch = p.C, xch0 = ch0;
    if constexpr (X3) {               // virtual channel block cb = (real block) * nseg + segment; the segment selects the plane of x
      int cbr = cb / p.sk.nseg, sg = cb - cbr * p.sk.nseg;
      xpitch = p.sk.x_pitch; xch0 = ((p.sk.seg_lo >> sg) & 1) * p.sk.cr + cbr * BK;
    }
    char* base = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      int R0 = (i * 4 + wid) * 8;
      int ih = x_ih0[i] + r, iw = x_iw0[i] + s;
      bool ok = x_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      unsigned off = (x_base[i] + (unsigned)((ih * p.W + iw) * xpitch + xch0 + x_q[i] * 8)) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_t*)(base + R0 * 128), 16, ok ? off : OOB, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      int R0 = (i * 4 + wid) * 8;
      unsigned off = w_off[i] + (unsigned)k0 * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_t*)(base + BM * 128 + R0 * 128), 16, w_ok[i] ? off : OOB, 0, 0, 0);
    }
  };

  f32x4 acc[FA][FB];
#pragma unroll
  for (int a = 0; a < FA; ++a)
#pragma unroll
    for (int b = 0; b < FB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  int nk = p.Kgemm / BK;
  int frow = lane & 15, fq = lane >> 4;
  // NS LDS stages, prefetch distance NS-1 k-tiles: a 64x128 tile multiplies one k-tile in ~256 MFMA cycles, far less than an
  // L2 round trip, so the DMA of k-tile kt+NS-1 is issued while kt is multiplied and each iteration waits only for the
  // OLDEST group (counted vmcnt: the (NS-2) younger groups of XI+WI DMAs each stay in flight).
  constexpr int PER = XI + WI;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) stage(s, s);
  int buf = 0, pbuf = NS - 1;          // buffer of k-tile kt ; buffer that k-tile kt+NS-1 goes to
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + NS - 1 <= nk) {           // steady state: NS-2 younger groups outstanding behind the one being waited for
      if constexpr ((NS - 2) * PER == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if constexpr ((NS - 2) * PER == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if constexpr ((NS - 2) * PER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr ((NS - 2) * PER == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();      // everyone's DMA of k-tile kt landed, and everyone finished reading buffer `pbuf` (k-tile kt-1)
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NS - 1 < nk) stage(kt + NS - 1, pbuf);
    const char* bx = smem + buf * BUF;
    const char* bw = bx + BM * 128;
    i32x4 fa[2][FA], fb[2][FB];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int a = 0; a < FA; ++a) fa[ks][a] = *reinterpret_cast<const i32x4*>(bw + swz128(wn * WNT + a * 16 + frow, ks * 4 + fq));
#pragma unroll
      for (int b = 0; b < FB; ++b) fb[ks][b] = *reinterpret_cast<const i32x4*>(bx + swz128(wm * WMT + b * 16 + frow, ks * 4 + fq));
    }
    __builtin_amdgcn_s_setprio(1);
#if UNIT_DBGMID == 2      // diagnostic build (tools/exp_feed.sh): no MFMA -> the loop runs at the operand feed rate
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int a = 0; a < FA; ++a) acc[a][0] += __builtin_bit_cast(f32x4, fa[ks][a]);
#pragma unroll
      for (int b = 0; b < FB; ++b) acc[0][b] += __builtin_bit_cast(f32x4, fb[ks][b]);
    }
#else
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[ks][a]), __builtin_bit_cast(bf16x8, fb[ks][b]), acc[a][b], 0, 0, 0);
#endif
    __builtin_amdgcn_s_setprio(0);
    pbuf = buf;
    buf = (buf + 1 == NS) ? 0 : buf + 1;
  }

  if constexpr (sizeof(TO) == 2) {
    if ((p.ldy & 7) == 0) {          // row-major epilogue through a wave-private LDS scratch (conv_epilogue.h)
      __syncthreads();               // every wave is done with the operand stages
      epilogue_rows_bf16_fast<FA, FB, X3>(acc, smem + wid * EpiCfg<FA>::BYTES, m0 + wm * WMT, n0 + wn * WNT, p, lane);
      return;
    }
  }
  TO* __restrict__ Y = (TO*)p.y;
  const TO* __restrict__ Rz = (const TO*)p.residual;
  const TO* __restrict__ Mk = (const TO*)p.mask_ref;
  bool plain = (p.oy_mul == 1 && p.OHf == p.OH && p.OWf == p.OW);
#pragma unroll
  for (int b = 0; b < FB; ++b) {
    int m = m0 + wm * WMT + b * 16 + frow;
    if (m >= p.M) continue;
    long off;
    if (plain) off = (long)m * p.ldy;
    else {
      int ow = m % p.OW; int t = m / p.OW; int oh = t % p.OH; int n = t / p.OH;
      off = (((long)n * p.OHf + (long)oh * p.oy_mul) * p.OWf + (long)ow * p.oy_mul) * p.ldy;
    }
#pragma unroll
    for (int a = 0; a < FA; ++a) {
      int n = n0 + wn * WNT + a * 16 + fq * 4;
      if (n >= p.ldy) continue;
      float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
      if (p.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += (n + j < p.K) ? p.bias[n + j] : 0.f;
      }
      if (Rz) {
        float rr[4]; Out4<TO>::load(Rz + off + n, rr);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += rr[j];
      }
      if (p.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (Mk) {
        float mm[4]; Out4<TO>::load(Mk + off + n, mm);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = mm[j] > 0.f ? v[j] : 0.f;
      }
      Out4<TO>::store(Y + off + n, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// In-workgroup split-K, ping-pong scheduled: for layers whose 128x128 tiling yields fewer workgroups than CUs (res4:
// 9 576 pixels x 256 channels = 150 tiles) a lone 4-wave workgroup per CU leaves every SIMD with ONE wave, whose LDS-DMA
// issue (8 pieces, ~60-100 cycles each) and MFMA issue (32 x 16 cycles) serialise: ~50 % MFMA utilisation.
// Here the workgroup has two 4-wave groups (waves g*4..g*4+3 of group g; waves i and i+4 share a SIMD). Group g multiplies
// the k-tiles kt = g, g+2, ... of the SAME 128x128 output tile into its own accumulators from its own two LDS stages
// (2 groups x 2 stages x 32 KB = 128 KB), and the groups alternate between a LOAD section (DMA issue of the group's next
// k-tile + the 16 fragment reads of the current one) and an MFMA section (32 MFMAs), one workgroup barrier per section:
// each SIMD's matrix pipe always has exactly one wave feeding it. Group 1 finally parks its accumulators in (its own)
// LDS, group 0 adds them and runs the epilogue. Deterministic (fixed summation order), no cross-workgroup traffic.
// ---------------------------------------------------------------------------------------------------
template <typename TO>
__global__ void __launch_bounds__(512, 2) conv_igemm_dma_ksplit_kernel(ConvDmaArgs p) {
  constexpr int BM = 128, BN = 128, BK = 64;
  constexpr int BUF = (BM + BN) * 128;           // 32 KB per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int nwg = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    int q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
  int m0 = tile_m * BM, n0 = tile_n * BN;

  const bf16_t* __restrict__ X = (const bf16_t*)p.x;
  const bf16_t* __restrict__ Wt = (const bf16_t*)p.w;
  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(X), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Wt), 0, (int)p.w_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;

  int tid = threadIdx.x, lane = tid & 63;
  int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int grp = wid >> 2, w4 = wid & 3;
  int wm = w4 >> 1, wn = w4 & 1;
  int lrow = lane >> 3, lc = lane & 7;
  char* gbase = smem + grp * 2 * BUF;

  int x_ih0[4], x_iw0[4]; unsigned x_base[4]; bool x_ok[4]; int x_q[4];
  unsigned w_off[4]; bool w_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = (i * 4 + w4) * 8 + lrow;
    int q = lc ^ ((row >> 1) & 7);
    x_q[i] = q;
    int m = m0 + row;
    x_ok[i] = m < p.M;
    int mm = x_ok[i] ? m : 0;
    int ow = mm % p.OW; int t = mm / p.OW; int oh = t % p.OH; int n = t / p.OH;
    x_ih0[i] = oh * p.stride - p.pad; x_iw0[i] = ow * p.stride - p.pad;
    x_base[i] = (unsigned)n * (unsigned)(p.H * p.W * p.C);
    int nn = n0 + row;
    w_ok[i] = nn < p.K;
    w_off[i] = ((unsigned)(w_ok[i] ? nn : 0) * (unsigned)p.Kgemm + (unsigned)q * 8u) * 2u;
  }

  auto stage = [&](int kt, int buf) {
    int RS = p.R * p.S;
    int cb = kt / RS; int rs = kt - cb * RS;
    int ch0 = cb * BK; int k0 = rs * p.C + ch0; int r = rs / p.S; int s = rs - r * p.S;
    char* base = gbase + buf * BUF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int R0 = (i * 4 + w4) * 8;
      int ih = x_ih0[i] + r, iw = x_iw0[i] + s;
      bool ok = x_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      unsigned off = (x_base[i] + (unsigned)((ih * p.W + iw) * p.C + ch0 + x_q[i] * 8)) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_t*)(base + R0 * 128), 16, ok ? off : OOB, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int R0 = (i * 4 + w4) * 8;
      unsigned off = w_off[i] + (unsigned)k0 * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_t*)(base + BM * 128 + R0 * 128), 16, w_ok[i] ? off : OOB, 0, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  int nk = p.Kgemm / BK;
  int nkg = (nk - grp + 1) >> 1;            // k-tiles of this group: kt = grp + 2*j
  int nk0 = (nk + 1) >> 1, nk1 = nk >> 1;
  int frow = lane & 15, fq = lane >> 4;
  if (nkg > 0) stage(grp, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (grp == 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }     // group 1 runs one section behind
  for (int j = 0; j < nkg; ++j) {
    // ---- LOAD section
    if (j + 1 < nkg) stage(grp + 2 * (j + 1), (j + 1) & 1);
    const char* bx = gbase + (j & 1) * BUF;
    const char* bw = bx + BM * 128;
    i32x4 fa[2][4], fb[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int a = 0; a < 4; ++a) fa[ks][a] = *reinterpret_cast<const i32x4*>(bw + swz128(wn * 64 + a * 16 + frow, ks * 4 + fq));
#pragma unroll
      for (int b = 0; b < 4; ++b) fb[ks][b] = *reinterpret_cast<const i32x4*>(bx + swz128(wm * 64 + b * 16 + frow, ks * 4 + fq));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- MFMA section (registers only)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[ks][a]), __builtin_bit_cast(bf16x8, fb[ks][b]), acc[a][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's DMA of the group's next k-tile has landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  // equalise the barrier counts: group 0 executed 1 + 2*nk0, group 1 executed 2 + 2*nk1
  {
    int mine = grp == 0 ? 2 * nk0 : 1 + 2 * nk1;
    int other = grp == 0 ? 1 + 2 * nk1 : 2 * nk0;
    for (int e = mine; e < other; ++e) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
  }
  // ---- reduction: group 1 parks its accumulators in its own LDS stages (64 KB), group 0 adds them
  float* red = reinterpret_cast<float*>(smem + 2 * BUF);
  if (grp == 1) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        *reinterpret_cast<f32x4*>(red + ((w4 * 16 + a * 4 + b) * 64 + lane) * 4) = acc[a][b];
  }
  __syncthreads();
  if (grp == 1) return;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      f32x4 o = *reinterpret_cast<const f32x4*>(red + ((w4 * 16 + a * 4 + b) * 64 + lane) * 4);
      acc[a][b] += o;
    }

  if constexpr (sizeof(TO) == 2) {
    if ((p.ldy & 7) == 0) {          // scratch = group 0's own operand stages (all waves passed the barrier above)
      epilogue_rows_bf16<4, 4>(acc, smem + w4 * EpiCfg<4>::BYTES, m0 + wm * 64, n0 + wn * 64, p, lane);
      return;
    }
  }
  TO* __restrict__ Y = (TO*)p.y;
  const TO* __restrict__ Rz = (const TO*)p.residual;
  const TO* __restrict__ Mk = (const TO*)p.mask_ref;
  bool plain = (p.oy_mul == 1 && p.OHf == p.OH && p.OWf == p.OW);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    int m = m0 + wm * 64 + b * 16 + frow;
    if (m >= p.M) continue;
    long off;
    if (plain) off = (long)m * p.ldy;
    else {
      int ow = m % p.OW; int t = m / p.OW; int oh = t % p.OH; int n = t / p.OH;
      off = (((long)n * p.OHf + (long)oh * p.oy_mul) * p.OWf + (long)ow * p.oy_mul) * p.ldy;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int n = n0 + wn * 64 + a * 16 + fq * 4;
      if (n >= p.ldy) continue;
      float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
      if (p.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += (n + j < p.K) ? p.bias[n + j] : 0.f;
      }
      if (Rz) {
        float rr[4]; Out4<TO>::load(Rz + off + n, rr);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += rr[j];
      }
      if (p.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (Mk) {
        float mm[4]; Out4<TO>::load(Mk + off + n, mm);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = mm[j] > 0.f ? v[j] : 0.f;
      }
      Out4<TO>::store(Y + off + n, v);
    }
  }
}

template <typename TO>
static int launch_ksplit(ConvDmaArgs& a, hipStream_t st) {
  a.tiles_m = cdiv(a.M, 128); a.tiles_n = cdiv(a.K, 128);
  size_t lds = 4 * (128 + 128) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_igemm_dma_ksplit_kernel<TO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  conv_igemm_dma_ksplit_kernel<TO><<<a.tiles_m * a.tiles_n, 512, lds, st>>>(a);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

template <typename TO, int BM, int BN, int NS, bool X3 = false>
static int launch_dma(ConvDmaArgs& a, hipStream_t st) {
  a.tiles_m = cdiv(a.M, BM); a.tiles_n = cdiv(a.K, BN);
  size_t lds = (size_t)NS * (BM + BN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_igemm_dma_kernel<TO, BM, BN, NS, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)conv_igemm_dma_kernel<TO, BM, BN, NS, X3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  if (a.second.on) {          // pair launch (conv_epilogue.h ConvSecond): the second problem's tiles follow the first's
    a.second.tiles_m = cdiv(a.second.M, BM);
    a.second.tiles0 = a.tiles_m * a.tiles_n;
    conv_igemm_dma_kernel<TO, BM, BN, NS, X3, true><<<(a.tiles_m + a.second.tiles_m) * a.tiles_n, 256, lds, st>>>(a);
  } else
  conv_igemm_dma_kernel<TO, BM, BN, NS, X3><<<a.tiles_m * a.tiles_n, 256, lds, st>>>(a);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// bf16x3 operands (unit_conv2d_fwd_x3, conv_igemm256.hip fills the argument block): tile 0 = 128x128, 1 = 64x128, 2 = 128x64 of the
// 4-wave kernel, >= 100 = loader / consumer tile code
int unit_conv_mid_x3_launch(ConvDmaArgs& a, int tile, hipStream_t st) {
  if (a.sk.nseg < 2 || (a.ldy & 7) != 0) { unit_set_error("conv_mid_x3: split operands with output rows of 16-byte vectors only"); return UNIT_ERR_UNSUPPORTED; }
  if (tile >= 100) return unit_conv_lc_launch(a, UNIT_BF16, tile, st);
  if (tile == 0) return launch_dma<bf16_t, 128, 128, 2, true>(a, st);
  if (tile == 1) return launch_dma<bf16_t, 64, 128, 3, true>(a, st);
  if (tile == 2) return launch_dma<bf16_t, 128, 64, 3, true>(a, st);
  unit_set_error("conv_mid_x3: tile must be 0, 1, 2 or a loader / consumer tile code");
  return UNIT_ERR_UNSUPPORTED;
}

// Same contract as unit_conv2d_fwd (include/unit_hip.h) restricted to bf16 inputs and C % 64 == 0.
// tile: 0 = 128x128, 1 = 64 (pixels) x 128 (channels), 2 = 128 x 64, 3 = 128x128 with in-workgroup split-K (8 waves).
extern "C" int unit_conv2d_fwd_mid(const void* x, const void* w, void* y, const float* bias, const void* residual,
                                   const void* mask_ref, int out_dtype, int N, int H, int W, int C, int K, int R, int S, int stride,
                                   int pad, int OH, int OW, int ldy, int oy_mul, int OHf, int OWf, int relu, int tile, void* stream) {
  return unit_conv_mid_impl(x, w, y, bias, residual, mask_ref, out_dtype, N, H, W, C, K, R, S, stride, pad, OH, OW, ldy, oy_mul, OHf, OWf, relu, tile,
                            nullptr, stream);
}

int unit_conv_mid_impl(const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref, int out_dtype, int N,
                       int H, int W, int C, int K, int R, int S, int stride, int pad, int OH, int OW, int ldy, int oy_mul, int OHf, int OWf, int relu,
                       int tile, const UnitConvSecond* second, void* stream) {
  UNIT_CHECK_ARG(C % 64 == 0, "conv_mid: C must be a multiple of 64");
  UNIT_CHECK_ARG(ldy % 4 == 0 && ldy >= K, "conv_mid: ldy must be a multiple of 4 and >= K");
  UNIT_CHECK_ARG(OH == (H + 2 * pad - R) / stride + 1 && OW == (W + 2 * pad - S) / stride + 1, "conv_mid: OH/OW mismatch");
  UNIT_CHECK_ARG((OH - 1) * oy_mul < OHf && (OW - 1) * oy_mul < OWf, "conv_mid: output scatter out of range");
  UNIT_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)y % 16 == 0), "conv_mid: 16B alignment");
  UNIT_CHECK_ARG((tile >= 0 && tile <= 5) || tile >= 100, "conv_mid: tile must be 0..5 or a loader / consumer tile code (>= 100)");
  ConvDmaArgs a;
  a.sk = SplitK{0, 0, 0, 0}; a.mask_pitch = 0; a.second.on = 0;
  a.x = x; a.w = w; a.y = y; a.bias = bias; a.residual = residual; a.mask_ref = mask_ref;
  a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad;
  a.OH = OH; a.OW = OW; a.ldy = ldy; a.oy_mul = oy_mul; a.OHf = OHf; a.OWf = OWf; a.relu = relu;
  a.Kgemm = R * S * C; a.M = N * OH * OW;
  size_t xb = (size_t)N * H * W * C * 2, wb = (size_t)K * R * S * C * 2;
  UNIT_CHECK_ARG(xb < 0xFFFFFFF0ull && wb < 0xFFFFFFF0ull, "conv_mid: operand larger than 4 GiB");
  a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
  { int rc = unit_fill_second(a.second, second, R, S, stride, pad, oy_mul, (size_t)C * 2); if (rc != UNIT_OK) return rc; }
  if (K == 0 || (a.M == 0 && !a.second.on)) return UNIT_OK;
  UNIT_CHECK_ARG(!a.second.on || tile != 3, "conv_mid: the split-K tile has no pair form");
  hipStream_t st = (hipStream_t)stream;
  if (tile >= 100) return unit_conv_lc_launch(a, out_dtype, tile, st);       // persistent loader / consumer workgroups (conv_igemm_lc.hip)
  if (out_dtype == UNIT_BF16) {
    if (tile == 3) return launch_ksplit<bf16_t>(a, st);
    if (tile == 0) return launch_dma<bf16_t, 128, 128, 2>(a, st);
    if (tile == 1) return launch_dma<bf16_t, 64, 128, 3>(a, st);
    if (tile == 4) return launch_dma<bf16_t, 96, 128, 3>(a, st);      // 100 x 2 workgroups on the res4 1x1 -> 256 layers: one round, one per CU
    if (tile == 5) return launch_dma<bf16_t, 96, 128, 2>(a, st);
    return launch_dma<bf16_t, 128, 64, 3>(a, st);
  }
  if (out_dtype == UNIT_F32) {
    if (tile == 3) return launch_ksplit<float>(a, st);
    if (tile == 0) return launch_dma<float, 128, 128, 2>(a, st);
    if (tile == 1) return launch_dma<float, 64, 128, 3>(a, st);
    if (tile == 4) return launch_dma<float, 96, 128, 3>(a, st);
    if (tile == 5) return launch_dma<float, 96, 128, 2>(a, st);
    return launch_dma<float, 128, 64, 3>(a, st);
  }
  unit_set_error("conv_mid: unsupported out dtype");
  return UNIT_ERR_UNSUPPORTED;
}
