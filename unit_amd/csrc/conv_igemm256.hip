// conv_igemm256.hip -- large-tile bf16 implicit-GEMM convolution (forward / dgrad) for the big-M layers of the hot path
// (Res5 heads on 1024-2048 RoIs: M = 50 176 .. 100 352, RPN 3x3 conv) -- same math and epilogue as conv_igemm.hip.
//
// 256 (pixels) x 256 (channels) x 64 (k) tile, 512 threads = 8 waves (2 x 4), each wave 128 x 64 = 8 x 4 MFMA 16x16x32 tiles
// (12 ds_read_b128 per 32 MFMAs).  Operand tiles go HBM/L2 -> LDS directly with LDS-DMA (`buffer_load_dwordx4 ... lds`):
// no staging VGPRs, no ds_write; out-of-range voffsets (im2col zero padding, tile edges) deliver zeros. The LDS image is
// lane-linear per wave instruction (8 rows x 128 B), so the bank-conflict XOR swizzle is applied to the per-lane SOURCE
// chunk (chunk ^= (row>>1)&7) and undone in the fragment reads. Two 64 KB LDS buffers: the DMA of k-tile t+1 is in flight
// while k-tile t is multiplied; one vmcnt(0)+barrier per k-tile.  Requires C % 64 == 0 (every layer except the stem).
#include "conv_igemm256.h"
#include "conv_igemm128.h"
#include "conv_pair.h"
#ifndef UNIT_P8M_DEFAULT
#define UNIT_P8M_DEFAULT 0
#endif
#include "conv_epilogue.h"

// diagnostic builds only (tools/exp256.sh): 1 = no operand DMA after the first k-tile (MFMA + LDS-read bound of the loop),
// 2 = MFMAs replaced by a few VALU adds on the fragments (DMA + LDS-read bound of the loop). Results are garbage in both.
#ifndef UNIT_DBG256
#define UNIT_DBG256 0
#endif

// FB = 16-row MFMA blocks per wave along the pixel dimension: tile = (32*FB) pixels x 256 channels. FB = 7 (224 rows) divides
// the Res5 problem sizes (50 176 = 224 * 224 pixels per 1024 RoIs) into whole rounds of 256 workgroups where 256-row tiles
// leave the last round 1/2 - 3/4 empty; the LDS image keeps 256 rows per stage either way.
template <typename TO, bool PP, int FB>
__global__ void __launch_bounds__(512, 2) conv_igemm256_kernel(Conv256Args p) {
  constexpr int BM = 32 * FB, BMR = 256, BN = 256, BK = 64;
  constexpr int BUF = (BMR + BN) * 128;         // 64 KB per stage
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
  int wm = wid >> 2, wn = wid & 3;
  int lrow = lane >> 3, lc = lane & 7;

  // staging pattern: wave `wid`, instruction i (0..3) covers tile rows R0 = (i*8 + wid)*8 .. +8 ; lane -> row R0 + lrow,
  // LDS chunk lc (linear), source chunk lc ^ f(row)
  int x_ih0[4], x_iw0[4]; unsigned x_base[4]; bool x_ok[4]; int x_q[4];
  unsigned w_off[4]; bool w_ok[4]; int w_q[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = (i * 8 + wid) * 8 + lrow;
    int q = lc ^ ((row >> 1) & 7);
    x_q[i] = q; w_q[i] = q;
    int m = m0 + row;
    x_ok[i] = m < p.M && row < BM;
    int mm = x_ok[i] ? m : 0;
    int ow = mm % p.OW; int t = mm / p.OW; int oh = t % p.OH; int n = t / p.OH;
    x_ih0[i] = oh * p.stride - p.pad; x_iw0[i] = ow * p.stride - p.pad;
    x_base[i] = (unsigned)n * (unsigned)(p.H * p.W * p.C);
    int nn = n0 + row;
    w_ok[i] = nn < p.K;
    w_off[i] = ((unsigned)(w_ok[i] ? nn : 0) * (unsigned)p.Kgemm + (unsigned)q * 8u) * 2u;
  }

  auto stage = [&](int kt, int buf) {
    // k-tile order: channel block outermost, the R*S taps innermost. Consecutive k-tiles then read the SAME input rows
    // shifted by one pixel / one row (L2- and mostly L1-resident), instead of coming back to them C/64 k-tiles later when
    // the other workgroups of the XCD have pushed them out of the 4 MB L2 (3x3: the input was fetched ~9x past L2).
    int RS = p.R * p.S;
    int cb = kt / RS; int rs = kt - cb * RS;  // wave-uniform: C % 64 == 0 -> the whole k-tile sits inside one (r,s)
    int ch0 = cb * BK; int k0 = rs * p.C + ch0; int r = rs / p.S; int s = rs - r * p.S;
    char* base = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int R0 = (i * 8 + wid) * 8;
      if (R0 >= BM) continue;                 // wave-uniform: pixel rows beyond a 224-row tile are never read
      int ih = x_ih0[i] + r, iw = x_iw0[i] + s;
      bool ok = x_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      unsigned off = (x_base[i] + (unsigned)((ih * p.W + iw) * p.C + ch0 + x_q[i] * 8)) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void*)(base + R0 * 128), 16, ok ? off : OOB, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int R0 = (i * 8 + wid) * 8;
      unsigned off = w_off[i] + (unsigned)k0 * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(base + BMR * 128 + R0 * 128), 16, w_ok[i] ? off : OOB, 0, 0, 0);
    }
  };

  f32x4 acc[4][FB];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < FB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  int nk = p.Kgemm / BK;
  int frow = lane & 15, fq = lane >> 4;
  stage(0, 0);
  __syncthreads();
  if constexpr (!PP) {
    for (int kt = 0; kt < nk; ++kt) {
      int buf = kt & 1;
      if (kt + 1 < nk && UNIT_DBG256 != 1) stage(kt + 1, buf ^ 1);
      const char* bx = smem + buf * BUF;
      const char* bw = bx + BMR * 128;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        i32x4 fa[4], fb[FB];
#pragma unroll
        for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const i32x4*>(bw + swz256(wn * 64 + a * 16 + frow, ks * 4 + fq));
#pragma unroll
        for (int b = 0; b < FB; ++b) fb[b] = *reinterpret_cast<const i32x4*>(bx + swz256(wm * (FB * 16) + b * 16 + frow, ks * 4 + fq));
        __builtin_amdgcn_s_setprio(1);
#if UNIT_DBG256 == 2
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][0] += __builtin_bit_cast(f32x4, fa[a]);
#pragma unroll
        for (int b = 0; b < FB; ++b) acc[0][b] += __builtin_bit_cast(f32x4, fb[b]);
#else
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < FB; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[a]), __builtin_bit_cast(bf16x8, fb[b]), acc[a][b], 0, 0, 0);
#endif
        __builtin_amdgcn_s_setprio(0);
      }
      __syncthreads();   // vmcnt(0): this wave's DMA of tile kt+1 landed ; barrier: everyone's did, and everyone finished reading `buf`
    }
  } else {
    // ---- ping-pong schedule: waves 0-3 (group 0) and waves 4-7 (group 1) share the four SIMDs pairwise (wave i and i+4).
    // Every interval between two workgroup barriers one group runs a pure-MFMA section (32 MFMAs, 512 cycles) while the
    // other runs its LOAD section (12 ds_read_b128 fragments + the LDS-DMA issue of the next k-tile): the matrix pipe of each
    // SIMD always has exactly one wave feeding it, and no wave ever waits for LDS data inside its MFMA section.
    // Group 1 runs one interval behind group 0 (one extra barrier up front, one fewer at the end).
    // Hazards (2 LDS buffers, prefetch distance 1 k-tile):
    //   WAR  DMA(t+1) -> buffer of tile t-1: issued by a group in its LOAD(t, ks=0) section; the other group's last reads of
    //        tile t-1 were completed (lgkmcnt(0)) before the barrier that precedes this interval.
    //   RAW  tile t+1 is first read one barrier after every wave waited vmcnt(0) for its own DMA (end of LOAD(t, ks=1)).
    const int grp = wm;     // wm == wid >> 2
    if (grp == 1) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
      int buf = kt & 1;
      const char* bx = smem + buf * BUF;
      const char* bw = bx + BMR * 128;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        // LOAD section
        if (ks == 0 && kt + 1 < nk) stage(kt + 1, buf ^ 1);
        i32x4 fa[4], fb[FB];
#pragma unroll
        for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const i32x4*>(bw + swz256(wn * 64 + a * 16 + frow, ks * 4 + fq));
#pragma unroll
        for (int b = 0; b < FB; ++b) fb[b] = *reinterpret_cast<const i32x4*>(bx + swz256(wm * (FB * 16) + b * 16 + frow, ks * 4 + fq));
        if (ks == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // MFMA section (registers only)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < FB; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[a]), __builtin_bit_cast(bf16x8, fb[b]), acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
  }

  if constexpr (sizeof(TO) == 2) {
    if ((p.ldy & 7) == 0) {          // row-major epilogue through a wave-private LDS scratch (conv_epilogue.h)
      __syncthreads();               // every wave is done with the operand stages
      epilogue_rows_bf16<4, FB>(acc, smem + wid * EpiCfg<4>::BYTES, m0 + wm * (FB * 16), n0 + wn * 64, p, lane);
      return;
    }
  }
  TO* __restrict__ Y = (TO*)p.y;
  const TO* __restrict__ Rz = (const TO*)p.residual;
  const TO* __restrict__ Mk = (const TO*)p.mask_ref;
  bool plain = (p.oy_mul == 1 && p.OHf == p.OH && p.OWf == p.OW);
#pragma unroll
  for (int b = 0; b < FB; ++b) {
    int m = m0 + wm * (FB * 16) + b * 16 + frow;
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
        float rr[4]; O4<TO>::load(Rz + off + n, rr);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += rr[j];
      }
      if (p.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (Mk) {
        float mm[4]; O4<TO>::load(Mk + off + n, mm);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = mm[j] > 0.f ? v[j] : 0.f;
      }
      O4<TO>::store(Y + off + n, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// variant 6: 3x3 / stride 1 / pad 1 convolution on 7x7 maps (conv2 of the Res5 blocks and its dgrad: 47 % of the Res5 FLOPs)
// with the INPUT tile shared by the nine taps. Output pixels of a tile are rows m0 .. m0+255 of the flattened [RoI][7][7]
// pixel list; tap (r,s) of output row m reads input row m + (r-1)*7 + (s-1) when that neighbour is inside the 7x7 map and
// zero otherwise. So one 272-row input "super-tile" (rows m0-8 .. m0+263, 34 KB) per 64-channel block serves all nine
// k-tiles of that block: the per-k-tile operand feed drops from 64 KB (pixels + weights) to 32 KB of weights + 1/9 of
// 34 KB, and the fragment reads of a tap address the super-tile through a per-lane row map (out-of-map neighbours point
// at an all-zero LDS row). LDS: 2 x 273 x 128 B input (double-buffered across channel blocks) + 2 x 32 KB weight stages.
// bf16 output, ldy % 8 == 0.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512, 2) conv_igemm256_halo7_kernel(Conv256Args p) {
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int XROWS = 272, XZERO = 272;         // super-tile rows; index of the all-zero row
  constexpr int XBUF = (XROWS + 1) * 128;          // 34 944 B
  constexpr int WBUF = BN * 128;                   // 32 KB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;                                 // 2 x XBUF
  char* ws = smem + 2 * XBUF;                      // 2 x WBUF

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
  int wm = wid >> 2, wn = wid & 3;
  int lrow = lane >> 3, lc = lane & 7;

  // zero row of both input buffers (never written by the DMA)
  if (tid < 16) {
    *reinterpret_cast<i32x4*>(xs + XZERO * 128 + (tid & 7) * 16 + (tid >> 3) * XBUF) = i32x4{0, 0, 0, 0};
  }

  // weight staging (as in the generic kernel): wave `wid`, instruction i covers channel rows (i*8 + wid)*8 .. +8
  unsigned w_off[4]; bool w_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = (i * 8 + wid) * 8 + lrow;
    int q = lc ^ ((row >> 1) & 7);
    int nn = n0 + row;
    w_ok[i] = nn < p.K;
    w_off[i] = ((unsigned)(w_ok[i] ? nn : 0) * (unsigned)p.Kgemm + (unsigned)q * 8u) * 2u;
  }
  // input super-tile staging: 34 pieces of 8 rows; wave `wid` takes pieces i*8 + wid (i = 0..4, piece < 34)
  unsigned xs_off[5]; bool xs_ok[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    int row = (i * 8 + wid) * 8 + lrow;            // super-tile row
    long g = (long)m0 - 8 + row;                   // flattened input pixel
    xs_ok[i] = row < XROWS && g >= 0 && g < (long)p.M;
    int q = lc ^ ((row >> 1) & 7);
    xs_off[i] = (unsigned)((xs_ok[i] ? g : 0) * p.C + q * 8) * 2u;
  }
  auto stage_x = [&](int cb, int buf) {
    char* base = xs + buf * XBUF;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      int R0 = (i * 8 + wid) * 8;
      if (R0 >= XROWS) continue;                   // wave-uniform
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void*)(base + R0 * 128), 16, xs_ok[i] ? xs_off[i] + (unsigned)cb * 128u : OOB, 0, 0, 0);
    }
  };
  auto stage_w = [&](int kt, int buf) {            // k-tile kt = (channel block cb, tap): weights k index = tap*C + cb*64
    int cb = kt / 9, tap = kt - cb * 9;
    int k0 = tap * p.C + cb * BK;
    char* base = ws + buf * WBUF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int R0 = (i * 8 + wid) * 8;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(base + R0 * 128), 16, w_ok[i] ? w_off[i] + (unsigned)k0 * 2u : OOB, 0, 0, 0);
    }
  };

  // per-lane map of the 8 fragment rows of this wave: position inside the 7x7 map (packed oh*8 + ow)
  int frow = lane & 15, fq = lane >> 4;
  int pos[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    int m = m0 + wm * 128 + b * 16 + frow;
    int px = m % 49;
    pos[b] = ((px / 7) << 3) | (px % 7);
  }

  f32x4 acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  int ncb = p.C / BK, nk = 9 * ncb;
  stage_x(0, 0);
  stage_w(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    int cb = kt / 9, tap = kt - cb * 9;
    if (kt + 1 < nk) stage_w(kt + 1, (kt + 1) & 1);
    if (tap == 0 && cb + 1 < ncb) stage_x(cb + 1, (cb + 1) & 1);
    const char* bx = xs + (cb & 1) * XBUF;
    const char* bw = ws + (kt & 1) * WBUF;
    int dr = tap / 3 - 1, dc = tap - (tap / 3) * 3 - 1;
    int delta = 8 + dr * 7 + dc;
    // byte offset of this lane's 8 pixel rows inside the super-tile for this tap (swizzle term added per k-substep)
    int rowoff[8], rowsw[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      int oh = (pos[b] >> 3) + dr, ow = (pos[b] & 7) + dc;
      bool ok = (unsigned)oh < 7u && (unsigned)ow < 7u;
      int row = ok ? wm * 128 + b * 16 + frow + delta : XZERO;
      rowoff[b] = row * 128; rowsw[b] = (row >> 1) & 7;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      i32x4 fa[4], fb[8];
#pragma unroll
      for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const i32x4*>(bw + swz256(wn * 64 + a * 16 + frow, ks * 4 + fq));
#pragma unroll
      for (int b = 0; b < 8; ++b) fb[b] = *reinterpret_cast<const i32x4*>(bx + rowoff[b] + (((ks * 4 + fq) ^ rowsw[b]) << 4));
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[a]), __builtin_bit_cast(bf16x8, fb[b]), acc[a][b], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();   // vmcnt(0) + barrier: next weight stage (and, after tap 0, the next input super-tile) landed; reads of this stage done
  }
  __syncthreads();
  epilogue_rows_bf16<4, 8>(acc, smem + wid * EpiCfg<4>::BYTES, m0 + wm * 128, n0 + wn * 64, p, lane);
}

static int launch256_halo7(Conv256Args& a, hipStream_t st) {
  a.tiles_m = cdiv(a.M, 256); a.tiles_n = cdiv(a.K, 256);
  size_t lds = 2 * (272 + 1) * 128 + 2 * 256 * 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_igemm256_halo7_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  conv_igemm256_halo7_kernel<<<a.tiles_m * a.tiles_n, 512, lds, st>>>(a);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// variant 2: the same 256x256 tile with FOUR 32 KB stages of 32 k each instead of two 64 KB stages of 64 k.
// The two-stage loop drains the LDS-DMA queue at every barrier (all of a k-tile's 64 KB is issued in one burst, waited
// for in full, and only then is the next burst issued): measured operand feed 14 TB/s chip-wide, co-limiting with the
// MFMA pipe (tools/exp256.sh). With 32-k stages three stages (96 KB) stay in flight behind the one being multiplied and
// every iteration waits only for the oldest group (counted vmcnt), which the 4-wave kernels show streams at 18-20 TB/s.
// LDS image per stage: [256 pixel rows][64 B] + [256 channel rows][64 B]; one DMA piece = 16 rows x 64 B; a row's four
// 16-B chunks are XOR-swizzled with g((row>>2)&3), g = {0,2,3,1}, which makes the 4x16-lane ds_read_b128 groups
// conflict-free for this pitch.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz_g(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

template <typename TO>
__global__ void __launch_bounds__(512, 2) conv_igemm256_k32_kernel(Conv256Args p) {
  constexpr int BM = 256, BN = 256, BK = 32, NS = 4;
  constexpr int BUF = (BM + BN) * 64;           // 32 KB per stage
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
  int wm = wid >> 2, wn = wid & 3;
  int lrow = lane >> 2, lc = lane & 3;

  // staging: wave `wid`, piece i (0..1) covers tile rows R0 = (i*8 + wid)*16 .. +16 ; lane -> row R0 + lrow, LDS chunk lc
  int x_ih0[2], x_iw0[2]; unsigned x_base[2]; bool x_ok[2]; int x_q[2];
  unsigned w_off[2]; bool w_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row = (i * 8 + wid) * 16 + lrow;
    int q = lc ^ swz_g(row);
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
    int RS = p.R * p.S;                       // k-tile order: channel block (32) outermost, taps innermost
    int cb = kt / RS; int rs = kt - cb * RS;
    int ch0 = cb * BK; int k0 = rs * p.C + ch0; int r = rs / p.S; int s = rs - r * p.S;
    char* base = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int R0 = (i * 8 + wid) * 16;
      int ih = x_ih0[i] + r, iw = x_iw0[i] + s;
      bool ok = x_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      unsigned off = (x_base[i] + (unsigned)((ih * p.W + iw) * p.C + ch0 + x_q[i] * 8)) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void*)(base + R0 * 64), 16, ok ? off : OOB, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int R0 = (i * 8 + wid) * 16;
      unsigned off = w_off[i] + (unsigned)k0 * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(base + BM * 64 + R0 * 64), 16, w_ok[i] ? off : OOB, 0, 0, 0);
    }
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  int nk = p.Kgemm / BK;
  int frow = lane & 15, fq = lane >> 4;
  // per-lane fragment offsets inside a stage (row*64 + swizzled chunk): fixed for the whole loop
  int offa[4], offb[8];
#pragma unroll
  for (int a = 0; a < 4; ++a) { int row = wn * 64 + a * 16 + frow; offa[a] = BM * 64 + row * 64 + ((fq ^ swz_g(row)) << 4); }
#pragma unroll
  for (int b = 0; b < 8; ++b) { int row = wm * 128 + b * 16 + frow; offb[b] = row * 64 + ((fq ^ swz_g(row)) << 4); }

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) stage(s, s);
  int buf = 0, pbuf = NS - 1;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + NS - 1 <= nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // 2 younger groups of 4 pieces stay in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // everyone's DMA of stage kt landed, everyone finished reading buffer `pbuf` (stage kt-1)
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NS - 1 < nk) stage(kt + NS - 1, pbuf);
    const char* bs = smem + buf * BUF;
    i32x4 fa[4], fb[8];
#pragma unroll
    for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const i32x4*>(bs + offa[a]);
#pragma unroll
    for (int b = 0; b < 8; ++b) fb[b] = *reinterpret_cast<const i32x4*>(bs + offb[b]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[a]), __builtin_bit_cast(bf16x8, fb[b]), acc[a][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    pbuf = buf;
    buf = (buf + 1) & (NS - 1);
  }

  if constexpr (sizeof(TO) == 2) {
    if ((p.ldy & 7) == 0) {
      __syncthreads();
      epilogue_rows_bf16<4, 8>(acc, smem + wid * EpiCfg<4>::BYTES, m0 + wm * 128, n0 + wn * 64, p, lane);
      return;
    }
  }
  TO* __restrict__ Y = (TO*)p.y;
  const TO* __restrict__ Rz = (const TO*)p.residual;
  const TO* __restrict__ Mk = (const TO*)p.mask_ref;
  bool plain = (p.oy_mul == 1 && p.OHf == p.OH && p.OWf == p.OW);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    int m = m0 + wm * 128 + b * 16 + frow;
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
        float rr[4]; O4<TO>::load(Rz + off + n, rr);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += rr[j];
      }
      if (p.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (Mk) {
        float mm[4]; O4<TO>::load(Mk + off + n, mm);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = mm[j] > 0.f ? v[j] : 0.f;
      }
      O4<TO>::store(Y + off + n, v);
    }
  }
}

template <typename TO>
static int launch256_k32(Conv256Args& a, hipStream_t st) {
  a.tiles_m = cdiv(a.M, 256); a.tiles_n = cdiv(a.K, 256);
  size_t lds = 4 * (256 + 256) * 64;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_igemm256_k32_kernel<TO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  conv_igemm256_k32_kernel<TO><<<a.tiles_m * a.tiles_n, 512, lds, st>>>(a);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

template <typename TO, bool PP, int FB>
static int launch256(Conv256Args& a, hipStream_t st) {
  a.tiles_m = cdiv(a.M, 32 * FB); a.tiles_n = cdiv(a.K, 256);
  size_t lds = 2 * (256 + 256) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_igemm256_kernel<TO, PP, FB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  conv_igemm256_kernel<TO, PP, FB><<<a.tiles_m * a.tiles_n, 512, lds, st>>>(a);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

static void set_div_magics(Conv256Args& a) {
  unsigned long long mx = (unsigned long long)(a.M + 512) * (unsigned long long)(a.OW > a.OH ? a.OW : a.OH);
  bool ok = mx < 0xFFFFFFFFull;
  a.magic_ow = ok ? div_magic((unsigned)a.OW) : 0u;
  a.magic_oh = ok ? div_magic((unsigned)a.OH) : 0u;
}

// Position classes of a 3x3 s1 p1 "same" conv (Conv256Args::pm_cls): the map rows split into runs with the same in-map tap rows
// (first row / middle rows / last row; fewer for maps of 1 or 2 rows), the columns likewise; a class = a row run x a column run.
// Sorted by taps, heaviest first (the kernel starts the heavy tiles first). Leaves pm_ncls = 0 when the row-major tiles need fewer
// k-tiles (few images: every class is padded to whole 256-row tiles).
static void build_position_classes(Conv256Args& a) {
  struct Run { int lo, n, taps; };
  auto runs = [](int L, Run (&out)[3]) {
    int cnt = 0;
    for (int o = 0; o < L; ++o) {
      int t0 = o > 0 ? 0 : 1, t1 = o < L - 1 ? 2 : 1;                       // in-map taps [t0, t1] of output index o (pad 1)
      if (cnt > 0) {
        int p = out[cnt - 1].lo, q0 = p > 0 ? 0 : 1, q1 = p < L - 1 ? 2 : 1;
        if (q0 == t0 && q1 == t1) { ++out[cnt - 1].n; continue; }
      }
      out[cnt++] = Run{o, 1, t1 - t0 + 1};
    }
    return cnt;
  };
  Run rr[3], cc[3];
  int nr = runs(a.OH, rr), nc = runs(a.OW, cc);
  struct Cls { PmClass c; int taps; long tiles; } cls[9];
  int n = 0;
  for (int i = 0; i < nr; ++i)
    for (int j = 0; j < nc; ++j) {
      Cls k; k.c = PmClass{0, rr[i].n * cc[j].n, rr[i].lo, rr[i].n, cc[j].lo, cc[j].n, 0u, 0u}; k.taps = rr[i].taps * cc[j].taps;
      // (class row index i < N * np + 256; i * np < 2^32 is what fast_div needs)
      if (((unsigned long long)a.N * k.c.np + 256ull) * (unsigned long long)k.c.np < 0xFFFFFFFFull) { k.c.magic_np = div_magic(k.c.np); k.c.magic_cw = div_magic(k.c.cw); }
      k.tiles = cdiv((long)a.N * k.c.np, 256);
      cls[n++] = k;
    }
  for (int i = 1; i < n; ++i)                                                // insertion sort, taps descending (stable)
    for (int j = i; j > 0 && cls[j].taps > cls[j - 1].taps; --j) { Cls t = cls[j]; cls[j] = cls[j - 1]; cls[j - 1] = t; }
  long ktiles = 0, tiles = 0;
  for (int i = 0; i < n; ++i) { cls[i].c.tile0 = (int)tiles; tiles += cls[i].tiles; ktiles += cls[i].tiles * cls[i].taps; }
  if (ktiles * 100 >= (long)cdiv(a.M, 256) * 9 * 95) return;
  a.pm_ncls = n; a.tiles_m = (int)tiles;
  for (int i = 0; i < n; ++i) a.pm_cls[i] = cls[i].c;
}

// which MFMA shape variant 0 means for the p8 schedule (compile-time: no process state; callers A/B through the variant argument)
int unit_conv256_use_m32() { return UNIT_P8M_DEFAULT; }

// Same contract as unit_conv2d_fwd (include/unit_hip.h) restricted to bf16 inputs and C % 64 == 0.
extern "C" int unit_conv2d_fwd_big(const void* x, const void* w, void* y, const float* bias, const void* residual,
                                   const void* mask_ref, int out_dtype, int N, int H, int W, int C, int K, int R, int S, int stride,
                                   int pad, int OH, int OW, int ldy, int oy_mul, int OHf, int OWf, int relu, int variant, void* stream) {
  return unit_conv_big_impl(x, w, y, bias, residual, mask_ref, out_dtype, N, H, W, C, K, R, S, stride, pad, OH, OW, ldy, oy_mul, OHf, OWf, relu,
                            variant, nullptr, stream);
}

int unit_conv_big_impl(const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref, int out_dtype, int N,
                       int H, int W, int C, int K, int R, int S, int stride, int pad, int OH, int OW, int ldy, int oy_mul, int OHf, int OWf, int relu,
                       int variant, const UnitConvSecond* second, void* stream) {
  UNIT_CHECK_ARG(C % 64 == 0, "conv_big: C must be a multiple of 64");
  UNIT_CHECK_ARG(ldy % 4 == 0 && ldy >= K, "conv_big: ldy must be a multiple of 4 and >= K");
  UNIT_CHECK_ARG(OH == (H + 2 * pad - R) / stride + 1 && OW == (W + 2 * pad - S) / stride + 1, "conv_big: OH/OW mismatch");
  UNIT_CHECK_ARG((OH - 1) * oy_mul < OHf && (OW - 1) * oy_mul < OWf, "conv_big: output scatter out of range");
  UNIT_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)y % 16 == 0), "conv_big: 16B alignment");
  Conv256Args a;
  a.sk = SplitK{0, 0, 0, 0}; a.mask_pitch = 0; a.second.on = 0;
  a.x = x; a.w = w; a.y = y; a.bias = bias; a.residual = residual; a.mask_ref = mask_ref;
  a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad;
  a.OH = OH; a.OW = OW; a.ldy = ldy; a.oy_mul = oy_mul; a.OHf = OHf; a.OWf = OWf; a.relu = relu;
  a.Kgemm = R * S * C; a.M = N * OH * OW;
  size_t xb = (size_t)N * H * W * C * 2, wb = (size_t)K * R * S * C * 2;
  UNIT_CHECK_ARG(xb < 0xFFFFFFF0ull && wb < 0xFFFFFFF0ull, "conv_big: operand larger than 4 GiB");
  a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
  a.ex = EpiExtra{nullptr, nullptr, nullptr, 0}; a.ex_on = 0;
  a.x2 = nullptr; a.x2_bytes = 0; a.cb_split = 0; a.ratio2 = 1; a.pm_ncls = 0;
  set_div_magics(a);
  { int rc = unit_fill_second(a.second, second, R, S, stride, pad, oy_mul, (size_t)C * 2); if (rc != UNIT_OK) return rc; }
  if (K == 0 || (a.M == 0 && !a.second.on)) return UNIT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (a.second.on) {          // pair launch: the phase-interleaved kernel on row-major 256-row tiles
    UNIT_CHECK_ARG(out_dtype == UNIT_BF16 && (ldy & 7) == 0 && (variant == 0 || variant == 8 || variant == 12), "conv_big: pair launches: bf16 output, ldy % 8 == 0, variant 0 / 8 / 12");
    return unit_conv256_p8_launch(a, out_dtype, true, false, st);
  }
  // variant 0 / 4: one barrier per k-tile, 256-row tiles; 3: 224-row tiles; 5: 224 or 256 rows, whichever needs fewer
  // rounds x rows (isolated launches gain 6-7 % on the Res5 shapes: 50 176 = 224 * 224 pixels; inside the multi-stream step
  // the other streams already fill the partial last round and the 7 % extra operand feed of the smaller tile costs 0.8 %);
  // 1: ping-pong wave groups; 2: four 32-k stages
  // 6 (and 0 with UNIT_NO_HALO=0 when the shape allows it): 3x3 s1 p1 on 7x7 maps with the input super-tile shared by the nine taps
  {
    static int no_halo = -1;
    if (no_halo < 0) { const char* e = getenv("UNIT_NO_HALO"); no_halo = e ? atoi(e) : 1; }   // default: the p8 kernel below is faster
    bool halo_ok = out_dtype == UNIT_BF16 && (ldy & 7) == 0 && R == 3 && S == 3 && stride == 1 && pad == 1 && H == 7 && W == 7 &&
                   OH == 7 && OW == 7;
    if (variant == 6 && !halo_ok) { unit_set_error("conv_big: variant 6 needs a bf16-out 3x3 s1 p1 conv on 7x7 maps"); return UNIT_ERR_UNSUPPORTED; }
    if (halo_ok && (variant == 6 || (variant == 0 && !no_halo))) return launch256_halo7(a, st);
  }
  // 7 / 8 (8 = default for variant 0; UNIT_P8=0 falls back to the two-stage kernels below, =1 selects 7): four phases per k-tile,
  // half-tile staging under a counted vmcnt; 8 also issues the fragment reads inside the MFMA sections (conv_igemm256p8.hip)
  {
    static int p8 = -1;
    if (p8 < 0) { const char* e = getenv("UNIT_P8"); p8 = e ? atoi(e) : 2; }
    // 9: variant 8 on 224-row tiles; 10 (and 0 with UNIT_P8_ROWS=0): 224 or 256 rows, whichever needs fewer rounds x rows
    // (+5 % per isolated launch on the Res5 shapes)
    static int p8rows = -1;
    if (p8rows < 0) { const char* e = getenv("UNIT_P8_ROWS"); p8rows = e ? atoi(e) : 256; }   // in the multi-stream step 256 rows win (19.4 vs 19.7 ms): the other streams fill the partial round
    bool auto_rows = variant == 10 || (variant == 0 && p8 == 2 && p8rows == 0);
    bool r224 = variant == 9 || (variant == 0 && p8 == 2 && p8rows == 224);
    if (auto_rows) {
      long n_tiles = cdiv(K, 256);
      long c256 = (long)cdiv((long)cdiv(a.M, 256) * n_tiles, 256) * 256, c224 = (long)cdiv((long)cdiv(a.M, 224) * n_tiles, 256) * 224;
      r224 = c224 < c256;
      // (not where the position-class tiles apply: they need 256 rows and gain more)
      if (variant == 0 && R == 3 && S == 3 && stride == 1 && pad == 1 && OH == H && OW == W && H * W <= 4096) r224 = false;
    }
    // 11 (and 0 when UNIT_P8M_DEFAULT is 1): the variant-8 schedule on v_mfma_f32_32x32x16_bf16 (conv_igemm256p8m.hip)
    if (variant == 11 || (variant == 0 && p8 == 2 && !r224 && unit_conv256_use_m32())) return unit_conv256_p8m_launch(a, out_dtype, st);
    // 12: variant 8 with row-major tiles even where position-major tiles apply (A/B and bit-identity tests)
    if (variant == 12) return unit_conv256_p8_launch(a, out_dtype, true, false, st);
    if (variant >= 7 && variant <= 10 || (variant == 0 && p8)) {
      bool rm = variant >= 8 || (variant == 0 && p8 == 2);
      // position-class tiles (Conv256Args::pm_ncls): 3x3 s1 p1 conv on a small map, plain bf16 output, when skipping the all-padding
      // taps saves more k-tiles than padding every class to whole tiles costs
      if (rm && !r224 && (variant == 0 || variant == 8) && out_dtype == UNIT_BF16 && (ldy & 7) == 0 && R == 3 && S == 3 && stride == 1 &&
          pad == 1 && OH == H && OW == W && oy_mul == 1 && OHf == OH && OWf == OW && H * W <= 4096 && (size_t)N * H * W * ldy * 2 < 0xFFFFFFF0ull)
        build_position_classes(a);
      return unit_conv256_p8_launch(a, out_dtype, rm, r224, st);
    }
  }
  bool rows224 = variant == 3;
  if (variant == 5) {
    long n_tiles = cdiv(K, 256);
    long c256 = cdiv((long)cdiv(a.M, 256) * n_tiles, 256) * 256, c224 = cdiv((long)cdiv(a.M, 224) * n_tiles, 256) * 224;
    rows224 = c224 < c256;
  }
  if (out_dtype == UNIT_BF16) {
    if (variant == 2) return launch256_k32<bf16_t>(a, st);
    if (variant == 1) return launch256<bf16_t, true, 8>(a, st);
    return rows224 ? launch256<bf16_t, false, 7>(a, st) : launch256<bf16_t, false, 8>(a, st);
  }
  if (out_dtype == UNIT_F32) {
    if (variant == 2) return launch256_k32<float>(a, st);
    if (variant == 1) return launch256<float, true, 8>(a, st);
    return rows224 ? launch256<float, false, 7>(a, st) : launch256<float, false, 8>(a, st);
  }
  unit_set_error("conv_big: unsupported out dtype");
  return UNIT_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16x3 ("split") convolution: fp32-grade forward / dgrad on the bf16 MFMA kernels (conv_epilogue.h SplitK, csrc/split.hip).
//   x        split tensor [N,H,W][2][C] bf16 (hi plane, lo plane)
//   w        [K][R][S][C / 64][3][64] bf16 = per 64-channel block [Wh | Wh | Wl] (unit_weight_prep_x3) against the planes [lo | hi | hi] of x
//   y        split tensor [.][2][ldy]; residual: like y; mask_ref: split tensor with mask_c channels per plane (plane 0 carries the sign)
// y = split(relu?(sum_k (lo.Wh + hi.Wh + hi.Wl) + bias + residual) * (mask_ref > 0)), fp32 accumulation: ~2^-17 relative per product, the
// reference's fp32 arithmetic (fast_rcnn.py:37-101, rpn.py:55-101 run on fp32 convs) at three bf16 MFMA passes instead of the 16x
// slower fp32 MFMA. tile: -1 = 256x256 phase-interleaved kernel (conv_igemm256p8.hip), 0 / 1 / 2 = 128x128 / 64x128 / 128x64 4-wave
// tiles (conv_igemm128.hip), >= 100 = loader / consumer tile code (conv_igemm_lc.hip).
extern "C" int unit_conv2d_fwd_x3(const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref,
                                  int mask_c, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int OH, int OW, int ldy,
                                  int oy_mul, int OHf, int OWf, int relu, int tile, void* stream) {
  return unit_conv_x3_impl(x, w, y, bias, residual, mask_ref, mask_c, N, H, W, C, K, R, S, stride, pad, OH, OW, ldy, oy_mul, OHf, OWf, relu, tile,
                           nullptr, stream, 3);
}

// segs = 3: as unit_conv2d_fwd_x3. segs = 2 (round 6, the dgrad chain of the bf16x3 mode): w = [K][R][S][C / 64][2][64] = [Wh | Wl] against the
// planes [hi | hi] of x -- y = hi(x).(Wh + Wl): the weights at full 16-bit precision, the input at its hi plane (its lo plane is not read). For a
// gradient map that is one fresh, unbiased 2^-9 rounding per element and layer, which the weight gradients' sums over >= 2 394 rows average
// out (weights rounded instead would be the SAME error in every row); the forward pass never uses it.
extern "C" int unit_conv2d_fwd_x3s(const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref,
                                   int mask_c, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int OH, int OW, int ldy,
                                   int oy_mul, int OHf, int OWf, int relu, int tile, int segs, void* stream) {
  UNIT_CHECK_ARG(segs == 2 || segs == 3, "conv_x3s: 2 or 3 k-segments per 64-channel block");
  return unit_conv_x3_impl(x, w, y, bias, residual, mask_ref, mask_c, N, H, W, C, K, R, S, stride, pad, OH, OW, ldy, oy_mul, OHf, OWf, relu, tile,
                           nullptr, stream, segs);
}

int unit_conv_x3_impl(const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref, int mask_c, int N, int H,
                      int W, int C, int K, int R, int S, int stride, int pad, int OH, int OW, int ldy, int oy_mul, int OHf, int OWf, int relu, int tile,
                      const UnitConvSecond* second, void* stream, int segs) {
  UNIT_CHECK_ARG(C % 64 == 0, "conv_x3: C must be a multiple of 64");
  UNIT_CHECK_ARG(ldy % 8 == 0 && ldy >= K, "conv_x3: ldy must be a multiple of 8 and >= K");
  UNIT_CHECK_ARG(OH == (H + 2 * pad - R) / stride + 1 && OW == (W + 2 * pad - S) / stride + 1, "conv_x3: OH/OW mismatch");
  UNIT_CHECK_ARG((OH - 1) * oy_mul < OHf && (OW - 1) * oy_mul < OWf, "conv_x3: output scatter out of range");
  UNIT_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)y % 16 == 0), "conv_x3: 16B alignment");
  UNIT_CHECK_ARG(mask_ref == nullptr || (mask_c >= ldy && mask_c % 8 == 0), "conv_x3: mask_c must be the mask tensor's channels per plane");
  const int NSEG = segs;
  size_t xb = (size_t)N * H * W * C * 4, wb = (size_t)K * R * S * C * NSEG * 2;
  UNIT_CHECK_ARG(xb < 0xFFFFFFF0ull && wb < 0xFFFFFFF0ull, "conv_x3: operand larger than 4 GiB");
  SplitK sk{NSEG, NSEG == 3 ? 0x1 : 0x0, C, 2 * C};          // segments [lo.Wh, hi.Wh, hi.Wl] | [hi.Wh, hi.Wl]
  hipStream_t st = (hipStream_t)stream;
  ConvSecond sec;
  { int rc = unit_fill_second(sec, second, R, S, stride, pad, oy_mul, (size_t)C * 4); if (rc != UNIT_OK) return rc; }
  if (K == 0 || ((long)N * OH * OW == 0 && !sec.on)) return UNIT_OK;
  if (tile >= 0) {
    ConvDmaArgs a;
    a.second = sec;
    a.sk = sk; a.mask_pitch = 2 * mask_c;
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.residual = residual; a.mask_ref = mask_ref;
    a.N = N; a.H = H; a.W = W; a.C = NSEG * C; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad;
    a.OH = OH; a.OW = OW; a.ldy = ldy; a.oy_mul = oy_mul; a.OHf = OHf; a.OWf = OWf; a.relu = relu;
    a.Kgemm = R * S * NSEG * C; a.M = N * OH * OW;
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
    return unit_conv_mid_x3_launch(a, tile, st);
  }
  Conv256Args a;
  a.second = sec;
  a.sk = sk; a.mask_pitch = 2 * mask_c;
  a.x = x; a.w = w; a.y = y; a.bias = bias; a.residual = residual; a.mask_ref = mask_ref;
  a.N = N; a.H = H; a.W = W; a.C = NSEG * C; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad;
  a.OH = OH; a.OW = OW; a.ldy = ldy; a.oy_mul = oy_mul; a.OHf = OHf; a.OWf = OWf; a.relu = relu;
  a.Kgemm = R * S * NSEG * C; a.M = N * OH * OW;
  a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
  a.ex = EpiExtra{nullptr, nullptr, nullptr, 0}; a.ex_on = 0;
  a.x2 = nullptr; a.x2_bytes = 0; a.cb_split = 0; a.ratio2 = 1; a.pm_ncls = 0;
  set_div_magics(a);
  // position-class tiles (3x3 s1 p1 on a small map: the k-tiles of all-padding taps are skipped) as in unit_conv2d_fwd_big
  if (!sec.on && R == 3 && S == 3 && stride == 1 && pad == 1 && OH == H && OW == W && oy_mul == 1 && OHf == OH && OWf == OW && H * W <= 4096 &&
      (size_t)N * H * W * ldy * 4 < 0xFFFFFFF0ull)
    build_position_classes(a);
  return unit_conv256_p8_launch(a, UNIT_BF16, true, false, st);
}

// Two problems of ONE layer in one grid (conv_epilogue.h ConvSecond): the supervised and the weak batch of a training step, each zero-padded to
// its own largest image (meta_arch/rcnn.py:438-452). kernel: 0 = unit_conv2d_fwd (tile = tile_cfg), 1 = unit_conv2d_fwd_mid (tile), 2 =
// unit_conv2d_fwd_big (tile = variant 0 / 8 / 12; row-major tiles), 3 = unit_conv2d_fwd_x3 (tile). Results per problem: bit-identical to the
// single launches (same tiles, same k order).
extern "C" int unit_conv2d_fwd_pair(int kernel, const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref,
                                    int mask_c, int in_dtype, int out_dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                                    int OH, int OW, int ldy, int oy_mul, int OHf, int OWf, int relu, int tile, const UnitConvSecond* second,
                                    void* stream) {
  UNIT_CHECK_ARG(second != nullptr, "conv_pair: no second problem");
  switch (kernel) {
    case 0: return unit_conv_generic_impl(x, w, y, bias, residual, mask_ref, in_dtype, out_dtype, N, H, W, C, K, R, S, stride, pad, OH, OW, ldy, oy_mul, OHf,
                                          OWf, relu, tile, second, stream);
    case 1: return unit_conv_mid_impl(x, w, y, bias, residual, mask_ref, out_dtype, N, H, W, C, K, R, S, stride, pad, OH, OW, ldy, oy_mul, OHf, OWf, relu, tile,
                                      second, stream);
    case 2: return unit_conv_big_impl(x, w, y, bias, residual, mask_ref, out_dtype, N, H, W, C, K, R, S, stride, pad, OH, OW, ldy, oy_mul, OHf, OWf, relu, tile,
                                      second, stream);
    case 3: return unit_conv_x3_impl(x, w, y, bias, residual, mask_ref, mask_c, N, H, W, C, K, R, S, stride, pad, OH, OW, ldy, oy_mul, OHf, OWf, relu, tile,
                                     second, stream, 3);
    case 4: return unit_conv_x3_impl(x, w, y, bias, residual, mask_ref, mask_c, N, H, W, C, K, R, S, stride, pad, OH, OW, ldy, oy_mul, OHf, OWf, relu, tile,
                                     second, stream, 2);
  }
  unit_set_error("conv_pair: kernel 0 (generic), 1 (mid), 2 (big), 3 (bf16x3), 4 (bf16x3, two segments)");
  return UNIT_ERR_ARG;
}


// ---------------------------------------------------------------------------------------------------------------------
// Conv with the extended epilogue (conv_epilogue.h EpiExtra): 1x1 / 3x3 stride-1 bf16 conv on the phase-interleaved 256x256
// kernel with any of  (a) relu_bits out: one bit per output element, (b) mask_bits in instead of a bf16 mask tensor,
// (c) pool_partial: global average pool over `pool_rows` consecutive output rows fused into the epilogue; y may then be null.
// Reference: box_head.py:80 (x.mean(dim=[2,3]) after the last Res5 block) and the ReLU backward of the Bottleneck outputs.
extern "C" size_t unit_conv_pool_partial_floats(int M, int ldy) { return (size_t)cdiv(M, 128) * 4 * (size_t)ldy; }

extern "C" int unit_conv2d_fwd_big_ex(const void* x, const void* w, void* y, const float* bias, const void* residual,
                                      const unsigned char* mask_bits, unsigned char* relu_bits, float* pool_partial, int pool_rows,
                                      int N, int H, int W, int C, int K, int R, int S, int pad, int ldy, int relu, const void* x2, int C2,
                                      int variant, void* stream) {
  UNIT_CHECK_ARG(variant == 0 || variant == 8 || variant == 11, "conv_big_ex: variant 0 (default), 8 (16x16x32 MFMA) or 11 (32x32x16 MFMA)");
  // x2 != NULL: 1x1 conv over the channel concatenation [x (C channels) | x2 (C2 channels)] of two tensors of the same N, H, W;
  // w = [K][C + C2]. C2 must be a multiple of C.
  UNIT_CHECK_ARG(x2 == nullptr || (R == 1 && S == 1 && pad == 0 && C2 > 0 && C2 % C == 0 && ((uintptr_t)x2 % 16 == 0)),
                 "conv_big_ex: a second input needs a 1x1 conv and C2 a multiple of C");
  UNIT_CHECK_ARG(C % 64 == 0 && (x2 == nullptr || C2 % 64 == 0), "conv_big_ex: C must be a multiple of 64");
  UNIT_CHECK_ARG(ldy % 8 == 0 && ldy >= K, "conv_big_ex: ldy must be a multiple of 8 and >= K");
  UNIT_CHECK_ARG(y != nullptr || pool_partial != nullptr, "conv_big_ex: no output requested");
  UNIT_CHECK_ARG((mask_bits == nullptr && relu_bits == nullptr) || ldy % 64 == 0, "conv_big_ex: bit masks need ldy % 64 == 0");
  UNIT_CHECK_ARG(pool_partial == nullptr || pool_rows >= 44, "conv_big_ex: a 128-row wave tile must span at most 4 pooling segments");
  UNIT_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)y % 16 == 0), "conv_big_ex: 16B alignment");
  int OH = H + 2 * pad - R + 1, OW = W + 2 * pad - S + 1;
  UNIT_CHECK_ARG(OH > 0 && OW > 0, "conv_big_ex: empty output");
  Conv256Args a;
  a.sk = SplitK{0, 0, 0, 0}; a.mask_pitch = 0; a.second.on = 0;
  a.x = x; a.w = w; a.y = y; a.bias = bias; a.residual = residual; a.mask_ref = nullptr;
  const int Ct = C + (x2 ? C2 : 0);
  a.N = N; a.H = H; a.W = W; a.C = Ct; a.K = K; a.R = R; a.S = S; a.stride = 1; a.pad = pad;
  a.OH = OH; a.OW = OW; a.ldy = ldy; a.oy_mul = 1; a.OHf = OH; a.OWf = OW; a.relu = relu;
  a.Kgemm = R * S * Ct; a.M = N * OH * OW;
  size_t xb = (size_t)N * H * W * C * 2, wb = (size_t)K * R * S * Ct * 2, x2b = x2 ? (size_t)N * H * W * C2 * 2 : 0;
  UNIT_CHECK_ARG(xb < 0xFFFFFFF0ull && wb < 0xFFFFFFF0ull && x2b < 0xFFFFFFF0ull, "conv_big_ex: operand larger than 4 GiB");
  a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
  a.x2 = x2; a.x2_bytes = (unsigned)x2b; a.cb_split = C / 64; a.ratio2 = x2 ? C2 / C : 1; a.pm_ncls = 0;
  set_div_magics(a);
  a.ex = EpiExtra{relu_bits, mask_bits, pool_partial, pool_rows}; a.ex_on = 1;
  if (a.M == 0 || K == 0) return UNIT_OK;
  if (variant == 11 || (variant == 0 && unit_conv256_use_m32())) return unit_conv256_p8m_launch(a, UNIT_BF16, (hipStream_t)stream);
  return unit_conv256_p8_launch(a, UNIT_BF16, true, false, (hipStream_t)stream);
}

// pooled[r][n] = (1 / rows) * sum of the partial sums of RoI r: its rows [r*rows, (r+1)*rows) lie in at most two 128-row wave tiles
template <typename T>
__global__ void pool_finish_kernel(const float* __restrict__ part, int R, int rows, int ldy, int K, T* __restrict__ out, int ldo) {
  int r = blockIdx.x;
  int w0 = (r * rows) / 128, w1 = (r * rows + rows - 1) / 128;
  float inv = (float)rows;                     // divided, not multiplied by a reciprocal: unit_global_avgpool_fwd's arithmetic
  int seg0 = r - (w0 * 128) / rows, seg1 = r - (w1 * 128) / rows;
  const float* p0 = part + ((size_t)w0 * 4 + seg0) * ldy;
  const float* p1 = part + ((size_t)w1 * 4 + seg1) * ldy;
  // eight consecutive channels per thread: the two 32-byte loads of each of the (at most two) pieces are issued before anything is used
  // (one channel per thread and loop iteration was eight dependent round trips: 27 us for 1024 RoIs x 2048 channels)
  if ((K & 7) == 0 && (ldy & 3) == 0) {
    for (int n = threadIdx.x * 8; n < K; n += blockDim.x * 8) {
      f32x4 a0 = *reinterpret_cast<const f32x4*>(p0 + n), a1 = *reinterpret_cast<const f32x4*>(p0 + n + 4);
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
      if (w1 != w0) { b0 = *reinterpret_cast<const f32x4*>(p1 + n); b1 = *reinterpret_cast<const f32x4*>(p1 + n + 4); }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = 0.f;
        t += j < 4 ? a0[j] : a1[j - 4];
        if (w1 != w0) t += j < 4 ? b0[j] : b1[j - 4];
        out[(size_t)r * ldo + n + j] = (T)(t / inv);
      }
    }
    return;
  }
  for (int n = threadIdx.x; n < K; n += blockDim.x) {
    float t = 0.f;
    for (int wt = w0; wt <= w1; ++wt) {
      int seg = r - (wt * 128) / rows;
      t += part[((size_t)wt * 4 + seg) * ldy + n];
    }
    out[(size_t)r * ldo + n] = (T)(t / inv);
  }
}

extern "C" int unit_pool_finish(const float* partial, int R, int rows, int ldy, int K, void* out, int ldo, int out_dtype, void* stream) {
  UNIT_CHECK_ARG(rows >= 44 && rows <= 128, "pool_finish: 44 <= rows <= 128");
  if (R == 0) return UNIT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == UNIT_BF16) pool_finish_kernel<bf16_t><<<R, 256, 0, st>>>(partial, R, rows, ldy, K, (bf16_t*)out, ldo);
  else pool_finish_kernel<float><<<R, 256, 0, st>>>(partial, R, rows, ldy, K, (float*)out, ldo);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// g[m][n] = bit(m0 + m, n) ? dfeat[m / rows][n] / rows : 0 -- backward of (global average pool o ReLU) from the bit mask the fused
// forward left behind (replaces unit_global_avgpool_bwd_relu's read of the whole output map). m0 = first row of this slice inside
// the map the bits were written for (the weak head backpropagates only its weak RoIs).
__global__ void avgpool_bwd_bits_kernel(const bf16_t* __restrict__ dfeat, const unsigned char* __restrict__ bits, long R, long m0, int rows,
                                        int C, bf16_t* __restrict__ g) {
  // one lane = 8 channels of SEVEN consecutive bins of one RoI: one IEEE division per channel for the seven outputs (the kernel was
  // VALU-heavy with one per output element), seven bit bytes in flight, then seven non-temporal 16-byte stores (the 205 MB map is
  // written once and read by later kernels: 73 -> 5x us at the Res5 size, tools/avgpool_bits_bench.py)
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cb = C / 8, ng = (rows + 6) / 7;
  if (i >= R * ng * cb) return;
  long t = i / cb; int c0 = (int)(i - t * cb) * 8;
  long roi = t / ng; int r0 = (int)(t - roi * ng) * 7;
  bf16x8 d = *reinterpret_cast<const bf16x8*>(dfeat + roi * C + c0);
  float inv = (float)rows;
  bf16x8 q;
#pragma unroll
  for (int j = 0; j < 8; ++j) q[j] = (bf16_t)((float)d[j] / inv);      // = unit_global_avgpool_bwd_relu's arithmetic
  const bf16_t zero = (bf16_t)0.f;
  unsigned b[7];
#pragma unroll
  for (int u = 0; u < 7; ++u) {
    long word; int bit;
    relu_bit_index(m0 + roi * rows + min(r0 + u, rows - 1), c0, C, word, bit);
    b[u] = bits[word * 16 + (bit >> 3)];
  }
#pragma unroll
  for (int u = 0; u < 7; ++u) {
    if (r0 + u >= rows) break;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = ((b[u] >> j) & 1u) ? q[j] : zero;
    typedef __attribute__((ext_vector_type(4))) int i4;
    __builtin_nontemporal_store(__builtin_bit_cast(i4, o), reinterpret_cast<i4*>(g + (roi * rows + r0 + u) * C + c0));
  }
}

extern "C" size_t unit_relu_bits_bytes(int M, int ldy) { return (size_t)cdiv(M, 128) * 128 * (size_t)(ldy / 8); }

extern "C" int unit_avgpool_bwd_bits(const void* dfeat, const unsigned char* bits, int R, int roi_offset, int rows, int C, void* g, void* stream) {
  UNIT_CHECK_ARG(C % 64 == 0, "avgpool_bwd_bits: C % 64");
  long M = (long)R * rows;
  if (M == 0) return UNIT_OK;
  long n = (long)R * ((rows + 6) / 7) * (C / 8);
  avgpool_bwd_bits_kernel<<<(unsigned)cdiv(n, 256L), 256, 0, (hipStream_t)stream>>>((const bf16_t*)dfeat, bits, (long)R, (long)roi_offset * rows, rows, C,
                                                                                    (bf16_t*)g);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
