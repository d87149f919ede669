// conv_wgrad128r.hip -- the 128 (k) x 128 (n) weight-gradient tile of conv_wgrad.hip for bf16 layers with C % 128 == 0 and
// K % 128 == 0 (every trainable backbone / RPN conv), staged by LDS-DMA through a RING of four 32-pixel stages under a counted
// vmcnt instead of through registers with one step of prefetch.
//
// Why: conv_wgrad_kernel issues the global loads of step t+1, multiplies step t (32 MFMAs per wave, ~0.2 us) and then needs
// the loaded registers for its ds_write: with M = 9 576 pixels split over ~20 workgroups per tile a workgroup runs 6-8 steps
// and spends each of them waiting ~1.5 us for rows that come from HBM / Infinity Cache (the kernels measured 17-29 us for
// 5-11 GFLOP). Here three stages (48 KB) stay in flight per workgroup, two workgroups per CU (64 KB of LDS each).
// Layout per stage and operand: [32 pixel rows][256 B = 128 channels]; one LDS-DMA piece = 4 rows; the eight 32-B column
// blocks of a row are XOR-swizzled with (row & 7) on the source side (conv_wgrad256p8.hip). Transposing reads through
// inline asm (conv_wgrad256.h: the intrinsic makes hipcc drain vmcnt before it). im2col rows: pointwise layers index
// directly; otherwise the (image, oh, ow) triple of each staged row is carried incrementally (+32 pixels per stage; needs
// OW >= 32 -- the backbone / RPN maps are 63 .. 250 wide) or comes from magic-number division.
// Same fragment permutation, accumulation order, split-M slabs and epilogue as conv_wgrad_kernel: bit-identical results.
#include "conv_wgrad256.h"

__device__ __forceinline__ bf16x8 tr_frag128(const char* tile, int col0, int lane) {
  // rows 16h + 4g + q of a 32-row stage with 256-B rows, 8 B at column block (col >> 4) ^ (row & 7)
  int g = lane >> 4, i = lane & 15, q = i >> 2, pq = i & 3;
  int row = 4 * g + q;
  const char* a0 = tile + row * 256 + ((((col0 >> 4) ^ (row & 7)) << 5) + 8 * pq);
  s16x4 lo = ds_tr16(a0);
  s16x4 hi = ds_tr16(a0 + 16 * 256);
  s16x8_w v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// one 128 x 128 tile of dW over the pixels of one split (p may live in the kernel-argument segment: every field read is uniform)
template <int NS>
__device__ __forceinline__ void wgrad128_ring_tile(const Wgrad256Args& p, int tile_k, int tile_n, int split, char* smem) {
  constexpr int MS = 32;
  constexpr int TILE = MS * 256;               // 8 KB per operand per stage
  int k0 = tile_k * 128, n0 = tile_n * 128;
  int m_begin = split * p.m_per_split, m_end = min(p.M, m_begin + p.m_per_split);
  int rs = k0 / p.C, ch0 = k0 - rs * p.C, kr = rs / p.S, ksx = rs - kr * p.S;

  const bf16_t* __restrict__ X = (const bf16_t*)p.x;
  const bf16_t* __restrict__ DY = (const bf16_t*)p.dy;
  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(X), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(DY), 0, (int)p.dy_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;

  int tid = threadIdx.x, lane = tid & 63;
  int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wk = wid >> 1, wn = wid & 1;
  const bool pointwise = (p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0);
  const bool incremental = !pointwise && p.OW >= MS;

  // staging: wave `wid`, piece i (0..1) = stage rows (i*4 + wid)*4 .. +4 ; lane -> row + (lane>>4), physical 16-B chunk lane&15,
  // logical chunk = (32-B block XOR (row & 7), 16-B half kept)
  int s_row[2]; unsigned s_col;
  {
    int r0 = wid * 4 + (lane >> 4);
    s_row[0] = r0; s_row[1] = 16 + r0;
    int jp = lane & 15;
    s_col = (unsigned)((((jp >> 1) ^ (r0 & 7)) << 1) | (jp & 1)) * 8u;      // (row & 7) is the same for both pieces
  }
  // incremental im2col state of the two staged rows: image, output row, output column of pixel m_stage + s_row[i]
  int in_[2] = {0, 0}, ioh[2] = {0, 0}, iow[2] = {0, 0};
  if (incremental) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned um = (unsigned)(m_begin + s_row[i]);
      unsigned ow = um % (unsigned)p.OW, tt = um / (unsigned)p.OW;
      iow[i] = (int)ow; ioh[i] = (int)(tt % (unsigned)p.OH); in_[i] = (int)(tt / (unsigned)p.OH);
    }
  }
  int mst = m_begin;        // first pixel of the next stage to be issued
  auto stage = [&](int buf) {      // issues stage `mst` into ring slot `buf` and advances: 4 LDS-DMA pieces per wave
    char* bx = smem + buf * 2 * TILE;
    char* bd = bx + TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int R0 = (i * 4 + wid) * 4;
      int m = mst + s_row[i];
      bool mok = m < m_end;
      unsigned xoff;
      bool ok = mok;
      if (pointwise) xoff = ((unsigned)m * (unsigned)p.x_pitch + (unsigned)ch0 + s_col) * 2u;
      else {
        int n, oh, ow;
        if (incremental) { n = in_[i]; oh = ioh[i]; ow = iow[i]; }
        else {
          unsigned um = (unsigned)m, un, uoh, uow;
          if (p.use_magic) {
            un = __umulhi(um, p.magic_ohw); unsigned rem = um - un * (unsigned)p.OHW;
            if (rem >= (unsigned)p.OHW) { rem -= p.OHW; ++un; }
            uoh = __umulhi(rem, p.magic_ow); uow = rem - uoh * (unsigned)p.OW;
            if (uow >= (unsigned)p.OW) { uow -= p.OW; ++uoh; }
          } else {
            uow = um % (unsigned)p.OW; unsigned tt = um / (unsigned)p.OW; uoh = tt % (unsigned)p.OH; un = tt / (unsigned)p.OH;
          }
          n = (int)un; oh = (int)uoh; ow = (int)uow;
        }
        int ih = oh * p.stride - p.pad + kr, iw = ow * p.stride - p.pad + ksx;
        ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        xoff = ((unsigned)n * (unsigned)(p.H * p.W * p.x_pitch) + (unsigned)((ih * p.W + iw) * p.x_pitch + ch0) + s_col) * 2u;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_w*)(bx + R0 * 256), 16, ok ? xoff : OOB, 0, 0, 0);
      unsigned doff = ((unsigned)m * (unsigned)p.ldy + (unsigned)n0 + s_col) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (lds_void_w*)(bd + R0 * 256), 16, mok ? doff : OOB, 0, 0, 0);
    }
    mst += MS;
    if (incremental) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        iow[i] += MS;
        if (iow[i] >= p.OW) { iow[i] -= p.OW; ioh[i] += 1; if (ioh[i] >= p.OH) { ioh[i] = 0; in_[i] += 1; } }
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nst = (m_end - m_begin + MS - 1) / MS;     // 32-pixel stages = the register kernel's sub-steps, in the same order
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nst) stage(s);
  for (int st = 0; st < nst; ++st) {
    int younger = min(NS - 2, nst - 1 - st);
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (st + NS - 1 < nst) stage((st + NS - 1) % NS);
    const char* bx = smem + (st % NS) * 2 * TILE;
    const char* bd = bx + TILE;
    bf16x8 fa[4], fb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) fa[a] = tr_frag128(bx, wk * 64 + a * 16, lane);
#pragma unroll
    for (int b = 0; b < 4; ++b) fb[b] = tr_frag128(bd, wn * 64 + b * 16, lane);
    tr_wait(fa); tr_wait(fb);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
  }

  // epilogue: D[row = k][col = n] -> partial[split][n][k..k+3]
  float* out = p.partial + (size_t)split * p.K * p.Kgemm;
  int fq = lane >> 4, fr = lane & 15;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    int n = n0 + wn * 64 + b * 16 + fr;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int k = k0 + wk * 64 + a * 16 + fq * 4;
#if UNIT_SLAB_NT
      __builtin_nontemporal_store(acc[a][b], reinterpret_cast<f32x4*>(out + (size_t)n * p.Kgemm + k));     // read back once, by a later kernel
#else
      *reinterpret_cast<f32x4*>(out + (size_t)n * p.Kgemm + k) = acc[a][b];
#endif
    }
  }
}

template <int NS>
__global__ void __launch_bounds__(256, 2) conv_wgrad128_ring_kernel(Wgrad256Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bid = blockIdx.x;
  int tile_k = bid % p.tiles_k; int t = bid / p.tiles_k;
  wgrad128_ring_tile<NS>(p, tile_k, t % p.tiles_n, t / p.tiles_n, smem);
}

// ---- grouped launch: the weight gradients of SEVERAL layers in one grid (include/unit_hip.h: unit_conv2d_wgrad_group).
// Why: a res4 layer has M = 9 576 pixels and 16-36 tiles -- launched alone it needs ~17 split-M slabs to fill the chip, so a
// workgroup runs 9 steps between a cold start and a 64 KB slab store, and the reduction reads 17 slabs back (17-29 us per layer for
// 5-11 GFLOP, 100 launches per step). The layers of a gradient bucket (18 of them for six res4 blocks) have 408 tiles between them:
// one grid, ONE slab per layer, 150-step loops. A "unit" = (layer, split): its tiles read the same pixel rows, so a unit is dealt
// to ONE XCD (workgroup b runs on XCD b % 8) and its rows come through that XCD's L2 once; the host deals units to XCDs
// longest-first (conv_wgrad.hip).
__device__ __forceinline__ int pin(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const void* pin_ptr(const void* q) {
  unsigned long long u = (unsigned long long)(uintptr_t)q;
  unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return (const void*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}

template <int NS>
__global__ void __launch_bounds__(256, 2) conv_wgrad128_group_kernel(WgradGroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int nu = g.n_units[xcd];
  if (slot >= (int)g.unit_start[xcd][nu]) return;
  int u = 0;
  for (int i = 1; i < nu; ++i)
    if (slot >= (int)g.unit_start[xcd][i]) u = i;
  unsigned code = g.unit_code[xcd][u];
  // the layer's arguments into SGPRs ONCE: read in place, hipcc re-loads fields from the argument segment inside the pixel loop, and
  // every such s_load is followed by an lgkmcnt(0) wait that also drains the LDS reads in flight
  const Wgrad256Args& src = g.p[code & 31];
  Wgrad256Args p;
  p.x = pin_ptr(src.x); p.dy = pin_ptr(src.dy); p.partial = (float*)pin_ptr(src.partial);
  p.N = pin(src.N); p.H = pin(src.H); p.W = pin(src.W); p.C = pin(src.C); p.K = pin(src.K); p.R = pin(src.R); p.S = pin(src.S);
  p.stride = pin(src.stride); p.pad = pin(src.pad); p.OH = pin(src.OH); p.OW = pin(src.OW); p.ldy = pin(src.ldy);
  p.Kgemm = pin(src.Kgemm); p.M = pin(src.M); p.tiles_k = pin(src.tiles_k); p.tiles_n = pin(src.tiles_n); p.splits = pin(src.splits);
  p.m_per_split = pin(src.m_per_split); p.x_bytes = (unsigned)pin((int)src.x_bytes); p.dy_bytes = (unsigned)pin((int)src.dy_bytes);
  p.magic_ohw = (unsigned)pin((int)src.magic_ohw); p.magic_ow = (unsigned)pin((int)src.magic_ow); p.OHW = pin(src.OHW);
  p.use_magic = pin(src.use_magic); p.valid_only = 0; p.x_pitch = pin(src.x_pitch);
  int t = slot - (int)g.unit_start[xcd][u] + (int)g.unit_tile0[xcd][u];
  wgrad128_ring_tile<NS>(p, t % p.tiles_k, t / p.tiles_k, (int)(code >> 9), smem);
}

int unit_wgrad128_group_launch(const WgradGroupArgs& g, int slots_per_xcd, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad128_group_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 32 * 256);
    attr_set = true;
  }
  conv_wgrad128_group_kernel<4><<<slots_per_xcd * 8, 256, 4 * 2 * 32 * 256, st>>>(g);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

int unit_wgrad128_ring_launch(const Wgrad256Args& a, hipStream_t st) {
  // UNIT_WG128_NS=3: three stages (48 KB of LDS, three workgroups per CU) instead of four (64 KB, two per CU)
  static int ns = -1;
  if (ns < 0) { const char* e = getenv("UNIT_WG128_NS"); ns = e ? atoi(e) : 4; }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad128_ring_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 32 * 256);
    (void)hipFuncSetAttribute((const void*)conv_wgrad128_ring_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 2 * 32 * 256);
    attr_set = true;
  }
  if (ns == 3) conv_wgrad128_ring_kernel<3><<<a.tiles_k * a.tiles_n * a.splits, 256, 3 * 2 * 32 * 256, st>>>(a);
  else conv_wgrad128_ring_kernel<4><<<a.tiles_k * a.tiles_n * a.splits, 256, 4 * 2 * 32 * 256, st>>>(a);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
