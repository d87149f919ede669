// conv_wgrad256p8.hip -- the 256 (k = (r,s,c)) x 256 (n = out channel) weight-gradient tile of conv_wgrad256.hip with the
// schedule of conv_igemm256p8.hip: every 64-pixel step is cut into FOUR phases of 16 MFMAs (one 64 x 32 quadrant of the
// wave's 128 x 64 tile, both 32-pixel sub-steps), the two operand tiles are cut into FOUR 16 KB half-tiles staged one per
// phase under a counted vmcnt, the two wave groups (waves 0-3 / 4-7) run half a phase apart, and the transposing fragment
// reads of phase p+1 are issued inside the MFMA section of phase p.
//
//   dW[n][k] = sum_m dy[m][n] * im2col(x)[m][k]     (contract, split-M slab layout and epilogue: conv_wgrad256.hip)
//
// LDS (128 KB): buffer d = step & 1, slots [X0 | D0 | D1 | X1], each 64 pixel rows x 256 B (128 columns):
//   X half q holds columns  wkr*128 + q*64 + i  of the 256-column x tile at half-column wkr*64 + i   (wkr = 0,1 ; i < 64)
//   D half q holds columns  wnr*64  + q*32 + j  of the 256-column dy tile at half-column wnr*32 + j  (wnr = 0..3 ; j < 32)
// so wave (wk, wn) still owns the contiguous 128 (k) x 64 (n) block. One LDS-DMA piece = 4 rows x 256 B; the 32-B column
// blocks of a row are XOR-swizzled with (row & 7) on the source side (8 blocks per row: a half-wave of ds_read_b64_tr_b16
// touches 8 rows x 32 B on 8 different bank groups).
// Schedule of step t (d = t & 1), fragments fx / fxb (x quadrant columns) and fd0 / fd1 (dy):
//   phase 0: stage X1(t+1) ; quadrant (0,0) = fx  x fd0 || read D1(t) -> fd1
//   phase 1: stage X0(t+2) ; quadrant (0,1) = fx  x fd1 || read X1(t) -> fxb
//   phase 2: stage D0(t+2), vmcnt wait (step t+1 landed) ; quadrant (1,0) = fxb x fd0
//   phase 3: stage D1(t+2) ; quadrant (1,1) = fxb x fd1 || read X0(t+1) -> fx, D0(t+1) -> fd0
// WAR / RAW distances are those of conv_igemm256p8.hip (reads retired by lgkmcnt(0) before the barrier that ends their
// MFMA section; a slot is restaged >= 2 phases after its last read; first read of step t+1 one phase after both groups'
// vmcnt wait; UNIT_P8_FINE_WAIT as there). Same m permutation inside a fragment and same accumulation order as conv_wgrad256.hip: bit-identical slabs.
#include "conv_wgrad256.h"
#ifndef UNIT_W8_AUX
#define UNIT_W8_AUX 0          // cache policy of the LDS-DMA loads (2 = nt; tools/exp_wait.sh)
#endif
#ifndef UNIT_W8_PER0
#define UNIT_W8_PER0 1          // transposing fragment reads per MFMA gap in phases 0 / 1 / 3 (tools/exp_wait.sh)
#define UNIT_W8_PER1 1
#define UNIT_W8_PER3 2
#endif
#ifndef UNIT_P8_FINE_WAIT
#define UNIT_P8_FINE_WAIT 0      // 1: one counted vmcnt wait per half-tile instead of one per k-tile / step (tools/exp_wait.sh: measured 1-8 % slower)
#endif

// one 256 x 256 tile of dW over the pixels [split * mps, +mps) of a contraction of Meff rows. vo (Wgrad256Args::valid_only): the rows are
// (image, valid output position of the tile's filter tap): positions = rows v_oh0.. of the map, columns v_ow0.. of v_cw, nv per image.
// TABN: entries of the pixel -> input offset table behind the operand stages (maps of up to TABN pixels avoid per-row divisions).
template <int TABN>
__device__ __forceinline__ void wgrad256_p8_tile(const Wgrad256Args& p, int tile_k, int tile_n, int split, const bool vo, int v_oh0, int v_ow0,
                                                 int v_cw, int nv, int Meff, int mps, char* smem) {
  constexpr int MS = 64;
  constexpr int HALF = MS * 256;               // 16 KB
  constexpr int SX0 = 0, SD0 = HALF, SD1 = 2 * HALF, SX1 = 3 * HALF, BUF = 4 * HALF;
  int k0 = tile_k * 256, n0 = tile_n * 256;
  int rs = k0 / p.C, ch0 = k0 - rs * p.C, kr = rs / p.S, ksx = rs - kr * p.S;
  int m_begin = min(Meff, split * mps), m_end = min(Meff, m_begin + mps);

  const bf16_t* __restrict__ X = (const bf16_t*)p.x;
  const bf16_t* __restrict__ DY = (const bf16_t*)p.dy;
  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(X), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(DY), 0, (int)p.dy_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;

  int tid = threadIdx.x, lane = tid & 63;
  int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wk = wid >> 2, wn = wid & 3;
  const int grp = wk;
  const bool pointwise = (p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0);

  // staging: piece j (0,1) of this wave in every half-tile = rows (j*8 + wid)*4 .. +4 ; lane -> row + (lane>>4), physical
  // 16-B chunk lane&15 ; logical chunk = (32-B block XOR (row & 7), 16-B half kept). (row & 7) does not depend on j.
  int s_row[2]; unsigned xcol[2], dcol[2];
  {
    int r0 = wid * 4 + (lane >> 4);
    s_row[0] = r0; s_row[1] = 32 + r0;
    int jp = lane & 15;
    int hcol = ((((jp >> 1) ^ (r0 & 7)) << 1) | (jp & 1)) * 8;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      xcol[q] = (unsigned)(ch0 + (hcol >> 6) * 128 + q * 64 + (hcol & 63));
      dcol[q] = (unsigned)(n0 + (hcol >> 5) * 64 + q * 32 + (hcol & 31));
    }
  }
  // im2col rows without per-step divisions: the (n, pixel-in-image) pair of each of this lane's two piece rows is carried
  // incrementally (+64 pixels per step) and a small LDS table maps the pixel to its input offset for THIS workgroup's filter
  // tap ((ih*W + iw)*C, or -1 outside the map). Table after the operand stages (OHW <= 1024 entries; larger maps divide).
  int* tab = reinterpret_cast<int*>(smem + 2 * BUF);
  int* tabd = tab + 512;                          // valid_only: position index -> output pixel of the dy row (OHW <= 512 then)
  const bool use_tab = !pointwise && p.OHW <= TABN;
  int xn[2] = {0, 0}, xp[2] = {0, 0};
  const int adv_q = MS / nv, adv_r = MS - adv_q * nv;
  if (use_tab) {
    for (int px = tid; px < nv; px += 512) {
      int oh = px / v_cw, ow = px - oh * v_cw;
      oh += v_oh0; ow += v_ow0;
      int ih = oh * p.stride - p.pad + kr, iw = ow * p.stride - p.pad + ksx;
      tab[px] = ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? (ih * p.W + iw) * p.x_pitch : -1;
      if (vo) tabd[px] = oh * p.OW + ow;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned um = (unsigned)(m_begin + s_row[j]);
      unsigned n = um / (unsigned)nv;
      xn[j] = (int)n; xp[j] = (int)(um - n * (unsigned)nv);
    }
    __syncthreads();
  }
  auto x_advance = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      xn[j] += adv_q; xp[j] += adv_r;
      if (xp[j] >= nv) { xp[j] -= nv; xn[j] += 1; }
    }
  };
  auto stage_x = [&](int q, int d, int mstep) {
    char* base = smem + d * BUF + (q ? SX1 : SX0);
    int tv[2] = {0, 0};
    if (use_tab) {
      // (asm: a plain LDS load here would make hipcc drain vmcnt in front of it, conv_wgrad256.h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(tab + xp[j]);
        asm volatile("ds_read_b32 %0, %1" : "=v"(tv[j]) : "v"(a) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tv[0]), "+v"(tv[1]) :: "memory");
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int m = mstep + s_row[j];
      bool ok = m < m_end;
      unsigned xoff;
      if (pointwise) xoff = ((unsigned)m * (unsigned)p.x_pitch + xcol[q]) * 2u;
      else if (use_tab) {
        ok = ok && tv[j] >= 0;
        xoff = ((unsigned)xn[j] * (unsigned)(p.H * p.W * p.x_pitch) + (unsigned)tv[j] + xcol[q]) * 2u;
      } else {
        unsigned um = (unsigned)m, n, oh, ow;
        if (p.use_magic) {
          n = __umulhi(um, p.magic_ohw); unsigned rem = um - n * (unsigned)p.OHW;
          if (rem >= (unsigned)p.OHW) { rem -= p.OHW; ++n; }
          oh = __umulhi(rem, p.magic_ow); ow = rem - oh * (unsigned)p.OW;
          if (ow >= (unsigned)p.OW) { ow -= p.OW; ++oh; }
        } else {
          ow = um % (unsigned)p.OW; unsigned t2 = um / (unsigned)p.OW; oh = t2 % (unsigned)p.OH; n = t2 / (unsigned)p.OH;
        }
        int ih = (int)oh * p.stride - p.pad + kr, iw = (int)ow * p.stride - p.pad + ksx;
        ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        xoff = ((unsigned)n * (unsigned)(p.H * p.W * p.x_pitch) + (unsigned)((ih * p.W + iw) * p.x_pitch) + xcol[q]) * 2u;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_w*)(base + (j * 8 + wid) * 1024), 16, ok ? xoff : OOB, 0, 0, UNIT_W8_AUX);
    }
  };
  auto stage_d = [&](int q, int d, int mstep) {
    char* base = smem + d * BUF + (q ? SD1 : SD0);
    int tv[2] = {0, 0};
    if (vo) {                                     // the dy row of (image xn, valid position xp): table lookup, as stage_x
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(tabd + xp[j]);
        asm volatile("ds_read_b32 %0, %1" : "=v"(tv[j]) : "v"(a) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tv[0]), "+v"(tv[1]) :: "memory");
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int m = mstep + s_row[j];
      unsigned doff = ((unsigned)(vo ? xn[j] * p.OHW + tv[j] : m) * (unsigned)p.ldy + dcol[q]) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (lds_void_w*)(base + (j * 8 + wid) * 1024), 16, m < m_end ? doff : OOB, 0, 0, UNIT_W8_AUX);
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addressing (tr_frag of conv_wgrad256.hip on 256-B rows): lane l: g = l>>4, i = l&15 = 4q+pq reads rows
  // 32*sub + 16h + 4g + q, 8 B at column block (col>>4) ^ (row & 7); element j = 4h+q' of lane i <-> m-row 32*sub+16h+4g+q'
  int offx[4], offd[2];
  {
    int g = lane >> 4, i = lane & 15, q4 = i >> 2, pq = i & 3;
    int rowl = 4 * g + q4, r7 = rowl & 7;
    int basel = rowl * 256 + 8 * pq;
#pragma unroll
    for (int a = 0; a < 4; ++a) offx[a] = basel + (((wk * 4 + a) ^ r7) << 5);
#pragma unroll
    for (int b = 0; b < 2; ++b) offd[b] = basel + (((wn * 2 + b) ^ r7) << 5);
  }
  auto frag = [&](const char* half, int off, int sub) -> bf16x8 {
    const char* a0 = half + off + sub * 8192;
    s16x4 lo = ds_tr16(a0);
    s16x4 hi = ds_tr16(a0 + 4096);
    s16x8_w v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  bf16x8 fx[4][2], fxb[4][2], fd0[2][2], fd1[2][2];
#define W8_READ_X(HALFP, FX)                                                          \
  do {                                                                                \
    _Pragma("unroll") for (int a = 0; a < 4; ++a)                                     \
      _Pragma("unroll") for (int sub = 0; sub < 2; ++sub) FX[a][sub] = frag(HALFP, offx[a], sub); \
  } while (0)
#define W8_READ_D(HALFP, FD)                                                          \
  do {                                                                                \
    _Pragma("unroll") for (int b = 0; b < 2; ++b)                                     \
      _Pragma("unroll") for (int sub = 0; sub < 2; ++sub) FD[b][sub] = frag(HALFP, offd[b], sub); \
  } while (0)
#define W8_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#if defined(UNIT_DBGW8) && (UNIT_DBGW8 & 4)
#define W8_FMA(ACC, A, B) do { if (b == 0) ACC += __builtin_bit_cast(f32x4, A); if (a == 0) ACC += __builtin_bit_cast(f32x4, B); } while (0)
#else
#define W8_FMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, ACC, 0, 0, 0)
#endif
#if defined(UNIT_DBGW8) && (UNIT_DBGW8 & 8)
#define W8_RD(stmt) do { if (t == 0) { stmt; } } while (0)
#else
#define W8_RD(stmt) do { stmt; } while (0)
#endif
  // MFMA section: quadrant (QX, QN) = acc[QX*4 ..][QN*2 ..]; NR ds_read_b64_tr_b16 of READS spread PER per MFMA gap
#define W8_MM(QX, QN, FX, FD, NR, PER, READS)                                          \
  do {                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                     \
    READS;                                                                             \
    _Pragma("unroll") for (int sub = 0; sub < 2; ++sub)                                \
      _Pragma("unroll") for (int a = 0; a < 4; ++a)                                    \
        _Pragma("unroll") for (int b = 0; b < 2; ++b)                                  \
          W8_FMA(acc[(QX) * 4 + a][(QN) * 2 + b], FX[a][sub], FD[b][sub]);             \
    _Pragma("unroll") for (int i = 0; i < (NR) / ((PER) > 0 ? (PER) : 1); ++i) {      \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                               \
      __builtin_amdgcn_sched_group_barrier(0x100, (PER), 0);                           \
    }                                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x008, 16 - (NR) / ((PER) > 0 ? (PER) : 1), 0); \
    __builtin_amdgcn_s_setprio(0);                                                     \
    if ((NR) > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   \
  } while (0)

  const int nsteps = (m_end - m_begin + MS - 1) / MS;
  if (nsteps > 0) {
    int mst = m_begin;
    stage_x(0, 0, mst); stage_d(0, 0, mst); stage_d(1, 0, mst); stage_x(1, 0, mst);
    mst += MS; x_advance();
    if (nsteps > 1) {
      stage_x(0, 1, mst); stage_d(0, 1, mst); stage_d(1, 1, mst);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    W8_BAR();
    W8_READ_D(smem + SD0, fd0);
    W8_READ_X(smem + SX0, fx);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (grp == 1) W8_BAR();
    for (int t = 0; t < nsteps; ++t) {
      const int d = t & 1;
      const char* buf = smem + d * BUF;
      const char* bnx = smem + (d ^ 1) * BUF;
#if defined(UNIT_DBGW8) && (UNIT_DBGW8 & 2)
      const bool n1 = false, n2 = false;
#else
      const bool n1 = t + 1 < nsteps, n2 = t + 2 < nsteps;
#endif
      // phase 0
      if (n1) stage_x(1, d ^ 1, mst);
      mst += MS; x_advance();
#if UNIT_P8_FINE_WAIT
      if (n1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // X1(t) landed (read in M(t, 1)); X0, D0, D1, X1 of t+1 younger
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      W8_BAR();
      W8_MM(0, 0, fx, fd0, 8, UNIT_W8_PER0, W8_RD(W8_READ_D(buf + SD1, fd1)));
      W8_BAR();
      // phase 1
      if (n2) stage_x(0, d, mst);
      W8_BAR();
      W8_MM(0, 1, fx, fd1, 16, UNIT_W8_PER1, W8_RD(W8_READ_X(buf + SX1, fxb)));
      W8_BAR();
      // phase 2
#if UNIT_P8_FINE_WAIT
      if (n2) {
        stage_d(0, d, mst);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // X0, D0 of t+1 landed (read in M(t, 3)); D1, X1 of t+1, X0, D0 of t+2 younger
      } else if (n1) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");             // D1, X1 of t+1 younger
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
#else
      if (n2) {
        stage_d(0, d, mst);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
#endif
      W8_BAR();
      W8_MM(1, 0, fxb, fd0, 0, 0, (void)0);
      W8_BAR();
      // phase 3 (after the last step the reads fetch stale, in-bounds LDS that nobody uses)
      if (n2) stage_d(1, d, mst);
#if UNIT_P8_FINE_WAIT
      if (n2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // D1(t+1) landed (read in M(t+1, 0)); X1(t+1), X0, D0, D1 of t+2 younger
      else if (n1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // X1(t+1) younger
#endif
      W8_BAR();
      W8_MM(1, 1, fxb, fd1, 24, UNIT_W8_PER3, W8_RD(W8_READ_D(bnx + SD0, fd0); W8_READ_X(bnx + SX0, fx)));
      W8_BAR();
    }
    if (grp == 0) W8_BAR();
  }
#undef W8_MM
#undef W8_FMA
#undef W8_RD
#undef W8_BAR
#undef W8_READ_D
#undef W8_READ_X

  // epilogue: D[row = k][col = n] -> partial[split][n][k..k+3]
  float* out = p.partial + (size_t)split * p.K * p.Kgemm;
  int fq = lane >> 4, fr = lane & 15;
#if defined(UNIT_DBGW8) && (UNIT_DBGW8 & 1)
  // diagnostic build only (tools/exp_w8.sh): one store per lane instead of 32 -- what does the slab store cost?
  f32x4 ssum = acc[0][0];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int a = 0; a < 8; ++a) if (a + b) ssum += acc[a][b];
  *reinterpret_cast<f32x4*>(out + (size_t)(n0 + wn * 64 + fr) * p.Kgemm + k0 + wk * 128 + fq * 4) = ssum;
#else
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    int n = n0 + wn * 64 + b * 16 + fr;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      int k = k0 + wk * 128 + a * 16 + fq * 4;
#if UNIT_SLAB_NT
      __builtin_nontemporal_store(acc[a][b], reinterpret_cast<f32x4*>(out + (size_t)n * p.Kgemm + k));     // read back once, by a later kernel
#else
      *reinterpret_cast<f32x4*>(out + (size_t)n * p.Kgemm + k) = acc[a][b];
#endif
    }
  }
#endif
}


__global__ void __launch_bounds__(512, 2) conv_wgrad256_p8_kernel(Wgrad256Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware remap (workgroup b runs on XCD b % 8): every XCD gets a contiguous chunk of (split, tile_n, tile_k) ids, i.e.
  // the tiles of one or two split-M slabs. They walk the same pixel rows at the same pace, so an x / dy row block is
  // fetched from HBM once per XCD and then served to the other tiles of the slab out of that XCD's 4 MB L2.
  int bid = blockIdx.x;
  const bool vo = p.valid_only != 0;
  int tile_k, tile_n, split;
  // valid_only (Wgrad256Args): the contraction index of a tile runs over (image, valid output position of ITS filter tap) instead of
  // all pixels; valid positions = rows v_oh0 .. of the map, columns v_ow0 .. of v_cw (nv per image). The taps differ in work (36 / 42
  // / 49 positions on 7x7) and the grid is one round of workgroups, so with the same split count for every tap the centre tap would
  // set the time: the p.splits slabs include one spare, and 9 * (p.splits - 1) workgroup slots per (channel block, n tile) are dealt to
  // the taps in proportion to their positions (7 x 7, 8 slabs: corners 6, edges 7, centre 8 splits -> 6.0 / 6.0 / 6.1 positions per
  // split instead of 49 / 7). A workgroup past its tap's count ("filler") only leaves its slab tile zero.
  int v_oh0 = 0, v_ow0 = 0, v_cw = p.OW, nv = p.OHW, Meff = p.M, mps = p.m_per_split;
  if (!vo) {
    int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    tile_k = bid % p.tiles_k; int tt = bid / p.tiles_k;
    tile_n = tt % p.tiles_n; split = tt / p.tiles_n;
  } else {
    const int ntap = p.R * p.S, ncb = p.tiles_k / ntap;
    int nh[3], nw[3], sum_h = 0, sum_w = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      nh[t] = max(0, min(p.OH - 1, p.H - 1 + p.pad - t) - max(0, p.pad - t) + 1);
      nw[t] = max(0, min(p.OW - 1, p.W - 1 + p.pad - t) - max(0, p.pad - t) + 1);
      sum_h += nh[t]; sum_w += nw[t];
    }
    const int tot = sum_h * sum_w, slots = ntap * max(1, p.splits - 1);
    int stt[9], sum_st = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      int v = nh[t / 3] * nw[t % 3];
      stt[t] = max(1, min(p.splits, (slots * v + tot / 2) / tot));
      sum_st += stt[t];
    }
    // id map: real work first on every XCD (workgroup b runs on XCD b % 8 and the XCD's workgroups start in id order), fillers last;
    // real items ordered split-major (split j of every tap that has one, all n tiles and channel blocks: they walk about the same
    // pixel rows at the same pace -- shared through the XCD's L2), each XCD a contiguous run of them
    const int per = p.tiles_n * ncb;
    const int RI = per * sum_st, nwg = gridDim.x;
    const int xcd = bid % 8, loc = bid / 8;
    const int start_r = xcd * (RI / 8) + min(xcd, RI % 8), real_x = RI / 8 + (xcd < RI % 8 ? 1 : 0);
    const int start_t = xcd * (nwg / 8) + min(xcd, nwg % 8);
    int tap = 0, cb = 0;
    if (loc < real_x) {
      int L = start_r + loc, j = 0;
      for (;; ++j) {
        int act = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) act += stt[t] > j ? 1 : 0;
        if (L < per * act) {
          tile_n = L / (ncb * act); int rem = L - tile_n * (ncb * act);
          int a = rem / ncb; cb = rem - a * ncb;
#pragma unroll
          for (int t = 8; t >= 0; --t) { int before = 0; for (int u = 0; u < t; ++u) before += stt[u] > j ? 1 : 0; if (stt[t] > j && before == a) tap = t; }
          break;
        }
        L -= per * act;
      }
      split = j;
    } else {
      int z = (start_t - start_r) + (loc - real_x);
      tile_n = 0; split = p.splits;             // (overwritten below; z always lands in a tap)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        int cnt = (p.splits - stt[t]) * per;
        if (z >= 0 && z < cnt) {
          int jj = z / per, rem = z - jj * per;
          split = stt[t] + jj; tile_n = rem / ncb; cb = rem - tile_n * ncb; tap = t;
          z = -1;
        } else if (z >= 0) z -= cnt;
      }
    }
    tile_k = tap * ncb + cb;
    int kr_ = tap / p.S, ks_ = tap - kr_ * p.S;
    v_oh0 = max(0, p.pad - kr_); v_ow0 = max(0, p.pad - ks_);
    v_cw = nw[ks_]; nv = nh[kr_] * v_cw;
    Meff = p.N * nv;
    int st = stt[tap];
    mps = ((Meff + st - 1) / st + 63) / 64 * 64;
    if (split >= st) Meff = 0;
  }
  wgrad256_p8_tile<1024>(p, tile_k, tile_n, split, vo, v_oh0, v_ow0, v_cw, nv, Meff, mps, smem);
}

int unit_wgrad256_p8_launch(const Wgrad256Args& a, hipStream_t st) {
  size_t lds = 8 * 64 * 256 + 4096;      // operand stages + the pixel -> input offset table
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad256_p8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  conv_wgrad256_p8_kernel<<<a.tiles_k * a.tiles_n * a.splits, 512, lds, st>>>(a);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---- grouped launch (include/unit_hip.h: unit_conv2d_wgrad_group): the 256x256 tiles of SEVERAL layers in one grid. A unit = the tiles of
// one (layer, split) -- for valid_only layers of one (layer, filter tap, split): they contract over the same rows -- dealt to ONE XCD by
// the host (conv_wgrad.hip), longest first. With the layers of a Res5 head in one grid every layer needs 3-4 split-M slabs instead of
// 8-16 (launched alone, 16 tiles have to become 256 workgroups): a quarter of the slab traffic, 200-step loops; the res4 layers of a
// gradient bucket get 256x256 tiles at all (half the operand bytes per flop of the 128x128 ring kernel).
__device__ __forceinline__ int pinw(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const void* pinw_ptr(const void* q) {
  unsigned long long u = (unsigned long long)(uintptr_t)q;
  unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return (const void*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}

__global__ void __launch_bounds__(512, 2) conv_wgrad256_group_kernel(WgradGroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int nu = g.n_units[xcd];
  if (slot >= (int)g.unit_start[xcd][nu]) return;
  int u = 0;
  for (int i = 1; i < nu; ++i)
    if (slot >= (int)g.unit_start[xcd][i]) u = i;
  unsigned code = g.unit_code[xcd][u];
  const Wgrad256Args& src = g.p[code & 31];     // into SGPRs once (conv_wgrad128r.hip)
  Wgrad256Args p;
  p.x = pinw_ptr(src.x); p.dy = pinw_ptr(src.dy); p.partial = (float*)pinw_ptr(src.partial);
  p.N = pinw(src.N); p.H = pinw(src.H); p.W = pinw(src.W); p.C = pinw(src.C); p.K = pinw(src.K); p.R = pinw(src.R); p.S = pinw(src.S);
  p.stride = pinw(src.stride); p.pad = pinw(src.pad); p.OH = pinw(src.OH); p.OW = pinw(src.OW); p.ldy = pinw(src.ldy);
  p.Kgemm = pinw(src.Kgemm); p.M = pinw(src.M); p.tiles_k = pinw(src.tiles_k); p.tiles_n = pinw(src.tiles_n); p.splits = pinw(src.splits);
  p.m_per_split = pinw(src.m_per_split); p.x_bytes = (unsigned)pinw((int)src.x_bytes); p.dy_bytes = (unsigned)pinw((int)src.dy_bytes);
  p.magic_ohw = (unsigned)pinw((int)src.magic_ohw); p.magic_ow = (unsigned)pinw((int)src.magic_ow); p.OHW = pinw(src.OHW);
  p.use_magic = pinw(src.use_magic); p.valid_only = pinw(src.valid_only); p.x_pitch = pinw(src.x_pitch);
  int t = slot - (int)g.unit_start[xcd][u] + (int)g.unit_tile0[xcd][u];
  int tap = (int)((code >> 5) & 15), split = (int)(code >> 9);
  if (p.valid_only) {
    int ncb = p.tiles_k / (p.R * p.S);
    int kr_ = tap / p.S, ks_ = tap - kr_ * p.S;
    int nh = max(0, min(p.OH - 1, p.H - 1 + p.pad - kr_) - max(0, p.pad - kr_) + 1);
    int nw = max(0, min(p.OW - 1, p.W - 1 + p.pad - ks_) - max(0, p.pad - ks_) + 1);
    int Meff = p.N * nh * nw;
    int mps = ((Meff + p.splits - 1) / p.splits + 63) / 64 * 64;
    int tile_n = t / ncb, cb = t - tile_n * ncb;
    wgrad256_p8_tile<4096>(p, tap * ncb + cb, tile_n, split, true, max(0, p.pad - kr_), max(0, p.pad - ks_), nw, nh * nw, Meff, mps, smem);
  } else {
    wgrad256_p8_tile<4096>(p, t % p.tiles_k, t / p.tiles_k, split, false, 0, 0, p.OW, p.OHW, p.M, p.m_per_split, smem);
  }
}

int unit_wgrad256_group_launch(const WgradGroupArgs& g, int slots_per_xcd, hipStream_t st) {
  size_t lds = 8 * 64 * 256 + 4 * 4096;      // operand stages + a 4096-entry pixel -> input offset table
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad256_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  conv_wgrad256_group_kernel<<<slots_per_xcd * 8, 512, lds, st>>>(g);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
