// conv_igemm256p8.hip -- 256 (pixels) x 256 (channels) x 64 (k) implicit-GEMM conv tile, 8 waves (2 x 4, 128 x 64 each), with the
// k-tile cut into FOUR phases and the operand stage cut into FOUR 16 KB half-tiles that are staged one per phase, six
// phases ahead of their first read, under a COUNTED vmcnt (the LDS-DMA queue never drains inside the loop).
//
// Why (MI355X_MICROARCH.md cycle constants; tools/exp256.sh): one LDS-DMA piece costs the issuing wave 60-185 cycles and
// the two-stage kernel of conv_igemm256.hip issues the 8 pieces of a k-tile in one burst, waits vmcnt(0) for all of them
// at the next barrier and only then issues the next burst: operand feed 27 B/clk/CU, MFMA pipe idle ~55 % of the loop.
// Here every phase = {LOAD section: <= 12 ds_read_b128 + 2 LDS-DMA pieces} barrier {16 MFMA} barrier, the two wave groups
// (waves 0-3 / 4-7, one of each per SIMD) run half a phase apart so that a SIMD always has one wave in its MFMA section
// while the other issues loads, and `s_waitcnt vmcnt(4)` once per k-tile leaves two half-tiles in flight across it.
//
// LDS (128 KB): buffer d = k-tile & 1, four half-tile slots [X0 | W0 | W1 | X1] of 128 rows x 128 B each.
//   X half q holds tile pixels  wmr*128 + q*64 + i  at row wmr*64 + i   (wmr = 0,1 ; i < 64)
//   W half q holds tile channels wnr*64 + q*32 + j  at row wnr*32 + j   (wnr = 0..3 ; j < 32)
// so a wave (wm, wn) still owns the CONTIGUOUS 128 pixels x 64 channels block (wm*128.., wn*64..) -- the row permutation
// lives only in the source addresses of the staging -- while each of its four quadrants (64 pixels x 32 channels x 64 k)
// reads one X sub-tile (8 ds_read_b128) and one W sub-tile (4):
//   phase 0: read X0,W0 sub-tiles ; quadrant (0,0)        phase 1: read W1 ; quadrant (0,1)
//   phase 2: read X1 (over X0's registers) ; (1,0)        phase 3: no reads, the k-tile's vmcnt wait ; (1,1)
// Staging order (one half-tile = 2 pieces per wave per phase): k-tile t phase 0 stages W1(t+1), phase 1 X1(t+1), phase 2
// X0(t+2), phase 3 W0(t+2).
//   WAR: a slot is restaged >= 2 phases after the phase that last read it (X0(t): read phase 0, restaged phase 2; the
//        others 3) -- both groups' reads are retired (lgkmcnt(0) right after the barrier that ends their LOAD section) at
//        least one whole barrier interval before either group issues the DMA.
//   RAW: every wave waits vmcnt(4) [vmcnt(0) once nothing younger was issued] before the first barrier of phase 3: all of
//        k-tile t+1 has landed for that wave; the later group does so one interval later, and the first read of k-tile
//        t+1 by either group comes after the barrier that ends that interval.
// Same swizzle / k order / accumulation order / epilogue as conv_igemm256.hip: results are bit-identical to it.
#include "conv_igemm256.h"
#ifndef UNIT_P8_X_AUX
#define UNIT_P8_X_AUX 0       // cache policy of the LDS-DMA loads (buffer instruction aux bits; 2 = nt): pixel rows / weights (tools/exp_wait.sh)
#endif
#ifndef UNIT_P8_W_AUX
#define UNIT_P8_W_AUX 0
#endif
#ifndef UNIT_P8_RD_PER
#define UNIT_P8_RD_PER 1        // fragment reads per MFMA gap inside the MFMA sections of phases 1 and 3 (tools/exp_wait.sh)
#endif
#ifndef UNIT_P8_FINE_WAIT
#define UNIT_P8_FINE_WAIT 0      // 1: one counted vmcnt wait per half-tile instead of one per k-tile / step (tools/exp_wait.sh: measured 1-8 % slower)
#endif
#include "conv_epilogue.h"

// diagnostic builds only (tools/exp_p8.sh): 1 = no LDS-DMA inside the loop, 2 = MFMAs replaced by one VALU add per fragment,
// 3 = fragment reads only in the first k-tile, 4 = 1 + 2, 5 = 2 + 3. Results are garbage in all of them.
#ifndef UNIT_DBGP8
#define UNIT_DBGP8 0
#endif

#ifdef UNIT_EPI_STAMP
// diagnostic build (tools/epi_stamp.sh): wave 0 of every 97th workgroup stamps s_memtime at kernel entry [0], first barrier [1], end of the
// main loop [2], after the __syncthreads in front of the epilogue [3], before the first block [4], after block b [5 + b], end [13];
// [14] = blockIdx.x, [15] = s_memrealtime at entry (100 MHz)
__device__ unsigned long long g_stamp[32 * 16];
extern "C" int unit_debug_read_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 32 * 16, 0, hipMemcpyDeviceToHost);
}
#define P8_STAMP(i) do { if (stamp) stamp[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define P8_STAMP(i) do { } while (0)
#endif

#define MFMA_BF16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)

// B1 = 16-row pixel blocks of the SECOND quadrant row of a wave (the first always has 4): 4 -> 256-row tiles, 3 -> 224-row
// tiles (RM schedule only). 224 rows divide the Res5 problem sizes (50 176 = 224 * 224 pixels per 1024 RoIs) into whole
// rounds of 256 workgroups where 256-row tiles leave the last round 1/2 - 3/4 empty. The LDS image keeps its 64-row groups
// (the 16 unused rows of the X1 half are staged as zeros and never read).
// X3: bf16x3 operands (conv_epilogue.h SplitK): x is a split tensor, the k extent holds three segments per 64-channel block, the output
// (and residual / mask_ref) are split tensors. Same schedule; only the staging offsets (scalars) and the epilogue's stores differ.
// PAIR: two problems of one layer in one grid (conv_epilogue.h ConvSecond; row-major tiles only).
// PERS (round 5): one workgroup per CU walks the row-major tiles bid, bid + gridDim.x, ... and issues the first k-tiles of its NEXT tile before the
// epilogue of the current one, whose scratch moves behind the operand slots that prologue fills (LDS 146 KB): the 4.7 k cycles between a workgroup's
// start and its first MFMA -- 12 % of a K = 512 tile, profiles/r05_exp_epilogue_diet.txt -- run under the epilogue's 7.8 k instead of after it.
template <typename TO, bool RM, int B1, bool X3 = false, bool PAIR = false, bool PERS = false>
__global__ void __launch_bounds__(512, 2) conv_igemm256_p8_kernel(Conv256Args p) {
  static_assert(RM || B1 == 4, "224-row tiles: RM schedule only");
  static_assert(!PAIR || (RM && B1 == 4), "pair launches: 256-row RM schedule");
  static_assert(!X3 || (RM && sizeof(TO) == 2), "bf16x3 operands: RM schedule, split bf16 output");
  static_assert(!PERS || (RM && !PAIR && sizeof(TO) == 2), "persistent tiles: RM schedule, bf16 rows through the LDS epilogue, one problem");
  constexpr int FBT = 4 + B1;
  constexpr int BM = 32 * FBT, BN = 256, BK = 64;
  constexpr int HALF = 128 * 128;               // 16 KB half-tile
  constexpr int SX0 = 0, SW0 = HALF, SW1 = 2 * HALF, SX1 = 3 * HALF, BUF = 4 * HALF;
  // epilogue scratch: over the operand stages, or (PERS) from buffer 1's X1 slot on -- the only slot the prologue of a tile leaves empty
  constexpr int EPI_OFF = PERS ? BUF + SX1 : 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int bid = blockIdx.x;
  if constexpr (PAIR) {
    if (p.second.on && bid >= p.second.tiles0) { bid -= p.second.tiles0; pair_swap_common(p); p.magic_ow = p.second.magic_ow; p.magic_oh = p.second.magic_oh; }
  }
  int nwg = p.tiles_m * p.tiles_n;
  const bool pm = !PAIR && !PERS && RM && B1 == 4 && p.pm_ncls > 0;
  int tile_n, tile_m;
  PmRows pmr = {0, 1, 0, 0, 1, p.OW, p.OH * p.OW, p.N, 0u, 0u};
  int pm_nh = 1;
  if (pm) {
    // position-class tiles (Conv256Args::pm_ncls): workgroup b runs on XCD b % 8. The tiles of a class are ordered by image, and XCD x
    // takes the x-th eighth of EVERY class -- the interior, edge and corner tiles of the same images, whose filter taps read the same
    // input pixels, go through one L2 -- heaviest classes first, and all channel tiles of a row tile one after the other.
    int xcd = bid % 8, loc = bid / 8;
    tile_n = loc % p.tiles_n;
    int idx = loc / p.tiles_n, c = 0;
    tile_m = -1;
    for (; c < p.pm_ncls; ++c) {
      int t0 = p.pm_cls[c].tile0, tc = (c + 1 < p.pm_ncls ? p.pm_cls[c + 1].tile0 : p.tiles_m) - t0;
      int lo = (xcd * tc + 7) / 8, hi = ((xcd + 1) * tc + 7) / 8;
      if (idx < hi - lo) { tile_m = t0 + lo + idx; break; }
      idx -= hi - lo;
    }
    if (tile_m < 0) return;
    const PmClass& k = p.pm_cls[c];
    pmr.i0 = (tile_m - k.tile0) * BM; pmr.np = k.np; pmr.oh0 = k.oh0; pmr.ow0 = k.ow0; pmr.cw = k.cw;
    pmr.magic_np = k.magic_np; pmr.magic_cw = k.magic_cw;
    pm_nh = k.nh;
  }
  // row-major tiles: workgroup (or, PERS, virtual workgroup) v runs on XCD v % 8 and takes the v / 8-th tile of that XCD's contiguous share
  auto map_tile = [&](int v) {
    int q = nwg / 8, r = nwg % 8, xcd = v % 8, loc = v / 8;
    int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    tile_n = t % p.tiles_n; tile_m = t / p.tiles_n;
  };
  if (!pm) map_tile(bid);
  int m0 = tile_m * BM, n0 = tile_n * BN;

  const bf16_t* __restrict__ X = (const bf16_t*)p.x;
  const bf16_t* __restrict__ Wt = (const bf16_t*)p.w;
  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(X), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Wt), 0, (int)p.w_bytes, 0x00020000);
  // dual-source 1x1 conv (Conv256Args::x2): the first cb_split 64-channel blocks of the k extent come from x (row pitch Cx), the rest from x2
  const bool dual = !X3 && p.x2 != nullptr;
  const int Cx = X3 ? p.sk.x_pitch : (dual ? p.cb_split * 64 : p.C);
  __amdgpu_buffer_rsrc_t rsX2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>((const bf16_t*)(dual ? p.x2 : p.x)), 0, (int)(dual ? p.x2_bytes : p.x_bytes), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;

  int tid = threadIdx.x, lane = tid & 63;
  int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wm = wid >> 2, wn = wid & 3;
  const int grp = wm;                          // waves 4-7 run half a phase behind waves 0-3
  unsigned long long* stamp = nullptr;
#ifdef UNIT_EPI_STAMP
  if (tid == 0 && blockIdx.x % 97 == 0 && blockIdx.x / 97 < 32) {
    stamp = g_stamp + (blockIdx.x / 97) * 16;
    stamp[14] = blockIdx.x; stamp[15] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  P8_STAMP(0);
  int lrow = lane >> 3, lc = lane & 7;

  // (PERS launches are pointwise by the launcher's choice: the per-row tap origins (x_ih0 / x_iw0, 8 registers) do not exist there -- a row past
  //  M is marked in its offset instead -- which is what lets the tile loop compile without spills)
  const bool pointwise = PERS || (p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0);      // (the bounds test of the single tap then only sees x_ih0 = 0 or the "past M" marker)
  // staging descriptors: half q (0,1), piece j (0,1) of this wave = half-tile rows R0 = (j*8 + wid)*8 .. +8 ; lane -> row
  // R0 + lrow, LDS chunk lc (lane-linear), SOURCE chunk lc ^ ((row>>1)&7)
  int x_ih0[4], x_iw0[4]; unsigned x_off0[4], w_off[4];
  const unsigned sw16 = (unsigned)((lc ^ ((((wid * 8 + lrow) >> 1)) & 7)) * 16);      // = sw * 16 of every piece of this lane (j * 64 rows do not change (R >> 1) & 7)
  auto setup_tile = [&](int lrow, int lc) {          // the staging descriptors of the tile at (m0, n0); (lrow, lc) = lane >> 3, lane & 7
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int R = (j * 8 + wid) * 8 + lrow;
      int sw = lc ^ ((R >> 1) & 7);
      int m = m0 + (R >> 6) * (FBT * 16) + q * 64 + (R & 63);
      bool ok = m < p.M && (q == 0 || (R & 63) < B1 * 16);
      int mm = ok ? m : 0;
      int ow, oh, n;
      if (pm) {                                  // row of the tile -> (image, position) of its class
        ok = pmr.map(m - m0, n, oh, ow);
        if (!ok) n = 0;
      } else if (pointwise) {                    // 1x1 s1 p0: input pixel = output pixel m; written as "image 0, row 0, column m" of a one-row map
        n = 0; oh = 0; ow = 0;
      } else {
        unsigned t = fast_div((unsigned)mm, (unsigned)p.OW, p.magic_ow);
        ow = mm - (int)t * p.OW;
        n = (int)fast_div(t, (unsigned)p.OH, p.magic_oh);
        oh = (int)t - n * p.OH;
      }
      int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
      x_off0[q * 2 + j] = pointwise ? ((unsigned)mm * (unsigned)Cx + (unsigned)(sw * 8)) * 2u
                                    : ((unsigned)n * (unsigned)(p.H * p.W * Cx) + (unsigned)((ih0 * p.W + iw0) * Cx + sw * 8)) * 2u;  // tap (0,0), wraps for negative ih0/iw0
      if constexpr (PERS) {
        if (!ok) x_off0[q * 2 + j] = OOB;
      } else {
        x_ih0[q * 2 + j] = ok ? ih0 : -(1 << 20);                    // rows past M fail the bounds test of every tap
        x_iw0[q * 2 + j] = iw0;
      }
      int nn = n0 + (R >> 5) * 64 + q * 32 + (R & 31);
      w_off[q * 2 + j] = nn < p.K ? ((unsigned)nn * (unsigned)p.Kgemm + (unsigned)sw * 8u) * 2u : OOB;
    }
  };
  setup_tile(lrow, lc);

  // k-tile order: channel block outermost, the R*S taps innermost (conv_igemm256.hip). (cb, r, s) of the k-tile that the
  // staging is currently working on, and the two byte offsets derived from them, live in scalar registers.
  // taps of this tile: all of them, or (position-major tiles) those inside the map at the tile's position -- wave-uniform scalars
  int r_lo = 0, r_hi = p.R - 1, s_lo = 0, s_hi = p.S - 1;
  if (pm) {                                      // the same for every position of the class (that is what makes it a class)
    r_lo = max(0, p.pad - pmr.oh0); r_hi = min(p.R - 1, p.H - 1 + p.pad - (pmr.oh0 + pm_nh - 1));
    s_lo = max(0, p.pad - pmr.ow0); s_hi = min(p.S - 1, p.W - 1 + p.pad - (pmr.ow0 + pmr.cw - 1));
  }
  int st_cb = 0, st_r = r_lo, st_s = s_lo;
  int st_sg = 0, st_cbr = 0;            // X3: segment of the virtual channel block st_cb, and the real 64-channel block it reads (st_cb = st_cbr * nseg + st_sg)
  unsigned st_kx = (unsigned)((st_r * p.W + st_s) * Cx) * 2u, st_kw = (unsigned)((st_r * p.S + st_s) * p.C) * 2u;
  if constexpr (X3) st_kx += (unsigned)((p.sk.seg_lo & 1) * p.sk.cr) * 2u;
  auto st_reset = [&]() {               // back to the first k-tile (PERS: the next tile of this workgroup)
    st_cb = 0; st_r = r_lo; st_s = s_lo; st_sg = 0; st_cbr = 0;
    st_kx = (unsigned)((st_r * p.W + st_s) * Cx) * 2u; st_kw = (unsigned)((st_r * p.S + st_s) * p.C) * 2u;
    if constexpr (X3) st_kx += (unsigned)((p.sk.seg_lo & 1) * p.sk.cr) * 2u;
  };
  auto st_advance = [&]() {
    if (++st_s > s_hi) {
      st_s = s_lo;
      if (++st_r > r_hi) {
        st_r = r_lo; ++st_cb;
        if constexpr (X3) { if (++st_sg == p.sk.nseg) { st_sg = 0; ++st_cbr; } }
      }
    }
    if constexpr (X3) st_kx = (unsigned)((st_r * p.W + st_s) * Cx + ((p.sk.seg_lo >> st_sg) & 1) * p.sk.cr + st_cbr * BK) * 2u;
    else st_kx = (unsigned)((st_r * p.W + st_s) * Cx + st_cb * BK) * 2u;
    st_kw = (unsigned)((st_r * p.S + st_s) * p.C + st_cb * BK) * 2u;
  };
  auto stage_x = [&](int q, int d) {
    char* base = smem + d * BUF + (q ? SX1 : SX0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bool ok;
      if constexpr (PERS) ok = x_off0[q * 2 + j] != OOB;
      else {
        int ih = x_ih0[q * 2 + j] + st_r, iw = x_iw0[q * 2 + j] + st_s;
        ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      }
      if (dual && st_cb >= p.cb_split) {      // (scalar branch) the row of x2: same pixel, ratio2 times the pitch; the 16-B chunk swizzle term stays
        unsigned o2 = (x_off0[q * 2 + j] - sw16) * (unsigned)p.ratio2 + sw16 + (unsigned)((st_cb - p.cb_split) * BK) * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX2, (lds_void*)(base + (j * 8 + wid) * 1024), 16, ok ? o2 : OOB, 0, 0, UNIT_P8_X_AUX);
      } else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void*)(base + (j * 8 + wid) * 1024), 16, ok ? x_off0[q * 2 + j] + st_kx : OOB, 0, 0, UNIT_P8_X_AUX);
    }
  };
  auto stage_w = [&](int q, int d) {
    char* base = smem + d * BUF + (q ? SW1 : SW0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned o = w_off[q * 2 + j];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(base + (j * 8 + wid) * 1024), 16, o == OOB ? OOB : o + st_kw, 0, 0, UNIT_P8_W_AUX);
    }
  };

  f32x4 acc[4][FBT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < FBT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  const int nk = pm ? (p.C / BK) * (r_hi - r_lo + 1) * (s_hi - s_lo + 1) : p.Kgemm / BK;
  const int frow = lane & 15, fq = lane >> 4;
  // per-lane fragment offsets inside a half-tile, k-substep 0 (substep 1 = ^ 64); tile rows b*16 / a*16 add b*2048 / a*2048
  const int fsw = (fq ^ ((frow >> 1) & 7)) << 4;
  const int offx = (wm * 64 + frow) * 128 + fsw;
  const int offw = (wn * 32 + frow) * 128 + fsw;

  i32x4 fx[4][2], fw0[2][2], fw1[2][2];
  auto read_x = [&](const char* half) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      fx[b][0] = *reinterpret_cast<const i32x4*>(half + b * 2048 + offx);
      fx[b][1] = *reinterpret_cast<const i32x4*>(half + b * 2048 + (offx ^ 64));
    }
  };
  auto read_w = [&](const char* half, i32x4 (&fw)[2][2]) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      fw[a][0] = *reinterpret_cast<const i32x4*>(half + a * 2048 + offw);
      fw[a][1] = *reinterpret_cast<const i32x4*>(half + a * 2048 + (offw ^ 64));
    }
  };
#define P8_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#if UNIT_DBGP8 == 2 || UNIT_DBGP8 == 4 || UNIT_DBGP8 == 5
#define P8_FMA(ACC, A, B) do { if (b == 0) ACC += __builtin_bit_cast(f32x4, A); if (a == 0) ACC += __builtin_bit_cast(f32x4, B); } while (0)
#else
#define P8_FMA(ACC, A, B) ACC = MFMA_BF16(A, B, ACC)
#endif
  // MFMA section of a phase: quadrant (qx, qw) = acc[qw*2 ..][qx*4 ..]
#define P8_MFMA(QX, QW, FW)                                                            \
  do {                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    __builtin_amdgcn_s_setprio(1);                                                     \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                   \
      _Pragma("unroll") for (int a = 0; a < 2; ++a)                                    \
        _Pragma("unroll") for (int b = 0; b < 4; ++b)                                  \
          P8_FMA(acc[(QW) * 2 + a][(QX) * 4 + b], FW[a][ks], fx[b][ks]);               \
    __builtin_amdgcn_s_setprio(0);                                                     \
  } while (0)

  // the LDS epilogue of the tile at (m0e, n0e) (bf16 rows of 16-byte vectors; conv_epilogue.h) -- the caller has passed the barrier behind the last
  // fragment read. PERS: the scratch sits behind the slots the next tile's prologue is filling (EPI_OFF).
  auto lds_epilogue = [&](int m0e, int n0e, int lane) {
    if constexpr (sizeof(TO) == 2) {
      if constexpr (X3) {              // split outputs: two bf16 planes per row (conv_epilogue.h SPL)
        if constexpr (B1 == 4) {
          if (pm) {
            epilogue_rows_bf16_dispatch<4, FBT, false, true, true, EPI_BIAS | EPI_RELU | EPI_Y, EPI_MK | EPI_Y, EPI_Y>(
                acc, smem + EPI_OFF + wid * EpiCfg<4>::BYTES, nullptr, wm * (FBT * 16), n0e + wn * 64, p, lane, &pmr);
            return;
          }
        }
        epilogue_rows_bf16_dispatch<4, FBT, false, false, true, EPI_FULL | EPI_BIAS | EPI_RELU | EPI_Y, EPI_FULL | EPI_BIAS | EPI_RES | EPI_RELU | EPI_Y,
                                    EPI_FULL | EPI_BIAS | EPI_Y, EPI_FULL | EPI_MK | EPI_Y, EPI_FULL | EPI_RES | EPI_MK | EPI_Y, EPI_FULL | EPI_Y>(
            acc, smem + EPI_OFF + wid * EpiCfg<4>::BYTES, nullptr, m0e + wm * (FBT * 16), n0e + wn * 64, p, lane);
        return;
      }
      if constexpr (RM && !X3) {
        if (p.ex_on) {               // fused average pool / ReLU bit mask / bit-mask input (unit_conv2d_fwd_big_ex)
          // (layers.py BottleneckBlock: conv3 with / without the shortcut in the k extent, the pooled last block, the mask-bit dgrads)
          // (the Res5 problem sizes are whole numbers of tiles: only the all-rows-exist form of each combination is instantiated)
          constexpr int F = EPI_FULL;
          epilogue_rows_bf16_dispatch<4, FBT, true, false, false,
                                      F | EPI_BIAS | EPI_RES | EPI_RELU | EPI_RB | EPI_Y, F | EPI_BIAS | EPI_RELU | EPI_RB | EPI_Y, F | EPI_BIAS | EPI_RELU | EPI_Y,
                                      F | EPI_BIAS | EPI_RES | EPI_RELU | EPI_RB | EPI_PP, F | EPI_BIAS | EPI_RES | EPI_RELU | EPI_PP,
                                      F | EPI_RES | EPI_MB | EPI_Y, F | EPI_MB | EPI_Y, F | EPI_Y>(
              acc, smem + EPI_OFF + wid * EpiCfg<4>::BYTES, (float*)(smem + 36864 + wid * 8192), m0e + wm * (FBT * 16), n0e + wn * 64, p, lane);
          return;
        }
      }
      if constexpr (RM && B1 == 4) {
        if (pm) {
          epilogue_rows_bf16_dispatch<4, FBT, false, true, false, EPI_BIAS | EPI_RELU | EPI_Y, EPI_MK | EPI_Y, EPI_Y>(
              acc, smem + EPI_OFF + wid * EpiCfg<4>::BYTES, nullptr, wm * (FBT * 16), n0e + wn * 64, p, lane, &pmr);
          return;
        }
      }
      epilogue_rows_bf16_dispatch<4, FBT, false, false, false, EPI_FULL | EPI_BIAS | EPI_RELU | EPI_Y, EPI_FULL | EPI_BIAS | EPI_Y, EPI_FULL | EPI_Y,
                                  EPI_FULL | EPI_MK | EPI_Y, EPI_FULL | EPI_RES | EPI_MK | EPI_Y, EPI_BIAS | EPI_RELU | EPI_Y, EPI_MK | EPI_Y>(
          acc, smem + EPI_OFF + wid * EpiCfg<4>::BYTES, nullptr, m0e + wm * (FBT * 16), n0e + wn * 64, p, lane, nullptr, stamp ? stamp + 4 : nullptr);
    }
  };

  if constexpr (RM) {
    // ---- RM schedule: the fragment reads of phase p+1 are issued INSIDE the MFMA section of phase p (one per MFMA gap, a
    // second X register set), so a LOAD section is only {2 LDS-DMA pieces, the k-tile's vmcnt wait}: tools/exp_p8.sh showed
    // the plain schedule bound by its LOAD sections (reads + their latency + DMA issue ~ 440 clk against 256 clk of MFMA).
    // Everything moves one phase earlier relative to the quadrants:
    //   phase 0: stage X1(t+1) ; (0,0) || read W1(t)        phase 1: stage X0(t+2) ; (0,1) || read X1(t)
    //   phase 2: stage W0(t+2), vmcnt wait ; (1,0)           phase 3: stage W1(t+2) ; (1,1) || read X0(t+1), W0(t+1)
    // Reads are retired (lgkmcnt(0)) before the barrier that ends their MFMA section.
    //   WAR: X0/W0(t) last read in M(t-1, 3), restaged in L(t, 1) / L(t, 2): >= 2 phases; X1(t-1): M(t-1, 1) -> L(t, 0): 3;
    //        W1(t): M(t, 0) -> L(t, 3): 3.
    //   RAW: vmcnt in L(t, 2) of both groups (two half-tiles younger than k-tile t+1 may stay in flight); first read of
    //        k-tile t+1 in M(t, 3), which for either group starts after the barrier that ends the later group's L(t, 2).
    //        (UNIT_P8_FINE_WAIT=1: every half-tile waited for separately, with the four half-tiles issued after it still in flight, one
    //        phase before the MFMA section that reads it -- X1(t+1) then has four phases to land instead of two. Measured 1-8 % SLOWER
    //        on every Res5 / RPN shape, profiles/r03_exp_fine_vmcnt.txt: the loop is not waiting for that half-tile.)
    i32x4 fxb[B1][2];
    auto read_xb = [&](const char* half) {
#pragma unroll
      for (int b = 0; b < B1; ++b) {
        fxb[b][0] = *reinterpret_cast<const i32x4*>(half + b * 2048 + offx);
        fxb[b][1] = *reinterpret_cast<const i32x4*>(half + b * 2048 + (offx ^ 64));
      }
    };
#define P8_MM(QX, QW, FW, FX, NB, NR, PER, READS)                                        \
    do {                                                                                 \
      __builtin_amdgcn_s_setprio(1);                                                     \
      READS;                                                                             \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                   \
        _Pragma("unroll") for (int a = 0; a < 2; ++a)                                    \
          _Pragma("unroll") for (int b = 0; b < (NB); ++b)                               \
            acc[(QW) * 2 + a][(QX) * 4 + b] = MFMA_BF16(FW[a][ks], FX[b][ks], acc[(QW) * 2 + a][(QX) * 4 + b]); \
      _Pragma("unroll") for (int i = 0; i < (NR) / (PER); ++i) {                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                               \
        __builtin_amdgcn_sched_group_barrier(0x100, (PER), 0);                           \
      }                                                                                  \
      if (4 * (NB) - (NR) / (PER) > 0) __builtin_amdgcn_sched_group_barrier(0x008, 4 * (NB) - (NR) / (PER) > 0 ? 4 * (NB) - (NR) / (PER) : 1, 0); \
      __builtin_amdgcn_s_setprio(0);                                                     \
      if ((NR) > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   \
    } while (0)
    auto prologue_stage = [&]() {          // k-tile 0 whole, k-tile 1 without its X1 half (staged in phase 0 of the loop)
      stage_x(0, 0); stage_w(0, 0); stage_w(1, 0); stage_x(1, 0);
      st_advance();
      if (nk > 1) { stage_x(0, 1); stage_w(0, 1); stage_w(1, 1); }
    };
    prologue_stage();
    if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int pers_v = blockIdx.x;             // PERS: the virtual workgroup (tile) this pass of the loop below works on
    while (true) {
    P8_BAR();
    P8_STAMP(1);
    read_w(smem + SW0, fw0);
    read_x(smem + SX0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (grp == 1) P8_BAR();
    for (int t = 0; t < nk; ++t) {
      const int d = t & 1;
      const char* buf = smem + d * BUF;
      const char* bnx = smem + (d ^ 1) * BUF;
      const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
      // phase 0
      if (n1) stage_x(1, d ^ 1);
      st_advance();
#if UNIT_P8_FINE_WAIT
      if (n1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // X1(t) landed (read in M(t, 1)); X0, W0, W1, X1 of t+1 younger
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      P8_BAR();
      P8_MM(0, 0, fw0, fx, 4, 4, 1, read_w(buf + SW1, fw1));
      P8_BAR();
      // phase 1
      if (n2) stage_x(0, d);
      P8_BAR();
      P8_MM(0, 1, fw1, fx, 4, 2 * B1, UNIT_P8_RD_PER, read_xb(buf + SX1));
      P8_BAR();
      // phase 2
#if UNIT_P8_FINE_WAIT
      if (n2) {
        stage_w(0, d);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // X0, W0 of t+1 landed (read in M(t, 3)); W1, X1 of t+1, X0, W0 of t+2 younger
      } else if (n1) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");             // W1, X1 of t+1 younger
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
#else
      if (n2) {
        stage_w(0, d);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
#endif
      P8_BAR();
      P8_MM(1, 0, fw0, fxb, B1, 0, 1, (void)0);
      P8_BAR();
      // phase 3 (after the last k-tile the reads fetch stale, in-bounds LDS that nobody uses)
      if (n2) stage_w(1, d);
#if UNIT_P8_FINE_WAIT
      if (n2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // W1(t+1) landed (read in M(t+1, 0)); X1(t+1), X0, W0, W1 of t+2 younger
      else if (n1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // X1(t+1) younger
#endif
      P8_BAR();
      P8_MM(1, 1, fw1, fxb, B1, 12, UNIT_P8_RD_PER, read_w(bnx + SW0, fw0); read_x(bnx + SX0));
      P8_BAR();
    }
    if (grp == 0) P8_BAR();
    if constexpr (!PERS) break;
    else {
      // ---- next tile of this workgroup: its first k-tiles go out BEFORE the epilogue of the current one
      P8_STAMP(2);
      __syncthreads();                   // every wave is done with the operand stages
      P8_STAMP(3);
      const int m0e = m0, n0e = n0, nv = pers_v + (int)gridDim.x;
      const bool more = nv < nwg;
      // (the fused-pool epilogue keeps its per-wave sums in the operand area: no early prologue under it)
      const bool early = more && !(p.ex_on && p.ex.pool_partial != nullptr);
      // (the lane number is made opaque here: what the epilogue and the descriptor set-up derive from it is then recomputed per tile instead of
      //  being hoisted out of the tile loop and kept in registers across the main loop, which has none to spare)
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      auto next_tile = [&]() { map_tile(nv); m0 = tile_m * BM; n0 = tile_n * BN; setup_tile(lane_o >> 3, lane_o & 7); st_reset(); prologue_stage(); };
      if (early) next_tile();
      lds_epilogue(m0e, n0e, lane_o);
      if (!more) return;
      if (!early) { __syncthreads(); next_tile(); }
      else {                             // recomputed rather than kept: 16 descriptor registers less across the epilogue (which otherwise spills)
        asm volatile("" : "+s"(m0), "+s"(n0));          // (opaque to common-subexpression elimination)
        setup_tile(lane_o >> 3, lane_o & 7);
      }
      // the tile's k-tile 0 (and the epilogue's own loads / stores, which share the counter) has landed; the three half-tiles of k-tile 1 are
      // covered by the loop's own wait in phase 2
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      zero_acc();
      pers_v = nv;
    }
    }
#undef P8_MM
  } else {
  // ---- prologue: k-tile 0 (4 half-tiles) and X0, W0 of k-tile 1 in flight
  stage_x(0, 0); stage_w(0, 0); stage_w(1, 0); stage_x(1, 0);
  st_advance();
  if (nk > 1) {
    stage_x(0, 1); stage_w(0, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  P8_BAR();
  if (grp == 1) P8_BAR();

  for (int t = 0; t < nk; ++t) {
    const int d = t & 1;
    const char* buf = smem + d * BUF;
    const bool n1 = t + 1 < nk && UNIT_DBGP8 != 1 && UNIT_DBGP8 != 4, n2 = t + 2 < nk && UNIT_DBGP8 != 1 && UNIT_DBGP8 != 4;
#if UNIT_DBGP8 == 3 || UNIT_DBGP8 == 5
#define P8_RD(stmt) do { if (t == 0) { stmt; } } while (0)
#else
#define P8_RD(stmt) do { stmt; } while (0)
#endif
    // phase 0
    P8_RD(read_w(buf + SW0, fw0));
    __builtin_amdgcn_sched_barrier(0);
    P8_RD(read_x(buf + SX0));
    if (n1) stage_w(1, d ^ 1);
    P8_BAR();
    P8_MFMA(0, 0, fw0);
    P8_BAR();
    // phase 1
    P8_RD(read_w(buf + SW1, fw1));
    if (n1) stage_x(1, d ^ 1);
    st_advance();
    P8_BAR();
    P8_MFMA(0, 1, fw1);
    P8_BAR();
    // phase 2
    P8_RD(read_x(buf + SX1));
    if (n2) stage_x(0, d);
    P8_BAR();
    P8_MFMA(1, 0, fw0);
    P8_BAR();
    // phase 3
    if (n2) {
      stage_w(0, d);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    P8_BAR();
    P8_MFMA(1, 1, fw1);
    P8_BAR();
  }
  if (grp == 0) P8_BAR();
  }
#undef P8_MFMA
#undef P8_FMA
#undef P8_RD
#undef P8_BAR

  if constexpr (sizeof(TO) == 2) {
    if ((p.ldy & 7) == 0) {          // row-major epilogue through a wave-private LDS scratch (conv_epilogue.h)
      P8_STAMP(2);
      __syncthreads();               // every wave is done with the operand stages
      P8_STAMP(3);
      lds_epilogue(m0, n0, lane);
      P8_STAMP(13);
      return;
    }
  }
  TO* __restrict__ Y = (TO*)p.y;
  const TO* __restrict__ Rz = (const TO*)p.residual;
  const TO* __restrict__ Mk = (const TO*)p.mask_ref;
  bool plain = (p.oy_mul == 1 && p.OHf == p.OH && p.OWf == p.OW);
#pragma unroll
  for (int b = 0; b < FBT; ++b) {
    int m = m0 + wm * (FBT * 16) + b * 16 + frow;
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

template <typename TO, bool RM, int B1, bool X3 = false>
static int launch256_p8(Conv256Args& a, hipStream_t st) {
  if (a.second.on) {          // pair launch (conv_epilogue.h ConvSecond): row-major 256-row tiles of both problems in one grid
    if constexpr (RM && B1 == 4 && sizeof(TO) == 2) {
      a.pm_ncls = 0;
      a.tiles_m = cdiv(a.M, 256); a.tiles_n = cdiv(a.K, 256);
      a.second.tiles_m = cdiv(a.second.M, 256);
      a.second.tiles0 = a.tiles_m * a.tiles_n;
      size_t lds2 = 8 * 128 * 128;
      static bool attr2 = false;
      if (!attr2) {
        (void)hipFuncSetAttribute((const void*)conv_igemm256_p8_kernel<TO, RM, B1, X3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        attr2 = true;
      }
      conv_igemm256_p8_kernel<TO, RM, B1, X3, true><<<(a.tiles_m + a.second.tiles_m) * a.tiles_n, 512, lds2, st>>>(a);
      UNIT_LAUNCH_CHECK();
      return UNIT_OK;
    } else {
      unit_set_error("conv_big: pair launches need the 256-row RM schedule with bf16 output");
      return UNIT_ERR_UNSUPPORTED;
    }
  }
  if (a.pm_ncls == 0) a.tiles_m = cdiv(a.M, 32 * (4 + B1));
  a.tiles_n = cdiv(a.K, 256);
  int grid = a.tiles_m * a.tiles_n;
  if (a.pm_ncls > 0) {
    if (!RM || B1 != 4) { unit_set_error("conv_big: position-class tiles need the 256-row RM schedule"); return UNIT_ERR_UNSUPPORTED; }
    // a.tiles_m was set with the class table; every XCD gets an eighth of every class (rounded): the grid holds the longest such list
    int most = 0;
    for (int x = 0; x < 8; ++x) {
      int cnt = 0;
      for (int c = 0; c < a.pm_ncls; ++c) {
        int tc = (c + 1 < a.pm_ncls ? a.pm_cls[c + 1].tile0 : a.tiles_m) - a.pm_cls[c].tile0;
        cnt += ((x + 1) * tc + 7) / 8 - (x * tc + 7) / 8;
      }
      most = cnt > most ? cnt : most;
    }
    grid = most * 8 * a.tiles_n;
  }
  size_t lds = 8 * 128 * 128;
  if constexpr (RM && B1 == 4 && sizeof(TO) == 2) {
    // persistent tiles (PERS): row-major launches of more tiles than CUs whose rows go through the LDS epilogue. UNIT_P8_PERSIST=0: off (A/B)
    const char* pe = getenv("UNIT_P8_PERSIST");          // (read per launch: the bit-identity test flips it inside one process)
    const int pers = pe ? atoi(pe) : 1;
    if (pers && a.pm_ncls == 0 && (a.ldy & 7) == 0 && grid > 256 && a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0) {
      const size_t lds_p = 7 * 128 * 128 + 8 * EpiCfg<4>::BYTES;          // operand slots up to buffer 1's X1 + the epilogue scratch from there on
      static bool attr_p = false;
      if (!attr_p) {
        (void)hipFuncSetAttribute((const void*)conv_igemm256_p8_kernel<TO, RM, B1, X3, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p);
        attr_p = true;
      }
      static int ncu = 0;                    // one workgroup per CU (256 on MI355X); a multiple of 8 so that a workgroup's tiles stay on its XCD's share
      if (ncu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        ncu = n / 8 * 8;
      }
      conv_igemm256_p8_kernel<TO, RM, B1, X3, false, true><<<grid < ncu ? grid : ncu, 512, lds_p, st>>>(a);
      UNIT_LAUNCH_CHECK();
      return UNIT_OK;
    }
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_igemm256_p8_kernel<TO, RM, B1, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  conv_igemm256_p8_kernel<TO, RM, B1, X3><<<grid, 512, lds, st>>>(a);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

int unit_conv256_p8_launch(Conv256Args& a, int out_dtype, bool reads_in_mfma, bool rows224, hipStream_t st) {
  if (a.sk.nseg > 1) {             // bf16x3 operands (unit_conv2d_fwd_x3)
    if (out_dtype != UNIT_BF16 || (a.ldy & 7) != 0 || rows224 || !reads_in_mfma) { unit_set_error("conv_big: bf16x3 operands need the 256-row RM schedule and split output rows of 16-byte vectors"); return UNIT_ERR_UNSUPPORTED; }
    return launch256_p8<bf16_t, true, 4, true>(a, st);          // (a pair launch goes through the same launcher)
  }
  if (rows224 && !reads_in_mfma) { unit_set_error("conv_big: 224-row tiles need the reads-in-MFMA schedule"); return UNIT_ERR_UNSUPPORTED; }
  if (out_dtype == UNIT_BF16) return rows224 ? launch256_p8<bf16_t, true, 3>(a, st) : reads_in_mfma ? launch256_p8<bf16_t, true, 4>(a, st) : launch256_p8<bf16_t, false, 4>(a, st);
  if (out_dtype == UNIT_F32) return rows224 ? launch256_p8<float, true, 3>(a, st) : reads_in_mfma ? launch256_p8<float, true, 4>(a, st) : launch256_p8<float, false, 4>(a, st);
  unit_set_error("conv_big: unsupported out dtype");
  return UNIT_ERR_UNSUPPORTED;
}
