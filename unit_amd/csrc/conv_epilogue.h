// conv_epilogue.h -- row-major epilogue of the LDS-DMA implicit-GEMM conv kernels (bf16 output).
//
// The MFMA accumulator layout gives every lane 4 consecutive channels of ONE pixel row (16 rows x 32 B per wave
// instruction): stores, residual reads and ReLU-mask reads issued straight from it touch 16 cache lines for 512 useful
// bytes and re-request every 128-B line four times. The Res5 conv3 layers (512 -> 2048 + residual + ReLU, 410 MB of
// epilogue traffic per 1024 RoIs for 105 GFLOP) ran at 1.35 TB/s that way -- half their time.
// Here each 16-row x (FA*16)-channel accumulator block goes through a wave-private LDS scratch (fp32, padded pitch) and
// comes back row-major: a lane owns 8 consecutive channels (16 B of bf16) of one row, a wave instruction covers whole
// 128-B lines (64-channel wave tiles: 8 rows x 128 B per instruction). The residual / mask lines of block b+1 are
// requested before block b goes through the scratch, so their latency hides under the LDS round trip and the stores.
#pragma once
#include "common.h"

// Position-class tiles (conv_igemm256p8.hip, Conv256Args::pm): the rows of a 3x3 s1 p1 "same" conv over small maps are regrouped by
// the CLASS of their output position -- rectangles of positions that see the same set of in-map filter taps (interior, four edges, four
// corners) -- image-major inside a class: class row i = image i / np, position (oh0 + (i % np) / cw, ow0 + (i % np) % cw).
// x / d for 0 <= x with x * d < 2^32, magic = ceil(2^32 / d) (d >= 2; d == 1: magic 0 -> plain path): exact -- the quotient estimate
// exceeds x / d by less than 1 / d. The conv kernels' tile set-up is a chain of integer divisions by run-time constants (~25 instructions
// each without this): in-kernel stamps put the set-up at 2-3 k cycles of a 42 k-cycle 512 -> 2048 tile (profiles/r03_exp_p8_tile_stamps.txt).
__host__ __device__ __forceinline__ unsigned div_magic(unsigned d) { return d > 1 ? (unsigned)((0x100000000ull + d - 1) / d) : 0u; }
__device__ __forceinline__ unsigned fast_div(unsigned x, unsigned d, unsigned magic) { return magic ? __umulhi(x, magic) : (d > 1 ? x / d : x); }

// bf16x3 ("split") operands of the LDS-DMA conv kernels (their X3 instantiations; csrc/split.hip has the format and the converters).
// An fp32 activation x travels as two bf16 planes per pixel row, [row][2][Cr]: hi = bf16(x), lo = bf16(x - hi) -- 16 significant bits
// in 4 bytes -- and a weight likewise as Wh, Wl. x . W ~ lo.Wh + hi.Wh + hi.Wl with fp32 accumulation (the dropped lo.Wl term and the
// representation error are ~2^-17 relative per product) is ONE GEMM over a three times longer contraction: each 64-channel block of
// the k extent becomes `nseg` = 3 k-tiles, [lo | hi | hi] of x against [Wh | Wh | Wl] of a weight tensor laid out as
// [K][R][S][Cr / 64][nseg][64]. To the kernels this is a conv with C = nseg * Cr virtual channels whose x k-tiles come from row
// pitch `x_pitch`, plane (seg_lo >> segment) & 1, channel block cb. Only the staging addresses change: scalar arithmetic.
struct SplitK {
  int nseg;        // k segments per 64-channel block (3: [lo.Wh, hi.Wh, hi.Wl]); 0 / 1 = plain bf16 operands
  int seg_lo;      // bit s: segment s reads the lo plane of x (offset cr elements inside the row)
  int cr;          // real channels per plane
  int x_pitch;     // elements per pixel row of x (2 * cr)
};

// "Pair" launches (PAIR kernel instantiations): ONE grid over two independent problems of the SAME layer -- same weights, channels, taps, stride,
// epilogue -- that differ in their tensors and map sizes: the supervised and the weak batch of a training step, each zero-padded to its OWN
// largest image (meta_arch/rcnn.py:438-452, data/build.py:476-486 aspect-ratio grouping), so that a 3x3 / strided layer of both runs as one
// launch instead of two half-empty ones (pointwise stride-1 layers need nothing: their rows are independent, the two batches are simply
// concatenated). Workgroups [0, tiles0) run the first problem, the rest the second: the kernel swaps these fields into its argument block.
struct ConvSecond {
  int on, tiles0;
  const void* x; void* y; const void* residual; const void* mask_ref;
  int N, H, W, OH, OW, OHf, OWf, M, tiles_m;
  unsigned x_bytes, magic_ow, magic_oh;
};
template <typename A>
__device__ __forceinline__ void pair_swap_common(A& p) {
  p.x = p.second.x; p.y = p.second.y; p.residual = p.second.residual; p.mask_ref = p.second.mask_ref;
  p.N = p.second.N; p.H = p.second.H; p.W = p.second.W; p.OH = p.second.OH; p.OW = p.second.OW; p.OHf = p.second.OHf; p.OWf = p.second.OWf;
  p.M = p.second.M; p.tiles_m = p.second.tiles_m; p.x_bytes = p.second.x_bytes;
}
// non-persistent kernels: called once at entry with the workgroup id
template <typename A>
__device__ __forceinline__ void pair_enter(A& p, int& bid) {
  if (p.second.on && bid >= p.second.tiles0) { bid -= p.second.tiles0; pair_swap_common(p); }
}

struct PmClass { int tile0, np, oh0, nh, ow0, cw; unsigned magic_np, magic_cw; };      // tiles [tile0, next class's tile0) ; np = nh * cw positions
struct PmRows {                                           // rows of one tile: class row i0 + (row of the tile)
  int i0, np, oh0, ow0, cw, OW, OHW, N;
  unsigned magic_np, magic_cw;
  __device__ __forceinline__ bool map(int row, int& img, int& oh, int& ow) const {
    int i = i0 + row;
    img = (int)fast_div((unsigned)i, (unsigned)np, magic_np);
    int q = i - img * np, dh = (int)fast_div((unsigned)q, (unsigned)cw, magic_cw);
    oh = oh0 + dh; ow = ow0 + (q - dh * cw);
    return img < N;
  }
};

template <int FA, int RB = 16> struct EpiCfg {
  static constexpr int CH = FA * 16;            // channels of the wave tile
  static constexpr int LPR = CH / 8;            // lanes per row (8 channels each)
  static constexpr int RPP = 64 / LPR;          // rows per pass
  static constexpr int NP = RB / RPP;           // passes per RB-row block (RB = 16: 16x16 MFMA accumulators, 32: 32x32)
  static constexpr int PITCH = CH * 4 + 16;     // bytes per scratch row (fp32 + 16 B pad: rows land on different banks)
  static constexpr int BYTES = RB * PITCH;      // scratch per wave
};

// Optional extras of the epilogue (EX = true instantiations; conv_igemm256p8.hip):
//   relu_bits   : one bit per output element, (stored value) > 0, in the epilogue's own order so that a lane writes its 16 rows x 8
//                 channels as ONE 16-byte store per wave tile (byte stores, one per row pass, cost as many store instructions as the
//                 map itself did): 16-byte word [(m / 128) * (ldy / 64) + n / 64][lane], lane = (m % 8) * 8 + (n % 64) / 8,
//                 bit ((m % 128) / 8) * 8 + n % 8 of the word (relu_bit_index() below). 1/16 of the bytes of the bf16 map.
//   mask_bits   : the same layout as an INPUT: v = bit ? v : 0 (dgrad epilogue on the same geometry), replaces mask_ref.
//   pool_partial: per-RoI channel sums of the (bf16-rounded) outputs, for a global average pool over `pool_rows` consecutive
//                 rows (49 = 7x7 bins) fused into the conv: fp32 [wave tile of FB*16 rows][segment 0..3][ldy], a wave tile's rows
//                 cover at most 4 RoIs; unit_pool_finish adds the <= 2 pieces of a RoI in fixed order. Deterministic (no atomics).
//                 With y == nullptr the output tensor itself is never written.
// (m, n) -> (16-byte word index, bit inside the word) of relu_bits for an [M][ldy] map, ldy % 64 == 0
__host__ __device__ __forceinline__ void relu_bit_index(long m, int n, int ldy, long& word, int& bit) {
  word = ((m >> 7) * (ldy >> 6) + (n >> 6)) * 64 + ((m & 7) * 8 + ((n & 63) >> 3));
  bit = (int)((m & 127) >> 3) * 8 + (n & 7);
}

struct EpiExtra {
  unsigned char* relu_bits;
  const unsigned char* mask_bits;
  float* pool_partial;
  int pool_rows;
};

// The wave tile is FA*16 channels x FB*16 pixel rows, handed over in blocks of RB rows: put_block(b, scr) writes the fp32 values of
// rows b*RB .. b*RB+RB-1 into the scratch as [row][channel] with pitch EpiCfg::PITCH (whatever the MFMA accumulator layout is);
// m_w / n_w = first pixel row / channel of the wave tile; scr = this wave's scratch (EpiCfg<FA, RB>::BYTES, 16-B aligned).
// Requires p.ldy % 8 == 0.
// EX: `pool` = this wave's 8 KB LDS area [4 segments][8 row classes][64 channels] fp32 (FA == 4 only); plain output layout only.
// PM: position-class tiles: m_w = the wave's first row INSIDE its tile, `rows` maps a tile row to (image, position); the stored row is
// image * OH*OW + position.
#ifdef UNIT_EPI_STAMP      // diagnostic build only (tools/epi_stamp.sh): s_memtime stamps of one wave around the epilogue's steps
#define EPI_STAMP(i) do { if (stamp) stamp[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define EPI_STAMP(i) do { } while (0)
#endif
// the epilogue's global accesses: a residual / mask segment is read once, an output segment is written once and read by a LATER kernel.
// UNIT_EPI_NT: 2 (default) = the residual / mask loads are non-temporal, so that they do not push the operand tiles (weights, the pixel
// rows the other channel tiles of the XCD are about to read) out of the 4 MB L2 -- FETCH_SIZE of a 512 -> 2048 launch is 3x its operands;
// 1 = the stores too, 3 = the stores only, 0 = none. Same values in every form. Measured (profiles/r03_exp_epilogue_nontemporal.txt): with
// loads AND stores the Res5 / RPN launches are 2-7 % faster in isolation (512 -> 2048 + residual 173 -> 161 us) but the step is not (the
// consumers of the outputs then miss the Infinity Cache); loads only: step 16.40 -> 16.30 ms over five alternating runs.
#ifndef UNIT_EPI_NT
#define UNIT_EPI_NT 2
#endif
// diagnostic builds only (tools/epi_issue.sh): bit 0 = the residual / mask loads come from registers instead of memory, bit 1 = the output
// stores are issued only for a value that never occurs -- what is left is the epilogue's INSTRUCTION time (LDS round trip + VALU).
#ifndef UNIT_EPI_DBG
#define UNIT_EPI_DBG 0
#endif
typedef __attribute__((ext_vector_type(4))) int epi_i32x4;
__device__ __forceinline__ bf16x8 epi_load8(const bf16_t* q) {
#if UNIT_EPI_DBG & 1
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (bf16_t)(float)(((uintptr_t)q >> (4 + j)) & 3);
  return v;
#elif UNIT_EPI_NT == 1 || UNIT_EPI_NT == 2
  return __builtin_bit_cast(bf16x8, __builtin_nontemporal_load(reinterpret_cast<const epi_i32x4*>(q)));
#else
  return *reinterpret_cast<const bf16x8*>(q);
#endif
}
__device__ __forceinline__ void epi_store8(bf16_t* q, bf16x8 v) {
#if UNIT_EPI_DBG & 2
  if (__builtin_bit_cast(epi_i32x4, v)[0] == 0x7fc17fc3) *reinterpret_cast<bf16x8*>(q) = v;
#elif UNIT_EPI_NT == 1 || UNIT_EPI_NT == 3
  __builtin_nontemporal_store(__builtin_bit_cast(epi_i32x4, v), reinterpret_cast<epi_i32x4*>(q));
#else
  *reinterpret_cast<bf16x8*>(q) = v;
#endif
}

// Instruction diet of a pass (round 4: the epilogue is bound by its ~150-200 instructions per 8-row pass, not by memory --
// profiles/r04_exp_epilogue_issue.txt). UNIT_EPI_SLIM=0 restores the plain C forms (identical values; A/B and bit-identity tests).
#ifndef UNIT_EPI_SLIM
#define UNIT_EPI_SLIM 1
#endif
// max(x, 0) as ONE v_max_f32: fmaxf() compiles to a canonicalising v_max_f32 x, x in front of it (quiets a signalling NaN, nothing else)
__device__ __forceinline__ float epi_relu(float x) {
#if UNIT_EPI_SLIM
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
#else
  return fmaxf(x, 0.f);
#endif
}
// bit ? x : +0 as bit-field extract + and (v_bfe_i32 gives 0 / -1) instead of and + compare + (VCC wait) + select
__device__ __forceinline__ float epi_keep_if_bit(float x, unsigned bits, int j) {
#if UNIT_EPI_SLIM
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & (unsigned)__builtin_amdgcn_sbfe((int)bits, j, 1));
#else
  return ((bits >> j) & 1u) ? x : 0.f;
#endif
}
// one bit per element of eight packed bf16 values, (value > 0). (A packed-integer form -- positive int16 per half word -- was tried in
// round 4: hipcc folded its short-vector max / min into two compares for the first word only, wrong bits, caught by
// test_conv_fused_pool_and_relu_bits; spelled out in 32-bit integer ops it needs more instructions than the eight compares + selects here.)
__device__ __forceinline__ unsigned epi_positive_bits(bf16x8 o) {
  unsigned bits = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) bits |= ((float)o[j] > 0.f ? 1u : 0u) << j;
  return bits;
}

// Packed forms of a pass's tail (round 5; UNIT_EPI_SLIM): the eight outputs of a lane live as four dwords of two bf16 each.
//   epi_cvt_pk   : two fp32 -> one dword, ONE v_cvt_pk_bf16_f32 (the C conversion of a single value costs one each plus a v_perm per pair)
//   epi_relu_pk  : max(x, 0) of both halves as signed 16-bit integers (a bf16 is negative iff its int16 is; -0 -> +0). Rounding commutes
//                  with max(., 0), so relu-after-rounding stores the same bits as rounding-after-relu. (A positive NaN stays a NaN where
//                  v_max_f32 would have returned 0: a diverged model, and the loss is NaN either way.)
//   epi_positive_bits_pk : (value > 0) of the eight halves = (int16 > 0): clamp to [0, 1] per half, then fold the four dwords into a byte.
// Spelled in inline assembly: hipcc's short-vector min / max folding produced wrong bits in round 4 (see epi_positive_bits above).
__device__ __forceinline__ unsigned epi_cvt_pk(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned epi_relu_pk(unsigned w) {
  unsigned r;
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(w));
  return r;
}
__device__ __forceinline__ unsigned epi_positive_bits_pk(u32x4 w, bool nonneg) {
  unsigned t[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned x = w[i];
    if (!nonneg) asm("v_pk_max_i16 %0, %1, 0" : "=v"(x) : "v"(x));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(t[i]) : "v"(x), "v"(0x00010001u));      // 0 / 1 per half: bit 0 = element 2i, bit 16 = element 2i + 1
  }
  unsigned r = t[0] | (t[1] << 2) | (t[2] << 4) | (t[3] << 6);
  return (r | (r >> 15)) & 0xffu;
}
typedef __attribute__((ext_vector_type(2))) float epi_f32x2;
__device__ __forceinline__ float epi_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float epi_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// Compile-time knowledge of a launch's epilogue switches: FL >= 0 names exactly which of them are on (the hot Res5 / RPN combinations get
// straight-line passes the compiler can schedule across: the dynamic form branches six times per 8-row pass), FL = -1 reads them from
// the arguments as before. Same values either way.
// EPI_FULL: every row and channel of the wave tile exists (no bounds tests, no exec masks around the loads and stores).
enum { EPI_RES = 1, EPI_RELU = 2, EPI_MK = 4, EPI_MB = 8, EPI_RB = 16, EPI_PP = 32, EPI_Y = 64, EPI_BIAS = 128, EPI_FULL = 256 };
template <bool EX, typename Args>
__device__ __forceinline__ int epi_flags(const Args& p) {
  int f = (p.residual ? EPI_RES : 0) | (p.relu ? EPI_RELU : 0) | (p.y ? EPI_Y : 0) | (p.bias ? EPI_BIAS : 0);
  if constexpr (EX) f |= (p.ex.mask_bits ? EPI_MB : 0) | (p.ex.relu_bits ? EPI_RB : 0) | (p.ex.pool_partial ? EPI_PP : 0);
  else f |= p.mask_ref ? EPI_MK : 0;
  return f;
}

// SPL (bf16x3 "split" tensors, split.hip): y and the residual are [row][2][ldy] bf16 -- plane 0 = bf16(v), plane 1 = bf16(v - plane 0),
// 4 bytes per element like fp32 -- and mask_ref is a split tensor of p.mask_pitch elements per row whose plane 0 carries the sign.
template <int FA, int FB, bool EX, int RB, bool PM = false, bool SPL = false, int FL = -1, typename Put, typename Args>
__device__ __forceinline__ void epilogue_rows_bf16_blocks(Put put_block, char* scr, float* pool, int m_w, int n_w, const Args& p, int lane, const PmRows* rows = nullptr,
                                                          unsigned long long* stamp = nullptr) {
  typedef EpiCfg<FA, RB> E;
  static_assert(!(EX && SPL), "the fused pool / bit-mask epilogue has no split form");
  constexpr int NBLK = FB * 16 / RB;
  bf16_t* __restrict__ Y = (bf16_t*)p.y;
  const bf16_t* __restrict__ Rz = (FL >= 0 && !(FL & EPI_RES)) ? nullptr : (const bf16_t*)p.residual;
  const bf16_t* __restrict__ Mk = (EX || (FL >= 0 && !(FL & EPI_MK))) ? nullptr : (const bf16_t*)p.mask_ref;
  const bool has_res = FL >= 0 ? (FL & EPI_RES) != 0 : Rz != nullptr;
  const bool has_mk = FL >= 0 ? (!EX && (FL & EPI_MK) != 0) : Mk != nullptr;
  const bool do_relu = FL >= 0 ? (FL & EPI_RELU) != 0 : p.relu != 0;
  const bool has_y = FL >= 0 ? (FL & EPI_Y) != 0 : Y != nullptr;
  constexpr bool FULLT = FL >= 0 && (FL & EPI_FULL) != 0;
  static_assert(!(FULLT && PM), "position-class tiles carry their own row validity");
  const bool plain = EX || (p.oy_mul == 1 && p.OHf == p.OH && p.OWf == p.OW);
  const long pitch = SPL ? 2L * p.ldy : (long)p.ldy;           // elements per output (and residual) row
  long mpitch = pitch;                                         // ... per mask_ref row
  if constexpr (SPL) mpitch = p.mask_pitch;
  const int rr = lane / E::LPR, c0 = (lane % E::LPR) * 8;
  const int n = n_w + c0;
  const bool n_ok = FULLT || n < p.ldy;
  float bias8[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias8[j] = ((FL < 0 || (FL & EPI_BIAS)) && p.bias && (FULLT || n + j < p.K)) ? p.bias[n + j] : 0.f;
  const bool has_bias = FL >= 0 ? (FL & EPI_BIAS) != 0 : p.bias != nullptr;      // (the accumulators start at +0: no -0 that "+ 0.f" would have had to turn)

  const unsigned char* __restrict__ Mb = nullptr;
  unsigned char* __restrict__ Rb = nullptr;
  float* __restrict__ Pp = nullptr;
  bool has_mb = false, has_rb = false, has_pp = false;
  int prow = 1, roi0 = 0, cur_roi = 0, nxt_edge = 0;
  long bword = 0;
  u32x4 mw = {0u, 0u, 0u, 0u}, rw = {0u, 0u, 0u, 0u};
  float run[8];
  if constexpr (EX) {
    static_assert(FA == 4, "pooling epilogue: 64-channel wave tiles");
    static_assert(FB <= 8 && E::RPP == 8, "bit words: 16 row passes of 8 rows per wave tile");
    Mb = (FL >= 0 && !(FL & EPI_MB)) ? nullptr : p.ex.mask_bits;
    Rb = (FL >= 0 && !(FL & EPI_RB)) ? nullptr : p.ex.relu_bits;
    Pp = (FL >= 0 && !(FL & EPI_PP)) ? nullptr : p.ex.pool_partial;
    prow = p.ex.pool_rows > 0 ? p.ex.pool_rows : 1;
    has_mb = FL >= 0 ? (FL & EPI_MB) != 0 : Mb != nullptr;
    has_rb = FL >= 0 ? (FL & EPI_RB) != 0 : Rb != nullptr;
    has_pp = FL >= 0 ? (FL & EPI_PP) != 0 : Pp != nullptr;
    bword = (((long)(m_w >> 7)) * (p.ldy >> 6) + (n_w >> 6)) * 64 + lane;          // this lane's word of the wave tile (m_w % 128 == 0)
    if (has_mb && n_ok && m_w < p.M) mw = reinterpret_cast<const u32x4*>(Mb)[bword];
    if (has_pp) {
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(pool + (i * 64 + lane) * 4) = z;      // 4 x 8 x 64 floats
      roi0 = m_w / prow;
      int m_first = m_w + rr;                       // this lane's first row
      cur_roi = m_first / prow;
      nxt_edge = (cur_roi + 1) * prow;
#pragma unroll
      for (int j = 0; j < 8; ++j) run[j] = 0.f;
    }
  }
  auto flush = [&]() {
    float* d = pool + (((cur_roi - roi0) * 8 + rr) * 64 + c0);
    *reinterpret_cast<f32x4*>(d) = f32x4{run[0], run[1], run[2], run[3]};
    *reinterpret_cast<f32x4*>(d + 4) = f32x4{run[4], run[5], run[6], run[7]};
  };

  struct Pre { long off[E::NP]; long moff[SPL ? E::NP : 1]; bool ok[E::NP]; bf16x8 res[E::NP], msk[E::NP]; bf16x8 res2[SPL ? E::NP : 1]; };
  auto prefetch = [&](int b, Pre& q) {
#pragma unroll
    for (int h = 0; h < E::NP; ++h) {
      int m = m_w + b * RB + h * E::RPP + rr;
      q.ok[h] = FULLT || (n_ok && m < p.M);
      int mm = q.ok[h] ? m : 0;
      if constexpr (PM) {
        int img, oh, ow;
        q.ok[h] = rows->map(m, img, oh, ow) && n_ok;
        long orow = (long)(q.ok[h] ? img : 0) * rows->OHW + oh * rows->OW + ow;
        q.off[h] = orow * pitch + n;
        if constexpr (SPL) q.moff[h] = orow * mpitch + n;
      } else
      if (plain) {
        q.off[h] = (long)mm * pitch + n;
        if constexpr (SPL) q.moff[h] = (long)mm * mpitch + n;
      } else {
        int ow = mm % p.OW; int t = mm / p.OW; int oh = t % p.OH; int nimg = t / p.OH;
        long orow = ((long)nimg * p.OHf + (long)oh * p.oy_mul) * p.OWf + (long)ow * p.oy_mul;
        q.off[h] = orow * pitch + n;
        if constexpr (SPL) q.moff[h] = orow * mpitch + n;
      }
      if (has_res && q.ok[h]) q.res[h] = epi_load8(Rz + q.off[h]);
      if constexpr (SPL) {
        if (has_res && q.ok[h]) q.res2[h] = epi_load8(Rz + q.off[h] + p.ldy);
        if (has_mk && q.ok[h]) q.msk[h] = epi_load8(Mk + q.moff[h]);
      } else {
        if (has_mk && q.ok[h]) q.msk[h] = epi_load8(Mk + q.off[h]);
      }
    }
  };

  Pre cur;
  prefetch(0, cur);
  EPI_STAMP(0);
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    Pre nxt;
    if (b + 1 < NBLK) prefetch(b + 1, nxt);
    put_block(b, scr);
#pragma unroll
    for (int h = 0; h < E::NP; ++h) {
      int r = h * E::RPP + rr;
      f32x4 v0 = *reinterpret_cast<const f32x4*>(scr + r * E::PITCH + c0 * 4);
      f32x4 v1 = *reinterpret_cast<const f32x4*>(scr + r * E::PITCH + c0 * 4 + 16);
      // (acc + bias) + residual as packed fp32 pairs (v_pk_add_f32: two adds per instruction)
      epi_f32x2 t2[4] = {{v0[0], v0[1]}, {v0[2], v0[3]}, {v1[0], v1[1]}, {v1[2], v1[3]}};
      if (has_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) t2[i] += epi_f32x2{bias8[2 * i], bias8[2 * i + 1]};
      }
      if (has_res && !SPL) {
        const u32x4 rz = __builtin_bit_cast(u32x4, cur.res[h]);
#pragma unroll
        for (int i = 0; i < 4; ++i) t2[i] += epi_f32x2{epi_lo(rz[i]), epi_hi(rz[i])};
      }
      if constexpr (SPL) {
        if (has_res) {
          const u32x4 rz = __builtin_bit_cast(u32x4, cur.res[h]), rz2 = __builtin_bit_cast(u32x4, cur.res2[h]);
#pragma unroll
          for (int i = 0; i < 4; ++i)      // (hi + lo is exact in fp32)
            t2[i] += epi_f32x2{epi_lo(rz[i]), epi_hi(rz[i])} + epi_f32x2{epi_lo(rz2[i]), epi_hi(rz2[i])};
        }
      }
      float v[8] = {t2[0][0], t2[0][1], t2[1][0], t2[1][1], t2[2][0], t2[2][1], t2[3][0], t2[3][1]};
      // max(., 0) commutes with the rounding: where nothing sits between them it runs on the packed result (4 instructions instead of 8)
      const bool relu_packed = UNIT_EPI_SLIM && !SPL && do_relu && !has_mk && !has_mb;
      if (do_relu && !relu_packed) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = epi_relu(v[j]);
      }
      if (has_mk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)cur.msk[h][j] > 0.f ? v[j] : 0.f;
      }
      if constexpr (EX) {
        if (has_mb) {
          unsigned mbits = mw[(b * E::NP + h) >> 2] >> (((b * E::NP + h) & 3) * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = epi_keep_if_bit(v[j], mbits, j);
        }
      }
#if UNIT_EPI_SLIM
      u32x4 ow;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ow[i] = epi_cvt_pk(v[2 * i], v[2 * i + 1]);
        if (relu_packed) ow[i] = epi_relu_pk(ow[i]);
      }
      const bf16x8 o = __builtin_bit_cast(bf16x8, ow);
#else
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
#endif
      if constexpr (EX) {
        if (has_rb && cur.ok[h]) {
#if UNIT_EPI_SLIM
          unsigned bits = epi_positive_bits_pk(ow, do_relu);      // of the stored (rounded) value, as a mask_ref read would see it
#else
          unsigned bits = epi_positive_bits(o);
#endif
          rw[(b * E::NP + h) >> 2] |= bits << (((b * E::NP + h) & 3) * 8);
        }
        if (has_pp && cur.ok[h]) {
          int m = m_w + b * RB + r;
          if (m >= nxt_edge) {                     // this lane's rows entered the next RoI (rows only grow: at most 3 times)
            flush();
            cur_roi = m / prow; nxt_edge = (cur_roi + 1) * prow;
#pragma unroll
            for (int j = 0; j < 8; ++j) run[j] = 0.f;
          }
#if UNIT_EPI_SLIM
#pragma unroll
          for (int i = 0; i < 4; ++i) { run[2 * i] += epi_lo(ow[i]); run[2 * i + 1] += epi_hi(ow[i]); }
#else
#pragma unroll
          for (int j = 0; j < 8; ++j) run[j] += (float)o[j];
#endif
        }
        if (has_y && cur.ok[h]) epi_store8(Y + cur.off[h], o);
      } else if constexpr (SPL) {
#if UNIT_EPI_SLIM
        u32x4 ow2;
#pragma unroll
        for (int i = 0; i < 4; ++i) ow2[i] = epi_cvt_pk(v[2 * i] - epi_lo(ow[i]), v[2 * i + 1] - epi_hi(ow[i]));
        const bf16x8 o2 = __builtin_bit_cast(bf16x8, ow2);
#else
        bf16x8 o2;
#pragma unroll
        for (int j = 0; j < 8; ++j) o2[j] = (bf16_t)(v[j] - (float)o[j]);
#endif
        if (cur.ok[h]) { epi_store8(Y + cur.off[h], o); epi_store8(Y + cur.off[h] + p.ldy, o2); }
      } else {
        if (cur.ok[h]) epi_store8(Y + cur.off[h], o);
      }
    }
    if (b + 1 < NBLK) cur = nxt;
    EPI_STAMP(1 + b);
  }
  if constexpr (EX) {
    if (has_rb && n_ok && m_w < p.M) reinterpret_cast<u32x4*>(Rb)[bword] = rw;      // (a wave tile past the last row owns no word)
    if (has_pp) {
      if (n_ok && m_w + rr < p.M) flush();
      // row classes -> one sum per (segment, channel), fixed order; lane = channel of the wave tile
      int last_row = m_w + FB * 16 - 1; if (last_row >= p.M) last_row = p.M - 1;
      int nseg = last_row >= m_w ? last_row / prow - roi0 + 1 : 0;
      long wt = (long)(m_w / (FB * 16));
      if (n_w + lane < p.ldy) {
        for (int sgm = 0; sgm < nseg; ++sgm) {
          float t = 0.f;
#pragma unroll
          for (int q = 0; q < 8; ++q) t += pool[(sgm * 8 + q) * 64 + lane];
          Pp[(wt * 4 + sgm) * p.ldy + n_w + lane] = t;
        }
      }
    }
  }
}

// 16x16 MFMA accumulators: acc[a][b] = 16x16 tile (channels a*16.., pixel rows b*16..); lane holds 4 consecutive channels of row lane & 15
template <int FA, int FB, bool EX, bool PM = false, bool SPL = false, int FL = -1, typename Args>
__device__ __forceinline__ void epilogue_rows_bf16_impl(const f32x4 (&acc)[FA][FB], char* scr, float* pool, int m_w, int n_w, const Args& p, int lane,
                                                        const PmRows* rows = nullptr, unsigned long long* stamp = nullptr) {
  typedef EpiCfg<FA, 16> E;
  const int frow = lane & 15, fq = lane >> 4;
  auto put = [&](int b, char* sc) {
#pragma unroll
    for (int a = 0; a < FA; ++a)
      *reinterpret_cast<f32x4*>(sc + frow * E::PITCH + (a * 16 + fq * 4) * 4) = acc[a][b];
  };
  epilogue_rows_bf16_blocks<FA, FB, EX, 16, PM, SPL, FL>(put, scr, pool, m_w, n_w, p, lane, rows, stamp);
}

// the switch combinations named in FLS... run their straight-line instantiation, anything else the dynamic form
template <int FA, int FB, bool EX, bool PM, bool SPL, int... FLS, typename Args>
__device__ __forceinline__ void epilogue_rows_bf16_dispatch(const f32x4 (&acc)[FA][FB], char* scr, float* pool, int m_w, int n_w, const Args& p, int lane,
                                                            const PmRows* rows = nullptr, unsigned long long* stamp = nullptr) {
#if UNIT_EPI_SLIM
  int fl = SPL ? -2 : epi_flags<EX>(p);          // (split outputs: the straight-line forms measured 2 % SLOWER over the bf16x3 step, 40.15 -> 40.85-41.26 ms: dynamic form)
  if (!PM && m_w + FB * 16 <= p.M && n_w + FA * 16 <= p.K && n_w + FA * 16 <= p.ldy) fl |= EPI_FULL;
  bool done = false;
  if constexpr (!SPL)
    (void)((fl == FLS ? (epilogue_rows_bf16_impl<FA, FB, EX, PM, SPL, FLS>(acc, scr, pool, m_w, n_w, p, lane, rows, stamp), done = true) : false) || ...);
  if (done) return;
#endif
  epilogue_rows_bf16_impl<FA, FB, EX, PM, SPL, -1>(acc, scr, pool, m_w, n_w, p, lane, rows, stamp);
}

template <int FA, int FB, bool SPL = false, typename Args>
__device__ __forceinline__ void epilogue_rows_bf16(const f32x4 (&acc)[FA][FB], char* scr, int m_w, int n_w, const Args& p, int lane) {
  epilogue_rows_bf16_impl<FA, FB, false, false, SPL>(acc, scr, nullptr, m_w, n_w, p, lane);
}

// the same with straight-line passes for the backbone's combinations on whole tiles (conv_igemm128.hip, conv_igemm_lc.hip): conv + folded
// FrozenBN (+ shortcut) + ReLU forward, dgrad with the ReLU mask of the layer input (+ the shortcut's gradient)
template <int FA, int FB, bool SPL = false, typename Args>
__device__ __forceinline__ void epilogue_rows_bf16_fast(const f32x4 (&acc)[FA][FB], char* scr, int m_w, int n_w, const Args& p, int lane) {
  epilogue_rows_bf16_dispatch<FA, FB, false, false, SPL, EPI_FULL | EPI_BIAS | EPI_RELU | EPI_Y, EPI_FULL | EPI_BIAS | EPI_RES | EPI_RELU | EPI_Y,
                                   EPI_FULL | EPI_BIAS | EPI_Y, EPI_FULL | EPI_MK | EPI_Y, EPI_FULL | EPI_RES | EPI_MK | EPI_Y, EPI_FULL | EPI_Y>(
      acc, scr, nullptr, m_w, n_w, p, lane);
}

// 32x32 MFMA accumulators (v_mfma_f32_32x32x16_bf16, A = channels, B = pixels): acc[a][b] = 32 channels a*32.. x 32 pixel rows
// b*32..; lane holds, for row lane & 31, the channels 8*g + 4*(lane >> 5) + (0..3) in registers 4g .. 4g+3 (g = 0..3)
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int FA2, int FB2, bool EX, typename Args>
__device__ __forceinline__ void epilogue_rows32_bf16_impl(const f32x16 (&acc)[FA2][FB2], char* scr, float* pool, int m_w, int n_w, const Args& p, int lane) {
  typedef EpiCfg<FA2 * 2, 32> E;
  const int frow = lane & 31, fh = lane >> 5;
  auto put = [&](int b, char* sc) {
#pragma unroll
    for (int a = 0; a < FA2; ++a)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(sc + frow * E::PITCH + (a * 32 + g * 8 + fh * 4) * 4) =
            f32x4{acc[a][b][g * 4], acc[a][b][g * 4 + 1], acc[a][b][g * 4 + 2], acc[a][b][g * 4 + 3]};
  };
  epilogue_rows_bf16_blocks<FA2 * 2, FB2 * 2, EX, 32>(put, scr, pool, m_w, n_w, p, lane);
}
