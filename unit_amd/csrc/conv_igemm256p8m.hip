// conv_igemm256p8m.hip -- the 256 x 256 x 64 implicit-GEMM conv tile of conv_igemm256p8.hip (8 waves of 128 pixels x 64 channels,
// four phases per k-tile, half-tile staging by LDS-DMA under a counted vmcnt, wave groups half a phase apart, fragment reads inside
// the MFMA sections) on v_mfma_f32_32x32x16_bf16 instead of v_mfma_f32_16x16x32_bf16.
//
// What changes and what does not:
//   * LDS image, staging addresses, swizzle, slot order, WAR / RAW distances: those of conv_igemm256p8.hip, unchanged. The
//     16-B-chunk XOR swizzle (row >> 1) & 7 is also conflict-free for 32-row fragments: a ds_read_b128 is served in groups of 16
//     lanes {0-3,12-15,20-27} / {4-11,16-19,28-31} (+32), whose rows carry all eight values of (row >> 1) & 7 in both parities
//     of row & 1 -- sixteen different 16-byte slots of the 256-byte bank line.
//   * A fragment = 32 rows x 16 k: lane l reads row l & 31, 16-B chunk (l >> 5) + 2 * ks of the row, ks = 0..3 (four k-substeps of 16
//     per 64-deep k-tile). A quadrant (64 pixels x 32 channels x 64 k) is 2 x 1 x 4 = 8 MFMAs of 32 cycles where the 16x16x32 form
//     issues 16 of 16 cycles: the same 256 MFMA cycles and the same 12 ds_read_b128 per wave and phase (LDS bytes depend on the wave
//     tile, not on the MFMA shape), but half as many MFMA issues: an MFMA holds the SIMD's vector issue for 8 of its cycles
//     (MI355X_MICROARCH.md), i.e. 25 % instead of 50 % of the section -- issue slots the partner wave's LOAD section (address VALU + LDS-DMA
//     issue) and this wave's own fragment reads run in.
//   * Accumulators: acc[a][b] = f32x16, channels a*32.. x pixels b*32.. (a < 2, b < 4): 128 VGPRs as before. Lane l holds pixel
//     l & 31, channels 8g + 4(l >> 5) + (0..3) in registers 4g..4g+3: four consecutive channels per pixel as in the 16x16 layout,
//     so the row-major epilogue (conv_epilogue.h) only changes its scratch write: 32-row blocks, four 8-row passes each.
//   * The k-summation order inside a k-tile differs (16-deep substeps), so results are NOT bit-identical to conv_igemm256p8.hip;
//     both accumulate in fp32 and are tested against the same references at the same tolerances.
#include "conv_igemm256.h"
#include "conv_epilogue.h"

#define MFMA32_BF16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)

template <typename TO>
__global__ void __launch_bounds__(512, 2) conv_igemm256_p8m_kernel(Conv256Args p) {
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int HALF = 128 * 128;               // 16 KB half-tile
  constexpr int SX0 = 0, SW0 = HALF, SW1 = 2 * HALF, SX1 = 3 * HALF, BUF = 4 * HALF;
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
  const bool dual = p.x2 != nullptr;           // dual-source 1x1 conv (Conv256Args::x2)
  const int Cx = dual ? p.cb_split * 64 : p.C;
  __amdgpu_buffer_rsrc_t rsX2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>((const bf16_t*)(dual ? p.x2 : p.x)), 0, (int)(dual ? p.x2_bytes : p.x_bytes), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;

  int tid = threadIdx.x, lane = tid & 63;
  int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wm = wid >> 2, wn = wid & 3;
  const int grp = wm;                          // waves 4-7 run half a phase behind waves 0-3
  int lrow = lane >> 3, lc = lane & 7;

  // staging descriptors (conv_igemm256p8.hip): half q, piece j of this wave = half-tile rows (j*8 + wid)*8 .. +8
  int x_ih0[4], x_iw0[4]; unsigned x_off0[4], w_off[4];
  const unsigned sw16 = (unsigned)((lc ^ ((((wid * 8 + lrow) >> 1)) & 7)) * 16);
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int R = (j * 8 + wid) * 8 + lrow;
      int sw = lc ^ ((R >> 1) & 7);
      int m = m0 + (R >> 6) * 128 + q * 64 + (R & 63);
      bool ok = m < p.M;
      int mm = ok ? m : 0;
      int ow = mm % p.OW; int t = mm / p.OW; int oh = t % p.OH; int n = t / p.OH;
      int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
      x_off0[q * 2 + j] = ((unsigned)n * (unsigned)(p.H * p.W * Cx) + (unsigned)((ih0 * p.W + iw0) * Cx + sw * 8)) * 2u;
      x_ih0[q * 2 + j] = ok ? ih0 : -(1 << 20);
      x_iw0[q * 2 + j] = iw0;
      int nn = n0 + (R >> 5) * 64 + q * 32 + (R & 31);
      w_off[q * 2 + j] = nn < p.K ? ((unsigned)nn * (unsigned)p.Kgemm + (unsigned)sw * 8u) * 2u : OOB;
    }

  int st_cb = 0, st_r = 0, st_s = 0;
  unsigned st_kx = 0, st_kw = 0;
  auto st_advance = [&]() {
    if (++st_s == p.S) { st_s = 0; if (++st_r == p.R) { st_r = 0; ++st_cb; } }
    st_kx = (unsigned)((st_r * p.W + st_s) * Cx + st_cb * BK) * 2u;
    st_kw = (unsigned)((st_r * p.S + st_s) * p.C + st_cb * BK) * 2u;
  };
  auto stage_x = [&](int q, int d) {
    char* base = smem + d * BUF + (q ? SX1 : SX0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int ih = x_ih0[q * 2 + j] + st_r, iw = x_iw0[q * 2 + j] + st_s;
      bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      if (dual && st_cb >= p.cb_split) {
        unsigned o2 = (x_off0[q * 2 + j] - sw16) * (unsigned)p.ratio2 + sw16 + (unsigned)((st_cb - p.cb_split) * BK) * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX2, (lds_void*)(base + (j * 8 + wid) * 1024), 16, ok ? o2 : OOB, 0, 0, 0);
      } else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void*)(base + (j * 8 + wid) * 1024), 16, ok ? x_off0[q * 2 + j] + st_kx : OOB, 0, 0, 0);
    }
  };
  auto stage_w = [&](int q, int d) {
    char* base = smem + d * BUF + (q ? SW1 : SW0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned o = w_off[q * 2 + j];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(base + (j * 8 + wid) * 1024), 16, o == OOB ? OOB : o + st_kw, 0, 0, 0);
    }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  const int nk = p.Kgemm / BK;
  const int frow = lane & 31, fh = lane >> 5;
  // per-lane fragment offset inside a half-tile, k-substep 0; substep ks = logical chunk fh + 2*ks = byte offset ^ (ks * 32)
  const int fsw = (fh ^ ((frow >> 1) & 7)) << 4;
  const int offx = (wm * 64 + frow) * 128 + fsw;
  const int offw = (wn * 32 + frow) * 128 + fsw;

  i32x4 fx[2][4], fxb[2][4], fw0[4], fw1[4];
  auto read_x = [&](const char* half, i32x4 (&f)[2][4]) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) f[b][ks] = *reinterpret_cast<const i32x4*>(half + b * 4096 + (offx ^ (ks * 32)));
  };
  auto read_w = [&](const char* half, i32x4 (&f)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const i32x4*>(half + (offw ^ (ks * 32)));
  };
#define P8_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
  // MFMA section of a phase: quadrant (QX, QW) = acc[QW][QX*2 .. +2]; NR ds_read_b128 of the NEXT phase's fragments are spread over
  // the 8 MFMA gaps (at most two per gap: MI355X_MICROARCH.md, LDS -- a third per 32x32x16 gap would saturate the LDS array)
#define P8_GAP(NR, I) do { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           \
      if ((NR) > (I)) __builtin_amdgcn_sched_group_barrier(0x100, ((NR) + 7 - (I)) / 8 > 0 ? ((NR) + 7 - (I)) / 8 : 1, 0); } while (0)
#define P8_MM(QX, QW, FW, FX, NR, READS)                                               \
    do {                                                                               \
      __builtin_amdgcn_s_setprio(1);                                                   \
      READS;                                                                           \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                 \
        _Pragma("unroll") for (int b = 0; b < 2; ++b)                                  \
          acc[(QW)][(QX) * 2 + b] = MFMA32_BF16(FW[ks], FX[b][ks], acc[(QW)][(QX) * 2 + b]); \
      P8_GAP(NR, 0); P8_GAP(NR, 1); P8_GAP(NR, 2); P8_GAP(NR, 3);                      \
      P8_GAP(NR, 4); P8_GAP(NR, 5); P8_GAP(NR, 6); P8_GAP(NR, 7);                      \
      __builtin_amdgcn_s_setprio(0);                                                   \
      if ((NR) > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 \
    } while (0)

  // Schedule of k-tile t (d = t & 1), conv_igemm256p8.hip RM:
  //   phase 0: stage X1(t+1) ; (0,0) = fx  x fw0 || read W1(t) -> fw1     phase 1: stage X0(t+2) ; (0,1) = fx  x fw1 || read X1(t) -> fxb
  //   phase 2: stage W0(t+2), vmcnt wait ; (1,0) = fxb x fw0               phase 3: stage W1(t+2) ; (1,1) = fxb x fw1 || read X0(t+1), W0(t+1)
  stage_x(0, 0); stage_w(0, 0); stage_w(1, 0); stage_x(1, 0);
  st_advance();
  if (nk > 1) {
    stage_x(0, 1); stage_w(0, 1); stage_w(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  P8_BAR();
  read_w(smem + SW0, fw0);
  read_x(smem + SX0, fx);
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
    P8_BAR();
    P8_MM(0, 0, fw0, fx, 4, read_w(buf + SW1, fw1));
    P8_BAR();
    // phase 1
    if (n2) stage_x(0, d);
    P8_BAR();
    P8_MM(0, 1, fw1, fx, 8, read_x(buf + SX1, fxb));
    P8_BAR();
    // phase 2
    if (n2) {
      stage_w(0, d);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    P8_BAR();
    P8_MM(1, 0, fw0, fxb, 0, (void)0);
    P8_BAR();
    // phase 3 (after the last k-tile the reads fetch stale, in-bounds LDS that nobody uses)
    if (n2) stage_w(1, d);
    P8_BAR();
    P8_MM(1, 1, fw1, fxb, 12, read_w(bnx + SW0, fw0); read_x(bnx + SX0, fx));
    P8_BAR();
  }
  if (grp == 0) P8_BAR();
#undef P8_MM
#undef P8_GAP
#undef P8_BAR

  if constexpr (sizeof(TO) == 2) {
    if ((p.ldy & 7) == 0) {          // row-major epilogue through a wave-private LDS scratch (conv_epilogue.h), 32-row blocks
      __syncthreads();               // every wave is done with the operand stages
      typedef EpiCfg<4, 32> E;
      if (p.ex_on) {                 // fused average pool / ReLU bit mask / bit-mask input (unit_conv2d_fwd_big_ex)
        epilogue_rows32_bf16_impl<2, 4, true>(acc, smem + wid * E::BYTES, (float*)(smem + 8 * E::BYTES + wid * 8192), m0 + wm * 128, n0 + wn * 64, p, lane);
        return;
      }
      epilogue_rows32_bf16_impl<2, 4, false>(acc, smem + wid * E::BYTES, nullptr, m0 + wm * 128, n0 + wn * 64, p, lane);
      return;
    }
  }
  // generic epilogue (fp32 output, strided scatter with ldy % 8 != 0): straight from the accumulator layout
  TO* __restrict__ Y = (TO*)p.y;
  const TO* __restrict__ Rz = (const TO*)p.residual;
  const TO* __restrict__ Mk = (const TO*)p.mask_ref;
  bool plain = (p.oy_mul == 1 && p.OHf == p.OH && p.OWf == p.OW);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    int m = m0 + wm * 128 + b * 32 + frow;
    if (m >= p.M) continue;
    long off;
    if (plain) off = (long)m * p.ldy;
    else {
      int ow = m % p.OW; int t = m / p.OW; int oh = t % p.OH; int n = t / p.OH;
      off = (((long)n * p.OHf + (long)oh * p.oy_mul) * p.OWf + (long)ow * p.oy_mul) * p.ldy;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int n = n0 + wn * 64 + a * 32 + g * 8 + fh * 4;
        if (n >= p.ldy) continue;
        float v[4] = {acc[a][b][g * 4], acc[a][b][g * 4 + 1], acc[a][b][g * 4 + 2], acc[a][b][g * 4 + 3]};
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
static int launch256_p8m(Conv256Args& a, hipStream_t st) {
  a.tiles_m = cdiv(a.M, 256); a.tiles_n = cdiv(a.K, 256);
  // operand stages 128 KB; the epilogue reuses them: 8 x 8704 B scratch + 8 x 8 KB pooling areas = 135 168 B
  size_t lds = 8 * EpiCfg<4, 32>::BYTES + 8 * 8192;
  static_assert(8 * EpiCfg<4, 32>::BYTES + 8 * 8192 >= 8 * 128 * 128, "LDS covers the operand stages");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_igemm256_p8m_kernel<TO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  conv_igemm256_p8m_kernel<TO><<<a.tiles_m * a.tiles_n, 512, lds, st>>>(a);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

int unit_conv256_p8m_launch(Conv256Args& a, int out_dtype, hipStream_t st) {
  if (out_dtype == UNIT_BF16) return launch256_p8m<bf16_t>(a, st);
  if (out_dtype == UNIT_F32) return launch256_p8m<float>(a, st);
  unit_set_error("conv_big: unsupported out dtype");
  return UNIT_ERR_UNSUPPORTED;
}
