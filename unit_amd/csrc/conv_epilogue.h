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

template <int FA> struct EpiCfg {
  static constexpr int CH = FA * 16;            // channels of the wave tile
  static constexpr int LPR = CH / 8;            // lanes per row (8 channels each)
  static constexpr int RPP = 64 / LPR;          // rows per pass
  static constexpr int NP = 16 / RPP;           // passes per 16-row block
  static constexpr int PITCH = CH * 4 + 16;     // bytes per scratch row (fp32 + 16 B pad: rows land on different banks)
  static constexpr int BYTES = 16 * PITCH;      // scratch per wave
};

// acc[a][b] = 16x16 tile (channels a*16.., pixel rows b*16..) of this wave; m_w / n_w = first pixel row / channel of the
// wave tile; scr = this wave's scratch (EpiCfg<FA>::BYTES, 16-B aligned). Requires p.ldy % 8 == 0.
template <int FA, int FB, typename Args>
__device__ __forceinline__ void epilogue_rows_bf16(const f32x4 (&acc)[FA][FB], char* scr, int m_w, int n_w, const Args& p, int lane) {
  typedef EpiCfg<FA> E;
  bf16_t* __restrict__ Y = (bf16_t*)p.y;
  const bf16_t* __restrict__ Rz = (const bf16_t*)p.residual;
  const bf16_t* __restrict__ Mk = (const bf16_t*)p.mask_ref;
  const bool plain = (p.oy_mul == 1 && p.OHf == p.OH && p.OWf == p.OW);
  const int frow = lane & 15, fq = lane >> 4;
  const int rr = lane / E::LPR, c0 = (lane % E::LPR) * 8;
  const int n = n_w + c0;
  const bool n_ok = n < p.ldy;
  float bias8[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias8[j] = (p.bias && n + j < p.K) ? p.bias[n + j] : 0.f;

  struct Pre { long off[E::NP]; bool ok[E::NP]; bf16x8 res[E::NP], msk[E::NP]; };
  auto prefetch = [&](int b, Pre& q) {
#pragma unroll
    for (int h = 0; h < E::NP; ++h) {
      int m = m_w + b * 16 + h * E::RPP + rr;
      q.ok[h] = n_ok && m < p.M;
      int mm = q.ok[h] ? m : 0;
      if (plain) q.off[h] = (long)mm * p.ldy + n;
      else {
        int ow = mm % p.OW; int t = mm / p.OW; int oh = t % p.OH; int nimg = t / p.OH;
        q.off[h] = (((long)nimg * p.OHf + (long)oh * p.oy_mul) * p.OWf + (long)ow * p.oy_mul) * p.ldy + n;
      }
      if (Rz && q.ok[h]) q.res[h] = *reinterpret_cast<const bf16x8*>(Rz + q.off[h]);
      if (Mk && q.ok[h]) q.msk[h] = *reinterpret_cast<const bf16x8*>(Mk + q.off[h]);
    }
  };

  Pre cur;
  prefetch(0, cur);
#pragma unroll
  for (int b = 0; b < FB; ++b) {
    Pre nxt;
    if (b + 1 < FB) prefetch(b + 1, nxt);
#pragma unroll
    for (int a = 0; a < FA; ++a)
      *reinterpret_cast<f32x4*>(scr + frow * E::PITCH + (a * 16 + fq * 4) * 4) = acc[a][b];
#pragma unroll
    for (int h = 0; h < E::NP; ++h) {
      int r = h * E::RPP + rr;
      f32x4 v0 = *reinterpret_cast<const f32x4*>(scr + r * E::PITCH + c0 * 4);
      f32x4 v1 = *reinterpret_cast<const f32x4*>(scr + r * E::PITCH + c0 * 4 + 16);
      float v[8] = {v0[0] + bias8[0], v0[1] + bias8[1], v0[2] + bias8[2], v0[3] + bias8[3],
                    v1[0] + bias8[4], v1[1] + bias8[5], v1[2] + bias8[6], v1[3] + bias8[7]};
      if (Rz) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)cur.res[h][j];
      }
      if (p.relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (Mk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)cur.msk[h][j] > 0.f ? v[j] : 0.f;
      }
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
      if (cur.ok[h]) *reinterpret_cast<bf16x8*>(Y + cur.off[h]) = o;
    }
    if (b + 1 < FB) cur = nxt;
  }
}
