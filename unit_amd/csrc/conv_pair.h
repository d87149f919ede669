// conv_pair.h -- internal entry points of the conv kernel families with an optional second problem (pair launches, conv_epilogue.h
// ConvSecond); the extern "C" functions of include/unit_hip.h are thin wrappers (second == nullptr), unit_conv2d_fwd_pair dispatches here.
#pragma once
#include "common.h"
#include "conv_epilogue.h"

struct UnitConvSecond {          // mirrors include/unit_hip.h
  const void* x; void* y; const void* residual; const void* mask_ref;
  int N, H, W, OHf, OWf;
};

// ConvSecond of a layer (R, S, stride, pad, scatter multiplier shared with the first problem) from the public descriptor; x_row_bytes = bytes of
// one input pixel row (C * element size; 4 * C for a bf16x3 split tensor)
static inline int unit_fill_second(ConvSecond& s, const UnitConvSecond* u, int R, int S, int stride, int pad, int oy_mul, size_t x_row_bytes) {
  s.on = 0; s.tiles0 = 0;
  if (u == nullptr) return UNIT_OK;
  UNIT_CHECK_ARG(u->x != nullptr && u->y != nullptr && u->N >= 0 && u->H > 0 && u->W > 0, "conv pair: second problem needs x, y and positive sizes");
  UNIT_CHECK_ARG(((uintptr_t)u->x % 16 == 0) && ((uintptr_t)u->y % 16 == 0), "conv pair: 16B alignment");
  s.x = u->x; s.y = u->y; s.residual = u->residual; s.mask_ref = u->mask_ref;
  s.N = u->N; s.H = u->H; s.W = u->W;
  s.OH = (u->H + 2 * pad - R) / stride + 1; s.OW = (u->W + 2 * pad - S) / stride + 1;
  s.OHf = u->OHf; s.OWf = u->OWf;
  UNIT_CHECK_ARG(s.OH > 0 && s.OW > 0 && (s.OH - 1) * oy_mul < s.OHf && (s.OW - 1) * oy_mul < s.OWf, "conv pair: second output scatter out of range");
  s.M = u->N * s.OH * s.OW;
  size_t xb = (size_t)u->N * u->H * u->W * x_row_bytes;
  UNIT_CHECK_ARG(xb < 0xFFFFFFF0ull, "conv pair: operand larger than 4 GiB");
  s.x_bytes = (unsigned)xb;
  unsigned long long mx = (unsigned long long)(s.M + 512) * (unsigned long long)(s.OW > s.OH ? s.OW : s.OH);
  bool ok = mx < 0xFFFFFFFFull;
  s.magic_ow = ok ? div_magic((unsigned)s.OW) : 0u; s.magic_oh = ok ? div_magic((unsigned)s.OH) : 0u;
  s.tiles_m = 0;
  s.on = s.M > 0 ? 1 : 0;
  return UNIT_OK;
}

int unit_conv_generic_impl(const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref, int in_dtype,
                           int out_dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int OH, int OW, int ldy, int oy_mul,
                           int OHf, int OWf, int relu, int tile_cfg, const UnitConvSecond* second, void* stream);
int unit_conv_mid_impl(const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref, int out_dtype, int N,
                       int H, int W, int C, int K, int R, int S, int stride, int pad, int OH, int OW, int ldy, int oy_mul, int OHf, int OWf, int relu,
                       int tile, const UnitConvSecond* second, void* stream);
int unit_conv_big_impl(const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref, int out_dtype, int N,
                       int H, int W, int C, int K, int R, int S, int stride, int pad, int OH, int OW, int ldy, int oy_mul, int OHf, int OWf, int relu,
                       int variant, const UnitConvSecond* second, void* stream);
int unit_conv_x3_impl(const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref, int mask_c, int N, int H,
                      int W, int C, int K, int R, int S, int stride, int pad, int OH, int OW, int ldy, int oy_mul, int OHf, int OWf, int relu, int tile,
                      const UnitConvSecond* second, void* stream, int segs);
