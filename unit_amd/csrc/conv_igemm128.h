// conv_igemm128.h -- argument block shared by the mid-size LDS-DMA conv kernels (conv_igemm128.hip: 4-wave tiles, two workgroups per
// CU; conv_igemm_lc.hip: persistent loader / consumer workgroups)
#pragma once
#include "common.h"
#include "conv_epilogue.h"

struct ConvDmaArgs {
  const void* x; const void* w; void* y;
  const float* bias; const void* residual; const void* mask_ref;
  int N, H, W, C;
  int K, R, S, stride, pad;
  int OH, OW;
  int ldy, oy_mul, OHf, OWf;
  int relu;
  int Kgemm, M;
  int tiles_m, tiles_n;
  unsigned x_bytes, w_bytes;
  ConvSecond second; // conv_epilogue.h: pair launches (PAIR kernel instantiations; second.on == 0 otherwise)
  SplitK sk;         // conv_epilogue.h: bf16x3 operands (X3 kernel instantiations only; nseg == 0 otherwise)
  int mask_pitch;    // split epilogue: elements per row of mask_ref
};

// conv_igemm_lc.hip: tile code = 100 + 10 * (BM / 16) + (BN / 64)   (BM 64..128 pixels, BN 128 or 256 channels)
int unit_conv_lc_launch(ConvDmaArgs& a, int out_dtype, int code, hipStream_t st);
// conv_igemm128.hip: the bf16x3 instantiations of the 4-wave kernel / the loader-consumer kernel (a.sk filled by the caller)
int unit_conv_mid_x3_launch(ConvDmaArgs& a, int tile, hipStream_t st);
