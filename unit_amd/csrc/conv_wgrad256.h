// conv_wgrad256.h -- argument block shared by the 256x256-tile weight-gradient kernels (conv_wgrad256.hip: two-stage loop;
// conv_wgrad256p8.hip: four phases per 64-pixel step, half-tile staging under a counted vmcnt).
#pragma once
#include "common.h"

struct Wgrad256Args {
  const void* x; const void* dy; float* partial;
  int N, H, W, C;
  int K, R, S, stride, pad;
  int OH, OW;
  int ldy;
  int Kgemm, M;
  int tiles_k, tiles_n, splits, m_per_split;
  unsigned x_bytes, dy_bytes;
  unsigned magic_ohw, magic_ow; int OHW; int use_magic;
};

typedef __attribute__((address_space(3))) void lds_void_w;
typedef __attribute__((ext_vector_type(8))) short s16x8_w;

// conv_wgrad256p8.hip
int unit_wgrad256_p8_launch(const Wgrad256Args& a, hipStream_t st);
