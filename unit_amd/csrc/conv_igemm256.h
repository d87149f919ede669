// conv_igemm256.h -- argument block and LDS layout helpers shared by the 256x256-tile implicit-GEMM conv kernels
// (conv_igemm256.hip: 8-wave kernels; conv_igemm256p8.hip: phase-interleaved schedule with counted vmcnt).
#pragma once
#include "common.h"

#include "conv_epilogue.h"

struct Conv256Args {
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
  EpiExtra ex;      // conv_epilogue.h; all-null unless launched through unit_conv2d_fwd_big_ex
  int ex_on;
  // second input tensor of a 1x1 conv (conv_igemm256p8.hip only): k-tiles >= cb_split read x2 [M][ratio2 * cb_split * 64] -- the
  // GEMM [x | x2] . [w_a ; w_b] in one launch (Bottleneck conv3 + shortcut; conv1 dgrad + shortcut dgrad). C = total channels.
  const void* x2; unsigned x2_bytes; int cb_split, ratio2;
  // position-class tiles (conv_igemm256p8.hip, RM schedule, 3x3 s1 p1 "same" convs on small maps: conv2 of the Res5 blocks and its
  // dgrad on 7x7 bins): pm_ncls > 0 = the rows are regrouped by position class (conv_epilogue.h PmClass; classes sorted by their
  // number of in-map taps, heaviest first) so that a filter tap is inside the map for ALL rows of a tile or for none -- the k-tiles of
  // taps that only read zero padding (18 % of them on 7x7) are skipped, staging and MFMAs alike. Those k-tiles added exact zeros:
  // results are bit-identical.
  int pm_ncls;
  PmClass pm_cls[9];
  // ceil(2^32 / OW), ceil(2^32 / OH) when every pixel index m satisfies m * max(OW, OH) < 2^32 (fast_div, conv_epilogue.h), else 0
  unsigned magic_ow, magic_oh;
  ConvSecond second; // conv_epilogue.h: pair launches (PAIR kernel instantiations; second.on == 0 otherwise)
  SplitK sk;         // conv_epilogue.h: bf16x3 operands (X3 kernel instantiations only; nseg == 0 otherwise)
  int mask_pitch;    // split epilogue: elements per row of mask_ref
};

// LDS image of an operand stage: [row][128 B = 64 k]; 16-B chunks XOR-swizzled with (row>>1)&7 (applied to the SOURCE
// chunk of the lane-linear LDS-DMA, undone here by the fragment reads)
__device__ __forceinline__ int swz256(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

typedef __attribute__((address_space(3))) void lds_void;

template <typename TO> struct O4;
template <> struct O4<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) { f32x4 a = *reinterpret_cast<const f32x4*>(p); v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { f32x4 a = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(p) = a; }
};
template <> struct O4<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
    bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)a[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
    bf16x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x4*>(p) = a;
  }
};

// conv_igemm256p8.hip: the 8-phase (4 per k-tile) schedule of the same tile
int unit_conv256_p8_launch(Conv256Args& a, int out_dtype, bool reads_in_mfma, bool rows224, hipStream_t st);
// conv_igemm256p8m.hip: the same schedule on v_mfma_f32_32x32x16_bf16 (256-row tiles)
int unit_conv256_p8m_launch(Conv256Args& a, int out_dtype, hipStream_t st);
// 1 = the 32x32x16 kernel is what variant 0 of unit_conv2d_fwd_big / unit_conv2d_fwd_big_ex launches (UNIT_P8M_DEFAULT)
int unit_conv256_use_m32();
