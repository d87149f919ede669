// conv_wgrad256.h -- argument block shared by the 256x256-tile weight-gradient kernels (conv_wgrad256.hip: two-stage loop;
// conv_wgrad256p8.hip: four phases per 64-pixel step, half-tile staging under a counted vmcnt).
#pragma once
#include "common.h"
#ifndef UNIT_SLAB_NT
#define UNIT_SLAB_NT 0      // 1: the fp32 slab tiles are stored non-temporal (tools/exp_wait.sh)
#endif

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
  // conv_wgrad256p8.hip, 3x3 s1 p1 "same" convs on small maps (conv2 of the Res5 blocks on 7x7 bins): valid_only = 1 contracts, for
  // the filter tap of a tile, only over the output pixels whose input pixel lies inside the map (a rectangle of positions per image,
  // image-major) instead of staging zero rows for the others: 18 % fewer 64-pixel steps on 7x7. The skipped rows contributed exact
  // zeros, but the fp32 partial sums associate differently (the rows of a step and of a split change): equal to the full contraction
  // within fp32 rounding, deterministic, not bit-identical to it.
  int valid_only;
  // elements per pixel row of x (== C for a plain tensor). A bf16x3 weight gradient (split x [.][2][C], split dy [.][2][K], csrc/split.hip)
  // runs as three passes of these kernels -- planes (hi, hi), (hi, lo), (lo, hi): x / dy point at the plane, x_pitch = 2 * C, ldy = 2 * K,
  // every pass writes its own slabs and the reduction adds them.
  int x_pitch;
};

// grouped launches (conv_wgrad128r.hip: 128x128 ring tiles; conv_wgrad256p8.hip: 256x256 phase-interleaved tiles): up to 20 layers,
// their units -- the tiles of a (layer, split), or (layer, filter tap, split) for valid_only layers, in chunks of at most one XCD's
// workgroup slots -- dealt to the 8 XCDs.
// Passed BY VALUE (3.4 KB of the 4 KB kernel-argument segment): the operand pointers change every step, a device-side table
// would cost a host-to-device copy per launch.
constexpr int WG_GROUP_MAX_PROBLEMS = 20;
constexpr int WG_GROUP_MAX_UNITS = 24;           // per XCD
constexpr int WG_GROUP_MAX_SPLITS = 127;
struct WgradGroupArgs {
  Wgrad256Args p[WG_GROUP_MAX_PROBLEMS];
  unsigned short unit_start[8][WG_GROUP_MAX_UNITS + 1];   // per XCD: first workgroup slot of unit i ([n_units] = the XCD's total)
  unsigned short unit_code[8][WG_GROUP_MAX_UNITS];        // problem (5 bits) | filter tap << 5 (4 bits) | split << 9
  unsigned short unit_tile0[8][WG_GROUP_MAX_UNITS];       // first tile of the unit within its (layer [, tap], split): layers with more
                                                          // tiles than an XCD has workgroup slots are cut into several units
  int n_units[8];
};
static_assert(sizeof(WgradGroupArgs) <= 4000, "kernel-argument segment is 4 KB");
int unit_wgrad128_group_launch(const WgradGroupArgs& g, int slots_per_xcd, hipStream_t st);
int unit_wgrad256_group_launch(const WgradGroupArgs& g, int slots_per_xcd, hipStream_t st);

typedef __attribute__((address_space(3))) void lds_void_w;
typedef __attribute__((ext_vector_type(8))) short s16x8_w;

// conv_wgrad256p8.hip / conv_wgrad256r.hip
int unit_wgrad256_p8_launch(const Wgrad256Args& a, hipStream_t st);
int unit_wgrad256_ring_launch(const Wgrad256Args& a, hipStream_t st);
// conv_wgrad128r.hip: 128x128 tile, LDS-DMA ring (bf16, C % 128 == 0, K % 128 == 0)
int unit_wgrad128_ring_launch(const Wgrad256Args& a, hipStream_t st);

// ds_read_b64_tr_b16 through inline asm. Reason: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the first
// __builtin_amdgcn_ds_read_tr16_b64 of every step when LDS-DMA loads are in flight (the intrinsic carries no alias information,
// so the waitcnt pass assumes it may read what the DMA is still writing) -- which silently removes the whole prefetch: every
// step then waits for the stage it has just issued. The asm form is invisible to that pass; the price is that its result is
// not tracked by lgkmcnt either: callers MUST pass the fragments through tr_wait() before the first use.
__device__ __forceinline__ s16x4 ds_tr16(const char* p) {
  s16x4 v;
  unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a) : "memory");
  return v;
}
// s_waitcnt lgkmcnt(0) that the consumers of the listed fragments depend on (so no MFMA can be scheduled above it)
template <int N>
__device__ __forceinline__ void tr_wait(bf16x8 (&f)[N]) {
  static_assert(N == 4 || N == 8 || N == 2, "fragment array size");
  if constexpr (N == 8)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) :: "memory");
  else if constexpr (N == 4)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) :: "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]) :: "memory");
}
