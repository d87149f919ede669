// linear_wgrad.hip -- weight and bias gradient of a group of Linear layers on the same input (the box predictors: cls_score /
// bbox_pred of the supervised head, classifier / detection streams + OICR predictors + regression branch of the weak head; the 1x1
// RPN predictors), bf16 operands, ONE launch:
//     dW[k][c] = sum_r dy[r][k] * x[r][c]        db[k] = sum_r dy[r][k]         k < K <= 128, C % 128 == 0
// Why a kernel of its own: as a "1x1 convolution" the problem has 16 tiles of 128 channels and K = 101..112 output columns that do
// not fill a 128-wide tile; the generic register-staged weight-gradient kernel + its slab reduction + the one-workgroup bias
// column sum measured 41 + 37 + 35 us per head (profiles/r04_exp_linear_wgrad.txt) on the step's critical path, with the chip idle.
// Here: workgroup = (128-channel slice, split of the rows); the rows of a split are streamed through a ring of four 32-row LDS-DMA
// stages (layout and transposing fragment reads of conv_wgrad128r.hip); wave w multiplies channels [32w, 32w+32) against all KB
// 16-column blocks of dy. Wave 0 of slice 0 also multiplies a fragment of ones against dy: the bias gradient. Each workgroup stores
// its partial tile into a slab; a second small kernel adds the slabs in split order (deterministic) into dW and db. Two launches
// instead of five (weight gradient, its reduction, two memsets, bias column sum).
// Tried first: ONE launch, the last workgroup of a slice to take an atomic ticket adds the slice's slabs. Correct, but the
// __threadfence() every workgroup needs before its ticket is a buffer_wbl2 + buffer_inv of the XCD's whole L2 on gfx950: the kernel
// took 71-78 us whatever the shape (profiles/r04_exp_linear_wgrad.txt).
#include "conv_wgrad256.h"
#include <stdlib.h>

namespace {

struct LinWgradArgs {
  const bf16_t* x; const bf16_t* dy;
  float* dw; float* db;
  float* slab;          // [splits][K][C] then [splits][128] (bias partials)
  int R, C, K, ldy, splits, rows_per_split;
  unsigned x_bytes, dy_bytes;
};

__device__ __forceinline__ bf16x8 lw_frag(const char* tile, int col0, int lane) {
  // rows 16h + 4g + q of a 32-row stage with 256-B rows, 8 B at 32-B column block (col >> 4) ^ (row & 7)   (conv_wgrad128r.hip)
  int g = lane >> 4, i = lane & 15, q = i >> 2, pq = i & 3;
  int row = 4 * g + q;
  const char* a0 = tile + row * 256 + ((((col0 >> 4) ^ (row & 7)) << 5) + 8 * pq);
  s16x4 lo = ds_tr16(a0);
  s16x4 hi = ds_tr16(a0 + 16 * 256);
  s16x8_w v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ void lw_wait(bf16x8& f) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f) :: "memory"); }

template <int KB>
__global__ void __launch_bounds__(256, 2) linear_wgrad_kernel(LinWgradArgs p) {
  constexpr int NS = 4, MS = 32, TILE = MS * 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int split = blockIdx.x % p.splits, slice = blockIdx.x / p.splits;
  const int c0 = slice * 128;
  const int m_begin = split * p.rows_per_split, m_end = min(p.R, m_begin + p.rows_per_split);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;

  // staging (conv_wgrad128r.hip): wave `wid`, piece i = stage rows (i*4 + wid)*4 .. +4; lane -> row + (lane >> 4), physical 16-B chunk
  // lane & 15 holds logical chunk ((jp >> 1) ^ (row & 7)) << 1 | (jp & 1)
  const int r0 = wid * 4 + (lane >> 4);
  const int jp = lane & 15;
  const unsigned s_col = (unsigned)((((jp >> 1) ^ (r0 & 7)) << 1) | (jp & 1)) * 8u;
  const bool dcol_ok = (int)s_col < p.ldy;       // ldy % 8 == 0: a 16-B chunk is inside the row or outside it
  int mst = m_begin;
  auto stage = [&](int buf) {
    char* bx = smem + buf * 2 * TILE;
    char* bd = bx + TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int R0 = (i * 4 + wid) * 4;
      int m = mst + r0 + 16 * i;
      bool mok = m < m_end;
      unsigned xoff = ((unsigned)m * (unsigned)p.C + (unsigned)c0 + s_col) * 2u;
      unsigned doff = ((unsigned)m * (unsigned)p.ldy + s_col) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_w*)(bx + R0 * 256), 16, mok ? xoff : OOB, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (lds_void_w*)(bd + R0 * 256), 16, (mok && dcol_ok) ? doff : OOB, 0, 0, 0);
    }
    mst += MS;
  };

  f32x4 acc[2][KB], accb[KB];
#pragma unroll
  for (int b = 0; b < KB; ++b) { acc[0][b] = acc[1][b] = f32x4{0.f, 0.f, 0.f, 0.f}; accb[b] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const bool bias_wave = (slice == 0 && wid == 0);
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)1.0f;

  const int nst = (m_end - m_begin + MS - 1) / MS;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nst) stage(s);
  for (int st = 0; st < nst; ++st) {
    int younger = min(NS - 2, nst - 1 - st);
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (st + NS - 1 < nst) stage((st + NS - 1) % NS);
    const char* bx = smem + (st % NS) * 2 * TILE;
    const char* bd = bx + TILE;
    bf16x8 fa[2], fb[KB];
    fa[0] = lw_frag(bx, wid * 32, lane);
    fa[1] = lw_frag(bx, wid * 32 + 16, lane);
#pragma unroll
    for (int b = 0; b < KB; ++b) fb[b] = lw_frag(bd, b * 16, lane);
    lw_wait(fa[0]); lw_wait(fa[1]);
#pragma unroll
    for (int b = 0; b < KB; ++b) lw_wait(fb[b]);
#pragma unroll
    for (int b = 0; b < KB; ++b) {
      acc[0][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[b], acc[0][b], 0, 0, 0);
      acc[1][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[b], acc[1][b], 0, 0, 0);
    }
    if (bias_wave) {
#pragma unroll
      for (int b = 0; b < KB; ++b) accb[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fb[b], accb[b], 0, 0, 0);
    }
  }

  // D[row = channel][col = k]: lane holds channels fq*4 .. +3 of column fr
  const int fq = lane >> 4, fr = lane & 15;
  const bool direct = (p.splits == 1);
  float* out = direct ? p.dw : p.slab + (size_t)split * p.K * p.C;
#pragma unroll
  for (int b = 0; b < KB; ++b) {
    int k = b * 16 + fr;
    if (k < p.K) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
        *reinterpret_cast<f32x4*>(out + (size_t)k * p.C + c0 + wid * 32 + a * 16 + fq * 4) = acc[a][b];
    }
  }
  if (bias_wave && fq == 0) {
    float* ob = direct ? p.db : p.slab + (size_t)p.splits * p.K * p.C + (size_t)split * 128;
#pragma unroll
    for (int b = 0; b < KB; ++b) {
      int k = b * 16 + fr;
      if (k < p.K) ob[k] = accb[b][0];
    }
  }
}

// dW = slabs added in split order; db likewise (workgroup 0's first K threads)
__global__ void __launch_bounds__(256) linear_wgrad_sum_kernel(const float* __restrict__ slab, int splits, int K, int C, float* __restrict__ dw,
                                                               float* __restrict__ db) {
  const size_t n4 = (size_t)K * C / 4, stride4 = n4;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    const f32x4* src = reinterpret_cast<const f32x4*>(slab) + i;
    f32x4 s = src[0];
#pragma unroll 8
    for (int sp = 1; sp < splits; ++sp) s += src[sp * stride4];
    reinterpret_cast<f32x4*>(dw)[i] = s;
  }
  if (blockIdx.x == 0 && threadIdx.x < K) {
    const float* src = slab + (size_t)splits * K * C + threadIdx.x;
    float s = src[0];
    for (int sp = 1; sp < splits; ++sp) s += src[sp * 128];
    db[threadIdx.x] = s;
  }
}

int pick_splits(int R, int C) {
  // ~128 workgroups (tools/linear_wgrad_ab.py: 64 / 128 / 256 / 512 tried), at least 2 stages of 32 rows per split, at most 32 splits
  static int target = -1;
  if (target < 0) { const char* e = getenv("UNIT_LINW_WORKGROUPS"); target = e ? atoi(e) : 128; }      // tuning knob
  int slices = C / 128;
  int s = (target + slices - 1) / slices;
  int max_by_rows = R / 64; if (max_by_rows < 1) max_by_rows = 1;
  if (s > max_by_rows) s = max_by_rows;
  if (s > 32) s = 32;
  return s;
}

}  // namespace

extern "C" size_t unit_linear_wgrad_workspace_bytes(int R, int C, int K) {
  int s = pick_splits(R, C);
  return ((size_t)s * K * C + (size_t)s * 128) * sizeof(float);
}

extern "C" int unit_linear_wgrad(const void* x, const void* dy, int dtype, int R, int C, int K, int ldy, float* dw, float* db,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  UNIT_CHECK_ARG(dtype == UNIT_BF16, "unit_linear_wgrad: bf16 operands only (fp32 layers use unit_conv2d_wgrad + unit_bias_grad)");
  UNIT_CHECK_ARG(R > 0 && K > 0 && K <= 128 && C > 0 && C % 128 == 0, "unit_linear_wgrad: needs K <= 128 and C % 128 == 0");
  UNIT_CHECK_ARG(ldy % 8 == 0 && ldy >= K, "unit_linear_wgrad: ldy must be a multiple of 8 and >= K");
  UNIT_CHECK_ARG((size_t)R * C * 2 < 0xFFFFFFF0ull && (size_t)R * ldy * 2 < 0xFFFFFFF0ull, "unit_linear_wgrad: operand over 4 GB");
  LinWgradArgs a;
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.dw = dw; a.db = db;
  a.R = R; a.C = C; a.K = K; a.ldy = ldy;
  a.splits = pick_splits(R, C);
  a.rows_per_split = ((R + a.splits - 1) / a.splits + 31) / 32 * 32;
  a.x_bytes = (unsigned)((size_t)R * C * 2); a.dy_bytes = (unsigned)((size_t)R * ldy * 2);
  a.slab = (float*)workspace;
  if (a.splits > 1) {
    if (workspace == nullptr || workspace_bytes < unit_linear_wgrad_workspace_bytes(R, C, K)) {
      unit_set_error("unit_linear_wgrad: workspace missing or too small");
      return UNIT_ERR_WORKSPACE;
    }
  }
  const int KB = (K + 15) / 16;
  const int grid = (C / 128) * a.splits;
  constexpr int LDS = 4 * 2 * 32 * 256;
  hipStream_t st = (hipStream_t)stream;
  static bool attr_set[9] = {};
#define LW_CASE(kb)                                                                                                             \
  case kb:                                                                                                                      \
    if (!attr_set[kb]) {                                                                                                        \
      (void)hipFuncSetAttribute((const void*)linear_wgrad_kernel<kb>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);         \
      attr_set[kb] = true;                                                                                                      \
    }                                                                                                                           \
    linear_wgrad_kernel<kb><<<grid, 256, LDS, st>>>(a);                                                                         \
    break;
  switch (KB) {
    LW_CASE(1) LW_CASE(2) LW_CASE(3) LW_CASE(4) LW_CASE(5) LW_CASE(6) LW_CASE(7) LW_CASE(8)
  }
#undef LW_CASE
  UNIT_LAUNCH_CHECK();
  if (a.splits > 1) {
    linear_wgrad_sum_kernel<<<cdiv((long)K * C / 4, 256), 256, 0, st>>>(a.slab, a.splits, K, C, dw, db);
    UNIT_LAUNCH_CHECK();
  }
  return UNIT_OK;
}
