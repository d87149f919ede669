// conv_wgrad.hip -- weight gradient of the NHWC implicit-GEMM convolution on CDNA4 MFMA (gfx950).
//
//   dW[n][k] = sum_m dy[m][n] * im2col(x)[m][k]        n = out channel, k = (r,s,c), m = output pixel
// Both operands are contraction(m)-major in memory, so the MFMA fragments (8 consecutive m per lane) are produced by
// the CDNA4 transposing LDS read ds_read_b64_tr_b16 from row-major [m][k] / [m][n] LDS tiles (bf16); the fp32 parity
// path uses v_mfma_f32_16x16x4_f32 whose one-float-per-lane operands are plain ds_read_b32.
//   MFMA A (rows) = im2col(x)^T [k][m],  MFMA B (cols) = dy [m][n]  -> lane holds 4 consecutive k of one n: 16-B stores
// The m-range is split across workgroups (split-M); partial fp32 slabs go to the workspace and unit_wgrad_reduce sums
// them in a fixed order (bit-reproducible), applies the FrozenBN scale[n] fold and writes / accumulates dW [K][R][S][C].
#include "common.h"
#include "conv_wgrad256.h"

struct WgradArgs {
  const void* x; const void* dy; float* partial;
  int N, H, W, C;
  int K, R, S, stride, pad;
  int OH, OW;
  int ldy;     // dy row stride (elements)
  int Kgemm, M;
  int tiles_k, tiles_n, splits, m_per_split;
  unsigned x_bytes, dy_bytes;
  unsigned magic_ohw, magic_ow; int OHW; int use_magic;
  int x_pitch;     // elements per pixel row of x (Wgrad256Args::x_pitch)
};

template <typename TI> struct WgCfg;
template <> struct WgCfg<bf16_t> { static constexpr int ROWB = 288, MS = 64, EPC = 8; };   // 128 elems * 2 B + 32 pad
template <> struct WgCfg<float>  { static constexpr int ROWB = 576, MS = 32, EPC = 4; };   // 128 elems * 4 B + 64 pad

// fragment fetch: returns the per-lane operand for k-substep `sub` (bf16: 32 m-rows; fp32: 16 m-rows = 4 MFMAs)
template <typename TI> struct WgFrag;
template <> struct WgFrag<bf16_t> {
  typedef bf16x8 frag_t;
  // lane l: g = l>>4, i = l&15 = 4q+p ; reads rows (32*sub + 16h + 4g + q), cols col0 + 4p..4p+3 ; element j=4h+q' of
  // lane i <-> m-row 32*sub + 16h + 4g + q', column col0 + i.  (same m permutation for both operands)
  static __device__ __forceinline__ frag_t load(const char* tile, int sub, int col0, int lane) {
    int g = lane >> 4, i = lane & 15, q = i >> 2, pq = i & 3;
    const char* a0 = tile + (32 * sub + 4 * g + q) * 288 + (col0 + 4 * pq) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 16 * 288));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
  static __device__ __forceinline__ void mma(const frag_t& a, const frag_t& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
};
template <> struct WgFrag<float> {
  typedef f32x4 frag_t;   // 4 consecutive MFMA k-steps: element e <-> m-row 16*sub + 4e + (l>>4)
  static __device__ __forceinline__ frag_t load(const char* tile, int sub, int col0, int lane) {
    int g = lane >> 4, i = lane & 15;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = *reinterpret_cast<const float*>(tile + (16 * sub + 4 * e + g) * 576 + (col0 + i) * 4);
    return v;
  }
  static __device__ __forceinline__ void mma(const frag_t& a, const frag_t& b, f32x4& acc) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
  }
};

template <typename TI>
__global__ void __launch_bounds__(256, 2) conv_wgrad_kernel(WgradArgs p) {
  typedef WgCfg<TI> Cfg;
  constexpr int ROWB = Cfg::ROWB, MS = Cfg::MS, EPC = Cfg::EPC;
  constexpr int BK = 128, BN = 128;
  constexpr int CPR = 128 / EPC;                // 16-B chunks per tile row (16 bf16 / 32 fp32)
  constexpr int NL = MS * CPR / 256;            // chunks per thread per operand per step (4)
  constexpr int TILE_BYTES = MS * ROWB;
  constexpr int SUBS = 2;                       // MFMA sub-steps per staged tile (bf16: 2x32 rows, fp32: 2x16 rows)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int bid = blockIdx.x;
  int tile_k = bid % p.tiles_k; int t = bid / p.tiles_k;
  int tile_n = t % p.tiles_n; int split = t / p.tiles_n;
  int k0 = tile_k * BK, n0 = tile_n * BN;
  int m_begin = split * p.m_per_split, m_end = min(p.M, m_begin + p.m_per_split);

  const TI* __restrict__ X = (const TI*)p.x;
  const TI* __restrict__ DY = (const TI*)p.dy;
  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<TI*>(X), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<TI*>(DY), 0, (int)p.dy_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;

  int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int wk = wid >> 1, wn = wid & 1;
  int lc = tid % CPR, lr = tid / CPR;           // chunk column and first row of this thread
  constexpr int RSTEP = 256 / CPR;              // row step between this thread's chunks (16 bf16 / 8 fp32)
  // this thread's k-chunk is fixed for the whole kernel: decompose once
  int kk = k0 + lc * EPC;
  bool k_ok = kk < p.Kgemm;
  int rs = kk / p.C, ch = kk - rs * p.C, kr = rs / p.S, ksx = rs - kr * p.S;
  int nn = n0 + lc * EPC;
  bool n_ok = nn < p.K;   // K multiple of EPC assumed for full chunks; partial chunk columns are masked in the epilogue
  bool pointwise = (p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0);

  i32x4 rx[NL], rd[NL];
  auto gload = [&](int mstep) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      int m = mstep + lr + RSTEP * i;
      bool mok = m < m_end;
      unsigned xoff;
      bool ok = mok && k_ok;
      if (pointwise) xoff = ((unsigned)m * (unsigned)p.x_pitch + (unsigned)ch) * (unsigned)sizeof(TI);
      else {
        // m -> (n, oh, ow) with multiply-high "magic" division (the rows change every step here, unlike the forward
        // kernel): q = umulhi(m, ceil(2^32/d)) is exact while m*d < 2^32 (checked on the host), else one correction
        unsigned um = (unsigned)m, n, oh, ow;
        if (p.use_magic) {
          n = __umulhi(um, p.magic_ohw); unsigned rem = um - n * (unsigned)p.OHW;
          if (rem >= (unsigned)p.OHW) { rem -= p.OHW; ++n; }
          oh = __umulhi(rem, p.magic_ow); ow = rem - oh * (unsigned)p.OW;
          if (ow >= (unsigned)p.OW) { ow -= p.OW; ++oh; }
        } else {
          ow = um % (unsigned)p.OW; unsigned tt = um / (unsigned)p.OW; oh = tt % (unsigned)p.OH; n = tt / (unsigned)p.OH;
        }
        int ih = (int)oh * p.stride - p.pad + kr, iw = (int)ow * p.stride - p.pad + ksx;
        ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        xoff = ((unsigned)n * (unsigned)(p.H * p.W * p.x_pitch) + (unsigned)((ih * p.W + iw) * p.x_pitch + ch)) * (unsigned)sizeof(TI);
      }
      rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? xoff : OOB, 0, 0);
      unsigned doff = ((unsigned)m * (unsigned)p.ldy + (unsigned)nn) * (unsigned)sizeof(TI);
      rd[i] = __builtin_amdgcn_raw_buffer_load_b128(rsD, (mok && n_ok) ? doff : OOB, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
    char* bx = smem + buf * 2 * TILE_BYTES;
    char* bd = bx + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      int row = lr + RSTEP * i;
      *reinterpret_cast<i32x4*>(bx + row * ROWB + lc * 16) = rx[i];
      *reinterpret_cast<i32x4*>(bd + row * ROWB + lc * 16) = rd[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  int nsteps = (m_end - m_begin + MS - 1) / MS;
  if (nsteps > 0) {
    gload(m_begin);
    lstore(0);
  }
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    int buf = st & 1;
    if (st + 1 < nsteps) gload(m_begin + (st + 1) * MS);
    const char* bx = smem + buf * 2 * TILE_BYTES;
    const char* bd = bx + TILE_BYTES;
#pragma unroll
    for (int sub = 0; sub < SUBS; ++sub) {
      typename WgFrag<TI>::frag_t fa[4], fb[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) fa[a] = WgFrag<TI>::load(bx, sub, wk * 64 + a * 16, lane);
#pragma unroll
      for (int b = 0; b < 4; ++b) fb[b] = WgFrag<TI>::load(bd, sub, wn * 64 + b * 16, lane);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) WgFrag<TI>::mma(fa[a], fb[b], acc[a][b]);
    }
    if (st + 1 < nsteps) lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: D[row = k][col = n] -> partial[split][n][k..k+3]
  float* out = p.partial + (size_t)split * p.K * p.Kgemm;
  int fq = lane >> 4, fr = lane & 15;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    int n = n0 + wn * 64 + b * 16 + fr;
    if (n >= p.K) continue;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int k = k0 + wk * 64 + a * 16 + fq * 4;
      if (k >= p.Kgemm) continue;
      *reinterpret_cast<f32x4*>(out + (size_t)n * p.Kgemm + k) = acc[a][b];
    }
  }
}

// dW[n][k] = (accumulate ? dW : 0) + scale[n] * sum_s partial[s][n][k]     (fixed summation order)
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int splits, long KK, int Kgemm, const float* __restrict__ scale,
                                    float* __restrict__ dw, int accumulate) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= KK) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(partial + i);
  for (int sp = 1; sp < splits; ++sp) {
    f32x4 v = *reinterpret_cast<const f32x4*>(partial + (size_t)sp * KK + i);
    s += v;
  }
  if (scale) { float sc = scale[i / Kgemm]; s *= sc; }
  if (accumulate) s += *reinterpret_cast<const f32x4*>(dw + i);
  *reinterpret_cast<f32x4*>(dw + i) = s;
}

// conv_wgrad256.hip: 256x256 LDS-DMA kernel for the big-M bf16 layers (same slab layout)
extern "C" int unit_wgrad_use_big(int in_dtype, long M, int K, int C, int RS);
extern "C" int unit_wgrad_big_splits(long M, int tiles, int R, int S, int OHW);
// (x_pitch: elements per pixel row of x, x_span / dy_span: bytes from x / dy to the end of their tensors -- Wgrad256Args::x_pitch)
int unit_conv2d_wgrad_big_launch_p(const void* x, const void* dy, float* partial, int N, int H, int W, int C, int K, int R, int S,
                                   int stride, int pad, int OH, int OW, int ldy, int variant, size_t workspace_bytes, int x_pitch, size_t x_span,
                                   size_t dy_span, void* stream);

// 128x128 tile: the LDS-DMA ring kernel (conv_wgrad128r.hip) where it applies (bf16, C % 128 == 0, K % 128 == 0), variant 4 = the
// register-staged kernel below everywhere. Isolated the two are equal on the backbone shapes (tools/wgrad128_bench.py: 18.2 vs 18.5 us,
// 31.9 vs 30.4 us; RPN 3x3 306 vs 284 us -- at M = 9 576 these launches are bound by their fp32 slab store and input streaming, not by
// the loop's load latency); inside the step the ring form is 0.05-0.1 ms ahead on the same box (18.36 vs 18.44-18.48 ms).
static int choose_splits(int M, int tiles, int ms) {
  // 2 workgroups of this kernel are co-resident per CU (72 KB LDS each): 512 slots per "round" on 256 CUs. Pick the
  // split count whose grid fills whole rounds best (tile quantisation), preferring fewer splits (less slab traffic).
  // Cost model (us), fitted to tools/microbench.py on the res3/res4/RPN shapes: a workgroup needs ~4 us of fixed time
  // (launch ramp, first loads, slab store) plus ~1.0 us per staged 64-row step (operand-feed bound at this tile); the
  // grid runs in rounds of 512 workgroups; every split writes one fp32 slab of the whole dW (64 KB per tile) that the
  // reduction reads back: ~2 x 64 KB per tile and split at ~4 TB/s.
  int maxs = (M + 4 * ms - 1) / (4 * ms);   // at least 4 staged steps per split
  if (maxs < 1) maxs = 1;
  if (maxs > 64) maxs = 64;
  int best = 1; double best_cost = 1e30;
  double steps_total = (double)((M + ms - 1) / ms);
  for (int s = 1; s <= maxs; ++s) {
    long blocks = (long)tiles * s;
    long rounds = (blocks + 511) / 512;
    double per_block = 4.0 + 1.0 * (steps_total / s) * (ms / 64.0);
    double slab = (double)blocks * 2.0 * 65536.0 / 4.0e6;      // bytes / (4 TB/s) in us
    double cost = rounds * per_block + slab;
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}

extern "C" size_t unit_conv2d_wgrad_workspace_bytes(int in_dtype, int N, int OH, int OW, int K, int R, int S, int C) {
  long M = (long)N * OH * OW;
  int Kgemm = R * S * C;
  if (unit_wgrad_use_big(in_dtype, M, K, C, R * S)) return (size_t)unit_wgrad_big_splits(M, (Kgemm / 256) * (K / 256), R, S, OH * OW) * K * Kgemm * sizeof(float);
  int tiles = cdiv(Kgemm, 128) * cdiv(K, 128);
  int splits = choose_splits((int)M, tiles, in_dtype == UNIT_BF16 ? 64 : 32);
  return (size_t)splits * K * Kgemm * sizeof(float);
}

// one pass of split-M slabs into `workspace` (no reduction); returns the number of slabs written or a negative status.
// x_pitch: elements per pixel row of x; x_span / dy_span: bytes from x / dy to the end of their tensors (the buffer range of the loads).
static int wgrad_pass(const void* x, const void* dy, int in_dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int OH,
                      int OW, int ldy, int variant, void* workspace, size_t workspace_bytes, int x_pitch, size_t x_span, size_t dy_span,
                      void* stream) {
  UNIT_CHECK_ARG(variant >= 0 && variant <= 4, "wgrad: variant 0..4");
  int epc = in_dtype == UNIT_BF16 ? 8 : 4;
  UNIT_CHECK_ARG(C % epc == 0 && K % epc == 0 && ldy % epc == 0, "wgrad: C, K, ldy must be multiples of 8 (bf16) / 4 (fp32)");
  UNIT_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)dy % 16 == 0), "wgrad: 16B alignment");
  WgradArgs a;
  a.x = x; a.dy = dy; a.partial = (float*)workspace; a.x_pitch = x_pitch;
  a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.OH = OH; a.OW = OW;
  a.ldy = ldy; a.Kgemm = R * S * C; a.M = N * OH * OW;
  UNIT_CHECK_ARG(a.Kgemm % 4 == 0, "wgrad: R*S*C % 4 != 0");
  size_t xb = x_span, db = dy_span;
  UNIT_CHECK_ARG(xb < 0xFFFFFFF0ull && db < 0xFFFFFFF0ull, "wgrad: operand larger than 4 GiB");
  a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)db;
  a.OHW = OH * OW;
  a.use_magic = ((unsigned long long)(a.M + 64) * (unsigned long long)a.OHW < 0xFFFFFFFFull) ? 1 : 0;
  // ceil(2^32 / d); for d == 1 the quotient is m itself: magic 0xFFFFFFFF gives m-1 for m>0 and the correction fixes it
  a.magic_ohw = a.OHW > 1 ? (unsigned)((0x100000000ull + a.OHW - 1) / (unsigned long long)a.OHW) : 0xFFFFFFFFu;
  a.magic_ow = OW > 1 ? (unsigned)((0x100000000ull + OW - 1) / (unsigned long long)OW) : 0xFFFFFFFFu;
  hipStream_t st = (hipStream_t)stream;
  if (unit_wgrad_use_big(in_dtype, a.M, K, C, R * S))
    return unit_conv2d_wgrad_big_launch_p(x, dy, (float*)workspace, N, H, W, C, K, R, S, stride, pad, OH, OW, ldy, variant <= 3 ? variant : 0,
                                          workspace_bytes, x_pitch, x_span, dy_span, stream);
  a.tiles_k = cdiv(a.Kgemm, 128); a.tiles_n = cdiv(K, 128);
  int ms = in_dtype == UNIT_BF16 ? 64 : 32;
  a.splits = choose_splits(a.M, a.tiles_k * a.tiles_n, ms);
  int mps = cdiv(a.M > 0 ? a.M : 1, a.splits);
  a.m_per_split = cdiv(mps, ms) * ms;
  size_t need = (size_t)a.splits * K * a.Kgemm * sizeof(float);
  if (workspace_bytes < need) { unit_set_error("wgrad: workspace too small"); return UNIT_ERR_WORKSPACE; }
  int grid = a.tiles_k * a.tiles_n * a.splits;
  // bf16 layers whose tiles are full (every trainable backbone / RPN conv): LDS-DMA ring kernel (conv_wgrad128r.hip), same slabs
  if (in_dtype == UNIT_BF16 && C % 128 == 0 && K % 128 == 0 && variant != 4) {
    Wgrad256Args b;
    b.x = a.x; b.dy = a.dy; b.partial = a.partial; b.N = a.N; b.H = a.H; b.W = a.W; b.C = a.C; b.K = a.K; b.R = a.R; b.S = a.S;
    b.stride = a.stride; b.pad = a.pad; b.OH = a.OH; b.OW = a.OW; b.ldy = a.ldy; b.Kgemm = a.Kgemm; b.M = a.M;
    b.tiles_k = a.tiles_k; b.tiles_n = a.tiles_n; b.splits = a.splits; b.m_per_split = a.m_per_split;
    b.x_bytes = a.x_bytes; b.dy_bytes = a.dy_bytes; b.magic_ohw = a.magic_ohw; b.magic_ow = a.magic_ow; b.OHW = a.OHW; b.use_magic = a.use_magic; b.valid_only = 0;
    b.x_pitch = a.x_pitch;
    int rc = unit_wgrad128_ring_launch(b, st);
    if (rc != UNIT_OK) return rc;
  } else if (in_dtype == UNIT_BF16) {
    size_t lds = (size_t)2 * 2 * WgCfg<bf16_t>::MS * WgCfg<bf16_t>::ROWB;
    static bool set1 = false;
    if (!set1) { (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set1 = true; }
    conv_wgrad_kernel<bf16_t><<<grid, 256, lds, st>>>(a);
  } else {
    size_t lds = (size_t)2 * 2 * WgCfg<float>::MS * WgCfg<float>::ROWB;
    static bool set2 = false;
    if (!set2) { (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set2 = true; }
    conv_wgrad_kernel<float><<<grid, 256, lds, st>>>(a);
  }
  UNIT_LAUNCH_CHECK();
  return a.splits;
}

// x [N,H,W,C], dy [M][ldy] (pixels of the conv's OUTPUT grid, row-major), dw fp32 [K][R][S][C].
extern "C" int unit_conv2d_wgrad(const void* x, const void* dy, float* dw, const float* scale_k, int in_dtype, int N,
                                 int H, int W, int C, int K, int R, int S, int stride, int pad, int OH, int OW, int ldy,
                                 int accumulate, int variant, void* workspace, size_t workspace_bytes, void* stream) {
  UNIT_CHECK_ARG((uintptr_t)dw % 16 == 0, "wgrad: 16B alignment");
  size_t esz = in_dtype == UNIT_BF16 ? 2 : 4;
  int sp = wgrad_pass(x, dy, in_dtype, N, H, W, C, K, R, S, stride, pad, OH, OW, ldy, variant, workspace, workspace_bytes, C,
                      (size_t)N * H * W * C * esz, (size_t)N * OH * OW * ldy * esz, stream);
  if (sp < 0) return sp;
  if (dw == nullptr) return UNIT_OK;   // partial slabs only: the caller reduces later (unit_multi_wgrad_reduce)
  long KK = (long)K * R * S * C;
  wgrad_reduce_kernel<<<cdiv(KK / 4, 256), 256, 0, (hipStream_t)stream>>>((const float*)workspace, sp, KK, R * S * C, scale_k, dw, accumulate);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// bf16x3 weight gradient (csrc/split.hip): x split [N,H,W][2][C], dy split [M][2][ldy] -> dW ~ hi^T.hi + hi^T.lo + lo^T.hi as THREE passes of
// the bf16 kernels above over the planes, each pass with its own split-M slabs (3 * unit_conv2d_wgrad_splits(UNIT_BF16, ...) in all, at
// workspace + s*K*R*S*C floats like unit_conv2d_wgrad's); dw != NULL: reduced here, else the caller folds them (unit_multi_wgrad_reduce).
extern "C" int unit_conv2d_wgrad_x3(const void* x, const void* dy, float* dw, const float* scale_k, int N, int H, int W, int C, int K, int R,
                                    int S, int stride, int pad, int OH, int OW, int ldy, int accumulate, int variant, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  UNIT_CHECK_ARG((uintptr_t)dw % 16 == 0, "wgrad_x3: 16B alignment");
  UNIT_CHECK_ARG(C % 8 == 0 && ldy % 8 == 0, "wgrad_x3: C, ldy must be multiples of 8");
  const size_t xt = (size_t)N * H * W * C * 4, dt = (size_t)N * OH * OW * ldy * 4;      // bytes of the split tensors
  const size_t slab = (size_t)K * R * S * C * sizeof(float);
  const int xpl[3] = {0, 0, 1}, dpl[3] = {0, 1, 0};                                    // plane of x / dy per pass: (hi, hi), (hi, lo), (lo, hi)
  // variant bits 8-9: how many of the three passes run (0 = all; 1 = hi^T.hi only: bf16-grade products of the rounded operands, fp32 sums)
  const int npass = ((variant >> 8) & 3) ? ((variant >> 8) & 3) : 3;
  variant &= 0xff;
  int total = 0;
  for (int ps = 0; ps < npass; ++ps) {
    size_t used = (size_t)total * slab;
    if (workspace_bytes < used) { unit_set_error("wgrad_x3: workspace too small"); return UNIT_ERR_WORKSPACE; }
    const char* xp = (const char*)x + (size_t)xpl[ps] * C * 2;
    const char* dp = (const char*)dy + (size_t)dpl[ps] * ldy * 2;
    int sp = wgrad_pass(xp, dp, UNIT_BF16, N, H, W, C, K, R, S, stride, pad, OH, OW, 2 * ldy, variant, (char*)workspace + used,
                        workspace_bytes - used, 2 * C, xt - (size_t)xpl[ps] * C * 2, dt - (size_t)dpl[ps] * ldy * 2, stream);
    if (sp < 0) return sp;
    total += sp;
  }
  if (dw == nullptr) return UNIT_OK;
  long KK = (long)K * R * S * C;
  wgrad_reduce_kernel<<<cdiv(KK / 4, 256), 256, 0, (hipStream_t)stream>>>((const float*)workspace, total, KK, R * S * C, scale_k, dw, accumulate);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// number of split-M slabs unit_conv2d_wgrad writes for this shape (slab s at workspace + s*K*R*S*C floats)
extern "C" int unit_conv2d_wgrad_splits(int in_dtype, int N, int OH, int OW, int K, int R, int S, int C) {
  long M = (long)N * OH * OW;
  if (unit_wgrad_use_big(in_dtype, M, K, C, R * S)) return unit_wgrad_big_splits(M, (R * S * C / 256) * (K / 256), R, S, OH * OW);
  int tiles = cdiv(R * S * C, 128) * cdiv(K, 128);
  return choose_splits((int)M, tiles, in_dtype == UNIT_BF16 ? 64 : 32);
}

// ---- grouped launches (conv_wgrad128r.hip: conv_wgrad128_group_kernel; conv_wgrad256p8.hip: conv_wgrad256_group_kernel) -------------
// mirrors include/unit_hip.h
struct UnitWgradProblem {
  const void* x; const void* dy; void* partial;
  int N, H, W, C, K, R, S, stride, pad, OH, OW, ldy;
  int splits, kind;
  int x_pitch;            // 0 = C; one pass of a bf16x3 weight gradient: 2 * C (x, dy point at the pass's planes, ldy = 2 * K), x_back / dy_back =
  int x_back, dy_back;    // elements between the tensor's start and the x / dy pointer (0 or one plane: keeps the loads' buffer range exact)
};
extern "C" size_t unit_wgrad_problem_bytes(void) { return sizeof(UnitWgradProblem); }

// 0 = not eligible; 1 = 128x128 ring tiles; 2 = 256x256 phase-interleaved tiles (half the operand bytes per flop; needs enough pixels
// for a few 64-pixel steps per split)
extern "C" int unit_conv2d_wgrad_group_supported(int in_dtype, int N, int OH, int OW, int K, int R, int S, int C) {
  if (in_dtype != UNIT_BF16 || (C % 128) != 0 || (K % 128) != 0) return 0;
  long M = (long)N * OH * OW;
  if (M <= 0 || M > 0x7FFFFFFFl) return 0;
  if ((C % 256) == 0 && (K % 256) == 0 && M >= 2048 && (R * S * C / 256) * (K / 256) <= 4096) return 2;
  if ((R * S * C / 128) * (K / 128) > 4096) return 0;
  return 1;
}

static bool group_valid_only(const UnitWgradProblem& q) {
  return q.R == 3 && q.S == 3 && q.stride == 1 && q.pad == 1 && q.OH == q.H && q.OW == q.W && q.OH * q.OW <= 512;
}

// split counts of the layers of a grouped launch, per tile kind. All workgroups of a grid should run about the same number of pixels, so
// layer p gets s_p ~ s_ref * M_p / M_max slabs;
//  kind 1 (two workgroups per CU): s_ref from choose_splits()'s cost model over the whole grid: rounds of 512 slots, ~4 us fixed + ~1 us
//  per 64 pixels per workgroup, two transfers of every workgroup's 64 KB slab tile at ~4 TB/s;
//  kind 2 (one per CU): rounds of 256 slots with the constants measured below.
extern "C" int unit_conv2d_wgrad_group_plan(UnitWgradProblem* pr, int n, int splits_hint) {
  UNIT_CHECK_ARG(pr != nullptr && n > 0, "wgrad_group_plan: no problems");
  for (int i = 0; i < n; ++i) {
    pr[i].kind = unit_conv2d_wgrad_group_supported(UNIT_BF16, pr[i].N, pr[i].OH, pr[i].OW, pr[i].K, pr[i].R, pr[i].S, pr[i].C);
    UNIT_CHECK_ARG(pr[i].kind != 0, "wgrad_group_plan: layer not eligible (unit_conv2d_wgrad_group_supported)");
  }
  // a few 256x256 tiles alone cannot fill 256 CUs with long loops (res3's one 256 -> 512 shortcut: 2 tiles): they join the 128x128 grid
  long tiles2 = 0;
  for (int i = 0; i < n; ++i)
    if (pr[i].kind == 2) tiles2 += (long)(pr[i].R * pr[i].S * pr[i].C / 256) * (pr[i].K / 256);
  if (tiles2 < 16)
    for (int i = 0; i < n; ++i)
      if (pr[i].kind == 2) pr[i].kind = 1;
  for (int kind = 1; kind <= 2; ++kind) {
    long mmax = 0;
    for (int i = 0; i < n; ++i)
      if (pr[i].kind == kind) { long M = (long)pr[i].N * pr[i].OH * pr[i].OW; if (M > mmax) mmax = M; }
    if (mmax == 0) continue;
    const int T = kind == 2 ? 256 : 128;
    auto splits_of = [&](int i, int sref) {
      long M = (long)pr[i].N * pr[i].OH * pr[i].OW;
      long s = (M * sref + mmax / 2) / mmax;
      long maxs = (M + 255) / 256;              // >= 4 staged 64-pixel steps per split
      if (s > maxs) s = maxs;
      if (s < 1) s = 1;
      if (s > 64) s = 64;
      return (int)s;
    };
    int best = 1;
    if (splits_hint > 0) best = splits_hint;
    else if (kind == 1) {
      double best_cost = 1e30;
      for (int sref = 1; sref <= 32; ++sref) {
        long wgs = 0;
        for (int i = 0; i < n; ++i)
          if (pr[i].kind == kind) wgs += (long)(pr[i].R * pr[i].S * pr[i].C / T) * (pr[i].K / T) * splits_of(i, sref);
        long rounds = (wgs + 511) / 512;
        double cost = rounds * (4.0 + 1.0 * ((double)mmax / sref) / 64.0) + (double)wgs * 2.0 * 65536.0 / 4.0e6;
        if (cost < best_cost) { best_cost = cost; best = sref; }
      }
    } else {
      // rounds of 256 workgroups (one per CU), each ~12 us of fixed time (launch ramp, first loads, 256 KB slab store) + ~2.2 us per
      // 64-pixel step when every CU streams. Fits tools/wgrad_group_bench.py within a few percent where the loops are short -- a res4
      // bucket (102 tiles, 150 steps): 315 / 177 / 219 / 176 / 191 / 209 us measured with 1 / 2 / 3 / 4 / 6 / 8 slabs per layer, model 342 /
      // 177 / 244 / 188 / 201 / 212; the RPN's 3x3 (144 tiles, 75 steps): 180 / 188 / 134 measured with 1 / 2 / 3, model 177 / 188 / 134.
      // For the long loops of a Res5 head (236 tiles, 784 steps) the model is flat (1.74-1.82 ms) while 3-8 slabs measured 5-8 % faster
      // than 1-2 (1.49-1.55 against 1.62 ms: units of 16-36 tiles quantise on an XCD's 32 CUs until there are several rounds of them).
      // At least 24 steps per workgroup.
      long tiles = 0;
      for (int i = 0; i < n; ++i)
        if (pr[i].kind == kind) tiles += (long)(pr[i].R * pr[i].S * pr[i].C / T) * (pr[i].K / T);
      double steps = (double)((mmax + 63) / 64), tmin = 1e30;
      int smax = (int)(steps / 24.0);
      if (smax < 1) smax = 1;
      if (smax > 64) smax = 64;
      // (a round counts as full at 90 % of its slots: the units are dealt to the XCDs by weight, not evenly by workgroup count)
      auto model = [&](int sref) { return ceil((double)(tiles * sref) / (0.9 * 256.0)) * (12.0 + 2.2 * steps / sref); };
      int argmin = 1;
      for (int sref = 1; sref <= smax; ++sref)
        if (model(sref) < tmin) { tmin = model(sref); argmin = sref; }
      // every slab is read back by the reduction (a Res5 head: 60 MB each): of the split counts within 3 % of the best, the SMALLEST that
      // still gives 2.5 rounds of workgroups (3 slabs per layer for a Res5 head: 1 522 us against 1 541 with 5); none such: the model's best
      best = argmin;
      for (int sref = smax; sref >= 1; --sref)
        if (model(sref) <= 1.03 * tmin && tiles * sref >= 640) best = sref;
    }
    for (int i = 0; i < n; ++i)
      if (pr[i].kind == kind) pr[i].splits = splits_of(i, best);
  }
  return UNIT_OK;
}

static int wgrad_group_fill(const UnitWgradProblem& q, Wgrad256Args& b) {
  const int T = q.kind == 2 ? 256 : 128;
  UNIT_CHECK_ARG(q.kind == 1 || q.kind == 2, "wgrad_group: kind 1 / 2 (unit_conv2d_wgrad_group_plan)");
  UNIT_CHECK_ARG(q.C % T == 0 && q.K % T == 0 && q.ldy % 8 == 0, "wgrad_group: C, K must be multiples of the tile, ldy of 8");
  UNIT_CHECK_ARG(((uintptr_t)q.x % 16 == 0) && ((uintptr_t)q.dy % 16 == 0) && ((uintptr_t)q.partial % 16 == 0) && q.partial != nullptr,
                 "wgrad_group: 16B alignment");
  UNIT_CHECK_ARG(q.splits >= 1 && q.splits <= WG_GROUP_MAX_SPLITS, "wgrad_group: splits 1..127 (unit_conv2d_wgrad_group_plan)");
  b.x = q.x; b.dy = q.dy; b.partial = (float*)q.partial;
  b.N = q.N; b.H = q.H; b.W = q.W; b.C = q.C; b.K = q.K; b.R = q.R; b.S = q.S; b.stride = q.stride; b.pad = q.pad; b.OH = q.OH; b.OW = q.OW;
  b.ldy = q.ldy; b.Kgemm = q.R * q.S * q.C; b.M = q.N * q.OH * q.OW;
  UNIT_CHECK_ARG(b.M > 0, "wgrad_group: empty problem");
  b.x_pitch = q.x_pitch > 0 ? q.x_pitch : q.C;
  UNIT_CHECK_ARG(b.x_pitch >= q.C && b.x_pitch % 8 == 0 && q.x_back >= 0 && q.dy_back >= 0, "wgrad_group: x_pitch / x_back / dy_back");
  size_t xb = ((size_t)q.N * q.H * q.W * b.x_pitch - q.x_back) * 2, db = ((size_t)b.M * q.ldy - q.dy_back) * 2;
  UNIT_CHECK_ARG(xb < 0xFFFFFFF0ull && db < 0xFFFFFFF0ull, "wgrad_group: operand larger than 4 GiB");
  b.x_bytes = (unsigned)xb; b.dy_bytes = (unsigned)db;
  b.OHW = q.OH * q.OW;
  b.use_magic = ((unsigned long long)(b.M + 64) * (unsigned long long)b.OHW < 0xFFFFFFFFull) ? 1 : 0;
  b.magic_ohw = b.OHW > 1 ? (unsigned)((0x100000000ull + b.OHW - 1) / (unsigned long long)b.OHW) : 0xFFFFFFFFu;
  b.magic_ow = q.OW > 1 ? (unsigned)((0x100000000ull + q.OW - 1) / (unsigned long long)q.OW) : 0xFFFFFFFFu;
  b.tiles_k = b.Kgemm / T; b.tiles_n = q.K / T;
  b.splits = q.splits;
  b.m_per_split = cdiv(cdiv(b.M, q.splits), 64) * 64;
  b.valid_only = (q.kind == 2 && group_valid_only(q)) ? 1 : 0;
  UNIT_CHECK_ARG(b.tiles_k * b.tiles_n <= 4096, "wgrad_group: more than 4096 tiles in one layer");
  return UNIT_OK;
}

// units of a layer: one per split (x 9 filter taps for valid_only layers), each cut into chunks of at most one XCD's workgroup slots
// (32 CUs: one 256x256 workgroup or two 128x128 ones per CU) -- whole rows of k tiles (they share the dy columns) where a row fits
static int unit_chunk(int tiles_k, int tiles, int kind) {
  const int L = kind == 2 ? 32 : 64;
  if (tiles <= L + L / 4) return tiles;
  return tiles_k <= L ? tiles_k * (L / tiles_k) : L;
}
static int units_of(const UnitWgradProblem& q) {
  const int T = q.kind == 2 ? 256 : 128;
  int tiles_k = q.R * q.S * q.C / T, tiles_n = q.K / T;
  if (q.kind == 2 && group_valid_only(q)) {
    int per = tiles_n * (tiles_k / 9), c = unit_chunk(tiles_k / 9, per, q.kind);
    return q.splits * 9 * cdiv(per, c);
  }
  int c = unit_chunk(tiles_k, tiles_k * tiles_n, q.kind);
  return q.splits * cdiv(tiles_k * tiles_n, c);
}

// x / dy / partial as unit_conv2d_wgrad(dw = NULL) takes them, for n layers at once; pr[i].splits / kind from unit_conv2d_wgrad_group_plan.
// Slab s of layer i at partial + s*K*R*S*C floats, same layout as unit_conv2d_wgrad's (unit_multi_wgrad_reduce folds them).
// layout != nullptr: nothing is launched; one row of 9 ints per unit instead -- {launch, tile kind, XCD, first workgroup slot on that XCD, tiles, index
// into pr, filter tap, split, first tile} (unit_conv2d_wgrad_group_layout: what the CPU tests check the dealing on)
constexpr int WG_FIXED_STEPS = 6;          // a tile's fixed time in 64-pixel steps (~12 us against ~2.2 us per step, unit_conv2d_wgrad_group_plan)
static int wgrad_group_run(const UnitWgradProblem* pr, int n, void* stream, int* layout, int layout_rows, int* rows_out) {
  static thread_local WgradGroupArgs g;         // 3.8 KB; filled and passed by value
  int launch_no = 0, rows = 0, pidx[WG_GROUP_MAX_PROBLEMS];
  struct U { long w; int p, tap, s, tiles, tile0, gang; long gw; int d; };
  static thread_local U us[8 * WG_GROUP_MAX_UNITS];
  // gangs (UNIT_WGRAD_GANG, profiles/r06_exp_wgrad_gangs.txt): the nine filter-tap units of one split of a valid_only 3x3 layer contract over the
  // SAME x / dy rows; dealt one by one (0) they land on up to eight XCDs and every one of those L2s fetches the rows for itself.
  // 1: all nine on one XCD (36 tiles on 32 CUs, and the taps drift apart: 36 / 42 / 49 valid positions per 7x7 image);
  // 2 (default): the taps that walk at the same pace -- 4 corner, 4 edge, the centre tap -- form a gang: a Res5 head's grid 4.00 -> 2.67 GB past L2
  const char* gang_env = getenv("UNIT_WGRAD_GANG");          // read per call (a handful per step): the tests switch it in-process
  const int use_gangs = gang_env ? atoi(gang_env) : 2;
  // UNIT_WGRAD_DEAL: 0 = by summed weight alone (round 5), 1 = by makespan. Default: makespan for gangs, weight for single units (measured,
  // isolated Res5-head grid: gangs 1 599 / 1 625 -> 1 564 / 1 588 us; single units 1 534 / 1 495 -> 1 643 / 1 611: the model's equal step time
  // per tile is only roughly true)
  const char* deal_env = getenv("UNIT_WGRAD_DEAL");
  const bool deal_by_makespan = deal_env ? deal_env[0] != '0' : use_gangs != 0;
  for (int kind = 2; kind >= 1; --kind) {           // the long 256-tile grid first
    int i0 = 0;
    while (i0 < n) {
      // one launch: the next problems of this kind that fit WG_GROUP_MAX_PROBLEMS and the unit table
      int cnt = 0, units = 0, i = i0;
      for (; i < n; ++i) {
        if (pr[i].kind != kind) continue;
        if (cnt == WG_GROUP_MAX_PROBLEMS || units + units_of(pr[i]) > 8 * WG_GROUP_MAX_UNITS) break;
        int rc = wgrad_group_fill(pr[i], g.p[cnt]);
        if (rc != UNIT_OK) return rc;
        units += units_of(pr[i]);
        pidx[cnt] = i;
        ++cnt;
      }
      if (cnt == 0) {
        UNIT_CHECK_ARG(i >= n, "wgrad_group: a layer with more units than one grid holds");
        break;
      }
      i0 = i;
      // units (gangs of units), heaviest first, each to the least-loaded XCD that still has free unit entries
      int nu = 0, ngang = 0;
      for (int p = 0; p < cnt; ++p) {
        const Wgrad256Args& a = g.p[p];
        if (a.valid_only) {
          int ncb = a.tiles_k / 9, per = a.tiles_n * ncb, ch = unit_chunk(ncb, per, kind);
          for (int tap = 0; tap < 9; ++tap) {
            int kr = tap / 3, ks = tap % 3;
            int nh = a.OH - (kr == 1 ? 0 : 1), nw = a.OW - (ks == 1 ? 0 : 1);      // 3x3 s1 p1 "same": the border taps lose a row / column
            if (nh < 0) nh = 0;
            if (nw < 0) nw = 0;
            long meff = (long)a.N * nh * nw;
            for (int sp = 0; sp < a.splits; ++sp)
              for (int t0 = 0; t0 < per; t0 += ch) {
                int nt = per - t0 < ch ? per - t0 : ch;
                int cls = (kr != 1) + (ks != 1);            // 0 centre, 1 edge, 2 corner
                us[nu++] = U{(long)nt * (meff / a.splits + 1), p, tap, sp, nt, t0,
                             use_gangs == 1 ? ngang + sp : use_gangs == 2 ? ngang + sp * 3 + cls : -1, 0,
                             (int)(((meff + a.splits - 1) / a.splits + 63) / 64) + WG_FIXED_STEPS};
              }
          }
          ngang += a.splits * 3;
        } else {
          int tiles = a.tiles_k * a.tiles_n, ch = unit_chunk(a.tiles_k, tiles, kind);
          for (int sp = 0; sp < a.splits; ++sp) {
            int mb = sp * a.m_per_split, me = a.M < mb + a.m_per_split ? a.M : mb + a.m_per_split;
            for (int t0 = 0; t0 < tiles; t0 += ch) {
              int nt = tiles - t0 < ch ? tiles - t0 : ch;
              us[nu++] = U{(long)nt * (me > mb ? me - mb : 0), p, 0, sp, nt, t0, -1, 0, (me > mb ? (me - mb + 63) / 64 : 0) + WG_FIXED_STEPS};
            }
          }
        }
      }
      for (int a = 0; a < nu; ++a) {           // a gang is dealt as one item of its units' total weight; a lone unit is its own gang
        if (us[a].gang < 0) { us[a].gang = ngang++; us[a].gw = us[a].w; continue; }
        long t = 0;
        for (int b = 0; b < nu; ++b) if (us[b].gang == us[a].gang) t += us[b].w;
        us[a].gw = t;
      }
      // insertion sort, stable (<= 192 entries): gangs by descending weight, inside a gang the heaviest unit first (the centre tap: the units past
      // an XCD's 32 workgroup slots start late, they should be the short ones)
      auto before = [](const U& p, const U& q) { return p.gw != q.gw ? p.gw > q.gw : p.gang != q.gang ? p.gang < q.gang : p.w > q.w; };
      for (int a = 1; a < nu; ++a) {
        U v = us[a]; int b = a - 1;
        while (b >= 0 && before(v, us[b])) { us[b + 1] = us[b]; --b; }
        us[b + 1] = v;
      }
      long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      int slots[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int x = 0; x < 8; ++x) g.n_units[x] = 0;
      // the makespan rule costs ~4 ms for a Res5 head's 100 units and depends on shapes only: the last few problem lists' tables are kept
      struct Dealt { int nkey, key[3 + 12 * WG_GROUP_MAX_PROBLEMS]; unsigned short start[8][WG_GROUP_MAX_UNITS + 1], code[8][WG_GROUP_MAX_UNITS],
                     tile0[8][WG_GROUP_MAX_UNITS]; int n_units[8]; };
      static thread_local Dealt dealt[8];
      static thread_local int dealt_next = 0;
      int key[3 + 12 * WG_GROUP_MAX_PROBLEMS], nkey = 0;
      Dealt* hit = nullptr;
      if (kind == 2 && deal_by_makespan) {
        key[nkey++] = kind; key[nkey++] = cnt; key[nkey++] = use_gangs;
        for (int p = 0; p < cnt; ++p) {
          const Wgrad256Args& a = g.p[p];
          const int f[12] = {a.N, a.H, a.W, a.C, a.K, a.R, a.S, a.stride, a.pad, a.OH, a.OW, a.splits};
          for (int v : f) key[nkey++] = v;
        }
        for (int e = 0; e < 8 && !hit; ++e)
          if (dealt[e].nkey == nkey && memcmp(dealt[e].key, key, sizeof(int) * nkey) == 0) hit = &dealt[e];
      }
      if (hit) {
        memcpy(g.unit_start, hit->start, sizeof(g.unit_start)); memcpy(g.unit_code, hit->code, sizeof(g.unit_code));
        memcpy(g.unit_tile0, hit->tile0, sizeof(g.unit_tile0)); memcpy(g.n_units, hit->n_units, sizeof(g.n_units));
        for (int x = 0; x < 8; ++x) slots[x] = g.unit_start[x][g.n_units[x]];
      } else if (kind == 2 && deal_by_makespan) {
        // 256-tiles, one workgroup per CU: an XCD's 32 CUs take its slots in order, so what ends the grid is the XCD's list-scheduling makespan,
        // not its summed weight (a Res5 head: 2.7 rounds of tiles that last 265 / 228 / 196 steps -- dealt by weight alone the 16-tile gangs cost
        // 798 steps on the slowest XCD against 728, profiles/r06_exp_wgrad_gangs.txt). Every gang goes to the XCD whose makespan with it is the
        // shortest (then the lighter one), an XCD's units run longest-first. Shapes only: cached per problem list.
        static thread_local int xl[8][WG_GROUP_MAX_UNITS];
        auto makespan = [&](int x, int a0, int na) {          // XCD x's units + us[a0 .. a0 + na), longest tiles first, on 32 CUs
          int cu[32] = {0};
          int ia = 0, ib = 0, nx = g.n_units[x], worst = 0;
          while (ia < nx || ib < na) {
            const U& v = (ib >= na || (ia < nx && us[xl[x][ia]].d >= us[a0 + ib].d)) ? us[xl[x][ia++]] : us[a0 + ib++];
            for (int t = 0; t < v.tiles; ++t) {
              int m = 0;
              for (int c = 1; c < 32; ++c) if (cu[c] < cu[m]) m = c;
              cu[m] += v.d;
              if (cu[m] > worst) worst = cu[m];
            }
          }
          return worst;
        };
        for (int a = 0; a < nu;) {
          int need = 1;
          while (a + need < nu && us[a + need].gang == us[a].gang) ++need;
          for (int b = a + 1; b < a + need; ++b) {             // the gang's units longest first (they are merged into sorted lists)
            U v = us[b]; int c = b - 1;
            while (c >= a && us[c].d < v.d) { us[c + 1] = us[c]; --c; }
            us[c + 1] = v;
          }
          int bx = -1, bm = 0, take = need;
          for (;;) {
            for (int x = 0; x < 8; ++x) {
              if (g.n_units[x] + take > WG_GROUP_MAX_UNITS) continue;
              int m = makespan(x, a, take);
              if (bx < 0 || m < bm || (m == bm && load[x] < load[bx])) { bx = x; bm = m; }
            }
            if (bx >= 0 || take == 1) break;
            take = 1;                                          // no XCD has entries for the whole gang: unit by unit
          }
          UNIT_CHECK_ARG(bx >= 0, "wgrad_group: unit table full");
          for (int b = a; b < a + take; ++b) {                 // merge into the XCD's list, longest first, a gang's units adjacent
            int k = g.n_units[bx]++;
            while (k > 0 && us[xl[bx][k - 1]].d < us[b].d) { xl[bx][k] = xl[bx][k - 1]; --k; }
            xl[bx][k] = b;
            load[bx] += us[b].w;
          }
          a += take;
        }
        for (int x = 0; x < 8; ++x)
          for (int k = 0; k < g.n_units[x]; ++k) {
            const U& v = us[xl[x][k]];
            g.unit_start[x][k] = (unsigned short)slots[x];
            g.unit_code[x][k] = (unsigned short)(v.p | (v.tap << 5) | (v.s << 9));
            g.unit_tile0[x][k] = (unsigned short)v.tile0;
            slots[x] += v.tiles;
            UNIT_CHECK_ARG(slots[x] < 65536, "wgrad_group: more than 65535 workgroups on one XCD");
          }
      } else {
      int bx = -1;
      for (int a = 0; a < nu; ++a) {
        if (a == 0 || us[a].gang != us[a - 1].gang) {
          int need = 1;
          while (a + need < nu && us[a + need].gang == us[a].gang) ++need;
          bx = -1;
          for (int x = 0; x < 8; ++x)
            if (g.n_units[x] + need <= WG_GROUP_MAX_UNITS && (bx < 0 || load[x] < load[bx])) bx = x;
        }
        if (bx < 0 || g.n_units[bx] >= WG_GROUP_MAX_UNITS) {          // no XCD has room for the whole gang: unit by unit
          bx = -1;
          for (int x = 0; x < 8; ++x)
            if (g.n_units[x] < WG_GROUP_MAX_UNITS && (bx < 0 || load[x] < load[bx])) bx = x;
        }
        int k = g.n_units[bx]++;
        g.unit_start[bx][k] = (unsigned short)slots[bx];
        g.unit_code[bx][k] = (unsigned short)(us[a].p | (us[a].tap << 5) | (us[a].s << 9));
        g.unit_tile0[bx][k] = (unsigned short)us[a].tile0;
        slots[bx] += us[a].tiles; load[bx] += us[a].w;
        UNIT_CHECK_ARG(slots[bx] < 65536, "wgrad_group: more than 65535 workgroups on one XCD");
      }
      }
      int most = 0;
      for (int x = 0; x < 8; ++x) {
        g.unit_start[x][g.n_units[x]] = (unsigned short)slots[x];
        if (slots[x] > most) most = slots[x];
      }
      if (nkey > 0 && !hit) {
        Dealt& e = dealt[dealt_next++ & 7];
        e.nkey = nkey; memcpy(e.key, key, sizeof(int) * nkey);
        memcpy(e.start, g.unit_start, sizeof(g.unit_start)); memcpy(e.code, g.unit_code, sizeof(g.unit_code));
        memcpy(e.tile0, g.unit_tile0, sizeof(g.unit_tile0)); memcpy(e.n_units, g.n_units, sizeof(g.n_units));
      }
      if (layout != nullptr) {
        for (int x = 0; x < 8; ++x)
          for (int k = 0; k < g.n_units[x]; ++k) {
            UNIT_CHECK_ARG(rows < layout_rows, "wgrad_group_layout: more units than rows");
            int* o = layout + 9 * rows++;
            unsigned code = g.unit_code[x][k];
            o[0] = launch_no; o[1] = kind; o[2] = x; o[3] = g.unit_start[x][k]; o[4] = g.unit_start[x][k + 1] - g.unit_start[x][k];
            o[5] = pidx[code & 31]; o[6] = (code >> 5) & 15; o[7] = code >> 9; o[8] = g.unit_tile0[x][k];
          }
        ++launch_no;
        continue;
      }
      int rc = kind == 2 ? unit_wgrad256_group_launch(g, most, (hipStream_t)stream) : unit_wgrad128_group_launch(g, most, (hipStream_t)stream);
      if (rc != UNIT_OK) return rc;
    }
  }
  if (rows_out) *rows_out = rows;
  return UNIT_OK;
}

extern "C" int unit_conv2d_wgrad_group(const UnitWgradProblem* pr, int n, int in_dtype, void* stream) {
  UNIT_CHECK_ARG(in_dtype == UNIT_BF16, "wgrad_group: bf16 only");
  UNIT_CHECK_ARG(pr != nullptr && n >= 0, "wgrad_group: no problems");
  return wgrad_group_run(pr, n, stream, nullptr, 0, nullptr);
}

extern "C" int unit_conv2d_wgrad_group_layout(const UnitWgradProblem* pr, int n, int* layout, int layout_rows, int* rows) {
  UNIT_CHECK_ARG(pr != nullptr && n >= 0 && layout != nullptr && rows != nullptr, "wgrad_group_layout: null argument");
  return wgrad_group_run(pr, n, nullptr, layout, layout_rows, rows);
}
