// conv_igemm.hip -- NHWC implicit-GEMM convolution on CDNA4 MFMA (gfx950), forward and dgrad.
//
// Replaces the cuDNN/ATen conv calls the reference reaches through Detectron2 (ResNet-C4 backbone
// configs/VOC/VOC-RCNN-101-C4-split1.yaml:6-10, RPN head modeling/proposal_generator/rpn.py:24, Res5 heads
// modeling/roi_heads/box_head.py:65-80, Linear predictors modeling/roi_heads/fast_rcnn.py:386-387).
//
// GEMM view: D[n][m] = sum_k Wt[n][k] * X[m][k],  m = output pixel (img,oh,ow), n = output channel, k = (r,s,c).
//   * MFMA A operand (rows)  = weights  [n][k]  (k contiguous: [K][R][S][C])
//   * MFMA B operand (cols)  = im2col(x)[m][k]  (k contiguous inside one (r,s): NHWC)
//   => each lane ends up with 4 consecutive output channels of one pixel: 8 B (bf16) / 16 B (fp32) vector epilogue.
// bf16 inputs: v_mfma_f32_16x16x32_bf16 ; fp32 inputs (parity mode): v_mfma_f32_16x16x4_f32 (exact fp32 fma chain).
// Tiles: 256 threads = 4 waves (2x2), per wave TM x TN MFMA tiles, BK = 128 bytes of k per row per step,
// LDS rows of 128 B with a 16-B-chunk XOR swizzle (chunk ^= (row>>1)&7) -> conflict-free ds_read_b128 fragments,
// register-staged double buffering (global_load_dwordx4 of step t+1 issued before the MFMAs of step t, written to the
// other LDS buffer after them, one barrier per step).
// Fused epilogue: + bias[n] (FrozenBN shift / conv bias) + residual, ReLU or ReLU-mask (dgrad), strided scatter
// (1x1 stride-2 dgrad writes every other pixel of a pre-zeroed tensor).
#include "common.h"
#include "conv_epilogue.h"
#include "conv_pair.h"

struct ConvArgs {
  const void* x; const void* w; void* y;
  const float* bias; const void* residual; const void* mask_ref;
  int N, H, W, C;
  int K, R, S, stride, pad;
  int OH, OW;
  int ldy, oy_mul, OHf, OWf;
  int relu;
  int Kgemm;   // R*S*C
  int M;       // N*OH*OW
  int tiles_m, tiles_n;
  unsigned x_bytes, w_bytes;
  ConvSecond second;  // conv_epilogue.h: pair launches
};

template <typename T> struct ElemsPerChunk { static constexpr int v = 16 / sizeof(T); };

template <typename TI> struct Mma;
template <> struct Mma<bf16_t> {
  // one 16-B chunk per lane = 8 bf16 = the whole k-fragment of v_mfma_f32_16x16x32_bf16
  static __device__ __forceinline__ void run(const i32x4& a, const i32x4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // one 16-B chunk per lane = 4 floats: 4 x v_mfma_f32_16x16x4_f32, element i of every lane forms one k-slice
  static __device__ __forceinline__ void run(const i32x4& a, const i32x4& b, f32x4& acc) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[i], acc, 0, 0, 0);
  }
};

template <typename TO> struct Out4;
template <> struct Out4<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p); v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    f32x4 a = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(p) = a;
  }
};
template <> struct Out4<bf16_t> {
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

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename TI, typename TO, int TM, int TN, bool PAIR = false>
__global__ void __launch_bounds__(256, 2) conv_igemm_kernel(ConvArgs p) {
  constexpr int BM = 2 * TM * 16;   // pixels per block
  constexpr int BN = 2 * TN * 16;   // channels per block
  constexpr int EPC = ElemsPerChunk<TI>::v;
  constexpr int BK = 8 * EPC;       // elements of k per step (128 bytes)
  constexpr int XL = BM / 32;       // 16-B chunks of X per thread per step
  constexpr int WL = BN / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BUF_BYTES = (BM + BN) * 128;

  // XCD-aware tile order: blocks b, b+8, ... share an XCD/L2 -> give each XCD a contiguous run of tiles, n fastest.
  int bid = blockIdx.x;
  if constexpr (PAIR) pair_enter(p, bid);
  int nwg = p.tiles_m * p.tiles_n;
  {
    int q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
  int m0 = tile_m * BM, n0 = tile_n * BN;

  const TI* __restrict__ X = (const TI*)p.x;
  const TI* __restrict__ Wt = (const TI*)p.w;
  int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int wm = wid >> 1, wn = wid & 1;
  int lc = tid & 7, lr = tid >> 3;   // this thread's chunk column / first row in the staging pattern

  // buffer descriptors: out-of-range voffset (predicated-off lanes) returns 0 -> im2col zero padding for free,
  // no branches around the loads (cdna guide section 5 trap (c)), 32-bit offsets.
  constexpr unsigned OOB = 0xFFFFFFF0u;
  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<TI*>(X), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<TI*>(Wt), 0, (int)p.w_bytes, 0x00020000);
  // per-row (pixel) decomposition for the X loads of this thread
  int x_ih0[XL], x_iw0[XL]; unsigned x_base[XL]; bool x_ok[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    int m = m0 + lr + 32 * i;
    x_ok[i] = m < p.M;
    int mm = x_ok[i] ? m : 0;
    int ow = mm % p.OW; int t = mm / p.OW; int oh = t % p.OH; int n = t / p.OH;
    x_ih0[i] = oh * p.stride - p.pad; x_iw0[i] = ow * p.stride - p.pad;
    x_base[i] = (unsigned)n * (unsigned)(p.H * p.W * p.C);
  }
  unsigned w_base[WL]; bool w_ok[WL];
#pragma unroll
  for (int i = 0; i < WL; ++i) {
    int n = n0 + lr + 32 * i;
    w_ok[i] = n < p.K;
    w_base[i] = (unsigned)(w_ok[i] ? n : 0) * (unsigned)p.Kgemm;
  }

  i32x4 rx[XL], rw[WL];
  auto gload = [&](int kt) {
    int k = kt * BK + lc * EPC;
    bool kok = k < p.Kgemm;
    int rs = k / p.C; int ch = k - rs * p.C; int r = rs / p.S; int s = rs - r * p.S;
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      int ih = x_ih0[i] + r, iw = x_iw0[i] + s;
      bool ok = kok && x_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      unsigned off = (x_base[i] + (unsigned)((ih * p.W + iw) * p.C + ch)) * (unsigned)sizeof(TI);
      rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? off : OOB, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      unsigned off = (w_base[i] + (unsigned)k) * (unsigned)sizeof(TI);
      rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, (kok && w_ok[i]) ? off : OOB, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XL; ++i) *reinterpret_cast<i32x4*>(smem + buf * BUF_BYTES + swz(lr + 32 * i, lc)) = rx[i];
#pragma unroll
    for (int i = 0; i < WL; ++i) *reinterpret_cast<i32x4*>(smem + buf * BUF_BYTES + BM * 128 + swz(lr + 32 * i, lc)) = rw[i];
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  int nk = (p.Kgemm + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  int frow = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      i32x4 fa[TN], fb[TM];
#pragma unroll
      for (int a = 0; a < TN; ++a) fa[a] = *reinterpret_cast<const i32x4*>(smem + buf * BUF_BYTES + BM * 128 + swz(wn * TN * 16 + a * 16 + frow, ks * 4 + fq));
#pragma unroll
      for (int b = 0; b < TM; ++b) fb[b] = *reinterpret_cast<const i32x4*>(smem + buf * BUF_BYTES + swz(wm * TM * 16 + b * 16 + frow, ks * 4 + fq));
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) Mma<TI>::run(fa[a], fb[b], acc[a][b]);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds channels n..n+3 of pixel m for every (a,b) tile
  TO* __restrict__ Y = (TO*)p.y;
  const TO* __restrict__ Rz = (const TO*)p.residual;
  const TO* __restrict__ Mk = (const TO*)p.mask_ref;
  bool plain = (p.oy_mul == 1 && p.OHf == p.OH && p.OWf == p.OW);
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    int m = m0 + wm * TM * 16 + b * 16 + frow;
    if (m >= p.M) continue;
    long off;
    if (plain) off = (long)m * p.ldy;
    else {
      int ow = m % p.OW; int t = m / p.OW; int oh = t % p.OH; int n = t / p.OH;
      off = (((long)n * p.OHf + (long)oh * p.oy_mul) * p.OWf + (long)ow * p.oy_mul) * p.ldy;
    }
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      int n = n0 + wn * TN * 16 + a * 16 + fq * 4;
      if (n >= p.ldy) continue;
      float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
      if (p.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += (n + j < p.K) ? p.bias[n + j] : 0.f;
      }
      if (Rz) {
        float rr[4]; Out4<TO>::load(Rz + off + n, rr);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += rr[j];
      }
      if (p.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (Mk) {
        float mm[4]; Out4<TO>::load(Mk + off + n, mm);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = mm[j] > 0.f ? v[j] : 0.f;
      }
      Out4<TO>::store(Y + off + n, v);
    }
  }
}

template <typename TI, typename TO, int TM, int TN>
static int launch_conv(ConvArgs& a, hipStream_t st) {
  constexpr int BM = 2 * TM * 16, BN = 2 * TN * 16;
  a.tiles_m = cdiv(a.M, BM); a.tiles_n = cdiv(a.K, BN);
  size_t lds = (size_t)(BM + BN) * 128 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<TI, TO, TM, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<TI, TO, TM, TN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  if (a.second.on) {          // pair launch: the second problem's tiles follow the first's
    a.second.tiles_m = cdiv(a.second.M, BM);
    a.second.tiles0 = a.tiles_m * a.tiles_n;
    conv_igemm_kernel<TI, TO, TM, TN, true><<<(a.tiles_m + a.second.tiles_m) * a.tiles_n, 256, lds, st>>>(a);
  } else
  conv_igemm_kernel<TI, TO, TM, TN><<<a.tiles_m * a.tiles_n, 256, lds, st>>>(a);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

template <typename TI, typename TO>
static int dispatch_tile(ConvArgs& a, int tile_cfg, hipStream_t st) {
  // tile_cfg: 0 auto, 1 = 128x128, 2 = 64(m)x128(n), 3 = 128(m)x64(n), 4 = 64x64
  if (tile_cfg == 0) {
    long M = (long)a.M + (a.second.on ? a.second.M : 0);          // a pair launch fills the chip with both problems' tiles
    long t128 = (long)cdiv(M, 128) * cdiv(a.K, 128);
    if (a.K <= 64) tile_cfg = ((long)cdiv(M, 128) >= 384) ? 3 : 4;
    else if (t128 >= 384) tile_cfg = 1;
    else if ((long)cdiv(M, 64) * cdiv(a.K, 128) >= 256) tile_cfg = 2;
    else tile_cfg = 4;
  }
  switch (tile_cfg) {
    case 1: return launch_conv<TI, TO, 4, 4>(a, st);
    case 2: return launch_conv<TI, TO, 2, 4>(a, st);
    case 3: return launch_conv<TI, TO, 4, 2>(a, st);
    default: return launch_conv<TI, TO, 2, 2>(a, st);
  }
}

// C ABI -------------------------------------------------------------------------------------------------
// x [N,H,W,C] (in_dtype), w [K][R][S][C] (in_dtype), y addressed as pixel (n, oh*oy_mul, ow*oy_mul) of [N,OHf,OWf,ldy]
// (out_dtype). bias fp32[K] or null; residual / mask_ref: same addressing and dtype as y, or null.
extern "C" int unit_conv2d_fwd(const void* x, const void* w, void* y, const float* bias, const void* residual,
                               const void* mask_ref, int in_dtype, int out_dtype, int N, int H, int W, int C, int K,
                               int R, int S, int stride, int pad, int OH, int OW, int ldy, int oy_mul, int OHf, int OWf,
                               int relu, int tile_cfg, void* stream) {
  return unit_conv_generic_impl(x, w, y, bias, residual, mask_ref, in_dtype, out_dtype, N, H, W, C, K, R, S, stride, pad, OH, OW, ldy, oy_mul, OHf, OWf,
                                relu, tile_cfg, nullptr, stream);
}

int unit_conv_generic_impl(const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref, int in_dtype,
                           int out_dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int OH, int OW, int ldy, int oy_mul,
                           int OHf, int OWf, int relu, int tile_cfg, const UnitConvSecond* second, void* stream) {
  int epc = in_dtype == UNIT_BF16 ? 8 : 4;
  UNIT_CHECK_ARG(C % epc == 0, "conv: C must be a multiple of 8 (bf16) / 4 (fp32)");
  UNIT_CHECK_ARG(ldy % 4 == 0 && ldy >= K, "conv: ldy must be a multiple of 4 and >= K");
  UNIT_CHECK_ARG(OH == (H + 2 * pad - R) / stride + 1 && OW == (W + 2 * pad - S) / stride + 1, "conv: OH/OW mismatch");
  UNIT_CHECK_ARG((OH - 1) * oy_mul < OHf && (OW - 1) * oy_mul < OWf, "conv: output scatter out of range");
  UNIT_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)y % 16 == 0), "conv: 16B alignment");
  ConvArgs a;
  a.x = x; a.w = w; a.y = y; a.bias = bias; a.residual = residual; a.mask_ref = mask_ref;
  a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad;
  a.OH = OH; a.OW = OW; a.ldy = ldy; a.oy_mul = oy_mul; a.OHf = OHf; a.OWf = OWf; a.relu = relu;
  a.Kgemm = R * S * C; a.M = N * OH * OW;
  size_t esz = in_dtype == UNIT_BF16 ? 2 : 4;
  size_t xb = (size_t)N * H * W * C * esz, wb = (size_t)K * R * S * C * esz;
  UNIT_CHECK_ARG(xb < 0xFFFFFFF0ull && wb < 0xFFFFFFF0ull, "conv: operand larger than 4 GiB (32-bit buffer offsets)");
  a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
  { int rc = unit_fill_second(a.second, second, R, S, stride, pad, oy_mul, (size_t)C * esz); if (rc != UNIT_OK) return rc; }
  if (K == 0 || (a.M == 0 && !a.second.on)) return UNIT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == UNIT_BF16 && out_dtype == UNIT_BF16) return dispatch_tile<bf16_t, bf16_t>(a, tile_cfg, st);
  if (in_dtype == UNIT_BF16 && out_dtype == UNIT_F32) return dispatch_tile<bf16_t, float>(a, tile_cfg, st);
  if (in_dtype == UNIT_F32 && out_dtype == UNIT_F32) return dispatch_tile<float, float>(a, tile_cfg, st);
  unit_set_error("conv: unsupported dtype combination");
  return UNIT_ERR_UNSUPPORTED;
}
