// conv_wgrad256.hip -- large-tile bf16 weight gradient for the big-M layers (Res5 heads: M = 50 176 pixels per 1024 RoIs).
//
//   dW[n][k] = sum_m dy[m][n] * im2col(x)[m][k]        n = out channel, k = (r,s,c), m = output pixel
// Same contract and slab layout as conv_wgrad.hip (split-M partial slabs [split][n][k], reduced later in a fixed order),
// but a 256 (k) x 256 (n) tile per workgroup: the 128x128 kernel needs 64 B/clk/CU of operand feed at the MFMA rate and
// gets ~27-37, this one needs half. 8 waves (2 along k x 4 along n), each 128 x 64 = 8 x 4 MFMA 16x16x32 tiles.
// Operands go HBM/L2 -> LDS by LDS-DMA in their memory order ([m][c] and [m][n] rows of 512 B; one 1 KB DMA piece = two rows,
// two 64 KB stages of 64 pixels each); the contraction index m is the ROW of both tiles, so the MFMA fragments (8
// consecutive m per lane) come from the transposing LDS read ds_read_b64_tr_b16. Rows are 512 B apart (all on the same
// banks), so 32-B column blocks are XOR-swizzled with (row & 7) on the DMA source side; a half-wave of the transposing
// read then touches 8 rows x 32 B on 8 different bank groups.
// Requires bf16, C % 256 == 0 (a k-tile stays inside one filter tap) and K % 256 == 0.
#include "conv_wgrad256.h"

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int sub, int col0, int lane) {
  // lane l: g = l>>4, i = l&15 = 4q+p ; reads rows (32*sub + 16h + 4g + q), cols col0 + 4p..4p+3 ; element j=4h+q' of
  // lane i <-> m-row 32*sub + 16h + 4g + q', column col0 + i   (same m permutation for both operands)
  int g = lane >> 4, i = lane & 15, q = i >> 2, pq = i & 3;
  int row = 32 * sub + 4 * g + q;
  int sw = (((col0 >> 4) ^ (row & 7)) << 5) + 8 * pq;      // (row + 16) & 7 == row & 7
  const char* a0 = tile + row * 512 + sw;
  s16x4 lo = ds_tr16(a0);
  s16x4 hi = ds_tr16(a0 + 16 * 512);
  s16x8_w v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ void __launch_bounds__(512, 2) conv_wgrad256_kernel(Wgrad256Args p) {
  constexpr int MS = 64;
  constexpr int TILE = MS * 512;               // 32 KB per operand per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware remap (workgroup b runs on XCD b % 8): every XCD gets a contiguous chunk of (split, tile_n, tile_k) ids, i.e.
  // the tiles of one or two split-M slabs. They walk the same pixel rows at the same pace, so an x / dy row block is
  // fetched from HBM once per XCD and then served to the other tiles of the slab out of that XCD's 4 MB L2.
  int bid = blockIdx.x;
  {
    int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  int tile_k = bid % p.tiles_k; int t = bid / p.tiles_k;
  int tile_n = t % p.tiles_n; int split = t / p.tiles_n;
  int k0 = tile_k * 256, n0 = tile_n * 256;
  int m_begin = split * p.m_per_split, m_end = min(p.M, m_begin + p.m_per_split);
  int rs = k0 / p.C, ch0 = k0 - rs * p.C, kr = rs / p.S, ksx = rs - kr * p.S;

  const bf16_t* __restrict__ X = (const bf16_t*)p.x;
  const bf16_t* __restrict__ DY = (const bf16_t*)p.dy;
  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(X), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(DY), 0, (int)p.dy_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;

  int tid = threadIdx.x, lane = tid & 63;
  int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wk = wid >> 2, wn = wid & 3;
  bool pointwise = (p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0);

  // staging: wave `wid`, piece i (0..3) covers tile rows R0 = (i*8 + wid)*2, R0+1 ; lane -> row R0 + (lane>>5),
  // physical 16-B chunk lane&31 ; logical source chunk = 32-B block index XOR (row & 7), 16-B half kept
  int s_row[4]; unsigned s_col[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = (i * 8 + wid) * 2 + (lane >> 5);
    int jp = lane & 31;
    int j = ((((jp >> 1) ^ (row & 7)) << 1) | (jp & 1));
    s_row[i] = row; s_col[i] = (unsigned)j * 8u;           // element offset inside the 256-wide row
  }

  auto stage = [&](int mstep, int buf) {
    char* bx = smem + buf * 2 * TILE;
    char* bd = bx + TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int R0 = (i * 8 + wid) * 2;
      int m = mstep + s_row[i];
      bool mok = m < m_end;
      unsigned xoff;
      bool ok = mok;
      if (pointwise) xoff = ((unsigned)m * (unsigned)p.x_pitch + (unsigned)ch0 + s_col[i]) * 2u;
      else {
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
        xoff = ((unsigned)n * (unsigned)(p.H * p.W * p.x_pitch) + (unsigned)((ih * p.W + iw) * p.x_pitch + ch0) + s_col[i]) * 2u;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_w*)(bx + R0 * 512), 16, ok ? xoff : OOB, 0, 0, 0);
      unsigned doff = ((unsigned)m * (unsigned)p.ldy + (unsigned)n0 + s_col[i]) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (lds_void_w*)(bd + R0 * 512), 16, mok ? doff : OOB, 0, 0, 0);
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  int nsteps = (m_end - m_begin + MS - 1) / MS;
  if (nsteps > 0) stage(m_begin, 0);
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    int buf = st & 1;
    if (st + 1 < nsteps) stage(m_begin + (st + 1) * MS, buf ^ 1);
    const char* bx = smem + buf * 2 * TILE;
    const char* bd = bx + TILE;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      bf16x8 fa[8], fb[4];
#pragma unroll
      for (int a = 0; a < 8; ++a) fa[a] = tr_frag(bx, sub, wk * 128 + a * 16, lane);
#pragma unroll
      for (int b = 0; b < 4; ++b) fb[b] = tr_frag(bd, sub, wn * 64 + b * 16, lane);
      tr_wait(fa); tr_wait(fb);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();   // vmcnt(0): this wave's DMA of the next m-step landed ; barrier: everyone's did, everyone finished reading `buf`
  }

  // epilogue: D[row = k][col = n] -> partial[split][n][k..k+3]
  float* out = p.partial + (size_t)split * p.K * p.Kgemm;
  int fq = lane >> 4, fr = lane & 15;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    int n = n0 + wn * 64 + b * 16 + fr;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      int k = k0 + wk * 128 + a * 16 + fq * 4;
      *reinterpret_cast<f32x4*>(out + (size_t)n * p.Kgemm + k) = acc[a][b];
    }
  }
}

// Schedules of this tile (all bit-identical): 1 = the two-stage kernel above; 0 = phase-interleaved (conv_wgrad256p8.hip);
// 2 = ring of four 32-pixel stages with a counted vmcnt (conv_wgrad256r.hip); 3 (default) = 0 whenever its staging needs no
// per-step divisions (pointwise layers, or maps of <= 1024 pixels whose im2col offsets come from its LDS table), 2 otherwise.
// History worth keeping: until the transposing reads went through inline asm (conv_wgrad256.h, ds_tr16) hipcc put
// `s_waitcnt vmcnt(0)` in front of the first __builtin_amdgcn_ds_read_tr16_b64 of every step -- each step waited for the stage
// it had just issued, so no schedule could prefetch, the loop looked "bound by miss latency" (147 us -> 81 us with the DMA
// removed) and the interleaved schedules measured SLOWER than the two-stage loop. With the wait gone (tools/wgrad_bench.py):
// 512->2048: two-stage 139 -> 121 us, ring 115, interleaved 113; 2048->512: 139 -> 126 / 112 / 103 us.

// shared with conv_wgrad.hip: which kernel handles a shape, and with how many split-M slabs
extern "C" int unit_wgrad_use_big(int in_dtype, long M, int K, int C, int RS) {
  if (in_dtype != UNIT_BF16 || (C % 256) != 0 || (K % 256) != 0) return 0;
  // few pixels need many tiles: with >= 96 tiles (RPN 3x3 1024->1024 on 4 images: 144) two or three split-M slabs fill the chip
  // and each workgroup still runs >= 50 steps; the backbone layers (4-16 tiles) would need ~64 slabs of 2-3 steps each
  long tiles = (long)(RS * C / 256) * (K / 256);
  return M >= 16384 || (M >= 8192 && tiles >= 96);
}

// 3x3 convs on small maps get one spare slab: the valid-only contraction of conv_wgrad256p8.hip deals its workgroups to the filter
// taps unevenly (Wgrad256Args::valid_only); where that path does not apply the spare is simply one more split
extern "C" int unit_wgrad_big_splits_base(long M, int tiles);
extern "C" int unit_wgrad_big_splits(long M, int tiles, int R, int S, int OHW) {
  int s = unit_wgrad_big_splits_base(M, tiles);
  return (R == 3 && S == 3 && OHW <= 512) ? s + 1 : s;
}

extern "C" int unit_wgrad_big_splits_base(long M, int tiles) {
  // one workgroup per CU (128 KB of LDS): 256 slots per round; >= 8 staged steps per split. Cost model (us): a workgroup
  // needs ~6 us of fixed time plus ~1.3 us per 64-pixel step; the grid runs in rounds of 256 workgroups; every workgroup
  // writes a 256 KB fp32 slab that the reduction reads back (~2 x 256 KB at ~4 TB/s).
  int maxs = (int)((M + 8 * 64 - 1) / (8 * 64));
  if (maxs < 1) maxs = 1;
  if (maxs > 64) maxs = 64;
  int best = 1; double best_cost = 1e30;
  double steps_total = (double)((M + 63) / 64);
  for (int s = 1; s <= maxs; ++s) {
    long blocks = (long)tiles * s;
    long rounds = (blocks + 255) / 256;
    double cost = rounds * (6.0 + 1.3 * steps_total / s) + (double)blocks * 2.0 * 262144.0 / 4.0e6;
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}

int unit_conv2d_wgrad_big_launch_p(const void* x, const void* dy, float* partial, int N, int H, int W, int C, int K, int R, int S,
                                   int stride, int pad, int OH, int OW, int ldy, int variant, size_t workspace_bytes, int x_pitch, size_t x_span,
                                   size_t dy_span, void* stream) {
  Wgrad256Args a;
  a.x = x; a.dy = dy; a.partial = partial; a.x_pitch = x_pitch;
  a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.OH = OH; a.OW = OW;
  a.ldy = ldy; a.Kgemm = R * S * C; a.M = N * OH * OW;
  UNIT_CHECK_ARG(ldy % 8 == 0, "wgrad_big: ldy must be a multiple of 8");
  size_t xb = x_span, db = dy_span;
  UNIT_CHECK_ARG(xb < 0xFFFFFFF0ull && db < 0xFFFFFFF0ull, "wgrad_big: operand larger than 4 GiB");
  a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)db;
  a.OHW = OH * OW;
  a.use_magic = ((unsigned long long)(a.M + 64) * (unsigned long long)a.OHW < 0xFFFFFFFFull) ? 1 : 0;
  a.magic_ohw = a.OHW > 1 ? (unsigned)((0x100000000ull + a.OHW - 1) / (unsigned long long)a.OHW) : 0xFFFFFFFFu;
  a.magic_ow = OW > 1 ? (unsigned)((0x100000000ull + OW - 1) / (unsigned long long)OW) : 0xFFFFFFFFu;
  a.tiles_k = a.Kgemm / 256; a.tiles_n = K / 256;
  a.splits = unit_wgrad_big_splits(a.M, a.tiles_k * a.tiles_n, R, S, OH * OW);
  int mps = cdiv(a.M, a.splits);
  a.m_per_split = cdiv(mps, 64) * 64;
  size_t need = (size_t)a.splits * K * a.Kgemm * sizeof(float);
  if (workspace_bytes < need) { unit_set_error("wgrad_big: workspace too small"); return UNIT_ERR_WORKSPACE; }
  // variant (include/unit_hip.h): 0 = policy, 1 = two-stage, 2 = ring, 3 = phase-interleaved over all pixels
  a.valid_only = 0;
  if (variant == 0) {
    variant = ((R == 1 && S == 1 && stride == 1 && pad == 0) || OH * OW <= 1024) ? 3 : 2;
    // 3x3 s1 p1 "same" conv on a small map: contract only over the pixels whose tap lies inside the map (Wgrad256Args::valid_only)
    if (variant == 3 && R == 3 && S == 3 && stride == 1 && pad == 1 && OH == H && OW == W && OH * OW <= 512) a.valid_only = 1;
  }
  if (variant == 3) {
    int rc = unit_wgrad256_p8_launch(a, (hipStream_t)stream);
    return rc == UNIT_OK ? a.splits : rc;
  }
  if (variant == 2) {
    int rc = unit_wgrad256_ring_launch(a, (hipStream_t)stream);
    return rc == UNIT_OK ? a.splits : rc;
  }
  size_t lds = 2 * 2 * 64 * 512;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  conv_wgrad256_kernel<<<a.tiles_k * a.tiles_n * a.splits, 512, lds, (hipStream_t)stream>>>(a);
  UNIT_LAUNCH_CHECK();
  return a.splits;
}

extern "C" int unit_conv2d_wgrad_big_launch(const void* x, const void* dy, float* partial, int N, int H, int W, int C, int K, int R, int S,
                                            int stride, int pad, int OH, int OW, int ldy, int variant, size_t workspace_bytes, void* stream) {
  return unit_conv2d_wgrad_big_launch_p(x, dy, partial, N, H, W, C, K, R, S, stride, pad, OH, OW, ldy, variant, workspace_bytes, C,
                                        (size_t)N * H * W * C * 2, (size_t)N * OH * OW * ldy * 2, stream);
}
