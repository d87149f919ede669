// conv_wgrad256r.hip -- the 256 x 256 weight-gradient tile of conv_wgrad256.hip with a RING of four 32-pixel stages
// (4 x 32 KB) instead of two 64-pixel stages, and a counted vmcnt instead of a drained one.
//
// Why (tools/exp_w8.sh, profiles/r01_exp_p8_schedule.txt): this kernel streams x / dy rows that are fetched from HBM /
// Infinity Cache once per XCD; its loop is bound by miss latency x the LDS bytes in flight, not by LDS-DMA issue, fragment
// reads or MFMAs (1.3 PF-equivalent with the DMA removed). The two-stage loop issues the next 64-pixel stage at the top of
// a step and waits vmcnt(0) at the bottom: the stage has one step's compute time (~1.2 us at the MFMA rate) to land, an
// HBM miss under load takes longer, and the difference is a stall in every step. Here stage t+3 is issued when stage t
// starts being multiplied: 96 KB stay in flight, every stage has three stage-times to land, and the wait at the top of a
// stage (`vmcnt(8)`: the two younger stages of this wave stay in flight) is normally already satisfied.
// One workgroup barrier per 32-pixel stage:
//   RAW  every wave waits for its own pieces of stage t (vmcnt) before the barrier; reads of stage t come after it.
//   WAR  stage t+3 goes into the buffer of stage t-1, whose fragment reads were consumed by the MFMAs of stage t-1, which
//        every wave issued before it arrived at this barrier.
// Same operand layout, swizzle, fragment permutation and accumulation order as conv_wgrad256.hip: bit-identical slabs.
#include "conv_wgrad256.h"

__device__ __forceinline__ bf16x8 tr_frag32(const char* tile, int col0, int lane) {
  // conv_wgrad256.hip tr_frag for a 32-row stage (sub = 0)
  int g = lane >> 4, i = lane & 15, q = i >> 2, pq = i & 3;
  int row = 4 * g + q;
  int sw = (((col0 >> 4) ^ (row & 7)) << 5) + 8 * pq;
  const char* a0 = tile + row * 512 + sw;
  s16x4 lo = ds_tr16(a0);
  s16x4 hi = ds_tr16(a0 + 16 * 512);
  s16x8_w v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ void __launch_bounds__(512, 2) conv_wgrad256_ring_kernel(Wgrad256Args p) {
  constexpr int MS = 32, NS = 4;
  constexpr int TILE = MS * 512;               // 16 KB per operand per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];

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

  // staging: wave `wid`, piece i (0..1) covers stage rows R0 = (i*8 + wid)*2, R0+1 ; lane -> row R0 + (lane>>5),
  // physical 16-B chunk lane&31 ; logical source chunk = 32-B block index XOR (row & 7), 16-B half kept
  int s_row[2]; unsigned s_col[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row = (i * 8 + wid) * 2 + (lane >> 5);
    int jp = lane & 31;
    int j = ((((jp >> 1) ^ (row & 7)) << 1) | (jp & 1));
    s_row[i] = row; s_col[i] = (unsigned)j * 8u;
  }

  // maps at least 32 pixels wide (the RPN conv: 63): the (image, oh, ow) triple of the two staged rows is carried incrementally
  // (+32 pixels per stage, stages are issued in order) instead of being divided out of m for every piece
  const bool incremental = !pointwise && p.OW >= MS;
  int in_[2] = {0, 0}, ioh[2] = {0, 0}, iow[2] = {0, 0};
  if (incremental) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned um = (unsigned)(m_begin + s_row[i]);
      unsigned ow = um % (unsigned)p.OW, tt = um / (unsigned)p.OW;
      iow[i] = (int)ow; ioh[i] = (int)(tt % (unsigned)p.OH); in_[i] = (int)(tt / (unsigned)p.OH);
    }
  }
  auto stage = [&](int mstep, int buf) {      // 4 LDS-DMA pieces per wave
    char* bx = smem + buf * 2 * TILE;
    char* bd = bx + TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int R0 = (i * 8 + wid) * 2;
      int m = mstep + s_row[i];
      bool mok = m < m_end;
      unsigned xoff;
      bool ok = mok;
      if (pointwise) xoff = ((unsigned)m * (unsigned)p.x_pitch + (unsigned)ch0 + s_col[i]) * 2u;
      else if (incremental) {
        int ih = ioh[i] * p.stride - p.pad + kr, iw = iow[i] * p.stride - p.pad + ksx;
        ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        xoff = ((unsigned)in_[i] * (unsigned)(p.H * p.W * p.x_pitch) + (unsigned)((ih * p.W + iw) * p.x_pitch + ch0) + s_col[i]) * 2u;
        iow[i] += MS;
        if (iow[i] >= p.OW) { iow[i] -= p.OW; ioh[i] += 1; if (ioh[i] >= p.OH) { ioh[i] = 0; in_[i] += 1; } }
      } else {
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

  const int nst = (m_end - m_begin + MS - 1) / MS;     // 32-pixel stages (the two-stage kernel's sub-steps, in the same order)
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nst) stage(m_begin + s * MS, s);
  for (int st = 0; st < nst; ++st) {
    // my pieces of stage st have landed once at most the pieces of the younger stages that were actually issued are pending
    int younger = min(NS - 2, nst - 1 - st);
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (st + NS - 1 < nst) stage(m_begin + (st + NS - 1) * MS, (st + NS - 1) & (NS - 1));
    const char* bx = smem + (st & (NS - 1)) * 2 * TILE;
    const char* bd = bx + TILE;
    bf16x8 fa[8], fb[4];
#pragma unroll
    for (int a = 0; a < 8; ++a) fa[a] = tr_frag32(bx, wk * 128 + a * 16, lane);
#pragma unroll
    for (int b = 0; b < 4; ++b) fb[b] = tr_frag32(bd, wn * 64 + b * 16, lane);
    tr_wait(fa); tr_wait(fb);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
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

int unit_wgrad256_ring_launch(const Wgrad256Args& a, hipStream_t st) {
  size_t lds = 4 * 2 * 32 * 512;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad256_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  conv_wgrad256_ring_kernel<<<a.tiles_k * a.tiles_n * a.splits, 512, lds, st>>>(a);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
