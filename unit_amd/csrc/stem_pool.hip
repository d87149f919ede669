// stem_pool.hip -- a2: the frozen ResNet stem as ONE persistent kernel, bf16:
//     conv 7x7 stride 2 pad 3 (3 -> 64 channels, input channels padded to 8) + FrozenBN (scale folded into the weights, shift added)
//     + ReLU + max_pool2d(3, 2, 1)                       detectron2 BasicStem.forward (modeling/backbone/resnet.py) behind
//                                                        configs/VOC/VOC-RCNN-101-C4-split1.yaml:6-10
// Why: as a generic implicit GEMM the 7x7 conv gathers 49 taps x 16 B per output pixel from L2 for 64 output channels (90 us for
// 4 x 600 x 1000, 330 TFLOP/s, 1.3 TB/s), writes 77 MB that the pooling kernel reads back (22 us) to keep a quarter of it. The stem is
// frozen (FREEZE_AT = 2): nothing of the conv output is needed afterwards.
// Here: a workgroup owns a 4 x 16 tile of POOLED pixels = a 9 x 33 region of conv outputs (the pooling windows overlap by one row /
// column: 16 % of the conv outputs are computed twice) = a 23 x 71 patch of input pixels, staged once in LDS by LDS-DMA (16 B per
// pixel; pixels outside the image come back as zeros through the buffer bounds check). A wave owns two of the four 16-channel blocks and
// a quarter of the region's 19 pixel blocks; its 2 x 13 weight fragments stay in REGISTERS for the life of the (persistent) workgroup
// (first version: weights in LDS, every wave reading all four blocks per k-step: 72 us, LDS-bound by 663 KB of fragment reads per tile,
// two thirds of them weights). MFMA 16x16x32: a k-step = 4 filter taps x 8 channels, so a lane's 8 contraction values are the
// 16 B of ONE input pixel -- one ds_read_b128 straight from the patch, no im2col buffer. D = [64 channels] x [16 region pixels]; the
// lane holds 4 consecutive channels of a pixel -> +shift, ReLU, bf16, 8-B store into an LDS image of the region; after a barrier
// 512 threads take the 3 x 3 max per pooled pixel and 8 channels and store 16 B. The next tile's patch is in flight meanwhile.
// ReLU output >= 0, so a zero for conv positions outside the map is the -inf padding of max_pool2d (every window has a valid element).
#include "common.h"
#ifndef UNIT_STEM_DBG
#define UNIT_STEM_DBG 0      // diagnostic builds (tools/stem_dbg.sh): bit 0 no patch staging after the first, bit 1 no MFMA loop, bit 2 no epilogue + pooling
#endif

namespace {

typedef __attribute__((address_space(3))) void lds_void_s;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_s;

constexpr int PT_H = 4, PT_W = 16;                        // pooled tile
constexpr int CR_H = 2 * PT_H + 1, CR_W = 2 * PT_W + 1;   // conv region 9 x 33
constexpr int NPX = CR_H * CR_W;                          // 297
constexpr int NMT = (NPX + 15) / 16;                      // 19 MFMA column blocks
constexpr int IP_H = 2 * (CR_H - 1) + 7, IP_W = 2 * (CR_W - 1) + 7;      // input patch 23 x 71
constexpr int PATCH_CHUNKS = IP_H * IP_W;                 // 1633 pixels of 16 B
constexpr int PATCH_PIECES = (PATCH_CHUNKS + 63) / 64;    // 26 LDS-DMA pieces of 1 KB
constexpr int PATCH_BYTES = PATCH_PIECES * 1024;
constexpr int KTAPS = 49, KSTEPS = 13;                    // 13 x 4 taps (3 padding taps)
constexpr int W_ROW = KTAPS * 16;                         // 784 B per output channel
constexpr int CT_PITCH = 144;                             // region image row: 64 ch bf16 = 128 B + 16 B (a 128-B pitch puts the 16 pixels of an
                                                          // epilogue store on two bank groups: 8-way conflict)
constexpr int CT_BYTES = NMT * 16 * CT_PITCH;             // region image [pixel][64 ch] (rows of the unused pixels included)
constexpr int LDS_BYTES = PATCH_BYTES + CT_BYTES;         // 64.5 KB: two workgroups per CU
static_assert(2 * LDS_BYTES <= 160 * 1024, "LDS budget");

struct StemArgs {
  const bf16_t* x; const bf16_t* w; const float* shift; bf16_t* y;
  int N, H, W, OH, OW, PH, PW, tiles_x, tiles_y, ntiles;
  unsigned x_bytes;
};

// patch layout: row pitch IP_W slots of 16 B; input column ix sits in slot (ix >> 1) + (ix & 1) * EVEN_SLOTS -- the 16 region pixels of an
// MFMA column block read columns 2 rx + kx, i.e. one parity per tap: consecutive slots, no LDS bank conflict (plain order: 32-B stride, 2-way)
constexpr int EVEN_SLOTS = (IP_W + 1) / 2;                // 36
constexpr int NMI = 5;                                    // pixel blocks per wave and pass (2 halves x 2 passes x 5 >= 19)

// Two 4-wave workgroups per CU, each walking its own tiles through {patch landed; MFMAs + epilogue, two passes of five pixel blocks; barrier;
// next patch issued; pooling + stores}: a workgroup's phases are serial (one patch buffer, one region image), the CU overlaps the MFMA phase
// of one workgroup with the VALU / LDS phase (epilogue, pooling: ~500 VALU instructions per wave and tile) and the patch latency of the
// other. One 8-wave workgroup per CU with a double-buffered patch had the same two waves per SIMD but all of them in the same phase: 66-72 us
// (tools/stem_dbg.sh: 38 us with the MFMA loop removed, 11 us with the epilogue + pooling removed too).
__global__ void __launch_bounds__(256, 2) stem_pool_kernel(StemArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch = smem;
  char* ct = smem + PATCH_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fq = lane >> 4;
  const int npair = wid & 1, mhalf = wid >> 1;            // this wave: channel blocks 2 npair, 2 npair + 1; pixel blocks mhalf * 10 + pass * 5 + mi

  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;

  auto stage = [&](int tile) {          // the tile's input patch: pieces wid, wid + 4, ...
    int tx = tile % p.tiles_x, t2 = tile / p.tiles_x, ty = t2 % p.tiles_y, n = t2 / p.tiles_y;
    int iy0 = 2 * (2 * ty * PT_H - 1) - 3, ix0 = 2 * (2 * tx * PT_W - 1) - 3;
    for (int pc = wid; pc < PATCH_PIECES; pc += 4) {
      int c = pc * 64 + lane;
      int iy = c / IP_W, slot = c - iy * IP_W;
      int ix = slot < EVEN_SLOTS ? 2 * slot : 2 * (slot - EVEN_SLOTS) + 1;
      int y = iy0 + iy, x = ix0 + ix;
      bool ok = c < PATCH_CHUNKS && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      unsigned off = ((unsigned)(n * p.H + y) * (unsigned)p.W + (unsigned)x) * 16u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_s*)(patch + pc * 1024), 16, ok ? off : OOB, 0, 0, 0);
    }
  };

  int tile = blockIdx.x;
  if (tile < p.ntiles) stage(tile);
  // this lane's weight fragments, for the life of the workgroup: channel (2 npair + a) * 16 + fi, tap 4 s + fq (padding taps: zeros)
  bf16x8 fw[2][KSTEPS];
  int toff[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    int t = 4 * s + fq;
    bool real = t < KTAPS;
    int tt = real ? t : 0;
    int ky = (tt * 37) >> 8, kx = tt - 7 * ky;          // t / 7 for t < 64
    toff[s] = (ky * IP_W + (kx >> 1) + (kx & 1) * EVEN_SLOTS) * 16;      // a padding tap reads tap 0's pixel and multiplies it by zero
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      i32x4 v = {0, 0, 0, 0};
      if (real) v = *reinterpret_cast<const i32x4*>(reinterpret_cast<const char*>(p.w) + (size_t)((2 * npair + a) * 16 + fi) * W_ROW + t * 16);
      fw[a][s] = __builtin_bit_cast(bf16x8, v);
    }
  }
  float sh[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[a][j] = p.shift[(2 * npair + a) * 16 + fq * 4 + j];

  for (; tile < p.ntiles; tile += gridDim.x) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();          // the patch has landed; everybody is done pooling the previous tile out of the region image
    int tx = tile % p.tiles_x, t2 = tile / p.tiles_x, ty = t2 % p.tiles_y, n = t2 / p.tiles_y;
    int cy0 = 2 * ty * PT_H - 1, cx0 = 2 * tx * PT_W - 1;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      f32x4 acc[NMI][2];
      int ppix[NMI];
#pragma unroll
      for (int mi = 0; mi < NMI; ++mi) {
        acc[mi][0] = acc[mi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        ppix[mi] = (mhalf * 10 + pass * 5 + mi) * 16 + fi;
      }
#pragma unroll
      for (int mi = 0; mi < NMI; ++mi) {
        if (!(UNIT_STEM_DBG & 2) && mhalf * 10 + pass * 5 + mi < NMT) {
          int pc = ppix[mi] < NPX ? ppix[mi] : NPX - 1;
          int ry = pc / CR_W, rx = pc - ry * CR_W;
          const char* src = patch + (2 * ry * IP_W + rx) * 16;
          // the pixel fragments of 7 (then 6) k-steps first, then their MFMAs (left to itself hipcc issues read, wait, two MFMAs, read, wait
          // ...: one LDS round trip per k-step); all 13 at once would not fit the 256 registers of two waves per SIMD next to the weights
          constexpr int H0 = 7;
          bf16x8 fx[H0];
#pragma unroll
          for (int s = 0; s < H0; ++s) fx[s] = *reinterpret_cast<const bf16x8*>(src + toff[s]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = 0; s < H0; ++s) {
            acc[mi][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0][s], fx[s], acc[mi][0], 0, 0, 0);
            acc[mi][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1][s], fx[s], acc[mi][1], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = H0; s < KSTEPS; ++s) fx[s - H0] = *reinterpret_cast<const bf16x8*>(src + toff[s]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = H0; s < KSTEPS; ++s) {
            acc[mi][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0][s], fx[s - H0], acc[mi][0], 0, 0, 0);
            acc[mi][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1][s], fx[s - H0], acc[mi][1], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (UNIT_STEM_DBG & 4) { if (acc[0][0][0] == 123.456f) p.y[tile] = (bf16_t)1.f; continue; }
      // D[row = channel (2 npair + a)*16 + fq*4 + j][col = region pixel]: +shift, ReLU, bf16 -> region image
#pragma unroll
      for (int mi = 0; mi < NMI; ++mi) {
        if (mhalf * 10 + pass * 5 + mi < NMT) {
          int px = ppix[mi];
          int ry = px / CR_W, rx = px - ry * CR_W;
          bool valid = px < NPX && (unsigned)(cy0 + ry) < (unsigned)p.OH && (unsigned)(cx0 + rx) < (unsigned)p.OW;
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (bf16_t)fmaxf(acc[mi][a][j] + sh[a][j], 0.f);
            i32x2 ob = __builtin_bit_cast(i32x2, o) & i32x2{0x7fff7fff, 0x7fff7fff};      // (a -0 would win the unsigned max below)
            if (!valid) ob = i32x2{0, 0};
            *reinterpret_cast<i32x2*>(ct + px * CT_PITCH + ((2 * npair + a) * 16 + fq * 4) * 2) = ob;
          }
        }
      }
    }
    __syncthreads();          // region image complete; nobody reads the patch any more
    if (!(UNIT_STEM_DBG & 1) && tile + (int)gridDim.x < p.ntiles) stage(tile + gridDim.x);
    if (UNIT_STEM_DBG & 4) continue;
    // 3x3 / stride 2 max: item = (pooled pixel of the tile, 8 channels), two items per thread, one 16-B store each. The values are >= 0:
    // their bf16 patterns order like unsigned integers, so the max is taken on the packed 16-bit halves without unpacking.
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      int item = tid + 256 * k;
      int pp = item >> 3, c8 = item & 7;
      int ppy = pp / PT_W, ppx = pp - ppy * PT_W;
      int py = ty * PT_H + ppy, pxo = tx * PT_W + ppx;
      if (py < p.PH && pxo < p.PW) {
        u16x8_s m = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            u16x8_s v = *reinterpret_cast<const u16x8_s*>(ct + ((2 * ppy + dy) * CR_W + 2 * ppx + dx) * CT_PITCH + c8 * 16);
            m = __builtin_elementwise_max(m, v);
          }
        *reinterpret_cast<u16x8_s*>(reinterpret_cast<char*>(p.y) + ((((size_t)n * p.PH + py) * p.PW + pxo) * 64 + c8 * 8) * 2) = m;
      }
    }
  }
}

}  // namespace

extern "C" int unit_stem_conv_pool(const void* x, const void* w, const float* shift, void* y, int dtype, int N, int H, int W, void* stream) {
  UNIT_CHECK_ARG(dtype == UNIT_BF16, "unit_stem_conv_pool: bf16 only (fp32 runs unit_conv2d + unit_maxpool3x3s2_fwd)");
  UNIT_CHECK_ARG(N > 0 && H > 0 && W > 0 && (size_t)N * H * W * 16 < 0xFFFFFFF0ull, "unit_stem_conv_pool: bad shape");
  StemArgs a;
  a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.shift = shift; a.y = (bf16_t*)y;
  a.N = N; a.H = H; a.W = W;
  a.OH = (H - 1) / 2 + 1; a.OW = (W - 1) / 2 + 1;
  a.PH = (a.OH - 1) / 2 + 1; a.PW = (a.OW - 1) / 2 + 1;
  a.tiles_x = cdiv(a.PW, PT_W); a.tiles_y = cdiv(a.PH, PT_H);
  a.ntiles = a.tiles_x * a.tiles_y * N;
  a.x_bytes = (unsigned)((size_t)N * H * W * 16);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)stem_pool_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  int grid = a.ntiles < 512 ? a.ntiles : 512;
  stem_pool_kernel<<<grid, 256, LDS_BYTES, (hipStream_t)stream>>>(a);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
