// conv_igemm_lc.hip -- implicit-GEMM convolution (forward / dgrad) for the backbone layers as PERSISTENT workgroups of four LOADER
// waves and four CONSUMER waves on an LDS ring (MI355X_MICROARCH.md "ring-gemm"), one workgroup per CU.
//
// Why: the 4-wave kernel of conv_igemm128.hip (every wave issues its own LDS-DMA pieces, then its fragment reads, then its MFMAs,
// one barrier per k-step) spends ~1450 cycles on a k-step whose MFMAs take 256: an LDS-DMA piece (`buffer_load_dwordx4 ... lds`,
// 1 KB) occupies the issuing wave for 60-185 cycles, six of them per wave and k-step, serial with the wave's own reads and MFMAs
// (DESIGN.md section 8). At M = 9 576 pixels (res4 on four 600x1000 images) a layer is 150-600 such tiles of 4-36 k-steps: the
// launch is its pipeline fill, its serial k-steps and its drain.
// Here the roles are split by wave: waves 4-7 (one per SIMD) do nothing but issue LDS-DMA pieces into an NS-slot ring, D = NS-1
// k-steps ahead, under a counted vmcnt; waves 0-3 (one per SIMD) do nothing but read fragments and multiply -- a SIMD's matrix pipe
// and its DMA issue port are fed by different waves at the same time. The workgroup is persistent: it walks a contiguous run of
// output tiles, and because the loaders' cursor runs ahead across tile boundaries, the first k-steps of the next tile land while
// the consumers write the previous tile's outputs (no per-tile pipeline fill or drain).
//
// Tile = BM pixels x BN channels x 64 k, BM = FB*16 in {64 .. 128}, BN = FA*64 in {128, 256}; consumer c owns all BM pixels x the
// FA*16 channels c*FA*16 .. of the tile (acc[FA][FB], v_mfma_f32_16x16x32_bf16). The tile shape is chosen per layer so that the
// tile count is a little under a multiple of the 256 CUs (M = 9 576, 256 channels: 80 x 128 -> 240 tiles).
// LDS: NS slots of [X: XR rows][W: BN rows] x 128 B (XR = BM rounded up to 32: every loader issues the same number of pieces per
// k-step -- the counted vmcnt needs a compile-time count -- and the rows past BM are zero pieces nobody reads), then the consumers'
// wave-private epilogue scratch. Same LDS image / source-side XOR swizzle / k order (channel block outermost, taps innermost) /
// MFMA order / epilogue as conv_igemm128.hip: results are bit-identical to it.
//
// Synchronisation: ONE s_barrier per k-step, executed by all eight waves (G = tiles * k-steps of them in all). Global k-step g
// lives in slot g % NS.
//   loader   : [prologue: issue steps 0 .. D-1, wait until step 0 has landed]
//              loop g: barrier(g) ; issue step g+D into slot (g+D) % NS = (g-1) % NS ; s_waitcnt vmcnt(P*(D-1)) -> step g+1 landed
//   consumer : loop g: barrier(g) ; read the fragments of slot g % NS ; MFMAs ; [last k-step of a tile: epilogue]
//   RAW: barrier(g) follows the loaders' wait for step g.   WAR: slot (g-1) % NS was read in iteration g-1, its reads retired before
//   that iteration's MFMAs, i.e. before the consumers reached barrier(g); the loaders overwrite it after barrier(g).
#include "conv_igemm128.h"
#include "conv_epilogue.h"

typedef __attribute__((address_space(3))) void lds_void_lc;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 40, "vmcnt immediate");
#define LC_W(n) else if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  LC_W(1); LC_W(2); LC_W(3); LC_W(4); LC_W(5); LC_W(6); LC_W(7); LC_W(8); LC_W(9); LC_W(10); LC_W(11); LC_W(12); LC_W(13); LC_W(14);
  LC_W(15); LC_W(16); LC_W(17); LC_W(18); LC_W(19); LC_W(20); LC_W(21); LC_W(22); LC_W(23); LC_W(24); LC_W(25); LC_W(26); LC_W(27);
  LC_W(28); LC_W(29); LC_W(30); LC_W(31); LC_W(32); LC_W(33); LC_W(34); LC_W(35); LC_W(36); LC_W(37); LC_W(38); LC_W(39); LC_W(40);
#undef LC_W
}

// wave-uniform runtime count (the operand-reuse form below issues a different number of pieces per step)
__device__ __forceinline__ void wait_vmcnt_rt(int n) {
#define LC_R(v) case v: asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory"); break
  switch (n) {
    LC_R(0); LC_R(1); LC_R(2); LC_R(3); LC_R(4); LC_R(5); LC_R(6); LC_R(7); LC_R(8); LC_R(9); LC_R(10); LC_R(11); LC_R(12); LC_R(13); LC_R(14);
    LC_R(15); LC_R(16); LC_R(17); LC_R(18); LC_R(19); LC_R(20); LC_R(21); LC_R(22); LC_R(23); LC_R(24); LC_R(25); LC_R(26); LC_R(27); LC_R(28);
    LC_R(29); LC_R(30); LC_R(31); LC_R(32); LC_R(33); LC_R(34); LC_R(35); LC_R(36);
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef LC_R
}

// WD ("weights direct", round 6): the weight fragments do not pass through LDS at all. A consumer wave owns FA*16 channels of the tile
// exclusively -- nobody else reads its weight rows -- so staging them in LDS buys no sharing and costs the CU's LDS port twice (the
// LDS-DMA write and the fragment read): of the ~83 KB that cross the port per k-step of an 80 x 128 tile (26.6 KB staged + 4 x 10 KB of
// pixel fragments + 16 KB of weight fragments) 32 KB are weights, and at 128 B/clk the port, not the MFMA pipe (282 clk) or the L2, is what
// the ~900 clk k-step waits for. With WD every consumer lane fetches its 16-byte fragment piece straight from L2 into registers
// (buffer_load_dwordx4: row = its channel, the same k-slice the LDS image would have held -- same MFMA operands, bit-identical results),
// PD k-steps ahead; the loaders stage pixels only.
template <int FB, int FA, int NL, int NSMAX = 3, bool WD = false> struct LcCfg {
  static constexpr int BM = FB * 16, BN = FA * 64;
  static constexpr int XP = BM / 8, WP = WD ? 0 : BN / 8, TP = XP + WP;     // LDS-DMA pieces (8 rows x 128 B) per k-step: X, W, all
  static constexpr int XPL = (XP + NL - 1) / NL, WPL = (WP + NL - 1) / NL;      // at most so many per loader wave
  static constexpr int XR = BM;                                     // X rows of a slot
  static constexpr int SLOT = (XR + (WD ? 0 : BN)) * 128;
  static constexpr int THREADS = (4 + NL) * 64;
  static constexpr int SCR = 4 * EpiCfg<FA>::BYTES;
  static constexpr int NS = ((160 * 1024 - SCR) / SLOT >= NSMAX) ? NSMAX : (160 * 1024 - SCR) / SLOT;
  static constexpr int LDS = NS * SLOT + SCR;
  static_assert((160 * 1024 - SCR) / SLOT >= (NSMAX < 3 ? NSMAX : 3), "the ring slots must fit the 160 KB of LDS");
  // NSMAX == 2: two slots, at most 80 KB -- TWO workgroups per CU (code + 2000): the epilogue of one beside the k-steps of the other
  static_assert(NSMAX != 2 || LDS <= 80 * 1024, "two-per-CU form: 80 KB of LDS per workgroup");
};

// X3: bf16x3 operands (conv_epilogue.h SplitK) -- the loaders' cursor carries the segment of the virtual channel block, the consumers
// write split output planes; everything else is the same kernel.
// PAIR: two problems of one layer in one grid (conv_epilogue.h ConvSecond): tile ids >= second.tiles0 belong to the second problem; the loaders'
// cursor and the consumers swap the problem's fields in per tile (a workgroup's run of tiles may cross from one problem into the other).
// R3 (X3 pointwise layers, round 5): operand reuse inside a 64-channel block. Its three k-steps are lo.Wh, hi.Wh, hi.Wl -- four distinct operand
// tiles, not six -- and this kernel is bound by what a CU takes in (header), so the loaders stage each tile ONCE: the ring is split into three X
// slots and three W slots, block b keeps lo(b) in X slot 2b % 3, hi(b) in (2b + 1) % 3, Wh(b) in W slot 2b % 3, Wl(b) in (2b + 1) % 3, and during
// the steps of block b the loaders issue [lo, Wh](b + 1), then hi(b + 1), then Wl(b + 1) -- each into the slot whose previous tile was last read
// one step earlier, three steps before its own first use. Same products in the same order: bit-identical to the plain X3 form.
//   RAW: after issuing in iteration g a loader lets only its pieces of iterations g and g - 1 stay in flight; what step g + 1 reads was issued
//        in iteration g - 2 or earlier.   WAR: see the slot table above; a slot's last reader retired its reads before the barrier that the
//        loaders pass before overwriting it.
template <int FB, int FA, int NL, int NSMAX = 3, bool X3 = false, bool PAIR = false, bool R3 = false, bool WD = false>
__global__ void __launch_bounds__((4 + NL) * 64) conv_igemm_lc_kernel(ConvDmaArgs p) {
  static_assert(!R3 || (X3 && !PAIR && NSMAX == 3), "operand reuse: bf16x3 operands, one problem, the three-slot ring");
  static_assert(!WD || !R3, "weights direct: not with the operand-reuse rings");
  typedef LcCfg<FB, FA, NL, NSMAX, WD> Cf;
  constexpr int BM = Cf::BM, BN = Cf::BN, BK = 64;
  constexpr int XP = Cf::XP, WP = Cf::WP, TP = Cf::TP, XPL = Cf::XPL, WPL = Cf::WPL, XR = Cf::XR, SLOT = Cf::SLOT, NS = Cf::NS;
  constexpr int D = NS - 1, PLO = TP / NL, NHI = TP % NL;           // loaders 0 .. NHI-1 issue PLO + 1 pieces per k-step, the others PLO
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // this workgroup's run of tiles: the XCD-aware id map of the other conv kernels (workgroup b runs on XCD b % 8; an XCD's
  // workgroups get neighbouring runs: the channel tiles of a pixel block share its rows through that XCD's L2), then an even split
  const int total = (p.tiles_m + (PAIR ? p.second.tiles_m : 0)) * p.tiles_n, nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    int q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tq = total / nwg, tr = total % nwg;
  const int t_first = bid * tq + min(bid, tr), t_count = tq + (bid < tr ? 1 : 0);
  const int nk = p.Kgemm / BK;
  const int G = t_count * nk;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

  if (wid >= 4) {
    // ------------------------------------------------------------------------------------------------ loader waves
    const int l = wid - 4;
    const bf16_t* __restrict__ X = (const bf16_t*)p.x;
    const bf16_t* __restrict__ Wt = (const bf16_t*)p.w;
    __amdgpu_buffer_rsrc_t rsX0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(X), 0, (int)p.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsX1 = rsX0;
    if constexpr (PAIR) rsX1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>((const bf16_t*)p.second.x), 0, (int)p.second.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Wt), 0, (int)p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const int lrow = lane >> 3, lc = lane & 7;
    const int RS = p.R * p.S;
    int cH = p.H, cW = p.W;                         // map size of the problem the cursor's tile belongs to
    bool cur_second = false;
    // The TP pieces of a k-step (X pieces 0 .. XP-1, then the W pieces) are dealt round-robin to the NL loaders: X piece i*NL + l, W piece
    // i*NL + lw with XP + lw = l (mod NL); piece = 8 rows: lane -> row + lrow, LDS chunk lc (lane-linear image), SOURCE chunk
    // lc ^ ((row >> 1) & 7)
    const int lw = (l + NL - XP % NL) % NL;
    // (arrays of fixed size: hipcc's host pass silently drops a kernel instantiation whose lambdas capture arrays sized by a template-
    // dependent constant -- the launch stub then is an undefined symbol, conv_igemm128.hip has the same note)
    static_assert(XPL <= 4 && WPL <= 8, "staging tables");
    int x_ih0[4], x_iw0[4]; unsigned x_base[4]; bool x_ok[4]; int x_q[4];
    unsigned w_off[8]; bool w_ok[8];
    int cur_t = 0, cur_kt = 0, cur_cb = 0, cur_rs = 0;
    int cur_sg = 0, cur_cbr = 0;                    // X3: cur_cb = cur_cbr * nseg + cur_sg
    const int xpitch = X3 ? p.sk.x_pitch : p.C;
    auto setup_tile = [&](int t) {
      int id = t_first + t;
      int M = p.M, OW = p.OW, OH = p.OH;
      cH = p.H; cW = p.W; cur_second = false;
      if constexpr (PAIR) {
        if (id >= p.second.tiles0) { id -= p.second.tiles0; M = p.second.M; OW = p.second.OW; OH = p.second.OH; cH = p.second.H; cW = p.second.W; cur_second = true; }
      }
      int tile_n = id % p.tiles_n, tile_m = id / p.tiles_n;
      int m0 = tile_m * BM, n0 = tile_n * BN;
#pragma unroll
      for (int i = 0; i < XPL; ++i) {
        int row = (i * NL + l) * 8 + lrow;
        x_q[i] = lc ^ ((row >> 1) & 7);
        int m = m0 + row;
        x_ok[i] = row < BM && m < M;
        int mm = x_ok[i] ? m : 0;
        int ow = mm % OW; int tt = mm / OW; int oh = tt % OH; int n = tt / OH;
        x_ih0[i] = oh * p.stride - p.pad; x_iw0[i] = ow * p.stride - p.pad;
        x_base[i] = (unsigned)n * (unsigned)(cH * cW * xpitch);
      }
#pragma unroll
      for (int i = 0; i < WPL; ++i) {
        int row = (i * NL + lw) * 8 + lrow;
        int q = lc ^ ((row >> 1) & 7);
        int nn = n0 + row;
        w_ok[i] = row < BN && nn < p.K;
        w_off[i] = ((unsigned)(w_ok[i] ? nn : 0) * (unsigned)p.Kgemm + (unsigned)q * 8u) * 2u;
      }
    };
    // issue the k-step at the cursor into `slot`, advance the cursor (k order: channel block outermost, the R*S taps innermost)
    auto issue = [&](int slot) {
      int ch0 = cur_cb * BK, k0 = cur_rs * p.C + ch0, r = cur_rs / p.S, s = cur_rs - r * p.S;
      if constexpr (X3) ch0 = ((p.sk.seg_lo >> cur_sg) & 1) * p.sk.cr + cur_cbr * BK;      // (k0 keeps the virtual block: the weights' layout)
      char* base = smem + slot * SLOT;
#pragma unroll
      for (int i = 0; i < XPL; ++i) {
        if (i * NL + l >= XP) continue;             // (wave-uniform)
        int R0 = (i * NL + l) * 8;
        int ih = x_ih0[i] + r, iw = x_iw0[i] + s;
        bool ok = x_ok[i] && (unsigned)ih < (unsigned)cH && (unsigned)iw < (unsigned)cW;
        unsigned off = (x_base[i] + (unsigned)((ih * cW + iw) * xpitch + ch0 + x_q[i] * 8)) * 2u;
        if (PAIR && cur_second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX1, (lds_void_lc*)(base + R0 * 128), 16, ok ? off : OOB, 0, 0, 0);      // (scalar branch)
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX0, (lds_void_lc*)(base + R0 * 128), 16, ok ? off : OOB, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < WPL; ++i) {
        if (i * NL + lw >= WP) continue;
        int R0 = (i * NL + lw) * 8;
        unsigned off = w_off[i] + (unsigned)k0 * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_lc*)(base + XR * 128 + R0 * 128), 16, w_ok[i] ? off : OOB, 0, 0, 0);
      }
      if (++cur_rs == RS) {
        cur_rs = 0; ++cur_cb;
        if constexpr (X3) { if (++cur_sg == p.sk.nseg) { cur_sg = 0; ++cur_cbr; } }
      }
      if (++cur_kt == nk) {
        cur_kt = 0; cur_cb = 0; cur_rs = 0; cur_sg = 0; cur_cbr = 0;
        if (++cur_t < t_count) setup_tile(cur_t);
      }
    };
    if constexpr (R3) {
      // blocks = (tile, real 64-channel block); NB of them in this workgroup's run; step g = 3 * b + sg
      const int nb = nk / 3, NB = t_count * nb;
      int cx = 0, cw = 0;                            // this loader's pieces per X tile / W tile
#pragma unroll
      for (int i = 0; i < XPL; ++i) cx += (i * NL + l < XP) ? 1 : 0;
#pragma unroll
      for (int i = 0; i < WPL; ++i) cw += (i * NL + lw < WP) ? 1 : 0;
      char* xring = smem;
      char* wring = smem + 3 * XR * 128;
      auto issue_x = [&](int plane_bit, int cbr, int xs) {
        const int ch0 = plane_bit * p.sk.cr + cbr * BK;
        char* base = xring + xs * (XR * 128);
#pragma unroll
        for (int i = 0; i < XPL; ++i) {
          if (i * NL + l >= XP) continue;
          int R0 = (i * NL + l) * 8;
          bool ok = x_ok[i] && (unsigned)x_ih0[i] < (unsigned)cH && (unsigned)x_iw0[i] < (unsigned)cW;
          unsigned off = (x_base[i] + (unsigned)((x_ih0[i] * cW + x_iw0[i]) * xpitch + ch0 + x_q[i] * 8)) * 2u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX0, (lds_void_lc*)(base + R0 * 128), 16, ok ? off : OOB, 0, 0, 0);
        }
      };
      auto issue_w = [&](int seg, int cbr, int ws) {
        const int k0 = (cbr * 3 + seg) * BK;          // the weights' layout keeps all three segments ([Wh | Wh | Wl], split.hip)
        char* base = wring + ws * (BN * 128);
#pragma unroll
        for (int i = 0; i < WPL; ++i) {
          if (i * NL + lw >= WP) continue;
          int R0 = (i * NL + lw) * 8;
          unsigned off = w_off[i] + (unsigned)k0 * 2u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_lc*)(base + R0 * 128), 16, w_ok[i] ? off : OOB, 0, 0, 0);
        }
      };
      const int lo_bit = p.sk.seg_lo & 1, hi_bit = (p.sk.seg_lo >> 1) & 1;      // plane of segment 0 (lo) / segments 1, 2 (hi)
      int set_t = 0;                                 // tile whose descriptors are loaded
      setup_tile(0);
      auto at_block = [&](int b) {                   // descriptors of block b's tile
        int t = b / nb;
        if (t != set_t) { setup_tile(t); set_t = t; }
        return b - t * nb;                           // its real channel block
      };
      if (NB > 0) {                                  // block 0 whole
        int c0 = at_block(0);
        issue_x(lo_bit, c0, 0); issue_w(0, c0, 0); issue_x(hi_bit, c0, 1); issue_w(2, c0, 1);
      }
      wait_vmcnt<0>();
      for (int g = 0; g < G; ++g) {
        __builtin_amdgcn_s_barrier();
        const int b = g / 3, sg = g - b * 3, nx = b + 1;
        if (nx < NB) {
          int c1 = at_block(nx);
          if (sg == 0) { issue_x(lo_bit, c1, (2 * nx) % 3); issue_w(0, c1, (2 * nx) % 3); wait_vmcnt_rt(cx + 2 * cw); }
          else if (sg == 1) { issue_x(hi_bit, c1, (2 * nx + 1) % 3); wait_vmcnt_rt(2 * cx + cw); }
          else { issue_w(2, c1, (2 * nx + 1) % 3); wait_vmcnt_rt(cx + cw); }
        } else {
          wait_vmcnt<0>();
        }
      }
      return;
    }
    setup_tile(0);
    int issued = 0, islot = 0;
    const bool hi = l < NHI;                        // this loader issues PLO + 1 pieces per k-step
    for (; issued < D && issued < G; ++issued) { issue(islot); islot = islot + 1 == NS ? 0 : islot + 1; }
    auto wait_ahead = [&]() {                       // all but the youngest D-1 k-steps of this wave's pieces have landed
      if (hi) wait_vmcnt<(PLO + 1) * (D - 1)>(); else wait_vmcnt<PLO * (D - 1)>();
    };
    if (G >= D) wait_ahead(); else wait_vmcnt<0>();
    for (int g = 0; g < G; ++g) {
      __builtin_amdgcn_s_barrier();
      if (issued < G) {
        issue(islot); islot = islot + 1 == NS ? 0 : islot + 1; ++issued;
        wait_ahead();
      } else {
        wait_vmcnt<0>();
      }
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- consumer waves
  const int c = wid;
  const int frow = lane & 15, fq = lane >> 4;
  // fragment offsets inside a slot (k-substep 0; substep 1 = chunk + 4 = byte offset ^ 64)
  static_assert(FB <= 8 && FA <= 4, "fragment tables");
  int offx[8], offw[4];
#pragma unroll
  for (int b = 0; b < FB; ++b) { int row = b * 16 + frow; offx[b] = row * 128 + ((fq ^ ((row >> 1) & 7)) << 4); }
#pragma unroll
  for (int a = 0; a < FA; ++a) { int row = c * FA * 16 + a * 16 + frow; offw[a] = XR * 128 + row * 128 + ((fq ^ ((row >> 1) & 7)) << 4); }
  char* scr = smem + NS * SLOT + c * EpiCfg<FA>::BYTES;
  int slot = 0;
  // WD: this lane's piece of every weight fragment comes straight from memory -- row = channel n0 + c*FA*16 + a*16 + frow, bytes
  // (k0 + ks*32 + fq*8) * 2 .. +16 of it: exactly the 16 bytes the LDS image holds at offw[a] ^ (ks * 64)
  constexpr int PD = WD ? (FA <= 2 ? 2 : 1) : 0, SETS = PD + 1;      // k-steps of prefetch; register sets of fragments
  __amdgpu_buffer_rsrc_t rsWc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>((const bf16_t*)p.w), 0, (int)p.w_bytes, 0x00020000);
  const int RSc = p.R * p.S;
  for (int t = 0; t < t_count; ++t) {
    int id = t_first + t;
    bool second = false;
    if constexpr (PAIR) { if (id >= p.second.tiles0) { id -= p.second.tiles0; second = true; } }
    int tile_n = id % p.tiles_n, tile_m = id / p.tiles_n;
    const int blk0 = R3 ? t * (nk / 3) : 0;          // R3: the run's block number of this tile's first block
    f32x4 acc[FA][FB];
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
      for (int b = 0; b < FB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (WD) {
      constexpr unsigned OOBW = 0xFFFFFFF0u;
      unsigned wrow[FA]; bool wok[FA];
#pragma unroll
      for (int a = 0; a < FA; ++a) {
        int nn = tile_n * BN + c * FA * 16 + a * 16 + frow;
        wok[a] = nn < p.K;
        wrow[a] = ((unsigned)(wok[a] ? nn : 0) * (unsigned)p.Kgemm + (unsigned)fq * 8u) * 2u;
      }
      i32x4 fw[SETS][2][FA];
      int w_rs = 0, w_cb = 0;                        // (channel block, tap) of the next k-step to fetch: k order = block outermost, taps innermost
      auto fetch = [&](i32x4 (&dst)[2][FA]) {
        const unsigned koff = (unsigned)(w_rs * p.C + w_cb * BK) * 2u;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int a = 0; a < FA; ++a)
            dst[ks][a] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsWc, (int)(wok[a] ? wrow[a] + koff + (unsigned)ks * 64u : OOBW), 0, 0));
        if (++w_rs == RSc) { w_rs = 0; ++w_cb; }
      };
      // one k-step: wait for the slot's pixels (the barrier all eight waves meet at), read the pixel fragments, multiply with the weight
      // fragments of register set `fwu`
      auto kstep = [&](const i32x4 (&fwu)[2][FA]) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* base = smem + slot * SLOT;
        i32x4 fb[2][FB];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int b = 0; b < FB; ++b) fb[ks][b] = *reinterpret_cast<const i32x4*>(base + (offx[b] ^ (ks * 64)));
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int a = 0; a < FA; ++a)
#pragma unroll
            for (int b = 0; b < FB; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fwu[ks][a]), __builtin_bit_cast(bf16x8, fb[ks][b]), acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        slot = slot + 1 == NS ? 0 : slot + 1;
      };
      // (nk > PD: unit_conv_lc_launch sends shorter contractions to the staged form.) Prologue and steady state are free of conditional
      // fetches: hipcc's vmcnt bookkeeping merges the pending loads of both branches at every join and falls back to vmcnt(0) -- no
      // prefetch at all -- wherever a fetch is conditional; this way the count at the loop header is the same from both of its edges.
#pragma unroll
      for (int u = 0; u < PD; ++u) fetch(fw[u]);
      int kt = 0;
      for (; kt + SETS - 1 + PD < nk; kt += SETS) {
#pragma unroll
        for (int u = 0; u < SETS; ++u) {
          fetch(fw[(u + PD) % SETS]);
          kstep(fw[u]);
        }
      }
      const int rem = nk - kt;                       // PD .. SETS + PD - 1 steps left; set of step kt + j = j % SETS
#pragma unroll
      for (int j = 0; j < SETS + PD - 1; ++j) {
        if (j < rem) {
          if (j + PD < rem) fetch(fw[(j + PD) % SETS]);
          kstep(fw[j % SETS]);
        }
      }
    } else
    for (int kt = 0; kt < nk; ++kt) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const char* base = smem + slot * SLOT;
      const char* basew = base;
      if constexpr (R3) {          // X and W come from their own rings (slot table in the header); offw carries XR * 128, the W ring starts at 3 * XR * 128
        const int sg = kt % 3, b2 = 2 * (blk0 + kt / 3);
        base = smem + ((b2 + (sg > 0 ? 1 : 0)) % 3) * (XR * 128);
        basew = smem + 2 * XR * 128 + ((b2 + (sg == 2 ? 1 : 0)) % 3) * (BN * 128);
      }
      i32x4 fa[2][FA], fb[2][FB];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int a = 0; a < FA; ++a) fa[ks][a] = *reinterpret_cast<const i32x4*>(basew + (offw[a] ^ (ks * 64)));
#pragma unroll
        for (int b = 0; b < FB; ++b) fb[ks][b] = *reinterpret_cast<const i32x4*>(base + (offx[b] ^ (ks * 64)));
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int a = 0; a < FA; ++a)
#pragma unroll
          for (int b = 0; b < FB; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[ks][a]), __builtin_bit_cast(bf16x8, fb[ks][b]), acc[a][b], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      slot = slot + 1 == NS ? 0 : slot + 1;
    }
    // this wave's BM x FA*16 block through its private scratch, row-major (conv_epilogue.h); the loaders are already D k-steps
    // into the next tile
    if (PAIR && second) {          // the second problem's output / residual / mask tensors and sizes
      ConvDmaArgs q = p;
      pair_swap_common(q);
      epilogue_rows_bf16_fast<FA, FB, X3>(acc, scr, tile_m * BM, tile_n * BN + c * FA * 16, q, lane);
    } else
    epilogue_rows_bf16_fast<FA, FB, X3>(acc, scr, tile_m * BM, tile_n * BN + c * FA * 16, p, lane);
  }
}

template <int FB, int FA, int NL, int NSMAX = 3, bool X3 = false, bool WD = false>
static int launch_lc(ConvDmaArgs& a, hipStream_t st) {
  typedef LcCfg<FB, FA, NL, NSMAX, WD> Cf;
  a.tiles_m = cdiv(a.M, Cf::BM); a.tiles_n = cdiv(a.K, Cf::BN);
  int total = a.tiles_m * a.tiles_n;
  if (a.second.on) {          // pair launch (only the default three-slot / four-loader forms carry a PAIR instantiation)
    if constexpr (NL == 4 && NSMAX == 3) {
      a.second.tiles_m = cdiv(a.second.M, Cf::BM);
      a.second.tiles0 = total;
      total += a.second.tiles_m * a.tiles_n;
      int grid = total < 256 ? total : 256;
      static bool attr2 = false;
      if (!attr2) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_lc_kernel<FB, FA, NL, NSMAX, X3, true, false, WD>, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS);
        attr2 = true;
      }
      conv_igemm_lc_kernel<FB, FA, NL, NSMAX, X3, true, false, WD><<<grid, Cf::THREADS, Cf::LDS, st>>>(a);
      UNIT_LAUNCH_CHECK();
      return UNIT_OK;
    } else {
      unit_set_error("conv_lc: pair launches take the tile codes 142 .. 182, 144 .. 164");
      return UNIT_ERR_UNSUPPORTED;
    }
  }
  const int slots = NSMAX == 2 ? 512 : 256;          // persistent workgroups: one per CU, two with the two-slot ring
  int grid = total < slots ? total : slots;
  if constexpr (X3 && NL == 4 && NSMAX == 3 && !WD) {
    // operand reuse (R3) for the pointwise layers: lo / hi / Wh / Wl of a block staged once each. UNIT_X3_REUSE=0: off (A/B, bit-identity test)
    const char* e = getenv("UNIT_X3_REUSE");
    if ((e ? atoi(e) : 1) && a.R == 1 && a.S == 1 && a.sk.nseg == 3 && (a.Kgemm / 64) % 3 == 0) {
      static bool attr3 = false;
      if (!attr3) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_lc_kernel<FB, FA, NL, NSMAX, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS);
        attr3 = true;
      }
      conv_igemm_lc_kernel<FB, FA, NL, NSMAX, true, false, true><<<grid, Cf::THREADS, Cf::LDS, st>>>(a);
      UNIT_LAUNCH_CHECK();
      return UNIT_OK;
    }
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_igemm_lc_kernel<FB, FA, NL, NSMAX, X3, false, false, WD>, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS);
    attr_set = true;
  }
  conv_igemm_lc_kernel<FB, FA, NL, NSMAX, X3, false, false, WD><<<grid, Cf::THREADS, Cf::LDS, st>>>(a);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

int unit_conv_lc_launch(ConvDmaArgs& a, int out_dtype, int code, hipStream_t st) {
  if (out_dtype != UNIT_BF16 || (a.ldy & 7) != 0) { unit_set_error("conv_lc: bf16 output with ldy % 8 == 0 only"); return UNIT_ERR_UNSUPPORTED; }
  // code = 100 + 10 * (BM / 16) + BN / 64  [+ 1000: eight loader waves instead of four; + 4000: four ring slots instead of three].
  // Measured on the res4 shapes (tools/lc_sweep.py, profiles/r03_exp_loader_consumer.txt): neither a deeper ring (3 / 4 / 5 / 6 slots)
  // nor eight loaders change the time -- a k-step costs what its (BM + BN) * 128 bytes cost at the CU's intake from L2 / Infinity Cache
  // (~27 B/clk) -- so the default is the smallest footprint: three slots, four loaders.
  if (a.sk.nseg > 1) {             // bf16x3 operands: the default form (three slots, four loaders) of every tile shape
    switch (code) {
      case 142: return launch_lc<4, 2, 4, 3, true>(a, st);
      case 152: return launch_lc<5, 2, 4, 3, true>(a, st);
      case 162: return launch_lc<6, 2, 4, 3, true>(a, st);
      case 172: return launch_lc<7, 2, 4, 3, true>(a, st);
      case 182: return launch_lc<8, 2, 4, 3, true>(a, st);
      case 144: return launch_lc<4, 4, 4, 3, true>(a, st);
      case 154: return launch_lc<5, 4, 4, 3, true>(a, st);
      case 164: return launch_lc<6, 4, 4, 3, true>(a, st);
    }
    unit_set_error("conv_lc: bf16x3 operands: tile code 142 .. 182, 144 .. 164");
    return UNIT_ERR_UNSUPPORTED;
  }
  if (code >= 8000 && a.Kgemm / 64 < 4) code -= 8000;          // (the prefetch pipeline of the WD form wants >= 4 k-steps)
  switch (code) {          // + 8000: weights direct (WD), the default three-slot / four-loader form of every tile shape
    case 8142: return launch_lc<4, 2, 4, 3, false, true>(a, st);
    case 8152: return launch_lc<5, 2, 4, 3, false, true>(a, st);
    case 8162: return launch_lc<6, 2, 4, 3, false, true>(a, st);
    case 8172: return launch_lc<7, 2, 4, 3, false, true>(a, st);
    case 8182: return launch_lc<8, 2, 4, 3, false, true>(a, st);
    case 8144: return launch_lc<4, 4, 4, 3, false, true>(a, st);
    case 8154: return launch_lc<5, 4, 4, 3, false, true>(a, st);
    case 8164: return launch_lc<6, 4, 4, 3, false, true>(a, st);
    case 142: return launch_lc<4, 2, 4>(a, st);
    case 152: return launch_lc<5, 2, 4>(a, st);
    case 162: return launch_lc<6, 2, 4>(a, st);
    case 172: return launch_lc<7, 2, 4>(a, st);
    case 182: return launch_lc<8, 2, 4>(a, st);
    case 144: return launch_lc<4, 4, 4>(a, st);
    case 154: return launch_lc<5, 4, 4>(a, st);
    case 164: return launch_lc<6, 4, 4>(a, st);
    case 1142: return launch_lc<4, 2, 8>(a, st);
    case 1152: return launch_lc<5, 2, 8>(a, st);
    case 1162: return launch_lc<6, 2, 8>(a, st);
    case 1172: return launch_lc<7, 2, 8>(a, st);
    case 1182: return launch_lc<8, 2, 8>(a, st);
    case 1144: return launch_lc<4, 4, 8>(a, st);
    case 1154: return launch_lc<5, 4, 8>(a, st);
    case 2142: return launch_lc<4, 2, 4, 2>(a, st);
    case 2152: return launch_lc<5, 2, 4, 2>(a, st);
    case 2162: return launch_lc<6, 2, 4, 2>(a, st);
    case 4152: return launch_lc<5, 2, 4, 4>(a, st);
    case 4142: return launch_lc<4, 2, 4, 4>(a, st);
  }
  unit_set_error("conv_lc: tile code must be 100 + 10 * (BM / 16) + BN / 64 with BM 64..128 x BN 128, or BM 64..96 x BN 256");
  return UNIT_ERR_UNSUPPORTED;
}
