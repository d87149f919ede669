/* unit_hip.h -- C ABI of libunit_hip.so: the MI355X (gfx950) operator library behind the Faster-R-CNN-C4 hot path of
 * ubc-vision/UniT.  This is the drop-in boundary: every entry point is `extern "C"`, takes raw
 * device pointers + explicit shapes/strides + a caller-owned workspace, launches asynchronously on the given
 * hipStream_t (passed as void*), never allocates, never synchronises, never throws.  Return 0 on success, <0 on
 * error (unit_last_error() gives the message).  One process per GPU.  Process-wide state, all of it listed here: the last error
 * string (thread-local); `hipFuncSetAttribute` one-time flags of the kernels that need more than 64 KB of LDS. Which kernel serves
 * a call is decided per call (explicit `variant` / `tile` arguments, 0 = the production policy): there is no kernel-selection state. No operator result depends on call history; there is no allocator, cache or handle to thread
 * through calls (which is why the `unit_ctx*` of SURVEY 8b was not needed).
 *
 * The reference reaches these operators through PyTorch's dispatcher into ATen/cuDNN, Detectron2 `_C` and torchvision;
 * each declaration cites the UniT call site (path under /root/reference) whose operator it replaces.
 * dtype codes: 0 = fp32, 1 = bf16.  All activation tensors are NHWC; "ld" arguments are row strides in elements.
 * Data-dependent sizes (proposal counts, RoI counts, ...) live in device int32 arrays so the step never syncs.
 */
#ifndef UNIT_HIP_H
#define UNIT_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int unit_version(void);
/* sha256 of the sources this library was built from (unit_amd/build.py:source_hash; csrc/build_stamp.hip): the loader compares it with
 * the sources lying next to the library, bench.py prints it, profiles/pmc_traffic.json records it */
const char* unit_build_hash(void);
const char* unit_last_error(void);

/* ---- a1 preprocess_image: modeling/meta_arch/rcnn.py:257-266 (+ ImageList.from_tensors zero padding) ---- */
int unit_preprocess_image(const float* img_chw, int C, int H, int W, const float* mean3, const float* std3, float prescale,
                          void* out_nhwc, int out_dtype, int Hmax, int Wmax, int Cpad, void* stream);
/* ---- input pipeline on the device: data/dataset_mapper.py:13-31, data/build.py:476-497 -> Detectron2
 * DatasetMapper = ResizeShortestEdge (Pillow BILINEAR on uint8 HWC) + RandomFlip, image.astype(float32) CHW.
 * unit_resize_u8_pass: one pass of Pillow's 8-bit resampler (src/libImaging/Resample.c) along axis 1 (x) or 0 (y); bounds
 * [out][2] = (first source index, taps), kk [out][ksize] 22-bit fixed-point taps, both device int32, computed by the caller
 * exactly as Pillow does (unit_amd/data_pipeline.py). Horizontal pass first, then vertical, through a uint8 intermediate.
 * unit_preprocess_u8: unit_preprocess_image for a uint8 HWC source with an optional horizontal flip. */
int unit_resize_u8_pass(const unsigned char* src, int H, int W, int C, int axis, const int* bounds, const int* kk, int ksize,
                        int out_size, unsigned char* dst, void* stream);
int unit_preprocess_u8(const unsigned char* img_hwc, int C, int H, int W, int hflip, const float* mean3, const float* std3,
                       float prescale, void* out_nhwc, int out_dtype, int Hmax, int Wmax, int Cpad, void* stream);
/* NCHW fp32 <-> NHWC converters for the plugin boundary (reference tensors are NCHW fp32) */
int unit_nchw_to_nhwc(const float* x, void* y, int dtype, int N, int C, int H, int W, int Cp, void* stream);
int unit_nhwc_to_nchw(const void* x, int dtype, float* y, int N, int C, int H, int W, int Cp, void* stream);
/* diagnostic: one wave spinning for `cycles` shader clocks on `stream` (sink: any 4 writable device bytes, never written). Two of them on two
 * streams overlap iff the streams sit on different hardware queues (tools/queue_probe.py, unit_amd/modeling/rcnn.py stream set-up). */
int unit_debug_spin(long long cycles, void* sink, void* stream);
int unit_cast(const void* x, int in_dtype, void* y, int out_dtype, long n, void* stream);
/* zero nbytes at p (16-byte aligned): replaces the torch.zeros / Tensor.zero_ fills of the reference's step (loss accumulators of
 * engine/defaults.py:276-279's loss_dict, the zero-initialised gradient of a strided slice in autograd's conv backward) */
int unit_fill_zero(void* p, size_t nbytes, void* stream);
int unit_add_cast(const float* a32, const void* b, const void* mask_ref, void* y, int dtype, long n, void* stream);

/* ---- a2/a3/a9/a10 convolution as implicit GEMM on MFMA: backbone (configs/VOC/VOC-RCNN-101-C4-split1.yaml:6-10 ->
 * detectron2 build_resnet_backbone), RPN head (modeling/proposal_generator/rpn.py:24), Res5 heads
 * (modeling/roi_heads/box_head.py:65-80), Linear predictors (modeling/roi_heads/fast_rcnn.py:386-387,
 * modeling/roi_heads/weak_detector_fast_rcnn.py:150-156).  unit_conv2d_fwd is also the dgrad kernel (flipped,
 * transposed weights from unit_weight_prep; strided scatter + ReLU-mask epilogue). ---- */
int unit_conv2d_fwd(const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref,
                    int in_dtype, int out_dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int OH,
                    int OW, int ldy, int oy_mul, int OHf, int OWf, int relu, int tile_cfg, void* stream);
/* large-tile variant (256x256x64, 8 waves, LDS-DMA operand staging) for the big-M layers; bf16 inputs, C % 64 == 0.
 * variant: 0 = default (= 8); 8 = four phases per k-tile, half-tile staging under a counted vmcnt, fragment reads inside the
 * MFMA sections (csrc/conv_igemm256p8.hip); 7 = 8 without the reads-in-MFMA step; 9 = 8 on 224-row tiles; 10 = 224 or 256
 * rows, whichever needs fewer rounds x rows; 4 = two-stage loop (csrc/conv_igemm256.hip), 1 / 2 / 3 / 5 / 6 = its ping-pong,
 * 4 x 32-k, 224-row, auto-row and shared-input-super-tile (3x3 s1 p1 on 7x7 maps) forms: identical results bit for bit.
 * 8 (and 0) on a 3x3 stride-1 pad-1 conv over a small map (the Res5 heads' conv2 and its dgrad on 7x7) cut the output into tiles of
 * ONE output position x 256 images and skip the filter taps that read only zero padding there (18 % of the k-tiles); 12 = 8 without that
 * (same results bit for bit).
 * 11 = the schedule of 8 on v_mfma_f32_32x32x16_bf16 (csrc/conv_igemm256p8m.hip): same fp32 accumulation, another summation order
 * inside a k-tile (equal to the others within fp32 rounding of the accumulation, not bit for bit). */
int unit_conv2d_fwd_big(const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref,
                        int out_dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int OH, int OW, int ldy,
                        int oy_mul, int OHf, int OWf, int relu, int variant, void* stream);
/* unit_conv2d_fwd_big (variant 0 = default, 8 or 11; stride 1, plain output layout, bf16 in / out) with an extended epilogue -- any of:
 *   relu_bits    out: one bit per output element, (stored value) > 0, unit_relu_bits_bytes(M, ldy) bytes, ldy % 64 == 0. Layout = the
 *                     epilogue's own order (one 16-byte store per lane and 128-row x 64-channel wave tile): 16-byte word
 *                     [(m / 128) * (ldy / 64) + n / 64][lane], lane = (m % 8) * 8 + (n % 64) / 8, bit ((m % 128) / 8) * 8 + n % 8;
 *   mask_bits    in : same layout and geometry; the output is zeroed where the bit is clear (instead of a bf16 mask_ref tensor);
 *   pool_partial out: global average pool over `pool_rows` (>= 44) consecutive output rows fused into the conv -- the
 *                     x.mean(dim=[2,3]) of Res5BoxHead (/root/reference/modeling/roi_heads/box_head.py:80) without writing the
 *                     res5 map: fp32 [ceil(M/128)][4][ldy] partial sums (unit_conv_pool_partial_floats), folded per RoI in a fixed
 *                     order by unit_pool_finish (no atomics: bit-reproducible). y may be NULL when pool_partial is given.
 *   x2 / C2         : a second input tensor of the same N, H, W for a 1x1 conv over the channel concatenation [x | x2], w = [K][C + C2], C2 a
 *                     multiple of C: out = relu(conv3(y2) + shortcut(x)) of a Bottleneck block (detectron2 BottleneckBlock.forward,
 *                     reached from box_head.py:65-75) as ONE GEMM -- the shortcut's output is never written and never read back as
 *                     the residual; likewise conv1's dgrad + the shortcut's dgrad.
 * unit_avgpool_bwd_bits: g[m][n] = bit(roi_offset * rows + m, n) ? dfeat[m / rows][n] / rows : 0 for R RoIs starting at RoI
 * `roi_offset` of the map the bits belong to: the backward of (average pool o ReLU) from relu_bits (= unit_global_avgpool_bwd_relu
 * without reading the map). */
size_t unit_conv_pool_partial_floats(int M, int ldy);
size_t unit_relu_bits_bytes(int M, int ldy);
int unit_conv2d_fwd_big_ex(const void* x, const void* w, void* y, const float* bias, const void* residual, const unsigned char* mask_bits,
                           unsigned char* relu_bits, float* pool_partial, int pool_rows, int N, int H, int W, int C, int K, int R, int S,
                           int pad, int ldy, int relu, const void* x2, int C2, int variant, void* stream);
int unit_pool_finish(const float* partial, int R, int rows, int ldy, int K, void* out, int ldo, int out_dtype, void* stream);
int unit_avgpool_bwd_bits(const void* dfeat, const unsigned char* bits, int R, int roi_offset, int rows, int C, void* g, void* stream);
/* mid-size variants for the backbone layers; bf16 inputs, C % 64 == 0.
 * tile 0..5 (4 waves, LDS-DMA, two workgroups per CU, csrc/conv_igemm128.hip): 0 = 128x128, 1 = 64 pixels x 128 channels, 2 = 128 x 64,
 * 3 = 128x128 with in-workgroup split-K (few-tile layers), 4 / 5 = 96 x 128 with three / two LDS stages;
 * tile >= 100 (csrc/conv_igemm_lc.hip): ONE persistent workgroup per CU of four LDS-DMA loader waves and four MFMA consumer waves on a
 * three-slot LDS ring, walking a contiguous run of output tiles: 100 + 10 * (BM / 16) + BN / 64 with BM 64..128 x BN 128 or BM 64..96 x
 * BN 256 (+ 1000: eight loader waves; + 4000: four ring slots -- measured no faster; + 2000, BM 64..96 x BN 128: TWO workgroups per CU on
 * a two-slot ring, for layers with two or more tiles per CU -- one's epilogue beside the other's k-steps: 7-20 % faster in isolation
 * on such layers, no gain inside the step, not used by the policy); bf16 output with ldy % 8 == 0. Same results
 * bit for bit as tiles 0..5. For the layers with a long contraction and few output tiles (res4 on four 600x1000 images). */
int unit_conv2d_fwd_mid(const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref,
                        int out_dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int OH, int OW, int ldy,
                        int oy_mul, int OHf, int OWf, int relu, int tile, void* stream);
/* ---- bf16x3 ("split") operands: the parity-grade fast mode (csrc/split.hip) ------------------------------------------------------------
 * The reference computes its convolutions in fp32 (/root/reference/modeling/roi_heads/fast_rcnn.py:37-101, modeling/proposal_generator/
 * rpn.py:55-101 over fp32 cuDNN convs); gfx950's fp32 MFMA runs at 1/16 of the bf16 rate. A fp32 value travels here as TWO bf16 numbers,
 * hi = bf16(x), lo = bf16(x - hi) (16 significant bits), a "split tensor" [rows][2][C] bf16 (plane 0 = hi, plane 1 = lo; 4 bytes per
 * element), and a product is three bf16 MFMA products with fp32 accumulation, x . w ~ lo.Wh + hi.Wh + hi.Wl: ~2^-17 relative per product.
 *   unit_x3_split / unit_x3_merge   fp32 [rows][C] <-> split [rows][2][C] (C % 8 == 0; merge is exact)
 *   unit_weight_prep_x3             fp32 [K][R][S][C] (x scale[k], the FrozenBN fold) -> w_fwd [K][R][S][C/64][3][64] = per 64-channel block
 *                                   the k-segments [Wh | Wh | Wl] that meet the planes [lo | hi | hi] of x; w_dgrad [C][R][S][K/64][3][64] with the
 *                                   taps flipped (either may be NULL)
 *   unit_conv2d_fwd_x3              y = split(relu?(conv(x, w) + bias + residual) masked by (mask_ref > 0)); x, y, residual split tensors (y /
 *                                   residual rows of ldy channels per plane, ldy % 8 == 0), mask_ref a split tensor of mask_c channels per plane
 *                                   (its hi plane is tested); strided scatter (oy_mul, OHf, OWf) as unit_conv2d_fwd. tile: -1 = 256x256
 *                                   phase-interleaved kernel (csrc/conv_igemm256p8.hip, position-class tiles on small 3x3 maps), 0 / 1 / 2 =
 *                                   128x128 / 64x128 / 128x64 4-wave tiles (csrc/conv_igemm128.hip), 142 .. 182, 144 .. 164 = loader / consumer
 *                                   tile codes (csrc/conv_igemm_lc.hip). C % 64 == 0.
 *   unit_conv2d_wgrad_x3            dW ~ hi^T.hi + hi^T.lo + lo^T.hi: three passes of unit_conv2d_wgrad's bf16 kernels over the planes of split
 *                                   x [N,H,W][2][C] and split dy [M][2][ldy]; 3 * unit_conv2d_wgrad_splits(UNIT_BF16, ...) slabs, workspace
 *                                   3 * unit_conv2d_wgrad_workspace_bytes(UNIT_BF16, ...); dw == NULL leaves the slabs. Bits 8-9 of
 *                                   `variant` = the number of passes to run (0 = all three; 1 = hi^T.hi only: products of the bf16-rounded
 *                                   operands with fp32 accumulation -- p passes write p * splits slabs)
 *   unit_global_avgpool_x3_fwd      mean over `rows` consecutive rows of a split map [R][rows][2][C] -> fp32 [R][C] (box_head.py:80)
 *   unit_global_avgpool_x3_bwd_relu g = split((y > 0) ? dfeat / rows : 0), y the split forward map */
int unit_x3_split(const float* x, void* out, long rows, int C, void* stream);
int unit_x3_merge(const void* in, float* out, long rows, int C, void* stream);
int unit_weight_prep_x3(const float* w_krsc, const float* scale_k, int K, int R, int S, int C, void* w_fwd, void* w_dgrad, void* stream);
int unit_conv2d_fwd_x3(const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref, int mask_c,
                       int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int OH, int OW, int ldy, int oy_mul, int OHf,
                       int OWf, int relu, int tile, void* stream);
/* round 6: the dgrad chain of the bf16x3 mode with TWO k-segments per 64-channel block -- w_dgrad [C][R][S][K/64][2][64] = [Wh | Wl]
 * (unit_weight_prep_x3s dgrad_segs = 2) against the hi plane of the gradient map twice: dx = hi(dy).(Wh + Wl), the weights at 16 bits, dy at
 * its hi plane (one fresh 2^-9 rounding per element and layer, averaged out by the weight gradients' sums over the pixel rows). segs = 3 is
 * unit_conv2d_fwd_x3. The forward pass always runs three segments. */
int unit_weight_prep_x3s(const float* w_krsc, const float* scale_k, int K, int R, int S, int C, void* w_fwd, void* w_dgrad, int dgrad_segs,
                         void* stream);
int unit_conv2d_fwd_x3s(const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref, int mask_c,
                        int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int OH, int OW, int ldy, int oy_mul, int OHf,
                        int OWf, int relu, int tile, int segs, void* stream);
int unit_conv2d_wgrad_x3(const void* x, const void* dy, float* dw, const float* scale_k, int N, int H, int W, int C, int K, int R, int S,
                         int stride, int pad, int OH, int OW, int ldy, int accumulate, int variant, void* workspace, size_t workspace_bytes,
                         void* stream);
int unit_global_avgpool_x3_fwd(const void* y, float* out, int R, int rows, int C, void* stream);
int unit_global_avgpool_x3_bwd_relu(const float* dfeat, const void* y, void* g, int R, int rows, int C, void* stream);
/* Pair launches: ONE grid over two independent problems of the SAME layer (same weights, channels, taps, stride, epilogue) that differ in their
 * tensors and map sizes -- the supervised and the weak batch of a training step, each zero-padded to its OWN largest image
 * (/root/reference/modeling/meta_arch/rcnn.py:438-452 runs the backbone once per batch; data/build.py:476-486 groups each loader's images by
 * aspect ratio, so the two padded sizes almost never agree). Pointwise stride-1 layers need nothing (concatenate the rows); every other layer
 * takes the second problem here instead of a second, half-empty launch. kernel: 0 = unit_conv2d_fwd (tile = tile_cfg), 1 = unit_conv2d_fwd_mid
 * (tile 0 / 1 / 2 / 4 / 5 or a loader-consumer code 142 .. 182, 144 .. 164), 2 = unit_conv2d_fwd_big (tile = variant 0 / 8 / 12: row-major 256-row
 * tiles), 3 = unit_conv2d_fwd_x3 (mask_c as there), 4 = unit_conv2d_fwd_x3s with two segments. The second problem's OH / OW follow from its H / W; OHf / OWf = its scatter target's map
 * size (= OH / OW without scatter). Each problem's result is what the single launch writes, bit for bit. */
typedef struct UnitConvSecond {
  const void* x; void* y; const void* residual; const void* mask_ref;
  int N, H, W, OHf, OWf;
} UnitConvSecond;
int unit_conv2d_fwd_pair(int kernel, const void* x, const void* w, void* y, const float* bias, const void* residual, const void* mask_ref, int mask_c,
                         int in_dtype, int out_dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int OH, int OW, int ldy,
                         int oy_mul, int OHf, int OWf, int relu, int tile, const UnitConvSecond* second, void* stream);
size_t unit_conv2d_wgrad_workspace_bytes(int in_dtype, int N, int OH, int OW, int K, int R, int S, int C);
/* variant: 0 = production policy. Big-M bf16 layers (C % 256 == 0, K % 256 == 0, M >= 16384) use a 256x256 tile: policy = the
 * phase-interleaved schedule (csrc/conv_wgrad256p8.hip) for pointwise layers and maps of <= 1024 pixels -- on 3x3 s1 p1 convs over maps
 * of <= 512 pixels contracting only over the pixels whose filter tap lies inside the map (18 % fewer steps on 7x7; equal to the full
 * contraction within fp32 rounding, deterministic) -- and a ring of four 32-pixel stages (csrc/conv_wgrad256r.hip) otherwise;
 * 1 = two-stage loop (csrc/conv_wgrad256.hip), 2 = the ring, 3 = the phase-interleaved schedule over ALL pixels: same slabs bit for bit.
 * All other layers use a 128x128 tile: policy = LDS-DMA ring kernel for bf16 layers with C % 128 == 0 and K % 128 == 0
 * (csrc/conv_wgrad128r.hip); 4 = the register-staged kernel everywhere (csrc/conv_wgrad.hip): same slabs bit for bit. */
int unit_conv2d_wgrad(const void* x, const void* dy, float* dw, const float* scale_k, int in_dtype, int N, int H, int W, int C,
                      int K, int R, int S, int stride, int pad, int OH, int OW, int ldy, int accumulate, int variant, void* workspace,
                      size_t workspace_bytes, void* stream);
/* dw == NULL: unit_conv2d_wgrad leaves unit_conv2d_wgrad_splits() partial slabs in the workspace (no reduction); the
 * multi-tensor kernels below reduce all layers of a gradient bucket / refresh all prepared weight copies in ONE launch.
 * descs_dev: array of {const float* partial, *scale; void* wf, *wd; long offset; int splits,K,R,S,C,block0} (64 B each) */
int unit_conv2d_wgrad_splits(int in_dtype, int N, int OH, int OW, int K, int R, int S, int C);
/* The weight gradients of SEVERAL layers in one grid (csrc/conv_wgrad128r.hip): the layers of a gradient bucket -- 18 convs for six
 * res4 blocks, 9 576 pixels and 16-36 tiles each -- launched one by one need ~17 split-M slabs apiece to fill 256 CUs (9-step loops,
 * 17 slabs to reduce); launched together they have 408 tiles: one slab per layer and 150-step loops. The reference reaches these
 * through autograd's per-layer cudnn wgrad calls (engine/defaults.py:279-284 `losses.backward()`); grouping them is this plan's own.
 * Eligible (unit_conv2d_wgrad_group_supported != 0): bf16, C % 128 == 0, K % 128 == 0; returns the tile kind: 2 = 256x256 tiles
 * (csrc/conv_wgrad256p8.hip; C % 256 == 0, K % 256 == 0, >= 2048 pixels: the Res5 / RPN / res4 layers -- a Res5 head's ten layers in one
 * grid need 3-4 slabs each instead of 8-16), 1 = 128x128 ring tiles. One call launches one grid per kind (more if the list is long).
 * unit_conv2d_wgrad_group_plan fills pr[i].kind and pr[i].splits for the launch (splits_hint > 0: that many for the layer with the most pixels, the
 * others in proportion); slab s of layer i is written at pr[i].partial + s*K*R*S*C floats, the layout unit_conv2d_wgrad(dw = NULL)
 * leaves (unit_multi_wgrad_reduce folds them). Per layer, kind 1 equals unit_conv2d_wgrad's 128x128 path with the same split count bit for bit; kind 2 contracts
 * 3x3 s1 p1 layers on maps of <= 512 pixels only over in-map pixels with the same split count for every filter tap. */
typedef struct UnitWgradProblem {
  const void* x; const void* dy; void* partial;
  int N, H, W, C, K, R, S, stride, pad, OH, OW, ldy;
  int splits, kind;      /* filled by unit_conv2d_wgrad_group_plan: split-M slabs of the layer; tile kind 1 = 128x128, 2 = 256x256 */
  int x_pitch;           /* 0 = C. One PASS of a bf16x3 weight gradient (below) is a problem of its own: x / dy point at the pass's plane of the
                          * split tensors, x_pitch = 2 * C, ldy = 2 * K, x_back / dy_back = elements between the tensor's start and the pointer
                          * (0 or one plane), partial = the pass's own slabs */
  int x_back, dy_back;
} UnitWgradProblem;
size_t unit_wgrad_problem_bytes(void);
int unit_conv2d_wgrad_group_supported(int in_dtype, int N, int OH, int OW, int K, int R, int S, int C);
int unit_conv2d_wgrad_group_plan(UnitWgradProblem* pr, int n, int splits_hint);
int unit_conv2d_wgrad_group(const UnitWgradProblem* pr, int n, int in_dtype, void* stream);
/* The grid(s) unit_conv2d_wgrad_group would launch for these (planned) problems, without launching (host only; pointers are checked for
 * alignment, never read): one row of 9 ints per unit -- {launch, tile kind, XCD, first workgroup slot on that XCD, tiles, index into pr,
 * filter tap, split, first tile of the unit within its (layer [, tap], split)}. Units that contract over the same rows share an XCD's L2:
 * the tiles of one (layer, split), and for the in-map 3x3 form the filter taps that walk at the same pace (corner / edge / centre). */
int unit_conv2d_wgrad_group_layout(const UnitWgradProblem* pr, int n, int* layout, int layout_rows, int* rows);
/* stream fork / join without host-side event objects: everything enqueued on `waiter` after this call waits for everything enqueued on
 * `signaller` before it (the reference reaches this through torch.cuda.Stream.wait_stream / Event; here the step forks ~100 weight-gradient
 * launches per step to a side stream: engine/defaults.py:279-284's backward has no such structure, it is the explicit plan's own). */
int unit_stream_wait_stream(void* waiter, void* signaller);
size_t unit_tensor_desc_bytes(void);
int unit_multi_wgrad_reduce(const void* descs_dev, int n, int total_blocks, float* grads_flat, void* stream);
int unit_multi_weight_prep(const void* descs_dev, int n, int total_blocks, const float* params_flat, int dtype, void* stream);
/* local half of the "direct" data-parallel gradient exchange (unit_amd/parallel.py; replaces the sum DDP's bucket all-reduce does inside
 * NCCL, reached from engine/defaults.py:256): out[i] = parts[0][i] + ... + parts[nparts-1][i], fp32, in that order; parts = [nparts][n]
 * fp32 or bf16 (received by an all-to-all over all xGMI links at once), out fp32 [n] = this rank's shard of the gradient bucket */
int unit_shard_sum(const void* parts, int dtype, int nparts, long n, float* out, void* stream);
/* The gradient exchange from the C ABI (csrc/comm.hip): what DistributedDataParallel's reducer does on NCCL for the reference
 * (engine/defaults.py:256, scripts/train_VOC.py:67-77), for a host without torch.distributed. RCCL is resolved with dlopen at the first call (the copy
 * already in the process, else the system's): not a link-time dependency. One process per GPU.
 *   unit_comm_unique_id          rank 0: 128 bytes (UNIT_COMM_ID_BYTES) that the host hands to the other ranks out of band
 *   unit_comm_init               every rank, on its device: *comm = opaque communicator
 *   unit_allreduce_bucket_async  buf[0..count) <- sum over ranks, in place, enqueued on `stream`; dtype 0 = fp32, 1 = bf16
 *   unit_comm_wait               work enqueued on compute_stream afterwards waits for the collectives enqueued on comm_stream so far (no host wait)
 *   unit_comm_destroy, unit_comm_rccl_version (major * 10000 + minor * 100 + patch; 0 = RCCL not loadable) */
int unit_comm_unique_id(void* id, int id_bytes);
int unit_comm_init(int rank, int world, const void* id, int id_bytes, void** comm);
int unit_allreduce_bucket_async(void* comm, void* buf, long count, int dtype, void* stream);
int unit_comm_wait(void* compute_stream, void* comm_stream);
int unit_comm_destroy(void* comm);
int unit_comm_rccl_version(void);
/* FrozenBatchNorm2d fold (detectron2 layers/batch_norm.py, eps 1e-5) and weight re-layout / cast */
int unit_frozen_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float eps, float* scale, float* shift,
                        int C, void* stream);
int unit_weight_prep(const float* w_krsc, const float* scale_k, int K, int R, int S, int C, int Cp, void* w_fwd, void* w_dgrad,
                     int dtype, void* stream);
/* scratch (unit_bias_grad_scratch_bytes, may be NULL): tall bf16 inputs are summed per 512-row block into it and combined in block
 * order (deterministic, parallel over rows); without it one workgroup per 256 columns walks all rows */
size_t unit_bias_grad_scratch_bytes(int M, int K);
int unit_bias_grad(const void* dy, int dtype, int M, int K, int ld, float* db, int accumulate, float* scratch, size_t scratch_bytes,
                   void* stream);
/* weight + bias gradient of a group of Linear layers that share their input, bf16 operands, ONE launch (replaces autograd's
 * addmm backward of the box predictors: modeling/roi_heads/fast_rcnn.py:315-316 cls_score_delta / bbox_pred_delta, :477-478 the _ft
 * pair; weak_detector_fast_rcnn.py:75-95 classifier / detection streams, OICR predictors, regression branch):
 * dw[k][c] = sum_r dy[r][k] x[r][c] (fp32 [K][C], overwritten), db[k] = sum_r dy[r][k]; x [R][C], dy [R][ldy] (columns >= K zero),
 * K <= 128, C % 128 == 0, ldy % 8 == 0. Partial sums of row splits are added in split order: deterministic. */
size_t unit_linear_wgrad_workspace_bytes(int R, int C, int K);
int unit_linear_wgrad(const void* x, const void* dy, int dtype, int R, int C, int K, int ldy, float* dw, float* db,
                      void* workspace, size_t workspace_bytes, void* stream);
/* a2 the frozen stem in one launch, bf16: conv 7x7 s2 p3 (3 -> 64, input channels padded to 8) + FrozenBN + ReLU + max_pool2d(3, 2, 1)
 * (detectron2 BasicStem.forward behind configs/VOC/VOC-RCNN-101-C4-split1.yaml:6-10). x [N][H][W][8], w [64][7][7][8] with the
 * FrozenBN scale folded in (unit_weight_prep), shift [64] fp32, y [N][PH][PW][64], PH = ((H-1)/2+1 - 1)/2 + 1 */
int unit_stem_conv_pool(const void* x, const void* w, const float* shift, void* y, int dtype, int N, int H, int W, void* stream);
int unit_maxpool3x3s2_fwd(const void* x, void* y, int dtype, int N, int H, int W, int C, void* stream);
/* Res5BoxHead.forward x.mean(dim=[2,3]): modeling/roi_heads/box_head.py:80 */
int unit_global_avgpool_fwd(const void* x, void* y, int dtype, int R, int P, int C, void* stream);
int unit_global_avgpool_bwd_relu(const void* dfeat, const void* out, void* g, int dtype, int R, int P, int C, void* stream);

/* ---- K5 anchors (DefaultAnchorGenerator via modeling/proposal_generator/rpn.py:22) ---- */
int unit_anchor_grid(float* out, int H, int W, int A, float stride, float offset, const float* cell_dev, void* stream);

/* ---- K6/K7 pairwise IoU + Matcher: modeling/matcher.py:54-120 (and the stock d2 matcher via rpn.py:41, roi_heads.py:563) */
size_t unit_iou_match_workspace_bytes(int B, int Mcap);
int unit_iou_match(const float* gt, const int* gt_count, int B, int Mcap, const float* boxes, long box_batch_stride,
                   const int* box_count, int Ncap, const float* thresholds, const int* labels, int n_thresh,
                   int allow_low_quality, int64_t* match_idx, int8_t* match_label, float* match_val, void* workspace,
                   size_t workspace_bytes, void* stream);
int unit_pairwise_iou(const float* b1, int M, const float* b2, int Nb, float* out, void* stream);
/* Matcher.__call__(match_quality_matrix): modeling/matcher.py:54 on a precomputed M x N matrix; rowmax_ws: M floats */
int unit_match_matrix(const float* q, int M, int N, const float* thresholds, const int* labels, int n_thresh, int allow_low_quality,
                      int64_t* match_idx, int8_t* match_label, float* match_val, float* rowmax_ws, void* stream);

/* ---- K8 subsample_labels with the explicit-permutation contract (detectron2.modeling.sampling via rpn.py:41,
 * roi_heads.py:563, roi_heads.py:415) ---- */
int unit_subsample_labels(const void* labels, int labels_are_int64, const int* count, int B, int Ncap, const int* perm, int Pcap,
                          int num_samples, int max_pos, int bg_label, int8_t* out_labels, int* sampled_idx, int* out_counts,
                          void* stream);

/* ---- K9 Box2BoxTransform (rpn.py:70, fast_rcnn.py:71, weak_detector_fast_rcnn.py:275) ---- */
int unit_box_encode(const float* src, const float* tgt, const float* weights4, float* out, int n, void* stream);
int unit_box_decode(const float* deltas, int ld, int col0, int K, const float* boxes, const float* weights4, float scale_clamp,
                    float* out, int n, void* stream);

/* ---- K10/K11/a6 find_top_rpn_proposals (detectron2 via rpn.py:48): stable descending sort, decode+clip+filter, NMS */
size_t unit_sort_workspace_bytes(int B, int n);
int unit_sort_desc_stable(const float* src, long batch_stride, int ld, int A, int col0, int B, int n, float* out_keys, int* out_idx,
                          void* workspace, size_t workspace_bytes, void* stream);
/* top-k form (chip-wide select + rank sort): the first min(topk, #keys > min_exclusive) entries of each output row are the ranked
 * keys (NaN keys are never ranked: torch.topk would put them first and detectron2's find_top_rpn_proposals then drops them as
 * non-finite or, in training, raises "Training has diverged" -- here TrainerNoMeta.loss_dict() raises on the NaN losses instead); every
 * entry that is not a ranked candidate holds index -1 / key -INFINITY (unit_rpn_decode_select skips negative indices); pass -INFINITY
 * to rank every key */
int unit_sort_desc_stable_topk(const float* src, long batch_stride, int ld, int A, int col0, int B, int n, int topk,
                               float min_exclusive, float* out_keys, int* out_idx, void* workspace, size_t workspace_bytes,
                               void* stream);
int unit_rpn_decode_select(const float* head, long head_batch_stride, int ld, int A, int delta_col0, const float* anchors,
                           const int* sorted_idx, const float* sorted_logit, int B, int Ncap, int topk, const float* image_hw_dev,
                           float scale_clamp, float min_size, float* cand_boxes, float* cand_scores, int* cand_count, void* stream);
/* entries behind keep_count[b] are written too: keep_idx = -1, out_boxes / out_scores = 0 */
size_t unit_nms_workspace_bytes(int B, int cap);
int unit_nms(const float* boxes_sorted, const float* scores_sorted, const int* count, int B, int cap, float thresh, int max_keep,
             int* keep_idx, int* keep_count, float* out_boxes, float* out_scores, void* workspace, size_t workspace_bytes,
             void* stream);

/* ---- a7 ROIHeads.label_and_sample_proposals plumbing (roi_heads.py:563, :566-572) ---- */
int unit_append_gt(const float* props, const int* pcount, int Pcap, const float* gt, const int* gcount, int Mcap, int B, float* cat,
                   int* ccount, void* stream);
int unit_roi_classes(const int64_t* match_idx, const int8_t* match_label, const int* count, const int64_t* gt_classes,
                     const int* gcount, int Mcap, int B, int Ncap, int K, int64_t* cls, void* stream);
int unit_gather_rois(const float* cat, int Ncap, const int* sampled_idx, int S, const int64_t* cls, const int64_t* match_idx,
                     const float* gt, const int* gcount, int Mcap, int B, float* rois, int* roi_cls, float* roi_gt, void* stream);
int unit_first_k_rois(const float* props, const int* pcount, int Pcap, int S, int B, int batch_index_offset, float* rois, int* valid,
                      void* stream);

/* ---- a8 RoIAlign (ROIAlignV2): roi_heads.py:499,511,708 via detectron2 ROIPooler ---- */
int unit_roi_align_fwd(const void* feat_nhwc, int dtype, int N, int H, int W, int C, const float* rois, const int* roi_count_dev,
                       int R, int pooled_size, int out_size, int bin_step, float spatial_scale, int sampling_ratio, int aligned,
                       void* out, void* stream);
int unit_roi_align_bwd(const void* gout, int dtype, int N, int H, int W, int C, const float* rois, const int* roi_count_dev, int R,
                       int pooled_size, int out_size, int bin_step, float spatial_scale, int sampling_ratio, int aligned,
                       float* dfeat_f32, void* stream);

/* deterministic gather-form backward (no atomics); optional fused "+ addend, * (mask_ref > 0), cast" consumer epilogue */
size_t unit_roi_align_bwd_gather_workspace_bytes(int R);
int unit_roi_align_bwd_gather(const void* gout, int dtype, int N, int H, int W, int C, const float* rois, const int* roi_count_dev,
                              int R, int rois_per_image, int image_offset, int pooled_size, int out_size, int bin_step,
                              float spatial_scale, int sampling_ratio, int aligned, const void* addend, int addend_images,
                              const void* mask_ref, void* dfeat, int out_dtype, void* workspace, size_t workspace_bytes,
                              void* stream);

/* ---- a5/a10/a11/a12 fused loss forward+backward kernels ---- */
/* scratch (unit_rpn_loss_scratch_bytes, contents irrelevant): per-workgroup partial sums, added by the last workgroup in a fixed
 * order -- the two loss scalars are bit-reproducible */
size_t unit_rpn_loss_scratch_bytes(int B, int Ncap);
int unit_rpn_loss(const float* head, int ld, int A, int dcol0, const int8_t* labels, const int64_t* match_idx, const float* gt_boxes,
                  int Mcap, const float* anchors, int B, int Ncap, float normalizer, float gscale, float* loss2, void* dhead,
                  int dhead_dtype, float* scratch, size_t scratch_bytes, void* stream);
/* unit_rpn_loss with Detectron2's per-loss weights (rpn.py:100 `v * self.loss_weight.get(k, 1.0)`: MODEL.RPN.LOSS_WEIGHT for loss_rpn_cls,
 * LOSS_WEIGHT * BBOX_REG_LOSS_WEIGHT for loss_rpn_loc): loss values and gradients are scaled */
int unit_rpn_loss_w(const float* head, int ld, int A, int dcol0, const int8_t* labels, const int64_t* match_idx, const float* gt_boxes,
                    int Mcap, const float* anchors, int B, int Ncap, float normalizer, float gscale, float w_cls, float w_loc, float* loss2,
                    void* dhead, int dhead_dtype, float* scratch, size_t scratch_bytes, void* stream);
int unit_sup_scores(const float* delta, int ldd, int dcol0, const float* weak, int ldw, int wcol0, int n_oicr, int ncls,
                    const unsigned char* novel_mask_dev, const float* extra, int lde, int ecol0, float* out, int ldo, int R,
                    void* stream);
/* acc (may be NULL): 8 device bytes, zero at the call and handed back zero, private to the stream -- with it the rows are spread over one-wave
 * workgroups and their partial losses are added through it in fixed point (bit-reproducible whatever the arrival order); without it one
 * workgroup walks all rows */
int unit_softmax_ce(const float* logits, int ld, int col0, int ncls, const int* labels, const float* weights, int R, float gscale,
                    float* loss, void* dy, int dy_dtype, int ldd, int dcol0, unsigned long long* acc, void* stream);
int unit_box_reg_loss(const float* bbox, int ld, int col0, int K, const int* labels, const float* rois5, const float* gt_boxes,
                      const float* weights4, int R, float gscale, float* loss, void* dy, int dy_dtype, int ldd, int dcol0,
                      unsigned long long* acc, void* stream);
int unit_wsddn_mil(const float* streams, int ld, int ccol0, int dcol0, int K, const int* valid, int S, int B,
                   const unsigned char* multihot, float cls_temp, float det_temp, float mil_multiplier, float gscale, float* loss,
                   float* xr_out, void* dy, int dy_dtype, int ldd, int dyc0, int dyd0, void* stream);
int unit_oicr_targets(const float* src, int ld, int col0, int mode, int K, const float* rois5, const int* valid, int S, int B,
                      const unsigned char* multihot, float fg_thresh, float bg_thresh, int* labels, float* weights, void* stream);
int unit_sum_losses(const float* losses, int n, float* out, void* stream);
/* sampling permutations (d2 `subsample_labels` -> torch.randperm; call sites rpn.py:41, roi_heads.py:563): keys [B][n] = hash of
 * (seed, *counter_dev, stream_id, b, i) as positive finite floats; `unit_sort_desc_stable` of them yields the permutation.
 * The device-resident counter is advanced by unit_counter_bump (graph-replay safe). */
int unit_perm_keys(unsigned long long seed, const long long* counter_dev, int stream_id, int B, int n, float* keys, void* stream);
int unit_counter_bump(long long* counter_dev, long long delta, void* stream);

/* ---- a14 base->novel similarity transfer: modeling/roi_heads/roi_heads.py:245-336, fast_rcnn.py:376-382,401-423,504-533 ---- */
int unit_embedding_similarity(const float* emb, int ld, int dim, const int* novel_rows, int n_novel, const int* base_rows, int n_base,
                              float* out, void* stream);
int unit_similarity(const float* lin_weak, int ld, int col0, int n_oicr, int ncls, const int* base_dev, int n_base, const float* lingual,
                    int n_novel, float visual_threshold, int use_lingual, int use_visual, float* sim, int R, void* stream);
int unit_transfer_predictions(const float* lin, int ld, int ccol0, int bcol0, int K, const float* weak, int ldw, int wcol0, int n_oicr,
                              const float* ft, int ldf, int fccol0, int fbcol0, const float* sim_cls, const float* sim_bbox,
                              const int* base_dev, int n_base, const int* novel_dev, int n_novel, const int8_t* role_dev,
                              const int* slot_dev, float* scores, int lds, float* bbox, int ldb, int R, void* stream);
/* backward of the two above for the fine-tune configurations whose box head trains (COCO-RCNN-50-C4-split1-segm-ft.yaml; the reference
 * computes the similarity with grad in training, roi_heads.py:852): dy = d(loss)/d[scores | bbox] in the ft heads' layout ->
 * dlin (delta heads' layout, zero-filled first) and dsim [R][n_novel][n_base] (written); then dsim -> the OICR logit columns of dlin_weak */
int unit_transfer_predictions_bwd(const void* dy, int dy_dtype, int ldd, int dccol0, int dbcol0, const float* lin, int ld, int ccol0,
                                  int bcol0, int K, const float* sim_cls, const float* sim_bbox, const int* base_dev, int n_base,
                                  const int* novel_dev, int n_novel, const int8_t* role_dev, const int* slot_dev, void* dlin, int ldl,
                                  float* dsim, int R, void* stream);
int unit_similarity_bwd(const float* lin_weak, int ld, int col0, int n_oicr, int ncls, const int* base_dev, int n_base,
                        const float* lingual, int n_novel, float visual_threshold, int use_lingual, int use_visual, const float* dsim,
                        void* dlin, int dlin_dtype, int ldl, int dcol0, int R, void* stream);
/* ---- a15 detections: fast_rcnn.py:455-468 -> detectron2 fast_rcnn_inference; rcnn.py:411-429 detector_postprocess ---- */
int unit_softmax_rows(const float* x, int ld, int ncls, float* y, int ldy, int R, void* stream);
int unit_detection_candidates(const float* probs, int ldp, const float* deltas, int ldd, const float* props, const int* pcount, int B,
                              int Rcap, int K, const float* weights4, float scale_clamp, const float* image_hw_dev, float score_thresh,
                              int cap, float* cand_boxes, float* cand_scores, int* cand_class, int* cand_roi, int* cand_count,
                              float* cand_max, void* stream);
int unit_detection_offset_gather(const float* cand_boxes, const int* cand_class, const int* order, const int* cand_count,
                                 const float* cand_max, int B, int cap, float* out, void* stream);
int unit_detection_finalize(const float* cand_boxes, const float* cand_scores, const int* cand_class, const int* cand_roi, const int* order,
                            const int* keep, const int* keep_count, int B, int cap, int topk, float* out_boxes, float* out_scores,
                            int* out_class, int* out_roi, int* out_count, void* stream);
int unit_detector_postprocess(float* boxes, const int* count, int B, int topk, const float* scale_xy_dev, const float* out_hw_dev,
                              unsigned char* nonempty, void* stream);
/* output assembly without stock operators (the reference: boolean-mask indexing per image and field, detectron2 detector_postprocess via
 * meta_arch/rcnn.py:411-429; torch.cat of per-image slices in roi_heads.py:691-710):
 *   unit_compact_detections  stable compaction of the kept detections of every image (j < count[b] [&& nonempty[b][j]]) to the front of its
 *                            block; classes come out as int64 (pred_classes); masks [B][topk][mask_elems] optional; out_count[b] = kept
 *   unit_boxes_to_rois5      [B][T][4] boxes -> [B*T][5] RoIAlign rows (image index, box)
 *   unit_gather_rows         out[b*T + j] = src[b*rcap + max(idx[b][j], 0)] (rows of row_bytes bytes, % 4 == 0)
 *   unit_gather_blocks       the first `take` rows of each of nb blocks of block_rows rows -> dense [nb*take] rows */
int unit_compact_detections(const float* boxes, const float* scores, const int* cls, const int* roi, const float* masks, int mask_elems,
                            const int* count, const unsigned char* nonempty, int B, int topk, float* oboxes, float* oscores, long* ocls,
                            int* oroi, float* omasks, int* out_count, void* stream);
int unit_boxes_to_rois5(const float* boxes, int B, int T, float* rois5, void* stream);
int unit_gather_rows(const void* src, const int* idx, int B, int T, int rcap, int row_bytes, void* out, void* stream);
int unit_gather_blocks(const void* src, int nb, int block_rows, int take, int row_bytes, void* out, void* stream);

/* ---- a16 mask head: modeling/roi_heads/mask_head.py:16-37, roi_heads.py:654-710 (+ detectron2 mask_rcnn_loss/inference,
 * BitMasks.crop_and_resize). The 2x2/s2 ConvTranspose2d is one 1x1 GEMM with 4*Cout columns run by unit_conv2d_fwd. ---- */
int unit_deconv2x2_weight_prep(const float* w, int Cin, int Cout, void* w_fwd, void* w_dgrad, int dtype, void* stream);
int unit_deconv2x2_grad_unpack(const float* dw_gemm, const float* db_gemm, int Cin, int Cout, float* dw, float* db, void* stream);
int unit_mask_targets(const unsigned char* gt_masks, int Mcap, int Hm, int Wm, const float* rois5, const int* gt_index, const int* cls,
                      int K, int S, int M, unsigned char* out, void* stream);
/* the same targets from POLYGON ground truth (the reference's COCO-segm yaml leaves INPUT.MASK_FORMAT at "polygon": mask_head.py:34 ->
 * Detectron2 mask_rcnn_loss -> PolygonMasks.crop_and_resize = rasterize_polygons_within_box -> pycocotools frPyObjects / merge / decode).
 * poly_xy [V][2] fp64 image coordinates of every polygon vertex; poly_start [P + 1] vertex ranges; inst_start [I + 1] polygon ranges of the
 * flat instances; image_inst0 [B] flat index of an image's first instance; slot s uses instance image_inst0[rois5[s][0]] + gt_index[s].
 * Arithmetic = pycocotools' rleFrPoly on the box-relative, M / side scaled vertices (fp64 on the device); cls outside [0, K): zeros. */
int unit_mask_targets_polygon(const double* poly_xy, const int* poly_start, const int* inst_start, const int* image_inst0,
                              const float* rois5, const int* gt_index, const int* cls, int K, int S, int M, unsigned char* out, void* stream);
int unit_mask_bce_loss(const float* logits, int K, int ldk, const int* cls, const unsigned char* targets, int S, int M, float gscale,
                       float* loss, void* dlogits, int d_dtype, void* stream);
/* training form of the fine-tune mask head (mask_head.py:74-93 with similarity['seg'][fg], roi_heads.py:888-906): gt-class logit =
 * transfer(predictor columns; sim[sim_rows[s]]) + predictor_delta column; loss + d(loss)/d(logits) of both column groups, and
 * d(loss)/d(sim) ADDED into dsim[sim_rows[s]] (sim / dsim [R][n_novel][n_base]; NULL sim: no transfer) */
int unit_mask_bce_loss_ft(const float* logits, int K, int ldk, int delta_col0, const int* cls, const unsigned char* targets,
                          const float* sim, const int* sim_rows, const int* base_dev, int n_base, int n_novel, const int8_t* role_dev,
                          const int* slot_dev, int S, int M, float gscale, float* loss, void* dlogits, int d_dtype, float* dsim,
                          void* stream);
/* delta_col0 >= 0: the logits carry K more columns from there (`predictor_delta` of MaskRCNNConvUpsampleHeadWithFineTune,
 * mask_head.py:39-94), added AFTER the base->novel transfer (:91); -1: none */
int unit_mask_probs(const float* logits, int K, int ldk, int delta_col0, const int* cls, const float* sim, const int* base_dev, int n_base,
                    int n_novel, const int8_t* role_dev, const int* slot_dev, int S, int M, float* out, void* stream);
int unit_gather_match_index(const int* sampled_idx, int S, const int64_t* match_idx, int Ncap, int B, int* out, void* stream);
/* paste_masks_in_image of detector_postprocess (modeling/meta_arch/rcnn.py:423 -> d2 layers/mask_ops.py): probs [S][M][M],
 * boxes [S][4] in output-image coordinates, valid [S] or NULL -> uint8 [S][H][W] = (bilinear sample >= threshold) */
int unit_paste_masks(const float* probs, const float* boxes, const unsigned char* valid, int S, int M, int H, int W, float threshold,
                     unsigned char* out, void* stream);

/* ---- K18 SGD momentum (solver/build.py:110-112) ---- */
/* lr_dev (may be NULL): device float holding the step's learning rate; `lr` is then the parameter group's multiplier (a captured
 * hipGraph of the step follows the LR schedule: the host rewrites that one float before every replay) */
int unit_sgd_momentum(float* p, const float* g, float* buf, long n, float lr, float momentum, float wd, float grad_scale,
                      int first_step, const float* lr_dev, void* stream);

/* ---- the step's launch sequence as a call list walked in C (csrc/replay.hip; unit_amd/_lib.py Recorder, engine.ReplayedStep) ----
 * The reference's step is `loss_dict = self.model(data, ...); losses.backward(); self.optimizer.step()` (engine/defaults.py:279-284):
 * ~650 operator launches issued one by one from Python. Here the sequence of C-ABI calls of one eagerly executed step (a constant for
 * a given batch key: the step has no host sync and no data-dependent host branch) is recorded and re-issued by ONE call per segment:
 * same launches, same in-order streams, same schedule as eager -- minus the interpreter. calls = [n] records of unit_call_bytes() each:
 * {void* fn; int n_int, n_flt; long long i[32]; float f[8]} = a function of this header, its integer-class arguments (ints, sizes,
 * pointers) and its float arguments in declaration order (x86-64 SysV: the two classes travel in separate register files). Every
 * pointer in the list must stay valid and every shape unchanged between replays (the recorder pins the recorded step's allocations
 * in a private memory pool and keeps the host structs alive). *failed = index of the first call that did not return 0, or -1. */
size_t unit_call_bytes(void);
int unit_replay(const void* calls, int n, int* failed);
/* hipEventRecord / hipStreamWaitEvent on a caller-owned hipEvent_t: what torch.cuda.Event.record / .wait do, as list entries */
int unit_event_record_raw(void* event, void* stream);
int unit_stream_wait_event_raw(void* stream, void* event);
/* host-only self-test of the generic call (tests/test_replay_cpu.py): out[0] = sum of (i + 1) * i-th integer-class argument,
 * out[1] = the four floats x 1000, packed in decimal */
int unit_replay_selftest(int a0, const void* p1, float f0, long a2, int a3, float f1, int a4, int a5, size_t a6, int a7, int a8, float f2,
                         int a9, int a10, int a11, int a12, int a13, int a14, int a15, int a16, int a17, int a18, int a19, int a20,
                         int a21, int a22, int a23, int a24, int a25, int a26, float f3, int a27, long long* out);

#ifdef __cplusplus
}
#endif
#endif
