"""Functional Python wrappers over the libunit_hip.so C ABI (include/unit_hip.h).

PyTorch is used here only as plumbing: device memory (torch.empty on the ROCm device), the current HIP stream and
dtype bookkeeping. Every arithmetic step is one of the hand-written HIP kernels; there is no CPU / eager fallback --
a missing extension or a CPU tensor raises.
"""
import ctypes
import os
import math

import torch

from ._lib import _DEBUG_SYNC, check, lib

F32, BF16 = 0, 1
SCALE_CLAMP = math.log(1000.0 / 16)  # Box2BoxTransform scale_clamp (detectron2 default)


def dt(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {dtype}")


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("unit_amd ops need tensors on the ROCm device (no CPU fallback)")
    if type(t) is X3:
        raise TypeError("a bf16x3 split tensor (ops.X3) was handed to a kernel that reads plain fp32: convert with ops.as_f32 first")
    return ctypes.c_void_p(t.data_ptr())


class X3(torch.Tensor):
    """A bf16x3 "split" activation (csrc/split.hip): logically fp32 [..., C], physically two bf16 planes per row, [...][2][C] --
    hi = bf16(x), lo = bf16(x - hi) -- in the 4 bytes per element of a float32 tensor of the SAME shape. The subclass is the marker:
    it survives slicing / views / empty_like / cat (row-wise operations are all the plan does to activations), the conv wrappers take
    it, every other wrapper refuses it (`_p`) until `as_f32` has merged the planes. Never do arithmetic on it with torch operators."""


def _px(t):
    """device pointer of a split tensor (or None)"""
    if t is None:
        return None
    if type(t) is not X3:
        raise TypeError("expected a bf16x3 split tensor (ops.X3): convert with ops.as_x3 first")
    return ctypes.c_void_p(t.data_ptr())


def x3_split(x, out=None):
    """fp32 [..., C] -> X3 of the same shape (unit_x3_split)"""
    assert type(x) is not X3 and x.dtype == torch.float32 and x.is_contiguous(), "x3_split: contiguous fp32"
    c = x.shape[-1]
    y = out if out is not None else torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib().unit_x3_split(_p(x), ctypes.c_void_p(y.data_ptr()), x.numel() // c, c, _s()), "x3_split")
    return y.as_subclass(X3)


def x3_merge(x, out=None):
    """X3 -> plain fp32 of the same shape (hi + lo, exact)"""
    assert type(x) is X3 and x.is_contiguous()
    c = x.shape[-1]
    y = out if out is not None else torch.empty(x.shape, dtype=torch.float32, device=x.device).as_subclass(torch.Tensor)
    check(lib().unit_x3_merge(_px(x), _p(y), x.numel() // c, c, _s()), "x3_merge")
    return y


def as_x3(x):
    return x if (x is None or type(x) is X3) else x3_split(x)


def as_f32(x):
    return x3_merge(x) if type(x) is X3 else x


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


_STREAM_OVERRIDE = []          # stack of raw HIP stream handles: `on_stream` below


def raw_stream(device_index=None):
    """the HIP stream the next launch goes to, as an int: the innermost `on_stream` handle, else torch's current stream.
    torch.cuda.current_stream() builds a Stream object through several Python layers (~4 us, ~750 calls per step = a quarter of
    the step's host time); the raw getter is one C call"""
    if _STREAM_OVERRIDE:
        return _STREAM_OVERRIDE[-1]
    if _raw_stream is None:
        return torch.cuda.current_stream().cuda_stream
    return _raw_stream(torch.cuda.current_device() if device_index is None else device_index)


def _s():
    return ctypes.c_void_p(raw_stream())


class on_stream:
    """`with on_stream(side):` -- the C-ABI launches inside go to the torch.cuda.Stream `side` WITHOUT switching torch's current stream
    (torch.cuda.stream() costs ~15 us of Python per enter / exit, and the step forks ~100 weight-gradient launches per step). Only for
    blocks that launch through this module and allocate nothing whose lifetime depends on the stream; while bench.py's event profiler
    is on, it degrades to torch.cuda.stream (the profiler records torch events on the current stream)."""

    def __init__(self, stream):
        self.stream = stream
        self.ctx = None

    def __enter__(self):
        if PROFILER is not None or torch.cuda.is_current_stream_capturing():
            self.ctx = torch.cuda.stream(self.stream)
            self.ctx.__enter__()
        else:
            _STREAM_OVERRIDE.append(self.stream.cuda_stream)
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        else:
            _STREAM_OVERRIDE.pop()
        return False


def streams_on_distinct_queues(device, n, candidates=12, spin_cycles=200000):
    """-> n torch.cuda.Stream objects that sit on n DIFFERENT hardware queues, none of them the queue of the calling thread's current stream.
    The ROCm runtime maps all HIP streams of a process onto GPU_MAX_HW_QUEUES (4) hardware queues in an order that depends on every stream
    created before -- PyTorch's pool, RCCL's process group ... -- and two streams on one queue run strictly one after the other. Which of the
    step's streams share a queue decides ~10 % of its time (head stream on the main stream's queue, as happens once a process group exists:
    18.2 instead of 16.3 ms), so the choice is made by measurement: two single-wave spin kernels (unit_debug_spin) on two
    streams take one spin time on different queues and two on the same. Costs ~10 ms once. Falls back to fewer distinct streams (roles then
    share a stream OBJECT) when the runtime offers fewer queues."""
    import time
    cur = torch.cuda.current_stream(device)
    sink = torch.zeros(4, dtype=torch.int32, device=device)
    cyc = ctypes.c_longlong(int(spin_cycles))

    def spin(st):
        check(lib().unit_debug_spin(cyc, _p(sink), ctypes.c_void_p(st.cuda_stream)), "debug_spin")

    def pair(a, b):
        best = 1e9
        for _ in range(2):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            spin(a)
            spin(b)
            torch.cuda.synchronize(device)
            best = min(best, time.perf_counter() - t0)
        return best

    cands = [torch.cuda.Stream(device) for _ in range(candidates)]
    spin(cur)
    one = pair(cands[0], cands[0]) / 2.0          # two spins on ONE stream: strictly serial
    share = lambda a, b: pair(a, b) > 1.5 * one
    reps = []
    for c in cands:
        if len(reps) == n:
            break
        if share(c, cur) or any(share(c, r) for r in reps):
            continue
        reps.append(c)
    while len(reps) < n:
        # Fewer distinct queues FOUND than roles -- the runtime offers fewer, or the measurement was disturbed (another process on the same
        # GPU: the two-ranks-one-GPU rehearsals). The roles still get stream OBJECTS of their own (round 6): two of them may then share a
        # hardware queue and run in turn, which costs time, not correctness. Until round 5 the roles were aliased onto one object here; with
        # the weight-gradient and the RPN-branch role on ONE stream object the step's capture makes ROCm 7.2's hipStreamEndCapture segfault
        # (deterministic: UNIT_STREAM_MERGE=wr / all in a single process, profiles/r06_exp_capture_crash_root_cause.txt) -- the "1 in 12-24"
        # crash of DESIGN section 8 round 5 item 10 was this fallback firing when the spin probe of a rank was disturbed by the other rank.
        reps.append(torch.cuda.Stream(device))
    return reps


def stream_wait_stream(waiter, signaller_raw=None):
    """everything enqueued on torch.cuda.Stream `waiter` from now on waits for what is on the stream with raw handle `signaller_raw`
    (default: the current launch stream) so far: one C call (unit_stream_wait_stream) instead of Event() + record + wait_event"""
    sig = raw_stream() if signaller_raw is None else signaller_raw
    check(lib().unit_stream_wait_stream(ctypes.c_void_p(waiter.cuda_stream), ctypes.c_void_p(sig)), "unit_stream_wait_stream")


_WS = {}



def _big_tile_default(dtype, m, k, c, kgemm):
    """use the 256x256 LDS-DMA kernel when it pays (measured, tools/microbench.py): bf16, C % 64 == 0, at least half a
    round of 256x256 tiles and a k-extent long enough to amortise the 128 KB-per-tile epilogue"""
    if dtype != torch.bfloat16 or c % 64 != 0 or k < 256 or kgemm < 512:
        return False
    tiles = ((m + 255) // 256) * ((k + 255) // 256)
    return tiles >= 128


_NO_MID = bool(int(__import__("os").environ.get("UNIT_NO_MID_TILE", "0")))   # A/B switch for tools/ and debugging


_LC_TWO = bool(int(os.environ.get("UNIT_LC_TWO", "0")))      # A/B switch: 1 = the two-workgroups-per-CU form of the loader / consumer kernel where it won in isolation
_NO_LC = bool(int(os.environ.get("UNIT_NO_LC", "0")))      # A/B switch: 1 = never the persistent loader / consumer conv kernel (csrc/conv_igemm_lc.hip)
# 1 = the loader / consumer kernel's weights-direct form (csrc/conv_igemm_lc.hip WD: weight fragments from L2 straight into the consumers'
# registers, only pixels through LDS); 0 (default) = both operands staged in LDS. Measured SLOWER in round 6 (profiles/r06_exp_weights_direct.txt:
# res4 1x1 1024 -> 256 12.6 -> 14.0 us, 3x3 20.5 -> 24.2 us, step 15.12 -> 15.69 ms): the k-step is not waiting for the LDS port, it waits for
# what one CU takes in from L2 (~29 B/clk, the guide's "rows shared by every workgroup" rate), and 64-byte fragment rows are a worse shape for
# that path than the 128-byte rows of the LDS-DMA pieces. Kept as a bit-identical variant (tile codes + 8000) and as the evidence.
_LC_WD = int(os.environ.get("UNIT_LC_WD", "0"))
_MID96 = int(os.environ.get("UNIT_MID96", "0"))      # 0: off; 1: 96x128 tiles where tools/mid_sweep.py found them faster in isolation; 2: only the two-per-CU form


def mid_tile_dims(mid):
    """(BM pixels, BN channels) of a `mid` tile code: 0..5 = the 4-wave LDS-DMA tiles, 100 + 10 * (BM / 16) + BN / 64 = loader / consumer tiles
    (+ 2000: their two-workgroups-per-CU form; + 8000: weights fetched straight into registers)"""
    if mid >= 8000:
        mid -= 8000
    code = mid - 2000 if mid >= 2000 else mid
    if code >= 100:
        return (code - 100) // 10 * 16, (code % 10) * 64
    return {0: (128, 128), 1: (64, 128), 2: (128, 64), 3: (64, 128), 4: (96, 128), 5: (96, 128)}[code]


def lc_tile_code(m, k, kgemm):
    """tile of the persistent loader / consumer kernel (csrc/conv_igemm_lc.hip) for an [m pixels] x [k channels] x [kgemm] layer:
    100 + 10 * (BM / 16) + BN / 64. A workgroup's time is its tiles x k-steps x the (BM + BN) * 128 bytes a k-step stages (the CU's
    intake is what bounds it) plus an epilogue per tile; the grid is one workgroup per CU. (M = 9 576, 256 channels: 80 x 128 = 240
    tiles, one per CU, measured best of the eight shapes; 1024 channels: 80 x 256.)"""
    best = None
    for fb, fa in ((4, 2), (5, 2), (6, 2), (7, 2), (8, 2), (4, 4), (5, 4), (6, 4)):
        bm, bn = fb * 16, fa * 64
        tiles = ((m + bm - 1) // bm) * ((k + bn - 1) // bn)
        per_wg = (tiles + 255) // 256
        cost = per_wg * ((kgemm // 64) * (bm + bn) + 0.5 * bm * bn / 64.0)
        if best is None or cost < best[0]:
            best = (cost, 100 + 10 * fb + fa)
    return best[1]


def _mid_tile_default(dtype, m, k, c, kgemm, allow_lc=True):
    """allow_lc=False: the caller's output cannot be written by the loader / consumer kernel (fp32 output, rows not a multiple of 16 bytes):
    choose among the 4-wave tiles only.
    -1: use the register-staged conv_igemm.hip kernel; 0..5: LDS-DMA 4-wave kernel with that tile (csrc/conv_igemm128.hip);
    >= 100: persistent loader / consumer workgroups (csrc/conv_igemm_lc.hip, `lc_tile_code`) -- the layers with a long contraction
    and few output tiles (res4 and the res3 -> res4 transition on four 600x1000 images: 1x1 1024 -> 256 15.0 -> 11.9 us, 3x3 256 -> 256
    26.8 -> 18.9 us, tools/lc_sweep.py); short contractions (K = 256 -> 1024: four k-steps per tile) and the large res2 / res3 maps stay
    on the 4-wave tiles, which measured equal or better there.
    Measured on the backbone shapes (tools/microbench.py): 128x64 for 64-channel outputs, 128x128 when that tiling still
    gives every CU a workgroup or two, 64x128 below that."""
    if dtype != torch.bfloat16 or c % 64 != 0 or k < 64 or _NO_MID:
        return -1
    if k <= 64:
        return 2
    tiles = ((m + 127) // 128) * ((k + 127) // 128)
    # two loader / consumer workgroups per CU on a two-slot ring (code + 2000; off unless UNIT_LC_TWO=1): where every CU has two or more
    # 80 x 128 tiles, one workgroup's epilogue runs beside the other's k-steps. Isolated (tools/lc_sweep.py two): res4 256 -> 1024
    # 17.3 -> 16.0 us, res3 512 -> 128 14.4 -> 12.8, res3 3x3 21.9 -> 17.6 (with one tile per CU -- res4 -> 256 layers -- the three-slot
    # one-per-CU form stays ahead, with two k-steps per tile -- res3 128 -> 512 -- the 4-wave kernel, the 1024-wide transition layer is a
    # tie). In the step: 16.31 ms without vs 16.35 ms with it (three alternating runs each) -- not enabled.
    t80 = ((m + 79) // 80) * ((k + 127) // 128)
    lc = allow_lc and not _NO_LC
    if lc and _LC_TWO and k % 8 == 0 and kgemm >= 256 and 384 <= t80 <= 1024 and not (kgemm >= 512 and k >= 1024):
        return 2152
    if lc and kgemm >= 512 and tiles <= 640 and k % 8 == 0:
        return lc_tile_code(m, k, kgemm) + (8000 if _LC_WD else 0)
    if _MID96:
        # 96-row tiles (tools/mid_sweep.py, profiles/r02_exp_mid_sweep_96_row_tiles.txt): the res4 1x1 -> 256 layers become 100 x 2 = 200
        # workgroups, one round with one workgroup per CU (14.9 vs 15.6 us; 3x3: 26.7 vs 27.9); between one and 2.5 rounds of 128x128
        # tiles the 2-stage 96x128 form (two workgroups per CU) wins: res3 512 -> 128 12.7 vs 15.4 us, 3x3 19.6 vs 22.1, res4 256 -> 1024 16.6 vs 17.8
        if tiles < 256:
            return (4 if ((m + 95) // 96) * ((k + 127) // 128) <= 256 else 1) if _MID96 == 1 else 1
        if tiles <= 640:
            return 5
        return 0
    return 0 if tiles >= 256 else 1      # tools/mid_sweep.py: res3 (293 tiles) 22.4 vs 26.2 us with 128x128; res4 (150) 27.5 vs 32.9 with 64x128


MID_TILE_POLICY = _mid_tile_default
BIG_TILE_POLICY = _big_tile_default
# 0: one barrier per k-tile, 256-row tiles; 1: ping-pong wave groups; 2: four 32-k stages; 3: 224-row tiles; 5: 224 or 256
# rows per launch, whichever needs fewer rounds x rows (faster for isolated launches, see csrc/conv_igemm256.hip)
BIG_TILE_VARIANT = int(__import__("os").environ.get("UNIT_BIG_VARIANT", "0"))

# side HIP stream for the weight-gradient kernels (set by the model when stream overlap is enabled; None = inline)
WGRAD_STREAM = None
# True while a SECOND backbone backward of the same step runs (ragged supervised / weak batches): weight gradients accumulate
WGRAD_ACCUMULATE = False
# True inside the module-level training backward (modeling/train_modules.py): conv weight gradients go straight into `.grad`, not through the
# fused step's multi-tensor plan (slabs + bucket reductions)
WGRAD_DIRECT = False

# bench.py sets this to a dict to time every conv_igemm launch with HIP events on the launch stream (roofline evidence)
PROFILER = None


class _timed:
    """`with _timed("name", flops, bytes):` -- when bench.py has set PROFILER, brackets the launches inside with a HIP-event pair on
    the launch stream and files (e0, e1, flops, algorithmic bytes) under `name`; free otherwise"""

    def __init__(self, name, flops=0.0, nbytes=0.0, ref_bytes=None):
        self.name, self.flops, self.nbytes, self.ref_bytes = name, flops, nbytes, ref_bytes

    def __enter__(self):
        self.prof = PROFILER
        if self.prof is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.prof is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.prof.setdefault(self.name, []).append((self.e0, e1, self.flops, self.nbytes, self.ref_bytes))
        return False


_WS_RETIRED = []
_GRAPHS_ALIVE = [False]          # set by engine.GraphedStep at its first capture: from then on an outgrown buffer's address may be baked into a graph


def retain_retired_buffers():
    """engine.GraphedStep, before its first capture: outgrown workspaces / weight-gradient slabs are kept alive from now on (a captured
    hipGraph replays into the addresses it recorded). Without a graph they are returned to the allocator."""
    _GRAPHS_ALIVE[0] = True


WS_GROWTHS = [0]          # scratch / slab buffers replaced by bigger ones so far (bench.py --shapes voc reports the count per run)


def _retire(buf):
    """an outgrown scratch buffer: kept for ever only while a captured graph may replay into it; otherwise handed back to the caching
    allocator -- after the work already queued on the LAUNCH stream (which need not be torch's current stream: `on_stream`), hence the
    record_stream. With multi-scale inputs (ResizeShortestEdge 480-800) the per-conv slabs regrow many times; keeping every old one pinned
    tens of MB each for the life of the process."""
    WS_GROWTHS[0] += 1
    if _GRAPHS_ALIVE[0]:
        _WS_RETIRED.append(buf)
    elif buf.is_cuda:
        buf.record_stream(torch.cuda.ExternalStream(raw_stream(buf.device.index), device=buf.device))


def _grown(nbytes, old):
    """new capacity for a buffer that must hold nbytes: geometric (x1.25 over the old size at least) so that a slowly rising maximum does
    not reallocate at every step"""
    have = old.numel() if old is not None else 0
    return max(int(nbytes), have + have // 4)


def workspace(nbytes, device, slot=0):
    """scratch buffer for the kernels that need one; one buffer per (slot, HIP stream): the step runs the RPN-loss branch,
    the proposal chain and the weight gradients on different streams at the same time, and scratch must never be shared
    between kernels that are not ordered by a stream"""
    key = (device, slot, raw_stream(device.index) if device.type == "cuda" else 0)
    w = _WS.get(key)
    if w is None or w.numel() < nbytes:
        if w is not None:
            _retire(w)      # a captured hipGraph (engine.GraphedStep) may have this address baked into its launches: see _retire
        w = torch.empty(max(_grown(nbytes, w), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = w
    return w


# ------------------------------------------------------------------------------------------------ a1
def preprocess_images(images, pixel_mean, pixel_std, dtype=torch.bfloat16, cpad=8, normalize_images=False, out=None):
    """images: list of CHW fp32 device tensors (0..255). -> ([N,Hmax,Wmax,cpad] NHWC, [(h,w)...])  rcnn.py:257-266"""
    sizes = [(int(x.shape[-2]), int(x.shape[-1])) for x in images]
    hm, wm = max(s[0] for s in sizes), max(s[1] for s in sizes)
    n = len(images)
    if out is None:
        out = torch.empty((n, hm, wm, cpad), dtype=dtype, device=images[0].device)
    mean = (ctypes.c_float * 3)(*[float(v) for v in pixel_mean])
    std = (ctypes.c_float * 3)(*[float(v) for v in pixel_std])
    for i, x in enumerate(images):
        x = x.contiguous()
        if x.dtype != torch.float32:
            raise TypeError("preprocess_images expects fp32 CHW images")
        check(lib().unit_preprocess_image(_p(x), x.shape[0], sizes[i][0], sizes[i][1], mean, std,
                                          255.0 if normalize_images else 1.0, _p(out[i]), dt(dtype), hm, wm, cpad, _s()),
              "unit_preprocess_image")
    return out, sizes


def nchw_to_nhwc(x, dtype=torch.bfloat16, cpad=None):
    n, c, h, w = x.shape
    cp = cpad or c
    y = torch.empty((n, h, w, cp), dtype=dtype, device=x.device)
    check(lib().unit_nchw_to_nhwc(_p(x.contiguous().float()), _p(y), dt(dtype), n, c, h, w, cp, _s()), "nchw_to_nhwc")
    return y


def nhwc_to_nchw(x, c=None):
    n, h, w, cp = x.shape
    c = c or cp
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    check(lib().unit_nhwc_to_nchw(_p(x.contiguous()), dt(x.dtype), _p(y), n, c, h, w, cp, _s()), "nhwc_to_nchw")
    return y


def zeros(shape, dtype, device):
    """torch.zeros by the library's own fill kernel (unit_fill_zero): the step launches no stock fill"""
    t = torch.empty(shape, dtype=dtype, device=device)
    check(lib().unit_fill_zero(_p(t), t.numel() * t.element_size(), _s()), "fill_zero")
    return t


def cast(x, dtype):
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib().unit_cast(_p(x.contiguous()), dt(x.dtype), _p(y), dt(dtype), x.numel(), _s()), "cast")
    return y


def shard_sum(parts, out):
    """out[i] = parts[0][i] + parts[1][i] + ... in that order, fp32 (parts: [nparts, n] fp32 / bf16, contiguous; out: fp32 [n]);
    the owner-side sum of the "direct" gradient exchange (parallel.GradBuckets(mode="direct"))"""
    assert parts.dim() == 2 and parts.is_contiguous() and out.is_contiguous() and out.dtype == torch.float32 and out.numel() == parts.shape[1]
    check(lib().unit_shard_sum(_p(parts), dt(parts.dtype), parts.shape[0], parts.shape[1], _p(out), _s()), "shard_sum")
    return out


def add_cast(a32, b, dtype, mask_ref=None, out=None):
    """y = cast((a32 [+ b]) [* (mask_ref > 0)])"""
    y = out if out is not None else torch.empty(a32.shape, dtype=dtype, device=a32.device)
    check(lib().unit_add_cast(_p(a32), _p(b), _p(mask_ref), _p(y), dt(dtype), a32.numel(), _s()), "add_cast")
    return y


# ------------------------------------------------------------------------------------------------ conv
def conv_out_size(h, w, r, s, stride, pad):
    return (h + 2 * pad - r) // stride + 1, (w + 2 * pad - s) // stride + 1


_POLICY_CACHE = {}


_POLICY_SIG = {}


def _policy_takes_allow_lc(fn):
    """does this tile policy accept the `allow_lc` keyword? Looked at once per policy object (a TypeError raised INSIDE a policy must
    surface, not be mistaken for the old five-argument signature)."""
    r = _POLICY_SIG.get(fn)
    if r is None:
        import inspect
        try:
            ps = inspect.signature(fn).parameters
            r = "allow_lc" in ps or any(q.kind is inspect.Parameter.VAR_KEYWORD for q in ps.values())
        except (TypeError, ValueError):
            r = True
        _POLICY_SIG[fn] = r
    return r


def conv2d(x, w, k, r, s, stride=1, pad=0, bias=None, residual=None, mask_ref=None, relu=False, out_dtype=None,
           out=None, ldy=None, scatter=None, tile_cfg=0):
    """x [N,H,W,C] NHWC ; w [k][r][s][C] (same dtype). Returns y [N,OH,OW,ldy] (or writes the strided scatter target).
    scatter = (oy_mul, OHf, OWf): output pixel (n,oh,ow) lands at (n, oh*oy_mul, ow*oy_mul) of `out` [N,OHf,OWf,ldy].
    x an ops.X3 (bf16x3 split tensor, w prepared by weight_prep_x3): conv2d_x3."""
    if type(x) is X3:
        assert out_dtype is None and ldy is None
        return conv2d_x3(x, w, k, r, s, stride, pad, bias=bias, residual=residual, mask_ref=mask_ref, relu=relu, out=out, scatter=scatter,
                         tile=None if tile_cfg == 0 else tile_cfg)
    n, h, wd, c = x.shape
    oh, ow = conv_out_size(h, wd, r, s, stride, pad)
    out_dtype = out_dtype or x.dtype
    ldy = ldy or ((k + 3) // 4 * 4)
    if scatter is None:
        oy_mul, ohf, owf = 1, oh, ow
    else:
        oy_mul, ohf, owf = scatter
    if out is None:
        if scatter is not None:
            out = zeros((n, ohf, owf, ldy), out_dtype, x.device)
        else:
            out = torch.empty((n, ohf, owf, ldy), dtype=out_dtype, device=x.device)
    prof = PROFILER
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    if tile_cfg == 0:          # the two policy functions cost ~5 us per call (the loader / consumer tile search): cached per shape
        pkey = (x.dtype, n * oh * ow, k, c, r * s * c, out_dtype, ldy % 8, BIG_TILE_POLICY, MID_TILE_POLICY, _NO_LC, _MID96, _LC_TWO, _NO_MID)
        pol = _POLICY_CACHE.get(pkey)
    else:
        pkey = pol = None
    big = tile_cfg in (5, 6, 11, 12, 13, 14, 15, 16, 17, 18, 21, 22) or (tile_cfg == 0 and (pol[0] if pol is not None else BIG_TILE_POLICY(x.dtype, n * oh * ow, k, c, r * s * c)))
    mid = -1
    if pol is not None:
        mid = pol[1]
    elif tile_cfg >= 100:          # persistent loader / consumer workgroups: 100 + 10 * (BM / 16) + BN / 64 (csrc/conv_igemm_lc.hip)
        mid = tile_cfg
    elif tile_cfg in (7, 8, 9, 10, 19, 20):
        mid = {19: 4, 20: 5}.get(tile_cfg, tile_cfg - 7)
    elif tile_cfg == 0 and not big:
        lc_ok = out_dtype == torch.bfloat16 and ldy % 8 == 0        # the loader / consumer kernel writes bf16 rows of 16-byte vectors
        if _policy_takes_allow_lc(MID_TILE_POLICY):
            mid = MID_TILE_POLICY(x.dtype, n * oh * ow, k, c, r * s * c, allow_lc=lc_ok)
        else:                      # a user-supplied policy with the five-argument signature
            mid = MID_TILE_POLICY(x.dtype, n * oh * ow, k, c, r * s * c)
        if mid >= 100 and not lc_ok:          # ... that asked for the loader / consumer kernel anyway: the best 4-wave tile instead
            tiles = ((n * oh * ow + 127) // 128) * ((k + 127) // 128)
            mid = 2 if k <= 64 else (0 if tiles >= 256 else 1)
    if pkey is not None and pol is None:
        _POLICY_CACHE[pkey] = (big, mid)
    if mid >= 0:
        check(lib().unit_conv2d_fwd_mid(_p(x), _p(w), _p(out), _p(bias), _p(residual), _p(mask_ref), dt(out_dtype),
                                        n, h, wd, c, k, r, s, stride, pad, oh, ow, ldy, oy_mul, ohf, owf, int(relu), mid, _s()),
              "unit_conv2d_fwd_mid")
    elif big:
        check(lib().unit_conv2d_fwd_big(_p(x), _p(w), _p(out), _p(bias), _p(residual), _p(mask_ref), dt(out_dtype),
                                        n, h, wd, c, k, r, s, stride, pad, oh, ow, ldy, oy_mul, ohf, owf, int(relu),
                                        BIG_TILE_VARIANT if tile_cfg == 0 else {5: 0, 6: 1, 11: 2, 12: 3, 13: 4, 14: 6, 15: 7, 16: 8, 17: 9, 18: 10, 21: 11, 22: 12}[tile_cfg], _s()),
              "unit_conv2d_fwd_big")
    else:
        check(lib().unit_conv2d_fwd(_p(x), _p(w), _p(out), _p(bias), _p(residual), _p(mask_ref), dt(x.dtype), dt(out_dtype),
                                    n, h, wd, c, k, r, s, stride, pad, oh, ow, ldy, oy_mul, ohf, owf, int(relu), tile_cfg, _s()),
              "unit_conv2d_fwd")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        es = out.element_size()
        nbytes = (x.numel() + w.numel()) * x.element_size() + n * oh * ow * ldy * es * (1 + (residual is not None) + (mask_ref is not None))
        feed = None
        if mid >= 0:          # operand bytes the busiest CU stages into LDS (feed-bound ceiling of the backbone layers, bench.py backbone_ceiling)
            bm, bn = mid_tile_dims(mid)
            tiles = ((n * oh * ow + bm - 1) // bm) * ((k + bn - 1) // bn)
            feed = ((tiles + 255) // 256) * (r * s * c // 64) * (bm + bn) * 128.0
        prof.setdefault("conv_igemm256" if (big and mid < 0) else ("conv_igemm_dma" if mid >= 0 else "conv_igemm"), []).append(
            (e0, e1, 2.0 * n * oh * ow * k * r * s * c, nbytes, feed))
    return out


FUSE_EPILOGUE = os.environ.get("UNIT_FUSE_EPILOGUE", "1") != "0"          # Res5 heads: average pool + ReLU bit mask inside the last conv's epilogue (conv2d_ex); False = separate kernels


FUSE_DUAL = os.environ.get("UNIT_FUSE_DUAL", "1") != "0"          # first Res5 block: conv3 + shortcut (and their dgrads) as dual-input GEMMs


def conv_ex_supported(dtype, c, ldy):
    return dtype == torch.bfloat16 and c % 64 == 0 and ldy % 64 == 0


class ReluBits:
    """(map > 0) of an [R, bins, C] activation as one bit per element in the conv epilogue's order (include/unit_hip.h,
    unit_conv2d_fwd_big_ex); `self[a:b]` = the RoIs a..b of it (same storage, RoI offset)."""

    def __init__(self, data, r, bins, c, roi0=0):
        self.data, self.r, self.bins, self.c, self.roi0 = data, r, bins, c, roi0

    def __getitem__(self, sl):
        start, stop, step = sl.indices(self.r)
        assert step == 1
        return ReluBits(self.data, stop - start, self.bins, self.c, self.roi0 + start)

    def aligned(self):
        """the same RoIs as a ReluBits with RoI offset 0 (what a conv epilogue can read as mask_bits), or None when the offset is not
        a whole number of 128-row wave tiles"""
        m0 = self.roi0 * self.bins
        if m0 % 128:
            return None
        return self if m0 == 0 else ReluBits(self.data[(m0 // 128) * (self.c // 64) * 1024:], self.r, self.bins, self.c, 0)

    def unpack(self):
        """-> bool [r, bins, c] (tests)"""
        dev = self.data.device
        m = torch.arange(self.roi0 * self.bins, (self.roi0 + self.r) * self.bins, device=dev).view(-1, 1)
        n = torch.arange(self.c, device=dev).view(1, -1)
        word = ((m >> 7) * (self.c >> 6) + (n >> 6)) * 64 + ((m & 7) * 8 + ((n & 63) >> 3))
        byte = self.data[word * 16 + ((m & 127) >> 3)]
        return ((byte.int() >> (n & 7)) & 1).bool().view(self.r, self.bins, self.c)


def conv2d_ex(x, w, k, r, s, pad=0, bias=None, residual=None, relu=False, mask_bits=None, want_bits=False, pool_rows=0, want_y=True, x2=None, variant=0,
              pooled_out=None):
    """stride-1 bf16 conv on the 256x256 kernel with the extended epilogue (unit_conv2d_fwd_big_ex): returns (y | None, ReluBits |
    None, pooled [N, k] | None). pool_rows: must be OH*OW -- global average pool of each image (= RoI) fused; want_y=False then
    skips writing the map. mask_bits: ReluBits of an [N, OH*OW, k] map (RoI offset 0). x2 [N,H,W,C2]: second input of a 1x1 conv over
    the channel concatenation [x | x2] with w = [k][1][1][C + C2] (C2 a multiple of C)."""
    n, h, wd, c = x.shape
    c2 = 0
    if variant == 0 and BIG_TILE_VARIANT in (8, 11):
        variant = BIG_TILE_VARIANT
    if x2 is not None:
        c2 = x2.shape[3]
        assert r == 1 and s == 1 and pad == 0 and x2.shape[:3] == x.shape[:3] and c2 % c == 0 and x2.dtype == x.dtype and w.shape[-1] == c + c2
    oh, ow = conv_out_size(h, wd, r, s, 1, pad)
    ldy = k
    assert conv_ex_supported(x.dtype, c, ldy), "conv2d_ex: bf16, C % 64 == 0, K % 8 == 0"
    m = n * oh * ow
    y = torch.empty((n, oh, ow, ldy), dtype=x.dtype, device=x.device) if want_y else None
    bits = None
    if want_bits:
        bits = ReluBits(torch.empty(lib().unit_relu_bits_bytes(m, ldy), dtype=torch.uint8, device=x.device), n, oh * ow, k)
    if mask_bits is not None:
        assert mask_bits.roi0 == 0 and mask_bits.r * mask_bits.bins == m and mask_bits.c == k
    part = None
    if pool_rows:
        assert pool_rows == oh * ow
        part = torch.empty(lib().unit_conv_pool_partial_floats(m, ldy), dtype=torch.float32, device=x.device)
    prof = PROFILER
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib().unit_conv2d_fwd_big_ex(_p(x), _p(w), _p(y), _p(bias), _p(residual), _p(mask_bits.data if mask_bits is not None else None),
                                       _p(bits.data if bits is not None else None), _p(part), pool_rows,
                                       n, h, wd, c, k, r, s, pad, ldy, int(relu), _p(x2), c2, int(variant), _s()), "unit_conv2d_fwd_big_ex")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        nbytes = (x.numel() + w.numel() + (x2.numel() if x2 is not None else 0)) * 2 + m * ldy * 2 * (int(want_y) + (residual is not None)) \
            + (m * ldy // 8) * (int(want_bits) + (mask_bits is not None))
        prof.setdefault("conv_igemm256", []).append((e0, e1, 2.0 * m * k * r * s * (c + c2), nbytes))
    pooled = None
    if pool_rows:
        pooled = pooled_out if pooled_out is not None else torch.empty((n, k), dtype=x.dtype, device=x.device)
        assert pooled.shape == (n, k) and pooled.is_contiguous() and pooled.dtype == x.dtype
        check(lib().unit_pool_finish(_p(part), n, pool_rows, ldy, k, _p(pooled), k, dt(x.dtype), _s()), "unit_pool_finish")
    return y, bits, pooled


def avgpool_bwd_bits(dfeat, bits, ph, pw):
    """dfeat [R,C] bf16, bits: ReluBits of R RoIs -> g [R,PH,PW,C] = bit ? dfeat / (PH*PW) : 0"""
    assert bits.bins == ph * pw and bits.r == dfeat.shape[0] and bits.c == dfeat.shape[1]
    g = torch.empty((bits.r, ph, pw, bits.c), dtype=dfeat.dtype, device=dfeat.device)
    check(lib().unit_avgpool_bwd_bits(_p(dfeat), _p(bits.data), bits.r, bits.roi0, bits.bins, bits.c, _p(g), _s()), "avgpool_bwd_bits")
    return g


def conv2d_wgrad(x, dy, k, r, s, stride=1, pad=0, scale=None, out=None, accumulate=False, ldy=None, variant=0):
    """x [N,H,W,C], dy [N,OH,OW,ldy] -> dw fp32 [k,r,s,C] (scale[k] folded)."""
    if type(x) is X3:
        assert ldy is None
        return conv2d_wgrad_x3(x, dy, k, r, s, stride, pad, scale=scale, out=out, accumulate=accumulate, variant=variant)
    n, h, wd, c = x.shape
    oh, ow = conv_out_size(h, wd, r, s, stride, pad)
    ldy = ldy or dy.shape[-1]
    if out is None:
        out = torch.empty((k, r, s, c), dtype=torch.float32, device=x.device)
    nbytes = lib().unit_conv2d_wgrad_workspace_bytes(dt(x.dtype), n, oh, ow, k, r, s, c)
    ws = workspace(nbytes, x.device, slot=2)   # own slot: these launches may run on a side stream next to sort/NMS (slot 0)
    prof = PROFILER
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib().unit_conv2d_wgrad(_p(x), _p(dy), _p(out), _p(scale), dt(x.dtype), n, h, wd, c, k, r, s, stride, pad, oh, ow,
                                  ldy, int(accumulate), int(variant), _p(ws), ws.numel(), _s()), "unit_conv2d_wgrad")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        prof.setdefault("conv_wgrad", []).append((e0, e1, 2.0 * n * oh * ow * k * r * s * c,
                                                  (x.numel() + n * oh * ow * ldy) * x.element_size() + 4 * k * r * s * c))
    return out


def conv2d_wgrad_partial(x, dy, k, r, s, stride, pad, slab=None, variant=0):
    """split-M partial slabs only (no reduction): returns (slab uint8 tensor, n_splits); slab i = floats [i*k*r*s*C, ...).
    x / dy ops.Parts: the parts' slabs follow each other (n_splits = all of them)."""
    if isinstance(x, Parts):
        c = x[0].shape[-1]
        x3 = type(x[0]) is X3
        need, per = [], []
        for xp, dp in zip(x, dy):
            n, h, wd, _ = xp.shape
            oh, ow = conv_out_size(h, wd, r, s, stride, pad)
            mul = X3_WGRAD_PASSES if x3 else 1
            need.append(mul * lib().unit_conv2d_wgrad_workspace_bytes(BF16 if x3 else dt(xp.dtype), n, oh, ow, k, r, s, c))
            per.append(mul * lib().unit_conv2d_wgrad_splits(BF16 if x3 else dt(xp.dtype), n, oh, ow, k, r, s, c))
        nbytes = sum(need)
        if slab is None or slab.numel() < nbytes:
            old = slab
            if slab is not None:
                _retire(slab)
            slab = torch.empty(_grown(nbytes, old), dtype=torch.uint8, device=x[0].device)
        one = k * r * s * c * 4
        at = 0
        for xp, dp, sp in zip(x, dy, per):
            n, h, wd, _ = xp.shape
            oh, ow = conv_out_size(h, wd, r, s, stride, pad)
            view = slab[at * one:]
            with _timed("conv_wgrad", (X3_WGRAD_PASSES if x3 else 1) * 2.0 * n * oh * ow * k * r * s * c, (xp.numel() + dp.numel()) * xp.element_size() + 4 * k * r * s * c):
                if x3:
                    check(lib().unit_conv2d_wgrad_x3(_px(xp), _px(dp), None, None, n, h, wd, c, k, r, s, stride, pad, oh, ow, k, 0, _x3v(variant), _p(view),
                                                     view.numel(), _s()), "unit_conv2d_wgrad_x3(part)")
                else:
                    check(lib().unit_conv2d_wgrad(_p(xp), _p(dp), None, None, dt(xp.dtype), n, h, wd, c, k, r, s, stride, pad, oh, ow, dp.shape[-1], 0,
                                                  int(variant), _p(view), view.numel(), _s()), "unit_conv2d_wgrad(part)")
            at += sp
        return slab, at
    if type(x) is X3:
        return _wgrad_partial_x3(x, dy, k, r, s, stride, pad, slab, variant)
    n, h, wd, c = x.shape
    oh, ow = conv_out_size(h, wd, r, s, stride, pad)
    nbytes = lib().unit_conv2d_wgrad_workspace_bytes(dt(x.dtype), n, oh, ow, k, r, s, c)
    if slab is None or slab.numel() < nbytes:
        old = slab
        if slab is not None:
            _retire(slab)       # as workspace(): a captured step may replay into the old slab
        slab = torch.empty(_grown(nbytes, old), dtype=torch.uint8, device=x.device)
    splits = lib().unit_conv2d_wgrad_splits(dt(x.dtype), n, oh, ow, k, r, s, c)
    ldy = dy.shape[-1]
    prof = PROFILER
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib().unit_conv2d_wgrad(_p(x), _p(dy), None, None, dt(x.dtype), n, h, wd, c, k, r, s, stride, pad, oh, ow, dy.shape[-1], 0,
                                  int(variant), _p(slab), slab.numel(), _s()), "unit_conv2d_wgrad(partial)")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        prof.setdefault("conv_wgrad", []).append((e0, e1, 2.0 * n * oh * ow * k * r * s * c,
                                                  (x.numel() + n * oh * ow * ldy) * x.element_size() + 4 * k * r * s * c))
    return slab, splits



# ------------------------------------------------------------------------------------------------ ragged batches (two image groups)
class Ragged:
    """Activations of TWO image groups -- the supervised and the weak batch of a training step, each zero-padded to its own largest image
    (meta_arch/rcnn.py:438-452) -- as ONE tensor: the pixel rows of group 0 followed by those of group 1, `flat` [M0 + M1, C] (plain or X3).
    Pointwise stride-1 layers run over `flat` as one GEMM (rows are independent); every other layer runs both groups in one PAIR launch
    (unit_conv2d_fwd_pair) on the group views; weight gradients take the groups as parts of one layer. `group(i)` is a zero-copy
    [n, h, w, C] view."""

    def __init__(self, flat, dims):
        assert flat.dim() == 2 and len(dims) == 2 and sum(n * h * w for n, h, w in dims) == flat.shape[0], (flat.shape, dims)
        self.flat, self.dims = flat, [tuple(d) for d in dims]

    @property
    def dtype(self):
        return self.flat.dtype

    @property
    def device(self):
        return self.flat.device

    @property
    def channels(self):
        return self.flat.shape[1]

    def rows(self, i):
        lo = 0 if i == 0 else self.dims[0][0] * self.dims[0][1] * self.dims[0][2]
        n, h, w = self.dims[i]
        return lo, lo + n * h * w

    def group(self, i):
        lo, hi = self.rows(i)
        n, h, w = self.dims[i]
        return self.flat[lo:hi].view(n, h, w, self.flat.shape[1])

    def groups(self):
        return self.group(0), self.group(1)

    def as_gemm(self):
        """[1, 1, M0 + M1, C]: what a pointwise stride-1 layer sees"""
        return self.flat.view(1, 1, self.flat.shape[0], self.flat.shape[1])

    def like(self, flat):
        return Ragged(flat, self.dims)

    def record_stream(self, stream):
        self.flat.record_stream(stream)

    @staticmethod
    def empty(dims, c, like):
        """uninitialised Ragged of `c` channels with the dtype / device / X3-ness of the tensor `like`"""
        m = sum(n * h * w for n, h, w in dims)
        t = torch.empty((m, c), dtype=like.dtype, device=like.device)
        return Ragged(t.as_subclass(X3) if type(like) is X3 else t, dims)

    @staticmethod
    def zeros(dims, c, like):
        m = sum(n * h * w for n, h, w in dims)
        t = zeros((m, c), like.dtype, like.device)
        return Ragged(t.as_subclass(X3) if type(like) is X3 else t, dims)


class Parts(tuple):
    """the operand of a layer's weight gradient as several PARTS (the image groups of an ops.Ragged through a non-pointwise layer): x = Parts of
    [n, h, w, C] tensors, dy = Parts of the matching [n, oh, ow, K] tensors. The parts' split-M slabs follow each other in the layer's buffer and
    one reduction adds them -- a part is to the weight gradient what one more split-M range is."""

    def record_stream(self, stream):
        for t in self:
            t.record_stream(stream)


class ConvSecond(ctypes.Structure):
    """include/unit_hip.h: UnitConvSecond"""
    _fields_ = [("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("mask_ref", ctypes.c_void_p)] + \
               [(f, ctypes.c_int) for f in ("N", "H", "W", "OHf", "OWf")]


def conv2d_pair(xs, w, k, r, s, stride=1, pad=0, bias=None, residuals=None, mask_refs=None, relu=False, outs=None, scatters=None, force=None):
    """the same conv layer over TWO inputs of different map sizes in ONE launch (unit_conv2d_fwd_pair; csrc/conv_epilogue.h ConvSecond):
    xs = (x0, x1) NHWC of one dtype / channel count (plain or X3), residuals / mask_refs / outs / scatters pairs (or None). The kernel and
    tile are chosen for the SUM of the two problems' pixels (force = (kernel, tile) of unit_conv2d_fwd_pair overrides: tests). Returns (y0, y1);
    each equals the single launch's result bit for bit."""
    x0, x1 = xs
    x3 = type(x0) is X3
    assert (type(x1) is X3) == x3 and x0.dtype == x1.dtype and x0.shape[3] == x1.shape[3]
    c = x0.shape[3]
    residuals = residuals or (None, None)
    mask_refs = mask_refs or (None, None)
    scatters = scatters or (None, None)
    geo = []
    for x, sc in zip(xs, scatters):
        n, h, wd, _ = x.shape
        oh, ow = conv_out_size(h, wd, r, s, stride, pad)
        oy_mul, ohf, owf = (1, oh, ow) if sc is None else sc
        geo.append((n, h, wd, oh, ow, oy_mul, ohf, owf))
    assert geo[0][5] == geo[1][5], "conv2d_pair: one scatter multiplier for both problems"
    ldy = k if x3 else (k + 3) // 4 * 4
    if outs is None:
        mk = (lambda shape: zeros(shape, x0.dtype, x0.device)) if scatters[0] is not None else (lambda shape: torch.empty(shape, dtype=x0.dtype, device=x0.device))
        outs = tuple(mk((g[0], g[6], g[7], ldy)) for g in geo)
        if x3:
            outs = tuple(o.as_subclass(X3) for o in outs)
    ptr = (lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())) if x3 else _p
    if x3:
        for t in tuple(xs) + tuple(outs) + tuple(q for q in residuals + mask_refs if q is not None):
            assert type(t) is X3 and t.is_contiguous()
    m_tot = sum(g[0] * g[3] * g[4] for g in geo)
    mask_c = mask_refs[0].shape[-1] if (x3 and mask_refs[0] is not None) else 0
    if force is not None:
        kernel, tile = force
    elif x3:
        segs = w.shape[-1] // c          # (conv2d_x3)
        assert segs in (2, 3) and w.shape[-1] == segs * c
        kernel, tile = (3 if segs == 3 else 4), X3_TILE_POLICY(m_tot, k, c, segs * r * s * c)
    else:
        big = BIG_TILE_POLICY(x0.dtype, m_tot, k, c, r * s * c) and ldy % 8 == 0
        mid = -1
        if not big:
            lc_ok = x0.dtype == torch.bfloat16 and ldy % 8 == 0
            mid = MID_TILE_POLICY(x0.dtype, m_tot, k, c, r * s * c, allow_lc=lc_ok)
            if 2000 <= mid < 8000 or mid == 3:          # forms without a pair instantiation: the plain 4-wave tile instead
                tiles = ((m_tot + 127) // 128) * ((k + 127) // 128)
                mid = 2 if k <= 64 else (0 if tiles >= 256 else 1)
        kernel, tile = (2, 0) if big else ((1, mid) if mid >= 0 else (0, 0))
    g0, g1 = geo
    sec = ConvSecond()
    sec.x, sec.y = xs[1].data_ptr(), outs[1].data_ptr()
    sec.residual = residuals[1].data_ptr() if residuals[1] is not None else None
    sec.mask_ref = mask_refs[1].data_ptr() if mask_refs[1] is not None else None
    sec.N, sec.H, sec.W, sec.OHf, sec.OWf = g1[0], g1[1], g1[2], g1[6], g1[7]
    with _timed("conv_igemm256" if kernel == 2 or (kernel in (3, 4) and tile < 0) else ("conv_igemm_dma" if kernel in (1, 3, 4) else "conv_igemm"),
                ((3 if kernel == 3 else 2) if x3 else 1) * 2.0 * m_tot * k * r * s * c, (sum(x.numel() for x in xs) + m_tot * ldy) * x0.element_size()):
        check(lib().unit_conv2d_fwd_pair(kernel, ptr(xs[0]), _p(w), ptr(outs[0]), _p(bias), ptr(residuals[0]), ptr(mask_refs[0]), mask_c,
                                         BF16 if x3 else dt(x0.dtype), BF16 if x3 else dt(x0.dtype), g0[0], g0[1], g0[2], c, k, r, s, stride, pad, g0[3], g0[4],
                                         ldy, g0[5], g0[6], g0[7], int(relu), int(tile), ctypes.byref(sec), _s()), "unit_conv2d_fwd_pair")
    return outs


# ------------------------------------------------------------------------------------------------ bf16x3 (split) convolutions
def x3_tile_policy(m, k, c, kgemm_v):
    """kernel / tile of a bf16x3 conv with m output pixels, k filters, c real input channels, kgemm_v = 3 * r * s * c virtual k extent:
    -1 = the 256x256 phase-interleaved kernel, 0 / 1 / 2 = 4-wave tiles, >= 100 = loader / consumer tile code (the plain bf16 policies
    applied to the three times longer contraction)"""
    # (at least 200 tiles: the res4 256 -> 1024 layers -- 152 tiles on 256 CUs, 36.7 us -- run faster on the loader / consumer tiles: 36.2;
    #  tools/x3_conv_sweep.py)
    if k >= 256 and kgemm_v >= 512 and ((m + 255) // 256) * ((k + 255) // 256) >= 200:
        return -1
    if k <= 64:
        return 2
    tiles = ((m + 127) // 128) * ((k + 127) // 128)
    if not _NO_LC and kgemm_v >= 512 and tiles <= 640 and k % 8 == 0:
        return lc_tile_code(m, k, kgemm_v)
    return 0 if tiles >= 256 else 1


X3_TILE_POLICY = x3_tile_policy


def conv2d_x3(x, w, k, r, s, stride=1, pad=0, bias=None, residual=None, mask_ref=None, relu=False, out=None, scatter=None, tile=None):
    """bf16x3 convolution (unit_conv2d_fwd_x3): x X3 [N,H,W,C], w = weight_prep_x3's forward (or dgrad) copy, residual / mask_ref X3;
    returns an X3 [N,OH,OW,k] (or writes the strided scatter target)."""
    n, h, wd, c = x.shape
    assert x.is_contiguous() and c % 64 == 0 and k % 8 == 0, "conv2d_x3: C % 64 == 0, K % 8 == 0"
    oh, ow = conv_out_size(h, wd, r, s, stride, pad)
    if scatter is None:
        oy_mul, ohf, owf = 1, oh, ow
    else:
        oy_mul, ohf, owf = scatter
    if out is None:
        out = (zeros((n, ohf, owf, k), torch.float32, x.device) if scatter is not None
               else torch.empty((n, ohf, owf, k), dtype=torch.float32, device=x.device)).as_subclass(X3)
    assert type(out) is X3 and out.is_contiguous()
    if residual is not None:
        assert type(residual) is X3 and residual.is_contiguous() and residual.shape[-1] == k
    mask_c = 0
    if mask_ref is not None:
        assert type(mask_ref) is X3 and mask_ref.is_contiguous()
        mask_c = mask_ref.shape[-1]
    m = n * oh * ow
    segs = w.shape[-1] // c          # 3 = [Wh | Wh | Wl]; 2 = [Wh | Wl], the two-segment dgrad copy (weight_prep_x3 dgrad_segs)
    assert segs in (2, 3) and w.shape[-1] == segs * c, "conv2d_x3: w is a weight_prep_x3 copy"
    if tile is None:
        pkey = ("x3", m, k, c, r * s, X3_TILE_POLICY, _NO_LC, segs)
        tile = _POLICY_CACHE.get(pkey)
        if tile is None:
            tile = _POLICY_CACHE[pkey] = X3_TILE_POLICY(m, k, c, segs * r * s * c)
    prof = PROFILER
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib().unit_conv2d_fwd_x3s(_px(x), _p(w), _px(out), _p(bias), _px(residual), _px(mask_ref), mask_c, n, h, wd, c, k, r, s, stride, pad,
                                    oh, ow, k, oy_mul, ohf, owf, int(relu), int(tile), segs, _s()), "unit_conv2d_fwd_x3s")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        nbytes = (x.numel() + m * k * (1 + (residual is not None))) * 4 + (m * k * 2 if mask_ref is not None else 0) + w.numel() * 2
        feed = None
        if tile >= 0:
            bm, bn = mid_tile_dims(tile)
            tiles = ((m + bm - 1) // bm) * ((k + bn - 1) // bn)
            feed = ((tiles + 255) // 256) * (segs * r * s * c // 64) * (bm + bn) * 128.0
        # flops = the MFMA work issued: three (dgrad with two segments: two) bf16 products per fp32 product
        prof.setdefault("conv_igemm256" if tile < 0 else "conv_igemm_dma", []).append((e0, e1, segs * 2.0 * m * k * r * s * c, nbytes, feed))
    return out


# k-segments of the DGRAD copies of the bf16x3 mode (round 6): 2 = [Wh | Wl] against the gradient map's hi plane (dx = hi(dy).(Wh + Wl)),
# 3 = the forward's three products. UNIT_X3_DGRAD_SEGS=3 restores round 5 (A/B; profiles/r06_exp_x3_wgrad_passes.txt)
X3_DGRAD_SEGS = int(os.environ.get("UNIT_X3_DGRAD_SEGS", "2"))
assert X3_DGRAD_SEGS in (2, 3)


def weight_prep_x3(w_krsc, scale, k, r, s, c, want_fwd=True, want_dgrad=True, w_fwd=None, w_dgrad=None):
    """fp32 [k][r][s][c] storage -> (bf16 [k][r][s][c/64][3][64], bf16 [c][r][s][k/64][3][64]) = the k-segments [Wh | Wh | Wl] of the bf16x3
    convs (forward: c % 64 == 0; dgrad: k % 64 == 0; either may be skipped)"""
    dev = w_krsc.device
    if want_fwd and w_fwd is None:
        w_fwd = torch.empty((k, r, s, 3 * c), dtype=torch.bfloat16, device=dev)
    dsegs = X3_DGRAD_SEGS
    if want_dgrad and (w_dgrad is None or w_dgrad.shape[-1] != dsegs * k):
        w_dgrad = torch.empty((c, r, s, dsegs * k), dtype=torch.bfloat16, device=dev)
    check(lib().unit_weight_prep_x3s(_p(w_krsc), _p(scale), k, r, s, c, _p(w_fwd) if want_fwd else None, _p(w_dgrad) if want_dgrad else None,
                                     dsegs, _s()), "weight_prep_x3s")
    return (w_fwd if want_fwd else None), (w_dgrad if want_dgrad else None)


def _x3v(variant):
    """`variant` of unit_conv2d_wgrad_x3 with the pass count in bits 8-9 (0 = all three)"""
    return int(variant) | ((X3_WGRAD_PASSES if X3_WGRAD_PASSES != 3 else 0) << 8)


def conv2d_wgrad_x3(x, dy, k, r, s, stride=1, pad=0, scale=None, out=None, accumulate=False, variant=0, slab=None):
    """bf16x3 weight gradient: x X3 [N,H,W,C], dy X3 [N,OH,OW,k] -> dw fp32 [k,r,s,C] (scale[k] folded); out=None and slab given (or
    partial=True via conv2d_wgrad_partial): leaves the 3 * splits slabs"""
    n, h, wd, c = x.shape
    oh, ow = conv_out_size(h, wd, r, s, stride, pad)
    assert type(x) is X3 and type(dy) is X3 and x.is_contiguous() and dy.is_contiguous() and dy.shape[-1] == k
    if out is None:
        out = torch.empty((k, r, s, c), dtype=torch.float32, device=x.device)
    nbytes = X3_WGRAD_PASSES * lib().unit_conv2d_wgrad_workspace_bytes(BF16, n, oh, ow, k, r, s, c)
    ws = workspace(nbytes, x.device, slot=2)
    with _timed("conv_wgrad", X3_WGRAD_PASSES * 2.0 * n * oh * ow * k * r * s * c, (x.numel() + dy.numel()) * 4 + 4 * k * r * s * c):
        check(lib().unit_conv2d_wgrad_x3(_px(x), _px(dy), _p(out), _p(scale), n, h, wd, c, k, r, s, stride, pad, oh, ow, k, int(accumulate),
                                         _x3v(variant), _p(ws), ws.numel(), _s()), "unit_conv2d_wgrad_x3")
    return out


def _wgrad_partial_x3(x, dy, k, r, s, stride, pad, slab, variant):
    n, h, wd, c = x.shape
    oh, ow = conv_out_size(h, wd, r, s, stride, pad)
    assert type(dy) is X3 and x.is_contiguous() and dy.is_contiguous() and dy.shape[-1] == k
    nbytes = X3_WGRAD_PASSES * lib().unit_conv2d_wgrad_workspace_bytes(BF16, n, oh, ow, k, r, s, c)
    if slab is None or slab.numel() < nbytes:
        old = slab
        if slab is not None:
            _retire(slab)
        slab = torch.empty(_grown(nbytes, old), dtype=torch.uint8, device=x.device)
    splits = X3_WGRAD_PASSES * lib().unit_conv2d_wgrad_splits(BF16, n, oh, ow, k, r, s, c)
    with _timed("conv_wgrad", X3_WGRAD_PASSES * 2.0 * n * oh * ow * k * r * s * c, (x.numel() + dy.numel()) * 4 + 4 * k * r * s * c):
        check(lib().unit_conv2d_wgrad_x3(_px(x), _px(dy), None, None, n, h, wd, c, k, r, s, stride, pad, oh, ow, k, 0, _x3v(variant), _p(slab),
                                         slab.numel(), _s()), "unit_conv2d_wgrad_x3(partial)")
    return slab, splits


class WgradProblem(ctypes.Structure):
    """include/unit_hip.h: UnitWgradProblem"""
    _fields_ = [("x", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("partial", ctypes.c_void_p)] + \
               [(f, ctypes.c_int) for f in ("N", "H", "W", "C", "K", "R", "S", "stride", "pad", "OH", "OW", "ldy", "splits", "kind",
                                            "x_pitch", "x_back", "dy_back")]


def wgrad_group_supported(x, dy, k, r, s, stride, pad):
    """may this layer's weight gradient go into a grouped launch (csrc/conv_wgrad128r.hip)? bf16 tensors, or X3 split tensors (each of
    the three plane passes of a bf16x3 weight gradient is a bf16 problem of its own in the grid); ops.Parts: every part"""
    if isinstance(x, Parts):
        return all(wgrad_group_supported(xp, dp, k, r, s, stride, pad) for xp, dp in zip(x, dy))
    n, h, wd, c = x.shape
    oh, ow = conv_out_size(h, wd, r, s, stride, pad)
    if type(x) is X3:
        return type(dy) is X3 and bool(lib().unit_conv2d_wgrad_group_supported(BF16, n, oh, ow, k, r, s, c))
    return x.dtype == torch.bfloat16 and bool(lib().unit_conv2d_wgrad_group_supported(dt(x.dtype), n, oh, ow, k, r, s, c))


_X3_PASSES = ((0, 0), (0, 1), (1, 0))          # (plane of x, plane of dy) per pass: hi^T.hi + hi^T.lo + lo^T.hi
# How many of those passes a bf16x3 WEIGHT gradient runs (round 6). 3 = fp32-grade products (round 5). 1 = the hi planes only: every
# product is a bf16 x bf16 product of the ROUNDED operands, accumulated in fp32 over the >= 2 394 pixel rows of the contraction -- the
# forward and the dgrad chain (what the losses, the index decisions and every upstream gradient depend on) keep all three products. The
# parity tests hold weight gradients to 2e-3 of a tensor's largest entry, and what moves them at fp32-grade precision already is ReLU masks
# flipping within rounding of zero (1.6e-3, profiles/r05_fullsize_parity_metrics.json); a 2^-9 relative rounding of the operands averages
# out over the contraction. UNIT_X3_WGRAD_PASSES=3 restores the three passes (A/B; profiles/r06_exp_x3_wgrad_passes.txt).
X3_WGRAD_PASSES = int(os.environ.get("UNIT_X3_WGRAD_PASSES", "1"))
assert X3_WGRAD_PASSES in (1, 3)


def conv2d_wgrad_group(items, slabs=None, splits_hint=0):
    """split-M partial slabs of SEVERAL layers from one launch (unit_conv2d_wgrad_group). items: [(x, dy, k, r, s, stride, pad)];
    slabs: per item a uint8 tensor to reuse or None. Returns [(slab, n_splits)]; slab i of a layer = floats [i*k*r*s*C, ...) as
    conv2d_wgrad_partial leaves them. An item is one or more PROBLEMS of the grid whose slabs follow each other in the layer's buffer
    (n_splits = all of them): one per part of an ops.Parts operand (the image groups of a ragged batch), times one per plane pass for X3
    (bf16x3 split) tensors."""
    n_items = len(items)
    if n_items == 0:
        return []
    assert ctypes.sizeof(WgradProblem) == lib().unit_wgrad_problem_bytes()
    probs = []          # (item index, x part, dy part, pass | None)
    for i, it in enumerate(items):
        parts = list(zip(it[0], it[1])) if isinstance(it[0], Parts) else [(it[0], it[1])]
        for xp, dp in parts:
            if type(xp) is X3:
                assert type(dp) is X3 and dp.shape[-1] == it[2]
                probs += [(i, xp, dp, ps) for ps in range(X3_WGRAD_PASSES)]
            else:
                probs.append((i, xp, dp, None))
    pr = (WgradProblem * len(probs))()
    flops = nbytes = 0
    for j, (i, x, dy, ps) in enumerate(probs):
        _, _, k, r, s, stride, pad = items[i]
        n, h, wd, c = x.shape
        oh, ow = conv_out_size(h, wd, r, s, stride, pad)
        q = pr[j]
        q.N, q.H, q.W, q.C, q.K, q.R, q.S, q.stride, q.pad, q.OH, q.OW = n, h, wd, c, k, r, s, stride, pad, oh, ow
        if ps is None:
            q.x, q.dy, q.ldy = x.data_ptr(), dy.data_ptr(), dy.shape[-1]
            q.x_pitch = q.x_back = q.dy_back = 0
            nbytes += (x.numel() + n * oh * ow * dy.shape[-1]) * 2 + 4 * k * r * s * c
        else:
            px, pd = _X3_PASSES[ps]
            q.x, q.dy, q.ldy = x.data_ptr() + px * c * 2, dy.data_ptr() + pd * k * 2, 2 * k
            q.x_pitch, q.x_back, q.dy_back = 2 * c, px * c, pd * k
            nbytes += (x.numel() + n * oh * ow * k) * 2 + 4 * k * r * s * c
        flops += 2.0 * n * oh * ow * k * r * s * c
    check(lib().unit_conv2d_wgrad_group_plan(pr, len(probs), int(splits_hint)), "unit_conv2d_wgrad_group_plan")
    total = [0] * n_items
    for j, (i, _, _, _) in enumerate(probs):
        total[i] += pr[j].splits
    out = []
    for i, it in enumerate(items):
        k, r, s = it[2], it[3], it[4]
        x0 = it[0][0] if isinstance(it[0], Parts) else it[0]
        one = k * r * s * x0.shape[-1] * 4
        need = total[i] * one
        slab = slabs[i] if slabs is not None else None
        if slab is None or slab.numel() < need:
            old = slab
            if slab is not None:
                _retire(slab)       # (conv2d_wgrad_partial)
            slab = torch.empty(_grown(need, old), dtype=torch.uint8, device=x0.device)
        out.append((slab, total[i]))
    at = [0] * n_items
    for j, (i, x, _, _) in enumerate(probs):
        k, r, s = items[i][2], items[i][3], items[i][4]
        pr[j].partial = out[i][0].data_ptr() + at[i] * k * r * s * x.shape[-1] * 4
        at[i] += pr[j].splits
    with _timed("conv_wgrad", flops, nbytes):
        check(lib().unit_conv2d_wgrad_group(pr, len(probs), dt(torch.bfloat16), _s()), "unit_conv2d_wgrad_group")
    return out


def frozen_bn_fold(weight, bias, running_mean, running_var, eps=1e-5):
    c = weight.numel()
    scale = torch.empty(c, dtype=torch.float32, device=weight.device)
    shift = torch.empty(c, dtype=torch.float32, device=weight.device)
    check(lib().unit_frozen_bn_fold(_p(weight), _p(bias), _p(running_mean), _p(running_var), eps, _p(scale), _p(shift), c, _s()),
          "frozen_bn_fold")
    return scale, shift


def weight_prep(w_krsc, scale, k, r, s, c, cp, dtype, want_dgrad=True, w_fwd=None, w_dgrad=None):
    """w_krsc: fp32 storage in [k][r][s][c] order (channels_last view of the [k,c,r,s] parameter)."""
    dev = w_krsc.device
    if w_fwd is None:
        w_fwd = torch.empty((k, r, s, cp), dtype=dtype, device=dev)
    if want_dgrad and w_dgrad is None:
        w_dgrad = torch.empty((c, r, s, k), dtype=dtype, device=dev)
    check(lib().unit_weight_prep(_p(w_krsc), _p(scale), k, r, s, c, cp, _p(w_fwd), _p(w_dgrad) if want_dgrad else None,
                                 dt(dtype), _s()), "weight_prep")
    return w_fwd, w_dgrad


def bias_grad(dy2d, k, out=None, accumulate=False):
    m, ld = dy2d.shape
    if out is None:
        out = torch.empty(k, dtype=torch.float32, device=dy2d.device)
    nb = lib().unit_bias_grad_scratch_bytes(m, k) if m > 1024 else 0
    ws = workspace(nb, dy2d.device, slot=3) if nb else None
    check(lib().unit_bias_grad(_p(dy2d), dt(dy2d.dtype), m, k, ld, _p(out), int(accumulate), _p(ws), nb, _s()), "bias_grad")
    return out


def linear_wgrad(x2d, dy2d, k, dw, db):
    """bf16 x [R,C], dy [R,ldy] (pad columns zero) -> dw fp32 [k,C], db fp32 [k] from one launch (csrc/linear_wgrad.hip)."""
    r, c = x2d.shape
    ldy = dy2d.shape[1]
    dev = x2d.device
    nb = lib().unit_linear_wgrad_workspace_bytes(r, c, k)
    ws = workspace(nb, dev, slot=4)
    with _timed("conv_wgrad", 2.0 * r * k * c, (x2d.numel() + r * ldy) * 2 + 4 * k * c):
        check(lib().unit_linear_wgrad(_p(x2d), _p(dy2d), dt(x2d.dtype), r, c, k, ldy, _p(dw), _p(db), _p(ws), ws.numel(), _s()),
              "unit_linear_wgrad")


def stem_conv_pool(x, w_fwd, shift, out=None):
    """bf16 x [N,H,W,8] -> [N,PH,PW,64]: 7x7 s2 conv + folded FrozenBN + ReLU + 3x3 s2 max pool in one launch (csrc/stem_pool.hip)"""
    n, h, w, c = x.shape
    assert c == 8 and tuple(w_fwd.shape) == (64, 7, 7, 8) and x.dtype == torch.bfloat16 and w_fwd.dtype == torch.bfloat16
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    ph, pw = (oh - 1) // 2 + 1, (ow - 1) // 2 + 1
    y = out if out is not None else torch.empty((n, ph, pw, 64), dtype=x.dtype, device=x.device)
    assert tuple(y.shape) == (n, ph, pw, 64) and y.is_contiguous()
    with _timed("stem", 2.0 * n * oh * ow * 64 * 147, x.numel() * 2 + y.numel() * 2):
        check(lib().unit_stem_conv_pool(_p(x), _p(w_fwd), _p(shift), _p(y), dt(x.dtype), n, h, w, _s()), "unit_stem_conv_pool")
    return y


def maxpool3x3s2(x, out=None):
    n, h, w, c = x.shape
    oh, ow = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    y = out if out is not None else torch.empty((n, oh, ow, c), dtype=x.dtype, device=x.device)
    assert tuple(y.shape) == (n, oh, ow, c) and y.is_contiguous() and y.dtype == x.dtype
    check(lib().unit_maxpool3x3s2_fwd(_p(x), _p(y), dt(x.dtype), n, h, w, c, _s()), "maxpool")
    return y


def global_avgpool(x, out=None):
    """[R,PH,PW,C] -> [R,C] (an X3 map: fp32 features, unit_global_avgpool_x3_fwd); out: contiguous rows of a larger feature matrix"""
    r, ph, pw, c = x.shape
    if out is not None:
        assert out.shape == (r, c) and out.is_contiguous() and out.dtype == (torch.float32 if type(x) is X3 else x.dtype)
    if type(x) is X3:
        y = out if out is not None else torch.empty((r, c), dtype=torch.float32, device=x.device)
        check(lib().unit_global_avgpool_x3_fwd(_px(x), _p(y), r, ph * pw, c, _s()), "avgpool_x3_fwd")
        return y
    y = out if out is not None else torch.empty((r, c), dtype=x.dtype, device=x.device)
    check(lib().unit_global_avgpool_fwd(_p(x), _p(y), dt(x.dtype), r, ph * pw, c, _s()), "avgpool_fwd")
    return y


def global_avgpool_bwd_relu(dfeat, out):
    r, ph, pw, c = out.shape
    if type(out) is X3:          # -> X3 gradient map
        g = torch.empty(out.shape, dtype=torch.float32, device=out.device).as_subclass(X3)
        check(lib().unit_global_avgpool_x3_bwd_relu(_p(dfeat), _px(out), _px(g), r, ph * pw, c, _s()), "avgpool_x3_bwd")
        return g
    g = torch.empty_like(out)
    check(lib().unit_global_avgpool_bwd_relu(_p(dfeat), _p(out), _p(g), dt(out.dtype), r, ph * pw, c, _s()), "avgpool_bwd")
    return g


# ------------------------------------------------------------------------------------------------ boxes
def cell_anchors(sizes=(32, 64, 128, 256, 512), ratios=(0.5, 1.0, 2.0)):
    """DefaultAnchorGenerator.generate_cell_anchors: python double math, then fp32."""
    out = []
    for size in sizes:
        area = size ** 2.0
        for r in ratios:
            w = math.sqrt(area / r)
            h = r * w
            out.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(out, dtype=torch.float32)


def anchor_grid(h, w, cell, stride=16, offset=0.0):
    a = cell.shape[0]
    out = torch.empty((h * w * a, 4), dtype=torch.float32, device=cell.device)
    check(lib().unit_anchor_grid(_p(out), h, w, a, float(stride), float(offset), _p(cell), _s()), "anchor_grid")
    return out


def iou_match(gt, gt_count, boxes, box_count, thresholds, labels, allow_low_quality, want_vals=True):
    """gt [B,Mcap,4]; boxes [Ncap,4] (shared) or [B,Ncap,4]; counts: device int32 [B] or None."""
    b, mcap = gt.shape[0], gt.shape[1]
    shared = boxes.dim() == 2
    ncap = boxes.shape[-2]
    dev = gt.device
    idx = torch.empty((b, ncap), dtype=torch.int64, device=dev)
    lab = torch.empty((b, ncap), dtype=torch.int8, device=dev)
    val = torch.empty((b, ncap), dtype=torch.float32, device=dev) if want_vals else None
    th = (ctypes.c_float * len(thresholds))(*thresholds)
    lb = (ctypes.c_int * len(labels))(*labels)
    nb = lib().unit_iou_match_workspace_bytes(b, mcap)
    ws = workspace(nb, dev)
    check(lib().unit_iou_match(_p(gt), _p(gt_count), b, mcap, _p(boxes), 0 if shared else ncap * 4, _p(box_count), ncap,
                               th, lb, len(thresholds), int(allow_low_quality), _p(idx), _p(lab), _p(val), _p(ws), ws.numel(),
                               _s()), "iou_match")
    return idx, lab, val


def pairwise_iou(b1, b2):
    out = torch.empty((b1.shape[0], b2.shape[0]), dtype=torch.float32, device=b1.device)
    check(lib().unit_pairwise_iou(_p(b1), b1.shape[0], _p(b2), b2.shape[0], _p(out), _s()), "pairwise_iou")
    return out


def match_matrix(q, thresholds, labels, allow_low_quality):
    """Matcher.__call__ on an [M,N] quality matrix (modeling/matcher.py:54-98) -> (idx int64, label int8, val fp32)"""
    m, n = q.shape
    dev = q.device
    idx = torch.empty((n,), dtype=torch.int64, device=dev)
    lab = torch.empty((n,), dtype=torch.int8, device=dev)
    val = torch.empty((n,), dtype=torch.float32, device=dev)
    ws = torch.empty((max(m, 1),), dtype=torch.float32, device=dev)
    th = (ctypes.c_float * len(thresholds))(*thresholds)
    lb = (ctypes.c_int * len(labels))(*labels)
    check(lib().unit_match_matrix(_p(q.contiguous()), m, n, th, lb, len(thresholds), int(allow_low_quality), _p(idx), _p(lab), _p(val),
                                  _p(ws), _s()), "match_matrix")
    return idx, lab, val


def subsample_labels(labels, count, perm, num_samples, positive_fraction, bg_label, want_labels=True, want_idx=True):
    """labels [B,Ncap] int8|int64 ; perm [B,Pcap] int32 -> (out_labels int8 [B,Ncap], sampled_idx int32 [B,S], counts [B,2])"""
    b, ncap = labels.shape
    dev = labels.device
    out_labels = torch.empty((b, ncap), dtype=torch.int8, device=dev) if want_labels else None
    sidx = torch.empty((b, num_samples), dtype=torch.int32, device=dev) if want_idx else None
    counts = torch.empty((b, 2), dtype=torch.int32, device=dev)
    max_pos = int(num_samples * positive_fraction)
    check(lib().unit_subsample_labels(_p(labels), int(labels.dtype == torch.int64), _p(count), b, ncap, _p(perm), perm.shape[1],
                                      num_samples, max_pos, bg_label, _p(out_labels), _p(sidx), _p(counts), _s()),
          "subsample_labels")
    return out_labels, sidx, counts


def random_permutations(b, n, seed, counter, stream_id, device):
    """-> int32 [b, n]: one uniformly random permutation of range(n) per row, drawn from (seed, *counter, stream_id) by
    unit_perm_keys + the stable descending sort (replaces torch.randperm in the step; `counter` is a device int64 scalar)"""
    keys = torch.empty((b, n), dtype=torch.float32, device=device)
    check(lib().unit_perm_keys(int(seed) & 0xFFFFFFFFFFFFFFFF, _p(counter), int(stream_id), b, n, _p(keys), _s()), "perm_keys")
    _, idx = sort_desc(keys, b, n, topk=n)        # chip-wide select + rank form (the plain form is one workgroup per row)
    return idx


def counter_bump(counter, delta=1):
    check(lib().unit_counter_bump(_p(counter), int(delta), _s()), "counter_bump")


def box_encode(src, tgt, weights):
    out = torch.empty_like(src)
    w = (ctypes.c_float * 4)(*weights)
    check(lib().unit_box_encode(_p(src), _p(tgt), w, _p(out), src.shape[0], _s()), "box_encode")
    return out


def box_decode(deltas, boxes, weights, k=None, col0=0):
    n, ld = deltas.shape
    k = k or (ld - col0) // 4
    out = torch.empty((n, k * 4), dtype=torch.float32, device=deltas.device)
    w = (ctypes.c_float * 4)(*weights)
    check(lib().unit_box_decode(_p(deltas), ld, col0, k, _p(boxes), w, SCALE_CLAMP, _p(out), n, _s()), "box_decode")
    return out


def sort_desc(src, b, n, ld=1, a=1, col0=0, batch_stride=None, topk=None, min_exclusive=None):
    """stable descending sort of n keys per batch row; keys read as src[b*bstride + (i//a)*ld + col0 + i%a].
    topk: only the first min(topk, n) entries of each output row are needed (chip-wide select + rank sort);
    min_exclusive (with topk): keys <= it are not ranked at all (rows are defined up to the number of larger keys)."""
    dev = src.device
    keys = torch.empty((b, n), dtype=torch.float32, device=dev)
    idx = torch.empty((b, n), dtype=torch.int32, device=dev)
    if batch_stride is None:
        batch_stride = (n // a) * ld
    nb = lib().unit_sort_workspace_bytes(b, n)
    ws = workspace(nb, dev)
    if topk is not None:
        with _timed("sort_topk", 0.0, b * n * 4.0):
            check(lib().unit_sort_desc_stable_topk(_p(src), batch_stride, ld, a, col0, b, n, int(topk),
                                                   float("-inf") if min_exclusive is None else float(min_exclusive), _p(keys), _p(idx),
                                                   _p(ws), ws.numel(), _s()), "sort_desc_stable_topk")
        return keys, idx
    check(lib().unit_sort_desc_stable(_p(src), batch_stride, ld, a, col0, b, n, _p(keys), _p(idx), _p(ws), ws.numel(), _s()),
          "sort_desc_stable")
    return keys, idx


def rpn_decode_select(head, a, delta_col0, anchors, sorted_idx, sorted_logit, topk, image_hw, min_size=0.0):
    """head [B,HW,ld] fp32 -> (cand_boxes [B,topk,4], cand_scores [B,topk], cand_count [B])"""
    b, hw, ld = head.shape
    ncap = anchors.shape[0]
    dev = head.device
    cb = torch.empty((b, topk, 4), dtype=torch.float32, device=dev)
    cs = torch.empty((b, topk), dtype=torch.float32, device=dev)
    cc = torch.empty((b,), dtype=torch.int32, device=dev)
    check(lib().unit_rpn_decode_select(_p(head), hw * ld, ld, a, delta_col0, _p(anchors), _p(sorted_idx), _p(sorted_logit), b, ncap,
                                       topk, _p(image_hw), SCALE_CLAMP, float(min_size), _p(cb), _p(cs), _p(cc), _s()),
          "rpn_decode_select")
    return cb, cs, cc


def nms(boxes_sorted, scores_sorted, count, thresh, max_keep, out=None):
    """boxes [B,cap,4] in descending score order -> (keep_idx [B,max_keep], keep_count [B], out_boxes, out_scores).
    out = (boxes [B,max_keep,4], scores [B,max_keep], keep_count int32 [B]): rows of larger buffers to write into (a two-pass step keeps the
    proposals of both passes in one tensor)"""
    b, cap = boxes_sorted.shape[0], boxes_sorted.shape[1]
    dev = boxes_sorted.device
    keep = torch.empty((b, max_keep), dtype=torch.int32, device=dev)      # unit_nms writes the tails (-1 / 0) itself
    if out is not None:
        ob, osc, kc = out
        assert ob.shape == (b, max_keep, 4) and osc.shape == (b, max_keep) and kc.shape == (b,) and ob.is_contiguous() and osc.is_contiguous()
    else:
        kc = torch.empty((b,), dtype=torch.int32, device=dev)
        ob = torch.empty((b, max_keep, 4), dtype=torch.float32, device=dev)
        osc = torch.empty((b, max_keep), dtype=torch.float32, device=dev)
    nb = lib().unit_nms_workspace_bytes(b, cap)
    ws = workspace(nb, dev)
    with _timed("nms", 0.0, b * cap * 20.0):
        check(lib().unit_nms(_p(boxes_sorted), _p(scores_sorted), _p(count), b, cap, float(thresh), max_keep, _p(keep), _p(kc), _p(ob),
                             _p(osc), _p(ws), ws.numel(), _s()), "nms")
    return keep, kc, ob, osc


# ------------------------------------------------------------------------------------------------ a7 plumbing
def append_gt(props, pcount, gt, gcount):
    b, pcap = props.shape[0], props.shape[1]
    mcap = gt.shape[1]
    cat = torch.empty((b, pcap + mcap, 4), dtype=torch.float32, device=props.device)
    cc = torch.empty((b,), dtype=torch.int32, device=props.device)
    check(lib().unit_append_gt(_p(props), _p(pcount), pcap, _p(gt), _p(gcount), mcap, b, _p(cat), _p(cc), _s()), "append_gt")
    return cat, cc


def roi_classes(match_idx, match_label, count, gt_classes, gcount, k):
    b, ncap = match_idx.shape
    cls = torch.empty((b, ncap), dtype=torch.int64, device=match_idx.device)
    check(lib().unit_roi_classes(_p(match_idx), _p(match_label), _p(count), _p(gt_classes), _p(gcount), gt_classes.shape[1], b, ncap,
                                 k, _p(cls), _s()), "roi_classes")
    return cls


def gather_rois(cat, sampled_idx, cls, match_idx, gt, gcount, rois_out=None):
    """rois_out: [b * s, 5] fp32 rows of a larger buffer to write the RoIs into (the step keeps supervised and weak RoIs in one tensor)"""
    b, ncap = cat.shape[0], cat.shape[1]
    s = sampled_idx.shape[1]
    dev = cat.device
    rois = rois_out if rois_out is not None else torch.empty((b * s, 5), dtype=torch.float32, device=dev)
    assert rois.shape == (b * s, 5) and rois.is_contiguous()
    rcls = torch.empty((b * s,), dtype=torch.int32, device=dev)
    rgt = torch.empty((b * s, 4), dtype=torch.float32, device=dev)
    check(lib().unit_gather_rois(_p(cat), ncap, _p(sampled_idx), s, _p(cls), _p(match_idx), _p(gt), _p(gcount), gt.shape[1], b,
                                 _p(rois), _p(rcls), _p(rgt), _s()), "gather_rois")
    return rois, rcls, rgt


def first_k_rois(props, pcount, s, batch_index_offset=0, rois_out=None):
    b, pcap = props.shape[0], props.shape[1]
    rois = rois_out if rois_out is not None else torch.empty((b * s, 5), dtype=torch.float32, device=props.device)
    assert rois.shape == (b * s, 5) and rois.is_contiguous()
    valid = torch.empty((b * s,), dtype=torch.int32, device=props.device)
    check(lib().unit_first_k_rois(_p(props), _p(pcount), pcap, s, b, batch_index_offset, _p(rois), _p(valid), _s()), "first_k_rois")
    return rois, valid


# ------------------------------------------------------------------------------------------------ a8
def roi_align(feat, rois, pooled_size=14, out_size=None, bin_step=1, spatial_scale=1.0 / 16, sampling_ratio=0, aligned=True,
              roi_count=None, out=None, image_offset=0):
    """image_offset: the RoIs' batch indices count from `image_offset` (feat[0] is image `image_offset` of the step's batch: the weak images
    of a two-pass step). The kernel addresses image b at base + b * H * W * C, so the base handed over is moved back by that many images."""
    n, h, w, c = feat.shape
    r = rois.shape[0]
    out_size = out_size or pooled_size
    if out is None:
        out = torch.empty((r, out_size, out_size, c), dtype=feat.dtype, device=feat.device)
    # algorithmic bytes: the maps once + the pooled tensor once; `ref` = the reference-equivalent full 14x14 output
    es = feat.element_size()
    # ref_bytes: what the reference's ROIAlignV2 call moves for the same RoIs -- the full pooled_size x pooled_size grid (roi_heads.py:499:
    # 14 x 14), of which the strided mode materialises only the bins Res5's stride-2 1x1 convs read
    with _timed("roi_align_fwd", 0.0, (feat.numel() + r * out_size * out_size * c) * es, (feat.numel() + r * pooled_size * pooled_size * c) * es):
        _p(feat)
        base = ctypes.c_void_p(feat.data_ptr() - image_offset * h * w * c * es)
        check(lib().unit_roi_align_fwd(base, dt(feat.dtype), n + image_offset, h, w, c, _p(rois), _p(roi_count), r, pooled_size, out_size, bin_step,
                                       float(spatial_scale), sampling_ratio, int(aligned), _p(out), _s()), "roi_align_fwd")
    return out


def roi_align_bwd(gout, feat_shape, rois, dfeat32=None, pooled_size=14, bin_step=1, spatial_scale=1.0 / 16, sampling_ratio=0,
                  aligned=True, roi_count=None):
    n, h, w, c = feat_shape
    r, out_size = gout.shape[0], gout.shape[1]
    if dfeat32 is None:
        dfeat32 = torch.zeros((n, h, w, c), dtype=torch.float32, device=gout.device)
    check(lib().unit_roi_align_bwd(_p(gout), dt(gout.dtype), n, h, w, c, _p(rois), _p(roi_count), r, pooled_size, out_size, bin_step,
                                   float(spatial_scale), sampling_ratio, int(aligned), _p(dfeat32), _s()), "roi_align_bwd")
    return dfeat32


def roi_align_bwd_gather(gout, n_images, h, w, rois, out, pooled_size=14, bin_step=1, spatial_scale=1.0 / 16, sampling_ratio=0,
                         aligned=True, rois_per_image=0, image_offset=0, addend=None, addend_images=0, mask_ref=None, roi_count=None):
    """deterministic gather-form RoIAlign backward; `out` [n_images,h,w,C] (fp32 or gout.dtype) is fully overwritten with
    cast((sum_of_roi_contributions [+ addend]) [* (mask_ref > 0)])."""
    r, out_size, c = gout.shape[0], gout.shape[1], gout.shape[3]
    nb = lib().unit_roi_align_bwd_gather_workspace_bytes(r)
    ws = workspace(nb, gout.device, slot=1)
    nb_alg = gout.numel() * gout.element_size() + out.numel() * out.element_size() * (1 + (addend is not None) + (mask_ref is not None))
    with _timed("roi_align_bwd_gather", 0.0, nb_alg, nb_alg + (pooled_size * pooled_size - out_size * out_size) * r * c * gout.element_size()):
        check(lib().unit_roi_align_bwd_gather(_p(gout), dt(gout.dtype), n_images, h, w, c, _p(rois), _p(roi_count), r, rois_per_image,
                                              image_offset, pooled_size, out_size, bin_step, float(spatial_scale), sampling_ratio,
                                              int(aligned), _p(addend), addend_images, _p(mask_ref), _p(out), dt(out.dtype), _p(ws),
                                              ws.numel(), _s()), "roi_align_bwd_gather")
    return out


# ------------------------------------------------------------------------------------------------ losses
def rpn_loss(head, a, dcol0, labels, match_idx, gt_boxes, anchors, normalizer, grad_dtype, gscale=1.0, loss_out=None, weights=(1.0, 1.0)):
    """weights = (loss_rpn_cls, loss_rpn_loc) factors of Detectron2's RPN `loss_weight` (rpn.py:100)"""
    b, hw, ld = head.shape
    ncap = anchors.shape[0]
    loss2 = loss_out if loss_out is not None else torch.empty(2, dtype=torch.float32, device=head.device)
    dhead = torch.empty((b, hw, ld), dtype=grad_dtype, device=head.device)
    nbytes = lib().unit_rpn_loss_scratch_bytes(b, ncap)
    scratch = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=head.device)
    check(lib().unit_rpn_loss_w(_p(head), ld, a, dcol0, _p(labels), _p(match_idx), _p(gt_boxes), gt_boxes.shape[1], _p(anchors), b, ncap,
                                float(normalizer), float(gscale), float(weights[0]), float(weights[1]), _p(loss2), _p(dhead), dt(grad_dtype),
                                _p(scratch), nbytes, _s()), "rpn_loss")
    return loss2, dhead


def sup_scores(delta, dcol0, weak, wcol0, n_oicr, ncls, novel_mask=None, extra=None, ecol0=0):
    r = delta.shape[0]
    out = torch.empty((r, ncls), dtype=torch.float32, device=delta.device)
    check(lib().unit_sup_scores(_p(delta), delta.shape[1], dcol0, _p(weak), weak.shape[1] if weak is not None else 0, wcol0, n_oicr,
                                ncls, _p(novel_mask), _p(extra), extra.shape[1] if extra is not None else 0, ecol0, _p(out), ncls, r,
                                _s()), "sup_scores")
    return out


_LOSS_ACC = {}
_MULTI_WG_LOSSES = os.environ.get("UNIT_MULTI_WG_LOSSES", "1") != "0"


def _loss_acc(device):
    """the 8-byte accumulator of the multi-workgroup loss kernels (include/unit_hip.h: unit_softmax_ce): one per (device, HIP stream), zeroed
    once -- the kernels hand it back zero, launches on a stream are ordered"""
    if not _MULTI_WG_LOSSES or device.type != "cuda":
        return None
    key = (device, raw_stream(device.index))
    a = _LOSS_ACC.get(key)
    if a is None:
        a = _LOSS_ACC[key] = zeros(2, torch.int32, device)
    elif _DEBUG_SYNC:          # diagnostic runs follow faults: a loss kernel that was aborted mid-ticket left its count behind
        check(lib().unit_fill_zero(_p(a), a.numel() * a.element_size(), _s()), "fill_zero")
    return a


def softmax_ce(logits, col0, ncls, labels, weights=None, dy=None, dcol0=0, gscale=1.0, loss_out=None):
    """weights: per-row, NON-NEGATIVE (the reference's are OICR pseudo-label scores, weak_detector_fast_rcnn.py:198-226): with more than 256
    rows the per-workgroup partial sums travel as unsigned fixed point (csrc/losses.hip packed_sum_finish), where a negative, NaN or
    >= 2^23 partial is reported as a NaN loss -- checked here under UNIT_DEBUG_SYNC=1"""
    r, ld = logits.shape
    if _DEBUG_SYNC and weights is not None and weights.numel():
        assert float(weights.min()) >= 0.0, "softmax_ce: negative row weights"
    loss = loss_out if loss_out is not None else torch.empty(1, dtype=torch.float32, device=logits.device)
    check(lib().unit_softmax_ce(_p(logits), ld, col0, ncls, _p(labels), _p(weights), r, float(gscale), _p(loss), _p(dy),
                                dt(dy.dtype) if dy is not None else 0, dy.shape[1] if dy is not None else 0, dcol0, _p(_loss_acc(logits.device)),
                                _s()), "softmax_ce")
    return loss


def box_reg_loss(bbox, col0, k, labels, rois5, gt_boxes, weights, dy=None, dcol0=0, gscale=1.0, loss_out=None):
    r, ld = bbox.shape
    loss = loss_out if loss_out is not None else torch.empty(1, dtype=torch.float32, device=bbox.device)
    w = (ctypes.c_float * 4)(*weights)
    check(lib().unit_box_reg_loss(_p(bbox), ld, col0, k, _p(labels), _p(rois5), _p(gt_boxes), w, r, float(gscale), _p(loss), _p(dy),
                                  dt(dy.dtype) if dy is not None else 0, dy.shape[1] if dy is not None else 0, dcol0, _p(_loss_acc(bbox.device)),
                                  _s()), "box_reg_loss")
    return loss


def wsddn_mil(streams, ccol0, dcol0, k, valid, s, b, multihot, cls_temp, det_temp, mil_multiplier, dy=None, dyc0=0, dyd0=0,
              gscale=1.0, loss_out=None):
    rtot, ld = streams.shape
    loss = loss_out if loss_out is not None else torch.empty(1, dtype=torch.float32, device=streams.device)
    xr = torch.empty((rtot, k), dtype=torch.float32, device=streams.device)
    check(lib().unit_wsddn_mil(_p(streams), ld, ccol0, dcol0, k, _p(valid), s, b, _p(multihot), float(cls_temp), float(det_temp),
                               float(mil_multiplier), float(gscale), _p(loss), _p(xr), _p(dy), dt(dy.dtype) if dy is not None else 0,
                               dy.shape[1] if dy is not None else 0, dyc0, dyd0, _s()), "wsddn_mil")
    return loss, xr


def oicr_targets(src, col0, mode, k, rois5, valid, s, b, multihot, fg_thresh=0.5, bg_thresh=0.1):
    rtot = rois5.shape[0]
    labels = torch.empty((rtot,), dtype=torch.int32, device=src.device)
    weights = torch.empty((rtot,), dtype=torch.float32, device=src.device)
    check(lib().unit_oicr_targets(_p(src), src.shape[1], col0, mode, k, _p(rois5), _p(valid), s, b, _p(multihot), float(fg_thresh),
                                  float(bg_thresh), _p(labels), _p(weights), _s()), "oicr_targets")
    return labels, weights


# ------------------------------------------------------------------------------------------------ a14 / a15
def embedding_similarity(emb, novel_rows, base_rows):
    """fast_rcnn.py:376-382: E[novel] @ E[base]^T (row index lists are device int32 tensors)"""
    out = torch.empty((novel_rows.numel(), base_rows.numel()), dtype=torch.float32, device=emb.device)
    check(lib().unit_embedding_similarity(_p(emb), emb.shape[1], emb.shape[1], _p(novel_rows), novel_rows.numel(), _p(base_rows),
                                          base_rows.numel(), _p(out), _s()), "embedding_similarity")
    return out


def similarity(lin_weak, col0, n_oicr, ncls, base_dev, lingual, n_novel, visual_threshold, use_lingual=True, use_visual=True):
    r = lin_weak.shape[0]
    nb = base_dev.numel()
    sim = torch.empty((r, n_novel, nb), dtype=torch.float32, device=lin_weak.device)
    check(lib().unit_similarity(_p(lin_weak), lin_weak.shape[1], col0, n_oicr, ncls, _p(base_dev), nb, _p(lingual), n_novel,
                                float(visual_threshold), int(use_lingual), int(use_visual), _p(sim), r, _s()), "similarity")
    return sim


def transfer_predictions(lin, ccol0, bcol0, k, weak, wcol0, n_oicr, sim_cls, sim_bbox, base_dev, novel_dev, role_dev, slot_dev,
                         ft=None, fccol0=0, fbcol0=0):
    """-> (scores [R,K+1], bbox [R,4K]) of SupervisedDetectorOutputs*.forward (eval / fine-tune branches)."""
    r = lin.shape[0]
    scores = torch.empty((r, k + 1), dtype=torch.float32, device=lin.device)
    bbox = torch.empty((r, 4 * k), dtype=torch.float32, device=lin.device)
    check(lib().unit_transfer_predictions(_p(lin), lin.shape[1], ccol0, bcol0, k, _p(weak), weak.shape[1] if weak is not None else 0, wcol0,
                                          n_oicr, _p(ft), ft.shape[1] if ft is not None else 0, fccol0, fbcol0, _p(sim_cls), _p(sim_bbox),
                                          _p(base_dev), base_dev.numel(), _p(novel_dev), novel_dev.numel(), _p(role_dev), _p(slot_dev),
                                          _p(scores), k + 1, _p(bbox), 4 * k, r, _s()), "transfer_predictions")
    return scores, bbox


def transfer_predictions_bwd(dy, dccol0, dbcol0, lin, ccol0, bcol0, k, sim_cls, sim_bbox, t, ldl):
    """backward of transfer_predictions w.r.t. the delta heads' outputs and the similarity -> (dlin [R, ldl] in dy's dtype, dsim fp32)"""
    r = lin.shape[0]
    dlin = torch.empty((r, ldl), dtype=dy.dtype, device=lin.device)
    dsim = torch.empty(sim_cls.shape, dtype=torch.float32, device=lin.device)
    check(lib().unit_transfer_predictions_bwd(_p(dy), dt(dy.dtype), dy.shape[1], dccol0, dbcol0, _p(lin), lin.shape[1], ccol0, bcol0, k,
                                              _p(sim_cls), _p(sim_bbox), _p(t["base"]), t["base"].numel(), _p(t["novel"]), t["novel"].numel(),
                                              _p(t["role"]), _p(t["slot"]), _p(dlin), ldl, _p(dsim), r, _s()), "transfer_predictions_bwd")
    return dlin, dsim


def similarity_bwd(lin_weak, col0, n_oicr, ncls, base_dev, lingual, n_novel, visual_threshold, use_lingual, use_visual, dsim, grad_dtype):
    """backward of `similarity` -> d(loss)/d(lin_weak) [R, ld] (only the OICR logit columns are non-zero)"""
    r, ld = lin_weak.shape
    dlin = torch.empty((r, ld), dtype=grad_dtype, device=lin_weak.device)
    check(lib().unit_similarity_bwd(_p(lin_weak), ld, col0, n_oicr, ncls, _p(base_dev), base_dev.numel(), _p(lingual), n_novel,
                                    float(visual_threshold), int(use_lingual), int(use_visual), _p(dsim), _p(dlin), dt(grad_dtype), ld, col0, r,
                                    _s()), "similarity_bwd")
    return dlin


def softmax_rows(x, ncls):
    y = torch.empty((x.shape[0], ncls), dtype=torch.float32, device=x.device)
    check(lib().unit_softmax_rows(_p(x), x.shape[1], ncls, _p(y), ncls, x.shape[0], _s()), "softmax_rows")
    return y


def detections(probs, deltas, props, pcount, image_hw, weights, score_thresh, nms_thresh, topk, cand_cap=None):
    """fast_rcnn_inference for B images: probs [B*Rcap, K+1], deltas [B*Rcap, 4K], props [B,Rcap,4], pcount [B] (device)
    -> (boxes [B,topk,4], scores [B,topk], classes [B,topk], roi_idx [B,topk], count [B])"""
    b, rcap = props.shape[0], props.shape[1]
    k = deltas.shape[1] // 4
    dev = probs.device
    cap = cand_cap or min(rcap * k, 65536)
    cb = torch.empty((b, cap, 4), dtype=torch.float32, device=dev)
    cs = zeros((b, cap), torch.float32, dev)
    cc = torch.empty((b, cap), dtype=torch.int32, device=dev)
    cr = torch.empty((b, cap), dtype=torch.int32, device=dev)
    cnt = torch.empty((b,), dtype=torch.int32, device=dev)
    cmax = torch.empty((b,), dtype=torch.float32, device=dev)
    w = (ctypes.c_float * 4)(*weights)
    check(lib().unit_detection_candidates(_p(probs), probs.shape[1], _p(deltas), deltas.shape[1], _p(props), _p(pcount), b, rcap, k, w,
                                          SCALE_CLAMP, _p(image_hw), float(score_thresh), cap, _p(cb), _p(cs), _p(cc), _p(cr), _p(cnt),
                                          _p(cmax), _s()), "detection_candidates")
    # stable descending sort of the candidate scores (unused tail slots hold 0 <= thresh and sort behind every candidate)
    # (chip-wide select + rank sort over the valid candidates only: every valid score is > score_thresh >= 0 = the fill value)
    _, order = sort_desc(cs, b, cap, topk=cap, min_exclusive=0.0 if score_thresh >= 0 else None)
    ob = torch.empty((b, cap, 4), dtype=torch.float32, device=dev)
    check(lib().unit_detection_offset_gather(_p(cb), _p(cc), _p(order), _p(cnt), _p(cmax), b, cap, _p(ob), _s()), "detection_offset_gather")
    keep, kc, _, _ = nms(ob, cs, cnt, nms_thresh, topk)
    oboxes = torch.empty((b, topk, 4), dtype=torch.float32, device=dev)
    oscores = torch.empty((b, topk), dtype=torch.float32, device=dev)
    ocls = torch.empty((b, topk), dtype=torch.int32, device=dev)
    oroi = torch.empty((b, topk), dtype=torch.int32, device=dev)
    ocnt = torch.empty((b,), dtype=torch.int32, device=dev)
    check(lib().unit_detection_finalize(_p(cb), _p(cs), _p(cc), _p(cr), _p(order), _p(keep), _p(kc), b, cap, topk, _p(oboxes), _p(oscores),
                                        _p(ocls), _p(oroi), _p(ocnt), _s()), "detection_finalize")
    return oboxes, oscores, ocls, oroi, ocnt


def detector_postprocess(boxes, count, scale_xy, out_hw):
    b, topk = boxes.shape[0], boxes.shape[1]
    nonempty = torch.empty((b, topk), dtype=torch.uint8, device=boxes.device)
    check(lib().unit_detector_postprocess(_p(boxes), _p(count), b, topk, _p(scale_xy), _p(out_hw), _p(nonempty), _s()), "detector_postprocess")
    return nonempty


def compact_detections(boxes, sc, cls, roi, cnt, nonempty=None, masks=None):
    """stable compaction of each image's kept detections (j < cnt[b] and nonempty[b][j]) to the front of its block (unit_compact_detections)
    -> (boxes, scores, classes int64, roi index, masks | None, kept count int32 [B])"""
    b, topk = boxes.shape[0], boxes.shape[1]
    dev = boxes.device
    ob, osc = torch.empty_like(boxes), torch.empty_like(sc)
    ocls = torch.empty((b, topk), dtype=torch.int64, device=dev)
    oroi = torch.empty_like(roi)
    om = torch.empty_like(masks) if masks is not None else None
    kept = torch.empty((b,), dtype=torch.int32, device=dev)
    me = masks[0, 0].numel() if masks is not None else 0
    check(lib().unit_compact_detections(_p(boxes), _p(sc), _p(cls), _p(roi), _p(masks), me, _p(cnt), _p(nonempty), b, topk, _p(ob), _p(osc), _p(ocls),
                                        _p(oroi), _p(om), _p(kept), _s()), "compact_detections")
    return ob, osc, ocls, oroi, om, kept


def boxes_to_rois5(boxes):
    """[B,T,4] -> [B*T,5] RoIAlign rows (image index, box)"""
    b, t = boxes.shape[0], boxes.shape[1]
    out = torch.empty((b * t, 5), dtype=torch.float32, device=boxes.device)
    check(lib().unit_boxes_to_rois5(_p(boxes), b, t, _p(out), _s()), "boxes_to_rois5")
    return out


def gather_rows(src, idx, rcap):
    """src [B*rcap, ...] rows, idx int32 [B,T] (negative = row 0) -> [B*T, ...] = src[b*rcap + idx[b][j]]"""
    b, t = idx.shape
    row = src[0].numel() * src.element_size()
    out = torch.empty((b * t,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    check(lib().unit_gather_rows(_p(src), _p(idx), b, t, rcap, row, _p(out), _s()), "gather_rows")
    return out


def gather_blocks(src, nb, block_rows, take):
    """src [nb*block_rows, ...] (or more rows) -> dense [nb*take, ...]: the first `take` rows of every block (one launch instead of a
    torch.cat of nb slices)"""
    assert src.is_contiguous() and src.shape[0] >= nb * block_rows
    x3 = type(src) is X3
    row = src[0].numel() * src.element_size()
    out = torch.empty((nb * take,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    if x3:
        out = out.as_subclass(X3)
    ptr = (lambda t: ctypes.c_void_p(t.data_ptr())) if x3 else _p
    check(lib().unit_gather_blocks(ptr(src), nb, block_rows, take, row, ptr(out), _s()), "gather_blocks")
    return out


def paste_masks(probs, boxes, out_hw, threshold=0.5, valid=None):
    """probs [S,M,M] fp32, boxes [S,4] (output-image coordinates) -> uint8 [S,H,W]  (detector_postprocess mask pasting)"""
    s, m = probs.shape[0], probs.shape[-1]
    h, w = int(out_hw[0]), int(out_hw[1])
    out = torch.empty((s, h, w), dtype=torch.uint8, device=probs.device)
    check(lib().unit_paste_masks(_p(probs.contiguous()), _p(boxes.contiguous()), _p(valid), s, m, h, w, float(threshold), _p(out), _s()),
          "paste_masks")
    return out


def sum_losses(losses, out=None):
    out = out if out is not None else torch.empty(1, dtype=torch.float32, device=losses.device)
    check(lib().unit_sum_losses(_p(losses), losses.numel(), _p(out), _s()), "sum_losses")
    return out


def sgd_momentum(p, g, buf, lr, momentum, weight_decay, grad_scale=1.0, first_step=False, lr_dev=None):
    check(lib().unit_sgd_momentum(_p(p), _p(g), _p(buf), p.numel(), float(lr), float(momentum), float(weight_decay), float(grad_scale),
                                  int(first_step), _p(lr_dev), _s()), "sgd_momentum")
