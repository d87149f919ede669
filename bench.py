#!/usr/bin/env python
"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one UniT base-training step exactly as TrainerNoMeta.run_step (engine/defaults.py:266-288): 2 supervised + 2
weak 3x600x1000 images per GPU through ResNet-101-C4 + RPN + RoIAlign + the two Res5 heads + all eight losses, explicit
backward, fused SGD step (variant "S1"). Synthetic VOC-shaped inputs already resident in HBM,
random-init weights of the real architecture. `value` = supervised images / second over ALL ranks (weak scaling: 2
supervised images per GPU). Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOPs per step per GPU, SURVEY.md section 8d (1 MAC = 2 FLOP; fwd + bwd of trainable convs)
STEP_TFLOP = {("s1", 101): 12.70, ("s1", 50): 11.61, ("s0", 101): 5.69}
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16, MI355X_MICROARCH.md (never the 2:1-sparse figure)
HBM_PEAK_GBS = 8000.0       # HBM3E spec, same guide (6.3 TB/s is what a float4 copy reaches)
FEED_B_PER_CLK = 27.0       # bytes per clock a CU takes in by LDS-DMA inside a conv loop (8 rows x 128 B pieces beside MFMAs and fragment reads:
FEED_CLOCK_HZ = 2.1e9       # tools/feed_matrix.hip "conv-like" rows, profiles/r04_exp_feed_matrix.txt), at the clock the chip holds on those loops


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--depth", type=int, default=101)
    ap.add_argument("--variant", default="s1", choices=["s1", "s0"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "bf16x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="single HIP stream for the whole run (profiling aid: every "
                    "kernel's rocprof duration is then that kernel alone)")
    ap.add_argument("--graph", action="store_true", help="capture the whole step in a hipGraph and replay it (engine.GraphedStep; "
                    "N > 1: two graphs around one eager all-reduce of the flat gradient buffer). Measured at N=1: host enqueue per step "
                    "drops, the device runs the replay ~2.5 %% slower than the eager four-stream schedule: off by default")
    ap.add_argument("--launch", default=None, choices=["replay", "eager", "graph"], help="how the step's ~650 launches reach the device: replay "
                    "(default: engine.ReplayedStep -- the eager step's C-ABI call sequence recorded once per batch key and re-issued by one C call per "
                    "segment, csrc/replay.hip: the eager four-stream schedule without the interpreter), eager (Python issues every launch), graph "
                    "(= --graph: hipGraph replay)")
    ap.add_argument("--sustain-steps", type=int, default=300, help="N=1: after the timed region the same step runs this many more times and the last "
                    "two thirds are timed (`sustained_images_per_sec`: held clocks next to the burst figure); 0 = skip")
    ap.add_argument("--early-update", action="store_true", help="per-bucket optimizer updates beside the backward instead of one "
                    "update after it (engine.EarlyUpdate; measured 18.56 vs 18.43 ms per step: off by default)")
    ap.add_argument("--high-priority", action="store_true", help="run the steps on a high-priority HIP stream instead of PyTorch's default stream "
                    "(GeneralizedRCNN.high_priority_stream: the forward / dgrad chain ahead of the weight-gradient and head side streams; measured "
                    "between -0.17 and +0.05 ms per step over four A/B series: inside the noise, off by default)")
    ap.add_argument("--tail-overlap", action="store_true", help="let the end of a step (last weight gradients, SGD, weight re-preparation) "
                    "overlap the next step's frozen layers instead of joining the weight-gradient stream first "
                    "(GeneralizedRCNN.overlap_optimizer_tail; measured 119.3 vs 121.4 images/s at N=1: off by default)")
    ap.add_argument("--bucket-mb", type=float, default=None, help="gradient bucket size in MB (parallel.GradBuckets; default 64 / UNIT_BUCKET_MB)")
    ap.add_argument("--reduce-mode", default=None, choices=["allreduce", "rs_ag", "direct", "cabi"], help="how a gradient bucket crosses xGMI (cabi = one all-reduce through the library's own RCCL binding): one all-reduce "
                    "(default), reduce-scatter + all-gather in place, or all-to-all + ordered owner-side sum + all-gather (all 7 links at once, "
                    "bit-reproducible); parallel.py")
    ap.add_argument("--bf16-buckets", action="store_true", help="exchange bf16 copies of the gradient buckets (half the bytes)")
    ap.add_argument("--force-collectives", action="store_true", help="N=1 only: initialise RCCL with one rank and keep every collective of the "
                    "data-parallel step in the timed region (what an N>1 run adds on the device side, minus the wire)")
    ap.add_argument("--no-tune", action="store_true", help="N > 1: skip the ~30 s exchange probe (reduce mode x bucket size x collective stream) "
                    "and run the timed region on the flags / environment as given")
    ap.add_argument("--collective-stream", default=None, choices=["rpn", "own", "internal"], help="where a bucket's collectives are enqueued "
                    "(parallel.GradBuckets; default rpn / UNIT_COLLECTIVE_STREAM)")
    ap.add_argument("--shapes", default="fixed", choices=["fixed", "voc"], help="fixed: the headline 600x1000 workload; voc: a SECONDARY labelled line "
                    "over steps whose image sizes are drawn like ResizeShortestEdge((480..800), 1333) on VOC aspect ratios "
                    "(configs/VOC/VOC-RCNN-101-C4-split1.yaml:27-29)")
    ap.add_argument("--config", default=None, choices=["r50_s1", "s2", "coco_mask", "eval"],
                    help="SECONDARY labelled lines for the other BASELINE.json configurations at their own size (N = 1): r50_s1 = config 2 (the headline "
                         "path at --depth 50), s2 = config 4 (R101 1-shot fine-tune step), coco_mask = config 5 (COCO K = 80 + mask head step), "
                         "eval = the inference path (VOC R101 1 / 4 images per call, COCO K = 80 with the 80-class NMS)")
    ap.add_argument("--dry-launch", action="store_true", help="launcher self-test: the rank processes print their rendezvous environment and exit "
                    "(no GPU call anywhere)")
    ap.add_argument("--launch-timeout", type=float, default=3600.0, help="seconds the launcher waits for its rank processes")
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpu_count():
    """GPUs this process may use, counted WITHOUT touching the HIP runtime (the launcher parent must stay GPU-free: a process that has
    initialised the GPU may not start others on this pool): KFD topology nodes with SIMDs, else /dev/dri render nodes, capped by the
    *_VISIBLE_DEVICES lists. None when neither source is readable (the ranks then find out themselves)."""
    import glob
    import re
    n = None
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        cnt = 0
        for d in os.listdir(base):
            m = re.search(r"simd_count\s+(\d+)", open(os.path.join(base, d, "properties")).read())
            cnt += 1 if (m and int(m.group(1)) > 0) else 0
        n = cnt
    except OSError:
        r = glob.glob("/dev/dri/renderD*")
        n = len(r) if r else None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:          # (a visibility list caps the count even where sysfs is not readable)
            k = len([x for x in v.split(",") if x.strip() != ""])
            n = k if n is None else min(n, k)
    return n


def pick_exchange(table):
    """table: [{"mode", "bucket_mb", "stream", "ms" (max over ranks) | None, "error" | None}] -> the entry the timed region runs on: the
    fastest measured one (ties: fewer bytes in flight per collective first, then the table's order); None when nothing ran"""
    ok = [(e["ms"], i) for i, e in enumerate(table) if e.get("ms") is not None and not e.get("error")]
    if not ok:
        return None
    best = min(ok)
    return table[best[1]]


def _thread_cpu():
    """{tid: (name, CPU seconds)} of this process's threads (Linux /proc): which thread burns the CPU time `host_cpu_ms_per_step` reports"""
    out = {}
    try:
        tck = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            with open(f"/proc/self/task/{tid}/stat") as f:
                st = f.read()
            name = st[st.index("(") + 1:st.rindex(")")]
            rest = st[st.rindex(")") + 2:].split()
            out[tid] = (name, (int(rest[11]) + int(rest[12])) / tck)
    except (OSError, ValueError):
        pass
    return out


def launch_mode(args):
    """replay | eager | graph for this run: --graph and the modes that need Python inside the step (early update, tail overlap, the single-stream
    profiling run) keep their form; everything else replays the recorded call list"""
    if args.graph:
        return "graph"
    if args.launch:
        return args.launch
    if args.early_update or args.tail_overlap:
        return "eager"
    if args.shapes == "voc":
        # multi-scale batches: a batch key (image sizes, GT capacity) almost never comes back inside a run -- 40 eager + 10 recorded + 0 replayed
        # steps in profiles/r06_bench_voc_shapes_replay_per_key.json -- so recording buys nothing there and costs the recording
        return "eager"
    return "replay"


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process -- which has made NO GPU call and makes
    none -- starts N fresh copies of itself, one rank per GPU (the reference: scripts/train_VOC.py:67-77 -> detectron2 launch ->
    mp.spawn + init_process_group("NCCL", tcp://127.0.0.1:port)), waits for them, prints rank 0's JSON line if and only if every
    rank ended with status 0, and returns the first non-zero status otherwise. Children are ended by PID, never by pattern."""
    import subprocess
    import tempfile
    n = args.gpus
    backend = os.environ.get("UNIT_DIST_BACKEND", "nccl")
    if not args.dry_launch and backend == "nccl":
        ndev = visible_gpu_count()          # sysfs only: this process never initialises the GPU runtime
        if ndev is not None and ndev < n:
            print(f"bench.py: --gpus {n} needs {n} visible GPUs (one rank per GPU over RCCL), this box shows {ndev}. Nothing was run. "
                  f"(A single-GPU rehearsal of the multi-process path exists: UNIT_DIST_BACKEND=gloo python bench.py --gpus {n}.)", file=sys.stderr)
            return 2
    port = _free_port()
    outs = [tempfile.NamedTemporaryFile(prefix=f"unit_bench_rank{r}_", suffix=".out", delete=False) for r in range(n)]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), UNIT_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=outs[r]))
    deadline = time.time() + args.launch_timeout
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            c = p.poll()
            if c is not None:
                live.remove(p)
                if c != 0 and rc == 0:
                    rc = c if c > 0 else 1
        if (rc != 0 or time.time() > deadline) and live:
            if rc == 0:
                rc = 124
                print(f"bench.py: rank processes still running after {args.launch_timeout:.0f} s", file=sys.stderr)
            for p in live:          # one rank failed: the others would wait in a collective for ever
                p.terminate()
            t_end = time.time() + 20
            for p in live:
                try:
                    p.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
            live = []
        else:
            time.sleep(0.05)
    texts = []
    for f in outs:
        f.close()
        texts.append(open(f.name).read())
        os.unlink(f.name)
    text = "".join(texts) if args.dry_launch else texts[0]          # only rank 0 prints the result line
    if rc != 0:
        print(f"bench.py: a rank process ended with status {rc}; no result line", file=sys.stderr)
        sys.stderr.write(text[-2000:])
        return rc
    sys.stdout.write(text)
    sys.stdout.flush()
    return 0


def cpu_baseline(depth, variant):
    """The oracle (CPU restatement of the reference path, kind "port") timed on this box's host cores -- as many threads as a short probe
    finds fastest, up to the whole affinity set (torch.set_num_threads(os.cpu_count())) -- on a BOUNDED sample: one S1 step (forward + backward) on 2 supervised + 2 weak
    600x1000 images -- first with 256 RoIs per image (half of the bench workload's; ~10 s), then, when that took under 40 s, with the full
    512: `value` is the full workload's figure when it ran, the half sample's otherwise (both are reported)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import unit_oracle as orc
    from unit_amd import config
    from unit_amd.modeling import build_model
    from unit_amd.synthetic import init_synthetic_weights, synthetic_batch
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # How many threads: every core of the affinity set is the plan -- but the GPU boxes show 256 logical CPUs to a container whose
    # CPU share is a fraction of them, and PyTorch-CPU convolutions then THRASH (measured on a box, tools/cpu_threads_probe.py: the Res5 3x3 conv
    # 2.1 / 2.7 / 1.8 / 1.0 / 0.33 TFLOP/s at 16 / 32 / 64 / 128 / 256 threads). So the count is measured, not assumed: a 2-second probe of one
    # res4-sized convolution per candidate, the fastest wins, the table goes into the line.
    import torch.nn.functional as F
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else round(float(q) / float(per), 2)
    except (OSError, ValueError):
        pass
    cands = sorted({c for c in (8, 16, 32, 64, 128, avail) if c <= avail})
    px, pw = torch.randn(4, 256, 38, 63), torch.randn(256, 256, 3, 3)
    probe = {}
    for cnd in cands:
        torch.set_num_threads(cnd)
        F.conv2d(px, pw, padding=1)
        t0 = time.time()
        reps = 0
        while reps < 3 or (time.time() - t0 < 0.25 and reps < 50):
            F.conv2d(px, pw, padding=1)
            reps += 1
        probe[cnd] = round((time.time() - t0) / reps * 1e3, 2)          # ms per convolution
        if probe[cnd] > 4 * min(probe.values()):
            break                                                        # (past the knee: more threads only get slower)
    cores = min(probe, key=probe.get)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    torch.set_num_threads(cores)
    cfg = config.voc_rcnn_c4_split1(depth)
    cfg.MODEL.DEVICE = "cpu"
    m = build_model(cfg)
    init_synthetic_weights(m, seed=1)
    trainable = {n for n, p in m.named_parameters() if p.requires_grad}
    sd = m.state_dict()
    sup, weak = synthetic_batch(2, 2 if variant == "s1" else 0, seed=0)
    g = torch.Generator().manual_seed(2)
    perms = dict(rpn=[torch.randperm(38 * 63 * 15, generator=g) for _ in range(2)], roi=[torch.randperm(2000 + 8, generator=g) for _ in range(2)])
    full = STEP_TFLOP.get((variant, depth))
    heads = {"s1": 7, "s0": 3}[variant] * 1.499

    def sample(rois):
        p = {}
        for k, v in sd.items():
            t = v.detach().clone().contiguous()
            if k in trainable:
                t.requires_grad_(True)
            p[k] = t
        ocfg = dict(depth=depth, num_classes=20, novel_classes=list(cfg.DATASETS.FEWSHOT.NOVEL_CLASSES_ID), pixel_mean=cfg.MODEL.PIXEL_MEAN,
                    pixel_std=cfg.MODEL.PIXEL_STD, rois_per_image=rois, pre_nms_topk=12000, post_nms_topk=2000, multi_box_head=True)
        t0 = time.time()
        losses, _ = orc.step_losses(p, [x["image"] for x in sup], [x["instances"].gt_boxes.tensor for x in sup],
                                    [x["instances"].gt_classes for x in sup], [x["image"] for x in weak] if weak else None,
                                    [x["instances"].gt_classes for x in weak] if weak else None, perms, ocfg)
        sum(losses.values()).backward()
        dt = time.time() - t0
        # FLOP normalisation: the sample runs every conv of the step except that the three Res5 passes see rois / 512 of the
        # RoIs -- R101 S1: 4 x 166.1 + 4 x 45.6 (+ backward 4 x 2 x 147.3 + 2 x 2 x 45.6) GFLOP in full, 7 x 1499 * rois / 512 for the heads
        tf = round(full - heads * (1 - rois / 512.0), 2) if full else None
        return {"rois_per_image": rois, "seconds": round(dt, 1), "images_per_sec": round(2.0 / dt, 4), "sample_tflop": tf,
                "cpu_tflops": round(tf / dt, 3) if tf else None, "images_per_sec_flop_normalised": round(2.0 / dt * tf / full, 4) if tf else None}

    half = sample(256)
    whole = sample(512) if half["seconds"] < 40.0 else None
    best = whole or half
    cpu_model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": best["images_per_sec"], "unit": "images/sec", "cores": cores, "cpu_model": cpu_model, "host_cores_available": avail,
            "cgroup_cpu_quota": quota, "thread_probe_ms_per_res4_conv": probe, "kind": "port",
            "sample": f"oracle (PyTorch-CPU fp32 + C) S1 fwd+bwd, R{depth}-C4, 2 supervised + 2 weak 3x600x1000 images, "
                      f"{best['rois_per_image']} RoIs/image" + (" (the bench workload)" if whole else " (1/2 of 512)") + f", {best['seconds']} s on {cores} threads",
            "sample_tflop": best["sample_tflop"], "step_tflop": full, "cpu_tflops": best["cpu_tflops"],
            "value_flop_normalised": best["images_per_sec_flop_normalised"], "half_sample": half, "full_sample": whole}


def _kernel_rates(prof, n_iter):
    """per kernel family of an ops.PROFILER pass: launches per iteration, average launch time, TFLOP/s (MFMA) or GB/s (HBM) on the algorithmic work"""
    out = {}
    for k, ev in prof.items():
        if not ev:
            continue
        tot = sum(e[0].elapsed_time(e[1]) for e in ev)      # ms
        fl, by = sum(e[2] for e in ev), sum(e[3] for e in ev)
        d = {"launches_per_iter": round(len(ev) / n_iter, 1), "avg_launch_us": round(tot / len(ev) * 1e3, 2), "ms_per_iter": round(tot / n_iter, 3)}
        if fl:
            d["tflops"] = round(fl / tot / 1e9, 1)
        elif by:
            d["gbs_on_algorithmic_bytes"] = round(by / tot / 1e6, 1)
        out[k] = d
    return out


def other_config_run(args, dev):
    """SECONDARY lines (never the headline): BASELINE.json configs 4 / 5 and the inference path at their own size on one GPU, each with the
    dominant kernel's roofline entry from a single-stream HIP-event pass and the latency kernels (top-k sort, NMS) in us."""
    from unit_amd import _lib, config, ops
    from unit_amd.modeling import build_model
    from unit_amd.parallel import GradBuckets
    from unit_amd.solver import FlatSGD
    from unit_amd.synthetic import init_synthetic_weights, synthetic_batch

    def timed(fn, iters, warm):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    def profiled(fn, model, iters=3):
        was = getattr(model, "overlap_streams", True)
        model.overlap_streams = False
        fn()
        torch.cuda.synchronize()
        prof = {}
        ops.PROFILER = prof
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        ops.PROFILER = None
        model.overlap_streams = was
        return _kernel_rates(prof, iters)

    def roofline_of(rates):
        conv = {k: v for k, v in rates.items() if k in ("conv_igemm256", "conv_igemm_dma", "conv_igemm", "conv_wgrad") and "tflops" in v}
        if not conv:
            return None
        key = max(conv, key=lambda k: conv[k]["ms_per_iter"])
        r = conv[key]
        return {"kernel": {"conv_igemm256": "conv_igemm256_p8_kernel", "conv_igemm_dma": "conv_igemm_lc / conv_igemm_dma kernels", "conv_wgrad": "conv_wgrad group kernels"}.get(key, key),
                "bound": "mfma", "achieved": r["tflops"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(r["tflops"] / MFMA_PEAK_TFLOPS, 4),
                "traffic": None, "launches_per_iter": r["launches_per_iter"], "avg_launch_us": r["avg_launch_us"], "ms_per_iter": r["ms_per_iter"],
                "measured_over": "3 iterations on ONE HIP stream with a HIP-event pair around every launch", "families": rates}

    common = {"unit": "images/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
              "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", "secondary": True, "build_hash": _lib.build_hash()}
    lines = []
    if args.config == "s2":
        cfg = config.voc_rcnn_c4_split1_ft(args.depth)
        cfg.MODEL.DEVICE, cfg.SEED = str(dev), 0
        m = build_model(cfg)
        init_synthetic_weights(m, seed=1)
        m.train()
        m.compute_mode = args.dtype
        sup, _ = synthetic_batch(2, 0, seed=100, base_ids=list(range(20)))
        batch = m.pack_batch(sup, None)
        b = GradBuckets(m)
        opt = FlatSGD(m, cfg, grad_scale=b.grad_scale)

        def step():
            st = m.forward_train(batch, early_backward=True)
            m.backward_train(st)
            b.finish()
            opt.step()
        ms = timed(step, args.steps, args.warmup)
        rates = profiled(step, m)
        lines.append(dict(common, metric=f"SECONDARY images/sec (fwd+bwd+SGD) R{args.depth}-C4 VOC 1-shot fine-tune step S2, 600x1000 bs=2/GPU (BASELINE.json config 4)",
                          value=round(2e3 / ms, 2), ms_per_step=round(ms, 3),
                          config={"workload": f"UniT fine-tune step S2 (TrainerFineTune.run_step, engine/defaults.py:442-463; configs/VOC/FT/1_shot/VOC-RCNN-101-C4-split1-ft.yaml): "
                                              f"ResNet-{args.depth}-C4, 2 images 3x600x1000, 512 RoIs/image, both Res5 heads forward, similarity transfer in training, only "
                                              "cls_score_ft / bbox_pred_ft train", "step_tflop_per_gpu": 3.42, "step_tflops_achieved_per_gpu": round(3.42 / ms * 1e3, 1)},
                          roofline=roofline_of(rates), latency_kernels_us={k: rates[k]["avg_launch_us"] for k in ("sort_topk", "nms") if k in rates}))
    elif args.config == "coco_mask":
        cfg = config.coco_rcnn_c4_split1_segm(args.depth)
        cfg.MODEL.DEVICE, cfg.SEED = str(dev), 0
        m = build_model(cfg)
        init_synthetic_weights(m, seed=1)
        m.train()
        m.compute_mode = args.dtype
        sup, weak = synthetic_batch(2, 2, num_classes=80, base_ids=list(cfg.DATASETS.FEWSHOT.BASE_CLASSES_ID), seed=100)
        yy, xx = torch.meshgrid(torch.arange(600.0), torch.arange(1000.0), indexing="ij")
        for x in sup:          # bitmask ground truth: the ellipse inscribed in every box
            bx = x["instances"].gt_boxes.tensor
            x["instances"].gt_masks = torch.stack([(((xx - (q[0] + q[2]) / 2) / ((q[2] - q[0]) / 2)) ** 2 + ((yy - (q[1] + q[3]) / 2) / ((q[3] - q[1]) / 2)) ** 2) <= 1.0 for q in bx])
        batch = m.pack_batch(sup, weak)
        b = GradBuckets(m)
        opt = FlatSGD(m, cfg, grad_scale=b.grad_scale)

        def step():
            st = m.forward_train(batch, early_backward=True)
            m.backward_train(st)
            b.finish()
            opt.step()
            return st.losses
        ms = timed(step, args.steps, args.warmup)
        assert torch.isfinite(step()).all()
        rates = profiled(step, m)
        bb = {50: 75.4, 101: 166.1}[args.depth], {50: 56.6, 101: 147.3}[args.depth]
        tf = (4 * bb[0] + 4 * 45.6 + 2 * 1499 + 4 * 2 * bb[1] + 2 * 2 * 45.6 + 2 * 2 * 1499) / 1e3
        lines.append(dict(common, metric=f"SECONDARY images/sec (fwd+bwd+SGD) R{args.depth}-C4 COCO K=80 + mask head step, 600x1000 bs=2/GPU (BASELINE.json config 5)",
                          value=round(2e3 / ms, 2), ms_per_step=round(ms, 3),
                          config={"workload": f"UniT base-training step with the mask head (configs/COCO/COCO-RCNN-50-C4-split1-segm.yaml at RESNETS.DEPTH {args.depth}): K = 80, "
                                              "2 supervised + 2 weak 3x600x1000 images, ONE Res5 head over 2048 RoIs (MULTI_BOX_HEAD False) keeping its map, mask head "
                                              "(deconv 2x2 + 1x1) on <= 128 fg RoIs / image with bitmask targets, nine losses, SGD",
                                  "step_tflop_per_gpu": round(tf, 2), "step_tflops_achieved_per_gpu": round(tf / ms * 1e3, 1)},
                          roofline=roofline_of(rates), latency_kernels_us={k: rates[k]["avg_launch_us"] for k in ("sort_topk", "nms") if k in rates}))
    else:          # eval
        for name, mk, k, ns in (("VOC K=20", lambda: config.voc_rcnn_c4_split1(args.depth), 20, (1, 4)), ("COCO K=80 + masks", lambda: config.coco_rcnn_c4_split1_segm(args.depth), 80, (1,))):
            cfg = mk()
            cfg.MODEL.DEVICE = str(dev)
            m = build_model(cfg)
            init_synthetic_weights(m, seed=1)
            m.eval()
            m.compute_mode = args.dtype
            for n in ns:
                sup, _ = synthetic_batch(n, 0, num_classes=k, seed=7)
                inp = [{"image": x["image"].to(dev), "height": 600, "width": 1000} for x in sup]
                call = lambda: m(inp)
                ms = timed(call, args.steps, args.warmup)
                rates = profiled(call, m)
                lines.append(dict(common, metric=f"SECONDARY images/sec inference R{args.depth}-C4 {name}, {n} image(s) 600x1000 per call",
                                  value=round(n * 1e3 / ms, 2), ms_per_step=round(ms, 3),
                                  config={"workload": f"WeaklySupervisedRCNNNoMeta.inference (meta_arch/rcnn.py:493-542): backbone, RPN 6000 -> 1000 proposals, both Res5 heads, "
                                                      f"similarity transfer, per-class NMS over {k} classes, top-100, detector_postprocess"
                                                      + (", mask head + paste" if k == 80 else "") + "; one host sync for the variable-length Instances",
                                          "images_per_call": n},
                                  roofline=roofline_of(rates), latency_kernels_us={kk: rates[kk]["avg_launch_us"] for kk in ("sort_topk", "nms") if kk in rates}))
            del m
    for ln in lines:
        emit(json.dumps(ln))
    return 0



def voc_shapes_run(args, cfg, model, buckets, opt, rank, world, dev):
    """SECONDARY line (never the headline: BASELINE.json's metric is quoted on fixed 3x600x1000 images): the same S1 step over batches whose
    image sizes change every step the way the reference's loader makes them (synthetic.voc_shaped_steps: VOC aspect ratios,
    ResizeShortestEdge 480-800 / max 1333, one orientation per batch). The supervised and the weak batch of a step almost never pad to the
    same size: such a step runs ONE ragged backbone + RPN-head pass over both groups (ops.Ragged; two passes until round 4); tile policies, weight-gradient split plans and scratch sizes are met
    for the first time inside the timed region, as they are in training."""
    import torch.distributed as dist
    from unit_amd import _lib, ops
    from unit_amd.synthetic import synthetic_batch, voc_shaped_steps
    steps, warm = max(args.steps, 40), max(args.warmup, 3)
    plan = voc_shaped_steps(warm + steps, cfg, seed=100 + rank)
    packed = []
    for i, (sup_hw, weak_hw) in enumerate(plan):
        sup = [synthetic_batch(1, 0, hw=hw, seed=1000 * rank + 10 * i + j)[0][0] for j, hw in enumerate(sup_hw)]
        weak = [synthetic_batch(0, 1, hw=hw, seed=1000 * rank + 10 * i + 5 + j)[1][0] for j, hw in enumerate(weak_hw)]
        packed.append(model.pack_batch(sup, weak, gt_buckets=(8, 16, 32)))          # resident in HBM before the timed region
    gs = None
    mode = launch_mode(args)
    if mode == "graph":
        from unit_amd.engine import GraphedStep
        gs = GraphedStep(model, opt, warmup_steps=2, buckets=buckets)
    elif mode == "replay":
        from unit_amd.engine import ReplayedStep
        gs = ReplayedStep(model, opt, warmup_steps=2, buckets=buckets)

    def run(b):
        if gs is not None:
            return gs.run(packed=b)
        st = model.forward_train(b, early_backward=True)
        model.backward_train(st)
        buckets.finish()
        opt.step()
        return st.losses

    for b in packed[:warm]:
        run(b)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    pol0, ws0, gstat0 = len(ops._POLICY_CACHE), ops.WS_GROWTHS[0], dict(gs.stats) if gs else None
    mem0 = torch.cuda.memory_stats(dev)
    t0 = time.perf_counter()
    for b in packed[warm:]:
        losses = run(b)
    t_host = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mem1 = torch.cuda.memory_stats(dev)
    # the host's own share: the same batches again, each enqueued onto an IDLE device (synchronize before, none after the clock stops) --
    # `host_enqueue_ms_per_step` above includes queue back-pressure whenever the device is the slower side
    idle = []
    for b in packed[warm:warm + 10]:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(b)
        idle.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    host_idle_ms = sorted(idle)[len(idle) // 2] * 1e3
    host_ms = t_host / steps * 1e3
    if world > 1:
        t = torch.tensor([dt, host_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, host_ms = (float(v) for v in t.tolist())
    assert torch.isfinite(losses).all(), f"non-finite losses {losses}"
    if rank == 0:
        two_pass = sum(1 for s_hw, w_hw in plan[warm:] if (max(h for h, _ in s_hw), max(w for _, w in s_hw)) != (max(h for h, _ in w_hw), max(w for _, w in w_hw)))
        px = sum(h * w for s_hw, w_hw in plan[warm:] for h, w in s_hw + w_hw) / steps
        # what the kernels see: every batch of two is padded to its own largest height x largest width (ImageList.from_tensors)
        padded = sum(len(b) * max(h for h, _ in b) * max(w for _, w in b) for s_hw, w_hw in plan[warm:] for b in (s_hw, w_hw)) / steps
        out = {"metric": "images/sec (fwd+bwd+SGD) R101-C4 VOC, multi-scale VOC-shaped batches (SECONDARY line, not BASELINE.json's metric)",
               "headline": False, "value": round(2 * world * steps / dt, 3), "unit": "images/sec", "n_gpus": world, "steps": steps, "warmup": warm,
               "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
               "data": "synthetic", "launch": {"graph": "hipGraph replay per batch key", "replay": "call-list replay per batch key (engine.ReplayedStep)", "eager": "eager"}[mode], "host_enqueue_ms_per_step": round(host_ms, 3),
               "host_enqueue_ms_from_idle_device": round(host_idle_ms, 3),
               "dist": buckets.describe(), "build_hash": _lib.build_hash(),
               "config": {"workload": f"UniT base-training step S1, ResNet-{args.depth}-C4, 2 supervised + 2 weak images per GPU whose sizes change every "
                                      "step: VOC raw sizes through ResizeShortestEdge((480, ..., 800), max 1333) with one orientation per batch "
                                      "(configs/VOC/VOC-RCNN-101-C4-split1.yaml:27-29, data/build.py:476-497), 512 RoIs/image, 12000->2000",
                          "images_per_gpu": 2, "global_batch": 2 * world, "parallelism": f"dp{world}",
                          "mean_pixels_per_step": round(px), "pixels_relative_to_the_600x1000_workload": round(px / (4 * 600 * 1000), 3),
                          "mean_padded_pixels_per_step": round(padded), "padded_pixels_relative_to_the_600x1000_workload": round(padded / (4 * 600 * 1000), 3),
                          "steps_whose_two_batches_pad_differently": two_pass,
                          "backbone_passes_per_such_step": 1 if getattr(model, "ragged_single_pass", False) else 2},
               "shape_churn": {"tile_policy_cache_misses": len(ops._POLICY_CACHE) - pol0, "scratch_buffer_growths": ops.WS_GROWTHS[0] - ws0,
                               "device_mallocs": mem1.get("num_device_alloc", 0) - mem0.get("num_device_alloc", 0),
                               "allocator_retries": mem1.get("num_alloc_retries", 0) - mem0.get("num_alloc_retries", 0),
                               "reserved_gb": round(mem1.get("reserved_bytes.all.peak", 0) / 2 ** 30, 2),
                               "graph": ({k: gs.stats[k] - gstat0[k] for k in gs.stats} if gs else None)}}
        emit(json.dumps(out))
    return 0


_RESULT = None          # the process's ORIGINAL stdout, kept for the one result line (claim_stdout)


def claim_stdout():
    """stdout carries exactly one JSON line. RCCL prints a version banner and its warnings to file descriptor 1 (at communicator creation and
    again at teardown, i.e. after the result line), so the descriptor is duplicated for the result and fd 1 itself is pointed at stderr
    before any library is initialised: whatever a library writes to "stdout" from then on lands in stderr."""
    global _RESULT
    if _RESULT is None:
        sys.stdout.flush()
        _RESULT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(line):
    (_RESULT or sys.stdout).write(line + "\n")
    (_RESULT or sys.stdout).flush()


def _trace(msg):
    """UNIT_BENCH_TRACE=1: milestones on stderr (diagnosing a run that ends without its line)"""
    if os.environ.get("UNIT_BENCH_TRACE"):
        print(f"[bench {os.getpid()} {time.time():.3f}] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    _trace(f"start argv={sys.argv[1:]} WORLD_SIZE={os.environ.get('WORLD_SIZE')} MASTER_PORT={os.environ.get('MASTER_PORT')}")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)          # this process never touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to print a line whose n_gpus is not "
              "what was asked for", file=sys.stderr)
        return 2
    claim_stdout()
    if args.dry_launch:
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")
        emit("DRY_LAUNCH " + json.dumps({k: os.environ.get(k) for k in keys}))
        if os.environ.get("UNIT_DRY_FAIL_RANK") == str(rank):          # launcher self-test: one rank dies, the others hang in a "collective"
            return 3
        if os.environ.get("UNIT_DRY_FAIL_RANK") is not None:
            time.sleep(600)
        return 0
    import torch.distributed as dist
    # UNIT_DIST_BACKEND=gloo + more ranks than GPUs is a single-GPU-box rehearsal of the multi-process path (the ranks then
    # share device 0 and the collectives go through the host); the real run is RCCL ("nccl"), one rank per GPU
    backend = os.environ.get("UNIT_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if world > 1 and backend == "nccl" and torch.cuda.device_count() < world:
        print(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible GPUs: one rank per GPU over RCCL", file=sys.stderr)
        return 2
    if world > 1 or args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        torch.cuda.set_device(local_rank)
        for attempt in range(3):
            try:
                if backend == "nccl":
                    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
                else:
                    dist.init_process_group(backend, rank=rank, world_size=world)
                break
            except Exception as e:          # noqa: a port this process probed free a moment ago can be taken when the store binds it
                if world > 1 or attempt == 2 or "EADDRINUSE" not in str(e):
                    raise
                os.environ["MASTER_PORT"] = str(_free_port())
        _trace("process group initialised")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from unit_amd import _lib, config, ops
    from unit_amd.modeling import build_model
    from unit_amd.parallel import GradBuckets
    from unit_amd.solver import FlatSGD
    from unit_amd.synthetic import init_synthetic_weights, synthetic_batch

    if args.config == "r50_s1":
        args.depth = 50          # BASELINE.json config 2: the headline path itself at RESNETS.DEPTH 50 (the line says so in metric / config)
    elif args.config is not None:
        if world != 1:
            print("bench.py: --config lines are single-GPU figures", file=sys.stderr)
            return 2
        return other_config_run(args, dev)
    cfg = config.voc_rcnn_c4_split1(args.depth)
    cfg.MODEL.DEVICE = f"cuda:{local_rank}"
    cfg.SOLVER.IMS_PER_BATCH = 2 * world
    cfg.SEED = 0
    model = build_model(cfg)
    init_synthetic_weights(model, seed=1)
    model.train()
    model.compute_mode = args.dtype
    model.overlap_streams = not args.no_overlap
    n_weak = 2 if args.variant == "s1" else 0
    sup, weak = synthetic_batch(2, n_weak, seed=100 + rank)   # rank r's shard of the global batch
    batch = model.pack_batch(sup, weak)                      # inputs resident in HBM before the timed region
    def make_buckets(mode=None, bucket_mb=None, stream=None):
        mb = bucket_mb if bucket_mb is not None else args.bucket_mb
        return GradBuckets(model, bucket_bytes=None if mb is None else int(mb * (1 << 20)), bf16=args.bf16_buckets, mode=mode or args.reduce_mode,
                           force=True if args.force_collectives else None, collective_stream=stream or args.collective_stream)

    buckets = make_buckets()
    buckets.broadcast_parameters()
    _trace("model built, parameters broadcast")
    opt = FlatSGD(model, cfg, grad_scale=buckets.grad_scale)
    from unit_amd.engine import EarlyUpdate
    early = EarlyUpdate(model, buckets, opt) if args.early_update else None
    model.overlap_optimizer_tail = args.tail_overlap and not (args.graph or args.early_update or args.no_overlap)

    def one_step():
        step = model.forward_train(batch, early_backward=True)
        model.backward_train(step)
        buckets.finish()
        if early is not None:
            early.join()
        opt.step()
        return step.losses

    # N > 1: the driver's scaling run is ONE run per N -- it tunes itself. ~30 s of 8-step probes over how a gradient bucket crosses xGMI
    # (reduce mode x bucket size x the stream the collectives are enqueued on), every rank timing the same configuration between barriers,
    # the slowest rank's time agreed by one all-reduce; the timed region then runs on the fastest, and the whole table is printed in `dist`.
    tuning = None
    if world > 1 and not (args.no_tune or args.graph or args.early_update or args.shapes != "fixed"):
        modes = [args.reduce_mode] if args.reduce_mode else ["allreduce", "rs_ag", "direct"]
        sizes_mb = [args.bucket_mb] if args.bucket_mb is not None else [25.0, 64.0]
        streams = [args.collective_stream] if args.collective_stream else ["rpn", "own"]
        table = []
        for _ in range(3):
            one_step()          # first steps: allocations, weight copies
        for md in modes:
            for mb in sizes_mb:
                for cs in streams:
                    entry = {"mode": md, "bucket_mb": mb, "stream": cs, "ms": None, "error": None}
                    try:
                        buckets = make_buckets(md, mb, cs)
                        for _ in range(2):
                            one_step()
                        dist.barrier()
                        torch.cuda.synchronize()
                        tp = time.perf_counter()
                        for _ in range(8):
                            one_step()
                        torch.cuda.synchronize()
                        ms_p = (time.perf_counter() - tp) / 8 * 1e3
                    except Exception as e:          # noqa: a backend without this collective (gloo rehearsal): every rank raises alike, before any transfer
                        ms_p, entry["error"] = float("inf"), f"{type(e).__name__}: {str(e)[:120]}"
                    tt = torch.tensor([ms_p if ms_p != float("inf") else 1e30], device=dev, dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)          # the slowest rank's time; 1e30 = some rank could not run it
                    if float(tt) < 1e29:
                        entry["ms"], entry["error"] = round(float(tt), 3), None
                    elif entry["error"] is None:
                        entry["error"] = "failed on another rank"
                    table.append(entry)
        chosen = pick_exchange(table)
        if chosen is None:
            print("bench.py: no exchange configuration ran on every rank", file=sys.stderr)
            return 4
        buckets = make_buckets(chosen["mode"], chosen["bucket_mb"], chosen["stream"])
        tuning = {"probe": "8 steps per configuration after 2 warm-up steps, max over ranks", "table": table,
                  "chosen": {k: chosen[k] for k in ("mode", "bucket_mb", "stream", "ms")}}
        _trace(f"exchange tuned: {tuning['chosen']}")

    if args.shapes == "voc":
        rc = voc_shapes_run(args, cfg, model, buckets, opt, rank, world, dev)
        buckets.close()          # (mode "cabi": the library's own RCCL communicator goes before the process group does)
        if dist.is_initialized():
            dist.destroy_process_group()
        return rc

    if args.high_priority and not (args.no_overlap or args.graph):
        torch.cuda.synchronize()
        torch.cuda.set_stream(model.high_priority_stream())          # the step's main chain ahead of the side streams it forks (rcnn.py)

    timed_step = one_step
    mode = launch_mode(args) if early is None else "eager"
    gs = None
    if mode == "graph":
        from unit_amd.engine import GraphedStep
        gs = GraphedStep(model, opt, warmup_steps=max(1, args.warmup - 2), buckets=buckets)      # the last warm-up steps already replay the graph
        timed_step = lambda: gs.run(packed=batch)
    elif mode == "replay":
        from unit_amd.engine import ReplayedStep
        gs = ReplayedStep(model, opt, warmup_steps=max(1, args.warmup - 3), buckets=buckets)     # eager steps, one recorded step, then replays
        timed_step = lambda: gs.run(packed=batch)
    for _ in range(max(args.warmup, 4 if gs is not None else 0)):
        timed_step()
    if gs is not None:
        assert gs.stats["replayed"] > 0 or args.warmup < 4, gs.stats          # the timed region measures the mode the line names
    buckets.exposed_events = [] if buckets.active else None          # two event records per step around the bucket waits
    buckets.exposed_per_bucket = [] if (buckets.active and world > 1) else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    th0 = _thread_cpu()
    c0 = time.process_time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = timed_step()
    t_host = time.perf_counter() - t0          # the host's share: all K steps enqueued (nothing in a step waits for the device)
    _trace("timed steps enqueued")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cpu_ms = (time.process_time() - c0) / args.steps * 1e3          # CPU time of this process (all threads) per step over the timed region, waits included
    th1 = _thread_cpu()
    cpu_threads = sorted(((round((th1[t][1] - th0.get(t, (None, 0.0))[1]) / args.steps * 1e3, 3), th1[t][0]) for t in th1), reverse=True)[:4]
    # the host's OWN share of a step: the same step enqueued onto an IDLE device (synchronize before, clock stopped before anything is waited
    # for). `host_wall_ms_per_step_timed_region` below is the timed region's figure and includes waiting: the runtime's queues hold a few
    # steps' worth of launches, after that the host enqueues at the pace the device retires them -- whenever the device is the slower side
    # that figure tends to the device's time per step, whatever the host needs.
    idle = []
    for _ in range(8):          # (every rank runs the same eight steps: their collectives match)
        torch.cuda.synchronize()
        ti = time.perf_counter()
        timed_step()
        idle.append(time.perf_counter() - ti)
    torch.cuda.synchronize()
    host_idle_ms = sorted(idle)[len(idle) // 2] * 1e3 if idle else None
    if world > 1:
        ti_all = torch.tensor([host_idle_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(ti_all, op=dist.ReduceOp.MAX)          # the slowest rank's host
        host_idle_ms = float(ti_all)
    # sustained clocks: the timed region is a burst of `steps` x ~15 ms from a cool chip; the same step for a few seconds more, timed over its
    # tail (the driver line carries both figures). N = 1 only: no collective may differ between ranks' loop counts.
    sustained = None
    if world == 1 and args.sustain_steps > 0:
        k_sus = args.sustain_steps
        for _ in range(k_sus // 3):
            timed_step()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(k_sus - k_sus // 3):
            timed_step()
        torch.cuda.synchronize()
        ms_sus = (time.perf_counter() - ts) / (k_sus - k_sus // 3) * 1e3
        sustained = {"images_per_sec": round(2e3 / ms_sus, 2), "ms_per_step": round(ms_sus, 3), "steps_timed": k_sus - k_sus // 3,
                     "steps_before": args.warmup + args.steps + len(idle) + k_sus // 3}
    exposed_ms = 0.0
    if buckets.exposed_events:
        exposed_ms = sum(a.elapsed_time(b) for a, b in buckets.exposed_events) / args.steps
    per_bucket = None
    if buckets.exposed_per_bucket:
        acc = {}
        for tags, e0, marks in buckets.exposed_per_bucket:
            prev = e0
            for tg, mk in zip(tags, marks):
                a = acc.setdefault(tg, [0.0, 0])
                a[0] += prev.elapsed_time(mk)
                a[1] += 1
                prev = mk
        per_bucket = {tg: round(v[0] / args.steps, 4) for tg, v in acc.items()}          # ms per step the compute stream waited for this tag's buckets (rank 0)
    buckets.exposed_events = buckets.exposed_per_bucket = None
    # roofline passes (outside the `value` timing: ~650 extra HIP-event records per step perturb it by a few per cent):
    # the same K steps again with a HIP-event pair around every conv launch, recorded on the launch stream --
    #  (1) on ONE stream (overlap off): an event interval is then that kernel alone -> `achieved`
    #  (2) with the production 3-stream schedule: the interval includes time shared with other streams -> `insitu`
    prof = prof_insitu = None
    dt_events = None
    if not args.no_roofline:          # every rank runs the passes (the steps contain collectives); rank 0 records events
        was = model.overlap_streams
        model.overlap_streams = False
        one_step()
        torch.cuda.synchronize()
        prof = {} if rank == 0 else None
        ops.PROFILER = prof
        t1 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        torch.cuda.synchronize()
        dt_events = time.perf_counter() - t1
        ops.PROFILER = None
        model.overlap_streams = was
        if was:
            one_step()
            torch.cuda.synchronize()
            prof_insitu = {} if rank == 0 else None
            ops.PROFILER = prof_insitu
            for _ in range(args.steps):
                one_step()
            torch.cuda.synchronize()
            ops.PROFILER = None
    if world > 1:
        dist.barrier()
    host_ms = t_host / args.steps * 1e3
    if world > 1:
        t = torch.tensor([dt, host_ms, exposed_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, host_ms, exposed_ms = (float(v) for v in t.tolist())
    assert torch.isfinite(losses).all(), f"non-finite losses {losses}"
    # the same step in fp32 parity mode (the reference's arithmetic precision), a secondary figure: N = 1 only, after everything else
    fp32_mode = bf16x3_mode = None
    if world == 1 and args.dtype == "bf16" and not args.no_roofline and not args.graph:
        model.compute_dtype = torch.float32
        one_step()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        k32 = max(2, min(5, args.steps))
        for _ in range(k32):
            l32 = one_step()
        torch.cuda.synchronize()
        ms32 = (time.perf_counter() - t2) / k32 * 1e3
        fp32_mode = {"ms_per_step": round(ms32, 2), "images_per_sec": round(2e3 / ms32, 2), "steps": k32,
                     "note": "same workload with fp32 activations / weights / MFMA-free fp32 kernels (the parity mode the full-size tests "
                             "run against the oracle); not the headline"}
        assert torch.isfinite(l32).all()
        # ... and in the parity-GRADE fast mode: split-bf16 operands on the bf16 MFMA kernels (compute_mode "bf16x3", csrc/split.hip): the
        # full-size tests hold it to the fp32 mode's bar (losses 1e-4, index decisions exact, gradients 2e-3)
        model.compute_mode = "bf16x3"
        for _ in range(2):
            one_step()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        kx = max(3, min(10, args.steps))
        for _ in range(kx):
            lx = one_step()
        torch.cuda.synchronize()
        msx = (time.perf_counter() - t3) / kx * 1e3
        assert torch.isfinite(lx).all()
        was = model.overlap_streams
        model.overlap_streams = False
        one_step()
        torch.cuda.synchronize()
        profx = {}
        ops.PROFILER = profx
        for _ in range(3):
            one_step()
        torch.cuda.synchronize()
        ops.PROFILER = None
        model.overlap_streams = was
        evx = profx.get("conv_igemm256") or []
        totx = sum(e[0].elapsed_time(e[1]) for e in evx) or 1.0
        issued = sum(e[2] for e in evx) / totx / 1e9
        bf16x3_mode = {"ms_per_step": round(msx, 2), "images_per_sec": round(2e3 / msx, 2), "steps": kx,
                       "vs_fp32_mode": round(ms32 / msx, 2),
                       "roofline": {"kernel": "conv_igemm256_p8_kernel<X3> (the dominant kernel on split operands: three bf16 MFMA products per fp32 product)",
                                    "bound": "mfma", "achieved": round(issued, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s (MFMA work issued)",
                                    "frac": round(issued / MFMA_PEAK_TFLOPS, 4), "algorithmic_tflops": round(issued / 3.0, 1),
                                    "launches_per_step": len(evx) // 3, "avg_launch_us": round(totx / max(1, len(evx)) * 1e3, 2)},
                       "note": "same workload, every conv with 64-multiple channels as hi.Wh + hi.Wl + lo.Wh on split bf16 operands with fp32 "
                               "accumulation (~2^-17 per product); the parity-grade mode: tests/test_fullsize_gpu.py::test_r101_s1_fullsize_bf16x3_*"}
        model.compute_dtype = torch.bfloat16

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = 2 * world * args.steps / dt
        out = {
            "metric": "images/sec (fwd+bwd+SGD) R101-C4 VOC 600x1000 bs=2/GPU" if args.depth == 101 else
                      f"SECONDARY images/sec (fwd+bwd+SGD) R{args.depth}-C4 VOC 600x1000 bs=2/GPU" + (" (BASELINE.json config 2)" if args.depth == 50 else ""),
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic", "launch": {"graph": ("hipGraph replay" if world == 1 else "hipGraph replay (one graph per gradient-bucket stage, the bucket all-reduces launched between the replays | optimizer graph)"),
                                                                     "replay": "call-list replay (engine.ReplayedStep: the eager step's C-ABI calls recorded once, re-issued by unit_replay" + ("" if world == 1 else "; the buckets' collectives launched live between its segments") + ")",
                                                                     "eager": "eager"}[mode],
            "launch_stats": (dict(gs.stats) if gs is not None else None),
            "dist": dict(buckets.describe(), tuning=tuning, exposed_ms_per_bucket_tag=per_bucket), "build_hash": _lib.build_hash(),
            # what the host needs to enqueue one step: median of eight steps each enqueued onto an IDLE device (max over ranks; with gloo on
            # host-synchronous collectives it includes them). Until round 5 this field was the timed region's wall time per step, which includes waiting -- queue
            # back-pressure then, the replay's deliberate run-ahead bound (a sleeping poll, engine.ReplayedStep) now: that figure tends to the
            # DEVICE's time per step whenever the device is the slower side and is kept as host_wall_ms_per_step_timed_region.
            "host_enqueue_ms_per_step": round(host_idle_ms if host_idle_ms is not None else host_ms, 3),
            "host_enqueue_ms_from_idle_device": (round(host_idle_ms, 3) if host_idle_ms is not None else None),
            "host_wall_ms_per_step_timed_region": round(host_ms, 3),      # max over ranks; includes the waits described above
            "host_cpu_ms_per_step_by_thread": [{"thread": n, "ms": v} for v, n in cpu_threads if v > 0.05],
            "host_cpu_ms_per_step": round(cpu_ms, 3),      # process CPU time (user + sys, all threads) per step of the timed region (rank 0): what a rank costs the node's cores
            "sustained_images_per_sec": (sustained["images_per_sec"] if sustained else None), "sustained": sustained,
            "allreduce_exposed_ms": round(exposed_ms, 3),       # max over ranks: compute-stream time inside GradBuckets.finish() per step
            "config": {"workload": f"UniT base-training step {args.variant.upper()} (TrainerNoMeta.run_step): ResNet-{args.depth}-C4, "
                                   f"VOC split1 K=20, 2 supervised + {n_weak} weak 3x600x1000 images per GPU, 512 RoIs/image, "
                                   "two Res5 heads, RPN 12000->2000, all 8 losses, SGD momentum",
                       "images_per_gpu": 2, "global_batch": 2 * world, "parallelism": f"dp{world}",
                       "step_tflop_per_gpu": STEP_TFLOP.get((args.variant, args.depth)),
                       "step_tflops_achieved_per_gpu": round(STEP_TFLOP.get((args.variant, args.depth), 0) / (ms / 1e3), 1)},
        }
        if prof is not None and (prof.get("conv_igemm256") or prof.get("conv_igemm_dma") or prof.get("conv_igemm")):
            # dominant kernel of the step: conv_igemm256_kernel (Res5 heads + RPN conv, forward and dgrad: ~60 % of the
            # step's FLOPs, ~45 % of its kernel time)
            def rate(pr, key):
                ev = pr.get(key) or []
                if not ev:
                    return None
                tot = sum(e[0].elapsed_time(e[1]) for e in ev)      # ms
                fl = sum(e[2] for e in ev)
                return {"tflops": round(fl / tot / 1e9, 1), "launches_per_step": len(ev) // args.steps,
                        "avg_launch_us": round(tot / len(ev) * 1e3, 2), "gflop_per_launch": round(fl / len(ev) / 1e9, 2),
                        "algorithmic_bytes_per_launch": round(sum(e[3] for e in ev) / len(ev))}
            key = next(k for k in ("conv_igemm256", "conv_igemm_dma", "conv_igemm") if prof.get(k))
            r = rate(prof, key)
            traffic, tdoc = None, {}
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")     # written by tools/pmc_traffic.py from a
            if os.path.exists(tpath):                                     # rocprofv3 --pmc run of this same command
                try:
                    tdoc = json.load(open(tpath))
                    traffic = tdoc.get(key + "_kernel", {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            # the PMC passes are a separate rocprofv3 run of this command (tools/pmc_bench.sh): their numbers describe THIS build only while the
            # kernel sources have not changed since -- compared by content hash, never assumed
            traffic_stale = traffic is not None and tdoc.get("_build_hash") != _lib.build_hash()
            if traffic_stale:
                traffic = None
            traffic_other_mode = traffic is not None and args.dtype != "bf16"          # the PMC passes run the default (bf16) step: their bytes
            if traffic_other_mode:                                                     # per launch say nothing about this mode's launches
                traffic = None
            out["roofline"] = {"kernel": ("conv_igemm256_p8_kernel" if key == "conv_igemm256" else key + "_kernel") +
                                         " (implicit-GEMM conv fwd/dgrad, bf16 MFMA 16x16x32, 256x256x64 LDS-DMA tiles)",
                               "bound": "mfma", "achieved": r["tflops"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(r["tflops"] / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                               "traffic_source": (f"profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command, kernel "
                                                  f"sources {tdoc.get('_build_hash')} = this build (not collected inside this run)") if traffic else
                                                 (f"withheld: profiles/pmc_traffic.json was measured on kernel sources {tdoc.get('_build_hash')}, this build is "
                                                  f"{_lib.build_hash()} -- re-run tools/pmc_bench.sh") if traffic_stale else
                                                 ("withheld: profiles/pmc_traffic.json holds the bf16 step's launches, not this mode's" if traffic_other_mode else None),
                               "launches_per_step": r["launches_per_step"], "avg_launch_us": r["avg_launch_us"],
                               "algorithmic_gflop_per_launch": r["gflop_per_launch"],
                               "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"],
                               "measured_over": f"a second pass of the same {args.steps} steps on ONE HIP stream with a HIP-event pair "
                                                f"around every launch ({dt_events / args.steps * 1e3:.2f} ms/step in that mode)",
                               "other_conv_kernels": {k2: rate(prof, k2) for k2 in ("conv_igemm_dma", "conv_igemm", "conv_wgrad")
                                                      if k2 != key and prof.get(k2)}}
            # weight-gradient kernels: bytes past the L2 per STEP over all of their grids against the operands' size per step (x + dy once,
            # the fp32 result once) -- per step, because the PMC families and the timed regions do not cut the launches the same way
            wg = out["roofline"]["other_conv_kernels"].get("conv_wgrad")
            if wg and traffic is not None and tdoc.get(key + "_kernel", {}).get("launches"):
                pmc_steps = tdoc[key + "_kernel"]["launches"] / r["launches_per_step"]
                tot = sum(v["launches"] * v["hbm_bytes_per_launch"] for kk, v in tdoc.items() if isinstance(v, dict) and kk.startswith(("conv_wgrad", "void conv_wgrad")))
                wg["algorithmic_bytes_per_step"] = wg["algorithmic_bytes_per_launch"] * wg["launches_per_step"]
                wg["pmc_bytes_per_step"] = round(tot / pmc_steps)
                wg["traffic_ratio"] = round(wg["pmc_bytes_per_step"] / max(1, wg["algorithmic_bytes_per_step"]), 3)
            # the HBM- / latency-bound kernels north_star names: RoIAlign (GB/s on algorithmic bytes = res4 maps once + pooled
            # tensor once, strided 7x7 bins; bytes past the L2 from the same PMC file) and the proposal chain (us per launch)
            def hbm(pr, k2, pmc_key):
                ev = pr.get(k2) or []
                if not ev:
                    return None
                tot = sum(e[0].elapsed_time(e[1]) for e in ev)
                byt = sum(e[3] for e in ev) / len(ev)
                us = tot / len(ev) * 1e3
                pm = {} if traffic_stale else next((v for kk, v in tdoc.items() if isinstance(v, dict) and pmc_key in kk), {})
                ref = sum((e[4] or 0) for e in ev) / len(ev)
                return {"launches_per_step": len(ev) // args.steps, "avg_launch_us": round(us, 1), "algorithmic_bytes_per_launch": round(byt),
                        "achieved_gbs": round(byt / us / 1e3, 1), "frac_of_hbm_peak": round(byt / us / 1e3 / HBM_PEAK_GBS, 4),
                        "pmc_bytes_per_launch": pm.get("hbm_bytes_per_launch"),
                        # the bytes the reference's full 14x14 ROIAlignV2 grid would move for the same RoIs, and the rate that is
                        "reference_equivalent_bytes_per_launch": round(ref) if ref else None,
                        "reference_equivalent_gbs": round(ref / us / 1e3, 1) if ref else None}
            out["roofline"]["hbm_kernels"] = {"peak_gbs": HBM_PEAK_GBS, "roi_align_fwd": hbm(prof, "roi_align_fwd", "roi_align_fwd"),
                                              "roi_align_bwd_gather": hbm(prof, "roi_align_bwd_gather", "roi_align_bwd_gather")}
            # backbone convs (north_star: >= 60 % MFMA): achieved fraction of the dense bf16 peak over every forward / dgrad launch of the
            # backbone + RPN-dgrad family (loader / consumer and 4-wave LDS-DMA kernels), and the CEILING the per-CU operand feed puts on it:
            # a launch cannot end before its busiest CU has staged its tiles' operand bytes -- tiles per CU x k-steps x (BM + BN) x 128 B --
            # at the rate a CU takes LDS-DMA bytes in inside a conv loop (FEED_B_PER_CLK, tools/feed_matrix.hip)
            bb = prof.get("conv_igemm_dma") or []
            if bb:
                t_meas = sum(e[0].elapsed_time(e[1]) for e in bb) * 1e-3
                fl = sum(e[2] for e in bb)
                t_mfma = [e[2] / (MFMA_PEAK_TFLOPS * 1e12) for e in bb]
                t_feed = [(e[4] or 0.0) / (FEED_B_PER_CLK * FEED_CLOCK_HZ) for e in bb]
                out["roofline"]["backbone"] = {
                    "launches_per_step": len(bb) // args.steps, "ms_per_step_single_stream": round(t_meas / args.steps * 1e3, 3),
                    "backbone_frac": round(fl / t_meas / 1e12 / MFMA_PEAK_TFLOPS, 4),
                    "backbone_ceiling": round(sum(t_mfma) / sum(max(a, b) for a, b in zip(t_mfma, t_feed)), 4),
                    "ceiling_ms_per_step": round(sum(max(a, b) for a, b in zip(t_mfma, t_feed)) / args.steps * 1e3, 3),
                    "feed_model": f"{FEED_B_PER_CLK} B/clk/CU at {FEED_CLOCK_HZ / 1e9:.1f} GHz through LDS-DMA inside a conv loop; ceiling = sum of MFMA-peak times / "
                                  "sum of max(MFMA-peak time, busiest CU's staged bytes / feed rate) over the family's launches"}
            out["roofline"]["latency_kernels_us"] = {k2: round(sum(e[0].elapsed_time(e[1]) for e in prof[k2]) / len(prof[k2]) * 1e3, 1)
                                                     for k2 in ("sort_topk", "nms") if prof.get(k2)}
            if prof_insitu:
                out["roofline"]["insitu"] = {k2: rate(prof_insitu, k2) for k2 in ("conv_igemm256", "conv_igemm_dma", "conv_igemm", "conv_wgrad")
                                             if prof_insitu.get(k2)}
                out["roofline"]["insitu"]["note"] = ("same events under the production schedule (Res5 heads on two streams, wgrad on a "
                                                     "third): an interval includes the time the launch shares the chip")
        if fp32_mode is not None:
            out["fp32_mode"] = fp32_mode
        if bf16x3_mode is not None:
            out["bf16x3_mode"] = bf16x3_mode
        if not args.no_cpu_baseline and world == 1:
            # bounded: the oracle runs in a child process with a wall-clock limit (never part of the timed region)
            import subprocess
            code = ("import sys, json; sys.argv=['bench.py']; sys.path.insert(0, %r); import bench; "
                    "print('CPU_BASELINE ' + json.dumps(bench.cpu_baseline(%d, %r)))" % (ROOT, args.depth, args.variant))
            try:
                env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
                env.pop("OMP_NUM_THREADS", None)          # the child takes every core of its affinity set
                r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, env=env)
                line = [l for l in r.stdout.splitlines() if l.startswith("CPU_BASELINE ")]
                out["cpu_baseline"] = json.loads(line[-1][len("CPU_BASELINE "):]) if line else {"error": r.stderr[-300:]}
            except subprocess.TimeoutExpired:
                out["cpu_baseline"] = {"error": "oracle sample exceeded the 240 s bound on this host"}
        _trace("printing the line")
        emit(json.dumps(out))
    buckets.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    _trace("done")
    return 0


if __name__ == "__main__":
    sys.exit(main())
