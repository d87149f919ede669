"""bench.py's own N-rank launcher (`python bench.py --gpus N` without a torchrun around it), on CPU: the parent never touches the GPU,
starts N fresh rank processes with the rendezvous environment of scripts/train_VOC.py:67-77 (detectron2 launch: one process per GPU,
tcp://127.0.0.1:port), relays rank 0's output only when every rank succeeded, ends the survivors by PID when one rank dies, and never
prints a result line whose n_gpus is not what --gpus asked for."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e)


@pytest.mark.parametrize("n", [4, 8])
def test_dry_launch_starts_n_ranks_with_one_rendezvous(n):
    """n = 8: the node's real rank count -- eight fresh processes from a parent that never touches the GPU"""
    r = _run(["--gpus", str(n), "--dry-launch"])
    assert r.returncode == 0, r.stderr
    envs = [json.loads(l[len("DRY_LAUNCH "):]) for l in r.stdout.splitlines() if l.startswith("DRY_LAUNCH ")]
    assert sorted(int(e["RANK"]) for e in envs) == list(range(n))
    assert all(e["RANK"] == e["LOCAL_RANK"] for e in envs)          # one node: rank r drives GPU r
    assert {e["WORLD_SIZE"] for e in envs} == {str(n)} and {e["MASTER_ADDR"] for e in envs} == {"127.0.0.1"}
    assert len({e["MASTER_PORT"] for e in envs}) == 1
    assert {e["HSA_ENABLE_IPC_MODE_LEGACY"] for e in envs} == {"0"}          # dmabuf IPC for RCCL across processes


def test_a_dead_rank_ends_the_run_non_zero_without_a_result_line():
    t0 = time.time()
    r = _run(["--gpus", "3", "--dry-launch"], env={"UNIT_DRY_FAIL_RANK": "1"})
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert time.time() - t0 < 120          # the surviving ranks (asleep, as in a collective) were terminated, not waited for
    assert r.stdout == ""
    assert "no result line" in r.stderr


def test_more_ranks_than_gpus_fails_loudly():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], env={"CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""})
    assert r.returncode == 2
    assert r.stdout == "" and "needs 2 visible GPUs" in r.stderr


def test_world_size_from_a_launcher_must_equal_gpus():
    r = _run(["--gpus", "8"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and r.stdout == ""
    assert "WORLD_SIZE=2" in r.stderr


def test_under_torch_distributed_run_the_ranks_run_main_directly():
    """the driver's N > 1 command: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` -- WORLD_SIZE comes from the launcher,
    bench.py must not start ranks of its own, and every rank sees the launcher's rendezvous"""
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", BENCH, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    envs = [json.loads(l[len("DRY_LAUNCH "):]) for l in r.stdout.splitlines() if l.startswith("DRY_LAUNCH ")]
    assert sorted(int(x["RANK"]) for x in envs) == [0, 1] and {x["MASTER_PORT"] for x in envs} == {"29517"}


def test_exchange_selection_and_gpu_count_without_hip():
    """the N > 1 self-tuning run (bench.py: 8-step probes of reduce mode x bucket size x collective stream): the selection rule, and the
    launcher's device count, which must not initialise the HIP runtime in the parent (sysfs / the *_VISIBLE_DEVICES lists only)"""
    sys.path.insert(0, ROOT)
    import bench
    table = [{"mode": "allreduce", "bucket_mb": 64.0, "stream": "rpn", "ms": 21.5, "error": None},
             {"mode": "rs_ag", "bucket_mb": 25.0, "stream": "own", "ms": None, "error": "RuntimeError: no reduce_scatter in this backend"},
             {"mode": "direct", "bucket_mb": 25.0, "stream": "own", "ms": 19.25, "error": None},
             {"mode": "direct", "bucket_mb": 64.0, "stream": "rpn", "ms": 19.25, "error": None},
             {"mode": "allreduce", "bucket_mb": 25.0, "stream": "own", "ms": 30.0, "error": None}]
    best = bench.pick_exchange(table)
    assert (best["mode"], best["bucket_mb"], best["stream"]) == ("direct", 25.0, "own")          # fastest; a tie goes to the table's order
    assert bench.pick_exchange([table[1]]) is None and bench.pick_exchange([]) is None
    src = open(BENCH).read()
    body = src[src.index("def launch_ranks"):src.index("def cpu_baseline")]
    assert "torch.cuda" not in body          # the parent makes no torch.cuda call at all
    old = {k: os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")}
    try:
        os.environ["HIP_VISIBLE_DEVICES"] = ""
        n = bench.visible_gpu_count()
        assert n is None or n == 0
        os.environ["HIP_VISIBLE_DEVICES"] = "0,1,2"
        n = bench.visible_gpu_count()
        assert n is None or 0 <= n <= 3
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
