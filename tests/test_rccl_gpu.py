"""RCCL itself in the step: the real "nccl" backend, one rank, collectives forced on (UNIT_FORCE_COLLECTIVES=1 disables the
world == 1 early-outs of parallel.GradBuckets) -- see tests/rccl_world1_worker.py for what that exercises. The reference's counterpart is
DDP's bucketed all-reduce behind engine/defaults.py:256,285 and the NCCL process group of scripts/train_VOC.py:67-77."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def result(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("rccl") / "res.json")
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_world1_worker.py"), str(_free_port()), out], capture_output=True, text=True,
                       timeout=1500)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    return json.load(open(out))


def test_rccl_is_the_backend(dev, result):
    assert result["backend"] == "nccl" and result["world"] == 1
    d = result["cases"]["fp32/allreduce"]["describe"]
    assert d["backend"].startswith("rccl") and d["ranks_seen"] == 1 and d["collectives_forced_at_world_1"] and d["rccl_version"]


def test_comm_exports_of_the_c_abi(dev, result):
    """unit_comm_unique_id / unit_comm_init / unit_allreduce_bucket_async / unit_comm_wait / unit_comm_destroy (csrc/comm.hip): RCCL
    resolved by dlopen, a one-rank communicator, an fp32 and a bf16 bucket summed in place on a side stream (= the identity), the compute
    stream ordered behind them by the event hand-off; and the same binding as the step's exchange (`reduce_mode="cabi"`, below)"""
    c = result["cabi"]
    assert c["fp32_identity"] and c["bf16_identity"] and c["rccl_version"] >= 20000, c


@pytest.mark.parametrize("case", ["allreduce", "rs_ag", "direct", "cabi", "graph_per_bucket", "graph_whole", "graph_per_bucket_direct", "tail_overlap",
                                  "replay", "replay_direct", "replay_cabi"])
def test_forced_collectives_fp32_buckets_are_the_identity(dev, result, case):
    """fp32 buckets through RCCL at world 1: launched from the weight-gradient stream inside the backward, waited for by the optimizer's
    stream -- parameters after 4 steps BIT-equal to the run without any collective (a missing stream dependency would show as a torn
    gradient: the optimizer reading a bucket before its exchange wrote it back)"""
    for dname in ("fp32", "bf16"):
        c = result["cases"].get(f"{dname}/{case}")
        if c is None:
            assert dname == "bf16" and (case.startswith("graph") or case == "tail_overlap")
            continue
        assert c["launched"] >= 6, c          # 2 broadcasts + one or more collectives per gradient bucket and step
        assert c["finite"] and c["bit_equal"], (dname, case, c["max_abs_diff"])
        if case == "graph_per_bucket":
            assert c["graph_segments"] >= 4
        if case.startswith("replay"):
            assert c["graph_segments"] >= 5, c["graph_segments"]          # the list is cut at every gradient bucket: its RCCL launch sits between two segments
        if case != "allreduce":
            assert c["describe"]["reduce_mode"] == ("direct" if "direct" in case else "cabi" if "cabi" in case else case if case == "rs_ag" else "allreduce")


@pytest.mark.parametrize("case", ["allreduce_bf16_buckets", "direct_bf16_buckets"])
def test_forced_collectives_bf16_buckets(dev, result, case):
    """bf16 buckets: cast -> exchange -> widen; after ONE step every parameter is within (2^-8 relative rounding of its gradient) x lr of the
    fp32-bucket run (the bound of test_dp_gpu.py::test_two_ranks_bf16_gradient_buckets); later steps only have to stay finite -- a last-bit
    parameter difference may re-draw near-tied proposals"""
    for dname in ("fp32", "bf16"):
        c = result["cases"][f"{dname}/{case}"]
        assert c["finite"] and not c["bit_equal"]
        assert 0 < c["max_abs_diff_step1"] <= 1e-4, c
