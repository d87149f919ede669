"""Data-parallel path on CPU: world_size 2, gloo backend (127.0.0.1 rendezvous). Covers the flat-bucket gradient
all-reduce launched from the backward plan's `on_grad_ready` hook, the initial parameter broadcast, the 1/world fold
into SGD and batch sharding (data/build.py:354-355). The RCCL run on MI355X uses the same code with backend "nccl"."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from unit_amd import config
from unit_amd.engine import shard_batch


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from unit_amd.modeling import build_model
        from unit_amd.parallel import GradBuckets
        cfg = config.voc_rcnn_c4_split1(50)
        cfg.MODEL.DEVICE = "cpu"
        torch.manual_seed(100 + rank)                 # different init per rank -> broadcast must equalise
        model = build_model(cfg)
        model.train()
        st = model.flatten_parameters()
        buckets = GradBuckets(model, bucket_bytes=8 << 20)
        # (1) broadcast from rank 0
        ref0 = st.params.clone()
        lst = [None, None]
        dist.all_gather_object(lst, float(ref0.double().sum()))
        assert lst[0] != lst[1]
        dist.broadcast(st.params, 0)
        dist.all_gather_object(lst, float(st.params.double().sum()))
        assert lst[0] == lst[1]
        # (2) bucketed all-reduce in backward order through the hook the backward plan calls
        st.grads.copy_(torch.arange(st.size, dtype=torch.float32) % 97 * (rank + 1))
        expect = torch.arange(st.size, dtype=torch.float32) % 97 * 3.0     # (1 + 2)
        tags = []
        for tag, _, _ in st.tags:
            if tag not in tags:
                tags.append(tag)
        assert tags[:2] == ["heads", "box_head"] and tags[-1] == "res3"
        for tag in tags:
            model.on_grad_ready(tag)
        assert len(buckets._works) >= len(st.tags)          # res4/res5 ranges are split into several buckets
        buckets.finish()
        assert torch.equal(st.grads, expect)
        assert buckets.grad_scale == 0.5
        # every gradient element is covered by exactly one bucket
        cover = torch.zeros(st.size, dtype=torch.int32)
        for chunks in buckets._plan.values():
            for a, b in chunks:
                cover[a:b] += 1
        assert int(cover.min()) == 1 and int(cover.max()) == 1
        # (3) the other exchange forms of a bucket (parallel.MODES): reduce-scatter + all-gather in place, and the "direct" form
        # (all-to-all of shard contributions, owner-side sum in rank order, all-gather) -- same sums, odd bucket sizes included
        # (a tag whose length is not a multiple of the world size leaves a tail that goes through a small all-reduce)
        for mode, mb in (("rs_ag", 8), ("direct", 8), ("direct", 0.37), ("rs_ag", 1.3)):
            b2 = GradBuckets(model, bucket_bytes=int(mb * (1 << 20)), mode=mode)
            assert b2.bucket_elems % world == 0
            st.grads.copy_(torch.arange(st.size, dtype=torch.float32) % 97 * (rank + 1))
            for tag in tags:
                model.on_grad_ready(tag)
            b2.finish()
            assert torch.equal(st.grads, expect), (mode, mb, (st.grads - expect).abs().max())
            assert b2.launched > len(tags) and b2.describe()["reduce_mode"] == mode and b2.describe()["ranks_seen"] == 2
        # reduce_all (one-graph mode of engine.GraphedStep): one exchange of the whole buffer, every mode
        for mode in ("allreduce", "rs_ag", "direct"):
            b3 = GradBuckets(model, mode=mode)
            st.grads.copy_(torch.arange(st.size, dtype=torch.float32) % 97 * (rank + 1))
            b3.reduce_all()
            assert torch.equal(st.grads, expect), mode
        q.put((rank, "ok"))
    except Exception as e:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_grad_buckets_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_grad_buckets_knobs_from_env(monkeypatch):
    """bucket size / exchange form come from the environment when the caller does not choose (bench.py --bucket-mb / --reduce-mode set them
    explicitly); forcing collectives without a process group is an error, not a silent no-op"""
    import pytest
    from unit_amd.parallel import GradBuckets

    class M:
        on_grad_ready = "untouched"

    monkeypatch.setenv("UNIT_BUCKET_MB", "16")
    monkeypatch.setenv("UNIT_REDUCE_MODE", "direct")
    b = GradBuckets(M())
    assert b.bucket_bytes == 16 << 20 and b.mode == "direct" and not b.active and b.model.on_grad_ready is None
    assert b.describe()["backend"] is None and b.describe()["ranks_seen"] == 1
    with pytest.raises(ValueError):
        GradBuckets(M(), mode="ring")
    with pytest.raises(RuntimeError):
        GradBuckets(M(), force=True)
    # where the collectives are enqueued: the library's stream by default, torch's internal one on request, anything else is an error; host
    # tensors (these gloo tests) never get a HIP stream
    import torch
    assert b.collective_stream == "rpn" and b.describe()["collective_stream"] == "rpn"
    assert b._collective_stream(torch.device("cpu")) is None
    monkeypatch.setenv("UNIT_COLLECTIVE_STREAM", "internal")
    assert GradBuckets(M()).collective_stream == "internal"
    monkeypatch.setenv("UNIT_COLLECTIVE_STREAM", "somewhere")
    with pytest.raises(ValueError):
        GradBuckets(M())


def test_shard_batch():
    g = list(range(16))
    parts = [shard_batch(g, r, 8) for r in range(8)]
    assert parts[0] == [0, 1] and parts[7] == [14, 15]
    assert sorted(sum(parts, [])) == g
