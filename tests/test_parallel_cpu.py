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
    torch.set_num_threads(2 if world <= 2 else 1)
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
        lst = [None] * world
        dist.all_gather_object(lst, float(ref0.double().sum()))
        assert len(set(lst)) == world
        dist.broadcast(st.params, 0)
        dist.all_gather_object(lst, float(st.params.double().sum()))
        assert len(set(lst)) == 1
        # (2) bucketed all-reduce in backward order through the hook the backward plan calls
        st.grads.copy_(torch.arange(st.size, dtype=torch.float32) % 97 * (rank + 1))
        expect = torch.arange(st.size, dtype=torch.float32) % 97 * (world * (world + 1) / 2.0)     # (1 + 2 + ... + world): integers, exact in any order
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
        assert buckets.grad_scale == 1.0 / world
        # every gradient element is covered by exactly one bucket
        cover = torch.zeros(st.size, dtype=torch.int32)
        for chunks in buckets._plan.values():
            for a, b in chunks:
                cover[a:b] += 1
        assert int(cover.min()) == 1 and int(cover.max()) == 1
        # (3) the other exchange forms of a bucket (parallel.MODES): reduce-scatter + all-gather in place, and the "direct" form
        # (all-to-all of shard contributions, owner-side sum in rank order, all-gather) -- same sums, odd bucket sizes included
        # (a tag whose length is not a multiple of the world size leaves a tail that goes through a small all-reduce)
        for mode, mb in ((("rs_ag", 8), ("direct", 8), ("direct", 0.37), ("rs_ag", 1.3)) if world == 2 else (("direct", 25), ("direct", 0.37), ("rs_ag", 1.3))):
            b2 = GradBuckets(model, bucket_bytes=int(mb * (1 << 20)), mode=mode)
            assert b2.bucket_elems % world == 0
            st.grads.copy_(torch.arange(st.size, dtype=torch.float32) % 97 * (rank + 1))
            for tag in tags:
                model.on_grad_ready(tag)
            b2.finish()
            assert torch.equal(st.grads, expect), (mode, mb, (st.grads - expect).abs().max())
            assert b2.launched > len(tags) and b2.describe()["reduce_mode"] == mode and b2.describe()["ranks_seen"] == world
            if world > 2:          # shards: every bucket a whole number of shards, tails (< world elements at the end of a tag) only at a tag's last bucket
                tails = 0
                for tag, chunks in b2._plan.items():
                    for i, (a, b) in enumerate(chunks):
                        assert (b - a) % world == 0 or i == len(chunks) - 1, (tag, a, b)
                        tails += (b - a) % world != 0
                assert tails == 0          # (the flat store pads every tag to a multiple of 8 elements: the real model has no tails at 8 ranks)
        # reduce_all (one-graph mode of engine.GraphedStep): one exchange of the whole buffer, every mode
        for mode in ("allreduce", "rs_ag", "direct"):
            b3 = GradBuckets(model, mode=mode)
            st.grads.copy_(torch.arange(st.size, dtype=torch.float32) % 97 * (rank + 1))
            b3.reduce_all()
            assert torch.equal(st.grads, expect), mode
        # (3b) tails: a store whose tags are NOT multiples of the world size (another padding, another world size): the last bucket of a tag
        # carries < world leftover elements through a small all-reduce, buckets smaller than the world size go through all-reduce whole
        class _Store:
            tags = [("a", 0, 1003), ("b", 1003, 1080), ("c", 1080, 1085), ("d", 1085, 1085 + 8 * 40)]
            size = 1085 + 8 * 40
            grads = torch.zeros(1085 + 8 * 40)

        class _Model:
            store = _Store()
            on_grad_ready = None

        fake = _Model()
        pat = torch.arange(_Store.size, dtype=torch.float32) % 89
        for mode in ("allreduce", "rs_ag", "direct"):
            bf = GradBuckets(fake, bucket_bytes=4 * 128, mode=mode)
            assert bf.bucket_elems == 128 // world * world
            _Store.grads.copy_(pat * (rank + 1))
            for tag, _, _ in _Store.tags:
                fake.on_grad_ready(tag)
            bf.finish()
            assert torch.equal(_Store.grads, pat * (world * (world + 1) / 2.0)), mode
            n_tail = sum((b - a) % world != 0 for ch in bf._plan.values() for a, b in ch)
            assert n_tail == (3 if world == 8 else 3), (n_tail, world)          # tags a (1003), b (77), c (5) end off a multiple of 2 and of 8
        # (4) the module-level training surface under a foreign trainer (parallel.allreduce_module_grads: what replaces DDP there)
        from unit_amd.parallel import allreduce_module_grads
        lin = torch.nn.Linear(5, 3)
        frozen = torch.nn.Linear(2, 2)
        for prm in frozen.parameters():
            prm.requires_grad = False
        mod = torch.nn.Sequential(lin, frozen)
        lin.weight.grad = torch.full((3, 5), float(rank + 1))
        lin.bias.grad = torch.full((3,), 2.0 * (rank + 1))
        assert allreduce_module_grads(mod) == 2
        mean = (world + 1) / 2.0
        assert torch.equal(lin.weight.grad, torch.full((3, 5), mean)) and torch.equal(lin.bias.grad, torch.full((3,), 2.0 * mean))
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


def test_grad_buckets_gloo_world8():
    """the node's real rank count: eight gloo ranks over the R50 model's flat buffer -- the bucket plan cut into whole shards
    of eight, tags whose length is not a multiple of eight (tail all-reduce), `direct` (all-to-all + ordered owner-side sum + all-gather) and
    `rs_ag` on odd bucket sizes, one exchange of the whole buffer, the parameter broadcast from rank 0"""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
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
