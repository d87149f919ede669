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


def test_shard_batch():
    g = list(range(16))
    parts = [shard_batch(g, r, 8) for r in range(8)]
    assert parts[0] == [0, 1] and parts[7] == [14, 15]
    assert sorted(sum(parts, [])) == g
