"""RCCL with ONE rank, every collective of the data-parallel step forced on (tests/test_rccl_gpu.py starts this as a fresh child process).

The pool's boxes have one GPU, so RCCL never sees a second rank here -- but everything an N-GPU run does on the device side except
the wire does happen: ProcessGroupNCCL initialisation with `device_id`, the initial broadcasts, every bucket's exchange launched
asynchronously from the weight-gradient stream in the middle of the backward, RCCL's own stream ordered behind it, `finish()` making
the optimizer's stream wait, the bf16 bucket cast / widen, the "rs_ag" in-place pair and the "direct" chain (all-to-all -> unit_shard_sum
on the communication stream -> all-gather), and the graph modes that cut the capture at bucket boundaries. A sum over one rank is the
identity, so with fp32 buckets the parameters after K steps must be BIT-equal to the plain single-process run.
usage: python tests/rccl_world1_worker.py <port> <out.json>"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

STEPS = 4


def main():
    port, out = sys.argv[1], sys.argv[2]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import dp_rehearsal_worker as W
    from unit_amd import engine
    from unit_amd.modeling import build_model
    from unit_amd.synthetic import init_synthetic_weights
    cfg = W.rehearsal_cfg()
    sup, weak = W.global_batch()
    sup, weak = sup[:2], weak[:2]

    def run(dtype, **kw):
        model = build_model(cfg)
        init_synthetic_weights(model, seed=1)
        model.train()
        model.compute_dtype = dtype
        tr = engine.TrainerNoMeta(cfg, model, **kw)
        first = None
        for it in range(STEPS):
            losses = tr.run_step(sup, weak)
            if it == 0:
                model.join_optimizer_tail()
                first = model.store.params.clone()
        model.join_optimizer_tail()
        torch.cuda.synchronize()
        nseg = 0
        if isinstance(tr.graphed, engine.ReplayedStep):
            assert tr.graphed.stats["replayed"] >= 1, tr.graphed.stats
            nseg = max(sum(1 for it in ent[0].items if it[0] == "calls") for ent in tr.graphed.plans.values())
        elif tr.graphed is not None and tr.graphed.graphs:
            g = next(iter(tr.graphed.graphs.values()))[0]
            nseg = len(g[0]) if (isinstance(g, tuple) and isinstance(g[0], list)) else 1
        return model.store.params.clone(), losses.clone(), tr.buckets, nseg, first

    res = {"backend": dist.get_backend(), "world": dist.get_world_size(), "cases": {}}
    for dtype, dname in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        os.environ["UNIT_FORCE_COLLECTIVES"] = "0"
        ref_p, ref_l, b0, _, ref_first = run(dtype)
        assert not b0.active and b0.launched == 0
        os.environ["UNIT_FORCE_COLLECTIVES"] = "1"
        cases = [("allreduce", {}), ("rs_ag", dict(reduce_mode="rs_ag", bucket_bytes=3 << 20)), ("direct", dict(reduce_mode="direct", bucket_bytes=5 << 20)),
                 ("cabi", dict(reduce_mode="cabi", bucket_bytes=7 << 20)),          # RCCL through the library's own comm exports (csrc/comm.hip)
                 ("direct_bf16_buckets", dict(reduce_mode="direct", bf16_buckets=True)), ("allreduce_bf16_buckets", dict(bf16_buckets=True))]
        # the call-list replay (engine.ReplayedStep): the recorded step's launches re-issued in C, every bucket's RCCL exchange launched live
        # between two segments of the list, from the weight-gradient stream, where the eager backward launches it
        cases += [("replay", dict(use_replay=True)), ("replay_direct", dict(use_replay=True, reduce_mode="direct", bucket_bytes=5 << 20)),
                  ("replay_cabi", dict(use_replay=True, reduce_mode="cabi", bucket_bytes=7 << 20))]
        if dname == "fp32":
            cases += [("graph_per_bucket", dict(use_graph=True, graph_per_bucket=True)), ("graph_whole", dict(use_graph=True, graph_per_bucket=False)),
                      ("graph_per_bucket_direct", dict(use_graph=True, graph_per_bucket=True, reduce_mode="direct")),
                      ("tail_overlap", dict(overlap_tail=True))]
        for name, kw in cases:
            p, l, b, nseg, first = run(dtype, **kw)
            d = b.describe()
            res["cases"][f"{dname}/{name}"] = {
                "bit_equal": bool(torch.equal(p, ref_p)), "max_abs_diff": float((p - ref_p).abs().max()),
                "max_abs_diff_step1": float((first - ref_first).abs().max()), "loss_diff": float((l - ref_l).abs().max()),
                "launched": b.launched, "describe": d, "graph_segments": nseg, "finite": bool(torch.isfinite(p).all())}
    # the comm exports on their own: id -> communicator -> in-place sum of a bucket on a side stream -> event hand-off -> destroy
    from unit_amd import parallel, _lib
    comm = parallel.CAbiComm(0, 1, parallel.CAbiComm.unique_id())
    side = torch.cuda.Stream()
    a = torch.randn(1 << 20, device="cuda")
    b = a.to(torch.bfloat16)
    a0, b0c = a.clone(), b.clone()
    side.wait_stream(torch.cuda.current_stream())
    comm.all_reduce_(a, side)
    comm.all_reduce_(b, side)
    parallel.CAbiComm.wait(torch.cuda.current_stream(), side)
    a.mul_(2.0)                                       # ordered behind the collective by the hand-off
    torch.cuda.synchronize()
    res["cabi"] = {"fp32_identity": bool(torch.equal(a, a0 * 2.0)), "bf16_identity": bool(torch.equal(b, b0c)),
                   "rccl_version": int(_lib.lib().unit_comm_rccl_version())}
    comm.close()
    json.dump(res, open(out, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
