"""One rank of the 2-rank data-parallel rehearsal (tests/test_dp_gpu.py starts two of these as fresh child processes; both
share cuda:0 and talk gloo through 127.0.0.1 -- the RCCL run differs only in the backend name).
usage: python tests/dp_rehearsal_worker.py <rank> <world> <port> <out.pt> [fp32|bf16_buckets|graph|graph_whole|replay|eager][_ragged] [steps]
(replay = engine.ReplayedStep: the step's recorded C-ABI call list re-issued in C, the buckets' all-reduces launched live between its segments)
(graph = one graph per gradient-bucket stage with the bucket all-reduces between the replays; graph_whole = one graph + one all-reduce;
_ragged: rank 1 meets a NEW batch key (another image size) at step 3 and again at step 5 while rank 0 keeps replaying its first key -- the ranks
then disagree about eager vs replay in those steps and must still issue the same collectives)"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rehearsal_cfg():
    from unit_amd import config
    cfg = config.voc_rcnn_c4_split1(50)
    cfg.MODEL.DEVICE = "cuda"
    cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 32
    cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN = 600, 100
    cfg.SEED = 11
    return cfg


def global_batch(hw=(128, 192)):
    from unit_amd.synthetic import synthetic_batch
    return synthetic_batch(4, 4, hw=hw, seed=5, max_gt=4)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unit_amd import engine
    from unit_amd.modeling import build_model
    from unit_amd.synthetic import init_synthetic_weights
    cfg = rehearsal_cfg()
    model = build_model(cfg)
    init_synthetic_weights(model, seed=1 + rank)          # ranks start DIFFERENT: the initial broadcast must equalise them
    model.train()
    model.compute_dtype = torch.float32
    mode = sys.argv[5] if len(sys.argv) > 5 else "fp32"
    ragged = mode.endswith("_ragged")
    mode = mode[:-len("_ragged")] if ragged else mode
    steps = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    tr = engine.TrainerNoMeta(cfg, model, bf16_buckets=(mode == "bf16_buckets"), use_graph=mode in ("graph", "graph_whole"),
                              graph_per_bucket=(mode == "graph"), use_replay=(mode == "replay"))
    if mode == "eager":          # the graphed trainer's packing capacity and device-resident learning rate, launched eagerly
        tr.optimizer._bind()
    sup, weak = global_batch()
    sup2, weak2 = global_batch(hw=(128, 160))
    for it in range(steps):
        if ragged:
            sup, weak = (sup2, weak2) if (rank == 1 and it in (3, 5)) else global_batch()
        if mode == "eager":
            tr.optimizer.use_device_lr(model.device)
            batch = model.pack_batch(engine.shard_batch(sup, rank, world), engine.shard_batch(weak, rank, world),
                                     gt_buckets=engine.GraphedStep.GT_BUCKETS)
            step = model.forward_train(batch, early_backward=True)
            model.backward_train(step)
            tr.buckets.finish()
            tr.optimizer.step()
            losses = step.losses
        else:
            losses = tr.run_step(engine.shard_batch(sup, rank, world), engine.shard_batch(weak, rank, world))
    torch.cuda.synchronize()
    nseg = 0
    if tr.graphed is not None and mode != "replay" and tr.graphed.graphs:
        g = next(iter(tr.graphed.graphs.values()))[0]
        nseg = len(g[0]) if (isinstance(g, tuple) and isinstance(g[0], list)) else 1
    stats, py_items = None, 0
    if mode == "replay":
        stats = dict(tr.graphed.stats)
        py_items = max(sum(1 for it in ent[0].items if it[0] == "py") for ent in tr.graphed.plans.values()) if tr.graphed.plans else 0
    torch.save({"params": model.store.params.cpu(), "losses": losses.cpu(), "graph_segments": nseg, "replay_stats": stats, "replay_py_items": py_items}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
