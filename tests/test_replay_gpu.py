"""The training step re-issued from a recorded call list (engine.ReplayedStep, csrc/replay.hip) against the eager step it was recorded
from: the list holds the eager step's own launches on the eager step's own streams, so losses of every iteration and the weights after
several iterations with changing data must agree BIT FOR BIT; sampling permutations and the learning rate advance on every replay; keys
first seen later get their own lists; the recorded step's memory is never handed to anyone else."""
import pytest
import torch

from unit_amd import _lib, config, engine
from unit_amd.modeling import build_model
from unit_amd.solver import FlatSGD
from unit_amd.synthetic import init_synthetic_weights, synthetic_batch

pytestmark = pytest.mark.gpu


def _setup(seed=3, mode="bf16"):
    cfg = config.voc_rcnn_c4_split1(50)
    cfg.MODEL.DEVICE = "cuda"
    cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 32
    cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN = 600, 100
    cfg.SOLVER.WARMUP_ITERS = 4           # the learning rate changes on every one of the test's iterations
    cfg.SEED = seed
    model = build_model(cfg)
    init_synthetic_weights(model, seed=1)
    model.train()
    model.compute_mode = mode
    return cfg, model


def _eager(seq, mode="bf16"):
    cfg, m = _setup(mode=mode)
    o = FlatSGD(m, cfg)
    ref = []
    for d in seq:
        b = m.pack_batch(*d, gt_buckets=engine.GraphedStep.GT_BUCKETS)
        o._bind()
        o.use_device_lr(m.device)
        step = m.forward_train(b, early_backward=True)
        m.backward_train(step)
        o.step()
        ref.append(step.losses.clone())
    torch.cuda.synchronize()
    return m, o, ref


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_replayed_step_equals_eager_step(dev, mode):
    data = [synthetic_batch(2, 2, hw=(128, 192), seed=50 + i, max_gt=4) for i in range(3)]
    seq = [data[i] for i in (0, 1, 2, 1, 0, 2, 2, 0)]
    m1, o1, ref = _eager(seq, mode)
    cfg, m2 = _setup(mode=mode)
    o2 = FlatSGD(m2, cfg)
    rs = engine.ReplayedStep(m2, o2, warmup_steps=2)
    n0 = _lib.LAUNCHES[0]
    got = []
    for d in seq:
        got.append(rs.run(*d).clone())
        junk = [torch.full((1 << 18,), float("nan"), device=dev) for _ in range(8)]          # churn what the allocator would hand out again
        del junk
    torch.cuda.synchronize()
    assert rs.stats == {"eager": 2, "captured": 1, "replayed": 5} and len(rs.plans) == 1 and o2.iter == o1.iter == len(seq)
    plan = next(iter(rs.plans.values()))[0]
    assert plan.n_calls > 200 and _lib.LAUNCHES[0] - n0 > 7 * plan.n_calls          # a replay counts its launches like an eager step
    names = [n for it in plan.items if it[0] == "calls" for n in it[3]]
    assert "unit_event_record_raw" in names and "unit_stream_wait_event_raw" in names and "unit_sgd_momentum" in names
    for k, (a, b) in enumerate(zip(got, ref)):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b), (k, a.tolist(), b.tolist())
    assert torch.equal(m2.store.params, m1.store.params)
    assert not torch.equal(got[1], got[3])          # same data, later weights / other permutations: the replay is not a recording of values
    # the trainer switch
    cfg, m3 = _setup(mode=mode)
    tr = engine.TrainerNoMeta(cfg, m3, use_replay=True)
    for d in seq:
        tr.run_step(*d)
    torch.cuda.synchronize()
    assert isinstance(tr.graphed, engine.ReplayedStep) and tr.graphed.stats["replayed"] == 5
    assert torch.equal(m3.store.params, m1.store.params)


def test_replayed_step_new_keys_after_warmup(dev):
    """image sizes and ground-truth counts change from batch to batch: a key first seen after the warm-up runs its first step eagerly, is
    recorded at its second occurrence, and the lists of earlier keys keep replaying correctly while later recordings allocate"""
    small = [synthetic_batch(2, 2, hw=(96, 128), seed=70 + i, max_gt=3) for i in range(2)]
    big = [synthetic_batch(2, 2, hw=(160, 224), seed=80 + i, max_gt=4) for i in range(2)]
    crowded = [synthetic_batch(2, 2, hw=(96, 128), seed=sd, max_gt=40) for sd in (91, 102)]
    seq = [small[0], small[1], small[0], big[0], small[1], big[1], crowded[0], big[0], crowded[1], small[0], crowded[0], big[1], small[1]]
    m1, o1, ref = _eager(seq)
    cfg, m2 = _setup()
    o2 = FlatSGD(m2, cfg)
    rs = engine.ReplayedStep(m2, o2, warmup_steps=2)
    got = [rs.run(*d).clone() for d in seq]
    torch.cuda.synchronize()
    assert len(rs.plans) == 3 and len(rs.seen) == 3 and rs.stats["replayed"] == 6
    for k, (a, b) in enumerate(zip(got, ref)):
        assert torch.equal(a, b), (k, a.tolist(), b.tolist())
    assert torch.equal(m2.store.params, m1.store.params)


def test_replayed_step_mixed_image_sizes_and_no_weak_batch(dev):
    """a ragged step (supervised and weak batch pad differently: pair launches with host structs in the list) and the fine-tune form of the
    step (no weak batch) replay like the plain one"""
    ragged = [(synthetic_batch(2, 0, hw=(128, 160), seed=20 + i, max_gt=3)[0], synthetic_batch(0, 2, hw=(96, 192), seed=30 + i)[1]) for i in range(2)]
    sup_only = [(synthetic_batch(2, 0, hw=(128, 160), seed=40 + i, max_gt=3)[0], None) for i in range(2)]
    seq = [ragged[0], ragged[1], ragged[0], sup_only[0], ragged[1], sup_only[1], sup_only[0], ragged[0]]
    m1, o1, ref = _eager(seq)
    cfg, m2 = _setup()
    o2 = FlatSGD(m2, cfg)
    rs = engine.ReplayedStep(m2, o2, warmup_steps=1)
    got = [rs.run(*d).clone() for d in seq]
    torch.cuda.synchronize()
    assert len(rs.plans) == 2 and rs.stats["replayed"] >= 3
    plan = rs.plans[m2.pack_batch(*ragged[0], gt_buckets=engine.GraphedStep.GT_BUCKETS).key()][0]
    assert any(n == "unit_conv2d_fwd_pair" for it in plan.items if it[0] == "calls" for n in it[3])
    for k, (a, b) in enumerate(zip(got, ref)):
        assert torch.equal(a, b), (k, a.tolist(), b.tolist())
    assert torch.equal(m2.store.params, m1.store.params)
