"""The whole training step captured in a hipGraph (engine.GraphedStep) against the eager step: same losses every iteration, same
parameters after several iterations with changing data, fresh sampling permutations on every replay, learning-rate schedule
followed without re-capture."""
import pytest
import torch

from unit_amd import config, engine
from unit_amd.modeling import build_model
from unit_amd.solver import FlatSGD
from unit_amd.synthetic import init_synthetic_weights, synthetic_batch

pytestmark = pytest.mark.gpu


def _setup(seed=3):
    cfg = config.voc_rcnn_c4_split1(50)
    cfg.MODEL.DEVICE = "cuda"
    cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 32
    cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN = 600, 100
    cfg.SOLVER.WARMUP_ITERS = 4           # the learning rate changes on every one of the test's iterations
    cfg.SEED = seed
    model = build_model(cfg)
    init_synthetic_weights(model, seed=1)
    model.train()
    return cfg, model


def test_graphed_step_equals_eager_step(dev):
    data = [synthetic_batch(2, 2, hw=(128, 192), seed=50 + i, max_gt=4) for i in range(3)]
    order = [0, 1, 2, 1, 0, 2]
    # eager reference: same packing capacity, same device-resident learning rate
    cfg, m1 = _setup()
    o1 = FlatSGD(m1, cfg)
    ref_losses = []
    for i in order:
        b = m1.pack_batch(*data[i], gt_buckets=engine.GraphedStep.GT_BUCKETS)
        o1._bind()
        o1.use_device_lr(m1.device)
        step = m1.forward_train(b, early_backward=True)
        m1.backward_train(step)
        o1.step()
        ref_losses.append(step.losses.clone())
    torch.cuda.synchronize()
    # graphed: two eager warm-up iterations, capture at the third, replays afterwards (with OTHER data than at capture)
    cfg, m2 = _setup()
    o2 = FlatSGD(m2, cfg)
    gs = engine.GraphedStep(m2, o2, warmup_steps=2)
    got = []
    for i in order:
        got.append(gs.run(*data[i]).clone())
    torch.cuda.synchronize()
    assert len(gs.graphs) == 1 and o2.iter == o1.iter == len(order)
    # The eager steps run the decoupled backward plan (rcnn.py `decoupled_sup_chain` / `early_sup_backward`: the weak head's feature GEMM over
    # two row subsets, the supervised predictor backward on the head stream), captured steps the single-GEMM plan: rows of a GEMM do not
    # depend on which launch computed them, so the two plans must agree BIT FOR BIT -- losses of every iteration and the weights after six.
    for k, (a, b) in enumerate(zip(got, ref_losses)):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b), (k, a.tolist(), b.tolist())
    assert torch.equal(m2.store.params, m1.store.params)
    assert not torch.equal(got[1], got[3])          # same data, later weights / other permutations: the replay is not a recording
    # the trainer switch
    cfg, m3 = _setup()
    tr = engine.TrainerNoMeta(cfg, m3, use_graph=True)
    for i in order:
        l3 = tr.run_step(*data[i])
    torch.cuda.synchronize()
    assert tr.graphed is not None and len(tr.graphed.graphs) == 1
    assert torch.allclose(m3.store.params, m1.store.params, rtol=1e-5, atol=1e-7)


def test_graphed_step_new_keys_after_warmup(dev):
    """image sizes and ground-truth counts change from batch to batch in real training: a key first seen AFTER the warm-up steps (other
    image size; an image with more boxes than the 32-slot bucket) runs its first step eagerly -- its host-built constants are uploaded
    and its workspaces sized outside any capture -- is captured at its second occurrence, and the graphs of the earlier keys keep
    replaying correctly afterwards (workspaces that grew meanwhile are never freed). Losses and parameters equal the eager run."""
    small = [synthetic_batch(2, 2, hw=(96, 128), seed=70 + i, max_gt=3) for i in range(2)]
    big = [synthetic_batch(2, 2, hw=(160, 224), seed=80 + i, max_gt=4) for i in range(2)]
    crowded = [synthetic_batch(2, 2, hw=(96, 128), seed=sd, max_gt=40) for sd in (91, 102)]
    assert all(max(len(x["instances"].gt_classes) for x in c[0]) > 32 for c in crowded)
    seq = [small[0], small[1], small[0], big[0], small[1], big[1], crowded[0], big[0], crowded[1], small[0], crowded[0], big[1]]
    cfg, m1 = _setup()
    o1 = FlatSGD(m1, cfg)
    ref = []
    for d in seq:
        b = m1.pack_batch(*d, gt_buckets=engine.GraphedStep.GT_BUCKETS)
        o1._bind()
        o1.use_device_lr(m1.device)
        step = m1.forward_train(b, early_backward=True)
        m1.backward_train(step)
        o1.step()
        ref.append(step.losses.clone())
    torch.cuda.synchronize()
    cfg, m2 = _setup()
    o2 = FlatSGD(m2, cfg)
    gs = engine.GraphedStep(m2, o2, warmup_steps=2)
    got = [gs.run(*d).clone() for d in seq]
    torch.cuda.synchronize()
    assert len(gs.graphs) == 3 and len(gs.seen) == 3          # small, big, crowded (64-slot bucket)
    for k, (a, b) in enumerate(zip(got, ref)):
        assert torch.isfinite(a).all()
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (k, a.tolist(), b.tolist())
    assert torch.allclose(m2.store.params, m1.store.params, rtol=1e-5, atol=1e-7)


def test_graph_replays_survive_anchor_grid_eviction(dev):
    """the anchor-grid cache is the grids' only owner and a captured step has a grid's address baked in. With more map
    sizes than the cache holds (cap forced to 2 here; 128 in production, which multi-scale VOC exceeds) an evicted grid must outlive the
    graphs: it is parked in ops._WS_RETIRED, eviction is least-recently-USED, and the replays of the evicted size still equal the eager run
    after the freed-looking memory has been churned."""
    from unit_amd import ops
    sizes = [(96, 128), (160, 224), (128, 160), (112, 208)]
    data = [[synthetic_batch(2, 2, hw=hw, seed=300 + 10 * i + j, max_gt=3) for j in range(2)] for i, hw in enumerate(sizes)]
    seq = [data[0][0], data[0][1], data[1][0], data[1][1], data[2][0], data[2][1], data[3][0], data[3][1], data[0][0], data[1][1], data[0][1], data[2][0]]
    cfg, m1 = _setup()
    o1 = FlatSGD(m1, cfg)
    ref = []
    for d in seq:
        b = m1.pack_batch(*d, gt_buckets=engine.GraphedStep.GT_BUCKETS)
        o1._bind()
        o1.use_device_lr(m1.device)
        step = m1.forward_train(b, early_backward=True)
        m1.backward_train(step)
        o1.step()
        ref.append(step.losses.clone())
    torch.cuda.synchronize()
    cfg, m2 = _setup()
    gen = m2.proposal_generator.anchor_generator
    gen.CACHE_CAP = 2
    o2 = FlatSGD(m2, cfg)
    gs = engine.GraphedStep(m2, o2, warmup_steps=2)
    retired0 = len(ops._WS_RETIRED)
    got = []
    for d in seq:
        got.append(gs.run(*d).clone())
        junk = [torch.full((1 << 18,), float("nan"), device=dev) for _ in range(8)]          # churn what the allocator would hand out again
        del junk
    torch.cuda.synchronize()
    assert len(gen._cache) <= 2 and len(ops._WS_RETIRED) > retired0          # grids were evicted, and kept alive for the graphs
    for k, (a, b) in enumerate(zip(got, ref)):
        assert torch.isfinite(a).all(), (k, a.tolist())
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (k, a.tolist(), b.tolist())
    # LRU, not FIFO: a hit moves the size to the back
    g0 = gen.grid(6, 8)
    gen.grid(10, 14)
    assert gen.grid(6, 8) is g0
    gen.grid(8, 10)                                                           # evicts (10, 14), the least recently used
    assert gen.grid(6, 8) is g0


def test_pack_batch_rejects_a_capacity_that_does_not_hold_the_batch(dev):
    cfg, m = _setup()
    sup, weak = synthetic_batch(1, 1, hw=(96, 128), seed=5, max_gt=40)
    with pytest.raises(ValueError, match="gt_capacity"):
        m.pack_batch(sup, weak, gt_capacity=8)


def test_capture_refuses_aliased_stream_roles(dev, monkeypatch):
    """Root cause of round 5's "1 in 12-24" hipStreamEndCapture SIGSEGV: the weight-gradient and the RPN-branch role of the
    step on ONE stream object. The stream-placement probe can no longer produce that (distinct objects even when it finds fewer queues); the
    experiment switch that still can is refused by GraphedStep with an error instead of a crash inside the ROCm runtime; the call-list replay,
    which captures nothing, runs such a placement and equals the eager step."""
    from unit_amd import ops
    st = ops.streams_on_distinct_queues(dev, 3, candidates=1)          # one candidate: at most one distinct queue can be FOUND
    assert len({s.cuda_stream for s in st}) == 3
    monkeypatch.setenv("UNIT_STREAM_MERGE", "wr")
    data = [synthetic_batch(2, 2, hw=(96, 128), seed=60 + i, max_gt=3) for i in range(2)]
    cfg, m = _setup()
    o = FlatSGD(m, cfg)
    gs = engine.GraphedStep(m, o, warmup_steps=1)
    gs.run(*data[0])
    assert m._wgrad_stream is m._rpn_stream
    with pytest.raises(RuntimeError, match="share one HIP stream object"):
        gs.run(*data[1])
    cfg, m1 = _setup()
    o1 = FlatSGD(m1, cfg)
    ref = []
    for d in (data[0], data[1], data[0], data[1]):
        b = m1.pack_batch(*d, gt_buckets=engine.GraphedStep.GT_BUCKETS)
        o1._bind()
        o1.use_device_lr(m1.device)
        step = m1.forward_train(b, early_backward=True)
        m1.backward_train(step)
        o1.step()
        ref.append(step.losses.clone())
    cfg, m2 = _setup()
    o2 = FlatSGD(m2, cfg)
    rs = engine.ReplayedStep(m2, o2, warmup_steps=1)
    got = [rs.run(*d).clone() for d in (data[0], data[1], data[0], data[1])]
    torch.cuda.synchronize()
    assert rs.stats["replayed"] == 2 and all(torch.equal(a, b) for a, b in zip(got, ref)) and torch.equal(m2.store.params, m1.store.params)
