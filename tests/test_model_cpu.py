"""CPU tests of the host-side model mirror: registry names, state-dict keys (the checkpoint contract),
freeze lists, flat parameter store layout. No kernels are launched."""
import torch

from unit_amd import config
from unit_amd.modeling import build_model
from unit_amd.structures import (FAST_RCNN_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, ROI_BOX_HEAD_REGISTRY,
                                 ROI_HEADS_REGISTRY)


def _cfg(depth=50, ft=False):
    c = config.voc_rcnn_c4_split1_ft(depth) if ft else config.voc_rcnn_c4_split1(depth)
    c.MODEL.DEVICE = "cpu"
    return c


def test_registry_names_match_reference_yaml():
    for reg, names in [(META_ARCH_REGISTRY, ["WeaklySupervisedRCNNNoMeta"]), (PROPOSAL_GENERATOR_REGISTRY, ["WSRPN"]),
                       (ROI_HEADS_REGISTRY, ["WSROIHeadNoMeta", "WSROIHeadFineTune"]),
                       (ROI_BOX_HEAD_REGISTRY, ["Res5BoxHead", "Res5BoxHeadWithMask"]),
                       (FAST_RCNN_REGISTRY, ["SupervisedDetectorOutputsBase", "SupervisedDetectorOutputsFineTune"])]:
        for n in names:
            assert n in reg, n


def test_state_dict_keys_and_shapes():
    m = build_model(_cfg(50))
    sd = m.state_dict()
    expect = {
        "backbone.stem.conv1.weight": (64, 3, 7, 7), "backbone.stem.conv1.norm.running_var": (64,),
        "backbone.res2.0.shortcut.weight": (256, 64, 1, 1), "backbone.res3.0.conv1.weight": (128, 256, 1, 1),
        "backbone.res4.5.conv2.weight": (256, 256, 3, 3), "backbone.res4.0.shortcut.norm.bias": (1024,),
        "proposal_generator.rpn_head.conv.weight": (1024, 1024, 3, 3), "proposal_generator.rpn_head.conv.bias": (1024,),
        "proposal_generator.rpn_head.objectness_logits.weight": (15, 1024, 1, 1),
        "proposal_generator.rpn_head.anchor_deltas.bias": (60,), "proposal_generator.anchor_generator.cell_anchors.0": (15, 4),
        "roi_heads.box_head.res5.0.shortcut.weight": (2048, 1024, 1, 1), "roi_heads.weak_box_head.res5.2.conv3.weight": (2048, 512, 1, 1),
        "roi_heads.box_predictor.cls_score_delta.weight": (21, 2048), "roi_heads.box_predictor.bbox_pred_delta.bias": (80,),
        "roi_heads.box_predictor.weak_detector_head.classifier_stream.weight": (20, 2048),
        "roi_heads.box_predictor.weak_detector_head.oicr_predictors.2.bias": (21,),
        "roi_heads.box_predictor.embeddings.weight": (80, 300),
    }
    for k, shp in expect.items():
        assert k in sd and tuple(sd[k].shape) == shp, k
    assert not any(k.startswith("backbone.res5") for k in sd)
    assert len([k for k in sd if k.startswith("backbone.res4.")]) == 6 * (3 * 5) + 5   # R50: 6 blocks, one shortcut


def test_freeze_at_and_finetune_freeze_lists():
    m = build_model(_cfg(50))
    req = {n: p.requires_grad for n, p in m.named_parameters()}
    assert not req["backbone.stem.conv1.weight"] and not req["backbone.res2.2.conv3.weight"]
    assert req["backbone.res3.0.conv1.weight"] and req["roi_heads.weak_box_head.res5.0.conv1.weight"]
    assert not req["roi_heads.box_predictor.embeddings.weight"]
    ft = build_model(_cfg(50, ft=True))
    trainable = sorted(n for n, p in ft.named_parameters() if p.requires_grad)
    assert trainable == ["roi_heads.box_predictor.bbox_pred_ft.bias", "roi_heads.box_predictor.bbox_pred_ft.weight",
                         "roi_heads.box_predictor.cls_score_ft.bias", "roi_heads.box_predictor.cls_score_ft.weight"]
    assert sum(p.numel() for p in ft.parameters() if p.requires_grad) == 21 * 2048 + 21 + 80 * 2048 + 80   # 0.207 M


def test_flat_store_layout_and_fused_heads():
    m = build_model(_cfg(50))
    before = {k: v.clone() for k, v in m.state_dict().items()}
    st = m.flatten_parameters()
    after = m.state_dict()
    for k in before:
        assert torch.equal(before[k], after[k]), k          # values preserved
    n_train = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert st.size >= n_train and st.size < n_train * 1.01
    w = m.backbone.res4[0].conv2.weight
    assert w.data.permute(0, 2, 3, 1).is_contiguous() and w.grad.permute(0, 2, 3, 1).is_contiguous()   # physical [K][R][S][C]
    bp = m.roi_heads.box_predictor
    for grp in (bp.group, bp.weak_detector_head.group, m.proposal_generator.rpn_head.pred):
        wv, bv = grp._fused_views("data")
        assert wv is not None and wv.shape == (grp.kp, grp.cin) and bv.shape == (grp.k,)
        gv, gb = grp._fused_views("grad")
        assert gv is not None
        assert torch.count_nonzero(wv[grp.k:]) == 0          # pad rows
    tags = [t for t, _, _ in st.tags]
    assert tags.index("heads") < tags.index("box_head") < tags.index("rpn") < tags.index("res4") < tags.index("res3")
    # R50 trainable parameter count (~48.1 M)
    assert abs(n_train - 48.1e6) < 0.5e6, n_train


def test_r101_parameter_count():
    m = build_model(_cfg(101))
    n_train = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert abs(n_train - 67.05e6) < 0.5e6, n_train      # 67.0 M trainable


def test_gradient_bucket_tags_follow_backward_order():
    """flat store tags = data-parallel buckets in the order the backward completes them; the long res4 stage of R101 is four
    buckets of six blocks (blocks 22..17 first), R50's six blocks are one"""
    m = build_model(_cfg(101))
    st = m.flatten_parameters()
    tags = [t for t, a, b in st.tags]
    assert tags[-6:] == ["res4", "res4.1", "res4.2", "res4.3", "res3"][-6:] or tags[-5:] == ["res4", "res4.1", "res4.2", "res4.3", "res3"]
    assert all(b0 == a1 for (_, _, b0), (_, a1, _) in zip(st.tags, st.tags[1:])) and st.tags[0][1] == 0 and st.tags[-1][2] == st.size
    off = {e["name"]: e["offset"] for e in st.entries}
    rng = {t: (a, b) for t, a, b in st.tags if t.startswith("res4")}
    assert rng["res4"][0] <= off["backbone.res4.22.conv3.weight"] < rng["res4"][1]
    assert rng["res4"][0] <= off["backbone.res4.17.conv1.weight"] < rng["res4"][1]
    assert rng["res4.1"][0] <= off["backbone.res4.16.conv3.weight"] < rng["res4.1"][1]
    assert rng["res4.3"][0] <= off["backbone.res4.0.shortcut.weight"] < rng["res4.3"][1]
    r4 = m.backbone.res4
    assert [r4.last_block_of_bucket(k) for k in range(4)] == [17, 11, 5, 0] and r4.bucket_of_block(5) == 2 and r4.bucket_of_block(4) == 3
    m50 = build_model(_cfg(50))
    assert [t for t, a, b in m50.flatten_parameters().tags if t.startswith("res4")] == ["res4"]


def test_config_merges_reference_style_yaml(tmp_path):
    """yacs semantics of CfgNode.merge_from_file: string leaves that are python literals are evaluated (the reference's yaml files
    write tuples as strings: `STEPS: (12000, 24000)`), `_BASE_` chains, type coercion list <-> tuple; the LR schedule then works."""
    from unit_amd import config, solver
    base = tmp_path / "Base.yaml"
    base.write_text("MODEL:\n  RPN:\n    PRE_NMS_TOPK_TEST: 6000\n    POST_NMS_TOPK_TEST: 1000\nSOLVER:\n  IMS_PER_BATCH: 16\n")
    y = tmp_path / "child.yaml"
    y.write_text('_BASE_: "Base.yaml"\nMODEL:\n  RESNETS:\n    DEPTH: 101\n  ROI_HEADS:\n    NAME: "WSROIHeadNoMeta"\n    MULTI_BOX_HEAD: True\n'
                 '  ROI_MASK_HEAD:\n    POOLER_TYPE: "None"\nINPUT:\n  MIN_SIZE_TRAIN: (480, 512, 800)\nSOLVER:\n  STEPS: (12000, 24000)\n'
                 '  BASE_LR: 0.02\n  WARMUP_ITERS: 100\nDATASETS:\n  FEWSHOT:\n    NOVEL_CLASSES_ID: [2, 5, 9, 13, 17]\n')
    c = config.get_cfg()
    c.merge_from_file(str(y))
    assert c.SOLVER.STEPS == (12000, 24000) and isinstance(c.SOLVER.STEPS, tuple)
    assert c.INPUT.MIN_SIZE_TRAIN == (480, 512, 800) and c.SOLVER.IMS_PER_BATCH == 16 and c.MODEL.RESNETS.DEPTH == 101
    assert c.MODEL.ROI_HEADS.NAME == "WSROIHeadNoMeta" and c.MODEL.ROI_MASK_HEAD.POOLER_TYPE in ("None", None)
    sched = solver.WarmupMultiStepLR(c)
    assert abs(sched(200) - 0.02) < 1e-12 and abs(sched(12000) - 0.002) < 1e-12 and abs(sched(24000) - 0.0002) < 1e-12
    c.merge_from_list(["SOLVER.STEPS", "(50,)", "MODEL.ROI_HEADS.NAME", "WSROIHeadFineTune", "SOLVER.BASE_LR", "0.001"])
    assert c.SOLVER.STEPS == (50,) and c.MODEL.ROI_HEADS.NAME == "WSROIHeadFineTune" and c.SOLVER.BASE_LR == 0.001


def test_presets_equal_the_reference_yaml_files_when_present():
    """the presets of unit_amd.config against the reference's own yaml files (authoring container only)."""
    import os
    import pytest
    from unit_amd import config
    root = "/root/reference/configs"
    if not os.path.isdir(root):
        pytest.skip("reference not present")
    skip = ("DATASETS.TRAIN", "DATASETS.TEST", "MODEL.WEIGHTS", "MODEL.ROI_HEADS.EMBEDDING_PATH", "INPUT.MIN_SIZE_TRAIN", "TEST.EVAL_PERIOD",
            "SOLVER.CHECKPOINT_PERIOD")

    def diff(a, b, pre=""):
        out = []
        for k in sorted(set(a) & set(b)):
            if isinstance(a[k], dict):
                out += diff(a[k], b[k], pre + k + ".")
            elif (pre + k) not in skip and a[k] != b[k] and not (isinstance(a[k], (list, tuple)) and list(a[k]) == list(b[k])) \
                    and not (a[k] in ("None", None) and b[k] in ("None", None)):
                out.append((pre + k, a[k], b[k]))
        return out
    for y, preset in (("VOC/VOC-RCNN-101-C4-split1.yaml", config.voc_rcnn_c4_split1(101)),
                      ("VOC/FT/1_shot/VOC-RCNN-101-C4-split1-ft.yaml", config.voc_rcnn_c4_split1_ft(101)),
                      ("COCO/COCO-RCNN-50-C4-split1.yaml", config.coco_rcnn_c4_split1(50)),
                      ("COCO/COCO-RCNN-50-C4-split1-segm.yaml", config.coco_rcnn_c4_split1_segm(50)),
                      ("COCO/COCO-RCNN-50-C4-split1-segm-ft.yaml", config.coco_rcnn_c4_split1_segm_ft(50))):
        c = config.get_cfg()
        c.merge_from_file(os.path.join(root, y))
        assert diff(c, preset) == [], (y, diff(c, preset))


def test_loader_consumer_tile_choice_is_host_arithmetic():
    """ops.lc_tile_code (the tile of csrc/conv_igemm_lc.hip per layer) minimises (tiles per CU) x (BM + BN) + an epilogue term over the
    eight instantiated tiles; the values the res4 shapes get are the ones measured fastest on the device (profiles/r03_exp_loader_consumer.txt)."""
    from unit_amd import ops
    m = 4 * 38 * 63
    assert ops.lc_tile_code(m, 256, 1024) == 152          # 80 x 128: 240 tiles, one per CU
    assert ops.lc_tile_code(m, 256, 9 * 256) == 152
    assert ops.lc_tile_code(m, 1024, 512) == 154          # 80 x 256: 480 tiles, two per CU
    for code in (ops.lc_tile_code(mm, k, kg) for mm in (100, 4788, 9576, 37500) for k in (64, 128, 256, 1024) for kg in (512, 2304)):
        fb, fa = (code - 100) // 10, (code - 100) % 10
        assert fa in (2, 4) and 4 <= fb <= (8 if fa == 2 else 6)
    import torch
    # short contractions and big maps stay on the 4-wave tiles
    assert ops.MID_TILE_POLICY(torch.bfloat16, m, 1024, 256, 256) < 100
    assert ops.MID_TILE_POLICY(torch.bfloat16, 4 * 150 * 250, 256, 64, 576) < 100


def test_wgrad_group_plan_is_host_arithmetic():
    """unit_conv2d_wgrad_group_plan (tile kind and split count of every layer of a grouped weight-gradient launch) runs on the host:
    Res5-sized layers get 256x256 tiles and 3 slabs, a res4 gradient bucket 2, the RPN's 3x3 conv 3; 128-channel layers and a lone
    two-tile layer go to the 128x128 grid; ineligible layers are refused (csrc/conv_wgrad.hip, profiles/r03_exp_grouped_wgrad.txt)."""
    from unit_amd import ops

    def plan(cases, hint=0):
        pr = (ops.WgradProblem * len(cases))()
        for q, (n, h, w, c, k, r, stride, pad) in zip(pr, cases):
            oh, ow = ops.conv_out_size(h, w, r, r, stride, pad)
            q.N, q.H, q.W, q.C, q.K, q.R, q.S, q.stride, q.pad, q.OH, q.OW, q.ldy = n, h, w, c, k, r, r, stride, pad, oh, ow, k
        rc = ops.lib().unit_conv2d_wgrad_group_plan(pr, len(cases), hint)
        return rc, [(q.kind, q.splits) for q in pr]

    n = 1024
    head = [(n, 14, 14, 1024, 512, 1, 2, 0), (n, 7, 7, 512, 512, 3, 1, 1), (n, 7, 7, 512, 2048, 1, 1, 0), (n, 14, 14, 1024, 2048, 1, 2, 0)] + \
           [(n, 7, 7, 2048, 512, 1, 1, 0), (n, 7, 7, 512, 512, 3, 1, 1), (n, 7, 7, 512, 2048, 1, 1, 0)] * 2
    assert plan(head) == (0, [(2, 3)] * 10)
    bucket = [(4, 38, 63, 1024, 256, 1, 1, 0), (4, 38, 63, 256, 256, 3, 1, 1), (4, 38, 63, 256, 1024, 1, 1, 0)] * 6
    assert plan(bucket) == (0, [(2, 2)] * 18)
    assert plan([(2, 38, 63, 1024, 1024, 3, 1, 1)]) == (0, [(2, 3)])
    assert plan(bucket, hint=5)[1] == [(2, 5)] * 18                                  # the caller's split count
    assert plan([(4, 150, 250, 256, 512, 1, 2, 0)])[1][0][0] == 1                    # two 256x256 tiles alone: the 128x128 grid
    res3 = [(4, 75, 125, 512, 128, 1, 1, 0), (4, 75, 125, 128, 128, 3, 1, 1), (4, 75, 125, 128, 512, 1, 1, 0)] * 3
    rc, ks = plan(res3)
    assert rc == 0 and all(k == 1 for k, _ in ks) and len({sp for _, sp in ks}) == 1 and 2 <= ks[0][1] <= 16
    assert plan([(2, 8, 8, 64, 128, 1, 1, 0)])[0] != 0                                # C % 128 != 0: not eligible
    assert ops.lib().unit_conv2d_wgrad_group_supported(ops.dt(torch.bfloat16), 4, 38, 63, 256, 1, 1, 1024) == 2
    assert ops.lib().unit_conv2d_wgrad_group_supported(ops.dt(torch.float32), 4, 38, 63, 256, 1, 1, 1024) == 0


def test_voc_shaped_steps_follow_the_yaml_and_group_by_orientation():
    """synthetic.voc_shaped_steps (bench.py --shapes voc): sizes as the reference's loader makes them -- short side from INPUT.MIN_SIZE_TRAIN,
    long side <= MAX_SIZE_TRAIN (configs/VOC/VOC-RCNN-101-C4-split1.yaml:27-29), every batch one orientation (data/build.py:476-497)"""
    from unit_amd import config
    from unit_amd.synthetic import voc_shaped_steps
    cfg = config.voc_rcnn_c4_split1(101)
    assert tuple(cfg.INPUT.MIN_SIZE_TRAIN) == (480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800) and cfg.INPUT.MAX_SIZE_TRAIN == 1333
    steps = voc_shaped_steps(200, cfg, seed=3)
    assert steps == voc_shaped_steps(200, cfg, seed=3) and steps != voc_shaped_steps(200, cfg, seed=4)
    shorts, portrait_batches = set(), 0
    for sup, weak in steps:
        for batch in (sup, weak):
            assert len(batch) == 2
            assert len({w > h for h, w in batch}) == 1, batch          # one aspect-ratio bucket per batch (width > height or not)
            portrait_batches += not (batch[0][1] > batch[0][0])
            for h, w in batch:
                assert max(h, w) <= 1333
                shorts.add(min(h, w))
    assert shorts <= set(cfg.INPUT.MIN_SIZE_TRAIN) | set(range(470, 801)) and {480, 800} <= shorts
    assert 20 < portrait_batches < 160          # about a fifth of VOC is portrait
