"""f3 on the device (reference checkpoint/detection_checkpoint.py:8-52, MODEL.WEIGHTS of every yaml):
weights loaded from a Detectron2-format `.pth` and from a Caffe2 / MSRA `.pkl` reach the PREPARED device copies (FrozenBN-folded
bf16 / fp32 NHWC weights, dgrad-transposed copies) of a model that had already run with other weights; `save_checkpoint` round-trips."""
import pickle

import numpy as np
import pytest
import torch

import unit_oracle as orc
from unit_amd import checkpoint as ck
from unit_amd import config
from unit_amd.modeling import build_model
from unit_amd.synthetic import init_synthetic_weights, synthetic_batch

pytestmark = pytest.mark.gpu


def _cfg():
    c = config.voc_rcnn_c4_split1(50)
    c.MODEL.DEVICE = "cuda"
    c.MODEL.RPN.PRE_NMS_TOPK_TEST, c.MODEL.RPN.POST_NMS_TOPK_TEST = 400, 80
    return c


def _ocfg(cfg):
    return dict(depth=50, num_classes=20, novel_classes=list(cfg.DATASETS.FEWSHOT.NOVEL_CLASSES_ID), base_classes=list(cfg.DATASETS.FEWSHOT.BASE_CLASSES_ID),
                coco_indexer=orc.VOC_COCO_INDEXER, pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD, pre_nms_topk_test=400,
                post_nms_topk_test=80, multi_box_head=True)


def _spread(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        model.roi_heads.box_predictor.cls_score_delta.weight.copy_(torch.randn(21, 2048, generator=g) * 0.02)


def _check_inference(model, cfg, image):
    model.eval()
    model.compute_dtype = torch.float32
    out = model([{"image": image, "height": 128, "width": 192}])[0]["instances"]
    p = {k: v.detach().cpu().clone().contiguous() for k, v in model.state_dict().items()}
    b, s, c, r, _ = orc.inference(p, image, _ocfg(cfg), out_hw=(128, 192))
    # as sets (a detection whose score sits within fp32 noise of the 0.05 threshold, or a pair at the NMS threshold, may
    # legitimately flip): >= 95 % of either side's detections have a partner of the same class, box within 0.05 px, score 1e-4
    hb, hs, hc = out.pred_boxes.tensor.cpu(), out.scores.cpu(), out.pred_classes.cpu()
    assert len(b) > 3 and abs(len(hb) - len(b)) <= 2
    d = torch.cdist(hb.double(), b.double(), p=float("inf"))
    ok = (d < 0.05) & (hc[:, None] == c[None, :]) & ((hs[:, None] - s[None, :]).abs() < 1e-4)
    assert ok.any(1).float().mean() >= 0.95 and ok.any(0).float().mean() >= 0.95, (ok.any(1).float().mean(), ok.any(0).float().mean())
    return out


def test_pth_checkpoint_reaches_prepared_device_copies(dev, tmp_path):
    cfg = _cfg()
    src = build_model(cfg)
    init_synthetic_weights(src, seed=21)
    _spread(src, 1)
    path = ck.save_checkpoint(src, str(tmp_path / "model_final.pth"), iteration=500, AP50=12.5)
    model = build_model(cfg)
    init_synthetic_weights(model, seed=4)        # OTHER weights first ...
    _spread(model, 2)
    sup, _ = synthetic_batch(1, 0, hw=(128, 192), seed=8)
    first = _check_inference(model, cfg, sup[0]["image"])          # ... and a run that prepares the device copies from them
    rep = ck.load_checkpoint(model, path)
    assert rep["missing"] == [] and rep["unexpected"] == [] and rep["extras"] == {"iteration": 500, "AP50": 12.5}
    second = _check_inference(model, cfg, sup[0]["image"])         # the oracle now runs on the LOADED weights: stale copies would fail
    assert not (len(first) == len(second) and torch.allclose(first.scores, second.scores))
    # training path: the dgrad / flat-store copies refresh too (one fp32 step, losses vs oracle)
    model.train()
    sup2, weak2 = synthetic_batch(1, 1, hw=(96, 128), seed=3, max_gt=3)
    cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 16
    model.roi_heads.batch_size_per_image = 16
    batch = model.pack_batch(sup2, weak2)
    model._ensure_ready()
    perms = model.sampling_permutations(1, 6 * 8 * 15, model.proposal_generator.post_nms_topk[True] + batch.gt_boxes.shape[1])
    step = model.forward_train(batch, perms)
    model.backward_train(step)
    trainable = {n for n, q in model.named_parameters() if q.requires_grad}
    p = {k: v.detach().cpu().clone().contiguous().requires_grad_(k in trainable) for k, v in model.state_dict().items()}
    oc = _ocfg(cfg)
    oc.update(rois_per_image=16, pre_nms_topk=cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, post_nms_topk=cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN)
    ref, _ = orc.step_losses(p, [x["image"] for x in sup2], [x["instances"].gt_boxes.tensor for x in sup2], [x["instances"].gt_classes for x in sup2],
                             [x["image"] for x in weak2], [x["instances"].gt_classes for x in weak2],
                             dict(rpn=[x.long().cpu() for x in perms["rpn"]], roi=[x.long().cpu() for x in perms["roi"]]), oc)
    from unit_amd.modeling.rcnn import LOSS_NAMES
    got = dict(zip(LOSS_NAMES, step.losses.cpu().tolist()))
    for k, v in ref.items():
        assert abs(got[k] - v.item()) <= 1e-4 * max(1.0, abs(v.item())), (k, got[k], v.item())
    # round trip: what save_checkpoint writes is bit for bit what was loaded
    again = ck.save_checkpoint(model, str(tmp_path / "again.pth"))
    a, b = torch.load(path, weights_only=False)["model"], torch.load(again, weights_only=False)["model"]
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k


def _c2_name(k):
    if k.startswith("stem.conv1.norm."):
        return {"weight": "res_conv1_bn_s", "bias": "res_conv1_bn_b"}[k.rsplit(".", 1)[-1]]
    k = k.replace("stem.conv1.", "conv1.")
    k = k.replace(".shortcut.", ".branch1.").replace(".conv1.", ".branch2a.").replace(".conv2.", ".branch2b.").replace(".conv3.", ".branch2c.")
    k = k.replace("norm.weight", "bn_s").replace("norm.bias", "bn_b")
    k = k.replace(".weight", "_w").replace(".bias", "_b")
    return k.replace(".", "_")


def test_caffe2_pkl_trunk_reaches_device(dev, tmp_path):
    """an MSRA-style `R-50.pkl` (Caffe2 blob names, BN as scale / bias) into a model that already ran: backbone AND both Res5
    heads take the trunk's tensors; inference afterwards equals the oracle on the loaded state."""
    cfg = _cfg()
    model = build_model(cfg)
    init_synthetic_weights(model, seed=4)
    _spread(model, 2)
    sup, _ = synthetic_batch(1, 0, hw=(128, 192), seed=8)
    _check_inference(model, cfg, sup[0]["image"])
    donor = build_model(cfg)
    init_synthetic_weights(donor, seed=33)
    sd = donor.state_dict()
    trunk = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.") and "running" not in k}
    trunk.update({k[len("roi_heads.box_head."):]: v for k, v in sd.items() if k.startswith("roi_heads.box_head.res5.") and "running" not in k})
    blobs = {_c2_name(k): v.detach().cpu().numpy().astype(np.float32) for k, v in trunk.items()}
    blobs["fc1000_w"] = np.zeros((1000, 2048), np.float32)
    path = tmp_path / "R-50.pkl"
    with open(path, "wb") as f:
        pickle.dump({"blobs": blobs, "__author__": "Caffe2"}, f)
    rep = ck.load_checkpoint(model, str(path))
    assert rep["unexpected"] == []
    new = model.state_dict()
    assert torch.equal(new["backbone.res4.5.conv3.weight"].cpu(), sd["backbone.res4.5.conv3.weight"].cpu())
    assert torch.equal(new["roi_heads.weak_box_head.res5.1.conv2.weight"].cpu(), sd["roi_heads.box_head.res5.1.conv2.weight"].cpu())
    _check_inference(model, cfg, sup[0]["image"])
