"""CPU tests of the checkpoint adapter: DetectionCheckpointer-style containers, suffix alignment,
Caffe2 / MSRA ResNet names. No kernels are launched."""
import pickle

import numpy as np
import pytest
import torch

from unit_amd import checkpoint as ck
from unit_amd import config
from unit_amd.modeling import build_model


def _model(depth=50):
    c = config.voc_rcnn_c4_split1(depth)
    c.MODEL.DEVICE = "cpu"
    return build_model(c)


def test_save_load_roundtrip_with_sidecar_entries(tmp_path):
    m = _model()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn_like(p) * 0.01)
    path = ck.save_checkpoint(m, str(tmp_path / "best_model_final.pth"), iteration=1234, AP50=41.5)   # detection_checkpoint.py:40-45
    m2 = _model()
    rep = ck.load_checkpoint(m2, path)
    assert rep["missing"] == [] and rep["unexpected"] == [] and rep["extras"] == {"iteration": 1234, "AP50": 41.5}
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_suffix_alignment_and_res5_heads_from_trunk():
    m = _model()
    sd = m.state_dict()
    # an ImageNet trunk as Detectron2 stores it after conversion: no "backbone." prefix, res5 present, nothing else
    trunk = {k[len("backbone."):]: torch.randn_like(v) for k, v in sd.items() if k.startswith("backbone.") and v.dtype.is_floating_point}
    for k, v in sd.items():
        if k.startswith("roi_heads.box_head.res5."):
            trunk[k[len("roi_heads.box_head."):]] = torch.randn_like(v)
    rep = ck.load_checkpoint(m, {"model": trunk})
    new = m.state_dict()
    assert torch.equal(new["backbone.res3.1.conv2.weight"], trunk["res3.1.conv2.weight"])
    assert torch.equal(new["roi_heads.box_head.res5.0.conv1.weight"], trunk["res5.0.conv1.weight"])
    assert torch.equal(new["roi_heads.weak_box_head.res5.2.conv3.norm.bias"], trunk["res5.2.conv3.norm.bias"])
    assert rep["unexpected"] == []
    assert all(not k.startswith("backbone.") and ".res5." not in k for k in rep["missing"])
    assert "proposal_generator.rpn_head.conv.weight" in rep["missing"]
    m2 = _model()
    rep2 = ck.load_checkpoint(m2, {"model": trunk}, res5_from_trunk=False)
    assert any(k.startswith("roi_heads.box_head.res5.") for k in rep2["missing"])
    with pytest.raises(ValueError):
        ck.load_checkpoint(_model(), {"model": {"res3.1.conv2.weight": torch.zeros(1, 2, 3, 3)}})      # shape mismatch is an error


def _c2_name(k):
    """inverse of convert_basic_c2_names for trunk keys (test helper)"""
    if k.startswith("stem.conv1.norm."):
        return {"weight": "res_conv1_bn_s", "bias": "res_conv1_bn_b"}[k.rsplit(".", 1)[-1]]
    k = k.replace("stem.conv1.", "conv1.")
    k = k.replace(".shortcut.", ".branch1.").replace(".conv1.", ".branch2a.").replace(".conv2.", ".branch2b.").replace(".conv3.", ".branch2c.")
    k = k.replace("norm.weight", "bn_s").replace("norm.bias", "bn_b")
    k = k.replace(".weight", "_w").replace(".bias", "_b")
    return k.replace(".", "_")


def test_caffe2_msra_pickle(tmp_path):
    m = _model()
    sd = m.state_dict()
    trunk = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.") and "running" not in k}
    for k, v in sd.items():
        if k.startswith("roi_heads.box_head.res5.") and "running" not in k:
            trunk[k[len("roi_heads.box_head."):]] = v
    blobs = {_c2_name(k): np.random.RandomState(len(k)).randn(*v.shape).astype(np.float32) for k, v in trunk.items()}
    assert "conv1_w" in blobs and "res_conv1_bn_s" in blobs and "res2_0_branch2a_w" in blobs and "res4_5_branch2c_bn_b" in blobs
    assert "res5_0_branch1_w" in blobs
    assert sorted(ck.convert_basic_c2_names(list(blobs))) == sorted(trunk)
    blobs["fc1000_w"] = np.zeros((1000, 2048), np.float32)
    blobs["conv1_w_momentum"] = np.zeros((64, 3, 7, 7), np.float32)
    path = tmp_path / "R-50.pkl"
    with open(path, "wb") as f:
        pickle.dump({"blobs": blobs}, f)
    rep = ck.load_checkpoint(m, str(path))
    new = m.state_dict()
    assert np.array_equal(new["backbone.stem.conv1.weight"].numpy(), blobs["conv1_w"])
    assert np.array_equal(new["backbone.res4.5.conv3.norm.bias"].numpy(), blobs["res4_5_branch2c_bn_b"])
    assert np.array_equal(new["roi_heads.weak_box_head.res5.0.shortcut.weight"].numpy(), blobs["res5_0_branch1_w"])
    assert rep["unexpected"] == []
    assert all(("running_" in k) or not (k.startswith("backbone.") or ".res5." in k) for k in rep["missing"])
