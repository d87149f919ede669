"""Checkpoint adapter: the reference trains from `MODEL.WEIGHTS` files written by Detectron2's
`DetectionCheckpointer` (`configs/VOC/VOC-RCNN-101-C4-split1.yaml:3` -> resnet_101_MSRA_C4.pth; base-training / fine-tune
checkpoints `model_final.pth`, `best_model_final.pth` with the AP50 side entries of
`checkpoint/detection_checkpoint.py:13-17,40-45`). This module's model keeps the reference's state-dict keys and logical NCHW
shapes (tests/test_model_cpu.py), so loading is a key alignment, not a re-layout: the device-side NHWC / bf16 copies are
refreshed by `Conv2d.prepare` after the load (`layers._FROZEN_EPOCH`).

Restated from Detectron2 v0.3 (not under /root/reference; parity unpinned):
  * `checkpoint/c2_model_loading.py::convert_basic_c2_names` -- Caffe2 / MSRA names -> Detectron2 names for the ResNet trunk
    (`R-101.pkl`-style files: `conv1_w`, `res2_0_branch2a_w`, `res2_0_branch2a_bn_s`, ...);
  * `align_and_update_state_dicts` -- every model key takes the checkpoint key that is its longest suffix ("backbone.res2.0.
    conv1.weight" <- "res2.0.conv1.weight"), ambiguity is an error, unmatched keys on either side are reported;
  * the container: a `.pth` holds `{"model": state_dict, ...extras}` (extras = iteration, AP50, ...), a legacy `.pkl` holds
    `{"model" | "blobs": {name: ndarray}, "__author__": "Caffe2"?}`.
"""
import pickle
import re

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------- Caffe2 / MSRA names
def convert_basic_c2_names(names):
    """Caffe2 ResNet blob names -> Detectron2 parameter names (same order). d2 `convert_basic_c2_names`."""
    out = []
    for k in names:
        k = k.replace("_", ".")
        k = re.sub(r"\.b$", ".bias", k)
        k = re.sub(r"\.w$", ".weight", k)
        k = re.sub(r"bn\.s$", "norm.weight", k)        # "affine channel" scale / bias of the frozen BN
        k = re.sub(r"bn\.bias$", "norm.bias", k)
        k = re.sub(r"bn\.rm", "norm.running_mean", k)
        k = re.sub(r"bn\.running.mean$", "norm.running_mean", k)
        k = re.sub(r"bn\.riv$", "norm.running_var", k)
        k = re.sub(r"bn\.running.var$", "norm.running_var", k)
        k = re.sub(r"bn\.gamma$", "norm.weight", k)
        k = re.sub(r"bn\.beta$", "norm.bias", k)
        k = re.sub(r"gn\.s$", "norm.weight", k)
        k = re.sub(r"gn\.bias$", "norm.bias", k)
        k = re.sub(r"^res\.conv1\.norm\.", "conv1.norm.", k)   # stem
        k = re.sub(r"^conv1\.", "stem.conv1.", k)
        k = k.replace(".branch1.", ".shortcut.")
        k = k.replace(".branch2a.", ".conv1.")
        k = k.replace(".branch2b.", ".conv2.")
        k = k.replace(".branch2c.", ".conv3.")
        out.append(k)
    return out


def convert_c2_state(blobs):
    """{c2 name: ndarray} -> {d2 name: tensor} for a classification-pretrained ResNet: momentum blobs and the 1000-way
    classifier are dropped; a frozen-BN layer without stored statistics gets mean 0 / var 1 at load time (strict=False)."""
    keep = [k for k in blobs if not k.endswith("_momentum") and not k.startswith("fc1000") and not k.startswith("pred")]
    new = convert_basic_c2_names(keep)
    return {n: torch.from_numpy(np.ascontiguousarray(blobs[o])) for o, n in zip(keep, new)}


# ---------------------------------------------------------------------------------------------- key alignment
def align_keys(model_keys, ckpt_keys):
    """-> {model key: checkpoint key}. A checkpoint key matches a model key when it equals it or is a '.'-bounded suffix of it;
    the longest suffix wins; two equally long candidates are an error (d2 `align_and_update_state_dicts`). One checkpoint key
    may serve several model keys (the trunk's `res5.*` feeds both Res5 box heads)."""
    by_leaf = {}
    for c in ckpt_keys:
        by_leaf.setdefault(c.rsplit(".", 1)[-1], []).append(c)
    mapping = {}
    for m in model_keys:
        best, tie = None, False
        for c in by_leaf.get(m.rsplit(".", 1)[-1], ()):
            if m == c or m.endswith("." + c):
                if best is None or len(c) > len(best):
                    best, tie = c, False
                elif len(c) == len(best):
                    tie = True
        if best is not None:
            if tie:
                raise ValueError(f"ambiguous checkpoint match for {m}")
            mapping[m] = best
    return mapping


def load_file(path):
    """-> (state {name: tensor}, extras dict, legacy: bool)"""
    if str(path).endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1")
        blobs = data.get("model", data.get("blobs", data))
        if data.get("__author__", "Caffe2").lower().startswith("caffe2") or "blobs" in data:
            return convert_c2_state(blobs), {}, True
        return {k: torch.as_tensor(v) for k, v in blobs.items()}, {k: v for k, v in data.items() if k != "model"}, False
    data = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(data, dict) and "model" in data and isinstance(data["model"], dict):
        return dict(data["model"]), {k: v for k, v in data.items() if k != "model"}, False
    return dict(data), {}, False


def load_checkpoint(model, path_or_state, res5_from_trunk=True):
    """Loads into `model` (keys aligned by suffix). Returns {"missing": [...], "unexpected": [...], "extras": {...}}.
    res5_from_trunk: an ImageNet ResNet checkpoint stores `res5.*`; the C4 model has no `backbone.res5` but Res5 box heads
    (`modeling/roi_heads/box_head.py:65-75`): `roi_heads.box_head.res5.*` and `roi_heads.weak_box_head.res5.*` both take those
    tensors, exactly what the suffix rule of Detectron2 does for `res5.0.conv1.weight`."""
    if isinstance(path_or_state, dict):
        state, extras = (dict(path_or_state["model"]), {k: v for k, v in path_or_state.items() if k != "model"}) \
            if "model" in path_or_state and isinstance(path_or_state["model"], dict) else (dict(path_or_state), {})
    else:
        state, extras, _ = load_file(path_or_state)
    own = model.state_dict()
    mapping = align_keys(own.keys(), state.keys())
    if not res5_from_trunk:
        mapping = {m: c for m, c in mapping.items() if m == c or not (".res5." in m and c.startswith("res5."))}
    new = {}
    for m, c in mapping.items():
        t = torch.as_tensor(state[c])
        if tuple(t.shape) != tuple(own[m].shape):
            raise ValueError(f"shape mismatch for {m} <- {c}: {tuple(t.shape)} vs {tuple(own[m].shape)}")
        new[m] = t.to(own[m].dtype)
    model.load_state_dict(new, strict=False)
    missing = [k for k in own if k not in new]
    unexpected = sorted(set(state) - set(mapping.values()))
    return {"missing": missing, "unexpected": unexpected, "extras": extras}


def save_checkpoint(model, path, **extras):
    """`DetectionCheckpointer.save(name, **extras)` format: {"model": state_dict (reference keys, NCHW fp32), **extras}; the
    reference's best-model sidecar passes iteration= and AP50= (`checkpoint/detection_checkpoint.py:40-45`)."""
    data = {"model": {k: v.detach().cpu() for k, v in model.state_dict().items()}}
    data.update(extras)
    torch.save(data, path)
    return path
