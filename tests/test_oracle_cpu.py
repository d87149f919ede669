"""CPU tests (no GPU): the oracle against the reference-generated golden vectors, oracle self-consistency between
its independent restatements (torch vs C), and the C-ABI library's exported symbols."""
import ctypes
import os

import numpy as np
import torch

import unit_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden", "matcher_golden.npz")


def test_matcher_oracle_vs_reference_golden():
    """tests/golden/matcher_golden.npz was produced by importing /root/reference/modeling/matcher.py."""
    gold = np.load(GOLD)
    names = sorted({k.split("/")[0] for k in gold.files})
    assert len(names) >= 8
    for name in names:
        q = torch.from_numpy(gold[f"{name}/q"])
        for cn, cfg in {"rpn": orc.RPN_MATCHER, "roi": orc.ROI_MATCHER}.items():
            idx, lab, val = orc.Matcher(**cfg)(q.clone())
            assert np.array_equal(idx.numpy(), gold[f"{name}/{cn}/idx"]), (name, cn)
            assert np.array_equal(lab.numpy(), gold[f"{name}/{cn}/label"]), (name, cn)
            assert np.array_equal(val.numpy(), gold[f"{name}/{cn}/val"]), (name, cn)
            if f"{name}/gt" in gold.files:  # fused C restatement from boxes
                ci, cl, cv = orc.iou_match_c(gold[f"{name}/gt"], gold[f"{name}/pr"], cfg["thresholds"], cfg["labels"],
                                             cfg["allow_low_quality_matches"])
                assert np.array_equal(ci, gold[f"{name}/{cn}/idx"]) and np.array_equal(cl, gold[f"{name}/{cn}/label"])
                assert np.array_equal(cv, gold[f"{name}/{cn}/val"])


def test_pairwise_iou_matches_golden_quality_matrix():
    gold = np.load(GOLD)
    for name in sorted({k.split("/")[0] for k in gold.files if k.endswith("/gt")}):
        q = orc.pairwise_iou(torch.from_numpy(gold[f"{name}/gt"]), torch.from_numpy(gold[f"{name}/pr"]))
        assert np.array_equal(q.numpy(), gold[f"{name}/q"])


def _roi_align_numpy(feat, rois, P=14, scale=1 / 16.0):
    """independent tiny pure-numpy restatement of SURVEY A.12 (fp64) used to cross-check the C oracle."""
    n, c, h, w = feat.shape
    out = np.zeros((len(rois), c, P, P))
    for r, roi in enumerate(rois):
        b = int(roi[0])
        sw, sh, ew, eh = [float(np.float32(v) * np.float32(scale) - np.float32(0.5)) for v in roi[1:]]
        rw, rh = ew - sw, eh - sh
        bh, bw = rh / P, rw / P
        gh, gw = int(np.ceil(rh / P)), int(np.ceil(rw / P))
        cnt = max(gh * gw, 1)
        for ph in range(P):
            for pw in range(P):
                acc = np.zeros(c)
                for iy in range(gh):
                    y = sh + ph * bh + (iy + .5) * bh / gh
                    for ix in range(gw):
                        x = sw + pw * bw + (ix + .5) * bw / gw
                        if y < -1 or y > h or x < -1 or x > w:
                            continue
                        yy, xx = max(y, 0), max(x, 0)
                        yl, xl = int(yy), int(xx)
                        if yl >= h - 1:
                            yh = yl = h - 1; yy = yl
                        else:
                            yh = yl + 1
                        if xl >= w - 1:
                            xh = xl = w - 1; xx = xl
                        else:
                            xh = xl + 1
                        ly, lx = yy - yl, xx - xl
                        acc += (1 - ly) * (1 - lx) * feat[b, :, yl, xl] + (1 - ly) * lx * feat[b, :, yl, xh] + \
                            ly * (1 - lx) * feat[b, :, yh, xl] + ly * lx * feat[b, :, yh, xh]
                out[r, :, ph, pw] = acc / cnt
    return out


def test_roi_align_c_vs_numpy_and_autograd():
    g = torch.Generator().manual_seed(0)
    feat = torch.randn(2, 3, 12, 17, generator=g)
    rois = torch.tensor([[0, 10.0, 10.0, 10.0, 10.0], [1, -40.0, -30.0, 90.0, 70.0], [0, 20.0, 8.0, 250.0, 180.0],
                         [1, 200.0, 150.0, 330.0, 230.0]])
    ref = _roi_align_numpy(feat.numpy().astype(np.float64), rois.numpy())
    got = orc.roi_align_forward(feat.numpy(), rois.numpy())
    assert np.allclose(got, ref, rtol=1e-4, atol=1e-5)
    assert np.all(got[0] == 0)  # zero-area RoI: empty grid, divisor max(0,1)
    f = feat.clone().requires_grad_(True)
    out = orc.roi_align(f, rois)
    w = torch.randn(out.shape, generator=g)
    (out * w).sum().backward()
    # finite-difference check of one input element
    eps = 1e-2
    i = (0, 1, 3, 4)
    fp = feat.clone(); fp[i] += eps
    fm = feat.clone(); fm[i] -= eps
    num = ((torch.from_numpy(orc.roi_align_forward(fp.numpy(), rois.numpy())) * w).sum() -
           (torch.from_numpy(orc.roi_align_forward(fm.numpy(), rois.numpy())) * w).sum()) / (2 * eps)
    assert abs(num.item() - f.grad[i].item()) < 1e-2 * max(1.0, abs(num.item()))


def test_nms_oracle_properties():
    g = torch.Generator().manual_seed(1)
    b = torch.rand(400, 4, generator=g) * 200
    b[:, 2:] = b[:, :2] + 5 + torch.rand(400, 2, generator=g) * 80
    s = torch.rand(400, generator=g)
    keep = orc.nms(b, s, 0.5)
    assert torch.all(s[keep][:-1] >= s[keep][1:])
    iou = orc.pairwise_iou(b[keep], b[keep])
    iou.fill_diagonal_(0)
    assert iou.max() <= 0.5
    # idempotence: NMS of the kept set keeps everything
    assert len(orc.nms(b[keep], s[keep], 0.5)) == len(keep)
    # every dropped box is suppressed by a higher-scored kept box
    dropped = torch.tensor(sorted(set(range(400)) - set(keep.tolist())))
    q = orc.pairwise_iou(b[dropped], b[keep])
    assert torch.all(((q > 0.5) & (s[keep][None] >= s[dropped][:, None])).any(dim=1))


def test_subsample_contract():
    g = torch.Generator().manual_seed(2)
    lab = torch.randint(-1, 2, (1000,), generator=g)
    perm = torch.randperm(1200, generator=g)   # capacity-sized permutation with out-of-range entries
    pos, neg = orc.subsample_labels(lab, 256, 0.5, 0, perm)
    assert len(pos) == min(128, int((lab == 1).sum())) and len(neg) == min(int((lab == 0).sum()), 256 - len(pos))
    assert torch.all(lab[pos] == 1) and torch.all(lab[neg] == 0)
    rank = torch.empty(1200, dtype=torch.long); rank[perm] = torch.arange(1200)
    assert torch.all(rank[pos][:-1] < rank[pos][1:])   # permutation order preserved


def test_box_codec_roundtrip_and_clamp():
    g = torch.Generator().manual_seed(3)
    src = torch.rand(100, 4, generator=g) * 100
    src[:, 2:] += src[:, :2] + 10
    tgt = torch.rand(100, 4, generator=g) * 100
    tgt[:, 2:] += tgt[:, :2] + 10
    for w in [(1.0, 1.0, 1.0, 1.0), (10.0, 10.0, 5.0, 5.0)]:
        d = orc.get_deltas(src, tgt, w)
        back = orc.apply_deltas(d, src, w)
        assert torch.allclose(back, tgt, rtol=1e-4, atol=1e-3)
    d = torch.tensor([[0.0, 0.0, 100.0, 100.0]])
    out = orc.apply_deltas(d, torch.tensor([[0.0, 0.0, 16.0, 16.0]]), (1.0, 1.0, 1.0, 1.0))
    assert torch.allclose(out[0, 2] - out[0, 0], torch.tensor(1000.0), rtol=1e-4)


def test_anchors_shape_and_order():
    a = orc.grid_anchors(38, 63)
    assert a.shape == (35910, 4)
    assert torch.allclose(a[0], torch.tensor([-22.627417, -11.313708, 22.627417, 11.313708]))
    assert torch.allclose(a[15] - a[0], torch.tensor([16.0, 0.0, 16.0, 0.0]))       # x-minor
    assert torch.allclose(a[15 * 63] - a[0], torch.tensor([0.0, 16.0, 0.0, 16.0]))  # y-major


def test_library_exports_every_declared_symbol():
    """-m 'not gpu': the C-ABI library loads and exports every symbol include/unit_hip.h declares (no compute calls)."""
    from unit_amd import _lib, build
    build.build()
    protos = _lib.parse_header()
    assert len(protos) >= 40
    l = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(l, name), name
    assert _lib.lib().unit_version() >= 100


def test_oracle_reproduces_step_golden():
    """tests/golden/step_golden.npz (tiny end-to-end S1 step, made by tests/golden/gen_step_golden.py): the oracle still
    produces the same losses, index decisions and gradients -- a regression pin of the checker itself."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_step_golden", os.path.join(os.path.dirname(__file__), "golden", "gen_step_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "step_golden.npz"))
    cfg = gen.tiny_cfg()
    m, sup, weak, perms = gen.inputs(cfg)
    assert np.array_equal(perms["rpn"][0].numpy(), gold["perm_rpn"]) and np.array_equal(perms["roi"][0].numpy(), gold["perm_roi"])
    losses, aux, p = gen.oracle_step(m, cfg, sup, weak, perms)
    for k, v in zip(gold["loss_names"], gold["losses"]):
        assert abs(losses[str(k)].item() - v) <= 1e-6 * max(1.0, abs(v)), (k, losses[str(k)].item(), v)
    assert np.array_equal(torch.stack(aux["anchor_labels"]).numpy(), gold["anchor_labels"])
    assert np.array_equal(aux["sampled"][0]["gt_classes"].numpy(), gold["roi_classes"])
    for k in gen.GRAD_KEYS:
        assert abs(p[k].grad.double().norm().item() - gold["gradnorm/" + k]) <= 1e-5 * gold["gradnorm/" + k], k
