"""Why do the bf16 step's proposals differ from the fp32 oracle's?  python tools/bf16_proposal_probe.py
R101, 2 images 600x1000, random-init weights (the full-size test fixture): objectness logits / anchor deltas of the HIP path in bf16
and in fp32 against the CPU oracle, then the proposal sets under several matching criteria."""
import sys

import torch

sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import unit_oracle as orc
from unit_amd import config, ops
from unit_amd.modeling import build_model
from unit_amd.synthetic import init_synthetic_weights, synthetic_batch

HW = (600, 1000)
cfg = config.voc_rcnn_c4_split1(101)
cfg.MODEL.DEVICE = "cuda"
model = build_model(cfg)
init_synthetic_weights(model, seed=1)
model.train()
sup, weak = synthetic_batch(2, 0, hw=HW, seed=3)
imgs = [x["image"] for x in sup]
p = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
with torch.no_grad():
    x = orc.preprocess_image(imgs, cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD)[0]
    feat = orc.resnet_c4(x, p, 101, "backbone.")
    lo, do = orc.rpn_head(feat, p)
print("oracle logits: std per image", lo.std(dim=1).tolist(), "range", lo.min().item(), lo.max().item())
rpn = model.proposal_generator
a = rpn.num_anchors
anchors = rpn.anchor_generator.grid(38, 63)
ora = orc.find_top_rpn_proposals(anchors.cpu(), lo, do, [HW] * 2, 0.7, 12000, 2000)
hw = torch.tensor([HW] * 2, dtype=torch.float32, device="cuda")


def iou_mat(b1, b2):
    return orc.pairwise_iou(b1, b2)


for dt in (torch.float32, torch.bfloat16):
    model.compute_dtype = dt
    model._ensure_ready()
    xx, _ = ops.preprocess_images([i.cuda() for i in imgs], model._pixel_mean, model._pixel_std, dt, 8, model.normalize_images)
    f, _ = model.backbone.fwd(xx)
    head, _ = rpn.rpn_head.fwd(f)
    head = head.float()
    n = head.shape[0]
    lg = head[:, :, :a].reshape(n, -1).cpu()
    dl = head[:, :, a:5 * a].reshape(n, -1, 4).cpu()
    el = (lg - lo).abs()
    ed = (dl - do.view(n, -1, 4)).abs()
    print(f"{dt}: logit abs err mean {el.mean():.4f} p99 {el.flatten().kthvalue(int(0.99 * el.numel())).values:.4f} max {el.max():.4f} | "
          f"delta abs err mean {ed.mean():.4f} max {ed.max():.4f} (delta std {do.std():.3f})")
    for i in range(n):
        t_h = set(lg[i].topk(12000).indices.tolist()); t_o = set(lo[i].topk(12000).indices.tolist())
        print(f"   image {i}: top-12000 set overlap {len(t_h & t_o) / 12000:.4f}")
    b2, s2, c2 = rpn.predict_proposals(head.cuda().to(torch.float32), anchors, hw, True)
    for i in range(n):
        hb = b2[i, : int(c2[i])].cpu()
        ob = ora[i][0]
        d = torch.cdist(hb.double(), ob.double(), p=float("inf")).min(1).values
        iou = iou_mat(hb, ob).max(1).values
        size = (hb[:, 2:] - hb[:, :2]).max(1).values
        rel = d / size
        print(f"   image {i}: partner within 0.01 px {float((d < 0.01).float().mean()):.3f} | 2 px {float((d < 2).float().mean()):.3f} | 8 px {float((d < 8).float().mean()):.3f} "
              f"| 2 % of the box size {float((rel < 0.02).float().mean()):.3f} | IoU >= 0.9 {float((iou >= 0.9).float().mean()):.3f} | IoU >= 0.8 {float((iou >= 0.8).float().mean()):.3f} "
              f"| IoU >= 0.7 {float((iou >= 0.7).float().mean()):.3f}")
    # the same with the ORACLE's deltas and the path's logits (selection only) / the oracle's logits and the path's deltas (geometry only)
    for tag, L, D in (("own logits, oracle deltas", lg, do.view(n, -1, 4)), ("oracle logits, own deltas", lo, dl)):
        pr = orc.find_top_rpn_proposals(anchors.cpu(), L, D.reshape(n, -1, 4), [HW] * 2, 0.7, 12000, 2000)
        for i in range(n):
            d = torch.cdist(pr[i][0].double(), ora[i][0].double(), p=float("inf")).min(1).values
            iou = iou_mat(pr[i][0], ora[i][0]).max(1).values
            print(f"   [{tag}] image {i}: within 2 px {float((d < 2).float().mean()):.3f} | IoU >= 0.9 {float((iou >= 0.9).float().mean()):.3f}")
