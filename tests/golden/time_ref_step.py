"""One reference-EXECUTED timing point: the reference's own `WeaklySupervisedRCNNNoMeta.forward`
(/root/reference/modeling/meta_arch/rcnn.py:433-491) + `sum(losses).backward()` (engine/defaults.py:280) on the "s1" fixture -- its own RPN / ROI-head /
predictor / loss code orchestrating the Detectron2 blocks the image lacks (supplied by the CPU oracle, as for the golden vectors) -- timed beside the oracle's
own step on the same inputs. Runs only in the authoring container (imports /root/reference by file); the numbers go into BASELINE.md.
  python tests/golden/time_ref_step.py [iterations]"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_unit_golden as G  # noqa: E402  (installs the Detectron2 stand-ins, loads the reference's modules)
import gen_ref_step as S  # noqa: E402

orc = G.orc


def main(iters=3, name="s1"):
    torch.set_num_threads(os.cpu_count())
    G.load_meta_arch()          # the reference's meta_arch/rcnn.py, loaded by file like the other modules
    cfg, model, sup, weak, perms, masks = S.step_inputs(name)
    p = S.oracle_params(model)
    ocfg = S.oracle_cfg(cfg)
    ref, trace = G.build_reference_model(p, ocfg, perms, roi_cls=cfg.MODEL.ROI_HEADS.NAME, pred_cls=cfg.MODEL.ROI_HEADS.FAST_RCNN.NAME, mask_cls=None)
    ref.roi_heads.visual_threshold = ocfg["visual_threshold"]
    ref.roi_heads.box_predictor._freeze_layers(list(cfg.MODEL.FREEZE_LAYERS.FAST_RCNN))
    ref.train()
    t_ref, t_orc = [], []
    for it in range(iters + 1):
        for q in p.values():
            q.grad = None
        t0 = time.perf_counter()
        losses = ref(G.to_d2_inputs(sup, masks), G.to_d2_inputs(weak) if weak else None)
        sum(losses.values()).backward()
        t1 = time.perf_counter()
        for q in p.values():
            q.grad = None
        lo, _ = orc.step_losses(p, [x["image"] for x in sup], [x["instances"].gt_boxes.tensor for x in sup], [x["instances"].gt_classes for x in sup],
                                [x["image"] for x in weak], [x["instances"].gt_classes for x in weak], perms, ocfg)
        sum(lo.values()).backward()
        t2 = time.perf_counter()
        if it:          # (first iteration: warm-up)
            t_ref.append(t1 - t0)
            t_orc.append(t2 - t1)
    hw = tuple(sup[0]["image"].shape[-2:])
    print(f"case {name}: {len(sup)} supervised + {len(weak)} weak images of {hw[0]}x{hw[1]}, R{cfg.MODEL.RESNETS.DEPTH}-C4, {cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE} RoIs/image, "
          f"{torch.get_num_threads()} threads")
    print(f"reference forward + backward (its own orchestration over oracle-supplied Detectron2 blocks): {min(t_ref) * 1e3:.0f} ms per step "
          f"({len(sup) / min(t_ref):.2f} supervised images/s)")
    print(f"oracle step (the CPU restatement bench.py times as cpu_baseline kind 'port'):               {min(t_orc) * 1e3:.0f} ms per step "
          f"({len(sup) / min(t_orc):.2f} supervised images/s)")
    print("losses agree:", {k: (round(float(losses[k]), 6), round(float(lo[k]), 6)) for k in sorted(lo)})


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
