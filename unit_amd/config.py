"""Minimal CfgNode with the Detectron2-v0.3 + UniT key schema the hot path reads.

Mirrors the keys consumed by the reference's `from_config` classmethods (configs/default_config.py:4-106 adds the UniT
keys onto detectron2's get_cfg(); SURVEY.md appendix B lists every value that fixes a shape). Only keys on the hot path
exist; a user yaml in the reference's schema can be merged with `merge_from_file` (`_BASE_` supported).
"""
import copy
import os


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def merge_from_dict(self, d):
        for k, v in d.items():
            if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                self[k].merge_from_dict(v)
            else:
                self[k] = _coerce(_to_cfg(_literal(v)), self.get(k))
        return self

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            d = yaml.safe_load(f) or {}
        base = d.pop("_BASE_", None)
        if base:
            self.merge_from_file(os.path.join(os.path.dirname(path), base))
        return self.merge_from_dict(d)

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0
        for k, v in zip(opts[0::2], opts[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = _coerce(_literal(v), node.get(parts[-1]))
        return self


CN = CfgNode


def _literal(v):
    """yacs semantics: a string leaf is a python literal when it parses as one ("(12000, 24000)" -> tuple, "0.02" -> float);
    anything else stays the raw string ("WSRPN", a path)."""
    if isinstance(v, str):
        import ast
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v
    return v


def _coerce(v, old):
    """yacs `_check_and_coerce_cfg_value_type`: tuple <-> list and int -> float follow the type of the value being replaced."""
    if old is None or isinstance(old, CfgNode) or type(v) is type(old):
        return v
    if isinstance(old, tuple) and isinstance(v, list):
        return tuple(v)
    if isinstance(old, list) and isinstance(v, tuple):
        return list(v)
    if isinstance(old, float) and isinstance(v, int) and not isinstance(v, bool):
        return float(v)
    return v


def _to_cfg(v):
    if isinstance(v, dict) and not isinstance(v, CfgNode):
        return CfgNode({k: _to_cfg(x) for k, x in v.items()})
    return v


def get_cfg():
    """detectron2.config.get_cfg() defaults (v0.3) for the keys the C4 hot path reads [d2-ext], then add_config."""
    c = CN()
    c.MODEL = CN()
    c.MODEL.DEVICE = "cuda"
    c.MODEL.META_ARCHITECTURE = "WeaklySupervisedRCNNNoMeta"
    c.MODEL.MASK_ON = False
    c.MODEL.WEIGHTS = ""
    c.MODEL.PIXEL_MEAN = [103.530, 116.280, 123.675]
    c.MODEL.PIXEL_STD = [1.0, 1.0, 1.0]
    c.MODEL.LOAD_PROPOSALS = False
    c.MODEL.BACKBONE = CN(NAME="build_resnet_backbone", FREEZE_AT=2)
    c.MODEL.RESNETS = CN(DEPTH=101, OUT_FEATURES=["res4"], NUM_GROUPS=1, WIDTH_PER_GROUP=64, STRIDE_IN_1X1=True,
                         RES2_OUT_CHANNELS=256, STEM_OUT_CHANNELS=64, NORM="FrozenBN", RES5_DILATION=1)
    c.MODEL.ANCHOR_GENERATOR = CN(NAME="DefaultAnchorGenerator", SIZES=[[32, 64, 128, 256, 512]], ASPECT_RATIOS=[[0.5, 1.0, 2.0]], OFFSET=0.0)
    c.MODEL.PROPOSAL_GENERATOR = CN(NAME="WSRPN", MIN_SIZE=0, WEAK_RPN_SCORE_TRESHOLD=0.0)
    c.MODEL.RPN = CN(HEAD_NAME="StandardRPNHead", IN_FEATURES=["res4"], BOUNDARY_THRESH=-1, IOU_THRESHOLDS=[0.3, 0.7],
                     IOU_LABELS=[0, -1, 1], BATCH_SIZE_PER_IMAGE=256, POSITIVE_FRACTION=0.5, BBOX_REG_LOSS_TYPE="smooth_l1",
                     BBOX_REG_LOSS_WEIGHT=1.0, BBOX_REG_WEIGHTS=(1.0, 1.0, 1.0, 1.0), SMOOTH_L1_BETA=0.0, LOSS_WEIGHT=1.0,
                     PRE_NMS_TOPK_TRAIN=12000, PRE_NMS_TOPK_TEST=6000, POST_NMS_TOPK_TRAIN=2000, POST_NMS_TOPK_TEST=1000,
                     NMS_THRESH=0.7)
    c.MODEL.ROI_HEADS = CN(NAME="WSROIHeadNoMeta", NUM_CLASSES=20, IN_FEATURES=["res4"], IOU_THRESHOLDS=[0.5], IOU_LABELS=[0, 1],
                           BATCH_SIZE_PER_IMAGE=512, POSITIVE_FRACTION=0.25, SCORE_THRESH_TEST=0.05, NMS_THRESH_TEST=0.5,
                           PROPOSAL_APPEND_GT=True)
    c.MODEL.ROI_BOX_HEAD = CN(NAME="Res5BoxHead", BBOX_REG_WEIGHTS=(10.0, 10.0, 5.0, 5.0), SMOOTH_L1_BETA=0.0,
                              POOLER_RESOLUTION=14, POOLER_SAMPLING_RATIO=0, POOLER_TYPE="ROIAlignV2", CLS_AGNOSTIC_BBOX_REG=False,
                              BBOX_REG_LOSS_TYPE="smooth_l1", BBOX_REG_LOSS_WEIGHT=1.0, TRAIN_ON_PRED_BOXES=False)
    c.MODEL.ROI_MASK_HEAD = CN(NAME="MaskRCNNConvUpsampleHeadWithSimilarity", POOLER_RESOLUTION=14, POOLER_SAMPLING_RATIO=0,
                               POOLER_TYPE="None", NUM_CONV=0, CONV_DIM=256, NORM="", CLS_AGNOSTIC_MASK=False)
    c.INPUT = CN(FORMAT="BGR", MIN_SIZE_TRAIN=(800,), MAX_SIZE_TRAIN=1333, MIN_SIZE_TEST=800, MAX_SIZE_TEST=1333)
    c.DATASETS = CN(TRAIN=("voc_2007_trainval_base1",), TEST=("voc_2007_test_all1",))
    c.DATALOADER = CN(NUM_WORKERS=2)
    c.SOLVER = CN(IMS_PER_BATCH=8, BASE_LR=0.02, MOMENTUM=0.9, NESTEROV=False, WEIGHT_DECAY=1e-4, WEIGHT_DECAY_NORM=0.0,
                  WEIGHT_DECAY_BIAS=1e-4, BIAS_LR_FACTOR=1.0, STEPS=(12000, 24000), MAX_ITER=30000, WARMUP_ITERS=100,
                  WARMUP_FACTOR=1.0 / 1000, GAMMA=0.1, CHECKPOINT_PERIOD=500)
    c.TEST = CN(DETECTIONS_PER_IMAGE=100, EVAL_PERIOD=0, AUG=CN(ENABLED=False))
    c.SEED = -1
    add_config(c)
    return c


def add_config(cfg):
    """UniT keys -- restates configs/default_config.py:4-106 for the keys on the C4 hot path (same names, same defaults)."""
    _C = cfg
    _C.MODEL.FREEZE_LAYERS = CN(ROI_HEADS=[], META_ARCH=[], FAST_RCNN=[], BOX_HEAD=[], MASK_HEAD=[])
    _C.MODEL.ROI_HEADS.EMBEDDING_PATH = ""
    _C.MODEL.ROI_HEADS.FINETUNE_TERMS = CN(CLASSIFIER=["lingual", "visual"], BBOX=["lingual", "visual"], MASK=["lingual", "visual"])
    _C.MODEL.ROI_HEADS.WEAK_CLASSIFIER_PROPOSAL_DIVISOR = 1
    _C.MODEL.ROI_HEADS.MULTI_BOX_HEAD = False
    _C.MODEL.ROI_HEADS.TRAIN_USING_WEAK = False
    _C.MODEL.ROI_HEADS.TRAIN_PROPOSAL_REGRESSOR = False
    _C.MODEL.ROI_HEADS.WEAK_PROPOSAL_DIVISOR = 1
    _C.MODEL.ROI_HEADS.FAST_RCNN = CN(NAME="SupervisedDetectorOutputsBase")
    _C.MODEL.ROI_HEADS.FAST_RCNN.WEAK_DETECTOR = CN(
        NAME="WeakDetectorOutputsBase", DETECTOR_TEMP=1.0, CLASSIFIER_TEMP=1.0, REGRESSION_BRANCH=False, OICR_ITER=3,
        FG_THRESHOLD=0.5, BG_THRESHOLD=0.1, MIL_MULTIPLIER=1.0, TYPE="OICR", OICR_REGRESSION_BRANCH=False,
        NUM_KMEANS_CLUSTER=3, GRAPH_IOU_THRESHOLD=0.4, MAX_PC_NUM=5)
    _C.MODEL.ROI_HEADS.VISUAL_ATTENTION_HEAD = CN(VISUAL_SIMILARITY_THRESHOLD=0.02, SIMILARITY_COMBINATION="Sum", TOPK=5)
    _C.INPUT.NORMALIZE_IMAGES = False
    _C.DATASETS.WEAK_CLASSIFIER_MUTLIPLIER = 1.0
    _C.DATASETS.BASE_MULTIPLIER = 1
    _C.DATASETS.FEWSHOT = CN(BASE_CLASSES_ID=[0, 1, 3, 4, 6, 7, 8, 10, 11, 12, 14, 15, 16, 18, 19], NOVEL_CLASSES_ID=[2, 5, 9, 13, 17])
    _C.SOLVER.REFINEMENT_LR_FACTOR = 1.0
    _C.SOLVER.MIL_LR_FACTOR = 1.0
    _C.SOLVER.DELTA_LR_FACTOR = 1.0
    return cfg


def voc_rcnn_c4_split1(depth=101):
    """configs/VOC/VOC-RCNN-101-C4-split1.yaml restated as a preset (R50: DEPTH 50)."""
    c = get_cfg()
    c.MODEL.RESNETS.DEPTH = depth
    c.MODEL.ROI_HEADS.MULTI_BOX_HEAD = True
    c.MODEL.ROI_HEADS.FAST_RCNN.WEAK_DETECTOR.DETECTOR_TEMP = 2.0
    c.MODEL.ROI_HEADS.FAST_RCNN.WEAK_DETECTOR.REGRESSION_BRANCH = False
    c.SOLVER.IMS_PER_BATCH = 8
    c.INPUT.MIN_SIZE_TRAIN = (480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800)          # yaml:27-29 (ResizeShortestEdge "choice", max 1333)
    return c


def voc_rcnn_c4_split1_ft(depth=101):
    """configs/VOC/FT/1_shot/VOC-RCNN-101-C4-split1-ft.yaml restated (freeze lists :6-9)."""
    c = voc_rcnn_c4_split1(depth)
    c.MODEL.ROI_HEADS.NAME = "WSROIHeadFineTune"
    c.MODEL.ROI_HEADS.FAST_RCNN.NAME = "SupervisedDetectorOutputsFineTune"
    c.MODEL.FREEZE_LAYERS.META_ARCH = ["backbone", "proposal_generator"]
    c.MODEL.FREEZE_LAYERS.ROI_HEADS = ["box_pooler", "box_head", "weak_box_head"]
    c.MODEL.FREEZE_LAYERS.FAST_RCNN = ["weak_detector_head", "cls_score_delta", "bbox_pred_delta", "embeddings"]
    c.SOLVER.BASE_LR, c.SOLVER.STEPS, c.SOLVER.MAX_ITER, c.SOLVER.WARMUP_ITERS, c.SOLVER.CHECKPOINT_PERIOD = 0.001, (50,), 50, 0, 50
    return c


COCO_SPLIT1_NOVEL = [0, 1, 2, 3, 4, 5, 6, 8, 14, 15, 16, 17, 18, 19, 39, 56, 57, 58, 60, 62]       # the 20 VOC classes inside COCO
COCO_SPLIT1_BASE = [i for i in range(80) if i not in COCO_SPLIT1_NOVEL]


def coco_rcnn_c4_split1(depth=50):
    """configs/COCO/COCO-RCNN-50-C4-split1.yaml restated: K = 80, ONE Res5 head (MULTI_BOX_HEAD False), 60 base / 20 novel classes
    (:13-14,38-39)."""
    c = get_cfg()
    c.MODEL.RESNETS.DEPTH = depth
    c.MODEL.ROI_HEADS.NUM_CLASSES = 80
    c.MODEL.ROI_HEADS.MULTI_BOX_HEAD = False
    c.MODEL.ROI_HEADS.FAST_RCNN.WEAK_DETECTOR.DETECTOR_TEMP = 2.0
    c.DATASETS.TRAIN = ("coco_base_training_query_train",)
    c.DATASETS.TEST = ("coco_base_training_query_val",)
    c.DATASETS.FEWSHOT.BASE_CLASSES_ID = list(COCO_SPLIT1_BASE)
    c.DATASETS.FEWSHOT.NOVEL_CLASSES_ID = list(COCO_SPLIT1_NOVEL)
    c.SOLVER.STEPS, c.SOLVER.MAX_ITER = (210000, 250000), 270000
    return c


def coco_rcnn_c4_split1_segm(depth=50):
    """configs/COCO/COCO-RCNN-50-C4-split1-segm.yaml restated (BASELINE config 5): the above + MASK_ON, WSROIHeadNoMetaWithMask,
    Res5BoxHeadWithMask, MaskRCNNConvUpsampleHeadWithSimilarity on the un-pooled res5 map (ROI_MASK_HEAD.POOLER_TYPE "None")."""
    c = coco_rcnn_c4_split1(depth)
    c.MODEL.MASK_ON = True
    c.MODEL.ROI_HEADS.NAME = "WSROIHeadNoMetaWithMask"
    c.MODEL.ROI_BOX_HEAD.NAME = "Res5BoxHeadWithMask"
    c.MODEL.ROI_MASK_HEAD.NAME = "MaskRCNNConvUpsampleHeadWithSimilarity"
    c.MODEL.ROI_MASK_HEAD.POOLER_TYPE = "None"
    return c


def coco_rcnn_c4_split1_segm_ft(depth=50):
    """configs/COCO/COCO-RCNN-50-C4-split1-segm-ft.yaml restated: fine-tune ROI heads / predictor / mask head, two Res5 heads,
    freeze lists :10-14, visual similarity threshold 0.04 (:31)."""
    c = coco_rcnn_c4_split1_segm(depth)
    c.MODEL.ROI_HEADS.NAME = "WSROIHeadWithMaskFineTune"
    c.MODEL.ROI_HEADS.MULTI_BOX_HEAD = True
    c.MODEL.ROI_HEADS.FAST_RCNN.NAME = "SupervisedDetectorOutputsFineTune"
    c.MODEL.ROI_MASK_HEAD.NAME = "MaskRCNNConvUpsampleHeadWithFineTune"
    c.MODEL.ROI_HEADS.VISUAL_ATTENTION_HEAD.VISUAL_SIMILARITY_THRESHOLD = 0.04
    c.MODEL.FREEZE_LAYERS.META_ARCH = ["backbone"]
    c.MODEL.FREEZE_LAYERS.ROI_HEADS = ["box_pooler", "weak_box_head"]
    c.MODEL.FREEZE_LAYERS.FAST_RCNN = ["weak_detector_head", "cls_score_delta", "bbox_pred_delta", "embeddings"]
    c.MODEL.FREEZE_LAYERS.MASK_HEAD = ["deconv", "deconv_relu", "predictor"]
    c.DATASETS.TRAIN = ("coco_fine_tuning_query_train",)
    c.SOLVER.BASE_LR, c.SOLVER.STEPS, c.SOLVER.MAX_ITER, c.SOLVER.WARMUP_ITERS = 0.001, (800,), 1000, 0
    return c
