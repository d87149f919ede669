from .backbone import ResNet, build_resnet_backbone  # noqa: F401
from .box_head import Res5BoxHead, Res5BoxHeadWithMask  # noqa: F401
from .matcher import Matcher  # noqa: F401
from .rpn import WSRPN  # noqa: F401
from .fast_rcnn import SupervisedDetectorOutputsBase, SupervisedDetectorOutputsFineTune, WeakDetectorOutputsBase  # noqa: F401
from .roi_heads import WSROIHeadNoMeta, WSROIHeadFineTune, WSROIHeadNoMetaWithMask, WSROIHeadWithMaskFineTune  # noqa: F401
from .mask_head import MaskRCNNConvUpsampleHeadWithSimilarity, MaskRCNNConvUpsampleHeadWithFineTune  # noqa: F401
from .rcnn import WeaklySupervisedRCNNNoMeta, build_model  # noqa: F401
