"""DTOID detector: forward (all templates) and the forward/backward of the online finetune step, behind the
reference's DtoidNet / Network interfaces (/root/reference/python/ossid/models/dtoid/)."""
from .model import DtoidNet, DtoidConfig, normalizeImageRange, binary_iou  # noqa: F401
from .network import (Network, BBoxTransform, ClipBoxes, ClassificationModel, RegressionModel, ImageFeatExtract,  # noqa: F401
                      TemplateFeatExtract, TemplateFeatExtractGlobal, CorrelationModel)
from .loss import DetectionLoss, calc_iou  # noqa: F401
from .anchors import Anchors  # noqa: F401
