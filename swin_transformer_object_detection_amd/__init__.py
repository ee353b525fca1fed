"""MI355X-native Swin detection hot path.

The package directory is also reachable as ``swin-transformer-object-detection_amd``
(a symlink): Python package names cannot contain hyphens.
"""
__version__ = "0.1.0"

from . import ops  # noqa: E402,F401
from .backbone import SwinTransformer  # noqa: E402,F401  (registers into BACKBONES)
from .fpn import FPN  # noqa: E402,F401                   (registers into NECKS)
from .detector import MaskRCNN, build_detector  # noqa: E402,F401
from .registry import BACKBONES, NECKS, ROI_EXTRACTORS, HEADS, DETECTORS, build_backbone, build_neck  # noqa: E402,F401
from .config import Config  # noqa: E402,F401
