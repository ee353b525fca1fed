"""Operator surface of the hot path.

``RoIAlign``, ``roi_align``, ``nms`` and ``batched_nms`` keep the names and signatures of
``mmcv.ops`` because the reference looks them up by name
(``getattr(ops, 'RoIAlign')``, base_roi_extractor.py:51-52).
"""
from .functional import (add_layer_norm, add_scaled, bias_gelu, conv3x3, layer_norm, linear, patch_im2row,  # noqa: F401
                         patch_merge_layer_norm, rel_bias_expand, upsample_add, window_attention)
from .nms import batched_nms, batched_nms_static, batched_nms_static_multi, nms, nms_static  # noqa: F401
from .roi_align import RoIAlign, roi_align, roi_align_multilevel, roi_align_multilevel_group  # noqa: F401
from .targets import (bbox_targets, delta2bbox, map_roi_levels, max_iou_assign, paste_masks, random_sample,  # noqa: F401
                      random_sample_raw, regress_by_class, roi_targets_pack, RoiStageBuffers, rpn_topk_decode)
from .batchnorm import batch_norm  # noqa: F401
from .losses import bbox_loss, mask_loss, rpn_flatten, rpn_loss  # noqa: F401
