"""Model hyper-parameters of the benchmark configurations, as plain dicts.

The GPU box has no access to the reference tree, so ``bench.py`` cannot read
``configs/swin/*.py`` there; these dicts carry the same values as
``configs/_base_/models/mask_rcnn_swin_fpn.py`` + ``configs/swin/mask_rcnn_swin_tiny_patch4_window7_
mstrain_480-800_adamw_1x_coco.py`` of the reference (tests/test_config_parity.py checks them key by
key against the reference files when the reference tree is present).
"""
import copy


def swin_backbone(variant="tiny", drop_path_rate=None):
    v = {
        "tiny": dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], drop_path_rate=0.1),
        "small": dict(embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], drop_path_rate=0.2),
        "base": dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], drop_path_rate=0.3),
    }[variant]
    cfg = dict(type='SwinTransformer', window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0.,
               attn_drop_rate=0., ape=False, patch_norm=True, out_indices=(0, 1, 2, 3), use_checkpoint=False, **v)
    if drop_path_rate is not None:
        cfg['drop_path_rate'] = drop_path_rate
    return cfg


def mask_rcnn_swin(variant="tiny"):
    bb = swin_backbone(variant)
    C = bb['embed_dim']
    model = dict(
        type='MaskRCNN',
        pretrained=None,
        backbone=bb,
        neck=dict(type='FPN', in_channels=[C, 2 * C, 4 * C, 8 * C], out_channels=256, num_outs=5),
        rpn_head=dict(
            type='RPNHead', in_channels=256, feat_channels=256,
            anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0], strides=[4, 8, 16, 32, 64]),
            bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0]),
            loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
            loss_bbox=dict(type='L1Loss', loss_weight=1.0)),
        roi_head=dict(
            type='StandardRoIHead',
            bbox_roi_extractor=dict(type='SingleRoIExtractor',
                                    roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=0),
                                    out_channels=256, featmap_strides=[4, 8, 16, 32]),
            bbox_head=dict(type='Shared2FCBBoxHead', in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=80,
                           bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.],
                                           target_stds=[0.1, 0.1, 0.2, 0.2]),
                           reg_class_agnostic=False,
                           loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                           loss_bbox=dict(type='L1Loss', loss_weight=1.0)),
            mask_roi_extractor=dict(type='SingleRoIExtractor',
                                    roi_layer=dict(type='RoIAlign', output_size=14, sampling_ratio=0),
                                    out_channels=256, featmap_strides=[4, 8, 16, 32]),
            mask_head=dict(type='FCNMaskHead', num_convs=4, in_channels=256, conv_out_channels=256, num_classes=80,
                           loss_mask=dict(type='CrossEntropyLoss', use_mask=True, loss_weight=1.0))),
        train_cfg=dict(
            rpn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3,
                                   match_low_quality=True, ignore_iof_thr=-1),
                     sampler=dict(type='RandomSampler', num=256, pos_fraction=0.5, neg_pos_ub=-1, add_gt_as_proposals=False),
                     allowed_border=-1, pos_weight=-1, debug=False),
            rpn_proposal=dict(nms_pre=2000, max_per_img=1000, nms=dict(type='nms', iou_threshold=0.7), min_bbox_size=0),
            rcnn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5,
                                    match_low_quality=True, ignore_iof_thr=-1),
                      sampler=dict(type='RandomSampler', num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True),
                      mask_size=28, pos_weight=-1, debug=False)),
        test_cfg=dict(
            rpn=dict(nms_pre=1000, max_per_img=1000, nms=dict(type='nms', iou_threshold=0.7), min_bbox_size=0),
            rcnn=dict(score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100, mask_thr_binary=0.5)))
    return copy.deepcopy(model)


def cascade_mask_rcnn_swin(variant="base"):
    """configs/_base_/models/cascade_mask_rcnn_swin_fpn.py + configs/swin/cascade_mask_rcnn_swin_{tiny,small,base}_*_giou_4conv1f_
    adamw_*_coco.py: three ConvFCBBoxHead stages (4 shared convs with SyncBN + 1 fc, GIoU on decoded boxes), one FCNMaskHead per
    stage, SmoothL1 RPN regression, per-stage assigner thresholds 0.5 / 0.6 / 0.7."""
    bb = swin_backbone(variant, drop_path_rate={"tiny": 0.2, "small": 0.2, "base": 0.3}[variant])
    C = bb['embed_dim']

    def bbox_head(stds):
        return dict(type='ConvFCBBoxHead', num_shared_convs=4, num_shared_fcs=1, in_channels=256, conv_out_channels=256,
                    fc_out_channels=1024, roi_feat_size=7, num_classes=80,
                    bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=stds),
                    reg_class_agnostic=False, reg_decoded_bbox=True, norm_cfg=dict(type='SyncBN', requires_grad=True),
                    loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                    loss_bbox=dict(type='GIoULoss', loss_weight=10.0))

    def rcnn(thr):
        return dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=thr, neg_iou_thr=thr, min_pos_iou=thr,
                                  match_low_quality=False, ignore_iof_thr=-1),
                    sampler=dict(type='RandomSampler', num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True),
                    mask_size=28, pos_weight=-1, debug=False)
    model = dict(
        type='CascadeRCNN',
        pretrained=None,
        backbone=bb,
        neck=dict(type='FPN', in_channels=[C, 2 * C, 4 * C, 8 * C], out_channels=256, num_outs=5),
        rpn_head=dict(
            type='RPNHead', in_channels=256, feat_channels=256,
            anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0], strides=[4, 8, 16, 32, 64]),
            bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0]),
            loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
            loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0)),
        roi_head=dict(
            type='CascadeRoIHead', num_stages=3, stage_loss_weights=[1, 0.5, 0.25],
            bbox_roi_extractor=dict(type='SingleRoIExtractor',
                                    roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=0),
                                    out_channels=256, featmap_strides=[4, 8, 16, 32]),
            bbox_head=[bbox_head([0.1, 0.1, 0.2, 0.2]), bbox_head([0.05, 0.05, 0.1, 0.1]), bbox_head([0.033, 0.033, 0.067, 0.067])],
            mask_roi_extractor=dict(type='SingleRoIExtractor',
                                    roi_layer=dict(type='RoIAlign', output_size=14, sampling_ratio=0),
                                    out_channels=256, featmap_strides=[4, 8, 16, 32]),
            mask_head=dict(type='FCNMaskHead', num_convs=4, in_channels=256, conv_out_channels=256, num_classes=80,
                           loss_mask=dict(type='CrossEntropyLoss', use_mask=True, loss_weight=1.0))),
        train_cfg=dict(
            rpn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3,
                                   match_low_quality=True, ignore_iof_thr=-1),
                     sampler=dict(type='RandomSampler', num=256, pos_fraction=0.5, neg_pos_ub=-1, add_gt_as_proposals=False),
                     allowed_border=0, pos_weight=-1, debug=False),
            rpn_proposal=dict(nms_across_levels=False, nms_pre=2000, nms_post=2000, max_per_img=2000,
                              nms=dict(type='nms', iou_threshold=0.7), min_bbox_size=0),
            rcnn=[rcnn(0.5), rcnn(0.6), rcnn(0.7)]),
        test_cfg=dict(
            rpn=dict(nms_across_levels=False, nms_pre=1000, nms_post=1000, max_per_img=1000,
                     nms=dict(type='nms', iou_threshold=0.7), min_bbox_size=0),
            rcnn=dict(score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100, mask_thr_binary=0.5)))
    return copy.deepcopy(model)


# optimizer of configs/swin/mask_rcnn_swin_tiny_..._1x_coco.py:64-67
OPTIMIZER = dict(type='AdamW', lr=0.0001, betas=(0.9, 0.999), weight_decay=0.05,
                 no_decay_keys=('absolute_pos_embed', 'relative_position_bias_table', 'norm'))
