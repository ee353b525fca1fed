"""Callers of the hot path: the Mask R-CNN training step around backbone -> FPN -> NMS -> RoIAlign.

The hot path is the Swin backbone, the FPN and the mmcv ops; to measure the
reference's headline metric (Mask R-CNN Swin-T training images/s) those kernels have to be
driven by a real ``forward_train``.  This file restates that orchestration with the reference's
wiring and registry names so ``configs/swin/mask_rcnn_*.py`` build unchanged:

    TwoStageDetector.forward_train     mmdet/models/detectors/two_stage.py:105-167
    RPNHead / AnchorHead loss+targets  mmdet/models/dense_heads/rpn_head.py:27-80, anchor_head.py:175-493
    RPNHead._get_bboxes -> batched_nms rpn_head.py:82-236
    MaxIoUAssigner / RandomSampler     mmdet/core/bbox/assigners/max_iou_assigner.py, samplers/random_sampler.py
    StandardRoIHead.forward_train      mmdet/models/roi_heads/standard_roi_head.py:70-194
    SingleRoIExtractor                 roi_heads/roi_extractors/single_level_roi_extractor.py:32-108
    Shared2FCBBoxHead / FCNMaskHead    roi_heads/bbox_heads/convfc_bbox_head.py, mask_heads/fcn_mask_head.py
    mask_target                        mmdet/core/mask/mask_target.py:66-122 (kept on the device)

The dense layers of the heads are plain library GEMM/conv calls (SURVEY 8(f) rank 1 is where they
get their own kernels); the hot-path ops (``ops.batched_nms``, ``ops.RoIAlign``) are the HIP ones.
Everything between the kernels runs on the device: the reference's numpy round trips in
``mask_target`` are gone.
"""
import contextlib
import os
import ctypes
import math

import numpy as np
import torch

from ._lib import half_dtype as _H
import torch.nn as nn
import torch.nn.functional as F

from . import mixed, ops
from .registry import BACKBONES, DETECTORS, HEADS, NECKS, ROI_EXTRACTORS, build_from_cfg, build_roi_extractor


def _cfg_get(cfg, key, default=None):
    if cfg is None:
        return default
    return cfg.get(key, default) if isinstance(cfg, dict) else getattr(cfg, key, default)


# ------------------------------------------------------------------------------------------
# box utilities (mmdet/core/bbox)
# ------------------------------------------------------------------------------------------
def bbox_overlaps(b1, b2, eps=1e-6):
    """IoU matrix (len(b1), len(b2)); mmdet/core/bbox/iou_calculators/iou2d_calculator.py."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[None, :, :2])
    rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = (a1[:, None] + a2[None, :] - inter).clamp(min=eps)
    return inter / union


def bbox2delta(proposals, gt, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)):
    """delta_xywh_bbox_coder.py:82-130."""
    px = (proposals[:, 0] + proposals[:, 2]) * 0.5
    py = (proposals[:, 1] + proposals[:, 3]) * 0.5
    pw = proposals[:, 2] - proposals[:, 0]
    ph = proposals[:, 3] - proposals[:, 1]
    gx = (gt[:, 0] + gt[:, 2]) * 0.5
    gy = (gt[:, 1] + gt[:, 3]) * 0.5
    gw = gt[:, 2] - gt[:, 0]
    gh = gt[:, 3] - gt[:, 1]
    d = torch.stack([(gx - px) / pw, (gy - py) / ph, torch.log(gw / pw), torch.log(gh / ph)], dim=-1)
    return (d - d.new_tensor(means)) / d.new_tensor(stds)


def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None, wh_ratio_clip=16 / 1000):
    """delta_xywh_bbox_coder.py:189-237 for (N,4) rois / deltas."""
    d = deltas * deltas.new_tensor(stds) + deltas.new_tensor(means)
    mr = abs(math.log(wh_ratio_clip))
    dw = d[:, 2].clamp(min=-mr, max=mr)
    dh = d[:, 3].clamp(min=-mr, max=mr)
    px = (rois[:, 0] + rois[:, 2]) * 0.5
    py = (rois[:, 1] + rois[:, 3]) * 0.5
    pw = rois[:, 2] - rois[:, 0]
    ph = rois[:, 3] - rois[:, 1]
    gw, gh = pw * dw.exp(), ph * dh.exp()
    gx, gy = px + pw * d[:, 0], py + ph * d[:, 1]
    b = torch.stack([gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5], dim=-1)
    if max_shape is not None:
        b[:, 0::2] = b[:, 0::2].clamp(min=0, max=float(max_shape[1]))
        b[:, 1::2] = b[:, 1::2].clamp(min=0, max=float(max_shape[0]))
    return b


_ROI_IDX = {}        # (sizes, device) -> constant batch-index column: fixed-size samples repeat the same sizes every step


def bbox2roi(bbox_list):
    """mmdet/core/bbox/transforms.py:69-77: rois = [batch index as float, x1, y1, x2, y2], images concatenated in order.
    The index column depends only on the per-image counts, which are fixed in training (512 / 128 slots per image): it is built
    once and cached, so a call is two concatenations instead of two per image plus one."""
    if not bbox_list:
        return torch.zeros((0, 5))
    sizes = tuple(int(b.size(0)) for b in bbox_list)
    b0 = bbox_list[0]
    key = (sizes, b0.device, b0.dtype)
    col = _ROI_IDX.get(key)
    if col is None:
        col = torch.cat([torch.full((n, 1), float(i)) for i, n in enumerate(sizes)], 0).to(device=b0.device, dtype=b0.dtype)
        if len(_ROI_IDX) < 64:
            _ROI_IDX[key] = col
    boxes = torch.cat([b[:, :4] for b in bbox_list], 0) if len(bbox_list) > 1 else b0[:, :4]
    return torch.cat([col, boxes], dim=1)


class AnchorGenerator:
    """anchor_generator.py:161-185, 255-270 (scale_major, center_offset 0)."""

    def __init__(self, strides, ratios, scales):
        self.strides = list(strides)
        self.ratios = torch.tensor(ratios, dtype=torch.float32)
        self.scales = torch.tensor(scales, dtype=torch.float32)
        self.num_base_anchors = len(ratios) * len(scales)
        self._cache = {}

    def _base(self, stride):
        h_r = torch.sqrt(self.ratios)
        w_r = 1 / h_r
        ws = (stride * w_r[:, None] * self.scales[None, :]).view(-1)
        hs = (stride * h_r[:, None] * self.scales[None, :]).view(-1)
        return torch.stack([-0.5 * ws, -0.5 * hs, 0.5 * ws, 0.5 * hs], dim=-1)

    def grid_anchors(self, featmap_sizes, device):
        key = (tuple(featmap_sizes), str(device))
        if key not in self._cache:
            out = []
            for (h, w), s in zip(featmap_sizes, self.strides):
                base = self._base(s).to(device)
                sx = torch.arange(0, w, device=device, dtype=torch.float32) * s
                sy = torch.arange(0, h, device=device, dtype=torch.float32) * s
                yy, xx = torch.meshgrid(sy, sx, indexing='ij')
                shifts = torch.stack([xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)], dim=-1)
                out.append((shifts[:, None, :] + base[None, :, :]).view(-1, 4))
            self._cache[key] = out
        return self._cache[key]

    def grid_anchors_cat(self, featmap_sizes, device):
        """All levels concatenated (anchor_head.py:451 `torch.cat(anchor_list[i])`), cached like the per-level lists."""
        key = ('cat', tuple(featmap_sizes), str(device))
        if key not in self._cache:
            self._cache[key] = torch.cat(self.grid_anchors(featmap_sizes, device), 0).contiguous()
        return self._cache[key]


def max_iou_assign(bboxes, gt_bboxes, pos_iou_thr, neg_iou_thr, min_pos_iou, match_low_quality=True, gt_labels=None, feeds=None):
    """MaxIoUAssigner.assign_wrt_overlaps (max_iou_assigner.py:130-212), ignore_iof_thr=-1.  ``feeds`` (n,) bool: boxes
    that take part in the per-gt maximum (the reference never sees the others: padding slots, anchors outside
    ``allowed_border``)."""
    n = bboxes.size(0)
    assigned = bboxes.new_full((n,), -1, dtype=torch.long)
    if gt_bboxes.size(0) == 0 or n == 0:
        assigned[:] = 0
        labels = None if gt_labels is None else bboxes.new_full((n,), -1, dtype=torch.long)
        return assigned, bboxes.new_zeros((n,)), labels
    overlaps = bbox_overlaps(gt_bboxes, bboxes)
    max_ov, argmax = overlaps.max(dim=0)
    gt_max, _ = (overlaps if feeds is None else torch.where(feeds[None, :], overlaps, overlaps.new_full((), -1.0))).max(dim=1)
    # masked writes as torch.where: boolean-mask assignment would cost a device->host sync each
    zero = torch.zeros_like(assigned)
    assigned = torch.where((max_ov >= 0) & (max_ov < neg_iou_thr), zero, assigned)
    assigned = torch.where(max_ov >= pos_iou_thr, argmax + 1, assigned)
    if match_low_quality:
        # later gts override earlier ones, as the reference's python loop does: take the LAST gt whose
        # row attains its own maximum at this box
        hit = (overlaps == gt_max[:, None]) & (gt_max[:, None] >= min_pos_iou)
        idx = torch.arange(1, gt_bboxes.size(0) + 1, device=bboxes.device)[:, None].expand_as(hit)
        last = torch.where(hit, idx, torch.zeros_like(idx)).max(dim=0)[0]
        assigned = torch.where(last > 0, last, assigned)
    labels = None
    if gt_labels is not None:
        labels = torch.where(assigned > 0, gt_labels[(assigned - 1).clamp(min=0)], assigned.new_full((n,), -1))
    return assigned, max_ov, labels


def random_sample(assigned, num, pos_fraction):
    """RandomSampler (random_sampler.py:31-78), neg_pos_ub=-1."""
    pos_inds = torch.nonzero(assigned > 0, as_tuple=False).view(-1)
    num_pos = int(num * pos_fraction)
    if pos_inds.numel() > num_pos:
        pos_inds = pos_inds[torch.randperm(pos_inds.numel(), device=pos_inds.device)[:num_pos]]
    neg_inds = torch.nonzero(assigned == 0, as_tuple=False).view(-1)
    num_neg = num - pos_inds.numel()
    if neg_inds.numel() > num_neg:
        neg_inds = neg_inds[torch.randperm(neg_inds.numel(), device=neg_inds.device)[:num_neg]]
    return pos_inds, neg_inds


def sample_static(assigned, num, pos_fraction):
    """RandomSampler (random_sampler.py:31-78) with a FIXED-size result and no host synchronisation.

    Same distribution as the reference: a uniformly random subset of min(n_pos, num*pos_fraction) positives,
    then uniformly random negatives up to ``num`` in total.  Implemented with random keys and two top-k
    selections instead of nonzero + randperm (whose output sizes would have to travel to the host).
    Returns (idx (k,), is_pos (k,), valid (k,)) with k = min(num, n); positives come first."""
    n = assigned.numel()
    key = torch.rand(n, device=assigned.device)
    is_p = assigned > 0
    num_pos = min(int(num * pos_fraction), n)
    if num_pos > 0:
        posk = torch.where(is_p, key, key.new_full((), 2.0))
        kth = torch.topk(posk, num_pos, largest=False, sorted=True).values[-1]
        pos_sel = is_p & (posk <= kth)
    else:
        pos_sel = torch.zeros_like(is_p)
    comb = torch.where(pos_sel, key, torch.where(assigned == 0, key + 1.0, key.new_full((), 3.0)))
    vals, idx = torch.topk(comb, min(num, n), largest=False, sorted=True)
    return idx, vals < 1.0, vals < 3.0


def assign_and_sample(bboxes, gt_bboxes, a_cfg, s_cfg, means, stds, gt_labels=None, num_leading_gt=0, valid=None, bg_label=0):
    """assigner.assign + sampler.sample + target encoding of one image (anchor_head.py:213-247,
    standard_roi_head.py:83-93 + bbox_head.py:140-186) with a fixed-size result:
    (boxes (num,4), deltas (num,4), labels (num,), gt_inds (num,), is_pos (num,), valid (num,), inds (num,),
    flags (num,) uint8: bit 0 used, bit 1 positive).
    On the GPU this is four HIP entry points (ops.max_iou_assign / random_sample_raw / bbox_targets); the torch
    forms above are the host restatement used on CPU."""
    num = s_cfg['num']
    if bboxes.is_cuda:
        assigned, _, lab = ops.max_iou_assign(bboxes, gt_bboxes, a_cfg['pos_iou_thr'], a_cfg['neg_iou_thr'], a_cfg['min_pos_iou'],
                                              a_cfg.get('match_low_quality', True), gt_labels, num_leading_gt, valid)
        inds, flags = ops.random_sample_raw(assigned, num, s_cfg['pos_fraction'])
        boxes, deltas, gt_inds, labels = ops.bbox_targets(bboxes, inds, flags, assigned, gt_bboxes, means, stds, lab, bg_label)
        return boxes, deltas, labels, gt_inds, flags >= 2, flags >= 1, inds, flags
    g = num_leading_gt
    assigned, _, lab = max_iou_assign(bboxes[g:], gt_bboxes, a_cfg['pos_iou_thr'], a_cfg['neg_iou_thr'], a_cfg['min_pos_iou'],
                                      a_cfg.get('match_low_quality', True), gt_labels, None if valid is None else valid[g:])
    if g > 0:
        assigned = torch.cat([torch.arange(1, g + 1, device=assigned.device), assigned], 0)
        if lab is not None:
            lab = torch.cat([gt_labels, lab], 0)
    if valid is not None:
        assigned = torch.where(valid, assigned, torch.full_like(assigned, -1))
    idx, is_pos, ok = sample_static(assigned, num, s_cfg['pos_fraction'])
    k = num - idx.numel()
    if k > 0:                                   # fewer boxes than samples: pad to the fixed size
        idx = torch.cat([idx, idx.new_zeros(k)]); is_pos = torch.cat([is_pos, is_pos.new_zeros(k)]); ok = torch.cat([ok, ok.new_zeros(k)])
    boxes = torch.where(ok[:, None], bboxes[idx], bboxes.new_tensor([0., 0., 1., 1.]).expand(num, 4))
    gt_inds = (assigned[idx] - 1).clamp(min=0)
    if gt_bboxes.size(0) > 0:
        deltas = bbox2delta(boxes, gt_bboxes[gt_inds], means, stds)
        deltas = torch.where(is_pos[:, None], deltas, torch.zeros_like(deltas))
    else:
        deltas = torch.zeros_like(boxes)
        is_pos = torch.zeros_like(is_pos)
    labels = None
    if lab is not None:
        labels = torch.where(is_pos, lab[idx], torch.full_like(idx, bg_label))
    return boxes, deltas, labels, gt_inds, is_pos, ok, idx, ok.to(torch.uint8) + 2 * is_pos.to(torch.uint8)


def _scaled(x, w):
    """x * loss_weight without the two launches (forward and backward) of a multiplication by one"""
    return x if w == 1.0 else x * w


def _cast(t, dtype):
    """compute-dtype view of a tensor; parameters resolve to their bf16 shadow (mixed.py)"""
    return mixed.weight(t, dtype)


def _conv(x, conv, dtype, padding=0, relu=False, x_is_relu=False):
    """conv on a channels-last activation.  1x1 convs are GEMMs over the token-major view (no conv
    library involved); 3x3 convs go to the conv library with PACKED channels-last operands (a strided
    view silently selects MIOpen's naive reference kernels)."""
    N, C, H, W = x.shape
    if conv.kernel_size == (1, 1):
        tok = x.permute(0, 2, 3, 1).reshape(N * H * W, C)
        y = ops.linear(tok, conv.weight, conv.bias, dtype)           # weight/bias gradients on the split-T kernel
        y = y.view(N, H, W, conv.out_channels).permute(0, 3, 1, 2)
    elif dtype == _H() and conv.kernel_size == (3, 3) and padding == 1 and C % 64 == 0 and conv.out_channels % 64 == 0:
        return ops.conv3x3(x, conv.weight, conv.bias, relu, x_is_relu=x_is_relu)  # HIP implicit-GEMM kernel (ReLU fused)
    else:
        x = x.contiguous(memory_format=torch.channels_last)
        w = _cast(conv.weight, dtype).contiguous(memory_format=torch.channels_last)
        y = F.conv2d(x, w, None if conv.bias is None else _cast(conv.bias, dtype), padding=padding)
    return F.relu(y, inplace=True) if relu else y


class _SyncBNHost(torch.autograd.Function):
    """torch restatement of ops.batch_norm's synchronised training pass for CPU tensors (gloo rehearsals and CPU tests):
    the same two collectives as torch.nn.SyncBatchNorm -- per-channel {sum, sum of squares, count} forward,
    {sum dy, sum dy*xhat} backward."""

    @staticmethod
    def forward(ctx, x, w, b, rm, rv, eps, momentum):
        import torch.distributed as dist
        C = x.shape[1]
        xf = x.float()
        s_ = torch.cat([xf.sum((0, 2, 3)), (xf * xf).sum((0, 2, 3)), xf.new_tensor([xf.numel() / C])])
        dist.all_reduce(s_)
        n = s_[-1]
        mean = s_[:C] / n
        var = (s_[C:2 * C] / n - mean * mean).clamp(min=0)
        invstd = torch.rsqrt(var + eps)
        if rm is not None:
            rm.mul_(1 - momentum).add_(mean * momentum)
            rv.mul_(1 - momentum).add_(var * (n / (n - 1).clamp(min=1)) * momentum)
        xhat = (xf - mean[None, :, None, None]) * invstd[None, :, None, None]
        ctx.save_for_backward(xhat, invstd, w.float(), n)
        return (xhat * w.float()[None, :, None, None] + b.float()[None, :, None, None]).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        xhat, invstd, w, n = ctx.saved_tensors
        dyf = dy.float()
        sdy, sdyx = dyf.sum((0, 2, 3)), (dyf * xhat).sum((0, 2, 3))
        tot = torch.cat([sdy, sdyx])
        dist.all_reduce(tot)
        C = sdy.numel()
        m1, m2 = tot[:C] / n, tot[C:] / n
        dx = (w * invstd)[None, :, None, None] * (dyf - m1[None, :, None, None] - xhat * m2[None, :, None, None])
        return dx.to(dy.dtype), sdyx, sdy, None, None, None, None


def _bn_act(x, cm, training, relu=True):
    """norm (+ ReLU) of a ConvModule built with norm_cfg: HIP kernels on the GPU (ops.batch_norm: statistics exchanged
    over the ranks for SyncBN), torch on the CPU."""
    import torch.distributed as dist
    bn = cm.bn
    if training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    multi = cm.sync and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if x.is_cuda:
        return ops.batch_norm(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.eps, bn.momentum, relu, sync=multi)
    if training and multi:
        y = _SyncBNHost.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum)
    else:
        y = F.batch_norm(x.float(), bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps).to(x.dtype)
    return F.relu(y) if relu else y


def giou_loss_elem(pred, target, eps=1e-6):
    """1 - GIoU of aligned (n,4) boxes: iou_loss.py:78-101 over bbox_overlaps(mode='giou', is_aligned=True)
    (iou2d_calculator.py:108-158).  Host restatement (differentiable) of csrc/det_losses.hip's giou_loss."""
    area1 = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
    area2 = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
    wh = (torch.min(pred[:, 2:], target[:, 2:]) - torch.max(pred[:, :2], target[:, :2])).clamp(min=0)
    overlap = wh[:, 0] * wh[:, 1]
    e = pred.new_tensor([eps])
    union = torch.max(area1 + area2 - overlap, e)
    ious = overlap / union
    ewh = (torch.max(pred[:, 2:], target[:, 2:]) - torch.min(pred[:, :2], target[:, :2])).clamp(min=0)
    earea = torch.max(ewh[:, 0] * ewh[:, 1], e)
    return 1 - (ious - (earea - union) / earea)


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1):
    """mmdet/core/post_processing/bbox_nms.py:7-93 (no score_factors): multi_bboxes (n, 4) or (n, 4*num_classes),
    multi_scores (n, num_classes + 1) with the background LAST.  -> (dets (k,5), labels (k,))."""
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 4)
    else:
        bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), num_classes, 4)
    scores = multi_scores[:, :-1]
    labels = torch.arange(num_classes, dtype=torch.long, device=scores.device).view(1, -1).expand_as(scores)
    bboxes, scores, labels = bboxes.reshape(-1, 4), scores.reshape(-1), labels.reshape(-1)
    inds = (scores > score_thr).nonzero(as_tuple=False).squeeze(1)            # test time: a host sync is acceptable
    bboxes, scores, labels = bboxes[inds], scores[inds], labels[inds]
    if bboxes.numel() == 0:
        return torch.cat([bboxes, scores[:, None]], -1), labels
    dets, keep = ops.batched_nms(bboxes.float().contiguous(), scores.float().contiguous(), labels, nms_cfg)   # HIP nms
    if max_num > 0:
        dets, keep = dets[:max_num], keep[:max_num]
    return dets, labels[keep]


def _sf4(scale_factor):
    """img_meta['scale_factor'] as (w, h, w, h) floats (a scalar or the 4-vector of the Resize transform)."""
    try:
        v = [float(t) for t in scale_factor]
    except TypeError:
        v = [float(scale_factor)] * 4
    return v if len(v) == 4 else [v[0]] * 4


def bbox2result(bboxes, labels, num_classes):
    """mmdet/core/bbox/transforms.py:99-117: per-class list of (k_c, 5) numpy arrays."""
    import numpy as np
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    b, l = bboxes.detach().float().cpu().numpy(), labels.detach().cpu().numpy()
    return [b[l == i, :] for i in range(num_classes)]


# ------------------------------------------------------------------------------------------
# RPN
# ------------------------------------------------------------------------------------------
class _RpnHeads(torch.autograd.Function):
    """rpn_cls and rpn_reg (rpn_head.py:41-47) as ONE GEMM per level on a concatenated bf16 weight, with the weight / bias
    gradients of ALL five levels accumulated in one fp32 step buffer by the split-T kernel and split into the two layers'
    gradients once, after the last level's backward.  (Through plain autograd every level cost two zero fills, two casts and the
    running sums of the concatenation's backward: ~30 small launches per step.)
    Inputs: tokens (T,C) bf16; w_cat (5A+pad, C) / b_cat bf16 constants; head; then the four real leaves (bf16 shadows or fp32
    masters) only so that autograd routes gradients to them."""

    @staticmethod
    def forward(ctx, tok, w_cat, b_cat, head, wc_leaf, wr_leaf, bc_leaf, br_leaf, relu_input=False):
        from .ops.functional import gemm_bf16
        ctx.save_for_backward(tok, w_cat)
        ctx.head = head
        ctx.relu_input = bool(relu_input)      # tok is a ReLU output (rpn_conv + ReLU, rpn_head.py:44): see backward
        ctx.dts = (wc_leaf.dtype, wr_leaf.dtype, bc_leaf.dtype, br_leaf.dtype)
        ctx.counted = any(ctx.needs_input_grad)
        if ctx.counted:
            mixed.use_begin(head.rpn_cls.weight)
        return gemm_bf16(tok.contiguous(), w_cat, b_cat)

    @staticmethod
    def backward(ctx, dy):
        from ._lib import call
        from .ops.functional import _p, _s, gemm_bf16
        tok, w_cat = ctx.saved_tensors
        head = ctx.head
        A = head.num_anchors
        dy = dy.contiguous()
        N1, C = w_cat.shape
        dx = None
        if ctx.needs_input_grad[0]:
            if (ctx.relu_input and tok.is_contiguous() and N1 % 8 == 0 and N1 <= 64 and C % 8 == 0 and N1 * C * 4 <= 65536):
                # a 16-deep "GEMM" is a bandwidth problem: one pass that also applies rpn_conv's ReLU backward (the 3x3 conv's
                # backward finds the gradient marked as gated and skips its threshold_backward launch)
                dx = torch.empty_like(tok)
                call("narrow_dgrad_gated_bf16", _p(dy), _p(w_cat), _p(tok), _p(dx), dy.shape[0], N1, C, _s())
                mixed.gated_mark(dx, tok)
            else:
                dx = gemm_bf16(dy, w_cat, None, b_is_kn=True)
        key = head.rpn_cls.weight
        dwb = mixed.step_buffer(key, 'rpn_dw', (N1, C), tok.device)
        dbb = mixed.step_buffer(key, 'rpn_db', (N1,), tok.device)
        parts = ((head.rpn_cls.weight, dwb[:A].view(A, C, 1, 1)), (head.rpn_reg.weight, dwb[A:5 * A].view(4 * A, C, 1, 1)),
                 (head.rpn_cls.bias, dbb[:A]), (head.rpn_reg.bias, dbb[A:5 * A]))
        sinks = [mixed.grad_sink(p) for p, _ in parts]
        all_sinks = all(sk is not None and sk[0].shape == g.shape for sk, (_, g) in zip(sinks, parts))
        tokc = tok.contiguous()
        last = mixed.use_end(key) == 0
        # with every parameter delivered through a reducer sink nothing on this stream reads the result: second stream
        sst = mixed.fork_to_side(tok.device, dy, tokc) if all_sinks else None
        call("wgrad_linear_bf16", _p(dy), _p(tokc), _p(dwb), _p(dbb), dy.shape[0], N1, C,
             ctypes.c_void_p(sst) if sst is not None else _s())
        if last and all_sinks:                          # last level: hand the accumulated gradients to the four parameters
            with mixed.on_side(tok.device):
                for sk, (_, g) in zip(sinks, parts):
                    sk[0].add_(g)
        if not last:
            return dx, None, None, None, None, None, None, None, None
        mixed.step_buffer_done(key, 'rpn_dw'); mixed.step_buffer_done(key, 'rpn_db')
        outs = []
        for sk, (p, g), dt_ in zip(sinks, parts, ctx.dts):
            if all_sinks:
                sk[1]()
                outs.append(None)
            elif sk is not None and sk[0].shape == g.shape:
                sk[0].add_(g)
                sk[1]()
                outs.append(None)
            else:
                outs.append(g.to(dt_).clone())
        return (dx, None, None, None) + tuple(outs) + (None,)


@HEADS.register_module()
class RPNHead(nn.Module):
    def __init__(self, in_channels, feat_channels=256, anchor_generator=None, bbox_coder=None, loss_cls=None,
                 loss_bbox=None, train_cfg=None, test_cfg=None, compute_dtype=torch.float32, **kwargs):
        super().__init__()
        ag = dict(anchor_generator)
        assert ag.pop('type', 'AnchorGenerator') == 'AnchorGenerator'
        self.anchor_generator = AnchorGenerator(ag['strides'], ag['ratios'], ag['scales'])
        self.num_anchors = self.anchor_generator.num_base_anchors
        bc = dict(bbox_coder or {})
        self.means = tuple(bc.get('target_means', (0., 0., 0., 0.)))
        self.stds = tuple(bc.get('target_stds', (1., 1., 1., 1.)))
        self.loss_cls_weight = _cfg_get(loss_cls, 'loss_weight', 1.0)
        self.loss_bbox_weight = _cfg_get(loss_bbox, 'loss_weight', 1.0)
        lb_type = _cfg_get(loss_bbox, 'type', 'L1Loss')
        if lb_type not in ('L1Loss', 'SmoothL1Loss') or not _cfg_get(loss_cls, 'use_sigmoid', True):
            raise NotImplementedError("RPN: sigmoid CE + L1 / SmoothL1 (the swin configs)")
        self.loss_bbox_beta = float(_cfg_get(loss_bbox, 'beta', 1.0)) if lb_type == 'SmoothL1Loss' else 0.0
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.compute_dtype = compute_dtype
        self.rpn_conv = nn.Conv2d(in_channels, feat_channels, 3, padding=1)
        self.rpn_cls = nn.Conv2d(feat_channels, self.num_anchors, 1)
        self.rpn_reg = nn.Conv2d(feat_channels, self.num_anchors * 4, 1)

    def init_weights(self):                                     # rpn_head.py:34-39
        for m in (self.rpn_conv, self.rpn_cls, self.rpn_reg):
            nn.init.normal_(m.weight, std=0.01)
            nn.init.constant_(m.bias, 0)

    def forward(self, feats):                                   # rpn_head.py:41-47
        dt = self.compute_dtype
        cls, reg = [], []
        A = self.num_anchors
        fused = dt == _H() and feats[0].is_cuda
        if dt == _H():
            # rpn_cls (A) and rpn_reg (4A) as ONE GEMM over the tokens, rows padded to a multiple of 8 for the kernels; the
            # concatenated bf16 weight / bias are built once per step (mixed.derived: constant until the next shadow refresh)
            C = self.rpn_cls.in_channels
            pad = (-5 * A) % 8

            def _wcat():
                return torch.cat([_cast(self.rpn_cls.weight, dt).detach().view(A, C), _cast(self.rpn_reg.weight, dt).detach().view(4 * A, C),
                                  torch.zeros(pad, C, device=self.rpn_cls.weight.device, dtype=dt)], 0)

            def _bcat():
                return torch.cat([self.rpn_cls.bias.detach(), self.rpn_reg.bias.detach(),
                                  torch.zeros(pad, device=self.rpn_cls.bias.device)], 0).to(dt)
            if fused:
                w, b = mixed.derived(self.rpn_cls.weight, 'rpn_wcat', _wcat), mixed.derived(self.rpn_cls.weight, 'rpn_bcat', _bcat)
            else:
                w = torch.cat([_cast(self.rpn_cls.weight, dt).view(A, C), _cast(self.rpn_reg.weight, dt).view(4 * A, C),
                               torch.zeros(pad, C, device=self.rpn_cls.weight.device, dtype=dt)], 0)
                b = torch.cat([self.rpn_cls.bias, self.rpn_reg.bias, torch.zeros(pad, device=w.device)], 0).to(dt)
        ys = []
        for x in feats:
            # a small level's conv + heads (P4-P6: 10-126 tiles, 40 us each) run on the second stream next to the big levels
            with (mixed.small_branch(x) if (fused and dt == _H()) else contextlib.nullcontext()) as sd:
                x = _conv(x, self.rpn_conv, dt, padding=1, relu=True)
                if dt == _H():
                    N, C, H, W = x.shape
                    tok = x.permute(0, 2, 3, 1).reshape(N * H * W, C)
                    if fused:
                        # the four leaves only route gradients (no cast: a parameter without a shadow is passed as it is)
                        y = _RpnHeads.apply(tok, w, b, self, mixed.leaf(self.rpn_cls.weight), mixed.leaf(self.rpn_reg.weight),
                                            self.rpn_cls.bias, self.rpn_reg.bias, True).view(N, H, W, -1)
                    else:
                        y = ops.linear(tok, w, b, dt).view(N, H, W, -1)
                    cls.append(y[..., :A].permute(0, 3, 1, 2))
                    reg.append(y[..., A:5 * A].permute(0, 3, 1, 2))
                    ys.append(y.view(N, H * W, -1))
                else:
                    cls.append(_conv(x, self.rpn_cls, dt))
                    reg.append(_conv(x, self.rpn_reg, dt))
            if sd is not None:
                mixed.side_outputs(x, y)
        mixed.side_join()            # the levels computed on the second stream (here and in the FPN) are read below / by the RoI heads
        self._flat = None
        if ys:
            # the anchor-major flattening that loss() and get_bboxes() need (anchor_head.py:474-486, rpn_head.py:119-125):
            # one strided concatenation per head straight from the token-major GEMM outputs
            N = ys[0].size(0)
            if ys[0].is_cuda:
                self._flat = (cls,) + tuple(ops.rpn_flatten(ys, A))             # one kernel each way (det_rpn_flatten_*)
            else:
                self._flat = (cls, torch.cat([y[:, :, :A] for y in ys], 1).reshape(N, -1),
                              torch.cat([y[:, :, A:5 * A] for y in ys], 1).reshape(N, -1, 4))
        return cls, reg

    def _flattened(self, cls_scores, bbox_preds):
        """(B, sum_l H_l W_l A) logits and (B, same, 4) deltas, levels concatenated in (h, w, a) order."""
        f = getattr(self, '_flat', None)
        if f is not None and f[0] is cls_scores:
            return f[1], f[2]
        B = cls_scores[0].size(0)
        return (torch.cat([c.permute(0, 2, 3, 1).reshape(B, -1) for c in cls_scores], 1),
                torch.cat([r.permute(0, 2, 3, 1).reshape(B, -1, 4) for r in bbox_preds], 1))

    def simple_test_rpn(self, x, img_metas):
        """dense_test_mixins.py:29-49: proposals of the test-time config, one (n, 5) tensor per image."""
        cls_scores, bbox_preds = self(x)
        return self.get_bboxes(cls_scores, bbox_preds, [m['img_shape'] for m in img_metas], self.test_cfg)

    # ---- training targets + loss (anchor_head.py:175-493) ----
    def _inside_flags(self, anchors, img_shapes, B):
        ab = self.train_cfg.get('allowed_border', -1)
        inside = [None] * B
        if ab >= 0:                                 # anchor_inside_flags (core/anchor/utils.py:28-50), anchor_head.py:200-207
            for i in range(B):
                h, w = img_shapes[i][:2]
                inside[i] = ((anchors[:, 0] >= -ab) & (anchors[:, 1] >= -ab) & (anchors[:, 2] < w + ab) & (anchors[:, 3] < h + ab))
        return inside

    def _sampled_targets(self, anchors, inside, gt_bboxes, device):
        """assign -> sample -> encode per image (anchor_head.py:175-262 for the sampled anchors), device path"""
        a_cfg, s_cfg = self.train_cfg['assigner'], self.train_cfg['sampler']
        B, num = len(gt_bboxes), s_cfg['num']
        inds = torch.empty(B, num, dtype=torch.long, device=device)
        flags = torch.empty(B, num, dtype=torch.uint8, device=device)
        tgts = torch.empty(B, num, 4, dtype=torch.float32, device=device)
        for i in range(B):
            assigned, _, _ = ops.max_iou_assign(anchors, gt_bboxes[i], a_cfg['pos_iou_thr'], a_cfg['neg_iou_thr'], a_cfg['min_pos_iou'],
                                                a_cfg.get('match_low_quality', True), None, 0, inside[i])
            ops.random_sample_raw(assigned, num, s_cfg['pos_fraction'], out=(inds[i], flags[i]))
            ops.bbox_targets(anchors, inds[i], flags[i], assigned, gt_bboxes[i], self.means, self.stds, out_deltas=tgts[i])
        return inds, flags, tgts

    @torch.no_grad()
    def early_targets(self, sizes, gt_bboxes, img_shapes, device):
        """The anchor targets depend on the anchors and the ground truth only, not on the predictions: computed (on the second stream)
        BEFORE the backbone runs, when the host is ahead of the GPU, instead of ~25 launches issued between proposal selection and the
        RoI heads, where the main stream sat idle for 0.25 ms waiting for the host.  ``sizes``: the pyramid's feature-map sizes the
        caller expects; ``loss`` recomputes the targets if the real ones differ.  Returns an opaque pair for ``loss(targets=...)``."""
        anchors = self.anchor_generator.grid_anchors_cat(list(sizes), device)
        inside = self._inside_flags(anchors, img_shapes, len(gt_bboxes))
        return tuple(sizes), self._sampled_targets(anchors, inside, gt_bboxes, device)

    def loss(self, cls_scores, bbox_preds, gt_bboxes, img_shapes, targets=None):
        """Sampled-anchor form of AnchorHead.loss: the reference builds (B, 255780) label / weight / target arrays and
        multiplies the per-anchor losses by 0/1 weights; only the <= 256 sampled anchors per image have non-zero
        weight, so the same sums are taken over those anchors directly.  No host synchronisation."""
        cfg = self.train_cfg
        a_cfg, s_cfg = cfg['assigner'], cfg['sampler']
        sizes = [tuple(c.shape[-2:]) for c in cls_scores]
        anchors = self.anchor_generator.grid_anchors_cat(sizes, cls_scores[0].device)
        B = cls_scores[0].size(0)
        cls, reg = self._flattened(cls_scores, bbox_preds)
        inside = self._inside_flags(anchors, img_shapes, B)
        beta = self.loss_bbox_beta
        if cls.is_cuda:
            # assign -> sample -> encode per image, each kernel writing its row of the batch-level tensors; then both losses, over
            # all images, in one forward and one backward launch (csrc/det_losses.hip)
            if targets is not None and targets[0] == tuple(sizes):
                inds, flags, tgts = targets[1]
            else:
                inds, flags, tgts = self._sampled_targets(anchors, inside, gt_bboxes, cls.device)
            lc, lb = ops.rpn_loss(cls, reg, inds, flags, tgts, beta)
            return dict(loss_rpn_cls=_scaled(lc, self.loss_cls_weight), loss_rpn_bbox=_scaled(lb, self.loss_bbox_weight))
        samples = [assign_and_sample(anchors, gt_bboxes[i], a_cfg, s_cfg, self.means, self.stds, valid=inside[i]) for i in range(B)]
        loss_cls = loss_bbox = total = 0.
        for i in range(B):
            _, tgt, _, _, is_pos, valid, idx, _ = samples[i]
            c_i, r_i = cls[i][idx].float(), reg[i][idx].float()
            lc = F.binary_cross_entropy_with_logits(c_i, is_pos.float(), reduction='none')     # fg -> 1, bg -> 0
            loss_cls = loss_cls + (lc * valid).sum()
            d = (r_i - tgt).abs()
            if beta > 0:
                d = torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)
            loss_bbox = loss_bbox + (d * is_pos[:, None]).sum()
            total = total + valid.sum()
        avg = total.clamp(min=1).float()                                                   # num_total_samples
        return dict(loss_rpn_cls=loss_cls / avg * self.loss_cls_weight, loss_rpn_bbox=loss_bbox / avg * self.loss_bbox_weight)

    # ---- proposals (rpn_head.py:82-236): detached, per level top-k by a full stable sort ----
    @torch.no_grad()
    def get_bboxes(self, cls_scores, bbox_preds, img_shapes, cfg, static=False):
        sizes = [tuple(c.shape[-2:]) for c in cls_scores]
        mlvl_anchors = self.anchor_generator.grid_anchors(sizes, cls_scores[0].device)
        B = cls_scores[0].size(0)
        nms_pre = cfg['nms_pre']
        sc_l, bp_l, an_l, id_l = [], [], [], []
        cls_all, reg_all = self._flattened(cls_scores, bbox_preds)
        if (static and cls_all.is_cuda and cfg.get('min_bbox_size', 0) <= 0
                and sum(min(a.size(0), nms_pre) for a in mlvl_anchors) < cfg['nms'].get('split_thr', 10000)
                and all(tuple(sh[:2]) == tuple(img_shapes[0][:2]) for sh in img_shapes)):
            # all images, all levels: ONE selection + decode launch (csrc/rpn_select.hip: radix select of the nms_pre-th
            # score per level, survivors compacted in anchor order and decoded), then one sort + one pair of NMS launches
            # (the per-image reductions run side by side).  Equal to the sort-based path below including ties.
            anchors_cat = self.anchor_generator.grid_anchors_cat(sizes, cls_all.device)
            scores, props, ids = ops.rpn_topk_decode(cls_all, reg_all, anchors_cat, [a.size(0) for a in mlvl_anchors], nms_pre,
                                                     self.means, self.stds, img_shapes[0])
            dets, valid = ops.batched_nms_static_multi(props, scores, ids, cfg['nms']['iou_threshold'], cfg['max_per_img'],
                                                       group_sizes=[min(a.size(0), nms_pre) for a in mlvl_anchors])
            return [(dets[i], valid[i]) for i in range(B)]
        s_all, d_all = cls_all.detach().float().sigmoid(), reg_all.detach().float()
        off = 0
        for lvl in range(len(cls_scores)):
            na = mlvl_anchors[lvl].size(0)
            s, d = s_all[:, off:off + na], d_all[:, off:off + na]
            off += na
            an = mlvl_anchors[lvl][None].expand(B, -1, -1)
            if s.shape[1] > nms_pre:
                ranked, rank_inds = s.sort(dim=1, descending=True, stable=True)
                topk = rank_inds[:, :nms_pre]
                s = ranked[:, :nms_pre]
                d = torch.gather(d, 1, topk[..., None].expand(-1, -1, 4))
                an = torch.gather(an, 1, topk[..., None].expand(-1, -1, 4))
            sc_l.append(s); bp_l.append(d); an_l.append(an)
            id_l.append(s.new_full((B, s.size(1)), lvl, dtype=torch.long))
        scores, deltas = torch.cat(sc_l, 1), torch.cat(bp_l, 1)
        anchors, ids = torch.cat(an_l, 1), torch.cat(id_l, 1)
        out = []
        if (static and anchors.is_cuda and cfg.get('min_bbox_size', 0) <= 0 and anchors.size(1) < cfg['nms'].get('split_thr', 10000)
                and all(tuple(sh[:2]) == tuple(img_shapes[0][:2]) for sh in img_shapes)):
            # all images at once: one decode, one sort, one pair of NMS launches (the per-image reductions run side by side)
            n = anchors.size(1)
            props = ops.delta2bbox(anchors.reshape(B * n, 4), deltas.reshape(B * n, 4), self.means, self.stds,
                                   max_shape=img_shapes[0]).view(B, n, 4)
            dets, valid = ops.batched_nms_static_multi(props, scores, ids, cfg['nms']['iou_threshold'], cfg['max_per_img'])
            return [(dets[i], valid[i]) for i in range(B)]
        for i in range(B):
            dec = ops.delta2bbox if anchors.is_cuda else delta2bbox
            props = dec(anchors[i], deltas[i], self.means, self.stds, max_shape=img_shapes[i])
            if cfg.get('min_bbox_size', 0) > 0:
                w, h = props[:, 2] - props[:, 0], props[:, 3] - props[:, 1]
                v = (w >= cfg['min_bbox_size']) & (h >= cfg['min_bbox_size'])
                props, sc, idl = props[v], scores[i][v], ids[i][v]
            else:
                sc, idl = scores[i], ids[i]
            if static and cfg.get('min_bbox_size', 0) <= 0 and props.size(0) < cfg['nms'].get('split_thr', 10000):
                # fixed-size (max_per_img, 5) result + validity mask: the kept count never leaves the device
                out.append(ops.batched_nms_static(props, sc, idl, cfg['nms']['iou_threshold'], cfg['max_per_img']))
                continue
            # max_num == the reference's `dets[:cfg.max_per_img]` (rpn_head.py:235); lets the device reduction stop early
            nms_cfg = dict(cfg['nms'], max_num=cfg['max_per_img'])
            dets, _ = ops.batched_nms(props, sc, idl, nms_cfg)                  # HIP nms
            out.append(dets[:cfg['max_per_img']])
        return out


# ------------------------------------------------------------------------------------------
# RoI extractor / heads
# ------------------------------------------------------------------------------------------
@ROI_EXTRACTORS.register_module()
class SingleRoIExtractor(nn.Module):
    """single_level_roi_extractor.py:32-108 over ops.RoIAlign (looked up by name as the reference does)."""

    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56):
        super().__init__()
        cfg = dict(roi_layer)
        layer_type = cfg.pop('type')
        assert hasattr(ops, layer_type)                          # base_roi_extractor.py:51
        layer_cls = getattr(ops, layer_type)
        self.roi_layers = nn.ModuleList([layer_cls(spatial_scale=1 / s, **cfg) for s in featmap_strides])
        self.out_channels, self.featmap_strides, self.finest_scale = out_channels, featmap_strides, finest_scale
        self.fp16_enabled = False

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def map_roi_levels(self, rois, num_levels):                  # :47-51
        scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
        lv = torch.floor(torch.log2(scale / self.finest_scale + 1e-6))
        return lv.clamp(min=0, max=num_levels - 1).long()

    def _fast_ok(self, feats, rois):
        f0 = feats[0]
        return (1 < len(feats) <= 4 and f0.is_cuda and f0.shape[1] % 4 == 0 and rois.size(0) > 0
                and f0.dtype in (torch.float32, _H()))

    def forward_with(self, other, feats, rois, valid, other_rois, other_valid):
        """This extractor's and ``other``'s RoI features from the same pyramid (the bbox and mask extractors of one
        R-CNN stage): values equal the two separate calls; on the one-launch path their backward shares one fp32
        gradient accumulator (ops.roi_align_multilevel_group)."""
        a, b = self.roi_layers[0], other.roi_layers[0]
        same = (len(feats) == other.num_inputs and tuple(self.featmap_strides) == tuple(other.featmap_strides)
                and self.finest_scale == other.finest_scale and a.sampling_ratio == b.sampling_ratio and a.aligned == b.aligned)
        if not (same and self._fast_ok(feats, rois) and other_rois.size(0) > 0):
            return self(feats, rois, valid=valid), other(feats[:other.num_inputs], other_rois, valid=other_valid)
        n = len(feats)
        groups = []
        for ext, r, v in ((self, rois, valid), (other, other_rois, other_valid)):
            lv = ops.map_roi_levels(r, n, ext.finest_scale, v)           # one launch: levels, -1 for unused slots
            groups.append((r, lv, ext.roi_layers[0].output_size))
        return tuple(ops.roi_align_multilevel_group(list(feats), groups, self.featmap_strides[:n], a.sampling_ratio, a.aligned,
                                                    out_dtype=feats[0].dtype))

    def forward(self, feats, rois, roi_scale_factor=None, valid=None):
        out_size = self.roi_layers[0].output_size
        num_levels = len(feats)
        f0 = feats[0]
        if (num_levels > 1 and num_levels <= 4 and f0.is_cuda and f0.shape[1] % 4 == 0 and rois.size(0) > 0
                and f0.dtype in (torch.float32, _H())):
            # one launch over the pyramid, level chosen per RoI on the device: no per-level nonzero / gather /
            # scatter and no host sync; every level receives a gradient tensor, so the reference's dummy-gradient
            # trick (:98-107) is not needed.  `valid` marks the used slots of a fixed-size sample.
            lvls = ops.map_roi_levels(rois, num_levels, self.finest_scale, valid)
            rl = self.roi_layers[0]
            return ops.roi_align_multilevel(list(feats), rois, lvls, out_size, self.featmap_strides[:num_levels],
                                            rl.sampling_ratio, rl.aligned, out_dtype=f0.dtype)   # bf16 in -> bf16 out: no cast pass
        cl = feats[0].is_contiguous(memory_format=torch.channels_last) and not feats[0].is_contiguous()
        roi_feats = torch.empty((rois.size(0), self.out_channels, *out_size), device=rois.device, dtype=torch.float32,
                                memory_format=torch.channels_last if cl else torch.contiguous_format).zero_()
        if rois.size(0) == 0:
            return roi_feats
        if num_levels == 1:
            return self.roi_layers[0](feats[0], rois)
        lvls = self.map_roi_levels(rois, num_levels)
        # one stable sort groups the rois by level (no per-level nonzero sync); scatter back by index_copy
        order = torch.sort(lvls, stable=True)[1]
        counts = torch.bincount(lvls, minlength=num_levels).tolist()
        start = 0
        pieces = []
        dummy = 0.
        for i in range(num_levels):
            n = counts[i]
            if n > 0:
                inds = order[start:start + n]
                pieces.append((inds, self.roi_layers[i](feats[i], rois[inds])))
            else:
                # keep every level in the graph (single_level_roi_extractor.py:98-107: avoids a DDP hang)
                dummy = dummy + feats[i].sum() * 0.
            start += n
        roi_feats = roi_feats.index_copy(0, torch.cat([p[0] for p in pieces]), torch.cat([p[1] for p in pieces]))
        if isinstance(dummy, torch.Tensor):
            roi_feats = roi_feats + dummy.float()
        return roi_feats


@HEADS.register_module()
class ConvFCBBoxHead(nn.Module):
    """convfc_bbox_head.py:9-178 (shared convs -> shared fcs -> fc_cls / fc_reg) + bbox_head.py loss / targets / decoding.
    The swin configs use two instances: Shared2FCBBoxHead (Mask R-CNN) and 4 shared convs with SyncBN + 1 shared fc,
    ``reg_decoded_bbox=True`` with GIoULoss (Cascade Mask R-CNN).  Separate cls / reg branches are not on this path."""

    def __init__(self, num_shared_convs=0, num_shared_fcs=0, num_cls_convs=0, num_cls_fcs=0, num_reg_convs=0, num_reg_fcs=0,
                 conv_out_channels=256, fc_out_channels=1024, conv_cfg=None, norm_cfg=None, with_avg_pool=False,
                 in_channels=256, roi_feat_size=7, num_classes=80, bbox_coder=None, reg_class_agnostic=False,
                 reg_decoded_bbox=False, loss_cls=None, loss_bbox=None, compute_dtype=torch.float32, **kwargs):
        super().__init__()
        if num_cls_convs or num_cls_fcs or num_reg_convs or num_reg_fcs or with_avg_pool or conv_cfg is not None:
            raise NotImplementedError("ConvFCBBoxHead: shared convs + shared fcs only (the swin configs)")
        if num_shared_fcs < 1:
            raise NotImplementedError("ConvFCBBoxHead: at least one shared fc (the swin configs use 1 or 2)")
        bc = dict(bbox_coder or {})
        self.means = tuple(bc.get('target_means', (0., 0., 0., 0.)))
        self.stds = tuple(bc.get('target_stds', (0.1, 0.1, 0.2, 0.2)))
        lb_type = _cfg_get(loss_bbox, 'type', 'L1Loss')
        if lb_type not in ('L1Loss', 'SmoothL1Loss', 'GIoULoss') or (lb_type == 'GIoULoss') != bool(reg_decoded_bbox):
            raise NotImplementedError("bbox head: L1 / SmoothL1 on deltas, or GIoU with reg_decoded_bbox=True")
        if _cfg_get(loss_cls, 'use_sigmoid', False):
            raise NotImplementedError("bbox head: softmax cross entropy")
        self.reg_beta = float(_cfg_get(loss_bbox, 'beta', 1.0)) if lb_type == 'SmoothL1Loss' else 0.0
        self.giou_eps = float(_cfg_get(loss_bbox, 'eps', 1e-6))
        self.reg_class_agnostic, self.reg_decoded_bbox = bool(reg_class_agnostic), bool(reg_decoded_bbox)
        self.num_classes = num_classes
        self.loss_cls_weight = _cfg_get(loss_cls, 'loss_weight', 1.0)
        self.loss_bbox_weight = _cfg_get(loss_bbox, 'loss_weight', 1.0)
        self.compute_dtype = compute_dtype
        from .fpn import ConvModule
        self.shared_convs = nn.ModuleList([ConvModule(in_channels if i == 0 else conv_out_channels, conv_out_channels, 3, padding=1,
                                                      norm_cfg=norm_cfg) for i in range(num_shared_convs)])
        last = (conv_out_channels if num_shared_convs else in_channels) * roi_feat_size * roi_feat_size
        self.shared_fcs = nn.ModuleList([nn.Linear(last if i == 0 else fc_out_channels, fc_out_channels) for i in range(num_shared_fcs)])
        self.fc_cls = nn.Linear(fc_out_channels, num_classes + 1)
        self.fc_reg = nn.Linear(fc_out_channels, 4 if reg_class_agnostic else 4 * num_classes)

    def init_weights(self):
        for cm in self.shared_convs:                        # mmcv ConvModule.init_weights: kaiming (fan_out, relu), norm 1/0
            nn.init.kaiming_normal_(cm.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
            if cm.conv.bias is not None:
                nn.init.constant_(cm.conv.bias, 0)
            if cm.with_norm:
                nn.init.constant_(cm.bn.weight, 1); nn.init.constant_(cm.bn.bias, 0)
        for m in self.shared_fcs:                           # convfc_bbox_head.py:116-122
            nn.init.xavier_uniform_(m.weight); nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.fc_cls.weight, 0, 0.01); nn.init.constant_(self.fc_cls.bias, 0)       # bbox_head.py:61-68
        nn.init.normal_(self.fc_reg.weight, 0, 0.001); nn.init.constant_(self.fc_reg.bias, 0)

    def forward(self, x):
        dt = self.compute_dtype
        if len(self.shared_convs):
            x = _cast(x, dt).contiguous(memory_format=torch.channels_last)
            for cm in self.shared_convs:                    # ConvModule: conv -> norm -> ReLU
                x = _conv(x, cm.conv, dt, padding=1, relu=not cm.with_norm)
                if cm.with_norm:
                    x = _bn_act(x, cm, self.training, relu=True)
        x = _cast(x.flatten(1), dt)                 # (K, C*7*7) in the reference's (C,7,7) order
        fast = dt == _H() and x.is_cuda
        for fc in self.shared_fcs:
            y = ops.linear(x, fc.weight, fc.bias, dt) if fast else F.linear(x, _cast(fc.weight, dt), _cast(fc.bias, dt))
            x = F.relu(y, inplace=True)
        if fast:
            return ops.linear(x, self.fc_cls.weight, self.fc_cls.bias, dt), ops.linear(x, self.fc_reg.weight, self.fc_reg.bias, dt)
        cls = F.linear(x, _cast(self.fc_cls.weight, dt), _cast(self.fc_cls.bias, dt))
        reg = F.linear(x, _cast(self.fc_reg.weight, dt), _cast(self.fc_reg.bias, dt))
        return cls, reg

    def loss(self, cls_score, bbox_pred, labels, bbox_targets, pos_mask, valid=None, rois=None, flags=None):
        """bbox_head.py:188-238: CE over the sampled RoIs (avg_factor = their count); on the positives -- divided by the
        number of samples -- L1 / SmoothL1 between the labelled (or class-agnostic) deltas and the encoded targets, or
        (reg_decoded_bbox) GIoU between the boxes decoded against ``rois`` (n,4) and the gt boxes in ``bbox_targets``.
        ``valid`` masks the unused slots of a fixed-size sample; ``flags`` (n,) uint8 = valid + 2 * (pos_mask & valid) when the
        caller already has it."""
        n = cls_score.size(0)
        if valid is None:
            valid = torch.ones(n, dtype=torch.bool, device=cls_score.device)
        if self.reg_decoded_bbox and rois is None:
            raise ValueError("reg_decoded_bbox=True: loss() needs the sampled rois")
        if cls_score.is_cuda and n > 0:
            if flags is None:                       # (the packed training path hands over the sampler's own flag bytes)
                flags = valid.to(torch.uint8) + 2 * (pos_mask & valid).to(torch.uint8)
            giou = (rois[:, -4:], self.means, self.stds, self.giou_eps) if self.reg_decoded_bbox else None
            lc, acc, lb = ops.bbox_loss(cls_score, bbox_pred, labels, bbox_targets, flags, self.num_classes,
                                        self.reg_class_agnostic, self.reg_beta, giou)
            return dict(loss_cls=_scaled(lc, self.loss_cls_weight), acc=acc, loss_bbox=_scaled(lb, self.loss_bbox_weight))
        cls_score, bbox_pred = cls_score.float(), bbox_pred.float()
        nv = valid.sum().clamp(min=1).float()
        ce = F.cross_entropy(cls_score, labels, reduction='none')
        loss_cls = (ce * valid).sum() / nv
        acc = (((cls_score.argmax(1) == labels) & valid).sum() / nv) * 100
        if self.reg_class_agnostic:
            pred = bbox_pred.view(n, 4)
        else:
            pred = bbox_pred.view(n, -1, 4)[torch.arange(n, device=labels.device), labels.clamp(max=self.num_classes - 1)]
        pos = (pos_mask & valid).float()
        if self.reg_decoded_bbox:
            dec = delta2bbox(rois[:, -4:].float(), pred, self.means, self.stds)                  # bbox_head.py:215-216
            per = giou_loss_elem(dec, bbox_targets.float(), self.giou_eps)
            loss_bbox = (per * pos).sum() / nv
        else:
            d = (pred - bbox_targets).abs()
            if self.reg_beta > 0:
                d = torch.where(d < self.reg_beta, 0.5 * d * d / self.reg_beta, d - 0.5 * self.reg_beta)
            loss_bbox = (d * pos[:, None]).sum() / nv
        return dict(loss_cls=loss_cls * self.loss_cls_weight, acc=acc, loss_bbox=loss_bbox * self.loss_bbox_weight)

    @torch.no_grad()
    def regress_by_class(self, rois, labels, cls_score, bbox_pred, img_shape):
        """bbox_head.py:409-436 with CascadeRoIHead's label choice: ``labels`` None (testing) or background entries take
        argmax(cls_score[:, :-1]) (cascade_roi_head.py:274-281, :316-317).  rois (n,4) -> refined (n,4)."""
        nc = self.num_classes
        if rois.is_cuda:
            return ops.regress_by_class(rois, labels, cls_score, bbox_pred, nc, self.reg_class_agnostic, self.means, self.stds,
                                        img_shape)
        am = cls_score[:, :nc].float().argmax(1)
        lab = am if labels is None else torch.where((labels >= nc) | (labels < 0), am, labels)
        d = bbox_pred.float()
        if not self.reg_class_agnostic:
            d = d.view(d.size(0), nc, 4)[torch.arange(d.size(0), device=d.device), lab]
        return delta2bbox(rois.float(), d, self.means, self.stds, max_shape=img_shape)

    @torch.no_grad()
    def get_bboxes(self, rois, cls_score, bbox_pred, img_shape, scale_factor, rescale=False, cfg=None):
        """bbox_head.py:270-373 for one image: rois (n,5), deltas (n, 4*num_classes) or class-agnostic (n,4);
        ``cls_score`` may be a list of per-stage scores, which are averaged (:300-301)."""
        if isinstance(cls_score, (list, tuple)):
            cls_score = sum(c.float() for c in cls_score) / float(len(cls_score))
        scores = F.softmax(cls_score.float(), dim=-1)
        n = scores.size(0)
        nc = 1 if self.reg_class_agnostic else self.num_classes
        deltas = bbox_pred.float().reshape(n * nc, 4)
        boxes = rois[:, 1:5].float()[:, None, :].expand(n, nc, 4).reshape(n * nc, 4)
        dec = ops.delta2bbox if boxes.is_cuda else delta2bbox
        bboxes = dec(boxes.contiguous(), deltas.contiguous(), self.means, self.stds, max_shape=img_shape).view(n, nc * 4)
        if rescale and n > 0:
            bboxes = (bboxes.view(n, nc, 4) / bboxes.new_tensor(_sf4(scale_factor))).view(n, nc * 4)
        if cfg is None:
            return bboxes, scores
        return multiclass_nms(bboxes, scores, cfg['score_thr'], cfg['nms'], cfg['max_per_img'])


@HEADS.register_module()
class Shared2FCBBoxHead(ConvFCBBoxHead):
    """convfc_bbox_head.py:181-192."""

    def __init__(self, fc_out_channels=1024, *args, **kwargs):
        super().__init__(*args, num_shared_convs=0, num_shared_fcs=2, fc_out_channels=fc_out_channels, **kwargs)


@HEADS.register_module()
class Shared4Conv1FCBBoxHead(ConvFCBBoxHead):
    """convfc_bbox_head.py:195-205."""

    def __init__(self, fc_out_channels=1024, *args, **kwargs):
        super().__init__(*args, num_shared_convs=4, num_shared_fcs=1, fc_out_channels=fc_out_channels, **kwargs)


@HEADS.register_module()
class FCNMaskHead(nn.Module):
    """fcn_mask_head.py:20-126: 4x (conv3x3+ReLU), deconv 2x2 s2 + ReLU, conv1x1 -> num_classes."""

    def __init__(self, num_convs=4, roi_feat_size=14, in_channels=256, conv_kernel_size=3, conv_out_channels=256,
                 num_classes=80, class_agnostic=False, upsample_cfg=dict(type='deconv', scale_factor=2), loss_mask=None,
                 compute_dtype=torch.float32, **kwargs):
        super().__init__()
        if upsample_cfg.get('type', 'deconv') != 'deconv' or class_agnostic or conv_kernel_size != 3:
            raise NotImplementedError("FCNMaskHead: deconv upsampling, class-specific masks (swin configs)")
        self.num_classes, self.compute_dtype = num_classes, compute_dtype
        self.loss_mask_weight = _cfg_get(loss_mask, 'loss_weight', 1.0)
        from .fpn import ConvModule
        self.convs = nn.ModuleList([ConvModule(in_channels if i == 0 else conv_out_channels, conv_out_channels, 3, padding=1)
                                    for i in range(num_convs)])
        self.upsample = nn.ConvTranspose2d(conv_out_channels, conv_out_channels, 2, stride=2)
        self.conv_logits = nn.Conv2d(conv_out_channels, num_classes, 1)

    def init_weights(self):
        for m in [self.upsample, self.conv_logits]:
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            nn.init.constant_(m.bias, 0)

    def forward(self, x):
        dt = self.compute_dtype
        x = _cast(x, dt).contiguous(memory_format=torch.channels_last)
        for i, c in enumerate(self.convs):
            x = _conv(x, c.conv, dt, padding=1, relu=True, x_is_relu=i > 0)
        x = self._deconv2x2_relu(x, dt)
        return _conv(x, self.conv_logits, dt)

    def _deconv_rows(self, x, dt):
        """ConvTranspose2d(k=2, s=2) + ReLU (fcn_mask_head.py:122-125) as one GEMM over tokens:
        rows[(n,y,x), (ky,kx,co)] = relu(sum_ci x[n,ci,y,x] W[ci,co,ky,kx] + b[co]) = out[n, co, 2y+ky, 2x+kx]."""
        P, C, H, W = x.shape
        Co = self.upsample.out_channels
        tok = x.permute(0, 2, 3, 1).reshape(P * H * W, C)
        w = _cast(self.upsample.weight, dt).permute(2, 3, 1, 0).reshape(4 * Co, C)     # rows (ky,kx,co)
        b = _cast(self.upsample.bias, dt).repeat(4)
        return ops.linear(tok, w, b, dt, relu=True)                                    # (P*H*W, 4*Co); ReLU in the GEMM's epilogue

    def _deconv2x2_relu(self, x, dt):
        """_deconv_rows followed by the 2x2 pixel shuffle -> (P, Co, 2H, 2W) channels-last view."""
        P, C, H, W = x.shape
        Co = self.upsample.out_channels
        y = self._deconv_rows(x, dt)
        y = y.view(P, H, W, 2, 2, Co).permute(0, 1, 3, 2, 4, 5).reshape(P, 2 * H, 2 * W, Co)
        return y.permute(0, 3, 1, 2)                                                   # channels-last view

    def forward_rows(self, x):
        """Training form of forward(): the same logits as rows in deconvolution order, (roi, y, x, ky, kx) x class, i.e.
        WITHOUT the pixel shuffle of the 256-channel activation and without the NHWC->NCHW copy of the logits -- the 1x1
        conv_logits is per pixel, so it commutes with the shuffle, and ops.mask_loss(deconv_order=True) indexes the rows
        directly.  forward(x)[n, c, 2y+ky, 2x+kx] == forward_rows(x)[((n*H + y)*W + x)*4 + ky*2 + kx, c]."""
        dt = self.compute_dtype
        x = _cast(x, dt).contiguous(memory_format=torch.channels_last)
        for i, c in enumerate(self.convs):             # conv -> ReLU chain: from the second on, the input is a ReLU output
            x = _conv(x, c.conv, dt, padding=1, relu=True, x_is_relu=i > 0)
        Co = self.upsample.out_channels
        rows = self._deconv_rows(x, dt).view(-1, Co)                                   # (P*H*W*4, Co)
        return ops.linear(rows, self.conv_logits.weight, self.conv_logits.bias, dt)    # (P*H*W*4, num_classes)

    def loss_rows(self, rows, mask_targets, labels, valid):
        """loss() on forward_rows() output (GPU)."""
        return dict(loss_mask=_scaled(ops.mask_loss(rows, mask_targets, labels, valid, deconv_order=True), self.loss_mask_weight))

    def loss(self, mask_pred, mask_targets, labels, valid=None):
        """mask_cross_entropy (cross_entropy_loss.py): mean BCE over (positives x 28 x 28) of the class channel."""
        n = mask_pred.size(0)
        if n == 0:
            return dict(loss_mask=mask_pred.sum() * 0)
        if mask_pred.is_cuda:
            v = valid if valid is not None else torch.ones(n, dtype=torch.bool, device=mask_pred.device)
            return dict(loss_mask=_scaled(ops.mask_loss(mask_pred, mask_targets, labels, v), self.loss_mask_weight))
        pred = mask_pred.float()[torch.arange(n, device=mask_pred.device), labels]
        per_roi = F.binary_cross_entropy_with_logits(pred, mask_targets, reduction='none').mean(dim=(1, 2))
        if valid is None:
            return dict(loss_mask=per_roi.mean() * self.loss_mask_weight)
        loss = (per_roi * valid).sum() / valid.sum().clamp(min=1)
        return dict(loss_mask=loss * self.loss_mask_weight)


    @torch.no_grad()
    def get_seg_masks(self, mask_pred, det_bboxes, det_labels, rcnn_test_cfg, ori_shape, scale_factor, rescale, is_prob=False):
        """fcn_mask_head.py:169-300: per-class lists of (img_h, img_w) bool numpy masks.  Sigmoid, class select,
        bilinear paste into the box and the `mask_thr_binary` test run as one HIP kernel (ops.paste_masks).
        ``is_prob``: mask_pred already holds probabilities (the reference's ndarray input, :209-213)."""
        import numpy as np
        cls_segms = [[] for _ in range(self.num_classes)]
        bboxes = det_bboxes[:, :4].float()
        if rescale:
            img_h, img_w = int(ori_shape[0]), int(ori_shape[1])
            bboxes = bboxes / bboxes.new_tensor(_sf4(scale_factor))
        else:
            img_h = int(np.round(ori_shape[0] * _sf4(scale_factor)[1]))
            img_w = int(np.round(ori_shape[1] * _sf4(scale_factor)[0]))
        thr = rcnn_test_cfg['mask_thr_binary']
        if thr < 0:
            raise NotImplementedError("get_seg_masks: soft (uint8-scaled) masks")
        n = mask_pred.size(0)
        if n == 0:
            return cls_segms
        im_mask = ops.paste_masks(mask_pred, det_labels, bboxes, img_h, img_w, thr, is_prob).cpu().numpy()
        labels = det_labels.cpu().numpy()
        for i in range(n):
            cls_segms[labels[i]].append(im_mask[i])
        return cls_segms


def mask_target(pos_proposals_list, pos_assigned_gt_inds_list, gt_masks_list, mask_size):
    """mmdet.core.mask_target (mask_target.py:6-122) over BitmapMasks.crop_and_resize (structures.py:328-358) on the
    device: proposals clipped to the mask's extent (:104-107), rois = [index of the assigned gt mask, box],
    roi_align(masks[:, None], rois, mask_size, 1.0, 0, 'avg', True) >= 0.5 -> float 0/1 targets (sum_i K_i, h, w).
    gt_masks_list[i]: (G_i, H, W) uint8 / bool tensor.  When every image has masks of one size, all images go through ONE
    RoIAlign launch (the masks stacked along the batch axis, a RoI's batch index offset by its image's first mask) --
    the per-image launches are latency-bound and would run back to back; the reference selects masks[inds] first
    (a K-times larger copy) and loops over images on the host.  Slot arithmetic stays on the device: no sync."""
    size = (mask_size, mask_size) if isinstance(mask_size, int) else tuple(mask_size)
    nimg = len(pos_proposals_list)
    m_roi, m_gt, gt_masks = pos_proposals_list, pos_assigned_gt_inds_list, gt_masks_list

    def rois_of(i, off):
        r = torch.cat([(m_gt[i] + off).to(m_roi[i].dtype)[:, None], m_roi[i][:, :4]], 1)
        maxh, maxw = gt_masks[i].shape[1:]
        r[:, 1::2].clamp_(0, maxw)
        r[:, 2::2].clamp_(0, maxh)
        return r
    same = (nimg > 0 and all(g_.size(0) > 0 for g_ in gt_masks) and all(r_.size(0) > 0 for r_ in m_roi)
            and all(g_.shape[1:] == gt_masks[0].shape[1:] for g_ in gt_masks) and m_roi[0].is_cuda)
    if same:
        m = torch.cat(list(gt_masks), 0).to(_H())[:, None]      # 0/1 exact in bf16
        offs, o_ = [], 0
        for g_ in gt_masks:
            offs.append(o_); o_ += g_.size(0)
        r = torch.cat([rois_of(i, offs[i]) for i in range(nimg)], 0)
        t = ops.roi_align(m, r, size, 1.0, 0, 'avg', True)                # structures.py:353-354
        return (t[:, 0] >= 0.5).float()
    tg = []
    for i in range(nimg):
        if m_roi[i].size(0) == 0 or gt_masks[i].size(0) == 0:
            tg.append(m_roi[i].new_zeros((m_roi[i].size(0),) + size))
            continue
        m = gt_masks[i].to(_H())[:, None].contiguous()
        t = ops.roi_align(m, rois_of(i, 0), size, 1.0, 0, 'avg', True)
        tg.append((t[:, 0] >= 0.5).float())
    return torch.cat(tg) if tg else tg


def _roi_stage_train(x, proposal_list, gt_bboxes, gt_labels, gt_masks, cfg, bbox_roi_extractor, bbox_head, mask_roi_extractor,
                     mask_head, bbox_branch=False):
    """One R-CNN stage of training (standard_roi_head.py:70-131; cascade_roi_head.py:228-270 runs it per stage) with
    fixed-size samples.  ``proposal_list[i]`` is ``dets (n,4|5)`` (the reference's form) or ``(dets (max,4|5), valid (max,))``
    from the static RPN / refinement path.  Every tensor below has a shape known on the host (512 RoIs and 128 mask slots per
    image); unused slots are masked out of the losses and skipped by RoIAlign, so the step needs no device->host
    synchronisation.  gt_masks: list of (G_i, H, W) uint8/bool tensors on the device.
    Returns (losses, state) -- state carries what CascadeRoIHead's refinement needs."""
    a, s = cfg['assigner'], cfg['sampler']
    nimg = len(proposal_list)
    nc = bbox_head.num_classes
    num, npos_max = s['num'], int(s['num'] * s['pos_fraction'])
    if _PACKED_STAGE and nimg > 0 and x[0].is_cuda and all(gb.size(0) > 0 for gb in gt_bboxes):
        return _roi_stage_train_packed(x, proposal_list, gt_bboxes, gt_labels, gt_masks, cfg, bbox_roi_extractor, bbox_head,
                                       mask_roi_extractor, mask_head, bbox_branch)
    roi_l, lab_l, tgt_l, pos_l, val_l, isgt_l = [], [], [], [], [], []
    m_roi, m_gt, m_lab, m_val = [], [], [], []
    for i in range(nimg):
        p = proposal_list[i]
        dets, pvalid = (p if isinstance(p, tuple) else (p, torch.ones(p.size(0), dtype=torch.bool, device=p.device)))
        props = dets[:, :4]
        g = gt_bboxes[i].size(0)
        if s.get('add_gt_as_proposals', True):
            props = torch.cat([gt_bboxes[i], props], 0)
            pvalid = torch.cat([torch.ones(g, dtype=torch.bool, device=props.device), pvalid], 0)
        lead = g if s.get('add_gt_as_proposals', True) else 0
        boxes, t_i, l_i, gt_ind, is_pos, valid, inds, _ = assign_and_sample(
            props, gt_bboxes[i], a, s, bbox_head.means, bbox_head.stds, gt_labels[i], lead, pvalid, bg_label=nc)
        if bbox_head.reg_decoded_bbox:                 # bbox_head.py:174-175: the targets are the matched gt boxes themselves
            t_i = gt_bboxes[i][gt_ind] if g > 0 else torch.zeros_like(boxes)
        roi_l.append(boxes); lab_l.append(l_i); tgt_l.append(t_i); pos_l.append(is_pos); val_l.append(valid)
        isgt_l.append(is_pos & (inds < lead))         # SamplingResult.pos_is_gt
        k = min(npos_max, num)                                    # positives come first in the sample
        m_roi.append(boxes[:k]); m_gt.append(gt_ind[:k]); m_lab.append(l_i[:k].clamp(max=nc - 1)); m_val.append(is_pos[:k])
    losses = {}
    rois = bbox2roi(roi_l)
    valid = torch.cat(val_l)
    labels = torch.cat(lab_l)
    feats = x[:bbox_roi_extractor.num_inputs]
    mask_feats = None
    if mask_head is not None and isinstance(bbox_roi_extractor, SingleRoIExtractor) and isinstance(mask_roi_extractor, SingleRoIExtractor):
        # both RoI sets of the stage are known here: pool them together (shared backward accumulator)
        bbox_feats, mask_feats = bbox_roi_extractor.forward_with(mask_roi_extractor, feats, rois, valid, bbox2roi(m_roi), torch.cat(m_val))
    else:
        bbox_feats = bbox_roi_extractor(feats, rois, valid=valid)                  # HIP RoIAlign, all levels at once
    cls_score, bbox_pred = bbox_head(bbox_feats)
    losses.update(bbox_head.loss(cls_score, bbox_pred, labels, torch.cat(tgt_l), torch.cat(pos_l), valid, rois=rois))
    state = dict(rois=roi_l, labels=labels, cls_score=cls_score, bbox_pred=bbox_pred, valid=val_l, pos_is_gt=isgt_l)
    if mask_head is not None:
        mvalid = torch.cat(m_val)
        if mask_feats is None:
            mask_feats = mask_roi_extractor(x[:mask_roi_extractor.num_inputs], bbox2roi(m_roi), valid=mvalid)
        rows_path = mask_feats.is_cuda and hasattr(mask_head, 'forward_rows') and mask_feats.size(0) > 0
        mask_pred = mask_head.forward_rows(mask_feats) if rows_path else mask_head(mask_feats)
        tg = [mask_target(m_roi, m_gt, gt_masks, cfg.get('mask_size', 28))]
        if rows_path:
            losses.update(mask_head.loss_rows(mask_pred, torch.cat(tg), torch.cat(m_lab), mvalid))
        else:
            losses.update(mask_head.loss(mask_pred, torch.cat(tg), torch.cat(m_lab), mvalid))
    return losses, state


def _roi_stage_train_packed(x, proposal_list, gt_bboxes, gt_labels, gt_masks, cfg, bbox_roi_extractor, bbox_head,
                            mask_roi_extractor, mask_head, bbox_branch=False):
    """_roi_stage_train on the GPU: per image assign -> sample -> ONE pack launch (ops.roi_targets_pack) that writes the
    image's rows of every batch-level tensor of the stage -- RoIs with their image column, targets, labels, flags, the mask
    slots' feature RoIs and crop_and_resize rows -- instead of ~20 gathers / comparisons / clamps / concatenations."""
    a, s = cfg['assigner'], cfg['sampler']
    nimg = len(proposal_list)
    nc = bbox_head.num_classes
    num = s['num']
    dev = x[0].device
    with_mask = mask_head is not None
    km = min(int(s['num'] * s['pos_fraction']), num) if with_mask else 0          # positives come first in the sample
    buf = ops.RoiStageBuffers(nimg, num, km, dev)
    add_gt = s.get('add_gt_as_proposals', True)
    # every image's masks of one size: ONE crop_and_resize launch over the stacked masks (mask_target)
    stacked = with_mask and all(m_.size(0) > 0 and m_.shape[1:] == gt_masks[0].shape[1:] for m_ in gt_masks)
    off = 0
    for i in range(nimg):
        p = proposal_list[i]
        dets, pvalid = p if isinstance(p, tuple) else (p, None)
        props = dets[:, :4]
        g = gt_bboxes[i].size(0)
        if add_gt:
            props = torch.cat([gt_bboxes[i], props], 0)
            if pvalid is not None:
                pvalid = torch.cat([_ones_bool(g, dev), pvalid], 0)
        lead = g if add_gt else 0
        assigned, _, lab = ops.max_iou_assign(props, gt_bboxes[i], a['pos_iou_thr'], a['neg_iou_thr'], a['min_pos_iou'],
                                              a.get('match_low_quality', True), gt_labels[i], lead, pvalid)
        inds, flags = ops.random_sample_raw(assigned, num, s['pos_fraction'], out=(buf.inds[i], buf.flags[i]))
        mh, mw = (gt_masks[i].shape[1:] if with_mask else (0, 0))
        ops.roi_targets_pack(buf, i, props, inds, flags, assigned, gt_bboxes[i], bbox_head.means, bbox_head.stds, lab, nc, lead,
                             bbox_head.reg_decoded_bbox, off if stacked else 0, (mh, mw))
        if with_mask:
            off += gt_masks[i].size(0)
    losses = {}
    rois, valid, labels = buf.rois, buf.valid, buf.labels
    feats = x[:bbox_roi_extractor.num_inputs]
    mask_feats = None
    if with_mask and isinstance(bbox_roi_extractor, SingleRoIExtractor) and isinstance(mask_roi_extractor, SingleRoIExtractor):
        bbox_feats, mask_feats = bbox_roi_extractor.forward_with(mask_roi_extractor, feats, rois, valid, buf.feat_rois, buf.mvalid)
    else:
        bbox_feats = bbox_roi_extractor(feats, rois, valid=valid)
    # bbox_branch (the standard RoI head, whose caller does not read the stage's box predictions): the box head and its loss -- two
    # fully connected layers of 64 output tiles and a handful of small launches -- run on the sub-graph stream next to the mask
    # head's convolutions (196 tiles of 256 CUs each): neither fills the chip.  Autograd runs their backward on that stream too.
    # (not inside a stream capture: the branch's outputs cross streams through Tensor.record_stream, which a captured graph's private
    # pool does not honour -- test_graph_replay_equals_eager_steps, which captures WITH the auxiliary streams, caught it)
    # (and one process only: with an overlapped gradient all-reduce a bucket is launched behind the stream its LAST gradient arrived on,
    # and nothing marks the sub-graph stream as busy during backward -- not rehearsed, so not enabled)
    branch = bool(bbox_branch and with_mask and _BBOX_BRANCH and mixed.side_enabled() and not torch.cuda.is_current_stream_capturing()
                  and not (torch.distributed.is_available() and torch.distributed.is_initialized()
                           and torch.distributed.get_world_size() > 1))
    state = {}

    def run_bbox():
        with (mixed.on_side(dev, bbox_feats, labels, buf.targets, buf.pos, valid, rois, buf.flags, kind='branch')
              if branch else contextlib.nullcontext()) as bsd:
            cls_score, bbox_pred = bbox_head(bbox_feats)
            bl = bbox_head.loss(cls_score, bbox_pred, labels, buf.targets, buf.pos, valid, rois=rois, flags=buf.flags.view(-1))
        if bsd is not None:
            mixed.side_outputs(cls_score, bbox_pred, *bl.values())
        losses.update(bl)
        rv = rois.view(nimg, num, 5)
        state.update(rois=[rv[i, :, 1:] for i in range(nimg)], labels=labels, cls_score=cls_score, bbox_pred=bbox_pred,
                     valid=list(valid.view(nimg, num).unbind(0)), pos_is_gt=list(buf.is_gt.view(nimg, num).unbind(0)))

    if not branch:
        run_bbox()
    if with_mask:
        if mask_feats is None:
            mask_feats = mask_roi_extractor(x[:mask_roi_extractor.num_inputs], buf.feat_rois, valid=buf.mvalid)
        size = cfg.get('mask_size', 28)
        size = (size, size) if isinstance(size, int) else tuple(size)
        # The mask targets (crop_and_resize of the gt masks: a 16 MB concatenation, a cast and a latency-bound RoIAlign, 0.25 ms)
        # feed only the loss: they are computed on the second stream while the mask head runs on this one.
        # (the mask head is enqueued first: issuing the branch takes the host ~100 us, which the main stream spends computing)
        rows_path = hasattr(mask_head, 'forward_rows') and mask_feats.size(0) > 0
        mask_pred = mask_head.forward_rows(mask_feats) if rows_path else mask_head(mask_feats)
        if branch:
            run_bbox()                        # issued behind the mask head: the host's ~0.25 ms for it no longer idle the main stream
        with mixed.on_side(dev, buf.mask_rois, *gt_masks) as sd:
            if stacked:
                m = _take_mask_stack(gt_masks) if sd is not None else None                              # stacked before the backbone ran
                if m is None:
                    m = (gt_masks[0] if nimg == 1 else torch.cat(list(gt_masks), 0)).to(_H())[:, None]  # 0/1 exact in bf16
                tg = (ops.roi_align(m, buf.mask_rois, size, 1.0, 0, 'avg', True)[:, 0] >= 0.5).float()
            else:
                mr = buf.mask_rois.view(nimg, km, 5)
                tg = torch.cat([(ops.roi_align(gt_masks[i].to(_H())[:, None].contiguous(), mr[i], size, 1.0, 0, 'avg',
                                               True)[:, 0] >= 0.5).float() for i in range(nimg)])
        if sd is not None:
            mixed.side_outputs(tg)            # produced on the second stream, read (and kept for backward) on this one
            mixed.side_join()
        loss_fn = mask_head.loss_rows if rows_path else mask_head.loss
        losses.update(loss_fn(mask_pred, tg, buf.mlabels, buf.mvalid))
    return losses, state


# The stacked 16-bit copy of the ground-truth masks (a 16 MB concatenation and a cast) depends on the batch only: forward_train makes it
# on the second stream before the backbone runs, the mask-target branch picks it up here instead of issuing it in front of its RoIAlign.
_MASK_STACKS = {}


def _mask_key(gt_masks):
    return tuple((t.data_ptr(), tuple(t.shape)) for t in gt_masks)


def _early_mask_stack(gt_masks):
    if not gt_masks or not all(m_.size(0) > 0 and m_.shape[1:] == gt_masks[0].shape[1:] for m_ in gt_masks):
        return
    _MASK_STACKS.clear()
    _MASK_STACKS[_mask_key(gt_masks)] = (gt_masks[0] if len(gt_masks) == 1 else torch.cat(list(gt_masks), 0)).to(_H())[:, None]


def _take_mask_stack(gt_masks):
    return _MASK_STACKS.pop(_mask_key(gt_masks), None)


_BBOX_BRANCH = os.environ.get("SWIN_BBOX_BRANCH", "1") != "0"       # 0: the box head on the main stream (A/B)
_ONES_BOOL = {}
_PACKED_STAGE = True          # False: _roi_stage_train's per-op body also on the GPU (A/B, tests)


def _ones_bool(n, device):
    """a cached all-True (n,) mask (constant: read only)"""
    k = (str(device), )
    t = _ONES_BOOL.get(k)
    if t is None or t.numel() < n:
        t = _ONES_BOOL[k] = torch.ones(max(n, 256), dtype=torch.bool, device=device)
    return t[:n]


@HEADS.register_module()
class StandardRoIHead(nn.Module):
    def __init__(self, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None, mask_head=None, shared_head=None,
                 train_cfg=None, test_cfg=None, compute_dtype=torch.float32):
        super().__init__()
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.bbox_roi_extractor = build_roi_extractor(bbox_roi_extractor)
        self.bbox_head = build_from_cfg(bbox_head, HEADS, dict(compute_dtype=compute_dtype))
        self.mask_roi_extractor = build_roi_extractor(mask_roi_extractor) if mask_roi_extractor else None
        self.mask_head = build_from_cfg(mask_head, HEADS, dict(compute_dtype=compute_dtype)) if mask_head else None

    def init_weights(self, pretrained=None):
        self.bbox_head.init_weights()
        if self.mask_head is not None:
            self.mask_head.init_weights()

    def forward_train(self, x, proposal_list, gt_bboxes, gt_labels, gt_masks):
        """standard_roi_head.py:70-131 with fixed-size samples (see _roi_stage_train)."""
        losses, _ = _roi_stage_train(x, proposal_list, gt_bboxes, gt_labels, gt_masks, self.train_cfg, self.bbox_roi_extractor,
                                     self.bbox_head, self.mask_roi_extractor, self.mask_head, bbox_branch=True)
        return losses


    # ---- test time (standard_roi_head.py:221-247, test_mixins.py:52-157, :246-317) ----
    @torch.no_grad()
    def simple_test(self, x, proposal_list, img_metas, rescale=False):
        cfg = self.test_cfg
        bbox_results, segm_results = [], []
        nc = self.bbox_head.num_classes
        feats = x[:self.bbox_roi_extractor.num_inputs]
        for i, (props, meta) in enumerate(zip(proposal_list, img_metas)):
            rois = bbox2roi([props[:, :4]])
            rois[:, 0] = i
            if rois.size(0) == 0:
                det_bboxes, det_labels = rois.new_zeros((0, 5)), rois.new_zeros((0,), dtype=torch.long)
            else:
                cls_score, bbox_pred = self.bbox_head(self.bbox_roi_extractor(feats, rois))
                det_bboxes, det_labels = self.bbox_head.get_bboxes(rois, cls_score, bbox_pred, meta['img_shape'],
                                                                   meta['scale_factor'], rescale, cfg)
            bbox_results.append(bbox2result(det_bboxes, det_labels, nc))
            if self.mask_head is None:
                continue
            if det_bboxes.size(0) == 0:
                segm_results.append([[] for _ in range(self.mask_head.num_classes)])
                continue
            # boxes back at the test scale for the RoI features (test_mixins.py:277-281)
            mb = det_bboxes[:, :4]
            if rescale:
                mb = mb * mb.new_tensor(_sf4(meta['scale_factor']))
            mask_rois = torch.cat([mb.new_full((mb.size(0), 1), i), mb], 1)
            mask_pred = self.mask_head(self.mask_roi_extractor(x[:self.mask_roi_extractor.num_inputs], mask_rois))
            # get_seg_masks receives the TEST-scale boxes and undoes the scale itself (test_mixins.py:299, fcn_mask_head.py:245)
            segm_results.append(self.mask_head.get_seg_masks(mask_pred, mb, det_labels, cfg, meta['ori_shape'],
                                                             meta['scale_factor'], rescale))
        if self.mask_head is None:
            return bbox_results
        return list(zip(bbox_results, segm_results))


@HEADS.register_module()
class CascadeRoIHead(nn.Module):
    """cascade_roi_head.py:13-420: ``num_stages`` bbox heads (and mask heads) applied in sequence, each trained on the
    boxes refined by the previous one, with per-stage assigner thresholds (train_cfg is a list) and loss weights."""

    def __init__(self, num_stages, stage_loss_weights, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None,
                 mask_head=None, shared_head=None, train_cfg=None, test_cfg=None, compute_dtype=torch.float32):
        super().__init__()
        assert bbox_roi_extractor is not None and bbox_head is not None
        assert shared_head is None, 'Shared head is not supported in Cascade RCNN anymore'
        self.num_stages, self.stage_loss_weights = num_stages, list(stage_loss_weights)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

        def per_stage(c):
            c = c if isinstance(c, (list, tuple)) else [c for _ in range(num_stages)]
            assert len(c) == num_stages
            return c
        self.bbox_roi_extractor = nn.ModuleList([build_roi_extractor(c) for c in per_stage(bbox_roi_extractor)])
        self.bbox_head = nn.ModuleList([build_from_cfg(c, HEADS, dict(compute_dtype=compute_dtype)) for c in per_stage(bbox_head)])
        self.mask_head = self.mask_roi_extractor = None
        self.share_roi_extractor = False
        if mask_head is not None:
            self.mask_head = nn.ModuleList([build_from_cfg(c, HEADS, dict(compute_dtype=compute_dtype)) for c in per_stage(mask_head)])
            if mask_roi_extractor is not None:
                self.mask_roi_extractor = nn.ModuleList([build_roi_extractor(c) for c in per_stage(mask_roi_extractor)])
            else:                                                    # cascade_roi_head.py:87-89
                self.share_roi_extractor = True
                self.mask_roi_extractor = self.bbox_roi_extractor
        if train_cfg is not None:
            assert len(train_cfg) == num_stages

    @property
    def with_mask(self):
        return self.mask_head is not None

    def init_weights(self, pretrained=None):
        for i in range(self.num_stages):
            self.bbox_head[i].init_weights()
            if self.with_mask:
                self.mask_head[i].init_weights()

    def forward_train(self, x, proposal_list, gt_bboxes, gt_labels, gt_masks, img_shapes=None):
        """cascade_roi_head.py:200-284.  Each stage is _roi_stage_train (fixed-size samples, no host sync); between
        stages the sampled RoIs are regressed by their (gt or predicted) class on the device and the ones that were gt
        boxes are dropped through the validity mask (bbox_head.py:376-407)."""
        losses = {}
        for i in range(self.num_stages):
            lw = self.stage_loss_weights[i]
            head = self.bbox_head[i]
            st_losses, st = _roi_stage_train(x, proposal_list, gt_bboxes, gt_labels, gt_masks, self.train_cfg[i],
                                             self.bbox_roi_extractor[i], head,
                                             self.mask_roi_extractor[i] if self.with_mask else None,
                                             self.mask_head[i] if self.with_mask else None)
            for name, value in st_losses.items():
                losses[f's{i}.{name}'] = value * lw if 'loss' in name else value
            if i < self.num_stages - 1:
                with torch.no_grad():
                    nimg = len(st['rois'])
                    per = st['rois'][0].size(0)
                    shapes = img_shapes if img_shapes is not None else [None] * nimg
                    if all(tuple(sh[:2]) == tuple(shapes[0][:2]) for sh in shapes) if shapes[0] is not None else True:
                        new = head.regress_by_class(torch.cat(st['rois']), st['labels'], st['cls_score'], st['bbox_pred'], shapes[0])
                        new = list(new.split(per))
                    else:
                        new = [head.regress_by_class(st['rois'][j], st['labels'][j * per:(j + 1) * per],
                                                     st['cls_score'][j * per:(j + 1) * per], st['bbox_pred'][j * per:(j + 1) * per],
                                                     shapes[j]) for j in range(nimg)]
                    proposal_list = [(new[j], st['valid'][j] & ~st['pos_is_gt'][j]) for j in range(nimg)]
        return losses

    # ---- test time (cascade_roi_head.py:286-411) ----
    @torch.no_grad()
    def simple_test(self, x, proposal_list, img_metas, rescale=False):
        cfg = self.test_cfg
        nc = self.bbox_head[-1].num_classes
        bbox_results, segm_results = [], []
        for i, (props, meta) in enumerate(zip(proposal_list, img_metas)):
            rois = bbox2roi([props[:, :4]])
            rois[:, 0] = i
            if rois.size(0) == 0:
                det_bboxes, det_labels = rois.new_zeros((0, 5)), rois.new_zeros((0,), dtype=torch.long)
            else:
                ms_scores = []
                for st in range(self.num_stages):
                    ext = self.bbox_roi_extractor[st]
                    cls_score, bbox_pred = self.bbox_head[st](ext(x[:ext.num_inputs], rois))
                    ms_scores.append(cls_score)
                    if st < self.num_stages - 1:                          # :315-323
                        ref = self.bbox_head[st].regress_by_class(rois[:, 1:], None, cls_score, bbox_pred, meta['img_shape'])
                        rois = torch.cat([rois[:, :1], ref], 1)
                det_bboxes, det_labels = self.bbox_head[-1].get_bboxes(rois, ms_scores, bbox_pred, meta['img_shape'],
                                                                       meta['scale_factor'], rescale, cfg)
            bbox_results.append(bbox2result(det_bboxes, det_labels, nc))
            if not self.with_mask:
                continue
            if det_bboxes.size(0) == 0:
                segm_results.append([[] for _ in range(self.mask_head[-1].num_classes)])
                continue
            mb = det_bboxes[:, :4]
            if rescale:
                mb = mb * mb.new_tensor(_sf4(meta['scale_factor']))
            mask_rois = torch.cat([mb.new_full((mb.size(0), 1), i), mb], 1)
            prob = 0.
            for st in range(self.num_stages):                               # merge_aug_masks without flips: the mean
                ext = self.mask_roi_extractor[st]
                prob = prob + self.mask_head[st](ext(x[:ext.num_inputs], mask_rois)).float().sigmoid()
            prob = prob / float(self.num_stages)
            segm_results.append(self.mask_head[-1].get_seg_masks(prob, mb, det_labels, cfg, meta['ori_shape'],
                                                                 meta['scale_factor'], rescale, is_prob=True))
        if not self.with_mask:
            return bbox_results
        return list(zip(bbox_results, segm_results))


@DETECTORS.register_module()
class MaskRCNN(nn.Module):
    """TwoStageDetector wiring (two_stage.py:17-167) for the Mask R-CNN swin configs."""

    def __init__(self, backbone, neck=None, rpn_head=None, roi_head=None, train_cfg=None, test_cfg=None, pretrained=None,
                 compute_dtype=torch.float32):
        super().__init__()
        self.compute_dtype = compute_dtype
        self.backbone = build_from_cfg(backbone, BACKBONES, dict(compute_dtype=compute_dtype))
        self.neck = build_from_cfg(neck, NECKS, dict(compute_dtype=compute_dtype)) if neck is not None else None
        rpn_train = _cfg_get(train_cfg, 'rpn')
        self.rpn_head = build_from_cfg(rpn_head, HEADS, dict(train_cfg=rpn_train, test_cfg=_cfg_get(test_cfg, 'rpn'),
                                                             compute_dtype=compute_dtype))
        self.roi_head = build_from_cfg(roi_head, HEADS, dict(train_cfg=_cfg_get(train_cfg, 'rcnn'),
                                                             test_cfg=_cfg_get(test_cfg, 'rcnn'),
                                                             compute_dtype=compute_dtype))
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.init_weights(pretrained)

    def init_weights(self, pretrained=None):                    # two_stage.py:50-68
        self.backbone.init_weights(pretrained=pretrained)
        if self.neck is not None:
            self.neck.init_weights()
        self.rpn_head.init_weights()
        self.roi_head.init_weights(pretrained)

    def parameters_in_forward_order(self):
        """Parameters in first-use order (for gradient buckets: reverse of it ~ the order gradients arrive in backward)."""
        out, seen = [], set()
        for m in (self.backbone, self.neck, self.rpn_head, self.roi_head):
            if m is None:
                continue
            ps = m.parameters_in_forward_order() if hasattr(m, 'parameters_in_forward_order') else m.parameters()
            for p in ps:
                if id(p) not in seen:
                    seen.add(id(p)); out.append(p)
        for p in self.parameters():
            if id(p) not in seen:
                seen.add(id(p)); out.append(p)
        return out

    def extract_feat(self, img):                                # two_stage.py:80-85
        x = self.backbone(img)
        return self.neck(x) if self.neck is not None else x

    def _pyramid_sizes(self, img):
        """feature-map sizes of the neck's outputs for this image size (patch embedding / PatchMerging pad to even sizes: ceil
        halving from stride 4; the extra FPN levels are stride-2 subsamplings), or None when the configuration is not the plain one"""
        try:
            n_out = int(self.neck.num_outs)
            n_in = len(self.neck.in_channels)
            ps = self.backbone.patch_embed.patch_size
            ps = int(ps[0] if isinstance(ps, (tuple, list)) else ps)
            if getattr(self.neck, 'start_level', 0) != 0:
                return None
        except (AttributeError, TypeError):
            return None
        h, w = -(-img.shape[-2] // ps), -(-img.shape[-1] // ps)
        sizes = []
        oi = tuple(getattr(self.backbone, 'out_indices', (0, 1, 2, 3)))
        if len(oi) != n_in or oi != tuple(range(oi[0], oi[0] + n_in)):
            return None
        for _ in range(oi[0]):
            h, w = -(-h // 2), -(-w // 2)
        for k in range(n_out):
            sizes.append((h, w))
            h, w = -(-h // 2), -(-w // 2)
        return sizes

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_masks=None, proposals=None):
        # the RPN's anchor targets first, on the second stream (RPNHead.early_targets)
        early = None
        if img.is_cuda and mixed.side_enabled() and hasattr(self.rpn_head, 'early_targets') and self.neck is not None:
            sizes = self._pyramid_sizes(img)
            if sizes is not None:
                with mixed.on_side(img.device, *gt_bboxes) as side0:
                    if side0 is not None:
                        early = self.rpn_head.early_targets(sizes, gt_bboxes, [m['img_shape'] for m in img_metas], img.device)
                        if gt_masks is not None and getattr(self.roi_head, 'with_mask', True):
                            _early_mask_stack(list(gt_masks))
        if self.neck is not None and hasattr(self.neck, 'defer_join'):
            # the small pyramid levels' convs run on the second stream (mixed.small_branch); here the RPN head, which continues
            # on that stream, joins -- a stand-alone extract_feat() joins at the end of the neck
            self.neck.defer_join = True
            try:
                x = self.extract_feat(img)
            finally:
                self.neck.defer_join = False
        else:
            x = self.extract_feat(img)
        losses = {}
        img_shapes = [m['img_shape'] for m in img_metas]
        cls_scores, bbox_preds = self.rpn_head(x)
        # The RPN's training branch (anchor assignment, sampling, loss: ~30 small latency-bound launches) does not feed the RoI
        # stage: it runs on the second stream next to proposal selection / NMS and the RoI heads; the main stream joins before the
        # losses are summed (parse_losses) -- autograd runs the branch's backward on that stream too.
        # proposals first: the RoI heads wait for them, nothing waits for the branch -- and issuing the branch takes the host ~150 us,
        # which the main stream now spends on proposal selection instead of idling in front of it
        proposal_cfg = _cfg_get(self.train_cfg, 'rpn_proposal', _cfg_get(self.test_cfg, 'rpn'))
        proposal_list = self.rpn_head.get_bboxes(cls_scores, bbox_preds, img_shapes, proposal_cfg, static=True)
        with mixed.on_side(cls_scores[0].device, *cls_scores, *bbox_preds) as side:
            rpn_losses = self.rpn_head.loss(cls_scores, bbox_preds, gt_bboxes, img_shapes, targets=early if side is not None else None)
        if side is not None:
            for v in rpn_losses.values():
                v.record_stream(torch.cuda.current_stream(v.device))
        losses.update(rpn_losses)
        if isinstance(self.roi_head, CascadeRoIHead):
            losses.update(self.roi_head.forward_train(x, proposal_list, gt_bboxes, gt_labels, gt_masks, img_shapes))
        else:
            losses.update(self.roi_head.forward_train(x, proposal_list, gt_bboxes, gt_labels, gt_masks))
        mixed.side_join()
        return losses

    @torch.no_grad()
    def simple_test(self, img, img_metas, proposals=None, rescale=False):
        """two_stage.py:187-204: per image (bbox_results, segm_results) in the reference's result format."""
        x = self.extract_feat(img)
        proposal_list = self.rpn_head.simple_test_rpn(x, img_metas) if proposals is None else proposals
        return self.roi_head.simple_test(x, proposal_list, img_metas, rescale=rescale)

    @staticmethod
    def parse_losses(losses):
        """base.py:185-218 without the per-scalar all-reduce/.item(): returns (loss tensor, dict of tensors)."""
        terms = [v.float().reshape(()) for k, v in losses.items() if 'loss' in k]
        total = torch.stack(terms).sum() if len(terms) > 1 else terms[0]      # one stack + one sum, not a chain of adds
        return total, losses


@DETECTORS.register_module()
class CascadeRCNN(MaskRCNN):
    """cascade_rcnn.py: TwoStageDetector with a CascadeRoIHead (configs/_base_/models/cascade_mask_rcnn_swin_fpn.py)."""


def build_detector(cfg, train_cfg=None, test_cfg=None, compute_dtype=torch.float32):
    """mmdet/models/builder.py:67-77."""
    cfg = dict(cfg)
    if compute_dtype in (torch.bfloat16, torch.float16):
        from . import _lib
        _lib.set_half_dtype(compute_dtype)        # the process's 16-bit type (and library build): before anything uses the library
    if train_cfg is not None:
        cfg['train_cfg'] = train_cfg
    if test_cfg is not None:
        cfg['test_cfg'] = test_cfg
    model = build_from_cfg(cfg, DETECTORS, dict(compute_dtype=compute_dtype))
    if compute_dtype == _H():
        mixed.khwc_resident_(model)          # 3x3 conv weights live in the HIP conv kernels' (Cout,ky,kx,Cin) memory layout
    return model
