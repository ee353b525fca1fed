"""Loss functions of the detector heads on the HIP kernels (csrc/det_losses.hip): one forward and one backward
launch each, over the fixed-size samples produced by ops.random_sample_raw / ops.bbox_targets.

Reference arithmetic: AnchorHead.loss_single (anchor_head.py:375-434), BBoxHead.loss (bbox_head.py:188-238),
FCNMaskHead.loss -> mask_cross_entropy (losses/cross_entropy_loss.py)."""
import ctypes

import torch

from .._lib import half_dtype as _H

from .._lib import SWIN_BF16, SWIN_F32, SwinHipError, call
from .functional import _p, _s


def _dt(t):
    if t.dtype == torch.float32:
        return SWIN_F32
    if t.dtype == _H():
        return SWIN_BF16
    raise SwinHipError(f"loss kernels: float32 / bfloat16 logits only, got {t.dtype}")


class _RPNLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls, reg, inds, flags, targets, beta):
        cls, reg = cls.contiguous(), reg.contiguous()
        B, A = cls.shape
        S = inds.numel() // B
        out = torch.empty(3, device=cls.device, dtype=torch.float32)
        call("det_rpn_loss_fwd", _p(cls), _p(reg), B, A, S, _p(inds), _p(flags), _p(targets), beta, _p(out), _dt(cls), _s())
        ctx.save_for_backward(cls, reg, inds, flags, targets, out)
        ctx.beta = beta
        ctx.mark_non_differentiable(inds, flags)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_cls, g_bbox):
        cls, reg, inds, flags, targets, out = ctx.saved_tensors
        B, A = cls.shape
        S = inds.numel() // B
        g = torch.stack([g_cls if g_cls is not None else out.new_zeros(()), g_bbox if g_bbox is not None else out.new_zeros(())]).float()
        dcls, dreg = torch.zeros_like(cls), torch.zeros_like(reg)
        call("det_rpn_loss_bwd", _p(cls), _p(reg), B, A, S, _p(inds), _p(flags), _p(targets), ctx.beta, _p(out), _p(g), _p(dcls),
             _p(dreg), _dt(cls), _s())
        return dcls, dreg, None, None, None, None


def rpn_loss(cls, reg, inds, flags, targets, beta=0.0):
    """cls (B,A) logits, reg (B,A,4) deltas; inds / flags (B,S) and targets (B,S,4) of the sampled anchors ->
    (loss_cls, loss_bbox): sums over the batch's samples divided by their count (anchor_head.py:375-434, 485-493).
    ``beta`` 0: L1Loss; > 0: SmoothL1Loss(beta)."""
    return _RPNLoss.apply(cls, reg, inds.contiguous(), flags.contiguous(), targets.contiguous().float(), float(beta))


def _f4(v):
    return (ctypes.c_float * 4)(*[float(x) for x in v])


class _BBoxLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls, bbox, labels, targets, flags, num_classes, reg):
        cls, bbox = cls.contiguous(), bbox.contiguous()
        n = cls.size(0)
        mode, agnostic, beta, eps, rois, means, stds = reg
        if bbox.numel() != n * (4 if agnostic else 4 * num_classes):
            raise SwinHipError(f"bbox_loss: bbox_pred has {bbox.numel()} elements for {n} RoIs (class_agnostic={agnostic})")
        out = torch.empty(4, device=cls.device, dtype=torch.float32)
        lse = torch.empty(n + 4 * ((n + 15) // 16), device=cls.device, dtype=torch.float32)      # + the kernel's partial-sum rows
        ctx.reg_args = (mode, 1 if agnostic else 0, beta, eps, _p(rois), _f4(means) if means is not None else None,
                        _f4(stds) if stds is not None else None)
        call("det_bbox_loss_fwd", _p(cls), _p(bbox), n, num_classes, _p(labels), _p(targets), _p(flags), *ctx.reg_args, _p(out),
             _p(lse), _dt(cls), _s())
        ctx.save_for_backward(cls, bbox, labels, targets, flags, out, lse, rois)
        ctx.nc = num_classes
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_cls, g_acc, g_bbox):
        cls, bbox, labels, targets, flags, out, lse, rois = ctx.saved_tensors
        n = cls.size(0)
        z = out.new_zeros(())
        g = torch.stack([g_cls if g_cls is not None else z, z, g_bbox if g_bbox is not None else z, z]).float()
        dcls, dbbox = torch.empty_like(cls), torch.empty_like(bbox)
        call("det_bbox_loss_bwd", _p(cls), _p(bbox), n, ctx.nc, _p(labels), _p(targets), _p(flags), *ctx.reg_args, _p(out), _p(lse),
             _p(g), _p(dcls), _p(dbbox), _dt(cls), _s())
        return dcls, dbbox, None, None, None, None, None


def bbox_loss(cls_score, bbox_pred, labels, targets, flags, num_classes, class_agnostic=False, beta=0.0, giou=None):
    """(loss_cls, acc %, loss_bbox) of BBoxHead.loss for a fixed-size sample (flags: bit 0 used, bit 1 positive).
    Regression term: L1 (beta 0) / SmoothL1(beta) between the labelled (or class-agnostic) deltas and ``targets``; or,
    with ``giou=(rois (n,4), means, stds, eps)``, GIoU between the DECODED boxes and the gt boxes in ``targets``
    (reg_decoded_bbox=True, bbox_head.py:215-216)."""
    if giou is None:
        reg = (0, bool(class_agnostic), float(beta), 1e-6, None, None, None)
    else:
        rois, means, stds, eps = giou
        reg = (2, bool(class_agnostic), 0.0, float(eps), rois.contiguous().float(), tuple(means), tuple(stds))
    return _BBoxLoss.apply(cls_score, bbox_pred, labels.contiguous(), targets.contiguous().float(), flags.contiguous(),
                           int(num_classes), reg)


class _MaskLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, labels, valid, deconv_w):
        pred = pred.contiguous()
        n = target.shape[0]
        P = target.shape[1] * target.shape[2]
        if deconv_w:
            nc = pred.shape[-1]
            if pred.numel() != n * P * nc or deconv_w * deconv_w != P:
                raise SwinHipError(f"mask_loss: deconv-order logits {tuple(pred.shape)} do not match {n} RoIs of {P} pixels")
        else:
            nc = pred.shape[1]
            if pred.shape[0] != n or pred.shape[2] * pred.shape[3] != P:
                raise SwinHipError(f"mask_loss: logits {tuple(pred.shape)} do not match targets {tuple(target.shape)}")
        out = torch.empty(2, device=pred.device, dtype=torch.float32)
        per_roi = torch.empty(n, device=pred.device, dtype=torch.float32)
        call("det_mask_loss_fwd", _p(pred), n, nc, P, int(deconv_w), _p(target), _p(labels), _p(valid), _p(out), _p(per_roi), _dt(pred),
             _s())
        ctx.save_for_backward(pred, target, labels, valid, out)
        ctx.meta = (n, nc, P, int(deconv_w))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        pred, target, labels, valid, out = ctx.saved_tensors
        n, nc, P, deconv_w = ctx.meta
        dpred = torch.zeros_like(pred)
        call("det_mask_loss_bwd", _p(pred), n, nc, P, deconv_w, _p(target), _p(labels), _p(valid), _p(out), _p(g.float().reshape(1)),
             _p(dpred), _dt(pred), _s())
        return dpred, None, None, None, None


def mask_loss(mask_pred, mask_targets, labels, valid, deconv_order=False):
    """mean sigmoid-BCE of the labelled class channel over the valid RoIs (mask_cross_entropy, reduction 'mean').
    mask_pred: (n, num_classes, h, w) logits; or, with ``deconv_order``, the (n * h/2 * w/2 * 4, num_classes) rows of
    FCNMaskHead.forward_rows -- the same logits before the 2x2 pixel shuffle."""
    t = mask_targets.contiguous().float()
    if t.shape[0] == 0:
        return mask_pred.sum() * 0
    return _MaskLoss.apply(mask_pred, t, labels.contiguous(), valid.to(torch.uint8).contiguous(), int(t.shape[2]) if deconv_order else 0)


class _RPNFlatten(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, *ys):
        ys = [y.contiguous() for y in ys]
        B, CH = ys[0].shape[0], ys[0].shape[2]
        L = len(ys)
        hw = [y.shape[1] for y in ys]
        tot = sum(hw) * A
        cls_all = torch.empty((B, tot), device=ys[0].device, dtype=ys[0].dtype)
        reg_all = torch.empty((B, tot, 4), device=ys[0].device, dtype=ys[0].dtype)
        ptrs = (ctypes.c_void_p * L)(*[y.data_ptr() for y in ys])
        hws = (ctypes.c_int * L)(*hw)
        call("det_rpn_flatten_fwd", ptrs, hws, L, B, A, CH, _p(cls_all), _p(reg_all), _dt(ys[0]), _s())
        ctx.meta = (A, B, CH, hw, ys[0].dtype, ys[0].device)
        return cls_all, reg_all

    @staticmethod
    def backward(ctx, dcls, dreg):
        A, B, CH, hw, dtype, dev = ctx.meta
        L = len(hw)
        tot = sum(hw) * A
        dcls = torch.zeros((B, tot), device=dev, dtype=dtype) if dcls is None else dcls.contiguous()
        dreg = torch.zeros((B, tot, 4), device=dev, dtype=dtype) if dreg is None else dreg.contiguous()
        dys = [torch.empty((B, n, CH), device=dev, dtype=dtype) for n in hw]
        ptrs = (ctypes.c_void_p * L)(*[d.data_ptr() for d in dys])
        hws = (ctypes.c_int * L)(*hw)
        call("det_rpn_flatten_bwd", ptrs, hws, L, B, A, CH, _p(dcls), _p(dreg), SWIN_F32 if dtype == torch.float32 else SWIN_BF16, _s())
        return (None,) + tuple(dys)


def rpn_flatten(ys, num_anchors):
    """per-level fused RPN head outputs (B, HW_l, CH) -> (cls_all (B, sum HW_l*A), reg_all (B, sum HW_l*A, 4)),
    anchors ordered (level, h, w, a) as anchor_head.py:474-486 flattens them."""
    return _RPNFlatten.apply(int(num_anchors), *ys)
