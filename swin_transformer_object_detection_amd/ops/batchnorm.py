"""(Sync)BatchNorm + optional ReLU over channel-last activations on the HIP kernels (csrc/batchnorm.hip).

Reference: the Cascade configs build ``ConvModule(conv3x3, norm_cfg=dict(type='SyncBN'))`` in ConvFCBBoxHead
(convfc_bbox_head.py:99-107; configs/swin/cascade_mask_rcnn_swin_*), i.e. torch.nn.SyncBatchNorm followed by ReLU.
Statistics are exchanged the way SyncBatchNorm does -- one small collective forward (per-channel sums + count) and one
backward (per-channel gradient sums) -- on ``torch.distributed``'s default group when it is initialised with more than
one rank; a single process degenerates to plain BatchNorm, as torch.nn.SyncBatchNorm does."""
import torch

from .._lib import half_dtype as _H
import torch.distributed as dist

from .._lib import SWIN_BF16, SWIN_F32, SwinHipError, call, lib
from .functional import _p, _s

_WS = {}


def _ws(dev, C):
    key = (dev, C)
    w = _WS.get(key)
    if w is None:
        w = _WS[key] = torch.empty(lib().det_bn_workspace_bytes(C), dtype=torch.uint8, device=dev)
    return w


def _dt(t):
    if t.dtype == torch.float32:
        return SWIN_F32
    if t.dtype == _H():
        return SWIN_BF16
    raise SwinHipError(f"batch_norm: float32 / bfloat16 activations only, got {t.dtype}")


def _sync():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _rows(x):
    """(N,C,H,W) channels-last or (R,C) contiguous -> (tensor with (R,C) memory, R, C)"""
    if x.dim() == 4:
        x = x.contiguous(memory_format=torch.channels_last)
        N, C, H, W = x.shape
        return x, N * H * W, C
    x = x.contiguous()
    return x, x.shape[0], x.shape[1]


class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu, sync):
        x, R, C = _rows(x)
        dev = x.device
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        sums = torch.empty(2 * C + 1, device=dev, dtype=torch.float32)
        call("det_bn_stats", _p(x), R, C, _p(sums), _p(_ws(dev, C)), _dt(x), _s())
        if sync:
            dist.all_reduce(sums)
        mi = torch.empty(2 * C, device=dev, dtype=torch.float32)
        call("det_bn_finalize", _p(sums), C, float(eps), float(momentum), _p(mi), _p(running_mean), _p(running_var), _s())
        y = torch.empty_like(x)
        call("det_bn_apply", _p(x), _p(y), R, C, _p(mi), _p(g), _p(b), 1 if relu else 0, _dt(x), _s())
        ctx.save_for_backward(x, g, b, mi, sums)
        ctx.meta = (R, C, relu, sync, gamma.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, b, mi, sums = ctx.saved_tensors
        R, C, relu, sync, pdt = ctx.meta
        dy = dy.contiguous(memory_format=torch.channels_last) if dy.dim() == 4 else dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dev = x.device
        bs = torch.empty(2 * C, device=dev, dtype=torch.float32)
        call("det_bn_bwd_reduce", _p(x), _p(dy), R, C, _p(mi), _p(g), _p(b), 1 if relu else 0, _p(bs), _p(_ws(dev, C)), _dt(x), _s())
        dbeta, dgamma = bs[:C].to(pdt), bs[C:].to(pdt)               # this rank's parameter gradients (DDP averages them)
        if sync:
            bs = bs.clone()
            dist.all_reduce(bs)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            call("det_bn_bwd_apply", _p(x), _p(dy), _p(dx), R, C, _p(mi), _p(g), _p(b), 1 if relu else 0, _p(bs), _p(sums[2 * C:]),
                 _dt(x), _s())
        return dx, dgamma, dbeta, None, None, None, None, None, None


def batch_norm(x, gamma, beta, running_mean, running_var, training, eps=1e-5, momentum=0.1, relu=False, sync=None):
    """y = relu?(BN(x)) for a channels-last (N,C,H,W) or (R,C) GPU activation.  Training: batch statistics (over all
    ranks when ``sync``; default: whenever torch.distributed runs with world_size > 1), running statistics updated in
    place.  Eval: running statistics."""
    if not x.is_cuda:
        raise SwinHipError("batch_norm: GPU tensors only")
    if training:
        return _BatchNormTrain.apply(x, gamma, beta, running_mean, running_var, eps, momentum, relu, _sync() if sync is None else sync)
    xr, R, C = _rows(x)
    mi = torch.cat([running_mean.float(), torch.rsqrt(running_var.float() + eps)]).contiguous()
    y = torch.empty_like(xr)
    call("det_bn_apply", _p(xr), _p(y), R, C, _p(mi), _p(gamma.detach().float().contiguous()), _p(beta.detach().float().contiguous()),
         1 if relu else 0, _dt(xr), _s())
    return y
