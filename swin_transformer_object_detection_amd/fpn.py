"""``FPN`` neck on the hot path.  Drop-in for ``mmdet/models/necks/fpn.py:66-221`` in the
configuration every swin config uses (``add_extra_convs=False``, no norm / activation):
same registry name, kwargs, ``lateral_convs.{i}.conv`` / ``fpn_convs.{i}.conv`` parameter
names (mmcv ConvModule nests the conv under ``.conv``), xavier-uniform init.

Execution plan: feature maps stay channels-last (the backbone's token-major buffers), so a
1x1 lateral conv is a plain GEMM over tokens; the top-down ``+= nearest_upsample`` is one HIP
kernel (``ops.upsample_add``); the 3x3 output convs go to the library conv (MIOpen) in
channels-last; the extra pyramid level ``max_pool2d(k=1, s=2)`` is a strided view.
"""
import contextlib

import torch

from ._lib import half_dtype as _H
import torch.nn as nn
import torch.nn.functional as F

from . import mixed, ops
from .registry import NECKS


class ConvModule(nn.Module):
    """mmcv.cnn.ConvModule as the swin configs use it: a Conv2d stored as ``.conv`` and, with
    ``norm_cfg=dict(type='BN'|'SyncBN')``, a bias-free conv followed by a BatchNorm stored as ``.bn`` (mmcv names the norm
    layer 'bn' for both types; bias='auto' drops the conv bias when a norm follows).  This module only HOLDS the parameters
    (state_dict keys ``conv.weight``, ``conv.bias`` / ``bn.weight``, ``bn.bias``, ``bn.running_mean``, ...): the callers
    run conv / norm / activation on the HIP kernels.  ``sync`` records whether statistics span the ranks (SyncBN)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, conv_cfg=None, norm_cfg=None,
                 act_cfg=None, inplace=False):
        super().__init__()
        if act_cfg is not None or conv_cfg is not None:
            raise NotImplementedError("ConvModule: plain Conv2d; the activation is applied by the caller")
        self.with_norm = norm_cfg is not None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=not self.with_norm)
        self.sync = False
        if self.with_norm:
            ntype = norm_cfg.get('type')
            if ntype not in ('BN', 'SyncBN'):
                raise NotImplementedError(f"ConvModule: norm type {ntype!r} (swin configs: SyncBN)")
            self.sync = ntype == 'SyncBN'
            self.bn = nn.BatchNorm2d(out_channels, eps=norm_cfg.get('eps', 1e-5), momentum=norm_cfg.get('momentum', 0.1))
            for p_ in self.bn.parameters():
                p_.requires_grad = norm_cfg.get('requires_grad', True)

    def forward(self, x):
        x = self.conv(x)
        return self.bn(x) if self.with_norm else x


def _to_cl(x, dtype):
    if x.dtype != dtype:
        x = x.to(dtype)
    return x.contiguous(memory_format=torch.channels_last)


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=True, relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None,
                 norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode='nearest'), compute_dtype=torch.float32):
        super().__init__()
        assert isinstance(in_channels, list)
        if add_extra_convs or start_level != 0 or end_level != -1:
            raise NotImplementedError("only the Mask R-CNN form (extra levels by max-pool) is on the Swin path")
        if upsample_cfg.get('mode', 'nearest') != 'nearest' or 'scale_factor' in upsample_cfg:
            raise NotImplementedError("upsample_cfg must be dict(mode='nearest')")
        if norm_cfg is not None or act_cfg is not None or conv_cfg is not None:
            raise NotImplementedError("swin configs build FPN convs without norm/activation/conv_cfg")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        assert num_outs >= self.num_ins
        self.fp16_enabled = False
        self.compute_dtype = compute_dtype
        self.defer_join = False          # True (set by the detector around its training forward): the caller joins the second stream
        self.upsample_cfg = upsample_cfg.copy()
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for c in in_channels:
            self.lateral_convs.append(ConvModule(c, out_channels, 1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, conv_cfg=conv_cfg,
                                             norm_cfg=norm_cfg, act_cfg=act_cfg))

    def init_weights(self):                                     # fpn.py:163-167
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def _w(self, t):
        return mixed.weight(t, self.compute_dtype)

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        dt = self.compute_dtype
        laterals = []
        for x, lc in zip(inputs, self.lateral_convs):           # fpn.py:175-178: 1x1 conv == GEMM over tokens
            x = _to_cl(x, dt)
            N, C, H, W = x.shape
            tok = x.permute(0, 2, 3, 1).reshape(N * H * W, C)
            y = ops.linear(tok, lc.conv.weight, lc.conv.bias, dt)       # (Cout,Cin,1,1) weight == (Cout,Cin) GEMM operand
            laterals.append(y.view(N, H, W, self.out_channels).permute(0, 3, 1, 2))
        for i in range(len(laterals) - 1, 0, -1):               # fpn.py:182-191
            laterals[i - 1] = ops.upsample_add(laterals[i - 1], laterals[i])
        branched = False
        if dt == _H() and self.out_channels % 64 == 0:
            outs = []
            for i, fc in enumerate(self.fpn_convs):
                # the small levels' convs (P4, P5: one block per CU on half the chip) go to the second stream
                with mixed.small_branch(laterals[i]) as sd:
                    o = ops.conv3x3(laterals[i], fc.conv.weight, fc.conv.bias)
                if sd is not None:
                    mixed.side_outputs(o)
                    branched = True
                outs.append(o)
        else:                                                   # fp32 parity path: library conv
            outs = [F.conv2d(laterals[i].contiguous(memory_format=torch.channels_last),
                             self._w(fc.conv.weight).contiguous(memory_format=torch.channels_last),
                             self._w(fc.conv.bias), padding=1)
                    for i, fc in enumerate(self.fpn_convs)]     # fpn.py:195-197
        for _ in range(self.num_outs - len(outs)):              # fpn.py:202-204: max_pool2d(k=1, s=2)
            with (mixed.small_branch(outs[-1]) if branched else contextlib.nullcontext()) as sd:
                o = outs[-1][:, :, ::2, ::2].contiguous(memory_format=torch.channels_last)
            if sd is not None:
                mixed.side_outputs(o)
            outs.append(o)
        if branched and not self.defer_join:
            mixed.side_join()                                    # a caller that does not join itself (the detector's RPN head does)
        return tuple(outs)
