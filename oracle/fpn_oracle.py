"""Torch-CPU fp32 restatement of FPN.forward -- TEST INFRASTRUCTURE ONLY.

Follows ``mmdet/models/necks/fpn.py:169-221`` for the configuration the swin
configs use (``add_extra_convs=False``, no norm / activation, nearest
upsampling by ``size``): 1x1 laterals, top-down nearest add, 3x3 output convs,
extra levels by ``max_pool2d(k=1, s=2)``.

Pinned: yes -- tests/test_oracle_fpn.py replays the reference-generated golden
vectors (tests/golden/make_golden.py).
"""
import torch
import torch.nn.functional as F


def fpn_forward(inputs, p, num_outs=5, prefix=""):
    """inputs: tuple of NCHW tensors; p: dict with the reference's state_dict keys
    ``lateral_convs.{i}.conv.{weight,bias}``, ``fpn_convs.{i}.conv.{weight,bias}``."""
    n = len(inputs)
    lat = [F.conv2d(inputs[i], p[f"{prefix}lateral_convs.{i}.conv.weight"],
                    p[f"{prefix}lateral_convs.{i}.conv.bias"]) for i in range(n)]  # fpn.py:175-178
    for i in range(n - 1, 0, -1):                                                   # fpn.py:182-191
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
    outs = [F.conv2d(lat[i], p[f"{prefix}fpn_convs.{i}.conv.weight"],
                     p[f"{prefix}fpn_convs.{i}.conv.bias"], padding=1) for i in range(n)]  # fpn.py:195-197
    for _ in range(num_outs - n):                                                   # fpn.py:202-204
        outs.append(F.max_pool2d(outs[-1], 1, stride=2))
    return tuple(outs)


def make_params(in_channels=(96, 192, 384, 768), out_channels=256, seed=0):
    """xavier-uniform weights, zero bias (fpn.py:163-167 via mmcv xavier_init)."""
    g = torch.Generator().manual_seed(seed)
    p = {}

    def xavier(co, ci, k):
        fan_in, fan_out = ci * k * k, co * k * k
        a = (6.0 / (fan_in + fan_out)) ** 0.5
        return (torch.rand(co, ci, k, k, generator=g) * 2 - 1) * a

    for i, ci in enumerate(in_channels):
        p[f"lateral_convs.{i}.conv.weight"] = xavier(out_channels, ci, 1)
        p[f"lateral_convs.{i}.conv.bias"] = 0.05 * torch.randn(out_channels, generator=g)
        p[f"fpn_convs.{i}.conv.weight"] = xavier(out_channels, out_channels, 3)
        p[f"fpn_convs.{i}.conv.bias"] = 0.05 * torch.randn(out_channels, generator=g)
    return p
