#!/usr/bin/env python
"""Run only the fused MLP kernels a few times (for rocprofv3 counter passes).  usage: mlp_only.py [C] [fwd|bwd] [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swin_transformer_object_detection_amd.ops import functional as Fn  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 96
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
T = {96: 128000, 192: 32000}[C]
x = torch.randn(T, C, device="cuda").bfloat16()
w1 = (torch.randn(4 * C, C, device="cuda") * 0.05).bfloat16()
w2 = (torch.randn(C, 4 * C, device="cuda") * 0.05).bfloat16()
b1 = torch.randn(4 * C, device="cuda") * 0.1
b2 = torch.randn(C, device="cuda") * 0.1
dy = torch.randn(T, C, device="cuda").bfloat16()
for _ in range(n):
    if which == "fwd":
        Fn.mlp_fwd_raw(x, w1, b1, w2, b2)
    else:
        Fn.mlp_bwd_raw(x, dy, w1, b1, w2)
torch.cuda.synchronize()
