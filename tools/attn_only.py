"""Run only the stage-1 window-attention forward (20 launches) -- target of rocprofv3 --pmc passes (development aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swin_transformer_object_detection_amd import ops
from swin_transformer_object_detection_amd.ops import functional as Fn
B, H, W, C, nH = 2, 200, 320, 96, 3
qkv = torch.randn(B, H * W, 3 * C, device="cuda").bfloat16()
qb = torch.randn(3 * C, device="cuda") * 0.1
be = ops.rel_bias_expand(torch.randn(169, nH, device="cuda") * 0.02)
out = torch.empty(B, H * W, C, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B * 29 * 46 * nH, 64, device="cuda")
for _ in range(20):
    Fn.call("swin_window_attn_fwd", Fn._p(qkv), Fn._p(qb), Fn._p(be), Fn._p(out), Fn._p(lse), B, H, W, C, nH, 3, 32 ** -0.5, Fn.SWIN_BF16, Fn._s())
torch.cuda.synchronize()
