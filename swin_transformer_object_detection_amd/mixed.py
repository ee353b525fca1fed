"""fp32 master parameters + bf16 shadow copies for the GEMM/conv operands.

The reference trains with apex O1 (``mmdet/apis/train.py:82-89``): fp32 master weights, half-precision
GEMM inputs, fp32 gradients.  Casting every weight on every use costs three tiny kernels per parameter
per step (cast, cast-backward, gradient accumulate); here all shadows live in ONE flat bf16 buffer that is
refreshed from the masters with a single multi-tensor copy after the optimizer step, and the shadows are
the autograd leaves (their bf16 gradients are gathered into the flat fp32 gradient buckets by ``ddp.py``).
"""
import torch

_SHADOW = {}          # id(master parameter) -> bf16 leaf tensor (a view of the flat shadow buffer)
_SINK = {}            # id(master parameter) -> (fp32 gradient view in a flat bucket, notify())  -- set by ddp.py


def grad_sink(p):
    """Where a weight-gradient kernel may ACCUMULATE the fp32 gradient of master parameter ``p`` directly
    (the flat all-reduce bucket), plus the callback that tells the reducer the gradient has arrived.
    None when no reducer is active: the op then returns the gradient through autograd as usual."""
    return None if p is None else _SINK.get(id(p))


def register_sinks(table):
    _SINK.update(table)


def clear_sinks(ids=None):
    if ids is None:
        _SINK.clear()
    else:
        for i in ids:
            _SINK.pop(i, None)


def weight(p, dtype):
    """The tensor a GEMM/conv should consume for parameter ``p`` in compute dtype ``dtype``."""
    if p is None or p.dtype == dtype:
        return p
    s = _SHADOW.get(id(p))
    if s is not None and s.dtype == dtype:
        return s
    return p.to(dtype)


class ShadowParams:
    def __init__(self, module, dtype=torch.bfloat16, min_numel=1024):
        """Shadows are created for the ``weight`` of Linear / Conv2d / ConvTranspose2d layers with >= min_numel
        elements (the GEMM / conv operands).  Everything else (LayerNorm affine, biases, relative-position bias
        tables, position embeddings) is consumed in fp32 by the kernels and keeps its direct gradient."""
        import torch.nn as nn
        self.dtype = dtype
        seen, self.masters = set(), []
        for m in module.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d, nn.ConvTranspose2d)):
                p = m.weight
                if p.requires_grad and p.dtype == torch.float32 and p.numel() >= min_numel and id(p) not in seen:
                    seen.add(id(p))
                    self.masters.append(p)
        n = sum(p.numel() for p in self.masters)
        dev = self.masters[0].device if self.masters else torch.device("cpu")
        self.flat = torch.empty(n, device=dev, dtype=dtype)
        self.shadows = []
        off = 0
        for p in self.masters:
            v = self.flat[off:off + p.numel()].view_as(p)
            v.requires_grad_(True)
            self.shadows.append(v)
            _SHADOW[id(p)] = v
            off += p.numel()
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """shadow <- master (after every optimizer step): one multi-tensor copy."""
        if self.masters:
            torch._foreach_copy_(self.shadows, self.masters)

    def leaf_of(self, p):
        return _SHADOW.get(id(p), p)

    def release(self):
        for p in self.masters:
            _SHADOW.pop(id(p), None)
