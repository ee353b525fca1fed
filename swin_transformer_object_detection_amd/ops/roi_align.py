"""``mmcv.ops.RoIAlign`` / ``roi_align`` surface over the HIP kernels.

Signature and semantics follow the reference's call sites
(``base_roi_extractor.py:49-55`` builds ``RoIAlign(spatial_scale=1/s, output_size, sampling_ratio)``
by name with ``getattr``; ``structures.py:353-354`` calls
``roi_align(input, rois, output_size, 1.0, 0, 'avg', True)`` positionally).
"""
import os

import torch

from .._lib import half_dtype as _H
import torch.nn as nn

from .. import _lib
from .._lib import SWIN_BF16, SWIN_F32, SwinHipError, call
from .functional import _p, _s


def _pair(x):
    return (int(x), int(x)) if isinstance(x, int) else (int(x[0]), int(x[1]))


class _RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, rois, output_size, spatial_scale, sampling_ratio, aligned):
        if not inp.is_cuda:
            raise SwinHipError("roi_align: HIP path needs GPU tensors (no CPU fallback)")
        N, C, H, W = inp.shape
        if inp.is_contiguous():
            cl = 0
        elif inp.is_contiguous(memory_format=torch.channels_last):
            cl = 1
        else:
            inp = inp.contiguous()
            cl = 0
        if inp.dtype == torch.float32:
            dt = SWIN_F32
        elif inp.dtype == _H():
            dt = SWIN_BF16
        else:
            raise SwinHipError(f"roi_align: unsupported feature dtype {inp.dtype}")
        rois = rois.contiguous().float()
        K = rois.shape[0]
        ph, pw = output_size
        out = torch.empty((K, C, ph, pw), device=inp.device, dtype=torch.float32,
                          memory_format=torch.channels_last if cl else torch.contiguous_format)
        if K > 0:
            call("roi_align_fwd", _p(inp), _p(rois), _p(out), N, C, H, W, K, ph, pw, float(spatial_scale),
                 int(sampling_ratio), int(bool(aligned)), cl, dt, _s())
        ctx.save_for_backward(rois)
        ctx.cfg = (N, C, H, W, K, ph, pw, float(spatial_scale), int(sampling_ratio), int(bool(aligned)), cl, inp.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        (rois,) = ctx.saved_tensors
        N, C, H, W, K, ph, pw, scale, sr, aligned, cl, in_dtype = ctx.cfg
        mf = torch.channels_last if cl else torch.contiguous_format
        gin = torch.empty((N, C, H, W), device=gout.device, dtype=torch.float32, memory_format=mf).zero_()
        if K > 0:
            gout = gout.float().contiguous(memory_format=mf)
            call("roi_align_bwd", _p(gout), _p(rois), _p(gin), N, C, H, W, K, ph, pw, scale, sr, aligned, cl, _s())
        return gin.to(in_dtype), None, None, None, None, None


def roi_align(input, rois, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True):
    """mmcv.ops.roi_align.  input (N,C,H,W), rois (K,5)=[batch_idx,x1,y1,x2,y2] -> (K,C,ph,pw) float32."""
    if pool_mode != 'avg':
        raise NotImplementedError("only pool_mode='avg' is on the Swin path (every configs/swin file)")
    assert rois.dim() == 2 and rois.size(1) == 5, 'RoI must be (idx, x1, y1, x2, y2)!'
    return _RoIAlignFn.apply(input, rois, _pair(output_size), spatial_scale, sampling_ratio, aligned)


class RoIAlign(nn.Module):
    """mmcv.ops.RoIAlign(output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True,
    use_torchvision=False)."""

    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True,
                 use_torchvision=False):
        super().__init__()
        self.output_size = _pair(output_size)     # read as a 2-tuple at single_level_roi_extractor.py:56
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = int(sampling_ratio)
        self.pool_mode = pool_mode
        self.aligned = aligned
        self.use_torchvision = use_torchvision    # accepted for config compatibility; never used

    def forward(self, input, rois):
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.pool_mode,
                         self.aligned)

    def __repr__(self):
        return (f'{self.__class__.__name__}(output_size={self.output_size}, spatial_scale={self.spatial_scale}, '
                f'sampling_ratio={self.sampling_ratio}, pool_mode={self.pool_mode}, aligned={self.aligned})')


class _RoIAlignMultiLevelFn(torch.autograd.Function):
    """All pyramid levels in one launch (fwd) / one launch (bwd); see csrc/roi_align.hip."""

    @staticmethod
    def forward(ctx, rois, lvls, output_size, strides, sampling_ratio, aligned, out_dtype, *feats):
        import ctypes
        n = len(feats)
        assert 1 <= n <= 4
        f0 = feats[0]
        if not f0.is_cuda:
            raise SwinHipError("roi_align_multilevel: GPU tensors only")
        feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
        C = f0.shape[1]
        dt = SWIN_F32 if f0.dtype == torch.float32 else SWIN_BF16
        rois = rois.contiguous().float()
        lvls = lvls.to(torch.int32).contiguous()
        K = rois.shape[0]
        ph, pw = output_size
        if out_dtype is None or f0.dtype == torch.float32:
            out_dtype = torch.float32
        out = torch.empty((K, C, ph, pw), device=f0.device, dtype=out_dtype, memory_format=torch.channels_last)
        ptrs = (ctypes.c_void_p * n)(*[f.data_ptr() for f in feats])
        Hs = (ctypes.c_int * n)(*[f.shape[2] for f in feats])
        Ws = (ctypes.c_int * n)(*[f.shape[3] for f in feats])
        sc = (ctypes.c_float * n)(*[1.0 / s for s in strides])
        if K > 0:
            call("roi_align_multilevel_fwd", ptrs, Hs, Ws, sc, n, _p(rois), _p(lvls), _p(out), C, K, ph, pw,
                 int(sampling_ratio), int(bool(aligned)), dt, SWIN_F32 if out_dtype == torch.float32 else SWIN_BF16, _s())
        ctx.save_for_backward(rois, lvls)
        ctx.cfg = (n, C, K, ph, pw, tuple(strides), int(sampling_ratio), int(bool(aligned)),
                   [tuple(f.shape) for f in feats], f0.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        import ctypes
        rois, lvls = ctx.saved_tensors
        n, C, K, ph, pw, strides, sr, aligned, shapes, in_dtype = ctx.cfg
        if K > 0:
            g = gout if gout.dtype in (torch.float32, _H()) else gout.float()
            grads = _gather_backward(shapes, strides, [(g, rois, lvls, ph, pw)], C, sr, aligned, in_dtype)
            if grads is not None:
                return (None, None, None, None, None, None, None) + grads
        # scatter form: one flat fp32 accumulator for the whole pyramid: one memset, one cast back to the feature dtype
        sizes = [s[0] * s[1] * s[2] * s[3] for s in shapes]
        flat = torch.zeros(sum(sizes), device=gout.device, dtype=torch.float32)
        offs = [sum(sizes[:i]) for i in range(n)]
        if K > 0:
            if gout.dtype not in (torch.float32, _H()):
                gout = gout.float()
            gout = gout.contiguous(memory_format=torch.channels_last)
            ptrs = (ctypes.c_void_p * n)(*[flat.data_ptr() + 4 * o for o in offs])
            Hs = (ctypes.c_int * n)(*[s[2] for s in shapes])
            Ws = (ctypes.c_int * n)(*[s[3] for s in shapes])
            sc = (ctypes.c_float * n)(*[1.0 / s for s in strides])
            call("roi_align_multilevel_bwd", ptrs, Hs, Ws, sc, n, _p(gout), _p(rois), _p(lvls), C, K, ph, pw, sr, aligned,
                 SWIN_F32 if gout.dtype == torch.float32 else SWIN_BF16, _s())
        flat = flat.to(in_dtype)
        grads = tuple(flat[o:o + m].view(s[0], s[2], s[3], s[1]).permute(0, 3, 1, 2) for o, m, s in zip(offs, sizes, shapes))
        return (None, None, None, None, None, None, None) + grads


_ACC_ON_SIDE = os.environ.get("SWIN_ROI_ACC_SIDE", "1") != "0"      # 0: zero the backward accumulator in backward, on the main stream (A/B)
_GATHER = os.environ.get("SWIN_ROI_GATHER", "1") != "0"             # 0: the scatter (float-atomic) backward of rounds 1-2 (A/B)
_GATHER_WS = {}       # (device index, bytes) -> persistent int32 workspace, zeroed once (the kernels leave its counters zero)


def _gather_backward(shapes, strides, sets, C, sr, aligned, in_dtype):
    """Gradient of the pyramid for RoI sets [(gout, rois, lvls, ph, pw), ...] in gather form (roi_align_multilevel_bwd_gather): a
    tuple of (N, C, H_l, W_l) channels-last gradients in ``in_dtype``, or None when the shapes do not fit that form."""
    import ctypes
    n = len(shapes)
    sets = [t for t in sets if t[0] is not None and t[1].shape[0] > 0]
    if not _GATHER or not sets or len(sets) > 4 or C % 4 != 0 or any(ph > 16 or pw > 16 for _, _, _, ph, pw in sets):
        return None
    dev = sets[0][0].device
    N = shapes[0][0]
    gdt = sets[0][0].dtype
    if gdt not in (torch.float32, _H()) or any(t[0].dtype != gdt for t in sets) or in_dtype not in (torch.float32, _H()):
        return None
    Hs = (ctypes.c_int * n)(*[s[2] for s in shapes])
    Ws = (ctypes.c_int * n)(*[s[3] for s in shapes])
    sc = (ctypes.c_float * n)(*[1.0 / s for s in strides])
    ktot = sum(t[1].shape[0] for t in sets)
    nb = int(_lib.lib().roi_align_gather_workspace_bytes(Hs, Ws, n, N, ktot))
    if nb <= 0:
        return None
    key = (dev.index, nb, N, tuple((s[2], s[3]) for s in shapes))      # the layout inside the workspace follows the pyramid's tile counts
    ws = _GATHER_WS.get(key)
    if ws is None:
        ws = _GATHER_WS[key] = torch.zeros((nb + 3) // 4, device=dev, dtype=torch.int32)
    sizes = [s[0] * s[1] * s[2] * s[3] for s in shapes]
    flat = torch.empty(sum(sizes), device=dev, dtype=in_dtype)          # fully written by the kernel: no fill, no cast
    offs = [sum(sizes[:i]) for i in range(n)]
    esz = flat.element_size()
    ptrs = (ctypes.c_void_p * n)(*[flat.data_ptr() + esz * o for o in offs])
    m = len(sets)
    gouts = [t[0].contiguous(memory_format=torch.channels_last) for t in sets]
    gp = (ctypes.c_void_p * m)(*[g.data_ptr() for g in gouts])
    rp = (ctypes.c_void_p * m)(*[t[1].data_ptr() for t in sets])
    lp = (ctypes.c_void_p * m)(*[t[2].data_ptr() for t in sets])
    Ks = (ctypes.c_int * m)(*[t[1].shape[0] for t in sets])
    phs = (ctypes.c_int * m)(*[t[3] for t in sets])
    pws = (ctypes.c_int * m)(*[t[4] for t in sets])
    code = lambda d: SWIN_F32 if d == torch.float32 else SWIN_BF16      # noqa: E731
    call("roi_align_multilevel_bwd_gather", ptrs, Hs, Ws, sc, n, N, m, gp, rp, lp, Ks, phs, pws, C, sr, aligned, code(gdt), code(in_dtype),
         _p(ws), nb, _s())
    return tuple(flat[o:o + q].view(s[0], s[2], s[3], s[1]).permute(0, 3, 1, 2) for o, q, s in zip(offs, sizes, shapes))


class _RoIAlignMultiLevelGroupFn(torch.autograd.Function):
    """Several RoI sets (e.g. the bbox head's 7x7 and the mask head's 14x14 RoIs of one R-CNN stage) pooled from the SAME
    pyramid: one forward launch per set, and in backward ONE fp32 accumulator for the pyramid shared by all sets -- one
    memset and one cast back to the feature dtype instead of one each per set, and no gradient additions between the
    sets (the accumulator is 174 MB at 2x800x1280: the memset + cast pair costs about as much as the backward kernel)."""

    @staticmethod
    def forward(ctx, specs, strides, sampling_ratio, aligned, out_dtype, n, *tensors):
        import ctypes
        feats, rest = tensors[:n], tensors[n:]
        assert 1 <= n <= 4 and len(rest) == 2 * len(specs)
        f0 = feats[0]
        if not f0.is_cuda:
            raise SwinHipError("roi_align_multilevel: GPU tensors only")
        feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
        C = f0.shape[1]
        dt = SWIN_F32 if f0.dtype == torch.float32 else SWIN_BF16
        if out_dtype is None or f0.dtype == torch.float32:
            out_dtype = torch.float32
        ptrs = (ctypes.c_void_p * n)(*[f.data_ptr() for f in feats])
        Hs = (ctypes.c_int * n)(*[f.shape[2] for f in feats])
        Ws = (ctypes.c_int * n)(*[f.shape[3] for f in feats])
        sc = (ctypes.c_float * n)(*[1.0 / s for s in strides])
        outs, saved = [], []
        for gi, (ph, pw) in enumerate(specs):
            rois = rest[2 * gi].contiguous().float()
            lvls = rest[2 * gi + 1].to(torch.int32).contiguous()
            K = rois.shape[0]
            out = torch.empty((K, C, ph, pw), device=f0.device, dtype=out_dtype, memory_format=torch.channels_last)
            if K > 0:
                call("roi_align_multilevel_fwd", ptrs, Hs, Ws, sc, n, _p(rois), _p(lvls), _p(out), C, K, ph, pw,
                     int(sampling_ratio), int(bool(aligned)), dt, SWIN_F32 if out_dtype == torch.float32 else SWIN_BF16, _s())
            outs.append(out)
            saved += [rois, lvls]
        ctx.save_for_backward(*saved)
        ctx.cfg = (n, C, tuple(specs), tuple(strides), int(sampling_ratio), int(bool(aligned)), [tuple(f.shape) for f in feats],
                   f0.dtype)
        # The backward's fp32 accumulator (174 MB at 2x800x1280) needs 35-85 us of zero fill at the head of the data-gradient chain.
        # With a second stream it is allocated and zeroed THERE, now, while this stream runs the heads; backward waits for the event.
        ctx.acc = None
        gather_ok = _GATHER and all(ph <= 16 and pw <= 16 for ph, pw in specs) and len(specs) <= 4
        if _ACC_ON_SIDE and not gather_ok and any(f.requires_grad for f in tensors[:n]):
            from .. import mixed
            with mixed.on_side(f0.device) as sd:
                if sd is not None:
                    acc = torch.zeros(sum(f.numel() for f in feats), device=f0.device, dtype=torch.float32)
                    ev = torch.cuda.Event()
                    ev.record(sd)
                    ctx.acc = (acc, ev)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        import ctypes
        saved = ctx.saved_tensors
        n, C, specs, strides, sr, aligned, shapes, in_dtype = ctx.cfg
        sizes = [s[0] * s[1] * s[2] * s[3] for s in shapes]
        dev = saved[0].device
        sets = []
        for gi, ((ph, pw), gout) in enumerate(zip(specs, gouts)):
            if gout is not None and gout.dtype not in (torch.float32, _H()):
                gout = gout.float()
            sets.append((gout, saved[2 * gi], saved[2 * gi + 1], ph, pw))
        grads = _gather_backward(shapes, strides, sets, C, sr, aligned, in_dtype)
        if grads is not None:
            ctx.acc = None
            return (None, None, None, None, None, None) + grads + (None,) * (2 * len(specs))
        if getattr(ctx, 'acc', None) is not None and ctx.acc[0].numel() == sum(sizes):
            flat, ev = ctx.acc
            ctx.acc = None
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ev)                       # zeroed on the second stream during the forward pass
            flat.record_stream(cur)                  # its memory belongs to that stream's pool; it is used (and freed) here
        else:
            flat = torch.zeros(sum(sizes), device=dev, dtype=torch.float32)
        offs = [sum(sizes[:i]) for i in range(n)]
        ptrs = (ctypes.c_void_p * n)(*[flat.data_ptr() + 4 * o for o in offs])
        Hs = (ctypes.c_int * n)(*[s[2] for s in shapes])
        Ws = (ctypes.c_int * n)(*[s[3] for s in shapes])
        sc = (ctypes.c_float * n)(*[1.0 / s for s in strides])
        for gi, ((ph, pw), gout) in enumerate(zip(specs, gouts)):
            rois, lvls = saved[2 * gi], saved[2 * gi + 1]
            K = rois.shape[0]
            if gout is None or K == 0:
                continue
            if gout.dtype not in (torch.float32, _H()):
                gout = gout.float()
            gout = gout.contiguous(memory_format=torch.channels_last)
            call("roi_align_multilevel_bwd", ptrs, Hs, Ws, sc, n, _p(gout), _p(rois), _p(lvls), C, K, ph, pw, sr, aligned,
                 SWIN_F32 if gout.dtype == torch.float32 else SWIN_BF16, _s())
        flat = flat.to(in_dtype)
        grads = tuple(flat[o:o + m].view(s[0], s[2], s[3], s[1]).permute(0, 3, 1, 2) for o, m, s in zip(offs, sizes, shapes))
        return (None, None, None, None, None, None) + grads + (None,) * (2 * len(specs))


def roi_align_multilevel_group(feats, groups, strides, sampling_ratio=0, aligned=True, out_dtype=None):
    """``groups``: list of (rois (K_i,5), lvls (K_i,), output_size) pooled from the same ``feats`` -> list of (K_i, C, ph_i, pw_i).
    Values and gradients equal separate roi_align_multilevel calls; the backward shares one fp32 accumulator."""
    specs = tuple(_pair(g[2]) for g in groups)
    flat = []
    for g in groups:
        flat += [g[0], g[1]]
    return list(_RoIAlignMultiLevelGroupFn.apply(specs, tuple(strides), sampling_ratio, aligned, out_dtype, len(feats), *feats, *flat))


def roi_align_multilevel(feats, rois, lvls, output_size, strides, sampling_ratio=0, aligned=True, out_dtype=None):
    """feats: list of (N,C,H_l,W_l) channels-last maps; rois (K,5); lvls (K,) level per RoI (< 0: skip, zero row).
    -> (K, C, ph, pw) float32 (channels-last), or ``out_dtype`` = bfloat16 for bf16 features (the fp32 result rounded
    once, what the bf16 heads would do with a separate cast).  Same arithmetic as RoIAlign level by level."""
    return _RoIAlignMultiLevelFn.apply(rois, lvls, _pair(output_size), tuple(strides), sampling_ratio, aligned, out_dtype, *feats)
