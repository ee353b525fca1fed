"""torch-facing wrappers of the C ABI (include/swin_hip.h): tensors in, tensors out.

Every function here launches hand-written HIP kernels through ctypes on torch's
current stream.  torch only provides device memory, streams and autograd
bookkeeping.  There is deliberately no CPU / eager fallback: a tensor that is
not on a GPU, or a missing library, raises.
"""
import contextlib
import ctypes
import os

import torch

from .._lib import half_dtype as _H

from .. import _lib
from .._lib import SWIN_BF16, SWIN_F32, SwinHipError, call

LN_EPS = 1e-5


def _dt(t):
    if t.dtype == _H():
        return SWIN_BF16
    if t.dtype == torch.float32:
        return SWIN_F32
    raise SwinHipError(f"unsupported activation dtype {t.dtype} (float32 or bfloat16)")


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise SwinHipError("HIP path needs GPU tensors (no CPU fallback)")
        if not t.is_contiguous():
            raise SwinHipError("HIP path needs contiguous tensors")


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _s():
    """the current HIP stream's handle.  torch.cuda.current_stream() builds a Stream object through three Python layers (~9 us;
    ~190 calls per step were 1.5 ms of a host-bound step); the raw query is one C call."""
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t):
    if t.dtype != torch.float32:
        raise SwinHipError("parameters must be float32")
    return t


def _grad_buf(param):
    """(fp32 buffer a kernel ACCUMULATES param's gradient into, value to return through autograd, notify()).

    With a reducer active (mixed.grad_sink) the buffer is the parameter's slice of the flat all-reduce bucket
    (zeroed once per step), autograd gets None and notify() tells the reducer; otherwise a fresh zero tensor
    that is returned to autograd."""
    from .. import mixed
    s = mixed.grad_sink(param)
    if s is not None:
        return s[0], None, s[1]
    z = torch.zeros_like(param, dtype=torch.float32)
    return z, z, (lambda: None)


# --------------------------------------------------------------------------------------
# LayerNorm
# --------------------------------------------------------------------------------------
def _ln_ws(rows, C, like):
    """scratch for the per-block partial sums of the LayerNorm parameter gradients (no initialisation needed)"""
    n = _lib.lib().swin_layernorm_bwd_workspace_bytes(rows, C, _dt(like))
    return torch.empty(n // 4, device=like.device, dtype=torch.float32)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        _chk(x, w, b)
        C = x.shape[-1]
        rows = x.numel() // C
        y = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        call("swin_layernorm_fwd", _p(x), _p(_f32(w)), _p(_f32(b)), _p(y), _p(mean), _p(rstd), rows, C, eps, _dt(x), _s())
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.bias = b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        C = x.shape[-1]
        rows = x.numel() // C
        dx = torch.empty_like(x)
        dwb, dw, nw = _grad_buf(w)
        dbb, db, nb = _grad_buf(ctx.bias)
        call("swin_layernorm_bwd", _p(dy), _p(x), _p(w), _p(mean), _p(rstd), None, _p(dx), None, None, 1,
             _p(dwb), _p(dbb), rows, C, _dt(x), _p(_ln_ws(rows, C, x)), _s())
        nw(); nb()
        return dx, dw, db, None


def layer_norm(x, weight, bias, eps=LN_EPS):
    return _LayerNorm.apply(x, weight, bias, eps)


class _AddLayerNorm(torch.autograd.Function):
    """xo = x + scale[b]*y ; n = LN(xo).  Returns (xo, n)."""

    @staticmethod
    def forward(ctx, x, y, scale, rows_per_sample, w, b, eps):
        _chk(x, y, scale, w, b)
        C = x.shape[-1]
        rows = x.numel() // C
        xo = torch.empty_like(x)
        n = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        call("swin_add_layernorm_fwd", _p(x), _p(y), _p(scale), rows_per_sample, _p(_f32(w)), _p(_f32(b)), _p(xo), _p(n),
             _p(mean), _p(rstd), rows, C, eps, _dt(x), _s())
        ctx.save_for_backward(xo, w, mean, rstd, scale)
        ctx.rps = rows_per_sample
        ctx.bias = b
        return xo, n

    @staticmethod
    def backward(ctx, dxo, dn):
        xo, w, mean, rstd, scale = ctx.saved_tensors
        C = xo.shape[-1]
        rows = xo.numel() // C
        dn = torch.zeros_like(xo) if dn is None else dn.contiguous()
        dxo = None if dxo is None else dxo.contiguous()
        dx = torch.empty_like(xo)
        dyb = torch.empty_like(xo) if scale is not None else None
        dwb, dw, nw = _grad_buf(w)
        dbb, db, nb = _grad_buf(ctx.bias)
        call("swin_layernorm_bwd", _p(dn), _p(xo), _p(w), _p(mean), _p(rstd), _p(dxo), _p(dx), _p(dyb), _p(scale),
             ctx.rps, _p(dwb), _p(dbb), rows, C, _dt(xo), _p(_ln_ws(rows, C, xo)), _s())
        nw(); nb()
        return dx, (dyb if scale is not None else dx), None, None, dw, db, None


def add_layer_norm(x, y, scale, rows_per_sample, weight, bias, eps=LN_EPS):
    return _AddLayerNorm.apply(x, y, scale, rows_per_sample, weight, bias, eps)


class _AddScaled(torch.autograd.Function):
    """xo = x + scale[b]*y (residual + DropPath) without a following norm."""

    @staticmethod
    def forward(ctx, x, y, scale, rows_per_sample):
        _chk(x, y, scale)
        C = x.shape[-1]
        rows = x.numel() // C
        xo = torch.empty_like(x)
        call("swin_add_layernorm_fwd", _p(x), _p(y), _p(scale), rows_per_sample, None, None, _p(xo), None, None, None,
             rows, C, LN_EPS, _dt(x), _s())
        ctx.save_for_backward(scale)
        ctx.rps = rows_per_sample
        return xo

    @staticmethod
    def backward(ctx, dxo):
        (scale,) = ctx.saved_tensors
        if scale is None:
            return dxo, dxo, None, None
        B = scale.numel()
        dy = (dxo.reshape(B, -1) * scale.to(dxo.dtype).view(B, 1)).view_as(dxo)
        return dxo, dy, None, None


def add_scaled(x, y, scale, rows_per_sample):
    return _AddScaled.apply(x, y, scale, rows_per_sample)


# --------------------------------------------------------------------------------------
# Window attention
# --------------------------------------------------------------------------------------
def rel_bias_expand(table):
    _chk(table)
    nH = table.shape[1]
    out = torch.empty(nH, 64, 64, device=table.device, dtype=torch.float32)
    call("swin_rel_bias_expand", _p(_f32(table)), _p(out), nH, _s())
    return out


def rel_bias_expand_multi(tables, outs):
    """outs[i] (nH_i, 64, 64) f32 <- the expansion of tables[i] (169, nH_i) f32, all in one launch"""
    n = len(tables)
    if n == 0:
        return
    for t, o in zip(tables, outs):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape[0] == 169 and t.is_contiguous()
                and o.dtype == torch.float32 and o.is_contiguous() and o.numel() == t.shape[1] * 4096):
            raise SwinHipError("rel_bias_expand_multi: contiguous f32 (169, nH) tables and (nH, 64, 64) outputs")
    tp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tables])
    op = (ctypes.c_void_p * n)(*[t.data_ptr() for t in outs])
    hh = (ctypes.c_int * n)(*[t.shape[1] for t in tables])
    call("swin_rel_bias_expand_multi", tp, op, hh, n, _s())


class _WindowAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, qkv_bias, table, B, H, W, nH, shift):
        _chk(qkv, qkv_bias, table)
        C = qkv.shape[-1] // 3
        assert qkv.numel() == B * H * W * 3 * C
        bias_exp = rel_bias_expand(table)
        out = torch.empty(B, H * W, C, device=qkv.device, dtype=qkv.dtype)
        nW = ((H + 6) // 7) * ((W + 6) // 7)
        lse = torch.empty(B * nW * nH, 64, device=qkv.device, dtype=torch.float32)
        scale = float((C // nH) ** -0.5)
        call("swin_window_attn_fwd", _p(qkv), _p(qkv_bias), _p(bias_exp), _p(out), _p(lse), B, H, W, C, nH, shift, scale,
             _dt(qkv), _s())
        ctx.save_for_backward(qkv, qkv_bias, bias_exp, lse)
        ctx.geom = (B, H, W, C, nH, shift, scale)
        ctx.table = table
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, qkv_bias, bias_exp, lse = ctx.saved_tensors
        B, H, W, C, nH, shift, scale = ctx.geom
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        dbexp = torch.zeros_like(bias_exp)
        padded = (H % 7 != 0) or (W % 7 != 0)
        dpb, dpad, npad = _grad_buf(qkv_bias) if padded else (None, None, lambda: None)
        ws_bytes = _lib.lib().swin_window_attn_bwd_workspace_bytes(B, H, W, nH, _dt(qkv))
        ws = torch.empty(max(ws_bytes, 16), device=qkv.device, dtype=torch.uint8)
        call("swin_window_attn_bwd", _p(qkv), _p(qkv_bias), _p(bias_exp), _p(lse), _p(dout), _p(dqkv), _p(dbexp), _p(dpb),
             _p(ws), B, H, W, C, nH, shift, scale, _dt(qkv), _s())
        dtb, dtable, nt = _grad_buf(ctx.table)
        call("swin_rel_bias_reduce", _p(dbexp), _p(dtb), nH, _s())
        nt()
        # qkv.bias also receives a gradient from the qkv Linear: its arrival is signalled there, not here
        return dqkv, dpad, dtable, None, None, None, None, None


def window_attention(qkv, qkv_bias, table, B, H, W, num_heads, shift):
    """qkv (B, H*W, 3C) on the natural token grid -> attention output (B, H*W, C).

    Pad, roll, window partition/reverse, bias, mask, softmax and PV are one kernel."""
    return _WindowAttention.apply(qkv, qkv_bias, table, B, H, W, num_heads, shift)


# --------------------------------------------------------------------------------------
# bias + GELU
# --------------------------------------------------------------------------------------
class _BiasGelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias):
        _chk(x, bias)
        C = x.shape[-1]
        rows = x.numel() // C
        y = torch.empty_like(x)
        call("swin_bias_gelu_fwd", _p(x), _p(bias), _p(y), rows, C, _dt(x), _s())
        ctx.save_for_backward(x, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, bias = ctx.saved_tensors
        dy = dy.contiguous()
        C = x.shape[-1]
        rows = x.numel() // C
        dx = torch.empty_like(x)
        dbb, dbias, nb = _grad_buf(bias) if bias is not None else (None, None, lambda: None)
        call("swin_bias_gelu_bwd", _p(dy), _p(x), _p(bias), _p(dx), _p(dbb), rows, C, _dt(x), _s())
        nb()
        return dx, dbias


def bias_gelu(x, bias):
    return _BiasGelu.apply(x, bias)


# --------------------------------------------------------------------------------------
# PatchMerging gather + LN, PatchEmbed im2row
# --------------------------------------------------------------------------------------
class _PatchMergeLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, B, H, W, eps):
        _chk(x, w, b)
        C = x.shape[-1]
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        y = torch.empty(B, Ho * Wo, 4 * C, device=x.device, dtype=x.dtype)
        mean = torch.empty(B * Ho * Wo, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        call("swin_patch_merge_ln_fwd", _p(x), _p(_f32(w)), _p(_f32(b)), _p(y), _p(mean), _p(rstd), B, H, W, C, eps,
             _dt(x), _s())
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.geom = (B, H, W, C)
        ctx.bias = b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        B, H, W, C = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dwb, dw, nw = _grad_buf(w)
        dbb, db, nb = _grad_buf(ctx.bias)
        ws = _ln_ws(B * ((H + 1) // 2) * ((W + 1) // 2), 4 * C, x)
        call("swin_patch_merge_ln_bwd", _p(dy), _p(x), _p(w), _p(mean), _p(rstd), _p(dx), _p(dwb), _p(dbb), B, H, W, C,
             _dt(x), _p(ws), _s())
        nw(); nb()
        return dx, dw, db, None, None, None, None


def patch_merge_layer_norm(x, weight, bias, B, H, W, eps=LN_EPS):
    return _PatchMergeLN.apply(x, weight, bias, B, H, W, eps)


class _PatchIm2Row(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, out_dtype):
        _chk(img)
        if img.dtype != torch.float32:
            raise SwinHipError("image must be float32")
        B, Cin, Hi, Wi = img.shape
        if Cin != 3:
            raise SwinHipError("patch_im2row: in_chans must be 3")
        Ho, Wo = (Hi + 3) // 4, (Wi + 3) // 4
        rows = torch.empty(B * Ho * Wo, 48, device=img.device, dtype=out_dtype)
        call("swin_patch_im2row", _p(img), _p(rows), B, Hi, Wi, _dt(rows), _s())
        ctx.geom = (B, Hi, Wi, Ho, Wo)
        return rows

    @staticmethod
    def backward(ctx, drows):
        # pure layout (non-overlapping patches): only needed when the image itself requires grad
        B, Hi, Wi, Ho, Wo = ctx.geom
        g = drows.float().view(B, Ho, Wo, 3, 4, 4).permute(0, 3, 1, 4, 2, 5).reshape(B, 3, Ho * 4, Wo * 4)
        return g[:, :, :Hi, :Wi].contiguous(), None


def patch_im2row(img, out_dtype):
    return _PatchIm2Row.apply(img, out_dtype)


# --------------------------------------------------------------------------------------
# FPN top-down
# --------------------------------------------------------------------------------------
def _nchw_layout(t):
    """(channels_last_flag, N, C, H, W) of a logically-NCHW tensor; raises if neither layout."""
    N, C, H, W = t.shape
    if t.is_contiguous():
        return 0, N, C, H, W
    if t.is_contiguous(memory_format=torch.channels_last):
        return 1, N, C, H, W
    raise SwinHipError("tensor must be NCHW-contiguous or channels_last")


class _UpsampleAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fine, coarse):
        cl, N, C, Hf, Wf = _nchw_layout(fine)
        cl2, _, _, Hc, Wc = _nchw_layout(coarse)
        if cl != cl2 or not fine.is_cuda:
            raise SwinHipError("fine/coarse must share a memory layout and live on the GPU")
        out = torch.empty_like(fine, memory_format=torch.preserve_format)
        call("fpn_upsample_add_out_fwd", _p(fine), _p(coarse), _p(out), N, C, Hf, Wf, Hc, Wc, cl, _dt(fine), _s())
        ctx.geom = (cl, N, C, Hf, Wf, Hc, Wc)
        return out

    @staticmethod
    def backward(ctx, dout):
        cl, N, C, Hf, Wf, Hc, Wc = ctx.geom
        dout = dout.contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)
        dcoarse = torch.empty((N, C, Hc, Wc), device=dout.device, dtype=dout.dtype,
                              memory_format=torch.channels_last if cl else torch.contiguous_format)
        call("fpn_upsample_add_out_bwd", _p(dout), _p(dcoarse), N, C, Hf, Wf, Hc, Wc, cl, _dt(dout), _s())
        return dout, dcoarse


def upsample_add(fine, coarse):
    """fine + nearest_upsample(coarse, size=fine.shape[2:])  (fpn.py:188-191)."""
    return _UpsampleAdd.apply(fine, coarse)


# --------------------------------------------------------------------------------------
# 3x3 convolution (bf16, channels-last) on the MFMA implicit-GEMM kernel
# --------------------------------------------------------------------------------------
_SPLITK_BYTES = {}


def _conv3x3_raw(x_cl, w_khwc, bias, relu, gate=None):
    """x_cl: (N,C,H,W) bf16 with channels_last strides; w_khwc: (Cout,3,3,Cin) bf16 contiguous.  gate (N,Cout,H,W) bf16
    channels-last: the output is zeroed where gate <= 0 (a data gradient that includes the ReLU backward of the layer below).
    Few-tile maps (coarse pyramid levels) go through the split-K entry with a workspace from the caching allocator."""
    N, Cin, H, W = x_cl.shape
    Cout = w_khwc.shape[0]
    y = torch.empty((N, Cout, H, W), device=x_cl.device, dtype=_H(), memory_format=torch.channels_last)
    key = (N, H, W, Cin, Cout)
    nb = _SPLITK_BYTES.get(key)
    if nb is None:
        nb = _SPLITK_BYTES[key] = int(_lib.lib().conv3x3_splitk_workspace_bytes(N, H, W, Cin, Cout))
    if nb:
        ws = torch.empty(nb, device=x_cl.device, dtype=torch.uint8)          # used on the current stream only
        call("conv3x3_nhwc_bf16_ws", _p(x_cl), _p(w_khwc), _p(bias), _p(gate), _p(y), N, H, W, Cin, Cout, int(relu), _p(ws), nb, _s())
    elif gate is not None:
        call("conv3x3_nhwc_bf16_gated", _p(x_cl), _p(w_khwc), _p(bias), _p(gate), _p(y), N, H, W, Cin, Cout, _s())
    else:
        call("conv3x3_nhwc_bf16", _p(x_cl), _p(w_khwc), _p(bias), _p(y), N, H, W, Cin, Cout, int(relu), _s())
    return y


def _im2col3x3(x_cl):
    """(N,C,H,W) channels-last -> (N*H*W, 9*C) with column order (ky, kx, c) -- used for the weight gradient."""
    N, C, H, W = x_cl.shape
    xp = torch.nn.functional.pad(x_cl.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1))      # (N, H+2, W+2, C)
    cols = torch.cat([xp[:, ky:ky + H, kx:kx + W, :] for ky in range(3) for kx in range(3)], dim=-1)
    return cols.reshape(N * H * W, 9 * C)


def conv_dgrad_layout_multi(srcs, dsts):
    """dsts[i] (Cin,3,3,Cout) <- data-gradient layout of srcs[i], a (Cout,Cin,3,3) bf16 tensor in channels-last memory
    (= (Cout,3,3,Cin) contiguous); every tensor of the list in one launch."""
    n = len(srcs)
    if n == 0:
        return
    for s_, d_ in zip(srcs, dsts):
        if not (s_.is_cuda and s_.dtype == _H() and s_.dim() == 4 and s_.permute(0, 2, 3, 1).is_contiguous()
                and d_.is_contiguous() and d_.dtype == _H() and d_.numel() == s_.numel()):
            raise SwinHipError("conv_dgrad_layout_multi: (Cout,Cin,3,3) bf16 sources in channels-last memory, contiguous bf16 outputs")
    sp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs])
    dp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dsts])
    co = (ctypes.c_int * n)(*[t.shape[0] for t in srcs])
    ci = (ctypes.c_int * n)(*[t.shape[1] for t in srcs])
    call("conv_dgrad_layout_multi", sp, dp, co, ci, n, _s())


def linear_t_layout_multi(srcs, dsts):
    """dsts[i] (cols, rows) <- srcs[i] (rows, cols)^T for a list of contiguous bf16 matrices, in as few launches as the kernel's
    table allows (the transposed weights of the data-gradient GEMMs on the hand-written kernel)."""
    n = len(srcs)
    if n == 0:
        return
    for s_, d_ in zip(srcs, dsts):
        if not (s_.is_cuda and s_.dtype == _H() and s_.dim() == 2 and s_.is_contiguous() and d_.is_contiguous()
                and d_.dtype == _H() and tuple(d_.shape) == (s_.shape[1], s_.shape[0])):
            raise SwinHipError("linear_t_layout_multi: contiguous bf16 (rows, cols) sources and (cols, rows) outputs")
    sp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs])
    dp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dsts])
    rr = (ctypes.c_int * n)(*[t.shape[0] for t in srcs])
    cc = (ctypes.c_int * n)(*[t.shape[1] for t in srcs])
    call("linear_t_layout_multi", sp, dp, rr, cc, n, _s())


class _Conv3x3(torch.autograd.Function):
    """Inputs: x, weight (compute-dtype leaf: the bf16 shadow or a cast of the master), bias (fp32 master), relu,
    weight_master.  The re-laid-out weights are cached per step (mixed.derived); with a reducer active the weight
    gradient of ALL uses of a shared conv (the RPN conv runs on five levels) is summed in one (Cout,3,3,Cin) fp32
    accumulator and folded into the parameter's all-reduce bucket after the last use's backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, weight_master, x_is_relu=False):
        from .. import mixed
        ctx.x_is_relu = bool(x_is_relu)
        if x.dtype != _H() or not x.is_cuda:
            raise SwinHipError("conv3x3: bf16 GPU activations only (fp32 parity runs use the library conv)")
        x = x.contiguous(memory_format=torch.channels_last)
        if weight.dtype == _H() and mixed.is_khwc(weight):
            w = weight.detach().permute(0, 2, 3, 1)          # resident in the kernel's layout (mixed.khwc_resident_): a view
        else:
            w = mixed.derived(weight_master, 'khwc',
                              lambda: weight.detach().to(_H()).permute(0, 2, 3, 1).contiguous())   # (Cout,3,3,Cin)
        b = None if bias is None else _f32(bias.detach().float()).contiguous()
        y = _conv3x3_raw(x, w, b, relu)
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.relu, ctx.has_bias = relu, bias is not None
        ctx.masters = (weight_master, bias)
        ctx.counted = False
        if mixed.grad_sink(weight_master) is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            mixed.use_begin(weight_master)
            ctx.counted = True
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import mixed
        x, weight, y = ctx.saved_tensors
        w_master, b_master = ctx.masters
        dy = dy.contiguous(memory_format=torch.channels_last)
        if ctx.relu and not mixed.gated_take(dy, y):       # the producer of dy may have applied this ReLU's backward already
            dy = torch.ops.aten.threshold_backward(dy, y, 0)
        N, Cin, H, W = x.shape
        Cout = weight.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dx = conv(dy, rot180(w) with in/out swapped): (Cin, 3, 3, Cout)
            sh = mixed.shadow_of(w_master)
            if sh is not None and sh.data_ptr() == weight.data_ptr() and sh.dtype == _H() and mixed.is_khwc(sh):
                wt = mixed.conv_dgrad_weight(w_master, sh)          # persistent, rebuilt for all convs by one launch per step
            else:
                wt = mixed.derived(w_master, 'dgrad',
                                   lambda: weight.detach().to(_H()).flip(2, 3).permute(1, 2, 3, 0).contiguous())
            if ctx.x_is_relu:
                # x is the ReLU output of the layer below: its ReLU backward rides in this kernel's epilogue
                dx = _conv3x3_raw(dy, wt, None, False, gate=x)
                mixed.gated_mark(dx, x)
            else:
                dx = _conv3x3_raw(dy, wt, None, False)
        need_w = ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        if need_w or need_b:
            ws = mixed.grad_sink(w_master) if ctx.counted else None
            bs = mixed.grad_sink(b_master) if (need_b and ctx.counted) else None
            if ws is not None and tuple(ws[0].shape) != (Cout, Cin, 3, 3):
                ws = None
            direct = ws is not None and mixed.is_khwc(ws[0])      # the bucket view itself has the kernel's layout: no fold
            if direct:
                dwf = ws[0].permute(0, 2, 3, 1)
            elif ws is not None:
                dwf = mixed.step_buffer(w_master, 'dw_khwc', (Cout, 3, 3, Cin), x.device)
            else:
                dwf = torch.zeros(Cout, 3, 3, Cin, device=x.device, dtype=torch.float32)
            dbf = None
            if need_b:
                dbf = bs[0] if bs is not None else torch.zeros(Cout, device=x.device, dtype=torch.float32)
            if ws is not None:
                # sink mode: nothing on the main stream reads the result before the reducer -> weight-gradient stream
                last = mixed.use_end(w_master) == 0
                side_ok = bs is not None or not need_b        # a bias gradient returned through autograd is read on the main stream
                if direct:
                    # one C call: fork the side stream behind the producer of dy and launch there (no stream context, no events
                    # made per call: the step had become host-bound on exactly that bookkeeping)
                    sst = mixed.fork_to_side(x.device, dy, x) if side_ok else None
                    call("wgrad_conv3x3_nhwc_bf16", _p(dy), _p(x), _p(dwf), _p(dbf), N, H, W, Cin, Cout,
                         ctypes.c_void_p(sst) if sst is not None else _s())
                else:
                    with (mixed.on_side(x.device, dy, x) if side_ok else contextlib.nullcontext()):
                        call("wgrad_conv3x3_nhwc_bf16", _p(dy), _p(x), _p(dwf), _p(dbf), N, H, W, Cin, Cout, _s())
                        if last:
                            ws[0].add_(dwf.permute(0, 3, 1, 2))

                def notify(ws=ws, bs=bs, w_master=w_master, direct=direct):
                    if not direct:
                        mixed.step_buffer_done(w_master, 'dw_khwc')
                    mixed.set_pending(w_master, None)
                    ws[1]()
                    if bs is not None:
                        bs[1]()
                if last:
                    notify()
                else:
                    def finalize(ws=ws, dwf=dwf, dev=x.device, notify=notify, direct=direct):     # a use whose backward never came
                        if not direct:
                            with mixed.on_side(dev):
                                ws[0].add_(dwf.permute(0, 3, 1, 2))
                        notify()
                    mixed.set_pending(w_master, finalize)
            else:
                call("wgrad_conv3x3_nhwc_bf16", _p(dy), _p(x), _p(dwf), _p(dbf), N, H, W, Cin, Cout, _s())   # implicit im2col
                if need_w:
                    dw = dwf.permute(0, 3, 1, 2).to(weight.dtype)
            if need_b and bs is None:
                db = dbf
        elif ctx.counted:
            mixed.use_end(w_master)
        return dx, dw, db, None, None, None


def conv3x3(x, weight, bias=None, relu=False, dtype=None, x_is_relu=False):
    """3x3, padding 1, stride 1 conv of a logically-NCHW bf16 tensor (channels-last memory).  ``weight`` / ``bias``
    are the fp32 master parameters (the compute-dtype weight is resolved through mixed.weight; gradients may be
    accumulated straight into the reducer's buckets).  ``x_is_relu``: the caller guarantees that ``x`` is the output of a
    ReLU (x == 0 wherever the ReLU was inactive); the data gradient is then produced already multiplied by [x > 0], i.e.
    as the gradient at that ReLU's input, and a conv3x3(relu=True) below recognises it and skips its own masking pass."""
    dtype = _H() if dtype is None else dtype
    from .. import mixed
    return _Conv3x3.apply(x, mixed.weight(weight, dtype), bias, relu, weight, x_is_relu)


# --------------------------------------------------------------------------------------
# Linear layer with the weight gradient on the split-T MFMA kernel
# --------------------------------------------------------------------------------------
_GEMM_WS = {}
_MIN_T = int(os.environ.get("SWIN_LINEAR_MIN_T", "1024"))   # rows below which nn.Linear stays on the framework path
_DIRECT_GEMM = os.environ.get("SWIN_TORCH_GEMM") != "1"      # A/B switch (development): 1 = torch.nn.functional.linear / mm


_TS_K = (96, 128, 192, 256, 384)


def gemm_bf16(a, b, bias=None, b_is_kn=False, out_shape=None, relu=False):
    """c (M,N) = a (M,K) @ (b.T if b is (N,K) else b) [+ bias] in bf16 with fp32 accumulation, straight on hipBLASLt
    through the C ABI (swin_gemm_bf16: cached plans, no framework dispatch).  2-D contiguous bf16 GPU tensors."""
    M, K = a.shape
    N = b.shape[1] if b_is_kn else b.shape[0]
    dev = a.device
    # one workspace per STREAM: the library's split-K algorithms write it, and the box head's GEMMs run on the sub-graph stream next
    # to the main stream's (detector._roi_stage_train_packed) -- a shared buffer was a race (test_graph_replay_equals_eager_steps)
    stream = _s()
    key = (dev, stream.value)
    ws = _GEMM_WS.get(key)
    if ws is None:
        ws = _GEMM_WS[key] = torch.empty(_lib.lib().swin_gemm_workspace_bytes(), dtype=torch.uint8, device=dev)
    c = torch.empty((M, N) if out_shape is None else out_shape, dtype=_H(), device=dev)   # not a view
    if not b_is_kn and K in _TS_K and M >= 4096 and N % 32 == 0:
        # narrow contraction, long token axis (qkv / proj of stages 1-2, the FPN laterals of those stages, the mask head's
        # deconvolution as a GEMM): HBM-bound -- the token-stationary kernel (csrc/ts_linear.hip) instead of the library
        rc = _lib.lib().swin_ts_linear_bf16(_p(a), _p(b), _p(bias), _p(c), M, N, K, 1 if relu else 0, stream)
        if rc == 0:
            return c
        if rc != 2:                                            # 2 = SWIN_ERR_UNSUPPORTED: this N has no chunking -> library
            raise SwinHipError(f"swin_ts_linear_bf16 failed with status {rc}")
    call("swin_gemm_bf16", _p(a), _p(b), _p(bias), _p(c), M, N, K, 1 if b_is_kn else 0, _p(ws), stream)
    return c.relu_() if relu else c


class _LinearBf16(torch.autograd.Function):
    """y = x w^T + b.  Forward / data gradient: library GEMM.  Weight AND bias gradient: the split-T kernel
    (wgrad_gemm.hip), accumulating in fp32 -- straight into the parameters' all-reduce buckets when a reducer
    has registered gradient sinks (mixed.grad_sink), else into fresh buffers returned through autograd.
    ``b`` may be a constant bf16 copy of ``b_master`` (mixed.const): the bias gradient then goes to ``b_master``."""

    @staticmethod
    def forward(ctx, x, w, b, w_master, b_master, relu=False):
        from .. import mixed
        ctx.relu = bool(relu)
        ctx.has_bias = b is not None
        ctx.masters = (w_master, b_master)
        ctx.bias_to_master = b is not None and not ctx.needs_input_grad[2] and b_master is not None and b_master.requires_grad
        ctx.counted = False
        ws = mixed.grad_sink(w_master)
        if ws is not None and ws[0].numel() == w.numel() and ws[0].is_contiguous() and ctx.needs_input_grad[1]:
            mixed.use_begin(w_master)
            ctx.counted = True
        x2 = x.reshape(-1, w.shape[1])
        if _DIRECT_GEMM and x2.is_contiguous() and w.is_contiguous() and (b is None or (b.dtype == _H() and b.is_contiguous())):
            y = gemm_bf16(x2, w, b, out_shape=tuple(x.shape[:-1]) + (w.shape[0],), relu=ctx.relu)      # ReLU in the kernel's epilogue where it has one
        else:
            y = torch.nn.functional.linear(x, w, b)
            if ctx.relu:
                y = y.relu_()
        if ctx.relu:
            ctx.save_for_backward(x, w, y)
        else:
            ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import mixed
        if ctx.relu:
            x, w, y = ctx.saved_tensors
            dy = torch.ops.aten.threshold_backward(dy, y, 0)        # the gradient at the ReLU's input
        else:
            x, w = ctx.saved_tensors
        w_master, b_master = ctx.masters
        N1, N2 = w.shape
        dy2 = dy.reshape(-1, N1)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        x2 = x.reshape(-1, N2)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        dx = dw = db = dbm = None
        if ctx.needs_input_grad[0]:
            dx = gemm_bf16(dy2, w, None, b_is_kn=True, out_shape=tuple(x.shape)) if (_DIRECT_GEMM and w.is_contiguous()) else (dy2 @ w).view(x.shape)
        need_w = ctx.needs_input_grad[1]
        need_b = ctx.has_bias and (ctx.needs_input_grad[2] or ctx.bias_to_master)
        if need_w or need_b:
            ws = mixed.grad_sink(w_master) if ctx.counted else None
            bs = mixed.grad_sink(b_master) if (need_b and ctx.counted) else None
            dwf = ws[0].view(N1, N2) if ws is not None else torch.zeros(N1, N2, device=x.device, dtype=torch.float32)
            dbf = None
            if need_b:
                dbf = bs[0] if bs is not None else torch.zeros(N1, device=x.device, dtype=torch.float32)
            sinks_only = ws is not None and (bs is not None or not need_b)      # nothing of it is returned through autograd
            if sinks_only and mixed.wgrad_group_active():
                mixed.wgrad_record(dy2, x2, dwf, dbf)                    # launched with the next group (mixed.wgrad_flush)
            else:
                sst = mixed.fork_to_side(x2.device, dy2, x2) if sinks_only else None
                call("wgrad_linear_bf16", _p(dy2), _p(x2), _p(dwf), _p(dbf), dy2.shape[0], N1, N2,
                     ctypes.c_void_p(sst) if sst is not None else _s())
            if ws is not None:
                def finalize(ws=ws, bs=bs, w_master=w_master):
                    mixed.set_pending(w_master, None)
                    ws[1]()
                    if bs is not None:
                        bs[1]()
                if mixed.use_end(w_master) == 0:        # the last backward of a layer used several times
                    finalize()
                else:
                    mixed.set_pending(w_master, finalize)
            elif need_w:
                dw = dwf.to(w.dtype).view(w.shape)
            if need_b and bs is None:
                if ctx.bias_to_master:
                    dbm = dbf
                else:
                    db = dbf.to(dy.dtype)
        elif ctx.counted:
            mixed.use_end(w_master)
        return dx, dw, db, None, dbm, None


def linear(x, weight, bias=None, dtype=None, relu=False):
    """nn.Linear on the compute-dtype copies of fp32 master parameters ``weight`` / ``bias`` (resolved through
    mixed.weight).  bf16 GPU tensors with 8-aligned widths and a long token axis use the hand-written
    weight/bias-gradient kernel; anything else is the plain library path."""
    from .. import mixed
    dtype = dtype or x.dtype
    w = mixed.weight(weight, dtype)
    if w.dim() == 4 and w.is_contiguous():       # a conv weight used as a GEMM over (Cin, ky, kx)-ordered rows: 1x1 convs, patch embedding
        w = w.view(w.shape[0], -1)
    if (x.dtype == _H() and x.is_cuda and w.dtype == _H() and w.dim() == 2 and w.shape[0] % 8 == 0
            and w.shape[1] % 8 == 0 and x.numel() // w.shape[1] >= _MIN_T):
        b = mixed.const(bias, dtype)                                # bf16 constant; gradient delivered to the master
        if b is None:
            b = mixed.weight(bias, dtype)
        return _LinearBf16.apply(x, w, b, weight, bias, relu)
    y = torch.nn.functional.linear(x, w, mixed.weight(bias, dtype))
    return torch.relu_(y) if relu else y


# --------------------------------------------------------------------------------------
# Token-stationary fused MLP (csrc/ts_mlp.hip): fc1 -> GELU -> fc2 in one launch, C in {96, 192}
# --------------------------------------------------------------------------------------
MLP_FUSED_C = (96, 192)


def mlp_fused_ok(x, C):
    return x.is_cuda and x.dtype == _H() and C in MLP_FUSED_C


def mlp_fwd_raw(x2, w1, b1, w2, b2):
    """y (T,C) = fc2(gelu(fc1(x))) on bf16 (T,C) rows; w1 (4C,C) / w2 (C,4C) bf16, b1 / b2 fp32.  No autograd."""
    T, C = x2.shape
    y = torch.empty_like(x2)
    call("swin_mlp_fwd_bf16", _p(x2), _p(w1), _p(_f32(b1)), _p(w2), _p(_f32(b2)), _p(y), T, C, _s())
    return y


def mlp_bwd_raw(x2, dy2, w1, b1, w2):
    """-> (dx (T,C), h (T,4C), dhpre (T,4C)) bf16: data gradient of the fused MLP plus the two operands the weight-gradient
    GEMMs need (dW2 = dy^T h, dW1 = dhpre^T x, db1 = colsum dhpre)."""
    T, C = x2.shape
    dx = torch.empty_like(x2)
    h = torch.empty((T, 4 * C), device=x2.device, dtype=_H())
    dhpre = torch.empty_like(h)
    call("swin_mlp_bwd_bf16", _p(x2), _p(dy2), _p(w1), _p(_f32(b1)), _p(w2), _p(dx), _p(h), _p(dhpre), T, C, _s())
    return dx, h, dhpre
