"""One autograd node per Swin block.

``SwinTransformerBlock.forward`` (swin_transformer.py:204-255) is thirteen kernels here (4 GEMMs, window attention,
two fused residual+LayerNorm passes, bias+GELU, and in backward their gradients plus 4 weight-gradient GEMMs).  As
separate autograd Functions each of them pays the Function.apply / graph-node / engine-dispatch overhead (~15 us
forward, ~20 us backward on this stack): ~0.25 ms of pure host time per block, 3 ms per step, on a step whose GPU
time is ~17 ms.  This module runs the same kernels, in the same order, inside ONE Function: the arithmetic is
identical to ``backbone._block`` built from the individual ops (tests compare the two)."""
import ctypes
import os

import torch

from .._lib import half_dtype as _H

from .. import _lib, mixed
from .._lib import SWIN_BF16, call
from .functional import LN_EPS, _f32, _ln_ws, _p, _s, gemm_bf16, rel_bias_expand


def _weight_grads(dy2, x2, w, w_master, b_master, need_w, need_b):
    """dW (and the bias gradient) of y = x w^T + b on the split-T kernel: into the reducer's bucket views when sinks
    are registered (returns (None, None)), else into fresh fp32 buffers -> (dw in w.dtype, db fp32 for the master)."""
    if not (need_w or need_b):
        return None, None
    N1, N2 = w.shape
    ws = mixed.grad_sink(w_master) if need_w else None
    if ws is not None and (ws[0].numel() != w.numel() or not ws[0].is_contiguous()):
        ws = None
    bs = mixed.grad_sink(b_master) if need_b else None
    dwf = ws[0].view(N1, N2) if ws is not None else torch.zeros(N1, N2, device=x2.device, dtype=torch.float32)
    dbf = None
    if need_b:
        dbf = bs[0] if bs is not None else torch.zeros(N1, device=x2.device, dtype=torch.float32)
    call("wgrad_linear_bf16", _p(dy2), _p(x2), _p(dwf), _p(dbf), dy2.shape[0], N1, N2, _s())
    dw = db = None
    if ws is not None:
        ws[1]()
    elif need_w:
        dw = dwf.to(w.dtype)
    if need_b:
        if bs is not None:
            bs[1]()
        else:
            db = dbf
    return dw, db


def _param_buf(p):
    """(fp32 accumulation buffer, gradient to hand to autograd or None, notify) for a small fp32 parameter."""
    s = mixed.grad_sink(p)
    if s is not None:
        return s[0], None, s[1]
    z = torch.zeros_like(p, dtype=torch.float32)
    return z, z, (lambda: None)


def _bias16(b):
    """bf16 copy of a bias for the GEMM epilogue (the per-step constant when there is one)."""
    if b is None:
        return None
    c = mixed.const(b, _H())
    return c if c is not None else b.detach().to(_H())


# ---- native runner (csrc/block_runner.hip): one C call per block and direction ------------------------------------
_RUNNER = os.environ.get("SWIN_BLOCK_RUNNER", "1") != "0"     # 0: issue the kernels one by one from Python (A/B, debugging)
_FUSED_MLP = os.environ.get("SWIN_FUSED_MLP", "1") != "0"     # 0: fc1 / GELU / fc2 as three launches everywhere (A/B)
_FUSED_MLP_C = (96, 192)                                      # widths the token-stationary MLP kernels are built for
_SCRATCH = {}


def _scratch(dev, key, nbytes):
    """persistent scratch (contents never outlive one call on one stream)"""
    k = (dev, key)
    t = _SCRATCH.get(k)
    if t is None or t.numel() < nbytes:
        t = _SCRATCH[k] = torch.empty(max(int(nbytes), 16), device=dev, dtype=torch.uint8)
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


def _carve(flat, sizes):
    out, off = [], 0
    for n in sizes:
        out.append(flat[off:off + n])
        off += n
    return out


def _acc_weight(w, w_master, need):
    """fp32 accumulator for a GEMM weight gradient -> (buffer | None, finish() -> gradient for autograd | None)."""
    if not need:
        return None, (lambda: None)
    sk = mixed.grad_sink(w_master)
    if sk is not None and sk[0].numel() == w.numel() and sk[0].is_contiguous():
        return sk[0], (lambda: (sk[1](), None)[1])
    z = torch.zeros(w.shape, device=w.device, dtype=torch.float32)
    return z, (lambda: z.to(w.dtype))


def _acc_param(p, need=True):
    """fp32 accumulator for a small fp32 parameter -> (buffer | None, finish() -> gradient | None)."""
    if p is None or not need:
        return None, (lambda: None)
    sk = mixed.grad_sink(p)
    if sk is not None:
        return sk[0], (lambda: (sk[1](), None)[1])
    z = torch.zeros_like(p, dtype=torch.float32)
    return z, (lambda: z)


class _SwinBlockFn(torch.autograd.Function):
    """(x, n1) -> (x_out, n_next).  Tensor inputs, in order:
    x, n1, dp0, dp1, wqkv, bqkv, table, wproj, bproj, n2w, n2b, w1, b1, w2, b2, nnw, nnb
    (w* = compute-dtype weight leaves, b* / norm parameters = fp32 masters; nnw/nnb = the NEXT norm or None)."""

    @staticmethod
    def forward(ctx, geom, masters, x, n1, dp0, dp1, wqkv, bqkv, table, wproj, bproj, n2w, n2b, w1, b1, w2, b2, nnw, nnb):
        B, H, W, nH, shift = geom
        C = x.shape[-1]
        L = H * W
        T = B * L
        dev = x.device
        x = x.contiguous()
        n1 = n1.contiguous()
        if _RUNNER:
            return _SwinBlockFn._forward_native(ctx, geom, masters, x, n1, dp0, dp1, wqkv, bqkv, table, wproj, bproj, n2w, n2b, w1,
                                                b1, w2, b2, nnw, nnb)
        # ---- attention branch (:211-247, :129-151) ----
        qkv = gemm_bf16(n1.view(T, C), wqkv, _bias16(bqkv), out_shape=(B, L, 3 * C))
        bias_exp = rel_bias_expand(table.detach())
        o = torch.empty(B, L, C, device=dev, dtype=_H())
        nW = ((H + 6) // 7) * ((W + 6) // 7)
        lse = torch.empty(B * nW * nH, 64, device=dev, dtype=torch.float32)
        scale = float((C // nH) ** -0.5)
        qkv_bias = bqkv.detach() if bqkv is not None else torch.zeros(3 * C, device=dev)
        call("swin_window_attn_fwd", _p(qkv), _p(qkv_bias), _p(bias_exp), _p(o), _p(lse), B, H, W, C, nH, shift, scale,
             SWIN_BF16, _s())
        y = gemm_bf16(o.view(T, C), wproj, _bias16(bproj), out_shape=(B, L, C))
        # ---- residual + DropPath + norm2 (:252-253) ----
        x1 = torch.empty_like(x)
        n2 = torch.empty_like(x)
        mean2 = torch.empty(T, device=dev, dtype=torch.float32)
        rstd2 = torch.empty_like(mean2)
        call("swin_add_layernorm_fwd", _p(x), _p(y), _p(dp0), L, _p(_f32(n2w)), _p(_f32(n2b)), _p(x1), _p(n2), _p(mean2),
             _p(rstd2), T, C, LN_EPS, SWIN_BF16, _s())
        # ---- MLP (:32-38) ----
        hpre = gemm_bf16(n2.view(T, C), w1, None, out_shape=(B, L, 4 * C))
        h = torch.empty_like(hpre)
        call("swin_bias_gelu_fwd", _p(hpre), _p(b1), _p(h), T, 4 * C, SWIN_BF16, _s())
        y2 = gemm_bf16(h.view(T, 4 * C), w2, _bias16(b2), out_shape=(B, L, C))
        # ---- second residual (+ the next block's norm1 / the stage's output norm) ----
        x2 = torch.empty_like(x)
        has_next = nnw is not None
        if has_next:
            nn_ = torch.empty_like(x)
            mean3 = torch.empty(T, device=dev, dtype=torch.float32)
            rstd3 = torch.empty_like(mean3)
            call("swin_add_layernorm_fwd", _p(x1), _p(y2), _p(dp1), L, _p(_f32(nnw)), _p(_f32(nnb)), _p(x2), _p(nn_), _p(mean3),
                 _p(rstd3), T, C, LN_EPS, SWIN_BF16, _s())
        else:
            nn_ = mean3 = rstd3 = None
            call("swin_add_layernorm_fwd", _p(x1), _p(y2), _p(dp1), L, None, None, _p(x2), None, None, None, T, C, LN_EPS,
                 SWIN_BF16, _s())
        ctx.save_for_backward(n1, qkv, bias_exp, lse, o, x1, mean2, rstd2, n2, hpre, h, x2, mean3, rstd3, dp0, dp1, wqkv, wproj, w1,
                              w2, n2w, nnw, b1, qkv_bias)
        ctx.geom = (B, H, W, C, nH, shift, scale)
        ctx.masters = masters
        ctx.params = (bqkv, table, bproj, n2b, b2, nnb)
        ctx.has_next = has_next
        return x2, nn_

    @staticmethod
    def backward(ctx, dx2, dnn):
        (n1, qkv, bias_exp, lse, o, x1, mean2, rstd2, n2, hpre, h, x2, mean3, rstd3, dp0, dp1, wqkv, wproj, w1, w2, n2w, nnw, b1,
         qkv_bias) = ctx.saved_tensors
        B, H, W, C, nH, shift, scale = ctx.geom
        m_wqkv, m_wproj, m_w1, m_w2 = ctx.masters
        bqkv, table, bproj, n2b, b2, nnb = ctx.params
        L = H * W
        T = B * L
        dev = x1.device
        if _RUNNER:
            return _SwinBlockFn._backward_native(ctx, dx2, dnn)
        # needs_input_grad indices: 0 geom, 1 masters, 2 x, 3 n1, 4 dp0, 5 dp1, 6 wqkv, 7 bqkv, 8 table, 9 wproj, 10 bproj,
        # 11 n2w, 12 n2b, 13 w1, 14 b1, 15 w2, 16 b2, 17 nnw, 18 nnb
        nig = ctx.needs_input_grad
        g = dict()
        # ---- second residual (+ next norm) backward: dx1 (residual stream) and dy2 (scaled by DropPath) ----
        if ctx.has_next:
            dnn = torch.zeros_like(x2) if dnn is None else dnn.contiguous()
            dres = None if dx2 is None else dx2.contiguous()
            dx1 = torch.empty_like(x2)
            dy2 = torch.empty_like(x2) if dp1 is not None else None
            wb, g['nnw'], n_w = _param_buf(nnw)
            bb, g['nnb'], n_b = _param_buf(nnb)
            call("swin_layernorm_bwd", _p(dnn), _p(x2), _p(nnw), _p(mean3), _p(rstd3), _p(dres), _p(dx1), _p(dy2), _p(dp1), L,
                 _p(wb), _p(bb), T, C, SWIN_BF16, _p(_ln_ws(T, C, x2)), _s())
            n_w(); n_b()
            if dy2 is None:
                dy2 = dx1
        else:
            dx1 = dx2.contiguous()
            if dp1 is None:
                dy2 = dx1
            else:
                dy2 = (dx1.view(B, -1) * dp1.to(dx1.dtype).view(B, 1)).view_as(dx1)
        # ---- fc2 ----
        dy2_2 = dy2.view(T, C)
        dh = gemm_bf16(dy2_2, w2, None, b_is_kn=True, out_shape=(B, L, 4 * C))
        g['w2'], g['b2'] = _weight_grads(dy2_2, h.view(T, 4 * C), w2, m_w2, b2, nig[15], b2 is not None and b2.requires_grad)
        # ---- GELU ----
        dhpre = torch.empty_like(hpre)
        b1b, g['b1'], n_b1 = _param_buf(b1)
        call("swin_bias_gelu_bwd", _p(dh), _p(hpre), _p(b1), _p(dhpre), _p(b1b), T, 4 * C, SWIN_BF16, _s())
        n_b1()
        # ---- fc1 ----
        dhp2 = dhpre.view(T, 4 * C)
        dn2 = gemm_bf16(dhp2, w1, None, b_is_kn=True, out_shape=(B, L, C))
        g['w1'], _ = _weight_grads(dhp2, n2.view(T, C), w1, m_w1, None, nig[13], False)
        # ---- first residual + norm2 backward ----
        dx = torch.empty_like(x1)
        dy = torch.empty_like(x1) if dp0 is not None else None
        wb, g['n2w'], n_w = _param_buf(n2w)
        bb, g['n2b'], n_b = _param_buf(n2b)
        call("swin_layernorm_bwd", _p(dn2), _p(x1), _p(n2w), _p(mean2), _p(rstd2), _p(dx1), _p(dx), _p(dy), _p(dp0), L,
             _p(wb), _p(bb), T, C, SWIN_BF16, _p(_ln_ws(T, C, x1)), _s())
        n_w(); n_b()
        if dy is None:
            dy = dx
        # ---- proj ----
        dy_2 = dy.view(T, C)
        do = gemm_bf16(dy_2, wproj, None, b_is_kn=True, out_shape=(B, L, C))
        g['wproj'], g['bproj'] = _weight_grads(dy_2, o.view(T, C), wproj, m_wproj, bproj, nig[9],
                                               bproj is not None and bproj.requires_grad)
        # ---- window attention ----
        dqkv = torch.empty_like(qkv)
        dbexp = torch.zeros_like(bias_exp)
        padded = (H % 7 != 0) or (W % 7 != 0)
        if padded and bqkv is not None:
            dpb, g['bqkv_pad'], n_pad = _param_buf(bqkv)
        else:
            dpb, g['bqkv_pad'], n_pad = None, None, (lambda: None)
        ws_bytes = _lib.lib().swin_window_attn_bwd_workspace_bytes(B, H, W, nH, SWIN_BF16)
        ws = torch.empty(max(ws_bytes, 16), device=dev, dtype=torch.uint8)
        call("swin_window_attn_bwd", _p(qkv), _p(qkv_bias), _p(bias_exp), _p(lse), _p(do), _p(dqkv), _p(dbexp), _p(dpb), _p(ws),
             B, H, W, C, nH, shift, scale, SWIN_BF16, _s())
        tb, g['table'], n_t = _param_buf(table)
        call("swin_rel_bias_reduce", _p(dbexp), _p(tb), nH, _s())
        n_t()
        # ---- qkv ----
        dqkv2 = dqkv.view(T, 3 * C)
        dn1 = gemm_bf16(dqkv2, wqkv, None, b_is_kn=True, out_shape=(B, L, C))
        g['wqkv'], db_qkv = _weight_grads(dqkv2, n1.view(T, C), wqkv, m_wqkv, bqkv, nig[6], bqkv is not None and bqkv.requires_grad)
        n_pad()
        # qkv.bias: GEMM-bias gradient + the gradient through the padded tokens (they ARE the bias)
        gb = db_qkv
        if g['bqkv_pad'] is not None:
            gb = g['bqkv_pad'] if gb is None else gb + g['bqkv_pad']
        return (None, None, dx, dn1, None, None, g['wqkv'], gb, g['table'], g['wproj'], g['bproj'], g['n2w'], g['n2b'], g['w1'],
                g['b1'], g['w2'], g['b2'], g.get('nnw'), g.get('nnb'))


    # ---------------------------------------------------------------------------------------------- native runner
    @staticmethod
    def _forward_native(ctx, geom, masters, x, n1, dp0, dp1, wqkv, bqkv, table, wproj, bproj, n2w, n2b, w1, b1, w2, b2, nnw, nnb):
        B, H, W, nH, shift = geom
        C = x.shape[-1]
        L = H * W
        T = B * L
        dev = x.device
        has_next = nnw is not None
        scale = float((C // nH) ** -0.5)
        TC = T * C
        fused = _FUSED_MLP and C in _FUSED_MLP_C       # fc1 -> GELU -> fc2 in one launch; hpre / h are recomputed in backward
        if fused:
            flat = torch.empty(8 * TC, device=dev, dtype=_H())
            qkv, o, y, x1, n2, y2 = _carve(flat, [3 * TC, TC, TC, TC, TC, TC])
            hpre = h = None
        else:
            flat = torch.empty(16 * TC, device=dev, dtype=_H())
            qkv, o, y, x1, n2, hpre, h, y2 = _carve(flat, [3 * TC, TC, TC, TC, TC, 4 * TC, 4 * TC, TC])
        nW = ((H + 6) // 7) * ((W + 6) // 7)
        nl = B * nW * nH * 64
        # the expanded bias table: this step's persistent copy (rebuilt for all blocks in one launch after the optimizer step) when
        # the block's weights are shadowed, else expanded here by the runner
        bias_exp = mixed.rel_bias_expanded(table, masters[0])
        step_exp = bias_exp is not None
        f32 = torch.empty(nl + 4 * T + (0 if step_exp else nH * 4096), device=dev, dtype=torch.float32)
        if step_exp:
            lse, mean2, rstd2, mean3, rstd3 = _carve(f32, [nl, T, T, T, T])
            bias_exp = bias_exp.view(-1)
        else:
            lse, mean2, rstd2, mean3, rstd3, bias_exp = _carve(f32, [nl, T, T, T, T, nH * 4096])
        x2 = torch.empty_like(x)
        nn_ = torch.empty_like(x) if has_next else None
        qkv_bias = bqkv.detach() if bqkv is not None else torch.zeros(3 * C, device=dev)
        gws = _scratch(dev, 'gemm', _lib.lib().swin_gemm_workspace_bytes())
        tab = None if step_exp else _f32(table.detach())
        f_n2w, f_n2b = _f32(n2w), _f32(n2b)
        f_nnw, f_nnb = (_f32(nnw), _f32(nnb)) if has_next else (None, None)
        bq16, bp16, b216 = _bias16(bqkv), _bias16(bproj), _bias16(b2)
        f_b2 = _f32(b2.detach()) if (fused and b2 is not None) else (torch.zeros(C, device=dev) if fused else None)
        ptrs = (ctypes.c_void_p * 36)(
            _ptr(x), _ptr(n1), _ptr(dp0), _ptr(dp1), _ptr(wqkv), _ptr(bq16), _ptr(qkv_bias), _ptr(tab), _ptr(bias_exp), _ptr(wproj),
            _ptr(bp16), _ptr(f_n2w), _ptr(f_n2b), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b216), _ptr(f_nnw), _ptr(f_nnb), _ptr(qkv),
            _ptr(lse), _ptr(o), _ptr(y), _ptr(x1), _ptr(n2), _ptr(mean2), _ptr(rstd2), _ptr(hpre), _ptr(h), _ptr(y2), _ptr(x2),
            _ptr(nn_), _ptr(mean3) if has_next else None, _ptr(rstd3) if has_next else None, _ptr(gws), _ptr(f_b2))
        iv = (ctypes.c_int64 * 7)(B, H, W, C, nH, shift, 1 if fused else 0)
        fv = (ctypes.c_float * 2)(scale, LN_EPS)
        call("swin_block_fwd", ptrs, iv, fv, _s())
        shp = (B, L, C)
        ctx.fused_mlp = fused
        ctx.save_for_backward(n1, qkv.view(B, L, 3 * C), bias_exp.view(nH, 64, 64), lse, o.view(shp), x1.view(shp), mean2, rstd2,
                              n2.view(shp), None if fused else hpre.view(B, L, 4 * C), None if fused else h.view(B, L, 4 * C), x2,
                              mean3 if has_next else None,
                              rstd3 if has_next else None, dp0, dp1, wqkv, wproj, w1, w2, n2w, nnw, b1, qkv_bias)
        ctx.geom = (B, H, W, C, nH, shift, scale)
        ctx.masters = masters
        ctx.params = (bqkv, table, bproj, n2b, b2, nnb)
        ctx.has_next = has_next
        return x2, nn_

    @staticmethod
    def _backward_native(ctx, dx2, dnn):
        (n1, qkv, bias_exp, lse, o, x1, mean2, rstd2, n2, hpre, h, x2, mean3, rstd3, dp0, dp1, wqkv, wproj, w1, w2, n2w, nnw, b1,
         qkv_bias) = ctx.saved_tensors
        B, H, W, C, nH, shift, scale = ctx.geom
        m_wqkv, m_wproj, m_w1, m_w2 = ctx.masters
        bqkv, table, bproj, n2b, b2, nnb = ctx.params
        L = H * W
        T = B * L
        TC = T * C
        dev = x1.device
        nig = ctx.needs_input_grad
        has_next = ctx.has_next
        flat = torch.empty(16 * TC, device=dev, dtype=_H())
        dx1, dy2, dh, dhpre, dn2, dy, do, dqkv = _carve(flat, [TC, TC, 4 * TC, 4 * TC, TC, TC, TC, 3 * TC])
        if has_next:
            dnn = torch.zeros_like(x2) if dnn is None else dnn.contiguous()
            dx2 = None if dx2 is None else dx2.contiguous()
        else:
            dx2 = dx2.contiguous()
            dy2 = dx2 if dp1 is None else (dx2.view(B, -1) * dp1.to(dx2.dtype).view(B, 1)).view_as(dx2)
        dx = torch.empty_like(x1)
        dn1 = torch.empty_like(x1)
        dbexp = None          # entry 36: no longer used by the runner (the bias slabs go straight into the table, csrc/tail_reduce.hip)
        padded = (H % 7 != 0) or (W % 7 != 0)
        a_wqkv, f_wqkv = _acc_weight(wqkv, m_wqkv, nig[6])
        a_wproj, f_wproj = _acc_weight(wproj, m_wproj, nig[9])
        a_w1, f_w1 = _acc_weight(w1, m_w1, nig[13])
        a_w2, f_w2 = _acc_weight(w2, m_w2, nig[15])
        need_bq = bqkv is not None and bqkv.requires_grad
        a_bqkv, f_bqkv = _acc_param(bqkv, need_bq and a_wqkv is not None)
        a_bpad = a_bqkv if (padded and need_bq and a_bqkv is not None) else None
        f_bpad = (lambda: None)
        if padded and need_bq and a_bqkv is None:
            a_bpad, f_bpad = _acc_param(bqkv)
        a_bproj, f_bproj = _acc_param(bproj, bproj is not None and bproj.requires_grad and a_wproj is not None)
        a_b2, f_b2 = _acc_param(b2, b2 is not None and b2.requires_grad and a_w2 is not None)
        a_b1, f_b1 = _acc_param(b1)
        a_n2w, f_n2w = _acc_param(n2w)
        a_n2b, f_n2b = _acc_param(n2b)
        a_nnw, f_nnw = _acc_param(nnw) if has_next else (None, (lambda: None))
        a_nnb, f_nnb = _acc_param(nnb) if has_next else (None, (lambda: None))
        a_tab, f_tab = _acc_param(table)
        lib = _lib.lib()
        ws_attn = _scratch(dev, 'attn_bwd', lib.swin_window_attn_bwd_workspace_bytes(B, H, W, nH, SWIN_BF16))
        ln_bytes = lib.swin_layernorm_bwd_workspace_bytes(T, C, SWIN_BF16)
        if getattr(ctx, 'fused_mlp', False):        # norm2's backward runs in the fused MLP's epilogue: one partial row per thread block
            ln_bytes = max(ln_bytes, lib.swin_mlp_ln_bwd_partial_rows(T, C) * 2 * C * 4)
        ws_ln2 = _scratch(dev, 'ln2', ln_bytes)
        ws_ln3 = _scratch(dev, 'ln3', ln_bytes)
        gws = _scratch(dev, 'gemm', lib.swin_gemm_workspace_bytes())
        # weight-gradient stream: only when every weight / bias accumulator of the four GEMMs is a reducer sink (a fresh
        # buffer returned through autograd would be read on the main stream)
        side = mixed.side_stream(dev)
        all_sinks = True
        for q_ in (m_wqkv, m_wproj, m_w1, m_w2, bqkv, bproj, b1, b2, n2w, n2b, table) + ((nnw, nnb) if has_next else ()):
            if q_ is not None and q_.requires_grad and mixed.grad_sink(q_) is None:
                all_sinks = False
                break
        if not all_sinks:
            side = None
        # the four weight-gradient GEMMs are RECORDED for the next grouped launch (mixed.wgrad_flush) instead of launched here
        record = all_sinks and mixed.wgrad_group_active() and all(a_ is not None for a_ in (a_wqkv, a_wproj, a_w1, a_w2))
        if side is not None:
            # the reductions behind attention / LayerNorm backward run on the side stream too (csrc/abi.hip): their workspaces
            # must not be the shared scratch (the next block's kernels would overwrite it under them)
            ws_attn = torch.empty(max(lib.swin_window_attn_bwd_workspace_bytes(B, H, W, nH, SWIN_BF16), 16), device=dev, dtype=torch.uint8)
            ws_ln2 = torch.empty(max(ln_bytes, 16), device=dev, dtype=torch.uint8)
            ws_ln3 = torch.empty(max(ln_bytes, 16), device=dev, dtype=torch.uint8) if has_next else ws_ln2
            # dy2 is a tensor of its own (not a slice of `flat`) when the block has no next norm and DropPath scales dx2; the fc2
            # weight gradient reads it on the side stream after backward() has returned
            mixed.side_keep(flat, qkv, n1, dx2, dy2, dnn if has_next else None, ws_attn, ws_ln2, ws_ln3, dbexp, mean2, lse)
            mixed.side_mark(dev)
        w2t = None if getattr(ctx, 'fused_mlp', False) or C % 64 != 0 else mixed.linear_t_weight(m_w2, w2)
        # narrow stages: the proj data gradient on the token-stationary kernel (csrc/ts_linear.hip) needs the transposed weight
        wpt = mixed.linear_t_weight(m_wproj, wproj) if C in (96, 128, 192, 256, 384) else None
        ptrs = (ctypes.c_void_p * 58)(
            _ptr(n1), _ptr(qkv), _ptr(bias_exp), _ptr(lse), _ptr(o), _ptr(x1), _ptr(mean2), _ptr(rstd2), _ptr(n2), _ptr(hpre),
            _ptr(h), _ptr(x2), _ptr(mean3), _ptr(rstd3), _ptr(dp0), _ptr(dp1), _ptr(wqkv), _ptr(wproj), _ptr(w1), _ptr(w2),
            _ptr(n2w), _ptr(nnw) if has_next else None, _ptr(b1), _ptr(qkv_bias), _ptr(dx2), _ptr(dnn) if has_next else None, _ptr(dx),
            _ptr(dn1), _ptr(dx1), _ptr(dy2), _ptr(dh), _ptr(dhpre), _ptr(dn2), _ptr(dy), _ptr(do), _ptr(dqkv), _ptr(dbexp),
            _ptr(a_wqkv), _ptr(a_bqkv), _ptr(a_bpad), _ptr(a_wproj), _ptr(a_bproj), _ptr(a_w1), _ptr(a_b1), _ptr(a_w2), _ptr(a_b2),
            _ptr(a_n2w), _ptr(a_n2b), _ptr(a_nnw), _ptr(a_nnb), _ptr(a_tab), _ptr(ws_attn), _ptr(ws_ln2), _ptr(ws_ln3), _ptr(gws),
            side.cuda_stream if side is not None else None, _ptr(w2t), _ptr(wpt))
        iv = (ctypes.c_int64 * 8)(B, H, W, C, nH, shift, 1 if getattr(ctx, 'fused_mlp', False) else 0, 1 if record else 0)
        fv = (ctypes.c_float * 1)(scale)
        call("swin_block_bwd", ptrs, iv, fv, _s())
        if record:
            t128 = lambda a_, b_: ((a_ + 127) // 128) * ((b_ + 127) // 128)      # noqa: E731
            mixed.wgrad_note(dev, 4, t128(3 * C, C) + t128(C, C) + 2 * t128(4 * C, C), flat, n1, o, n2, h, dx2, dy2,
                             tail=(2 if (C in (192, 256) and shift == 0) else 1) if C <= 256 else 0)
        g_bq = f_bqkv()
        g_pad = f_bpad()
        if g_pad is not None:
            g_bq = g_pad if g_bq is None else g_bq + g_pad
        return (None, None, dx, dn1, None, None, f_wqkv(), g_bq, f_tab(), f_wproj(), f_bproj(), f_n2w(), f_n2b(), f_w1(), f_b1(),
                f_w2(), f_b2(), f_nnw(), f_nnb())


def swin_block(x, n1, dp, geom, blk, next_norm, dtype):
    """Fused forward of one SwinTransformerBlock on bf16 GPU tensors: (x, norm1(x)) -> (x_out, next_norm(x_out) | None)."""
    a = blk.attn
    mw = (a.qkv.weight, a.proj.weight, blk.mlp.fc1.weight, blk.mlp.fc2.weight)
    ws = [mixed.weight(p, dtype) for p in mw]
    nnw, nnb = (next_norm.weight, next_norm.bias) if next_norm is not None else (None, None)
    return _SwinBlockFn.apply(geom, mw, x, n1, dp[0], dp[1], ws[0], a.qkv.bias, a.relative_position_bias_table, ws[1], a.proj.bias,
                              blk.norm2.weight, blk.norm2.bias, ws[2], blk.mlp.fc1.bias, ws[3], blk.mlp.fc2.bias, nnw, nnb)
