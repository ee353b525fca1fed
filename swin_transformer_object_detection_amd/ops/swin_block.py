"""One autograd node per Swin block.

``SwinTransformerBlock.forward`` (swin_transformer.py:204-255) is thirteen kernels here (4 GEMMs, window attention,
two fused residual+LayerNorm passes, bias+GELU, and in backward their gradients plus 4 weight-gradient GEMMs).  As
separate autograd Functions each of them pays the Function.apply / graph-node / engine-dispatch overhead (~15 us
forward, ~20 us backward on this stack): ~0.25 ms of pure host time per block, 3 ms per step, on a step whose GPU
time is ~17 ms.  This module runs the same kernels, in the same order, inside ONE Function: the arithmetic is
identical to ``backbone._block`` built from the individual ops (tests compare the two)."""
import torch

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
    c = mixed.const(b, torch.bfloat16)
    return c if c is not None else b.detach().to(torch.bfloat16)


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
        # ---- attention branch (:211-247, :129-151) ----
        qkv = gemm_bf16(n1.view(T, C), wqkv, _bias16(bqkv), out_shape=(B, L, 3 * C))
        bias_exp = rel_bias_expand(table.detach())
        o = torch.empty(B, L, C, device=dev, dtype=torch.bfloat16)
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


def swin_block(x, n1, dp, geom, blk, next_norm, dtype):
    """Fused forward of one SwinTransformerBlock on bf16 GPU tensors: (x, norm1(x)) -> (x_out, next_norm(x_out) | None)."""
    a = blk.attn
    mw = (a.qkv.weight, a.proj.weight, blk.mlp.fc1.weight, blk.mlp.fc2.weight)
    ws = [mixed.weight(p, dtype) for p in mw]
    nnw, nnb = (next_norm.weight, next_norm.bias) if next_norm is not None else (None, None)
    return _SwinBlockFn.apply(geom, mw, x, n1, dp[0], dp[1], ws[0], a.qkv.bias, a.relative_position_bias_table, ws[1], a.proj.bias,
                              blk.norm2.weight, blk.norm2.bias, ws[2], blk.mlp.fc1.bias, ws[3], blk.mlp.fc2.bias, nnw, nnb)
