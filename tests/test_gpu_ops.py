"""GPU parity tests: every HIP kernel, called through the C ABI (ctypes), against the CPU oracle.

fp32 path: atol 1e-4 (the north_star gate).  bf16 path: the oracle is evaluated in fp32 on the SAME
bf16-rounded inputs; tolerance = a few bf16 ulps of the output scale (stated per test).
NMS: bit-exact index order.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import callers_oracle as CO  # noqa: E402
from oracle import det_ops_oracle as D  # noqa: E402
from oracle import swin_oracle as S  # noqa: E402

ATOL32 = 1e-4


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import swin_transformer_object_detection_amd.ops as o
    return o


def dev(t, dtype=None):
    t = t.cuda()
    return t.to(dtype) if dtype is not None else t


def close(a, b, atol, rtol=0.0, msg=""):
    a = a.detach().float().cpu().numpy()
    b = b.detach().float().cpu().numpy()
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol, err_msg=msg)


def bf16_tol(ref, ulps=4.0):
    """absolute tolerance = `ulps` bf16 ulps (2^-8 relative) of the reference's max magnitude"""
    return float(ulps * 2.0 ** -8 * max(ref.detach().abs().max().item(), 1e-3))


# ------------------------------------------------------------------------------------------
# window attention
# ------------------------------------------------------------------------------------------
def oracle_attention_natural(qkv, qkv_bias, table, B, H, W, nH, shift):
    """Reference semantics on the natural grid: pad (padded tokens are 0 before the qkv Linear, so
    their q|k|v equal qkv.bias -- swin_transformer.py:211-218), roll, partition, core, reverse,
    roll back, crop (:222-247)."""
    C3 = qkv.shape[-1]
    C = C3 // 3
    Hp, Wp = S.padded_hw(H, W)
    x = qkv.view(B, H, W, C3)
    full = qkv_bias.view(1, 1, 1, C3).expand(B, Hp, Wp, C3).clone()
    full = torch.cat([torch.cat([x, full[:, :H, W:, :]], 2), full[:, H:, :, :]], 1)
    mask = None
    if shift > 0:
        full = torch.roll(full, shifts=(-shift, -shift), dims=(1, 2))
        mask = S.shift_attn_mask(H, W, 7, shift)
    win = S.window_partition(full, 7).view(-1, 49, C3)
    o = S.window_attention_core(win, table, nH, mask)
    o = S.window_reverse(o.view(-1, 7, 7, C), 7, Hp, Wp)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return o[:, :H, :W, :].reshape(B, H * W, C)


ATTN_CASES = [
    # B, H, W, nH, shift
    (2, 14, 21, 2, 0),
    (2, 14, 21, 2, 3),
    (2, 19, 25, 1, 0),    # pads bottom and right
    (2, 19, 25, 3, 3),    # pads + shift + 3 heads
    (1, 5, 7, 4, 3),      # single window row (every window is a "last row" window)
    (3, 30, 9, 2, 3),
]


@pytest.mark.parametrize("B,H,W,nH,shift", ATTN_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_window_attention_fwd_bwd(ops, B, H, W, nH, shift, dtype):
    C = 32 * nH
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W + shift)
    qkv = torch.randn(B, H * W, 3 * C, generator=g) * 0.7
    qb = torch.randn(3 * C, generator=g) * 0.3
    table = torch.randn(169, nH, generator=g) * 0.5
    wgt = torch.randn(B, H * W, C, generator=g)
    if dtype == torch.bfloat16:
        qkv = qkv.bfloat16().float()
        wgt = wgt.bfloat16().float()
    # oracle (fp32, CPU)
    q0, b0, t0 = qkv.clone().requires_grad_(True), qb.clone().requires_grad_(True), table.clone().requires_grad_(True)
    ref = oracle_attention_natural(q0, b0, t0, B, H, W, nH, shift)
    (ref * wgt).sum().backward()
    # HIP
    q1 = dev(qkv, dtype).requires_grad_(True)
    b1 = dev(qb).requires_grad_(True)
    t1 = dev(table).requires_grad_(True)
    out = ops.window_attention(q1, b1, t1, B, H, W, nH, shift)
    assert out.shape == (B, H * W, C) and out.dtype == dtype
    (out.float() * dev(wgt)).sum().backward()
    torch.cuda.synchronize()
    padded = H % 7 != 0 or W % 7 != 0
    if not padded:        # no padded token exists: qkv.bias gets no gradient from the attention op
        assert b1.grad is None or float(b1.grad.abs().max()) == 0.0
        assert float(b0.grad.abs().max()) == 0.0
    if dtype == torch.float32:
        close(out, ref, ATOL32, msg="out")
        close(q1.grad, q0.grad, ATOL32, 1e-4, msg="dqkv")
        if padded:
            close(b1.grad, b0.grad, 2e-4, 1e-4, msg="dqkv_bias(pad)")
        close(t1.grad, t0.grad, 5e-4, 1e-4, msg="dtable")
    else:
        # bf16 storage of P (8 bits) and of the outputs: 4 ulps of the output scale; the table
        # gradient sums B*nW window contributions -> scale the tolerance with its own magnitude
        close(out, ref, bf16_tol(ref), msg="out")
        close(q1.grad, q0.grad, bf16_tol(q0.grad, 6), msg="dqkv")
        if padded:
            close(b1.grad, b0.grad, bf16_tol(b0.grad, 8) + 1e-3, msg="dqkv_bias(pad)")
        close(t1.grad, t0.grad, bf16_tol(t0.grad, 8), msg="dtable")


def test_window_attention_rejects_bad_shapes(ops):
    from swin_transformer_object_detection_amd._lib import SwinHipError
    qkv = torch.zeros(1, 49, 3 * 48, device="cuda")
    with pytest.raises(SwinHipError):      # head_dim 48 is not the Swin head_dim
        ops.window_attention(qkv, torch.zeros(144, device="cuda"), torch.zeros(169, 1, device="cuda"), 1, 7, 7, 1, 0)
    with pytest.raises(SwinHipError):      # CPU tensors: no fallback
        ops.window_attention(torch.zeros(1, 49, 96), torch.zeros(96), torch.zeros(169, 1), 1, 7, 7, 1, 0)


# ------------------------------------------------------------------------------------------
# LayerNorm family, GELU, gather kernels
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,C", [(37, 32), (200, 96), (64, 192), (33, 384), (17, 768), (9, 1536), (5, 3072)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm(ops, rows, C, dtype):
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g) * 2 + 0.5
    w = 1 + 0.2 * torch.randn(C, generator=g)
    b = 0.2 * torch.randn(C, generator=g)
    gy = torch.randn(rows, C, generator=g)
    if dtype == torch.bfloat16:
        x, gy = x.bfloat16().float(), gy.bfloat16().float()
    x0, w0, b0 = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = S.layer_norm(x0, w0, b0)
    (ref * gy).sum().backward()
    x1, w1, b1 = dev(x, dtype).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    y = ops.layer_norm(x1, w1, b1)
    (y.float() * dev(gy)).sum().backward()
    if dtype == torch.float32:
        close(y, ref, ATOL32)
        close(x1.grad, x0.grad, ATOL32, 1e-4)
        close(w1.grad, w0.grad, 1e-3, 1e-4)
        close(b1.grad, b0.grad, 1e-3, 1e-4)
    else:
        close(y, ref, bf16_tol(ref, 2))
        close(x1.grad, x0.grad, bf16_tol(x0.grad, 2))
        close(w1.grad, w0.grad, bf16_tol(w0.grad, 2) + 1e-2)
        close(b1.grad, b0.grad, bf16_tol(b0.grad, 2) + 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_scale", [False, True])
def test_add_layernorm(ops, dtype, with_scale):
    B, L, C = 3, 50, 96
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(B, L, C, generator=g), torch.randn(B, L, C, generator=g)
    w, b = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    sc = torch.tensor([0.0, 2.0, 2.0]) if with_scale else None
    g1, g2 = torch.randn(B, L, C, generator=g), torch.randn(B, L, C, generator=g)
    if dtype == torch.bfloat16:
        x, y, g1, g2 = [t.bfloat16().float() for t in (x, y, g1, g2)]
    x0, y0, w0, b0 = [t.clone().requires_grad_(True) for t in (x, y, w, b)]
    xo_ref = x0 + S.drop_path(y0, sc)
    n_ref = S.layer_norm(xo_ref, w0, b0)
    ((xo_ref * g1).sum() + (n_ref * g2).sum()).backward()
    x1, y1 = dev(x, dtype).requires_grad_(True), dev(y, dtype).requires_grad_(True)
    w1, b1 = dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    xo, n = ops.add_layer_norm(x1, y1, None if sc is None else dev(sc), L, w1, b1)
    ((xo.float() * dev(g1)).sum() + (n.float() * dev(g2)).sum()).backward()
    tol = (lambda r, u=3: ATOL32) if dtype == torch.float32 else bf16_tol
    close(xo, xo_ref, tol(xo_ref)); close(n, n_ref, tol(n_ref))
    close(x1.grad, x0.grad, tol(x0.grad) * (1 if dtype == torch.float32 else 1.5), 1e-4)
    close(y1.grad, y0.grad, tol(y0.grad) * (1 if dtype == torch.float32 else 1.5), 1e-4)
    close(w1.grad, w0.grad, 1e-3 if dtype == torch.float32 else bf16_tol(w0.grad) + 2e-2, 1e-4)
    # residual-only form
    xo2 = ops.add_scaled(dev(x, dtype), dev(y, dtype), None if sc is None else dev(sc), L)
    close(xo2, xo_ref, tol(xo_ref))


@pytest.mark.parametrize("H,W", [(6, 8), (7, 9), (5, 6)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_patch_merge_ln(ops, H, W, dtype):
    B, C = 2, 32
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(B, H * W, C, generator=g)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    p = {"d.norm.weight": 1 + 0.2 * torch.randn(4 * C, generator=g), "d.norm.bias": 0.2 * torch.randn(4 * C, generator=g),
         "d.reduction.weight": torch.eye(4 * C)}
    x0 = x.clone().requires_grad_(True)
    for v in p.values():
        v.requires_grad_(True)
    ref = S.patch_merging(x0, H, W, p, "d.")          # identity reduction -> the gathered+normed rows
    gy = torch.randn(ref.shape, generator=g)
    (ref * gy).sum().backward()
    x1 = dev(x, dtype).requires_grad_(True)
    w1, b1 = dev(p["d.norm.weight"].detach()).requires_grad_(True), dev(p["d.norm.bias"].detach()).requires_grad_(True)
    y = ops.patch_merge_layer_norm(x1, w1, b1, B, H, W)
    (y.float() * dev(gy)).sum().backward()
    if dtype == torch.float32:
        close(y, ref, ATOL32); close(x1.grad, x0.grad, ATOL32, 1e-4)
        close(w1.grad, p["d.norm.weight"].grad, 1e-3); close(b1.grad, p["d.norm.bias"].grad, 1e-3)
    else:
        close(y, ref, bf16_tol(ref, 2)); close(x1.grad, x0.grad, bf16_tol(x0.grad, 2))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bias_gelu(ops, dtype):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(77, 384, generator=g) * 2
    b = torch.randn(384, generator=g) * 0.3
    gy = torch.randn(77, 384, generator=g)
    if dtype == torch.bfloat16:
        x, gy = x.bfloat16().float(), gy.bfloat16().float()
    x0, b0 = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = S.gelu(x0 + b0)
    (ref * gy).sum().backward()
    x1, b1 = dev(x, dtype).requires_grad_(True), dev(b).requires_grad_(True)
    y = ops.bias_gelu(x1, b1)
    (y.float() * dev(gy)).sum().backward()
    if dtype == torch.float32:
        close(y, ref, 1e-5); close(x1.grad, x0.grad, 1e-5); close(b1.grad, b0.grad, 1e-4)
    else:
        close(y, ref, bf16_tol(ref, 1)); close(x1.grad, x0.grad, bf16_tol(x0.grad, 1))
        close(b1.grad, b0.grad, bf16_tol(b0.grad, 2) + 2e-2)


@pytest.mark.parametrize("Hi,Wi", [(16, 24), (18, 21)])
def test_patch_im2row(ops, Hi, Wi):
    g = torch.Generator().manual_seed(1)
    img = torch.randn(2, 3, Hi, Wi, generator=g)
    wt = torch.randn(8, 3, 4, 4, generator=g)
    p = {"patch_embed.proj.weight": wt, "patch_embed.proj.bias": torch.zeros(8)}
    ref = S.patch_embed(img, p)                                  # (B, 8, Ho, Wo)
    rows = ops.patch_im2row(dev(img), torch.float32)
    y = rows @ dev(wt).view(8, 48).t()
    Ho, Wo = ref.shape[2:]
    close(y.view(2, Ho, Wo, 8).permute(0, 3, 1, 2), ref, 1e-5)


@pytest.mark.parametrize("cl", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hf,wf,hc,wc", [(10, 14, 5, 7), (25, 40, 13, 20), (7, 9, 4, 5)])
def test_upsample_add(ops, cl, dtype, hf, wf, hc, wc):
    g = torch.Generator().manual_seed(hf)
    fine, coarse = torch.randn(2, 16, hf, wf, generator=g), torch.randn(2, 16, hc, wc, generator=g)
    gy = torch.randn(2, 16, hf, wf, generator=g)
    if dtype == torch.bfloat16:
        fine, coarse, gy = fine.bfloat16().float(), coarse.bfloat16().float(), gy.bfloat16().float()
    f0, c0 = fine.clone().requires_grad_(True), coarse.clone().requires_grad_(True)
    ref = f0 + F.interpolate(c0, size=(hf, wf), mode="nearest")    # fpn.py:188-191
    (ref * gy).sum().backward()
    mf = torch.channels_last if cl else torch.contiguous_format
    f1 = dev(fine, dtype).contiguous(memory_format=mf).requires_grad_(True)
    c1 = dev(coarse, dtype).contiguous(memory_format=mf).requires_grad_(True)
    out = ops.upsample_add(f1, c1)
    (out.float() * dev(gy)).sum().backward()
    tol = 1e-6 if dtype == torch.float32 else bf16_tol(ref, 1)
    close(out, ref, tol); close(f1.grad, f0.grad, tol)
    close(c1.grad, c0.grad, 1e-5 if dtype == torch.float32 else bf16_tol(c0.grad, 2))


# ------------------------------------------------------------------------------------------
# RoIAlign / nms
# ------------------------------------------------------------------------------------------
def _rand_rois(rng, K, N, Wimg, Himg):
    xy = rng.rand(K, 2) * np.array([Wimg, Himg]) * 1.1 - 0.05 * np.array([Wimg, Himg])
    wh = rng.rand(K, 2) * np.array([Wimg, Himg]) * 0.6
    return np.concatenate([rng.randint(0, N, (K, 1)), xy, xy + wh], 1).astype(np.float32)


@pytest.mark.parametrize("cl", [False, True])
@pytest.mark.parametrize("out_size,scale", [(7, 0.25), (14, 0.125), ((3, 5), 1.0)])
def test_roi_align_fwd_bwd(ops, cl, out_size, scale):
    rng = np.random.RandomState(7)
    N, C, H, W = 2, 8, 25, 40
    inp = rng.randn(N, C, H, W).astype(np.float32)
    rois = _rand_rois(rng, 60, N, W / scale, H / scale)
    rois[0, 1:] = [5, 5, 5, 5]                 # zero-size roi
    rois[1, 1:] = [1e4, 1e4, 1e4 + 8, 1e4 + 8]  # fully outside
    rois[2, 1:] = [-20, 10 / scale, W / scale + 30, 10.4 / scale]   # thin, wider than the map: large sampling grid on one axis
    rois[3, 1:] = [0, 0, W / scale, H / scale]                      # the whole map
    ref = D.roi_align_c(inp, rois, out_size, scale, 0, True)
    mf = torch.channels_last if cl else torch.contiguous_format
    x = torch.from_numpy(inp).cuda().contiguous(memory_format=mf).requires_grad_(True)
    out = ops.roi_align(x, torch.from_numpy(rois).cuda(), out_size, scale, 0, 'avg', True)
    assert out.shape == ref.shape and out.dtype == torch.float32
    close(out, torch.from_numpy(ref), ATOL32)
    gy = rng.randn(*ref.shape).astype(np.float32)
    (out * torch.from_numpy(gy).cuda()).sum().backward()
    gref = D.roi_align_bwd_c(gy, rois, inp.shape, scale, 0, True)
    close(x.grad, torch.from_numpy(gref).float(), 2e-4, 1e-4)
    # the module form, bf16 features (converted on load == force_fp32), empty roi set
    layer = ops.RoIAlign(out_size, scale, 0)
    assert isinstance(layer.output_size, tuple) and len(layer.output_size) == 2
    xb = x.detach().bfloat16()
    ob = layer(xb, torch.from_numpy(rois).cuda())
    refb = D.roi_align_c(xb.float().cpu().numpy(), rois, out_size, scale, 0, True)
    close(ob, torch.from_numpy(refb), ATOL32)
    empty = layer(x.detach(), torch.zeros(0, 5, device="cuda"))
    assert empty.shape == (0, C) + layer.output_size


def test_roi_align_not_aligned_and_fixed_grid(ops):
    rng = np.random.RandomState(8)
    inp = rng.randn(1, 4, 16, 16).astype(np.float32)
    rois = _rand_rois(rng, 20, 1, 64, 64)
    for aligned, sr in [(False, 0), (True, 2), (False, 3)]:
        ref = D.roi_align_c(inp, rois, 7, 0.25, sr, aligned)
        out = ops.roi_align(torch.from_numpy(inp).cuda(), torch.from_numpy(rois).cuda(), 7, 0.25, sr, 'avg', aligned)
        close(out, torch.from_numpy(ref), ATOL32)


@pytest.mark.parametrize("n,thr,offset", [(1, 0.5, 0), (63, 0.5, 0), (64, 0.7, 0), (65, 0.7, 1), (777, 0.3, 0),
                                           (2000, 0.7, 0), (8780, 0.7, 0)])
def test_nms_bit_exact(ops, n, thr, offset):
    rng = np.random.RandomState(n)
    xy = rng.rand(n, 2).astype(np.float32) * (300 if n < 5000 else 1200)
    wh = rng.rand(n, 2).astype(np.float32) * 80 + 1
    boxes = np.concatenate([xy, xy + wh], 1)
    scores = np.round(rng.rand(n), 3).astype(np.float32)          # ties on purpose
    dref, kref = D.nms_c(boxes, scores, thr, offset)
    dets, keep = ops.nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), thr, offset)
    assert keep.dtype == torch.int64
    np.testing.assert_array_equal(keep.cpu().numpy(), kref)
    np.testing.assert_array_equal(dets.cpu().numpy(), dref)


def test_nms_empty_and_batched(ops):
    dets, keep = ops.nms(torch.zeros(0, 4, device="cuda"), torch.zeros(0, device="cuda"), 0.5)
    assert dets.shape == (0, 5) and keep.shape == (0,) and keep.dtype == torch.int64
    rng = np.random.RandomState(11)
    n = 3000
    xy = rng.rand(n, 2).astype(np.float32) * 500
    boxes = np.concatenate([xy, xy + rng.rand(n, 2).astype(np.float32) * 90 + 1], 1)
    scores = rng.rand(n).astype(np.float32)
    ids = rng.randint(0, 5, n).astype(np.int64)
    for cfg in (dict(type="nms", iou_threshold=0.7), dict(type="nms", iou_threshold=0.5, split_thr=1000),
                dict(type="nms", iou_threshold=0.5, class_agnostic=True)):
        dref, kref = D.batched_nms(boxes, scores, ids, cfg)
        dets, keep = ops.batched_nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(),
                                     torch.from_numpy(ids).cuda(), cfg)
        np.testing.assert_array_equal(keep.cpu().numpy(), kref)
        np.testing.assert_array_equal(dets.cpu().numpy(), dref)
    d0, k0 = ops.batched_nms(torch.zeros(0, 4, device="cuda"), torch.zeros(0, device="cuda"),
                             torch.zeros(0, dtype=torch.long, device="cuda"), dict(type="nms", iou_threshold=0.5))
    assert d0.shape == (0, 5) and k0.shape == (0,)


# ------------------------------------------------------------------------------------------
# MFMA implicit-GEMM 3x3 convolution (bf16)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,H,W,Cin,Cout,relu", [(2, 13, 20, 64, 64, False), (1, 25, 40, 256, 256, True),
                                                  (3, 7, 9, 128, 192, False), (2, 50, 80, 256, 256, False),
                                                  (2, 100, 160, 256, 256, True)])      # P3: the 256-row tile (257..512 tiles of 128 rows)
def test_conv3x3_bf16(ops, N, H, W, Cin, Cout, relu):
    """Oracle: fp32 F.conv2d on the CPU with the same bf16-rounded operands (fpn.py:195-197 semantics).
    Tolerance: bf16 output rounding (2^-8 relative) of sums of 9*Cin bf16 products accumulated in fp32."""
    g = torch.Generator().manual_seed(N * H + Cin)
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).bfloat16().float()
    b = torch.randn(Cout, generator=g) * 0.1
    gy = torch.randn(N, Cout, H, W, generator=g).bfloat16().float()
    x0, w0, b0 = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(x0, w0, b0, padding=1)
    if relu:
        ref = F.relu(ref)
    (ref * gy).sum().backward()
    x1 = x.cuda().bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w1, b1 = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = ops.conv3x3(x1, w1, b1, relu)
    assert y.shape == ref.shape and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    (y.float() * gy.cuda()).sum().backward()
    close(y, ref, bf16_tol(ref, 2), msg="y")
    if relu:
        # an output within rounding of zero can land on the other side of the ReLU than the fp32 oracle's: the gradient of that ONE
        # pixel then enters (or leaves) dx on its 3 x 3 x Cin footprint with its full magnitude.  Such pixels are a 1e-5 fraction;
        # everything else must agree as without the ReLU, and no element may be off by more than one such gradient term.
        err = (x1.grad.float().cpu() - x0.grad).abs()
        bad = err > bf16_tol(x0.grad, 3)
        assert float(bad.float().mean()) < 2e-5, int(bad.sum())
        assert float(err.max()) <= float(gy.abs().max() * w.abs().max()) * 1.01 + bf16_tol(x0.grad, 3)
    else:
        close(x1.grad, x0.grad, bf16_tol(x0.grad, 3), msg="dx")
    close(w1.grad, w0.grad, bf16_tol(w0.grad, 3), msg="dw")
    close(b1.grad, b0.grad, bf16_tol(b0.grad, 3) + 1e-2, msg="db")


def test_nms_max_num_early_stop(ops):
    """max_num (mmcv nms) == slicing the full result; exercised across 64-box block boundaries."""
    rng = np.random.RandomState(21)
    n = 3000
    xy = rng.rand(n, 2).astype(np.float32) * 600
    boxes = np.concatenate([xy, xy + rng.rand(n, 2).astype(np.float32) * 60 + 1], 1)
    scores = rng.rand(n).astype(np.float32)
    dref, kref = D.nms_c(boxes, scores, 0.5)
    for m in (1, 63, 64, 65, 500, len(kref), len(kref) + 10):
        dets, keep = ops.nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 0.5, 0, 0, m)
        np.testing.assert_array_equal(keep.cpu().numpy(), kref[:m])
        np.testing.assert_array_equal(dets.cpu().numpy(), dref[:m])


@pytest.mark.parametrize("T,N1,N2", [(5000, 288, 96), (2048, 96, 384), (9000, 768, 192), (2500, 1536, 384), (4100, 256, 48)])
def test_linear_wgrad_bf16(ops, T, N1, N2):
    """ops.linear: forward/dx through the library, dW through the split-T MFMA kernel; oracle = fp32 autograd on CPU."""
    g = torch.Generator().manual_seed(T + N1)
    x = torch.randn(T, N2, generator=g).bfloat16().float()
    w = (torch.randn(N1, N2, generator=g) * 0.05).bfloat16().float()
    b = torch.randn(N1, generator=g).bfloat16().float()
    gy = torch.randn(T, N1, generator=g).bfloat16().float()
    x0, w0, b0 = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    (F.linear(x0, w0, b0) * gy).sum().backward()
    x1 = x.cuda().bfloat16().requires_grad_(True)
    w1, b1 = w.cuda().bfloat16().requires_grad_(True), b.cuda().bfloat16().requires_grad_(True)
    y = ops.linear(x1, w1, b1)
    (y.float() * gy.cuda()).sum().backward()
    close(w1.grad, w0.grad, bf16_tol(w0.grad, 2), msg="dW")      # fp32 accumulation, one bf16 rounding at the end
    close(x1.grad, x0.grad, bf16_tol(x0.grad, 3), msg="dx")
    close(b1.grad, b0.grad, bf16_tol(b0.grad, 2), msg="db")


def test_linear_wgrad_grouped_launch(ops):
    """swin_wgrad_record / swin_wgrad_flush: several Linear weight gradients of different shapes and token counts in ONE launch
    (csrc/wgrad_dma.hip) -- widths that are not multiples of the 128-wide tile, token counts that are not multiples of the 64-row
    stage, a problem without bias gradient, two problems accumulating into the SAME dw, and more problems than one launch's table
    holds -- against fp32 matmuls."""
    import ctypes
    from swin_transformer_object_detection_amd import _lib
    from swin_transformer_object_detection_amd.ops.functional import _p, _s, call
    shapes = [(8000, 1536, 384), (8000, 384, 384), (2000, 768, 3072), (5003, 288, 96), (777, 96, 96), (33, 1024, 1024),
              (8000, 384, 1536), (4100, 256, 48), (64, 8, 8)]
    shapes = shapes + [(1500 + 7 * k, 128 + 8 * (k % 5), 96 + 8 * (k % 3)) for k in range(30)]        # > 32 problems: two launches
    g = torch.Generator(device="cuda").manual_seed(11)
    probs, refs = [], []
    for k, (T, N1, N2) in enumerate(shapes):
        dy = (torch.randn(T, N1, device="cuda", generator=g) * 0.1).bfloat16()
        x = torch.randn(T, N2, device="cuda", generator=g).bfloat16()
        dw = torch.zeros(N1, N2, device="cuda")
        db = torch.zeros(N1, device="cuda") if k != 1 else None
        probs.append((dy, x, dw, db))
        refs.append((dy.float().t() @ x.float(), dy.float().sum(0)))
    # the same accumulator twice (a layer applied to two inputs)
    dy2 = (torch.randn(900, 1536, device="cuda", generator=g) * 0.1).bfloat16()
    x2 = torch.randn(900, 384, device="cuda", generator=g).bfloat16()
    probs.append((dy2, x2, probs[0][2], probs[0][3]))
    refs[0] = (refs[0][0] + dy2.float().t() @ x2.float(), refs[0][1] + dy2.float().sum(0))
    for dy, x, dw, db in probs:
        call("swin_wgrad_record", _p(dy), _p(x), _p(dw), _p(db), dy.shape[0], dy.shape[1], x.shape[1])
    tiles = ctypes.c_int64(0)
    assert _lib.lib().swin_wgrad_pending(ctypes.byref(tiles)) == len(probs) and tiles.value > 0
    call("swin_wgrad_flush", _s())
    assert _lib.lib().swin_wgrad_pending(None) == 0
    torch.cuda.synchronize()
    for (dy, x, dw, db), (rw, rb) in zip(probs[:len(shapes)], refs):
        assert float((dw - rw).abs().max()) <= 2e-3 * float(rw.abs().max()) + 1e-4, tuple(dw.shape)
        if db is not None:
            assert float((db - rb).abs().max()) <= 2e-3 * float(rb.abs().max()) + 1e-3, tuple(dw.shape)


def test_linear_wgrad96_group(ops):
    """swin_wgrad96_group (csrc/wgrad96.hip): every tile class of the Linear layers whose dimensions are multiples of 96 -- qkv / proj /
    fc1 / fc2 at C = 96 ... 768, PatchMerging.reduction -- in ONE launch: equal ranges of the work sequence per group of blocks (tile
    segments that start and end mid-contraction, several segments per block, clusters of 1 / 2 / 6 / 8 / 24 / 32 tiles); token counts that are not multiples of the stage rows, accumulation INTO non-zero buffers, one problem without bias gradient, two
    problems adding into the same dW.  Then the same problems through swin_wgrad_record / swin_wgrad_flush (which routes them here)
    next to a shape of the 128-tile form.  Against fp32 matmuls."""
    import ctypes
    from swin_transformer_object_detection_amd import _lib
    from swin_transformer_object_detection_amd.ops.functional import _p, _s, call
    shapes = [(20011, 288, 96), (16400, 96, 96), (17000, 384, 96), (16999, 96, 384), (16384, 576, 192), (18001, 192, 192),
              (16500, 768, 192), (16500, 192, 768), (16390, 192, 384), (2000, 1536, 384), (1999, 384, 1536), (2001, 1152, 384),
              (500, 2304, 768), (777, 768, 3072), (130, 480, 96), (33, 96, 672), (40000, 96, 96)]
    g = torch.Generator(device="cuda").manual_seed(5)
    probs, refs = [], []
    for k, (T, N1, N2) in enumerate(shapes):
        dy = (torch.randn(T, N1, device="cuda", generator=g) * 0.1).bfloat16()
        x = torch.randn(T, N2, device="cuda", generator=g).bfloat16()
        dw = torch.full((N1, N2), 0.5, device="cuda")
        db = torch.full((N1,), -0.25, device="cuda") if k != 2 else None
        probs.append((dy, x, dw, db))
        refs.append((0.5 + dy.float().t() @ x.float(), -0.25 + dy.float().sum(0)))
    last = len(shapes) - 1
    probs[last] = (probs[last][0], probs[last][1], probs[1][2], probs[1][3])           # the same accumulator twice
    refs[1] = (refs[1][0] + refs[last][0] - 0.5, refs[1][1] + refs[last][1] + 0.25)

    def run(direct):
        n = len(probs)
        if direct:
            pa = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() if t is not None else None for t in ts])
            call("swin_wgrad96_group", pa([q[0] for q in probs]), pa([q[1] for q in probs]), pa([q[2] for q in probs]),
                 pa([q[3] for q in probs]), (ctypes.c_int64 * n)(*[q[0].shape[0] for q in probs]),
                 (ctypes.c_int * n)(*[q[0].shape[1] for q in probs]), (ctypes.c_int * n)(*[q[1].shape[1] for q in probs]), n, _s())
        else:
            for dy, x, dw, db in probs:
                call("swin_wgrad_record", _p(dy), _p(x), _p(dw), _p(db), dy.shape[0], dy.shape[1], x.shape[1])
            call("swin_wgrad_flush", _s())
        torch.cuda.synchronize()

    def check():
        for (dy, x, dw, db), (rw, rb) in list(zip(probs, refs))[:last]:
            assert float((dw - rw).abs().max()) <= 2e-3 * float(rw.abs().max()) + 1e-4, tuple(dw.shape)
            if db is not None:
                assert float((db - rb).abs().max()) <= 2e-3 * float(rb.abs().max()) + 1e-3, tuple(dw.shape)

    run(True)
    check()
    # again through the recording API, next to a problem of the 128-tile form
    for dy, x, dw, db in probs[:last]:
        dw.fill_(0.5)
        if db is not None:
            db.fill_(-0.25)
    dy3 = (torch.randn(8000, 256, device="cuda", generator=g) * 0.1).bfloat16()
    x3 = torch.randn(8000, 1024, device="cuda", generator=g).bfloat16()
    dw3 = torch.zeros(256, 1024, device="cuda")
    call("swin_wgrad_record", _p(dy3), _p(x3), _p(dw3), _p(None), 8000, 256, 1024)
    run(False)
    check()
    r3 = dy3.float().t() @ x3.float()
    assert float((dw3 - r3).abs().max()) <= 2e-3 * float(r3.abs().max()) + 1e-4
    # a shape that is not this kernel's is refused, not mangled
    n = 1
    bad = _lib.lib().swin_wgrad96_group((ctypes.c_void_p * n)(probs[0][0].data_ptr()), (ctypes.c_void_p * n)(probs[0][1].data_ptr()),
                                        (ctypes.c_void_p * n)(probs[0][2].data_ptr()), None, (ctypes.c_int64 * n)(20011),
                                        (ctypes.c_int * n)(280), (ctypes.c_int * n)(96), n, _s())
    assert bad == 2                     # SWIN_ERR_UNSUPPORTED (include/swin_hip.h)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_roi_align_multilevel_group_equals_separate_calls(ops, dtype):
    """The grouped form (bbox 7x7 + mask 14x14 RoIs of one stage, one shared fp32 backward accumulator) == two separate
    roi_align_multilevel calls (which are checked against the oracle): same outputs; summed feature gradients equal up to
    the rounding of ONE cast instead of two casts + an addition."""
    rng = np.random.RandomState(7)
    N, C = 2, 16
    strides = [4, 8, 16, 32]
    shapes = [(48, 64), (24, 32), (12, 16), (6, 8)]
    feats_np = [rng.randn(N, C, h, w).astype(np.float32) for h, w in shapes]
    sets = []
    for K, size in ((120, 7), (40, 14)):
        rois = _rand_rois(rng, K, N, 256, 192)
        lvls = rng.randint(-1, 4, K).astype(np.int32)
        sets.append((torch.from_numpy(rois).cuda(), torch.from_numpy(lvls).cuda(), size))
    gys = [torch.from_numpy(rng.randn(K, C, sz, sz).astype(np.float32)).cuda() for (K, sz) in ((120, 7), (40, 14))]

    def leaves():
        return [torch.from_numpy(f).cuda().to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True) for f in feats_np]
    fa = leaves()
    outs_a = [ops.roi_align_multilevel(fa, r, l, sz, strides, 0, True, out_dtype=dtype) for r, l, sz in sets]
    sum((o.float() * g).sum() for o, g in zip(outs_a, gys)).backward()
    fb = leaves()
    outs_b = ops.roi_align_multilevel_group(fb, sets, strides, 0, True, out_dtype=dtype)
    sum((o.float() * g).sum() for o, g in zip(outs_b, gys)).backward()
    for a, b in zip(outs_a, outs_b):
        assert a.dtype == b.dtype and torch.equal(a, b)
    for a, b in zip(fa, fb):
        tol = 1e-5 if dtype == torch.float32 else 2.0 ** -7
        close(b.grad.float(), a.grad.float(), tol * float(a.grad.float().abs().max()) + 1e-6, tol)
    # a set without an incoming gradient is skipped
    fc = leaves()
    oc = ops.roi_align_multilevel_group(fc, sets, strides, 0, True, out_dtype=dtype)
    (oc[0].float() * gys[0]).sum().backward()
    fd = leaves()
    (ops.roi_align_multilevel(fd, *sets[0][:2], sets[0][2], strides, 0, True, out_dtype=dtype).float() * gys[0]).sum().backward()
    for a, b in zip(fd, fc):        # same launches; the fp32 atomics of the backward kernel may land in another order
        tol = 1e-5 if dtype == torch.float32 else 2.0 ** -7
        close(b.grad.float(), a.grad.float(), tol * float(a.grad.float().abs().max()) + 1e-6, tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("out_size", [7, 14])
def test_roi_align_multilevel(ops, dtype, out_size):
    """One launch over the pyramid == per-level mmcv roi_align on the rois of that level (single_level_roi_extractor.py
    :84-107); lvl = -1 rows are skipped (zero output, no gradient)."""
    rng = np.random.RandomState(31)
    N, C = 2, 8
    strides = [4, 8, 16, 32]
    shapes = [(48, 64), (24, 32), (12, 16), (6, 8)]
    feats_np = [rng.randn(N, C, h, w).astype(np.float32) for h, w in shapes]
    if dtype == torch.bfloat16:
        feats_np = [torch.from_numpy(f).bfloat16().float().numpy() for f in feats_np]
    K = 90
    rois = _rand_rois(rng, K, N, 256, 192)
    lvls = rng.randint(-1, 4, K).astype(np.int32)
    lvls[:4] = [0, 1, 2, 3]
    rois[4, 1:] = [0, 40, 255, 46]; lvls[4] = 0          # thin, full-width: 64-px footprint, sampling grid 10 x 1
    rois[5, 1:] = [0, 0, 255, 191]; lvls[5] = 0          # whole image on the finest level
    feats = [torch.from_numpy(f).cuda().to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
             for f in feats_np]
    out = ops.roi_align_multilevel(feats, torch.from_numpy(rois).cuda(), torch.from_numpy(lvls).cuda(), out_size, strides, 0, True)
    assert out.dtype == torch.float32 and out.shape == (K, C, out_size, out_size)
    ref = np.zeros((K, C, out_size, out_size), np.float32)
    for l in range(4):
        sel = np.nonzero(lvls == l)[0]
        ref[sel] = D.roi_align_c(feats_np[l], rois[sel], out_size, 1.0 / strides[l], 0, True)
    close(out, torch.from_numpy(ref), ATOL32)
    if dtype == torch.bfloat16:      # bf16 output option: the fp32 result rounded once; bf16 gradient in
        ob = ops.roi_align_multilevel([f.detach() for f in feats], torch.from_numpy(rois).cuda(), torch.from_numpy(lvls).cuda(),
                                      out_size, strides, 0, True, out_dtype=torch.bfloat16)
        assert ob.dtype == torch.bfloat16 and torch.equal(ob, out.detach().bfloat16())
    gy = rng.randn(*ref.shape).astype(np.float32)
    if dtype == torch.bfloat16:
        gy = torch.from_numpy(gy).bfloat16().float().numpy()
        f2 = [f.detach().clone().requires_grad_(True) for f in feats]
        o2 = ops.roi_align_multilevel(f2, torch.from_numpy(rois).cuda(), torch.from_numpy(lvls).cuda(), out_size, strides, 0, True,
                                      out_dtype=torch.bfloat16)
        (o2.float() * torch.from_numpy(gy).cuda()).sum().backward()          # gradient reaches the op in bf16
    (out * torch.from_numpy(gy).cuda()).sum().backward()
    for l in range(4):
        sel = np.nonzero(lvls == l)[0]
        gref = D.roi_align_bwd_c(gy[sel], rois[sel], feats_np[l].shape, 1.0 / strides[l], 0, True)
        if dtype == torch.bfloat16:
            close(f2[l].grad.float(), torch.from_numpy(gref).float(), 2e-4, 2.0 ** -8)
        assert feats[l].grad is not None and feats[l].grad.dtype == dtype
        if dtype == torch.float32:
            close(feats[l].grad, torch.from_numpy(gref).float(), 2e-4, 1e-4)
        else:   # fp32 accumulation, one rounding to bf16 at the end: <= 1/2 ulp(bf16) = 2^-9 relative
            close(feats[l].grad.float(), torch.from_numpy(gref).float(), 2e-4, 2.0 ** -8)


def test_nms_static_matches_dynamic(ops):
    """Fixed-size form: first max_num kept indices, -1 / invalid padded; identical to the dynamic result's prefix."""
    rng = np.random.RandomState(33)
    for n, m in [(3000, 1000), (500, 1000), (70, 64), (1, 5)]:
        xy = rng.rand(n, 2).astype(np.float32) * 400
        boxes = np.concatenate([xy, xy + rng.rand(n, 2).astype(np.float32) * 80 + 1], 1)
        scores = rng.rand(n).astype(np.float32)
        dref, kref = D.nms_c(boxes, scores, 0.7)
        inds, valid = ops.nms_static(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 0.7, m)
        assert inds.shape == (m,) and valid.shape == (m,)
        k = min(m, len(kref))
        assert int(valid.sum()) == k and bool(valid[:k].all())
        np.testing.assert_array_equal(inds[:k].cpu().numpy(), kref[:k])
        # batched (class-aware offsets, nms.py:258-262 of mmcv)
        idxs = rng.randint(0, 3, n)
        bref, bkeep = D.batched_nms(boxes, scores, idxs, dict(type='nms', iou_threshold=0.7))
        dets, v = ops.batched_nms_static(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(),
                                         torch.from_numpy(idxs).cuda(), 0.7, m)
        k = min(m, len(bkeep))
        assert dets.shape == (m, 5) and int(v.sum()) == k
        np.testing.assert_array_equal(dets[:k].cpu().numpy(), bref[:k])


# ------------------------------------------------------------------------------------------
# training targets: assigner / sampler kernels
# ------------------------------------------------------------------------------------------
def _boxes(rng, n, size=640.0, wh=160.0):
    xy = rng.rand(n, 2).astype(np.float32) * size
    return np.concatenate([xy, xy + rng.rand(n, 2).astype(np.float32) * wh + 2], 1).astype(np.float32)


@pytest.mark.parametrize("n,g,pos,neg,minpos,lowq", [(20000, 8, 0.7, 0.3, 0.3, True), (1000, 5, 0.5, 0.5, 0.5, False),
                                                      (3000, 300, 0.6, 0.4, 0.0, True), (77, 1, 0.5, 0.5, 0.5, True),
                                                      (500, 0, 0.7, 0.3, 0.3, True)])
def test_max_iou_assign_bit_exact(ops, n, g, pos, neg, minpos, lowq):
    """Device MaxIoUAssigner == the numpy restatement of max_iou_assigner.py:128-212, index for index."""
    rng = np.random.RandomState(n + g)
    boxes, gts = _boxes(rng, n), _boxes(rng, g)
    if g:
        boxes[5] = gts[0]                       # an exact match (IoU 1) and a duplicate of it (tie on the gt maximum)
        boxes[9] = gts[0]
    labels = rng.randint(0, 80, g)
    a_ref, m_ref, l_ref = CO.max_iou_assign(boxes, gts, pos, neg, minpos, lowq, labels)
    a, m, l = ops.max_iou_assign(dev(torch.from_numpy(boxes)), dev(torch.from_numpy(gts)), pos, neg, minpos, lowq,
                                 dev(torch.from_numpy(labels)))
    np.testing.assert_array_equal(a.cpu().numpy(), a_ref)
    np.testing.assert_array_equal(l.cpu().numpy(), l_ref)
    np.testing.assert_array_equal(m.cpu().numpy(), m_ref)          # same fp32 operation order: bit-exact IoU


def test_max_iou_assign_leading_gts_and_valid_mask(ops):
    """add_gt_as_proposals (base_sampler.py:77-84): the gt rows are appended to the ASSIGNED result, i.e. they match
    themselves and do not take part in the per-gt maxima; masked slots (valid == 0: padding of fixed-size lists, anchors
    outside allowed_border which anchor_head.py:200-207 removes BEFORE assigning) are never assigned and do not feed
    the per-gt maxima either -- the result equals assigning the valid boxes alone."""
    rng = np.random.RandomState(5)
    g, n = 6, 400
    gts, props = _boxes(rng, g), _boxes(rng, n)
    labels = rng.randint(0, 80, g)
    valid = rng.rand(n) > 0.2
    _, m_ref, _ = CO.max_iou_assign(props, gts, 0.5, 0.5, 0.5, True, labels)
    a_v, _, l_v = CO.max_iou_assign(props[valid], gts, 0.5, 0.5, 0.5, True, labels)
    a_ref = np.full(n, -1, np.int64); a_ref[valid] = a_v
    l_ref = np.full(n, -1, np.int64); l_ref[valid] = l_v
    a_ref = np.concatenate([np.arange(1, g + 1), a_ref]); l_ref = np.concatenate([labels, l_ref])
    allb = np.concatenate([gts, props]); v = np.concatenate([np.ones(g, bool), valid])
    a, m, l = ops.max_iou_assign(dev(torch.from_numpy(allb)), dev(torch.from_numpy(gts)), 0.5, 0.5, 0.5, True,
                                 dev(torch.from_numpy(labels)), num_leading_gt=g, valid=dev(torch.from_numpy(v)))
    np.testing.assert_array_equal(a.cpu().numpy(), a_ref)
    np.testing.assert_array_equal(l.cpu().numpy(), l_ref)
    np.testing.assert_array_equal(m.cpu().numpy()[g:], m_ref)
    assert bool((m[:g] == 1).all())


def test_random_sample_counts_and_uniformity(ops):
    """Device RandomSampler: the counts of random_sampler.py:31-78, positives first, no duplicates, never an ignored
    box; deterministic per seed; every candidate equally likely (frequency test over 300 seeds)."""
    torch.manual_seed(0)
    for n_pos, n_neg, n_ign, num, frac in [(10, 100000, 50, 256, 0.5), (300, 1000, 0, 256, 0.5), (5, 20, 3, 512, 0.25),
                                           (0, 40, 0, 64, 0.25), (200, 30, 0, 256, 0.5), (0, 0, 9, 16, 0.5)]:
        a = torch.cat([torch.randint(1, 5, (n_pos,)), torch.zeros(n_neg, dtype=torch.long), -torch.ones(n_ign, dtype=torch.long)])
        a = a[torch.randperm(a.numel())].cuda()
        idx, is_pos, valid = ops.random_sample(a, num, frac, seed=1234)
        assert idx.shape == (num,) and is_pos.shape == (num,) and valid.shape == (num,)
        exp_pos = min(n_pos, int(num * frac)); exp_neg = min(n_neg, num - exp_pos)
        assert int(is_pos.sum()) == exp_pos and int(valid.sum()) == exp_pos + exp_neg
        assert bool((a[idx[is_pos]] > 0).all()) and bool((a[idx[valid & ~is_pos]] == 0).all())
        assert bool(is_pos[:exp_pos].all()) and bool(valid[:exp_pos + exp_neg].all()) and not bool(valid[exp_pos + exp_neg:].any())
        assert idx[valid].unique().numel() == exp_pos + exp_neg
        idx2, _, _ = ops.random_sample(a, num, frac, seed=1234)
        assert torch.equal(idx, idx2)
        if exp_neg < n_neg:
            idx3, _, _ = ops.random_sample(a, num, frac, seed=99)
            assert not torch.equal(idx, idx3)
    n, num, trials = 2000, 256, 300
    a = torch.zeros(n, dtype=torch.long, device="cuda"); a[:100] = 1
    hits = torch.zeros(n, device="cuda")
    for s in range(trials):
        idx, is_pos, valid = ops.random_sample(a, num, 0.25, seed=1000 + s)
        hits[idx[valid]] += 1
    hp, hn = hits[:100].cpu().numpy(), hits[100:].cpu().numpy()
    # positives: 64 of 100 per draw -> p = 0.64; negatives: 192 of 1900 -> p = 0.101; binomial 5-sigma bands
    for h, p in ((hp, 0.64), (hn, 192 / 1900)):
        sd = (trials * p * (1 - p)) ** 0.5
        assert abs(h.mean() - trials * p) < 1e-6 * trials + 1e-3 or abs(h.mean() - trials * p) < sd
        assert h.max() < trials * p + 5.5 * sd and h.min() > trials * p - 5.5 * sd


def test_delta2bbox_and_bbox_targets(ops):
    """DeltaXYWHBBoxCoder on the device vs the oracle's numpy decode (callers_oracle.delta2bbox) and an fp64 encode;
    fp32, atol 1e-4 px for decode (exp of a clamped delta times a box side <= ~1e3) / 1e-5 for the encoded deltas."""
    rng = np.random.RandomState(41)
    n = 5000
    rois = _boxes(rng, n, 1000.0, 300.0)
    deltas = (rng.randn(n, 4) * np.array([0.5, 0.5, 2.0, 2.0])).astype(np.float32)     # |dw| beyond the ratio clip too
    for means, stds, shape in [((0., 0., 0., 0.), (1., 1., 1., 1.), (800, 1216)), ((0., 0., 0., 0.), (.1, .1, .2, .2), None)]:
        ref = CO.delta2bbox(rois, deltas, means, stds, shape)
        out = ops.delta2bbox(dev(torch.from_numpy(rois)), dev(torch.from_numpy(deltas)), means, stds, shape)
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-6, atol=2e-4)
    # targets of a sample: positives encode against their gt, the rest is zero / background
    g = 7
    gts = _boxes(rng, g, 900.0, 250.0)
    assigned = rng.randint(-1, g + 1, n).astype(np.int64)
    labels_all = np.where(assigned > 0, rng.randint(0, 80, n), -1).astype(np.int64)
    k = 300
    inds = rng.choice(n, k, replace=False).astype(np.int64)
    flags = np.where(assigned[inds] > 0, 3, np.where(assigned[inds] == 0, 1, 0)).astype(np.uint8)
    means, stds = (0., 0., 0., 0.), (.1, .1, .2, .2)
    b, d, gi, lab = ops.bbox_targets(dev(torch.from_numpy(rois)), dev(torch.from_numpy(inds)), dev(torch.from_numpy(flags)),
                                     dev(torch.from_numpy(assigned)), dev(torch.from_numpy(gts)), means, stds,
                                     dev(torch.from_numpy(labels_all)), bg_label=80)
    pos, used = flags == 3, flags >= 1
    eb = np.where(used[:, None], rois[inds], np.array([0, 0, 1, 1], np.float32))
    np.testing.assert_array_equal(b.cpu().numpy(), eb)
    np.testing.assert_array_equal(gi.cpu().numpy(), np.where(used, np.maximum(assigned[inds] - 1, 0), 0))
    np.testing.assert_array_equal(lab.cpu().numpy(), np.where(pos, labels_all[inds], 80))
    p64, g64 = rois[inds].astype(np.float64), gts[np.maximum(assigned[inds] - 1, 0)].astype(np.float64)
    pw, ph = p64[:, 2] - p64[:, 0], p64[:, 3] - p64[:, 1]
    gw, gh = g64[:, 2] - g64[:, 0], g64[:, 3] - g64[:, 1]
    enc = np.stack([((g64[:, 0] + g64[:, 2]) - (p64[:, 0] + p64[:, 2])) * 0.5 / pw, ((g64[:, 1] + g64[:, 3]) - (p64[:, 1] + p64[:, 3])) * 0.5 / ph,
                    np.log(gw / pw), np.log(gh / ph)], 1) / np.array(stds)
    enc = np.where(pos[:, None], enc, 0.0)
    np.testing.assert_allclose(d.cpu().numpy(), enc, rtol=1e-5, atol=1e-4)
    # round trip: decode(encode(gt)) == gt for the positives
    back = ops.delta2bbox(b, d, means, stds, None, wh_ratio_clip=1e-9)
    np.testing.assert_allclose(back.cpu().numpy()[pos], g64[pos], rtol=1e-5, atol=2e-3)


def test_roi_align_nchw_thread_per_bin_path(ops):
    """> 2^18 output bins: the NCHW forward takes the thread-per-bin kernel (the wave-per-bin one serves small outputs)."""
    rng = np.random.RandomState(9)
    inp = rng.randn(1, 32, 20, 24).astype(np.float32)
    rois = _rand_rois(rng, 200, 1, 96, 80)
    ref = D.roi_align_c(inp, rois, 7, 0.25, 0, True)
    assert ref.size > (1 << 18)
    out = ops.roi_align(torch.from_numpy(inp).cuda(), torch.from_numpy(rois).cuda(), 7, 0.25, 0, 'avg', True)
    close(out, torch.from_numpy(ref), ATOL32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_roi_align_nchw_separable_rows_path(ops, dtype):
    """The NCHW forward of the mask targets (csrc/roi_align.hip roi_align_fwd_nchw_rows: a wave per bin-row, the bilinear sums
    taken separably) against the oracle's sample-by-sample sums, at the mask targets' geometry (28 x 28 bins, adaptive grids of up
    to 20 x 20 samples on a 1-channel full-resolution map) and on the corners of its rules: RoIs hanging over every border (samples
    beyond -1 / H are dropped, those in between clamp), a RoI far outside, zero-size and inverted RoIs under aligned=True (no samples),
    not-aligned RoIs narrower than a pixel, and two bins per axis over a 200-pixel RoI (footprint > 64 rows: the sample loop)."""
    rng = np.random.RandomState(12)
    H, W = 300, 420
    inp = (rng.rand(3, 1, H, W) > 0.5).astype(np.float32)                       # binary instance masks
    rois = _rand_rois(rng, 40, 3, W, H)
    rois[0, 1:] = [-30.5, -12.25, 90.0, 75.5]                                     # over the top-left corner
    rois[1, 1:] = [350.2, 250.7, 460.0, 333.3]                                    # over the bottom-right corner
    rois[2, 1:] = [-500, -500, -400, -420]                                        # nowhere near the map
    rois[3, 1:] = [50, 60, 50, 60]                                                # zero size
    rois[4, 1:] = [120, 90, 100, 70]                                              # inverted
    rois[5, 1:] = [10.3, 20.6, 10.9, 21.1]                                        # narrower than a pixel
    rois[6, 1:] = [0, 0, W, H]                                                    # the whole map
    rois[7, 1:] = [5, 5, 414.5, 295.5]
    x = torch.from_numpy(inp).cuda().to(dtype)
    for out_size, aligned in ((28, True), (28, False), (2, True), (7, True)):
        ref = D.roi_align_c(inp, rois, out_size, 1.0, 0, aligned)
        assert ref.size <= (1 << 18)
        out = ops.roi_align(x, torch.from_numpy(rois).cuda(), out_size, 1.0, 0, 'avg', aligned)
        close(out.float(), torch.from_numpy(ref), ATOL32, msg=f"out_size={out_size} aligned={aligned}")


@pytest.mark.parametrize("M,N,K", [(5000, 288, 96), (1024, 1024, 12544), (2000, 768, 3072), (130, 88, 1024)])
def test_gemm_bf16_direct(ops, M, N, K):
    """swin_gemm_bf16 (hipBLASLt with cached plans) == fp32 matmul of the same bf16 operands, both layouts, with bias;
    bf16 output: 1 ulp of bf16 on the result magnitude (fp32 accumulation inside)."""
    from swin_transformer_object_detection_amd.ops.functional import gemm_bf16
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    b = torch.randn(N, generator=g).bfloat16().cuda()
    ref = a.float() @ w.float().t() + b.float()
    out = gemm_bf16(a, w, b)
    assert out.dtype == torch.bfloat16 and out.shape == (M, N)
    tol = float(ref.abs().max()) * 2.0 ** -8
    assert float((out.float() - ref).abs().max()) <= tol
    dy = torch.randn(M, N, generator=g).bfloat16().cuda()
    ref2 = dy.float() @ w.float()
    out2 = gemm_bf16(dy, w, None, b_is_kn=True)
    assert out2.shape == (M, K)
    assert float((out2.float() - ref2).abs().max()) <= float(ref2.abs().max()) * 2.0 ** -8
    out3 = gemm_bf16(a, w, b)                          # cached plan, different bias pointer
    assert torch.equal(out, out3)


# ------------------------------------------------------------------------------------------
# loss kernels vs the reference formulas in torch fp32 (values and input gradients)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rpn_loss_kernel(ops, dtype):
    """anchor_head.py:375-434: sigmoid CE over the sampled anchors + L1 on the positives, / number of samples."""
    g = torch.Generator().manual_seed(3)
    B, A, S = 2, 5000, 256
    cls = (torch.randn(B, A, generator=g) * 2).to(dtype)
    reg = torch.randn(B, A, 4, generator=g).to(dtype)
    inds = torch.stack([torch.randperm(A, generator=g)[:S] for _ in range(B)])
    flags = torch.randint(0, 3, (B, S), generator=g).to(torch.uint8)
    flags = torch.where(flags == 2, torch.full_like(flags, 3), flags)          # 0 unused, 1 negative, 3 positive
    tgt = torch.randn(B, S, 4, generator=g)
    c0, r0 = cls.float().clone().requires_grad_(True), reg.float().clone().requires_grad_(True)
    valid, pos = (flags & 1).bool(), (flags & 2).bool()
    n = valid.sum().clamp(min=1).float()
    ci = torch.gather(c0, 1, inds); ri = torch.gather(r0, 1, inds[..., None].expand(-1, -1, 4))
    lc = (F.binary_cross_entropy_with_logits(ci, pos.float(), reduction='none') * valid).sum() / n
    lb = ((ri - tgt).abs() * pos[..., None]).sum() / n
    (lc * 1.5 + lb * 0.5).backward()
    c1, r1 = cls.cuda().requires_grad_(True), reg.cuda().requires_grad_(True)
    olc, olb = ops.rpn_loss(c1, r1, inds.cuda(), flags.cuda(), tgt.cuda())
    (olc * 1.5 + olb * 0.5).backward()
    close(olc, lc.detach(), 1e-5, 1e-5); close(olb, lb.detach(), 1e-5, 1e-5)
    gtol = 1e-7 if dtype == torch.float32 else 2.0 ** -8 * float(c0.grad.abs().max())
    close(c1.grad.float(), c0.grad, gtol, 1e-5 if dtype == torch.float32 else 2.0 ** -8)
    close(r1.grad.float(), r0.grad, gtol, 1e-5 if dtype == torch.float32 else 2.0 ** -8)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bbox_and_mask_loss_kernels(ops, dtype):
    """bbox_head.py:188-238 (CE, accuracy, class-specific L1) and mask_cross_entropy (mean BCE of the class channel)."""
    g = torch.Generator().manual_seed(4)
    n, nc = 300, 80
    cls = (torch.randn(n, nc + 1, generator=g) * 2).to(dtype)
    bbox = torch.randn(n, 4 * nc, generator=g).to(dtype)
    labels = torch.randint(0, nc + 1, (n,), generator=g)
    valid = torch.rand(n, generator=g) > 0.2
    pos = (labels < nc) & valid & (torch.rand(n, generator=g) > 0.3)
    labels = torch.where(pos, labels.clamp(max=nc - 1), torch.where(valid, torch.full_like(labels, nc), labels))
    tgt = torch.randn(n, 4, generator=g)
    flags = valid.to(torch.uint8) + 2 * pos.to(torch.uint8)
    c0, b0 = cls.float().clone().requires_grad_(True), bbox.float().clone().requires_grad_(True)
    nv = valid.sum().clamp(min=1).float()
    lc = (F.cross_entropy(c0, labels, reduction='none') * valid).sum() / nv
    acc = ((c0.argmax(1) == labels) & valid).sum() / nv * 100
    pred = b0.view(n, nc, 4)[torch.arange(n), labels.clamp(max=nc - 1)]
    lb = ((pred - tgt).abs() * pos[:, None]).sum() / nv
    (lc * 2 + lb).backward()
    c1, b1 = cls.cuda().requires_grad_(True), bbox.cuda().requires_grad_(True)
    olc, oacc, olb = ops.bbox_loss(c1, b1, labels.cuda(), tgt.cuda(), flags.cuda(), nc)
    (olc * 2 + olb).backward()
    close(olc, lc.detach(), 1e-5, 1e-5); close(olb, lb.detach(), 1e-5, 1e-5); close(oacc, acc, 1e-4)
    rel = 1e-5 if dtype == torch.float32 else 2.0 ** -8
    close(c1.grad.float(), c0.grad, rel * float(c0.grad.abs().max()) + 1e-8, rel)
    close(b1.grad.float(), b0.grad, rel * float(b0.grad.abs().max()) + 1e-8, rel)
    # mask
    m, P = 40, 28
    mp = (torch.randn(m, nc, P, P, generator=g) * 2).to(dtype)
    mt = (torch.rand(m, P, P, generator=g) > 0.5).float()
    ml = torch.randint(0, nc, (m,), generator=g)
    mv = torch.rand(m, generator=g) > 0.3
    p0 = mp.float().clone().requires_grad_(True)
    per = F.binary_cross_entropy_with_logits(p0[torch.arange(m), ml], mt, reduction='none').mean(dim=(1, 2))
    lm = (per * mv).sum() / mv.sum().clamp(min=1)
    (lm * 3).backward()
    p1 = mp.cuda().requires_grad_(True)
    olm = ops.mask_loss(p1, mt.cuda(), ml.cuda(), mv.cuda())
    (olm * 3).backward()
    close(olm, lm.detach(), 1e-5, 1e-5)
    close(p1.grad.float(), p0.grad, rel * float(p0.grad.abs().max()) + 1e-9, rel)


@pytest.mark.parametrize("beta", [1.0 / 9.0, 1.0])
def test_rpn_loss_smooth_l1(ops, beta):
    """SmoothL1Loss(beta) regression term (smooth_l1_loss.py:10-28; the Cascade configs' RPN uses beta = 1/9)."""
    g = torch.Generator().manual_seed(13)
    B, A, S = 2, 3000, 256
    cls = torch.randn(B, A, generator=g) * 2
    reg = torch.randn(B, A, 4, generator=g) * 0.3
    inds = torch.stack([torch.randperm(A, generator=g)[:S] for _ in range(B)])
    flags = torch.randint(0, 3, (B, S), generator=g).to(torch.uint8)
    flags = torch.where(flags == 2, torch.full_like(flags, 3), flags)
    tgt = torch.randn(B, S, 4, generator=g) * 0.3
    r0 = reg.clone().requires_grad_(True)
    valid, pos = (flags & 1).bool(), (flags & 2).bool()
    n = valid.sum().clamp(min=1).float()
    ri = torch.gather(r0, 1, inds[..., None].expand(-1, -1, 4))
    lb = (F.smooth_l1_loss(ri, tgt, beta=beta, reduction='none') * pos[..., None]).sum() / n
    lb.backward()
    ref = float((CO.smooth_l1((ri.detach() - tgt).numpy(), beta) * pos[..., None].numpy()).sum() / float(n))
    assert abs(ref - float(lb)) < 1e-5                     # the oracle's formula == torch's smooth_l1_loss
    c1, r1 = cls.cuda().requires_grad_(True), reg.cuda().requires_grad_(True)
    _, olb = ops.rpn_loss(c1, r1, inds.cuda(), flags.cuda(), tgt.cuda(), beta)
    olb.backward()
    close(olb, lb.detach(), 1e-5, 1e-5)
    close(r1.grad, r0.grad, 1e-7, 1e-5)


def _giou_case(g, n, nc, agnostic):
    xy = torch.rand(n, 2, generator=g) * 300
    rois = torch.cat([xy, xy + torch.rand(n, 2, generator=g) * 120 + 4], 1)
    txy = xy + (torch.rand(n, 2, generator=g) - 0.5) * 60
    tgt = torch.cat([txy, txy + torch.rand(n, 2, generator=g) * 120 + 4], 1)
    tgt[::7] = rois[::7] + 400                                  # disjoint pairs: the enclosing-box term alone
    bbox = torch.randn(n, 4 if agnostic else 4 * nc, generator=g) * 1.5
    bbox[::11] *= 8                                             # some deltas beyond the wh_ratio clip
    return rois, tgt, bbox


@pytest.mark.parametrize("agnostic", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bbox_loss_giou_decoded(ops, dtype, agnostic):
    """reg_decoded_bbox=True + GIoULoss (bbox_head.py:215-216, iou_loss.py:78-101): value against the numpy oracle, input
    gradients against autograd through the torch restatement (detector.giou_loss_elem over detector.delta2bbox)."""
    from swin_transformer_object_detection_amd import detector
    g = torch.Generator().manual_seed(21)
    n, nc = 400, 80
    means, stds = (0., 0., 0., 0.), (0.05, 0.05, 0.1, 0.1)
    cls = (torch.randn(n, nc + 1, generator=g) * 2).to(dtype)
    rois, tgt, bbox = _giou_case(g, n, nc, agnostic)
    bbox = bbox.to(dtype)
    labels = torch.randint(0, nc + 1, (n,), generator=g)
    valid = torch.rand(n, generator=g) > 0.2
    pos = (labels < nc) & valid
    labels = torch.where(valid & ~pos, torch.full_like(labels, nc), labels)
    flags = valid.to(torch.uint8) + 2 * pos.to(torch.uint8)
    b0 = bbox.float().clone().requires_grad_(True)
    nv = valid.sum().clamp(min=1).float()
    pred = b0.view(n, 4) if agnostic else b0.view(n, nc, 4)[torch.arange(n), labels.clamp(max=nc - 1)]
    dec = detector.delta2bbox(rois, pred, means, stds)
    lb = (detector.giou_loss_elem(dec, tgt, 1e-6) * pos).sum() / nv
    lb.backward()
    lc_ref, acc_ref, lb_ref = CO.bbox_head_loss(cls.float().numpy(), bbox.float().numpy(), labels.numpy(), tgt.numpy(), flags.numpy(),
                                                nc, agnostic, 0.0, (rois.numpy(), means, stds, 1e-6))
    assert abs(lb_ref - float(lb)) < 2e-5                   # numpy oracle == torch restatement
    c1, b1 = cls.cuda().requires_grad_(True), bbox.cuda().requires_grad_(True)
    olc, oacc, olb = ops.bbox_loss(c1, b1, labels.cuda(), tgt.cuda(), flags.cuda(), nc, agnostic, 0.0,
                                   (rois.cuda(), means, stds, 1e-6))
    (olc + olb * 10).backward()
    close(olb, torch.tensor(lb_ref), 2e-5, 1e-5); close(olc, torch.tensor(lc_ref), 1e-5, 1e-5); close(oacc, torch.tensor(acc_ref), 1e-3)
    rel = 1e-4 if dtype == torch.float32 else 2.0 ** -7
    gref = b0.grad * 10
    close(b1.grad.float(), gref, rel * float(gref.abs().max()) + 1e-8, rel)
    assert float(b1.grad.float().abs().sum()) > 0


@pytest.mark.parametrize("agnostic", [False, True])
def test_bbox_loss_smooth_l1_and_agnostic(ops, agnostic):
    """The _base_ cascade heads: class-agnostic SmoothL1(beta=1) on encoded deltas (cascade_mask_rcnn_swin_fpn.py:52-70)."""
    g = torch.Generator().manual_seed(8)
    n, nc = 300, 80
    cls = torch.randn(n, nc + 1, generator=g) * 2
    bbox = torch.randn(n, 4 if agnostic else 4 * nc, generator=g) * 1.2
    labels = torch.randint(0, nc + 1, (n,), generator=g)
    valid = torch.rand(n, generator=g) > 0.1
    pos = (labels < nc) & valid
    labels = torch.where(valid & ~pos, torch.full_like(labels, nc), labels)
    tgt = torch.randn(n, 4, generator=g)
    flags = valid.to(torch.uint8) + 2 * pos.to(torch.uint8)
    b0 = bbox.clone().requires_grad_(True)
    nv = valid.sum().clamp(min=1).float()
    pred = b0.view(n, 4) if agnostic else b0.view(n, nc, 4)[torch.arange(n), labels.clamp(max=nc - 1)]
    lb = (F.smooth_l1_loss(pred, tgt, beta=1.0, reduction='none') * pos[:, None]).sum() / nv
    lb.backward()
    _, _, lb_ref = CO.bbox_head_loss(cls.numpy(), bbox.numpy(), labels.numpy(), tgt.numpy(), flags.numpy(), nc, agnostic, 1.0)
    assert abs(lb_ref - float(lb)) < 1e-5
    c1, b1 = cls.cuda().requires_grad_(True), bbox.cuda().requires_grad_(True)
    _, _, olb = ops.bbox_loss(c1, b1, labels.cuda(), tgt.cuda(), flags.cuda(), nc, agnostic, 1.0)
    olb.backward()
    close(olb, lb.detach(), 1e-5, 1e-5)
    close(b1.grad, b0.grad, 1e-7, 1e-5)


@pytest.mark.parametrize("agnostic", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_regress_by_class_kernel(ops, dtype, agnostic):
    """bbox_head.py:409-436 + the label choice of cascade_roi_head.py:274-281 / :316-317 against the numpy oracle."""
    g = torch.Generator().manual_seed(5)
    n, nc = 700, 80
    means, stds = (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2)
    xy = torch.rand(n, 2, generator=g) * 500
    rois = torch.cat([xy, xy + torch.rand(n, 2, generator=g) * 200 + 2], 1)
    cls = torch.randn(n, nc + 1, generator=g).to(dtype)
    cls[:, nc] += 3                                               # background often the overall maximum: must be ignored
    bbox = torch.randn(n, 4 if agnostic else 4 * nc, generator=g).to(dtype)
    labels = torch.randint(0, nc + 1, (n,), generator=g)
    for lab in (labels, None):
        ref = CO.regress_by_class(rois.numpy(), None if lab is None else lab.numpy(), cls.float().numpy(), bbox.float().numpy(), nc,
                                  agnostic, means, stds, (480, 640))
        out = ops.regress_by_class(rois.cuda(), None if lab is None else lab.cuda(), cls.cuda(), bbox.cuda(), nc, agnostic, means,
                                   stds, (480, 640))
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=2e-3)
    assert float(out[:, 2].max()) <= 640 and float(out[:, 3].max()) <= 480


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("dtype,C", [(torch.float32, 256), (torch.bfloat16, 256), (torch.bfloat16, 96), (torch.float32, 36)])
def test_batch_norm_kernels(ops, dtype, C, relu):
    """csrc/batchnorm.hip (the SyncBN of ConvFCBBoxHead's ConvModules) against the float64 oracle: output, running
    statistics, dx / dgamma / dbeta; and eval mode against running statistics."""
    g = torch.Generator().manual_seed(2)
    N, H, W = 37, 7, 7
    x = (torch.randn(N, C, H, W, generator=g) * 1.7 + torch.randn(C, generator=g)[None, :, None, None]).to(dtype)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    dy = torch.randn(N, C, H, W, generator=g).to(dtype)
    rows = x.float().permute(0, 2, 3, 1).reshape(-1, C).numpy()
    y_ref, mean, var = CO.batch_norm_train(rows, gamma.numpy(), beta.numpy(), 1e-5, relu)
    dx_ref, dg_ref, db_ref = CO.batch_norm_train_bwd(rows, gamma.numpy(), beta.numpy(), dy.float().permute(0, 2, 3, 1).reshape(-1, C).numpy(),
                                                     1e-5, relu)
    # the oracle's formula == torch's own batch norm (known answer)
    yt = F.batch_norm(x.float(), None, None, gamma, beta, True, 0.1, 1e-5)
    yt = F.relu(yt) if relu else yt
    np.testing.assert_allclose(yt.permute(0, 2, 3, 1).reshape(-1, C).numpy(), y_ref, rtol=1e-4, atol=1e-4)
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gg, bg = gamma.cuda().requires_grad_(True), beta.cuda().requires_grad_(True)
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    y = ops.batch_norm(xg, gg, bg, rm, rv, True, 1e-5, 0.1, relu)
    y.backward(dy.cuda())
    tol = 1e-4 if dtype == torch.float32 else 2.0 ** -7
    R = rows.shape[0]
    np.testing.assert_allclose(y.detach().float().permute(0, 2, 3, 1).reshape(-1, C).cpu().numpy(), y_ref, rtol=tol, atol=tol * 4)
    np.testing.assert_allclose(rm.cpu().numpy(), 0.1 * mean, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rv.cpu().numpy(), 0.9 + 0.1 * var * R / (R - 1), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gg.grad.cpu().numpy(), dg_ref, rtol=1e-3, atol=2e-3 * np.abs(dg_ref).max())
    np.testing.assert_allclose(bg.grad.cpu().numpy(), db_ref, rtol=1e-3, atol=2e-3 * np.abs(db_ref).max())
    np.testing.assert_allclose(xg.grad.float().permute(0, 2, 3, 1).reshape(-1, C).cpu().numpy(), dx_ref, rtol=tol,
                               atol=tol * 4 * float(np.abs(dx_ref).max()))
    ye = ops.batch_norm(xg.detach(), gg.detach(), bg.detach(), rm, rv, False, 1e-5, 0.1, relu)
    yr = F.batch_norm(x.float(), rm.cpu(), rv.cpu(), gamma, beta, False, 0.1, 1e-5)
    yr = F.relu(yr) if relu else yr
    close(ye.float().cpu(), yr, tol * 4, tol)


def test_batched_nms_static_multi_equals_per_image(ops):
    """All images in one set of launches == batched_nms_static image by image (which is checked against the oracle)."""
    rng = np.random.RandomState(44)
    B, n, m = 3, 2500, 600
    xy = rng.rand(B, n, 2).astype(np.float32) * 500
    boxes = torch.from_numpy(np.concatenate([xy, xy + rng.rand(B, n, 2).astype(np.float32) * 90 + 1], 2)).cuda()
    scores = torch.from_numpy(rng.rand(B, n).astype(np.float32)).cuda()
    idxs = torch.from_numpy(rng.randint(0, 5, (B, n))).cuda()
    dets, valid = ops.batched_nms_static_multi(boxes, scores, idxs, 0.7, m)
    assert dets.shape == (B, m, 5) and valid.shape == (B, m)
    for i in range(B):
        d1, v1 = ops.batched_nms_static(boxes[i], scores[i], idxs[i], 0.7, m)
        assert torch.equal(valid[i], v1) and torch.equal(dets[i], d1)


@pytest.mark.parametrize("case", ["rpn", "dense", "tiny", "one_big_group", "early_stop"])
def test_batched_nms_grouped_scans_equal_the_single_scan(ops, case):
    """nms_sorted_batch_grouped (the suppression scans of the level / class ids side by side) == the single scan over the whole
    score-sorted list (nms_sorted_batch, itself checked against the oracle), bit for bit: kept boxes, their order, the validity mask
    and the max_num cut -- at the RPN's geometry (five levels of <= 2000 candidates, 1000 kept), with heavy overlap, with groups of
    very different sizes (an empty one included), and when max_num stops the scan early."""
    rng = np.random.RandomState({"rpn": 1, "dense": 2, "tiny": 3, "one_big_group": 4, "early_stop": 5}[case])
    if case == "rpn":
        B, sizes, m, span, wh = 2, [2000, 2000, 2000, 2000, 780], 1000, 1200.0, 220.0
    elif case == "dense":
        B, sizes, m, span, wh = 3, [700, 650, 90, 1200], 900, 120.0, 90.0          # most boxes suppressed
    elif case == "tiny":
        B, sizes, m, span, wh = 1, [3, 0, 70], 50, 60.0, 40.0
    elif case == "one_big_group":
        B, sizes, m, span, wh = 2, [4100, 5, 64, 0, 1, 129, 63, 2], 2000, 900.0, 100.0
    else:
        B, sizes, m, span, wh = 2, [1500, 1500, 1500], 40, 2000.0, 30.0             # almost nothing suppressed: the cut decides
    n = sum(sizes)
    ids = np.concatenate([np.full(k, g) for g, k in enumerate(sizes)])
    idxs = torch.from_numpy(np.stack([ids for _ in range(B)])).long().cuda()
    xy = rng.rand(B, n, 2).astype(np.float32) * span
    boxes = torch.from_numpy(np.concatenate([xy, xy + rng.rand(B, n, 2).astype(np.float32) * wh + 1], 2)).cuda()
    scores = torch.from_numpy(rng.rand(B, n).astype(np.float32))
    kt = (n - 1) // 7
    scores[:, 0:7 * kt:7] = scores[:, 1:7 * kt:7]                                  # ties: the stable order decides
    scores = scores.cuda()
    d0, v0 = ops.batched_nms_static_multi(boxes, scores, idxs, 0.7, m)
    d1, v1 = ops.batched_nms_static_multi(boxes, scores, idxs, 0.7, m, group_sizes=sizes)
    assert int(v0.sum()) > 0
    assert torch.equal(v0, v1) and torch.equal(d0, d1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rpn_flatten_kernel(ops, dtype):
    """det_rpn_flatten == slicing + concatenating the per-level head outputs with torch (anchor_head.py:474-486 order),
    values and gradients (pure data movement: exact)."""
    g = torch.Generator().manual_seed(9)
    B, A, CH = 2, 3, 16
    hws = [40 * 64, 20 * 32, 10 * 16, 5 * 8, 3 * 4]
    ys = [torch.randn(B, n, CH, generator=g).to(dtype).cuda().requires_grad_(True) for n in hws]
    ys2 = [y.detach().clone().requires_grad_(True) for y in ys]
    cls_ref = torch.cat([y[:, :, :A] for y in ys2], 1).reshape(B, -1)
    reg_ref = torch.cat([y[:, :, A:5 * A] for y in ys2], 1).reshape(B, -1, 4)
    cls, reg = ops.rpn_flatten(ys, A)
    assert torch.equal(cls, cls_ref) and torch.equal(reg, reg_ref)
    wc, wr = torch.randn_like(cls_ref), torch.randn_like(reg_ref)
    ((cls_ref * wc).sum() + (reg_ref * wr).sum()).backward()
    ((cls * wc).sum() + (reg * wr).sum()).backward()
    for a, b in zip(ys, ys2):
        assert torch.equal(a.grad, b.grad)


def test_gemm_plan_export_import_roundtrip(ops):
    """swin_gemm_plans_export / _import (the records data-parallel ranks exchange so that all of them run rank 0's hipBLASLt
    algorithm choices -- ddp.sync_gemm_plans): a plan exists after a call, re-importing the exported choices changes nothing,
    importing another candidate index changes exactly that plan and the GEMM still computes the same product."""
    import ctypes
    from swin_transformer_object_detection_amd import _lib, ddp
    from swin_transformer_object_detection_amd.ops import functional as Fn
    a = torch.randn(3000, 192, device="cuda").bfloat16()
    w = (torch.randn(576, 192, device="cuda") * 0.05).bfloat16()
    c0 = Fn.gemm_bf16(a, w)
    lib = _lib.lib()
    n = lib.swin_gemm_plans_export(None, 0)
    assert n >= 1
    buf = (ctypes.c_int64 * (6 * n))()
    assert lib.swin_gemm_plans_export(buf, n) == n
    recs = [list(buf[6 * i:6 * i + 6]) for i in range(n)]
    mine = [r for r in recs if r[:5] == [3000, 576, 192, 0, 0]]
    assert len(mine) == 1
    assert lib.swin_gemm_plans_import(buf, n) == 0                      # the same choices: nothing changes
    assert ddp.sync_gemm_plans() == 0                                    # one process: a no-op
    other = (ctypes.c_int64 * 6)(3000, 576, 192, 0, 0, 1 if mine[0][5] == 0 else 0)
    changed = lib.swin_gemm_plans_import(other, 1)
    assert changed in (0, 1)                                             # 0 only if the heuristic offered a single candidate
    c1 = Fn.gemm_bf16(a, w)
    torch.cuda.synchronize()
    assert float((c0.float() - c1.float()).abs().max()) <= 2.0 ** -7 * float(c0.float().abs().max())
    lib.swin_gemm_plans_import(buf, n)                                   # back to the tuned choice


def test_tail_reduce_equals_separate_reductions(ops):
    """csrc/tail_reduce.hip: LayerNorm parameter-gradient partial rows and the attention backward's bias-gradient slabs reduced in
    ONE launch, against torch column sums and against the two-kernel path (slab reduce into (nH,64,64), then the index-rule
    reduce of swin_transformer.py:105-110)."""
    import ctypes
    from swin_transformer_object_detection_amd import _lib
    from swin_transformer_object_detection_amd.ops import functional as Fn
    B, H, W, nH, shift = 2, 30, 45, 3, 3                       # padded grid: the pad-token part of the qkv bias gradient is exercised
    C = 32 * nH
    g = torch.Generator().manual_seed(11)
    qkv = dev(torch.randn(B, H * W, 3 * C, generator=g) * 0.7, torch.bfloat16)
    qb = dev(torch.randn(3 * C, generator=g) * 0.3)
    table = dev(torch.randn(169, nH, generator=g) * 0.5)
    dout = dev(torch.randn(B, H * W, C, generator=g), torch.bfloat16)
    bias_exp = ops.rel_bias_expand(table)
    out = torch.empty(B, H * W, C, device=qkv.device, dtype=qkv.dtype)
    nW = ((H + 6) // 7) * ((W + 6) // 7)
    lse = torch.empty(B * nW * nH, 64, device=qkv.device, dtype=torch.float32)
    scale = 32 ** -0.5
    Fn.call("swin_window_attn_fwd", Fn._p(qkv), Fn._p(qb), Fn._p(bias_exp), Fn._p(out), Fn._p(lse), B, H, W, C, nH, shift, scale,
            Fn.SWIN_BF16, Fn._s())
    ws_bytes = _lib.lib().swin_window_attn_bwd_workspace_bytes(B, H, W, nH, Fn.SWIN_BF16)
    ws = torch.zeros(ws_bytes // 4, device=qkv.device, dtype=torch.float32)
    dqkv = torch.empty_like(qkv)
    dbexp = torch.zeros_like(bias_exp)
    dpad_ref = torch.zeros(3 * C, device=qkv.device)
    Fn.call("swin_window_attn_bwd", Fn._p(qkv), Fn._p(qb), Fn._p(bias_exp), Fn._p(lse), Fn._p(dout), Fn._p(dqkv), Fn._p(dbexp),
            Fn._p(dpad_ref), Fn._p(ws), B, H, W, C, nH, shift, scale, Fn.SWIN_BF16, Fn._s())
    dtable_ref = torch.zeros(169, nH, device=qkv.device)
    Fn.call("swin_rel_bias_reduce", Fn._p(dbexp), Fn._p(dtable_ref), nH, Fn._s())
    torch.cuda.synchronize()
    slab = 64 * 64 + 3 * 32 + 32
    blocks = ws.numel() // (4 * slab)                          # the workspace is sized for 4 waves-pairs per block; 3 heads use 3
    assert blocks * 4 * slab == ws.numel()
    n_slabs = blocks * 3
    # two LayerNorm-style problems of different widths next to the bias problem
    parts = [dev(torch.randn(rows, 2 * c, generator=g)) for rows, c in ((768, 96), (37, 192))]
    dg = [torch.full((p.shape[1] // 2,), 0.5, device=qkv.device) for p in parts]      # accumulated INTO (+=)
    db = [torch.full((p.shape[1] // 2,), -0.25, device=qkv.device) for p in parts]
    dtable = torch.ones(169, nH, device=qkv.device)
    dpad = torch.ones(3 * C, device=qkv.device)
    srcs = parts + [ws]
    d0 = dg + [dtable]
    d1 = db + [dpad]
    kinds = [0, 0, 1]
    rows = [p.shape[0] for p in parts] + [n_slabs]
    cols = [p.shape[1] for p in parts] + [slab]
    a0 = [p.shape[1] // 2 for p in parts] + [nH]
    a1 = [0, 0, C]
    n = len(kinds)
    ia = lambda v: (ctypes.c_int * n)(*v)
    pa = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    Fn.call("swin_tail_reduce", ia(kinds), pa(srcs), pa(d0), pa(d1), ia(rows), ia(cols), ia(a0), ia(a1), n, Fn._s())
    torch.cuda.synchronize()
    for p, a, b in zip(parts, dg, db):
        c = p.shape[1] // 2
        s = p.double().sum(0)
        close(a, (s[:c] + 0.5).float(), 1e-3, 1e-5, msg="dgamma")
        close(b, (s[c:] - 0.25).float(), 1e-3, 1e-5, msg="dbeta")
    assert float(dtable_ref.abs().max()) > 0 and float(dpad_ref.abs().max()) > 0
    close(dtable - 1.0, dtable_ref, 1e-4 * float(dtable_ref.abs().max()) + 1e-5, 1e-5, msg="dtable")
    close(dpad - 1.0, dpad_ref, 1e-4 * float(dpad_ref.abs().max()) + 1e-5, 1e-5, msg="dbias_pad")


@pytest.mark.parametrize("N,H,W,Cin,Cout,nt", [
    (1, 9, 11, 64, 128, 2),          # one ragged tile (99 pixels), W smaller than the halo
    (3, 7, 9, 128, 192, 2),          # three images inside one tile: the y-border rule at every image boundary; Cout not a tile multiple
    (2, 23, 37, 32, 256, 4),         # one channel block, 256-wide tile, tile boundaries in the middle of rows
    (2, 50, 80, 256, 256, 4),        # several tiles, 8 channel blocks (the segment rotation wraps twice)
    (1, 16, 2, 96, 8, 2),            # W = 2: both x borders in every row; Cout = 8
])
def test_conv3x3_halo_form(ops, N, H, W, Cin, Cout, nt):
    """csrc/conv_halo.hip through its own entry point (conv3x3_halo_nhwc_bf16) against fp32 F.conv2d of the same bf16 operands
    (fpn.py:195-197 semantics: 3x3, pad 1), plus the ReLU and gate epilogues against the plain result bit for bit."""
    from swin_transformer_object_detection_amd.ops import functional as Fn
    g = torch.Generator().manual_seed(N * H + Cin + W)
    x = torch.randn(N, H, W, Cin, generator=g).cuda().bfloat16()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (2.0 / (9 * Cin)) ** 0.5).cuda().bfloat16()
    b = (torch.randn(Cout, generator=g) * 0.1).cuda()
    gate = torch.randn(N, H, W, Cout, generator=g).cuda().bfloat16()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, padding=1).permute(0, 2, 3, 1)
    y = torch.full((N, H, W, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
    Fn.call("conv3x3_halo_nhwc_bf16", Fn._p(x), Fn._p(w), Fn._p(b), None, Fn._p(y), N, H, W, Cin, Cout, 0, nt, Fn._s())
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y.float()).all())
    close(y, ref, bf16_tol(ref, 2), msg="y")
    yr = torch.empty_like(y)
    yg = torch.empty_like(y)
    Fn.call("conv3x3_halo_nhwc_bf16", Fn._p(x), Fn._p(w), Fn._p(b), None, Fn._p(yr), N, H, W, Cin, Cout, 1, nt, Fn._s())
    Fn.call("conv3x3_halo_nhwc_bf16", Fn._p(x), Fn._p(w), Fn._p(b), Fn._p(gate), Fn._p(yg), N, H, W, Cin, Cout, 0, nt, Fn._s())
    torch.cuda.synchronize()
    assert torch.equal(yr, torch.relu(y))
    assert torch.equal(yg, torch.where(gate.float() > 0, y, torch.zeros_like(y)))
    # no bias
    y0 = torch.empty_like(y)
    Fn.call("conv3x3_halo_nhwc_bf16", Fn._p(x), Fn._p(w), None, None, Fn._p(y0), N, H, W, Cin, Cout, 0, nt, Fn._s())
    torch.cuda.synchronize()
    close(y0, ref - b, bf16_tol(ref, 2), msg="y (no bias)")


def test_conv3x3_halo_rejects_unsupported_shapes(ops):
    from swin_transformer_object_detection_amd import _lib
    from swin_transformer_object_detection_amd.ops import functional as Fn
    x = torch.zeros(1, 8, 8, 48, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(16, 3, 3, 48, device="cuda", dtype=torch.bfloat16)
    y = torch.zeros(1, 8, 8, 16, device="cuda", dtype=torch.bfloat16)
    rc = _lib.lib().conv3x3_halo_nhwc_bf16(Fn._p(x), Fn._p(w), None, None, Fn._p(y), 1, 8, 8, 48, 16, 0, 2, Fn._s())     # Cin % 32 != 0
    assert rc == 2                      # SWIN_ERR_UNSUPPORTED (include/swin_hip.h)
    w2 = torch.zeros(128, 3, 3, 64, device="cuda", dtype=torch.bfloat16)
    x2 = torch.zeros(1, 8, 8, 64, device="cuda", dtype=torch.bfloat16)
    y2 = torch.zeros(1, 8, 8, 128, device="cuda", dtype=torch.bfloat16)
    rc = _lib.lib().conv3x3_halo_nhwc_bf16(Fn._p(x2), Fn._p(w2), None, None, Fn._p(y2), 1, 8, 8, 64, 128, 0, 4, Fn._s())  # nt = 4 needs Cout % 256 == 0
    assert rc == 2                      # SWIN_ERR_UNSUPPORTED (include/swin_hip.h)


@pytest.mark.parametrize("gdt,odt", [(torch.bfloat16, torch.bfloat16), (torch.float32, torch.float32)])
def test_roi_align_gather_backward_equals_scatter_backward(ops, gdt, odt):
    """roi_align_multilevel_bwd_gather (tiles own their gradient: no atomics, no zero fill, no cast) against the float-atomic
    scatter form on the same inputs: two RoI sets (7x7 and 14x14) over a 4-level pyramid with map sizes that are not tile multiples,
    100 RoIs stacked on one spot (single_level_roi_extractor.py:93-97 with many proposals on one object), RoIs partly and wholly
    outside the image, skipped slots (lvl < 0), degenerate (negative-width) boxes.  Run twice: the workspace is reusable, and the
    result is bit-identical (sorted lists)."""
    import ctypes
    from swin_transformer_object_detection_amd import _lib
    from swin_transformer_object_detection_amd.ops import functional as Fn
    N, C = 2, 256
    shapes = [(50, 80), (25, 40), (13, 20), (7, 10)]
    strides = [4, 8, 16, 32]
    g = torch.Generator().manual_seed(5)

    def make(K, out):
        r = torch.rand(K, 5, generator=g)
        r[:, 0] = (torch.arange(K) % N).float()
        wh = torch.exp(torch.rand(K, 2, generator=g) * 5.0 + 1.5)                    # 4.5 .. 665 px: every level gets RoIs
        r[:, 1:3] = r[:, 1:3] * torch.tensor([320., 200.]) - 20.0                   # some start outside
        r[:, 3:] = r[:, 1:3] + wh
        r[:100, 1:] = torch.tensor([100., 60., 140., 95.]) + torch.rand(100, 4, generator=g) * 2.0   # stacked on one object
        r[100, 1:] = torch.tensor([-500., -500., -400., -400.])                      # wholly outside
        r[101, 1:] = torch.tensor([50., 50., 40., 45.])                              # negative width / height
        scale = torch.sqrt((r[:, 3] - r[:, 1]).clamp(min=1) * (r[:, 4] - r[:, 2]).clamp(min=1))
        lv = torch.floor(torch.log2(scale / 56 + 1e-6)).clamp(0, 3).int()
        lv[102:110] = -1
        go = torch.randn(K, out, out, C, generator=g)
        return r.cuda(), lv.cuda(), go.cuda().to(gdt)
    sets = [make(300, 7) + (7,), make(150, 14) + (14,)]
    n = len(shapes)
    Hs = (ctypes.c_int * n)(*[s[0] for s in shapes]); Ws = (ctypes.c_int * n)(*[s[1] for s in shapes])
    sc = (ctypes.c_float * n)(*[1.0 / s for s in strides])
    # scatter form: fp32 accumulators, zeroed, then cast
    acc = [torch.zeros(N, h, w, C, device="cuda") for h, w in shapes]
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in acc])
    for r, lv, go, out in sets:
        Fn.call("roi_align_multilevel_bwd", ptrs, Hs, Ws, sc, n, Fn._p(go), Fn._p(r), Fn._p(lv), C, r.shape[0], out, out, 0, 1,
                Fn.SWIN_F32 if gdt == torch.float32 else Fn.SWIN_BF16, Fn._s())
    ref = [a.to(odt) for a in acc]
    # gather form
    ktot = sum(s[0].shape[0] for s in sets)
    nb = int(_lib.lib().roi_align_gather_workspace_bytes(Hs, Ws, n, N, ktot))
    assert nb > 0
    ws = torch.zeros(nb // 4 + 1, device="cuda", dtype=torch.int32)
    m = len(sets)
    code = lambda d: Fn.SWIN_F32 if d == torch.float32 else Fn.SWIN_BF16      # noqa: E731
    runs = []
    for _ in range(2):
        outs = [torch.full((N, h, w, C), float("nan"), device="cuda", dtype=odt) for h, w in shapes]
        op = (ctypes.c_void_p * n)(*[t.data_ptr() for t in outs])
        gp = (ctypes.c_void_p * m)(*[s[2].data_ptr() for s in sets])
        rp = (ctypes.c_void_p * m)(*[s[0].data_ptr() for s in sets])
        lp = (ctypes.c_void_p * m)(*[s[1].data_ptr() for s in sets])
        Ks = (ctypes.c_int * m)(*[s[0].shape[0] for s in sets])
        ps = (ctypes.c_int * m)(*[s[3] for s in sets])
        Fn.call("roi_align_multilevel_bwd_gather", op, Hs, Ws, sc, n, N, m, gp, rp, lp, Ks, ps, ps, C, 0, 1, code(gdt), code(odt),
                Fn._p(ws), nb, Fn._s())
        torch.cuda.synchronize()
        runs.append(outs)
    for l, (a, b) in enumerate(zip(runs[0], ref)):
        assert bool(torch.isfinite(a.float()).all()), l
        assert float(b.float().abs().max()) > 0
        tol = bf16_tol(b, 1.5) if odt == torch.bfloat16 else 2e-4 * float(b.abs().max())
        close(a, b, tol, msg=f"level {l}")
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b)
