"""Token-stationary fused MLP (csrc/ts_mlp.hip) through the C ABI against the oracle: Mlp.forward of the reference
(swin_transformer.py:32-38: fc1 -> nn.GELU (exact erf) -> fc2) evaluated in fp32 on the CPU on the same bf16-rounded
operands.  Tolerance: the kernel keeps the hidden activation in fp32 until ONE bf16 rounding feeds fc2 and rounds the output
once: a few bf16 ulps (2^-8 relative) of the output scale; stated per assertion."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fn():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from swin_transformer_object_detection_amd.ops import functional
    return functional


def _case(T, C, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(T, C, generator=g).bfloat16()
    w1 = (torch.randn(4 * C, C, generator=g) * (1.0 / C) ** 0.5).bfloat16()
    w2 = (torch.randn(C, 4 * C, generator=g) * (1.0 / (4 * C)) ** 0.5).bfloat16()
    b1 = torch.randn(4 * C, generator=g) * 0.3
    b2 = torch.randn(C, generator=g) * 0.3
    return x, w1, b1, w2, b2


def _tol(ref, ulps):
    return float(ulps * 2.0 ** -8 * max(ref.abs().max().item(), 1e-3))


@pytest.mark.parametrize("T,C", [(1, 96), (33, 96), (256, 96), (1000, 96), (128000, 96), (77, 192), (1024, 192), (32000, 192)])
def test_mlp_fused_forward_vs_oracle(fn, T, C):
    x, w1, b1, w2, b2 = _case(T, C, T + C)
    ref = F.linear(F.gelu(F.linear(x.float(), w1.float(), b1)), w2.float(), b2)       # swin_transformer.py:32-38
    y = fn.mlp_fwd_raw(x.cuda(), w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda())
    torch.cuda.synchronize()
    assert y.shape == (T, C) and y.dtype == torch.bfloat16
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.numpy(), rtol=0, atol=_tol(ref, 3))
    # the hidden activation's single bf16 rounding is the dominant error; the mean error must be far below one ulp
    assert (y.float().cpu() - ref).abs().mean().item() < _tol(ref, 0.25)


def test_mlp_fused_gelu_tails(fn):
    """large |fc1| outputs: gelu(x) -> x and -> -0 (the erfc form must not lose the negative tail to cancellation)"""
    C = 96
    x = torch.zeros(64, C).bfloat16()
    x[:, 0] = torch.linspace(-12, 12, 64).bfloat16()
    w1 = torch.zeros(4 * C, C).bfloat16()
    w1[:, 0] = 1.0
    w2 = torch.zeros(C, 4 * C).bfloat16()
    w2[0, 5] = 1.0
    b1, b2 = torch.zeros(4 * C), torch.zeros(C)
    ref = F.linear(F.gelu(F.linear(x.float(), w1.float(), b1)), w2.float(), b2)
    y = fn.mlp_fwd_raw(x.cuda(), w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda()).float().cpu()
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=2.0 ** -7, atol=1e-6)


@pytest.mark.parametrize("T,C", [(1, 96), (70, 96), (1000, 96), (128000, 96), (77, 192), (1024, 192), (32000, 192)])
def test_mlp_fused_backward_vs_oracle(fn, T, C):
    """dx, and the two operands handed to the weight-gradient GEMMs (h = gelu(fc1 x + b1), dhpre = (dy W2) * gelu'), against
    fp32 autograd of the reference's Mlp on the CPU.  The kernel rounds dhpre to bf16 once before the last product (as the
    unfused chain does), so dx carries ~2 bf16 ulps of its scale."""
    x, w1, b1, w2, b2 = _case(T, C, 7 * T + C)
    g = torch.Generator().manual_seed(T)
    dy = (torch.randn(T, C, generator=g) * 0.5).bfloat16()
    x0 = x.float().requires_grad_(True)
    hpre = F.linear(x0, w1.float(), b1)
    hpre.retain_grad()
    h = F.gelu(hpre)
    y = F.linear(h, w2.float(), b2)
    y.backward(dy.float())
    dx, hh, dhp = fn.mlp_bwd_raw(x.cuda(), dy.cuda(), w1.cuda(), b1.cuda(), w2.cuda())
    torch.cuda.synchronize()
    assert dx.shape == (T, C) and hh.shape == dhp.shape == (T, 4 * C)
    np.testing.assert_allclose(hh.float().cpu().numpy(), h.detach().numpy(), rtol=0, atol=_tol(h.detach(), 1.01))
    np.testing.assert_allclose(dhp.float().cpu().numpy(), hpre.grad.numpy(), rtol=0, atol=_tol(hpre.grad, 2))
    np.testing.assert_allclose(dx.float().cpu().numpy(), x0.grad.numpy(), rtol=0, atol=_tol(x0.grad, 3))
    assert (dx.float().cpu() - x0.grad).abs().mean().item() < _tol(x0.grad, 0.25)
    # the weight gradients the step forms from these operands (wgrad_linear_bf16) equal autograd's
    dw2 = dy.float().t() @ hh.float().cpu()
    dw1 = dhp.float().cpu().t() @ x.float()
    w1g, w2g = torch.autograd.grad(F.linear(F.gelu(F.linear(x.float(), w1.float().requires_grad_(True), b1)),
                                            w2.float().requires_grad_(True), b2), [], allow_unused=True) if False else (None, None)
    w1f, w2f = w1.float().requires_grad_(True), w2.float().requires_grad_(True)
    F.linear(F.gelu(F.linear(x.float(), w1f, b1)), w2f, b2).backward(dy.float())
    np.testing.assert_allclose(dw2.numpy(), w2f.grad.numpy(), rtol=0, atol=2e-2 * float(w2f.grad.abs().max()) + 1e-3)
    np.testing.assert_allclose(dw1.numpy(), w1f.grad.numpy(), rtol=0, atol=2e-2 * float(w1f.grad.abs().max()) + 1e-3)


# ---- every other width (Swin-T/S stages 3-4: 384, 768; Swin-B: 128 ... 1024): the GELU lives in the epilogues of the hand-written
# GEMM (csrc/conv_gemm.hip: swin_linear_gelu_hip_bf16 / swin_linear_dgelu_hip_bf16), the transposed fc2 weight comes from
# linear_t_layout_multi
@pytest.mark.parametrize("T,C", [(8000, 384), (2000, 768), (1999, 128), (4097, 256), (777, 512), (130, 1024), (1, 384)])
def test_mlp_gelu_epilogue_gemms_vs_oracle(fn, T, C):
    x, w1, b1, w2, b2 = _case(T, C, T + C)
    g = torch.Generator().manual_seed(T)
    dy = (torch.randn(T, C, generator=g) * 0.5).bfloat16()
    # oracle: Mlp.forward / backward in fp32 on the same bf16 operands, the fc1 output rounded to bf16 as under autocast
    hpre_ref = F.linear(x.float(), w1.float()).bfloat16().float()
    xx = (hpre_ref + b1).requires_grad_(True)
    h_ref = F.gelu(xx)
    dh_ref = (dy.float() @ w2.float()).bfloat16().float()                    # data gradient through fc2, a bf16 tensor under autocast
    (h_ref * dh_ref).sum().backward()
    dhpre_ref = xx.grad
    xc, w1c, b1c, w2c, dyc = x.cuda(), w1.cuda(), b1.cuda(), w2.cuda(), dy.cuda()
    hpre = torch.empty(T, 4 * C, device="cuda", dtype=torch.bfloat16)
    h = torch.empty_like(hpre)
    fn.call("swin_linear_gelu_hip_bf16", fn._p(xc), fn._p(w1c), fn._p(b1c), fn._p(hpre), fn._p(h), T, 4 * C, C, fn._s())
    w2t = torch.empty(4 * C, C, device="cuda", dtype=torch.bfloat16)
    fn.linear_t_layout_multi([w2c], [w2t])
    assert torch.equal(w2t, w2c.t().contiguous())
    dhpre = torch.empty_like(hpre)
    fn.call("swin_linear_dgelu_hip_bf16", fn._p(dyc), fn._p(w2t), fn._p(hpre), fn._p(b1c), fn._p(dhpre), T, 4 * C, C, fn._s())
    torch.cuda.synchronize()
    np.testing.assert_allclose(hpre.float().cpu().numpy(), hpre_ref.numpy(), rtol=0, atol=_tol(hpre_ref, 1.01))
    np.testing.assert_allclose(h.float().cpu().numpy(), h_ref.detach().numpy(), rtol=0, atol=_tol(h_ref.detach(), 2))
    np.testing.assert_allclose(dhpre.float().cpu().numpy(), dhpre_ref.numpy(), rtol=0, atol=_tol(dhpre_ref, 2.5))
    # and against the three-launch chain it replaces: the same bits (the product is rounded to bf16 before the activation in both)
    ws = torch.empty(fn._lib.lib().swin_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
    hp2 = torch.empty_like(hpre)
    fn.call("swin_gemm_bf16", fn._p(xc), fn._p(w1c), None, fn._p(hp2), T, 4 * C, C, 0, fn._p(ws), fn._s())
    h2 = torch.empty_like(hpre)
    fn.call("swin_bias_gelu_fwd", fn._p(hp2), fn._p(b1c), fn._p(h2), T, 4 * C, fn.SWIN_BF16, fn._s())
    torch.cuda.synchronize()
    assert float((hp2.float() - hpre.float()).abs().max()) <= _tol(hpre_ref, 1.01)       # two GEMM kernels: at most an ulp apart
    same = hp2 == hpre
    assert bool((h2[same] == h[same]).all())                                               # same pre-activation -> same GELU bits


@pytest.mark.parametrize("C,T", [(384, 8000), (384, 37), (384, 2 * 50 * 80 + 3), (192, 32000), (96, 700)])
def test_token_stationary_linear_chunk_split_and_stage3_width(C, T):
    """swin_ts_linear_bf16 with the output chunks dealt over blockIdx.y (fewer token blocks than CUs: stage 2 / stage 3 of the
    BASELINE geometry) and at C = 384 (stage 3: qkv, proj and the proj data gradient of swin_block_fwd / _bwd), against swin_gemm_bf16:
    same rounding points, at most an ulp apart."""
    from swin_transformer_object_detection_amd import _lib
    from swin_transformer_object_detection_amd.ops import functional as Fn
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    H = _lib.half_dtype()
    g = torch.Generator().manual_seed(C + T)
    dev = torch.device("cuda", 0)
    x = torch.randn(T, C, generator=g).to(dev, H)
    ws = torch.empty(_lib.lib().swin_gemm_workspace_bytes(), dtype=torch.uint8, device=dev)
    for N, relu, with_bias in ((3 * C, 0, True), (C, 0, True), (C, 0, False), (256, 1, True), (4 * C, 0, True)):
        w = (torch.randn(N, C, generator=g) * C ** -0.5).to(dev, H)
        b = (torch.randn(N, generator=g) * 0.1).to(dev, H) if with_bias else None
        y0 = torch.empty(T, N, device=dev, dtype=H)
        y1 = torch.full_like(y0, float("nan"))
        Fn.call("swin_gemm_bf16", Fn._p(x), Fn._p(w), Fn._p(b), Fn._p(y0), T, N, C, 0, Fn._p(ws), Fn._s())
        Fn.call("swin_ts_linear_bf16", Fn._p(x), Fn._p(w), Fn._p(b), Fn._p(y1), T, N, C, relu, Fn._s())
        torch.cuda.synchronize()
        assert bool(torch.isfinite(y1.float()).all()), N
        ref = torch.relu(y0) if relu else y0
        a, r = y1.float(), ref.float()
        tol = 2.0 ** (-10 if H == torch.float16 else -7) * r.abs().clamp(min=2.0 ** -6)
        bad = (a - r).abs() > tol
        assert float(bad.float().mean()) <= 2e-3 and float(((a - r).abs() / tol).max()) <= 2.01, (N, int(bad.sum()))


@pytest.mark.parametrize("C", [96, 128, 192, 256])
@pytest.mark.parametrize("T", [1, 37, 1000, 2 * 25 * 40 + 5])
def test_token_stationary_qkv_and_proj_ln_equal_the_library_chain(C, T):
    """csrc/ts_linear.hip against the launches it replaces in swin_block_fwd: swin_gemm_bf16 for qkv (swin_transformer.py:129) and
    swin_gemm_bf16 + swin_add_layernorm_fwd for proj / residual / DropPath scale / norm2 (:150-151, :252-253).  The rounding points
    are the same, the fp32 summation order is not: a 16-bit output may differ by one ulp where the sum sits on a rounding boundary."""
    import ctypes
    from swin_transformer_object_detection_amd import _lib
    from swin_transformer_object_detection_amd.ops import functional as Fn
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    H = _lib.half_dtype()
    g = torch.Generator().manual_seed(C + T)
    dev = torch.device("cuda", 0)
    x = torch.randn(T, C, generator=g).to(dev, H)
    o = torch.randn(T, C, generator=g).to(dev, H)
    n1 = torch.randn(T, C, generator=g).to(dev, H)
    wqkv = (torch.randn(3 * C, C, generator=g) * C ** -0.5).to(dev, H)
    bqkv = (torch.randn(3 * C, generator=g) * 0.1).to(dev, H)
    wproj = (torch.randn(C, C, generator=g) * C ** -0.5).to(dev, H)
    bproj = (torch.randn(C, generator=g) * 0.1).to(dev, H)
    gamma = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    ws = torch.empty(_lib.lib().swin_gemm_workspace_bytes(), dtype=torch.uint8, device=dev)

    def ulp_close(a, b, what):
        a, b = a.float(), b.float()
        tol = 2.0 ** (-10 if H == torch.float16 else -7) * b.abs().clamp(min=2.0 ** -6)       # one ulp of the value (floor for tiny values)
        bad = (a - b).abs() > tol
        assert float(bad.float().mean()) <= 2e-3 and float(((a - b).abs() / tol).max()) <= 2.01, (what, int(bad.sum()))
    # qkv
    for bias in (bqkv, None):
        q0 = torch.empty(T, 3 * C, device=dev, dtype=H)
        q1 = torch.full_like(q0, float("nan"))
        Fn.call("swin_gemm_bf16", Fn._p(n1), Fn._p(wqkv), Fn._p(bias), Fn._p(q0), T, 3 * C, C, 0, Fn._p(ws), Fn._s())
        Fn.call("swin_ts_linear_bf16", Fn._p(n1), Fn._p(wqkv), Fn._p(bias), Fn._p(q1), T, 3 * C, C, 0, Fn._s())
        torch.cuda.synchronize()
        assert bool(torch.isfinite(q1.float()).all())
        ulp_close(q1, q0, "qkv")
    # N = 256 (an FPN lateral: the 64-row chunk variant) with the ReLU epilogue
    wl = (torch.randn(256, C, generator=g) * C ** -0.5).to(dev, H)
    bl = (torch.randn(256, generator=g) * 0.1).to(dev, H)
    l0 = torch.empty(T, 256, device=dev, dtype=H)
    l1 = torch.full_like(l0, float("nan"))
    Fn.call("swin_gemm_bf16", Fn._p(n1), Fn._p(wl), Fn._p(bl), Fn._p(l0), T, 256, C, 0, Fn._p(ws), Fn._s())
    Fn.call("swin_ts_linear_bf16", Fn._p(n1), Fn._p(wl), Fn._p(bl), Fn._p(l1), T, 256, C, 1, Fn._s())
    torch.cuda.synchronize()
    ulp_close(l1, torch.relu(l0), "lateral + relu")
    # proj + residual + norm2, with and without a DropPath scale (two samples of unequal length: the row -> sample map)
    L = max(T // 2, 1)
    for dp in (None, torch.tensor([1.25, 0.0, 0.5], device=dev)[: (T + L - 1) // L]):
        y = torch.empty(T, C, device=dev, dtype=H)
        x1a, n2a = torch.empty_like(x), torch.empty_like(x)
        ma, ra = torch.empty(T, device=dev), torch.empty(T, device=dev)
        Fn.call("swin_gemm_bf16", Fn._p(o), Fn._p(wproj), Fn._p(bproj), Fn._p(y), T, C, C, 0, Fn._p(ws), Fn._s())
        Fn.call("swin_add_layernorm_fwd", Fn._p(x), Fn._p(y), Fn._p(dp), L, Fn._p(gamma), Fn._p(beta), Fn._p(x1a), Fn._p(n2a), Fn._p(ma), Fn._p(ra),
                T, C, 1e-5, Fn.SWIN_BF16, Fn._s())
        x1b, n2b = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
        mb, rb = torch.empty(T, device=dev), torch.empty(T, device=dev)
        Fn.call("swin_ts_proj_add_ln_bf16", Fn._p(o), Fn._p(wproj), Fn._p(bproj), Fn._p(x), Fn._p(dp), L, Fn._p(gamma), Fn._p(beta), Fn._p(x1b),
                Fn._p(n2b), Fn._p(mb), Fn._p(rb), T, C, 1e-5, Fn._s())
        torch.cuda.synchronize()
        assert bool(torch.isfinite(n2b.float()).all())
        # x1 = x + dp * y: a one-ulp difference of y arrives scaled by dp and need not be small against x1 itself
        eps16 = 2.0 ** (-10 if H == torch.float16 else -7)
        tol1 = eps16 * (x1a.float().abs() + 1.25 * y.float().abs() + 2.0 ** -6)
        d1 = (x1b.float() - x1a.float()).abs()
        assert float((d1 > tol1).float().mean()) == 0.0 and float((d1 > 0).float().mean()) <= 2e-3, int((d1 > tol1).sum())
        assert float((mb - ma).abs().max()) <= 2e-3 and float(((rb - ra).abs() / ra).max()) <= 2e-3
        # n2 follows x1: compare through the LayerNorm of the fused kernel's own x1 where x1 agrees, loosely elsewhere
        assert float((n2b.float() - n2a.float()).abs().max()) <= 0.06 * float(n2a.float().abs().max())
        same = (x1b == x1a).all(dim=1)
        if bool(same.any()):
            ulp_close(n2b[same], n2a[same], "n2")


@pytest.mark.parametrize("C", [96, 192])
@pytest.mark.parametrize("T,with_norm", [(1, True), (1000, True), (2 * 25 * 40 + 5, False)])
def test_fused_mlp_with_residual_and_next_norm_equals_the_two_launch_chain(C, T, with_norm):
    """swin_mlp_add_ln_fwd_bf16 (the fused MLP with x2 = x1 + dp * y and the next LayerNorm in its epilogue) against
    swin_mlp_fwd_bf16 followed by swin_add_layernorm_fwd: the MLP arithmetic is the same code, so y and x2 must agree bit for bit;
    the norm sees another summation order for its statistics (one ulp on isolated outputs)."""
    from swin_transformer_object_detection_amd import _lib
    from swin_transformer_object_detection_amd.ops import functional as Fn
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    H = _lib.half_dtype()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3 * C + T)
    n2 = torch.randn(T, C, generator=g).to(dev, H)
    x1 = torch.randn(T, C, generator=g).to(dev, H)
    w1 = (torch.randn(4 * C, C, generator=g) * C ** -0.5).to(dev, H)
    w2 = (torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5).to(dev, H)
    b1 = (torch.randn(4 * C, generator=g) * 0.1).to(dev)
    b2 = (torch.randn(C, generator=g) * 0.1).to(dev)
    gamma = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    L = max(T // 2, 1)
    dp = torch.tensor([1.25, 0.5, 0.0], device=dev)[: (T + L - 1) // L]
    y = torch.empty(T, C, device=dev, dtype=H)
    Fn.call("swin_mlp_fwd_bf16", Fn._p(n2), Fn._p(w1), Fn._p(b1), Fn._p(w2), Fn._p(b2), Fn._p(y), T, C, Fn._s())
    x2a, nna = torch.empty_like(x1), torch.empty_like(x1)
    ma, ra = torch.empty(T, device=dev), torch.empty(T, device=dev)
    if with_norm:
        Fn.call("swin_add_layernorm_fwd", Fn._p(x1), Fn._p(y), Fn._p(dp), L, Fn._p(gamma), Fn._p(beta), Fn._p(x2a), Fn._p(nna), Fn._p(ma), Fn._p(ra),
                T, C, 1e-5, Fn.SWIN_BF16, Fn._s())
    else:
        Fn.call("swin_add_layernorm_fwd", Fn._p(x1), Fn._p(y), Fn._p(dp), L, None, None, Fn._p(x2a), None, None, None, T, C, 1e-5, Fn.SWIN_BF16, Fn._s())
    x2b, nnb = torch.full_like(x1, float("nan")), torch.full_like(x1, float("nan"))
    mb, rb = torch.empty(T, device=dev), torch.empty(T, device=dev)
    Fn.call("swin_mlp_add_ln_fwd_bf16", Fn._p(n2), Fn._p(w1), Fn._p(b1), Fn._p(w2), Fn._p(b2), Fn._p(x1), Fn._p(dp), L,
            Fn._p(gamma) if with_norm else None, Fn._p(beta) if with_norm else None, Fn._p(x2b), Fn._p(nnb) if with_norm else None,
            Fn._p(mb) if with_norm else None, Fn._p(rb) if with_norm else None, T, C, 1e-5, Fn._s())
    torch.cuda.synchronize()
    assert torch.equal(x2b, x2a)
    if with_norm:
        assert float((mb - ma).abs().max()) <= 1e-5 and float(((rb - ra).abs() / ra).max()) <= 1e-5
        eps16 = 2.0 ** (-10 if H == torch.float16 else -7)
        d = (nnb.float() - nna.float()).abs()
        assert float((d > eps16 * (nna.float().abs() + 2.0 ** -6)).float().mean()) == 0.0
        assert float((d > 0).float().mean()) <= 2e-3


@pytest.mark.parametrize("C", [96, 192])
@pytest.mark.parametrize("T,with_dres,with_dp", [(1, True, True), (777, True, False), (2 * 25 * 40 + 5, False, True)])
def test_fused_mlp_backward_with_norm2_backward_equals_the_two_launch_chain(C, T, with_dres, with_dp):
    """swin_mlp_ln_bwd_bf16 (fused MLP backward with the backward of norm2 and of the first residual in its epilogue) against
    swin_mlp_bwd_bf16 followed by swin_layernorm_bwd: h / dhpre bit-equal (same code); dx, dy and the [dgamma | dbeta] sums equal up
    to the summation order of the row statistics / column sums."""
    import ctypes
    from swin_transformer_object_detection_amd import _lib
    from swin_transformer_object_detection_amd.ops import functional as Fn
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    H = _lib.half_dtype()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5 * C + T)
    x1 = torch.randn(T, C, generator=g).to(dev, H)
    gamma = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    n2 = torch.empty_like(x1); mean = torch.empty(T, device=dev); rstd = torch.empty(T, device=dev)
    Fn.call("swin_layernorm_fwd", Fn._p(x1), Fn._p(gamma), Fn._p(beta), Fn._p(n2), Fn._p(mean), Fn._p(rstd), T, C, 1e-5, Fn.SWIN_BF16, Fn._s())
    dy2 = (torch.randn(T, C, generator=g) * 0.1).to(dev, H)
    dres = (torch.randn(T, C, generator=g) * 0.1).to(dev, H) if with_dres else None
    w1 = (torch.randn(4 * C, C, generator=g) * C ** -0.5).to(dev, H)
    w2 = (torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5).to(dev, H)
    b1 = (torch.randn(4 * C, generator=g) * 0.1).to(dev)
    L = max(T // 2, 1)
    dp = torch.tensor([1.25, 0.5, 0.0], device=dev)[: (T + L - 1) // L] if with_dp else None
    # chain
    dn2 = torch.empty_like(x1); ha = torch.empty(T, 4 * C, device=dev, dtype=H); da = torch.empty_like(ha)
    Fn.call("swin_mlp_bwd_bf16", Fn._p(n2), Fn._p(dy2), Fn._p(w1), Fn._p(b1), Fn._p(w2), Fn._p(dn2), Fn._p(ha), Fn._p(da), T, C, Fn._s())
    dxa = torch.empty_like(x1); dya = torch.empty_like(x1)
    dga, dba = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ws = Fn._ln_ws(T, C, x1)
    Fn.call("swin_layernorm_bwd", Fn._p(dn2), Fn._p(x1), Fn._p(gamma), Fn._p(mean), Fn._p(rstd), Fn._p(dres), Fn._p(dxa), Fn._p(dya), Fn._p(dp), L,
            Fn._p(dga), Fn._p(dba), T, C, Fn.SWIN_BF16, Fn._p(ws), Fn._s())
    # fused
    rows = int(_lib.lib().swin_mlp_ln_bwd_partial_rows(T, C))
    assert rows >= 1
    part = torch.full((rows, 2 * C), float("nan"), device=dev)
    hb = torch.full_like(ha, float("nan")); db_ = torch.full_like(ha, float("nan"))
    dxb = torch.full_like(x1, float("nan")); dyb = torch.full_like(x1, float("nan"))
    Fn.call("swin_mlp_ln_bwd_bf16", Fn._p(n2), Fn._p(dy2), Fn._p(w1), Fn._p(b1), Fn._p(w2), Fn._p(hb), Fn._p(db_), Fn._p(x1), Fn._p(mean), Fn._p(rstd),
            Fn._p(gamma), Fn._p(dres), Fn._p(dp), L, Fn._p(dxb), Fn._p(dyb), Fn._p(part), T, C, Fn._s())
    torch.cuda.synchronize()
    assert torch.equal(hb, ha) and torch.equal(db_, da)
    eps16 = 2.0 ** (-10 if H == torch.float16 else -7)
    for a, b, nm in ((dxb, dxa, "dx"), (dyb, dya, "dy")):
        assert bool(torch.isfinite(a.float()).all()), nm
        d = (a.float() - b.float()).abs()
        tol = eps16 * (b.float().abs() + 1.25 * dn2.float().abs().amax(dim=1, keepdim=True) * 0.25 + 2.0 ** -8)
        assert float((d > tol).float().mean()) == 0.0, (nm, float(d.max()))
        assert float((d > 0).float().mean()) <= 5e-3, nm
    sums = part.sum(0)
    scale = float(torch.maximum(dga.abs().max(), dba.abs().max())) + 1e-6
    assert float((sums[:C] - dga).abs().max()) <= 2e-4 * scale + 1e-5
    assert float((sums[C:] - dba).abs().max()) <= 2e-4 * scale + 1e-5


@pytest.mark.parametrize("C", [96, 192])
@pytest.mark.parametrize("T,with_dres,with_dp", [(1, True, True), (777, False, False), (2 * 25 * 40 + 5, True, True)])
def test_mlp_half_of_the_block_backward_in_one_launch_equals_the_three_launch_chain(C, T, with_dres, with_dp):
    """swin_mlp_ln2_bwd_bf16 (next-norm backward as prologue, fused MLP backward, norm2 backward as epilogue) against
    swin_layernorm_bwd -> swin_mlp_bwd_bf16 -> swin_layernorm_bwd."""
    from swin_transformer_object_detection_amd import _lib
    from swin_transformer_object_detection_amd.ops import functional as Fn
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    H = _lib.half_dtype()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(7 * C + T)
    x1 = torch.randn(T, C, generator=g).to(dev, H)
    x2 = torch.randn(T, C, generator=g).to(dev, H)
    g2 = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev); b2n = (0.1 * torch.randn(C, generator=g)).to(dev)
    g3 = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev); b3n = (0.1 * torch.randn(C, generator=g)).to(dev)
    n2 = torch.empty_like(x1); m2 = torch.empty(T, device=dev); r2 = torch.empty(T, device=dev)
    nn = torch.empty_like(x1); m3 = torch.empty(T, device=dev); r3 = torch.empty(T, device=dev)
    Fn.call("swin_layernorm_fwd", Fn._p(x1), Fn._p(g2), Fn._p(b2n), Fn._p(n2), Fn._p(m2), Fn._p(r2), T, C, 1e-5, Fn.SWIN_BF16, Fn._s())
    Fn.call("swin_layernorm_fwd", Fn._p(x2), Fn._p(g3), Fn._p(b3n), Fn._p(nn), Fn._p(m3), Fn._p(r3), T, C, 1e-5, Fn.SWIN_BF16, Fn._s())
    dnn = (torch.randn(T, C, generator=g) * 0.1).to(dev, H)
    dres3 = (torch.randn(T, C, generator=g) * 0.1).to(dev, H) if with_dres else None
    w1 = (torch.randn(4 * C, C, generator=g) * C ** -0.5).to(dev, H)
    w2 = (torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5).to(dev, H)
    b1 = (torch.randn(4 * C, generator=g) * 0.1).to(dev)
    L = max(T // 2, 1)
    ns = (T + L - 1) // L
    dp0 = torch.tensor([1.25, 0.5, 0.0], device=dev)[:ns] if with_dp else None
    dp1 = torch.tensor([0.8, 1.25, 1.0], device=dev)[:ns] if with_dp else None
    ws = Fn._ln_ws(T, C, x1)
    # chain
    dx1a = torch.empty_like(x1); dy2a = torch.empty_like(x1)
    dg3a, db3a = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    Fn.call("swin_layernorm_bwd", Fn._p(dnn), Fn._p(x2), Fn._p(g3), Fn._p(m3), Fn._p(r3), Fn._p(dres3), Fn._p(dx1a), Fn._p(dy2a) if with_dp else None,
            Fn._p(dp1), L, Fn._p(dg3a), Fn._p(db3a), T, C, Fn.SWIN_BF16, Fn._p(ws), Fn._s())
    torch.cuda.synchronize()
    dy2_in = dy2a if with_dp else dx1a
    dn2 = torch.empty_like(x1); ha = torch.empty(T, 4 * C, device=dev, dtype=H); da = torch.empty_like(ha)
    Fn.call("swin_mlp_bwd_bf16", Fn._p(n2), Fn._p(dy2_in), Fn._p(w1), Fn._p(b1), Fn._p(w2), Fn._p(dn2), Fn._p(ha), Fn._p(da), T, C, Fn._s())
    dxa = torch.empty_like(x1); dya = torch.empty_like(x1)
    dg2a, db2a = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ws2 = Fn._ln_ws(T, C, x1)
    Fn.call("swin_layernorm_bwd", Fn._p(dn2), Fn._p(x1), Fn._p(g2), Fn._p(m2), Fn._p(r2), Fn._p(dx1a), Fn._p(dxa), Fn._p(dya), Fn._p(dp0), L,
            Fn._p(dg2a), Fn._p(db2a), T, C, Fn.SWIN_BF16, Fn._p(ws2), Fn._s())
    # one launch
    rows = int(_lib.lib().swin_mlp_ln_bwd_partial_rows(T, C))
    p2 = torch.full((rows, 2 * C), float("nan"), device=dev); p3 = torch.full((rows, 2 * C), float("nan"), device=dev)
    hb = torch.full_like(ha, float("nan")); db_ = torch.full_like(ha, float("nan"))
    dxb = torch.full_like(x1, float("nan")); dyb = torch.full_like(x1, float("nan"))
    dx1b = torch.full_like(x1, float("nan")); dy2b = torch.full_like(x1, float("nan"))
    Fn.call("swin_mlp_ln2_bwd_bf16", Fn._p(n2), Fn._p(w1), Fn._p(b1), Fn._p(w2), Fn._p(hb), Fn._p(db_), Fn._p(x1), Fn._p(m2), Fn._p(r2), Fn._p(g2),
            Fn._p(dp0), L, Fn._p(dxb), Fn._p(dyb), Fn._p(p2), Fn._p(dnn), Fn._p(x2), Fn._p(m3), Fn._p(r3), Fn._p(g3), Fn._p(dres3), Fn._p(dp1),
            Fn._p(dx1b), Fn._p(dy2b), Fn._p(p3), T, C, Fn._s())
    torch.cuda.synchronize()
    eps16 = 2.0 ** (-10 if H == torch.float16 else -7)

    def near(a, b, ref_scale, nm, frac=2e-2):
        assert bool(torch.isfinite(a.float()).all()), nm
        d = (a.float() - b.float()).abs()
        tol = 2 * eps16 * (b.float().abs() + ref_scale + 2.0 ** -8)
        assert float((d > tol).float().mean()) == 0.0, (nm, float(d.max()))
        assert float((d > 0).float().mean()) <= frac, (nm, float((d > 0).float().mean()))
    rs = float(dnn.float().abs().max())
    near(dx1b, dx1a, 0.25 * rs, "dx1", 5e-3)
    if with_dp:
        near(dy2b, dy2a, 0.25 * rs, "dy2", 5e-3)
    # downstream of dx1 / dy2, one-ulp differences there travel through the MLP: compare in the L2 norm
    rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-20))     # noqa: E731
    assert rel(hb, ha) == 0.0                                   # h depends on n2 only
    assert rel(db_, da) <= 5e-3 and rel(dxb, dxa) <= 5e-3 and rel(dyb, dya) <= 5e-3
    for part, dgr, dbr in ((p3, dg3a, db3a), (p2, dg2a, db2a)):
        sums = part.sum(0)
        scale = float(torch.maximum(dgr.abs().max(), dbr.abs().max())) + 1e-6
        assert float((sums[:C] - dgr).abs().max()) <= 5e-3 * scale + 1e-5
        assert float((sums[C:] - dbr).abs().max()) <= 5e-3 * scale + 1e-5
