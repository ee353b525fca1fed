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
