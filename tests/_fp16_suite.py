"""The fp16 build of the library (libswin_hip_f16.so: csrc/common.h with -DSWIN_HALF) against the same fp32 oracles as the bf16 build.
Run in a process of its own with SWIN_HALF_DTYPE=fp16 (a process works with ONE 16-bit type): tests/test_gpu_fp16.py does that.
Tolerances are stated in units of the fp16 spacing 2^-11 relative to each tensor's scale (fp32 accumulation, one rounding per
stored tensor).  Prints one line per check and 'FP16 SUITE OK' at the end."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert os.environ.get("SWIN_HALF_DTYPE") == "fp16"

from oracle import swin_oracle as S  # noqa: E402
from swin_transformer_object_detection_amd import _lib, backbone, data, ddp, detector, mixed, ops, presets  # noqa: E402
from swin_transformer_object_detection_amd.ops import functional as Fn  # noqa: E402
from swin_transformer_object_detection_amd.optim import FusedAdamW  # noqa: E402

H16 = torch.float16
assert _lib.half_dtype() == H16 and _lib.lib().swin_hip_half_type() == 1
ULP = 2.0 ** -11


def close(a, b, ulps, msg):
    b = b.detach().float().cpu()
    tol = ulps * ULP * max(float(b.abs().max()), 1e-3)
    err = float((a.detach().float().cpu() - b).abs().max())
    assert err <= tol, (msg, err, tol)
    print(f"ok {msg}: max err {err:.3e} <= {tol:.3e}", flush=True)


def oracle_attention_natural(qkv, qkv_bias, table, B, H, W, nH, shift):
    C3 = qkv.shape[-1]
    C = C3 // 3
    Hp, Wp = S.padded_hw(H, W)
    x = qkv.view(B, H, W, C3)
    full = qkv_bias.view(1, 1, 1, C3).expand(B, Hp, Wp, C3)
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


# ---- 1. window attention forward + backward (padded, shifted; 3 and 4 heads) against autograd through the oracle
for (B, H, W, nH, shift) in [(2, 20, 23, 3, 3), (1, 14, 14, 4, 0), (2, 9, 16, 6, 3)]:
    C = 32 * nH
    g = torch.Generator().manual_seed(H + nH)
    qkv = (torch.randn(B, H * W, 3 * C, generator=g) * 0.7).half().float()
    qb = torch.randn(3 * C, generator=g) * 0.2
    table = torch.randn(169, nH, generator=g) * 0.5
    go = (torch.randn(B, H * W, C, generator=g) * 0.1).half().float()
    q0, b0, t0 = qkv.clone().requires_grad_(True), qb.clone().requires_grad_(True), table.clone().requires_grad_(True)
    ref = oracle_attention_natural(q0, b0, t0, B, H, W, nH, shift)
    (ref * go).sum().backward()
    q1 = qkv.cuda().half().requires_grad_(True)
    b1, t1 = qb.cuda().requires_grad_(True), table.cuda().requires_grad_(True)
    out = ops.window_attention(q1, b1, t1, B, H, W, nH, shift)
    assert out.dtype == H16
    (out.float() * go.cuda()).sum().backward()
    close(out, ref, 3, f"attention out {H}x{W} nH={nH} shift={shift}")
    close(q1.grad, q0.grad, 4, "attention dqkv")
    close(t1.grad, t0.grad, 40, "attention dtable")

# ---- 2. LayerNorm, fused residual + LayerNorm
x = (torch.randn(500, 384) * 2 + 0.5).half()
w, b = torch.rand(384) + 0.5, torch.randn(384) * 0.1
x0 = x.float().requires_grad_(True)
w0_, b0_ = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
ref = F.layer_norm(x0, (384,), w0_, b0_, 1e-5)
gy = torch.randn(500, 384)
(ref * gy).sum().backward()
x1 = x.cuda().requires_grad_(True)
w1, b1 = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
y = ops.layer_norm(x1, w1, b1)
(y.float() * gy.cuda()).sum().backward()
close(y, ref, 2, "layer_norm")
close(x1.grad, x0.grad, 4, "layer_norm dx")
close(w1.grad, w0_.grad, 40, "layer_norm dgamma")

# ---- 3. MLP: the token-stationary kernel (C = 96) and the GELU-epilogue GEMMs (C = 384)
for T, C in [(1000, 96), (777, 192)]:
    g = torch.Generator().manual_seed(T)
    xx = torch.randn(T, C, generator=g).half()
    w1_ = (torch.randn(4 * C, C, generator=g) * (1.0 / C) ** 0.5).half()
    w2_ = (torch.randn(C, 4 * C, generator=g) * (1.0 / (4 * C)) ** 0.5).half()
    bb1, bb2 = torch.randn(4 * C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3
    ref = F.linear(F.gelu(F.linear(xx.float(), w1_.float(), bb1)), w2_.float(), bb2)
    y = Fn.mlp_fwd_raw(xx.cuda(), w1_.cuda(), bb1.cuda(), w2_.cuda(), bb2.cuda())
    close(y, ref, 3, f"fused mlp forward C={C}")
T, C = 2000, 384
g = torch.Generator().manual_seed(5)
xx = torch.randn(T, C, generator=g).half()
w1_ = (torch.randn(4 * C, C, generator=g) * (1.0 / C) ** 0.5).half()
bb1 = torch.randn(4 * C, generator=g) * 0.3
hpre_ref = F.linear(xx.float(), w1_.float()).half().float()
h_ref = F.gelu(hpre_ref + bb1)
hpre = torch.empty(T, 4 * C, device="cuda", dtype=H16)
h = torch.empty_like(hpre)
xg, wg, bg = xx.cuda(), w1_.cuda(), bb1.cuda()          # (held: the call takes raw pointers)
Fn.call("swin_linear_gelu_hip_bf16", Fn._p(xg), Fn._p(wg), Fn._p(bg), Fn._p(hpre), Fn._p(h), T, 4 * C, C, Fn._s())
close(hpre, hpre_ref, 1.01, "gelu-epilogue GEMM: pre-activation")
close(h, h_ref, 2, "gelu-epilogue GEMM: activation")

# ---- 4. 3x3 convolution forward / data gradient / weight gradient
N, Cc, Hh, Ww = 2, 64, 25, 40
g = torch.Generator().manual_seed(17)
xc = torch.randn(N, Cc, Hh, Ww, generator=g).half().float()
wc = (torch.randn(128, Cc, 3, 3, generator=g) * (2.0 / (9 * Cc)) ** 0.5).half().float()
bc = torch.randn(128, generator=g) * 0.1
gy = (torch.randn(N, 128, Hh, Ww, generator=g) * 0.05).half().float()
x0, w0, b0 = xc.clone().requires_grad_(True), wc.clone().requires_grad_(True), bc.clone().requires_grad_(True)
ref = F.conv2d(x0, w0, b0, padding=1)
(ref * gy).sum().backward()
x1 = xc.cuda().half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
w1, b1 = wc.cuda().requires_grad_(True), bc.cuda().requires_grad_(True)
y = ops.conv3x3(x1, w1, b1, False)
(y.float() * gy.cuda()).sum().backward()
close(y, ref, 2, "conv3x3")
close(x1.grad, x0.grad, 3, "conv3x3 dx")
close(w1.grad, w0.grad, 30, "conv3x3 dw")

# ---- 5. Linear weight gradients: one launch and the grouped launch
for T, N1, N2 in [(5003, 288, 96), (8000, 1536, 384), (2000, 768, 3072)]:
    g = torch.Generator(device="cuda").manual_seed(T)
    dy = (torch.randn(T, N1, device="cuda", generator=g) * 0.1).half()
    xx = torch.randn(T, N2, device="cuda", generator=g).half()
    ref = dy.float().t() @ xx.float()
    dw = torch.zeros(N1, N2, device="cuda"); db = torch.zeros(N1, device="cuda")
    Fn.call("wgrad_linear_bf16", Fn._p(dy), Fn._p(xx), Fn._p(dw), Fn._p(db), T, N1, N2, Fn._s())
    dw2 = torch.zeros(N1, N2, device="cuda"); db2 = torch.zeros(N1, device="cuda")
    Fn.call("swin_wgrad_record", Fn._p(dy), Fn._p(xx), Fn._p(dw2), Fn._p(db2), T, N1, N2)
    Fn.call("swin_wgrad_flush", Fn._s())
    torch.cuda.synchronize()
    for name, got in (("direct", dw), ("grouped", dw2)):
        assert float((got - ref).abs().max()) <= 2e-3 * float(ref.abs().max()), (name, T, N1, N2)
    assert float((db - dy.float().sum(0)).abs().max()) <= 2e-3 * float(dy.float().sum(0).abs().max()) + 1e-3
    print(f"ok wgrad {T} {N1}x{N2}", flush=True)

# ---- 6. the backbone against the fp32 oracle (a config that pads and shifts), forward
cfg = dict(embed_dim=32, depths=(2, 2), num_heads=(1, 2))
p = S.make_params(cfg["embed_dim"], cfg["depths"], cfg["num_heads"], seed=5, out_indices=(0, 1), randomize_norm=True)
m = backbone.SwinTransformer(embed_dim=32, depths=[2, 2], num_heads=[1, 2], out_indices=(0, 1), drop_path_rate=0.0, compute_dtype=H16)
m.load_state_dict(p, strict=False)
m.cuda().eval()
img = torch.randn(1, 3, 60, 76, generator=torch.Generator().manual_seed(1))
with torch.no_grad():
    got = m(img.cuda())
    ref = S.swin_forward(img, p, cfg["depths"], cfg["num_heads"], out_indices=(0, 1))
for i, (a, b) in enumerate(zip(got, ref)):
    assert a.dtype == H16
    close(a, b, 24, f"backbone out{i} (4 blocks of fp16 storage)")

# ---- 7. a training loop with dynamic loss scaling on the device: finite losses, the scale halves on an injected overflow and
# that step leaves the parameters untouched, then training continues
torch.manual_seed(0)
model = detector.build_detector(presets.mask_rcnn_swin("tiny"), compute_dtype=H16).cuda().train()
sh = mixed.ShadowParams(model, H16)
red = ddp.BucketedGradReducer(model.parameters_in_forward_order(), leaf_of=sh.leaf_of)
opt = FusedAdamW(model.parameters(), lr=1e-4)
scaler = mixed.LossScaler(opt, red, init_scale=2.0 ** 13)
batch = data.synthetic_batch(2, 256, 320, torch.device("cuda"), seed=3, num_boxes=5)


def step(poison=False):
    red.zero_grad()
    loss, logs = model.parse_losses(model.forward_train(**batch))
    scaler.scale(loss).backward()
    red.finish()
    if poison:
        red.buckets[0]['flat'][123] = float("inf")
    scaler.check()
    opt.step()
    scaler.update()
    return float(loss.detach())


losses = [step() for _ in range(4)]
assert all(np.isfinite(losses)), losses
s0 = scaler.get_scale()
assert not scaler.skipped_last_step() and s0 == 2.0 ** 13, (s0, scaler.skipped_last_step())
before = [q.detach().clone() for q in model.parameters()]
step(poison=True)
torch.cuda.synchronize()
assert scaler.get_scale() == s0 / 2
assert all(torch.equal(a, q.detach()) for a, q in zip(before, model.parameters())), "a skipped step changed a parameter"
l2 = step()
assert np.isfinite(l2) and any(not torch.equal(a, q.detach()) for a, q in zip(before, model.parameters()))
print(f"ok fp16 training loop: losses {['%.3f' % v for v in losses]}, scale {s0} -> {scaler.get_scale()} after an injected overflow", flush=True)
red.release()
print("FP16 SUITE OK", flush=True)
