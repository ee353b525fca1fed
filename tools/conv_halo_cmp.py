"""The halo-staged 3x3 convolution (conv3x3_halo_nhwc_bf16) against the implicit-GEMM kernel and an fp32 torch convolution of the same
bf16 operands, on the conv shapes of the step (development; SWIN_CONV_HALO_MIN=0 with the -DSWIN_DEV library keeps the implicit GEMM
on the reference side of the comparison)."""
import os, sys, torch
import torch.nn.functional as F
os.environ.setdefault("SWIN_CONV_HALO_MIN", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swin_transformer_object_detection_amd.ops import functional as Fn


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]


shapes = [(2, 200, 320, 256, 256), (2, 100, 160, 256, 256), (256, 14, 14, 256, 256), (2, 50, 80, 256, 256), (2, 256, 256, 256, 256),
          (1, 9, 11, 64, 128), (3, 7, 9, 128, 192)]
for N, H, W, Cin, Cout in shapes:
    g = torch.Generator().manual_seed(N * H + Cin)
    x = torch.randn(N, H, W, Cin, generator=g).cuda().bfloat16()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (2.0 / (9 * Cin)) ** 0.5).cuda().bfloat16()
    b = (torch.randn(Cout, generator=g) * 0.1).cuda()
    gate = torch.randn(N, H, W, Cout, generator=g).cuda().bfloat16()
    y0 = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.bfloat16)
    ref = None
    if N * H * W <= 40000:
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, padding=1).permute(0, 2, 3, 1)
    t0 = timeit(lambda: Fn.call("conv3x3_nhwc_bf16", Fn._p(x), Fn._p(w), Fn._p(b), Fn._p(y0), N, H, W, Cin, Cout, 0, Fn._s()))
    fl = 2.0 * N * H * W * Cin * Cout * 9
    line = f"{N}x{H}x{W} {Cin}->{Cout}: implicit GEMM {t0:7.1f} us ({fl / t0 / 1e6:5.0f} TF)"
    for nt in (2, 4):
        if nt == 4 and Cout % 256:
            continue
        y1 = torch.full_like(y0, float("nan"))
        th = timeit(lambda: Fn.call("conv3x3_halo_nhwc_bf16", Fn._p(x), Fn._p(w), Fn._p(b), None, Fn._p(y1), N, H, W, Cin, Cout, 0, nt, Fn._s()))
        err = float((y1.float() - y0.float()).abs().max() / y0.float().abs().max())
        line += f" | halo nt={nt} {th:7.1f} us ({fl / th / 1e6:5.0f} TF) vs-gemm {err:.1e}"
        if ref is not None:
            line += f" vs-fp32 {float((y1.float() - ref).abs().max() / ref.abs().max()):.1e}"
        # gate + relu epilogues
        y2 = torch.empty_like(y0); y3 = torch.empty_like(y0)
        Fn.call("conv3x3_halo_nhwc_bf16", Fn._p(x), Fn._p(w), Fn._p(b), Fn._p(gate), Fn._p(y2), N, H, W, Cin, Cout, 0, nt, Fn._s())
        Fn.call("conv3x3_halo_nhwc_bf16", Fn._p(x), Fn._p(w), Fn._p(b), None, Fn._p(y3), N, H, W, Cin, Cout, 1, nt, Fn._s())
        torch.cuda.synchronize()
        ok_g = bool(torch.equal(y2, torch.where(gate.float() > 0, y1, torch.zeros_like(y1))))
        ok_r = bool(torch.equal(y3, torch.relu(y1)))
        line += f" gate={'ok' if ok_g else 'BAD'} relu={'ok' if ok_r else 'BAD'}"
    print(line, flush=True)
