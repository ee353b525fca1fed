"""Launch-geometry sweep of the weight-gradient kernels (development; needs the -DSWIN_DEV library:
SWIN_DEV_BUILD=1 python -m swin_transformer_object_detection_amd.build, then SWIN_HIP_LIB=.../lib/libswin_hip_dev.so).
Every configuration is checked against an fp32 matmul / conv weight gradient before it is timed."""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swin_transformer_object_detection_amd.ops import functional as Fn  # noqa: E402


def timeit(fn, n=15, warm=3):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]


def setenv(**kw):
    for k in ("SWIN_WGRAD_GEN", "SWIN_WGRAD2_NBUF", "SWIN_WGRAD2_WKG", "SWIN_WGRAD2_BLOCKS", "SWIN_WGRAD2_XCD", "SWIN_WGRAD_FORM", "SWIN_WGRAD3_BLOCKS"):
        os.environ.pop(k, None)
    if kw and kw.get("SWIN_WGRAD_GEN") != 1 and "SWIN_WGRAD_FORM" not in kw:
        os.environ["SWIN_WGRAD2_FORCE"] = "1"
    else:
        os.environ.pop("SWIN_WGRAD2_FORCE", None)
    for k, v in kw.items():
        os.environ[k] = str(v)


LIN = [(128000, 288, 96), (128000, 96, 96), (128000, 384, 96), (128000, 96, 384), (32000, 576, 192), (32000, 768, 192), (32000, 192, 768),
       (8000, 1152, 384), (8000, 384, 384), (8000, 1536, 384), (8000, 384, 1536), (2000, 2304, 768), (2000, 3072, 768), (2000, 768, 3072),
       (128000, 256, 96), (50176, 1024, 256), (1024, 1024, 12544), (1024, 1024, 1024)]
CONV = [(2, 200, 320), (2, 100, 160), (2, 50, 80), (2, 25, 40), (256, 14, 14)]
which = sys.argv[1] if len(sys.argv) > 1 else "all"

cfgs = [dict(SWIN_WGRAD_GEN=1), dict()]
for nb, kg in ((2, 1), (3, 1), (4, 1), (2, 2)):
    for bl in (256, 384, 512, 768):
        cfgs.append(dict(SWIN_WGRAD2_NBUF=nb, SWIN_WGRAD2_WKG=kg, SWIN_WGRAD2_BLOCKS=bl, SWIN_WGRAD2_XCD=0))
cfgs += [dict(SWIN_WGRAD2_NBUF=2, SWIN_WGRAD2_WKG=1, SWIN_WGRAD2_BLOCKS=bl, SWIN_WGRAD2_XCD=1) for bl in (384, 512, 768)]
cfgs += [dict(SWIN_WGRAD2_NBUF=3, SWIN_WGRAD2_WKG=1, SWIN_WGRAD2_BLOCKS=bl, SWIN_WGRAD2_XCD=1) for bl in (256, 384)]
if os.environ.get("SWEEP_SHORT") == "1":
    cfgs = [dict(SWIN_WGRAD_GEN=1), dict(), dict(SWIN_WGRAD_FORM=2, SWIN_WGRAD2_FORCE=1)]
lin_cfgs = cfgs + [dict(SWIN_WGRAD_FORM=3, SWIN_WGRAD3_BLOCKS=bl) for bl in (192, 256, 320, 512)]


def tag(c):
    if c.get("SWIN_WGRAD_GEN") == 1:
        return "gen1"
    if not c:
        return "default"
    if c.get("SWIN_WGRAD_FORM") == 3:
        return f"f3b{c['SWIN_WGRAD3_BLOCKS']}"
    if c.get("SWIN_WGRAD_FORM") == 2:
        return "f2default"
    return f"nb{c['SWIN_WGRAD2_NBUF']}kg{c['SWIN_WGRAD2_WKG']}b{c['SWIN_WGRAD2_BLOCKS']}x{c['SWIN_WGRAD2_XCD']}"


if which in ("all", "lin"):
    for (T, N1, N2) in LIN:
        g = torch.Generator(device="cuda").manual_seed(T + N1)
        dy = (torch.randn(T, N1, device="cuda", generator=g) * 0.1).bfloat16()
        x = torch.randn(T, N2, device="cuda", generator=g).bfloat16()
        ref = dy.float().t() @ x.float()
        refb = dy.float().sum(0)
        res = []
        for c in lin_cfgs:
            setenv(**c)
            dw = torch.zeros(N1, N2, device="cuda"); db = torch.zeros(N1, device="cuda")
            Fn.call("wgrad_linear_bf16", Fn._p(dy), Fn._p(x), Fn._p(dw), Fn._p(db), T, N1, N2, Fn._s())
            err = float((dw - ref).abs().max() / ref.abs().max()); errb = float((db - refb).abs().max() / (refb.abs().max() + 1e-6))
            us = timeit(lambda: Fn.call("wgrad_linear_bf16", Fn._p(dy), Fn._p(x), Fn._p(dw), Fn._p(db), T, N1, N2, Fn._s()))
            res.append((us, tag(c), err, errb))
        base = res[0][0]
        res.sort()
        bad = [r for r in res if r[2] > 2e-3 or r[3] > 2e-3]
        print(f"lin T={T} {N1}x{N2}: gen1 {base:6.1f} us | " + "  ".join(f"{t} {u:.1f}" for u, t, _, _ in res[:5]) + (f"  BAD {bad[:3]}" if bad else ""), flush=True)

if which in ("all", "conv"):
    import torch.nn.functional as F
    for (N, H, W) in CONV:
        C = 256
        g = torch.Generator(device="cuda").manual_seed(N + H)
        dy = (torch.randn(N, H, W, C, device="cuda", generator=g) * 0.1).bfloat16()
        x = torch.randn(N, H, W, C, device="cuda", generator=g).bfloat16()
        ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).float(), (C, C, 3, 3), dy.permute(0, 3, 1, 2).float(), padding=1)
        ref = ref.permute(0, 2, 3, 1).contiguous()            # (Cout, ky, kx, Cin)
        res = []
        for c in cfgs:
            setenv(**c)
            dw = torch.zeros(C, 3, 3, C, device="cuda"); db = torch.zeros(C, device="cuda")
            Fn.call("wgrad_conv3x3_nhwc_bf16", Fn._p(dy), Fn._p(x), Fn._p(dw), Fn._p(db), N, H, W, C, C, Fn._s())
            err = float((dw - ref).abs().max() / ref.abs().max())
            us = timeit(lambda: Fn.call("wgrad_conv3x3_nhwc_bf16", Fn._p(dy), Fn._p(x), Fn._p(dw), Fn._p(db), N, H, W, C, C, Fn._s()), n=8)
            res.append((us, tag(c), err))
        base = res[0][0]
        res.sort()
        bad = [r for r in res if r[2] > 3e-3]
        fl = 2.0 * N * H * W * C * C * 9
        print(f"conv {N}x{H}x{W}: gen1 {base:6.1f} us | " + "  ".join(f"{t} {u:.1f} ({fl / u / 1e6:.0f}TF)" for u, t, _ in res[:5]) + (f"  BAD {bad[:3]}" if bad else ""), flush=True)
