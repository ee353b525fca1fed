"""Ablation timings of the halo-staged conv kernel (development build: SWIN_HALO_ABL bits: 1 no MFMAs, 2 no in-loop DMA, 4 no fragment reads)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from swin_transformer_object_detection_amd.ops import functional as Fn
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]
out = []
for (N, H, W) in ((2, 200, 320), (2, 256, 256), (2, 100, 160)):
    x = torch.randn(N, H, W, 256, device="cuda").bfloat16(); w = (torch.randn(256, 3, 3, 256, device="cuda") * 0.02).bfloat16()
    b = torch.zeros(256, device="cuda"); y = torch.empty(N, H, W, 256, device="cuda", dtype=torch.bfloat16)
    for nt in (2, 4):
        out.append("%%dx%%dx%%d nt%%d %%6.1f" %% (N, H, W, nt, timeit(lambda: Fn.call("conv3x3_halo_nhwc_bf16", Fn._p(x), Fn._p(w), Fn._p(b), None, Fn._p(y), N, H, W, 256, 256, 0, nt, Fn._s()))))
print("  ".join(out))
''' % ROOT
for abl in (0, 1, 2, 4, 3, 5, 6, 7):
    env = dict(os.environ, SWIN_HALO_ABL=str(abl))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"abl={abl}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
