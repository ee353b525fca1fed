"""hipBLASLt (swin_gemm_bf16) against the hand-written MFMA GEMM (swin_linear_hip_bf16) on the Linear shapes of the step (development)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swin_transformer_object_detection_amd.ops import functional as Fn
from swin_transformer_object_detection_amd import _lib


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]


ws = torch.empty(_lib.lib().swin_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
shapes = []
for T, C in ((128000, 96), (32000, 192), (8000, 384), (2000, 768)):
    shapes += [(T, 3 * C, C, "qkv"), (T, C, C, "proj"), (T, 4 * C, C, "fc1"), (T, C, 4 * C, "fc2"), (T, C, 3 * C, "dqkv")]
shapes += [(32000, 192, 384, "merge1"), (8000, 384, 768, "merge2"), (2000, 768, 1536, "merge3"), (128000, 256, 96, "lat0"), (1024, 1024, 12544, "bbox_fc1"),
           (1024, 12544, 1024, "d_bbox_fc1"), (50176, 1024, 256, "deconv")]
for M, N, K, name in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b16 = torch.randn(N, device="cuda").bfloat16()
    b32 = b16.float()
    c0 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    c1 = torch.empty_like(c0)
    lt = timeit(lambda: Fn.call("swin_gemm_bf16", Fn._p(a), Fn._p(w), Fn._p(b16), Fn._p(c0), M, N, K, 0, Fn._p(ws), Fn._s()))
    if K % 64 == 0:
        mine = timeit(lambda: Fn.call("swin_linear_hip_bf16", Fn._p(a), Fn._p(w), Fn._p(b32), Fn._p(c1), M, N, K, 0, Fn._s()))
        err = float((c0.float() - c1.float()).abs().max() / c0.float().abs().max())
    else:
        mine, err = float("nan"), float("nan")
    fl = 2.0 * M * N * K
    print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d}: hipBLASLt {lt:6.1f} us ({fl / lt / 1e6:5.0f} TF)   mine {mine:6.1f} us ({fl / mine / 1e6:5.0f} TF)  relerr {err:.1e}", flush=True)
