"""Token-stationary Linear (swin_ts_linear_bf16) against hipBLASLt (swin_gemm_bf16) on the narrow-contraction shapes of the step (development)."""
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
for M, N, K, name in ((128000, 288, 96, "qkv s1"), (32000, 576, 192, "qkv s2"), (128000, 96, 96, "proj s1"), (32000, 192, 192, "proj s2"),
                      (128000, 256, 96, "lateral 0"), (32000, 256, 192, "lateral 1"), (50176, 1024, 256, "deconv"), (131072, 384, 128, "qkv swin-b s1"),
                      (8000, 1152, 384, "qkv s3"), (8000, 384, 384, "proj s3"), (8000, 1536, 384, "fc1 s3"), (8000, 256, 384, "lateral 2")):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    c0 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); c1 = torch.empty_like(c0)
    lt = timeit(lambda: Fn.call("swin_gemm_bf16", Fn._p(a), Fn._p(w), Fn._p(b), Fn._p(c0), M, N, K, 0, Fn._p(ws), Fn._s()))
    ts = timeit(lambda: Fn.call("swin_ts_linear_bf16", Fn._p(a), Fn._p(w), Fn._p(b), Fn._p(c1), M, N, K, 0, Fn._s()))
    by = 2.0 * (M * K + M * N)
    print(f"{name:14s} M={M:6d} N={N:5d} K={K:4d}: hipBLASLt {lt:6.1f} us ({by / lt / 1e3:5.0f} GB/s)   token-stationary {ts:6.1f} us ({by / ts / 1e3:5.0f} GB/s)", flush=True)
