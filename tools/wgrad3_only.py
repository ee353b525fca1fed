"""a few launches of one Linear weight-gradient shape (counter passes): wgrad3_only.py T N1 N2 [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swin_transformer_object_detection_amd.ops import functional as Fn
T, N1, N2 = (int(v) for v in sys.argv[1:4]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dy = torch.randn(T, N1, device="cuda").bfloat16(); x = torch.randn(T, N2, device="cuda").bfloat16()
dw = torch.zeros(N1, N2, device="cuda"); db = torch.zeros(N1, device="cuda")
for _ in range(reps):
    Fn.call("wgrad_linear_bf16", Fn._p(dy), Fn._p(x), Fn._p(dw), Fn._p(db), T, N1, N2, Fn._s())
torch.cuda.synchronize()
