"""development: first difference between the grouped and the single-scan NMS (tests/test_gpu_ops.py cases)"""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swin_transformer_object_detection_amd.ops.functional import _p, _s, call
from swin_transformer_object_detection_amd._lib import lib

rng = np.random.RandomState(2)
B, sizes, m, span, wh = 3, [700, 650, 90, 1200], 900, 120.0, 90.0
n = sum(sizes)
ids = np.concatenate([np.full(k, g) for g, k in enumerate(sizes)])
idxs = torch.from_numpy(np.stack([ids for _ in range(B)])).long().cuda()
xy = rng.rand(B, n, 2).astype(np.float32) * span
boxes = torch.from_numpy(np.concatenate([xy, xy + rng.rand(B, n, 2).astype(np.float32) * wh + 1], 2)).cuda()
scores = torch.from_numpy(rng.rand(B, n).astype(np.float32))
kt = (n - 1) // 7
scores[:, 0:7 * kt:7] = scores[:, 1:7 * kt:7]
scores = scores.cuda()
dev = boxes.device
bs = torch.empty((B, n, 4), dtype=torch.float32, device=dev)
order = torch.empty((B, n), dtype=torch.int32, device=dev)
pws = torch.empty(lib().nms_prepare_workspace_bytes(B, n), dtype=torch.uint8, device=dev)
call("nms_prepare_sorted_batch", _p(boxes), _p(scores), _p(idxs), B, n, _p(bs), _p(order), _p(pws), _s())
out = []
for grouped in (0, 1):
    flags = torch.zeros((B, n), dtype=torch.uint8, device=dev)
    cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    pos = torch.zeros((B, m), dtype=torch.int32, device=dev)
    if grouped:
        G, gmax = len(sizes), max(sizes)
        gws = torch.full((lib().nms_grouped_workspace_bytes(B, n, G, gmax),), 0xA5, dtype=torch.uint8, device=dev)
        call("nms_sorted_batch_grouped", _p(bs), _p(order), _p(idxs), B, n, G, gmax, 0.7, 0, m, _p(flags), _p(cnt), _p(pos), m, _p(gws), _s())
    else:
        ws = torch.empty(B * lib().swin_nms_workspace_bytes(n), dtype=torch.uint8, device=dev)
        call("nms_sorted_batch", _p(bs), B, n, 0.7, 0, m, _p(flags), _p(cnt), _p(pos), m, _p(ws), _s())
    torch.cuda.synchronize()
    out.append((flags.cpu(), cnt.cpu(), pos.cpu()))
(f0, c0, p0), (f1, c1, p1) = out
print("counts", c0.tolist(), c1.tolist())
for b in range(B):
    d = (f0[b] != f1[b]).nonzero().flatten()
    print("image", b, "flag diffs", d.numel(), d[:10].tolist())
    if d.numel():
        i = int(d[0]); src = int(order[b, i]); g = int(idxs[b, src])
        same_g = [(j) for j in range(i) if int(idxs[b, int(order[b, j])]) == g]
        print("  first diff at sorted pos", i, "group", g, "rank in group", len(same_g), "single", int(f0[b, i]), "grouped", int(f1[b, i]))
    dp = (p0[b] != p1[b]).nonzero().flatten()
    print("  kept_pos diffs", dp.numel(), dp[:5].tolist())
