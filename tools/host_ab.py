"""A/B of the direct-hipBLASLt vs framework GEMM dispatch inside ONE process, alternating blocks of steps (the host
timing noise between processes on a shared box is larger than the effect; development aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swin_transformer_object_detection_amd import data, ddp, detector, mixed, presets
from swin_transformer_object_detection_amd.ops import functional as Fn
from swin_transformer_object_detection_amd.optim import FusedAdamW
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = detector.build_detector(presets.mask_rcnn_swin("tiny"), compute_dtype=torch.bfloat16).to(dev).train()
sh = mixed.ShadowParams(model, torch.bfloat16)
red = ddp.BucketedGradReducer(model.parameters(), leaf_of=sh.leaf_of)
opt = FusedAdamW(model.parameters(), lr=1e-4)
for (h, w) in [(128, 160), (800, 1280)]:
    batch = data.synthetic_batch(2, h, w, dev, seed=0)
    def step():
        red.zero_grad()
        loss, _ = model.parse_losses(model.forward_train(**batch)); loss.backward(); red.finish(); opt.step()
    res = {True: [], False: []}
    for rep in range(8):
        for direct in (True, False):
            Fn._DIRECT_GEMM = direct
            for _ in range(3): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(15): step()
            torch.cuda.synchronize()
            res[direct].append((time.perf_counter() - t0) / 15 * 1e3)
    for direct in (True, False):
        v = sorted(res[direct])
        print(f"{h}x{w} direct={direct}: min {v[0]:.2f}  median {v[len(v) // 2]:.2f}  max {v[-1]:.2f} ms/step")
