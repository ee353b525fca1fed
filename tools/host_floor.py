"""Wall time of a training step on a TINY input: the GPU work is negligible, so this is the host-side floor of the
step (Python + dispatch + launches), which does not depend on the image size (development aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swin_transformer_object_detection_amd import data, ddp, detector, mixed, presets
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
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    print(f"{h}x{w}: {(time.perf_counter() - t0) * 50:.2f} ms/step")
