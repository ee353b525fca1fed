"""development: gradients of ONE step with the box head on the main stream vs on the sub-graph stream (detector._BBOX_BRANCH) -- the
difference should be bf16 summation-order noise, not a race"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swin_transformer_object_detection_amd import data, ddp, detector, mixed, presets
from swin_transformer_object_detection_amd.ops import targets
targets._next_seed = lambda: 0x1234567
cfg = presets.mask_rcnn_swin("tiny"); cfg["backbone"]["drop_path_rate"] = 0.0
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = detector.build_detector(cfg, compute_dtype=torch.bfloat16).to(dev).train()
sh = mixed.ShadowParams(model, torch.bfloat16)
red = ddp.BucketedGradReducer(model.parameters_in_forward_order(), leaf_of=sh.leaf_of)
batch = data.synthetic_batch(2, 256, 320, dev, seed=3, num_boxes=5)
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())

def grads(flag):
    detector._BBOX_BRANCH = flag
    out = []
    with torch.cuda.stream(st):
        for _ in range(2):
            red.zero_grad()
            loss, _ = model.parse_losses(model.forward_train(**batch)); loss.backward(); red.finish()
        torch.cuda.synchronize()
        out = [p.grad.detach().float().clone() if p.grad is not None else None for p in model.parameters()]
    return out, float(loss)

names = [n for n, _ in model.named_parameters()]
g0, l0 = grads(False); g0b, _ = grads(False); g1, l1 = grads(True); g1b, _ = grads(True)
print("loss", l0, l1)
worst = []
for n, a, a2, b, b2 in zip(names, g0, g0b, g1, g1b):
    if a is None: continue
    den = float(a.norm()) + 1e-12
    worst.append((float((a - b).norm()) / den, float((a - a2).norm()) / den, float((b - b2).norm()) / den, n, tuple(a.shape)))
worst.sort(reverse=True)
for w in worst[:12]: print("rel diff branch-vs-main %.3e   run-to-run main %.3e  branch %.3e   %s %s" % w)
import statistics
print("median rel diff", statistics.median(w[0] for w in worst))
