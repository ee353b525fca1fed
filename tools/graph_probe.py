"""Probe: capture backbone+FPN+RPN-convs forward/backward in HIP graphs (torch.cuda.make_graphed_callables)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swin_transformer_object_detection_amd import data, detector, mixed, presets

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = detector.build_detector(presets.mask_rcnn_swin("tiny"), compute_dtype=torch.bfloat16).to(dev).train()
sh = mixed.ShadowParams(model, torch.bfloat16)
batch = data.synthetic_batch(2, 800, 1280, dev, seed=0)
img = batch["img"]

trunk_mods = [model.backbone, model.neck, model.rpn_head]
masters = [p for m in trunk_mods for p in m.parameters() if p.requires_grad]
leaves = [sh.leaf_of(p) for p in masters]
extra_small = [p for p in masters if sh.leaf_of(p) is not p]     # masters whose shadow is used (not needed as inputs)


def trunk(img, *leaf_args):
    x = model.extract_feat(img)
    cls, reg = model.rpn_head(x)
    return tuple(x) + tuple(cls) + tuple(reg)


def run_eager():
    outs = trunk(img, *leaves)
    loss = sum(o.float().square().mean() for o in outs)
    loss.backward()


for _ in range(3):
    run_eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    run_eager()
torch.cuda.synchronize()
print("eager trunk fwd+bwd ms:", (time.perf_counter() - t0) * 100)

g_trunk = torch.cuda.make_graphed_callables(trunk, (img,) + tuple(leaves), num_warmup_iters=3)


def run_graph():
    outs = g_trunk(img, *leaves)
    loss = sum(o.float().square().mean() for o in outs)
    loss.backward()


for _ in range(3):
    run_graph()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    run_graph()
torch.cuda.synchronize()
print("graphed trunk fwd+bwd ms:", (time.perf_counter() - t0) * 100)
# numerics: same grads?
for l in leaves:
    l.grad = None
run_eager(); ge = [l.grad.clone() for l in leaves[:5]]
for l in leaves:
    l.grad = None
run_graph(); gg = [l.grad.clone() for l in leaves[:5]]
print("max grad diff:", max(float((a.float() - b.float()).abs().max()) for a, b in zip(ge, gg)))
