import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swin_transformer_object_detection_amd import data, ddp, detector, mixed, presets
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = detector.build_detector(presets.mask_rcnn_swin("tiny"), compute_dtype=torch.bfloat16).to(dev).train()
sh = mixed.ShadowParams(model, torch.bfloat16)
red = ddp.BucketedGradReducer(model.parameters(), leaf_of=sh.leaf_of)
batch = data.synthetic_batch(2, 128, 160, dev, seed=0)
def run(kind, n=20):
    tf = tb = 0.0
    for it in range(n + 5):
        red.zero_grad()
        t0 = time.perf_counter()
        if kind == 'backbone':
            outs = model.backbone(batch["img"]); loss = sum(o.float().mean() for o in outs)
        elif kind == 'trunk':
            outs = model.extract_feat(batch["img"]); loss = sum(o.float().mean() for o in outs)
        else:
            loss, _ = model.parse_losses(model.forward_train(**batch))
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        red.finish()
        if it >= 5: tf += t1 - t0; tb += t2 - t1
    torch.cuda.synchronize()
    print(f"{kind:9s}: fwd host {tf / n * 1e3:6.2f} ms  bwd host {tb / n * 1e3:6.2f} ms")
for k in ('backbone', 'trunk', 'full'):
    run(k)
import collections
from torch.utils._python_dispatch import TorchDispatchMode
# count autograd nodes of the full graph
batch = data.synthetic_batch(2, 800, 1280, dev, seed=0)
loss, _ = model.parse_losses(model.forward_train(**batch))
seen, stack, names = set(), [loss.grad_fn], collections.Counter()
while stack:
    f = stack.pop()
    if f is None or f in seen: continue
    seen.add(f); names[type(f).__name__] += 1
    stack.extend(n for n, _ in f.next_functions)
print("autograd nodes:", len(seen)); print(names.most_common(40))
