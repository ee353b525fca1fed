"""Attribute launch-producing aten ops of one training step to call sites in the package (development aid)."""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swin_transformer_object_detection_amd import data, ddp, detector, mixed, presets
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = detector.build_detector(presets.mask_rcnn_swin("tiny"), compute_dtype=torch.bfloat16).to(dev).train()
sh = mixed.ShadowParams(model, torch.bfloat16)
red = ddp.BucketedGradReducer(model.parameters(), leaf_of=sh.leaf_of)
batch = data.synthetic_batch(2, 800, 1280, dev, seed=0)
SKIP = ('view', 'detach', 't.default', 'permute', 'slice', 'select', 'unsqueeze', 'expand', 'empty', '_unsafe_view', 'alias',
        'squeeze', 'transpose', 'as_strided', 'lift_fresh', 'reshape', 'unbind', 'split', 'new_empty', 'stride', 'size')
PK = 'swin_transformer_object_detection_amd'
class Sites(TorchDispatchMode):
    def __init__(self): super().__init__(); self.c = collections.Counter()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace('aten.', '')
        if not any(name.startswith(s) for s in SKIP):
            site = 'autograd-native'
            for fr in reversed(traceback.extract_stack(limit=30)):
                if PK in fr.filename and 'tools' not in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            shp = ''
            for a in args:
                if isinstance(a, torch.Tensor):
                    shp = f"{tuple(a.shape)}{str(a.dtype).replace('torch.', '')}"
                    break
            self.c[(name, site, shp)] += 1
        return func(*args, **(kwargs or {}))
for _ in range(2):
    red.zero_grad()
    loss, _ = model.parse_losses(model.forward_train(**batch)); loss.backward(); red.finish()
red.zero_grad()
with Sites() as s:
    losses = model.forward_train(**batch)
    loss, _ = model.parse_losses(losses)
    loss.backward()
    red.finish()
tot = sum(s.c.values())
print("launch-producing ops:", tot)
bysite = collections.Counter()
for (n, site, shp), k in s.c.items():
    bysite[site] += k
print("== by site"); [print(f"{k:5d} {site}") for site, k in bysite.most_common(45)]
print("== by (op, site, first-arg)"); [print(f"{k:5d} {n:28s} {site:26s} {shp}") for (n, site, shp), k in s.c.most_common()]
