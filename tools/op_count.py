"""Count aten ops per section of a training step with TorchDispatchMode (development aid)."""
import os, sys, collections
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
class Counter(TorchDispatchMode):
    def __init__(self): super().__init__(); self.c = collections.Counter()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        self.c[str(func).replace('aten.', '')] += 1
        return func(*args, **(kwargs or {}))
def run(name, fn):
    with Counter() as c:
        out = fn()
    tot = sum(c.c.values())
    print(f"== {name}: {tot} ops; top: {c.c.most_common(14)}")
    return out
red.zero_grad()
shapes = [m['img_shape'] for m in batch["img_metas"]]
x = run("trunk_fwd", lambda: model.extract_feat(batch["img"]))
cls, reg = run("rpn_convs", lambda: model.rpn_head(x))
losses = run("rpn_loss", lambda: model.rpn_head.loss(cls, reg, batch["gt_bboxes"], shapes))
props = run("proposals", lambda: model.rpn_head.get_bboxes(cls, reg, shapes, model.train_cfg['rpn_proposal'], static=True))
l2 = run("roi_head", lambda: model.roi_head.forward_train(x, props, batch["gt_bboxes"], batch["gt_labels"], batch["gt_masks"]))
losses.update(l2)
loss, _ = model.parse_losses(losses)
run("backward", lambda: loss.backward())
