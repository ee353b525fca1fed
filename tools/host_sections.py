"""Host issue time of the sections of one training step (no syncs inside; development aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swin_transformer_object_detection_amd import data, ddp, detector, mixed, presets
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = detector.build_detector(presets.mask_rcnn_swin("tiny"), compute_dtype=torch.bfloat16).to(dev).train()
sh = mixed.ShadowParams(model, torch.bfloat16)
red = ddp.BucketedGradReducer(model.parameters(), leaf_of=sh.leaf_of)
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
batch = data.synthetic_batch(2, 800, 1280, dev, seed=0)
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
def step(sync=False):
    t = time.perf_counter(); red.zero_grad(); tick("zero_grad", t)
    t = time.perf_counter(); x = model.extract_feat(batch["img"]);
    if sync: torch.cuda.synchronize()
    tick("trunk_fwd", t)
    t = time.perf_counter()
    shapes = [m['img_shape'] for m in batch["img_metas"]]
    cls, reg = model.rpn_head(x)
    if sync: torch.cuda.synchronize()
    tick("rpn_convs", t)
    t = time.perf_counter(); losses = model.rpn_head.loss(cls, reg, batch["gt_bboxes"], shapes)
    if sync: torch.cuda.synchronize()
    tick("rpn_loss", t)
    t = time.perf_counter(); props = model.rpn_head.get_bboxes(cls, reg, shapes, model.train_cfg['rpn_proposal'], static=True)
    if sync: torch.cuda.synchronize()
    tick("proposals", t)
    t = time.perf_counter(); losses.update(model.roi_head.forward_train(x, props, batch["gt_bboxes"], batch["gt_labels"], batch["gt_masks"]))
    if sync: torch.cuda.synchronize()
    tick("roi_head", t)
    t = time.perf_counter(); loss, _ = model.parse_losses(losses); loss.backward()
    if sync: torch.cuda.synchronize()
    tick("backward", t)
    t = time.perf_counter(); red.finish(); opt.step(); sh.refresh()
    if sync: torch.cuda.synchronize()
    tick("optim", t)
for _ in range(5): step()
torch.cuda.synchronize(); T.clear()
for _ in range(10): step(False)
torch.cuda.synchronize()
print("HOST issue ms/step:", {k: round(v * 100, 2) for k, v in T.items()}, "sum", round(sum(T.values()) * 100, 2))
T.clear()
for _ in range(10): step(True)
print("SYNCED (host+gpu) ms/step:", {k: round(v * 100, 2) for k, v in T.items()}, "sum", round(sum(T.values()) * 100, 2))
