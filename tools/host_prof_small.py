"""cProfile of a training step on a tiny input (pure host cost; development aid)."""
import cProfile, pstats, os, sys, time
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
batch = data.synthetic_batch(2, 128, 160, dev, seed=0)
T = {}
def tick(k, t0): T[k] = T.get(k, 0) + time.perf_counter() - t0
def step():
    t = time.perf_counter(); red.zero_grad(); tick('zero', t)
    t = time.perf_counter(); x = model.extract_feat(batch["img"]); tick('trunk_fwd', t)
    t = time.perf_counter(); shapes = [m['img_shape'] for m in batch["img_metas"]]; cls, reg = model.rpn_head(x); tick('rpn_convs', t)
    t = time.perf_counter(); losses = model.rpn_head.loss(cls, reg, batch["gt_bboxes"], shapes); tick('rpn_loss', t)
    t = time.perf_counter(); props = model.rpn_head.get_bboxes(cls, reg, shapes, model.train_cfg['rpn_proposal'], static=True); tick('proposals', t)
    t = time.perf_counter(); losses.update(model.roi_head.forward_train(x, props, batch["gt_bboxes"], batch["gt_labels"], batch["gt_masks"])); tick('roi_head', t)
    t = time.perf_counter(); loss, _ = model.parse_losses(losses); loss.backward(); tick('backward', t)
    t = time.perf_counter(); red.finish(); opt.step(); tick('optim', t)
for _ in range(5): step()
torch.cuda.synchronize(); T.clear()
for _ in range(20): step()
torch.cuda.synchronize()
print("host ms/step by section:", {k: round(v * 50, 2) for k, v in T.items()}, "sum", round(sum(T.values()) * 50, 2))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumtime").print_stats(45)
