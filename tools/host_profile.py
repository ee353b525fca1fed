"""cProfile of the host side of one training step (development aid)."""
import cProfile, pstats, os, sys, time
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

def step():
    red.zero_grad()
    loss, _ = model.parse_losses(model.forward_train(**batch))
    loss.backward()
    red.finish()
    opt.step()
    sh.refresh()

for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue time per step ms:", (t1 - t0) * 100, " wall per step ms:", (t2 - t0) * 100)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
