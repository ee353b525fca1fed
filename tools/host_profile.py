"""cProfile of the host side of the training step (development aid): where the ~12 ms of Python / dispatch time per step go."""
import cProfile, io, os, pstats, sys
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
batch = data.synthetic_batch(2, 800, 1280, dev, seed=0)


def step():
    red.zero_grad()
    loss, _ = model.parse_losses(model.forward_train(**batch)); loss.backward(); red.finish(); opt.step()


torch.autograd.set_multithreading_enabled(False)      # backward on THIS thread, so that the profile sees it
for _ in range(8): step()
torch.cuda.synchronize()
n = 20
pr = cProfile.Profile()
pr.enable()
for _ in range(n): step()
pr.disable()
torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s).sort_stats(key)
    st.print_stats(70)
    txt = s.getvalue()
    print(f"==== sorted by {key} (totals over {n} steps; divide by {n})")
    print("\n".join(l[:200] for l in txt.splitlines()[4:]))
