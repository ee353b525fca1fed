"""Is the training step host-bound?  Host time to ENQUEUE n steps (no synchronisation) against the time until the GPU has
finished them (development aid).  host ~ total: the GPU waits for the host."""
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
batch = data.synthetic_batch(2, 800, 1280, dev, seed=0)
_st = torch.cuda.Stream(device=dev); _st.wait_stream(torch.cuda.current_stream(dev)); torch.cuda.set_stream(_st)   # as bench.py: not the default stream


def step():
    red.zero_grad()
    loss, _ = model.parse_losses(model.forward_train(**batch)); loss.backward(); red.finish(); opt.step()


for _ in range(8): step()
torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == "engine":
    # A/B in ONE process (boxes differ by 30 % in host speed): autograd engine on its device thread (default) vs on the calling thread
    for rep in range(3):
        for single in (False, True):
            with torch.autograd.set_multithreading_enabled(not single):
                for _ in range(3): step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(30): step()
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
            print(f"engine on the {'calling' if single else 'device '} thread: host enqueue {(t1 - t0) / 30 * 1e3:.2f} ms/step, until GPU done {(t2 - t0) / 30 * 1e3:.2f} ms/step", flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "gc":
    # A/B in one process: Python's cyclic collector as it comes vs frozen survivors + a generation-0 threshold far above a step's allocations
    import gc
    torch.autograd.set_multithreading_enabled(False)
    for rep in range(3):
        for tuned in (False, True):
            if tuned:
                gc.collect(); gc.freeze(); gc.set_threshold(200000, 50, 50)
            else:
                gc.unfreeze(); gc.set_threshold(700, 10, 10)
            for _ in range(3): step()
            torch.cuda.synchronize()
            t0 = time.perf_counter(); marks = []
            for _ in range(40):
                step(); marks.append(time.perf_counter())
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            d = sorted((marks[i] - marks[i - 1]) * 1e3 for i in range(1, 40))
            print(f"gc {'frozen, threshold 200000' if tuned else 'default                 '}: host enqueue {(t1 - t0) / 40 * 1e3:.2f} ms/step "
                  f"(median {d[len(d) // 2]:.2f}, max {d[-1]:.2f}), until GPU done {(t2 - t0) / 40 * 1e3:.2f} ms/step", flush=True)
    sys.exit(0)
n = 30
t0 = time.perf_counter()
marks = []
for _ in range(n):
    step(); marks.append(time.perf_counter())
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {(t1 - t0) / n * 1e3:.2f} ms/step, until GPU done {(t2 - t0) / n * 1e3:.2f} ms/step, "
      f"GPU still busy for {(t2 - t1) * 1e3:.2f} ms after the last enqueue")
d = [(marks[i] - marks[i - 1]) * 1e3 for i in range(1, n)]
print("per-step host intervals (ms): first 5", [f"{x:.2f}" for x in d[:5]], " last 5", [f"{x:.2f}" for x in d[-5:]])
