"""Host time spent inside the Python backward functions of the custom autograd nodes (they run in the engine's thread, where
cProfile does not look) -- development aid.  usage: python tools/host_bwd.py"""
import collections, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swin_transformer_object_detection_amd import data, ddp, detector, mixed, presets
from swin_transformer_object_detection_amd.optim import FusedAdamW

acc = collections.defaultdict(lambda: [0.0, 0])


def wrap(cls, which):
    fn = getattr(cls, which)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc[f"{cls.__name__}.{which}"]
            e[0] += time.perf_counter() - t0; e[1] += 1
    setattr(cls, which, staticmethod(timed))


def all_functions():
    import importlib, inspect, pkgutil
    import swin_transformer_object_detection_amd as pkg
    seen = set()
    for m in pkgutil.walk_packages(pkg.__path__, pkg.__name__ + "."):
        if m.name.endswith(".build"):
            continue
        mod = importlib.import_module(m.name)
        for _, obj in inspect.getmembers(mod, inspect.isclass):
            if issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function and obj not in seen \
                    and obj.__module__.startswith(pkg.__name__):
                seen.add(obj)
                yield obj


for c in all_functions():
    wrap(c, "forward"); wrap(c, "backward")
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = detector.build_detector(presets.mask_rcnn_swin("tiny"), compute_dtype=torch.bfloat16).to(dev).train()
sh = mixed.ShadowParams(model, torch.bfloat16)
red = ddp.BucketedGradReducer(model.parameters(), leaf_of=sh.leaf_of)
opt = FusedAdamW(model.parameters(), lr=1e-4)
batch = data.synthetic_batch(2, 800, 1280, dev, seed=0)
_st = torch.cuda.Stream(device=dev); _st.wait_stream(torch.cuda.current_stream(dev)); torch.cuda.set_stream(_st)   # as bench.py: not the default stream
sect = collections.defaultdict(float)


def step():
    t0 = time.perf_counter(); red.zero_grad()
    t1 = time.perf_counter(); loss, _ = model.parse_losses(model.forward_train(**batch))
    t2 = time.perf_counter(); loss.backward()
    t3 = time.perf_counter(); red.finish()
    t4 = time.perf_counter(); opt.step()
    t5 = time.perf_counter()
    for k, v in (("zero_grad", t1 - t0), ("forward", t2 - t1), ("backward", t3 - t2), ("finish", t4 - t3), ("optimizer", t5 - t4)):
        sect[k] += v


for _ in range(8): step()
torch.cuda.synchronize(); acc.clear(); sect.clear()
n = 20
for _ in range(n): step()
torch.cuda.synchronize()
print("host ms per step by section: " + ", ".join(f"{k} {v / n * 1e3:.2f}" for k, v in sect.items()) + f"  (sum {sum(sect.values()) / n * 1e3:.2f})")
print(f"{'function':40s} {'ms/step':>8s} {'calls/step':>10s} {'us/call':>8s}")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:40s} {v[0] / n * 1e3:8.3f} {v[1] / n:10.1f} {v[0] / max(v[1], 1) * 1e6:8.1f}")
