"""development: parameters after 3 optimizer steps, box head on the main stream vs on the sub-graph stream"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from swin_transformer_object_detection_amd import detector, mixed
from swin_transformer_object_detection_amd.ops import targets
from swin_transformer_object_detection_amd.graph_step import GraphedTrainStep
import test_gpu_graph_step as T
targets._next_seed = lambda: 0x1234567
model, sh, red, opt, batch = T._setup()
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
names = [n for n, _ in model.named_parameters()]
with torch.cuda.stream(st):
    g = GraphedTrainStep(model, red, opt, warmup=2)
    for _ in range(2): g.eager(batch)
    snap = T._snapshot(model, opt)
    def run(flag, n=3):
        detector._BBOX_BRANCH = flag
        T._restore(model, opt, snap)
        for _ in range(n): g.eager(batch)
        torch.cuda.synchronize()
        return [p.detach().clone() for p in model.parameters()]
    for n in (1, 2, 3):
        p0, p0b, p1 = run(False, n), run(False, n), run(True, n)
        rows = []
        for nm, a, a2, b, ref in zip(names, p0, p0b, p1, snap[0]):
            moved = float((a - ref).norm()) + 1e-12
            rows.append((float((a - b).norm()) / moved, float((a - a2).norm()) / moved, nm))
        rows.sort(reverse=True)
        print(f"after {n} steps: worst on-vs-off / moved:", ["%.3f (noise %.4f) %s" % r for r in rows[:4]])
