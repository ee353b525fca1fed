"""One training step as ONE hipGraph launch.

A step of the Mask R-CNN Swin path is ~570 kernel launches of 5-250 us each on two or three HIP streams.  Issued eagerly from
Python they cost the host 9-13 ms per step -- as much as the GPU needs to run them -- so the measured rate follows the host's
speed, not the GPU's (round 2: 11.4 ms per step on one box, 13.2 ms on another with identical kernel times).  The step has
fixed shapes and no host synchronisation by construction (fixed-size proposal / sample lists, device-side NMS and samplers),
so the whole of it -- ``forward_train`` (mmdet/models/detectors/two_stage.py:106-167), backward, the gradient reducer's
gather and the fused AdamW -- is captured once and replayed with a single ``hipGraphLaunch``; the weight-gradient /
sub-graph streams become parallel branches of the graph.

What changes from step to step does not live in kernel arguments:
  * learning rates, weight decays and Adam's bias corrections: ``FusedAdamW.prepare_step`` writes them into the optimizer's
    device-resident state before every replay (one tiny launch); the captured ``swin_adamw_step_dev`` reads them there;
  * the samplers' randomness (RandomSampler, random_sampler.py:54): a device-resident step seed that the kernels mix into the
    seeds frozen into the graph (``ops.targets.set_step_seed``);
  * DropPath / torch.rand draws: torch's CUDA generator is graph-aware (the philox offset advances at every replay);
  * the batch: copied into the captured step's static input tensors before the replay.
A captured step is valid for one input signature (image size, ground-truth box counts); ``GraphedTrainStep`` keeps one graph
per signature and captures on first use.  With more than one rank the step runs eagerly unless ``capture_collectives`` is
set: the gradient all-reduce overlapped with backward would have to be captured with it, which this round could not
rehearse on more than one GPU.
"""
import torch

from . import mixed
from .ops import targets as _targets


def _signature(batch):
    sig = [tuple(batch['img'].shape), str(batch['img'].dtype)]
    for k in ('gt_bboxes', 'gt_labels', 'gt_masks'):
        v = batch.get(k)
        if v is not None:
            sig.append(tuple(tuple(t.shape) for t in v))
    sig.append(tuple((m.get('img_shape'), m.get('pad_shape')) for m in batch.get('img_metas', ())))
    return tuple(sig)


def _copy_batch(dst, src):
    """src's tensors into dst's (same structure and shapes), one multi-tensor copy"""
    d, s = [dst['img']], [src['img']]
    for k in ('gt_bboxes', 'gt_labels', 'gt_masks'):
        if dst.get(k) is not None:
            d += list(dst[k]); s += list(src[k])
    pairs = [(a, b) for a, b in zip(d, s) if a is not b and a.data_ptr() != b.data_ptr()]
    if pairs:
        torch._foreach_copy_([a for a, _ in pairs], [b for _, b in pairs])


class GraphedTrainStep:
    """step = GraphedTrainStep(model, reducer, optimizer); log_vars = step(batch)

    ``reducer``: ddp.BucketedGradReducer (world size 1 unless capture_collectives), ``optimizer``: optim.FusedAdamW.
    ``warmup``: eager steps run before a capture (they build the GEMM plans, scratch buffers and the optimizer's tables; a
    capture must not meet a first-time allocation that synchronises).  The returned ``log_vars`` are the captured step's own
    output tensors: valid until the next call."""

    def __init__(self, model, reducer, optimizer, warmup=3, capture_collectives=False, loss_scale=None):
        self.model, self.reducer, self.optim = model, reducer, optimizer
        self.warmup = int(warmup)
        self.loss_scale = loss_scale          # mixed.LossScaler or None
        if reducer.world > 1 and not capture_collectives:
            raise RuntimeError("GraphedTrainStep: more than one rank (pass capture_collectives=True to capture the all-reduces too)")
        self._graphs = {}                     # signature -> (graph, static batch, log_vars)
        self.stream = None

    # ---- the step body (identical to the eager loop of bench.py) ----
    def _body(self, batch, captured):
        self.reducer.zero_grad()
        losses = self.model.forward_train(**batch)
        loss, log_vars = self.model.parse_losses(losses)
        if self.loss_scale is not None:
            loss = self.loss_scale.scale(loss)
        with torch.autograd.set_multithreading_enabled(False):
            loss.backward()
        self.reducer.finish()
        if self.loss_scale is not None:
            self.loss_scale.check(self.optim)
        if captured:
            self.optim.apply_step()
        else:
            self.optim.step()
        if self.loss_scale is not None:
            self.loss_scale.update(self.optim)
        return log_vars

    def eager(self, batch):
        dev = batch['img'].device
        if _targets.step_seed_tensor(dev) is not None:
            _targets.set_step_seed(dev)
        return self._body(batch, False)

    def _capture(self, batch):
        dev = batch['img'].device
        _targets.set_step_seed(dev)           # turns the device-resident seed on (before warm-up: same code path as the capture)
        static = dict(batch)
        static['img'] = batch['img'].clone()
        for k in ('gt_bboxes', 'gt_labels', 'gt_masks'):
            if batch.get(k) is not None:
                static[k] = [t.clone() for t in batch[k]]
        for _ in range(self.warmup):
            self.eager(static)
        mixed.side_join()
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        _targets.set_step_seed(dev)
        self.optim.prepare_step()
        torch.cuda.synchronize(dev)
        with torch.cuda.graph(g, stream=self.stream):
            log_vars = self._body(static, True)
        g.replay()                            # the capture pass records, it does not execute: run the step it was prepared for
        return g, static, log_vars

    def __call__(self, batch):
        sig = _signature(batch)
        ent = self._graphs.get(sig)
        if ent is None:
            ent = self._graphs[sig] = self._capture(batch)
            return ent[2]
        g, static, log_vars = ent
        _copy_batch(static, batch)
        _targets.set_step_seed(batch['img'].device)
        self.optim.prepare_step()
        g.replay()
        return log_vars

    def graphs(self):
        return len(self._graphs)
