"""Data-parallel gradient exchange: bucketed all-reduce over RCCL, overlapped with backward.

Replaces the reference's ``MMDistributedDataParallel`` wrap (``mmdet/apis/train.py:91-99``; torch DDP with
``broadcast_buffers=False``) and its per-scalar loss all-reduce (``mmdet/models/detectors/base.py:211-216``).

Design for MI355X / xGMI (one process per GPU, backend "nccl" == RCCL):
  * the fp32 gradients of all parameters live in a few large flat buffers (``param.grad`` are views), so a
    bucket is all-reduced in place with no gather/scatter copies and the fused optimizer reads them directly;
  * autograd leaves may be bf16 shadow copies of the fp32 masters (``mixed.ShadowParams``): when the last
    gradient of a bucket has been produced, ONE multi-tensor copy moves (and up-converts) the bucket's fresh
    gradients into the flat fp32 buffer -- no per-parameter cast / accumulate / zero-fill kernels;
  * buckets are formed in reverse registration order (heads -> FPN -> Swin stage 4..1), the order backward
    produces them; a post-accumulate hook counts arrivals and launches the bucket's async all-reduce as soon
    as it is complete, so communication hides under the remaining backward;
  * bucket size defaults to 48 MiB: xGMI links are point-to-point (~153 GB/s each), a few large messages keep
    every link busy and the per-collective launch cost negligible (Mask R-CNN Swin-T: 192 MB of fp32
    gradients -> 4 buckets);
  * the loss scalars for logging are packed into ONE tensor and reduced once, without .item().
  * collectives are issued STRICTLY in bucket-index order on every rank (bucket i only after buckets 0..i-1), whatever
    order the buckets complete in: a rank-dependent graph (a parameter unused on one rank) must not pair mismatched
    collectives; a bucket that never completes is issued by finish(), still in index order;
  * one backward() per finish(): a gradient that arrives for a bucket whose all-reduce is already in flight would be
    dropped or race with the collective, so it raises (gradient accumulation over several backward calls is not
    supported -- accumulate in the loss instead);
  * every launch / completion is time-stamped relative to mark_backward_start() (``timeline``) so a multi-GPU run
    shows by itself how much of the exchange overlapped backward.
Works unchanged on CPU with the gloo backend (tests/test_ddp_gloo.py, world_size 2).
"""
import contextlib
import os
import time

import torch
import torch.distributed as dist

_DEBUG = os.environ.get("SWIN_DDP_DEBUG") == "1"     # record where a parameter's gradient first arrived (late-arrival diagnosis)


class BucketedGradReducer:
    def __init__(self, params, bucket_bytes=48 << 20, process_group=None, average=True, leaf_of=None):
        """params: the fp32 master parameters (what the optimizer updates).  leaf_of(p) -> the tensor autograd
        accumulates into for p (p itself, or its bf16 shadow)."""
        self.params = [p for p in params if p.requires_grad]
        self.leaf_of = leaf_of or (lambda p: p)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.average = average
        self.buckets = []
        self._l2b = {}
        self._next = 0                      # index of the next bucket whose collective may be issued
        self._t0 = None
        self._home = None                   # the stream the step is issued on (set by zero_grad)
        self.early_step = None              # one process: callable(params, stream) -> bool that applies the optimizer to a finished bucket
        self.force_overlap = os.environ.get("SWIN_DDP_OVERLAP") == "1"    # world size 1: still gather mid-backward on the launch stream (tests)
        self.timeline = []                  # per step: [(bucket, launch_s, done_s | None, bytes)] relative to mark_backward_start
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_bytes:
                self._make_bucket(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._make_bucket(cur)
        self._hooks = []
        for b in self.buckets:
            for i, leaf in enumerate(b['leaves']):
                # the engine runs a leaf's accumulate node (and both hooks) even when a custom Function returned None for it
                # -- exactly what the sink kernels do.  The tensor hook sees the incoming gradient and records whether it is a
                # real one, so that a None "arrival" after the bucket was issued is not mistaken for a late gradient.
                self._hooks.append(leaf.register_hook(lambda g, b=b, i=i: self._pre(b, i, g)))
                self._hooks.append(leaf.register_post_accumulate_grad_hook(self._on_grad))
        # gradient sinks: weight-gradient kernels may accumulate straight into the fp32 bucket views
        from . import mixed
        table = {}
        for b in self.buckets:
            for i, (p, v, leaf) in enumerate(zip(b['params'], b['views'], b['leaves'])):
                table[id(p)] = (v, (lambda b=b, i=i: self._on_direct(b, i)))
        mixed.register_sinks(table)
        self._sink_ids = list(table)
        mixed.wgrad_group_enable(True)          # Linear weight gradients may be recorded and launched in groups: finish() / every
                                                # bucket launch flushes them (mixed.side_join / fork_into)

    def _make_bucket(self, plist):
        n = sum(p.numel() for p in plist)
        flat = torch.zeros(n, device=plist[0].device, dtype=torch.float32)
        views, off = [], 0
        from .mixed import dense_view
        for p in plist:
            views.append(dense_view(flat, off, p))          # the parameter's shape AND memory layout (contiguous / channels-last)
            off += p.numel()
        leaves = [self.leaf_of(p) for p in plist]
        b = dict(flat=flat, params=list(plist), leaves=leaves, views=views, pending=len(plist), handle=None,
                 arrived=[False] * len(plist), direct=[False] * len(plist), real=[False] * len(plist), auto=[], launched=False, ready=False,
                 no=len(self.buckets), index={id(l): i for i, l in enumerate(leaves)})
        for l in leaves:
            self._l2b[id(l)] = b
        for p, v in zip(plist, views):
            p.grad = v
        self.buckets.append(b)

    def _gather(self, b, keep=False):
        """fresh leaf gradients -> flat fp32 bucket (one multi-tensor copy; converts bf16 -> fp32).  keep: the copies run on
        another stream than the one the gradients' memory belongs to -- hold the gradients until the next join."""
        src, dst, asrc, adst = [], [], [], []
        leaves, params, views = b['leaves'], b['params'], b['views']
        # only the leaves autograd actually delivered a gradient to in this step (recorded by the hooks): with the sink kernels
        # that is a handful of the ~230 parameters, and walking all of them cost 0.4 ms of host time per step
        for i in b['auto']:
            leaf, v = leaves[i], views[i]
            g = leaf.grad
            if g is None or g.data_ptr() == v.data_ptr():
                continue                                    # accumulated in place into the view itself
            if b['direct'][i]:
                asrc.append(g); adst.append(v)              # a kernel already accumulated into the view: add
            else:
                src.append(g); dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        for d_, s_ in zip(adst, asrc):
            d_.add_(s_)
        if keep and (src or asrc):
            from . import mixed
            mixed.side_keep(*src, *asrc)
        for i in b['auto']:
            leaf, p = leaves[i], params[i]
            if leaf is not p:
                leaf.grad = None
            if p.grad is not views[i]:
                p.grad = views[i]

    def mark_backward_start(self):
        """Time origin of ``timeline`` (call right before loss.backward())."""
        self._t0 = time.perf_counter()

    def _now(self):
        return time.perf_counter() - (self._t0 if self._t0 is not None else time.perf_counter())

    def _launch(self, b, final=False):
        from . import mixed
        if b.get('stepped'):
            return                                  # gathered and already updated by the optimizer (early_step)
        cuda = b['flat'].is_cuda
        home = self._home if (self._home is not None and cuda) else None
        overlap = cuda and not final and (self.world > 1 or self.force_overlap)
        if overlap:
            # Mid-backward: the bucket is gathered and reduced on a LAUNCH stream that waits for what has been enqueued so far on
            # the step's stream (autograd-delivered gradients) and on the auxiliary streams (weight-gradient kernels writing this
            # bucket's views) -- the step's stream itself does not wait for anything and goes on with backward.  finish() joins.
            dev = b['flat'].device
            L = mixed.side_stream(dev, kind='launch')
            if L is None:
                overlap = False
        if overlap:
            mixed.fork_into(dev, L, home)
            with torch.cuda.stream(L):
                self._gather(b, keep=True)
                b['launched'] = True
                if self.average and self.world > 1:
                    b['flat'].div_(self.world)
                if self.world > 1:
                    b['handle'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.timeline.append([b['no'], self._now(), None, b['flat'].numel() * 4])
            return
        # finish() (or CPU): on the step's own stream, after it has waited for the auxiliary streams.  (A bucket may complete
        # inside a backward node that autograd runs on another stream: never rely on the current one.)
        with (torch.cuda.stream(home) if home is not None else contextlib.nullcontext()):
            mixed.side_join()
            self._gather(b)
            b['launched'] = True
            if self.world > 1:
                if self.average:
                    b['flat'].div_(self.world)
                b['handle'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.timeline.append([b['no'], self._now(), None, b['flat'].numel() * 4])

    def _issue_ready(self, force=False):
        """issue the collectives of buckets _next, _next+1, ... while they are complete (all, when force)"""
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if not (b['ready'] or force):
                break
            self._launch(b, final=force)
            self._next += 1

    def _arrive(self, b, i):
        if _DEBUG and not b['arrived'][i]:
            import traceback
            b.setdefault('first', {})[i] = "".join(traceback.format_stack(limit=10))
        if b['launched']:
            if _DEBUG:
                print("first arrival of the parameter:\n" + b.get('first', {}).get(i, '?'), flush=True)
            raise RuntimeError(
                f"gradient for parameter {i} (shape {tuple(b['params'][i].shape)}) of bucket {b['no']} arrived after the bucket's all-reduce was issued: more "
                "than one backward() per finish(), or a parameter used both through a gradient-sink kernel and plain "
                "autograd in different backward passes.  Not supported (it would be dropped or race with the collective).")
        if not b['arrived'][i]:
            b['arrived'][i] = True
            b['pending'] -= 1
            if b['pending'] == 0:
                b['ready'] = True
                if self.world > 1 or self.force_overlap:    # one process: nothing to overlap -- all gathered at finish(), after ONE join
                    self._issue_ready()
                elif self.early_step is not None:
                    self._early(b)

    def _pre(self, b, i, g):
        b['real'][i] = g is not None
        return None

    def _on_grad(self, leaf):
        b = self._l2b[id(leaf)]
        i = b['index'][id(leaf)]
        real, b['real'][i] = b['real'][i], False
        if real:
            b['auto'].append(i)
        if not real and b['arrived'][i]:
            return                                      # autograd visited the leaf with no gradient (a sink kernel delivered it)
        self._arrive(b, i)

    def _early(self, b):
        """One process: the bucket's gradients are final -- run the optimizer for its parameters NOW, on the second stream, behind
        a fork (every kernel that still reads these parameters, and every autograd-delivered gradient, is already enqueued on
        the current stream; the weight-gradient kernels that write the bucket's views are on the second stream itself)."""
        from . import mixed
        if not b['flat'].is_cuda:
            return
        dev = b['flat'].device
        side = mixed.side_stream(dev)
        if side is None:
            return
        sp = mixed.fork_to_side(dev)
        with torch.cuda.stream(side):
            self._gather(b, keep=True)
        if self.early_step(b['params'], sp):
            b['launched'] = b['stepped'] = True
        # (False: the optimizer has no tables yet -- first step; the gather above is harmless, finish() repeats nothing: 'auto'
        # gradients were moved into the views and the leaves' .grad cleared)

    def _on_direct(self, b, i):
        """A kernel accumulated parameter i's gradient straight into its bucket view (mixed.grad_sink)."""
        b['direct'][i] = True
        self._arrive(b, i)

    def finish(self):
        """Call after backward: issues the buckets with parameters that got no gradient (in index order), waits for the
        collectives, re-arms the reducer for the next step."""
        from . import mixed
        mixed.flush_pending()
        self._issue_ready(force=True)
        mixed.side_join()
        for b in self.buckets:
            if b['handle'] is not None:
                b['handle'].wait()
                b['handle'] = None
                for rec in self.timeline:
                    if rec[0] == b['no'] and rec[2] is None:
                        rec[2] = self._now()
            b['pending'] = len(b['params'])
            b['arrived'] = [False] * len(b['params'])
            b['direct'] = [False] * len(b['params'])
            b['real'] = [False] * len(b['params'])
            b['auto'] = []
            b['launched'] = b['ready'] = b['stepped'] = False
        self._next = 0

    def zero_grad(self):
        """Before forward: ONE memset per bucket (kernels accumulate into the views) and drop the leaves' grads."""
        from . import mixed
        mixed.reset_step()
        if self.buckets and self.buckets[0]['flat'].is_cuda:
            self._home = torch.cuda.current_stream(self.buckets[0]['flat'].device)      # the stream the step is issued on
        self.timeline = []
        for b in self.buckets:
            b['flat'].zero_()
            for leaf, p, v in zip(b['leaves'], b['params'], b['views']):
                if leaf is not p:
                    leaf.grad = None            # a bf16 shadow: autograd hands it a fresh gradient, gathered into the view
                elif p.grad is not v:
                    p.grad = v                  # an fp32 leaf accumulates in place into its (just zeroed) bucket view

    def release(self):
        from . import mixed
        mixed.wgrad_group_enable(False)
        mixed.clear_sinks(self._sink_ids)
        for h in self._hooks:
            h.remove()

    def broadcast_parameters(self, src=0):
        """Initial parameter sync (DDP does the same at construction)."""
        if self.world > 1:
            for p in self.params:
                d = p.data
                # a channels-last resident conv weight: broadcast its memory as the contiguous (Cout,3,3,Cin) tensor it is
                dist.broadcast(d if d.is_contiguous() else d.permute(0, 2, 3, 1), src, group=self.group)


def sync_gemm_plans(group=None, src=0):
    """Every rank runs the library GEMMs with the algorithms rank ``src`` chose.  swin_gemm_bf16 picks a plan's algorithm by timing
    hipBLASLt's candidates on first use, so ranks can choose differently for the same shape (harmless for the replicas' parameters --
    the gradients are all-reduced -- but it makes per-rank step times and local gradients differ).  Call once after the warm-up
    steps: plans exported on ``src`` (csrc/gemm_lt.hip: swin_gemm_plans_export), broadcast, imported everywhere.
    Returns the number of plans changed on this rank."""
    import ctypes
    from . import _lib
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    lib = _lib.lib()
    n = max(lib.swin_gemm_plans_export(None, 0), 0)
    buf = (ctypes.c_int64 * (6 * max(n, 1)))()
    if n:
        lib.swin_gemm_plans_export(buf, n)
    if world == 1:
        return 0
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    cnt = torch.tensor([n], dtype=torch.int64, device=dev)
    dist.broadcast(cnt, src, group=group)
    m = int(cnt.item())
    rec = torch.tensor(list(buf)[:6 * n] if dist.get_rank(group) == src else [0] * (6 * m), dtype=torch.int64, device=dev).reshape(-1)
    if rec.numel() != 6 * m:
        rec = torch.zeros(6 * m, dtype=torch.int64, device=dev)
    dist.broadcast(rec, src, group=group)
    if m == 0:
        return 0
    host = rec.cpu().tolist()
    arr = (ctypes.c_int64 * (6 * m))(*host)
    return max(lib.swin_gemm_plans_import(arr, m), 0)


def reduce_log_vars(log_vars, group=None):
    """One packed all-reduce(mean) for all logged scalars (vs one per scalar in base.py:211-216)."""
    keys = sorted(log_vars)
    t = torch.stack([log_vars[k].detach().float().reshape(()) for k in keys])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        t = t / dist.get_world_size(group)
        dist.all_reduce(t, group=group)
    return dict(zip(keys, t))
