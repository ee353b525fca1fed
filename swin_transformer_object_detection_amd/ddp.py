"""Data-parallel gradient exchange: bucketed all-reduce over RCCL, overlapped with backward.

Replaces the reference's ``MMDistributedDataParallel`` wrap (``mmdet/apis/train.py:91-99``; torch DDP with
``broadcast_buffers=False``) and its per-scalar loss all-reduce (``mmdet/models/detectors/base.py:211-216``).

Design for MI355X / xGMI (one process per GPU, backend "nccl" == RCCL):
  * gradients live in a few large flat fp32 buffers (``param.grad`` are views), so a bucket is reduced
    in place with no gather/scatter copies;
  * buckets are formed in reverse registration order (heads -> FPN -> Swin stage 4..1), which is the
    order backward produces them; a post-accumulate hook counts arrivals and launches the bucket's
    async all-reduce as soon as it is complete, so communication hides under the remaining backward;
  * bucket size defaults to 48 MiB: xGMI links are point-to-point (~153 GB/s each), a few large
    messages keep every link busy and the per-collective launch cost negligible (Mask R-CNN Swin-T:
    192 MB of fp32 gradients -> 4 buckets);
  * the loss scalars for logging are packed into ONE tensor and reduced once, without .item().
Works unchanged on CPU with the gloo backend (tests/test_ddp_gloo.py, world_size 2).
"""
import torch
import torch.distributed as dist


class BucketedGradReducer:
    def __init__(self, params, bucket_bytes=48 << 20, process_group=None, average=True):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.average = average
        self.buckets = []          # list of dict(flat, params, pending, handle)
        self._p2b = {}
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_bytes:
                self._make_bucket(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._make_bucket(cur)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _make_bucket(self, plist):
        n = sum(p.numel() for p in plist)
        dev = plist[0].device
        flat = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p in plist:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        b = dict(flat=flat, params=list(plist), pending=len(plist), handle=None)
        for p in plist:
            self._p2b[p] = b
        self.buckets.append(b)

    def _launch(self, b):
        if self.world > 1 and b['handle'] is None:
            if self.average:
                b['flat'].div_(self.world)
            b['handle'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        b = self._p2b[p]
        if p.grad.data_ptr() != b['flat'].data_ptr() + self._offset(b, p):
            # autograd replaced the view (first accumulation into a None grad): copy back into the bucket
            self._view(b, p).copy_(p.grad)
            p.grad = self._view(b, p)
        b['pending'] -= 1
        if b['pending'] == 0:
            self._launch(b)

    @staticmethod
    def _offset(b, p):
        off = 0
        for q in b['params']:
            if q is p:
                return off * 4
            off += q.numel()
        raise KeyError

    def _view(self, b, p):
        off = self._offset(b, p) // 4
        return b['flat'][off:off + p.numel()].view_as(p)

    def finish(self):
        """Call after backward: launches buckets whose parameters got no gradient this step and waits."""
        for b in self.buckets:
            if b['pending'] > 0:
                self._launch(b)
        for b in self.buckets:
            if b['handle'] is not None:
                b['handle'].wait()
                b['handle'] = None
            b['pending'] = len(b['params'])

    def zero_grad(self):
        for b in self.buckets:
            b['flat'].zero_()

    def broadcast_parameters(self, src=0):
        """Initial parameter sync (DDP does the same at construction)."""
        if self.world > 1:
            for p in self.params:
                dist.broadcast(p.data, src, group=self.group)


def reduce_log_vars(log_vars, group=None):
    """One packed all-reduce(mean) for all logged scalars (vs one per scalar in base.py:211-216)."""
    keys = sorted(log_vars)
    t = torch.stack([log_vars[k].detach().float().reshape(()) for k in keys])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        t = t / dist.get_world_size(group)
        dist.all_reduce(t, group=group)
    return dict(zip(keys, t))
