"""The data-parallel exchange (ddp.BucketedGradReducer) on CPU with the gloo backend, world_size 2:
gradients after finish() equal the mean of the per-rank gradients, buckets launch during backward, unused
parameters do not hang, the packed log-var reduction averages, zero_grad keeps the flat views."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from swin_transformer_object_detection_amd import ddp
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(16, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 4))
    unused = nn.Linear(3, 3)                                  # never used in forward
    params = list(model.parameters()) + list(unused.parameters())
    if rank == 1:                                             # ranks start different: broadcast must fix it
        with torch.no_grad():
            for p in params:
                p.add_(1.0)
    red = ddp.BucketedGradReducer(params, bucket_bytes=1024)         # several buckets
    red.broadcast_parameters()
    assert len(red.buckets) >= 3
    ref = nn.Sequential(nn.Linear(16, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 4))
    ref.load_state_dict(model.state_dict())
    g = torch.Generator().manual_seed(100)
    xs = [torch.randn(8, 16, generator=g) for _ in range(world)]
    ok = True
    for it in range(2):
        red.zero_grad()
        loss = model(xs[rank]).square().mean()
        loss.backward()
        red.finish()
        # expected: mean over ranks of the gradient, computed locally from all ranks' data
        ref.zero_grad()
        sum(ref(x).square().mean() for x in xs).div(world).backward()
        for p, r in zip(model.parameters(), ref.parameters()):
            ok &= torch.allclose(p.grad, r.grad, atol=1e-6)
            flat = red._l2b[id(p)]['flat']
            ok &= flat.data_ptr() <= p.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4   # a view of the bucket
        for p in unused.parameters():
            ok &= float(p.grad.abs().max()) == 0.0
    logs = ddp.reduce_log_vars({"loss_a": torch.tensor(float(rank)), "loss_b": torch.tensor(2.0)})
    ok &= abs(float(logs["loss_a"]) - 0.5) < 1e-6 and abs(float(logs["loss_b"]) - 2.0) < 1e-6
    same = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(same, list(model.parameters())[0].detach().sum().view(1))
    ok &= bool(torch.allclose(same[0], same[1]))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bucketed_reducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _syncbn_worker(rank, world, port, q):
    """SyncBN of the Cascade heads' ConvModules (detector._bn_act, CPU form of ops.batch_norm): with the batch split over
    two ranks, outputs, input gradients and running statistics equal plain BatchNorm over the whole batch; the parameter
    gradients are each rank's own share (the bucketed reducer then averages them, as DDP does for SyncBatchNorm)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch.nn.functional as F
    from swin_transformer_object_detection_amd import detector
    from swin_transformer_object_detection_amd.fpn import ConvModule
    g = torch.Generator().manual_seed(7)
    C = 8
    x = torch.randn(6, C, 7, 7, generator=g) * 2 + 0.5
    dy = torch.randn(6, C, 7, 7, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    cm = ConvModule(C, C, 3, padding=1, norm_cfg=dict(type='SyncBN', requires_grad=True))
    with torch.no_grad():
        cm.bn.weight.copy_(gamma); cm.bn.bias.copy_(beta)
    cm.train()
    sl = slice(0, 2) if rank == 0 else slice(2, 6)                 # uneven split: counts must be exchanged too
    xl = x[sl].clone().requires_grad_(True)
    y = detector._bn_act(xl, cm, True, relu=True)
    y.backward(dy[sl])
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    yr = F.relu(F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5))
    yr.backward(dy)
    ok = torch.allclose(y, yr[sl].detach(), atol=1e-5) and torch.allclose(xl.grad, xr.grad[sl], atol=1e-5)
    ok &= torch.allclose(cm.bn.running_mean, rm, atol=1e-6) and torch.allclose(cm.bn.running_var, rv, atol=1e-5)
    both = [torch.zeros(2 * C) for _ in range(world)]
    dist.all_gather(both, torch.cat([cm.bn.weight.grad, cm.bn.bias.grad]))
    ok &= torch.allclose(both[0] + both[1], torch.cat([gr.grad, br.grad]), atol=1e-4)
    ok &= int(cm.bn.num_batches_tracked) == 1
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sync_batch_norm_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _order_worker(rank, world, port, q):
    """A parameter that only rank 1 uses sits in the FIRST bucket: on rank 0 that bucket never completes during backward
    while the later buckets do, i.e. the completion order differs between the ranks.  Collectives are issued in bucket-index
    order on both ranks all the same: no hang, and every gradient is the mean over ranks.  A second backward() before
    finish() must raise instead of dropping / racing gradients."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from swin_transformer_object_detection_amd import ddp
    torch.manual_seed(0)
    trunk = nn.Sequential(nn.Linear(16, 32), nn.ReLU(), nn.Linear(32, 32))
    head = nn.Linear(32, 4)
    extra = nn.Linear(32, 4)                                  # used by rank 1 only; registered last -> bucket 0
    params = list(trunk.parameters()) + list(head.parameters()) + list(extra.parameters())
    red = ddp.BucketedGradReducer(params, bucket_bytes=256)
    red.broadcast_parameters()
    assert id(extra.weight) in [id(p) for p in red.buckets[0]['params']] and len(red.buckets) >= 4

    def loss_of(r, x):
        f = trunk(x)
        out = head(f)
        if r == 1:
            out = out + extra(f)
        return out.square().mean()
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(8, 16, generator=g) for _ in range(world)]
    red.zero_grad()
    red.mark_backward_start()
    loss_of(rank, xs[rank]).backward()
    order = [rec[0] for rec in red.timeline]                  # buckets issued DURING backward
    red.finish()
    ok = [rec[0] for rec in red.timeline] == list(range(len(red.buckets)))          # index order, all issued
    ok &= all(rec[2] is not None and rec[2] >= rec[1] for rec in red.timeline)
    if rank == 0:
        ok &= 0 not in order                                  # bucket 0 could only be issued by finish() on rank 0 ...
        ok &= len(order) == 0                                 # ... and it held back every later bucket
    got = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    want = torch.autograd.grad(sum(loss_of(r, xs[r]) for r in range(world)) / world, params, allow_unused=True)
    for a, b in zip(got, want):
        ok &= torch.allclose(a, b if b is not None else torch.zeros_like(a), atol=1e-6)
    # second backward without finish(): raises as soon as a gradient arrives for a bucket whose collective is in flight
    # (rank 1: bucket 0 was issued during the first backward; rank 0 issued nothing yet, so plain accumulation is still safe)
    red.zero_grad()
    loss_of(rank, xs[rank]).backward()
    issued = len(red.timeline) > 0
    raised = False
    try:
        loss_of(rank, xs[rank]).backward()
    except RuntimeError as e:
        raised = "arrived after" in str(e)
    ok &= raised == issued and issued == (rank == 1)
    red.finish()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_collectives_in_index_order_with_rank_dependent_graph():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_order_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
