"""The training step replayed as one hipGraph launch (graph_step.GraphedTrainStep) against the same step issued eagerly:
same parameters after the same steps; per-step scalars (learning rate, Adam bias corrections) and the samplers' randomness
follow the host although the launches' arguments are frozen into the graph."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import swin_transformer_object_detection_amd as p
    return p


def _setup(drop_path=0.0, hw=(256, 320)):
    from swin_transformer_object_detection_amd import data, ddp, detector, mixed, presets
    from swin_transformer_object_detection_amd.optim import FusedAdamW
    cfg = presets.mask_rcnn_swin("tiny")
    cfg["backbone"]["drop_path_rate"] = drop_path
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    model = detector.build_detector(cfg, compute_dtype=torch.bfloat16).to(dev).train()
    sh = mixed.ShadowParams(model, torch.bfloat16)
    red = ddp.BucketedGradReducer(model.parameters_in_forward_order(), leaf_of=sh.leaf_of)
    opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.05)
    batch = data.synthetic_batch(2, hw[0], hw[1], dev, seed=3, num_boxes=5)
    return model, sh, red, opt, batch


def _snapshot(model, opt):
    return ([p.detach().clone() for p in model.parameters()],
            {p: (st['exp_avg'].clone(), st['exp_avg_sq'].clone()) for p, st in opt.state.items()}, opt.step_count)


def _restore(model, opt, snap):
    from swin_transformer_object_detection_amd import mixed
    ps, st, n = snap
    with torch.no_grad():
        for p, q in zip(model.parameters(), ps):
            p.copy_(q)
        for p, (m, v) in st.items():
            opt.state[p]['exp_avg'].copy_(m)
            opt.state[p]['exp_avg_sq'].copy_(v)
    opt.step_count = n
    mixed.refresh_all()


def test_graph_replay_equals_eager_steps(pkg, monkeypatch):
    from swin_transformer_object_detection_amd import mixed
    from swin_transformer_object_detection_amd.graph_step import GraphedTrainStep
    from swin_transformer_object_detection_amd.ops import targets
    monkeypatch.setattr(targets, "_next_seed", lambda: 0x1234567)     # the same host AND device seed for every call and step
    model, sh, red, opt, batch = _setup()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(st):
            g = GraphedTrainStep(model, red, opt, warmup=2)
            g(batch)                                   # warm-up, capture, first replay
            assert g.graphs() == 1
            snap = _snapshot(model, opt)

            def run(fn, n=3):
                _restore(model, opt, snap)
                for _ in range(n):
                    lv = fn(batch)
                torch.cuda.synchronize()
                return [p.detach().clone() for p in model.parameters()], {k: float(v) for k, v in lv.items()}
            pe, le = run(g.eager)
            pe2, le2 = run(g.eager)                    # run-to-run noise of the float atomics
            pg, lg = run(g)
            assert g.graphs() == 1
        assert all(v == v for v in lg.values())
        # AdamW moves an element by ~lr per step whatever the size of its gradient, so an element whose gradient is noise (float
        # atomics arrive in another order on one stream than on two) can end anywhere within +-steps*lr: tensors are compared in
        # the L2 norm, against the run-to-run noise of two eager runs and the distance the steps moved the tensor
        worst = 0.0
        for a, a2, b, ref in zip(pe, pe2, pg, snap[0]):
            moved = float((a - ref).norm())
            noise = float((a - a2).norm())
            err = float((a - b).norm())
            assert err <= 4 * noise + 0.25 * moved + 1e-7, (tuple(a.shape), err, noise, moved)
            worst = max(worst, moved)
        assert worst > 0.0                             # the steps really changed the parameters
        for k in le:       # the third step's losses: three bf16 steps apart, the two eager runs already differ by several per cent
            assert abs(le[k] - lg[k]) <= 4 * abs(le[k] - le2[k]) + 0.1 * abs(le[k]) + 1e-3, (k, le[k], le2[k], lg[k])
    finally:
        targets.disable_step_seed()
        red.release()
        sh.release()


def test_graph_replay_reads_this_steps_learning_rate(pkg):
    """lr lives in the optimizer's device-resident state, not in the captured launch's arguments: lr = 0 before a replay leaves the
    parameters untouched, the original lr moves them again."""
    from swin_transformer_object_detection_amd.graph_step import GraphedTrainStep
    from swin_transformer_object_detection_amd.ops import targets
    model, sh, red, opt, batch = _setup()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(st):
            g = GraphedTrainStep(model, red, opt, warmup=2)
            g(batch)
            lr0 = [grp['lr'] for grp in opt.param_groups]
            wd0 = [grp['weight_decay'] for grp in opt.param_groups]
            before = [p.detach().clone() for p in model.parameters()]
            for grp in opt.param_groups:
                grp['lr'] = 0.0
            g(batch)
            torch.cuda.synchronize()
            assert all(torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
            for grp, lr, wd in zip(opt.param_groups, lr0, wd0):
                grp['lr'], grp['weight_decay'] = lr, wd
            n = opt.step_count
            g(batch)
            torch.cuda.synchronize()
            assert opt.step_count == n + 1
            assert any(not torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
    finally:
        targets.disable_step_seed()
        red.release()
        sh.release()


def test_captured_sampler_follows_the_device_seed(pkg):
    """det_random_sample inside a graph: the host seed is frozen, the device-resident step seed still changes the draw."""
    from swin_transformer_object_detection_amd import ops
    from swin_transformer_object_detection_amd.ops import targets
    dev = torch.device("cuda", 0)
    assigned = (torch.arange(20000, device=dev) % 50 == 0).long()            # 400 positives, the rest negatives
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(st):
            targets.set_step_seed(dev, 1)
            ops.random_sample_raw(assigned, 256, 0.25)                       # warm-up (workspace sizes, first launches)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                inds, flags = ops.random_sample_raw(assigned, 256, 0.25)
            outs = []
            for seed in (1, 2, 2, 3):
                targets.set_step_seed(dev, seed)
                g.replay()
                torch.cuda.synchronize()
                outs.append((inds.clone(), flags.clone()))
        assert torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1])      # same seed, same sample
        assert not torch.equal(outs[0][0], outs[1][0]) and not torch.equal(outs[2][0], outs[3][0])
        for i, f in outs:
            assert int((f >= 2).sum()) == 64 and int((f >= 1).sum()) == 256
            pos = i[f >= 2]
            assert bool((assigned[pos] == 1).all()) and pos.unique().numel() == pos.numel()
            assert i[f >= 1].unique().numel() == 256
    finally:
        targets.disable_step_seed()
