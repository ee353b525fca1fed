#!/usr/bin/env python
"""Per-kernel timings on the GPU at the BASELINE configs[1] geometries (development aid).
usage: python tools/microbench.py [attn] [conv] [ln] [nms] [roi]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swin_transformer_object_detection_amd import ops  # noqa: E402
from swin_transformer_object_detection_amd.ops import functional as Fn  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3        # median, min in us


STAGES = [(200, 320, 96, 3), (100, 160, 192, 6), (50, 80, 384, 12), (25, 40, 768, 24)]


def attn():
    B = 2
    for (H, W, C, nH) in STAGES:
        for shift in (0, 3):
            qkv = torch.randn(B, H * W, 3 * C, device="cuda").bfloat16()
            qb = torch.randn(3 * C, device="cuda") * 0.1
            table = torch.randn(169, nH, device="cuda") * 0.02
            bias_exp = ops.rel_bias_expand(table)
            out = torch.empty(B, H * W, C, device="cuda", dtype=torch.bfloat16)
            nW = ((H + 6) // 7) * ((W + 6) // 7)
            lse = torch.empty(B * nW * nH, 64, device="cuda")
            sc = 32 ** -0.5

            def f():
                Fn.call("swin_window_attn_fwd", Fn._p(qkv), Fn._p(qb), Fn._p(bias_exp), Fn._p(out), Fn._p(lse), B, H, W, C, nH,
                        shift, sc, Fn.SWIN_BF16, Fn._s())
            med, mn = timeit(f)
            byts = 8.0 * B * H * W * C
            dout = torch.randn_like(out)
            dqkv = torch.empty_like(qkv)
            dbe = torch.zeros_like(bias_exp)
            dpad = torch.zeros(3 * C, device="cuda")
            ws = torch.empty(Fn._lib.lib().swin_window_attn_bwd_workspace_bytes(B, H, W, nH, Fn.SWIN_BF16), device="cuda",
                             dtype=torch.uint8)

            def fb():
                Fn.call("swin_window_attn_bwd", Fn._p(qkv), Fn._p(qb), Fn._p(bias_exp), Fn._p(lse), Fn._p(dout), Fn._p(dqkv),
                        Fn._p(dbe), Fn._p(dpad), Fn._p(ws), B, H, W, C, nH, shift, sc, Fn.SWIN_BF16, Fn._s())
            medb, mnb = timeit(fb)
            print(f"attn {H}x{W} C={C} nH={nH} shift={shift}: fwd {med:8.1f} us ({byts / med / 1e3:7.1f} GB/s)   "
                  f"bwd {medb:8.1f} us ({2 * byts / medb / 1e3:7.1f} GB/s)")


def conv():
    for (N, H, W) in [(2, 200, 320), (2, 100, 160), (2, 50, 80), (2, 25, 40), (2, 13, 20), (200, 14, 14)]:
        x = torch.randn(N, 256, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        w = torch.randn(256, 3, 3, 256, device="cuda").bfloat16() * 0.02
        b = torch.zeros(256, device="cuda")

        def f():
            Fn._conv3x3_raw(x, w, b, False)
        med, mn = timeit(f)
        fl = 2.0 * N * H * W * 256 * 256 * 9
        print(f"conv3x3 {N}x{H}x{W}x256->256: {med:8.1f} us  {fl / med / 1e6:7.1f} TFLOP/s")


def gemmbb():
    """forward (x W^T + b) and data-gradient (dy W) GEMMs of the backbone linears on the library (hipBLASLt)"""
    for (T, C) in [(128000, 96), (32000, 192), (8000, 384), (2000, 768)]:
        for (N, K, name) in [(3 * C, C, "qkv"), (C, C, "proj"), (4 * C, C, "fc1"), (C, 4 * C, "fc2")]:
            a = torch.randn(T, K, device="cuda").bfloat16()
            w = torch.randn(N, K, device="cuda").bfloat16() * 0.02
            bf = torch.zeros(N, device="cuda"); bb = bf.bfloat16()
            c = torch.empty(T, N, device="cuda", dtype=torch.bfloat16)
            dy = torch.randn(T, N, device="cuda").bfloat16()
            lib, _ = timeit(lambda: torch.nn.functional.linear(a, w, bb))
            dg, _ = timeit(lambda: dy @ w)
            byts = 2.0 * (T * K + T * N)
            print(f"T={T:6d} {name:5s} N={N:5d} K={K:5d}: fwd lib {lib:6.1f} us ({byts / lib / 1e3:6.0f} GB/s, {2.0 * T * N * K / lib / 1e6:5.0f} TF)"
                  f"   dgrad lib {dg:6.1f} us")


def wgrad():
    for (T, N1, N2) in [(128000, 288, 96), (128000, 384, 96), (128000, 96, 384), (32000, 768, 192), (8000, 1536, 384), (8000, 1152, 384),
                        (2000, 3072, 768), (2000, 768, 768), (1024, 1024, 12544)]:
        dy = torch.randn(T, N1, device="cuda").bfloat16()
        x = torch.randn(T, N2, device="cuda").bfloat16()
        dw = torch.zeros(N1, N2, device="cuda")
        db = torch.zeros(N1, device="cuda")
        med, _ = timeit(lambda: Fn.call("wgrad_linear_bf16", Fn._p(dy), Fn._p(x), Fn._p(dw), Fn._p(db), T, N1, N2, Fn._s()))
        medt, _ = timeit(lambda: dy.t() @ x)
        print(f"wgrad_linear T={T} {N1}x{N2}: mine {med:7.1f} us ({(dy.numel() + x.numel()) * 2 / med / 1e3:7.1f} GB/s)   hipBLASLt {medt:7.1f} us")
    for (N, H, W) in [(2, 200, 320), (2, 100, 160), (2, 50, 80), (200, 14, 14)]:
        dy = torch.randn(N, H, W, 256, device="cuda").bfloat16()
        x = torch.randn(N, H, W, 256, device="cuda").bfloat16()
        dw = torch.zeros(256, 3, 3, 256, device="cuda")
        db = torch.zeros(256, device="cuda")
        med, _ = timeit(lambda: Fn.call("wgrad_conv3x3_nhwc_bf16", Fn._p(dy), Fn._p(x), Fn._p(dw), Fn._p(db), N, H, W, 256, 256, Fn._s()))
        fl = 2.0 * N * H * W * 256 * 256 * 9
        print(f"wgrad_conv {N}x{H}x{W}: {med:7.1f} us {fl / med / 1e6:7.1f} TFLOP/s")


def wgradp():
    """the step's dominant weight-gradient shapes, few launches each (counter passes)"""
    for (T, N1, N2) in [(128000, 384, 96), (128000, 96, 384), (8000, 1536, 384)]:
        dy = torch.randn(T, N1, device="cuda").bfloat16()
        x = torch.randn(T, N2, device="cuda").bfloat16()
        dw = torch.zeros(N1, N2, device="cuda")
        db = torch.zeros(N1, device="cuda")
        med, _ = timeit(lambda: Fn.call("wgrad_linear_bf16", Fn._p(dy), Fn._p(x), Fn._p(dw), Fn._p(db), T, N1, N2, Fn._s()), n=5, warm=1)
        print(f"wgrad_linear T={T} {N1}x{N2}: {med:7.1f} us")
    for (N, H, W) in [(2, 200, 320), (256, 14, 14)]:
        dy = torch.randn(N, H, W, 256, device="cuda").bfloat16()
        x = torch.randn(N, H, W, 256, device="cuda").bfloat16()
        dw = torch.zeros(256, 3, 3, 256, device="cuda")
        db = torch.zeros(256, device="cuda")
        med, _ = timeit(lambda: Fn.call("wgrad_conv3x3_nhwc_bf16", Fn._p(dy), Fn._p(x), Fn._p(dw), Fn._p(db), N, H, W, 256, 256, Fn._s()), n=5, warm=1)
        fl = 2.0 * N * H * W * 256 * 256 * 9
        print(f"wgrad_conv {N}x{H}x{W}: {med:7.1f} us {fl / med / 1e6:7.1f} TFLOP/s")
        xx = x.permute(0, 3, 1, 2)
        w = torch.randn(256, 3, 3, 256, device="cuda").bfloat16() * 0.02
        med, _ = timeit(lambda: Fn._conv3x3_raw(xx, w, db, False), n=5, warm=1)
        print(f"conv3x3 fwd {N}x{H}x{W}: {med:7.1f} us {fl / med / 1e6:7.1f} TFLOP/s")


def wgradn():
    """the backbone's grouped Linear weight gradients, a stage's worth per launch (qkv / proj / fc1 / fc2 of its blocks), distinct
    operands per problem so that nothing is cache-resident; through swin_wgrad_record / _flush.  With the -DSWIN_DEV library,
    SWIN_WGRAD96=0 sends them to the 128-tile grouped kernel instead (before / after), SWIN_WGRAD96_MODE=0/1 forces the work split."""
    for name, T, C, nblk in (("stage 1", 128000, 96, 2), ("stage 2", 32000, 192, 2), ("stage 3", 8000, 384, 6), ("stage 4", 2000, 768, 2)):
        probs = []
        for _ in range(nblk):
            for (N1, N2) in ((3 * C, C), (C, C), (4 * C, C), (C, 4 * C)):
                dy = (torch.randn(T, N1, device="cuda") * 0.1).bfloat16()
                x = torch.randn(T, N2, device="cuda").bfloat16()
                probs.append((dy, x, torch.zeros(N1, N2, device="cuda"), torch.zeros(N1, device="cuda")))
        nbytes = sum(q[0].numel() + q[1].numel() for q in probs) * 2
        flops = sum(2.0 * T * q[2].numel() for q in probs)

        def run():
            for dy, x, dw, db in probs:
                Fn.call("swin_wgrad_record", Fn._p(dy), Fn._p(x), Fn._p(dw), Fn._p(db), T, dy.shape[1], x.shape[1])
            Fn.call("swin_wgrad_flush", Fn._s())
        med, mn = timeit(run, n=10)
        print(f"wgrad group {name} (T={T}, C={C}, {len(probs)} problems, {nbytes / 1e6:.0f} MB, {flops / 1e9:.0f} GFLOP): {med:7.1f} us (min {mn:.1f})  "
              f"{nbytes / med / 1e6:5.2f} TB/s  {flops / med / 1e6:6.1f} TFLOP/s")


def wgradn3():
    """stage 3's grouped Linear weight gradients only, few launches (counter passes: tools/pmc_run.sh ... wgrad96_kernel)"""
    T, C = 8000, 384
    probs = []
    for _ in range(6):
        for (N1, N2) in ((3 * C, C), (C, C), (4 * C, C), (C, 4 * C)):
            probs.append(((torch.randn(T, N1, device="cuda") * 0.1).bfloat16(), torch.randn(T, N2, device="cuda").bfloat16(),
                          torch.zeros(N1, N2, device="cuda"), torch.zeros(N1, device="cuda")))

    def run():
        for dy, x, dw, db in probs:
            Fn.call("swin_wgrad_record", Fn._p(dy), Fn._p(x), Fn._p(dw), Fn._p(db), T, dy.shape[1], x.shape[1])
        Fn.call("swin_wgrad_flush", Fn._s())
    med, mn = timeit(run, n=5, warm=2)
    print(f"wgrad group stage 3: {med:7.1f} us (min {mn:.1f})")


def rooflinep():
    """the two kernels bench.py prices against the MFMA roofline, at the P2 geometry and in bench.py's cache regime (launches rotate over
    three operand sets, 3 x 131 MB > the 256 MB Infinity Cache): the conv weight gradient (wgrad2_kernel<ConvSrc>) and the halo-staged
    conv (conv_halo_kernel) -- for the counter passes of tools/pmc_run.sh"""
    N, H, W = 2, 200, 320
    sets = [(torch.randn(N, H, W, 256, device="cuda").bfloat16(), torch.randn(N, H, W, 256, device="cuda").bfloat16()) for _ in range(3)]
    dw = torch.zeros(256, 3, 3, 256, device="cuda")
    db = torch.zeros(256, device="cuda")
    w = torch.randn(256, 3, 3, 256, device="cuda").bfloat16() * 0.02
    y = torch.empty(N, H, W, 256, device="cuda", dtype=torch.bfloat16)
    k = [0]

    def wg():
        dy, x = sets[k[0] % 3]; k[0] += 1
        Fn.call("wgrad_conv3x3_nhwc_bf16", Fn._p(dy), Fn._p(x), Fn._p(dw), Fn._p(db), N, H, W, 256, 256, Fn._s())

    def cv():
        _, x = sets[k[0] % 3]; k[0] += 1
        Fn.call("conv3x3_nhwc_bf16", Fn._p(x), Fn._p(w), Fn._p(db), Fn._p(y), N, H, W, 256, 256, 0, Fn._s())
    fl = 2.0 * N * H * W * 256 * 256 * 9
    med, _ = timeit(wg, n=9, warm=3)
    print(f"wgrad_conv {N}x{H}x{W}: {med:7.1f} us {fl / med / 1e6:7.1f} TFLOP/s")
    med, _ = timeit(cv, n=9, warm=3)
    print(f"conv3x3 {N}x{H}x{W}: {med:7.1f} us {fl / med / 1e6:7.1f} TFLOP/s")


def overlap():
    """would a second stream for the weight-gradient kernels pay?  A backward-like chain (data-gradient kernel, then the
    weight-gradient kernel of the same layer) run in one stream vs with the weight gradients on a side stream."""
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()

    def chain(pairs, reps, two):
        ev = torch.cuda.Event()
        for _ in range(reps):
            for dg, wg in pairs:
                dg()
                if two:
                    ev.record(main)
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        wg()
                else:
                    wg()
        if two:
            main.wait_stream(side)

    def report(name, pairs, reps=8):
        a, _ = timeit(lambda: chain(pairs, reps, False), n=10)
        b, _ = timeit(lambda: chain(pairs, reps, True), n=10)
        print(f"{name:34s} one stream {a / reps:8.1f} us   two streams {b / reps:8.1f} us   ({a / b:4.2f}x)")

    # stage-1 Swin linears (T = 128000): dgrad on the library, wgrad mine
    T = 128000
    pairs = []
    for (N, K) in [(288, 96), (96, 96), (384, 96), (96, 384)]:
        dy = torch.randn(T, N, device="cuda").bfloat16(); x = torch.randn(T, K, device="cuda").bfloat16()
        w = torch.randn(N, K, device="cuda").bfloat16() * 0.02
        dx = torch.empty(T, K, device="cuda", dtype=torch.bfloat16)
        dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
        pairs.append((lambda dy=dy, w=w, dx=dx: torch.mm(dy, w, out=dx),
                      lambda dy=dy, x=x, dw=dw, db=db, N=N, K=K: Fn.call("wgrad_linear_bf16", Fn._p(dy), Fn._p(x), Fn._p(dw), Fn._p(db), T, N, K, Fn._s())))
    report("stage-1 linears dgrad + wgrad", pairs)
    T = 8000
    pairs = []
    for (N, K) in [(1152, 384), (384, 384), (1536, 384), (384, 1536)]:
        dy = torch.randn(T, N, device="cuda").bfloat16(); x = torch.randn(T, K, device="cuda").bfloat16()
        w = torch.randn(N, K, device="cuda").bfloat16() * 0.02
        dx = torch.empty(T, K, device="cuda", dtype=torch.bfloat16)
        dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
        pairs.append((lambda dy=dy, w=w, dx=dx: torch.mm(dy, w, out=dx),
                      lambda dy=dy, x=x, dw=dw, db=db, N=N, K=K: Fn.call("wgrad_linear_bf16", Fn._p(dy), Fn._p(x), Fn._p(dw), Fn._p(db), T, N, K, Fn._s())))
    report("stage-3 linears dgrad + wgrad", pairs)
    for (N, H, W) in [(2, 200, 320), (2, 100, 160), (2, 50, 80), (200, 14, 14)]:
        dy = torch.randn(N, H, W, 256, device="cuda").bfloat16()
        x = torch.randn(N, H, W, 256, device="cuda").bfloat16()
        dw = torch.zeros(256, 3, 3, 256, device="cuda"); db = torch.zeros(256, device="cuda")
        w = torch.randn(256, 3, 3, 256, device="cuda").bfloat16() * 0.02
        dyv = dy.permute(0, 3, 1, 2)
        pairs = [(lambda: Fn._conv3x3_raw(dyv, w, None, False),
                  lambda: Fn.call("wgrad_conv3x3_nhwc_bf16", Fn._p(dy), Fn._p(x), Fn._p(dw), Fn._p(db), N, H, W, 256, 256, Fn._s()))]
        report(f"conv3x3 {N}x{H}x{W} dgrad + wgrad", pairs)


def rpn():
    """RPN proposal selection + decode (rpn_select.hip) at the five FPN shapes of 2x800x1280"""
    sizes = [(200, 320), (100, 160), (50, 80), (25, 40), (13, 20)]
    ls = [3 * h * w for h, w in sizes]
    tot = sum(ls)
    for dt in (torch.bfloat16, torch.float32):
        cls = (torch.randn(2, tot, device="cuda") * 0.05).to(dt)
        reg = (torch.randn(2, tot, 4, device="cuda") * 0.1).to(dt)
        anchors = torch.rand(tot, 4, device="cuda") * 500
        anchors[:, 2:] += anchors[:, :2]
        med, mn = timeit(lambda: ops.rpn_topk_decode(cls, reg, anchors, ls, 2000, (0., 0., 0., 0.), (1., 1., 1., 1.), (800, 1280)))
        print(f"rpn_topk_decode {dt}: {med:7.1f} us (min {mn:7.1f})")


def mlp():
    """fused MLP (ts_mlp.hip) against the unfused chain fc1 (library) -> bias+GELU kernel -> fc2 (library)"""
    B = 2
    for (H, W, C, nH) in STAGES[:2]:
        T = B * H * W
        x = torch.randn(T, C, device="cuda").bfloat16()
        w1 = (torch.randn(4 * C, C, device="cuda") * 0.05).bfloat16()
        w2 = (torch.randn(C, 4 * C, device="cuda") * 0.05).bfloat16()
        b1 = torch.randn(4 * C, device="cuda") * 0.1
        b2 = torch.randn(C, device="cuda") * 0.1
        b216 = b2.bfloat16()
        dy = torch.randn(T, C, device="cuda").bfloat16()

        def fused():
            return Fn.mlp_fwd_raw(x, w1, b1, w2, b2)

        def unfused():
            hp = Fn.gemm_bf16(x, w1)
            hh = torch.empty_like(hp)
            Fn.call("swin_bias_gelu_fwd", Fn._p(hp), Fn._p(b1), Fn._p(hh), T, 4 * C, Fn.SWIN_BF16, Fn._s())
            return Fn.gemm_bf16(hh, w2, b216)
        a, b = fused().float(), unfused().float()
        err = (a - b).abs().max().item()
        mf, nf = timeit(fused)
        mu, nu = timeit(unfused)
        fl = 16.0 * T * C * C
        print(f"mlp fwd T={T} C={C}: fused {mf:7.1f} us (min {nf:7.1f}; {fl / mf / 1e6:6.1f} TFLOP/s)  unfused {mu:7.1f} us   max|diff| {err:.4f}")
        if os.environ.get("MLP_BWD", "1") == "1":
            try:
                mb, nb = timeit(lambda: Fn.mlp_bwd_raw(x, dy, w1, b1, w2))
                print(f"mlp bwd T={T} C={C}: fused dgrad {mb:7.1f} us (min {nb:7.1f}; {24.0 * T * C * C / mb / 1e6:6.1f} TFLOP/s)")
            except Exception as ex:          # noqa: BLE001
                print("mlp bwd:", ex)


def ln():
    for (H, W, C, nH) in STAGES:
        x = torch.randn(2 * H * W, C, device="cuda").bfloat16()
        y = torch.randn_like(x)
        w = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
        med, _ = timeit(lambda: ops.add_layer_norm(x, y, None, H * W, w, b))
        xr = x.clone().requires_grad_(True)
        o = ops.layer_norm(xr, w.requires_grad_(True), b.requires_grad_(True))
        g = torch.randn_like(o)
        medb, _ = timeit(lambda: torch.autograd.grad(o, xr, g, retain_graph=True))
        h = torch.randn(2 * H * W, 4 * C, device="cuda").bfloat16()
        bb = torch.zeros(4 * C, device="cuda")
        medg, _ = timeit(lambda: ops.bias_gelu(h, bb))
        print(f"rows={2 * H * W} C={C}: add_ln fwd {med:7.1f} us ({8.0 * x.numel() / med / 1e3:7.1f} GB/s)  ln bwd {medb:7.1f} us  "
              f"bias_gelu {medg:7.1f} us ({4.0 * h.numel() / medg / 1e3:7.1f} GB/s)")


def nms():
    import numpy as np
    rng = np.random.RandomState(0)
    for n in (1000, 4000, 8780):
        xy = rng.rand(n, 2).astype("float32") * 1000
        boxes = torch.from_numpy(np.concatenate([xy, xy + rng.rand(n, 2).astype("float32") * 150 + 4], 1)).cuda()
        scores = torch.from_numpy(rng.rand(n).astype("float32")).cuda()
        ids = torch.from_numpy(rng.randint(0, 5, n)).cuda()
        med, _ = timeit(lambda: ops.batched_nms(boxes, scores, ids, dict(type="nms", iou_threshold=0.7)), n=10)
        print(f"batched_nms n={n}: {med:8.1f} us")


def roi():
    feats = [torch.randn(2, 256, 200 // s, 320 // s, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
             for s in (1, 2, 4, 8)]
    rois = torch.rand(1024, 5, device="cuda")
    rois[:, 0] = (rois[:, 0] > 0.5).float()
    rois[:, 1:3] *= 600
    rois[:, 3:] = rois[:, 1:3] + rois[:, 3:] * 400 + 8
    for lvl, sc in enumerate((0.25, 0.125, 0.0625, 0.03125)):
        x = feats[lvl].clone().requires_grad_(True)
        med, _ = timeit(lambda: ops.roi_align(x, rois, 7, sc, 0, 'avg', True))
        o = ops.roi_align(x, rois, 7, sc, 0, 'avg', True)
        g = torch.randn_like(o)
        medb, _ = timeit(lambda: torch.autograd.grad(o, x, g, retain_graph=True), n=10)
        print(f"roi_align 1024 rois lvl{lvl}: fwd {med:7.1f} us  bwd {medb:7.1f} us")


def lnbwd():
    for (H, W, C, nH) in STAGES:
        rows = 2 * H * W
        x = torch.randn(rows, C, device="cuda").bfloat16(); dy = torch.randn_like(x); dres = torch.randn_like(x)
        w = torch.ones(C, device="cuda"); mean = torch.zeros(rows, device="cuda"); rstd = torch.ones(rows, device="cuda")
        dx = torch.empty_like(x); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
        ws = Fn._ln_ws(rows, C, x)
        med, _ = timeit(lambda: Fn.call("swin_layernorm_bwd", Fn._p(dy), Fn._p(x), Fn._p(w), Fn._p(mean), Fn._p(rstd), Fn._p(dres),
                                        Fn._p(dx), None, None, 1, Fn._p(dg), Fn._p(db), rows, C, 1, Fn._p(ws), Fn._s()))
        print(f"ln_bwd rows={rows} C={C}: {med:7.1f} us ({8.0 * x.numel() / med / 1e3:7.1f} GB/s)")


def gelu():
    for (H, W, C, nH) in STAGES:
        rows, C4 = 2 * H * W, 4 * C
        h = torch.randn(rows, C4, device="cuda").bfloat16()
        bb = torch.zeros(C4, device="cuda")
        medf, _ = timeit(lambda: ops.bias_gelu(h, bb))
        g = torch.randn_like(h); dx = torch.empty_like(h); db = torch.zeros(C4, device="cuda")
        medb, _ = timeit(lambda: Fn.call("swin_bias_gelu_bwd", Fn._p(g), Fn._p(h), Fn._p(bb), Fn._p(dx), Fn._p(db), rows, C4, 1, Fn._s()))
        print(f"bias_gelu rows={rows} C={C4}: fwd {medf:7.1f} us ({4.0 * h.numel() / medf / 1e3:7.1f} GB/s)  "
              f"bwd {medb:7.1f} us ({6.0 * h.numel() / medb / 1e3:7.1f} GB/s)")


def roig():
    """RoIAlign backward of one R-CNN stage at the bench geometry: scatter form (2 launches + zero fill + cast) against the gather form"""
    import ctypes
    from swin_transformer_object_detection_amd import _lib
    torch.manual_seed(0)
    N, C = 2, 256
    shapes = [(200, 320), (100, 160), (50, 80), (25, 40)]
    n = 4
    sets = []
    for K, out in ((1024, 7), (256, 14)):
        rois = torch.rand(K, 5, device="cuda")
        rois[:, 0] = (torch.arange(K, device="cuda") >= K // 2).float()
        wh = torch.exp(torch.rand(K, 2, device="cuda") * 4.0 + 2.5)          # 12 .. 660 px
        rois[:, 1:3] = rois[:, 1:3] * torch.tensor([1280., 800.], device="cuda") * 0.8
        rois[:, 3:] = torch.minimum(rois[:, 1:3] + wh, torch.tensor([1279., 799.], device="cuda"))
        if len(sys.argv) > 2 and sys.argv[2] == "hot":                       # 16 objects, every RoI a jitter of one of them
            obj = rois[torch.randint(0, 16, (K,), device="cuda")]
            rois[:, 1:] = obj[:, 1:] + torch.randn(K, 4, device="cuda") * 6.0
            rois[:, 3:] = torch.maximum(rois[:, 3:], rois[:, 1:3] + 8.0)
        scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
        lv = torch.floor(torch.log2(scale / 56 + 1e-6)).clamp(0, 3).int()
        g = torch.randn(K, out, out, C, device="cuda").bfloat16()
        sets.append((rois, lv, g, out))
    Hs = (ctypes.c_int * n)(*[s[0] for s in shapes]); Ws = (ctypes.c_int * n)(*[s[1] for s in shapes])
    sc = (ctypes.c_float * n)(0.25, 0.125, 0.0625, 0.03125)
    sizes = [N * h * w * C for h, w in shapes]

    def scatter():
        flat = torch.zeros(sum(sizes), device="cuda")
        offs = [sum(sizes[:i]) for i in range(n)]
        ptrs = (ctypes.c_void_p * n)(*[flat.data_ptr() + 4 * o for o in offs])
        for r, lv, g, out in sets:
            Fn.call("roi_align_multilevel_bwd", ptrs, Hs, Ws, sc, n, Fn._p(g), Fn._p(r), Fn._p(lv), C, r.shape[0], out, out, 0, 1, 1, Fn._s())
        return flat.bfloat16()
    ktot = sum(s[0].shape[0] for s in sets)
    nb = int(_lib.lib().roi_align_gather_workspace_bytes(Hs, Ws, n, N, ktot))
    ws = torch.zeros(nb // 4 + 1, device="cuda", dtype=torch.int32)
    m = len(sets)
    gp = (ctypes.c_void_p * m)(*[s[2].data_ptr() for s in sets]); rp = (ctypes.c_void_p * m)(*[s[0].data_ptr() for s in sets])
    lp = (ctypes.c_void_p * m)(*[s[1].data_ptr() for s in sets])
    Ks = (ctypes.c_int * m)(*[s[0].shape[0] for s in sets]); ps = (ctypes.c_int * m)(*[s[3] for s in sets])

    def gather():
        flat = torch.empty(sum(sizes), device="cuda", dtype=torch.bfloat16)
        offs = [sum(sizes[:i]) for i in range(n)]
        ptrs = (ctypes.c_void_p * n)(*[flat.data_ptr() + 2 * o for o in offs])
        Fn.call("roi_align_multilevel_bwd_gather", ptrs, Hs, Ws, sc, n, N, m, gp, rp, lp, Ks, ps, ps, C, 0, 1, 1, 1, Fn._p(ws), nb, Fn._s())
        return flat
    a, b = scatter(), gather()
    torch.cuda.synchronize()
    total = 2 * sum(((h + 7) // 8) * ((w + 7) // 8) for h, w in shapes)
    w_ = ws.cpu()
    per = (w_[2 * total + 1:3 * total + 1] - w_[2 * total:3 * total])
    err = float((a.float() - b.float()).abs().max() / a.float().abs().max())
    ms, _ = timeit(scatter, n=10)
    mg, _ = timeit(gather, n=10)
    print(f"roi backward, 1024 x 7x7 + 256 x 14x14 RoIs: scatter {ms:7.1f} us   gather {mg:7.1f} us   max rel diff {err:.1e}   "
          f"(tile, RoI) pairs {int(w_[3 * total])}, max per tile {int(per.max())}, tiles {total}, empty {int((per == 0).sum())}")


def roiml():
    import ctypes
    torch.manual_seed(0)
    feats = [torch.randn(2, 256, 200 // s, 320 // s, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
             for s in (1, 2, 4, 8)]
    grads = [torch.zeros(f.shape, device="cuda").contiguous(memory_format=torch.channels_last) for f in feats]
    for K, out in ((1024, 7), (256, 14)):
        rois = torch.rand(K, 5, device="cuda")
        rois[:, 0] = (torch.arange(K, device="cuda") >= K // 2).float()
        wh = torch.exp(torch.rand(K, 2, device="cuda") * 4.0 + 2.5)          # 12 .. 660 px
        rois[:, 1:3] = rois[:, 1:3] * torch.tensor([1280., 800.], device="cuda") * 0.8
        rois[:, 3:] = torch.minimum(rois[:, 1:3] + wh, torch.tensor([1279., 799.], device="cuda"))
        scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
        lv = torch.floor(torch.log2(scale / 56 + 1e-6)).clamp(0, 3).int()
        med, _ = timeit(lambda: ops.roi_align_multilevel(feats, rois, lv, out, [4, 8, 16, 32], 0, True))
        g = torch.randn(K, 256, out, out, device="cuda").contiguous(memory_format=torch.channels_last)
        ptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in grads])
        Hs = (ctypes.c_int * 4)(*[t.shape[2] for t in grads]); Ws = (ctypes.c_int * 4)(*[t.shape[3] for t in grads])
        sc = (ctypes.c_float * 4)(0.25, 0.125, 0.0625, 0.03125)
        medb, _ = timeit(lambda: Fn.call("roi_align_multilevel_bwd", ptrs, Hs, Ws, sc, 4, Fn._p(g), Fn._p(rois), Fn._p(lv), 256, K, out, out, 0, 1, 0, Fn._s()), n=10)
        print(f"roi_align_multilevel K={K} out={out}: fwd {med:7.1f} us  bwd {medb:7.1f} us  levels {torch.bincount(lv, minlength=4).tolist()}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["attn", "conv", "ln", "nms", "roi"]
    for w in which:
        globals()[w]()
