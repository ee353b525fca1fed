#!/usr/bin/env python
"""bench.py -- Mask R-CNN Swin-T 800x1280 bf16 training throughput on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = forward_train + backward + gradient all-reduce (RCCL, overlapped) + AdamW step on a fixed
synthetic batch of 2 images per GPU (BASELINE.json configs[1]; weak scaling).  Rank 0 prints ONE JSON
line.  Besides the contract keys it carries
  roofline     : the step's dominant kernel -- the 3x3-conv weight gradient at the P2 geometry of this workload -- timed live
                 with HIP events on the launch stream, cache-cold, against the MFMA roofline; `traffic` = HBM
                 bytes per launch from the committed rocprofv3 --pmc passes in the same regime;
  roofline_kernels : the same measurement for the WindowAttention core (HBM-bound; with the BASELINE metric's "MFMA util%") and for
                 the other MFMA-bound kernels with large shares of the step (halo-staged 3x3 conv, linear weight gradient, fused MLP);
  cpu_baseline : the CPU oracle (torch-CPU fp32 restatement + C RoIAlign/NMS) timed on the host cores
                 on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IMG_H, IMG_W, PER_GPU_BATCH = 800, 1280, 2


def _HD():
    """torch dtype of the loaded library's 16-bit type"""
    from swin_transformer_object_detection_amd import _lib
    return _lib.half_dtype()
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"],
                    help="bf16: the headline metric.  fp16: the reference's own mixed precision (apex O1, BASELINE configs[4]) -- the "
                         "libswin_hip_f16.so build of the same kernels plus dynamic loss scaling on the device")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the whole step as ONE hipGraph launch (graph_step.GraphedTrainStep).  auto: on one GPU (the "
                         "multi-rank step, whose all-reduces overlap backward, is issued eagerly)")
    ap.add_argument("--workload", default="mask_rcnn_swin_t", choices=sorted(WORKLOADS),
                    help="default: BASELINE.json configs[1] (the headline metric); the others are the BASELINE parity configurations, "
                         "timed for DESIGN.md only")
    ap.add_argument("--no-check-sync", action="store_true",
                    help="skip the replica check that runs by default on N > 1 (all ranks must hold bit-identical parameters "
                         "after the timed steps)")
    return ap.parse_args()


def build_param_groups(model, opt_cfg):
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if any(k in n for k in opt_cfg["no_decay_keys"]) else decay).append(p)
    return [dict(params=decay, weight_decay=opt_cfg["weight_decay"]), dict(params=no_decay, weight_decay=0.0)]


def attention_roofline(device, steps=30):
    """WindowAttention forward (bf16 MFMA kernel) at the stage-1 geometry of the workload:
    B=2, 200x320 tokens, C=96, 3 heads, shifted.  Algorithmic bytes per launch = read qkv (3C) +
    write out (C) per real token in bf16 = 8*T*C bytes (4*T*C*bpe)."""
    from swin_transformer_object_detection_amd import ops
    B, H, W, C, nH = PER_GPU_BATCH, IMG_H // 4, IMG_W // 4, 96, 3
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B, H * W, 3 * C, generator=g).to(device=device, dtype=_HD())
    qb = (torch.randn(3 * C, generator=g) * 0.1).to(device)
    table = (torch.randn(169, nH, generator=g) * 0.02).to(device)
    for _ in range(5):
        ops.window_attention(qkv, qb, table, B, H, W, nH, 3)
    torch.cuda.synchronize()
    # the op wrapper also launches the (tiny) bias-expand kernel; time the attention launch alone through the ABI
    from swin_transformer_object_detection_amd.ops import functional as Fn
    bias_exp = ops.rel_bias_expand(table)
    out = torch.empty(B, H * W, C, device=device, dtype=_HD())
    lse = torch.empty(B * ((H + 6) // 7) * ((W + 6) // 7) * nH, 64, device=device, dtype=torch.float32)
    scale = 32 ** -0.5
    # cache-cold: the launches rotate over four input / output sets (4 x 98 MB > the 256 MB Infinity Cache), so every launch
    # reads its qkv from HBM as it does inside the step (where ~100 MB of other tensors pass between two attention launches)
    sets = [(qkv, out, lse)] + [(qkv.clone(), torch.empty_like(out), torch.empty_like(lse)) for _ in range(3)]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for i, (s, e) in enumerate(ev):
        q_, o_, l_ = sets[i % 4]
        s.record()
        Fn.call("swin_window_attn_fwd", Fn._p(q_), Fn._p(qb), Fn._p(bias_exp), Fn._p(o_), Fn._p(l_), B, H, W, C, nH, 3,
                scale, Fn.SWIN_BF16, Fn._s())
        e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in ev)
    avg_ms = sum(ms) / len(ms)
    alg_bytes = 8.0 * B * H * W * C
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    # HBM traffic per launch from the PMC passes committed under profiles/ (FETCH_SIZE / WRITE_SIZE, separate
    # rocprofv3 --pmc runs of the same kernel and geometry, gfx950 correction applied); null if not collected
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r02_pmc_attn_traffic.json")) as f:
            traffic = round(json.load(f)["kernels"]["win_attn_fwd_bf16@stage1_wpb3"]["hbm_bytes_per_launch_corrected"])
    except Exception:
        pass
    # the BASELINE metric also names "WindowAttn MFMA util%": the core is HBM-bound (AI ~ 24 flop/B, ridge ~ 310), so this is
    # small by construction.  useful flops = 4*B_*49^2*C (QK^T + PV); issued = the 49->64 padded tiles actually executed.
    n_win = B * ((H + 6) // 7) * ((W + 6) // 7)
    useful_tflops = 4.0 * n_win * 49 * 49 * C / (avg_ms * 1e-3) / 1e12
    issued_tflops = 4.0 * n_win * 64 * 64 * C / (avg_ms * 1e-3) / 1e12
    return {"kernel": "win_attn_fwd_bf16_kernel", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "avg_launch_ms": round(avg_ms, 5), "median_launch_ms": round(ms[len(ms) // 2], 5),
            "algorithmic_bytes_per_launch": alg_bytes, "cache": "cold (launches rotate over 4 x 98 MB of inputs/outputs)",
            "step_share": step_shares({"a": "win_attn_fwd"}).get("a"), "step_share_source": PROFILE_NOTE,
            "mfma_util": {"useful_tflops": round(useful_tflops, 1), "issued_tflops": round(issued_tflops, 1),
                          "issued_frac_of_dense_bf16_peak": round(issued_tflops / MFMA_PEAK_TFLOPS, 4)},
            "shape": {"B": B, "H": H, "W": W, "C": C, "heads": nH, "shift": 3, "windows": B * 29 * 46}}


# rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 3 --graph off` (two-stream eager step), committed under profiles/
PROFILE_STATS = os.path.join(ROOT, "profiles", "r03_step_kernel_stats.csv")
PROFILE_NOTE = "share of the kernel time of profiles/r03_step_kernel_stats.csv (committed rocprofv3 summary of this bench command), not measured in this run"
PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_roofline.json")      # FETCH_SIZE / WRITE_SIZE passes of tools/microbench.py rooflinep


def step_shares(patterns):
    """share of the profiled step's kernel time taken by the kernels whose name contains each pattern (committed rocprofv3
    summary; bench.py's own roofline launches are in that trace too -- a few dozen launches against 13 steps)."""
    import csv
    try:
        rows = list(csv.DictReader(open(PROFILE_STATS)))
    except OSError:
        return {}
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    out = {}
    for key, pat in patterns.items():
        ns = sum(float(r["TotalDurationNs"]) for r in rows if pat in r["Name"])
        out[key] = round(ns / total, 4) if total > 0 else None
    return out


def pmc_traffic(key):
    """HBM bytes per launch from the committed counter passes (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE, separate
    rocprofv3 --pmc runs in the same cold-cache regime as the timing below); None if not collected"""
    try:
        with open(PMC_FILE) as f:
            return round(json.load(f)["kernels"][key]["hbm_bytes_per_launch_corrected"])
    except Exception:
        return None


def _time_launches(fn, steps):
    """(average, median) launch duration in ms, HIP events on the launch stream.  The launches are enqueued back to back behind three
    warm-up launches, an event between every two: the host runs ahead of the GPU, so an interval is one launch's duration and a host
    hiccup between two enqueues (seen on one box: a 10x outlier that dragged a per-launch start/stop average down to 0.02 of peak)
    does not idle the GPU inside the timed region."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    for _ in range(3):
        fn()
    ev[0].record()
    for i in range(steps):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    return ev[0].elapsed_time(ev[steps]) / steps, ms[len(ms) // 2]


def gemm_rooflines(device, steps=18):
    """The MFMA-bound kernels with the largest shares of the step, each timed live (HIP events around the C-ABI launch on the launch
    stream) at the workload's geometry and CACHE-COLD: the launches rotate over three operand sets (3 x 131 MB of P2 maps > the
    256 MB Infinity Cache), the regime of the step itself and of the counter passes behind `traffic`.  First entry = the step's
    dominant kernel (the 3x3-conv weight gradient, csrc/wgrad_dma.hip), reported as `roofline`.
    Algorithmic flops per launch: SURVEY 8(d) formulas (2*B*H*W*256*256*9; 2*T*N1*N2)."""
    from swin_transformer_object_detection_amd.ops import functional as Fn
    B, H, W, C = PER_GPU_BATCH, IMG_H // 4, IMG_W // 4, 256
    g = torch.Generator().manual_seed(1)
    sets = [(torch.randn(B, H, W, C, generator=g).to(device=device, dtype=_HD()), torch.randn(B, H, W, C, generator=g).to(device=device, dtype=_HD()))
            for _ in range(3)]
    w = (torch.randn(C, 3, 3, C, generator=g) * 0.02).to(device=device, dtype=_HD())
    b = torch.zeros(C, device=device)
    y = torch.empty_like(sets[0][0])
    dw = torch.zeros(C, 3, 3, C, device=device)
    db = torch.zeros(C, device=device)
    # stage 3's Linear weight gradients as the step launches them: recorded, then ONE grouped launch (six blocks x qkv / proj / fc1 / fc2)
    T, C3 = PER_GPU_BATCH * (IMG_H // 16) * (IMG_W // 16), 384
    lin = []
    for _ in range(6):
        for (N1, N2) in ((3 * C3, C3), (C3, C3), (4 * C3, C3), (C3, 4 * C3)):
            lin.append(((torch.randn(T, N1, generator=g) * 0.1).to(device=device, dtype=_HD()), torch.randn(T, N2, generator=g).to(device=device, dtype=_HD()),
                        torch.zeros(N1, N2, device=device), torch.zeros(N1, device=device)))
    lin_flops = sum(2.0 * T * q[2].numel() for q in lin)
    lin_mb = sum(q[0].numel() + q[1].numel() for q in lin) * 2 / 1e6

    def lin_wgrad():
        for dy_, x_, dw_, db_ in lin:
            Fn.call("swin_wgrad_record", Fn._p(dy_), Fn._p(x_), Fn._p(dw_), Fn._p(db_), T, dy_.shape[1], x_.shape[1])
        Fn.call("swin_wgrad_flush", Fn._s())
    Tm, Cm = PER_GPU_BATCH * (IMG_H // 4) * (IMG_W // 4), 96
    mx = torch.randn(Tm, Cm, generator=g).to(device=device, dtype=_HD())
    mdy = torch.randn(Tm, Cm, generator=g).to(device=device, dtype=_HD())
    mw1 = (torch.randn(4 * Cm, Cm, generator=g) * 0.05).to(device=device, dtype=_HD())
    mw2 = (torch.randn(Cm, 4 * Cm, generator=g) * 0.05).to(device=device, dtype=_HD())
    mb1, mb2 = torch.zeros(4 * Cm, device=device), torch.zeros(Cm, device=device)
    my = torch.empty_like(mx)
    mh = torch.empty(Tm, 4 * Cm, device=device, dtype=_HD())
    mdh = torch.empty_like(mh)
    k = [0]

    def rot():
        k[0] += 1
        return sets[k[0] % 3]

    def conv_wgrad():
        dy_, x_ = rot()
        Fn.call("wgrad_conv3x3_nhwc_bf16", Fn._p(dy_), Fn._p(x_), Fn._p(dw), Fn._p(db), B, H, W, C, C, Fn._s())

    def conv_fwd():
        x_, _ = rot()
        Fn.call("conv3x3_nhwc_bf16", Fn._p(x_), Fn._p(w), Fn._p(b), Fn._p(y), B, H, W, C, C, 0, Fn._s())
    conv_flops = 2.0 * B * H * W * C * C * 9
    # (name, kernel-name pattern in the profile, PMC key, flops per launch, launcher, cache regime)
    cases = [
        ("wgrad2_kernel<ConvSrc> (3x3 conv weight gradient, P2 2x200x320x256: LDS-DMA ring, split over t, fp32 atomics)", "wgrad2_kernelINS_7ConvSrc",
         "wgrad2_conv@P2", conv_flops, conv_wgrad, "cold"),
        ("conv_halo_kernel<4> (3x3 conv forward / data gradient, P2 2x200x320x256, halo-staged)", "conv_halo_kernel", "conv_halo@P2", conv_flops,
         conv_fwd, "cold"),
        (f"wgrad96_kernel (the Linear weight gradients of stage 3 in one grouped launch, as in the step: 24 problems, T={T}, C={C3}; "
         "96x96 pieces per wave, groups of eight blocks per XCD)", "wgrad96_kernel", "wgrad96@stage3", lin_flops, lin_wgrad, f"cold ({lin_mb:.0f} MB of operands)"),
        (f"ts_mlp_bwd_kernel (fused MLP data gradient with fc1 recompute, stage 1: T={Tm}, C={Cm})", "ts_mlp_bwd_kernel", None, 24.0 * Tm * Cm * Cm,
         lambda: Fn.call("swin_mlp_bwd_bf16", Fn._p(mx), Fn._p(mdy), Fn._p(mw1), Fn._p(mb1), Fn._p(mw2), Fn._p(my), Fn._p(mh),
                         Fn._p(mdh), Tm, Cm, Fn._s()), "warm"),
        (f"ts_mlp_fwd_kernel (fused fc1+GELU+fc2, stage 1: T={Tm}, C={Cm})", "ts_mlp_fwd_kernel", None, 16.0 * Tm * Cm * Cm,
         lambda: Fn.call("swin_mlp_fwd_bf16", Fn._p(mx), Fn._p(mw1), Fn._p(mb1), Fn._p(mw2), Fn._p(mb2), Fn._p(my), Tm, Cm, Fn._s()), "warm"),
    ]
    shares = step_shares({i: c[1] for i, c in enumerate(cases)})
    out = []
    for i, (name, _pat, pmc_key, flops, fn, cache) in enumerate(cases):
        avg, med = _time_launches(fn, steps)
        tf = flops / (avg * 1e-3) / 1e12
        out.append({"kernel": name, "bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic(pmc_key) if pmc_key else None,
                    "avg_launch_ms": round(avg, 5), "median_launch_ms": round(med, 5), "algorithmic_flops_per_launch": flops, "cache": cache,
                    "step_share": shares.get(i), "step_share_source": PROFILE_NOTE})
    return out


def cpu_baseline():
    """Oracle on the host cores: Swin-T backbone + FPN forward+backward on ONE 800x1280 image, then the C
    RoIAlign (512 rois, 7x7, 4 levels) and NMS (8780 boxes) of one image.  Heads / losses are NOT included, so
    this over-states what a CPU would reach on the full step."""
    import numpy as np
    from oracle import callers_oracle, det_ops_oracle, fpn_oracle, swin_oracle
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, 16)      # a 1-GPU box owns a 16-core share of the host (more threads only oversubscribe)
    torch.set_num_threads(cores)
    p = swin_oracle.make_params(seed=0)
    fp = fpn_oracle.make_params(seed=0)
    for v in list(p.values()) + list(fp.values()):
        v.requires_grad_(True)
    n_img = 4                                   # ~10-15 s of CPU work on a 16-core share
    rng = np.random.RandomState(0)
    t0 = time.perf_counter()
    for k in range(n_img):
        img = torch.randn(1, 3, IMG_H, IMG_W, generator=torch.Generator().manual_seed(k))
        outs = fpn_oracle.fpn_forward(swin_oracle.swin_forward(img, p), fp, 5)
        sum(o.square().mean() for o in outs).backward()
        feats = [o.detach().numpy() for o in outs[:4]]
        xy = rng.rand(512, 2) * [IMG_W * 0.8, IMG_H * 0.8]
        rois = np.concatenate([np.zeros((512, 1)), xy, xy + rng.rand(512, 2) * [IMG_W * 0.3, IMG_H * 0.3] + 8], 1).astype(np.float32)
        callers_oracle.roi_extract(feats, rois, 7)
        bxy = rng.rand(8780, 2).astype(np.float32) * [IMG_W, IMG_H]
        boxes = np.concatenate([bxy, bxy + rng.rand(8780, 2).astype(np.float32) * 200 + 4], 1).astype(np.float32)
        det_ops_oracle.batched_nms(boxes, rng.rand(8780).astype(np.float32), rng.randint(0, 5, 8780),
                                   dict(type="nms", iou_threshold=0.7))
    dt = time.perf_counter() - t0
    return {"value": round(n_img / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n_img} images 3x800x1280, each: oracle Swin-T+FPN fwd+bwd (torch-CPU fp32) + C RoIAlign(512 rois) + "
                      "C batched NMS(8780 boxes); heads/losses/optimizer excluded", "seconds": round(dt, 2)}


# name -> (preset builder, variant, H, W, description).  configs[3] / configs[4] of BASELINE.json besides the headline configs[1].
WORKLOADS = {
    "mask_rcnn_swin_t": ("mask_rcnn_swin", "tiny", 800, 1280,
                         "Mask R-CNN Swin-T patch4 window7, 2x3x800x1280 per GPU, 8 GT boxes+masks/image, AdamW, DropPath 0.1 "
                         "(BASELINE.json configs[1])"),
    "cascade_swin_b": ("cascade_mask_rcnn_swin", "base", 800, 1280,
                       "Cascade Mask R-CNN Swin-B patch4 window7 (3 stages, 4conv1fc + SyncBN, GIoU), 2x3x800x1280 per GPU, AdamW, "
                       "DropPath 0.3 (BASELINE.json configs[3])"),
    "cascade_swin_t": ("cascade_mask_rcnn_swin", "tiny", 800, 1280,
                       "Cascade Mask R-CNN Swin-T patch4 window7 (3 stages, 4conv1fc + SyncBN, GIoU), 2x3x800x1280 per GPU, AdamW"),
    "mask_rcnn_swin_s_1024": ("mask_rcnn_swin", "small", 1024, 1024,
                              "Mask R-CNN Swin-S window7, 2x3x1024x1024 per GPU, AdamW, DropPath 0.2 (BASELINE.json configs[4]; bf16 "
                              "in place of fp16)"),
}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # rehearsal switch (development only): run all ranks on cuda:0 over gloo to exercise the N>1 code path on a
    # one-GPU box.  The driver's multi-GPU runs use one GPU per rank over RCCL ("nccl").
    rehearsal = os.environ.get("BENCH_REHEARSAL_GLOO") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from swin_transformer_object_detection_amd import data, ddp, detector, mixed, presets
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    if dtype != torch.float32:
        from swin_transformer_object_detection_amd import _lib as _swinlib
        _swinlib.set_half_dtype(dtype)
    torch.manual_seed(0)                       # identical weights on every rank
    builder, variant, img_h, img_w, workload_desc = WORKLOADS[args.workload]
    headline = args.workload == "mask_rcnn_swin_t" and args.dtype == "bf16"
    model = detector.build_detector(getattr(presets, builder)(variant), compute_dtype=dtype).to(device)
    model.train()
    shadows = mixed.ShadowParams(model, dtype) if dtype != torch.float32 else None
    # buckets in reverse first-use order (= the order gradients arrive): with plain registration order the backbone's output norms
    # sit in the first backbone bucket and hold it back until the end of backward
    reducer = ddp.BucketedGradReducer(model.parameters_in_forward_order(), leaf_of=shadows.leaf_of if shadows else None,
                                      bucket_bytes=32 << 20)
    reducer.broadcast_parameters()
    if shadows:
        shadows.refresh()
    opt_cfg = presets.OPTIMIZER
    from swin_transformer_object_detection_amd.optim import FusedAdamW
    optim = FusedAdamW(build_param_groups(model, opt_cfg), lr=opt_cfg["lr"], betas=opt_cfg["betas"])      # one launch, refreshes the bf16 shadows
    scaler = mixed.LossScaler(optim, reducer) if dtype == torch.float16 else None      # apex O1's dynamic loss scaling, on the device
    if world == 1 and scaler is None and os.environ.get("SWIN_EARLY_OPT", "1") == "1":
        # one process: a bucket's gradients are final as soon as its parameters have arrived -- the optimizer runs for it right then,
        # on the second stream, instead of for everything after the join at the end of backward (not under dynamic loss scaling: the
        # finite check needs every gradient first).  Round 2 measured no gain (the second stream's backlog was what the join waited
        # for); with 3.3 ms of work left on that stream it takes the optimizer's 0.26 ms out of the tail: 9.78 -> 9.69 ms per step,
        # twice (SWIN_EARLY_OPT=0: off).  A captured one-stream step ignores it.
        reducer.early_step = optim.step_partial
    batch = data.synthetic_batch(PER_GPU_BATCH, img_h, img_w, device, seed=rank)     # per-rank data
    torch.manual_seed(1000 + rank)             # per-rank sampling / DropPath randomness
    # The step runs on a stream of its own, not on the legacy default stream: on this runtime a launch on the default stream that
    # follows a hipStreamWaitEvent on it makes the HOST wait for that event (measured: 24 ms for a 24 ms backlog; 0.16 ms on any
    # other stream) -- every join with the weight-gradient stream would throttle the host to the GPU's pace.
    step_stream = torch.cuda.Stream(device=device)
    step_stream.wait_stream(torch.cuda.current_stream(device))
    torch.cuda.set_stream(step_stream)

    comm = {}

    def step():
        reducer.zero_grad()
        losses = model.forward_train(**batch)
        loss, log_vars = model.parse_losses(losses)
        if scaler is not None:
            loss = scaler.scale(loss)
        reducer.mark_backward_start()
        with torch.autograd.set_multithreading_enabled(False):    # engine on this thread: 9.4 vs 9.4-10.2 ms of host time per step (tools/host_time.py engine)
            loss.backward()
        comm["backward_issued_ms"] = round(reducer._now() * 1e3, 3)     # host time: every backward kernel has been issued
        reducer.finish()
        comm["finish_ms"] = round(reducer._now() * 1e3, 3)
        comm["buckets"] = [dict(bucket=b, bytes=n, issued_ms=round(t0_ * 1e3, 3), done_ms=None if t1_ is None else round(t1_ * 1e3, 3))
                           for b, t0_, t1_, n in reducer.timeline]
        if scaler is not None:
            scaler.check()
        optim.step()                            # AdamW + bf16 shadow refresh, one HIP launch
        if scaler is not None:
            scaler.update()
        return log_vars

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Launch mode.  Eager: ~520 launches per step issued from Python on two HIP streams (weight gradients overlap
    # the data-gradient chain) -- the host needs 8-9 ms per step for that on a fast box and 12-13 ms on a slow one, the GPU ~10.9 ms,
    # so the eager rate follows the HOST on a slow box.  hipGraph: the whole step captured on ONE stream and replayed with a single
    # hipGraphLaunch (0.3 ms of host time; this runtime replays single-queue graphs from pre-built packets, multi-queue graphs cost
    # 7-9 ms of host time per launch) -- no overlap between the streams, ~11.4 ms per step on every box.  auto (one GPU): both are
    # timed for a few steps and the faster one runs the measurement.
    eager_step = step
    launch_mode, graph_note, mode_trials = "eager", None, None
    if os.environ.get("SWIN_BENCH_GC", "1") != "0":
        for _ in range(2):
            eager_step()                        # everything long-lived exists before it is frozen
        mixed.host_gc_for_training()            # fewer interpreter garbage collections inside the eager step (mixed.host_gc_for_training)

    def time_steps(fn, n):
        fn(); barrier()
        t_ = time.perf_counter()
        for _ in range(n):
            fn()
        barrier()
        return (time.perf_counter() - t_) / n * 1e3

    want_graph = args.graph == "on" or (args.graph == "auto" and world == 1 and dtype != torch.float32)
    if want_graph:
        try:
            for _ in range(3):
                eager_step()                    # GEMM plans, scratch buffers, optimizer tables
            t_eager = time_steps(eager_step, 6) if args.graph == "auto" else None
            from swin_transformer_object_detection_amd.graph_step import GraphedTrainStep
            side_was = mixed.side_enabled()
            mixed.set_side_enabled(False)       # a single-queue graph: this runtime's fast replay path
            gstep = GraphedTrainStep(model, reducer, optim, warmup=2, capture_collectives=world > 1, loss_scale=scaler)
            gstep(batch)                        # 2 eager steps on one stream, capture, first replay
            graph_step = lambda: gstep(batch)   # noqa: E731
            t_graph = time_steps(graph_step, 6)
            mode_trials = {"eager_two_streams_ms": None if t_eager is None else round(t_eager, 3), "hipgraph_one_stream_ms": round(t_graph, 3)}
            if t_eager is None or t_graph <= t_eager:
                step, launch_mode = graph_step, "hipgraph"
            else:
                mixed.set_side_enabled(side_was)
        except Exception as e:                  # noqa: BLE001 -- report and measure the eager step instead
            graph_note = f"capture failed, eager step measured: {type(e).__name__}: {e}"[:400]
            torch.cuda.synchronize()

    plans_synced = None
    if world > 1:
        step()                                  # every shape's library GEMM plan exists
        plans_synced = ddp.sync_gemm_plans()    # ... and every rank runs rank 0's algorithm choices from here on
    for _ in range(args.warmup):
        log_vars = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        log_vars = step()
    t_enq = time.perf_counter() - t0          # host time to ISSUE the K steps (nothing waited for yet)
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    logs = {k: float(v) for k, v in ddp.reduce_log_vars(log_vars).items()}
    if not all(map(lambda v: v == v and abs(v) != float("inf"), logs.values())):
        raise SystemExit(f"non-finite loss in the timed region: {logs}")

    sync_check = None
    if world > 1 and not args.no_check_sync:
        # replicas start identical (broadcast) and apply the same averaged gradients: any rank-dependent gradient that
        # slipped past the all-reduce (a missed bucket, a kernel writing after its bucket was launched) shows up here
        with torch.no_grad():
            sig = torch.stack([torch.stack([p.double().sum(), p.double().abs().sum()]) for p in model.parameters()]).flatten()
        gathered = [torch.empty_like(sig) for _ in range(world)]
        dist.all_gather(gathered, sig)
        worst = max(float((g - gathered[0]).abs().max()) for g in gathered)
        if worst != 0.0:
            raise SystemExit(f"replicas diverged: max parameter-checksum difference {worst}")
        sync_check = "replicas bit-identical after %d steps" % (args.warmup + args.steps)

    # `roofline` = the step's dominant kernel (largest share of the profiled step's kernel time: the 3x3-conv weight gradient);
    # `roofline_kernels` = the other priced kernels, the WindowAttention core (the north_star's "WindowAttn MFMA util%") first
    roof_gemm = gemm_rooflines(device) if rank == 0 else None
    roof_attn = attention_roofline(device) if rank == 0 else None
    roof = roof_gemm[0] if roof_gemm else None
    roof_others = ([roof_attn] + roof_gemm[1:]) if roof_gemm else None
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and headline:
        cpu = cpu_baseline()
    if rank == 0:
        gb = PER_GPU_BATCH * world
        out = {
            "metric": ("images/sec/node Mask R-CNN Swin-T 800x1280 bf16 train (fwd+bwd+allreduce+AdamW)" if headline else
                       f"images/sec/node {args.workload} {img_h}x{img_w} {args.dtype} train (fwd+bwd+allreduce+AdamW)"),
            "value": round(gb * args.steps / elapsed, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            # how the step was issued, and the host's share of it: host_enqueue_ms = host time to issue one step (no waiting),
            # ms_per_step = until the GPU has finished it.  host_enqueue_ms ~ ms_per_step means the host, not the GPU, set the rate.
            "launch_mode": launch_mode, "graph_note": graph_note, "mode_trials": mode_trials,
            "host_enqueue_ms": round(1000 * t_enq / args.steps, 3), "gpu_ms": round(1000 * elapsed / args.steps, 3),
            "config": {"workload": workload_desc,
                       "global_batch": gb, "per_gpu_batch": PER_GPU_BATCH, "parallelism": f"dp{world}"},
            "losses": {k: round(v, 4) for k, v in logs.items()},
            "roofline": roof, "roofline_kernels": roof_others, "cpu_baseline": cpu, "sync_check": sync_check,
            # host-side timeline of the last step's gradient exchange, relative to the start of backward (N > 1 only has
            # entries): a bucket issued before backward_issued_ms overlapped the rest of backward
            "comm_timeline": comm if world > 1 else None, "gemm_plans_changed_to_rank0": plans_synced,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
