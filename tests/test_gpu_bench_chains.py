"""GPU parity of the exact call chains and kernel geometries the benchmark (BASELINE configs[1]) runs.

* RPN proposals: ``RPNHead.get_bboxes`` (both the fixed-shape path of the training step -- ``det_rpn_topk_decode`` +
  ``batched_nms_static_multi`` -- and the dynamic path) against ``callers_oracle.rpn_get_bboxes`` at the five FPN shapes
  of 2x800x1280, with score ties straddling ``nms_pre``; and against the fixture produced by the reference's own
  ``RPNHead._get_bboxes`` (tests/golden/callers_with_ops.npz).
* Mask targets: ``detector.mask_target`` (the stacked one-launch path of the step and the per-image path) against
  ``callers_oracle.mask_target`` and the reference-code fixture.
* Kernels at the bench's own sizes: conv3x3 forward / data gradient / weight gradient at P2 (2x200x320x256),
  ``wgrad_linear_bf16`` at T = 128 000, window attention backward at the full stage-1 geometry.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import callers_oracle as CO  # noqa: E402
from oracle import swin_oracle as S  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RPN_TRAIN = dict(nms_pre=2000, max_per_img=1000, nms=dict(type='nms', iou_threshold=0.7), min_bbox_size=0)


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import swin_transformer_object_detection_amd as p
    return p


def _rpn_head(compute_dtype=torch.float32):
    from swin_transformer_object_detection_amd.detector import RPNHead
    return RPNHead(256, 256, anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0], strides=[4, 8, 16, 32, 64]),
                   bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0]),
                   loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0), loss_bbox=dict(type='L1Loss', loss_weight=1.0),
                   compute_dtype=compute_dtype)


def _rpn_maps(sizes, B, seed, ties):
    """seeded cls / reg maps.  ties=True: logits on a coarse grid (thousands of exact score ties, also across the nms_pre
    cut).  ties=False: every anchor of an image gets a DIFFERENT logit from one evenly spaced pool over [-3, 3], shuffled
    over levels and positions -- neighbouring scores are then >= 1e-6 apart, so a 1-ulp difference between the device's
    and numpy's exp cannot reorder them (the comparison is about selection and order, not about exp)."""
    g = torch.Generator().manual_seed(seed)
    cls, reg = [], []
    counts = [3 * h * w for h, w in sizes]
    if not ties:
        pool = torch.stack([torch.linspace(-3, 3, sum(counts))[torch.randperm(sum(counts), generator=g)] for _ in range(B)])
    off = 0
    for (h, w), n in zip(sizes, counts):
        if ties:
            c = ((torch.randn(B, 3, h, w, generator=g) * 1.5) * 4).round() / 4
        else:
            c = pool[:, off:off + n].reshape(B, 3, h, w).contiguous()
        off += n
        cls.append(c)
        reg.append(torch.randn(B, 12, h, w, generator=g) * 0.4)
    return cls, reg


def _check_dets(got, valid, want, tag):
    """got (max,5) [+ valid mask] vs the oracle's (k,5): same count, same order; scores / boxes to fp32 rounding."""
    got = got.cpu().numpy()
    if valid is not None:
        v = valid.cpu().numpy()
        k = int(v.sum())
        assert v[:k].all() and not v[k:].any(), f"{tag}: valid slots are not a prefix"
        assert np.all(got[k:] == 0), f"{tag}: unused slots not zero"
        got = got[:k]
    assert got.shape == want.shape, f"{tag}: {got.shape} vs oracle {want.shape}"
    np.testing.assert_allclose(got[:, 4], want[:, 4], rtol=0, atol=2e-7, err_msg=f"{tag}: scores / order")
    np.testing.assert_allclose(got[:, :4], want[:, :4], rtol=1e-5, atol=2e-3, err_msg=f"{tag}: boxes")


@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("static", [True, False])
def test_rpn_get_bboxes_cfg2_shapes_vs_oracle(pkg, static, ties):
    sizes = [(200, 320), (100, 160), (50, 80), (25, 40), (13, 20)]
    B = 2
    cls, reg = _rpn_maps(sizes, B, seed=5 + ties, ties=ties)
    head = _rpn_head()
    out = head.get_bboxes([c.cuda() for c in cls], [r.cuda() for r in reg], [(800, 1280, 3)] * B, RPN_TRAIN, static=static)
    for i in range(B):
        want, (props, scores, ids) = CO.rpn_get_bboxes([c[i].numpy() for c in cls], [r[i].numpy() for r in reg], (800, 1280),
                                                        nms_pre=2000, max_per_img=1000, iou_threshold=0.7)
        if ties:   # the case must really tie at the cut of level 0: the 2000th and 2001st scores are equal
            s0 = np.sort(CO.sigmoid(np.transpose(cls[0][i].numpy(), (1, 2, 0)).reshape(-1)))[::-1]
            assert s0[1999] == s0[2000]
        if static:
            dets, valid = out[i]
            _check_dets(dets, valid, want, f"static img{i}")
        else:
            _check_dets(out[i], None, want, f"dynamic img{i}")


def test_rpn_topk_decode_candidates_equal_oracle(pkg):
    """The kernel's candidate list (before NMS): per level the same SET as the reference's sort-based selection, scores
    bit-equal, and after a stable sort by score the same ORDER as the oracle's list fed to batched_nms."""
    from swin_transformer_object_detection_amd import ops
    sizes = [(200, 320), (100, 160), (50, 80), (25, 40), (13, 20)]
    cls, reg = _rpn_maps(sizes, 2, seed=11, ties=True)
    head = _rpn_head()
    for dtype in (torch.float32, torch.bfloat16):
        c = [x.to(dtype) for x in cls]
        r = [x.to(dtype) for x in reg]
        cls_all = torch.cat([x.permute(0, 2, 3, 1).reshape(2, -1) for x in c], 1).cuda()
        reg_all = torch.cat([x.permute(0, 2, 3, 1).reshape(2, -1, 4) for x in r], 1).cuda()
        anchors = head.anchor_generator.grid_anchors_cat(sizes, torch.device("cuda"))
        ls = [h * w * 3 for h, w in sizes]
        sc, pr, ids = ops.rpn_topk_decode(cls_all, reg_all, anchors, ls, 2000, (0., 0., 0., 0.), (1., 1., 1., 1.), (800, 1280))
        assert sc.shape == (2, 8780)
        for i in range(2):
            _, (props, scores, lids) = CO.rpn_get_bboxes([x[i].float().numpy() for x in c], [x[i].float().numpy() for x in r],
                                                         (800, 1280), nms_pre=2000, max_per_img=1000)
            np.testing.assert_array_equal(ids[i].cpu().numpy(), lids)
            order = torch.sort(sc[i], descending=True, stable=True)[1].cpu().numpy()
            o_order = np.argsort(-scores, kind="stable")
            np.testing.assert_allclose(sc[i].cpu().numpy()[order], scores[o_order], rtol=0, atol=2e-7)
            np.testing.assert_allclose(pr[i].cpu().numpy()[order], props[o_order], rtol=1e-5, atol=2e-3)


def test_rpn_get_bboxes_matches_reference_code_fixture(pkg):
    """Inputs / outputs of the reference's own RPNHead._get_bboxes (nms = the oracle's): both product paths."""
    w = np.load(os.path.join(GOLD, "callers_with_ops.npz"))
    cls = [torch.from_numpy(w[f"rpn_cls_l{l}"]).cuda() for l in range(5)]
    reg = [torch.from_numpy(w[f"rpn_reg_l{l}"]).cuda() for l in range(5)]
    head = _rpn_head()
    for tag in ("train", "small"):
        nms_pre, max_per_img, thr = w[f"rpn_{tag}_cfg"]
        cfg = dict(nms_pre=int(nms_pre), max_per_img=int(max_per_img), nms=dict(type='nms', iou_threshold=float(thr)), min_bbox_size=0)
        for static in (True, False):
            out = head.get_bboxes(cls, reg, [(96, 128, 3)] * 2, cfg, static=static)
            for i in range(2):
                if static:
                    _check_dets(out[i][0], out[i][1], w[f"rpn_{tag}_dets{i}"], f"{tag} static img{i}")
                else:
                    _check_dets(out[i], None, w[f"rpn_{tag}_dets{i}"], f"{tag} dynamic img{i}")


# ------------------------------------------------------------------------------------------------------- mask targets
def _mask_case(seed, H, W, gts, npos):
    rng = np.random.RandomState(seed)
    masks, props, inds = [], [], []
    for g_, k in zip(gts, npos):
        m = np.zeros((g_, H, W), np.uint8)
        for j in range(g_):
            x0, y0 = rng.randint(0, W - 40), rng.randint(0, H - 40)
            m[j, y0:y0 + rng.randint(10, 200), x0:x0 + rng.randint(10, 300)] = 1
        m[0] = (rng.rand(H, W) > 0.5).astype(np.uint8)
        cx, cy = rng.rand(k) * W, rng.rand(k) * H
        bw, bh = rng.rand(k) * W * 0.5 + 4, rng.rand(k) * H * 0.5 + 4
        p = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1).astype(np.float32)
        p[0] = [-20.0, -10.0, W + 30.0, H + 5.0]                       # exercises the clip of mask_target.py:104-107
        masks.append(m); props.append(p); inds.append(rng.randint(0, g_, k).astype(np.int64))
    return masks, props, inds


@pytest.mark.parametrize("stacked", [True, False])
def test_mask_target_chain_vs_oracle(pkg, stacked):
    """detector.mask_target at the step's geometry (800x1280 gt masks, 128 slots per image, 28x28).  stacked=False forces the
    per-image path with a second image whose masks have another size."""
    from swin_transformer_object_detection_amd.detector import mask_target
    masks, props, inds = _mask_case(3, 800, 1280, [8, 5], [128, 128])
    if not stacked:
        masks[1] = masks[1][:, :640, :1000].copy()
    got = mask_target([torch.from_numpy(p).cuda() for p in props], [torch.from_numpy(i).cuda() for i in inds],
                      [torch.from_numpy(m).cuda() for m in masks], 28)
    want = CO.mask_target(props, inds, masks, 28)
    assert got.shape == want.shape == (256, 28, 28)
    diff = (got.cpu().numpy() != want)
    # thresholding a bilinear average at 0.5: a bin whose average is within fp32 rounding of 0.5 may flip; none may otherwise
    assert diff.mean() < 2e-5, f"{diff.sum()} of {diff.size} target pixels differ"


def test_mask_target_matches_reference_code_fixture(pkg):
    from swin_transformer_object_detection_amd.detector import mask_target
    w = np.load(os.path.join(GOLD, "callers_with_ops.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    props = [t(w["mt_props0"]), t(w["mt_props1"])]
    inds = [t(w["mt_inds0"]), t(w["mt_inds1"])]
    masks = [t(w["mt_masks0"]), t(w["mt_masks1"])]
    np.testing.assert_array_equal(mask_target(props, inds, masks, 28).cpu().numpy(), w["mt_out"])
    np.testing.assert_array_equal(mask_target(props, inds, masks, (7, 11)).cpu().numpy(), w["mt_out_7x11"])
    # empty image in the batch (mask_target.py:119-120)
    out = mask_target([props[0], props[1][:0]], [inds[0], inds[1][:0]], masks, 28)
    np.testing.assert_array_equal(out.cpu().numpy(), w["mt_out"][:props[0].size(0)])


def test_roi_targets_pack_vs_oracle(pkg):
    """ops.roi_targets_pack (one launch per image) == the reference's composition on the same sample: bbox2roi
    (transforms.py:117-137), bbox2delta on the positives (delta_xywh_bbox_coder.py:82-130), labels / background
    (bbox_head.py:158-186), pos_is_gt (sampling_result.py), the clipped crop_and_resize rows (mask_target.py:95-107),
    the mask labels and validity; rows of the other image stay untouched."""
    from swin_transformer_object_detection_amd import ops
    rng = np.random.RandomState(11)
    g, n, num, km, nc = 7, 1000, 512, 128, 80
    W, H = 1280.0, 800.0
    xy = rng.rand(g, 2) * [W * 0.6, H * 0.6]
    gts = np.concatenate([xy, xy + rng.rand(g, 2) * [W * 0.35, H * 0.35] + 8], 1).astype(np.float32)
    jit = gts[rng.randint(0, g, 300)] + rng.randn(300, 4).astype(np.float32) * 6        # boxes near the gts -> positives
    xy = rng.rand(n - 300, 2) * [W, H] - 30
    rnd = np.concatenate([xy, xy + rng.rand(n - 300, 2) * 300 + 2], 1).astype(np.float32)
    props = np.concatenate([gts, jit, rnd], 0).astype(np.float32)
    gl = rng.randint(0, nc, g).astype(np.int64)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()          # noqa: E731
    assigned, _, lab = ops.max_iou_assign(t(props), t(gts), 0.5, 0.5, 0.5, False, t(gl), g, None)
    inds, flags = ops.random_sample_raw(assigned, num, 0.25, seed=1234)
    means, stds = (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2)
    for reg_decoded in (False, True):
        buf = ops.RoiStageBuffers(2, num, km, torch.device("cuda"))
        for q in (buf.rois, buf.targets, buf.feat_rois, buf.mask_rois):
            q.fill_(-7.0)
        ops.roi_targets_pack(buf, 1, t(props), inds, flags, assigned, t(gts), means, stds, lab, nc, g, reg_decoded, 5, (H - 100, W - 200))
        torch.cuda.synchronize()
        i_, f_, a_ = inds.cpu().numpy(), flags.cpu().numpy(), assigned.cpu().numpy()
        used, pos = (f_ & 1) > 0, (f_ & 2) > 0
        assert pos.sum() > 20 and (~pos & used).sum() > 100
        boxes = np.where(used[:, None], props[np.where(used, i_, 0)], np.array([[0, 0, 1, 1]], np.float32))
        gi = np.where(used & (a_[np.where(used, i_, 0)] > 0), a_[np.where(used, i_, 0)] - 1, 0)
        want_t = np.zeros((num, 4), np.float32)
        if reg_decoded:
            want_t = gts[gi]
        else:
            want_t[pos] = CO.bbox2delta(boxes[pos], gts[gi[pos]], means, stds)
        want_lab = np.where(pos, gl[gi], nc)
        r = buf.rois.cpu().numpy()
        assert (r[:num] == -7.0).all(), "image 0's rows were written"
        np.testing.assert_array_equal(r[num:, 0], 1.0)
        np.testing.assert_array_equal(r[num:, 1:], boxes)
        np.testing.assert_allclose(buf.targets.cpu().numpy()[num:], want_t, atol=1e-5, rtol=1e-5)
        np.testing.assert_array_equal(buf.labels.cpu().numpy()[num:], want_lab)
        np.testing.assert_array_equal(buf.pos.cpu().numpy()[num:], pos)
        np.testing.assert_array_equal(buf.valid.cpu().numpy()[num:], used)
        np.testing.assert_array_equal(buf.is_gt.cpu().numpy()[num:], pos & (i_ < g))
        fr, mr = buf.feat_rois.cpu().numpy(), buf.mask_rois.cpu().numpy()
        assert (fr[:km] == -7.0).all() and (mr[:km] == -7.0).all()
        np.testing.assert_array_equal(fr[km:, 0], 1.0)
        np.testing.assert_array_equal(fr[km:, 1:], boxes[:km])
        np.testing.assert_array_equal(mr[km:, 0], gi[:km] + 5.0)
        clip = boxes[:km].copy()
        clip[:, 0::2] = np.clip(clip[:, 0::2], 0, W - 200)
        clip[:, 1::2] = np.clip(clip[:, 1::2], 0, H - 100)
        np.testing.assert_array_equal(mr[km:, 1:], clip)
        np.testing.assert_array_equal(buf.mlabels.cpu().numpy()[km:], np.minimum(want_lab[:km], nc - 1))
        np.testing.assert_array_equal(buf.mvalid.cpu().numpy()[km:], pos[:km])


def test_roi_stage_packed_equals_per_op_stage(pkg):
    """_roi_stage_train through the pack kernel gives the losses of the per-op composition (same sampler seeds)."""
    from swin_transformer_object_detection_amd import data, detector, presets
    torch.manual_seed(0)
    model = detector.build_detector(presets.mask_rcnn_swin("tiny"), compute_dtype=torch.bfloat16).cuda().train()
    batch = data.synthetic_batch(2, 384, 512, torch.device("cuda"), seed=5, num_boxes=6)
    with torch.no_grad():
        x = model.extract_feat(batch['img'])
        props = [torch.cat([b_, b_ + 3.0, b_ * 0.9, torch.rand(300, 4, device="cuda") * 200 + torch.tensor([0., 0., 210., 170.], device="cuda")], 0)
                 for b_ in batch['gt_bboxes']]
        rh = model.roi_head
        args = (x, props, batch['gt_bboxes'], batch['gt_labels'], batch['gt_masks'], rh.train_cfg, rh.bbox_roi_extractor, rh.bbox_head,
                rh.mask_roi_extractor, rh.mask_head)
        torch.manual_seed(77)
        l_new, st_new = detector._roi_stage_train(*args)
        try:
            detector._PACKED_STAGE = False                 # the per-op body (what CPU tensors / images without gt boxes run)
            torch.manual_seed(77)
            l_old, st_old = detector._roi_stage_train(*args)
        finally:
            detector._PACKED_STAGE = True
    assert set(l_new) == set(l_old)
    for k in l_new:
        np.testing.assert_allclose(float(l_new[k]), float(l_old[k]), rtol=1e-5, atol=1e-6, err_msg=k)
    for a_, b_ in zip(st_new['rois'], st_old['rois']):
        np.testing.assert_array_equal(a_.cpu().numpy(), b_.cpu().numpy())
    for a_, b_ in zip(st_new['pos_is_gt'], st_old['pos_is_gt']):
        np.testing.assert_array_equal(a_.cpu().numpy(), b_.cpu().numpy())


def test_relu_backward_rides_in_the_producing_kernel(pkg):
    """conv -> ReLU -> conv chains (fcn_mask_head.py:73-104) and rpn_conv -> ReLU -> 1x1 heads (rpn_head.py:41-47): the data
    gradient kernels write the gradient already multiplied by [relu output > 0] (conv3x3_nhwc_bf16_gated /
    narrow_dgrad_gated_bf16) and the conv below skips threshold_backward.  Checked against fp32 autograd of
    F.conv2d / F.relu on the same bf16-rounded operands, and against the un-fused path of this package bit for bit."""
    import torch.nn as nn
    from swin_transformer_object_detection_amd import mixed, ops
    from swin_transformer_object_detection_amd.ops import functional as Fn
    torch.manual_seed(0)
    c1, c2 = nn.Conv2d(64, 128, 3, padding=1).cuda(), nn.Conv2d(128, 64, 3, padding=1).cuda()
    x0 = torch.randn(3, 64, 14, 14, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(3, 64, 14, 14, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)

    def run(fused):
        x = x0.clone().requires_grad_(True)
        for q in (c1, c2):
            q.weight.grad = q.bias.grad = None
        h = ops.conv3x3(x, c1.weight, c1.bias, relu=True)
        y = ops.conv3x3(h, c2.weight, c2.bias, relu=True, x_is_relu=fused)
        y.backward(gy)
        return y.detach(), x.grad.clone(), c1.weight.grad.clone(), c1.bias.grad.clone(), c2.weight.grad.clone()
    calls = []
    orig = torch.ops.aten.threshold_backward

    class _Count:                                   # count the masking passes each variant launches
        def __call__(self, *a):
            calls.append(1)
            return orig(*a)
    try:
        torch.ops.aten.threshold_backward = _Count()
        plain = run(False); n_plain = len(calls); calls.clear()
        fused = run(True); n_fused = len(calls)
    finally:
        torch.ops.aten.threshold_backward = orig
    assert (n_plain, n_fused) == (2, 1)
    assert torch.equal(plain[0], fused[0]) and torch.equal(plain[1], fused[1])          # y and dx: deterministic kernels
    for a, b in zip(plain[2:], fused[2:]):                                              # weight gradients: fp32 atomics
        _close(a, b, _bf16_tol(a, 2), "weight gradient, fused vs plain")                    # returned rounded to bf16: an ulp may flip
    # against fp32 autograd
    xr = x0.float().requires_grad_(True)
    w1, w2 = c1.weight.detach().bfloat16().float().requires_grad_(True), c2.weight.detach().bfloat16().float().requires_grad_(True)
    hr = F.relu(F.conv2d(xr, w1, c1.bias.detach(), padding=1)).bfloat16().float()
    yr = F.relu(F.conv2d(hr, w2, c2.bias.detach(), padding=1))
    yr.backward(gy.float())
    _close(fused[1], xr.grad, _bf16_tol(xr.grad, 6), "dx through both gated ReLUs")
    _close(fused[2], w1.grad, 0.02 * float(w1.grad.abs().max()), "dW of the lower conv")

    # the narrow head: dx = [gate > 0] * dy w
    # (K = 16 / 32 with C % 32 == 0: the matrix-core form, a wave per 32 tokens; else the fp32-FMA form)
    for T, K, C in ((5003, 16, 256), (77, 32, 96), (4096, 16, 64), (333, 24, 256), (1000, 16, 40)):
        dy = torch.randn(T, K, device="cuda").bfloat16()
        w = (torch.randn(K, C, device="cuda") * 0.1).bfloat16()
        gate = torch.relu(torch.randn(T, C, device="cuda")).bfloat16()
        dx = torch.full((T, C), 7.0, device="cuda", dtype=torch.bfloat16)
        Fn.call("narrow_dgrad_gated_bf16", Fn._p(dy), Fn._p(w), Fn._p(gate), Fn._p(dx), T, K, C, Fn._s())
        ref = (dy.float() @ w.float()) * (gate > 0)
        _close(dx, ref, _bf16_tol(ref, 1.01), f"narrow gated dgrad {T}x{K}x{C}")
        assert bool(((dx == 0) | (gate > 0)).all())


# ------------------------------------------------------------------------------------- kernels at the bench's own sizes
def oracle_attention_natural(qkv, qkv_bias, table, B, H, W, nH, shift):
    """Reference semantics on the natural grid: pad (padded tokens are 0 before the qkv Linear, so their q|k|v equal
    qkv.bias -- swin_transformer.py:211-218), roll, partition, core, reverse, roll back, crop (:222-247)."""
    C3 = qkv.shape[-1]
    C = C3 // 3
    Hp, Wp = S.padded_hw(H, W)
    x = qkv.view(B, H, W, C3)
    full = qkv_bias.view(1, 1, 1, C3).expand(B, Hp, Wp, C3)
    full = torch.cat([torch.cat([x, full[:, :H, W:, :]], 2), full[:, H:, :, :]], 1)
    mask = None
    if shift > 0:
        full = torch.roll(full, shifts=(-shift, -shift), dims=(1, 2))
        mask = S.shift_attn_mask(H, W, 7, shift)
    win = S.window_partition(full, 7).view(-1, 49, C3)
    o = S.window_attention_core(win, table, nH, mask)
    o = S.window_reverse(o.view(-1, 7, 7, C), 7, Hp, Wp)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return o[:, :H, :W, :].reshape(B, H * W, C)


def _close(a, b, atol, msg):
    np.testing.assert_allclose(a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy(), atol=atol, rtol=0, err_msg=msg)


def _bf16_tol(ref, ulps):
    return float(ulps * 2.0 ** -8 * max(ref.detach().abs().max().item(), 1e-3))


@pytest.mark.parametrize("H,W", [(200, 320), (256, 256)])
def test_conv3x3_p2_full_size(pkg, H, W):
    """FPN / RPN 3x3 conv at P2 of 2x800x1280 (2x256x200x320: BASELINE configs[1] / [3]) and of 2x1024x1024 (2x256x256x256:
    configs[4]): the many-tile single-buffer schedule (forward and data gradient) and the implicit-im2col weight gradient,
    against fp32 autograd on the CPU on the same bf16-rounded operands."""
    from swin_transformer_object_detection_amd import ops
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    N, C = 2, 256
    g = torch.Generator().manual_seed(17)
    x = torch.randn(N, C, H, W, generator=g).bfloat16().float()
    w = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).bfloat16().float()
    b = torch.randn(C, generator=g) * 0.1
    gy = (torch.randn(N, C, H, W, generator=g) * 0.05).bfloat16().float()
    x0, w0, b0 = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(x0, w0, b0, padding=1)
    (ref * gy).sum().backward()
    x1 = x.cuda().bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w1, b1 = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = ops.conv3x3(x1, w1, b1, False)
    (y.float() * gy.cuda()).sum().backward()
    _close(y, ref, _bf16_tol(ref, 2), "y")
    _close(x1.grad, x0.grad, _bf16_tol(x0.grad, 3), "dx")
    # 128 000 products per weight element accumulated in fp32 (split over blocks, combined with atomics)
    _close(w1.grad, w0.grad, 3e-3 * float(w0.grad.abs().max()), "dw")
    _close(b1.grad, b0.grad, 3e-3 * float(b0.grad.abs().max()), "db")


def test_conv3x3_whole_tile_maps_fwd_and_gated(pkg):
    """Forward with bias + ReLU and the gated data-gradient form (conv3x3_nhwc_bf16_gated) against fp32 convolutions of the same
    bf16-rounded operands, on maps whose pixel count is a multiple of 256 (borders, all nine taps; the mask-head geometry
    256 x 14 x 14 and a 2 x 160 x 128 map)."""
    from swin_transformer_object_detection_amd.ops import functional as Fn
    torch.manual_seed(3)
    for (N, H, W) in [(2, 160, 128), (256, 14, 14)]:
        C = 256
        x = torch.randn(N, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(C, 3, 3, C, device="cuda") * (2.0 / (9 * C)) ** 0.5).bfloat16()
        b = torch.randn(C, device="cuda") * 0.1
        y = Fn._conv3x3_raw(x, w, b, True)
        ref = F.relu(F.conv2d(x.float(), w.float().permute(0, 3, 1, 2), b, padding=1))
        err = float((y.float() - ref).abs().max()); tol = 2 * 2.0 ** -8 * float(ref.abs().max())
        assert err <= tol, ("fwd", N, H, W, err, tol)
        gate = torch.relu(torch.randn(N, C, H, W, device="cuda")).bfloat16().contiguous(memory_format=torch.channels_last)
        z = torch.empty_like(y)
        Fn.call("conv3x3_nhwc_bf16_gated", Fn._p(x), Fn._p(w), None, Fn._p(gate), Fn._p(z), N, H, W, C, C, Fn._s())
        ref2 = F.conv2d(x.float(), w.float().permute(0, 3, 1, 2), None, padding=1) * (gate > 0)
        err = float((z.float() - ref2).abs().max()); tol = 2 * 2.0 ** -8 * float(ref2.abs().max())
        assert err <= tol, ("gated", N, H, W, err, tol)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 32, 32, 256, 256), (2, 25, 40, 256, 256), (2, 13, 20, 256, 256), (1, 7, 9, 64, 128)])
def test_conv3x3_split_k_small_maps(pkg, N, H, W, Cin, Cout):
    """conv3x3_nhwc_bf16_ws on the coarse pyramid levels (P5 of a 1024 x 1024 batch, P5 / P6 of the bench batch and a one-tile map): the contraction split
    over blockIdx.y + the finish launch against (a) an fp32 convolution of the same bf16 operands, with bias + ReLU and in the
    gated data-gradient form, and (b) the unsplit kernel -- equal up to one bf16 rounding of the differently ordered fp32 sum."""
    import torch.nn.functional as F
    from swin_transformer_object_detection_amd import _lib
    from swin_transformer_object_detection_amd.ops import functional as Fn
    nb = int(_lib.lib().conv3x3_splitk_workspace_bytes(N, H, W, Cin, Cout))
    assert nb > 0 and nb % (N * H * W * Cout * 4) == 0 and nb // (N * H * W * Cout * 4) >= 2
    assert int(_lib.lib().conv3x3_splitk_workspace_bytes(2, 200, 320, 256, 256)) == 0       # many tiles: never split
    assert int(_lib.lib().conv3x3_splitk_workspace_bytes(2, 50, 80, 256, 256)) == 0         # 126 tiles: measured, not worth it
    torch.manual_seed(N * H + W)
    x = torch.randn(N, Cin, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, 3, 3, Cin, device="cuda") * (2.0 / (9 * Cin)) ** 0.5).bfloat16()
    b = torch.randn(Cout, device="cuda") * 0.1
    gate = torch.relu(torch.randn(N, Cout, H, W, device="cuda")).bfloat16().contiguous(memory_format=torch.channels_last)
    y = Fn._conv3x3_raw(x, w, b, True)                       # split (the host wrapper asks for the workspace size)
    z = Fn._conv3x3_raw(x, w, None, False, gate=gate)
    y0, z0 = torch.empty_like(y), torch.empty_like(z)        # unsplit kernel
    Fn.call("conv3x3_nhwc_bf16", Fn._p(x), Fn._p(w), Fn._p(b), Fn._p(y0), N, H, W, Cin, Cout, 1, Fn._s())
    Fn.call("conv3x3_nhwc_bf16_gated", Fn._p(x), Fn._p(w), None, Fn._p(gate), Fn._p(z0), N, H, W, Cin, Cout, Fn._s())
    conv = F.conv2d(x.float(), w.float().permute(0, 3, 1, 2), None, padding=1)
    ref = F.relu(conv + b.view(1, -1, 1, 1))
    ref2 = conv * (gate > 0)
    for got, unsplit, want, what in ((y, y0, ref, "fwd"), (z, z0, ref2, "gated")):
        tol = 2 * 2.0 ** -8 * float(want.abs().max())
        assert float((got.float() - want).abs().max()) <= tol, what
        d = (got.float() - unsplit.float()).abs()
        assert bool((d <= 2.0 ** -7 * torch.maximum(got.float().abs(), unsplit.float().abs()) + 1e-6).all()), (what, float(d.max()))
    # too small a workspace is refused, none at all is the plain kernel
    ws = torch.empty(nb - 4, device="cuda", dtype=torch.uint8)
    with pytest.raises(Fn.SwinHipError):
        Fn.call("conv3x3_nhwc_bf16_ws", Fn._p(x), Fn._p(w), Fn._p(b), None, Fn._p(y0), N, H, W, Cin, Cout, 1, Fn._p(ws), nb - 4, Fn._s())
    y1 = torch.empty_like(y)
    Fn.call("conv3x3_nhwc_bf16_ws", Fn._p(x), Fn._p(w), Fn._p(b), None, Fn._p(y1), N, H, W, Cin, Cout, 1, None, 0, Fn._s())
    assert torch.equal(y1, y0)


# Swin-T stage 1 at 2x800x1280 (T = 128 000); Swin-S stage 1 at 2x1024x1024 (T = 131 072); Swin-B stage 1 (C = 128) at 2x800x1280
@pytest.mark.parametrize("T,N1,N2", [(128000, 288, 96), (128000, 384, 96), (128000, 96, 384), (128000, 96, 96), (131072, 288, 96),
                                     (131072, 96, 384), (128000, 384, 128), (128000, 512, 128), (128000, 128, 512)])
def test_linear_wgrad_stage1_full_T(pkg, T, N1, N2):
    """wgrad_linear_bf16 at the stage-1 token counts of the BASELINE configurations (split-T, two k-groups)."""
    from swin_transformer_object_detection_amd._lib import call
    from swin_transformer_object_detection_amd.ops.functional import _p, _s
    g = torch.Generator().manual_seed(N1 + N2)
    dy = (torch.randn(T, N1, generator=g) * 0.1).bfloat16()
    x = torch.randn(T, N2, generator=g).bfloat16()
    ref = dy.float().t() @ x.float()
    refb = dy.float().sum(0)
    dyc, xc = dy.cuda(), x.cuda()
    dw = torch.zeros(N1, N2, device="cuda")
    db = torch.zeros(N1, device="cuda")
    call("wgrad_linear_bf16", _p(dyc), _p(xc), _p(dw), _p(db), T, N1, N2, _s())
    torch.cuda.synchronize()
    _close(dw, ref, 2e-3 * float(ref.abs().max()), "dW")        # fp32 accumulation in another order than the CPU's
    _close(db, refb, 2e-3 * float(refb.abs().max()) + 1e-3, "db")


# (H, W, heads, shift): stage 1 of Swin-T at 2x800x1280 (configs[1]); stage 1 of Swin-S at 2x1024x1024 (configs[4]: 256x256 tokens
# padded to 259x259, 2 738 windows); stages 1 / 2 of Swin-B at 2x800x1280 (configs[3]: C = 128 / 256, 4 / 8 heads -- the four-wave
# block map instead of the three-heads one)
@pytest.mark.parametrize("H,W,nH,shift", [(200, 320, 3, 0), (200, 320, 3, 3), (256, 256, 3, 3), (256, 256, 3, 0), (200, 320, 4, 3),
                                          (100, 160, 8, 3)])
def test_window_attention_bwd_stage1_full(pkg, H, W, nH, shift):
    """win_attn fwd + bwd (fp32 and bf16 kernels) at the first-stage geometries of the BASELINE configurations against autograd
    through the oracle's attention on the natural grid (pad -> roll -> partition -> core -> reverse -> crop)."""
    from swin_transformer_object_detection_amd import ops
    B, C = 2, 32 * nH
    g = torch.Generator().manual_seed(31 + shift + H + nH)
    qkv = (torch.randn(B, H * W, 3 * C, generator=g) * 0.7).bfloat16().float()
    qb = (torch.randn(3 * C, generator=g) * 0.2)
    table = torch.randn(169, nH, generator=g) * 0.5
    go = (torch.randn(B, H * W, C, generator=g) * 0.1).bfloat16().float()
    q0, b0, t0 = qkv.clone().requires_grad_(True), qb.clone().requires_grad_(True), table.clone().requires_grad_(True)
    ref = oracle_attention_natural(q0, b0, t0, B, H, W, nH, shift)
    (ref * go).sum().backward()
    for dtype in (torch.float32, torch.bfloat16):
        q1 = qkv.cuda().to(dtype).requires_grad_(True)
        b1, t1 = qb.cuda().requires_grad_(True), table.cuda().requires_grad_(True)
        out = ops.window_attention(q1, b1, t1, B, H, W, nH, shift)
        (out.float() * go.cuda()).sum().backward()
        if dtype == torch.float32:
            _close(out, ref, 1e-4, "out f32")
            _close(q1.grad, q0.grad, 1e-4, "dqkv f32")
            _close(t1.grad, t0.grad, 2e-3 * float(t0.grad.abs().max()), "dtable f32")
            _close(b1.grad, b0.grad, 2e-3 * float(b0.grad.abs().max()) + 1e-4, "dbias(pad) f32")
        else:
            _close(out, ref, _bf16_tol(ref, 3), "out bf16")
            _close(q1.grad, q0.grad, _bf16_tol(q0.grad, 4), "dqkv bf16")
            _close(t1.grad, t0.grad, 2e-2 * float(t0.grad.abs().max()), "dtable bf16")
            _close(b1.grad, b0.grad, 2e-2 * float(b0.grad.abs().max()) + 1e-3, "dbias(pad) bf16")


def test_fpn_bf16_hip_conv_vs_oracle(pkg):
    """FPN (fpn.py:169-221) in the configuration of the swin configs (in_channels 96..768 -> 256, five outputs) on the bf16
    path: 1x1 laterals as GEMMs, upsample+add kernel, the HAND-WRITTEN MFMA 3x3 conv (not torch's conv, which the fp32 golden test
    of the reference fixture runs) -- against the pinned fp32 oracle on the same bf16-rounded inputs and parameters."""
    from oracle import fpn_oracle
    p = fpn_oracle.make_params((96, 192, 384, 768), 256, seed=3)
    p = {k: v.bfloat16().float() for k, v in p.items()}
    m = pkg.fpn.FPN([96, 192, 384, 768], 256, 5, compute_dtype=torch.bfloat16)
    m.init_weights()
    m.load_state_dict(p, strict=True)
    m.cuda()
    g = torch.Generator().manual_seed(4)
    shapes = [(50, 80), (25, 40), (13, 20), (7, 10)]
    xs = [torch.randn(2, c, h, w, generator=g).bfloat16().float() for c, (h, w) in zip((96, 192, 384, 768), shapes)]
    x0 = [x.clone().requires_grad_(True) for x in xs]
    ref = fpn_oracle.fpn_forward(tuple(x0), p, 5)
    ws = [torch.randn(o.shape, generator=g) for o in ref]
    sum((o * w).sum() for o, w in zip(ref, ws)).backward()
    x1 = [x.cuda().bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True) for x in xs]
    outs = m(tuple(x1))
    assert len(outs) == 5 and all(o.dtype == torch.bfloat16 for o in outs)
    sum((o.float() * w.cuda()).sum() for o, w in zip(outs, ws)).backward()
    for i, (o, r) in enumerate(zip(outs, ref)):
        _close(o, r, _bf16_tol(r, 4), f"out{i}")              # two chained bf16 roundings (lateral sum, conv output)
    for i in range(4):
        _close(x1[i].grad, x0[i].grad, _bf16_tol(x0[i].grad, 6), f"gin{i}")


@pytest.mark.parametrize("hot", [False, True])
def test_roi_align_backward_stage_geometry_gather_vs_fp32_oracle_sum(pkg, hot):
    """The RoIAlign backward of one R-CNN stage at the benchmark's geometry -- 1024 RoIs pooled 7x7 and 256 pooled 14x14 from the
    four levels of a 2x800x1280 pyramid (single_level_roi_extractor.py:93-97) -- through the autograd path the detector uses
    (ops.roi_align_multilevel_group, gather-form backward) against the fp32 scatter kernel on the same inputs; `hot` piles every
    RoI on one of 16 objects, as the sampled RoIs of a training step pile up on the ground-truth boxes."""
    import ctypes
    from swin_transformer_object_detection_amd import ops
    from swin_transformer_object_detection_amd.ops import functional as Fn
    torch.manual_seed(3)
    N, C = 2, 256
    shapes = [(200, 320), (100, 160), (50, 80), (25, 40)]
    strides = [4, 8, 16, 32]
    sets = []
    for K, out in ((1024, 7), (256, 14)):
        rois = torch.rand(K, 5, device="cuda")
        rois[:, 0] = (torch.arange(K, device="cuda") >= K // 2).float()
        wh = torch.exp(torch.rand(K, 2, device="cuda") * 4.0 + 2.5)
        rois[:, 1:3] = rois[:, 1:3] * torch.tensor([1280., 800.], device="cuda") * 0.8
        rois[:, 3:] = torch.minimum(rois[:, 1:3] + wh, torch.tensor([1279., 799.], device="cuda"))
        if hot:
            obj = rois[torch.randint(0, 16, (K,), device="cuda")]
            rois[:, 1:] = obj[:, 1:] + torch.randn(K, 4, device="cuda") * 6.0
            rois[:, 3:] = torch.maximum(rois[:, 3:], rois[:, 1:3] + 8.0)
        scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
        lv = torch.floor(torch.log2(scale / 56 + 1e-6)).clamp(0, 3).int()
        sets.append((rois, lv, out))
    feats = [torch.randn(N, C, h, w, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
             for h, w in shapes]
    outs = ops.roi_align_multilevel_group(feats, sets, strides, 0, True, out_dtype=torch.bfloat16)
    gys = [torch.randn(o.shape, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last) for o in outs]
    torch.autograd.backward(outs, gys)
    # fp32 scatter reference
    n = 4
    Hs = (ctypes.c_int * n)(*[s[0] for s in shapes]); Ws = (ctypes.c_int * n)(*[s[1] for s in shapes])
    sc = (ctypes.c_float * n)(*[1.0 / s for s in strides])
    acc = [torch.zeros(N, h, w, C, device="cuda") for h, w in shapes]
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in acc])
    for (r, lv, out), g in zip(sets, gys):
        gn = g.permute(0, 2, 3, 1).contiguous()
        Fn.call("roi_align_multilevel_bwd", ptrs, Hs, Ws, sc, n, Fn._p(gn), Fn._p(r.float().contiguous()), Fn._p(lv), C, r.shape[0], out, out, 0, 1,
                Fn.SWIN_BF16, Fn._s())
    torch.cuda.synchronize()
    for l, (f, a) in enumerate(zip(feats, acc)):
        ref = a.permute(0, 3, 1, 2)
        if float(ref.abs().max()) == 0:              # no RoI mapped to this level: the gather form must have written zeros
            assert float(f.grad.float().abs().max()) == 0, l
            continue
        _close(f.grad, ref, _bf16_tol(ref, 1.5), f"level {l}")
    assert float(acc[0].abs().max()) > 0
