"""Known-answer + property tests for the RoIAlign / nms / caller oracle (CPU only).

mmcv is absent, so this oracle is 'parity unpinned' (see oracle/__init__.py); what CAN
be pinned is: hand-computed answers, the C restatement against an independently written
dense formulation, and the reference's own known-answer test for delta2bbox."""
import numpy as np
import pytest

from oracle import callers_oracle as C
from oracle import det_ops_oracle as D


# ---- RoIAlign -------------------------------------------------------------
def test_roi_align_hand_computed():
    # 1x1x4x4 ramp, value = 4*y + x.  A linear image is reproduced exactly by bilinear
    # sampling, so each bin equals the ramp at the mean of its sample points.
    inp = np.arange(16, dtype=np.float32).reshape(1, 1, 4, 4)
    rois = np.array([[0, 0.5, 0.5, 2.5, 2.5]], np.float32)  # aligned: start 0, size 2, bins 1x1
    out = D.roi_align_c(inp, rois, 2, 1.0, 0, True)
    # bin (i,j) centre = (0.5+i, 0.5+j) -> 4*(0.5+i) + 0.5+j
    exp = np.array([[2.5, 3.5], [6.5, 7.5]], np.float32)
    np.testing.assert_allclose(out[0, 0], exp, atol=1e-6)
    # spatial_scale and adaptive grid: roi 8x8 at scale 0.5 -> 4x4 on the map, 2x2 bins -> grid 2
    rois = np.array([[0, 0.0, 0.0, 8.0, 8.0]], np.float32)
    out = D.roi_align_c(inp, rois, 2, 0.5, 0, True)
    # start = -0.5; bin 2; samples at -0.5+{0.5,1.5}=0,1 for bin0 and 2,3 for bin1 (clamped at <=0 / >=3)
    exp = np.array([[np.mean([0, 1, 4, 5]), np.mean([2, 3, 6, 7])],
                    [np.mean([8, 9, 12, 13]), np.mean([10, 11, 14, 15])]], np.float32)
    np.testing.assert_allclose(out[0, 0], exp, atol=1e-6)


def test_roi_align_out_of_range_and_empty():
    inp = np.ones((1, 2, 5, 5), np.float32)
    # entirely outside (x > W): every sample returns 0
    out = D.roi_align_c(inp, np.array([[0, 100, 100, 110, 110]], np.float32), 3, 1.0, 0, True)
    assert np.all(out == 0)
    # zero-size roi (aligned): grid = ceil(0/ph) = 0 -> count clamps to 1, output 0
    out = D.roi_align_c(inp, np.array([[0, 2, 2, 2, 2]], np.float32), 3, 1.0, 0, True)
    assert np.all(out == 0)
    out = D.roi_align_c(inp, np.zeros((0, 5), np.float32), 7, 0.25, 0, True)
    assert out.shape == (0, 2, 7, 7)


@pytest.mark.parametrize("aligned,sr", [(True, 0), (True, 2), (False, 0)])
def test_roi_align_c_vs_dense(aligned, sr):
    rng = np.random.RandomState(0)
    inp = rng.randn(2, 3, 13, 17).astype(np.float32)
    K = 40
    xy = rng.rand(K, 2) * np.array([70, 55]) - 5
    wh = rng.rand(K, 2) * np.array([40, 30])
    rois = np.concatenate([rng.randint(0, 2, (K, 1)), xy, xy + wh], 1).astype(np.float32)
    a = D.roi_align_c(inp, rois, (7, 5), 0.25, sr, aligned)
    b = D.roi_align_py(inp, rois, (7, 5), 0.25, sr, aligned)
    np.testing.assert_allclose(a, b, atol=2e-5, rtol=1e-5)


def test_roi_align_bwd_is_adjoint():
    rng = np.random.RandomState(1)
    inp = rng.randn(2, 2, 9, 11).astype(np.float32)
    rois = np.array([[0, 1, 1, 20, 17], [1, -3, 2, 30, 40], [1, 5, 5, 6, 6]], np.float32)
    g = rng.randn(3, 2, 7, 7).astype(np.float32)
    out = D.roi_align_c(inp, rois, 7, 0.25, 0, True)
    gi = D.roi_align_bwd_c(g, rois, inp.shape, 0.25, 0, True)
    # <out, g> == <inp, gi>  (the op is linear in inp)
    np.testing.assert_allclose((out.astype(np.float64) * g).sum(), (inp * gi).sum(), rtol=1e-5)


# ---- nms -------------------------------------------------------------------
def test_nms_hand_computed():
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.5]], np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.95], np.float32)
    dets, keep = D.nms_c(boxes, scores, 0.5)
    # order 3,0,1,2: 3 suppresses 0 (iou .952) and 1 (81/(105+100-81)=.653); 2 survives
    assert keep.tolist() == [3, 2]
    np.testing.assert_allclose(dets, np.concatenate([boxes[[3, 2]], scores[[3, 2], None]], 1))
    # threshold is strict: IoU exactly 0.5 is NOT suppressed
    b = np.array([[0, 0, 2, 1], [0, 0, 1, 1]], np.float32)
    _, keep = D.nms_c(b, np.array([1.0, 0.5], np.float32), 0.5)
    assert keep.tolist() == [0, 1]
    # ties: stable -> lower index first
    b = np.array([[0, 0, 1, 1], [5, 5, 6, 6], [0, 0, 1, 1]], np.float32)
    _, keep = D.nms_c(b, np.array([0.5, 0.5, 0.5], np.float32), 0.3)
    assert keep.tolist() == [0, 1]
    dets, keep = D.nms_c(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 0.5)
    assert dets.shape == (0, 5) and keep.shape == (0,) and keep.dtype == np.int64


@pytest.mark.parametrize("n,thr,offset", [(1, 0.5, 0), (63, 0.5, 0), (64, 0.7, 0), (65, 0.7, 1), (1000, 0.7, 0),
                                           (3000, 0.3, 0)])
def test_nms_c_vs_bruteforce(n, thr, offset):
    rng = np.random.RandomState(n)
    xy = rng.rand(n, 2).astype(np.float32) * 200
    wh = rng.rand(n, 2).astype(np.float32) * 60 + 1
    boxes = np.concatenate([xy, xy + wh], 1)
    scores = np.round(rng.rand(n), 2).astype(np.float32)  # many ties
    d1, k1 = D.nms_c(boxes, scores, thr, offset)
    d2, k2 = D.nms_py(boxes, scores, thr, offset)
    np.testing.assert_array_equal(k1, k2)
    np.testing.assert_array_equal(d1, d2)
    assert np.all(np.diff(d1[:, 4]) <= 0)          # sortedness
    _, k3 = D.nms_c(boxes[k1], scores[k1], thr, offset)  # idempotence
    np.testing.assert_array_equal(k3, np.arange(len(k1)))


def test_batched_nms_separates_classes_and_split_path():
    rng = np.random.RandomState(3)
    n = 500
    xy = rng.rand(n, 2).astype(np.float32) * 100
    boxes = np.concatenate([xy, xy + rng.rand(n, 2).astype(np.float32) * 40 + 1], 1)
    scores = rng.rand(n).astype(np.float32)
    ids = rng.randint(0, 4, n).astype(np.int64)
    cfg = dict(type="nms", iou_threshold=0.5)
    dets, keep = D.batched_nms(boxes, scores, ids, cfg)
    # same result as independent per-class nms merged by score
    ks = []
    for c in range(4):
        m = np.nonzero(ids == c)[0]
        _, k = D.nms_c(boxes[m], scores[m], 0.5)
        ks.append(m[k])
    ks = np.concatenate(ks)
    ks = ks[np.argsort(-scores[ks], kind="stable")]
    np.testing.assert_array_equal(keep, ks)
    np.testing.assert_array_equal(dets[:, :4], boxes[keep])      # un-offset boxes come back
    # split_thr branch gives the same keep set/order
    dets2, keep2 = D.batched_nms(boxes, scores, ids, dict(type="nms", iou_threshold=0.5, split_thr=100))
    np.testing.assert_array_equal(keep, keep2)
    np.testing.assert_array_equal(dets, dets2)
    d0, k0 = D.batched_nms(np.zeros((0, 4)), np.zeros(0), np.zeros(0, np.int64), cfg)
    assert d0.shape == (0, 5) and k0.shape == (0,)


# ---- callers ---------------------------------------------------------------
def test_delta2bbox_reference_known_answer():
    # reference tests/test_utils/test_coder.py:26-45
    rois = np.array([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]], np.float32)
    deltas = np.array([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]], np.float32)
    exp = np.array([[0.0000, 0.0000, 1.0000, 1.0000], [0.1409, 0.1409, 2.8591, 2.8591],
                    [0.0000, 0.3161, 4.1945, 0.6839], [5.0000, 5.0000, 5.0000, 5.0000]], np.float32)
    np.testing.assert_allclose(C.delta2bbox(rois, deltas, max_shape=(32, 32)), exp, atol=1e-4)
    assert C.delta2bbox(np.zeros((0, 4)), np.zeros((0, 4)), max_shape=(32, 32)).shape == (0, 4)


def test_anchors_layout():
    a = C.grid_anchors(2, 3, 4)
    assert a.shape == (18, 4)
    # ratios .5,1,2 scale 8 stride 4: w = 4*8/sqrt(r) ; h = 4*8*sqrt(r)
    np.testing.assert_allclose(a[1], [-16, -16, 16, 16], atol=1e-5)
    np.testing.assert_allclose(a[0], [-22.627417, -11.313708, 22.627417, 11.313708], atol=1e-4)
    np.testing.assert_allclose(a[3 * (1 * 3 + 2) + 1], [8 - 16, 4 - 16, 8 + 16, 4 + 16], atol=1e-5)  # (y=1,x=2)


def test_map_roi_levels_boundaries():
    def roi(s):
        return [0, 0, 0, s, s]
    r = np.array([roi(10), roi(111.9), roi(112), roi(223.9), roi(224), roi(447.9), roi(448), roi(2000)], np.float32)
    assert C.map_roi_levels(r, 4).tolist() == [0, 0, 1, 1, 2, 2, 3, 3]


def test_rpn_get_bboxes_properties():
    rng = np.random.RandomState(5)
    shapes = [(12, 20), (6, 10), (3, 5), (2, 3), (1, 2)]
    cls = [rng.randn(3, h, w).astype(np.float32) for h, w in shapes]
    reg = [rng.randn(12, h, w).astype(np.float32) * 0.3 for h, w in shapes]
    dets, (props, scores, ids) = C.rpn_get_bboxes(cls, reg, (48, 80, 3), nms_pre=100, max_per_img=50)
    assert dets.shape[1] == 5 and 0 < dets.shape[0] <= 50
    assert props.shape[0] == 100 + 100 + 45 + 18 + 6          # per-level top-k then cat
    assert np.all(np.diff(dets[:, 4]) <= 0)
    assert dets[:, 0::2][:, :2].min() >= 0 and dets[:, 2].max() <= 80 and dets[:, 3].max() <= 48
    # per-level entries are in descending score order
    assert np.all(np.diff(scores[:100]) <= 0) and np.all(np.diff(scores[100:200]) <= 0)


def test_multiclass_nms_and_roi_extract():
    rng = np.random.RandomState(6)
    n, nc = 50, 5
    xy = rng.rand(n, nc, 2).astype(np.float32) * 60
    bb = np.concatenate([xy, xy + rng.rand(n, nc, 2).astype(np.float32) * 30 + 1], 2).reshape(n, nc * 4)
    sc = rng.rand(n, nc + 1).astype(np.float32)
    dets, labels = C.multiclass_nms(bb, sc, 0.3, dict(type="nms", iou_threshold=0.5), 20)
    assert dets.shape == (20, 5) and labels.shape == (20,) and labels.max() < nc
    assert dets[:, 4].min() > 0.3 and np.all(np.diff(dets[:, 4]) <= 0)
    d0, l0 = C.multiclass_nms(bb, sc, 2.0, dict(type="nms", iou_threshold=0.5), 20)
    assert d0.shape == (0, 5) and l0.shape == (0,)
    feats = [rng.randn(2, 4, 64 // s, 96 // s).astype(np.float32) for s in (1, 2, 4, 8)]
    rois = C.bbox2roi([np.array([[4, 4, 40, 30], [0, 0, 380, 250]], np.float32), np.array([[10, 10, 200, 150]], np.float32)])
    out = C.roi_extract(feats, rois, 7)
    assert out.shape == (3, 4, 7, 7)
    lv = C.map_roi_levels(rois, 4)
    for k in range(3):
        ref = D.roi_align_c(feats[lv[k]], rois[k:k + 1], 7, 1.0 / (4, 8, 16, 32)[lv[k]], 0, True)
        np.testing.assert_array_equal(out[k], ref[0])


# ---- Cascade R-CNN pieces of the oracle: hand-computable known answers + the host restatements -------------------------
def test_giou_and_smooth_l1_known_answers():
    from oracle import callers_oracle as CO
    same = CO.giou_loss([[0, 0, 2, 2]], [[0, 0, 2, 2]])
    disjoint = CO.giou_loss([[0, 0, 1, 1]], [[2, 0, 3, 1]])       # iou 0, enclosing 3, union 2 -> giou = -1/3
    half = CO.giou_loss([[0, 0, 2, 2]], [[1, 0, 3, 2]])           # overlap 2, union 6, enclosing 6 -> giou = 1/3
    inside = CO.giou_loss([[0, 0, 4, 4]], [[1, 1, 3, 3]])         # overlap 4, union 16, enclosing 16 -> giou = 1/4
    np.testing.assert_allclose([same[0], disjoint[0], half[0], inside[0]], [0.0, 4.0 / 3.0, 2.0 / 3.0, 0.75], atol=1e-6)
    np.testing.assert_allclose(CO.smooth_l1([0.05, -0.05, 0.5, -2.0], 1.0 / 9.0), [0.01125, 0.01125, 0.5 - 1 / 18, 2.0 - 1 / 18], atol=1e-12)
    np.testing.assert_allclose(CO.smooth_l1([0.05, -2.0], 0.0), [0.05, 2.0])


def test_host_bbox_head_loss_and_regress_match_oracle():
    """detector.ConvFCBBoxHead's CPU loss / regress_by_class (the restatement the HIP kernels are also tested against)
    equal the numpy oracle for the three regression modes of the swin configs."""
    import torch
    from oracle import callers_oracle as CO
    from swin_transformer_object_detection_amd import detector
    g = torch.Generator().manual_seed(3)
    n, nc = 120, 80
    xy = torch.rand(n, 2, generator=g) * 300
    rois = torch.cat([torch.zeros(n, 1), xy, xy + torch.rand(n, 2, generator=g) * 100 + 4], 1)
    txy = xy + (torch.rand(n, 2, generator=g) - 0.5) * 40
    gtb = torch.cat([txy, txy + torch.rand(n, 2, generator=g) * 100 + 4], 1)
    cls = torch.randn(n, nc + 1, generator=g)
    labels = torch.randint(0, nc + 1, (n,), generator=g)
    valid = torch.rand(n, generator=g) > 0.2
    pos = (labels < nc) & valid
    labels = torch.where(valid & ~pos, torch.full_like(labels, nc), labels)
    flags = (valid.to(torch.uint8) + 2 * pos.to(torch.uint8)).numpy()
    coder = dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=[0.05, 0.05, 0.1, 0.1])
    cases = [
        (dict(reg_class_agnostic=False, reg_decoded_bbox=True, loss_bbox=dict(type='GIoULoss', loss_weight=10.0)), True),
        (dict(reg_class_agnostic=True, loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0)), False),
        (dict(reg_class_agnostic=False, loss_bbox=dict(type='L1Loss', loss_weight=1.0)), False),
    ]
    for kw, decoded in cases:
        head = detector.ConvFCBBoxHead(num_shared_fcs=1, in_channels=4, fc_out_channels=8, roi_feat_size=1, num_classes=nc,
                                       bbox_coder=coder, **kw)
        ag = kw['reg_class_agnostic']
        bbox = torch.randn(n, 4 if ag else 4 * nc, generator=g)
        tgt = gtb if decoded else torch.randn(n, 4, generator=g)
        out = head.loss(cls, bbox, labels, tgt, pos, valid, rois=rois)
        giou = (rois[:, 1:].numpy(), head.means, head.stds, 1e-6) if decoded else None
        lc, acc, lb = CO.bbox_head_loss(cls.numpy(), bbox.numpy(), labels.numpy(), tgt.numpy(), flags, nc, ag, head.reg_beta, giou)
        assert abs(float(out['loss_cls']) - lc) < 1e-5 and abs(float(out['acc']) - acc) < 1e-3
        assert abs(float(out['loss_bbox']) - lb * head.loss_bbox_weight) < 1e-4 * max(1.0, abs(lb * head.loss_bbox_weight))
        for lab in (labels, None):
            ref = CO.regress_by_class(rois[:, 1:].numpy(), None if lab is None else lab.numpy(), cls.numpy(), bbox.numpy(), nc, ag,
                                      head.means, head.stds, (300, 400))
            got = head.regress_by_class(rois[:, 1:], lab, cls, bbox, (300, 400))
            np.testing.assert_allclose(got.numpy(), ref, rtol=1e-5, atol=1e-3)


def test_batch_norm_oracle_matches_torch():
    import torch
    import torch.nn.functional as F
    from oracle import callers_oracle as CO
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(50, 12, generator=g) * 2 + 1).double().requires_grad_(True)
    gamma, beta = (torch.rand(12, generator=g) + 0.5).double().requires_grad_(True), torch.randn(12, generator=g).double().requires_grad_(True)
    dy = torch.randn(50, 12, generator=g).double()
    y = F.relu(F.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-5))
    y.backward(dy)
    yo, mean, var = CO.batch_norm_train(x.detach().numpy(), gamma.detach().numpy(), beta.detach().numpy(), 1e-5, True)
    dx, dg, db = CO.batch_norm_train_bwd(x.detach().numpy(), gamma.detach().numpy(), beta.detach().numpy(), dy.numpy(), 1e-5, True)
    np.testing.assert_allclose(yo, y.detach().numpy(), atol=1e-10)
    np.testing.assert_allclose(dx, x.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(dg, gamma.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(db, beta.grad.numpy(), atol=1e-10)
