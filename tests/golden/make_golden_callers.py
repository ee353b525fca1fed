#!/usr/bin/env python
"""Generate ``callers_*.npz`` FROM THE REFERENCE'S OWN caller code (build container only).

The detection glue around the hot-path ops -- anchors, the delta coder, IoU, MaxIoUAssigner, the losses, mask pasting,
RPN proposal selection, multiclass NMS, mask targets, RoI level mapping -- is plain torch / numpy inside
``/root/reference/mmdet``.  The files are loaded BY PATH (``importlib``) with stand-ins for the non-arithmetic
third-party names they import (``mmcv.jit`` decorators, registries, ``ConfigDict``, ``force_fp32`` ...), run on seeded
inputs, and inputs + outputs are stored as data.  Nothing of the reference's text is copied.

Two kinds of fixture, kept apart in the file names:

* ``callers_pure.npz``  -- every arithmetic step ran in the reference's code (coder, IoU, assigner, anchors, losses,
  ``_do_paste_mask``, ``map_roi_levels``, ``bbox2roi``, bbox targets).  These PIN ``oracle/callers_oracle.py``.
* ``callers_with_ops.npz`` -- the reference's caller code (``RPNHead._get_bboxes``, ``multiclass_nms``,
  ``mask_target`` / ``BitmapMasks.crop_and_resize``) ran with ``mmcv.ops.batched_nms`` / ``mmcv.ops.roi_align``
  REPLACED by this repo's CPU oracle of those two ops (mmcv-full is absent: SURVEY 8c).  They pin the caller LOGIC
  (selection order, concatenation, thresholds, truncation, index plumbing); the two ops' arithmetic stays
  "parity unpinned" and the fixture says so in its ``note`` entry.

Also reproduces the reference-held known-answer vectors of tests/test_utils/test_assigner.py:14-152 and
tests/test_utils/test_anchor.py:22-40 (inputs typed in here as data, expected outputs produced by running the
reference code, asserted equal to the values those tests state).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SWIN_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, ROOT)

from oracle import det_ops_oracle as D  # noqa: E402  (stand-in for the two absent mmcv ops only)


class ConfigDict(dict):
    """attribute-style dict (what the callers need of mmcv.ConfigDict)"""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return ConfigDict(v) if isinstance(v, dict) and not isinstance(v, ConfigDict) else v

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        import copy
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


class _Reg:
    def __init__(self):
        self.d = {}

    def register_module(self, *a, **k):
        def deco(cls):
            self.d[cls.__name__] = cls
            return cls
        return deco

    def build(self, cfg):
        cfg = dict(cfg)
        return self.d[cfg.pop("type")](**cfg)


def _mod(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        if "." in name:
            parent, leaf = name.rsplit(".", 1)
            setattr(_mod(parent), leaf, m)
    m.__dict__.update(attrs)
    return m


def _load(rel):
    name = rel[:-3].replace("/", ".")
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    m.__package__ = name.rsplit(".", 1)[0]
    _mod(m.__package__)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    setattr(sys.modules[m.__package__], name.rsplit(".", 1)[1], m)
    return m


def _ident_deco(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


# ---- stand-ins for the two absent mmcv ops: this repo's CPU oracle (parity unpinned, see module docstring) ----------
def _batched_nms_standin(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    dets, keep = D.batched_nms(boxes.detach().numpy(), scores.detach().numpy(), idxs.numpy(), dict(nms_cfg),
                               class_agnostic=class_agnostic)
    return torch.from_numpy(np.ascontiguousarray(dets)), torch.from_numpy(np.ascontiguousarray(keep))


def _roi_align_standin(inp, rois, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True):
    assert pool_mode == 'avg'
    out = D.roi_align_c(inp.detach().numpy(), rois.detach().numpy(), tuple(output_size), float(spatial_scale),
                        int(sampling_ratio), bool(aligned))
    return torch.from_numpy(out)


def install():
    regs = {n: _Reg() for n in ("BBOX_CODERS", "BBOX_ASSIGNERS", "BBOX_SAMPLERS", "IOU_CALCULATORS", "ANCHOR_GENERATORS",
                                "LOSSES", "HEADS", "ROI_EXTRACTORS")}
    _mod("mmcv", jit=_ident_deco, ConfigDict=ConfigDict, Config=ConfigDict,
         is_tuple_of=lambda x, t: isinstance(x, tuple) and all(isinstance(i, t) for i in x))
    _mod("mmcv.cnn", Conv2d=nn.Conv2d, ConvModule=nn.Module, build_upsample_layer=None,
         normal_init=lambda m, std=0.01, **k: nn.init.normal_(m.weight, std=std))
    _mod("mmcv.runner", auto_fp16=_ident_deco, force_fp32=_ident_deco)
    _mod("mmcv.ops", batched_nms=_batched_nms_standin, roi_align=_roi_align_standin)
    _mod("mmcv.ops.nms", batched_nms=_batched_nms_standin)
    _mod("mmcv.ops.roi_align", roi_align=_roi_align_standin)
    _mod("mmcv.ops.carafe", CARAFEPack=type("CARAFEPack", (), {}))
    _mod("cv2"); _mod("pycocotools"); _mod("pycocotools.mask")
    _mod("mmdet"); _mod("mmdet.utils"); _mod("mmdet.core"); _mod("mmdet.models")
    _load("mmdet/utils/util_mixins.py")
    _mod("mmdet.core.bbox.builder", BBOX_CODERS=regs["BBOX_CODERS"], BBOX_ASSIGNERS=regs["BBOX_ASSIGNERS"],
         BBOX_SAMPLERS=regs["BBOX_SAMPLERS"])
    _mod("mmdet.core.bbox.iou_calculators.builder", IOU_CALCULATORS=regs["IOU_CALCULATORS"])
    _mod("mmdet.core.anchor.builder", ANCHOR_GENERATORS=regs["ANCHOR_GENERATORS"])
    _mod("mmdet.models.builder", LOSSES=regs["LOSSES"], HEADS=regs["HEADS"], ROI_EXTRACTORS=regs["ROI_EXTRACTORS"],
         build_loss=lambda cfg: regs["LOSSES"].build(cfg))
    R = types.SimpleNamespace()
    R.iou2d = _load("mmdet/core/bbox/iou_calculators/iou2d_calculator.py")
    _mod("mmdet.core.bbox.iou_calculators", bbox_overlaps=R.iou2d.bbox_overlaps, BboxOverlaps2D=R.iou2d.BboxOverlaps2D,
         build_iou_calculator=lambda cfg: regs["IOU_CALCULATORS"].build(cfg))
    _mod("mmdet.core", bbox_overlaps=R.iou2d.bbox_overlaps)
    _load("mmdet/core/bbox/coder/base_bbox_coder.py")
    R.coder = _load("mmdet/core/bbox/coder/delta_xywh_bbox_coder.py")
    _load("mmdet/core/bbox/assigners/base_assigner.py")
    _load("mmdet/core/bbox/assigners/assign_result.py")
    R.assigner = _load("mmdet/core/bbox/assigners/max_iou_assigner.py")
    R.anchor = _load("mmdet/core/anchor/anchor_generator.py")
    R.transforms = _load("mmdet/core/bbox/transforms.py")
    _load("mmdet/models/losses/utils.py")
    R.iou_loss = _load("mmdet/models/losses/iou_loss.py")
    R.sl1 = _load("mmdet/models/losses/smooth_l1_loss.py")
    R.ce = _load("mmdet/models/losses/cross_entropy_loss.py")
    R.acc = _load("mmdet/models/losses/accuracy.py")
    R.mask_target = _load("mmdet/core/mask/mask_target.py")
    R.structures = _load("mmdet/core/mask/structures.py")
    _mod("mmdet.core", mask_target=R.mask_target.mask_target)
    R.fcn = _load("mmdet/models/roi_heads/mask_heads/fcn_mask_head.py")
    R.bbox_nms = _load("mmdet/core/post_processing/bbox_nms.py")
    _load("mmdet/models/roi_heads/roi_extractors/base_roi_extractor.py")
    R.extractor = _load("mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py")
    # RPNHead: only the _get_bboxes method body is exercised; its bases are placeholders
    _mod("mmdet.models.dense_heads.anchor_head", AnchorHead=type("AnchorHead", (nn.Module,), {}))
    _mod("mmdet.models.dense_heads.rpn_test_mixin", RPNTestMixin=type("RPNTestMixin", (), {}))
    R.rpn = _load("mmdet/models/dense_heads/rpn_head.py")
    return R


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _rand_boxes(g, n, W, H, min_size=1.0):
    cx = torch.rand(n, generator=g) * W
    cy = torch.rand(n, generator=g) * H
    w = torch.rand(n, generator=g) * W * 0.4 + min_size
    h = torch.rand(n, generator=g) * H * 0.4 + min_size
    b = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    b[:, 0::2] = b[:, 0::2].clamp(0, W)
    b[:, 1::2] = b[:, 1::2].clamp(0, H)
    return b


def gen_pure(R):
    d = {}
    g = torch.Generator().manual_seed(101)
    # ---- anchors (anchor_generator.py:161-185, 255-270; config of mask_rcnn_swin_fpn.py:30-34) ----
    ag = R.anchor.AnchorGenerator(strides=[4, 8, 16, 32, 64], ratios=[0.5, 1.0, 2.0], scales=[8])
    sizes = [(10, 14), (5, 7), (3, 4), (2, 2), (1, 1)]
    for l, a in enumerate(ag.grid_anchors(sizes, device='cpu')):
        d[f"anchors_l{l}"] = _np(a)
    d["anchor_sizes"] = np.asarray(sizes)
    for l, b in enumerate(ag.base_anchors):
        d[f"base_anchors_l{l}"] = _np(b)
    vf = ag.valid_flags(sizes, (37, 50, 3), device='cpu')
    for l, v in enumerate(vf):
        d[f"valid_flags_l{l}"] = _np(v)
    # reference-held vector: tests/test_utils/test_anchor.py:22-40
    a1 = R.anchor.AnchorGenerator([10], [1.], [1.], [10]).grid_anchors([(2, 2)], device='cpu')[0]
    assert torch.equal(a1, torch.tensor([[-5., -5., 5., 5.], [5., -5., 15., 5.], [-5., 5., 5., 15.], [5., 5., 15., 15.]]))
    a2 = R.anchor.AnchorGenerator([(10, 20)], [1.], [1.], [10]).grid_anchors([(2, 2)], device='cpu')[0]
    assert torch.equal(a2, torch.tensor([[-5., -5., 5., 5.], [5., -5., 15., 5.], [-5., 15., 5., 25.], [5., 15., 15., 25.]]))
    d["test_anchor_strides_sq"] = _np(a1)
    d["test_anchor_strides_xy"] = _np(a2)

    # ---- delta coder (delta_xywh_bbox_coder.py:87-237) ----
    rois = _rand_boxes(g, 64, 320, 200)
    gts = _rand_boxes(g, 64, 320, 200)
    for tag, means, stds in (("rpn", (0., 0., 0., 0.), (1., 1., 1., 1.)), ("rcnn", (0., 0., 0., 0.), (.1, .1, .2, .2)),
                             ("casc3", (0., 0., 0., 0.), (.033, .033, .067, .067))):
        enc = R.coder.bbox2delta(rois, gts, means, stds)
        d[f"coder_{tag}_bbox2delta"] = _np(enc)
        deltas = torch.randn(64, 4, generator=g) * 1.5
        d[f"coder_{tag}_deltas"] = _np(deltas)
        d[f"coder_{tag}_delta2bbox_clip"] = _np(R.coder.delta2bbox(rois, deltas, means, stds, max_shape=(200, 320, 3)))
        d[f"coder_{tag}_delta2bbox_noclip"] = _np(R.coder.delta2bbox(rois, deltas, means, stds, max_shape=None))
    d["coder_rois"], d["coder_gts"] = _np(rois), _np(gts)
    # multi-class deltas (N, 4*nc) as BBoxHead.get_bboxes feeds them (bbox_head.py:312-314)
    dm = torch.randn(16, 12, generator=g)
    d["coder_multi_deltas"] = _np(dm)
    d["coder_multi_out"] = _np(R.coder.delta2bbox(rois[:16], dm, (0., 0., 0., 0.), (.1, .1, .2, .2), max_shape=(200, 320, 3)))

    # ---- IoU / GIoU (iou2d_calculator.py:71-158) ----
    b1, b2 = _rand_boxes(g, 40, 100, 80), _rand_boxes(g, 23, 100, 80)
    b2[3] = b1[5]                                 # an exact duplicate
    b2[4] = torch.tensor([200., 200., 210., 210.])  # disjoint from everything
    d["iou_b1"], d["iou_b2"] = _np(b1), _np(b2)
    d["iou_matrix"] = _np(R.iou2d.bbox_overlaps(b1, b2, mode='iou'))
    d["giou_matrix"] = _np(R.iou2d.bbox_overlaps(b1, b2, mode='giou'))
    d["iof_matrix"] = _np(R.iou2d.bbox_overlaps(b1, b2, mode='iof'))
    d["iou_aligned"] = _np(R.iou2d.bbox_overlaps(b1[:23], b2, mode='iou', is_aligned=True))
    d["giou_aligned"] = _np(R.iou2d.bbox_overlaps(b1[:23], b2, mode='giou', is_aligned=True))

    # ---- MaxIoUAssigner (max_iou_assigner.py:81-212): RPN and R-CNN settings of mask_rcnn_swin_fpn.py ----
    n_cases = 0
    for tag, kw in (("rpn", dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True, ignore_iof_thr=-1)),
                    ("rcnn", dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False, ignore_iof_thr=-1)),
                    ("casc3", dict(pos_iou_thr=0.7, neg_iou_thr=0.7, min_pos_iou=0.7, match_low_quality=False, ignore_iof_thr=-1))):
        asg = R.assigner.MaxIoUAssigner(iou_calculator=dict(type='BboxOverlaps2D'), **kw)
        for ci, (nb, ng) in enumerate(((300, 7), (50, 1), (64, 12))):
            bb = _rand_boxes(g, nb, 160, 120)
            gt = _rand_boxes(g, ng, 160, 120, min_size=8.0)
            # make ties and exact matches: some boxes equal a gt, two gts share their best box
            bb[:min(ng, 3)] = gt[:min(ng, 3)]
            if ng >= 2:
                bb[10] = (gt[0] + gt[1]) / 2
            gl = torch.randint(0, 80, (ng,), generator=g)
            res = asg.assign(bb, gt, gt_labels=gl)
            k = f"assign_{tag}_{ci}"
            d[k + "_bboxes"], d[k + "_gt"], d[k + "_gt_labels"] = _np(bb), _np(gt), _np(gl)
            d[k + "_gt_inds"], d[k + "_max_overlaps"], d[k + "_labels"] = _np(res.gt_inds), _np(res.max_overlaps), _np(res.labels)
            n_cases += 1
    # reference-held vectors: tests/test_utils/test_assigner.py:14-35, 65-81, 84-105, 142-151
    asg = R.assigner.MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5)
    bb = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [5, 5, 15, 15], [32, 32, 38, 42]])
    gt = torch.FloatTensor([[0, 0, 10, 9], [0, 10, 10, 19]])
    res = asg.assign(bb, gt, gt_labels=torch.LongTensor([2, 3]))
    assert res.gt_inds.tolist() == [1, 0, 2, 0]
    d["test_assigner_basic_gt_inds"] = _np(res.gt_inds)
    d["test_assigner_basic_labels"] = _np(res.labels)
    res = asg.assign(bb, torch.empty(0, 4))
    assert res.gt_inds.tolist() == [0, 0, 0, 0]
    d["test_assigner_empty_gt_gt_inds"] = _np(res.gt_inds)
    res = asg.assign(torch.empty((0, 4)), gt, gt_labels=torch.LongTensor([2, 3]))
    assert len(res.gt_inds) == 0 and tuple(res.labels.shape) == (0,)
    res = asg.assign(torch.empty((0, 4)), torch.empty((0, 4)))
    assert len(res.gt_inds) == 0
    # with ignore (test_assigner.py:38-62): the swin configs use ignore_iof_thr=-1; kept as a vector all the same
    asg_i = R.assigner.MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, ignore_iof_thr=0.5, ignore_wrt_candidates=False)
    bb_i = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [5, 5, 15, 15], [30, 32, 40, 42]])
    res = asg_i.assign(bb_i, gt, gt_bboxes_ignore=torch.Tensor([[30, 30, 40, 40]]))
    assert res.gt_inds.tolist() == [1, 0, 2, -1]

    # ---- losses ----
    pred = torch.randn(50, 4, generator=g, dtype=torch.float64).float()
    tgt = torch.randn(50, 4, generator=g)
    for beta in (1.0, 1.0 / 9.0):
        d[f"smooth_l1_beta{beta:.3f}"] = _np(R.sl1.smooth_l1_loss(pred, tgt, beta=beta, reduction='none'))
    d["l1"] = _np(R.sl1.l1_loss(pred, tgt, reduction='none'))
    d["loss_pred"], d["loss_tgt"] = _np(pred), _np(tgt)
    w = (torch.rand(50, 4, generator=g) > 0.5).float()
    d["loss_weight"] = _np(w)
    d["l1_weighted_avg"] = _np(R.sl1.L1Loss(loss_weight=1.0)(pred, tgt, w, avg_factor=37.0))
    d["smooth_l1_weighted_avg"] = _np(R.sl1.SmoothL1Loss(beta=1.0, loss_weight=1.0)(pred, tgt, w, avg_factor=37.0))
    pb, tb = _rand_boxes(g, 40, 100, 80), _rand_boxes(g, 40, 100, 80)
    tb[:5] = pb[:5]
    pbg = pb.clone().requires_grad_(True)
    gl = R.iou_loss.giou_loss(pbg, tb, eps=1e-6, reduction='none')
    d["giou_pred"], d["giou_tgt"], d["giou_loss"] = _np(pb), _np(tb), _np(gl)
    gw = torch.randn(40, generator=g)
    d["giou_w"] = _np(gw)
    d["giou_grad"] = _np(torch.autograd.grad((gl * gw).sum(), pbg)[0])
    # cross entropy (cross_entropy_loss.py:9-138)
    logits = (torch.randn(64, 81, generator=g) * 3).requires_grad_(True)
    labels = torch.randint(0, 81, (64,), generator=g)
    lw = (torch.rand(64, generator=g) > 0.2).float()
    ce = R.ce.cross_entropy(logits, labels, weight=lw, reduction='mean', avg_factor=51.0)
    d["ce_logits"], d["ce_labels"], d["ce_weight"], d["ce_loss"] = _np(logits), _np(labels), _np(lw), _np(ce)
    d["ce_grad"] = _np(torch.autograd.grad(ce, logits)[0])
    d["ce_accuracy"] = _np(R.acc.accuracy(logits.detach(), labels))
    # RPN: binary cross entropy with use_sigmoid (labels 0 = fg, 1 = bg -> _expand_onehot_labels)
    bl = (torch.randn(200, 1, generator=g) * 2).requires_grad_(True)
    blab = torch.randint(0, 2, (200,), generator=g)
    bw = (torch.rand(200, generator=g) > 0.3).float()
    bce = R.ce.binary_cross_entropy(bl, blab, weight=bw, reduction='mean', avg_factor=128.0)
    d["bce_logits"], d["bce_labels"], d["bce_weight"], d["bce_loss"] = _np(bl), _np(blab), _np(bw), _np(bce)
    d["bce_grad"] = _np(torch.autograd.grad(bce, bl)[0])
    # mask head: mask_cross_entropy (class-specific 28x28 logits)
    ml = (torch.randn(12, 80, 28, 28, generator=g)).requires_grad_(True)
    mt = (torch.rand(12, 28, 28, generator=g) > 0.5).float()
    mlab = torch.randint(0, 80, (12,), generator=g)
    mce = R.ce.mask_cross_entropy(ml, mt, mlab)
    d["mask_labels"], d["mask_targets"], d["mask_loss"] = _np(mlab), _np(mt), _np(mce)
    d["mask_logits_sel"] = _np(ml[torch.arange(12), mlab])                    # only the labelled channels matter
    d["mask_grad_sel"] = _np(torch.autograd.grad(mce, ml)[0][torch.arange(12), mlab])

    # ---- _do_paste_mask (fcn_mask_head.py:303-377), whole-image form used on the GPU (skip_empty=False) ----
    pm = torch.rand(6, 1, 28, 28, generator=g)
    pbx = _rand_boxes(g, 6, 90, 70, min_size=4.0)
    pbx[0] = torch.tensor([10.3, 5.2, 10.3, 40.0])      # zero width -> inf grid -> zeroed (:350-355)
    pbx[1] = torch.tensor([-8.0, -4.0, 30.5, 22.25])    # partly outside
    out, _ = R.fcn._do_paste_mask(pm, pbx, 70, 90, skip_empty=False)
    d["paste_masks"], d["paste_boxes"], d["paste_out"] = _np(pm[:, 0]), _np(pbx), _np(out)
    out2, sl = R.fcn._do_paste_mask(pm[2:], pbx[2:], 70, 90, skip_empty=True)
    d["paste_skip_out"] = _np(out2)
    d["paste_skip_slice"] = np.asarray([int(sl[0].start), int(sl[0].stop), int(sl[1].start), int(sl[1].stop)])

    # ---- RoI plumbing: bbox2roi (transforms.py:69-97), map_roi_levels (single_level_roi_extractor.py:32-51) ----
    bl_ = [_rand_boxes(g, 5, 320, 200), torch.zeros(0, 4), _rand_boxes(g, 3, 320, 200)]
    d["bbox2roi_in0"], d["bbox2roi_in2"] = _np(bl_[0]), _np(bl_[2])
    d["bbox2roi_out"] = _np(R.transforms.bbox2roi(bl_))
    sizes_ = torch.tensor([1., 55., 56., 111.9, 112., 112.1, 223.9, 224., 447.9, 448., 449., 900., 3000.])
    rr = torch.stack([torch.zeros_like(sizes_), torch.zeros_like(sizes_), torch.zeros_like(sizes_), sizes_, sizes_], 1)
    rr2 = torch.cat([torch.zeros(60, 1), _rand_boxes(g, 60, 1280, 800)], 1)
    rois_l = torch.cat([rr, rr2])
    ext = types.SimpleNamespace(finest_scale=56)
    d["map_levels_rois"] = _np(rois_l)
    d["map_levels_out"] = _np(R.extractor.SingleRoIExtractor.map_roi_levels(ext, rois_l, 4))
    d["bbox2result_labels"] = np.asarray([2, 0, 2, 1, 0], np.int64)
    det5 = torch.rand(5, 5, generator=g)
    res = R.transforms.bbox2result(det5, torch.from_numpy(d["bbox2result_labels"]), 3)
    d["bbox2result_dets"] = _np(det5)
    for i, r in enumerate(res):
        d[f"bbox2result_out{i}"] = np.asarray(r)

    d["note"] = np.asarray("every arithmetic step in the reference's own code (torch CPU fp32); generated by "
                           "tests/golden/make_golden_callers.py")
    np.savez_compressed(os.path.join(HERE, "callers_pure.npz"), **d)
    print("callers_pure:", len(d), "arrays;", n_cases, "assigner cases")


def gen_with_ops(R):
    d = {}
    g = torch.Generator().manual_seed(202)
    # ---- RPNHead._get_bboxes (rpn_head.py:82-236), batch of 2, five levels, nms_pre smaller than levels 0-1 ----
    A = 3
    sizes = [(24, 32), (12, 16), (6, 8), (3, 4), (2, 2)]
    strides = [4, 8, 16, 32, 64]
    ag = R.anchor.AnchorGenerator(strides=strides, ratios=[0.5, 1.0, 2.0], scales=[8])
    anchors = ag.grid_anchors(sizes, device='cpu')
    cls = [torch.randn(2, A, h, w, generator=g) * 2 for h, w in sizes]
    reg = [torch.randn(2, A * 4, h, w, generator=g) * 0.5 for h, w in sizes]
    head = types.SimpleNamespace(use_sigmoid_cls=True, test_cfg=None,
                                 bbox_coder=R.coder.DeltaXYWHBBoxCoder(target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0]))
    for tag, cfg in (("train", dict(nms_pre=300, max_per_img=100, nms=dict(type='nms', iou_threshold=0.7), min_bbox_size=0)),
                     ("small", dict(nms_pre=50, max_per_img=1000, nms=dict(type='nms', iou_threshold=0.5), min_bbox_size=0))):
        res = R.rpn.RPNHead._get_bboxes(head, cls, reg, anchors, [(96, 128, 3), (96, 128, 3)], [None, None], ConfigDict(cfg))
        for i, r in enumerate(res):
            d[f"rpn_{tag}_dets{i}"] = _np(r)
        d[f"rpn_{tag}_cfg"] = np.asarray([cfg["nms_pre"], cfg["max_per_img"], cfg["nms"]["iou_threshold"]], np.float64)
    for l in range(5):
        d[f"rpn_cls_l{l}"], d[f"rpn_reg_l{l}"] = _np(cls[l]), _np(reg[l])
    d["rpn_sizes"], d["rpn_strides"], d["rpn_img_shape"] = np.asarray(sizes), np.asarray(strides), np.asarray([96, 128])

    # ---- multiclass_nms (bbox_nms.py:7-93) ----
    n, nc = 60, 5
    mb = torch.cat([_rand_boxes(g, n, 200, 150) for _ in range(nc)], 1)          # (n, nc*4)
    ms = torch.softmax(torch.randn(n, nc + 1, generator=g) * 2, 1)
    for tag, thr, mx in (("thr05", 0.05, 100), ("thr30_max10", 0.3, 10)):
        dets, labels = R.bbox_nms.multiclass_nms(mb, ms, thr, ConfigDict(type='nms', iou_threshold=0.5), mx)
        d[f"mcnms_{tag}_dets"], d[f"mcnms_{tag}_labels"] = _np(dets), _np(labels)
    d["mcnms_bboxes"], d["mcnms_scores"] = _np(mb), _np(ms)
    dets, labels = R.bbox_nms.multiclass_nms(mb[:, :4], ms, 0.05, ConfigDict(type='nms', iou_threshold=0.5), 100)   # shared boxes
    d["mcnms_shared_dets"], d["mcnms_shared_labels"] = _np(dets), _np(labels)

    # ---- mask_target (mask_target.py:6-122) over BitmapMasks.crop_and_resize (structures.py:328-358) ----
    H, W = 60, 84
    rng = np.random.RandomState(0)
    masks_l, props_l, inds_l = [], [], []
    for img in range(2):
        ng = 4 + img
        m = np.zeros((ng, H, W), np.uint8)
        for k in range(ng):                         # rectangles and one ragged blob per image
            x0, y0 = rng.randint(0, W - 20), rng.randint(0, H - 20)
            m[k, y0:y0 + rng.randint(6, 20), x0:x0 + rng.randint(6, 20)] = 1
        m[0] = (rng.rand(H, W) > 0.5).astype(np.uint8)
        masks_l.append(R.structures.BitmapMasks(m, H, W))
        npos = 6 + 3 * img
        pp = _rand_boxes(g, npos, W, H, min_size=3.0)
        pp[0] = torch.tensor([-5.0, -3.0, W + 10.0, H + 4.0])      # clipped to the mask's extent (:104-107)
        props_l.append(pp)
        inds_l.append(torch.randint(0, ng, (npos,), generator=g))
        d[f"mt_masks{img}"], d[f"mt_props{img}"], d[f"mt_inds{img}"] = m, _np(pp), _np(inds_l[-1])
    mt = R.mask_target.mask_target(props_l, inds_l, masks_l, ConfigDict(mask_size=28))
    d["mt_out"] = _np(mt)
    mt2 = R.mask_target.mask_target(props_l, inds_l, masks_l, ConfigDict(mask_size=(7, 11)))
    d["mt_out_7x11"] = _np(mt2)
    d["note"] = np.asarray("reference caller code with mmcv.ops.batched_nms / roi_align replaced by oracle.det_ops_oracle "
                           "(mmcv-full absent): pins caller logic only; generated by tests/golden/make_golden_callers.py")
    np.savez_compressed(os.path.join(HERE, "callers_with_ops.npz"), **d)
    print("callers_with_ops:", len(d), "arrays")


def main():
    torch.set_num_threads(4)
    D.build()
    R = install()
    gen_pure(R)
    gen_with_ops(R)


if __name__ == "__main__":
    main()
