"""Callers of the detection ops on the hot path (CPU, numpy) -- TEST INFRASTRUCTURE ONLY.

Restated from the reference's source text (the modules themselves need mmcv
and cannot be imported here):
  anchors          mmdet/core/anchor/anchor_generator.py:161-185, 255-270
  delta2bbox       mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:133-237
  rpn_get_bboxes   mmdet/models/dense_heads/rpn_head.py:82-236
  map_roi_levels   mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py:32-51
  roi_extract      .../single_level_roi_extractor.py:53-108
  multiclass_nms   mmdet/core/post_processing/bbox_nms.py:7-93
  bbox2roi         mmdet/core/bbox/transforms.py:69-77
  smooth_l1 / giou_loss / bbox_head_loss   mmdet/models/losses/{smooth_l1_loss,iou_loss}.py, bbox_head.py:188-238
  regress_by_class mmdet/models/roi_heads/bbox_heads/bbox_head.py:409-436 (+ cascade_roi_head.py:274-281)
  batch_norm_train torch.nn.SyncBatchNorm arithmetic (norm layer of ConvFCBBoxHead, convfc_bbox_head.py:99-107)
  bbox2delta       mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:87-130
  cross_entropy / binary_cross_entropy / mask_cross_entropy   mmdet/models/losses/cross_entropy_loss.py:9-138
  crop_and_resize / mask_target   mmdet/core/mask/structures.py:328-358, mmdet/core/mask/mask_target.py:6-122

PINNED (round 2) by fixtures generated from the reference's own files, loaded by path in the build container
(tests/golden/make_golden_callers.py -> tests/golden/callers_pure.npz: anchors, coder, IoU/GIoU, MaxIoUAssigner, losses,
_do_paste_mask, map_roi_levels, bbox2roi, bbox2result; callers_with_ops.npz: RPNHead._get_bboxes, multiclass_nms,
mask_target run from the reference's code with the two absent mmcv ops replaced by det_ops_oracle), checked in
tests/test_oracle_callers_golden.py, plus the reference's known-answer tests (tests/test_utils/test_coder.py:26-60,
test_assigner.py:14-152, test_anchor.py:22-40).  What stays PARITY UNPINNED is only the arithmetic of RoIAlign and nms
themselves (mmcv-full absent) -- oracle/__init__.py.
"""
import numpy as np

from . import det_ops_oracle as D


def base_anchors(stride, scales=(8,), ratios=(0.5, 1.0, 2.0)):
    """anchor_generator.py:161-185 (scale_major=True, center_offset=0)."""
    ratios = np.asarray(ratios, dtype=np.float32)
    scales = np.asarray(scales, dtype=np.float32)
    h_r = np.sqrt(ratios)
    w_r = np.float32(1) / h_r
    ws = (np.float32(stride) * w_r[:, None] * scales[None, :]).reshape(-1)
    hs = (np.float32(stride) * h_r[:, None] * scales[None, :]).reshape(-1)
    return np.stack([-0.5 * ws, -0.5 * hs, 0.5 * ws, 0.5 * hs], -1).astype(np.float32)


def grid_anchors(H, W, stride, **kw):
    """anchor_generator.py:255-270 -- index (y*W + x)*A + a."""
    ba = base_anchors(stride, **kw)
    sx = np.arange(W, dtype=np.float32) * stride
    sy = np.arange(H, dtype=np.float32) * stride
    xx, yy = np.meshgrid(sx, sy)  # row-major (y, x)
    shifts = np.stack([xx.ravel(), yy.ravel(), xx.ravel(), yy.ravel()], -1)
    return (shifts[:, None, :] + ba[None, :, :]).reshape(-1, 4).astype(np.float32)


def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None,
               wh_ratio_clip=16 / 1000):
    """delta_xywh_bbox_coder.py:189-237, (N,4) rois and (N,4) deltas, fp32."""
    f = np.float32
    rois = np.asarray(rois, f)
    d = np.asarray(deltas, f) * np.asarray(stds, f) + np.asarray(means, f)
    dx, dy, dw, dh = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
    mr = f(np.abs(np.log(wh_ratio_clip)))
    dw = np.clip(dw, -mr, mr)
    dh = np.clip(dh, -mr, mr)
    px = (rois[:, 0] + rois[:, 2]) * f(0.5)
    py = (rois[:, 1] + rois[:, 3]) * f(0.5)
    pw = rois[:, 2] - rois[:, 0]
    ph = rois[:, 3] - rois[:, 1]
    gw = pw * np.exp(dw)
    gh = ph * np.exp(dh)
    gx = px + pw * dx
    gy = py + ph * dy
    b = np.stack([gx - gw * f(.5), gy - gh * f(.5), gx + gw * f(.5), gy + gh * f(.5)], -1).astype(f)
    if max_shape is not None:
        Hm, Wm = f(max_shape[0]), f(max_shape[1])
        b[:, 0::2] = np.clip(b[:, 0::2], 0, Wm)   # clamp to W / H, not W-1 (:222-235)
        b[:, 1::2] = np.clip(b[:, 1::2], 0, Hm)
    return b


def sigmoid(x):
    x = np.asarray(x, np.float32)
    return (np.float32(1) / (np.float32(1) + np.exp(-x))).astype(np.float32)


def rpn_get_bboxes(cls_scores, bbox_preds, img_shape, strides=(4, 8, 16, 32, 64), nms_pre=2000,
                   max_per_img=1000, iou_threshold=0.7, nms_fn=None):
    """rpn_head.py:82-236 for ONE image.

    cls_scores[l]: (A, H, W) logits; bbox_preds[l]: (A*4, H, W).  Returns dets (<=max,5).
    Also returns the (boxes, scores, level_ids) fed to batched_nms, for op-level tests."""
    sc_l, bp_l, an_l, id_l = [], [], [], []
    for l, (cs, bp) in enumerate(zip(cls_scores, bbox_preds)):
        A, H, W = cs.shape
        s = sigmoid(np.transpose(cs, (1, 2, 0)).reshape(-1))                 # :130-133
        d = np.transpose(bp, (1, 2, 0)).reshape(-1, 4)                        # :141-142
        an = grid_anchors(H, W, strides[l])
        if s.shape[0] > nms_pre:                                              # :162-169
            order = np.argsort(-s, kind="stable")[:nms_pre]
            s, d, an = s[order], d[order], an[order]
        sc_l.append(s); bp_l.append(d); an_l.append(an)
        id_l.append(np.full(s.shape[0], l, dtype=np.int64))                  # :174-180
    scores = np.concatenate(sc_l); deltas = np.concatenate(bp_l)
    anchors = np.concatenate(an_l); ids = np.concatenate(id_l)
    props = delta2bbox(anchors, deltas, max_shape=img_shape)                  # :185-186
    dets, keep = D.batched_nms(props, scores, ids, dict(type="nms", iou_threshold=iou_threshold),
                               nms_fn=nms_fn)                                 # :233
    return dets[:max_per_img], (props, scores, ids)                           # :235


def bbox2roi(bbox_list):
    """transforms.py:69-77."""
    out = []
    for i, b in enumerate(bbox_list):
        b = np.asarray(b, np.float32)[:, :4]
        out.append(np.concatenate([np.full((b.shape[0], 1), i, np.float32), b], 1))
    return np.concatenate(out, 0)


def map_roi_levels(rois, num_levels, finest_scale=56):
    """single_level_roi_extractor.py:47-51."""
    f = np.float32
    rois = np.asarray(rois, f)
    scale = np.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lv = np.floor(np.log2(scale / f(finest_scale) + f(1e-6)))
    return np.clip(lv, 0, num_levels - 1).astype(np.int64)


def roi_extract(feats, rois, output_size, strides=(4, 8, 16, 32), sampling_ratio=0, finest_scale=56):
    """single_level_roi_extractor.py:53-108: per-level RoIAlign gathered back in roi order."""
    rois = np.asarray(rois, np.float32).reshape(-1, 5)
    C = feats[0].shape[1]
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    out = np.zeros((rois.shape[0], C, ph, pw), np.float32)
    if rois.shape[0] == 0:
        return out
    lv = map_roi_levels(rois, len(feats), finest_scale)
    for i, f in enumerate(feats):
        inds = np.nonzero(lv == i)[0]
        if inds.size:
            out[inds] = D.roi_align_c(f, rois[inds], (ph, pw), 1.0 / strides[i], sampling_ratio, True)
    return out


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, nms_fn=None):
    """bbox_nms.py:7-93 -> (dets (k,5), labels (k,))."""
    multi_bboxes = np.asarray(multi_bboxes, np.float32)
    multi_scores = np.asarray(multi_scores, np.float32)
    n, nc = multi_scores.shape[0], multi_scores.shape[1] - 1
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.reshape(n, -1, 4)
    else:
        bboxes = np.broadcast_to(multi_bboxes[:, None], (n, nc, 4))
    scores = multi_scores[:, :-1]
    labels = np.broadcast_to(np.arange(nc, dtype=np.int64)[None], scores.shape)
    bboxes, scores, labels = bboxes.reshape(-1, 4), scores.reshape(-1), labels.reshape(-1)
    inds = np.nonzero(scores > np.float32(score_thr))[0]                      # :54,66-67
    bboxes, scores, labels = bboxes[inds], scores[inds], labels[inds]
    if bboxes.size == 0:
        return np.zeros((0, 5), np.float32), labels
    dets, keep = D.batched_nms(bboxes, scores, labels, nms_cfg, nms_fn=nms_fn)  # :84
    if max_num > 0:
        dets, keep = dets[:max_num], keep[:max_num]
    return dets, labels[keep]


# ---- training targets (checker for csrc/det_targets.hip) ---------------------------------------------------------
def bbox_overlaps(b1, b2, eps=1e-6, mode='iou', is_aligned=False):
    """IoU / IoF / GIoU, matrix (len(b1), len(b2)) or aligned pairs, float32, operation order of
    mmdet/core/bbox/iou_calculators/iou2d_calculator.py:71-158 (bbox_overlaps)."""
    f = np.float32
    b1 = np.asarray(b1, f).reshape(-1, 4)
    b2 = np.asarray(b2, f).reshape(-1, 4)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    if is_aligned:
        B1, B2, A1, A2 = b1, b2, a1, a2
    else:
        B1, B2, A1, A2 = b1[:, None, :], b2[None, :, :], a1[:, None], a2[None, :]
    lt = np.maximum(B1[..., :2], B2[..., :2])
    rb = np.minimum(B1[..., 2:], B2[..., 2:])
    wh = np.maximum(rb - lt, f(0))
    inter = wh[..., 0] * wh[..., 1]
    union = (A1 + A2 - inter) if mode in ('iou', 'giou') else (A1 + f(0) * inter)
    union = np.maximum(union, f(eps))
    ious = (inter / union).astype(f)
    if mode in ('iou', 'iof'):
        return ious
    elt = np.minimum(B1[..., :2], B2[..., :2])
    erb = np.maximum(B1[..., 2:], B2[..., 2:])
    ewh = np.maximum(erb - elt, f(0))
    earea = np.maximum(ewh[..., 0] * ewh[..., 1], f(eps))
    return (ious - (earea - union) / earea).astype(f)


def max_iou_assign(bboxes, gt_bboxes, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0, match_low_quality=True, gt_labels=None):
    """MaxIoUAssigner.assign_wrt_overlaps (mmdet/core/bbox/assigners/max_iou_assigner.py:128-212) with
    ignore_iof_thr=-1, gt_max_assign_all=True.  -> (assigned_gt_inds int64, max_overlaps f32, labels int64 | None)."""
    bboxes = np.asarray(bboxes, np.float32).reshape(-1, 4)
    gt_bboxes = np.asarray(gt_bboxes, np.float32).reshape(-1, 4)
    n, g = len(bboxes), len(gt_bboxes)
    assigned = np.full(n, -1, np.int64)
    if g == 0 or n == 0:
        if g == 0:
            assigned[:] = 0
        labels = None if gt_labels is None else np.full(n, -1, np.int64)
        return assigned, np.zeros(n, np.float32), labels
    ov = bbox_overlaps(gt_bboxes, bboxes)                 # (g, n)
    max_ov = ov.max(0)
    argmax = ov.argmax(0)
    gt_max = ov.max(1)
    assigned[(max_ov >= 0) & (max_ov < neg_iou_thr)] = 0
    pos = max_ov >= pos_iou_thr
    assigned[pos] = argmax[pos] + 1
    if match_low_quality:
        for i in range(g):
            if gt_max[i] >= min_pos_iou:
                assigned[ov[i] == gt_max[i]] = i + 1
    labels = None
    if gt_labels is not None:
        labels = np.full(n, -1, np.int64)
        p = assigned > 0
        labels[p] = np.asarray(gt_labels, np.int64)[assigned[p] - 1]
    return assigned, max_ov, labels


# ---- test-time path (checker for detector.simple_test and csrc/mask_paste.hip) -----------------------------------
def paste_masks(mask_logits, labels, boxes, img_h, img_w, thr=0.5, is_prob=False):
    """FCNMaskHead.get_seg_masks + _do_paste_mask (mmdet/models/roi_heads/mask_heads/fcn_mask_head.py:218-300, :303-377;
    GPU branch: whole image, skip_empty=False): sigmoid, class channel, F.grid_sample(bilinear, zeros padding,
    align_corners=False) on the normalised grid of the box, `>= thr`.  -> (bool (N,img_h,img_w), float32 values).
    is_prob: the input holds probabilities already (ndarray input of get_seg_masks, :209-213)."""
    m = np.asarray(mask_logits, np.float32)
    N, nc, mh, mw = m.shape
    boxes = np.asarray(boxes, np.float32).reshape(-1, 4)
    vals = np.zeros((N, img_h, img_w), np.float32)
    ys = np.arange(img_h, dtype=np.float32) + np.float32(0.5)
    xs = np.arange(img_w, dtype=np.float32) + np.float32(0.5)
    for n in range(N):
        p = m[n, int(labels[n])] if is_prob else (np.float32(1) / (np.float32(1) + np.exp(-m[n, int(labels[n])]))).astype(np.float32)
        x0, y0, x1, y1 = boxes[n]
        with np.errstate(divide='ignore', invalid='ignore'):
            gy = (ys - y0) / (y1 - y0) * np.float32(2) - np.float32(1)
            gx = (xs - x0) / (x1 - x0) * np.float32(2) - np.float32(1)
        gy[np.isinf(gy)] = 0
        gx[np.isinf(gx)] = 0
        iy = ((gy + np.float32(1)) * np.float32(mh) - np.float32(1)) / np.float32(2)
        ix = ((gx + np.float32(1)) * np.float32(mw) - np.float32(1)) / np.float32(2)
        fy, fx = np.floor(iy), np.floor(ix)
        wy1, wx1 = (iy - fy).astype(np.float32), (ix - fx).astype(np.float32)
        wy0, wx0 = np.float32(1) - wy1, np.float32(1) - wx1
        y0i, x0i = fy.astype(np.int64), fx.astype(np.int64)

        def tap(yy, xx):
            ok = ((yy >= 0) & (yy < mh))[:, None] & ((xx >= 0) & (xx < mw))[None, :]
            v = p[np.clip(yy, 0, mh - 1)[:, None], np.clip(xx, 0, mw - 1)[None, :]]
            return np.where(ok, v, np.float32(0))
        v = tap(y0i, x0i) * (wy0[:, None] * wx0[None, :])
        v = v + tap(y0i, x0i + 1) * (wy0[:, None] * wx1[None, :])
        v = v + tap(y0i + 1, x0i) * (wy1[:, None] * wx0[None, :])
        v = v + tap(y0i + 1, x0i + 1) * (wy1[:, None] * wx1[None, :])
        vals[n] = np.nan_to_num(v.astype(np.float32), nan=0.0)
    return vals >= np.float32(thr), vals


def bbox2result(bboxes, labels, num_classes):
    """mmdet/core/bbox/transforms.py:99-117."""
    bboxes, labels = np.asarray(bboxes, np.float32).reshape(-1, 5), np.asarray(labels, np.int64)
    return [bboxes[labels == i] for i in range(num_classes)]


def bbox_head_get_bboxes(rois, cls_score, bbox_pred, img_shape, scale_factor, rescale, score_thr, nms_cfg, max_per_img,
                         means=(0., 0., 0., 0.), stds=(.1, .1, .2, .2)):
    """BBoxHead.get_bboxes (mmdet/models/roi_heads/bbox_heads/bbox_head.py:270-373) for one image, class-specific
    regression: softmax, per-class decode clipped to img_shape, optional division by scale_factor, multiclass_nms."""
    if isinstance(cls_score, (list, tuple)):                       # bbox_head.py:300-301: per-stage scores are averaged
        cls_score = sum(np.asarray(c, np.float32) for c in cls_score) / np.float32(len(cls_score))
    cls_score = np.asarray(cls_score, np.float32)
    e = np.exp(cls_score - cls_score.max(1, keepdims=True))
    scores = (e / e.sum(1, keepdims=True)).astype(np.float32)
    rois = np.asarray(rois, np.float32)
    n, nc = scores.shape[0], scores.shape[1] - 1
    d = np.asarray(bbox_pred, np.float32).reshape(n * nc, 4)
    r = np.repeat(rois[:, 1:5], nc, axis=0)
    boxes = delta2bbox(r, d, means, stds, img_shape).reshape(n, nc * 4)
    if rescale:
        boxes = (boxes.reshape(n, nc, 4) / np.asarray(scale_factor, np.float32)).reshape(n, nc * 4)
    return multiclass_nms(boxes, scores, score_thr, nms_cfg, max_per_img)


# ---- Cascade R-CNN pieces (checkers for csrc/det_losses.hip regression modes, det_regress_by_class, csrc/batchnorm.hip) --
def smooth_l1(diff, beta):
    """losses/smooth_l1_loss.py:10-28 element (beta <= 0: plain L1, :31-45)."""
    d = np.abs(np.asarray(diff, np.float64))
    if beta <= 0:
        return d
    return np.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)


def giou_loss(pred, target, eps=1e-6):
    """1 - GIoU per aligned box pair: losses/iou_loss.py:78-101 over bbox_overlaps(mode='giou', is_aligned=True)
    (core/bbox/iou_calculators/iou2d_calculator.py:108-158).  float64."""
    p, t = np.asarray(pred, np.float64).reshape(-1, 4), np.asarray(target, np.float64).reshape(-1, 4)
    area1 = (p[:, 2] - p[:, 0]) * (p[:, 3] - p[:, 1])
    area2 = (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])
    wh = np.clip(np.minimum(p[:, 2:], t[:, 2:]) - np.maximum(p[:, :2], t[:, :2]), 0, None)
    overlap = wh[:, 0] * wh[:, 1]
    union = np.maximum(area1 + area2 - overlap, eps)
    ious = overlap / union
    ewh = np.clip(np.maximum(p[:, 2:], t[:, 2:]) - np.minimum(p[:, :2], t[:, :2]), 0, None)
    earea = np.maximum(ewh[:, 0] * ewh[:, 1], eps)
    return 1.0 - (ious - (earea - union) / earea)


def bbox_head_loss(cls_score, bbox_pred, labels, targets, flags, num_classes, class_agnostic=False, beta=0.0, giou=None):
    """BBoxHead.loss (bbox_head.py:188-238) over a fixed-size sample (flags bit 0 used, bit 1 positive):
    -> (loss_cls, acc %, loss_bbox), float64.  giou = (rois (n,4), means, stds, eps) selects reg_decoded_bbox + GIoULoss."""
    c = np.asarray(cls_score, np.float64)
    n = c.shape[0]
    used = (np.asarray(flags) & 1) > 0
    pos = ((np.asarray(flags) & 2) > 0) & used & (np.asarray(labels) < num_classes)
    nv = max(int(used.sum()), 1)
    m = c.max(1, keepdims=True)
    lse = (m + np.log(np.exp(c - m).sum(1, keepdims=True)))[:, 0]
    ce = lse - c[np.arange(n), labels]
    loss_cls = float((ce * used).sum() / nv)
    acc = float(((c.argmax(1) == labels) & used).sum() / nv * 100.0)
    b = np.asarray(bbox_pred, np.float64)
    lab = np.minimum(np.asarray(labels), num_classes - 1)
    pred = b.reshape(n, 4) if class_agnostic else b.reshape(n, num_classes, 4)[np.arange(n), lab]
    t = np.asarray(targets, np.float64)
    if giou is not None:
        rois, means, stds, eps = giou
        dec = delta2bbox(np.asarray(rois, np.float64), pred, means, stds, None)
        per = giou_loss(dec, t, eps)
    else:
        per = smooth_l1(pred - t, beta).sum(1)
    return loss_cls, acc, float((per * pos).sum() / nv)


def regress_by_class(rois, labels, cls_score, bbox_pred, num_classes, class_agnostic, means, stds, img_shape):
    """BBoxHead.regress_by_class (bbox_head.py:409-436) with CascadeRoIHead's label choice (cascade_roi_head.py:274-281
    in training: background -> argmax of the foreground scores; :316-317 in testing: labels=None -> argmax)."""
    c = np.asarray(cls_score, np.float32)
    n = c.shape[0]
    am = c[:, :num_classes].argmax(1)
    lab = am if labels is None else np.where((np.asarray(labels) >= num_classes) | (np.asarray(labels) < 0), am, labels)
    d = np.asarray(bbox_pred, np.float32)
    if not class_agnostic:
        d = d.reshape(n, num_classes, 4)[np.arange(n), lab]
    return delta2bbox(np.asarray(rois, np.float32), d.reshape(n, 4), means, stds, img_shape)


def batch_norm_train(x, gamma, beta, eps=1e-5, relu=False):
    """Training-mode BatchNorm over (R, C) rows (torch.nn.BatchNorm2d / SyncBatchNorm on NHWC data: biased variance for
    normalisation), float64 -> (y, mean, var_biased)."""
    x = np.asarray(x, np.float64)
    mean, var = x.mean(0), x.var(0)
    y = (x - mean) / np.sqrt(var + eps) * np.asarray(gamma, np.float64) + np.asarray(beta, np.float64)
    return (np.maximum(y, 0) if relu else y), mean, var


def batch_norm_train_bwd(x, gamma, beta, dy, eps=1e-5, relu=False):
    """Gradients of batch_norm_train (SyncBatchNorm's backward_reduce / backward_elemt arithmetic), float64:
    -> (dx, dgamma, dbeta)."""
    x, dy = np.asarray(x, np.float64), np.asarray(dy, np.float64)
    g = np.asarray(gamma, np.float64)
    mean, var = x.mean(0), x.var(0)
    invstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * invstd
    if relu:
        dy = dy * ((xhat * g + np.asarray(beta, np.float64)) > 0)
    sdy, sdyx = dy.sum(0), (dy * xhat).sum(0)
    n = x.shape[0]
    dx = g * invstd * (dy - sdy / n - xhat * sdyx / n)
    return dx, sdyx, sdy


# ---- round 2: the remaining callers, each pinned by tests/golden/callers_*.npz ---------------------------------------
def bbox2delta(proposals, gt, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)):
    """delta_xywh_bbox_coder.py:87-130 (fp32)."""
    f = np.float32
    p, g = np.asarray(proposals, f), np.asarray(gt, f)
    px, py = (p[:, 0] + p[:, 2]) * f(0.5), (p[:, 1] + p[:, 3]) * f(0.5)
    pw, ph = p[:, 2] - p[:, 0], p[:, 3] - p[:, 1]
    gx, gy = (g[:, 0] + g[:, 2]) * f(0.5), (g[:, 1] + g[:, 3]) * f(0.5)
    gw, gh = g[:, 2] - g[:, 0], g[:, 3] - g[:, 1]
    d = np.stack([(gx - px) / pw, (gy - py) / ph, np.log(gw / pw), np.log(gh / ph)], -1).astype(f)
    return ((d - np.asarray(means, f)) / np.asarray(stds, f)).astype(f)


def cross_entropy(logits, labels, weight=None, avg_factor=None):
    """cross_entropy_loss.py:9-43 (+ weight_reduce_loss, losses/utils.py:27-58, reduction='mean'): float64 scalar and
    d loss / d logits."""
    c = np.asarray(logits, np.float64)
    n = c.shape[0]
    m = c.max(1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(c - m).sum(1))
    per = lse - c[np.arange(n), labels]
    w = np.ones(n) if weight is None else np.asarray(weight, np.float64)
    den = float(n) if avg_factor is None else float(avg_factor)
    p = np.exp(c - lse[:, None])
    p[np.arange(n), labels] -= 1.0
    return float((per * w).sum() / den), p * (w / den)[:, None]


def binary_cross_entropy(logits, labels, weight=None, avg_factor=None):
    """cross_entropy_loss.py:46-95 for (N,1) logits of a one-class sigmoid head (the RPN): label 0 = foreground ->
    target 1, label 1 (= num_classes) = background -> target 0 (_expand_onehot_labels :46-58).  float64 scalar, grad."""
    z = np.asarray(logits, np.float64).reshape(-1)
    t = (np.asarray(labels) == 0).astype(np.float64)
    per = np.maximum(z, 0) - z * t + np.log1p(np.exp(-np.abs(z)))
    w = np.ones_like(z) if weight is None else np.asarray(weight, np.float64)
    den = float(z.size) if avg_factor is None else float(avg_factor)
    sig = 1.0 / (1.0 + np.exp(-z))
    return float((per * w).sum() / den), ((sig - t) * w / den).reshape(np.asarray(logits).shape)


def mask_cross_entropy(sel_logits, targets):
    """cross_entropy_loss.py:98-138: mean BCE-with-logits over the labelled class's (N,h,w) logits.  float64 scalar,
    d loss / d those logits."""
    z, t = np.asarray(sel_logits, np.float64), np.asarray(targets, np.float64)
    per = np.maximum(z, 0) - z * t + np.log1p(np.exp(-np.abs(z)))
    return float(per.mean()), (1.0 / (1.0 + np.exp(-z)) - t) / z.size


def giou_loss_grad(pred, target, w, eps=1e-6):
    """d/dpred of sum_i w_i * giou_loss(pred_i, target_i) by central differences in float64 (checker for the analytic
    backward of det_bbox_loss_bwd; iou_loss.py:78-101)."""
    p = np.asarray(pred, np.float64).copy()
    g = np.zeros_like(p)
    h = 1e-6
    for j in range(4):
        pp, pm = p.copy(), p.copy()
        pp[:, j] += h
        pm[:, j] -= h
        g[:, j] = (giou_loss(pp, target, eps) - giou_loss(pm, target, eps)) / (2 * h) * np.asarray(w, np.float64)
    return g


def accuracy_top1(logits, labels):
    """losses/accuracy.py:5-49, topk=1 -> percent."""
    return float((np.asarray(logits).argmax(1) == np.asarray(labels)).mean() * 100.0)


def crop_and_resize(masks, bboxes, out_shape, inds):
    """BitmapMasks.crop_and_resize (structures.py:328-358): rois = [arange(K), bboxes]; gather masks[inds] as float;
    roi_align(m[:, None], rois, out_shape, 1.0, 0, 'avg', True) >= 0.5  -> bool (K, h, w)."""
    bboxes = np.asarray(bboxes, np.float32).reshape(-1, 4)
    K = bboxes.shape[0]
    oh, ow = (out_shape, out_shape) if isinstance(out_shape, int) else tuple(out_shape)
    if K == 0 or len(masks) == 0:
        return np.zeros((0, oh, ow), bool)
    rois = np.concatenate([np.arange(K, dtype=np.float32)[:, None], bboxes], 1)
    m = np.asarray(masks)[np.asarray(inds, np.int64)].astype(np.float32)[:, None]
    return D.roi_align_c(m, rois, (oh, ow), 1.0, 0, True)[:, 0] >= np.float32(0.5)


def mask_target_single(pos_proposals, pos_assigned_gt_inds, gt_masks, mask_size):
    """mask_target.py:66-122: clip the proposals to the mask's extent, crop_and_resize, float32 0/1."""
    p = np.asarray(pos_proposals, np.float32).reshape(-1, 4).copy()
    oh, ow = (mask_size, mask_size) if isinstance(mask_size, int) else tuple(mask_size)
    if p.shape[0] == 0:
        return np.zeros((0, oh, ow), np.float32)
    maxh, maxw = np.asarray(gt_masks).shape[1:]
    p[:, [0, 2]] = np.clip(p[:, [0, 2]], 0, maxw)
    p[:, [1, 3]] = np.clip(p[:, [1, 3]], 0, maxh)
    return crop_and_resize(gt_masks, p, (oh, ow), pos_assigned_gt_inds).astype(np.float32)


def mask_target(pos_proposals_list, pos_assigned_gt_inds_list, gt_masks_list, mask_size):
    """mask_target.py:6-63: per image, concatenated in image order."""
    out = [mask_target_single(p, i, m, mask_size) for p, i, m in zip(pos_proposals_list, pos_assigned_gt_inds_list, gt_masks_list)]
    return np.concatenate(out, 0) if out else out
