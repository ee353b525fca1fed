"""Device-side training targets: MaxIoUAssigner and RandomSampler behind the C ABI (csrc/det_targets.hip).

Reference call sites: anchor_head.py:213-219 (assign + sample per image for the RPN), standard_roi_head.py:83-93
(the same for the RoI head).  Results have fixed shapes, so nothing here synchronises with the host."""
import ctypes

import torch

from .._lib import half_dtype as _H

from .. import _lib
from .._lib import SwinHipError, call
from .functional import _p, _s


def max_iou_assign(bboxes, gt_bboxes, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0, match_low_quality=True, gt_labels=None,
                   num_leading_gt=0, valid=None):
    """MaxIoUAssigner.assign (max_iou_assigner.py:85-212, ignore_iof_thr=-1, gt_max_assign_all=True).

    bboxes (n,4), gt_bboxes (g,4) float32 xyxy on the GPU -> (assigned_gt_inds (n) int64, max_overlaps (n) float32,
    assigned_labels (n) int64 or None).  ``num_leading_gt``: the first rows of bboxes are the gts (add_gt_as_proposals)
    and are matched to themselves, as AssignResult.add_gt_ does.  ``valid`` (n) bool: False rows get -1."""
    if not bboxes.is_cuda:
        raise SwinHipError("max_iou_assign: GPU tensors only")
    if isinstance(neg_iou_thr, (tuple, list)):
        raise SwinHipError("max_iou_assign: a (lo, hi) neg_iou_thr range is not supported")
    bboxes = bboxes.detach().float().contiguous()
    gt_bboxes = gt_bboxes.detach().float().contiguous()
    n, g = bboxes.size(0), gt_bboxes.size(0)
    assigned = torch.empty(n, dtype=torch.long, device=bboxes.device)
    max_ov = torch.empty(n, dtype=torch.float32, device=bboxes.device)
    labels = torch.empty(n, dtype=torch.long, device=bboxes.device) if gt_labels is not None else None
    if n == 0:
        return assigned, max_ov, labels
    ws = torch.empty(_lib.lib().det_assign_workspace_bytes(n, g), dtype=torch.uint8, device=bboxes.device)
    gl = gt_labels.long().contiguous() if gt_labels is not None else None
    vm = None
    if valid is not None:
        vm = valid.contiguous().view(torch.uint8) if valid.dtype == torch.bool else valid.to(torch.uint8).contiguous()
    call("det_max_iou_assign", _p(bboxes), n, _p(gt_bboxes) if g else None, g, _p(gl) if gl is not None and g else None,
         float(pos_iou_thr), float(neg_iou_thr), float(min_pos_iou), int(bool(match_low_quality)), int(num_leading_gt),
         _p(vm) if vm is not None else None, _p(assigned), _p(max_ov), _p(labels) if labels is not None else None, _p(ws), _s())
    return assigned, max_ov, labels


def _next_seed():
    """64-bit seed drawn on the HOST from torch's CPU generator (follows torch.manual_seed; no device op, no sync)."""
    return int(torch.randint(0, 2 ** 63 - 1, (1,), dtype=torch.int64))


# Device-resident step seed (one int64 word per device).  None = off: every call's randomness is its host seed.  A training step
# that is captured in a hipGraph (graph_step.py) turns it on: the host seeds of the captured calls are frozen into the graph, the
# device word is rewritten before every replay (set_step_seed) and the kernels mix the two.
_STEP_SEED = {}


def step_seed_tensor(device, create=False):
    k = device.index if device.index is not None else torch.cuda.current_device()
    t = _STEP_SEED.get(k)
    if t is None and create:
        t = _STEP_SEED[k] = torch.zeros(1, dtype=torch.int64, device=device)
    return t


def set_step_seed(device, value=None):
    """Write this step's seed word on ``device`` (one tiny launch, the value travels as a kernel argument: no host->device copy).
    value None: drawn from torch's CPU generator."""
    t = step_seed_tensor(device, create=True)
    v = _next_seed() if value is None else int(value)
    with torch.cuda.device(t.device):
        call("swin_set_u64", _p(t), ctypes.c_uint64(v & (2 ** 64 - 1)), _s())
    return t


def disable_step_seed(device=None):
    if device is None:
        _STEP_SEED.clear()
    else:
        _STEP_SEED.pop(device.index if device.index is not None else torch.cuda.current_device(), None)


def _seed_dev(device):
    t = step_seed_tensor(device)
    return None if t is None else _p(t)


def random_sample(assigned_gt_inds, num, pos_fraction, seed=None):
    """RandomSampler.sample (random_sampler.py:31-78, neg_pos_ub=-1) with a fixed-size result.

    -> (inds (num,) int64, is_pos (num,) bool, valid (num,) bool): a uniformly random subset of
    min(#pos, int(num * pos_fraction)) positives first, then uniformly random negatives up to ``num`` in total;
    unused slots have valid == False."""
    a = assigned_gt_inds
    if not a.is_cuda:
        raise SwinHipError("random_sample: GPU tensors only")
    a = a.long().contiguous()
    inds = torch.empty(num, dtype=torch.long, device=a.device)
    flags = torch.empty(num, dtype=torch.uint8, device=a.device)
    ws = torch.empty(_lib.lib().det_random_sample_workspace_bytes(), dtype=torch.uint8, device=a.device)
    call("det_random_sample", _p(a) if a.numel() else None, a.numel(), int(num), int(num * pos_fraction),
         _next_seed() if seed is None else int(seed), _seed_dev(a.device) if seed is None else None, _p(inds), _p(flags), _p(ws),
         _s())
    return inds, flags >= 2, flags >= 1


def _f4(v):
    return (ctypes.c_float * 4)(*[float(x) for x in v])


def random_sample_raw(assigned_gt_inds, num, pos_fraction, seed=None, out=None):
    """random_sample returning the kernel's (inds, flags) pair (bit 0 used, bit 1 positive) for bbox_targets."""
    a = assigned_gt_inds.long().contiguous()
    if out is not None:                       # rows of batch-level tensors provided by the caller
        inds, flags = out
        if not (inds.is_contiguous() and flags.is_contiguous() and inds.numel() == num and flags.numel() == num
                and inds.dtype == torch.long and flags.dtype == torch.uint8):
            raise SwinHipError("random_sample_raw: out = (int64 (num,), uint8 (num,)) contiguous tensors")
    else:
        inds = torch.empty(num, dtype=torch.long, device=a.device)
        flags = torch.empty(num, dtype=torch.uint8, device=a.device)
    ws = torch.empty(_lib.lib().det_random_sample_workspace_bytes(), dtype=torch.uint8, device=a.device)
    call("det_random_sample", _p(a) if a.numel() else None, a.numel(), int(num), int(num * pos_fraction),
         _next_seed() if seed is None else int(seed), _seed_dev(a.device) if seed is None else None, _p(inds), _p(flags), _p(ws),
         _s())
    return inds, flags


def bbox_targets(bboxes, inds, flags, assigned_gt_inds, gt_bboxes, means, stds, assigned_labels=None, bg_label=0, out_deltas=None):
    """Targets of a fixed-size sample in one launch: the gathers of anchor_head.py:221-247 / bbox_head.py:140-186 and
    DeltaXYWHBBoxCoder.encode (delta_xywh_bbox_coder.py:82-130).

    -> (boxes (k,4) [unused slots (0,0,1,1)], deltas (k,4) [zero unless positive], gt_inds (k) int64,
        labels (k) int64 or None [bg_label unless positive])."""
    k = inds.numel()
    dev = bboxes.device
    boxes = torch.empty(k, 4, dtype=torch.float32, device=dev)
    deltas = torch.empty(k, 4, dtype=torch.float32, device=dev) if out_deltas is None else out_deltas
    if not (deltas.is_contiguous() and deltas.dtype == torch.float32 and deltas.numel() == 4 * k):
        raise SwinHipError("bbox_targets: out_deltas must be a contiguous float32 (k,4) tensor")
    gt_inds = torch.empty(k, dtype=torch.long, device=dev)
    labels = torch.empty(k, dtype=torch.long, device=dev) if assigned_labels is not None else None
    g = gt_bboxes.size(0)
    bb, gb = bboxes.detach().float().contiguous(), gt_bboxes.detach().float().contiguous()
    call("det_bbox_targets", _p(bb), _p(inds), _p(flags), _p(assigned_gt_inds), _p(gb) if g else None, g,
         _p(assigned_labels) if assigned_labels is not None else None, int(bg_label), _f4(means), _f4(stds), k, _p(boxes), _p(deltas),
         _p(gt_inds), _p(labels) if labels is not None else None, _s())
    return boxes, deltas, gt_inds, labels


class RoiStageBuffers:
    """Batch-level tensors of one R-CNN training stage (nimg images x ``num`` sample slots, the first ``km`` of each image
    being its mask slots); ``roi_targets_pack`` fills one image's rows per launch."""

    def __init__(self, nimg, num, km, device):
        e = lambda *sh, dt=torch.float32: torch.empty(*sh, dtype=dt, device=device)      # noqa: E731
        self.nimg, self.num, self.km = nimg, num, km
        self.rois = e(nimg * num, 5)
        self.targets = e(nimg * num, 4)
        self.labels = e(nimg * num, dt=torch.long)
        self.pos = e(nimg * num, dt=torch.bool)
        self.valid = e(nimg * num, dt=torch.bool)
        self.is_gt = e(nimg * num, dt=torch.bool)
        self.inds = e(nimg, num, dt=torch.long)       # the sampler's own outputs (det_random_sample), one row per image:
        self.flags = e(nimg, num, dt=torch.uint8)     # bit 0 used, bit 1 positive -- the form the loss kernels read
        self.feat_rois = e(nimg * km, 5)          # bbox2roi of the mask slots (RoI extractor input)
        self.mask_rois = e(nimg * km, 5)          # [gt mask index (+ offset), clipped box] (crop_and_resize input)
        self.mlabels = e(nimg * km, dt=torch.long)
        self.mvalid = e(nimg * km, dt=torch.bool)


def roi_targets_pack(buf, img, bboxes, inds, flags, assigned_gt_inds, gt_bboxes, means, stds, assigned_labels, bg_label,
                     num_leading_gt, reg_decoded, gt_offset, mask_hw):
    """Image ``img``'s rows of ``buf`` (RoiStageBuffers) from its fixed-size sample, one launch: the sampled boxes with the
    image index column (bbox2roi), regression targets (bbox_head.py:140-186), labels, positive / used / pos_is_gt flags,
    and for the mask slots the feature RoIs, the rows mask_target.py:95-107 hands to crop_and_resize (gt index + gt_offset,
    box clipped to mask_hw = (h, w)), the labels clamped below the background label and the validity."""
    k, km = buf.num, buf.km
    if inds.numel() != k:
        raise SwinHipError("roi_targets_pack: sample size differs from the buffers'")
    g = gt_bboxes.size(0)
    bb, gb = bboxes.detach().float().contiguous(), gt_bboxes.detach().float().contiguous()
    r0, m0 = img * k, img * km
    call("det_roi_targets_pack", _p(bb), _p(inds), _p(flags), _p(assigned_gt_inds), _p(gb) if g else None, g,
         _p(assigned_labels) if assigned_labels is not None else None, int(bg_label), _f4(means), _f4(stds), k, int(img),
         int(num_leading_gt), int(bool(reg_decoded)), _p(buf.rois[r0:]), _p(buf.targets[r0:]), _p(buf.labels[r0:]), _p(buf.pos[r0:]),
         _p(buf.valid[r0:]), _p(buf.is_gt[r0:]), km, int(gt_offset), float(mask_hw[0]), float(mask_hw[1]),
         _p(buf.feat_rois[m0:]) if km else None, _p(buf.mask_rois[m0:]) if km else None, _p(buf.mlabels[m0:]) if km else None,
         _p(buf.mvalid[m0:]) if km else None, _s())


def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None, wh_ratio_clip=16 / 1000):
    """DeltaXYWHBBoxCoder.decode (delta_xywh_bbox_coder.py:189-237) for (n,4) float32 rois / deltas on the GPU."""
    if not rois.is_cuda:
        raise SwinHipError("delta2bbox: GPU tensors only")
    r, d = rois.detach().float().contiguous(), deltas.detach().float().contiguous()
    out = torch.empty_like(r)
    mh, mw = (float(max_shape[0]), float(max_shape[1])) if max_shape is not None else (0.0, 0.0)
    call("det_delta2bbox", _p(r), _p(d), r.size(0), _f4(means), _f4(stds), mh, mw, float(wh_ratio_clip), _p(out), _s())
    return out


def map_roi_levels(rois, num_levels, finest_scale=56, valid=None):
    """SingleRoIExtractor.map_roi_levels (single_level_roi_extractor.py:32-51) for (K,5) float32 rois on the GPU -> (K,) int32
    levels; rows where ``valid`` (K,) bool is False get -1 (skipped by roi_align_multilevel).  One launch."""
    if not rois.is_cuda:
        raise SwinHipError("map_roi_levels: GPU tensors only")
    r = rois.detach().float().contiguous()
    K = r.size(0)
    out = torch.empty(K, device=r.device, dtype=torch.int32)
    v = None
    if valid is not None:
        v = valid.contiguous().view(torch.uint8) if valid.dtype == torch.bool else valid.to(torch.uint8).contiguous()
    call("det_map_roi_levels", _p(r), _p(v), K, int(num_levels), float(finest_scale), _p(out), _s())
    return out


def rpn_topk_decode(cls_all, reg_all, anchors, level_sizes, nms_pre, means, stds, max_shape):
    """RPNHead._get_bboxes up to batched_nms (rpn_head.py:126-187) in one launch: cls_all (B,total) logits, reg_all
    (B,total,4) deltas (float32 or bfloat16, anchors flattened (level,h,w,a)), anchors (total,4) float32 ->
    (scores (B,n) f32, proposals (B,n,4) f32, level ids (B,n) int64), n = sum_l min(n_l, nms_pre); inside a level the
    survivors come in ascending anchor order (sorting by score, stably, restores the reference's order)."""
    if not cls_all.is_cuda:
        raise SwinHipError("rpn_topk_decode: GPU tensors only")
    from .._lib import SWIN_BF16, SWIN_F32, lib
    c, r = cls_all.detach().contiguous(), reg_all.detach().contiguous()
    if c.dtype != r.dtype or c.dtype not in (torch.float32, _H()):
        c, r = c.float(), r.float()
    B, total = c.shape
    if sum(level_sizes) != total or r.shape != (B, total, 4) or anchors.shape != (total, 4):
        raise SwinHipError(f"rpn_topk_decode: shapes {tuple(c.shape)} / {tuple(r.shape)} / {tuple(anchors.shape)} vs levels {level_sizes}")
    a = anchors.detach().float().contiguous()
    n = sum(min(int(s), int(nms_pre)) for s in level_sizes)
    dev = c.device
    scores = torch.empty((B, n), device=dev, dtype=torch.float32)
    boxes = torch.empty((B, n, 4), device=dev, dtype=torch.float32)
    ids = torch.empty((B, n), device=dev, dtype=torch.int64)
    ws = torch.empty(max(lib().det_rpn_topk_decode_workspace_bytes(B, total), 16), device=dev, dtype=torch.uint8)
    ls = (ctypes.c_int * len(level_sizes))(*[int(s) for s in level_sizes])
    mh, mw = (float(max_shape[0]), float(max_shape[1])) if max_shape is not None else (0.0, 0.0)
    call("det_rpn_topk_decode", _p(c), _p(r), _p(a), ls, len(level_sizes), B, int(nms_pre), _f4(means), _f4(stds), mh, mw, _p(ws),
         _p(scores), _p(boxes), _p(ids), SWIN_F32 if c.dtype == torch.float32 else SWIN_BF16, _s())
    return scores, boxes, ids


def regress_by_class(rois, labels, cls_score, bbox_pred, num_classes, class_agnostic, means, stds, max_shape=None):
    """BBoxHead.regress_by_class (bbox_head.py:409-436) with CascadeRoIHead's label choice (cascade_roi_head.py:274-284,
    :316-323): rois (n,4); labels (n,) int64 or None -- None / background labels take argmax(cls_score[:, :-1]);
    returns the (n,4) refined boxes clipped to ``max_shape``."""
    if not rois.is_cuda:
        raise SwinHipError("regress_by_class: GPU tensors only")
    from .._lib import SWIN_BF16, SWIN_F32
    c, b = cls_score.detach().contiguous(), bbox_pred.detach().contiguous()
    if c.dtype != b.dtype or c.dtype not in (torch.float32, _H()):
        c, b = c.float(), b.float()
    r = rois.detach().float().contiguous()
    n = r.size(0)
    if c.shape != (n, num_classes + 1) or b.numel() != n * (4 if class_agnostic else 4 * num_classes):
        raise SwinHipError(f"regress_by_class: shapes {tuple(c.shape)} / {tuple(b.shape)} do not match {n} RoIs")
    out = torch.empty_like(r)
    mh, mw = (float(max_shape[0]), float(max_shape[1])) if max_shape is not None else (0.0, 0.0)
    lab = None if labels is None else labels.long().contiguous()
    call("det_regress_by_class", _p(r), _p(lab), _p(c), _p(b), n, int(num_classes), 1 if class_agnostic else 0, _f4(means), _f4(stds),
         mh, mw, _p(out), SWIN_F32 if c.dtype == torch.float32 else SWIN_BF16, _s())
    return out


def paste_masks(mask_logits, labels, boxes, img_h, img_w, thr, is_prob=False):
    """FCNMaskHead.get_seg_masks pasting (fcn_mask_head.py:257-300 with _do_paste_mask :303-377) in one kernel:
    mask_logits (N, num_classes, mh, mw), labels (N,), boxes (N,4) in output-image coordinates ->
    (N, img_h, img_w) bool: sigmoid mask of the labelled class resampled into its box, >= thr.  ``is_prob``: the input
    already holds probabilities (stage-averaged masks of CascadeRoIHead)."""
    if not mask_logits.is_cuda:
        raise SwinHipError("paste_masks: GPU tensors only")
    if mask_logits.dtype not in (torch.float32, _H()):
        mask_logits = mask_logits.float()
    m = mask_logits.detach().contiguous()
    N, nc, mh, mw = m.shape
    out = torch.empty((N, int(img_h), int(img_w)), dtype=torch.uint8, device=m.device)
    if N:
        from .._lib import SWIN_BF16, SWIN_F32
        call("det_paste_masks", _p(m), _p(labels.long().contiguous()), _p(boxes.detach().float().contiguous()), N, nc, mh, mw,
             int(img_h), int(img_w), float(thr), 1 if is_prob else 0, SWIN_F32 if m.dtype == torch.float32 else SWIN_BF16, _p(out),
             _s())
    return out.bool()
