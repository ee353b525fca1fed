"""``mmcv.ops.nms`` / ``batched_nms`` surface over the HIP kernels.

Call sites in the reference: ``rpn_head.py:233`` and ``bbox_nms.py:84``
(``batched_nms(boxes, scores, idxs, nms_cfg)``).  Build spec:
stable descending sort, suppression test ``inter > thr * (Sa + Sb - inter)`` in fp32.
The bitmask AND its greedy reduction run on the device; the only host
synchronisation is the one the caller's dynamic output shape requires (boolean
indexing of the keep flags), exactly as any torch masked_select.
"""
import torch

from .._lib import SwinHipError, call, lib
from .functional import _p, _s


def nms(boxes, scores, iou_threshold, offset=0, score_threshold=0, max_num=-1):
    """-> (dets (k,5), inds (k,) int64 into the input, descending score order)."""
    assert boxes.size(1) == 4 and boxes.size(0) == scores.size(0) and offset in (0, 1)
    if not boxes.is_cuda:
        raise SwinHipError("nms: HIP path needs GPU tensors (no CPU fallback)")
    boxes = boxes.float()
    scores = scores.float()
    valid_inds = None
    if score_threshold > 0:
        valid = scores > score_threshold
        valid_inds = torch.nonzero(valid, as_tuple=False).squeeze(1)
        boxes, scores = boxes[valid], scores[valid]
    n = boxes.size(0)
    if n == 0:
        inds = torch.zeros(0, dtype=torch.long, device=boxes.device)
    else:
        order = torch.sort(scores, descending=True, stable=True)[1]
        bs = boxes.index_select(0, order).contiguous()
        ws = torch.empty(lib().swin_nms_workspace_bytes(n), dtype=torch.uint8, device=boxes.device)
        flags = torch.empty(n, dtype=torch.uint8, device=boxes.device)
        cnt = torch.empty(1, dtype=torch.int32, device=boxes.device)
        call("nms_sorted", _p(bs), n, float(iou_threshold), int(offset), int(max(max_num, 0)), _p(flags), _p(cnt), None, 0,
             _p(ws), _s())
        inds = order[flags.bool()]
    if max_num > 0:
        inds = inds[:max_num]
    dets = torch.cat((boxes[inds], scores[inds].reshape(-1, 1)), dim=1)
    if valid_inds is not None:
        inds = valid_inds[inds]
    return dets, inds


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """mmcv.ops.batched_nms: NMS within each id of ``idxs`` -> (dets (k,5), keep (k,))."""
    nms_cfg_ = dict(nms_cfg)
    class_agnostic = nms_cfg_.pop('class_agnostic', class_agnostic)
    if boxes.numel() == 0:
        return boxes.new_zeros((0, 5)), torch.zeros(0, dtype=torch.long, device=boxes.device)
    if class_agnostic:
        boxes_for_nms = boxes
    else:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + 1.0)        # mmcv: `+ torch.tensor(1).to(boxes)`; no host->device copy
        boxes_for_nms = boxes + offsets[:, None]
    nms_type = nms_cfg_.pop('type', 'nms')
    if nms_type != 'nms':
        raise NotImplementedError(f"nms type {nms_type!r} is not on the Swin path")
    split_thr = nms_cfg_.pop('split_thr', 10000)
    if boxes_for_nms.shape[0] < split_thr:
        dets, keep = nms(boxes_for_nms, scores, **nms_cfg_)
        boxes = boxes[keep]
        scores = dets[:, 4]
    else:
        max_num = nms_cfg_.pop('max_num', -1)
        total_mask = scores.new_zeros(scores.size(), dtype=torch.bool)
        scores_after_nms = scores.new_zeros(scores.size())
        for id in torch.unique(idxs):
            mask = (idxs == id).nonzero(as_tuple=False).view(-1)
            dets, keep = nms(boxes_for_nms[mask], scores[mask], **nms_cfg_)
            total_mask[mask[keep]] = True
            scores_after_nms[mask[keep]] = dets[:, -1]
        keep = total_mask.nonzero(as_tuple=False).view(-1)
        scores, inds = scores_after_nms[keep].sort(descending=True, stable=True)
        keep = keep[inds]
        boxes = boxes[keep]
        if max_num > 0:
            keep, boxes, scores = keep[:max_num], boxes[:max_num], scores[:max_num]
    return torch.cat([boxes, scores[:, None]], -1), keep


def nms_static(boxes, scores, iou_threshold, max_num, offset=0):
    """Fixed-shape variant for callers that slice to ``max_num`` anyway (rpn_head.py:235): returns
    (inds (max_num,) int64 into the input, valid (max_num,) bool); kept boxes first, in descending-score order.
    No device->host synchronisation (the dynamic count never leaves the device)."""
    if not boxes.is_cuda:
        raise SwinHipError("nms: HIP path needs GPU tensors (no CPU fallback)")
    boxes = boxes.float()
    scores = scores.float()
    n = boxes.size(0)
    pos = torch.empty(max_num, dtype=torch.int32, device=boxes.device)
    if n == 0:
        return torch.zeros(max_num, dtype=torch.long, device=boxes.device), torch.zeros(max_num, dtype=torch.bool, device=boxes.device)
    order = torch.sort(scores, descending=True, stable=True)[1]
    bs = boxes.index_select(0, order).contiguous()
    ws = torch.empty(lib().swin_nms_workspace_bytes(n), dtype=torch.uint8, device=boxes.device)
    flags = torch.empty(n, dtype=torch.uint8, device=boxes.device)
    cnt = torch.empty(1, dtype=torch.int32, device=boxes.device)
    call("nms_sorted", _p(bs), n, float(iou_threshold), int(offset), int(max_num), _p(flags), _p(cnt), _p(pos), int(max_num),
         _p(ws), _s())
    valid = pos >= 0
    return order[pos.clamp(min=0).long()], valid


def batched_nms_static(boxes, scores, idxs, iou_threshold, max_num):
    """batched_nms (single-pass branch, n < split_thr) with a fixed-size result: (dets (max_num,5), valid (max_num,))."""
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + 1.0)        # mmcv: `+ torch.tensor(1).to(boxes)`; no host->device copy
    inds, valid = nms_static(boxes + offsets[:, None], scores, iou_threshold, max_num)
    dets = torch.cat([boxes[inds], scores[inds, None]], -1)
    return torch.where(valid[:, None], dets, torch.zeros_like(dets)), valid


def batched_nms_static_multi(boxes, scores, idxs, iou_threshold, max_num, group_sizes=None):
    """batched_nms_static for a BATCH of images with the same candidate count, in one set of launches:
    boxes (B,n,4), scores (B,n), idxs (B,n) -> (dets (B,max_num,5), valid (B,max_num)).  Per image identical to
    batched_nms_static (same per-image coordinate offset, stable descending sort, greedy suppression).
    ``group_sizes``: the caller's promise that idxs takes the values 0 .. len(group_sizes) - 1 with at most group_sizes[g] members
    per image (the RPN's per-level candidate counts) -- the suppression scans of the ids then run side by side
    (nms_sorted_batch_grouped); same result."""
    if not boxes.is_cuda:
        raise SwinHipError("nms: HIP path needs GPU tensors (no CPU fallback)")
    boxes, scores = boxes.float(), scores.float()
    B, n = scores.shape
    if n == 0:
        return boxes.new_zeros((B, max_num, 5)), torch.zeros((B, max_num), dtype=torch.bool, device=boxes.device)
    boxes, scores, idxs = boxes.contiguous(), scores.contiguous(), idxs.contiguous()
    dev = boxes.device
    grouped = group_sizes is not None and 1 < len(group_sizes) <= 8 and iou_threshold > 0 and n <= 16384 and idxs.dtype == torch.int64
    ws = None if grouped else torch.empty(B * lib().swin_nms_workspace_bytes(n), dtype=torch.uint8, device=dev)
    flags = torch.empty((B, n), dtype=torch.uint8, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    pos = torch.empty((B, max_num), dtype=torch.int32, device=dev)
    if n <= 16384 and idxs.dtype == torch.int64:
        # the whole front end (per-image max coordinate, class offsets, stable descending sort, gather) and the back end (kept
        # boxes + scores, validity) are one launch each (csrc/nms.hip): 5 launches per call instead of ~27
        bs = torch.empty((B, n, 4), dtype=torch.float32, device=dev)
        order = torch.empty((B, n), dtype=torch.int32, device=dev)
        pws = torch.empty(lib().nms_prepare_workspace_bytes(B, n), dtype=torch.uint8, device=dev)
        call("nms_prepare_sorted_batch", _p(boxes), _p(scores), _p(idxs), B, n, _p(bs), _p(order), _p(pws), _s())
        if grouped:
            G, gmax = len(group_sizes), int(max(group_sizes))
            gws = torch.empty(lib().nms_grouped_workspace_bytes(B, n, G, gmax), dtype=torch.uint8, device=dev)
            call("nms_sorted_batch_grouped", _p(bs), _p(order), _p(idxs), B, n, G, gmax, float(iou_threshold), 0, int(max_num), _p(flags),
                 _p(cnt), _p(pos), int(max_num), _p(gws), _s())
        else:
            call("nms_sorted_batch", _p(bs), B, n, float(iou_threshold), 0, int(max_num), _p(flags), _p(cnt), _p(pos), int(max_num), _p(ws),
                 _s())
        dets = torch.empty((B, max_num, 5), dtype=torch.float32, device=dev)
        valid = torch.empty((B, max_num), dtype=torch.bool, device=dev)
        call("nms_gather_dets", _p(boxes), _p(scores), _p(order), _p(pos), B, n, int(max_num), _p(dets), _p(valid), _s())
        return dets, valid
    max_coordinate = boxes.amax(dim=(1, 2))                                        # per image, as one call per image would
    offsets = idxs.to(boxes) * (max_coordinate + 1.0)[:, None]
    order = torch.sort(scores, dim=1, descending=True, stable=True)[1]
    bs = torch.gather(boxes + offsets[..., None], 1, order[..., None].expand(B, n, 4)).contiguous()
    call("nms_sorted_batch", _p(bs), B, n, float(iou_threshold), 0, int(max_num), _p(flags), _p(cnt), _p(pos), int(max_num), _p(ws),
         _s())
    valid = pos >= 0
    inds = torch.gather(order, 1, pos.clamp(min=0).long())
    dets = torch.cat([torch.gather(boxes, 1, inds[..., None].expand(B, max_num, 4)), torch.gather(scores, 1, inds)[..., None]], -1)
    return torch.where(valid[..., None], dets, torch.zeros_like(dets)), valid
