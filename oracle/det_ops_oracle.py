"""RoIAlign / nms / batched_nms oracle -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/__init__.py): mmcv-full is a third-party dependency
absent from the reference tree; these follow its published algorithm as used at
the reference call sites (rpn_head.py:233, bbox_nms.py:84,
base_roi_extractor.py:49-55, structures.py:353-354) under the build spec of
SURVEY Appendix B (stable descending sort, strict `>` in the multiplied form).

Two independent statements live here on purpose:
  * ``*_c``  : the plain-C restatement (oracle/csrc/det_ops_ref.c) through ctypes,
               fast enough for the CPU baseline and the full-size cases;
  * ``*_py`` : slow, differently-structured Python/numpy versions (dense
               bilinear weights, O(n^2) IoU matrix) used only to cross-check the C
               code on small cases.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdet_ops_ref.so")
_lib = None


def build(force=False):
    """gcc the C restatement into oracle/_build/ (building the checker is not using it)."""
    src = os.path.join(_HERE, "csrc", "det_ops_ref.c")
    if not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(src):
        return _SO
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-ffp-contract=off",
                           "-o", _SO, src, "-lm"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.nms_ref.restype = ctypes.c_int64
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


# ----------------------------------------------------------------------------
# RoIAlign
# ----------------------------------------------------------------------------
def roi_align_c(inp, rois, output_size, spatial_scale=1.0, sampling_ratio=0, aligned=True):
    """inp (N,C,H,W) f32, rois (K,5) f32 -> (K,C,ph,pw) f32."""
    lib = _load()
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    inp = np.ascontiguousarray(np.asarray(inp, dtype=np.float32))
    rois = np.ascontiguousarray(np.asarray(rois, dtype=np.float32)).reshape(-1, 5)
    N, C, H, W = inp.shape
    K = rois.shape[0]
    out = np.zeros((K, C, ph, pw), dtype=np.float32)
    lib.roi_align_fwd_ref(_fp(inp), _fp(rois), _fp(out), N, C, H, W, K, ph, pw,
                          ctypes.c_float(spatial_scale), int(sampling_ratio), int(bool(aligned)))
    return out


def roi_align_bwd_c(grad_out, rois, in_shape, spatial_scale=1.0, sampling_ratio=0, aligned=True):
    lib = _load()
    grad_out = np.ascontiguousarray(np.asarray(grad_out, dtype=np.float32))
    rois = np.ascontiguousarray(np.asarray(rois, dtype=np.float32)).reshape(-1, 5)
    N, C, H, W = in_shape
    K, _, ph, pw = grad_out.shape
    gi = np.zeros((N, C, H, W), dtype=np.float64)
    lib.roi_align_bwd_ref(_fp(grad_out), _fp(rois), gi.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                          N, C, H, W, K, ph, pw, ctypes.c_float(spatial_scale),
                          int(sampling_ratio), int(bool(aligned)))
    return gi


def _axis_weights(start, bin_sz, n_bins, grid, size):
    """Dense (n_bins, size) matrix of summed 1-D bilinear weights, float32 arithmetic,
    with mmcv's border rules.  Also returns per-sample validity for the 2-D test."""
    f = np.float32
    Wm = np.zeros((n_bins, grid, size), dtype=np.float64)
    valid = np.zeros((n_bins, grid), dtype=bool)
    for b in range(n_bins):
        for g in range(grid):
            t = f(start) + f(b) * f(bin_sz) + f(f(g) + f(.5)) * f(bin_sz) / f(grid)
            if t < -1.0 or t > size:
                continue
            valid[b, g] = True
            if t <= 0:
                t = f(0)
            lo = int(t)
            if lo >= size - 1:
                hi = lo = size - 1
                t = f(lo)
            else:
                hi = lo + 1
            l = f(t) - f(lo)
            h = f(1.) - l
            Wm[b, g, lo] += h
            Wm[b, g, hi] += l
    return Wm, valid


def roi_align_py(inp, rois, output_size, spatial_scale=1.0, sampling_ratio=0, aligned=True):
    """Separable dense-weight restatement (bilinear weights factor per axis):
    out[k,c] = Wy (ph,H) @ inp[b,c] @ Wx^T (W,pw) / count, invalid samples masked jointly."""
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    inp = np.asarray(inp, dtype=np.float64)
    rois = np.asarray(rois, dtype=np.float32).reshape(-1, 5)
    N, C, H, W = inp.shape
    out = np.zeros((rois.shape[0], C, ph, pw))
    f = np.float32
    for k, r in enumerate(rois):
        b = int(r[0])
        off = f(0.5) if aligned else f(0)
        sw, sh = r[1] * f(spatial_scale) - off, r[2] * f(spatial_scale) - off
        ew, eh = r[3] * f(spatial_scale) - off, r[4] * f(spatial_scale) - off
        rw, rh = ew - sw, eh - sh
        if not aligned:
            rw, rh = max(rw, f(1)), max(rh, f(1))
        bh, bw = f(rh) / f(ph), f(rw) / f(pw)
        gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil(f(rh) / f(ph)))
        gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil(f(rw) / f(pw)))
        cnt = max(gh * gw, 1)
        if gh <= 0 or gw <= 0:
            continue
        Wy, vy = _axis_weights(sh, bh, ph, gh, H)
        Wx, vx = _axis_weights(sw, bw, pw, gw, W)
        # a sample contributes only when BOTH coordinates are in range
        Wy = Wy * vy[:, :, None]
        Wx = Wx * vx[:, :, None]
        Wys, Wxs = Wy.sum(1), Wx.sum(1)  # (ph,H), (pw,W)
        out[k] = np.einsum("ih,chw,jw->cij", Wys, inp[b], Wxs) / cnt
    return out.astype(np.float32)


# ----------------------------------------------------------------------------
# nms / batched_nms
# ----------------------------------------------------------------------------
def nms_c(boxes, scores, iou_threshold, offset=0):
    """-> (dets (k,5) f32, keep (k,) int64 in descending-score order)."""
    lib = _load()
    boxes = np.ascontiguousarray(np.asarray(boxes, dtype=np.float32)).reshape(-1, 4)
    scores = np.ascontiguousarray(np.asarray(scores, dtype=np.float32)).reshape(-1)
    n = boxes.shape[0]
    keep = np.zeros(max(n, 1), dtype=np.int64)
    m = lib.nms_ref(_fp(boxes), _fp(scores), ctypes.c_int64(n), ctypes.c_float(iou_threshold),
                    int(offset), keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    keep = keep[:m]
    dets = np.concatenate([boxes[keep], scores[keep, None]], axis=1) if m else np.zeros((0, 5), np.float32)
    return dets, keep


def nms_py(boxes, scores, iou_threshold, offset=0):
    """Brute force: full pairwise 'suppresses' matrix in fp32, then the greedy scan."""
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    n = len(scores)
    order = np.argsort(-scores, kind="stable")
    b = boxes[order]
    off = np.float32(offset)
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    area = (x2 - x1 + off) * (y2 - y1 + off)
    w = np.maximum(np.minimum(x2[:, None], x2[None]) - np.maximum(x1[:, None], x1[None]) + off, np.float32(0))
    h = np.maximum(np.minimum(y2[:, None], y2[None]) - np.maximum(y1[:, None], y1[None]) + off, np.float32(0))
    inter = (w * h).astype(np.float32)
    sup = inter > np.float32(iou_threshold) * (area[:, None] + area[None] - inter)
    alive = np.ones(n, bool)
    keep = []
    for i in range(n):
        if alive[i]:
            keep.append(order[i])
            alive[i + 1:] &= ~sup[i, i + 1:]
    keep = np.asarray(keep, dtype=np.int64)
    dets = np.concatenate([boxes[keep], scores[keep, None]], 1) if len(keep) else np.zeros((0, 5), np.float32)
    return dets, keep


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False, nms_fn=None):
    """mmcv.ops.batched_nms as called at rpn_head.py:233 / bbox_nms.py:84.

    boxes (n,4) f32, scores (n,), idxs (n,) int64 -> (dets (k,5), keep (k,) int64)."""
    nms_fn = nms_fn or nms_c
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    idxs = np.asarray(idxs, dtype=np.int64).reshape(-1)
    cfg = dict(nms_cfg)
    class_agnostic = cfg.pop("class_agnostic", class_agnostic)
    assert cfg.pop("type", "nms") == "nms"
    split_thr = cfg.pop("split_thr", 10000)
    thr = cfg.pop("iou_threshold")
    max_num = cfg.pop("max_num", -1)
    if boxes.shape[0] == 0:
        return np.zeros((0, 5), np.float32), np.zeros((0,), np.int64)
    if class_agnostic:
        bfn = boxes
    else:
        max_coord = boxes.max()
        offs = idxs.astype(np.float32) * (max_coord + np.float32(1))
        bfn = boxes + offs[:, None]
    if boxes.shape[0] < split_thr:
        dets, keep = nms_fn(bfn, scores, thr)
        if max_num > 0:
            dets, keep = dets[:max_num], keep[:max_num]
        out_boxes, out_scores = boxes[keep], dets[:, 4]
    else:
        total = np.zeros(len(scores), bool)
        for i in np.unique(idxs):
            m = np.nonzero(idxs == i)[0]
            _, k = nms_fn(bfn[m], scores[m], thr)
            total[m[k]] = True
        keep = np.nonzero(total)[0]
        inds = np.argsort(-scores[keep], kind="stable")
        keep = keep[inds]
        if max_num > 0:
            keep = keep[:max_num]
        out_boxes, out_scores = boxes[keep], scores[keep]
    return np.concatenate([out_boxes, out_scores[:, None]], 1).astype(np.float32), keep.astype(np.int64)
