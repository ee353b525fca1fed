"""CPU oracle for the Swin detection hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU fp32 for the floating-point ops,
numpy / plain C for index and integer work) of the reference algorithm for the
hot path named in BASELINE.json.  Every function cites the reference file:line
it follows (paths relative to the upstream repository root).

Rules (enforced by tests/test_layout.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here;
  * the product package (``swin_transformer_object_detection_amd``) never
    imports it and never falls back to it: the product path raises when the
    HIP library is missing.

Pinning status
  * Swin backbone + FPN (swin_oracle.py, fpn_oracle.py): PINNED -- checked
    against golden vectors produced by importing the reference's own
    ``swin_transformer.py`` / ``fpn.py`` in the build container
    (tests/golden/make_golden.py; fixtures committed under tests/golden/).
  * RoIAlign / nms / batched_nms (roi_align_oracle.py, nms_oracle.py):
    PARITY UNPINNED -- the arithmetic lives in the third-party wheel
    mmcv-full (1.2.4 <= mmcv <= 1.4.0, ``mmdet/__init__.py:18-19``) whose
    sources are not under the reference tree and which is not installed.  The
    restatement follows mmcv's published algorithm and is pinned only by
    hand-computed known-answer cases and brute-force properties.
  * Python callers (RPN proposal selection, RoI level mapping,
    multiclass_nms, delta2bbox, anchors: rpn_oracle.py): restated from the
    reference's source text; the reference modules cannot be imported (mmcv),
    the known-answer cases of the reference's own tests that apply
    (``tests/test_utils/test_coder.py:26-60``) are reproduced in
    tests/test_oracle_callers.py.
"""
