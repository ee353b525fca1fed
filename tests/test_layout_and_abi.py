"""CPU-side contract checks: the C-ABI library loads and exports every symbol include/swin_hip.h declares
(no compute calls), the product never imports the oracle, and the product fails loudly without the
library / without a GPU instead of falling back."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "swin_transformer_object_detection_amd")


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "swin_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"^\s*(?:int|int64_t)\s+(\w+)\s*\(", src, flags=re.M)))


@pytest.fixture(scope="module")
def built_lib():
    from swin_transformer_object_detection_amd.build import build_library
    return build_library()


def test_header_symbols_exported(built_lib):
    lib = ctypes.CDLL(built_lib)
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/swin_hip.h but not exported"


def test_fp16_build_exports_the_same_abi():
    """libswin_hip_f16.so (-DSWIN_HALF: the 16-bit type is IEEE half, the reference's apex O1 precision) is the same ABI: every
    declared symbol, and it says which build it is (pure host calls)."""
    from swin_transformer_object_detection_amd.build import build_library
    lib = ctypes.CDLL(build_library(half=True))
    for s in _declared_symbols():
        assert hasattr(lib, s), f"{s} missing from the fp16 build"
    assert lib.swin_hip_half_type() == 1 and lib.swin_hip_abi_version() == 2


def test_half_dtype_choice_is_per_process():
    """SWIN_HALF_DTYPE=fp16 selects the fp16 library; the bf16 process refuses to switch once its library is loaded"""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import torch\n"
            "from swin_transformer_object_detection_amd import _lib\n"
            "assert _lib.half_dtype() == torch.float16 and _lib.LIB_PATH.endswith('libswin_hip_f16.so')\n"
            "assert _lib.lib().swin_hip_half_type() == 1\n"
            "try:\n    _lib.set_half_dtype(torch.bfloat16)\nexcept _lib.SwinHipError:\n    print('REFUSED')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, SWIN_HALF_DTYPE="fp16"))
    assert r.returncode == 0 and "REFUSED" in r.stdout, r.stdout + r.stderr


def test_ctypes_table_matches_header(built_lib):
    from swin_transformer_object_detection_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    assert _lib.lib().swin_hip_abi_version() == 2          # pure host call
    assert _lib.lib().swin_hip_half_type() == 0 and _lib.half_dtype() == torch.bfloat16


def test_header_cites_reference_for_every_entry_point():
    src = open(os.path.join(ROOT, "include", "swin_hip.h")).read()
    for ref in ("swin_transformer.py:", "fpn.py:", "rpn_head.py:233", "bbox_nms.py:84", "base_roi_extractor.py:49-55",
                "structures.py:353-354"):
        assert ref in src


def test_product_never_imports_oracle():
    bad = []
    for dp, _, fs in os.walk(PKG):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "oracle/" in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, f"product files reference the oracle: {bad}"
    for f in ("bench.py", "__graft_entry__.py"):
        assert os.path.exists(os.path.join(ROOT, f))


def test_no_cpu_fallback():
    from swin_transformer_object_detection_amd import ops
    from swin_transformer_object_detection_amd._lib import SwinHipError
    with pytest.raises(SwinHipError):
        ops.layer_norm(torch.zeros(4, 32), torch.ones(32), torch.zeros(32))
    with pytest.raises(SwinHipError):
        ops.nms(torch.zeros(3, 4), torch.zeros(3), 0.5)
    with pytest.raises(SwinHipError):
        ops.roi_align(torch.zeros(1, 4, 8, 8), torch.zeros(1, 5), 7, 1.0, 0, 'avg', True)


def test_missing_library_is_an_error(tmp_path):
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from swin_transformer_object_detection_amd import _lib\n"
            "_lib.LIB_PATH = %r\n"
            "try:\n    _lib.lib()\nexcept _lib.SwinHipError as e:\n    print('RAISED'); sys.exit(0)\nsys.exit(1)\n"
            % (ROOT, str(tmp_path / "nope.so")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and "RAISED" in r.stdout, r.stderr
