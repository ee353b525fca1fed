"""BASELINE configs[4] asks for fp16 (the reference's apex O1, mmdet/apis/train.py:82-89): the fp16 build of the library
(libswin_hip_f16.so, the same sources with IEEE half as the 16-bit type) runs in a process of its own -- a process works with one
16-bit type -- against the same fp32 oracles as the bf16 build, including a training loop with device-side dynamic loss scaling."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fp16_build_suite():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, SWIN_HALF_DTYPE="fp16")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fp16_suite.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "FP16 SUITE OK" in r.stdout, r.stdout[-3000:] + "\n" + r.stderr[-3000:]


def test_one_process_one_half_type():
    """the 16-bit type cannot change once the library is loaded"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from swin_transformer_object_detection_amd import _lib
    _lib.lib()
    other = torch.float16 if _lib.half_dtype() == torch.bfloat16 else torch.bfloat16
    with pytest.raises(_lib.SwinHipError):
        _lib.set_half_dtype(other)
