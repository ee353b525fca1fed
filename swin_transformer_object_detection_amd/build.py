"""Build the gfx950 shared library (C ABI in include/swin_hip.h) with hipcc, in-tree.

`python -m swin_transformer_object_detection_amd.build` or `build_library()`.
Objects are compiled in parallel (one hipcc per .hip) and linked into
lib/libswin_hip.so next to this package so the .so travels with the tree.
"""
import concurrent.futures as cf
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libswin_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics", "-Wno-comment"]
# per-file additions.  ts_mlp.hip: hipcc's SLP vectoriser packs the GELU arithmetic into v_pk_fma_f32 / v_pk_mul_f32, which
# cost several times two scalar v_fma_f32 next to MFMAs (guide, "packed f32 VALU ... an anti-lever beside MFMAs").
EXTRA_FLAGS = {"ts_mlp.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build the gfx950 library)")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_all(force=False, verbose=False):
    """both shipped builds: libswin_hip.so (16-bit type = bfloat16) and libswin_hip_f16.so (-DSWIN_HALF: IEEE half)"""
    return build_library(force, verbose), build_library(force, verbose, half=True)


def build_library(force=False, verbose=False, dev=False, half=False):
    """dev=True (or SWIN_DEV_BUILD=1 in the environment): a -DSWIN_DEV build in lib/libswin_hip_dev.so -- the only build in which
    the kernels' launchers read SWIN_* environment variables (geometry sweeps, A/B switches, ablation instantiations).  Select it
    with SWIN_HIP_LIB=<path>; the shipped lib/libswin_hip.so never reads the environment."""
    dev = dev or os.environ.get("SWIN_DEV_BUILD") == "1"
    hipcc = _hipcc()
    tag = ("_f16" if half else "") + ("_dev" if dev else "")
    objdir = os.path.join(LIBDIR, "obj" + tag)
    lib_path = os.path.join(LIBDIR, f"libswin_hip{tag}.so") if tag else LIB
    flags = FLAGS + (["-DSWIN_DEV"] if dev else []) + (["-DSWIN_HALF"] if half else [])
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append([hipcc] + flags + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and (r.stdout or r.stderr):
            print(r.stdout + r.stderr)

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _newer(lib_path, objs):
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib")
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib_path] + objs +
            [f"-L{rocm_lib}", "-lhipblaslt", f"-Wl,-rpath,{rocm_lib}"])     # gemm_lt.hip: plain GEMMs on hipBLASLt
    return lib_path


if __name__ == "__main__":
    import sys
    if "--dev" in sys.argv or "--half" in sys.argv:
        print(build_library(force="--force" in sys.argv, verbose=True, dev="--dev" in sys.argv, half="--half" in sys.argv))
    else:
        print(build_all(force="--force" in sys.argv, verbose=True))
