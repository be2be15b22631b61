"""Build the in-tree native pieces: the HIP library for gfx950 (the product)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
HIP_SRC = os.path.join(CSRC, "bk_api.hip")
HIP_LIB = os.path.join(_HERE, "libbreakmer_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _stale(target, deps):
    if not os.path.isfile(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] + [os.path.join(ROOT, "include", "breakmer_hip.h")]
    if force or _stale(HIP_LIB, deps):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result",
               "-o", HIP_LIB, HIP_SRC]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    return HIP_LIB
