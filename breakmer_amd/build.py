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


# Diagnostic variants of the library (never loaded by the product: tests and tools name them through
# hip_backend.load_library(path)).  The product build has no -D at all and reads nothing from the environment.
VARIANTS = {
    "": [],
    "diag": ["-DBK_DIAG"],                                   # the environment switches of bk_api.hip (BK_DEBUG_SPLIT, BK_POISON_*, ...)
    "jitter": ["-DBK_JITTER"],                               # a pseudo-random subset of the wavefronts sleeps behind every barrier (bk_common.h)
    "check": ["-DBK_SYNC_CHECK"],                            # every barrier verifies that all wavefronts stand at the same site
    "checkjit": ["-DBK_SYNC_CHECK", "-DBK_JITTER"],
    "stamps": ["-DBK_DIAG", "-DBK_PHASE_STAMPS"],            # per-phase times (tools/phase_probe_*.py)
}


def lib_path(variant=""):
    return HIP_LIB if not variant else os.path.join(_HERE, "libbreakmer_hip_%s.so" % variant)


def build_hip(force=False, verbose=False, variant="", extra=()):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] + [os.path.join(ROOT, "include", "breakmer_hip.h")]
    out = lib_path(variant)
    if force or _stale(out, deps):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result"] + VARIANTS[variant] + list(extra) + ["-o", out, HIP_SRC]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    return out


if __name__ == "__main__":
    import sys
    for v in (sys.argv[1:] or [""]):
        print(build_hip(verbose=True, variant=v))
