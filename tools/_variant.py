"""Tools and probes only: pick the build of the library to load (BK_VARIANT=checkjit|jitter|check|diag|stamps, or BK_LIB=<path>).
The product (breakmer_amd/) reads no such variable; this module calls hip_backend.load_library(path) before any engine exists."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breakmer_amd import build, hip_backend  # noqa: E402


def use():
    path = os.environ.get("BK_LIB")
    v = os.environ.get("BK_VARIANT", "")
    if not path and v:
        path = build.lib_path(v)
    if path:
        hip_backend.load_library(path)
        print("[tools] library: %s" % path, flush=True)
    return path
