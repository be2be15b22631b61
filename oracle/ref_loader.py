"""TEST INFRASTRUCTURE ONLY -- loader for the *real* reference (Python 2) in this container.

This module imports the reference sources where they lie under /root/reference
through an in-memory lib2to3 translation.
Nothing is copied into the repo: sources are read, patched, translated and
exec'd in memory.  It is used only by tools/make_golden.py (fixture generation)
and by tests that are skipped when /root/reference is absent (i.e. on the GPU
box).  The product (breakmer_amd/) never imports anything from oracle/.

Canonicalisation patches applied before translation (recorded in every fixture):
  P1  sv_assembly.py:129  len(seq)/2 -> len(seq)//2     (restores Py2 int division)
  P2  sv_assembly.py:575  list(x)    -> sorted(x)        (removes set-hash order)
  P3  Py2 round() (half away from zero, float result) injected in module globals
  P4  fq_recs passed as insertion-ordered dict in FASTQ order (caller's duty)
"""
import math
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("BREAKMER_REFERENCE", "/root/reference")
MODULES = ["utils", "olc", "sv_assembly", "sv_caller", "sv_processor"]
PATCHES = {
    "sv_assembly": [
        ("m = len(seq)/2 ", "m = len(seq)//2 "),                       # P1
        ("for mer in list(x) :", "for mer in sorted(x) :"),            # P2
    ],
    # the reference prints debugging output from get_brkpt_coverages (sv_caller.py:104-108)
    # and opens the BAM; both are outside the pinned path -> stubbed via pysam stub below.
}
PATCH_IDS = ["P1", "P2", "P3", "P4"]


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "sv_assembly.py"))


def py2_round(x, n=0):
    """Python 2.7 round(): correctly rounded on the exact binary value, exact
    halfway cases away from zero, float result (P3; Objects/floatobject.c
    _Py_double_round of CPython 2.7)."""
    from decimal import Decimal, ROUND_HALF_UP
    d = Decimal(float(x)).quantize(Decimal(1).scaleb(-int(n)), rounding=ROUND_HALF_UP)
    return float(d)


class _AlignedRead(object):
    pass


def _install_stubs():
    if "Bio" not in sys.modules:
        bio = types.ModuleType("Bio")
        seqio = types.ModuleType("Bio.SeqIO")
        bio.SeqIO = seqio
        sys.modules["Bio"] = bio
        sys.modules["Bio.SeqIO"] = seqio
    if "pysam" not in sys.modules:
        pysam = types.ModuleType("pysam")

        class Samfile(object):
            def __init__(self, *a, **k):
                pass

            def fetch(self, *a, **k):
                return []

            def close(self):
                pass

            def write(self, *a, **k):
                pass

        def sort(*a, **k):
            pass

        def index(*a, **k):
            pass

        pysam.Samfile = Samfile
        pysam.sort = sort
        pysam.index = index
        pysam.__all__ = ["Samfile", "sort", "index"]
        sys.modules["pysam"] = pysam


_loaded = {}


def load(quiet_prints=True):
    """Return dict name -> module object of the translated reference."""
    if _loaded:
        return _loaded
    if not available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor
    _install_stubs()
    fixers = refactor.get_fixers_from_package("lib2to3.fixes")
    tool = refactor.RefactoringTool(fixers)
    sources = {}
    for name in MODULES:
        with open(os.path.join(REFERENCE_ROOT, name + ".py")) as f:
            src = f.read().expandtabs(8)
        for old, new in PATCHES.get(name, []):
            assert old in src, (name, old)
            src = src.replace(old, new)
        if not src.endswith("\n"):
            src += "\n"
        sources[name] = str(tool.refactor_string(src, name + ".py"))
    # circular import sv_processor <-> sv_caller: create all module objects first
    mods = {}
    for name in MODULES:
        m = types.ModuleType(name)
        m.__file__ = os.path.join(REFERENCE_ROOT, name + ".py")
        m.round = py2_round                                            # P3
        if quiet_prints:
            m.print = lambda *a, **k: None
        mods[name] = m
    saved = {n: sys.modules.get(n) for n in MODULES}
    try:
        for name in MODULES:
            sys.modules[name] = mods[name]
        for name in MODULES:
            code = compile(sources[name], mods[name].__file__, "exec")
            exec(code, mods[name].__dict__)
        # `from sv_processor import *` inside sv_caller ran before sv_processor was
        # populated (circular); re-export now, as Python 2 would have resolved it.
        for k, v in mods["sv_processor"].__dict__.items():
            if not k.startswith("_") and k not in mods["sv_caller"].__dict__:
                mods["sv_caller"].__dict__[k] = v
        for k, v in mods["sv_caller"].__dict__.items():
            if not k.startswith("_") and k not in mods["sv_processor"].__dict__:
                mods["sv_processor"].__dict__[k] = v
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
    _loaded.update(mods)
    return _loaded
