import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def _gpu_present():
    """A device node the HIP runtime can open (no HIP call here: counting devices must not initialise the GPU)."""
    return os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped, not failed, on a host without a GPU (plain `pytest tests` in the build container);
    `-m gpu` on the GPU box runs them all -- there the library itself fails loudly if the device is missing."""
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU device node (/dev/kfd): run with -m gpu on an MI355X")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_bg():
    """the CPU oracle's results for the full-size GPU tests, computed by a spawned pool in the background (tests/oracle_worker.py)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_worker import OracleBackground
    bg = OracleBackground()
    yield bg
    bg.close()


@pytest.fixture(scope="session", autouse=True)
def _oracle_bg_autostart(request):
    """the pool starts with the session when a selected test will ask for it, so that its work overlaps the GPU tests before it"""
    if _gpu_present() and any("oracle_bg" in getattr(it, "fixturenames", ()) for it in request.session.items):
        request.getfixturevalue("oracle_bg")
