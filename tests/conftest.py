import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A fresh checkout has no built artefacts (*.so are git-ignored): build them once, exactly
    # as __graft_entry__.build() does, so that the ABI / oracle tests have something to load.
    lib = os.path.join(ROOT, "fast-match_amd", "libfastmatch_hip.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "fast-match_amd", "csrc"), "-j4"],
                              stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def ctx():
    """Process-wide HIP context; fails loudly (no fallback) when the device is missing."""
    import fastmatch_amd
    return fastmatch_amd.default_context(0)
