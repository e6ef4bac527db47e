import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built library yet: compile the HIP extension (hipcc cross-compiles
    # without a GPU).  This builds the product, it is not a fallback: without libgpsiq.so nothing runs.
    lib = os.path.join(ROOT, "multi-sdr-gps-sim_amd", "gpsiq", "libgpsiq.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "multi-sdr-gps-sim_amd", "csrc")], check=True)


@pytest.fixture(scope="session")
def oracle():
    import _oracle
    return _oracle.load_oracle()


@pytest.fixture(scope="session")
def ref():
    """oracle/_ref/libgpsref.so: the reference's own hot-path lines (built here from
    /root/reference; prebuilt on the GPU box).  Tests needing it skip when absent."""
    import _oracle
    r = _oracle.load_ref()
    if r is None:
        pytest.skip("oracle/_ref/libgpsref.so not built (no /root/reference here)")
    return r
