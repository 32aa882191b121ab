import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the fp64 / fp32 CPU references of the tests run on torch's intra-op threads: sized by the container's CPU quota, not by the
    # node's visible cores (declip_amd/hostinfo.py: a 128-thread OpenMP team on a 16-core quota gets the whole process throttled)
    try:
        from declip_amd import hostinfo
        hostinfo.limit_host_threads()
    except Exception:       # (torch missing / import problems surface in the tests themselves)
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
