"""pytest configuration: markers, import paths, shared fixtures."""
import os
import sys

import pytest

# torch bundles its own ROCm runtime (libamdhip64 7.0) while libivjoin_hip.so links the system one
# (/opt/rocm, 7.2).  Whichever loads first serves the whole process; torch only finds its GPUs when
# its own copy came first.  Tests that mix both (device_api, bench) therefore need torch imported
# before the engine library is dlopen'ed -- do it once, up front.
try:  # pragma: no cover
    import torch  # noqa: F401
except Exception:
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "polars-bio_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Fast parity tests first, the full-size configs after them, the multi-GPU tests last (a -x run then reports
    the cheap failures before it spends minutes on 100M-row inputs)."""
    def weight(item):
        name = item.nodeid
        if "two_rank_hip" in name or "self_spawn" in name:
            return 2
        if "full_size" in name:
            return 1
        return 0
    items.sort(key=weight)          # stable: the order inside a class of tests is kept


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
