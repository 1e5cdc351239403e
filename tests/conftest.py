import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Several "ranks" inside ONE process (tests/test_gpu_ep_peer.py) wait for each other on the GPU: their compute streams must
# sit on DIFFERENT hardware queues, or a waiting kernel can sit in front of the kernel it waits for.  The HIP runtime shares
# its hardware queues between streams beyond GPU_MAX_HW_QUEUES (default 4) per priority; read once, at device init.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """Build libmoeinf_hip.so if it is missing or stale (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g

    g.build()
