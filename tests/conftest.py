import os
import sys

import pytest

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before torch / HIP start: gnn_hex_amd/graphs.py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hexref():
    """Build (if needed) and return the C env oracle module."""
    from oracle import env_ref
    env_ref.build()
    return env_ref
