import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def pytest_sessionfinish(session, exitstatus):
    """DD_SAVE_TUNED=<path>: after the session, write the tile / split-K table INCLUDING every shape this session tuned at
    run time (merged with the file's other entries) — how dualdiff_amd/tuned/gfx950.json learns the shapes of the other
    resolutions and of the capacity-layout contexts, so that later sessions tune nothing."""
    path = os.environ.get("DD_SAVE_TUNED")
    if path:
        from dualdiff_amd import ops
        if ops.tuned_table():
            ops.save_tuned(path)
