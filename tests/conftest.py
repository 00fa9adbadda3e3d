import os
import sys

import pytest

try:
    # torch first: it brings its own HIP runtime, which must be the one the process initialises when a
    # test mixes torch device tensors with the library (tests of the *_dev entry points, bench.py order)
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    import libecc_amd
    # build the fixed-base comb tables already for the small batches the tests use (default: 4096 items)
    os.environ.setdefault("ECAMD_COMB_MIN_BATCH", "48")
    ctx = libecc_amd.Context(0)  # raises if no device / library: no silent fallback
    yield ctx
    ctx.close()
