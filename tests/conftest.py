import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "generative-audio_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_PARITY = {}


@pytest.fixture
def record_err(request):
    """tests call record_err(tag, measured, limit): asserts measured < limit and keeps the worst measured value per tag; the
    session writes them to gpurun_out/parity_errors.json (copied to profiles/ so tolerances can be audited)."""
    def rec(tag, measured, limit):
        key = f"{request.node.name}::{tag}"
        _PARITY[key] = {"measured": float(measured), "limit": float(limit)}
        assert measured < limit, (key, measured, limit)
    return rec


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    import json
    out = os.path.join(REPO, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_errors.json")
        old = {}
        if os.path.exists(path):
            try:
                old = json.load(open(path))
            except Exception:
                old = {}
        old.update(_PARITY)
        json.dump(old, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
