import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


def pytest_sessionfinish(session, exitstatus):
    """Element-wise gradient-parity statistics gathered by tests.test_gpu_parity.check_backward (max, p99.99, count
    over tolerance per tensor and case): written next to the other GPU artefacts so a round's numbers can be
    committed under profiles/."""
    try:
        from tests.test_gpu_parity import PARITY_STATS
    except Exception:
        return
    if not PARITY_STATS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_stats.json"), "w") as fh:
        json.dump(PARITY_STATS, fh, indent=1, sort_keys=True)
