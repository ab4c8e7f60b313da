import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)              # tests/margins.py (the parity-margin recorder) is imported by name

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """Write the parity margins the floating-point asserts recorded (tests/margins.py) where gpurun brings them home."""
    import json
    import margins
    rows = margins.rows()
    if not rows:
        return
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_margins.json"), "w") as f:
            json.dump(dict(exitstatus=int(exitstatus), rows=rows), f, indent=1)
    except OSError:
        pass
