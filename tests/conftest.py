import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_literals.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    """The oracle (CPU restatement). Test infrastructure only."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "libpa_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    import pa_oracle
    return pa_oracle
