import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_extended: the long campaigns (fuzzers under every switch, duplicate full-size runs): skipped unless "
                                       "PA_TEST_EXTENDED=1 (tools/verify_on_gpu.sh sets it) -- the driver's `pytest -m gpu` has a time budget")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("PA_TEST_EXTENDED") == "1":
        return
    skip = pytest.mark.skip(reason="extended campaign: PA_TEST_EXTENDED=1 runs it (tools/verify_on_gpu.sh)")
    for item in items:
        if "gpu_extended" in item.keywords:
            item.add_marker(skip)


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the library the way
    __graft_entry__.build() does before the tests import it.  (Building is not a fallback: the product still
    refuses to run without libpa_hip.so.)"""
    import subprocess
    so = os.path.join(ROOT, "partitionedarrays.jl_amd", "libpa_hip.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "partitionedarrays.jl_amd", "csrc")])


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
