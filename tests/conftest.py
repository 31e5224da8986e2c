import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def scenes():
    return importlib.import_module("slam-tricks_amd.scenes")


@pytest.fixture(scope="session")
def O():
    import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def known():
    import json
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        return json.load(f)
