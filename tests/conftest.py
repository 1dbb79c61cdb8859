import os
import sys
import json
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))      # tests may use the oracle (checker only)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def param_specs():
    with open(os.path.join(GOLDEN, "param_specs.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def specs():
    return param_specs()
