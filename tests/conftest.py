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


@pytest.fixture(autouse=True)
def _quiesce_gpu_between_tests(request):
    """GPU tests build policies whose HIP graphs are destroyed when the garbage collector gets to them: do that here, with the
    device idle, rather than at an arbitrary allocation inside the next test while its graphs replay."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import gc
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.synchronize()
