import importlib.util
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mmt-psm_amd")
GOLD = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_synth():
    spec = importlib.util.spec_from_file_location("mmtpsm_synthetic", os.path.join(PKG, "synthetic.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="session")
def synth():
    return load_synth()


def gold(name):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def state_shapes():
    with open(os.path.join(GOLD, "state_shapes.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def weights(synth, state_shapes):
    return synth.make_weights(state_shapes["shapes"], seed=0)


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(autouse=True)
def _restore_conv_arithmetic(request):
    """GPU tests switch the convolution arithmetic (`_hip.set_f16x2`, `set_conv_precision`): the next test starts from
    the process default again (mode 3 on the two-term fp16 split)"""
    yield
    if request.node.get_closest_marker("gpu") is not None and torch.cuda.is_available():
        from maskrcnn_benchmark import _hip as H
        H.set_f16x2(None)
