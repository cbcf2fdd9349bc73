import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths  # noqa: E402

tcs_paths.add_product_path()
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ops_golden():
    return dict(np.load(os.path.join(GOLDEN, "ops_small.npz")))


@pytest.fixture(scope="session")
def e2e_golden():
    return dict(np.load(os.path.join(GOLDEN, "e2e.npz")))


@pytest.fixture(scope="session")
def key_shapes():
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def synth_weights(key_shapes):
    from tcs_mi355.weights import synth_state_dict
    return synth_state_dict(key_shapes["shared_backbone"])


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tcs_oracle
    return tcs_oracle


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def epe(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.from_numpy(np.asarray(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.from_numpy(np.asarray(b)).double()
    return float((a - b).abs().mean())


def maxdiff(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.from_numpy(np.asarray(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.from_numpy(np.asarray(b)).double()
    return float((a - b).abs().max())
