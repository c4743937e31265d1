import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import numpy as np
    import torch
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a.copy()) if a.ndim > 0 else a.item()
    return out


def split_golden(g):
    """-> (params, grads, rest) with the 'p.' / 'g.' prefixes stripped."""
    params = {k[2:]: v for k, v in g.items() if k.startswith("p.")}
    grads = {k[2:]: v for k, v in g.items() if k.startswith("g.")}
    rest = {k: v for k, v in g.items() if not (k.startswith("p.") or k.startswith("g."))}
    return params, grads, rest
