import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """tests/golden/<name>.npz as {key: torch tensor} (0-d arrays become python scalars/strings)."""
    out = {}
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        for k in z.files:
            v = z[k]
            if v.dtype.kind in "US":
                out[k] = str(v)
            elif v.ndim == 0 and v.dtype.kind in "iu" and "/" not in k:
                out[k] = int(v)
            else:
                out[k] = torch.from_numpy(np.array(v))
    return out


def sub(d, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    return load_golden
