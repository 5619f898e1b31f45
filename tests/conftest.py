import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield


def load_golden(name):
    z = np.load(GOLDEN / f"{name}.npz")
    return {k: torch.from_numpy(np.asarray(z[k])) if z[k].ndim else z[k].item() for k in z.files}


@pytest.fixture(scope="session")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def built_lib():
    """The in-tree HIP library (hipcc cross-compiles without a GPU); built on demand."""
    from nerfmatch_amd.build import build
    return build()
