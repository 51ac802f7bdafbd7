import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def split_rows(counts, rows):
    """(counts, concatenated rows) -> List[Optional[ndarray]] as the reference returns it."""
    out, o = [], 0
    for k in counts.tolist():
        out.append(None if k == 0 else rows[o:o + k])
        o += k
    return out


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
