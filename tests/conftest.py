import importlib
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
    if os.environ.get("FAVIT_POISON") and torch.cuda.is_available():
        # one deterministic pass of the suite on poisoned LDS / slab workspace / fresh allocations (tools/poison.py)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        importlib.import_module("poison").install()


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def favit():
    """The product package (its directory name has a hyphen -> importlib)."""
    return importlib.import_module("focused-attention-vit_amd")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def case(npz, prefix):
    """All arrays under 'prefix/' as a dict with the prefix stripped."""
    pre = prefix + "/"
    return {k[len(pre):]: npz[k] for k in npz.files if k.startswith(pre)}


def sd_of(c):
    return {k[3:]: torch.from_numpy(v) for k, v in c.items() if k.startswith("sd/")}


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().cpu().to(torch.float64).flatten()
    b = torch.as_tensor(b).detach().cpu().to(torch.float64).flatten()
    den = b.norm().item()
    return (a - b).norm().item() / (den if den > 0 else 1.0)
