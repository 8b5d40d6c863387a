import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

LAYER_CASES = ["tiny", "selftest", "one_rel", "r16", "bases", "empty", "single_edge"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the library is a build product (git-ignored): make it if this checkout has none yet.  Not a
    # fallback - the tests still exercise nothing but the HIP path - just the build step.
    lib = os.path.join(ROOT, "primekg_rgcn_linkprediction_amd", "librgcn_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()


def load_golden(name):
    """-> dict of torch tensors (numpy scalars stay python ints)."""
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        out = {}
        for k in z.files:
            v = z[k]
            out[k] = v.item() if v.shape == () and v.dtype.kind in "iu" else torch.from_numpy(v.copy())
        return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
