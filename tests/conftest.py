import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def golden(name):
    path = os.path.join(GOLDEN, name)
    if name.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:         # materialise once: NpzFile re-inflates a member on every access
            return {k: z[k] for k in z.files}
    import json
    with open(path) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.lib()
    return orc
