import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def mlp_blob():
    import numpy as np
    return np.fromfile(os.path.join(ROOT, "ndp_nmpc_qd_amd", "weights", "downwash_sn4.bin"), dtype="<f4")


@pytest.fixture(scope="session")
def mlp_golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "mlp_golden.npz"))
