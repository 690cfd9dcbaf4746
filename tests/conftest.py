import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of CPU-oracle time on the GPU box (the headline workload, the reference's default schedule)")


@pytest.fixture(scope="session")
def mano_arrays():
    from ihmr_amd.assets import synthetic_mano
    return synthetic_mano(True), synthetic_mano(False)
