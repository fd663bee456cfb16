import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the compiled reference in oracle/_ref")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def gold():
    from golden_util import golden
    return golden()


@pytest.fixture(scope="session")
def gold_build():
    """Construction-side golden vectors (tests/golden/make_golden_build.py)."""
    import numpy as np
    from golden_util import GOLDEN_DIR
    return np.load(os.path.join(GOLDEN_DIR, "golden_build.npz"))
