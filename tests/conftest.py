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
def golden_params():
    from tests.common import load_golden
    return load_golden("nsr_params.npz")


@pytest.fixture(scope="session")
def oracle_field(oracle, golden_params):
    from tests.common import oracle_field_from_golden
    return oracle_field_from_golden(golden_params)
