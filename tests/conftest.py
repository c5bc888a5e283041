import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def water_check_rhf():
    """The H2O geometry of the reference's validation/check_rhf.f90:112-116 (Bohr)."""
    return [8, 1, 1], [[0.0, 0.0, -0.1364652], [0.0, 1.4304924, 1.0826636], [0.0, -1.4304924, 1.0826636]]
