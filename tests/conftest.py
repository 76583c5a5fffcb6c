"""pytest wiring: registers the ``gpu`` marker and puts the product tree on sys.path.

``stylegan-for-facerec_amd/`` is laid out like the reference checkout (``backbone/``, ``head/``, ``loss/``,
``util/``, ``configs/`` + the native ``frhip/`` package) so the reference's import lines work unchanged.
"""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT = os.path.join(REPO, "stylegan-for-facerec_amd")
for p in (PRODUCT, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
