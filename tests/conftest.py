import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def ref(orc):
    if not orc.have_ref():
        pytest.skip("oracle/_ref/ref_harness not built (needs /root/reference at build time)")
    return orc.RefHarness()
