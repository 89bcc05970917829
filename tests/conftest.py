import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # The library picks k_trim_ends_batched (64 reads per wave) for batches of >= 65 536 reads and the wave-per-read kernel
    # below that.  The test batches are small: by default they go through the batched kernel -- the one the bench and any
    # large batch take -- and the tests named *_wave_per_read_* clear the hook to cover the other one.
    os.environ.setdefault("FPL_TRIM_BATCH_MIN", "1")
    # Likewise k_scan: a wave of a large batch takes its reads in chunks, and the head of a chunk's next read rides in the lanes a
    # read's last tile leaves empty (pair packing).  Small batches would get chunks of one read; the suite asks for chunks of four,
    # and the tests named *_plain_scan_* clear the hook.
    os.environ.setdefault("FPL_SCAN_CHUNK", "4")


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
