"""No-GPU checks of the drop-in boundary: the HIP library builds for gfx950, loads, exports every
symbol include/fastplong_amd.h declares, and refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from fastplong_amd import abi, build, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_hip()
    return engine.load_library()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "fastplong_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fpl_[a-z_0-9]+)\s*\(", hdr))
    assert declared and declared == set(engine.EXPORTS), declared ^ set(engine.EXPORTS)
    raw = C.CDLL(engine.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name


def test_struct_layouts_match_header(lib):
    o = abi.FplOptions()
    lib.fpl_options_default(C.byref(o))
    assert o.as_dict() == abi.FplOptions.default().as_dict()
    assert C.sizeof(abi.FplReadResult) == 36
    assert lib.fpl_abi_version() == abi.FPL_ABI_VERSION


def test_no_cpu_fallback(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(engine.FplError, match="no usable HIP device"):
        engine.Engine()
    assert lib.fpl_strerror(abi.FPL_ERR_NO_DEVICE).decode().startswith("no usable HIP device")


def test_product_does_not_touch_the_oracle():
    """the product tree never imports, links or executes anything under oracle/"""
    pkg = os.path.join(ROOT, "fastplong_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in txt.lower() or f == "__init__.py" and "oracle" not in txt.lower(), os.path.join(dp, f)
