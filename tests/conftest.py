import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def load_pkg():
    """Import the product package (its directory name contains '-', so load it by path)."""
    name = "lte_cell_scanner_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "lte-cell-scanner_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def iq_u8_to_capbuf(iq):
    """(u8-127)/128 exactly as the reference converts dongle bytes (src/capbuf.cpp:172-181)."""
    iq = np.asarray(iq, np.uint8).astype(np.float64)
    return ((iq[0::2] - 127.0) / 128.0) + 1j * ((iq[1::2] - 127.0) / 128.0)


def f_search_set_for(fc, ppm):
    """src/CellSearch.cpp:463-464"""
    n_extra = int(np.floor((fc * ppm / 1e6 + 2.5e3) / 5e3))
    return np.arange(-n_extra, n_extra + 1) * 5000.0


@pytest.fixture(scope="session")
def capbuf_0000():
    g = golden("capbuf_0000")
    return iq_u8_to_capbuf(g["iq_u8"]), float(g["fc"][0])
