"""The C++ CellSearch front end (host/CellSearch.cpp): option handling on CPU, end-to-end on GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden, iq_u8_to_capbuf, load_pkg

EXE = os.path.join(ROOT, "host", "CellSearch")


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])


def _run(args, **kw):
    return subprocess.run([EXE] + args, capture_output=True, text=True, timeout=600, **kw)


def test_help_lists_reference_options():
    out = _run(["-h"]).stdout
    for opt in ("-s --freq-start", "-e --freq-end", "-p --ppm", "-c --correction", "-r --record", "-l --load",
                "-d --data-dir", "-i --device-index", "-v --verbose", "-b --brief"):
        assert opt in out


def test_argument_errors_match_reference_messages():
    assert "must specify a start frequency" in _run([]).stderr
    assert "cannot read and write captured data at the same time" in _run(["-s", "739e6", "-r", "-l"]).stderr
    assert "end frequency must be >= start frequency" in _run(["-s", "739e6", "-e", "700e6", "-l"]).stderr
    r = _run(["-s", "739049999", "-l", "-d", "/nonexistent"])
    assert "start frequency has been rounded to the nearest multiple of 100kHz" in r.stdout
    assert "use --load" in _run(["-s", "739e6"]).stderr


def test_no_gpu_is_a_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = _run(["-s", "739e6", "-l", "-d", "/nonexistent"])
    assert r.returncode != 0 and "MI355X is required" in r.stderr


@pytest.mark.gpu
def test_fulltest_known_answer(tmp_path):
    """The reference's (disabled) FullTest: `CellSearch -s 739000000 -l -d test` must match cell.ID..271."""
    pkg = load_pkg()
    g = golden("capbuf_0000")
    pkg_it = __import__("importlib").import_module("lte_cell_scanner_amd.itfile")
    pkg_it.write_it(str(tmp_path / "capbuf_0000.it"), {"capbuf": iq_u8_to_capbuf(g["iq_u8"]), "fc": g["fc"].astype(np.int32)})
    r = _run(["-s", "739000000", "-l", "-d", str(tmp_path)])
    assert r.returncode == 0, r.stderr
    assert re.search(r"cell.ID..271", r.stdout) and re.search(r"cell.ID..277", r.stdout)
    assert "Examining center frequency 739 MHz ..." in r.stdout
    assert "CID A      fc   foff RXPWR C nRB P  PR CrystalCorrectionFactor" in r.stdout
    rows = [l for l in r.stdout.splitlines() if re.match(r"^(277|271) 2    739M  35\.2k", l)]
    assert len(rows) == 2, r.stdout
    for l in rows:
        assert re.search(r" N  50 N one 1\.0000476", l), l
    # no cells on an empty carrier
    noise = np.random.default_rng(1).normal(0, 0.1, 153600) + 1j * np.random.default_rng(2).normal(0, 0.1, 153600)
    pkg_it.write_it(str(tmp_path / "capbuf_0000.it"), {"capbuf": noise, "fc": np.array([800000000], np.int32)})
    r = _run(["-s", "800000000", "-l", "-d", str(tmp_path), "-b"])
    assert r.returncode == 0 and "No LTE cells were found..." in r.stdout
