"""Population parity (hard part 1): the GPU chain and the oracle on the same buffers, every decision
compared, the disagreement count asserted to be ZERO.  The test runs a 160-buffer cut of tools/parity_population.py's
population (64 synthetic buffers: 8 scenes x 8 noise realisations at -12 .. +10 dB; 64 of the buffers bench.py times; 32 of its
dense band) in a process of its own -- the oracle's worker processes are forked before that process touches the GPU runtime.
The full population (1152 buffers) is run by hand through the same tool: profiles/r05/parity_population.json."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_population_cut_has_no_disagreement(tmp_path):
    out = tmp_path / "population.json"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_population.py"), "--limit", "64", "--dense-limit", "32", "--out", str(out)],
                       env=env, capture_output=True, text=True, timeout=1800)
    assert out.exists(), (r.stdout[-2000:], r.stderr[-3000:])
    j = json.loads(out.read_text())
    t = j["totals"]
    assert t["buffers"] == 64 + 4 * 16 + 32 and t["cells"] >= 40 and t["peaks"] > t["cells"]
    assert j["per_group"]["synthetic"]["cells"] >= 8 and j["per_group"]["dense"]["cells"] >= 32
    assert j["disagreements"] == 0, j["details"][:10]
    assert r.returncode == 0
    # every index of xc_incoherent_collapsed_frq was compared, near-ties included, and the library repaired some of them
    assert t["frq_positions"] == t["buffers"] * 3 * 9600 and t["frq_near_ties_4e_6"] > 0 and j["gpu_frq_positions_repaired"] > 0


def test_population_on_fading_multipath_channels_has_no_disagreement(tmp_path):
    """Round 6: a 64-buffer cut (8 scenes x 8 noise realisations) of the `channels` group -- every cell through an independent
    Rayleigh tapped delay line per antenna port (EPA / EVA / ETU, Doppler 5 / 70 / 300 Hz), DC spike, I/Q imbalance, clipped
    ADC.  Everything the tool compares for the other groups, plus the ARRAYS of every decoded cell through the stage entry
    points: extract_tfg 1e-10, tfoec's tfg_comp 1e-9, chan_est's ce_tfg of every port 1e-9 and its noise power 1e-11.  The
    whole group (512 buffers): profiles/r06/parity_population_channels.json."""
    out = tmp_path / "population_channels.json"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_population.py"), "--groups", "channels", "--limit", "64", "--out", str(out)],
                       env=env, capture_output=True, text=True, timeout=1800)
    assert out.exists(), (r.stdout[-2000:], r.stderr[-3000:])
    j = json.loads(out.read_text())
    t = j["totals"]
    assert t["buffers"] == 64 and t["cells"] >= 30 and t["peaks"] > 2 * t["cells"]      # fading: many PSS peaks, some cells lost at low SNR
    assert j["disagreements"] == 0, j["details"][:10]
    sa = j["stage_arrays"]
    assert sa["cells_compared"] == t["cells"] and sa["disagreements"] == 0
    w = sa["worst_relative_deviation"]
    assert w["tfg"] <= 1e-10 and w["tfg_comp"] <= 1e-9 and w["ce_tfg"] <= 1e-9 and w["np"] <= 1e-11
    assert r.returncode == 0


def test_population_on_wide_grids_has_no_disagreement(tmp_path):
    """Round 6: a 32-buffer cut (4 scenes x 8 noise realisations: n_f = 61, 87, 103, 125 -- 1.25 / 1.8 / 2.14 / 2.6 GHz at the CLI's
    default 120 ppm, src/CellSearch.cpp:463-465) of the `highband` group: cells through fading channels with LO errors out to 0.9
    of the grid's edge, every decision and every collapsed index compared, the stage arrays of every decoded cell too.  The
    whole group (96 buffers, also n_f = 169 and 141 with dongle parameters): profiles/r06/parity_population_highband.json."""
    out = tmp_path / "population_highband.json"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_population.py"), "--groups", "highband", "--limit", "32", "--out", str(out)],
                       env=env, capture_output=True, text=True, timeout=1800)
    assert out.exists(), (r.stdout[-2000:], r.stderr[-3000:])
    j = json.loads(out.read_text())
    t = j["totals"]
    assert t["buffers"] == 32 and t["cells"] >= 12 and t["frq_positions"] == 32 * 3 * 9600
    assert j["disagreements"] == 0, j["details"][:10]
    sa = j["stage_arrays"]
    assert sa["cells_compared"] == t["cells"] and sa["disagreements"] == 0
    assert r.returncode == 0
