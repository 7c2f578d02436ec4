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
