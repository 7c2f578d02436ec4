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


def test_streaming_mode_on_fading_channels():
    """BASELINE configs[4] (the searcher thread's loop, src/searcher_thread.cpp:83-246: one buffer at a time, ONE frequency hypothesis,
    the hipGraph-captured chain, two buffers in flight) on the `channels` population's scenes: six scenes x four noise realisations,
    pushed two at a time, every collected cell against the oracle's chain with the same single hypothesis; then the same buffers with
    the decoded identities handed over as already tracked (:157-177): no cell comes back, each is counted as seen again."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    import parity_population as P
    import __graft_entry__ as ge
    pkg = ge.load_package()
    O.set_legacy(False)
    O.set_threads(min(16, os.cpu_count() or 1))
    INT = ("ind", "n_id_2", "n_id_1", "cp_type", "n_ports", "n_rb_dl", "phich_duration", "phich_resource", "sfn")
    n_cells = n_dup = n_multi = 0
    with pkg.Searcher(0) as S:
        for s in (0, 2, 3, 5, 7, 11):
            sc = P.channel_scene(pkg.synth, s)
            f_hyp = float(5e3 * np.round(sc["planted"][0]["f_off"] / 5e3))                 # the tracked frequency offset, on the searcher's raster
            bufs = []
            for v in (1, 3, 4, 5):
                rng = np.random.default_rng(95_000 + 8 * s + v)
                sig = np.roll(sc["sig"], int(rng.integers(0, P.N_CAP)))
                bufs.append(pkg.synth.add_noise_and_quantise(rng, sig, sc["ref_pow"], P.CH_SNRS[v], rms=float(rng.uniform(0.08, 0.22)), front_end=sc["front_end"]))
            exp = []
            for b in bufs:
                x = b.astype(np.float64)
                cap = ((x[0::2] - 127.0) / 128.0) + 1j * ((x[1::2] - 127.0) / 128.0)
                exp.append(O.search_capbuf(cap, np.array([f_hyp]), sc["fc_req"], sc["fc_prog"], sc["fs_prog"])[0])
            S.stream_open(pkg.FMT_IQ_U8, P.N_CAP, sc["fc_req"], sc["fc_prog"], sc["fs_prog"])
            got = []
            for k in (0, 2):                                                                 # two buffers in flight
                S.stream_push(bufs[k], f_hyp)
                S.stream_push(bufs[k + 1], f_hyp)
                got.append(S.stream_collect())
                got.append(S.stream_collect())
            for k, ((cells, dup, _), e_all) in enumerate(zip(got, exp)):
                tag = f"scene {s} buffer {k}"
                # the reference appends a decoded cell to the tracked list at once (:233-236): later peaks of the same identity in the same
                # buffer (a second path of the fading channel) are "already being tracked" -- one record per identity, the first decoded
                e = [c for q, c in enumerate(e_all) if c.n_id_cell() not in [x.n_id_cell() for x in e_all[:q]]]
                assert [tuple(getattr(c, f) for f in INT) for c in cells] == [tuple(getattr(c, f) for f in INT) for c in e], tag
                assert dup >= len(e_all) - len(e), tag
                n_multi += len(e_all) - len(e)
                for a, b in zip(cells, e):
                    assert abs(a.frame_start - b.frame_start) < 1e-6 and abs(a.freq_superfine - b.freq_superfine) < 1e-3 and abs(a.pss_pow - b.pss_pow) <= 1e-5 * b.pss_pow, tag
                n_cells += len(e)
            # the decoded identities as tracked cells: skipped (not decoded again), counted
            for k in (0, 1):
                ids = [c.n_id_cell() for c in exp[k]]
                S.stream_push(bufs[k], f_hyp, tracked=ids)
                cells, dup, _ = S.stream_collect()
                assert cells == [] or all(c.n_id_cell() not in ids for c in cells), (s, k)
                assert dup >= len(set(ids)), (s, k, dup, ids)
                n_dup += dup
            S.stream_close()
    assert n_cells >= 12 and n_dup >= 6
