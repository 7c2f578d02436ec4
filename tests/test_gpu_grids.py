"""Every frequency grid the reference's CLI can build, not only the 700 MHz ones.

`CellSearch` sizes the grid from the carrier (src/CellSearch.cpp:463-465): n_f = 2 floor((fc ppm / 1e6 + 2500) / 5000) + 1,
and `xc_correlate` loops over whatever it is handed (src/searcher.cpp:113-174).  At the CLI's default 120 ppm that is 61
hypotheses at 1.25 GHz, 87 at 1.8 GHz (band 3), 103 at 2.14 GHz (band 1), 125 at 2.6 GHz (band 7), 169 at 3.5 GHz (bands
42/43) and 289 at 6 GHz -- 8 to 55 template groups per buffer instead of the 6-7 of the 700 MHz band, a different
`k_factor` spread of the window starts (it shrinks with fc), 2-4 GB of xc_incoherent_single per 128-buffer batch.  Rounds
1-5 verified nothing above n_f = 37 on hardware and refused n_f > 128.  Here: EVERY element of xc_incoherent_single /
collapsed pow / frq / sp_incoherent / Z_th1 against the oracle on the int8 (u8 I/Q) and fp16 x 3 (complex<float>) kernels,
the host entry point (complex<double>, incoherent included), the whole chain on planted cells, the CLI on a synthetic `.it`."""
import os
import re
import subprocess

import numpy as np
import pytest

import oracle as O
from conftest import ROOT, iq_u8_to_capbuf, f_search_set_for, load_pkg
from test_gpu_pss import _batch_arrays_vs_oracle, _check_xcorr

pytestmark = pytest.mark.gpu
FS = 1.92e6
GRIDS = [(1.25e9, 61), (1.8e9, 87), (2.14e9, 103), (2.6e9, 125), (3.5e9, 169)]
KEY = lambda c: (c.n_id_cell(), c.ind, c.freq, c.n_ports, c.n_rb_dl, c.phich_duration, c.phich_resource, c.sfn, c.cp_type)


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


@pytest.fixture(scope="module")
def S(pkg):
    s = pkg.Searcher(0)
    yield s
    s.close()


@pytest.fixture(scope="module", autouse=True)
def _threads():
    O.set_legacy(False)
    O.set_threads(min(16, os.cpu_count() or 1))


def _planted(pkg, fc, n_f, seed):
    """Two buffers for carrier fc: cells near BOTH edges of the grid (the hypotheses the 700 MHz tests never reach) with the
    sample-clock stretch that goes with such an offset, and a weaker mid-grid cell; the second buffer is recorded with a
    dongle whose programmed frequency differs from the requested one."""
    edge = 5e3 * (n_f // 2) - 3.1e3
    b0, _ = pkg.synth.make_capbuf(seed, fc, [dict(n_id_1=(7 * seed) % 168, n_id_2=seed % 3, f_off=edge, n_ports=2),
                                              dict(n_id_1=(11 * seed + 5) % 168, n_id_2=(seed + 1) % 3, f_off=-edge + 1.2e3, gain_db=-4, cp_normal=False, n_ports=4)], 6.0)
    b1, _ = pkg.synth.make_capbuf(seed + 1, fc + 200e3, [dict(n_id_1=(13 * seed + 2) % 168, n_id_2=(seed + 2) % 3, f_off=0.37 * edge, n_ports=1, n_rb_dl=100)], 3.0)
    return [b0, b1], np.array([fc, fc + 200e3])


@pytest.mark.parametrize("fc,n_f", GRIDS)
def test_every_array_element_at_the_cli_grid_of_a_high_band(S, pkg, fc, n_f):
    f = f_search_set_for(fc, 120)
    assert f.size == n_f
    bufs, fcs = _planted(pkg, fc, n_f, int(fc / 1e7))
    _batch_arrays_vs_oracle(S, pkg, bufs, f, fcs, 153600, f"fc {fc / 1e6:.0f} MHz, n_f = {n_f}")
    listed, left = S.last_frq_repair_stats()
    assert left == 0, (listed, left)


@pytest.mark.parametrize("fc,n_f", GRIDS)
def test_full_chain_on_planted_cells_at_the_cli_grid_of_a_high_band(S, pkg, fc, n_f):
    import torch
    f = f_search_set_for(fc, 120)
    bufs, fcs = _planted(pkg, fc, n_f, int(fc / 1e7))
    d = torch.from_numpy(np.stack(bufs)).cuda()
    res = S.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, len(bufs), 153600, f, fcs, fcs, FS, pkg.STAGE_FULL)
    n_dec = 0
    for b, iq in enumerate(bufs):
        exp, _ = O.search_capbuf(iq_u8_to_capbuf(iq), f, fcs[b], fcs[b], FS)
        assert [KEY(c) for c in res[b]] == [KEY(c) for c in exp], (fc, b)
        for a, e in zip(res[b], exp):
            assert abs(a.pss_pow - e.pss_pow) <= 1e-5 * e.pss_pow and abs(a.freq_superfine - e.freq_superfine) < 1e-3
            assert abs(a.frame_start - e.frame_start) < 1e-6
        n_dec += len(exp)
    assert n_dec >= 2      # the planted cells decode: the chain was exercised, not only an empty list compared
    # the edge cells were found at the edge hypotheses
    edge = 5e3 * (n_f // 2)
    assert any(abs(c.freq) >= edge - 5e3 for c in res[0]), [c.freq for c in res[0]]


@pytest.mark.parametrize("fc,n_f", [(2.6e9, 125), (3.5e9, 169)])
def test_host_entry_point_at_a_high_band_grid(S, pkg, fc, n_f):
    """lcs_xcorr_pss (the reference's call shape: complex<double> in, every output array out, xc_incoherent included) and
    lcs_search_capbuf; once as dongle data (int8 kernel chosen on the device) and once scaled (fp32 kernel)."""
    f = f_search_set_for(fc, 120)
    bufs, fcs = _planted(pkg, fc, n_f, int(fc / 1e7))
    cap = iq_u8_to_capbuf(bufs[0])
    ro = O.xcorr_pss(cap, f, 2, fc, fc, FS)
    r = S.xcorr_pss(cap, f, 2, fc, fc, FS)
    assert S.last_xcorr_info()[0] == "k_xcorr_i8x3"
    _check_xcorr(r, ro, f"host int8 n_f={n_f}")
    r32 = S.xcorr_pss(0.5 * cap, f, 2, fc, fc, FS, want_incoherent=False)
    assert S.last_xcorr_info()[0].startswith("k_xcorr_mfma_blk")
    assert np.array_equal(r32["frq"], ro["frq"])
    assert (np.abs(4.0 * r32["single"].astype(np.float64) - ro["single"]) / ro["single"]).max() < 1e-5
    cells, peaks = S.search_capbuf(cap, f, fc, fc, FS)
    exp, exp_pk = O.search_capbuf(cap, f, fc, fc, FS)
    assert [KEY(c) for c in cells] == [KEY(c) for c in exp]
    assert [(p.n_id_2, p.ind, p.freq) for p in peaks] == [(p.n_id_2, p.ind, p.freq) for p in exp_pk]


def test_grid_of_a_6_ghz_carrier_289_hypotheses(S, pkg):
    """The top of what a tuner reaches: 6 GHz at 120 ppm -> n_f = 289, 55 template groups, 34 MB of xc_incoherent_single per
    buffer (rounds 1-5: LCS_ERR_BAD_ARG).  One buffer, both batch kernels, every element; then the chain."""
    import torch
    fc = 6.0e9
    f = f_search_set_for(fc, 120)
    assert f.size == 289
    b0, _ = pkg.synth.make_capbuf(600, fc, [dict(n_id_1=33, n_id_2=1, f_off=-712e3), dict(n_id_1=140, n_id_2=0, f_off=655.5e3, gain_db=-3)], 5.0)
    _batch_arrays_vs_oracle(S, pkg, [b0], f, np.array([fc]), 153600, "fc 6 GHz, n_f = 289")
    d = torch.from_numpy(b0[None]).cuda()
    res = S.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, 1, 153600, f, np.array([fc]), np.array([fc]), FS, pkg.STAGE_FULL)[0]
    exp, _ = O.search_capbuf(iq_u8_to_capbuf(b0), f, fc, fc, FS)
    assert [KEY(c) for c in res] == [KEY(c) for c in exp] and sorted(c.n_id_cell() for c in res) == [100, 420]


def test_no_grid_below_the_sanity_bound_is_refused(S, pkg):
    """Any n_f up to 1024 is taken (the reference has no bound at all; ours is a sanity bound on memory); beyond it the call
    fails with a message, it does not truncate."""
    rng = np.random.default_rng(3)
    iq = np.clip(np.rint(rng.normal(127.0, 16.0, 2 * 153600)), 0, 255).astype(np.uint8)
    cap = iq_u8_to_capbuf(iq)
    f = np.arange(-256, 257) * 2500.0          # 513 hypotheses on a 2.5 kHz raster
    r = S.xcorr_pss(cap, f, 2, 2.0e9, 2.0e9, FS, want_incoherent=False)
    assert r["single"].shape == (3, 9600, 513)
    sub = np.array([0, 1, 255, 256, 257, 511, 512])
    ro = O.xcorr_pss(cap, f[sub], 2, 2.0e9, 2.0e9, FS)
    assert (np.abs(r["single"][:, :, sub].astype(np.float64) - ro["single"]) / ro["single"]).max() < 1e-5
    with pytest.raises(Exception, match="n_f out of range"):
        S.xcorr_pss(cap, np.arange(1025) * 100.0, 2, 2.0e9, 2.0e9, FS, want_incoherent=False)


def test_duplicated_hypotheses_do_not_blow_up_the_repair(S, pkg):
    """A degenerate f_search_set (the same hypothesis several times) makes every position an exact tie of the arg-max: the
    repair's work is bounded (lcs_last_frq_repair_stats), the first copy wins everywhere as in the reference (strict >,
    src/searcher.cpp:374), and the call returns in ordinary time."""
    import time
    import torch
    b0, _ = pkg.synth.make_capbuf(77, 739e6, [dict(n_id_1=9, n_id_2=2, f_off=11e3)], 8.0)
    f = np.array([10e3, 10e3, 15e3, 10e3, 15e3])
    d = torch.from_numpy(np.stack([b0] * 8)).cuda()
    fcs = np.full(8, 739e6)
    t0 = time.time()
    S.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, 8, 153600, f, fcs, fcs, FS, pkg.STAGE_PSS)
    dt = time.time() - t0
    listed, left = S.last_frq_repair_stats()
    assert listed >= 8 * 3 * 9600 * 0.99 and left > 0 and dt < 5.0, (listed, left, dt)
    ro = O.xcorr_pss(iq_u8_to_capbuf(b0), f, 2, 739e6, 739e6, FS)
    assert set(np.unique(ro["frq"])) <= {0, 2}
    zth = O.z_th1(ro["sp_incoherent"], ro["n_comb_xc"])
    for b in (0, 7):
        r = S.batch_readback(b, f.size)
        bad = np.argwhere(r["frq"] != ro["frq"])
        # what the bound leaves unrepaired are near-ties BETWEEN the two distinct hypotheses that no peak can come from
        assert len(bad) <= 2 and all(ro["pow"][t, i] < zth[i] for t, i in bad), bad
        assert set(np.unique(r["frq"])) <= {0, 2}


def test_cli_on_a_band_3_carrier(tmp_path, pkg):
    """`CellSearch -s 1.8e9 -l` (n_f = 87) on a synthetic recording: the report names the planted cell."""
    exe = os.path.join(ROOT, "host", "CellSearch")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    it = __import__("importlib").import_module("lte_cell_scanner_amd.itfile")
    iq, _ = pkg.synth.make_capbuf(1800, 1.8e9, [dict(n_id_1=61, n_id_2=2, f_off=-187e3, n_ports=2, n_rb_dl=75)], 8.0)
    it.write_it(str(tmp_path / "capbuf_0000.it"), {"capbuf": iq_u8_to_capbuf(iq), "fc": np.array([1800000000], np.int32)})
    r = subprocess.run([exe, "-s", "1.8e9", "-l", "-d", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    exp, _ = O.search_capbuf(iq_u8_to_capbuf(iq), f_search_set_for(1.8e9, 120), 1.8e9, 1.8e9, FS)
    assert [c.n_id_cell() for c in exp] == [185]
    rows = r.stdout.split("CrystalCorrectionFactor\n")[1].splitlines()
    assert len(rows) == 1 and re.match(r"^185 2   1800M\s+-18\dk", rows[0]) and " N  75 " in rows[0], r.stdout


def test_hypothesis_split_of_a_169_hypothesis_grid_over_eight_shares(pkg):
    """SURVEY 8(e) latency mode on a band-42 grid: the 169 hypotheses of one buffer split over eight contexts (the shares an
    8-GPU node would take: 22 x 7 + 15), MAX of the packed words standing in for the all-reduce, near-ties settled by
    lcs_foe_contend / _resolve: every index equals the oracle's, and the eight ranks' cells together are the fused chain's."""
    import torch
    fc, n_f = 3.5e9, 169
    f = f_search_set_for(fc, 120)
    bufs, _ = _planted(pkg, fc, n_f, int(fc / 1e7))
    cap = iq_u8_to_capbuf(bufs[0])
    ro = O.xcorr_pss(cap, f, 2, fc, fc, FS)
    exp, exp_pk = O.search_capbuf(cap, f, fc, fc, FS)
    shares = [(22 * r, 22 if r < 7 else n_f - 154) for r in range(8)]
    ctxs = [pkg.Searcher(0) for _ in shares]
    try:
        words = [torch.empty(3 * 9600, dtype=torch.int64, device="cuda") for _ in ctxs]
        meta = [torch.empty(9601, dtype=torch.float64, device="cuda") for _ in ctxs]
        for S_, (a, n), w, m in zip(ctxs, shares, words, meta):
            S_.foe_partial(cap, f, a, n, fc, fc, FS, w.data_ptr(), m.data_ptr())
        red = torch.stack(words).max(dim=0).values
        w2 = [torch.empty(3 * 9600, dtype=torch.int64, device="cuda") for _ in ctxs]
        for S_, x in zip(ctxs, w2):
            S_.foe_contend(f, red.data_ptr(), x.data_ptr())
        red2 = torch.stack(w2).max(dim=0).values
        ctxs[0].foe_resolve(red.data_ptr(), red2.data_ptr())
        pw, fq = pkg.sweep.unpack_pow_frq(red.cpu().numpy().reshape(3, 9600))
        assert np.array_equal(fq, ro["frq"]) and (np.abs(pw - ro["pow"]) / ro["pow"]).max() < 1e-5
        got = []
        for S_ in ctxs:
            cells, order, peaks = S_.foe_finish(red.data_ptr(), meta[0].data_ptr(), f)
            got += list(zip(order.tolist(), cells))
            assert [(p.n_id_2, p.freq) for p in peaks] == [(p.n_id_2, p.freq) for p in exp_pk]
        got.sort(key=lambda x: x[0])
        assert [KEY(c) for _, c in got] == [KEY(c) for c in exp] and len(exp) >= 1
    finally:
        for S_ in ctxs:
            S_.close()
