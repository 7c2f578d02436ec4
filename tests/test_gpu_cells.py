"""GPU parity tests for the per-cell stages (sss_detect, pss_sss_foe, extract_tfg, tfoec,
decode_mib) and the fused chain, through the C ABI, against the CPU oracle and the goldens.

Integer identities (n_id_1, cp_type, MIB fields, which peaks survive) must be exact; the
continuous quantities are fp64 on both sides and agree to ~1e-9 relative (different summation
order / libm only)."""
import numpy as np
import pytest

import oracle as O
from conftest import golden, iq_u8_to_capbuf, f_search_set_for, load_pkg

pytestmark = pytest.mark.gpu
FS = 1.92e6


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
    import os
    O.set_legacy(False)
    O.set_threads(min(16, os.cpu_count() or 1))


def _close(a, b, rtol=1e-9, atol=0.0):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() <= atol + rtol * np.abs(b).max()


def test_sss_detect_and_foe_on_golden_peaks(S, pkg):
    """All 24 input peaks of test/test_sss_detect.it: identities vs the golden file, every
    intermediate vs the oracle."""
    g = golden("test_sss_detect")
    fc = float(g["fc"][0])
    cap = g["capbuf"]
    n_found = 0
    for t in range(24):
        kw = dict(pss_pow=float(g["peaks_pow"][t]), ind=int(g["peaks_ind"][t] - 1), freq=float(g["peaks_freq"][t]),
                  n_id_2=int(g["peaks_n_id_2"][t]), fc_requested=fc, fc_programmed=fc)
        co, do = O.sss_detect(O.new_cell(**kw), cap, 3.0, fc, fc, FS)
        cg, dg = S.sss_detect(pkg.new_cell(**kw), cap, 3.0, fc, fc, FS)
        for k in ("h1_np", "h2_np", "h1_nrm", "h2_nrm", "h1_ext", "h2_ext"):
            assert _close(dg[k], do[k], 1e-9), (t, k, np.abs(dg[k] - do[k]).max())
        assert _close(dg["ll_nrm"], do["ll_nrm"], 1e-9) and _close(dg["ll_ext"], do["ll_ext"], 1e-9)
        assert (cg.n_id_1, cg.cp_type) == (co.n_id_1, co.cp_type), t
        gold = g["peaks_out_n_id_1"][t]
        if np.isfinite(gold):
            n_found += 1
            assert cg.n_id_1 == int(gold) and cg.cp_type == (2 if g["peaks_out_cp_type"][t] else 1)
            assert abs(cg.frame_start - co.frame_start) < 1e-9
            fo = O.pss_sss_foe(co, cap, fc, fc, FS)
            fg = S.pss_sss_foe(cg, cap, fc, fc, FS)
            assert abs(fg.freq_fine - fo.freq_fine) < 1e-6, (t, fg.freq_fine, fo.freq_fine)
        else:
            assert cg.n_id_1 == -1 and cg.cp_type == 0 and np.isnan(cg.frame_start)
    assert n_found == 22


def test_tfg_tfoec_mib_on_golden_cell(S, pkg):
    """Matlab/test_tfg.mat (mirrors test/test_tfg.cpp): the stored peak must decode to 50 RB."""
    g = golden("test_tfg")
    fc = float(g["fc"][0])
    kw = dict(fc_requested=fc, fc_programmed=fc, pss_pow=float(g["peak_pow"][0]), ind=int(g["peak_ind"][0]) - 1,
              freq=float(g["peak_freq"][0]), n_id_2=int(g["peak_n_id_2"][0]), n_id_1=int(g["peak_n_id_1"][0]), cp_type=1,
              frame_start=float(g["peak_frame_start"][0]) - 1, freq_fine=float(g["peak_freq_fine"][0]))
    co, cg = O.new_cell(**kw), pkg.new_cell(**kw)
    tfg_o, ts_o = O.extract_tfg(co, g["capbuf"], fc, fc, FS)
    tfg_g, ts_g = S.extract_tfg(cg, g["capbuf"], fc, fc, FS)
    assert tfg_g.shape == (854, 72)
    assert np.array_equal(ts_g, ts_o)
    assert _close(tfg_g, tfg_o, 1e-10), np.abs(tfg_g - tfg_o).max()
    c2o, tfgc_o, tsc_o = O.tfoec(co, tfg_o, ts_o, fc, fc)
    c2g, tfgc_g, tsc_g = S.tfoec(cg, tfg_o, ts_o, fc, fc)          # same input grid for both
    assert abs(c2g.freq_superfine - c2o.freq_superfine) < 1e-7
    assert _close(tsc_g, tsc_o, 1e-13)
    assert _close(tfgc_g, tfgc_o, 1e-9), np.abs(tfgc_g - tfgc_o).max()
    # chan_est + ce_interp_hex (src/searcher.cpp:1369-1477, 1223-1362) directly: all four antenna ports, the whole 854 x 72
    # estimate and the noise power (the GPU forms each triangle's plane in closed form where the reference -- and the
    # oracle -- solve the 3x3 system: same plane, rounding differs at the 1e-13 level)
    for port in range(4):
        ce_o, np_o = O.chan_est(c2o, tfgc_o, port)
        ce_g, np_g = S.chan_est(c2g, tfgc_o, port)
        assert abs(np_g - np_o) <= 1e-11 * np_o, port
        assert np.abs(ce_g - ce_o).max() <= 1e-9 * np.abs(ce_o).max(), (port, np.abs(ce_g - ce_o).max())
    c3o = O.decode_mib(c2o, tfgc_o)
    c3g = S.decode_mib(c2g, tfgc_o)
    assert c3g.n_rb_dl == 50 == int(g["expected_n_rb_dl"][0])
    for k in ("n_ports", "n_rb_dl", "phich_duration", "phich_resource", "sfn", "n_id_1", "n_id_2", "cp_type"):
        assert getattr(c3g, k) == getattr(c3o, k), k
    assert (c3g.n_ports, c3g.phich_duration, c3g.phich_resource, c3g.n_id_cell()) == (2, 1, 3, 277)


def _cells_equal(got, exp):
    assert len(got) == len(exp), ([c.n_id_cell() for c in got], [c.n_id_cell() for c in exp])
    for a, b in zip(got, exp):
        for k in ("ind", "n_id_2", "n_id_1", "cp_type", "n_ports", "n_rb_dl", "phich_duration", "phich_resource", "sfn"):
            assert getattr(a, k) == getattr(b, k), (k, a, b)
        assert a.freq == b.freq and a.fc_requested == b.fc_requested
        assert abs(a.pss_pow - b.pss_pow) < 1e-5 * b.pss_pow
        assert abs(a.frame_start - b.frame_start) < 1e-6
        assert abs(a.freq_fine - b.freq_fine) < 1e-3 and abs(a.freq_superfine - b.freq_superfine) < 1e-3


def test_full_chain_capbuf_0000(S, capbuf_0000):
    """FullTest known answer: cells 277 and 271 @ 739 MHz, 2 ports, 50 RB, PHICH normal / one."""
    cap, fc = capbuf_0000
    f = f_search_set_for(fc, 120)
    cells, peaks = S.search_capbuf(cap, f, fc, fc, FS)
    co, po = O.search_capbuf(cap, f, fc, fc, FS)
    assert [(p.n_id_2, p.ind, p.freq) for p in peaks] == [(p.n_id_2, p.ind, p.freq) for p in po]
    assert all(p.n_id_1 == -1 and np.isnan(p.frame_start) for p in peaks)
    assert [c.n_id_cell() for c in cells] == [277, 271]
    _cells_equal(cells, co)
    for c in cells:
        assert (c.n_ports, c.n_rb_dl, c.cp_type, c.phich_duration, c.phich_resource) == (2, 50, 1, 1, 3)
    assert (cells[0].sfn, cells[1].sfn) == (74, 22)


def test_full_chain_noisy_and_short_buffers(S):
    g = golden("test_sss_detect")       # -17 dB: PSS/SSS found, MIB CRC fails -> no cell reported
    fc = float(g["fc"][0])
    f = np.arange(20e3, 60e3 + 1, 5e3)
    cells, peaks = S.search_capbuf(g["capbuf"], f, fc, fc, FS)
    co, po = O.search_capbuf(g["capbuf"], f, fc, fc, FS)
    assert len(cells) == 0 == len(co) and [(p.n_id_2, p.ind) for p in peaks] == [(p.n_id_2, p.ind) for p in po]
    g = golden("test_xcorr_pss")        # 135360 samples
    cap = iq_u8_to_capbuf(g["iq_u8"])
    cells, _ = S.search_capbuf(cap, g["f_search_set"], 739e6, 739e6, FS)
    co, _ = O.search_capbuf(cap, g["f_search_set"], 739e6, 739e6, FS)
    _cells_equal(cells, co)
    assert [c.n_id_cell() for c in cells][:1] == [277]


def test_full_chain_batch_device(S, pkg, capbuf_0000):
    import torch
    cap, fc = capbuf_0000
    g = golden("capbuf_0000")
    f = f_search_set_for(fc, 100)
    rng = np.random.default_rng(5)
    noise = np.clip(np.rint(rng.normal(127.0, 10.0, g["iq_u8"].size)), 0, 255).astype(np.uint8)
    bufs = np.stack([noise, g["iq_u8"], np.roll(g["iq_u8"], 2 * 4321), noise])
    d = torch.from_numpy(bufs).cuda()
    res = S.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, 4, cap.size, f, fc, fc, FS, pkg.STAGE_FULL)
    for b in range(4):
        co, _ = O.search_capbuf(iq_u8_to_capbuf(bufs[b]), f, fc, fc, FS)
        _cells_equal(res[b], co)
    assert [c.n_id_cell() for c in res[1]] == [277, 271] and len(res[0]) == 0 and len(res[2]) >= 1


def test_stage_entry_points_reject_unknown_cells(S, pkg, capbuf_0000):
    cap, fc = capbuf_0000
    with pytest.raises(pkg.SearcherError):
        S.pss_sss_foe(pkg.new_cell(ind=100, freq=0.0, n_id_2=1), cap, fc, fc, FS)     # cp_type unknown: reference throws
    with pytest.raises(pkg.SearcherError):
        S.extract_tfg(pkg.new_cell(ind=100, freq=0.0, n_id_2=1), cap, fc, fc, FS)
