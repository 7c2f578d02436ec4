"""GPU parity tests for the configurations round 2 left uncovered, every one against the CPU oracle:

 (a) fc_programmed != fc_requested and fs_programmed != 1.92e6 -- the normal case with a dongle
     (ref src/CellSearch.cpp:380-390, 481; every stage forms k_factor = (fc_requested - f) / fc_programmed,
     src/searcher.cpp:147, 741, 875) -- through the fused chain, every per-cell stage entry point, the batch API with
     a DIFFERENT fc_programmed per buffer, the streaming graph and the tracker block;
 (b) the per-cell stage entry points on an extended-CP cell (732-row grid, src/searcher.cpp:895);
 (c) the hipGraph streaming chain against the oracle (round 2 compared it with the eager GPU chain only);
 (d) BASELINE configs[3]'s grid: n_f = 35 at 715 MHz, array by array.

Synthetic captures are generated with the matching physics (lte-cell-scanner_amd/synth.py: true sample rate
fs_programmed * k_factor), so the cells really decode under the mismatched parameters and a swapped argument anywhere
moves frame_start by samples and the frequency estimates by Hz -- far outside the tolerances below."""
import os

import numpy as np
import pytest

import oracle as O
from conftest import golden, iq_u8_to_capbuf, f_search_set_for, load_pkg
from test_gpu_pss import _check_frq

pytestmark = pytest.mark.gpu
FS = 1.92e6
FC = 739e6
FCP = FC + 1234.0                 # what the dongle says it was programmed to
FSP = FS * (1 + 2e-5)
INT_FIELDS = ("ind", "n_id_2", "n_id_1", "cp_type", "n_ports", "n_rb_dl", "phich_duration", "phich_resource", "sfn")


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


CELLS_MIX = [dict(n_id_1=92, n_id_2=1, f_off=35e3, t0=1000.3),
             dict(n_id_1=33, n_id_2=2, cp_normal=False, n_ports=4, n_rb_dl=15, f_off=-41e3, t0=7000.6, gain_db=-2)]


@pytest.fixture(scope="module")
def mixed(pkg):
    """One normal-CP 2-port and one extended-CP 4-port cell, recorded with fc_programmed / fs_programmed off nominal."""
    iq, truth = pkg.synth.make_capbuf(4242, FC, CELLS_MIX, 10.0, fc_programmed=FCP, fs_programmed=FSP)
    return iq, iq_u8_to_capbuf(iq), truth


def _same_cell(a, b, what=""):
    for k in INT_FIELDS:
        assert getattr(a, k) == getattr(b, k), (what, k, getattr(a, k), getattr(b, k))
    assert a.freq == b.freq and a.fc_requested == b.fc_requested and a.fc_programmed == b.fc_programmed, what
    assert abs(a.pss_pow - b.pss_pow) <= 1e-5 * b.pss_pow, what
    assert abs(a.frame_start - b.frame_start) < 1e-6, (what, a.frame_start, b.frame_start)
    assert abs(a.freq_fine - b.freq_fine) < 1e-3 and abs(a.freq_superfine - b.freq_superfine) < 1e-3, what


def _same_cells(got, exp, what=""):
    assert [c.n_id_cell() for c in got] == [c.n_id_cell() for c in exp], what
    for a, b in zip(got, exp):
        _same_cell(a, b, what)


def _close(a, b, rtol):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() <= rtol * np.abs(b).max()


# ------------------------------------------------------------------------------------------------ (a) fused chains
def test_fused_chain_with_dongle_parameters(S, pkg, mixed):
    import torch
    iq, cap, truth = mixed
    f = f_search_set_for(FC, 100)
    co, po = O.search_capbuf(cap, f, FC, FCP, FSP)
    assert sorted(c.n_id_cell() for c in co) == sorted(t["n_id_cell"] for t in truth)     # both cells decode
    # the decode really depends on the two parameters: with nominal values the estimates move
    c_nom, _ = O.search_capbuf(cap, f, FC, FC, FS)
    assert abs(c_nom[0].frame_start - co[0].frame_start) > 0.5 and abs(c_nom[0].freq_superfine - co[0].freq_superfine) > 0.1
    # host entry point, complex<double>
    cells, peaks = S.search_capbuf(cap, f, FC, FCP, FSP)
    assert [(p.n_id_2, p.ind, p.freq) for p in peaks] == [(p.n_id_2, p.ind, p.freq) for p in po]
    _same_cells(cells, co, "search_capbuf")
    # device-resident u8 batch: the same capture under THREE different (fc_requested, fc_programmed) pairs in one call
    fr = np.array([FC, FC, FC + 200e3])
    fp = np.array([FCP, FC - 5000.0, FC + 200e3 + 777.0])
    d = torch.from_numpy(np.stack([iq, iq, iq])).cuda()
    res = S.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, 3, cap.size, f, fr, fp, FSP, pkg.STAGE_FULL)
    exp = [O.search_capbuf(cap, f, fr[b], fp[b], FSP)[0] for b in range(3)]
    for b in range(3):
        _same_cells(res[b], exp[b], f"batch buffer {b}")
        ro = O.xcorr_pss(cap, f, 2, fr[b], fp[b], FSP)
        r = S.batch_readback(b, f.size)
        assert (np.abs(r["single"].astype(np.float64) - ro["single"]) / ro["single"]).max() < 1e-5, b
        assert (np.abs(r["pow"] - ro["pow"]) / ro["pow"]).max() < 1e-5, b
    assert len(exp[0]) == 2
    # the three parameter sets are distinguishable (a mixed-up buffer index would fail above)
    assert abs(exp[0][0].frame_start - exp[1][0].frame_start) > 1e-3 or abs(exp[0][0].freq_fine - exp[1][0].freq_fine) > 1e-2
    # and through the host-buffer batch entry point (the path the C++ CLI takes)
    res_h = S.search_batch_host(np.stack([iq, iq, iq]), pkg.FMT_IQ_U8, 3, cap.size, f, fr, fp, FSP, pkg.STAGE_FULL)
    for b in range(3):
        _same_cells(res_h[b], exp[b], f"host batch buffer {b}")


# ------------------------------------------------------------------- (a) + (b) every stage entry point, both CP types
@pytest.mark.parametrize("fcp,fsp", [(FCP, FSP), (FC, FS)])
def test_stage_entry_points_normal_and_extended_cp(S, pkg, mixed, fcp, fsp):
    iq, cap, truth = mixed
    f = f_search_set_for(FC, 100)
    ro = O.xcorr_pss(cap, f, 2, FC, fcp, fsp)
    r = S.xcorr_pss(cap, f, 2, FC, fcp, fsp)
    assert (np.abs(r["single"].astype(np.float64) - ro["single"]) / ro["single"]).max() < 1e-5
    _check_frq(r["frq"], ro, "stage entry")
    Zo = O.z_th1(ro["sp_incoherent"], ro["n_comb_xc"])
    po = O.peak_search(ro["pow"], ro["frq"], Zo, f, FC, fcp, ro["single"], 2)
    pg = S.peak_search(r["pow"], r["frq"], pkg.z_th1(r["sp_incoherent"], r["n_comb_xc"]), f, FC, fcp, r["single"], 2)
    assert [(p.n_id_2, p.ind, p.freq, p.fc_programmed) for p in pg] == [(p.n_id_2, p.ind, p.freq, p.fc_programmed) for p in po]
    seen_cp = set()
    for pk_o, pk_g in zip(po, pg):
        co, do = O.sss_detect(pk_o, cap, 3.0, FC, fcp, fsp)
        cg, dg = S.sss_detect(pk_g, cap, 3.0, FC, fcp, fsp)
        for k in do:
            assert _close(dg[k], do[k], 1e-9), k
        assert (cg.n_id_1, cg.cp_type) == (co.n_id_1, co.cp_type)
        if co.n_id_1 < 0:
            continue
        assert abs(cg.frame_start - co.frame_start) < 1e-9
        fo, fg = O.pss_sss_foe(co, cap, FC, fcp, fsp), S.pss_sss_foe(cg, cap, FC, fcp, fsp)
        assert abs(fg.freq_fine - fo.freq_fine) < 1e-6
        tfg_o, ts_o = O.extract_tfg(fo, cap, FC, fcp, fsp)
        tfg_g, ts_g = S.extract_tfg(fg, cap, FC, fcp, fsp)
        assert tfg_g.shape == tfg_o.shape == ((854, 72) if co.cp_type == 1 else (732, 72))
        assert np.array_equal(ts_g, ts_o) and _close(tfg_g, tfg_o, 1e-10)
        c2o, tc_o, tsc_o = O.tfoec(fo, tfg_o, ts_o, FC, fcp)
        c2g, tc_g, tsc_g = S.tfoec(fg, tfg_o, ts_o, FC, fcp)
        assert abs(c2g.freq_superfine - c2o.freq_superfine) < 1e-7 and _close(tsc_g, tsc_o, 1e-13) and _close(tc_g, tc_o, 1e-9)
        for port in range(4):
            ce_o, np_o = O.chan_est(c2o, tc_o, port)
            ce_g, np_g = S.chan_est(c2g, tc_o, port)
            assert abs(np_g - np_o) <= 1e-11 * np_o and _close(ce_g, ce_o, 1e-9), (co.cp_type, port)
        mo, mg = O.decode_mib(c2o, tc_o), S.decode_mib(c2g, tc_o)
        for k in INT_FIELDS:
            assert getattr(mg, k) == getattr(mo, k), k
        if mo.n_rb_dl > 0:
            seen_cp.add(mo.cp_type)
    assert seen_cp == {1, 2}           # both a normal-CP and an extended-CP cell went through every stage and decoded


def test_xcorr_pss_sees_the_argument_order(S, mixed):
    """fc_requested and fc_programmed are not interchangeable anywhere: swapping them changes k_factor's sign of deviation."""
    _, cap, _ = mixed
    f = np.array([-40e3, 35e3])
    a = S.xcorr_pss(cap, f, 2, FC, FC + 50e3, FSP)
    b = S.xcorr_pss(cap, f, 2, FC + 50e3, FC, FSP)
    assert np.abs(a["single"] - b["single"]).max() > 1e-3 * a["single"].max()
    ro = O.xcorr_pss(cap, f, 2, FC, FC + 50e3, FSP)
    assert (np.abs(a["single"].astype(np.float64) - ro["single"]) / ro["single"]).max() < 1e-5


# ------------------------------------------------------------------------------------------------ (c) streaming graph
def test_stream_graph_against_oracle(pkg, mixed):
    """lcs_stream_* (ref src/searcher_thread.cpp:83-246) against O.search_capbuf with the single hypothesis of each push."""
    iq, cap, _ = mixed
    g = golden("capbuf_0000")["iq_u8"]
    rng = np.random.default_rng(11)
    noise = np.clip(np.rint(rng.normal(127.0, 15.0, g.size)), 0, 255).astype(np.uint8)
    with pkg.Searcher(0) as S:
        S.stream_open(pkg.FMT_IQ_U8, 153600, FC, FCP, FSP)
        for buf, f_off in [(iq, 35e3), (iq, -41e3), (noise, 35e3), (iq, -40e3), (g, 35e3)]:
            S.stream_push(buf, f_off)
            cells, dup, _ = S.stream_collect()
            exp, _ = O.search_capbuf(iq_u8_to_capbuf(buf), np.array([f_off]), FC, FCP, FSP)
            first = [c for k, c in enumerate(exp) if c.n_id_cell() not in [e.n_id_cell() for e in exp[:k]]]      # the reference tracks a cell from its first decoded peak on
            _same_cells(cells, first, f"stream f_off={f_off}")
            assert dup >= len(exp) - len(first)      # (+ later peaks of a decoded identity whose own MIB decode fails: also "already being tracked" in the reference)
        S.stream_close()
        # complex<float> pushes take the fp32 kernel: same cells
        S.stream_open(pkg.FMT_C64, 153600, FC, FCP, FSP)
        S.stream_push(cap.astype(np.complex64), 35e3)
        cells, _, _ = S.stream_collect()
        exp, _ = O.search_capbuf(cap.astype(np.complex64).astype(np.complex128), np.array([35e3]), FC, FCP, FSP)
        _same_cells(cells, exp, "stream c64")
        assert [c.n_id_cell() for c in cells] == [277]


# --------------------------------------------------------------------------------------- (d) cfg4's grid, 715 MHz
def test_cfg4_grid_n_f_35_full_arrays(S, pkg):
    """BASELINE configs[3]: `CellSearch -s 715e6 -e 768e6` builds its grid from freq_start only (src/CellSearch.cpp:463):
    ppm 120 at 715 MHz -> n_f = 35, used for every carrier of the sweep.  Both ends of the band, every array element."""
    from test_gpu_pss import _batch_arrays_vs_oracle
    f = f_search_set_for(715e6, 120)
    assert f.size == 35
    s1, _ = pkg.synth.make_capbuf(715, 715e6, [dict(n_id_1=101, n_id_2=0, f_off=-83e3), dict(n_id_1=7, n_id_2=2, f_off=80e3, gain_db=-5)], 4.0)
    s2, _ = pkg.synth.make_capbuf(768, 768e6, [dict(n_id_1=55, n_id_2=1, f_off=61e3, cp_normal=False)], 2.0)
    _batch_arrays_vs_oracle(S, pkg, [s1, s2], f, np.array([715e6, 768e6]), 153600, "cfg4 grid n_f=35")
    import torch
    d = torch.from_numpy(np.stack([s1, s2])).cuda()
    res = S.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, 2, 153600, f, np.array([715e6, 768e6]), np.array([715e6, 768e6]), FS, pkg.STAGE_FULL)
    for b, (iq, fc) in enumerate(((s1, 715e6), (s2, 768e6))):
        _same_cells(res[b], O.search_capbuf(iq_u8_to_capbuf(iq), f, fc, fc, FS)[0], f"cfg4 buffer {b}")
    assert sorted(c.n_id_cell() for c in res[0]) == [23, 303] and [c.n_id_cell() for c in res[1]] == [166]


# ------------------------------------------------------------------------------------------- (a) the tracker block
def test_track_block_with_dongle_parameters(pkg, mixed):
    """lcs_track_block on both cells of the mismatched capture (one normal CP / 2 ports, one extended CP / 4 ports), cut
    with the producer's time base for (fc_requested, fc_programmed, fs_programmed): every array against the oracle."""
    _, cap, _ = mixed
    cells, _ = O.search_capbuf(cap, f_search_set_for(FC, 100), FC, FCP, FSP)
    assert len(cells) == 2
    blocks = []
    for c in cells:
        k_factor = (FC - c.freq_superfine) / FCP
        ft = c.frame_start * (30.72e6 / 16) / (FSP * k_factor)
        n_sym = 7 * (140 if c.cp_type == 1 else 120)        # 7 frames: the 40 ms PBCH alignment falls on one of offsets 0..3
        blocks.append((c,) + pkg.tracker.cut_symbols(cap, ft, c.cp_type, c.freq_superfine, FC, FCP, FSP, n_sym))
    with pkg.Searcher(0) as S:
        for c, td, late, ftv, fov in blocks:
            assert td.shape[0] == 7 * (140 if c.cp_type == 1 else 120)
            g = S.track_block([c], td, fov, ftv, late, FC, FCP, FSP)
            r = _oracle_block_x(c, td, late, ftv, fov)
            assert np.abs(g["syms"][0] - r["syms"]).max() < 1e-11 * np.abs(r["syms"]).max()
            assert abs(g["bpo"][0] - r["bpo"]) < 1e-9
            assert np.array_equal(g["n_meas"][0], r["n_meas"]) and np.array_equal(g["ce_upto"][0], r["ce_upto"])
            for p in range(c.n_ports):
                n = r["n_meas"][p]
                gm, om = g["meas"][0, p, :n], r["meas"][p, :n]
                assert np.array_equal(gm[:, 0], om[:, 0]) and np.abs(gm[:, 1:5] - om[:, 1:5]).max() < 1e-11 * om[:, 2].max()
                assert np.abs(gm[:, 5] - om[:, 5]).max() < 1e-6 and np.abs(gm[:, 7] - om[:, 7]).max() < 1e-8
                u = r["ce_upto"][p]
                assert np.abs(g["ce"][0, p, :u] - r["ce"][p, :u]).max() < 1e-11 * np.abs(r["ce"][p, :u]).max()
            locks = 0
            for o, m in enumerate(r["mib"]):
                if m is None:
                    assert g["mib_ok"][0, o] == -1
                    continue
                assert g["mib_ok"][0, o] == (1 if m[1] else 0) | (2 if m[2] else 0), (c.n_id_cell(), o)
                assert [(int(g["mib_bits"][0, o]) >> k) & 1 for k in range(40)] == list(m[0])
                locks += g["mib_ok"][0, o] == 3
            assert locks >= 1, c.n_id_cell()
            # the wrong parameters give different symbols (the block really uses them)
            g2 = S.track_block([c], td, fov, ftv, late, FC, FC, FS, want_ce=False)
            assert np.abs(g2["syms"][0] - g["syms"][0]).max() > 1e-6 * np.abs(r["syms"]).max()


def _oracle_block_x(c, td, late, ftv, fov):
    """tests/test_tracker.py's oracle block for arbitrary (fc_programmed, fs_programmed) and CP type."""
    syms, bpo, _ = O.trk_get_fd(c, td, 0, 0, fov, late, FC, FCP, FSP)
    r = O.trk_chan_est(c, syms, 0, 0, fov, ftv, FC, FCP, FSP)
    r.update(syms=syms, bpo=bpo)
    per_frame = 140 if c.cp_type == 1 else 120
    nsd = per_frame // 20
    upto = int(min(r["ce_upto"][:c.n_ports]))
    mib = []
    for o in range(td.shape[0] // per_frame - 3):
        ii = [(o + fr) * per_frame + nsd + s for fr in range(4) for s in range(4)]
        if ii[-1] >= upto:
            mib.append(None)
            continue
        mib.append(O.trk_mib(c, syms[ii], r["ce"][:c.n_ports][:, ii], r["ce_pw"][:c.n_ports][:, ii, 3]))
    r["mib"] = mib
    return r


# ------------------------------------------------------------------------- int8 accumulator bounds on real hardware
def test_saturated_full_scale_buffers(S, pkg):
    """nothing drove the int8 kernel's operands to full scale.  Buffers made of the two extreme codes only
    (0 -> int8 +127, 255 -> int8 -128: the corner of 127 - u8), one white, one held for 12 samples at a time -- the
    largest operands the shared digit-1/0 accumulator 256 S1 + S0 (csrc/pss_xcorr_i8.hip) can meet; its bound itself
    is arithmetic (tests/test_i8_digits.py) -- every array element against the oracle."""
    from test_gpu_pss import _batch_arrays_vs_oracle
    rng = np.random.default_rng(23)
    n = 2 * 153600
    sat = np.where(rng.random(n) < 0.5, 0, 255).astype(np.uint8)
    # I and Q held for 12 samples at a time: partial sums add coherently, but no 137-tap window is constant (a constant
    # window correlates to ~0 with the DC-free PSS and an element-wise relative comparison would be meaningless)
    runs = np.repeat(np.where(rng.random((n // 24 + 1, 2)) < 0.5, 0, 255), 12, axis=0).reshape(-1)[:n].astype(np.uint8)
    f = f_search_set_for(FC, 100)
    _batch_arrays_vs_oracle(S, pkg, [sat, runs], f, np.array([FC, FC]), 153600, "saturated buffers")


def test_track_stream_mixed_cp_types_and_dongle_parameters(pkg, mixed):
    """lcs_track_stream_block with a normal-CP and an extended-CP cell in ONE stream (frames of 140 and 120 symbols: the two
    carry different amounts of history and are processed as two groups inside the call), cut in the middle of frames, under
    fc_programmed / fs_programmed off nominal: row for row what lcs_track_block returns for each cell's whole stream."""
    _, cap, _ = mixed
    cells, _ = O.search_capbuf(cap, f_search_set_for(FC, 100), FC, FCP, FSP)
    assert sorted(c.cp_type for c in cells) == [1, 2]
    n_sym = 840                                    # 6 normal-CP frames / 7 extended-CP frames
    per = []
    for c in cells:
        k_factor = (FC - c.freq_superfine) / FCP
        ft = c.frame_start * (30.72e6 / 16) / (FSP * k_factor)
        per.append(pkg.tracker.cut_symbols(cap, ft, c.cp_type, c.freq_superfine, FC, FCP, FSP, n_sym))
        assert per[-1][0].shape[0] == n_sym
    td = np.stack([p[0] for p in per]); late = np.stack([p[1] for p in per])
    ftv = np.stack([p[2] for p in per]); fov = np.stack([p[3] for p in per])
    with pkg.Searcher(0) as S:
        ones = [S.track_block([c], td[i:i + 1], fov[i:i + 1], ftv[i:i + 1], late[i:i + 1], FC, FCP, FSP) for i, c in enumerate(cells)]
        parts, a = [], 0
        for n in (333, 200, 307):
            parts.append(S.track_stream_block(cells, td[:, a:a + n], fov[:, a:a + n], ftv[:, a:a + n], late[:, a:a + n], FC, FCP, FSP))
            a += n
        # the same stream with the symbols handed over in DEVICE memory (asynchronous copies inside the call; the two CP types go
        # through the per-cell copy path, not the one strided copy of a one-CP-type stream): identical rows
        import torch
        S.track_stream_reset()
        parts_dev, a = [], 0
        for n in (333, 200, 307):
            d_td = torch.from_numpy(np.ascontiguousarray(td[:, a:a + n])).cuda()
            parts_dev.append(S.track_stream_block(cells, None, fov[:, a:a + n], ftv[:, a:a + n], late[:, a:a + n], FC, FCP, FSP, td_device_ptr=d_td.data_ptr()))
            del d_td
            a += n
    for q, qd in zip(parts, parts_dev):
        for k in ("syms", "n_meas", "ce_n", "n_mib"):
            assert np.array_equal(q[k], qd[k]), k
        for i in range(len(cells)):
            for p in range(4):
                assert np.array_equal(q["meas"][i, p, :q["n_meas"][i, p]], qd["meas"][i, p, :qd["n_meas"][i, p]])
                assert np.array_equal(q["ce"][i, p, :q["ce_n"][i, p]], qd["ce"][i, p, :qd["ce_n"][i, p]])
            assert np.array_equal(q["mib_ok"][i, :q["n_mib"][i]], qd["mib_ok"][i, :qd["n_mib"][i]])
    for i, c in enumerate(cells):
        one = ones[i]
        assert np.array_equal(np.concatenate([q["syms"][i] for q in parts]), one["syms"][0])
        assert parts[-1]["bpo"][i] == one["bpo"][0]
        for p in range(4):
            rows = np.concatenate([q["meas"][i, p, :q["n_meas"][i, p]] for q in parts])
            assert np.array_equal(rows, one["meas"][0, p, :one["n_meas"][0, p]]), (c.n_id_cell(), p)
            ce = np.concatenate([q["ce"][i, p, :q["ce_n"][i, p]] for q in parts])
            assert ce.shape[0] == one["ce_upto"][0, p] and np.array_equal(ce, one["ce"][0, p, :one["ce_upto"][0, p]])
        ok = np.concatenate([q["mib_ok"][i, :q["n_mib"][i]] for q in parts])
        tried = one["mib_ok"][0] != -1
        assert np.array_equal(ok, one["mib_ok"][0][tried]) and np.array_equal(
            np.concatenate([q["mib_bits"][i, :q["n_mib"][i]] for q in parts]), one["mib_bits"][0][tried])


# ------------------------------------------------------------------------ complex<float> batches that are dongle data
def test_float_batches_of_dongle_data_take_the_u8_route(pkg):
    """lcs_set_float_batch_probe: a device-resident LCS_FMT_C64 batch whose every component is (u8 - 127) / 128 is recognised on the
    device and handed to the u8 route -- the int8 kernel, byte-identical records to the same captures handed over as bytes, equal to
    the oracle; a batch with ONE component off the 8-bit grid (or a NaN) takes the fp16 kernel as before; off by default."""
    import torch
    f = f_search_set_for(FC, 100)
    fcs = FC + 100e3 * np.arange(4)
    cells = [[dict(n_id_1=31, n_id_2=1, f_off=22e3)], [], [dict(n_id_1=140, n_id_2=0, cp_normal=False, n_ports=4, f_off=-48e3), dict(n_id_1=7, n_id_2=2, f_off=9e3, gain_db=-4)], []]
    bufs = [pkg.synth.make_capbuf(900 + k, fcs[k], cells[k], 5.0)[0] for k in range(4)]
    d8 = torch.from_numpy(np.ascontiguousarray(np.stack(bufs))).cuda()
    h32 = np.stack([iq_u8_to_capbuf(b).astype(np.complex64) for b in bufs])
    d32 = torch.from_numpy(h32).cuda()
    with pkg.Searcher(0) as S:
        as_bytes = S.search_batch(d8.data_ptr(), pkg.FMT_IQ_U8, 4, 153600, f, fcs, fcs, FS, pkg.STAGE_FULL)
        S.search_batch(d32.data_ptr(), pkg.FMT_C64, 4, 153600, f, fcs, fcs, FS, pkg.STAGE_FULL)
        assert S.last_xcorr_info()[0] == "k_xcorr_f16x3"                          # off by default
        S.set_float_batch_probe(True)
        routed = S.search_batch(d32.data_ptr(), pkg.FMT_C64, 4, 153600, f, fcs, fcs, FS, pkg.STAGE_FULL)
        assert S.last_xcorr_info()[0] == "k_xcorr_i8x3"
        for b in range(4):
            assert [bytes(c) for c in routed[b]] == [bytes(c) for c in as_bytes[b]], b      # the same records, NaNs and all
            _same_cells(routed[b], O.search_capbuf(iq_u8_to_capbuf(bufs[b]), f, fcs[b], fcs[b], FS)[0], f"routed buffer {b}")
            r, r8 = S.batch_readback(b, f.size), None
            ro = O.xcorr_pss(iq_u8_to_capbuf(bufs[b]), f, 2, fcs[b], fcs[b], FS)
            assert (np.abs(r["single"].astype(np.float64) - ro["single"]) / ro["single"]).max() < 1e-5 and np.array_equal(r["frq"], ro["frq"]), b
        assert sum(len(x) for x in routed) >= 3
        # one component off the grid (half a step), then a NaN: the fp16 kernel, and the results of a float front end
        for bad in (np.float32(0.5 / 128), np.float32(np.nan)):
            h = h32.copy()
            h[2, 77777] = h[2, 77777] + bad
            dd = torch.from_numpy(h).cuda()
            S.set_float_batch_probe(True)                                          # (resets the skip counter)
            got = S.search_batch(dd.data_ptr(), pkg.FMT_C64, 4, 153600, f, fcs, fcs, FS, pkg.STAGE_PSS)
            assert S.last_xcorr_info()[0] == "k_xcorr_f16x3", bad
            if not np.isnan(bad):
                po = O.peak_search(*(lambda ro: (ro["pow"], ro["frq"], O.z_th1(ro["sp_incoherent"], ro["n_comb_xc"])))(O.xcorr_pss(h[2].astype(np.complex128), f, 2, fcs[2], fcs[2], FS)),
                                   f, fcs[2], fcs[2], O.xcorr_pss(h[2].astype(np.complex128), f, 2, fcs[2], fcs[2], FS)["single"], 2)
                assert [(c.n_id_2, c.ind) for c in got[2]] == [(c.n_id_2, c.ind) for c in po]
        # after a batch that was not dongle data the next batches are not probed
        S.search_batch(d32.data_ptr(), pkg.FMT_C64, 4, 153600, f, fcs, fcs, FS, pkg.STAGE_PSS)
        assert S.last_xcorr_info()[0] == "k_xcorr_f16x3"
        S.set_float_batch_probe(True)
        S.search_batch(d32.data_ptr(), pkg.FMT_C64, 4, 153600, f, fcs, fcs, FS, pkg.STAGE_PSS)
        assert S.last_xcorr_info()[0] == "k_xcorr_i8x3"
