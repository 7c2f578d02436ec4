"""GPU parity tests for the PSS stage (xcorr_pss + peak_search) through the C ABI.

Bar (BASELINE.json north_star): PSS peak indices / n_id_2 / frequency indices bit-exact,
correlation magnitudes within 1e-5 relative of the CPU oracle."""
import numpy as np
import pytest

import oracle as O
from conftest import golden, iq_u8_to_capbuf, f_search_set_for, load_pkg

pytestmark = pytest.mark.gpu
FS = 1.92e6
RTOL = 1e-5     # north_star: "correlation magnitudes within 1e-5 relative"


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


def _check_xcorr(r, ro, what="", rtol=RTOL):
    assert r["n_comb_xc"] == ro["n_comb_xc"] and r["n_comb_sp"] == ro["n_comb_sp"]
    for k in ("single", "incoherent"):
        if r.get(k) is None:
            continue
        err = np.abs(r[k].astype(np.float64) - ro[k]) / ro[k]
        assert err.max() < rtol, f"{what} {k}: max rel err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
        assert np.abs(r[k].astype(np.float64) - ro[k]).max() < 1e-6 * ro[k].max()
    _check_frq(r["frq"], ro, what)      # equal: an integer output
    assert np.abs(r["pow"] - ro["pow"]).max() <= 1e-6 * ro["pow"].max()
    assert (np.abs(r["pow"] - ro["pow"]) / ro["pow"]).max() < rtol
    assert (np.abs(r["sp_incoherent"] - ro["sp_incoherent"]) / ro["sp_incoherent"]).max() < 1e-11


def test_xcorr_pss_capbuf_0000_default_grid(S, capbuf_0000):
    cap, fc = capbuf_0000
    f = f_search_set_for(fc, 120)                 # 37 hypotheses, the CLI default at 739 MHz
    r = S.xcorr_pss(cap, f, 2, fc, fc, FS)
    ro = O.xcorr_pss(cap, f, 2, fc, fc, FS)
    _check_xcorr(r, ro, "capbuf_0000")
    Z = load_pkg().z_th1(r["sp_incoherent"], r["n_comb_xc"])
    Zo = O.z_th1(ro["sp_incoherent"], ro["n_comb_xc"])
    assert (np.abs(Z - Zo) / Zo).max() < 1e-10
    peaks = S.peak_search(r["pow"], r["frq"], Z, f, fc, fc, r["single"], 2)
    assert [(p.n_id_2, p.ind, p.freq) for p in peaks] == [(1, 1410, 35000.0), (1, 6990, 35000.0),
                                                          (2, 1314, 45000.0), (0, 1327, 30000.0)]
    po = O.peak_search(ro["pow"], ro["frq"], Zo, f, fc, fc, ro["single"], 2)
    for a, b in zip(peaks, po):
        assert abs(a.pss_pow - b.pss_pow) < RTOL * b.pss_pow
        assert a.fc_requested == fc and a.fc_programmed == fc and a.n_id_1 == -1 and np.isnan(a.frame_start)

def _check_frq(frq, ro, tag):
    """xc_peak_freq (src/searcher.cpp:353-383) is an argmax over the frequency axis of float values; its result is an
    INTEGER output of the reference (include/searcher.h:31) and must be EQUAL.  The matrix-core kernels reproduce the
    float values to ~1e-7 relative; wherever the best two hypotheses are closer than LCS_FRQ_TIE_EPS the library
    recomputes the candidates in the reference's own arithmetic (k_frq_repair), so no tie is left to chance."""
    bad = np.argwhere(frq != ro["frq"])
    msg = ""
    for t, i in bad[:8]:
        a, b = ro["incoherent"][t, i, frq[t, i]], ro["incoherent"][t, i, ro["frq"][t, i]]
        msg += f" frq[{t},{i}] = {frq[t, i]} vs {ro['frq'][t, i]} (oracle values {a!r} vs {b!r}, margin {abs(float(a) - float(b)) / float(b):.2e});"
    assert len(bad) == 0, f"{tag}: {len(bad)} frequency indices differ:{msg}"


def _batch_arrays_vs_oracle(S, pkg, bufs_u8, f, fcs, n_cap, what):
    """Run `bufs_u8` through the device-resident batch entry point as raw u8 I/Q (int8 MFMA kernel when the grid
    fits it) and as complex<float> (fp16 three-product MFMA kernel), read back EVERY element of xc_incoherent_single / collapsed
    pow / frq / sp_incoherent / Z_th1 of every buffer and compare with the oracle (reference semantics:
    src/searcher.cpp:263-308, 353-383; tolerance as test/test_xcorr_pss.cpp:104-109 and the 1e-5 of north_star)."""
    import torch
    n_buf = len(bufs_u8)
    d8 = torch.from_numpy(np.ascontiguousarray(np.stack(bufs_u8))).cuda()
    d32 = torch.from_numpy(np.stack([iq_u8_to_capbuf(b).astype(np.complex64) for b in bufs_u8])).cuda()
    oracle = []
    for b in range(n_buf):
        ro = O.xcorr_pss(iq_u8_to_capbuf(bufs_u8[b]), f, 2, fcs[b], fcs[b], FS)
        ro["z_th1"] = O.z_th1(ro["sp_incoherent"], ro["n_comb_xc"])
        oracle.append(ro)
    for fmt, dptr, name, kernel in ((pkg.FMT_IQ_U8, d8.data_ptr(), "u8/int8", "k_xcorr_i8x3"), (pkg.FMT_C64, d32.data_ptr(), "c64/fp16x3", "k_xcorr_f16x3")):
        S.search_batch(dptr, fmt, n_buf, n_cap, f, fcs, fcs, FS, pkg.STAGE_PSS, max_cells_per_buf=64)
        assert S.last_xcorr_info()[0] == kernel
        for b in range(n_buf):
            r, ro = S.batch_readback(b, f.size), oracle[b]
            tag = f"{what} [{name}] buffer {b}"
            assert r["single"].shape == ro["single"].shape
            err = np.abs(r["single"].astype(np.float64) - ro["single"]) / ro["single"]
            assert err.max() < RTOL, f"{tag} single: max rel err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
            _check_frq(r["frq"], ro, tag)
            assert (np.abs(r["pow"] - ro["pow"]) / ro["pow"]).max() < RTOL, tag
            assert (np.abs(r["sp_incoherent"] - ro["sp_incoherent"]) / ro["sp_incoherent"]).max() < 1e-11, tag
            assert (np.abs(r["z_th1"] - ro["z_th1"]) / ro["z_th1"]).max() < 1e-10, tag


@pytest.mark.parametrize("ppm", [100, 120])
def test_batched_kernels_full_arrays_vs_oracle_capbuf_0000(S, pkg, ppm):
    """BASELINE configs[1]/[2] grids (n_f = 31 at +-100 ppm, 37 at the CLI default): the recorded buffer, a rotated
    copy, two synthetic buffers and a noise-only buffer -- 3 x 9600 x n_f values each, all compared."""
    g = golden("capbuf_0000")
    fc = float(g["fc"][0])
    f = f_search_set_for(fc, ppm)
    rng = np.random.default_rng(17)
    noise = np.clip(np.rint(rng.normal(127.0, 14.0, g["iq_u8"].size)), 0, 255).astype(np.uint8)
    s1, _ = pkg.synth.make_capbuf(901, fc, [dict(n_id_1=12, n_id_2=0, f_off=22e3), dict(n_id_1=150, n_id_2=2, f_off=21e3, gain_db=-6)], 3.0)
    s2, _ = pkg.synth.make_capbuf(902, fc + 300e3, [dict(n_id_1=77, n_id_2=1, f_off=-61e3, cp_normal=False)], 0.0)
    extreme = g["iq_u8"].copy()
    extreme[:4000:7] = 255       # both ends of the u8 range (127 - 255 = -128 is the int8 corner case)
    extreme[1:4000:11] = 0
    bufs = [g["iq_u8"], s1, s2, noise, extreme]
    fcs = np.array([fc, fc, fc + 300e3, fc - 100e3, fc])
    _batch_arrays_vs_oracle(S, pkg, bufs, f, fcs, 153600, f"capbuf_0000 set, n_f={f.size}")


def test_batched_kernels_full_arrays_vs_oracle_short_buffer_and_odd_grids(S, pkg):
    """The 135360-sample Matlab/test_xcorr_pss.mat buffer (14 combining windows) on its own 3-entry grid, on a
    single hypothesis (one template group with 3 of 16 columns in use) and on a 10 kHz grid whose window starts
    spread over many samples inside a group (int8 kernel with 5 full tap blocks), plus a 40 kHz grid that is too
    sparse for 16 templates per group (the groups are then packed with fewer hypotheses each)."""
    g = golden("test_xcorr_pss")
    fc = float(g["fc"][0])
    n = g["iq_u8"].size // 2
    assert n == 135360
    g0 = golden("capbuf_0000")["iq_u8"][: 2 * n]
    bufs = [g["iq_u8"], g0]
    fcs = np.array([fc, fc])
    for f in (g["f_search_set"].astype(float), np.array([35e3]), np.arange(-10, 11) * 10e3, np.arange(-4, 5) * 40e3):
        _batch_arrays_vs_oracle(S, pkg, bufs, f, fcs, n, f"135360-sample set, grid step {f[1] - f[0] if f.size > 1 else 0}")


def test_xcorr_pss_noisy_buffer_matches_golden_peaks(S):
    """test_sss_detect.it: the 24 golden input peaks sit where the collapsed arrays say."""
    g = golden("test_sss_detect")
    f = np.arange(20e3, 60e3 + 1, 5e3)
    fc = float(g["fc"][0])
    r = S.xcorr_pss(g["capbuf"], f, 2, fc, fc, FS)
    ro = O.xcorr_pss(g["capbuf"], f, 2, fc, fc, FS)
    _check_xcorr(r, ro, "test_sss_detect")
    ind, n2 = g["peaks_ind"] - 1, g["peaks_n_id_2"]
    assert np.array_equal(f[r["frq"][n2, ind]], g["peaks_freq"].astype(float))
    assert (np.abs(r["pow"][n2, ind] - g["peaks_pow"]) / g["peaks_pow"]).max() < 2e-4   # SURVEY 4.3 (MATLAB semantics)


def test_xcorr_pss_short_buffer_14_windows(S):
    g = golden("test_xcorr_pss")
    cap = iq_u8_to_capbuf(g["iq_u8"])
    fc = float(g["fc"][0])
    r = S.xcorr_pss(cap, g["f_search_set"], 2, fc, fc, FS)
    ro = O.xcorr_pss(cap, g["f_search_set"], 2, fc, fc, FS)
    assert r["n_comb_xc"] == 14 and r["n_comb_sp"] == 14
    _check_xcorr(r, ro, "135360-sample buffer")


def test_xcorr_pss_edge_grids(S, capbuf_0000):
    """n_f = 1 (tracker mode), unsorted grid, different ds_comb_arm, fc_programmed != fc_requested."""
    cap, fc = capbuf_0000
    for f, ds, fcp, fs in (([35e3], 2, fc, FS), ([40e3, -20e3, 35e3, 0.0], 1, fc + 1234.0, FS * (1 + 2e-5)),
                           ([35e3, 35e3], 0, fc, FS)):
        f = np.array(f)
        r = S.xcorr_pss(cap, f, ds, fc, fcp, fs)
        ro = O.xcorr_pss(cap, f, ds, fc, fcp, fs)
        _check_xcorr(r, ro, f"grid {f.tolist()} ds={ds}")


def test_xcorr_pss_minimum_length_buffer(S, capbuf_0000):
    """A buffer with exactly ONE combining window: nothing averages over windows, xc_incoherent_single is |xc|^2 of one 137-tap sum
    and a few hundred of its elements lie 40 dB and more below the mean -- where fixed-point templates (rounds 1-5: up to 1e-3
    relative there, a documented exception) cannot follow the reference's floating-point ones.  Since round 6 such buffers take
    k_single_exact: every element in the reference's own arithmetic (fp64 sums in tap order stored as complex<float>,
    src/searcher.cpp:160-169, 299-305).  Every element within 1e-6 (a last-bit difference of a template's sincos may flip one float
    rounding), the usual bars for the rest."""
    cap, fc = capbuf_0000
    n = 9600 + 136 + 137 + 100            # exactly one combining window
    f = np.array([30e3, 35e3, 40e3])
    r = S.xcorr_pss(cap[:n], f, 2, fc, fc, FS)
    ro = O.xcorr_pss(cap[:n], f, 2, fc, fc, FS)
    assert r["n_comb_xc"] == 1 and S.last_xcorr_info()[0] == "k_single_exact"
    _check_xcorr(r, ro, "one-window buffer")
    err = np.abs(r["single"].astype(np.float64) - ro["single"]) / ro["single"]
    assert err.max() < 1e-6, err.max()
    assert ro["single"].min() < 1e-4 * ro["single"].mean()          # the deep nulls are really there


@pytest.mark.parametrize("n_win", [2, 3])
def test_xcorr_pss_two_and_three_window_buffers(S, capbuf_0000, n_win):
    """The shortest buffers that DO take the matrix-core kernel: with two windows the sum of two exponentials leaves no element
    deep enough in a null for the fixed-point templates' floor to show -- every element within the usual 1e-5."""
    cap, fc = capbuf_0000
    n = n_win * 9600 + 136 + 100 + 41
    f = np.array([30e3, 35e3, 40e3])
    r = S.xcorr_pss(cap[:n], f, 2, fc, fc, FS)
    ro = O.xcorr_pss(cap[:n], f, 2, fc, fc, FS)
    assert r["n_comb_xc"] == n_win and S.last_xcorr_info()[0] == "k_xcorr_i8x3"
    _check_xcorr(r, ro, f"{n_win}-window buffer")


def test_one_window_buffers_in_batches(S):
    """The same through the batch entry point, as raw u8 I/Q and as complex<float>, on the CLI's 37-hypothesis grid (7 template
    groups, the last one partly empty), two buffers with different carriers: every element of single / pow / frq / Z_th1."""
    import torch
    pkg = load_pkg()
    n = 9600 + 136 + 137 + 100 + 57
    f = f_search_set_for(739e6, 120)
    fcs = np.array([739e6, 739.3e6])
    bufs = [pkg.synth.make_capbuf(60 + k, fcs[k], [dict(n_id_1=11 + k, n_id_2=k, f_off=(-1) ** k * 52e3)], 3.0)[0][:2 * n] for k in range(2)]
    d8 = torch.from_numpy(np.ascontiguousarray(np.stack(bufs))).cuda()
    d32 = torch.from_numpy(np.stack([iq_u8_to_capbuf(b).astype(np.complex64) for b in bufs])).cuda()
    for fmt, dptr in ((pkg.FMT_IQ_U8, d8.data_ptr()), (pkg.FMT_C64, d32.data_ptr())):
        S.search_batch(dptr, fmt, 2, n, f, fcs, fcs, FS, pkg.STAGE_PSS, max_cells_per_buf=64)
        assert S.last_xcorr_info()[0] == "k_single_exact"
        for b in range(2):
            ro = O.xcorr_pss(iq_u8_to_capbuf(bufs[b]), f, 2, fcs[b], fcs[b], FS)
            r = S.batch_readback(b, f.size)
            err = np.abs(r["single"].astype(np.float64) - ro["single"]) / ro["single"]
            assert err.max() < 1e-6, (fmt, b, err.max())
            _check_frq(r["frq"], ro, f"one-window batch buffer {b}")
            assert (np.abs(r["pow"] - ro["pow"]) / ro["pow"]).max() < 1e-6
            zo = O.z_th1(ro["sp_incoherent"], ro["n_comb_xc"])
            assert (np.abs(r["z_th1"] - zo) / zo).max() < 1e-10


def test_debug_outputs_xc_and_sp(S, capbuf_0000):
    cap, fc = capbuf_0000
    f = np.array([35e3, 40e3])
    r = S.xcorr_pss(cap, f, 2, fc, fc, FS, want_xc=True, want_sp=True)
    ro = O.xcorr_pss(cap, f, 2, fc, fc, FS, want_xc=True, want_sp=True)
    scale = np.abs(ro["xc"]).max()
    assert np.abs(r["xc"] - ro["xc"]).max() < 1e-6 * scale       # test/test_xcorr_pss.cpp:107 uses 1e-6
    assert (np.abs(r["sp"] - ro["sp"]) / ro["sp"]).max() < 1e-11


def test_peak_search_golden_fixture(S):
    """test/test_peak_search.it through the GPU peak_search (mirrors test/test_peak_search.cpp)."""
    g = golden("test_peak_search")
    pow_ = g["xc_incoherent_collapsed_pow"]
    frq = g["xc_incoherent_collapsed_frq"] - 1
    f = g["f_search_set"].astype(float)
    single = np.repeat(pow_[:, :, None], f.size, axis=2).astype(np.float32)
    cells = S.peak_search(pow_, frq, g["Z_th1"], f, 739e6, 739e6, single, 0)
    assert len(cells) == 20
    for c, p, i, fr, n2 in zip(cells, g["peaks_pow"], g["peaks_ind"] - 1, g["peaks_freq"], g["peaks_n_id_2"]):
        assert c.pss_pow == p and (c.ind, c.freq, c.n_id_2) == (i, fr, n2)


def test_peak_search_quirks(S):
    """Q2 (peak_ind < ds_comb_arm -> ind = -1), wrap-around cancellation, tie-break order."""
    f = np.array([0.0, 5000.0])
    pow_ = np.full((3, 9600), 1e-3)
    frq = np.zeros((3, 9600), np.int32)
    single = np.full((3, 9600, 2), 1e-3, np.float32)
    pow_[2, 1] = 5.0          # peak_ind = 1 < ds = 2 -> refined index -1
    pow_[0, 9599] = 4.0       # near the wrap
    single[0, 0, 0] = 9.0     # the refine step must pick idx 0 (= 9600 mod 9600)
    pow_[1, 4000] = 3.0
    pow_[2, 4000] = 3.0       # tie: PSS 1 must come first
    Z = np.full(9600, 2.0)
    got = S.peak_search(pow_, frq, Z, f, 1e9, 1e9, single, 2)
    exp = O.peak_search(pow_, frq, Z, f, 1e9, 1e9, single, 2)
    assert [(c.n_id_2, c.ind, c.freq, c.pss_pow) for c in got] == [(c.n_id_2, c.ind, c.freq, c.pss_pow) for c in exp]
    assert [(c.n_id_2, c.ind) for c in got] == [(2, -1), (0, 0), (1, 3998), (2, 3998)]


def test_batch_device_api_matches_single_calls(S, pkg, capbuf_0000):
    """Device-resident batch entry point, both ingest formats, per-buffer carrier frequencies."""
    import torch
    cap, fc = capbuf_0000
    g = golden("capbuf_0000")
    f = f_search_set_for(fc, 100)
    rng = np.random.default_rng(3)
    noise_iq = rng.integers(100, 156, size=g["iq_u8"].size, dtype=np.uint8)
    bufs_u8 = np.stack([g["iq_u8"], noise_iq, g["iq_u8"]])
    fcs = np.array([fc, fc + 100e3, fc - 200e3])
    d = torch.from_numpy(bufs_u8).cuda()
    res = S.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, 3, cap.size, f, fcs, fcs, FS, pkg.STAGE_PSS)
    d32 = torch.from_numpy(np.stack([iq_u8_to_capbuf(b).astype(np.complex64) for b in bufs_u8])).cuda()
    res32 = S.search_batch(d32.data_ptr(), pkg.FMT_C64, 3, cap.size, f, fcs, fcs, FS, pkg.STAGE_PSS)
    for b in range(3):
        capb = iq_u8_to_capbuf(bufs_u8[b])
        ro = O.xcorr_pss(capb, f, 2, fcs[b], fcs[b], FS)
        po = O.peak_search(ro["pow"], ro["frq"], O.z_th1(ro["sp_incoherent"], 15), f, fcs[b], fcs[b], ro["single"], 2)
        for got in (res[b], res32[b]):
            assert [(c.n_id_2, c.ind, c.freq) for c in got] == [(c.n_id_2, c.ind, c.freq) for c in po], b
            for x, y in zip(got, po):
                assert abs(x.pss_pow - y.pss_pow) < RTOL * y.pss_pow and x.fc_requested == fcs[b]
    assert len(res[1]) == 0 and len(res[0]) == 4


def test_host_entry_points_pick_the_kernel_from_the_data(S, pkg, capbuf_0000):
    """The reference's call shape hands over complex<double> (searcher.h: cvec capbuf).  A dongle capture is exactly
    (u8 - 127) / 128 per component (src/capbuf.cpp:172-181): the library detects that on the device and takes the int8
    kernel; any other buffer takes the fp32 kernel.  Both against the oracle; both corners of the u8 range count as exact."""
    cap, fc = capbuf_0000
    f = f_search_set_for(fc, 100)
    r = S.xcorr_pss(cap, f, 2, fc, fc, FS)
    assert S.last_xcorr_info()[0] == "k_xcorr_i8x3"
    ro = O.xcorr_pss(cap, f, 2, fc, fc, FS)
    _check_xcorr(r, ro, "exact capture, int8 kernel")
    corner = cap.copy()
    corner[:1000:3] = (255 - 127) / 128.0 + 1j * (0 - 127) / 128.0        # codes 255 and 0
    S.xcorr_pss(corner, f[:3], 2, fc, fc, FS)
    assert S.last_xcorr_info()[0] == "k_xcorr_i8x3"
    for bad in (cap * 0.5, cap + 1e-9, np.where(np.arange(cap.size) == 77777, 129 / 128.0, cap), cap.astype(np.complex64) * (1 + 1e-7)):
        r2 = S.xcorr_pss(bad, f[:5], 2, fc, fc, FS)
        assert S.last_xcorr_info()[0].startswith("k_xcorr_mfma_blk"), S.last_xcorr_info()
        _check_xcorr(r2, O.xcorr_pss(bad, f[:5], 2, fc, fc, FS), "inexact capture, fp32 kernel")
    # the fused host chain takes the same decision, and the two kernels agree on the result
    cells, _ = S.search_capbuf(cap, f, fc, fc, FS)
    assert S.last_xcorr_info()[0] == "k_xcorr_i8x3" and [c.n_id_cell() for c in cells] == [277, 271]
    cells2, _ = S.search_capbuf(cap * (1 + 2 ** -30), f, fc, fc, FS)
    assert S.last_xcorr_info()[0].startswith("k_xcorr_mfma_blk") and [c.n_id_cell() for c in cells2] == [277, 271]
    for a, b in zip(cells, cells2):
        assert abs(a.pss_pow - b.pss_pow) < 1e-5 * b.pss_pow and a.sfn == b.sfn and abs(a.freq_superfine - b.freq_superfine) < 1e-2


def test_host_fed_batches_pipelined(pkg, capbuf_0000):
    """lcs_batch_enqueue_host (the carrier loop with captures in host memory): page-locked sources are DMA'd in place,
    ordinary memory is staged; two contexts in flight; results identical to the device-resident entry point."""
    import torch
    cap, fc = capbuf_0000
    g = golden("capbuf_0000")["iq_u8"]
    f = f_search_set_for(fc, 100)
    rng = np.random.default_rng(8)
    noise = np.clip(np.rint(rng.normal(127.0, 11.0, g.size)), 0, 255).astype(np.uint8)
    batches = [np.stack([g, noise, np.roll(g, 2 * 999)]), np.stack([noise, np.roll(g, 2 * 5000), g]), np.stack([np.roll(g, 2 * 31), g, noise])]
    fcs = np.array([fc, fc + 100e3, fc + 200e3])
    key = lambda c: tuple(v for v in c.as_dict().values() if v == v)
    with pkg.Searcher(0) as R:
        ref = []
        for b in batches:
            d = torch.from_numpy(b).cuda()
            ref.append(R.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, 3, cap.size, f, fcs, fcs, FS, pkg.STAGE_FULL))
    assert [c.n_id_cell() for c in ref[0][0]] == [277, 271]
    with pkg.Searcher(0) as A, pkg.Searcher(0) as B:
        ctx = [A, B]
        assert pkg.device_count() >= 1
        pin = [A.host_alloc(batches[0].nbytes), B.host_alloc(batches[0].nbytes)]
        for pinned in (True, False):
            got = [None] * len(batches)
            for i in range(len(batches) + 1):
                if i < len(batches):
                    src = batches[i]
                    if pinned:
                        pin[i % 2][:] = batches[i].reshape(-1)
                        src = pin[i % 2]
                    ctx[i % 2].batch_enqueue_host(src, pkg.FMT_IQ_U8, 3, cap.size, f, fcs, fcs, FS)
                if i >= 1:
                    got[i - 1] = ctx[(i - 1) % 2].batch_collect(3)
            for gb, rb in zip(got, ref):
                assert [[key(c) for c in x] for x in gb] == [[key(c) for c in x] for x in rb]
        A.host_free(pin[0]); B.host_free(pin[1])


def test_large_batch_equals_small_batches(pkg, capbuf_0000):
    """bench.py's batch shape: 128 buffers in ONE correlation launch (16 buffers per XCD queue position instead of 8; plus a
    ragged 77 = 72 through the XCD-aware mapping + 5 through the plain one).  Every buffer's cells and every collapsed array
    must equal what the same buffer gives in a batch of 8."""
    import torch
    cap, fc = capbuf_0000
    g = golden("capbuf_0000")["iq_u8"]
    f = f_search_set_for(fc, 100)
    rng = np.random.default_rng(12)
    noise = np.clip(np.rint(rng.normal(127.0, 11.0, g.size)), 0, 255).astype(np.uint8)
    base = [g, noise, np.roll(g, 2 * 999), np.roll(g, 2 * 5000), np.roll(noise, 77), np.roll(g, 2 * 31), np.roll(g, 2 * 8000), np.roll(noise, 4001)]
    key = lambda c: tuple(v for v in c.as_dict().values() if v == v)
    with pkg.Searcher(0) as S:
        fcs8 = fc + 100e3 * np.arange(8)
        d8 = torch.from_numpy(np.stack(base)).cuda()
        ref = S.search_batch(d8.data_ptr(), pkg.FMT_IQ_U8, 8, cap.size, f, fcs8, fcs8, FS, pkg.STAGE_FULL)
        ref_arr = [S.batch_readback(i, f.size) for i in range(8)]
        assert [c.n_id_cell() for c in ref[0]] == [277, 271]
        for n in (128, 77):
            order = [(5 * i + 3) % 8 for i in range(n)]
            dn = torch.from_numpy(np.stack([base[k] for k in order])).cuda()
            fcsn = fcs8[order]
            got = S.search_batch(dn.data_ptr(), pkg.FMT_IQ_U8, n, cap.size, f, fcsn, fcsn, FS, pkg.STAGE_FULL)
            assert len(got) == n
            for i, k in enumerate(order):
                assert [key(c) for c in got[i]] == [key(c) for c in ref[k]], (n, i, k)
            for i in sorted({0, 7, 8, 63, 64, 71, 72, n - 1}):
                a, b = S.batch_readback(i, f.size), ref_arr[order[i]]
                for name in ("pow", "frq", "single"):
                    assert np.array_equal(a[name], b[name]), (n, i, name)


def test_bad_arguments_fail_loudly(S, pkg, capbuf_0000):
    cap, fc = capbuf_0000
    with pytest.raises(pkg.SearcherError):
        S.xcorr_pss(cap, np.array([]), 2, fc, fc, FS)
    with pytest.raises(pkg.SearcherError):
        S.xcorr_pss(cap[:5000], np.array([0.0]), 2, fc, fc, FS)


def test_sparse_frequency_grids_are_repacked(S, pkg, capbuf_0000):
    """Grids whose hypotheses drift apart by more samples than a tap block holds -- 40 kHz, 1 MHz and 6 MHz steps: window
    starts up to 1100 samples apart -- are packed with fewer hypotheses per template group (down to one) instead of being
    rejected; host entry point (fp32 kernel) and u8 batch (int8 kernel), full arrays against the oracle."""
    cap, fc = capbuf_0000
    g = golden("capbuf_0000")
    for f in (np.arange(-4, 5) * 40e3, np.array([-2e6, -1e6, 0.0, 35e3, 1e6]), np.arange(6) * 6e6):
        ro = O.xcorr_pss(cap, f, 2, fc, fc, FS)
        r = S.xcorr_pss(cap, f, 2, fc, fc, FS)
        _check_xcorr(r, ro, f"sparse grid step {f[1] - f[0]}")
        _batch_arrays_vs_oracle(S, pkg, [g["iq_u8"]], f, np.array([fc]), 153600, f"sparse grid step {f[1] - f[0]}")


def test_operand_row_delay_limit(S, pkg):
    """The matrix-core kernels keep one operand image per template group in LDS and apply a column's window-start delay
    when they read it; the rows hold delays 0 .. 15 (LCS_I8_OFF).  At 739 MHz a 16 kHz grid reaches exactly 15 inside a
    16-column group (kept dense), 17 kHz reaches 16 (repacked with 15 columns per group: 13) and 20 kHz reaches 15 with
    15 columns: the boundary from both sides, every element of every array against the oracle, int8 and fp16 kernels."""
    g = golden("capbuf_0000")
    fc = float(g["fc"][0])
    for step in (16e3, 17e3, 20e3):
        f = np.arange(-10, 11) * step
        _batch_arrays_vs_oracle(S, pkg, [g["iq_u8"]], f, np.array([fc]), 153600, f"delay limit, grid step {step}")


def test_fp16_kernel_on_unquantised_float_sources(S, pkg):
    """complex<float> batches whose samples are NOT dongle values take the fp16 three-product kernel (samples and templates
    as fp16 hi + lo parts, 22 bits each): genuinely 24-bit float data of very different scales in one batch (each buffer
    gets its own power-of-two scale), every element against the oracle run on the same float values."""
    import torch
    fc = 739e6
    f = f_search_set_for(fc, 100)
    rng = np.random.default_rng(41)
    a, _ = pkg.synth.make_capbuf(7001, fc, [dict(n_id_1=60, n_id_2=1, f_off=12e3), dict(n_id_1=9, n_id_2=0, f_off=-48e3, gain_db=-3)], 6.0, quantise=False)
    b, _ = pkg.synth.make_capbuf(7002, fc, [dict(n_id_1=130, n_id_2=2, f_off=30e3, cp_normal=False)], 0.0, quantise=False)
    noise = 0.1 * (rng.normal(size=153600) + 1j * rng.normal(size=153600))
    bufs = [a.astype(np.complex64), (b * 3.7e-4).astype(np.complex64), (noise * 250.0).astype(np.complex64), np.zeros(153600, np.complex64)]
    bufs[3][1000:1137] = 1e-3 * O.pss_td(1).astype(np.complex64)          # an all-zero buffer but for one PSS: extreme dynamic range
    d = torch.from_numpy(np.stack(bufs)).cuda()
    fcs = np.full(4, fc)
    res = S.search_batch(d.data_ptr(), pkg.FMT_C64, 4, 153600, f, fcs, fcs, FS, pkg.STAGE_FULL, max_cells_per_buf=16)
    assert S.last_xcorr_info()[0] == "k_xcorr_f16x3"
    worst = 0.0
    for k in range(3):
        ro = O.xcorr_pss(bufs[k].astype(np.complex128), f, 2, fc, fc, FS)
        r = S.batch_readback(k, f.size)
        err = np.abs(r["single"].astype(np.float64) - ro["single"]) / ro["single"]
        worst = max(worst, float(err.max()))
        assert err.max() < RTOL, (k, err.max())
        _check_frq(r["frq"], ro, f"float buffer {k}")
        assert (np.abs(r["pow"] - ro["pow"]) / ro["pow"]).max() < RTOL
        co, _ = O.search_capbuf(bufs[k].astype(np.complex128), f, fc, fc, FS)
        assert [(c.n_id_cell(), c.ind, c.sfn) for c in res[k]] == [(c.n_id_cell(), c.ind, c.sfn) for c in co], k
    # the planted identities (60, 1), (9, 0) and (130, 2) come back decoded, at the oracle's positions
    assert [(c.n_id_cell(), c.ind, c.sfn, c.n_rb_dl) for c in res[0]] == [(181, 5211, 656, 50), (27, 5179, 503, 50)]
    assert [(c.n_id_cell(), c.ind, c.sfn, c.n_rb_dl) for c in res[1]] == [(392, 474, 158, 50)]
    # the nearly empty buffer: absolute agreement (most positions correlate to exactly zero in both)
    ro = O.xcorr_pss(bufs[3].astype(np.complex128), f, 2, fc, fc, FS)
    r = S.batch_readback(3, f.size)
    assert np.abs(r["single"].astype(np.float64) - ro["single"]).max() < 1e-6 * ro["single"].max()
    print(f"fp16 three-product kernel: worst relative deviation of xc_incoherent_single from the oracle {worst:.2e}")


@pytest.mark.parametrize("ds", [0, 1, 3])
def test_xcorr_pss_other_delay_spread_arms_without_debug_copy(S, capbuf_0000, ds):
    """ds_comb_arm != 2 with incoherent == NULL (lcs.h: "incoherent may be NULL"): the generic collapse kernel must not
    store the debug copy it was not given a buffer for (round-3 advisory: it did), and the collapsed arrays must equal
    the ones of the call that asks for the copy and the oracle's (src/searcher.cpp:312-383)."""
    cap, fc = capbuf_0000
    f = np.array([30e3, 35e3, 40e3])
    a = S.xcorr_pss(cap, f, ds, fc, fc, FS, want_incoherent=False)
    b = S.xcorr_pss(cap, f, ds, fc, fc, FS, want_incoherent=True)
    assert a["incoherent"] is None and np.array_equal(a["pow"], b["pow"]) and np.array_equal(a["frq"], b["frq"])
    ro = O.xcorr_pss(cap, f, ds, fc, fc, FS)
    assert (np.abs(a["pow"] - ro["pow"]) / ro["pow"]).max() < RTOL
    _check_frq(a["frq"], ro, f"ds_comb_arm {ds}")
    assert (np.abs(b["incoherent"].astype(np.float64) - ro["incoherent"]) / ro["incoherent"]).max() < RTOL
