"""Size-independent properties of the GPU path at BASELINE sizes (153600 samples, n_f = 31):
things that must hold exactly, whatever the data, without needing the (slow) CPU oracle."""
import numpy as np
import pytest

from conftest import golden, iq_u8_to_capbuf, f_search_set_for, load_pkg

pytestmark = pytest.mark.gpu
FS = 1.92e6
FC = 739e6


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


@pytest.fixture(scope="module")
def S(pkg):
    s = pkg.Searcher(0)
    yield s
    s.close()


@pytest.fixture(scope="module")
def synth_buf(pkg):
    iq, _ = pkg.synth.make_capbuf(4242, FC, [dict(n_id_1=41, n_id_2=2, f_off=-18e3), dict(n_id_1=100, n_id_2=0, f_off=-17e3, gain_db=-5)], 6.0)
    return iq


def test_power_of_two_scaling_is_exact(S, pkg, synth_buf):
    """xcorr is quadratic in the buffer: scaling the input by 2 scales every correlation power by
    exactly 4 (power-of-two scaling commutes with fp32/fp64 rounding) and moves no decision.  Holds within one
    correlation kernel: the halved and quartered captures are not (u8 - 127) / 128 any more, so both take the fp32
    kernel (the byte-exact capture itself would take the int8 kernel, whose results agree to ~1e-7, not bit for bit)."""
    f = f_search_set_for(FC, 100)
    cap = pkg.synth.iq_u8_to_complex(synth_buf)
    a = S.xcorr_pss(0.25 * cap, f, 2, FC, FC, FS)
    assert S.last_xcorr_info()[0].startswith("k_xcorr_mfma_blk")
    b = S.xcorr_pss(0.5 * cap, f, 2, FC, FC, FS)
    assert S.last_xcorr_info()[0].startswith("k_xcorr_mfma_blk")
    assert np.array_equal(b["single"], 4.0 * a["single"])
    assert np.array_equal(b["incoherent"], 4.0 * a["incoherent"])
    assert np.array_equal(b["pow"], 4.0 * a["pow"]) and np.array_equal(b["frq"], a["frq"])
    assert np.array_equal(b["sp_incoherent"], 4.0 * a["sp_incoherent"])
    # and across the two kernels: the byte-exact capture (int8 kernel) against 16 x the quartered one (fp32 kernel)
    c = S.xcorr_pss(cap, f, 2, FC, FC, FS)
    assert S.last_xcorr_info()[0] == "k_xcorr_i8x3"
    assert (np.abs(c["single"] - 16.0 * a["single"]) / c["single"]).max() < 2e-5
    assert np.array_equal(c["sp_incoherent"], 16.0 * a["sp_incoherent"])


def test_full_grid_is_repeatable(S, pkg, synth_buf):
    f = f_search_set_for(FC, 100)
    cap = pkg.synth.iq_u8_to_complex(synth_buf)
    a = S.xcorr_pss(cap, f, 2, FC, FC, FS, want_incoherent=False)
    a2 = S.xcorr_pss(cap, f, 2, FC, FC, FS, want_incoherent=False)
    assert np.array_equal(a["single"], a2["single"]) and np.array_equal(a["pow"], a2["pow"]) and np.array_equal(a["frq"], a2["frq"])


def test_batch_slots_are_independent_and_placement_invariant(S, pkg, synth_buf):
    """19 slots (two XCD-mapped groups of 8 + 3 plainly mapped): a buffer gives the same cells in
    whatever slot it sits, and the same as the single-buffer entry point."""
    import torch
    f = f_search_set_for(FC, 100)
    g = golden("capbuf_0000")["iq_u8"]
    rng = np.random.default_rng(11)
    noise = np.clip(np.rint(rng.normal(127.0, 15.0, g.size)), 0, 255).astype(np.uint8)
    order = [0, 1, 2, 1, 0, 2, 2, 0, 1, 0, 1, 2, 1, 1, 0, 2, 0, 2, 1]
    src = [synth_buf, g, noise]
    d = torch.from_numpy(np.stack([src[i] for i in order])).cuda()
    res = S.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, len(order), 153600, f, FC, FC, FS, pkg.STAGE_FULL)
    ref = [S.search_capbuf(iq_u8_to_capbuf(x), f, FC, FC, FS)[0] for x in src]
    key = lambda c: tuple(v for k, v in c.as_dict().items() if k != "pss_pow")   # the batch ran the int8 kernel, the host call fp32
    for slot, i in enumerate(order):
        assert [key(c) for c in res[slot]] == [key(c) for c in ref[i]], slot
        assert all(abs(a.pss_pow - b.pss_pow) <= 2e-6 * b.pss_pow for a, b in zip(res[slot], ref[i]))
    for slot, i in enumerate(order):       # and a buffer gives bit-identical records in whatever slot it sits
        first = order.index(i)
        assert [tuple(c.as_dict().values()) for c in res[slot]] == [tuple(c.as_dict().values()) for c in res[first]]
    assert sorted(c.n_id_cell() for c in ref[0]) == [125, 300] and [c.n_id_cell() for c in ref[1]] == [277, 271] and ref[2] == []


def test_u8_and_complex_float_sources_agree(S, pkg, synth_buf):
    """u8 I/Q sources take the int8 three-digit MFMA kernel (24-bit integer templates, exact integer accumulation),
    the same samples handed over as complex<float> take the fp32 MFMA kernel.  Every identity must agree and the
    correlation powers must agree far inside the 1e-5 parity bar -- on 9 slots (8 XCD-mapped + 1), through the
    whole chain, and on a 10 kHz grid whose window starts spread over many samples inside one template group."""
    import torch
    f = f_search_set_for(FC, 100)
    g = golden("capbuf_0000")["iq_u8"]
    rng = np.random.default_rng(3)
    noise = np.clip(np.rint(rng.normal(127.0, 20.0, g.size)), 0, 255).astype(np.uint8)
    bufs = [synth_buf, g, noise, np.roll(g, 2 * 4321), np.roll(synth_buf, 2 * 777), g, noise, synth_buf, g]
    d8 = torch.from_numpy(np.stack(bufs)).cuda()
    d32 = torch.from_numpy(np.stack([iq_u8_to_capbuf(b).astype(np.complex64) for b in bufs])).cuda()
    strip = lambda c: tuple(None if v != v else v for k, v in c.as_dict().items() if k != "pss_pow")   # NaN -> None
    for grid, stage, floor in ((f, pkg.STAGE_PSS, 20), (f, pkg.STAGE_FULL, 8), (np.arange(-10, 11) * 10e3, pkg.STAGE_PSS, 16)):
        a = S.search_batch(d8.data_ptr(), pkg.FMT_IQ_U8, len(bufs), 153600, grid, FC, FC, FS, stage, max_cells_per_buf=64)
        b = S.search_batch(d32.data_ptr(), pkg.FMT_C64, len(bufs), 153600, grid, FC, FC, FS, stage, max_cells_per_buf=64)
        n = 0
        for ra, rb in zip(a, b):
            assert [strip(c) for c in ra] == [strip(c) for c in rb]
            assert all(abs(ca.pss_pow - cb.pss_pow) <= 2e-6 * cb.pss_pow for ca, cb in zip(ra, rb))
            n += len(ra)
        assert n >= floor
        if stage == pkg.STAGE_FULL:
            assert [c.n_id_cell() for c in a[1]] == [277, 271]


def test_per_cell_rounds_and_overflow(S, pkg, synth_buf):
    """The per-cell stages hold lcs_set_max_cells_in_flight cells per round.  With the round size forced down to 3
    cells, a 19-buffer batch (38 cells past SSS) is decoded in 13 rounds -- 7 launched by the enqueue (one cell per
    buffer on average), 6 more by the collect once the device-side count is known -- and must return exactly what
    one big round returns."""
    import torch
    f = f_search_set_for(FC, 100)
    g = golden("capbuf_0000")["iq_u8"]
    order = [0, 1, 1, 0, 1, 0, 0, 1, 1, 0, 1, 0, 1, 1, 0, 0, 0, 1, 1]
    src = [synth_buf, g]
    d = torch.from_numpy(np.stack([src[i] for i in order])).cuda()
    ref = S.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, len(order), 153600, f, FC, FC, FS, pkg.STAGE_FULL)
    key = lambda c: tuple(c.as_dict().values())
    with pkg.Searcher(0) as S3:
        S3.set_max_cells_in_flight(3)
        res = S3.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, len(order), 153600, f, FC, FC, FS, pkg.STAGE_FULL)
        # the second batch of a context sizes its rounds and grids from the first one's cell count (round 4): same result
        res2 = S3.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, len(order), 153600, f, FC, FC, FS, pkg.STAGE_FULL)
        res3 = S3.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, 2, 153600, f, FC, FC, FS, pkg.STAGE_FULL)      # and a small batch after a busy one
    assert [[key(c) for c in r] for r in res] == [[key(c) for c in r] for r in ref]
    assert [[key(c) for c in r] for r in res2] == [[key(c) for c in r] for r in ref]
    assert [[key(c) for c in r] for r in res3] == [[key(c) for c in r] for r in ref[:2]]
    assert sum(len(r) for r in ref) == 2 * len(order)


def test_all_zero_buffer_terminates(S, pkg):
    """Degenerate input (every threshold is 0): the reference's peak_search loop never terminates;
    the GPU loop is bounded and reports overflow instead of hanging."""
    with pytest.raises(pkg.SearcherError):
        S.search_capbuf(np.zeros(153600, np.complex128), f_search_set_for(FC, 100), FC, FC, FS)


def test_ragged_buffer_lengths(S, pkg, synth_buf):
    import oracle as O
    O.set_threads(16)
    cap = pkg.synth.iq_u8_to_complex(synth_buf)
    f = np.array([-20e3, -15e3])
    for n in (150001, 144237, 60000):
        r = S.xcorr_pss(cap[:n], f, 2, FC, FC, FS, want_incoherent=False)
        ro = O.xcorr_pss(cap[:n], f, 2, FC, FC, FS)
        assert r["n_comb_xc"] == ro["n_comb_xc"] == (n - 236) // 9600 and r["n_comb_sp"] == ro["n_comb_sp"]
        assert np.array_equal(r["frq"], ro["frq"])
        assert (np.abs(r["pow"] - ro["pow"]) / ro["pow"]).max() < 1e-5
