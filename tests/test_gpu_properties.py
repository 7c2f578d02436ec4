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
    exactly 4 (power-of-two scaling commutes with fp32/fp64 rounding) and moves no decision."""
    f = f_search_set_for(FC, 100)
    cap = pkg.synth.iq_u8_to_complex(synth_buf)
    a = S.xcorr_pss(cap, f, 2, FC, FC, FS)
    b = S.xcorr_pss(2.0 * cap, f, 2, FC, FC, FS)
    assert np.array_equal(b["single"], 4.0 * a["single"])
    assert np.array_equal(b["incoherent"], 4.0 * a["incoherent"])
    assert np.array_equal(b["pow"], 4.0 * a["pow"]) and np.array_equal(b["frq"], a["frq"])
    assert np.array_equal(b["sp_incoherent"], 4.0 * a["sp_incoherent"])


def test_full_grid_mfma_equals_valu_and_is_repeatable(S, pkg, synth_buf):
    f = f_search_set_for(FC, 100)
    cap = pkg.synth.iq_u8_to_complex(synth_buf)
    S.set_xcorr_variant(0)
    a = S.xcorr_pss(cap, f, 2, FC, FC, FS, want_incoherent=False)
    a2 = S.xcorr_pss(cap, f, 2, FC, FC, FS, want_incoherent=False)
    S.set_xcorr_variant(1)
    b = S.xcorr_pss(cap, f, 2, FC, FC, FS, want_incoherent=False)
    S.set_xcorr_variant(0)
    assert np.array_equal(a["single"], a2["single"]) and np.array_equal(a["pow"], a2["pow"])
    assert np.array_equal(a["single"], b["single"]) and np.array_equal(a["frq"], b["frq"])


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


def test_bf16_three_term_correlation_matches_fp32(S, pkg, synth_buf):
    """u8 I/Q sources take the int8 three-digit MFMA kernel (variant 0: 24-bit integer templates, exact
    integer accumulation) or the bf16 three-term kernel (variant 4: exact products, fp32 accumulation);
    variant 3 forces the fp32 kernel on the same device-resident bytes.  Every identity must agree and the
    correlation powers must agree far inside the 1e-5 parity bar."""
    import torch
    f = f_search_set_for(FC, 100)
    g = golden("capbuf_0000")["iq_u8"]
    rng = np.random.default_rng(3)
    noise = np.clip(np.rint(rng.normal(127.0, 20.0, g.size)), 0, 255).astype(np.uint8)
    bufs = [synth_buf, g, noise, np.roll(g, 2 * 4321), np.roll(synth_buf, 2 * 777), g, noise, synth_buf, g]   # 8 XCD-mapped + 1
    d = torch.from_numpy(np.stack(bufs)).cuda()
    out = {}
    for v in (0, 4, 3):
        S.set_xcorr_variant(v)
        out[v] = S.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, len(bufs), 153600, f, FC, FC, FS, pkg.STAGE_PSS, max_cells_per_buf=64)
    S.set_xcorr_variant(0)
    n = 0
    for v in (0, 4):
        for a, b in zip(out[v], out[3]):
            assert [(c.n_id_2, c.ind, c.freq) for c in a] == [(c.n_id_2, c.ind, c.freq) for c in b], v
            for ca, cb in zip(a, b):
                assert abs(ca.pss_pow - cb.pss_pow) <= 2e-6 * cb.pss_pow, v
                n += 1
    assert n >= 20
    # and the decoded cells of the full chain are the same records except for that last-digit power
    full = {}
    for v in (0, 4, 3):
        S.set_xcorr_variant(v)
        full[v] = S.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, len(bufs), 153600, f, FC, FC, FS, pkg.STAGE_FULL)
    S.set_xcorr_variant(0)
    strip = lambda c: tuple(v for k, v in c.as_dict().items() if k != "pss_pow")
    assert [[strip(c) for c in r] for r in full[0]] == [[strip(c) for c in r] for r in full[3]]
    assert [[strip(c) for c in r] for r in full[4]] == [[strip(c) for c in r] for r in full[3]]
    assert [c.n_id_cell() for c in full[0][1]] == [277, 271]
    # a 10 kHz grid spreads the window starts of one template group over more than 7 samples: more than 9
    # 16-tap blocks per window, which takes the looping bf16 kernel instead of the unrolled one (the int8
    # kernel's five 32-tap blocks still hold it)
    f10 = np.arange(-10, 11) * 10e3
    wide = {}
    for v in (0, 4, 3):
        S.set_xcorr_variant(v)
        wide[v] = S.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, len(bufs), 153600, f10, FC, FC, FS, pkg.STAGE_PSS, max_cells_per_buf=64)
    S.set_xcorr_variant(0)
    m = 0
    for v in (0, 4):
        for a, b in zip(wide[v], wide[3]):
            assert [(c.n_id_2, c.ind, c.freq) for c in a] == [(c.n_id_2, c.ind, c.freq) for c in b], v
            assert all(abs(ca.pss_pow - cb.pss_pow) <= 2e-6 * cb.pss_pow for ca, cb in zip(a, b)), v
            m += len(a)
    assert m >= 16


def test_bf16_kernel_short_buffers_and_single_hypothesis(S, pkg, synth_buf):
    """The bf16 kernel on shapes away from the benchmark's: 135360-sample buffers (14 combining windows, the
    Matlab/test_xcorr_pss.mat length), a single frequency hypothesis (one template group, 3 of its 16 columns
    used) and a 3-entry grid -- always against the fp32 kernel on the same device-resident bytes."""
    import torch
    g = golden("capbuf_0000")["iq_u8"]
    n_short = 135360
    bufs = np.stack([g[:2 * n_short], synth_buf[:2 * n_short], np.roll(g, 2 * 999)[:2 * n_short]])
    d = torch.from_numpy(np.ascontiguousarray(bufs)).cuda()
    for f in (np.array([35e3]), np.array([30e3, 35e3, 40e3]), f_search_set_for(FC, 100)):
        out = {}
        for v in (0, 4, 3):
            S.set_xcorr_variant(v)
            out[v] = S.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, 3, n_short, f, FC, FC, FS, pkg.STAGE_PSS, max_cells_per_buf=64)
        S.set_xcorr_variant(0)
        for v in (0, 4):
            for a, b in zip(out[v], out[3]):
                assert [(c.n_id_2, c.ind, c.freq) for c in a] == [(c.n_id_2, c.ind, c.freq) for c in b], v
                assert all(abs(ca.pss_pow - cb.pss_pow) <= 2e-6 * cb.pss_pow for ca, cb in zip(a, b)), v
        assert len(out[0][0]) >= 1


def test_per_cell_rounds_and_overflow(S, pkg, synth_buf, monkeypatch):
    """The per-cell stages take LCS_MAX_WORK cells per round.  With the round size forced down to 3
    cells, a 19-buffer batch (38 cells past SSS) is decoded in 13 non-empty rounds (plus empty ones)
    and must return exactly what one big round returns."""
    import torch
    f = f_search_set_for(FC, 100)
    g = golden("capbuf_0000")["iq_u8"]
    order = [0, 1, 1, 0, 1, 0, 0, 1, 1, 0, 1, 0, 1, 1, 0, 0, 0, 1, 1]
    src = [synth_buf, g]
    d = torch.from_numpy(np.stack([src[i] for i in order])).cuda()
    ref = S.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, len(order), 153600, f, FC, FC, FS, pkg.STAGE_FULL)
    key = lambda c: tuple(c.as_dict().values())
    monkeypatch.setenv("LCS_MAX_WORK", "3")
    with pkg.Searcher(0) as S3:
        res = S3.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, len(order), 153600, f, FC, FC, FS, pkg.STAGE_FULL)
    assert [[key(c) for c in r] for r in res] == [[key(c) for c in r] for r in ref]
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


_MODE_SCRIPT = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from conftest import golden, f_search_set_for, load_pkg
import torch
pkg = load_pkg()
g = golden("capbuf_0000")["iq_u8"]
bufs = [g, np.roll(g, 2 * 4321), g[::-1].copy()]
d = torch.from_numpy(np.stack(bufs)).cuda()
out = []
with pkg.Searcher(0) as S:
    for f in (f_search_set_for(739e6, 100), np.arange(-10, 11) * 10e3):
        r = S.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, len(bufs), 153600, f, 739e6, 739e6, 1.92e6, pkg.STAGE_PSS, max_cells_per_buf=64)
        out.append([[(c.n_id_2, c.ind, c.freq, float(c.pss_pow).hex()) for c in b] for b in r])
print("RESULT" + json.dumps(out))
"""


def test_int8_kernel_modes_are_bit_identical():
    """The int8 correlation is exact integer arithmetic with one fixed recombination, so the default kernel
    (LDS-DMA staging, prefetched operands), the gathering variant (operands fetched from the compact digit
    table through per-lane DMA addresses) and the first, register-staged kernel must return the same bits --
    on the bench grid and on a 10 kHz grid whose window starts spread over many samples."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for mode in ("", "gather", "rs"):
        env = dict(os.environ)
        env.pop("LCS_I8_KERNEL", None)
        if mode:
            env["LCS_I8_KERNEL"] = mode
        p = subprocess.run([sys.executable, "-c", _MODE_SCRIPT, here], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][-1]
        res[mode] = json.loads(line[len("RESULT"):])
    assert sum(len(b) for grid in res[""] for b in grid) >= 10
    assert res["gather"] == res[""]
    assert res["rs"] == res[""]
