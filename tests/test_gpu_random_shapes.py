"""Randomised shapes through the PSS stage and the fused chain, against the oracle.

The other suites fix the reference's shapes (153600 samples, the CLI's 5 kHz raster, ds_comb_arm = 2).  xcorr_pss itself takes ANY
capture length, any f_search_set and any delay-spread arm (src/searcher.cpp:389-419, 263-308, 311-350): here twelve seeded draws of
(buffer length from one combining window to 80 ms, 1-12 hypotheses that are neither sorted nor on a raster nor distinct from a
neighbour by more than a few hertz, arm 0-4, carriers from 700 MHz to 2.7 GHz, dongle parameters) -- every element of
xc_incoherent_single / xc_incoherent / the collapsed arrays / sp_incoherent through the host entry point, the peak list, and the cells
of the fused chain; three buffers of each shape through the batch entry point as bytes and as complex<float>."""
import os

import numpy as np
import pytest

import oracle as O
from conftest import iq_u8_to_capbuf, load_pkg
from test_gpu_pss import _check_frq, _check_xcorr

pytestmark = pytest.mark.gpu
FS = 1.92e6
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


def _draw(seed):
    rng = np.random.default_rng(7000 + seed)
    fc = float(rng.choice([700e6, 739e6, 1.8e9, 2.14e9, 2.6e9, 2.7e9])) + 100e3 * int(rng.integers(0, 5))
    n_win = int(rng.choice([1, 2, 3, 5, 8, 15]))
    n_cap = n_win * 9600 + 136 + 100 + int(rng.integers(1, 9000 if n_win < 15 else 500))
    n_f = int(rng.integers(1, 13))
    span = fc * 150e-6
    f = rng.uniform(-span, span, n_f)
    if n_f > 2:
        f[1] = f[0] + float(rng.uniform(1.0, 40.0))           # two hypotheses a few hertz apart: near-ties everywhere
    ds = int(rng.integers(0, 5))
    dongle = bool(seed % 2)
    fcp = fc * (1 + float(rng.uniform(-2e-5, 2e-5))) if dongle else fc
    fsp = FS * (1 + float(rng.uniform(-3e-5, 3e-5))) if dongle else FS
    cells = [dict(n_id_1=int(rng.integers(0, 168)), n_id_2=int(rng.integers(0, 3)), cp_normal=bool(rng.integers(0, 2)), n_ports=int(rng.choice([1, 2, 4])),
                  f_off=float(f[int(rng.integers(0, n_f))] + rng.uniform(-2e3, 2e3)), gain_db=-3.0 * j) for j in range(int(rng.integers(0, 3)))]
    return dict(fc=fc, n_cap=n_cap, f=f, ds=ds, fcp=fcp, fsp=fsp, cells=cells, n_win=n_win)


@pytest.mark.parametrize("seed", range(12))
def test_random_shape_host_entry_points(S, pkg, seed):
    d = _draw(seed)
    iq, _ = pkg.synth.make_capbuf(8000 + seed, d["fc"], d["cells"], 6.0, fc_programmed=d["fcp"], fs_programmed=d["fsp"])
    cap = iq_u8_to_capbuf(iq)[:d["n_cap"]]
    tag = f"seed {seed}: n_cap {d['n_cap']} ({d['n_win']} windows), n_f {d['f'].size}, arm {d['ds']}, fc {d['fc']:.4g}"
    r = S.xcorr_pss(cap, d["f"], d["ds"], d["fc"], d["fcp"], d["fsp"])
    ro = O.xcorr_pss(cap, d["f"], d["ds"], d["fc"], d["fcp"], d["fsp"])
    assert r["n_comb_xc"] == d["n_win"]
    _check_xcorr(r, ro, tag)
    Z = pkg.z_th1(r["sp_incoherent"], r["n_comb_xc"], d["ds"])
    Zo = O.z_th1(ro["sp_incoherent"], ro["n_comb_xc"], d["ds"])
    assert (np.abs(Z - Zo) / Zo).max() < 1e-10, tag
    pk = S.peak_search(r["pow"], r["frq"], Z, d["f"], d["fc"], d["fcp"], r["single"], d["ds"])
    po = O.peak_search(ro["pow"], ro["frq"], Zo, d["f"], d["fc"], d["fcp"], ro["single"], d["ds"])
    assert [(c.n_id_2, c.ind, c.freq) for c in pk] == [(c.n_id_2, c.ind, c.freq) for c in po], tag
    if d["n_win"] >= 8:      # the fused chain (the CLI's: arm 2, src/CellSearch.cpp:497) on buffers of 40 ms and more: four PBCH frames fit
        got, _ = S.search_capbuf(cap, d["f"], d["fc"], d["fcp"], d["fsp"])
        exp, _ = O.search_capbuf(cap, d["f"], d["fc"], d["fcp"], d["fsp"])
        assert [tuple(getattr(c, k) for k in INT_FIELDS) for c in got] == [tuple(getattr(c, k) for k in INT_FIELDS) for c in exp], tag


@pytest.mark.parametrize("seed", [1, 4, 6, 9, 10])
def test_random_shape_batches(S, pkg, seed):
    """Three buffers of the draw's length and grid (arm 2: the batch entry points are the CLI's chain) as bytes and as
    complex<float>: every element of single / pow / frq / Z_th1."""
    import torch
    d = _draw(seed)
    n = d["n_cap"]
    fcs = d["fc"] + 100e3 * np.arange(3)
    bufs = [pkg.synth.make_capbuf(8100 + 3 * seed + k, fcs[k], d["cells"][:1 + k % 2], 4.0)[0][:2 * n] for k in range(3)]
    d8 = torch.from_numpy(np.ascontiguousarray(np.stack(bufs))).cuda()
    d32 = torch.from_numpy(np.stack([iq_u8_to_capbuf(b).astype(np.complex64) for b in bufs])).cuda()
    ros = [O.xcorr_pss(iq_u8_to_capbuf(bufs[b]), d["f"], 2, fcs[b], fcs[b], FS) for b in range(3)]
    for fmt, dptr in ((pkg.FMT_IQ_U8, d8.data_ptr()), (pkg.FMT_C64, d32.data_ptr())):
        S.search_batch(dptr, fmt, 3, n, d["f"], fcs, fcs, FS, pkg.STAGE_PSS, max_cells_per_buf=pkg.MAX_PEAKS)
        for b in range(3):
            r, ro = S.batch_readback(b, d["f"].size), ros[b]
            tag = f"seed {seed} fmt {fmt} buffer {b}: n_cap {n}, n_f {d['f'].size}"
            err = np.abs(r["single"].astype(np.float64) - ro["single"]) / ro["single"]
            assert err.max() < 1e-5, (tag, err.max())
            _check_frq(r["frq"], ro, tag)
            assert (np.abs(r["pow"] - ro["pow"]) / ro["pow"]).max() < 1e-5, tag
            zo = O.z_th1(ro["sp_incoherent"], ro["n_comb_xc"])
            assert (np.abs(r["z_th1"] - zo) / zo).max() < 1e-10, tag
