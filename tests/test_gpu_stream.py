"""Streaming mode (BASELINE configs[4] / SURVEY 8f-2: LTE-Tracker's searcher thread,
ref src/searcher_thread.cpp:83-246): the hipGraph-captured one-buffer chain must return exactly what
the eager single-buffer entry point returns for the same buffer and the same single hypothesis, push
after push, and must leave already-tracked cells undecoded.  (Graph vs the CPU oracle:
tests/test_gpu_configs.py::test_stream_graph_against_oracle.)"""
import numpy as np
import pytest

from conftest import golden, iq_u8_to_capbuf, load_pkg

pytestmark = pytest.mark.gpu
FS = 1.92e6
FC = 739e6


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def _key(c):     # both sides run the int8 correlation kernel on these byte-exact buffers; pss_pow is left out of the key anyway
    return tuple(v for k, v in c.as_dict().items() if k != "pss_pow")


def test_stream_graph_matches_eager_chain(pkg):
    g = golden("capbuf_0000")["iq_u8"]                     # cells 277 and 271 at ~35.2 kHz offset
    rng = np.random.default_rng(5)
    noise = np.clip(np.rint(rng.normal(127.0, 15.0, g.size)), 0, 255).astype(np.uint8)
    syn, _ = pkg.synth.make_capbuf(77, FC, [dict(n_id_1=12, n_id_2=1, f_off=35e3)], 8.0)
    with pkg.Searcher(0) as S, pkg.Searcher(0) as E:
        S.stream_open(pkg.FMT_IQ_U8, 153600, FC, FC, FS)
        for buf, f_off in [(g, 35e3), (noise, 35e3), (syn, 35e3), (g, 35e3), (g, 30e3)]:
            S.stream_push(buf, f_off)
            cells, dup, ms = S.stream_collect()
            ref, _ = E.search_capbuf(iq_u8_to_capbuf(buf), np.array([f_off]), FC, FC, FS)
            assert [_key(c) for c in cells] == [_key(c) for c in ref]
            assert dup == 0 and 0.0 < ms < 50.0
        # tracked cells are seen again but not decoded; the other one is still reported
        S.stream_push(g, 35e3, tracked=[277])
        cells, dup, _ = S.stream_collect()
        assert [c.n_id_cell() for c in cells] == [271] and dup == 1
        S.stream_push(g, 35e3, tracked=[271, 277, 5])
        cells, dup, _ = S.stream_collect()
        assert cells == [] and dup == 2
        S.stream_close()
        # the context is usable for batches again after closing the stream
        ref, _ = S.search_capbuf(iq_u8_to_capbuf(g), np.array([35e3]), FC, FC, FS)
        assert [c.n_id_cell() for c in ref] == [277, 271]


def test_stream_complex64_input_and_misuse(pkg):
    g = golden("capbuf_0000")["iq_u8"]
    cap = iq_u8_to_capbuf(g).astype(np.complex64)
    with pkg.Searcher(0) as S:
        with pytest.raises(RuntimeError):
            S._chk(S._lib.lcs_stream_push(S._h, None, 0.0, None, 0), "push before open")
        S.stream_open(pkg.FMT_C64, 153600, FC, FC, FS)
        S.stream_push(cap, 35e3)
        S.stream_push(cap, 30e3)                  # a second buffer may be in flight ...
        with pytest.raises(RuntimeError):
            S.stream_push(cap, 35e3)              # ... a third may not
        cells, _, _ = S.stream_collect()          # oldest first
        assert [c.n_id_cell() for c in cells] == [277, 271]
        cells, _, _ = S.stream_collect()
        assert [round(c.freq) for c in cells] == [30000] * len(cells)
        with pytest.raises(RuntimeError):
            S.stream_collect()                    # nothing in flight


def test_stream_two_buffers_in_flight_keep_their_order(pkg):
    """Double-buffered pushes (lcs_stream_push while the previous graph launch is still running): results come back in
    push order, each with its own hypothesis and tracked list, identical to one-at-a-time pushes."""
    g = golden("capbuf_0000")["iq_u8"]
    rng = np.random.default_rng(9)
    noise = np.clip(np.rint(rng.normal(127.0, 15.0, g.size)), 0, 255).astype(np.uint8)
    seq = [(g, 35e3, ()), (noise, 35e3, ()), (g, 35e3, (277,)), (np.roll(g, 2 * 700), 35e3, ()), (g, 30e3, ()), (noise, 0.0, (5,)), (g, 35e3, (271, 277))]
    with pkg.Searcher(0) as S:
        S.stream_open(pkg.FMT_IQ_U8, 153600, FC, FC, FS)
        ref = []
        for buf, f_off, tr in seq:
            S.stream_push(buf, f_off, tracked=tr)
            cells, dup, _ = S.stream_collect()
            ref.append(([_key(c) for c in cells], dup))
        got = []
        for i, (buf, f_off, tr) in enumerate(seq):
            S.stream_push(buf, f_off, tracked=tr)
            if i >= 1:
                cells, dup, ms = S.stream_collect()
                got.append(([_key(c) for c in cells], dup))
                assert 0.0 < ms < 50.0
        cells, dup, _ = S.stream_collect()
        got.append(([_key(c) for c in cells], dup))
        assert got == ref
        assert [len(x[0]) for x in ref] == [2, 0, 1, 2, 2, 0, 0] and [x[1] for x in ref] == [0, 0, 1, 0, 0, 0, 2]


def test_sweep_tool_single_gpu(tmp_path):
    """tools/sweep_cellsearch.py (BASELINE configs[3]) on a 12-carrier slice of the band: the synthetic cells
    planted on every 4th carrier come back through the full chain, the sharding driver and dedup."""
    import json, os, subprocess, sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_cellsearch.py"), "-s", "739.0e6", "-e", "740.1e6",
                          "--occupied-every", "4", "--batch", "8", "--json"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["carriers"] == 12 and r["n_gpus"] == 1 and len(r["cells"]) >= 2
    assert all(0 <= cid < 504 and nrb in (6, 15, 25, 50, 75, 100) for cid, _, nrb, _ in r["cells"])


def test_kalibrate_on_recorded_buffer(pkg):
    """LTE-Tracker's calibration step (src/LTE-Tracker.cpp:565-741) on capbuf_0000: the strongest cell is 277
    and its residual offset gives the crystal correction the CellSearch table prints for it."""
    cap = iq_u8_to_capbuf(golden("capbuf_0000")["iq_u8"])
    with pkg.Searcher(0) as S:
        best, resid, corr = pkg.kalibrate(S, cap, FC, FC, FS, ppm=120.0, correction=1.0)
        assert best.n_id_cell() == 277 and abs(resid - 35228.46) < 1.0
        assert abs(corr - FC / (FC - resid)) < 1e-15
        # a correction that moves the grid by +35 kHz finds the same cell on the shifted grid
        best2, resid2, _ = pkg.kalibrate(S, cap, FC, FC, FS, ppm=20.0, correction=1.0 + 35e3 / FC)
        assert best2.n_id_cell() == 277 and abs(resid2 - resid) < 1.0


def test_complex64_batch_on_a_context_with_an_open_stream(pkg):
    """A context whose stream graph is open cannot allocate the fp16 operand buffers (the graph holds the workspace): a
    complex<float> batch of one buffer (what fits the stream's workspace) then keeps the fp32 correlation kernel instead of
    failing (round-3 advisory), and the results are those of a fresh context's fp16 kernel."""
    import torch
    g = golden("capbuf_0000")["iq_u8"]
    cap = iq_u8_to_capbuf(g).astype(np.complex64)
    d = torch.from_numpy(cap[None, :]).cuda()
    f = np.array([35e3])
    with pkg.Searcher(0) as S, pkg.Searcher(0) as F:
        S.stream_open(pkg.FMT_C64, 153600, FC, FC, FS)
        a = S.search_batch(d.data_ptr(), pkg.FMT_C64, 1, 153600, f, np.array([FC]), np.array([FC]), FS, pkg.STAGE_FULL)[0]
        assert S.last_xcorr_info()[0].startswith("k_xcorr_mfma_blk")
        b = F.search_batch(d.data_ptr(), pkg.FMT_C64, 1, 153600, f, np.array([FC]), np.array([FC]), FS, pkg.STAGE_FULL)[0]
        assert F.last_xcorr_info()[0] == "k_xcorr_f16x3"
        assert [_key(c)[:4] for c in a] == [_key(c)[:4] for c in b] and [c.n_id_cell() for c in a] == [277, 271]
        for x, y in zip(a, b):
            assert (x.ind, x.n_id_1, x.n_ports, x.n_rb_dl, x.sfn) == (y.ind, y.n_id_1, y.n_ports, y.n_rb_dl, y.sfn)
            assert abs(x.pss_pow - y.pss_pow) < 1e-5 * y.pss_pow and abs(x.freq_superfine - y.freq_superfine) < 1e-3
        # the stream still works afterwards
        S.stream_push(cap, 35e3)
        cells, _, _ = S.stream_collect()
        assert [c.n_id_cell() for c in cells] == [277, 271]
        S.stream_close()
