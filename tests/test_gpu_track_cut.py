"""lcs_track_cut on the GPU: the producer thread's symbol extraction (src/producer_thread.cpp:96-131, 196-246) for many tracked cells
on ONE capture buffer resident in HBM, against the host cutter (lte-cell-scanner_amd/tracker.py cut_symbols -- the definition the
tracker tests and host/TrackCells.cpp share; the closed form itself is pinned on the CPU by tests/test_track_cut_host.py).
Every symbol's 128 samples and its `late`, bit for bit, for the three source formats; then the tracker block on the device-resident
symbols equals the block on the host-cut ones."""
import numpy as np
import pytest

import oracle as O
from conftest import golden, iq_u8_to_capbuf, load_pkg

pytestmark = pytest.mark.gpu
FS, FC = 1.92e6, 739e6


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def _gpu_cut(pkg, S, src, fmt, n_cap, cps, fts, fos, fcp, fsp, n_sym):
    import torch
    d = torch.from_numpy(src).cuda()
    td = torch.empty((len(cps), n_sym, 128), dtype=torch.complex128, device="cuda")
    late, n_cut = S.track_cut(d.data_ptr(), fmt, n_cap, cps, fts, fos, FC, fcp, fsp, n_sym, td.data_ptr())
    return td, late, n_cut


def test_cut_equals_the_host_cutter_for_every_format(pkg):
    """Twelve 'cells' (both CP types, frame timings over the whole frame incl. one that puts symbol 0's window on the buffer's
    first samples, frequency offsets to +-50 kHz) on the reference's recorded capture, as dongle bytes, complex<float> and
    complex<double>; and under dongle parameters (fc_programmed / fs_programmed off nominal)."""
    iq = golden("capbuf_0000")["iq_u8"]
    cap = iq_u8_to_capbuf(iq)
    rng = np.random.default_rng(5)
    cps = [1 + k % 2 for k in range(12)]
    fts = [float(rng.uniform(0, 19200)) for _ in range(12)]
    fts[3] = 19200.0 - 32.0 + 1.7          # extended CP: target 32 -> the window covers sample 2
    fts[4] = 19200.0 - 10.0 - 0.3          # normal CP: |tdiff| < 0.5 at sample 0
    fos = [float(rng.uniform(-50e3, 50e3)) for _ in range(12)]
    n_sym = 1000                            # more than a normal-CP buffer holds after the first frame boundary for some timings
    with pkg.Searcher(0) as S:
        for fcp, fsp in ((FC, FS), (FC * (1 + 14e-6), FS * (1 - 27e-6))):
            want = [pkg.tracker.cut_symbols(cap, fts[i], cps[i], fos[i], FC, fcp, fsp, n_sym) for i in range(12)]
            for fmt, src in ((pkg.FMT_IQ_U8, iq), (pkg.FMT_C64, cap.astype(np.complex64)), (pkg.FMT_C128, cap)):
                td, late, n_cut = _gpu_cut(pkg, S, src, fmt, cap.size, cps, fts, fos, fcp, fsp, n_sym)
                tdh = td.cpu().numpy()
                for i, (w_td, w_late, _, _) in enumerate(want):
                    n = w_td.shape[0]
                    assert n_cut[i] == n and n >= 800, (fmt, i, n_cut[i], n)
                    assert np.array_equal(late[i, :n], w_late), (fmt, i)
                    assert np.array_equal(tdh[i, :n], w_td), (fmt, i)                  # (u8 - 127) / 128 is exact in all three
                    assert not tdh[i, n:].any() and not late[i, n:].any()


def test_cut_walks_a_cell_whose_sample_rate_breaks_the_closed_form(pkg):
    """fs_programmed 7 % / 20 % below 1.92 MHz: a 128-sample capture then runs into (or past) the next symbol's window, the closed
    form's premise fails and the kernel walks the cell sample by sample as the host does: the same symbols all the same."""
    cap = iq_u8_to_capbuf(golden("capbuf_0000")["iq_u8"])[:60000]
    with pkg.Searcher(0) as S:
        for fsp in (FS * 0.93, FS * 0.8):
            cps, fts, fos = [1, 2, 1], [100.25, 9000.5, 19100.0], [1e3, -2e3, 0.0]
            td, late, n_cut = _gpu_cut(pkg, S, cap, pkg.FMT_C128, cap.size, cps, fts, fos, FC, fsp, 300)
            tdh = td.cpu().numpy()
            for i in range(3):
                w_td, w_late, _, _ = pkg.tracker.cut_symbols(cap, fts[i], cps[i], fos[i], FC, FC, fsp, 300)
                n = w_td.shape[0]
                assert n_cut[i] == n and n >= 3, (fsp, i, n)
                assert np.array_equal(late[i, :n], w_late) and np.array_equal(tdh[i, :n], w_td), (fsp, i)


def test_cut_continues_a_stream_over_several_buffers(pkg):
    """The producer's state between two blocks of samples, on the device: the recorded capture handed over as four overlapping
    buffers (sub-ranges of the bytes in HBM), each call continuing with ts_first / sym_first / pos_first from the previous call's
    n_cut / pos_next -- per buffer equal to the host cutter given the same state (samples and `late` bit for bit), and together the
    symbols of the capture cut in one go."""
    import torch
    iq = golden("capbuf_0000")["iq_u8"]
    cap = iq_u8_to_capbuf(iq)
    d = torch.from_numpy(np.ascontiguousarray(iq)).cuda()
    cps, fts, fos = [1, 2, 1], [4321.75, 18000.5, 77.0], [12e3, -7e3, 31e3]
    fsp = FS * (1 - 2e-5)
    steps = [(30.72e6 / 16) / (fsp * ((FC - fo) / FC)) for fo in fos]
    whole = [pkg.tracker.cut_symbols(cap, fts[i], cps[i], fos[i], FC, FC, fsp, 10 ** 6) for i in range(3)]
    n_sym = 400
    td = torch.empty((3, n_sym, 128), dtype=torch.complex128, device="cuda")
    got = [[], [], []]
    with pkg.Searcher(0) as S:
        o, ts0 = 0, 0.0
        sym, pos = np.zeros(3, np.int64), np.zeros(3, np.int64)
        while True:
            n = min(45000, cap.size - o)
            late, n_cut, pos_next = S.track_cut(d.data_ptr() + 2 * o, pkg.FMT_IQ_U8, n, cps, fts, fos, FC, FC, fsp, n_sym, td.data_ptr(), ts_first=ts0,
                                                sym_first=sym, pos_first=pos, want_state=True)
            tdh = td.cpu().numpy()
            for i in range(3):
                w = pkg.tracker.cut_symbols(cap[o:o + n], fts[i], cps[i], fos[i], FC, FC, fsp, n_sym, ts_first=ts0, sym_first=int(sym[i]), pos_first=int(pos[i]),
                                            want_state=True)
                assert n_cut[i] == w[0].shape[0] and pos_next[i] == w[4], (o, i)
                assert np.array_equal(tdh[i, :n_cut[i]], w[0]) and np.array_equal(late[i, :n_cut[i]], w[1]), (o, i)
                got[i].append(tdh[i, :n_cut[i]].copy())
            if o + n >= cap.size:
                break
            sym = sym + n_cut
            adv = int(min(pos_next.min(), n - 300))                         # one stream: every cell continues from the same sample
            # (one timestamp base per stream: the reference's producer steps it with the GLOBAL frequency offset; here the three
            # 'cells' carry different offsets, so each gets the base of its own step -- the first cell's is used for all, the others'
            # targets simply sit elsewhere: host and device are given the same state either way)
            ts0 = float(pkg.tracker.wrap(ts0 + adv * steps[0], 0.0, 19200.0))
            pos = pos_next - adv
            o += adv
    # cell 0 (whose step stamped the buffers): the chunks together are the capture cut in one go
    all0 = np.concatenate(got[0])
    assert all0.shape == whole[0][0].shape and np.array_equal(all0, whole[0][0])


def test_track_block_on_device_cut_symbols_equals_host_cut(pkg):
    """The two cells of the recorded capture: symbols cut on the device from the dongle's BYTES and handed to lcs_track_block as a
    device pointer -- every output array equal to the block on the host-cut symbols (which tests/test_tracker.py holds against the
    oracle), bit for bit."""
    import torch
    iq = golden("capbuf_0000")["iq_u8"]
    cap = iq_u8_to_capbuf(iq)
    O.set_legacy(False)
    O.set_threads(8)
    cells, _ = O.search_capbuf(cap, np.array([30e3, 35e3, 40e3]), FC, FC, FS)
    assert len(cells) == 2
    n_sym = 980
    fts = [c.frame_start * (30.72e6 / 16) / (FS * ((FC - c.freq_superfine) / FC)) for c in cells]
    fos = [c.freq_superfine for c in cells]
    host = [pkg.tracker.cut_symbols(cap, fts[i], cells[i].cp_type, fos[i], FC, FC, FS, n_sym) for i in range(2)]
    with pkg.Searcher(0) as S:
        td, late, n_cut = _gpu_cut(pkg, S, iq, pkg.FMT_IQ_U8, cap.size, [c.cp_type for c in cells], fts, fos, FC, FS, n_sym)
        assert list(n_cut) == [n_sym, n_sym]
        fov = np.repeat(np.array(fos)[:, None], n_sym, 1)
        ftv = np.repeat(np.array(fts)[:, None], n_sym, 1)
        g = S.track_block(cells, None, fov, ftv, late, FC, FC, FS, td_device_ptr=td.data_ptr(), n_sym=n_sym)
        h = S.track_block(cells, np.stack([x[0] for x in host]), fov, ftv, np.stack([x[1] for x in host]), FC, FC, FS)
    for key in ("syms", "n_meas", "ce_upto", "mib_ok", "mib_bits", "bpo"):
        assert np.array_equal(g[key], h[key]), key
    for i in range(2):
        for p in range(cells[i].n_ports):
            n, u = h["n_meas"][i, p], h["ce_upto"][i, p]
            assert np.array_equal(g["meas"][i, p, :n], h["meas"][i, p, :n]) and np.array_equal(g["ce"][i, p, :u], h["ce"][i, p, :u])
    assert 3 in list(g["mib_ok"][0]) and 3 in list(g["mib_ok"][1])
    del torch


def test_cut_rejects_bad_arguments(pkg):
    import torch
    d = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    td = torch.empty((1, 4, 128), dtype=torch.complex128, device="cuda")
    with pkg.Searcher(0) as S:
        for kw in (dict(fmt=7), dict(cps=[0]), dict(n_cap=100), dict(fts=[float("nan")]), dict(fsp=-1.0)):
            a = dict(fmt=pkg.FMT_IQ_U8, cps=[1], n_cap=2048, fts=[0.0], fsp=FS)
            a.update(kw)
            with pytest.raises(pkg.SearcherError):
                S.track_cut(d.data_ptr(), a["fmt"], a["n_cap"], a["cps"], a["fts"], [0.0], FC, FC, a["fsp"], 4, td.data_ptr())
        late, n_cut = S.track_cut(d.data_ptr(), pkg.FMT_IQ_U8, 2048, [1], [0.0], [0.0], FC, FC, FS, 4, td.data_ptr())
        assert n_cut[0] == 4 and np.all(np.abs(late[0]) < 0.5)


def test_device_tracker_from_bytes_in_chunks(pkg):
    """tracker.DeviceTracker: the producer + tracker data path with the samples on the device.  The recorded capture's BYTES pushed
    as five chunks (cut anywhere, not at symbol or frame boundaries), both cells tracked open loop (feedback off): the measurement
    rows and MIB attempts that come out over the pushes are those of the host path on the whole capture -- symbols cut by
    tracker.cut_symbols, ONE Searcher.track_stream_block call -- to 1e-9 (the chunks' timestamp origins round differently: `late`
    moves by 1e-12 samples); then closed loop: both cells lock and the global frequency offset stays with the searcher's estimate."""
    iq = golden("capbuf_0000")["iq_u8"]
    cap = iq_u8_to_capbuf(iq)
    O.set_legacy(False)
    O.set_threads(8)
    cells, _ = O.search_capbuf(cap, np.array([30e3, 35e3, 40e3]), FC, FC, FS)
    cells = [pkg.new_cell(**c.as_dict()) for c in cells]
    fo = float(cells[0].freq_superfine)                                    # one crystal: the global frequency offset
    fts = [c.frame_start * (30.72e6 / 16) / (FS * ((FC - fo) / FC)) for c in cells]
    # the host path on the whole capture
    host = [pkg.tracker.cut_symbols(cap, fts[i], cells[i].cp_type, fo, FC, FC, FS, 10 ** 6) for i in range(2)]
    n_all = min(h[0].shape[0] for h in host)
    with pkg.Searcher(0) as S:
        S.track_stream_reset()
        ref = S.track_stream_block(cells, np.stack([h[0][:n_all] for h in host]), np.full((2, n_all), fo), np.repeat(np.array(fts)[:, None], n_all, 1),
                                   np.stack([h[1][:n_all] for h in host]), FC, FC, FS, want_syms=False, want_ce=False)
    with pkg.Searcher(0) as S:
        T = pkg.tracker.DeviceTracker(S, cells, fts, fo, FC, FC, FS, feedback=False)
        cuts = [0, 51234, 120001, 170002, 260000, iq.size]                 # bytes; even offsets = whole samples
        meas = [[], []]
        mib = [[], []]
        for a, b in zip(cuts[:-1], cuts[1:]):
            o = T.push(iq[a:b])
            if o is None:
                continue
            for i in range(2):
                meas[i].append(o["meas"][i, 0, :o["n_meas"][i, 0]])
                mib[i] += [int(v) for v in o["mib_ok"][i, :o["n_mib"][i]]]
        assert n_all - 2 <= T.symbols_done <= n_all      # (the block moves both cells by the same number of symbols: the later cell's last one may wait)
        for i in range(2):
            got = np.concatenate(meas[i])
            want = ref["meas"][i, 0, :ref["n_meas"][i, 0]]
            n = got.shape[0]
            assert n >= want.shape[0] - 2 and np.array_equal(got[:, 0], want[:n, 0])
            assert np.abs(got[:, 1:5] - want[:n, 1:5]).max() < 1e-9 * want[:, 2].max()
            assert np.abs(got[:, 5] - want[:n, 5]).max() < 1e-4 and np.abs(got[:, 7] - want[:n, 7]).max() < 1e-6              # Hz, samples
            assert mib[i] == [int(v) for v in ref["mib_ok"][i, :ref["n_mib"][i]]][:len(mib[i])] and 3 in mib[i]
            assert T.lock[i][1], i                                          # synchronised
    with pkg.Searcher(0) as S:
        T = pkg.tracker.DeviceTracker(S, cells, fts, fo, FC, FC, FS, feedback=True)
        for a, b in zip(cuts[:-1], cuts[1:]):
            T.push(iq[a:b])
        assert all(l[1] for l in T.lock)
        assert abs(T.frequency_offset - fo) < 200.0 and np.all(np.abs(pkg.tracker.wrap(T.frame_timing - np.array(fts), -9600.0, 9600.0)) < 2.0)
