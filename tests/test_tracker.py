"""LTE-Tracker's per-symbol pipeline on blocks of OFDM symbols.

CPU: the oracle's restatement of src/tracker_thread.cpp (get_fd, filter_ce, do_foe, do_toe_v2, interp2d,
pbch_extract_rt + do_mib_decode) is pinned to the reference's golden capture: symbols cut from test/capbuf_0000.it the
way the producer thread cuts them must re-decode the MIBs the searcher decodes there (cells 277 and 271, 50 RB, two
ports, PHICH normal/one -- src/CMakeLists.txt:34-35, doc/CellSearch.html:75-82), and the frequency / timing measurements
must agree with the searcher's estimates.  GPU: lcs_track_block against that oracle, array by array."""
import numpy as np
import pytest

import oracle as O
from conftest import golden, iq_u8_to_capbuf, load_pkg

FS, FC = 1.92e6, 739e6


@pytest.fixture(scope="module")
def tracked():
    """(capbuf, [(searcher record, td, late, frame_timing, freq_off)] for cells 277 and 271 of capbuf_0000)."""
    pkg = load_pkg()
    O.set_legacy(False)
    O.set_threads(8)
    cap = iq_u8_to_capbuf(golden("capbuf_0000")["iq_u8"])
    cells, _ = O.search_capbuf(cap, np.array([30e3, 35e3, 40e3]), FC, FC, FS)
    out = []
    for c in cells:
        k_factor = (FC - c.freq_superfine) / FC
        ft = c.frame_start * (30.72e6 / 16) / (FS * k_factor)          # src/searcher_thread.cpp:224 with capbuf_sync.late = 0
        td, late, ftv, fov = pkg.tracker.cut_symbols(cap, ft, c.cp_type, c.freq_superfine, FC, FC, FS, 7 * 140)
        out.append((c, td, late, ftv, fov))
    return cap, out


def _oracle_block(c, td, late, ftv, fov):
    syms, bpo, trace = O.trk_get_fd(c, td, 0, 0, fov, late, FC, FC, FS)
    r = O.trk_chan_est(c, syms, 0, 0, fov, ftv, FC, FC, FS)
    r.update(syms=syms, bpo=bpo)
    n_fr = td.shape[0] // 140
    upto = int(min(r["ce_upto"][:c.n_ports]))
    mib = []
    for o in range(n_fr - 3):
        ii = [(o + fr) * 140 + 7 + s for fr in range(4) for s in range(4)]
        if ii[-1] >= upto:
            mib.append(None)
            continue
        mib.append(O.trk_mib(c, syms[ii], r["ce"][:c.n_ports][:, ii], r["ce_pw"][:c.n_ports][:, ii, 3]))
    r["mib"] = mib
    return r


def test_producer_cut_is_symbol_aligned(tracked):
    _, cells = tracked
    pkg = load_pkg()
    for c, td, late, ftv, fov in cells:
        assert td.shape == (980, 128) and np.all(np.abs(late) < 0.5)
        # the cutter advances 137/138 LTE samples per symbol: 19200 per frame
        assert abs((late[140] - late[0])) < 0.2
    assert pkg.tracker.mib_lock_walk([False, False, True, False, False, False, True]) == (0.0, True, 4, False)
    assert pkg.tracker.mib_lock_walk([False] * 5)[0] == 1.25


def test_oracle_tracker_redecodes_the_golden_cells(tracked):
    _, cells = tracked
    pkg = load_pkg()
    assert [c.n_id_2 + 3 * c.n_id_1 for c, *_ in cells] == [277, 271]
    for c, td, late, ftv, fov in cells:
        r = _oracle_block(c, td, late, ftv, fov)
        assert list(r["n_meas"][:2]) == [278, 278] and r["n_meas"][2] == 0      # 2 reference symbols per slot and port, minus the two ends
        locked = [o for o, m in enumerate(r["mib"]) if m and m[1] and m[2]]
        assert len(locked) == 1                                                  # exactly one 40 ms alignment in a 70 ms block
        bits = r["mib"][locked[0]][0]
        sfn8 = int("".join(str(b) for b in bits[6:14]), 2)
        assert list(bits[:6]) == [0, 1, 1, 0, 1, 0]                              # 50 RB, PHICH normal / one
        assert (sfn8 * 4 - locked[0]) % 1024 == c.sfn                            # the searcher's SFN for the buffer's first frame
        others = [m for o, m in enumerate(r["mib"]) if m and o != locked[0]]
        assert others and not any(m[1] for m in others)
        # FOE / TOE measurements scatter around the searcher's estimates; folded through the reference's recurrences
        # they stay put (the 1e-6 prior weight of do_foe makes a 50 ms block move the offset by well under 1 Hz)
        m0 = r["meas"][0, :r["n_meas"][0]]
        assert abs(np.median(m0[:, 5]) - c.freq_superfine) < 40 and abs(np.median(m0[:, 7]) - ftv[0]) < 1.0
        assert abs(pkg.tracker.fold_frequency_offset(c.freq_superfine, m0) - c.freq_superfine) < 1.0
        # channel estimates: |ce| on the strong cell is stable across the block, phase continuous
        ce = r["ce"][0, :r["ce_upto"][0]]
        assert np.isfinite(ce).all() and 0.2 < np.abs(ce).mean() / np.sqrt(r["meas"][0, 5, 2]) < 2.0


@pytest.mark.gpu
def test_gpu_track_block_matches_oracle(tracked):
    """Both cells in one call (two tracked cells, 980 symbols each), every output array against the oracle."""
    pkg = load_pkg()
    _, cells = tracked
    recs = [c for c, *_ in cells]
    td = np.stack([x[1] for x in cells]); late = np.stack([x[2] for x in cells])
    ftv = np.stack([x[3] for x in cells]); fov = np.stack([x[4] for x in cells])
    with pkg.Searcher(0) as S:
        g = S.track_block(recs, td, fov, ftv, late, FC, FC, FS)
        g2 = S.track_block(recs, td, fov, ftv, late, FC, FC, FS)
    assert np.array_equal(g["syms"], g2["syms"]) and np.array_equal(g["mib_bits"], g2["mib_bits"])
    for i, (c, td_i, late_i, ft_i, fo_i) in enumerate(cells):
        r = _oracle_block(c, td_i, late_i, ft_i, fo_i)
        scale = np.abs(r["syms"]).max()
        assert np.abs(g["syms"][i] - r["syms"]).max() < 1e-11 * scale
        assert abs(g["bpo"][i] - r["bpo"]) < 1e-9
        assert np.array_equal(g["n_meas"][i], r["n_meas"]) and np.array_equal(g["ce_upto"][i], r["ce_upto"])
        for p in range(c.n_ports):
            n = r["n_meas"][p]
            gm, om = g["meas"][i, p, :n], r["meas"][p, :n]
            assert np.array_equal(gm[:, 0], om[:, 0])
            assert np.abs(gm[:, 1:5] - om[:, 1:5]).max() < 1e-11 * om[:, 2].max()
            assert np.abs(gm[:, 5] - om[:, 5]).max() < 1e-6 and np.abs(gm[:, 7] - om[:, 7]).max() < 1e-8      # Hz, samples
            assert np.abs(gm[:, 6] / om[:, 6] - 1).max() < 1e-9 and np.abs(gm[:, 8] / om[:, 8] - 1).max() < 1e-9
            u = r["ce_upto"][p]
            assert np.abs(g["ce"][i, p, :u] - r["ce"][p, :u]).max() < 1e-11 * np.abs(r["ce"][p, :u]).max()
            assert np.abs(g["ce_pw"][i, p, :u] - r["ce_pw"][p, :u]).max() < 1e-11 * np.abs(r["ce_pw"][p, :u]).max()
        for o, m in enumerate(r["mib"]):
            if m is None:
                assert g["mib_ok"][i, o] == -1
                continue
            bits, crc, fields = m
            assert g["mib_ok"][i, o] == (1 if crc else 0) | (2 if fields else 0), (i, o)
            assert [(int(g["mib_bits"][i, o]) >> k) & 1 for k in range(40)] == list(bits), (i, o)
        assert 3 in list(g["mib_ok"][i])


def test_oracle_tracker_statistics_on_the_golden_cells(tracked):
    """do_ac_fd / do_ac_td / do_pss_sss_sigpower_ce (src/tracker_thread.cpp:318-371, 754-820) on the golden capture: the
    PSS/SSS signal-to-noise measurement agrees with the searcher's picture of the two cells (277 strong, 271 weaker),
    the autocorrelations are Hermitian-consistent and normalised like the reference's."""
    _, cells = tracked
    pkg = load_pkg()
    snr = []
    for c, td, late, ftv, fov in cells:
        r = _oracle_block(c, td, late, ftv, fov)
        st = O.trk_stats(c, r["syms"], 0, 0, r["meas"], r["n_meas"])
        assert st["n_hf"] == 14                                           # 7 frames: two PSS/SSS pairs each
        sy = st["sync"][:14]
        assert np.isfinite(sy).all() and (sy[:, 0] > 0).all() and (sy[:, 2] > 0).all()
        assert np.allclose(sy[:, 1], sy[:, 0] - sy[:, 2] / 13, rtol=0, atol=1e-15)          # sp = tp - np/13
        assert np.all(st["sync_ce"][:14, :5] == 0) and np.all(st["sync_ce"][:14, 67:] == 0)
        assert np.allclose(np.mean(np.abs(st["sync_ce"][:14, 5:67]) ** 2, axis=1), sy[:, 0])
        snr.append(10 * np.log10(np.median(sy[:, 1] / sy[:, 2])))
        n = r["n_meas"][0]
        fd, tdc = st["ac_fd"][0, :n], st["ac_td"][0, :n]
        # lag 0 of both autocorrelations is the raw power of the current reference symbol over its signal power: real, >= ~1
        assert np.abs(fd[:, 0].imag).max() < 1e-12 and np.abs(tdc[71:, 0].imag).max() < 1e-12
        assert np.allclose(fd[71:, 0].real, tdc[71:, 0].real, rtol=1e-12)
        assert np.isnan(tdc[:71].real).all() and np.isfinite(tdc[71:]).all()
        assert 0.8 < np.median(fd[:, 0].real) < 3.0
        # the running averages, folded as the reference does: coherent across frequency on a line-of-sight capture
        ac = pkg.tracker.fold_ac_fd(np.zeros(12), fd, r["meas"][0, :n])
        assert abs(ac[1]) > 0.5 * abs(ac[0]) > 0
        av = pkg.tracker.fold_sync_power(None, sy)
        assert av.shape == (4,) and abs(av[0] - sy[0, 0]) < 0.02 * sy[0, 0]
    assert snr[0] > snr[1] and snr[0] > 5.0                                # cell 277 is the strong one (doc/CellSearch.html:75-82)


def _wrap_walk(b, fov, cp_normal=True):
    """get_fd's bulk phase (tracker_thread.cpp:151-153) in IEEE doubles, expression by expression: WRAP of include/macros.h."""
    import math
    sm, lg = -math.pi, math.pi
    n = lg - sm
    for i, f in enumerate(fov):
        L = (128 + 32) if not cp_normal else ((128 + 10) if i % 7 == 0 else (128 + 9))
        x = b + 2 * math.pi * L * (1 / (30720000.0 / 16)) * -float(f)
        k = x - sm
        b = (k - n * float(int(math.floor(k / n)))) + sm
    return b


@pytest.mark.gpu
@pytest.mark.parametrize("start", [0.0, 2.5, -17.25, 3.0e7])
def test_gpu_bulk_phase_walk_is_wrap_bit_for_bit(tracked, start):
    """k_trk_prep takes the turns each step folds away from a parallel prefix sum and walks with additions only; the phase it
    hands back must be the reference's sequential WRAP walk to the last bit -- from a phase inside [-pi, pi), from ones outside it
    (the first step then folds several turns), and from 3e7 (4.8 M turns: beyond the range in which the kernel trusts its turn
    count, so that chunk runs WRAP as written)."""
    import types
    pkg = load_pkg()
    _, cells = tracked
    recs = []
    for c, *_ in cells:
        recs.append(types.SimpleNamespace(bulk_phase_offset=start, **{k: int(getattr(c, k)) for k in
                    ("n_id_1", "n_id_2", "cp_type", "n_ports", "n_rb_dl", "phich_duration", "phich_resource")}))
    td = np.stack([x[1] for x in cells]); late = np.stack([x[2] for x in cells])
    ftv = np.stack([x[3] for x in cells]); fov = np.stack([x[4] for x in cells])
    assert np.abs(fov).max() > 2e4            # the golden capture's LO error: more than two turns per symbol
    with pkg.Searcher(0) as S:
        g = S.track_block(recs, td, fov, ftv, late, FC, FC, FS, want_syms=False, want_ce=False)
    for i, (c, *_rest) in enumerate(cells):
        assert int(c.cp_type) == 1            # LCS_CP_NORMAL: both golden cells
        want = _wrap_walk(start, fov[i])
        assert g["bpo"][i] == want, (i, g["bpo"][i], want)


@pytest.mark.gpu
def test_gpu_track_stats_match_oracle(tracked):
    pkg = load_pkg()
    _, cells = tracked
    recs = [c for c, *_ in cells]
    td = np.stack([x[1] for x in cells]); late = np.stack([x[2] for x in cells])
    ftv = np.stack([x[3] for x in cells]); fov = np.stack([x[4] for x in cells])
    with pkg.Searcher(0) as S:
        S.track_block(recs, td, fov, ftv, late, FC, FC, FS, want_syms=False, want_ce=False)
        g = S.track_stats(len(recs), td.shape[1])
        with pytest.raises(pkg.SearcherError):
            S.track_stats(len(recs), td.shape[1] - 140)                    # not the block that was processed
    for i, (c, td_i, late_i, ft_i, fo_i) in enumerate(cells):
        r = _oracle_block(c, td_i, late_i, ft_i, fo_i)
        st = O.trk_stats(c, r["syms"], 0, 0, r["meas"], r["n_meas"])
        assert g["n_hf"][i] == st["n_hf"] == 14
        assert np.abs(g["sync"][i, :14] - st["sync"][:14]).max() < 1e-11 * st["sync"][:14].max()
        assert np.abs(g["sync_ce"][i, :14] - st["sync_ce"][:14]).max() < 1e-11 * np.abs(st["sync_ce"][:14]).max()
        for p in range(c.n_ports):
            n = r["n_meas"][p]
            assert np.abs(g["ac_fd"][i, p, :n] - st["ac_fd"][p, :n]).max() < 1e-10 * np.abs(st["ac_fd"][p, :n]).max()
            assert np.isnan(g["ac_td"][i, p, :71].real).all()
            assert np.abs(g["ac_td"][i, p, 71:n] - st["ac_td"][p, 71:n]).max() < 1e-10 * np.abs(st["ac_td"][p, 71:n]).max()
        assert np.isnan(g["ac_fd"][i, c.n_ports:].real).all()


@pytest.mark.gpu
def test_gpu_track_block_shapes_and_errors(tracked):
    """One cell / a block too short for any MIB attempt / bad identities."""
    pkg = load_pkg()
    _, cells = tracked
    c, td, late, ftv, fov = cells[0]
    with pkg.Searcher(0) as S:
        g = S.track_block([c], td[:280], fov[:280], ftv[:280], late[:280], FC, FC, FS)
        r = O.trk_chan_est(c, O.trk_get_fd(c, td[:280], 0, 0, fov[:280], late[:280], FC, FC, FS)[0], 0, 0, fov[:280], ftv[:280], FC, FC, FS)
        assert np.array_equal(g["ce_upto"][0], r["ce_upto"]) and (g["mib_ok"] == -1).all()
        bad = pkg.new_cell(n_id_1=-1, n_id_2=1, cp_type=1, n_ports=2, n_rb_dl=50, phich_duration=1, phich_resource=3)
        with pytest.raises(pkg.SearcherError):
            S.track_block([bad], td[:280], fov[:280], ftv[:280], late[:280], FC, FC, FS)


@pytest.mark.gpu
@pytest.mark.parametrize("cuts", [(490, 490), (300, 301, 379), (140, 840), (977, 3)])
def test_gpu_track_stream_blocks_equal_one_block(tracked, cuts):
    """lcs_track_stream_block: the 980 symbols of both golden cells delivered in pieces (cut anywhere, also in the middle
    of a frame) must give, row for row and BIT FOR BIT, what one lcs_track_block call over all 980 symbols gives: the
    filter window, the channel-estimate interpolation, the 72-deep ac_td history and the four-frame MIB fifo all reach
    across the cuts (ref src/tracker_thread.cpp:176-201, 343-371, 383-477, 552-745)."""
    pkg = load_pkg()
    _, cells = tracked
    recs = [c for c, *_ in cells]
    td = np.stack([x[1] for x in cells]); late = np.stack([x[2] for x in cells])
    ftv = np.stack([x[3] for x in cells]); fov = np.stack([x[4] for x in cells])
    assert sum(cuts) == 980
    with pkg.Searcher(0) as S:
        one = S.track_block(recs, td, fov, ftv, late, FC, FC, FS)
        st = S.track_stats(2, 980)
        parts, a = [], 0
        for n in cuts:
            parts.append(S.track_stream_block(recs, td[:, a:a + n], fov[:, a:a + n], ftv[:, a:a + n], late[:, a:a + n], FC, FC, FS, want_stats=True))
            a += n
        # a second stream on the same context starts from scratch after a reset -- this time with its symbols handed over in
        # DEVICE memory (round 5: the call takes pageable, page-locked or device memory alike)
        S.track_stream_reset()
        import torch
        d0 = torch.from_numpy(np.ascontiguousarray(td[:, :cuts[0]])).cuda()
        again = S.track_stream_block(recs, None, fov[:, :cuts[0]], ftv[:, :cuts[0]], late[:, :cuts[0]], FC, FC, FS, td_device_ptr=d0.data_ptr())
        assert np.array_equal(again["syms"], parts[0]["syms"])
        assert np.array_equal(again["n_meas"], parts[0]["n_meas"]) and np.array_equal(again["meas"], parts[0]["meas"], equal_nan=True)
        # ... and a third one from page-locked host memory
        S.track_stream_reset()
        h = S.host_alloc(td[:, :cuts[0]].nbytes)
        h[:] = np.ascontiguousarray(td[:, :cuts[0]]).view(np.uint8).reshape(-1)
        third = S.track_stream_block(recs, h.view(np.complex128).reshape(2, cuts[0], 128), fov[:, :cuts[0]], ftv[:, :cuts[0]], late[:, :cuts[0]], FC, FC, FS)
        assert np.array_equal(third["syms"], parts[0]["syms"])
        S.host_free(h)
    assert np.array_equal(np.concatenate([p["syms"] for p in parts], axis=1), one["syms"])
    assert np.array_equal(parts[-1]["bpo"], one["bpo"])
    for i, c in enumerate(recs):
        for p in range(4):
            rows = np.concatenate([q["meas"][i, p, :q["n_meas"][i, p]] for q in parts])
            n = one["n_meas"][i, p]
            assert rows.shape[0] == n and np.array_equal(rows, one["meas"][i, p, :n]), (i, p)
            if n:
                fd = np.concatenate([q["ac_fd"][i, p, :q["n_meas"][i, p]] for q in parts])
                tdc = np.concatenate([q["ac_td"][i, p, :q["n_meas"][i, p]] for q in parts])
                assert np.array_equal(fd, st["ac_fd"][i, p, :n])
                assert np.array_equal(tdc[71:], st["ac_td"][i, p, 71:n]) and np.isnan(tdc[:71].real).all()
            # channel estimates: every symbol below ce_upto exactly once, in order
            pos = 0
            for q in parts:
                assert q["ce_from"][i, p] == pos
                k = q["ce_n"][i, p]
                assert np.array_equal(q["ce"][i, p, :k], one["ce"][i, p, pos:pos + k]) and np.array_equal(q["ce_pw"][i, p, :k], one["ce_pw"][i, p, pos:pos + k])
                pos += k
            assert pos == one["ce_upto"][i, p]
        ok = np.concatenate([q["mib_ok"][i, :q["n_mib"][i]] for q in parts])
        bits = np.concatenate([q["mib_bits"][i, :q["n_mib"][i]] for q in parts])
        tried = one["mib_ok"][i] != -1
        assert np.array_equal(ok, one["mib_ok"][i][tried]) and np.array_equal(bits, one["mib_bits"][i][tried]) and 3 in list(ok)
        assert [q["mib_from"][i] for q in parts] == list(np.cumsum([0] + [q["n_mib"][i] for q in parts[:-1]]))
