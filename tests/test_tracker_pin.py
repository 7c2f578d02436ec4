"""The tracker restatement of the oracle pinned to the searcher restatement.

The reference ships no vectors for src/tracker_thread.cpp, but its per-symbol front end does the same arithmetic as the
searcher's extract_tfg / chan_est, which ARE pinned to the reference's goldens (tests/test_oracle_golden.py:
Matlab/test_tfg.mat, test/capbuf_0000.it).  On the same 128 samples with the same timing error the two must therefore
agree exactly, up to the conventions each one documents:
  * get_fd (tracker_thread.cpp:91-174) rotates every symbol's frequency correction from phase 0 and carries the phase
    between symbols in bulk_phase_offset, advancing by the NOMINAL symbol length; extract_tfg (searcher.cpp:892) rotates
    the whole buffer from sample 0.  Row i of the two differs by exp(j (2 pi f loc_i / (fs k) + bulk_phase_i)).
  * get_fd removes the searcher's deliberate 2-sample early timing (:129-134; searcher.cpp:741 "-2"): a linear phase
    exp(j 2 pi 2 cn / 128) over the subcarriers cn.
  * the late-sample phase ramp exp(-j 2 pi late cn / 128) is the same in both (:158-165; searcher.cpp:923-931).
  * the raw reference-signal estimates (:868-890 vs searcher.cpp:1404-1419) and the hexagonal filter (filter_ce :176-201
    vs searcher.cpp:1431-1467) are the same sums in the same order: bit-identical on the same grid.
CPU only (the GPU side of f4 is compared with this oracle in tests/test_tracker.py)."""
import numpy as np
import pytest

import oracle as O
from conftest import golden, iq_u8_to_capbuf, load_pkg

FS, FC = 1.92e6, 739e6
CN = np.r_[np.arange(-36, 0), np.arange(1, 37)]


def _cases():
    pkg = load_pkg()
    O.set_legacy(False)
    O.set_threads(8)
    out = []
    cap = iq_u8_to_capbuf(golden("capbuf_0000")["iq_u8"])                       # normal CP, 2 ports, recorded
    cells, _ = O.search_capbuf(cap, np.array([30e3, 35e3, 40e3]), FC, FC, FS)
    out.append(("capbuf_0000/277", cap, cells[0], FC, FS))
    fcp, fsp = FC + 1234.0, FS * (1 + 2e-5)                                     # extended CP, 4 ports, fc/fs as a dongle reports them
    iq, _ = pkg.synth.make_capbuf(4243, FC, [dict(n_id_1=33, n_id_2=2, cp_normal=False, n_ports=4, n_rb_dl=15, f_off=-41e3, t0=7000.6)],
                                  12.0, fc_programmed=fcp, fs_programmed=fsp)
    cap2 = iq_u8_to_capbuf(iq)
    cells2, _ = O.search_capbuf(cap2, np.array([-45e3, -40e3, -35e3]), FC, fcp, fsp)
    assert [c.n_id_cell() for c in cells2] == [101] and cells2[0].n_ports == 4 and cells2[0].cp_type == 2
    out.append(("synthetic ext-CP 4-port", cap2, cells2[0], fcp, fsp))
    return out


@pytest.fixture(scope="module")
def cases():
    return _cases()


def test_get_fd_equals_extract_tfg_rows(cases):
    for name, cap, c, fcp, fsp in cases:
        tfg, ts = O.extract_tfg(c, cap, FC, fcp, fsp)
        n = tfg.shape[0]
        assert n == (854 if c.cp_type == 1 else 732)
        loc = np.rint(ts).astype(np.int64)
        td = np.stack([cap[l:l + 128] for l in loc])
        late = loc - ts                                                         # searcher.cpp:925-927
        f = c.freq_fine
        syms, bpo, trace = O.trk_get_fd(c, td, 0, 0, np.full(n, f), late, FC, fcp, fsp)
        k_factor = (FC - f) / fcp
        ph = 2 * np.pi * f * loc / (fsp * k_factor) + trace
        exp = tfg * np.exp(1j * ph)[:, None] * np.exp(2j * np.pi * 2 * CN / 128)[None, :]
        err = np.abs(syms - exp).max() / np.abs(exp).max()
        assert err < 1e-10, (name, err)
        # the bulk phase itself: the reference's recurrence (:152) in closed form, nominal symbol lengths
        if c.cp_type == 1:
            elapsed = np.cumsum(np.where(np.arange(n) % 7 == 0, 138, 137))
        else:
            elapsed = np.cumsum(np.full(n, 160))
        closed = np.angle(np.exp(-2j * np.pi * elapsed * f / 1.92e6))
        assert np.abs(np.angle(np.exp(1j * (trace - closed)))).max() < 1e-9, name
        assert abs(bpo - trace[-1]) == 0


def test_tracker_reference_signal_estimates_equal_chan_est(cases):
    for name, cap, c, fcp, fsp in cases:
        tfg, ts = O.extract_tfg(c, cap, FC, fcp, fsp)
        c2, grid, _ = O.tfoec(c, tfg, ts, FC, fcp)
        for port in range(c.n_ports):
            raw_s, filt_s, rows = O.chan_est_dbg(c2, grid, port)
            raw_t, idx_t, filt_t, fidx_t = O.trk_raw_filt(c2, grid, 0, 0, port)
            assert len(rows) == (244 if port < 2 else 122) and np.array_equal(idx_t, rows), (name, port)
            assert np.array_equal(raw_t, raw_s), (name, port)                  # same products: bit-identical
            # filter_ce has no first / last row (it needs both neighbours); inside, the same seven-point sums
            assert np.array_equal(fidx_t, rows[1:-1]) and np.array_equal(filt_t, filt_s[1:-1]), (name, port)
            # and the powers the tracker derives from them (:908-916)
            n_sym = grid.shape[0]
            r = O.trk_chan_est(c2, grid, 0, 0, np.zeros(n_sym), np.zeros(n_sym), FC, fcp, fsp)
            m = r["meas"][port, :r["n_meas"][port]]
            tp = np.mean(np.abs(filt_s[1:-1]) ** 2, axis=1)
            npw = np.mean(np.abs(raw_s[1:-1] - filt_s[1:-1]) ** 2, axis=1) * 7 / 6
            assert np.array_equal(m[:, 0].astype(int), rows[1:-1])
            assert np.abs(m[:, 2] / tp - 1).max() < 1e-13 and np.abs(m[:, 1] / npw - 1).max() < 1e-12, (name, port)
        # ports the cell does not have produce nothing
        if c.n_ports < 4:
            assert O.trk_chan_est(c2, grid, 0, 0, np.zeros(grid.shape[0]), np.zeros(grid.shape[0]), FC, fcp, fsp)["n_meas"][c.n_ports] == 0


def test_mib_lock_walk_accepts_raw_codes():
    """mib_ok codes are bit0 = CRC, bit1 = fields match, -1 = not attempted; only 3 is a lock
    (ref src/tracker_thread.cpp:689-694, 703-708) and the walk stops at the first attempt that was never made."""
    pkg = load_pkg()
    walk = pkg.tracker.mib_lock_walk
    assert walk(np.array([1, 2, 0, 3, -1, -1], np.int32)) == (0.0, True, 4, False)          # 1 and 2 are failures
    assert walk(np.array([1, 2, -1, 3], np.int32)) == (0.5, False, 2, False)                 # stops at -1
    assert walk(np.array([3, 0, 0, 0, 1, 0, 0, 0, 2], np.int32)) == (2.0, True, 3, False)    # synchronised failures eat 4 frames
    assert walk([False, False, True, False, False, False, True]) == (0.0, True, 4, False)    # booleans still work
    with pytest.raises(ValueError):
        walk(np.array([0.5, 1.0]))


def test_cpp_tracker_recurrences_equal_the_python_ones(tmp_path):
    """include/searcher_amd.h (lcs::track): fold_frequency_offset / fold_frame_timing / mib_lock_walk in C++, for a
    reference-side caller that links the C ABI (ref src/tracker_thread.cpp:235-242, 283-287, 552-745) -- bit-identical to
    lte-cell-scanner_amd/tracker.py on random measurement tables (no GPU: stub library symbols are never called)."""
    import subprocess
    from conftest import ROOT
    import os
    pkg = load_pkg()
    exe = tmp_path / "track_host"
    subprocess.check_call(["g++", "-O2", "-std=c++11", "-ffp-contract=off", "-o", str(exe), os.path.join(ROOT, "tests", "host", "track_host.cpp"),
                           "-L" + os.path.join(ROOT, "lte-cell-scanner_amd"), "-llcs_amd", "-Wl,-rpath," + os.path.join(ROOT, "lte-cell-scanner_amd"),
                           "-Wl,-rpath,/opt/rocm/lib"])
    rng = np.random.default_rng(3)
    for trial in range(4):
        n = int(rng.integers(1, 300))
        meas = np.zeros((n, 9))
        meas[:, 0] = np.arange(n) * 3.5
        meas[:, 5] = 35e3 + rng.normal(0, 30, n)
        meas[:, 6] = rng.uniform(1e-3, 5.0, n)
        meas[:, 7] = (15000.0 + rng.normal(0, 0.4, n) + (19190.0 if trial == 3 else 0.0)) % 19200.0
        meas[:, 8] = rng.uniform(1e-3, 0.5, n)
        codes = rng.choice(np.array([-1, 0, 1, 2, 3], np.int32), size=int(rng.integers(1, 40)), p=[0.03, 0.5, 0.1, 0.1, 0.27])
        f0, t0 = 35010.0, 14999.5
        inp = f"{f0!r} {t0!r} {n}\n" + " ".join(repr(float(v)) for v in meas.reshape(-1)) + f"\n{codes.size}\n" + " ".join(str(int(v)) for v in codes) + "\n"
        out = subprocess.run([str(exe)], input=inp, capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        got = out.stdout.split()
        assert float(got[0]) == pkg.tracker.fold_frequency_offset(f0, meas)
        assert float(got[1]) == pkg.tracker.fold_frame_timing(t0, meas)
        fl, sy, at, dr = pkg.tracker.mib_lock_walk(codes)
        assert (float(got[2]), bool(int(got[3])), int(got[4]), bool(int(got[5]))) == (fl, sy, at, dr)
