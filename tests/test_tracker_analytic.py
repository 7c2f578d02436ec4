"""Analytic known-answer tests of the tracker's per-symbol pipeline: the reference ships no
tracker vectors, so the rows of the oracle restatement that no golden file reaches -- get_fd, do_foe, do_toe_v2, interp2d,
do_ac_fd, do_ac_td -- are pinned to CLOSED FORMS derived from the reference's own formulas (src/tracker_thread.cpp), on a
synthetic cell whose channel is known:

* a time-domain OFDM symbol built as the inverse DFT of a known 72-subcarrier row, advanced by the two samples get_fd
  takes back (:131-137), must come out of get_fd as exactly that row (:138-144, zero frequency offset / lateness);
* reference symbols  h_p * exp(j 2 pi f_res t) * exp(-j 2 pi sc tau / 128)  (port gain, residual frequency f_res, delay of
  tau samples, sc = subcarrier number) give, for every filtered reference symbol,
    - do_foe (:210-232): every element of conj(rs_prev.ce) * rs_next.ce has the phase 2 pi f_res dT, so
      residual_f = f_res * dT / 0.5 ms: f_res for ports 0/1 (their reference symbols of the same shift are one slot
      apart) and 2 f_res for ports 2/3 (two slots apart, divided by the same 0.0005 -- the reference's arithmetic);
    - do_toe_v2 (:253-271): toe1 and toe2 pair estimates three subcarriers apart (never across DC) with opposite time
      order, so the temporal phase cancels in arg(toe1) + arg(toe2) = -2 * 2 pi 3 tau / 128 and delay = tau;
* a static flat channel gives np = 0 up to the planted 1e-8 perturbation, tp = sp = |h|^2, interp2d (:402-477) = h on
  every symbol and subcarrier, ac_fd (:325-335) = 1 at all 12 lags and ac_td (:361-364) = 1 at all 72.

The same inputs go through lcs_track_block / lcs_track_stats under -m gpu."""
import numpy as np
import pytest

import oracle as O
from conftest import load_pkg

FS, FC = 1.92e6, 739e6
SC = np.concatenate([np.arange(-36, 0), np.arange(1, 37)])        # subcarrier number of column 0..71 (:139-143)


def _cell(n_id_1, n_id_2, cp_type, n_ports):
    return O.new_cell(n_id_1=n_id_1, n_id_2=n_id_2, cp_type=cp_type, n_ports=n_ports, n_rb_dl=50, phich_duration=1, phich_resource=3)


def _grid(cell, n_frames, gains, f_res=0.0, tau=0.0, eps=0.0, seed=1):
    """[n_sym][72] frequency-domain symbols carrying only the cell's reference signals through the channel above."""
    n_symb = 7 if cell.cp_type == 1 else 6
    rs, sh = O.rs_dl(cell.n_id_cell(), cell.cp_type)
    n_sym = n_frames * 20 * n_symb
    rng = np.random.default_rng(seed)
    g = np.zeros((n_sym, 72), np.complex128)
    for s in range(n_sym):
        slot, l = (s // n_symb) % 20, s % n_symb
        t = (s // n_symb) * 0.0005 + l * (0.0005 / n_symb)        # any clock that puts equal symbols of adjacent slots 0.5 ms apart
        for p in range(cell.n_ports):
            if np.isnan(sh[slot * n_symb + l, p]):
                continue
            cols = int(round(sh[slot * n_symb + l, p])) + 6 * np.arange(12)
            h = gains[p] * np.exp(2j * np.pi * f_res * t) * np.exp(-2j * np.pi * SC[cols] * tau / 128.0)
            if eps:
                h = h * (1 + eps * (rng.normal(size=12) + 1j * rng.normal(size=12)))
            g[s, cols] = h * rs[slot * n_symb + l]
    return g


def _to_td(grid):
    """Time-domain symbols whose get_fd output (zero offset, zero lateness) is `grid`."""
    D = np.zeros((grid.shape[0], 128), np.complex128)
    D[:, 1:37] = grid[:, 36:]
    D[:, 92:128] = grid[:, :36]
    # the reference's dft is unitary (include/dsp.h:34: fft / sqrt(N)); get_fd reads dft_in[t] = data[t + 2]
    return np.roll(np.fft.ifft(D, axis=1) * np.sqrt(128.0), 2, axis=1)


CASES = [  # n_id_1, n_id_2, cp_type, n_ports
    (92, 1, 1, 2), (17, 0, 1, 4), (140, 2, 2, 2), (5, 1, 2, 4),
]
GAINS = [0.7 * np.exp(0.3j), 0.45 * np.exp(-1.1j), 0.3 * np.exp(2.0j), 0.6 * np.exp(-2.5j)]


def _check_static(cell, r, st, n_sym, eps_tol=1e-6):
    for p in range(cell.n_ports):
        n = int(r["n_meas"][p])
        assert n > 20
        m = r["meas"][p, :n]
        a2 = abs(GAINS[p]) ** 2
        assert np.all(m[:, 1] < 1e-12 * a2)                                            # np: only the 1e-8 perturbation
        assert np.abs(m[:, 2] / a2 - 1).max() < eps_tol and np.abs(m[:, 4] / a2 - 1).max() < eps_tol
        assert np.abs(m[:, 5] - (-1234.5)).max() < 1e-3 and np.abs(m[:, 7] - 4321.25).max() < 1e-5
        u = int(r["ce_upto"][p])
        assert u > n_sym - 3 * 7 and np.abs(r["ce"][p, :u] - GAINS[p]).max() < eps_tol * abs(GAINS[p])      # interp2d of a constant
        assert np.abs(r["ce_pw"][p, :u, 0] / a2 - 1).max() < eps_tol
        assert np.abs(st["ac_fd"][p, :n] - 1).max() < eps_tol                         # flat channel: ac_fd == 1 at every lag
        assert np.abs(st["ac_td"][p, 71:n] - 1).max() < eps_tol                       # static channel: ac_td == 1 once the history is full
        assert np.isnan(st["ac_td"][p, :71].real).all()


def _check_moving(cell, r, f_res, tau):
    for p in range(cell.n_ports):
        n = int(r["n_meas"][p])
        m = r["meas"][p, :n]
        want_f = f_res * (1 if p < 2 else 2)          # ports 2/3: reference symbols two slots apart over the same 0.0005 (:230)
        assert np.abs(m[:, 5] - (-1234.5) - want_f).max() < 1e-8, (p, np.abs(m[:, 5] + 1234.5 - want_f).max())
        assert np.abs(m[:, 7] - 4321.25 - tau).max() < 1e-10, (p, np.abs(m[:, 7] - 4321.25 - tau).max())
        assert np.all(m[:, 6] >= .001) and np.all(m[:, 8] >= .001)


@pytest.mark.parametrize("n_id_1,n_id_2,cp_type,n_ports", CASES)
def test_oracle_tracker_closed_forms(n_id_1, n_id_2, cp_type, n_ports):
    cell = _cell(n_id_1, n_id_2, cp_type, n_ports)
    n_fr = 5
    # get_fd: the designed row comes back
    g = _grid(cell, n_fr, GAINS, eps=1e-8)
    n_sym = g.shape[0]
    z = np.zeros(n_sym)
    syms, bpo, _ = O.trk_get_fd(cell, _to_td(g), 0, 0, z, z, FC, FC, FS)
    assert np.abs(syms - g).max() < 1e-13 and bpo == 0.0
    fo, ft = np.full(n_sym, -1234.5), np.full(n_sym, 4321.25)
    r = O.trk_chan_est(cell, syms, 0, 0, fo, ft, FC, FC, FS)
    _check_static(cell, r, O.trk_stats(cell, syms, 0, 0, r["meas"], r["n_meas"]), n_sym)
    for f_res, tau in ((137.5, 0.8), (-61.0, -2.25), (12.0, 5.5)):
        g = _grid(cell, n_fr, GAINS, f_res=f_res, tau=tau)
        r = O.trk_chan_est(cell, g, 0, 0, fo, ft, FC, FC, FS)
        _check_moving(cell, r, f_res, tau)


def test_oracle_get_fd_phase_ramp_and_bulk_phase():
    """get_fd's compensation terms in closed form (:146-170): a symbol cut `late` samples late carries the ramp
    exp(+j 2 pi sc late / 128), which the exp(-j k t) factors take out again; a constant frequency offset f accumulates the
    bulk phase -2 pi f n_samp / 1.92 MHz per symbol (138 samples for symbol 0 of a slot, 137 otherwise; 160 extended)."""
    for cp_type, per_slot in ((1, 138 + 6 * 137), (2, 6 * 160)):
        cell = _cell(50, 2, cp_type, 2)
        g = _grid(cell, 1, GAINS)
        n_sym = g.shape[0]
        late = np.linspace(-0.4, 0.4, n_sym)
        ramp = np.exp(2j * np.pi * SC[None, :] * late[:, None] / 128.0)
        z = np.zeros(n_sym)
        syms, bpo, _ = O.trk_get_fd(cell, _to_td(g * ramp), 0, 0, z, late, FC, FC, FS)
        assert np.abs(syms - g).max() < 1e-13
        # frequency offset: the samples rotate by exp(j 2 pi f n / fs); get_fd derotates them (fshift by -f at fs * k_factor)
        # and applies the accumulated bulk phase: symbol s comes out as  row * exp(j bulk_s),  bulk_s = -2 pi f (samples so far) / fs
        f = 2500.0
        kf = (FC - f) / FC
        td = _to_td(g) * np.exp(2j * np.pi * f * np.arange(128)[None, :] / (FS * kf))
        syms, bpo, trace = O.trk_get_fd(cell, td, 0, 0, np.full(n_sym, f), z, FC, FC, FS)
        n_symb = 7 if cp_type == 1 else 6
        elapsed = np.cumsum([(138 if s % 7 == 0 else 137) if cp_type == 1 else 160 for s in range(n_sym)])
        want = np.angle(np.exp(-2j * np.pi * f * elapsed / 1.92e6))
        assert np.abs(np.angle(np.exp(1j * (trace - want)))).max() < 1e-9
        assert elapsed[n_symb - 1] == per_slot
        assert np.abs(syms - g * np.exp(1j * trace)[:, None]).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("n_id_1,n_id_2,cp_type,n_ports", CASES)
def test_gpu_tracker_closed_forms(n_id_1, n_id_2, cp_type, n_ports):
    """lcs_track_block / lcs_track_stats on the same synthetic cells: get_fd returns the designed rows, the measurements
    are the closed forms, the statistics of a static flat channel are identically one."""
    pkg = load_pkg()
    cell = _cell(n_id_1, n_id_2, cp_type, n_ports)
    n_fr = 5
    g = _grid(cell, n_fr, GAINS, eps=1e-8)
    n_sym = g.shape[0]
    z = np.zeros((1, n_sym))
    fo, ft = np.full((1, n_sym), -1234.5), np.full((1, n_sym), 4321.25)
    with pkg.Searcher(0) as S:
        o = S.track_block([cell], _to_td(g)[None], z, z, z, FC, FC, FS)
        assert np.abs(o["syms"][0] - g).max() < 1e-13 and o["bpo"][0] == 0.0
        # the measurements depend on the frequency-domain rows and the metadata only: feed rows that get_fd leaves alone
        # (zero offset in the derotation) but label them with the offset / timing the closed forms are written for
        o = S.track_block([cell], _to_td(g)[None], z, ft, z, FC, FC, FS)
        st = S.track_stats(1, n_sym)
        r = dict(meas=o["meas"][0].copy(), n_meas=o["n_meas"][0], ce=o["ce"][0], ce_pw=o["ce_pw"][0], ce_upto=o["ce_upto"][0])
        r["meas"][:, :, 5] += -1234.5                  # frequency_offset + residual_f with frequency_offset = 0 here
        _check_static(cell, r, dict(ac_fd=st["ac_fd"][0], ac_td=st["ac_td"][0]), n_sym)
        for f_res, tau in ((137.5, 0.8), (-61.0, -2.25), (12.0, 5.5)):
            g2 = _grid(cell, n_fr, GAINS, f_res=f_res, tau=tau)
            o = S.track_block([cell], _to_td(g2)[None], z, ft, z, FC, FC, FS)
            assert np.abs(o["syms"][0] - g2).max() < 1e-13
            r = dict(meas=o["meas"][0].copy(), n_meas=o["n_meas"][0])
            r["meas"][:, :, 5] += -1234.5
            _check_moving(cell, r, f_res, tau)
