"""Deterministic synthetic 1.92 Msps LTE downlink capture buffers (host-side utility).

The encode-side mirror of the searcher path, used by bench.py and by the parity tests to
produce inputs the reference's shipped vectors do not cover (extended CP, 1/2/4 antenna
ports, arbitrary timing / frequency offsets, several cells per buffer, empty buffers).
Structure follows Matlab/create_dl_sig.m:45-112 (6 RB, 128-point IDFT, DC empty, CP 10/9 or
32 samples, CRS, PSS in the last and SSS in the second-to-last symbol of slots 0 and 10,
random QPSK load) plus what that script lacks and the full chain needs: a PBCH carrying a
valid MIB (36.212 5.1.1 / 5.1.3.1 / 5.1.4.2, 36.211 6.6: CRC16 xor antenna mask, tail-biting
convolutional code, rate matching to 1920/1728 bits, cell-specific scrambling, QPSK,
SFBC / SFBC-FSTD), the receiver's crystal error (carrier offset f_off together with the
matching sample-clock stretch k_factor = (fc - f_off)/fc, ref src/searcher.cpp:18-43), AWGN
and the RTL-SDR's 8-bit quantisation (x -> clip(round(128 x + 127)), ref src/capbuf.cpp:174).

Only numpy and the product's own table functions (C ABI, CPU side) are used here.
"""
from __future__ import annotations

import numpy as np

from . import capi
import ctypes as C

N_CAP = 153600
FS = 1.92e6
_BW_IDX = {6: 0, 15: 1, 25: 2, 50: 3, 75: 4, 100: 5}
_PERM = [1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31, 0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30]


def _pss_fd(n_id_2):
    o = np.empty(62, np.complex128)
    capi.load().lcs_table_pss_fd(n_id_2, o.ctypes.data_as(C.POINTER(C.c_double)))
    return o


def _sss_fd(n_id_1, n_id_2, slot):
    o = np.empty(62, np.int32)
    capi.load().lcs_table_sss_fd(n_id_1, n_id_2, slot, o.ctypes.data_as(C.POINTER(C.c_int32)))
    return o.astype(np.float64)


def _pn(c_init, n):
    o = np.empty(n, np.uint8)
    capi.load().lcs_table_lte_pn(int(c_init), n, o.ctypes.data_as(C.POINTER(C.c_uint8)))
    return o


def crc16(bits):
    """CRC-16-CCITT (x^16+x^12+x^5+1), zero initial state, MSB first (36.212 5.1.1)."""
    poly = np.array([1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1], np.uint8)
    buf = np.concatenate([np.asarray(bits, np.uint8), np.zeros(16, np.uint8)])
    for i in range(len(bits)):
        if buf[i]:
            buf[i:i + 17] ^= poly
    return buf[-16:]


def conv_encode_tailbite(c):
    """Rate-1/3 tail-biting convolutional code, K=7, G = (133, 171, 165) octal (36.212 5.1.3.1)."""
    G = (0o133, 0o171, 0o165)
    n = len(c)
    state = 0
    for b in c[-6:]:                      # shift register starts with the last 6 information bits
        state = (state >> 1) | (int(b) << 5)
    d = np.zeros((3, n), np.uint8)
    for k in range(n):
        reg = (int(c[k]) << 6) | state
        for j in range(3):
            d[j, k] = bin(reg & G[j]).count("1") & 1
        state = reg >> 1
    return d


def conv_ratematch(d, n_e):
    """36.212 5.1.4.2: sub-block interleaver (same for the three streams), bit collection, selection."""
    D = d.shape[1]
    R = -(-D // 32)
    K = R * 32
    w = []
    for s in range(3):
        y = np.concatenate([np.full(K - D, -1, np.int64), d[s].astype(np.int64)]).reshape(R, 32)
        w.append(y[:, _PERM].T.reshape(-1))
    w = np.concatenate(w)
    out = np.empty(n_e, np.uint8)
    k = j = 0
    while k < n_e:
        if w[j] >= 0:
            out[k] = w[j]
            k += 1
        j = (j + 1) % (3 * K)
    return out


def mib_bits(n_rb_dl, phich_duration_ext, phich_res, sfn):
    """24-bit MasterInformationBlock: dl-Bandwidth(3) phich-Duration(1) phich-Resource(2) SFN MSBs(8) spare(10)."""
    b = [(_BW_IDX[n_rb_dl] >> 2) & 1, (_BW_IDX[n_rb_dl] >> 1) & 1, _BW_IDX[n_rb_dl] & 1, int(phich_duration_ext),
         (phich_res >> 1) & 1, phich_res & 1]
    sfn8 = (sfn >> 2) & 0xFF
    b += [(sfn8 >> (7 - i)) & 1 for i in range(8)]
    b += [0] * 10
    return np.array(b, np.uint8)


def pbch_symbols(n_id_cell, n_ports, n_rb_dl, phich_duration_ext, phich_res, sfn, cp_normal):
    """QPSK symbols of one 40 ms PBCH period (960 normal CP / 864 extended CP)."""
    a = mib_bits(n_rb_dl, phich_duration_ext, phich_res, sfn)
    p = crc16(a)
    if n_ports == 2:
        p = p ^ 1
    elif n_ports == 4:
        p = p ^ (np.arange(16) & 1).astype(np.uint8)
    c = np.concatenate([a, p])
    d = conv_encode_tailbite(c)
    n_e = 1920 if cp_normal else 1728
    e = conv_ratematch(d, n_e) ^ _pn(n_id_cell, n_e)
    return ((1 - 2.0 * e[0::2]) + 1j * (1 - 2.0 * e[1::2])) / np.sqrt(2.0)


def _crs(n_id_cell, slot, sym, cp_normal):
    """12 CRS values of the 6 centre RBs (36.211 6.10.1.1), same for every port."""
    c_init = (1 << 10) * (7 * (slot + 1) + sym + 1) * (2 * n_id_cell + 1) + 2 * n_id_cell + (1 if cp_normal else 0)
    c = _pn(c_init, 440).astype(np.float64)
    m = np.arange(104, 116)
    return ((1 - 2 * c[2 * m]) + 1j * (1 - 2 * c[2 * m + 1])) / np.sqrt(2.0)


def cell_waveform(n_frames, n_id_1, n_id_2, cp_normal=True, n_ports=2, n_rb_dl=50, phich_duration_ext=0,
                  phich_res=2, sfn0=0, load=0.5, rng=None, port_gains=None, per_port=False):
    """Baseband samples at the nominal 1.92 Msps, n_frames*19200 long, starting at the boundary of frame sfn0.
    per_port: the antenna ports' waveforms separately, [n_ports, n] each scaled by 1/sqrt(n_ports) and without the flat
    per-port gains -- what a channel model per (cell, port) starts from (fading_channel)."""
    rng = rng or np.random.default_rng(0)
    n_id_cell = n_id_2 + 3 * n_id_1
    n_symb = 7 if cp_normal else 6
    v_shift = n_id_cell % 6
    if per_port:
        port_gains = np.eye(n_ports)
    elif port_gains is None:
        port_gains = np.exp(2j * np.pi * rng.random(n_ports)) * (0.8 + 0.4 * rng.random(n_ports))
    port_gains = np.asarray(port_gains, np.complex128) / np.sqrt(n_ports)
    pss = _pss_fd(n_id_2)
    crs_cache = {}
    out = np.zeros((n_ports, n_frames * 19200) if per_port else n_frames * 19200, np.complex128)
    pos = 0
    pbch = None
    for fr in range(n_frames):
        sfn = (sfn0 + fr) % 1024
        if pbch is None or sfn % 4 == 0:
            pbch = pbch_symbols(n_id_cell, n_ports, n_rb_dl, phich_duration_ext, phich_res, sfn - (sfn % 4), cp_normal)
        quarter = pbch.reshape(4, -1)[sfn % 4]
        pb_i = 0
        for slot in range(20):
            for sym in range(n_symb):
                grid = np.zeros((n_ports, 72), np.complex128)
                reserved = np.zeros(72, bool)
                # cell-specific reference signals
                rs_here = {}
                if sym == 0:
                    rs_here = {0: 0, 1: 3}
                elif sym == n_symb - 3:
                    rs_here = {0: 3, 1: 0}
                elif sym == 1:
                    rs_here = {2: 3 * (slot & 1), 3: 3 + 3 * (slot & 1)}
                if rs_here:
                    key = (slot, sym)
                    if key not in crs_cache:
                        crs_cache[key] = _crs(n_id_cell, slot, sym, cp_normal)
                    for port, v in rs_here.items():
                        idx = (v + v_shift) % 6 + 6 * np.arange(12)
                        reserved[idx] = True
                        if port < n_ports:
                            grid[port, idx] = crs_cache[key]
                is_sync = (slot % 10 == 0) and sym >= n_symb - 2
                is_pbch = (slot == 1) and sym <= 3
                if is_sync:
                    seq = pss if sym == n_symb - 1 else _sss_fd(n_id_1, n_id_2, slot)
                    grid[:, :] = 0
                    grid[0, 5:67] = seq * np.sqrt(n_ports)      # sync signals go out on port 0
                elif is_pbch:
                    skip = (sym in (0, 1)) or (sym == 3 and not cp_normal)
                    sc = np.array([k for k in range(72) if not (skip and k % 3 == v_shift % 3)])
                    s = quarter[pb_i:pb_i + len(sc)]
                    pb_i += len(sc)
                    if n_ports == 1:
                        grid[0, sc] = s
                    else:
                        s0, s1 = s[0::2], s[1::2]
                        a = np.empty((2, len(sc)), np.complex128)
                        a[0, 0::2], a[0, 1::2] = s0, s1
                        a[1, 0::2], a[1, 1::2] = -np.conj(s1), np.conj(s0)
                        a /= np.sqrt(2.0)
                        if n_ports == 2:
                            grid[0, sc], grid[1, sc] = a[0], a[1]
                        else:   # SFBC-FSTD: pairs alternate between ports (0,2) and (1,3)
                            pair = (np.arange(len(sc)) // 2) & 1
                            for q, (p0, p1) in enumerate(((0, 2), (1, 3))):
                                m = pair == q
                                grid[p0, sc[m]] = a[0, m]
                                grid[p1, sc[m]] = a[1, m]
                else:
                    free = np.flatnonzero(~reserved)
                    on = free[rng.random(free.size) < load]
                    d = ((1 - 2.0 * rng.integers(0, 2, on.size)) + 1j * (1 - 2.0 * rng.integers(0, 2, on.size))) / np.sqrt(2.0)
                    grid[0, on] = d * np.sqrt(n_ports)     # single-layer data through port 0
                mix = port_gains @ grid                      # per_port: [n_ports, 72], otherwise the ports' sum [72]
                X = np.zeros(mix.shape[:-1] + (128,), np.complex128)
                X[..., 92:128] = mix[..., :36]
                X[..., 1:37] = mix[..., 36:]
                td = np.fft.ifft(X, axis=-1) * np.sqrt(128.0)
                cp = (10 if sym == 0 else 9) if cp_normal else 32
                out[..., pos:pos + cp] = td[..., -cp:]
                out[..., pos + cp:pos + cp + 128] = td
                pos += cp + 128
    assert pos == out.shape[-1]
    return out


# Multipath delay profiles of 36.101 Annex B.2 (excess tap delay [ns], relative power [dB]): Extended Pedestrian A, Extended
# Vehicular A, Extended Typical Urban.  At 1.92 Msps a sample is 521 ns: EPA stays inside one sample, EVA spreads over 4.8,
# ETU over 9.6 -- just beyond the normal cyclic prefix of 9 samples.
CHANNEL_PROFILES = {
    "EPA": ((0, 30, 70, 90, 110, 190, 410), (0.0, -1.0, -2.0, -3.0, -8.0, -17.2, -20.8)),
    "EVA": ((0, 30, 150, 310, 370, 710, 1090, 1730, 2510), (0.0, -1.5, -1.4, -3.6, -0.6, -9.1, -7.0, -12.0, -16.9)),
    "ETU": ((0, 50, 120, 200, 230, 500, 1600, 2300, 5000), (-1.0, -1.0, -1.0, 0.0, 0.0, 0.0, -3.0, -5.0, -7.0)),
}


def fading_channel(rng, w, profile, doppler_hz=0.0, fs=FS, n_sin=16):
    """One transmit waveform (nominal rate fs) through a tapped delay line with independent Rayleigh taps: tap k delays by
    profile's tau_k (a fraction of a sample in general: applied as a linear phase over the whole waveform's spectrum) and
    multiplies by a_k(t) = sqrt(P_k / n_sin) sum_m exp(j (2 pi f_d cos(alpha_m) t + phi_m)) -- a sum-of-sinusoids Jakes process
    with maximum Doppler f_d (f_d = 0: a constant complex Gaussian-like gain).  Total average power 1.
    profile: a CHANNEL_PROFILES name or (delays_ns, powers_db)."""
    delays_ns, powers_db = CHANNEL_PROFILES[profile] if isinstance(profile, str) else profile
    p = 10.0 ** (np.asarray(powers_db, np.float64) / 10.0)
    p /= p.sum()
    n = w.size
    W = np.fft.fft(w)
    fr = np.fft.fftfreq(n, 1.0 / fs)
    t = np.arange(n, dtype=np.float64) / fs
    out = np.zeros(n, np.complex128)
    for tau_ns, pk in zip(delays_ns, p):
        wk = np.fft.ifft(W * np.exp(-2j * np.pi * fr * (tau_ns * 1e-9))) if tau_ns else w
        alpha = rng.uniform(0.0, 2 * np.pi, n_sin)
        phi = rng.uniform(0.0, 2 * np.pi, n_sin)
        if doppler_hz:
            a = np.zeros(n, np.complex128)
            for am, pm in zip(alpha, phi):
                a += np.exp(1j * (2 * np.pi * doppler_hz * np.cos(am) * t + pm))
        else:
            a = np.exp(1j * phi).sum()
        out += np.sqrt(pk / n_sin) * a * wk
    return out


def frac_resample(x, u, half=24, beta=9.0):
    """x evaluated at fractional positions u (Kaiser-windowed sinc, 2*half taps)."""
    out = np.empty(u.size, np.complex128)
    k = np.arange(-half + 1, half + 1)
    for a in range(0, u.size, 16384):
        uu = u[a:a + 16384]
        base = np.floor(uu).astype(np.int64)
        frac = uu - base
        t = k[None, :] - frac[:, None]
        h = np.sinc(t) * np.i0(beta * np.sqrt(np.clip(1 - (t / half) ** 2, 0, 1))) / np.i0(beta)
        idx = base[:, None] + k[None, :]
        ok = (idx >= 0) & (idx < x.size)
        out[a:a + 16384] = np.sum(np.where(ok, x[np.clip(idx, 0, x.size - 1)], 0) * h, axis=1)
    return out


def make_signal(rng, fc, cells=(), n_cap=N_CAP, fc_programmed=None, fs_programmed=FS):
    """The noise-free part of a capture buffer: the sum of the cells' waveforms as the receiver samples them.
    -> (signal, ref_pow = sample power of the first cell's PSS/SSS symbols or None without cells, truth)."""
    n = np.arange(n_cap, dtype=np.float64)
    sig = np.zeros(n_cap, np.complex128)
    truth = []
    ref_pow = None
    for cd in cells:
        f_off = float(cd.get("f_off", 0.0))
        k_factor = (fc - f_off) / (fc if fc_programmed is None else fc_programmed)
        rate = k_factor if fs_programmed == FS else fs_programmed * k_factor / FS      # receiver samples per nominal 1.92 MHz transmitter sample
        t0 = float(cd.get("t0", rng.uniform(0, 19200)))
        n_frames = int(np.ceil((n_cap / rate + t0) / 19200)) + 1
        chan = cd.get("channel")
        w = cell_waveform(n_frames, cd["n_id_1"], cd["n_id_2"], cd.get("cp_normal", True), cd.get("n_ports", 2),
                          cd.get("n_rb_dl", 50), cd.get("phich_duration_ext", 0), cd.get("phich_res", 2),
                          cd.get("sfn0", int(rng.integers(0, 1024))), cd.get("load", 0.5), rng, cd.get("port_gains"), per_port=chan is not None)
        if chan is not None:      # an independent fading multipath channel per (cell, antenna port), 36.101 B.2
            w = sum(fading_channel(rng, wp, chan, float(cd.get("doppler_hz", 0.0))) for wp in w)
        # receiver sample n is taken at transmitter time (t0 + n / rate) nominal samples
        y = frac_resample(w, t0 + n / rate)
        y = y * np.exp(2j * np.pi * f_off * n / (fs_programmed * k_factor))
        g = 10 ** (cd.get("gain_db", 0.0) / 20)
        sig += g * y
        if ref_pow is None:
            ref_pow = g * g * 62.0 / 128.0       # sample power of a PSS/SSS symbol (62 of 128 bins, unit RE power)
        truth.append(dict(cd, t0=t0, n_id_cell=cd["n_id_2"] + 3 * cd["n_id_1"], k_factor=k_factor))
    return sig, ref_pow, truth


def add_noise_and_quantise(rng, sig, ref_pow, snr_db=10.0, rms=0.15, quantise=True, front_end=None):
    """AWGN at `snr_db` below the first cell's PSS/SSS sample power (unit noise without cells), AGC to `rms`, and the
    RTL-SDR's 8-bit quantisation (x -> clip(round(128 x + 127)), ref src/capbuf.cpp:174).
    front_end (optional dict): what a zero-IF dongle adds before its ADC -- dc (complex, in units of the signal's rms: the LO
    leakage spike at 0 Hz), iq_gain_db / iq_phase_deg (gain and quadrature error of the Q branch: an image at -f).  Hard
    clipping needs no switch: rms >= ~0.35 drives the 8-bit range into saturation."""
    n_cap = sig.size
    noise_pow = 1.0 if ref_pow is None else ref_pow / 10 ** (snr_db / 10)
    x = sig + np.sqrt(noise_pow / 2) * (rng.standard_normal(n_cap) + 1j * rng.standard_normal(n_cap))
    x *= rms / np.sqrt(np.mean(np.abs(x) ** 2))
    if front_end:
        g = 10.0 ** (float(front_end.get("iq_gain_db", 0.0)) / 20.0)
        ph = np.deg2rad(float(front_end.get("iq_phase_deg", 0.0)))
        x = x.real + 1j * g * (x.imag * np.cos(ph) + x.real * np.sin(ph))
        x = x + complex(front_end.get("dc", 0.0)) * rms
    if not quantise:
        return x
    iq = np.empty(2 * n_cap, np.uint8)
    iq[0::2] = np.clip(np.rint(128 * x.real + 127), 0, 255).astype(np.uint8)
    iq[1::2] = np.clip(np.rint(128 * x.imag + 127), 0, 255).astype(np.uint8)
    return iq


def make_capbuf(seed, fc, cells=(), snr_db=10.0, n_cap=N_CAP, rms=0.15, quantise=True, fc_programmed=None, fs_programmed=FS, front_end=None):
    """One capture buffer as the receiver would record it.

    fc is the frequency the caller asked for (fc_requested); fc_programmed / fs_programmed are what the dongle reports
    it was actually set to (ref src/CellSearch.cpp:380-390): the crystal error then is k_factor = (fc - f_off) /
    fc_programmed and the true sample rate fs_programmed * k_factor (src/searcher.cpp:147).

    cells: dicts with n_id_1, n_id_2 and optionally cp_normal, n_ports, n_rb_dl, phich_duration_ext,
    phich_res, sfn0, load, f_off (Hz, the dongle's LO error: +f_off means the cell appears f_off
    above DC), t0 (samples into the first frame), gain_db, channel ("EPA" / "EVA" / "ETU" or (delays_ns, powers_db): an
    independent Rayleigh tapped delay line per antenna port, fading_channel) with doppler_hz.  front_end: see
    add_noise_and_quantise.  Returns (iq_u8 or complex128, truth)."""
    rng = np.random.default_rng(seed)
    sig, ref_pow, truth = make_signal(rng, fc, cells, n_cap, fc_programmed, fs_programmed)
    return add_noise_and_quantise(rng, sig, ref_pow, snr_db, rms, quantise, front_end), truth


def iq_u8_to_complex(iq):
    """The receiver-side conversion (ref src/capbuf.cpp:172-181)."""
    f = np.asarray(iq, np.uint8).astype(np.float64)
    return ((f[0::2] - 127.0) / 128.0) + 1j * ((f[1::2] - 127.0) / 128.0)


def make_batch_u8(n_buf, seed, fc_list, occupied_every=4, n_distinct=8, cells_cycle=(1, 2)):
    """Synthetic sweep: every `occupied_every`-th carrier holds 1-2 cells (SNR 0..10 dB, LO error within
    +-60 kHz), the others are noise only -- like a band scan where most raster points are empty.
    To keep host-side generation quick, `n_distinct` different occupied buffers are generated and
    re-used cyclically with an independent circular time shift.  cells_cycle: cells planted in the i-th distinct
    occupied buffer, cyclically."""
    rng = np.random.default_rng(seed)
    occ = []
    for i in range(min(n_distinct, max(1, n_buf // occupied_every + 1))):
        n_cells = int(cells_cycle[i % len(cells_cycle)])
        cells = [dict(n_id_1=int(rng.integers(0, 168)), n_id_2=int(rng.integers(0, 3)), cp_normal=bool(i % 5 != 4),
                      n_ports=int((1, 2, 2, 4)[i % 4]), n_rb_dl=int((6, 15, 25, 50, 75, 100)[i % 6]),
                      f_off=float(rng.uniform(-60e3, 60e3)), gain_db=-3.0 * j) for j in range(n_cells)]
        occ.append(make_capbuf(seed * 1000 + i, float(fc_list[0]), cells, snr_db=float(rng.uniform(0, 10)))[0])
    out = np.empty((n_buf, 2 * N_CAP), np.uint8)
    k = 0
    for b in range(n_buf):
        if b % occupied_every == 0:
            out[b] = np.roll(occ[k % len(occ)], 2 * int(rng.integers(0, N_CAP)))
            k += 1
        else:
            out[b] = np.clip(np.rint(rng.normal(127.0, 19.0, 2 * N_CAP)), 0, 255).astype(np.uint8)
    return out
