"""ctypes binding of the CPU oracle (oracle/liblcs_oracle.so).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblcs_oracle.so")


class Cell(C.Structure):
    """POD mirror of the reference's ``class Cell`` (include/common.h.in:101-129)."""
    _fields_ = [
        ("fc_requested", C.c_double), ("fc_programmed", C.c_double), ("pss_pow", C.c_double),
        ("freq", C.c_double), ("frame_start", C.c_double), ("freq_fine", C.c_double),
        ("freq_superfine", C.c_double),
        ("ind", C.c_int32), ("n_id_2", C.c_int32), ("n_id_1", C.c_int32), ("cp_type", C.c_int32),
        ("n_ports", C.c_int32), ("n_rb_dl", C.c_int32), ("phich_duration", C.c_int32),
        ("phich_resource", C.c_int32), ("sfn", C.c_int32), ("reserved", C.c_int32),
    ]

    def n_id_cell(self) -> int:
        return self.n_id_2 + 3 * self.n_id_1 if (self.n_id_1 >= 0 and self.n_id_2 >= 0) else -1

    def as_dict(self) -> dict:
        return {f: getattr(self, f) for f, _ in self._fields_ if f != "reserved"}

    def __repr__(self) -> str:  # pragma: no cover
        return "Cell(" + ", ".join(f"{k}={v}" for k, v in self.as_dict().items()) + ")"


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "lcs_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        dp, ip, fp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_float)
        cp = C.POINTER(Cell)
        L.orc_chi2cdf_inv.restype = C.c_double
        L.orc_chi2cdf_inv.argtypes = [C.c_double, C.c_double]
        L.orc_xcorr_pss.argtypes = [dp, C.c_uint32, dp, C.c_uint32, C.c_uint32, C.c_double, C.c_double,
                                    C.c_double, dp, ip, fp, fp, dp, fp, dp,
                                    C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]
        L.orc_peak_search.argtypes = [dp, ip, dp, dp, C.c_uint32, C.c_double, C.c_double, fp, C.c_uint32,
                                      cp, C.c_int, C.POINTER(C.c_int)]
        L.orc_sss_detect.argtypes = [cp, dp, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_double, cp,
                                     dp, dp, dp, dp, dp, dp, dp, dp]
        L.orc_pss_sss_foe.argtypes = [cp, dp, C.c_uint32, C.c_double, C.c_double, C.c_double, cp]
        L.orc_extract_tfg.argtypes = [cp, dp, C.c_uint32, C.c_double, C.c_double, C.c_double, dp, dp,
                                      C.POINTER(C.c_int)]
        L.orc_tfoec.argtypes = [cp, dp, dp, C.c_int, C.c_double, C.c_double, dp, dp, cp]
        L.orc_chan_est.argtypes = [cp, dp, C.c_int, C.c_int, dp, dp]
        L.orc_decode_mib.argtypes = [cp, dp, C.c_int, cp]
        L.orc_search_capbuf.argtypes = [dp, C.c_uint32, dp, C.c_uint32, C.c_double, C.c_double, C.c_double,
                                        cp, C.c_int, C.POINTER(C.c_int), cp, C.c_int, C.POINTER(C.c_int)]
        L.orc_pss_td.argtypes = [C.c_int, dp]
        L.orc_pss_fd.argtypes = [C.c_int, dp]
        L.orc_sss_fd.argtypes = [C.c_int, C.c_int, C.c_int, ip]
        L.orc_lte_pn.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8)]
        L.orc_rs_dl.argtypes = [C.c_int, C.c_int, dp, dp]
        L.orc_fft128.argtypes = [dp, dp]
        L.orc_set_legacy.argtypes = [C.c_int]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_cell_init.argtypes = [cp]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _cap(capbuf) -> np.ndarray:
    c = np.ascontiguousarray(capbuf, dtype=np.complex128)
    return c


def new_cell(**kw) -> Cell:
    c = Cell()
    lib().orc_cell_init(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def set_legacy(on: bool) -> None:
    lib().orc_set_legacy(int(bool(on)))


def set_threads(n: int) -> None:
    lib().orc_set_threads(int(n))


def chi2cdf_inv(p: float, k: float) -> float:
    return lib().orc_chi2cdf_inv(p, k)


def pss_td(n_id_2: int) -> np.ndarray:
    o = np.empty(137, np.complex128)
    lib().orc_pss_td(n_id_2, _dp(o))
    return o


def pss_fd(n_id_2: int) -> np.ndarray:
    o = np.empty(62, np.complex128)
    lib().orc_pss_fd(n_id_2, _dp(o))
    return o


def sss_fd(n_id_1: int, n_id_2: int, slot: int) -> np.ndarray:
    o = np.empty(62, np.int32)
    lib().orc_sss_fd(n_id_1, n_id_2, slot, _ip(o))
    return o


def lte_pn(c_init: int, n: int) -> np.ndarray:
    o = np.empty(n, np.uint8)
    lib().orc_lte_pn(c_init, n, o.ctypes.data_as(C.POINTER(C.c_uint8)))
    return o


def rs_dl(n_id_cell: int, cp_type: int):
    n_symb = 7 if cp_type == 1 else 6
    rs = np.zeros((20 * n_symb, 12), np.complex128)
    sh = np.zeros((20 * n_symb, 4), np.float64)
    lib().orc_rs_dl(n_id_cell, cp_type, _dp(rs), _dp(sh))
    return rs, sh


def fft128(x) -> np.ndarray:
    x = np.ascontiguousarray(x, np.complex128)
    o = np.empty(128, np.complex128)
    lib().orc_fft128(_dp(x), _dp(o))
    return o


def xcorr_pss(capbuf, f_search_set, ds_comb_arm, fc_requested, fc_programmed, fs_programmed,
              want_xc=False, want_sp=False):
    """Returns dict with pow[3,9600] f64, frq[3,9600] i32, single/incoherent [3,9600,n_f] f32,
    sp_incoherent[9600], n_comb_xc, n_comb_sp (+ xc[3,n_cap-136,n_f] c64, sp)."""
    cap = _cap(capbuf)
    f = np.ascontiguousarray(f_search_set, np.float64)
    n_cap, n_f = cap.size, f.size
    out = dict(pow=np.empty((3, 9600)), frq=np.empty((3, 9600), np.int32),
               single=np.empty((3, 9600, n_f), np.float32), incoherent=np.empty((3, 9600, n_f), np.float32),
               sp_incoherent=np.empty(9600))
    xc = np.empty((3, n_cap - 136, n_f), np.complex64) if want_xc else None
    n_comb_sp_max = (n_cap - 136 - 137) // 9600
    sp = np.empty(n_comb_sp_max * 9600) if want_sp else None
    ncx, ncs = C.c_uint16(0), C.c_uint16(0)
    rc = lib().orc_xcorr_pss(_dp(cap), n_cap, _dp(f), n_f, int(ds_comb_arm), fc_requested, fc_programmed,
                             fs_programmed, _dp(out["pow"]), _ip(out["frq"]), _fp(out["single"]),
                             _fp(out["incoherent"]), _dp(out["sp_incoherent"]),
                             xc.ctypes.data_as(C.POINTER(C.c_float)) if want_xc else None,
                             _dp(sp) if want_sp else None, C.byref(ncx), C.byref(ncs))
    if rc:
        raise RuntimeError(f"orc_xcorr_pss rc={rc}")
    out.update(n_comb_xc=ncx.value, n_comb_sp=ncs.value, xc=xc, sp=sp)
    return out


def z_th1(sp_incoherent, n_comb_xc, ds_comb_arm=2, thresh1_n_nines=12):
    """Threshold recipe of the reference's main loop (src/CellSearch.cpp:500-503)."""
    R_th1 = chi2cdf_inv(1 - pow(10.0, -thresh1_n_nines), 2 * n_comb_xc * (2 * ds_comb_arm + 1))
    rx_cutoff = (6 * 12 * 15e3 / 2 + 4 * 15e3) / (30720000.0 / 16 / 2)
    return R_th1 * np.asarray(sp_incoherent) / rx_cutoff / 137 / 2 / n_comb_xc / (2 * ds_comb_arm + 1)


def peak_search(pow_, frq, Z_th1, f_search_set, fc_requested, fc_programmed, single, ds_comb_arm, max_cells=256):
    pow_ = np.ascontiguousarray(pow_, np.float64)
    frq = np.ascontiguousarray(frq, np.int32)
    Z = np.ascontiguousarray(Z_th1, np.float64)
    f = np.ascontiguousarray(f_search_set, np.float64)
    single = np.ascontiguousarray(single, np.float32)
    cells = (Cell * max_cells)()
    n = C.c_int(0)
    lib().orc_peak_search(_dp(pow_), _ip(frq), _dp(Z), _dp(f), f.size, fc_requested, fc_programmed,
                          _fp(single), int(ds_comb_arm), cells, max_cells, C.byref(n))
    return [cells[i] for i in range(min(n.value, max_cells))]


def _copy(c: Cell) -> Cell:
    o = Cell()
    C.memmove(C.byref(o), C.byref(c), C.sizeof(Cell))
    return o


def sss_detect(cell, capbuf, thresh2_n_sigma, fc_requested, fc_programmed, fs_programmed):
    cap = _cap(capbuf)
    out = Cell()
    d = dict(h1_np=np.empty(62), h2_np=np.empty(62), h1_nrm=np.empty(62, np.complex128),
             h2_nrm=np.empty(62, np.complex128), h1_ext=np.empty(62, np.complex128),
             h2_ext=np.empty(62, np.complex128), ll_nrm=np.empty((168, 2)), ll_ext=np.empty((168, 2)))
    rc = lib().orc_sss_detect(C.byref(cell), _dp(cap), cap.size, thresh2_n_sigma, fc_requested, fc_programmed,
                              fs_programmed, C.byref(out), _dp(d["h1_np"]), _dp(d["h2_np"]), _dp(d["h1_nrm"]),
                              _dp(d["h2_nrm"]), _dp(d["h1_ext"]), _dp(d["h2_ext"]), _dp(d["ll_nrm"]), _dp(d["ll_ext"]))
    if rc:
        raise RuntimeError(f"orc_sss_detect rc={rc}")
    return out, d


def pss_sss_foe(cell, capbuf, fc_requested, fc_programmed, fs_programmed) -> Cell:
    cap = _cap(capbuf)
    out = Cell()
    rc = lib().orc_pss_sss_foe(C.byref(cell), _dp(cap), cap.size, fc_requested, fc_programmed, fs_programmed,
                               C.byref(out))
    if rc:
        raise RuntimeError(f"orc_pss_sss_foe rc={rc}")
    return out


def extract_tfg(cell, capbuf, fc_requested, fc_programmed, fs_programmed):
    cap = _cap(capbuf)
    tfg = np.zeros((854, 72), np.complex128)
    ts = np.zeros(854)
    n = C.c_int(0)
    rc = lib().orc_extract_tfg(C.byref(cell), _dp(cap), cap.size, fc_requested, fc_programmed, fs_programmed,
                               _dp(tfg), _dp(ts), C.byref(n))
    if rc:
        raise RuntimeError(f"orc_extract_tfg rc={rc}")
    return tfg[:n.value].copy(), ts[:n.value].copy()


def tfoec(cell, tfg, ts, fc_requested, fc_programmed):
    tfg = np.ascontiguousarray(tfg, np.complex128)
    ts = np.ascontiguousarray(ts, np.float64)
    n = tfg.shape[0]
    tfgc = np.empty_like(tfg)
    tsc = np.empty_like(ts)
    out = Cell()
    rc = lib().orc_tfoec(C.byref(cell), _dp(tfg), _dp(ts), n, fc_requested, fc_programmed, _dp(tfgc), _dp(tsc),
                         C.byref(out))
    if rc:
        raise RuntimeError(f"orc_tfoec rc={rc}")
    return out, tfgc, tsc


def chan_est(cell, tfg, port):
    tfg = np.ascontiguousarray(tfg, np.complex128)
    ce = np.empty_like(tfg)
    np_ = C.c_double(0)
    lib().orc_chan_est(C.byref(cell), _dp(tfg), tfg.shape[0], port, _dp(ce), C.byref(np_))
    return ce, np_.value


def chan_est_dbg(cell, tfg, port):
    """chan_est's internal estimates -> (raw [n_rs][12], filtered [n_rs][12], grid rows [n_rs])"""
    tfg = np.ascontiguousarray(tfg, np.complex128)
    raw, filt = np.empty((512, 12), np.complex128), np.empty((512, 12), np.complex128)
    rows, n = np.empty(512, np.int32), C.c_int(0)
    L = lib()
    L.orc_chan_est_dbg.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int)]
    rc = L.orc_chan_est_dbg(C.byref(cell), _dp(tfg), tfg.shape[0], port, _dp(raw), _dp(filt), _ip(rows), C.byref(n))
    if rc:
        raise RuntimeError(f"orc_chan_est_dbg rc={rc}")
    return raw[:n.value].copy(), filt[:n.value].copy(), rows[:n.value].copy()


def trk_raw_filt(cell, syms, slot0, sym0, port):
    """The tracker's raw / filter_ce estimates of one port -> (raw [n][12], their symbol indices, filt [m][12], indices)"""
    syms = np.ascontiguousarray(syms, np.complex128)
    n = syms.shape[0]
    raw, filt = np.empty((n, 12), np.complex128), np.empty((n, 12), np.complex128)
    ri, fi = np.empty(n, np.int32), np.empty(n, np.int32)
    nr, nf = C.c_int(0), C.c_int(0)
    L = lib()
    L.orc_trk_raw_filt.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int)]
    rc = L.orc_trk_raw_filt(C.byref(cell), _dp(syms), n, slot0, sym0, port, _dp(raw), _ip(ri), C.byref(nr), _dp(filt), _ip(fi), C.byref(nf))
    if rc:
        raise RuntimeError(f"orc_trk_raw_filt rc={rc}")
    return raw[:nr.value].copy(), ri[:nr.value].copy(), filt[:nf.value].copy(), fi[:nf.value].copy()


def decode_mib(cell, tfg) -> Cell:
    tfg = np.ascontiguousarray(tfg, np.complex128)
    out = Cell()
    lib().orc_decode_mib(C.byref(cell), _dp(tfg), tfg.shape[0], C.byref(out))
    return out


def search_capbuf(capbuf, f_search_set, fc_requested, fc_programmed, fs_programmed, max_cells=64):
    """Full per-buffer chain of CellSearch's main loop.  Returns (cells, peaks)."""
    cap = _cap(capbuf)
    f = np.ascontiguousarray(f_search_set, np.float64)
    cells = (Cell * max_cells)()
    peaks = (Cell * 256)()
    n, npk = C.c_int(0), C.c_int(0)
    rc = lib().orc_search_capbuf(_dp(cap), cap.size, _dp(f), f.size, fc_requested, fc_programmed, fs_programmed,
                                 cells, max_cells, C.byref(n), peaks, 256, C.byref(npk))
    if rc:
        raise RuntimeError(f"orc_search_capbuf rc={rc}")
    return [_copy(cells[i]) for i in range(min(n.value, max_cells))], [_copy(peaks[i]) for i in range(min(npk.value, 256))]


# ---- LTE-Tracker per-symbol pipeline on a block of OFDM symbols (src/tracker_thread.cpp) ----
def trk_get_fd(cell, td, slot0, sym0, freq_off, late, fc_requested, fc_programmed, fs_programmed, bpo=0.0):
    """-> (syms [n_sym][72], bulk_phase_offset after the block, its value at every symbol)"""
    td = np.ascontiguousarray(td, np.complex128)
    n = td.shape[0]
    fo, lt = np.ascontiguousarray(freq_off, np.float64), np.ascontiguousarray(late, np.float64)
    syms, trace = np.empty((n, 72), np.complex128), np.empty(n)
    b = C.c_double(bpo)
    rc = lib().orc_trk_get_fd(C.byref(cell), _dp(td), n, slot0, sym0, _dp(fo), _dp(lt), C.c_double(fc_requested),
                              C.c_double(fc_programmed), C.c_double(fs_programmed), C.byref(b), _dp(syms), _dp(trace))
    if rc:
        raise RuntimeError(f"orc_trk_get_fd rc={rc}")
    return syms, b.value, trace


def trk_chan_est(cell, syms, slot0, sym0, freq_off, frame_timing, fc_requested, fc_programmed, fs_programmed, max_rs=None):
    """-> dict(meas [4][max_rs][9], n_meas [4], ce [4][n_sym][72], ce_pw [4][n_sym][4], ce_upto [4])"""
    syms = np.ascontiguousarray(syms, np.complex128)
    n = syms.shape[0]
    max_rs = max_rs or n
    fo, ft = np.ascontiguousarray(freq_off, np.float64), np.ascontiguousarray(frame_timing, np.float64)
    out = dict(meas=np.full((4, max_rs, 9), np.nan), n_meas=np.zeros(4, np.int32), ce=np.full((4, n, 72), np.nan + 0j, np.complex128),
               ce_pw=np.full((4, n, 4), np.nan), ce_upto=np.zeros(4, np.int32))
    rc = lib().orc_trk_chan_est(C.byref(cell), _dp(syms), n, slot0, sym0, _dp(fo), _dp(ft), C.c_double(fc_requested),
                                C.c_double(fc_programmed), C.c_double(fs_programmed), _dp(out["meas"]), max_rs, _ip(out["n_meas"]),
                                _dp(out["ce"]), _dp(out["ce_pw"]), _ip(out["ce_upto"]))
    if rc:
        raise RuntimeError(f"orc_trk_chan_est rc={rc}")
    return out


def trk_stats(cell, syms, slot0, sym0, meas, n_meas):
    """do_ac_fd / do_ac_td / do_pss_sss_sigpower_ce without their running averages (see orc_trk_stats).
    -> dict(ac_fd [4][max_rs][12], ac_td [4][max_rs][72], sync [max_hf][4], sync_ce [max_hf][72], n_hf)"""
    syms = np.ascontiguousarray(syms, np.complex128)
    n = syms.shape[0]
    meas = np.ascontiguousarray(meas, np.float64)
    max_rs, max_hf = meas.shape[1], n // 60 + 2
    nm = np.ascontiguousarray(n_meas, np.int32)
    out = dict(ac_fd=np.full((4, max_rs, 12), np.nan + 0j, np.complex128), ac_td=np.full((4, max_rs, 72), np.nan + 0j, np.complex128),
               sync=np.full((max_hf, 4), np.nan), sync_ce=np.full((max_hf, 72), np.nan + 0j, np.complex128))
    nh = C.c_int(0)
    L = lib()
    L.orc_trk_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int,
                                C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    rc = L.orc_trk_stats(C.byref(cell), _dp(syms), n, slot0, sym0, _dp(meas), max_rs, _ip(nm), _dp(out["ac_fd"]), _dp(out["ac_td"]),
                         _dp(out["sync"]), _dp(out["sync_ce"]), max_hf, C.byref(nh))
    if rc:
        raise RuntimeError(f"orc_trk_stats rc={rc}")
    out["n_hf"] = nh.value
    return out


def trk_mib(cell, syms16, ce16, np16):
    """-> (c_est [40] uint8, crc_ok, fields_ok)"""
    s = np.ascontiguousarray(syms16, np.complex128)
    ce = np.ascontiguousarray(ce16, np.complex128)
    npw = np.ascontiguousarray(np16, np.float64)
    bits = np.zeros(40, np.uint8)
    a, b = C.c_int(0), C.c_int(0)
    rc = lib().orc_trk_mib(C.byref(cell), _dp(s), _dp(ce), _dp(npw), bits.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(a), C.byref(b))
    if rc:
        raise RuntimeError(f"orc_trk_mib rc={rc}")
    return bits, bool(a.value), bool(b.value)


def producer_cut(n_cap, frame_timing, cp_type, frequency_offset, fc_requested, fc_programmed, fs_programmed, n_sym_max, ts_first=0.0):
    """The producer thread's sample loop for one tracked cell (src/producer_thread.cpp:96-131, 196-246) on a buffer of n_cap samples
    whose first sample has timestamp ts_first: -> (hit [n] int32: first sample of every complete 128-sample capture, late [n])."""
    hit = np.zeros(n_sym_max, np.int32)
    late = np.zeros(n_sym_max, np.float64)
    L = lib()
    L.orc_producer_cut.argtypes = [C.c_uint32, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                   C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    n = L.orc_producer_cut(int(n_cap), float(ts_first), float(frame_timing), int(cp_type), float(frequency_offset), float(fc_requested),
                           float(fc_programmed), float(fs_programmed), int(n_sym_max), _ip(hit), _dp(late))
    return hit[:n], late[:n]


def solve3(M, V):
    M = np.ascontiguousarray(M, np.complex128); V = np.ascontiguousarray(V, np.complex128)
    o = np.empty(3, np.complex128)
    lib().orc_solve3(_dp(M), _dp(V), _dp(o))
    return o


def qpsk_llr(syms, np_):
    s = np.ascontiguousarray(syms, np.complex128); n = np.ascontiguousarray(np_, np.float64)
    out = np.empty(2 * s.size)
    lib().orc_qpsk_llr(_dp(s), _dp(n), s.size, _dp(out))
    return out
