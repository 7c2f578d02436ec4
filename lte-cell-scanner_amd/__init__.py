"""lte-cell-scanner_amd -- MI355X-native LTE cell-search hot path.

Python host-side mirror of the reference's searcher interface (include/searcher.h:22-119):
the same seven functions with the same argument meaning, each a thin call through the C ABI
of liblcs_amd.so (include/lcs.h) into hand-written HIP kernels.  numpy arrays stand in for
the IT++ containers; 2-D results are [3][9600] row-major, 3-D results [t][idx][foi].

There is deliberately no CPU implementation here: without the HIP library / a GPU every
call raises (``SearcherError``).  The CPU oracle used by the tests lives in oracle/ and is
never imported from this package.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from . import synth
from . import sweep
from . import itfile
from . import tracker
from .capi import LcsCell, LcsTrackCell, FMT_C64, FMT_IQ_U8, FMT_C128, STAGE_PSS, STAGE_FULL, MAX_PEAKS

FS_LTE = 30720000.0        # include/constants.h:32
DS_COMB_ARM = 2            # src/CellSearch.cpp:484
THRESH2_N_SIGMA = 3        # src/CellSearch.cpp:528


class SearcherError(RuntimeError):
    pass


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def new_cell(**kw) -> LcsCell:
    c = LcsCell()
    capi.load().lcs_cell_init(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def f_search_set_for(freq_start: float, ppm: float) -> np.ndarray:
    """Frequency-offset grid of the CLI (src/CellSearch.cpp:463-464)."""
    n_extra = int(np.floor((freq_start * ppm / 1e6 + 2.5e3) / 5e3))
    return np.arange(-n_extra, n_extra + 1) * 5000.0


class Searcher:
    """One context = one GPU + stream + workspace (lcs_create / lcs_destroy)."""

    def __init__(self, device: int = -1):
        self._lib = capi.load()
        h = C.c_void_p()
        rc = self._lib.lcs_create(device, C.byref(h))
        if rc != 0:
            raise SearcherError(f"lcs_create failed: {capi.ERRORS.get(rc, rc)} (an MI355X is required; no CPU fallback)")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lcs_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc, what, allow_overflow=False):
        if rc == 0 or (allow_overflow and rc == -4):
            return rc
        raise SearcherError(f"{what}: {capi.ERRORS.get(rc, rc)}: {self._lib.lcs_last_error(self._h).decode()}")

    def set_max_cells_in_flight(self, n: int):
        self._chk(self._lib.lcs_set_max_cells_in_flight(self._h, n), "lcs_set_max_cells_in_flight")

    # ---- searcher.h:22-41 -------------------------------------------------------------
    def xcorr_pss(self, capbuf, f_search_set, ds_comb_arm, fc_requested, fc_programmed, fs_programmed,
                  want_incoherent=True, want_xc=False, want_sp=False):
        cap = np.ascontiguousarray(capbuf, np.complex128)
        f = np.ascontiguousarray(f_search_set, np.float64)
        n_cap, n_f = cap.size, f.size
        out = dict(pow=np.empty((3, 9600)), frq=np.empty((3, 9600), np.int32),
                   single=np.empty((3, 9600, n_f), np.float32),
                   incoherent=np.empty((3, 9600, n_f), np.float32) if want_incoherent else None,
                   sp_incoherent=np.empty(9600))
        xc = np.empty((3, n_cap - 136, n_f), np.complex64) if want_xc else None
        sp = np.empty(((n_cap - 136 - 137) // 9600) * 9600) if want_sp else None
        ncx, ncs = C.c_uint16(0), C.c_uint16(0)
        rc = self._lib.lcs_xcorr_pss(self._h, _dp(cap), n_cap, _dp(f), n_f, int(ds_comb_arm), fc_requested,
                                     fc_programmed, fs_programmed, _dp(out["pow"]), _ip(out["frq"]),
                                     _fp(out["single"]), _fp(out["incoherent"]), _dp(out["sp_incoherent"]),
                                     xc.ctypes.data_as(C.POINTER(C.c_float)) if want_xc else None, _dp(sp),
                                     C.byref(ncx), C.byref(ncs))
        self._chk(rc, "lcs_xcorr_pss")
        out.update(n_comb_xc=ncx.value, n_comb_sp=ncs.value, xc=xc, sp=sp)
        return out

    # ---- searcher.h:44-56 -------------------------------------------------------------
    def peak_search(self, pow_, frq, Z_th1, f_search_set, fc_requested, fc_programmed, single, ds_comb_arm,
                    max_cells=MAX_PEAKS):
        pow_ = np.ascontiguousarray(pow_, np.float64)
        frq = np.ascontiguousarray(frq, np.int32)
        Z = np.ascontiguousarray(Z_th1, np.float64)
        f = np.ascontiguousarray(f_search_set, np.float64)
        single = np.ascontiguousarray(single, np.float32)
        cells = (LcsCell * max_cells)()
        n = C.c_int(0)
        rc = self._lib.lcs_peak_search(self._h, _dp(pow_), _ip(frq), _dp(Z), _dp(f), f.size, fc_requested,
                                       fc_programmed, _fp(single), int(ds_comb_arm), cells, max_cells, C.byref(n))
        self._chk(rc, "lcs_peak_search")
        return [cells[i].copy() for i in range(n.value)]

    # ---- searcher.h:59-76 -------------------------------------------------------------
    def sss_detect(self, cell, capbuf, thresh2_n_sigma, fc_requested, fc_programmed, fs_programmed):
        cap = np.ascontiguousarray(capbuf, np.complex128)
        out = LcsCell()
        d = dict(h1_np=np.empty(62), h2_np=np.empty(62), h1_nrm=np.empty(62, np.complex128),
                 h2_nrm=np.empty(62, np.complex128), h1_ext=np.empty(62, np.complex128),
                 h2_ext=np.empty(62, np.complex128), ll_nrm=np.empty((168, 2)), ll_ext=np.empty((168, 2)))
        rc = self._lib.lcs_sss_detect(self._h, C.byref(cell), _dp(cap), cap.size, thresh2_n_sigma, fc_requested,
                                      fc_programmed, fs_programmed, C.byref(out), _dp(d["h1_np"]), _dp(d["h2_np"]),
                                      _dp(d["h1_nrm"]), _dp(d["h2_nrm"]), _dp(d["h1_ext"]), _dp(d["h2_ext"]),
                                      _dp(d["ll_nrm"]), _dp(d["ll_ext"]))
        self._chk(rc, "lcs_sss_detect")
        return out, d

    # ---- searcher.h:79-85 -------------------------------------------------------------
    def pss_sss_foe(self, cell, capbuf, fc_requested, fc_programmed, fs_programmed):
        cap = np.ascontiguousarray(capbuf, np.complex128)
        out = LcsCell()
        rc = self._lib.lcs_pss_sss_foe(self._h, C.byref(cell), _dp(cap), cap.size, fc_requested, fc_programmed,
                                       fs_programmed, C.byref(out))
        self._chk(rc, "lcs_pss_sss_foe")
        return out

    # ---- searcher.h:88-98 -------------------------------------------------------------
    def extract_tfg(self, cell, capbuf, fc_requested, fc_programmed, fs_programmed):
        cap = np.ascontiguousarray(capbuf, np.complex128)
        tfg = np.zeros((854, 72), np.complex128)
        ts = np.zeros(854)
        n = C.c_int(0)
        rc = self._lib.lcs_extract_tfg(self._h, C.byref(cell), _dp(cap), cap.size, fc_requested, fc_programmed,
                                       fs_programmed, _dp(tfg), _dp(ts), C.byref(n))
        self._chk(rc, "lcs_extract_tfg")
        return tfg[:n.value].copy(), ts[:n.value].copy()

    # ---- searcher.h:101-112 -----------------------------------------------------------
    def tfoec(self, cell, tfg, tfg_timestamp, fc_requested, fc_programmed):
        tfg = np.ascontiguousarray(tfg, np.complex128)
        ts = np.ascontiguousarray(tfg_timestamp, np.float64)
        tfgc, tsc = np.empty_like(tfg), np.empty_like(ts)
        out = LcsCell()
        rc = self._lib.lcs_tfoec(self._h, C.byref(cell), _dp(tfg), _dp(ts), tfg.shape[0], fc_requested,
                                 fc_programmed, _dp(tfgc), _dp(tsc), C.byref(out))
        self._chk(rc, "lcs_tfoec")
        return out, tfgc, tsc

    # ---- searcher.h:115-119 -----------------------------------------------------------
    def decode_mib(self, cell, tfg):
        tfg = np.ascontiguousarray(tfg, np.complex128)
        out = LcsCell()
        rc = self._lib.lcs_decode_mib(self._h, C.byref(cell), _dp(tfg), tfg.shape[0], C.byref(out))
        self._chk(rc, "lcs_decode_mib")
        return out

    # ---- searcher.cpp:1369-1477 (internal to decode_mib; exported for direct testing) ------
    def chan_est(self, cell, tfg, port: int):
        tfg = np.ascontiguousarray(tfg, np.complex128)
        ce = np.empty_like(tfg)
        npw = C.c_double(0)
        self._chk(self._lib.lcs_chan_est(self._h, C.byref(cell), _dp(tfg), tfg.shape[0], int(port), _dp(ce), C.byref(npw)), "lcs_chan_est")
        return ce, npw.value

    # ---- CellSearch.cpp:484-558, one buffer -----------------------------------------
    def search_capbuf(self, capbuf, f_search_set, fc_requested, fc_programmed, fs_programmed, max_cells=MAX_PEAKS):
        cap = np.ascontiguousarray(capbuf, np.complex128)
        f = np.ascontiguousarray(f_search_set, np.float64)
        cells, peaks = (LcsCell * max_cells)(), (LcsCell * MAX_PEAKS)()
        n, npk = C.c_int(0), C.c_int(0)
        rc = self._lib.lcs_search_capbuf(self._h, _dp(cap), cap.size, _dp(f), f.size, fc_requested, fc_programmed,
                                         fs_programmed, cells, max_cells, C.byref(n), peaks, MAX_PEAKS, C.byref(npk))
        self._chk(rc, "lcs_search_capbuf")
        return [cells[i].copy() for i in range(min(n.value, max_cells))], [peaks[i].copy() for i in range(min(npk.value, MAX_PEAKS))]

    # ---- one buffer, hypotheses split over GPUs (lcs_foe_*; driver: sweep.search_capbuf_foe_split_dev) ----
    def foe_partial(self, capbuf, f_search_set, f_first: int, f_count: int, fc_requested, fc_programmed, fs_programmed,
                    d_words_ptr: int, d_meta_ptr: int):
        cap = np.ascontiguousarray(capbuf, np.complex128)
        f = np.ascontiguousarray(f_search_set, np.float64)
        self._chk(self._lib.lcs_foe_partial(self._h, _dp(cap), cap.size, _dp(f), f.size, int(f_first), int(f_count), fc_requested,
                                            fc_programmed, fs_programmed, C.c_void_p(d_words_ptr), C.c_void_p(d_meta_ptr)), "lcs_foe_partial")

    def foe_contend(self, f_search_set, d_words_ptr: int, d_words2_ptr: int):
        """After the MAX all-reduce of the words: this rank's exact packed maxima at the positions it contends for (-1 elsewhere)
        into d_words2, which the caller MAX-all-reduces in turn (lcs_foe_contend)."""
        f = np.ascontiguousarray(f_search_set, np.float64)
        self._chk(self._lib.lcs_foe_contend(self._h, _dp(f), f.size, C.c_void_p(d_words_ptr), C.c_void_p(d_words2_ptr)), "lcs_foe_contend")

    def foe_resolve(self, d_words_ptr: int, d_words2_ptr: int):
        """The reduced exact words take the place of the approximate ones (lcs_foe_resolve); then foe_finish."""
        self._chk(self._lib.lcs_foe_resolve(self._h, C.c_void_p(d_words_ptr), C.c_void_p(d_words2_ptr)), "lcs_foe_resolve")

    def foe_finish(self, d_words_ptr: int, d_meta_ptr: int, f_search_set, max_cells: int = MAX_PEAKS):
        """-> (cells this rank decoded, their positions in the peak list, the whole peak list)"""
        f = np.ascontiguousarray(f_search_set, np.float64)
        cells, peaks = (LcsCell * max_cells)(), (LcsCell * MAX_PEAKS)()
        order = np.zeros(max_cells, np.int32)
        n, npk = C.c_int(0), C.c_int(0)
        rc = self._lib.lcs_foe_finish(self._h, C.c_void_p(d_words_ptr), C.c_void_p(d_meta_ptr), _dp(f), f.size, cells, _ip(order), max_cells,
                                      C.byref(n), peaks, MAX_PEAKS, C.byref(npk))
        self._chk(rc, "lcs_foe_finish")
        return [cells[i].copy() for i in range(n.value)], order[:n.value].copy(), [peaks[i].copy() for i in range(min(npk.value, MAX_PEAKS))]

    # ---- batched, device-resident ---------------------------------------------------
    def batch_enqueue(self, d_ptr: int, fmt: int, n_buf: int, n_cap: int, f_search_set, fc_requested, fc_programmed,
                      fs_programmed: float, stage_mask: int = STAGE_FULL):
        f = np.ascontiguousarray(f_search_set, np.float64)
        fr = np.ascontiguousarray(np.broadcast_to(np.asarray(fc_requested, np.float64), (n_buf,)))
        fp_ = np.ascontiguousarray(np.broadcast_to(np.asarray(fc_programmed, np.float64), (n_buf,)))
        rc = self._lib.lcs_batch_enqueue(self._h, C.c_void_p(d_ptr), fmt, n_buf, n_cap, _dp(f), f.size, _dp(fr),
                                         _dp(fp_), fs_programmed, stage_mask)
        self._chk(rc, "lcs_batch_enqueue")

    def batch_collect(self, n_buf: int, max_cells_per_buf: int = 16):
        """-> per-buffer cell lists.  LCS_ERR_OVERFLOW (more peaks / cells than the library's or the caller's
        arrays hold: results truncated) is surfaced as a warning and kept in ``self.last_overflow``."""
        cells = (LcsCell * (n_buf * max_cells_per_buf))()
        cnt = (C.c_int * n_buf)()
        rc = self._lib.lcs_batch_collect(self._h, cells, max_cells_per_buf, cnt)
        self._note_overflow(self._chk(rc, "lcs_batch_collect", allow_overflow=True), "lcs_batch_collect")
        return [[cells[b * max_cells_per_buf + i].copy() for i in range(min(cnt[b], max_cells_per_buf))]
                for b in range(n_buf)]

    def _note_overflow(self, rc, what):
        self.last_overflow = rc == -4
        if self.last_overflow:
            import warnings
            warnings.warn(f"{what}: LCS_ERR_OVERFLOW: {self._lib.lcs_last_error(self._h).decode()} (results truncated)",
                          RuntimeWarning, stacklevel=3)

    def batch_readback(self, buf: int, n_f: int):
        """xcorr_pss outputs of buffer `buf` of the last batch in the reference's layouts (debug)."""
        out = dict(single=np.empty((3, 9600, n_f), np.float32), pow=np.empty((3, 9600)), frq=np.empty((3, 9600), np.int32),
                   sp_incoherent=np.empty(9600), z_th1=np.empty(9600))
        self._chk(self._lib.lcs_batch_readback(self._h, buf, _fp(out["single"]), _dp(out["pow"]), _ip(out["frq"]),
                                               _dp(out["sp_incoherent"]), _dp(out["z_th1"])), "lcs_batch_readback")
        return out

    def search_batch_host(self, h_capbufs, fmt: int, n_buf: int, n_cap: int, f_search_set, fc_requested, fc_programmed,
                          fs_programmed: float, stage_mask: int = STAGE_FULL, max_cells_per_buf: int = 16):
        a = np.ascontiguousarray(h_capbufs, dtype=np.uint8 if fmt == FMT_IQ_U8 else np.complex64)
        assert a.size == n_buf * n_cap * (2 if fmt == FMT_IQ_U8 else 1)
        f = np.ascontiguousarray(f_search_set, np.float64)
        fr = np.ascontiguousarray(np.broadcast_to(np.asarray(fc_requested, np.float64), (n_buf,)))
        fp_ = np.ascontiguousarray(np.broadcast_to(np.asarray(fc_programmed, np.float64), (n_buf,)))
        cells = (LcsCell * (n_buf * max_cells_per_buf))()
        cnt = (C.c_int * n_buf)()
        rc = self._lib.lcs_search_batch_host(self._h, a.ctypes.data_as(C.c_void_p), fmt, n_buf, n_cap, _dp(f), f.size, _dp(fr),
                                             _dp(fp_), fs_programmed, stage_mask, cells, max_cells_per_buf, cnt)
        self._note_overflow(self._chk(rc, "lcs_search_batch_host", allow_overflow=True), "lcs_search_batch_host")
        return [[cells[b * max_cells_per_buf + i].copy() for i in range(min(cnt[b], max_cells_per_buf))]
                for b in range(n_buf)]

    def host_alloc(self, n_bytes: int) -> np.ndarray:
        """Page-locked host memory as a uint8 array (lcs_host_alloc): batch_enqueue_host DMAs from it without staging.
        Keep the Searcher alive while the array is in use; host_free() releases it."""
        p = C.c_void_p()
        self._chk(self._lib.lcs_host_alloc(self._h, n_bytes, C.byref(p)), "lcs_host_alloc")
        a = np.ctypeslib.as_array((C.c_uint8 * n_bytes).from_address(p.value))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p
        return a

    def host_free(self, a: np.ndarray):
        p = self._pinned.pop(a.ctypes.data)
        self._chk(self._lib.lcs_host_free(self._h, p), "lcs_host_free")

    def batch_enqueue_host(self, h_capbufs, fmt: int, n_buf: int, n_cap: int, f_search_set, fc_requested, fc_programmed,
                           fs_programmed: float, stage_mask: int = STAGE_FULL):
        """Asynchronous host-fed batch (lcs_batch_enqueue_host): returns once the copy and the kernels are queued;
        results come from batch_collect / batch_collect_raw.  The array must stay untouched until then."""
        # (the same lifetime rule holds for batch_enqueue's DEVICE buffers: complex<float> batches are read in place by
        # every stage until batch_collect returns -- include/lcs.h)
        a = h_capbufs if (isinstance(h_capbufs, np.ndarray) and h_capbufs.flags.c_contiguous) else np.ascontiguousarray(h_capbufs)
        assert a.nbytes == n_buf * n_cap * (2 if fmt == FMT_IQ_U8 else 8)
        f = np.ascontiguousarray(f_search_set, np.float64)
        fr = np.ascontiguousarray(np.broadcast_to(np.asarray(fc_requested, np.float64), (n_buf,)))
        fp_ = np.ascontiguousarray(np.broadcast_to(np.asarray(fc_programmed, np.float64), (n_buf,)))
        self._keep = a
        rc = self._lib.lcs_batch_enqueue_host(self._h, a.ctypes.data_as(C.c_void_p), fmt, n_buf, n_cap, _dp(f), f.size, _dp(fr),
                                              _dp(fp_), fs_programmed, stage_mask)
        self._chk(rc, "lcs_batch_enqueue_host")

    def search_batch(self, d_ptr: int, fmt: int, n_buf: int, n_cap: int, f_search_set, fc_requested, fc_programmed,
                     fs_programmed: float, stage_mask: int = STAGE_FULL, max_cells_per_buf: int = 16):
        self.batch_enqueue(d_ptr, fmt, n_buf, n_cap, f_search_set, fc_requested, fc_programmed, fs_programmed, stage_mask)
        return self.batch_collect(n_buf, max_cells_per_buf)

    # ---- LTE-Tracker's per-symbol pipeline on a block of symbols (src/tracker_thread.cpp:823-1068) ----
    def track_block(self, cells, td, freq_off, frame_timing, late, fc_requested, fc_programmed, fs_programmed,
                    want_syms=True, want_ce=True, td_device_ptr=None, n_sym=None, out=None):
        """cells: list of searcher records (LcsCell with n_id_1/2, cp_type, n_ports, n_rb_dl, PHICH fields) or LcsTrackCell;
        td [n_cells][n_sym][128] complex128 (or a device pointer via td_device_ptr + n_sym); the metadata arrays are
        [n_cells][n_sym].  Returns a dict (see lcs_track_block in include/lcs.h); 'bpo' = bulk phase after the block."""
        n_cells = len(cells)
        tc = (capi.LcsTrackCell * n_cells)()
        for i, c in enumerate(cells):
            for f in ("n_id_1", "n_id_2", "cp_type", "n_ports", "n_rb_dl", "phich_duration", "phich_resource"):
                setattr(tc[i], f, int(getattr(c, f)))
            tc[i].bulk_phase_offset = float(getattr(c, "bulk_phase_offset", 0.0))
        fo = np.ascontiguousarray(freq_off, np.float64).reshape(n_cells, -1)
        n_sym = fo.shape[1] if n_sym is None else n_sym
        ft = np.ascontiguousarray(frame_timing, np.float64).reshape(n_cells, n_sym)
        lt = np.ascontiguousarray(late, np.float64).reshape(n_cells, n_sym)
        if td_device_ptr is None:
            tdh = np.ascontiguousarray(td, np.complex128).reshape(n_cells, n_sym, 128)
            tdp, on_dev = tdh.ctypes.data_as(C.c_void_p), 0
        else:
            tdp, on_dev = C.c_void_p(td_device_ptr), 1
        max_rs, max_off = n_sym // 3 + 4, max(1, n_sym // 120 - 3)
        if out is not None:      # a dict this method returned for the same block shape: its arrays are reused (rows beyond n_meas keep old values)
            assert out["meas"].shape == (n_cells, 4, max_rs, 9) and out["mib_ok"].shape == (n_cells, max_off)
            assert (out["syms"] is not None) == bool(want_syms) and (out["ce"] is not None) == bool(want_ce)
        else:
            out = dict(syms=np.empty((n_cells, n_sym, 72), np.complex128) if want_syms else None,
                       ce=np.full((n_cells, 4, n_sym, 72), np.nan + 0j, np.complex128) if want_ce else None,
                       ce_pw=np.full((n_cells, 4, n_sym, 4), np.nan) if want_ce else None,
                       ce_upto=np.zeros((n_cells, 4), np.int32), meas=np.full((n_cells, 4, max_rs, 9), np.nan),
                       n_meas=np.zeros((n_cells, 4), np.int32), mib_ok=np.full((n_cells, max_off), -1, np.int32),
                       mib_bits=np.zeros((n_cells, max_off), np.uint64))
        ms = C.c_float(0)
        rc = self._lib.lcs_track_block(self._h, tc, n_cells, n_sym, tdp, on_dev, _dp(fo), _dp(ft), _dp(lt), fc_requested, fc_programmed,
                                       fs_programmed, _dp(out["syms"]), _dp(out["ce"]), _dp(out["ce_pw"]), _ip(out["ce_upto"]),
                                       _dp(out["meas"]), max_rs, _ip(out["n_meas"]), _ip(out["mib_ok"]),
                                       out["mib_bits"].ctypes.data_as(C.POINTER(C.c_uint64)), max_off, C.byref(ms))
        self._chk(rc, "lcs_track_block")
        out["bpo"] = np.array([tc[i].bulk_phase_offset for i in range(n_cells)])
        out["gpu_ms"] = ms.value
        return out

    def set_float_batch_probe(self, on: bool):
        """complex<float> batches (FMT_C64) are checked for dongle data on the device and then take the u8 / int8 route
        (lcs_set_float_batch_probe, include/lcs.h); off by default."""
        self._chk(self._lib.lcs_set_float_batch_probe(self._h, 1 if on else 0), "lcs_set_float_batch_probe")

    def track_cut(self, d_capbuf_ptr, fmt, n_cap, cp_types, frame_timing, freq_off, fc_requested, fc_programmed, fs_programmed, n_sym,
                  d_td_ptr, ts_first=0.0, sym_first=None, pos_first=None, want_state=False):
        """The producer thread's symbol extraction on the device (lcs_track_cut, src/producer_thread.cpp:96-131, 196-246): the
        OFDM symbols of len(cp_types) tracked cells cut out of ONE capture buffer resident in HBM (fmt FMT_IQ_U8 / FMT_C64 /
        FMT_C128 at device pointer d_capbuf_ptr) into the device array d_td_ptr [n_cells][n_sym][128] complex128 -- what
        track_block takes as td_device_ptr.  ts_first / sym_first / pos_first continue a stream over several buffers (see
        include/lcs.h).  Returns (late [n_cells][n_sym], n_cut [n_cells]) and, with want_state, pos_next [n_cells]."""
        n_cells = len(cp_types)
        cp = np.ascontiguousarray(cp_types, np.int32)
        ft = np.ascontiguousarray(frame_timing, np.float64).reshape(n_cells)
        fo = np.ascontiguousarray(freq_off, np.float64).reshape(n_cells)
        sf = None if sym_first is None else np.ascontiguousarray(sym_first, np.int64).reshape(n_cells)
        pf = None if pos_first is None else np.ascontiguousarray(pos_first, np.int64).reshape(n_cells)
        late = np.zeros((n_cells, n_sym), np.float64)
        n_cut = np.zeros(n_cells, np.int32)
        pos_next = np.zeros(n_cells, np.int64)
        i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64)) if a is not None else None
        rc = self._lib.lcs_track_cut(self._h, C.c_void_p(d_capbuf_ptr), int(fmt), int(n_cap), float(ts_first), n_cells, _ip(cp), _dp(ft), _dp(fo),
                                     i64(sf), i64(pf), float(fc_requested), float(fc_programmed), float(fs_programmed), int(n_sym),
                                     C.c_void_p(d_td_ptr), _dp(late), _ip(n_cut), i64(pos_next))
        self._chk(rc, "lcs_track_cut")
        return (late, n_cut, pos_next) if want_state else (late, n_cut)

    def track_stream_block(self, cells, td, freq_off, frame_timing, late, fc_requested, fc_programmed, fs_programmed, want_stats=False,
                           want_syms=True, want_ce=True, td_device_ptr=None):
        """Continuous tracking (lcs_track_stream_block): the next block of a symbol stream.  `cells` as in track_block; the
        objects' bulk_phase_offset attribute (if any) seeds the first call.  Returns a dict whose rows carry their index in
        the whole stream: syms [c][n_sym][72]; meas [c][4][n_meas][9] (+ ac_fd, ac_td with want_stats); ce / ce_pw lists per
        (cell, port) with ce_from; mib_ok / mib_bits lists per cell with mib_from."""
        n_cells = len(cells)
        if not hasattr(self, "_trk_stream_cells"):
            tc = (capi.LcsTrackCell * n_cells)()
            for i, c in enumerate(cells):
                for f in ("n_id_1", "n_id_2", "cp_type", "n_ports", "n_rb_dl", "phich_duration", "phich_resource"):
                    setattr(tc[i], f, int(getattr(c, f)))
                tc[i].bulk_phase_offset = float(getattr(c, "bulk_phase_offset", 0.0))
            self._trk_stream_cells = tc
        tc = self._trk_stream_cells
        fo = np.ascontiguousarray(freq_off, np.float64).reshape(n_cells, -1)
        n_sym = fo.shape[1]
        ft = np.ascontiguousarray(frame_timing, np.float64).reshape(n_cells, n_sym)
        lt = np.ascontiguousarray(late, np.float64).reshape(n_cells, n_sym)
        if td_device_ptr is None:      # host samples (a page-locked array from host_alloc is DMA'd in place); else [n_cells][n_sym][128] complex128 in HBM
            tdh = np.ascontiguousarray(td, np.complex128).reshape(n_cells, n_sym, 128)
            td_arg = tdh.ctypes.data_as(C.c_void_p)
        else:
            td_arg = C.c_void_p(td_device_ptr)
        max_rs, ce_cap, max_off = n_sym // 3 + 8, n_sym + 64, n_sym // 120 + 4
        o = dict(syms=np.empty((n_cells, n_sym, 72), np.complex128) if want_syms else None,
                 ce=np.full((n_cells, 4, ce_cap, 72), np.nan + 0j, np.complex128) if want_ce else None,
                 ce_pw=np.full((n_cells, 4, ce_cap, 4), np.nan) if want_ce else None, ce_from=np.zeros((n_cells, 4), np.int64), ce_n=np.zeros((n_cells, 4), np.int32),
                 meas=np.full((n_cells, 4, max_rs, 9), np.nan), n_meas=np.zeros((n_cells, 4), np.int32),
                 ac_fd=np.full((n_cells, 4, max_rs, 12), np.nan + 0j, np.complex128) if want_stats else None,
                 ac_td=np.full((n_cells, 4, max_rs, 72), np.nan + 0j, np.complex128) if want_stats else None,
                 mib_ok=np.full((n_cells, max_off), -1, np.int32), mib_bits=np.zeros((n_cells, max_off), np.uint64),
                 mib_from=np.zeros(n_cells, np.int64), n_mib=np.zeros(n_cells, np.int32))
        i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
        rc = self._lib.lcs_track_stream_block(self._h, tc, n_cells, n_sym, td_arg, _dp(fo), _dp(ft), _dp(lt),
                                              fc_requested, fc_programmed, fs_programmed, _dp(o["syms"]), _dp(o["ce"]), _dp(o["ce_pw"]), ce_cap,
                                              i64(o["ce_from"]), _ip(o["ce_n"]), _dp(o["meas"]), _dp(o["ac_fd"]), _dp(o["ac_td"]), max_rs,
                                              _ip(o["n_meas"]), _ip(o["mib_ok"]), o["mib_bits"].ctypes.data_as(C.POINTER(C.c_uint64)), max_off,
                                              i64(o["mib_from"]), _ip(o["n_mib"]))
        self._chk(rc, "lcs_track_stream_block")
        o["bpo"] = np.array([tc[i].bulk_phase_offset for i in range(n_cells)])
        return o

    def track_stream_reset(self):
        self._chk(self._lib.lcs_track_stream_reset(self._h), "lcs_track_stream_reset")
        if hasattr(self, "_trk_stream_cells"):
            del self._trk_stream_cells

    def track_stats(self, n_cells, n_sym, want_ac_td=True):
        """Display statistics (do_ac_fd, do_ac_td, do_pss_sss_sigpower_ce) of the block the last track_block call
        processed; see lcs_track_stats in include/lcs.h.  -> dict(ac_fd [c][4][max_rs][12], ac_td [c][4][max_rs][72],
        sync [c][max_hf][4] = (tp, sp, np, np_blank), sync_ce [c][max_hf][72], n_hf [c])."""
        max_rs, max_hf = n_sym // 3 + 4, n_sym // 60 + 2
        out = dict(ac_fd=np.full((n_cells, 4, max_rs, 12), np.nan + 0j, np.complex128),
                   ac_td=np.full((n_cells, 4, max_rs, 72), np.nan + 0j, np.complex128) if want_ac_td else None,
                   sync=np.full((n_cells, max_hf, 4), np.nan), sync_ce=np.full((n_cells, max_hf, 72), np.nan + 0j, np.complex128),
                   n_hf=np.zeros(n_cells, np.int32))
        rc = self._lib.lcs_track_stats(self._h, n_cells, n_sym, _dp(out["ac_fd"]), _dp(out["ac_td"]), max_rs, _dp(out["sync"]),
                                       _dp(out["sync_ce"]), max_hf, _ip(out["n_hf"]))
        self._chk(rc, "lcs_track_stats")
        return out

    # ---- streaming mode (LTE-Tracker's searcher thread, src/searcher_thread.cpp:83-246) ----
    def stream_open(self, fmt: int, n_cap: int, fc_requested: float, fc_programmed: float, fs_programmed: float):
        """Capture the one-buffer, single-hypothesis chain as a hipGraph (see include/lcs.h)."""
        self._chk(self._lib.lcs_stream_open(self._h, fmt, n_cap, fc_requested, fc_programmed, fs_programmed), "lcs_stream_open")
        self._stream_fmt, self._stream_n = fmt, n_cap

    def stream_push(self, samples, f_off: float, tracked=()):
        """samples: host array, uint8 I/Q (2*n_cap) or complex64 (n_cap).  Returns immediately."""
        a = np.ascontiguousarray(samples, dtype=np.uint8 if self._stream_fmt == FMT_IQ_U8 else np.complex64)
        assert a.size == (2 if self._stream_fmt == FMT_IQ_U8 else 1) * self._stream_n
        t = np.ascontiguousarray(np.asarray(list(tracked), dtype=np.int16))
        self._chk(self._lib.lcs_stream_push(self._h, a.ctypes.data_as(C.c_void_p), float(f_off),
                                            t.ctypes.data_as(C.POINTER(C.c_int16)), int(t.size)), "lcs_stream_push")

    def stream_collect(self, max_cells: int = 16):
        """-> (new cells, number of tracked cells seen again, GPU milliseconds of the pass)."""
        cells = (LcsCell * max_cells)()
        n, dup, ms = C.c_int(0), C.c_int(0), C.c_float(0)
        self._note_overflow(self._chk(self._lib.lcs_stream_collect(self._h, cells, max_cells, C.byref(n), C.byref(dup), C.byref(ms)),
                                      "lcs_stream_collect", allow_overflow=True), "lcs_stream_collect")
        return [cells[i].copy() for i in range(min(n.value, max_cells))], dup.value, ms.value

    def stream_close(self):
        self._chk(self._lib.lcs_stream_close(self._h), "lcs_stream_close")

    def last_xcorr_ms(self):
        ms, n = C.c_float(0), C.c_int(0)
        self._chk(self._lib.lcs_last_xcorr_ms(self._h, C.byref(ms), C.byref(n)), "lcs_last_xcorr_ms")
        return ms.value, n.value

    def last_frq_repairs(self) -> int:
        """Positions of xc_incoherent_collapsed_frq the last correlation call recomputed in the reference's arithmetic
        (near-ties of the arg-max, lcs_last_frq_repairs)."""
        n = C.c_int(0)
        self._chk(self._lib.lcs_last_frq_repairs(self._h, C.byref(n)), "lcs_last_frq_repairs")
        return n.value

    def last_frq_repair_stats(self):
        """-> (positions listed as near-ties, positions left unrepaired by the work bound) of the last correlation call
        (lcs_last_frq_repair_stats; the second is 0 on any real data)."""
        a, b = C.c_int(0), C.c_int(0)
        self._chk(self._lib.lcs_last_frq_repair_stats(self._h, C.byref(a), C.byref(b)), "lcs_last_frq_repair_stats")
        return a.value, b.value

    def last_batch_stats(self):
        """Counters of the last collected batch (lcs_last_batch_stats): dict with cells_past_sss and pbch_candidates_decoded among them."""
        a = (C.c_int * 8)()
        self._chk(self._lib.lcs_last_batch_stats(self._h, a), "lcs_last_batch_stats")
        return dict(records=a[0], overflow=a[1], cells_last_round=a[4], cells_past_sss=a[5], redetected=a[6], pbch_candidates_decoded=a[7])

    def last_collect_host_us(self) -> float:
        """Host microseconds the last batch_collect spent outside its wait for the GPU (lcs_last_collect_host_us)."""
        us = C.c_double(0)
        self._chk(self._lib.lcs_last_collect_host_us(self._h, C.byref(us)), "lcs_last_collect_host_us")
        return us.value

    def last_xcorr_info(self):
        """-> (kernel name, matrix-core operations executed by the last enqueue's correlation launches)."""
        ops, name = C.c_double(0), C.c_char_p()
        self._chk(self._lib.lcs_last_xcorr_info(self._h, C.byref(ops), C.byref(name)), "lcs_last_xcorr_info")
        return (name.value or b"").decode(), ops.value

    def batch_collect_raw(self, n_buf: int, max_cells_per_buf: int = 16):
        """Like batch_collect but without building Python objects: (records [n_buf][max_cells] as a numpy structured
        array with lcs_cell's layout, counts [n_buf])."""
        rec = np.zeros((n_buf, max_cells_per_buf), capi.cell_dtype())
        cnt = np.zeros(n_buf, np.int32)
        rc = self._lib.lcs_batch_collect(self._h, rec.ctypes.data_as(C.POINTER(LcsCell)), max_cells_per_buf,
                                         cnt.ctypes.data_as(C.POINTER(C.c_int)))
        self._note_overflow(self._chk(rc, "lcs_batch_collect", allow_overflow=True), "lcs_batch_collect")
        return rec, cnt

    def sync(self):
        self._chk(self._lib.lcs_sync(self._h), "lcs_sync")


def device_count() -> int:
    """GPUs the library can use (lcs_device_count)."""
    return capi.load().lcs_device_count()


def kalibrate(searcher: "Searcher", capbuf, fc_requested: float, fc_programmed: float, fs_programmed: float,
              ppm: float = 120.0, correction: float = 1.0):
    """LO calibration step of LTE-Tracker (src/LTE-Tracker.cpp:565-741) on one capture buffer: search
    the +-ppm grid shifted by the current correction (:586), take the strongest decoded cell (:712-722)
    and return (best cell, residual frequency offset, correction_residual of :724-731), or None when no
    cell was decoded (the reference then tries again on a fresh capture)."""
    n_extra = int(np.floor((fc_requested * ppm / 1e6 + 2.5e3) / 5e3))
    f = (fc_requested * correction - fc_requested) + np.arange(-n_extra, n_extra + 1) * 5000.0
    cells, _ = searcher.search_capbuf(capbuf, f, fc_requested, fc_programmed, fs_programmed)
    if not cells:
        return None
    best = max(cells, key=lambda c: c.pss_pow)
    crystal_freq_actual = fc_programmed - best.freq_superfine
    return best, best.freq_superfine, (fc_requested / fc_requested * fc_programmed) / crystal_freq_actual


def z_th1(sp_incoherent, n_comb_xc, ds_comb_arm=DS_COMB_ARM, thresh1_n_nines=12):
    """Detection threshold of the CLI main loop (src/CellSearch.cpp:500-503)."""
    L = capi.load()
    R_th1 = L.lcs_chi2cdf_inv(1 - pow(10.0, -thresh1_n_nines), 2.0 * n_comb_xc * (2 * ds_comb_arm + 1))
    rx_cutoff = (6 * 12 * 15e3 / 2 + 4 * 15e3) / (FS_LTE / 16 / 2)
    return R_th1 * np.asarray(sp_incoherent) / rx_cutoff / 137 / 2 / n_comb_xc / (2 * ds_comb_arm + 1)


# ---- table accessors (used by tests) ------------------------------------------------
def table_pss_td(n_id_2):
    o = np.empty(137, np.complex128)
    capi.load().lcs_table_pss_td(n_id_2, _dp(o))
    return o


def table_pss_fd(n_id_2):
    o = np.empty(62, np.complex128)
    capi.load().lcs_table_pss_fd(n_id_2, _dp(o))
    return o


def table_sss_fd(n_id_1, n_id_2, slot):
    o = np.empty(62, np.int32)
    capi.load().lcs_table_sss_fd(n_id_1, n_id_2, slot, _ip(o))
    return o


def table_lte_pn(c_init, n):
    o = np.empty(n, np.uint8)
    capi.load().lcs_table_lte_pn(c_init, n, o.ctypes.data_as(C.POINTER(C.c_uint8)))
    return o


def chi2cdf_inv(p, k):
    return capi.load().lcs_chi2cdf_inv(p, k)
