"""ctypes view of the C ABI declared in include/lcs.h (liblcs_amd.so).

The shared library is the product; this module only loads it and declares prototypes.
There is no fallback: if the library is missing or no MI355X is visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblcs_amd.so")

LCS_OK = 0
FMT_C64, FMT_IQ_U8, FMT_C128 = 0, 1, 2      # FMT_C128: lcs_track_cut only
STAGE_PSS, STAGE_FULL = 1, 3
MAX_PEAKS = 104            # LCS_MAX_PEAKS: the longest list peak_search can return (include/lcs.h)
ERRORS = {-1: "LCS_ERR_NO_DEVICE", -2: "LCS_ERR_BAD_ARG", -3: "LCS_ERR_HIP", -4: "LCS_ERR_OVERFLOW", -5: "LCS_ERR_NOMEM"}


class LcsCell(C.Structure):
    """lcs_cell: POD mirror of the reference's class Cell (include/common.h.in:101-129)."""
    _fields_ = [
        ("fc_requested", C.c_double), ("fc_programmed", C.c_double), ("pss_pow", C.c_double),
        ("freq", C.c_double), ("frame_start", C.c_double), ("freq_fine", C.c_double),
        ("freq_superfine", C.c_double),
        ("ind", C.c_int32), ("n_id_2", C.c_int32), ("n_id_1", C.c_int32), ("cp_type", C.c_int32),
        ("n_ports", C.c_int32), ("n_rb_dl", C.c_int32), ("phich_duration", C.c_int32),
        ("phich_resource", C.c_int32), ("sfn", C.c_int32), ("reserved", C.c_int32),
    ]

    def n_id_cell(self) -> int:          # src/common.cpp:29-31
        return self.n_id_2 + 3 * self.n_id_1 if (self.n_id_1 >= 0 and self.n_id_2 >= 0) else -1

    def n_symb_dl(self) -> int:          # src/common.cpp:32-34
        return 7 if self.cp_type == 1 else (6 if self.cp_type == 2 else -1)

    def as_dict(self) -> dict:
        return {f: getattr(self, f) for f, _ in self._fields_ if f != "reserved"}

    def copy(self) -> "LcsCell":
        o = LcsCell()
        C.memmove(C.byref(o), C.byref(self), C.sizeof(LcsCell))
        return o

    def __repr__(self) -> str:  # pragma: no cover
        return "LcsCell(" + ", ".join(f"{k}={v}" for k, v in self.as_dict().items()) + ")"


def cell_dtype():
    """numpy structured dtype with the layout of lcs_cell (zero-copy views of result arrays)."""
    import numpy as np
    m = {C.c_double: "<f8", C.c_int32: "<i4"}
    return np.dtype([(n, m[t]) for n, t in LcsCell._fields_])


class LcsTrackCell(C.Structure):
    """lcs_track_cell: what the tracker's per-symbol pipeline reads of a tracked cell (include/lcs.h)."""
    _fields_ = [("n_id_1", C.c_int32), ("n_id_2", C.c_int32), ("cp_type", C.c_int32), ("n_ports", C.c_int32),
                ("n_rb_dl", C.c_int32), ("phich_duration", C.c_int32), ("phich_resource", C.c_int32), ("reserved", C.c_int32),
                ("bulk_phase_offset", C.c_double)]


EXPORTS = [
    "lcs_create", "lcs_destroy", "lcs_last_error", "lcs_version", "lcs_cell_init", "lcs_set_max_cells_in_flight", "lcs_set_float_batch_probe",
    "lcs_xcorr_pss", "lcs_peak_search", "lcs_sss_detect", "lcs_pss_sss_foe", "lcs_extract_tfg", "lcs_tfoec",
    "lcs_decode_mib", "lcs_chan_est", "lcs_search_capbuf", "lcs_search_batch_dev", "lcs_search_batch_host", "lcs_batch_enqueue",
    "lcs_batch_collect", "lcs_batch_readback", "lcs_batch_enqueue_host", "lcs_host_alloc", "lcs_host_free", "lcs_device_alloc", "lcs_device_free", "lcs_device_upload", "lcs_device_count",
    "lcs_foe_partial", "lcs_foe_finish", "lcs_foe_contend", "lcs_foe_resolve", "lcs_track_block", "lcs_track_stats", "lcs_track_stream_block", "lcs_track_stream_reset", "lcs_track_cut", "lcs_stream_open", "lcs_stream_push", "lcs_stream_collect", "lcs_stream_close",
    "lcs_last_xcorr_ms", "lcs_last_xcorr_info", "lcs_last_frq_repairs", "lcs_last_frq_repair_stats", "lcs_last_batch_stats", "lcs_last_collect_host_us", "lcs_stream", "lcs_sync", "lcs_table_pss_td", "lcs_table_pss_fd", "lcs_table_sss_fd",
    "lcs_table_lte_pn", "lcs_chi2cdf_inv",
]

_lib = None


def load() -> C.CDLL:
    """Load liblcs_amd.so (built by __graft_entry__.build()); raises if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                           "there is no CPU fallback for the searcher path")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64; when torch
    # is installed it must be loaded first so that this library binds to the same runtime
    # (torch tensors' device pointers are handed to lcs_batch_enqueue).
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional plumbing, not a dependency of the C ABI
        pass
    L = C.CDLL(LIB_PATH)
    vp, dp, ip, fp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_float)
    cp, u16p = C.POINTER(LcsCell), C.POINTER(C.c_uint16)
    L.lcs_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.lcs_destroy.argtypes = [vp]
    L.lcs_destroy.restype = None
    L.lcs_last_error.argtypes = [vp]
    L.lcs_last_error.restype = C.c_char_p
    L.lcs_version.restype = C.c_char_p
    L.lcs_cell_init.argtypes = [cp]
    L.lcs_cell_init.restype = None
    L.lcs_set_max_cells_in_flight.argtypes = [vp, C.c_int]
    L.lcs_xcorr_pss.argtypes = [vp, dp, C.c_uint32, dp, C.c_uint16, C.c_uint8, C.c_double, C.c_double, C.c_double,
                                dp, ip, fp, fp, dp, fp, dp, u16p, u16p]
    L.lcs_peak_search.argtypes = [vp, dp, ip, dp, dp, C.c_uint16, C.c_double, C.c_double, fp, C.c_uint8, cp, C.c_int,
                                  C.POINTER(C.c_int)]
    L.lcs_sss_detect.argtypes = [vp, cp, dp, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_double, cp,
                                 dp, dp, dp, dp, dp, dp, dp, dp]
    L.lcs_pss_sss_foe.argtypes = [vp, cp, dp, C.c_uint32, C.c_double, C.c_double, C.c_double, cp]
    L.lcs_extract_tfg.argtypes = [vp, cp, dp, C.c_uint32, C.c_double, C.c_double, C.c_double, dp, dp, C.POINTER(C.c_int)]
    L.lcs_tfoec.argtypes = [vp, cp, dp, dp, C.c_int, C.c_double, C.c_double, dp, dp, cp]
    L.lcs_decode_mib.argtypes = [vp, cp, dp, C.c_int, cp]
    L.lcs_chan_est.argtypes = [vp, cp, dp, C.c_int, C.c_int, dp, dp]
    L.lcs_search_capbuf.argtypes = [vp, dp, C.c_uint32, dp, C.c_uint16, C.c_double, C.c_double, C.c_double,
                                    cp, C.c_int, C.POINTER(C.c_int), cp, C.c_int, C.POINTER(C.c_int)]
    L.lcs_search_batch_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.c_uint32, dp, C.c_uint16, dp, dp, C.c_double,
                                       C.c_int, cp, C.c_int, C.POINTER(C.c_int)]
    L.lcs_search_batch_host.argtypes = L.lcs_search_batch_dev.argtypes
    L.lcs_batch_readback.argtypes = [vp, C.c_int, fp, dp, ip, dp, dp]
    L.lcs_batch_enqueue.argtypes = [vp, vp, C.c_int, C.c_int, C.c_uint32, dp, C.c_uint16, dp, dp, C.c_double, C.c_int]
    L.lcs_batch_enqueue_host.argtypes = L.lcs_batch_enqueue.argtypes
    L.lcs_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.lcs_host_free.argtypes = [vp, vp]
    if hasattr(L, "lcs_device_alloc"):
        L.lcs_device_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        L.lcs_device_free.argtypes = [vp, vp]
        L.lcs_device_upload.argtypes = [vp, vp, vp, C.c_size_t]
    L.lcs_device_count.argtypes = []
    L.lcs_batch_collect.argtypes = [vp, cp, C.c_int, C.POINTER(C.c_int)]
    L.lcs_foe_partial.argtypes = [vp, dp, C.c_uint32, dp, C.c_uint16, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, vp, vp]
    L.lcs_foe_finish.argtypes = [vp, vp, vp, dp, C.c_uint16, cp, ip, C.c_int, C.POINTER(C.c_int), cp, C.c_int, C.POINTER(C.c_int)]
    if hasattr(L, "lcs_foe_contend"):
        L.lcs_foe_contend.argtypes = [vp, dp, C.c_uint16, vp, vp]
        L.lcs_foe_resolve.argtypes = [vp, vp, vp]
    L.lcs_track_block.argtypes = [vp, C.POINTER(LcsTrackCell), C.c_int, C.c_int, vp, C.c_int, dp, dp, dp, C.c_double, C.c_double, C.c_double,
                                  dp, dp, dp, ip, dp, C.c_int, ip, ip, C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_float)]
    L.lcs_track_stats.argtypes = [vp, C.c_int, C.c_int, dp, dp, C.c_int, dp, dp, C.c_int, ip]
    i64p = C.POINTER(C.c_int64)
    L.lcs_track_stream_block.argtypes = [vp, C.POINTER(LcsTrackCell), C.c_int, C.c_int, vp, dp, dp, dp, C.c_double, C.c_double, C.c_double,
                                         dp, dp, dp, C.c_int, i64p, ip, dp, dp, dp, C.c_int, ip, ip, C.POINTER(C.c_uint64), C.c_int, i64p, ip]
    L.lcs_track_stream_reset.argtypes = [vp]
    if hasattr(L, "lcs_set_float_batch_probe"):
        L.lcs_set_float_batch_probe.argtypes = [vp, C.c_int]
    if hasattr(L, "lcs_track_cut"):
        L.lcs_track_cut.argtypes = [vp, vp, C.c_int, C.c_uint32, C.c_double, C.c_int, ip, dp, dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                    C.c_double, C.c_double, C.c_double, C.c_int, vp, dp, ip, C.POINTER(C.c_int64)]
    L.lcs_stream_open.argtypes = [vp, C.c_int, C.c_uint32, C.c_double, C.c_double, C.c_double]
    L.lcs_stream_push.argtypes = [vp, vp, C.c_double, C.POINTER(C.c_int16), C.c_int]
    L.lcs_stream_collect.argtypes = [vp, cp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]
    L.lcs_stream_close.argtypes = [vp]
    L.lcs_last_xcorr_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.lcs_last_xcorr_info.argtypes = [vp, dp, C.POINTER(C.c_char_p)]
    if hasattr(L, "lcs_last_frq_repairs"):      # (absent from older developer builds loaded through bench.py --lib)
        L.lcs_last_frq_repairs.argtypes = [vp, C.POINTER(C.c_int)]
        L.lcs_last_collect_host_us.argtypes = [vp, dp]
    if hasattr(L, "lcs_last_batch_stats"):
        L.lcs_last_batch_stats.argtypes = [vp, C.POINTER(C.c_int)]
    if hasattr(L, "lcs_last_frq_repair_stats"):
        L.lcs_last_frq_repair_stats.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.lcs_stream.argtypes = [vp]
    L.lcs_stream.restype = vp
    L.lcs_sync.argtypes = [vp]
    L.lcs_table_pss_td.argtypes = [C.c_int, dp]
    L.lcs_table_pss_fd.argtypes = [C.c_int, dp]
    L.lcs_table_sss_fd.argtypes = [C.c_int, C.c_int, C.c_int, ip]
    L.lcs_table_lte_pn.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8)]
    L.lcs_chi2cdf_inv.argtypes = [C.c_double, C.c_double]
    L.lcs_chi2cdf_inv.restype = C.c_double
    _lib = L
    return L
