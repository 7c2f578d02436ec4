#!/usr/bin/env python3
"""Developer probe (GPU box): phase timestamps of workgroup (0,0,0) of the per-cell kernels from a -DLCS_PHASE_TS build.
   python tools/phase_ts.py build_exp/liblcs_phts.so"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
pkg.capi.LIB_PATH = sys.argv[1]
import torch
g = np.load(os.path.join(ROOT, "tests", "golden", "capbuf_0000.npz"))
d = torch.from_numpy(np.ascontiguousarray(g["iq_u8"])).to("cuda:0")
f = np.arange(-15, 16) * 5000.0
fc = float(g["fc"][0])
with pkg.Searcher(0) as S:
    for rep in range(3):
        cells = S.search_batch(d.data_ptr(), pkg.FMT_IQ_U8, 1, 153600, f, fc, fc, 1.92e6, pkg.STAGE_FULL)[0]
    ts = (C.c_ulonglong * 256)()
    lib = pkg.capi.load()
    lib.lcs_debug_phase_ts.argtypes = [C.POINTER(C.c_ulonglong)]
    print("rc", lib.lcs_debug_phase_ts(ts), "cells", [c.n_id_cell() for c in cells])
    t = np.array(ts[:], dtype=np.float64)
    def d_us(a, b): return (t[b] - t[a]) / 100.0          # wall_clock64: 100 MHz
    print("k_pbch   after the trellises: winner reduction %.1f us, retrace forward %.1f us, traceback %.1f us, CRC + stores %.1f us" % (d_us(3, 5), d_us(5, 6), d_us(6, 7), d_us(7, 4)))
    print("k_pbch   LLR phase %.1f us, de-ratematch %.1f us, 64 trellises %.1f us, traceback + CRC %.1f us" % (d_us(0, 1), d_us(1, 2), d_us(2, 3), d_us(3, 4)))
    print("k_tfg    fill %.1f us, FFT + output %.1f us" % (d_us(30, 31), d_us(31, 32)))
    print("k_tfoec  %.1f / %.1f / %.1f us" % (d_us(10, 11), d_us(11, 13), d_us(13, 14)))
    print("k_chan_est  corrections + PBCH rows %.1f | raw estimates %.1f | filter %.1f | noise + first row %.1f | interpolation %.1f us" % (d_us(19, 20), d_us(20, 21), d_us(21, 22), d_us(22, 23), d_us(23, 24)))
