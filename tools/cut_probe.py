#!/usr/bin/env python3
"""Developer probe (GPU box): lcs_track_cut on the recorded capture for 64 cells x 980 symbols, 30 calls -- run under
rocprofv3 --kernel-trace --stats for the three kernels' durations (tools/cut_probe.py; profiles/r06/kernel_stats_track_cut.csv)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
g = np.load(os.path.join(ROOT, "tests", "golden", "capbuf_0000.npz"))
iq = np.ascontiguousarray(g["iq_u8"])
C, n_sym, FS, fc = 64, 980, 1.92e6, float(g["fc"][0])
rng = np.random.default_rng(1)
cps = [1 + (i % 2) for i in range(C)]
fts = rng.uniform(0, 19200, C)
fos = rng.uniform(-40e3, 40e3, C)
d = torch.from_numpy(iq).cuda()
td = torch.empty((C, n_sym, 128), dtype=torch.complex128, device="cuda")
with pkg.Searcher(0) as S:
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            late, n_cut = S.track_cut(d.data_ptr(), pkg.FMT_IQ_U8, iq.size // 2, cps, fts, fos, fc, fc, FS, n_sym, td.data_ptr())
        torch.cuda.synchronize()
        print("ms per call", 1e2 * (time.perf_counter() - t0), "n_cut", n_cut.min(), n_cut.max())
