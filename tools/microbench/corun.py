#!/usr/bin/env python3
"""What slows the PSS correlation kernel when other work shares the GPU?  (developer experiment)
Times lcs_last_xcorr_ms of back-to-back STAGE_PSS batches while a side stream runs (a) nothing, (b) a saturating
device-to-device copy (HBM + L2 pressure, few registers), (c) a throttled copy (~800 MB per correlation launch, what
the chain's memory-bound kernels move), (d) small fp64 elementwise kernels on a few workgroups."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
N_CAP, FS, FC, B = 153600, 1.92e6, 739e6, 64
f = pkg.f_search_set_for(FC, 100.0)
fcs = FC + 100e3 * np.arange(B)
host = pkg.synth.make_batch_u8(B, 1234, fcs)
d = torch.from_numpy(host).cuda()
ctxs = [pkg.Searcher(0) for _ in range(2)]

def xcorr_loop(n=40):
    ms = []
    for i in range(n + 1):
        ctxs[i % 2].batch_enqueue(d.data_ptr(), pkg.FMT_IQ_U8, B, N_CAP, f, fcs, fcs, FS, pkg.STAGE_PSS)
        if i >= 1:
            ctxs[(i - 1) % 2].batch_collect_raw(B, 16)
            ms.append(ctxs[(i - 1) % 2].last_xcorr_ms()[0])
    ctxs[n % 2].batch_collect_raw(B, 16)
    return float(np.mean(ms[5:]))

side = torch.cuda.Stream()
stop = False
def bg(kind):
    a = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    x = torch.randn(1 << 16, dtype=torch.float64, device="cuda")
    with torch.cuda.stream(side):
        while not stop:
            if kind == "copy":
                b.copy_(a)
            elif kind == "copy_throttled":
                b[: 200 << 20].copy_(a[: 200 << 20]); side.synchronize(); time.sleep(0.0006)
            elif kind == "fp64_small":
                for _ in range(20):
                    x = torch.sin(x) * 1.0000001 + 0.1
            side.synchronize()

print("alone            : %.3f ms" % xcorr_loop())
for kind in ("copy", "copy_throttled", "fp64_small"):
    stop = False
    t = threading.Thread(target=bg, args=(kind,)); t.start()
    time.sleep(0.2)
    r = xcorr_loop()
    stop = True; t.join()
    print("%-17s: %.3f ms" % ("with " + kind, r))
# PSS-stage chain alone, 2 contexts pipelined, is what "alone" above already is: its small kernels co-run too
