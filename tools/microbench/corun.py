#!/usr/bin/env python3
"""What slows the PSS correlation kernel when other work shares the GPU?  (developer experiment, GPU box)

The correlation launch is timed alone (one context, enqueue + collect, lcs_last_xcorr_ms = HIP events on its stream) while
a side thread keeps ONE kind of synthetic load running on its own stream (tools/microbench/sideload.hip): fp64 ALU
waves, "fat" one-wave workgroups that cannot share a CU slot with two resident correlation workgroups, streaming reads
/ writes through L2 to HBM, L2-resident reads, bursts of empty workgroups (dispatcher pressure).  Every load runs
unconfined and confined to 16 CUs (CU-masked stream): a slowdown that survives the confinement is not competition for
CU slots but pressure on something the whole chip shares."""
import ctypes as C
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libsideload.so")
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "sideload.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", SO, os.path.join(HERE, "sideload.hip")])
pkg = ge.load_package()
L = C.CDLL(SO)
L.sl_stream.restype = C.c_void_p
L.sl_stream.argtypes = [C.c_int]
L.sl_alloc.restype = C.c_void_p
L.sl_alloc.argtypes = [C.c_size_t]
for fn, at in (("sl_alu64", [C.c_void_p, C.c_int, C.c_int, C.c_void_p]), ("sl_fat", [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
               ("sl_read", [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]), ("sl_write", [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
               ("sl_empty", [C.c_void_p, C.c_int, C.c_int]), ("sl_sync", [C.c_void_p])):
    getattr(L, fn).argtypes = at

N_CAP, FS, FC, B = 153600, 1.92e6, 739e6, 64
f = pkg.f_search_set_for(FC, 100.0)
fcs = FC + 100e3 * np.arange(B)
cache = "/tmp/synth/batch_64_1234.npy"
host = np.load(cache) if os.path.exists(cache) else pkg.synth.make_batch_u8(B, 1234, fcs)
d = torch.from_numpy(host).cuda()
ctx = pkg.Searcher(0)
big = L.sl_alloc(512 << 20)
scratch = L.sl_alloc(1 << 20)


def xcorr_ms(n=30):
    ms = []
    for i in range(n):
        ctx.batch_enqueue(d.data_ptr(), pkg.FMT_IQ_U8, B, N_CAP, f, fcs, fcs, FS, pkg.STAGE_PSS)
        ctx.batch_collect_raw(B, 16)
        ms.append(ctx.last_xcorr_ms()[0])
    return float(np.median(ms[5:]))


LOADS = {
    "alu64 x64 waves": lambda s: L.sl_alu64(s, 64, 20000, scratch),
    "alu64 x1024 waves": lambda s: L.sl_alu64(s, 1024, 20000, scratch),
    "fat x64 wg (48 dregs + 32 KB LDS)": lambda s: L.sl_fat(s, 64, 2000, scratch),
    "fat x512 wg": lambda s: L.sl_fat(s, 512, 2000, scratch),
    "read 256 MB (HBM)": lambda s: L.sl_read(s, 256, big, 256 << 20, scratch),
    "read 2 MB x64 (L2 resident)": lambda s: [L.sl_read(s, 256, big, 2 << 20, scratch) for _ in range(64)],
    "write 256 MB": lambda s: L.sl_write(s, 256, big, 256 << 20),
    "empty 4096 wg x50 launches": lambda s: L.sl_empty(s, 4096, 50),
}

print("correlation alone: %.3f ms" % xcorr_ms(), flush=True)
for n_cu in (0, 16):
    s = L.sl_stream(n_cu)
    if not s:
        print("CU-masked stream not available"); continue
    for name, fn in LOADS.items():
        stop, cnt = False, [0]

        def bg():
            while not stop:
                fn(s)
                L.sl_sync(s)
                cnt[0] += 1
        t = threading.Thread(target=bg)
        t.start()
        time.sleep(0.1)
        c0, t0 = cnt[0], time.perf_counter()
        r = xcorr_ms()
        rate = (cnt[0] - c0) / (time.perf_counter() - t0)
        stop = True
        t.join()
        print("%-8s %-36s: correlation %.3f ms   (side load: %.0f iterations/s)" % ("16 CUs" if n_cu else "all CUs", name, r, rate), flush=True)
print("correlation alone: %.3f ms" % xcorr_ms(), flush=True)
