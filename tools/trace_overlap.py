#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of the pipelined bench: how long each small kernel overlaps a correlation
launch, the gaps between correlation launches, and a least-squares fit  duration = a + sum_k b_k * overlap_k  (what
a microsecond of each co-running kernel costs the correlation).   python tools/trace_overlap.py <..._kernel_trace.csv>"""
import csv, sys
import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows), key=lambda x: x[1])
xc = [e for e in ev if e[0].startswith("k_xcorr_i8x3")]
xc = xc[len(xc) // 2:]                  # steady-state half of the run
names = sorted({e[0] for e in ev if not e[0].startswith("k_xcorr")})
small = {nm: np.array([(s, e) for n, s, e in ev if n == nm]) for nm in names}
X, y = [], []
for _, s, e in xc:
    X.append([np.clip(np.minimum(e, small[nm][:, 1]) - np.maximum(s, small[nm][:, 0]), 0, None).sum() / 1e3 for nm in names])
    y.append((e - s) / 1e3)
X, y = np.array(X), np.array(y)
print(f"correlation launches: {len(y)}, duration us mean {y.mean():.0f} min {y.min():.0f} max {y.max():.0f}")
gaps = [(xc[i + 1][1] - xc[i][2]) / 1e3 for i in range(len(xc) - 1)]
print(f"gap between consecutive correlation launches us: mean {np.mean(gaps):.1f} median {np.median(gaps):.1f}")
print(f"small-kernel time overlapping one correlation launch (sum over kernels): {X.sum(1).mean():.0f} us")
coef, *_ = np.linalg.lstsq(np.c_[np.ones(len(y)), X], y, rcond=None)
print(f"fit: duration = {coef[0]:.0f} us + ...")
for nm, m, c in sorted(zip(names, X.mean(0), coef[1:]), key=lambda t: -t[1]):
    print(f"  {nm[:34]:34s} overlap {m:7.1f} us   slope {c:6.2f}   -> {m * c:7.1f} us")
