#!/usr/bin/env python3
"""Per-kernel register / LDS footprint of the HIP sources (the compiler's view): what decides which of the small
kernels fit on a CU beside two resident correlation workgroups.  Usage: python tools/kernel_resources.py [hipcc flags]"""
import os, re, subprocess, sys

csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lte-cell-scanner_amd", "csrc")
pat = re.compile(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)")
for f in ("pss_xcorr", "pss_xcorr_i8", "pss_xcorr_f16", "peak_search", "sss_foe", "tfg_mib", "tracker"):
    src = os.path.join(csrc, f + ".hip")
    if not os.path.exists(src):
        continue
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c",
                        "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", src, "-o", "/dev/null"] + sys.argv[1:],
                       capture_output=True, text=True)
    cur = {}
    for line in p.stderr.splitlines():
        m = pat.search(line)
        if not m:
            continue
        k, v = m.group(1).split()[0], m.group(2)
        if k == "Function":
            cur = {"name": v}
        else:
            cur[k] = v
        if k == "LDS":
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.split("(")[0].strip()
            print("%-44s VGPR %4s AGPR %4s scratch %4s occ %2s LDS %6s" % (name[:44], cur.get("VGPRs"), cur.get("AGPRs"),
                                                                          cur.get("ScratchSize"), cur.get("Occupancy"), cur.get("LDS")))
