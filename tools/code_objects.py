#!/usr/bin/env python3
"""Per-kernel metadata of the gfx950 code objects INSIDE a built library (the shipped artefact, not a recompilation): registers,
spills, private segment (scratch), LDS.  The library's `.hip_fatbin` holds one uncompressed clang offload bundle per source file.
Usage: python tools/code_objects.py [path/to/liblcs_amd.so]          (also imported by tests/test_tables_abi.py)"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FIELDS = ("agpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count",
          "vgpr_spill_count", "max_flat_workgroup_size")


def kernels_of(lib):
    """-> {demangled-ish kernel name: {field: int}} for every gfx950 kernel of `lib`."""
    d = open(lib, "rb").read()
    out = {}
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", d):
        base = m.start()
        n = struct.unpack_from("<Q", d, base + 24)[0]
        p = base + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", d, p)
            p += 24
            triple = d[p:p + tl].decode()
            p += tl
            if "gfx950" not in triple or size == 0:
                continue
            with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
                f.write(d[base + off: base + off + size])
                path = f.name
            try:
                txt = subprocess.run([READELF, "--notes", path], capture_output=True, text=True, timeout=120).stdout
            finally:
                os.unlink(path)
            cur = None
            for line in txt.splitlines():
                mm = re.match(r"^\s*(-\s+)?\.(\w+):\s*(\S*)\s*$", line)
                if not mm:
                    continue
                dash, k, v = mm.group(1), mm.group(2), mm.group(3)
                if k == "agpr_count" and dash:          # first key of a kernel record (keys are sorted)
                    cur = {"agpr_count": int(v)}
                elif cur is not None and k in FIELDS:
                    cur[k] = int(v)
                elif cur is not None and k == "symbol":      # (vgpr_count / vgpr_spill_count / wavefront_size follow: the record stays open)
                    out[v[:-3] if v.endswith(".kd") else v] = cur
    return out


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "lte-cell-scanner_amd", "liblcs_amd.so")
    ks = kernels_of(lib)
    names = subprocess.run(["c++filt"] + list(ks), capture_output=True, text=True).stdout.splitlines()
    for sym, nm in sorted(zip(ks, names), key=lambda x: x[1]):
        k = ks[sym]
        print("%-46s VGPR %3d AGPR %3d spills %3d scratch %4d LDS %6d" % (nm.split("(")[0][:46], k["vgpr_count"], k["agpr_count"], k["vgpr_spill_count"],
                                                                         k["private_segment_fixed_size"], k["group_segment_fixed_size"]))
    print(len(ks), "kernels")
