set -u
O=gpurun_out/r05l; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
python - <<'PY'
import os, sys, struct, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
from conftest import load_pkg, golden, iq_u8_to_capbuf
pkg = load_pkg()
g = golden("capbuf_0000"); cap = iq_u8_to_capbuf(g["iq_u8"]); fc, FS = float(g["fc"][0]), 1.92e6
with pkg.Searcher(0) as S:
    found, _ = S.search_capbuf(cap, np.array([30e3, 35e3, 40e3]), fc, fc, FS)
per = []
for c in found:
    kf = (fc - c.freq_superfine) / fc
    per.append(pkg.tracker.cut_symbols(cap, c.frame_start * (30.72e6 / 16) / (FS * kf), c.cp_type, c.freq_superfine, fc, fc, FS, 980))
C = 64
cells = [found[i % 2] for i in range(C)]
td = np.stack([per[i % 2][0] for i in range(C)]); late = np.stack([per[i % 2][1] for i in range(C)])
ftv = np.stack([per[i % 2][2] for i in range(C)]); fov = np.stack([per[i % 2][3] for i in range(C)])
tc = (pkg.capi.LcsTrackCell * C)()
for i, c in enumerate(cells):
    for fld in ("n_id_1", "n_id_2", "cp_type", "n_ports", "n_rb_dl", "phich_duration", "phich_resource"):
        setattr(tc[i], fld, int(getattr(c, fld)))
with open("/tmp/blk.trkblock", "wb") as fh:
    fh.write(struct.pack("<ii3d", C, 980, fc, fc, FS)); fh.write(bytes(tc))
    for a in (fov, ftv, late): fh.write(np.ascontiguousarray(a, np.float64).tobytes())
    fh.write(np.ascontiguousarray(td, np.complex128).tobytes())
PY
for n in 1 2 3 4 6 8; do ./host/TrackBench /tmp/blk.trkblock $n 400 20 >> $O/trackbench.txt 2>&1; done
TL=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
echo "--- with torch's bundled HIP runtime ($TL)" >> $O/trackbench.txt
for n in 2 4; do LD_LIBRARY_PATH=$TL ./host/TrackBench /tmp/blk.trkblock $n 400 20 >> $O/trackbench.txt 2>&1; done
