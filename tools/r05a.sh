set -u
O=gpurun_out/r05a; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/pytest.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5) > $O/smoke.log
python - > $O/repairs.log 2>&1 <<'PY'
import numpy as np, torch, sys, time
import __graft_entry__ as ge
pkg = ge.load_package()
f = pkg.f_search_set_for(739e6, 100)
fcs = 739e6 + 100e3*np.arange(128)
host = pkg.synth.make_batch_u8(128, 1234, fcs)
d = torch.from_numpy(host).cuda()
with pkg.Searcher(0) as S:
    for st in (pkg.STAGE_PSS, pkg.STAGE_FULL):
        S.batch_enqueue(d.data_ptr(), pkg.FMT_IQ_U8, 128, 153600, f, fcs, fcs, 1.92e6, st)
        rec, cnt = S.batch_collect_raw(128, 16)
        print("stage", st, "repairs per 128-buffer batch:", S.last_frq_repairs(), "cells", int(cnt.sum()))
PY
python tools/ab.py r05a '--steps 10 --warmup 3 --lib build_exp/liblcs_r04.so' '--steps 10 --warmup 3' '--steps 10 --warmup 3 --lib build_exp/liblcs_r04.so' '--steps 10 --warmup 3'
