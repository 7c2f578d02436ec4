#!/usr/bin/env python3
"""Developer A/B runner (GPU box): tools/ab.py TAG 'bench args' ... -> one compact line per run in gpurun_out/ab_TAG.txt"""
import json, os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = os.path.join(root, "gpurun_out", f"ab_{tag}.txt")
os.makedirs(os.path.dirname(out), exist_ok=True)
for spec in sys.argv[2:]:
    t0 = time.time()
    p = subprocess.run(f"timeout 240 python {root}/bench.py --no-cpu-baseline --no-power-probe --synth-cache /tmp/synth {spec}", shell=True, capture_output=True, text=True, cwd="/tmp")
    line = f"{spec:60s} "
    try:
        j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        c, r = j["config"], j.get("roofline", {})
        line += (f"value {j['value']:9.0f}  ms/batch64 {1e3 * 64 / j['value']:.4f}  ms_per_batch {c.get('ms_per_batch', 0):.3f}  xc_in {r.get('kernel_ms', 0):.3f} "
                 f"xc_iso {r.get('kernel_ms_isolated', 0):.3f}  verified {j.get('verified')}  dense {(c.get('dense_band') or {}).get('ms_per_batch', 0):.3f}  "
                 f"step_ms {'/'.join('%.1f' % v for v in (c.get('step_ms') or {}).values())}  pbch/cell {c.get('pbch_candidates_decoded_per_cell_past_sss')} cells/buf {c.get('cells_past_sss_per_buffer')}")
    except Exception as e:
        line += f"FAILED rc={p.returncode} {e!r} :: {p.stderr[-400:]!r}"
    line += f"  [{time.time() - t0:.0f}s]"
    print(line, flush=True)
    open(out, "a").write(line + "\n")
