"""The collective code of the multi-GPU paths executed over RCCL (torch.distributed backend "nccl") on the ONE GPU a test box
has: world size 1 with the process group live.  Every collective the 8-GPU run issues is issued here on device tensors --
the device-identity all-gather, the per-step asynchronous all-gather of the cell records, the timing all-gather and MAX
all-reduce of bench.py; the record all-gather of the carrier sweep; the int64 MAX all-reduce, the fp64 broadcast and the
byte all-gather of the hypothesis split -- so the first multi-GPU run is not their first execution.  (Two ranks cannot
share one GPU under RCCL: it refuses duplicate devices.  The two-rank logic is covered over gloo in test_sweep_dist.py.)"""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

ENV = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
ENV.update(MASTER_ADDR="127.0.0.1", GPU_MAX_HW_QUEUES="8", HSA_ENABLE_IPC_MODE_LEGACY="0")


def _json_line(stdout):
    return json.loads([l for l in stdout.splitlines() if l.startswith("{")][-1])


def test_bench_forced_dist_over_rccl():
    """bench.py --gpus 1 --force-dist --dist-backend nccl takes the world > 1 branch with one rank: same verified result,
    collectives reported, and a rate close to the plain run's (the all-gather is asynchronous and waited for a step later)."""
    steps, K = 4, 12
    common = ["--steps", str(steps), "--warmup", "1", "--batch", "64", "--batches-per-step", str(K), "--no-cpu-baseline", "--no-dense"]
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, env=ENV, capture_output=True, text=True, timeout=900)
    assert plain.returncode == 0, plain.stderr[-3000:]
    forced = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--dist-backend", "nccl"] + common,
                            env=dict(ENV, MASTER_PORT="29561"), capture_output=True, text=True, timeout=900)
    assert forced.returncode == 0, forced.stderr[-3000:]
    a, b = _json_line(plain.stdout), _json_line(forced.stdout)
    assert a["verified"] is True and b["verified"] is True and b["n_gpus"] == 1
    assert a["config"]["collectives"] is None
    c = b["config"]["collectives"]
    assert c["backend"] == "nccl" and c["world"] == 1 and len(c["gathered_records_last_step"]) == 1
    # the records of the last step went through the all-gather: as many as the run found per step
    assert c["gathered_records_last_step"][0] == sum(b["config"]["cells_per_distinct_batch"][d % 4] for d in range(steps * K - K, steps * K))
    assert len(b["config"]["devices"]) == 1 and "RCCL" in b["config"]["parallelism"]
    assert b["config"]["cells_per_distinct_batch"] == a["config"]["cells_per_distinct_batch"]
    # round 4 found the multi-rank branch at 0.80 x (per-buffer Python packing of the gathered records); short runs are noisy,
    # the 1 % comparison at full size is profiles/r04/bench_forced_dist_n1.json
    assert b["value"] > 0.88 * a["value"], (a["value"], b["value"])


WORKER = textwrap.dedent("""
    import os, sys, json, hashlib
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    from conftest import load_pkg, golden, iq_u8_to_capbuf
    pkg = load_pkg()
    sw = pkg.sweep
    live = {live!r}
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if live:
        dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    d = dist if live else None
    out = {{}}
    g = golden("capbuf_0000")["iq_u8"]
    cap = iq_u8_to_capbuf(g)
    with pkg.Searcher(0) as S:
        # (1) hypothesis split, device-resident: int64 MAX all-reduce + fp64 broadcast on device tensors, byte all-gather
        f = np.array([25e3, 30e3, 35e3, 35e3, 40e3, 45e3, 50e3])
        cells, peaks = sw.search_capbuf_foe_split_dev(S, cap, f, 739e6, 739e6, 1.92e6, 0, 1, d, dev)
        out["foe_dev"] = [(c["n_id_cell"], c["n_rb_dl"], c["sfn"], c["ind"], c["freq"], repr(c["pss_pow"]), repr(c["freq_superfine"])) for c in cells]
        out["peaks"] = [(p.n_id_2, p.ind, p.freq) for p in peaks]
        # (2) the host-array driver of the same split with device tensors for the collectives
        cells2, arr = sw.search_capbuf_foe_split(sw.SearcherStages(S, pkg.z_th1), cap, f, 739e6, 739e6, 1.92e6, 0, 1, d, dev if live else None)
        out["foe_host"] = [(c["n_id_cell"], c["n_rb_dl"], c["sfn"], c["ind"], c["freq"]) for c in cells2]
        out["pow"] = hashlib.sha256(arr["pow"].tobytes()).hexdigest()
        out["frq"] = hashlib.sha256(arr["frq"].tobytes()).hexdigest()
        # (3) carrier sweep: five carriers through the batch API, one byte all-gather of the records on the device
        rng = np.random.default_rng(7)
        noise = np.clip(np.rint(rng.normal(127.0, 12.0, g.size)), 0, 255).astype(np.uint8)
        fcs = sw.fc_search_set(738.9e6, 739.3e6)
        res = torch.from_numpy(np.stack([noise, g, g, noise, noise])).to(dev)
        fset = pkg.f_search_set_for(739e6, 100)
        def search_fn(bufs, fc):
            S.batch_enqueue(bufs.data_ptr(), pkg.FMT_IQ_U8, len(fc), g.size // 2, fset, fc, fc, 1.92e6, pkg.STAGE_FULL)
            return S.batch_collect_raw(len(fc), sw.MAXC)
        final, detected = sw.run_sweep(search_fn, lambda idx: res[int(idx[0]): int(idx[0]) + len(idx)], fcs, 0, 1, d, dev if live else None, batch=2)
        out["sweep_final"] = [(c["n_id_cell"], c["fc_requested"], c["n_rb_dl"], repr(c["pss_pow"])) for c in final]
        out["sweep_per"] = [[c["n_id_cell"] for c in dd] for dd in detected]
    if live:
        out["backend"] = dist.get_backend()
        dist.destroy_process_group()
    print("RESULT " + json.dumps(out))
""")


def _worker(tmp_path, live, port):
    script = tmp_path / f"rccl_worker_{int(live)}.py"
    script.write_text(WORKER.format(root=ROOT, live=live))
    r = subprocess.run([sys.executable, str(script)], env=dict(ENV, MASTER_PORT=port, RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def test_sweep_and_foe_split_collectives_over_rccl(tmp_path):
    """run_sweep, search_capbuf_foe_split and search_capbuf_foe_split_dev with the RCCL process group live give what they
    give without one, and that is the recorded capture's two cells."""
    off = _worker(tmp_path, False, "29563")
    on = _worker(tmp_path, True, "29563")
    assert on.pop("backend") == "nccl"
    assert on == off
    assert [c[0] for c in on["foe_dev"]] == [277, 271] and [c[2] for c in on["foe_dev"]] == [74, 22]
    assert [tuple(c[:5]) for c in on["foe_dev"]] == [tuple(c) for c in on["foe_host"]]
    assert on["sweep_per"] == [[], [277, 271], [277, 271], [], []] and sorted(c[0] for c in on["sweep_final"]) == [271, 277]


def test_sweep_tool_forced_dist_over_rccl():
    """tools/sweep_cellsearch.py --force-dist --dist-backend nccl: the tool's own process-group set-up and device all-gather."""
    tool = os.path.join(ROOT, "tools", "sweep_cellsearch.py")
    args = ["-s", "738e6", "-e", "739.5e6", "--occupied-every", "5", "--json"]
    a = subprocess.run([sys.executable, tool] + args, env=ENV, capture_output=True, text=True, timeout=600)
    assert a.returncode == 0, a.stderr[-2000:]
    b = subprocess.run([sys.executable, tool] + args + ["--force-dist", "--dist-backend", "nccl"], env=dict(ENV, MASTER_PORT="29565"),
                       capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    ja, jb = _json_line(a.stdout), _json_line(b.stdout)
    assert ja["collective"] is None and jb["collective"] == "nccl all-gather, world 1"
    assert ja["cells"] == jb["cells"] and ja["carriers"] == jb["carriers"] == 16 and len(ja["cells"]) >= 2
