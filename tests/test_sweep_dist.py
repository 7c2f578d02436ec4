"""The multi-GPU path (carrier sweep sharded over ranks + one all-gather of cell records) on CPU:
world_size 2 over gloo.  The per-buffer searcher is stood in for by the CPU oracle here (tests
only) -- what is under test is the sharding, the record packing, the collective and dedup."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT, load_pkg


def test_shard_and_dedup_logic():
    pkg = load_pkg()
    sw = pkg.sweep
    assert list(sw.shard(7, 0, 2)) == [0, 2, 4, 6] and list(sw.shard(7, 1, 2)) == [1, 3, 5]
    assert np.array_equal(np.sort(np.concatenate([sw.shard(531, r, 8) for r in range(8)])), np.arange(531))
    assert len(sw.fc_search_set(715e6, 768e6)) == 531          # BASELINE config 4
    a = dict(n_id_cell=277, fc_requested=739e6, freq_superfine=35e3, pss_pow=0.06)
    b = dict(n_id_cell=277, fc_requested=739.1e6, freq_superfine=-65e3, pss_pow=0.09)   # same cell seen from the next carrier
    c = dict(n_id_cell=271, fc_requested=739e6, freq_superfine=35e3, pss_pow=0.01)
    d = dict(n_id_cell=277, fc_requested=751e6, freq_superfine=0.0, pss_pow=0.5)        # same ID, 12 MHz away: kept
    out = sw.dedup([[a, c], [b], [d]])
    assert [(x["n_id_cell"], x["fc_requested"]) for x in out] == [(277, 739.1e6), (271, 739e6), (277, 751e6)]


WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle")); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import oracle as O
    from conftest import load_pkg, golden, iq_u8_to_capbuf
    pkg = load_pkg()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if world > 1:
        dist.init_process_group("gloo")
    O.set_threads(2)
    g = golden("capbuf_0000")["iq_u8"]
    rng = np.random.default_rng(7)
    noise = np.clip(np.rint(rng.normal(127.0, 12.0, g.size)), 0, 255).astype(np.uint8)
    fcs = pkg.sweep.fc_search_set(738.9e6, 739.3e6)            # 5 carriers
    bufs = [noise, g, g, noise, noise]                         # the recorded cells show up on two adjacent carriers
    f = np.array([30e3, 35e3, 40e3])
    def get_capbufs(idx): return np.stack([bufs[i] for i in idx])
    def search_fn(b, fc):                                      # TEST stand-in for Searcher.search_batch
        return [O.search_capbuf(iq_u8_to_capbuf(x), f, c, c, 1.92e6)[0] for x, c in zip(b, fc)]
    final, detected = pkg.sweep.run_sweep(search_fn, get_capbufs, fcs, rank, world, dist if world > 1 else None, batch=2)
    if rank == 0:
        print("RESULT " + json.dumps(dict(final=[(c["n_id_cell"], c["fc_requested"], c["n_rb_dl"]) for c in final],
                                          per=[[c["n_id_cell"] for c in d] for d in detected])))
    if world > 1:
        dist.destroy_process_group()
""")


def _run(world, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", OMP_NUM_THREADS="2")
    if world == 1:
        env.update(RANK="0", WORLD_SIZE="1")
        cmd = [sys.executable, str(script)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    import json
    return json.loads(line[7:])


def test_two_rank_sweep_matches_single_rank(tmp_path):
    one = _run(1, tmp_path)
    two = _run(2, tmp_path)
    assert one == two
    assert one["per"][0] == [] and one["per"][3] == [] and one["per"][4] == []
    assert one["per"][1] == [277, 271] and one["per"][2] == [277, 271]
    # each cell is reported once after dedup (the two sightings are 100 kHz apart)
    assert sorted(c[0] for c in one["final"]) == [271, 277] and all(c[2] == 50 for c in one["final"])
