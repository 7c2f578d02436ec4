"""World size 8 before the hardware sees it: every multi-rank driver run with EIGHT ranks on the one GPU a test box has
(gloo for the collectives, `--share-gpu0`): rank arithmetic of the carrier shard (rank 7's carriers), the shape of the
per-step record all-gather (8 x (1 + 5 MAXREC)), the per-rank identity check, eight per-rank rates; the sweep tool's table at
world 8 equal to world 1; host/CellSearch with eight device threads byte-identical to one.  No scaling curve comes from
this -- the eight ranks share one GPU -- it is the logic that an 8-GPU node then only has to run faster."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden, iq_u8_to_capbuf, load_pkg

pytestmark = pytest.mark.gpu

ENV = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
ENV.update(MASTER_ADDR="127.0.0.1", GPU_MAX_HW_QUEUES="8", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")


def _json_line(stdout):
    return json.loads([l for l in stdout.splitlines() if l.startswith("{")][-1])


def test_bench_world_8_on_one_gpu():
    """`python bench.py --gpus 8` starts eight ranks itself; every rank owns its carriers FC + 100 kHz * (r B + b), keeps two
    batches in flight, packs its step's records, and the asynchronous all-gather brings 8 x (1 + 5 MAXREC) doubles to everyone."""
    B, K, steps = 8, 2, 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu0", "--dist-backend", "gloo", "--steps", str(steps),
                        "--warmup", "1", "--batch", str(B), "--batches-per-step", str(K), "--no-cpu-baseline", "--no-dense", "--no-power-probe"],
                       env=dict(ENV, MASTER_PORT="29571"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                            # rank 0 alone prints
    j = json.loads(lines[0])
    c = j["config"]
    assert j["n_gpus"] == 8 and j["verified"] is True and j["scaling"] == "weak"
    assert len(c["per_rank_buffers_per_s"]) == 8 and len(c["devices"]) == 8
    assert c["buffers_timed"] == 8 * B * K * steps
    assert j["value"] <= sum(c["per_rank_buffers_per_s"]) * 1.001    # whole job / the slowest rank's time
    coll = c["collectives"]
    assert coll["world"] == 8 and coll["backend"] == "gloo" and len(coll["gathered_records_last_step"]) == 8
    maxrec = max(64, B) * K
    assert coll["gather_out_shape"] == [8, 1 + 5 * maxrec]
    # every rank found its planted cells (buffer 0 of every batch of 8 is occupied) and names only its own carriers
    FC = 739e6
    for rank, (n, rng) in enumerate(zip(coll["gathered_records_last_step"], coll["gathered_fc_min_max_per_rank"])):
        assert n >= K, (rank, n)
        lo, hi = FC + 100e3 * rank * B, FC + 100e3 * (rank * B + B - 1)
        assert lo <= rng[0] <= rng[1] <= hi, (rank, rng, lo, hi)
    assert coll["gathered_fc_min_max_per_rank"][7][0] >= FC + 100e3 * 56


def test_sweep_tool_world_8_equals_world_1():
    """tools/sweep_cellsearch.py over 24 carriers: eight ranks of three carriers each (block-cyclic), one all-gather of the
    raw records, the merged table equal to the single-rank table."""
    tool = os.path.join(ROOT, "tools", "sweep_cellsearch.py")
    args = ["-s", "738e6", "-e", "740.3e6", "--occupied-every", "5", "--json"]
    a = subprocess.run([sys.executable, tool] + args, env=ENV, capture_output=True, text=True, timeout=900)
    assert a.returncode == 0, a.stderr[-2000:]
    b = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                        "--master-port", "29573", tool] + args + ["--share-gpu0", "--dist-backend", "gloo"], env=ENV,
                       capture_output=True, text=True, timeout=1500)
    assert b.returncode == 0, b.stderr[-2000:]
    ja, jb = _json_line(a.stdout), _json_line(b.stdout)
    assert ja["cells"] == jb["cells"] and ja["carriers"] == jb["carriers"] == 24 and len(ja["cells"]) >= 3
    assert jb["n_gpus"] == 8 and ja["n_gpus"] == 1


def test_cellsearch_cli_eight_device_threads(tmp_path):
    """host/CellSearch -g 0,0,0,0,0,0,0,0: eight device threads (two contexts each) share the one GPU; batches go to them
    block-cyclically and the report must be byte-identical to -g 0."""
    pkg = load_pkg()
    it = __import__("importlib").import_module("lte_cell_scanner_amd.itfile")
    g = golden("capbuf_0000")
    cap = iq_u8_to_capbuf(g["iq_u8"])
    rng = np.random.default_rng(8)
    fc0 = int(g["fc"][0])
    n = 18
    for k in range(n):                     # carriers 2, 9, 16 hold the recorded capture, the rest noise: more batches (-B 1) than threads
        if k % 7 == 2:
            buf = cap
        else:
            buf = iq_u8_to_capbuf(np.clip(np.rint(rng.normal(127.0, 12.0, g["iq_u8"].size)), 0, 255).astype(np.uint8))
        it.write_it(str(tmp_path / f"capbuf_{k:04d}.it"), {"capbuf": buf, "fc": np.array([fc0 + 100000 * (k - 2)], np.int32)})
    exe = os.path.join(ROOT, "host", "CellSearch")
    base = [exe, "-s", str(fc0 - 200000), "-e", str(fc0 + 100000 * (n - 3)), "-l", "-d", str(tmp_path)]
    one = subprocess.run(base + ["-g", "0", "-B", "1"], capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    eight = subprocess.run(base + ["-g", "0,0,0,0,0,0,0,0", "-B", "1"], capture_output=True, text=True, timeout=1500)
    assert eight.returncode == 0, eight.stderr[-2000:]
    assert one.stdout == eight.stdout and "277" in one.stdout and "271" in one.stdout


def test_contexts_give_their_memory_back():
    """Twelve contexts created, driven through differently shaped calls (single buffers, batches as bytes and floats with the probe on, a wide
    grid, the streaming graph, tracker block + cutter) and destroyed: the device's free memory returns to where it was -- every buffer a
    context grows (per-cell stages, operand tables, tracker / cutter workspaces, the float-batch bytes, graphs) goes with it."""
    import numpy as np
    import torch
    from conftest import golden, iq_u8_to_capbuf, f_search_set_for, load_pkg
    pkg = load_pkg()
    iq = golden("capbuf_0000")["iq_u8"]
    cap = iq_u8_to_capbuf(iq)
    FC, FS = 739e6, 1.92e6
    d8 = torch.from_numpy(np.ascontiguousarray(np.stack([iq] * 6))).cuda()
    d32 = torch.from_numpy(np.stack([cap.astype(np.complex64)] * 6)).cuda()
    td = torch.empty((2, 300, 128), dtype=torch.complex128, device="cuda")
    torch.cuda.synchronize()

    def one(k):
        with pkg.Searcher(0) as S:
            f = f_search_set_for(FC, 100) if k % 3 else f_search_set_for(2.6e9, 120)
            fcs = np.full(6, FC)
            if k % 2:
                S.set_float_batch_probe(True)
            S.search_batch(d8.data_ptr(), pkg.FMT_IQ_U8, 6, 153600, f, fcs, fcs, FS, pkg.STAGE_FULL)
            S.search_batch(d32.data_ptr(), pkg.FMT_C64, 3 + k % 3, 153600, f, fcs[:3 + k % 3], fcs[:3 + k % 3], FS, pkg.STAGE_FULL)
            cells, _ = S.search_capbuf(cap[:153600 - 1000 * k], np.array([30e3, 35e3, 40e3]), FC, FC, FS)
            S.stream_open(pkg.FMT_IQ_U8, 153600, FC, FC, FS)
            S.stream_push(iq, 35e3)
            S.stream_collect()
            S.stream_close()
            tr = [c for c in cells if c.n_rb_dl > 0][:2]
            if len(tr) == 2:
                late, n_cut = S.track_cut(d8.data_ptr(), pkg.FMT_IQ_U8, 153600, [c.cp_type for c in tr], [100.0, 7000.5], [c.freq_superfine for c in tr], FC, FC, FS,
                                          300, td.data_ptr())
                S.track_block(tr, None, np.zeros((2, 300)) + 35e3, np.zeros((2, 300)), late, FC, FC, FS, td_device_ptr=td.data_ptr(), n_sym=300, want_ce=False)

    one(0)                                  # first use pays for one-off allocations of the runtime itself
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for k in range(1, 13):
        one(k)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert abs(free1 - free0) < 64 << 20, (free0, free1)


def test_searcher_and_tracker_contexts_side_by_side_from_two_threads():
    """A searcher context (full-chain batches) and a tracker context (cutter + block) driven from two host threads at once -- the shape of
    LTE-Tracker's searcher and tracker threads on one GPU: every result identical to the same calls made alone."""
    import threading
    import numpy as np
    import torch
    from conftest import golden, iq_u8_to_capbuf, f_search_set_for, load_pkg
    pkg = load_pkg()
    iq = golden("capbuf_0000")["iq_u8"]
    cap = iq_u8_to_capbuf(iq)
    FC, FS = 739e6, 1.92e6
    f = f_search_set_for(FC, 100)
    fcs = np.full(8, FC)
    d8 = torch.from_numpy(np.ascontiguousarray(np.stack([np.roll(iq, 2 * 1000 * k) for k in range(8)]))).cuda()
    td = torch.empty((2, 600, 128), dtype=torch.complex128, device="cuda")
    torch.cuda.synchronize()
    with pkg.Searcher(0) as A, pkg.Searcher(0) as B:
        cells, _ = A.search_capbuf(cap, np.array([30e3, 35e3, 40e3]), FC, FC, FS)
        tr = [c for c in cells if c.n_rb_dl > 0]
        assert len(tr) == 2
        fts = [c.frame_start * (30.72e6 / 16) / (FS * ((FC - c.freq_superfine) / FC)) for c in tr]
        fos = [c.freq_superfine for c in tr]

        def search():
            r = A.search_batch(d8.data_ptr(), pkg.FMT_IQ_U8, 8, 153600, f, fcs, fcs, FS, pkg.STAGE_FULL)
            return [[bytes(c) for c in b] for b in r]

        def track():
            late, n_cut = B.track_cut(d8.data_ptr(), pkg.FMT_IQ_U8, 153600, [c.cp_type for c in tr], fts, fos, FC, FC, FS, 600, td.data_ptr())
            o = B.track_block(tr, None, np.repeat(np.array(fos)[:, None], 600, 1), np.repeat(np.array(fts)[:, None], 600, 1), late, FC, FC, FS,
                              td_device_ptr=td.data_ptr(), n_sym=600, want_ce=False)
            return late.tobytes(), o["syms"].tobytes(), o["mib_ok"].tobytes(), o["meas"][:, :2, :100].tobytes()

        want_s, want_t = search(), track()
        got = {"s": [], "t": []}
        ts = [threading.Thread(target=lambda: got["s"].extend(search() for _ in range(20))), threading.Thread(target=lambda: got["t"].extend(track() for _ in range(40)))]
        for t in ts: t.start()
        for t in ts: t.join()
        assert len(got["s"]) == 20 and all(x == want_s for x in got["s"])
        assert len(got["t"]) == 40 and all(x == want_t for x in got["t"])
        assert sum(len(b) for b in want_s) >= 8
