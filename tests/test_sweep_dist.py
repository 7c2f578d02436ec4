"""The multi-GPU paths on CPU, world_size 2 over gloo: (1) the carrier sweep sharded over ranks + one all-gather of
cell records, (2) the frequency-hypothesis split of ONE buffer with its packed MAX-with-index all-reduce.  The
per-buffer searcher is stood in for by the CPU oracle here (tests only) -- what is under test is the sharding, the
record packing, the collectives, the tie-break and dedup.  The `-m gpu` tests at the bottom run the same drivers with
the real GPU searcher under two ranks sharing one GPU."""
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


FOE_WORKER = textwrap.dedent("""
    import os, sys, json, hashlib
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle")); sys.path.insert(0, os.path.join({root!r}, "tests"))
    from conftest import load_pkg, golden, iq_u8_to_capbuf
    pkg = load_pkg()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    use_gpu = {use_gpu!r}
    if world > 1:
        dist.init_process_group("gloo")
    if use_gpu:
        S = pkg.Searcher(0)
        stages = pkg.sweep.SearcherStages(S, pkg.z_th1)
    else:
        import oracle as O
        O.set_threads(2)
        stages = O
    cap = iq_u8_to_capbuf(golden("capbuf_0000")["iq_u8"])
    f = np.array([25e3, 30e3, 35e3, 35e3, 40e3, 45e3, 50e3])        # duplicate hypothesis: the tie must go to the lower index
    if {dev!r}:      # everything between correlation and peak search stays on the GPU (lcs_foe_partial / lcs_foe_finish)
        torch.cuda.set_device(0)
        cells, peaks = pkg.sweep.search_capbuf_foe_split_dev(S, cap, f, 739e6, 739e6, 1.92e6, rank, world, dist if world > 1 else None, torch.device("cuda", 0))
        arr = dict(pow=np.array([p.pss_pow for p in peaks]), frq=np.array([p.freq for p in peaks]))
    else:
        cells, arr = pkg.sweep.search_capbuf_foe_split(stages, cap, f, 739e6, 739e6, 1.92e6, rank, world, dist if world > 1 else None)
    if rank == 0:
        print("RESULT " + json.dumps(dict(cells=[(c["n_id_cell"], c["n_rb_dl"], c["sfn"], c["ind"], c["freq"], repr(c["pss_pow"]), repr(c["freq_superfine"])) for c in cells],
                                          pow=hashlib.sha256(arr["pow"].tobytes()).hexdigest(), frq=hashlib.sha256(arr["frq"].tobytes()).hexdigest(),
                                          n_dup_wins=int(np.count_nonzero(arr["frq"] == 3)))))
    if world > 1:
        dist.destroy_process_group()
""")


def _run(world, tmp_path, worker=None, port="29541", **fmt):
    if worker is FOE_WORKER:
        fmt.setdefault("dev", False)
    script = tmp_path / "worker.py"
    script.write_text((worker or WORKER).format(root=ROOT, **fmt))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2", GPU_MAX_HW_QUEUES="8")
    if world == 1:
        env.update(RANK="0", WORLD_SIZE="1")
        cmd = [sys.executable, str(script)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", port, str(script)]
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


def test_pack_unpack_and_tie_break():
    """bits(pow) << 32 | ~foi: MAX over ranks = the reference's first maximum over the frequency axis (:369-382)."""
    sw = load_pkg().sweep
    pow_a, frq_a = np.array([[0.5, 0.25, 0.0, 1e-30]]), np.array([[4, 2, 0, 1]])
    pow_b, frq_b = np.array([[0.5, 0.26, 0.0, 1e-30]]), np.array([[1, 7, 3, 0]])
    w = np.maximum(sw.pack_pow_frq(pow_a, frq_a), sw.pack_pow_frq(pow_b, frq_b))
    p, f = sw.unpack_pow_frq(w)
    assert np.array_equal(f, [[1, 7, 0, 0]]) and np.array_equal(p, np.array([[0.5, 0.26, 0.0, 1e-30]]).astype(np.float32).astype(np.float64))
    assert [list(b) for b in sw.foe_blocks(7, 2)] == [[0, 1, 2, 3], [4, 5, 6]] and [len(b) for b in sw.foe_blocks(3, 8)].count(0) == 5
    assert np.array_equal(np.concatenate(sw.foe_blocks(37, 8)), np.arange(37))


def test_foe_split_two_ranks_equal_single_rank(tmp_path):
    """Collapsed (pow, frq) after the packed all-reduce are bit-identical to the single-rank arrays, the duplicate
    hypothesis (index 3 == index 2) never wins, and the decoded cells come back in the single-rank order."""
    one = _run(1, tmp_path, FOE_WORKER, "29543", use_gpu=False)
    two = _run(2, tmp_path, FOE_WORKER, "29543", use_gpu=False)
    assert one == two
    assert one["n_dup_wins"] == 0 and [c[0] for c in one["cells"]] == [277, 271] and [c[2] for c in one["cells"]] == [74, 22]


@pytest.mark.gpu
def test_foe_split_two_ranks_real_searcher(tmp_path):
    one = _run(1, tmp_path, FOE_WORKER, "29545", use_gpu=True)
    two = _run(2, tmp_path, FOE_WORKER, "29545", use_gpu=True)
    assert one == two and [c[0] for c in one["cells"]] == [277, 271]


@pytest.mark.gpu
def test_foe_split_device_resident(tmp_path):
    """lcs_foe_partial / lcs_foe_finish: the packed words and the power estimate stay in device tensors, torch.distributed
    reduces them in place (gloo on CUDA tensors here, two ranks sharing GPU 0; RCCL on a real node), and the result is
    the single-rank result -- which in turn is what the ordinary fused chain finds."""
    one = _run(1, tmp_path, FOE_WORKER, "29549", use_gpu=True, dev=True)
    two = _run(2, tmp_path, FOE_WORKER, "29549", use_gpu=True, dev=True)
    host = _run(1, tmp_path, FOE_WORKER, "29549", use_gpu=True)
    assert one == two and [c[0] for c in one["cells"]] == [277, 271] and [c[2] for c in one["cells"]] == [74, 22]
    for a, b in zip(one["cells"], host["cells"]):
        assert a[:5] == b[:5] and abs(float(a[5]) / float(b[5]) - 1) < 1e-5 and abs(float(a[6]) - float(b[6])) < 1e-2


@pytest.mark.gpu
def test_foe_split_two_contexts_emulate_two_ranks():
    """The same without processes: two contexts take one share of the hypotheses each, the MAX of their word tensors
    stands in for the all-reduce; every peak is refined by exactly one of them and the union is the fused chain's list.
    Includes a rank with NO hypotheses (world > n_f)."""
    import torch
    import oracle as O
    from conftest import golden, iq_u8_to_capbuf, f_search_set_for
    pkg = load_pkg()
    cap = iq_u8_to_capbuf(golden("capbuf_0000")["iq_u8"])
    fc, fs = 739e6, 1.92e6
    f = f_search_set_for(fc, 120)        # 37 hypotheses: this buffer then has two near-ties of the arg-max (none with the 31 of ppm 100)
    O.set_threads(8)
    exp, exp_peaks = O.search_capbuf(cap, f, fc, fc, fs)
    shares = [(0, 19), (19, 18), (0, 0)]
    with pkg.Searcher(0) as A, pkg.Searcher(0) as B, pkg.Searcher(0) as Z:
        ctxs = [A, B, Z]
        words = [torch.empty(3 * 9600, dtype=torch.int64, device="cuda") for _ in ctxs]
        meta = [torch.empty(9601, dtype=torch.float64, device="cuda") for _ in ctxs]
        for S, (a, n), w, m in zip(ctxs, shares, words, meta):
            S.foe_partial(cap, f, a, n, fc, fc, fs, w.data_ptr(), m.data_ptr())
        assert int(words[2].max()) == -1                       # the empty share never wins
        red = torch.maximum(torch.maximum(words[0], words[1]), words[2])
        # round 5: near-ties of the arg-max -- inside a share or across the two -- are settled in the reference's arithmetic: every
        # context packs the exact maxima of the positions it contends for, their MAX replaces the approximate words
        w2 = [torch.empty(3 * 9600, dtype=torch.int64, device="cuda") for _ in ctxs]
        for S, x in zip(ctxs, w2):
            S.foe_contend(f, red.data_ptr(), x.data_ptr())
        assert int(w2[2].max()) == -1
        red2 = torch.maximum(torch.maximum(w2[0], w2[1]), w2[2])
        reds = [red.clone() for _ in ctxs]
        for S, x in zip(ctxs, reds):
            S.foe_resolve(x.data_ptr(), red2.data_ptr())
        assert torch.equal(reds[0], reds[1]) and torch.equal(reds[0], reds[2])
        ro = O.xcorr_pss(cap, f, 2, fc, fc, fs)
        pw, fq = pkg.sweep.unpack_pow_frq(reds[0].cpu().numpy().reshape(3, 9600))
        settled = (red2.cpu().numpy().reshape(3, 9600) >= 0)
        assert settled.sum() >= 1 and np.array_equal(fq, ro["frq"])                      # EVERY index equal to the oracle's
        assert np.array_equal(pw[settled], ro["pow"][settled])                            # and the settled powers are the reference's own floats
        assert (np.abs(pw - ro["pow"]) / ro["pow"]).max() < 1e-5
        # the same buffer split differently, and not split at all: identical words
        for shares_b in ([(0, 7), (7, 30)], [(0, 37)]):
            wb = [torch.empty(3 * 9600, dtype=torch.int64, device="cuda") for _ in shares_b]
            for S, (a, n), w in zip(ctxs, shares_b, wb):
                S.foe_partial(cap, f, a, n, fc, fc, fs, w.data_ptr(), meta[1].data_ptr())
            rb = wb[0] if len(wb) == 1 else torch.maximum(wb[0], wb[1])
            wb2 = [torch.empty(3 * 9600, dtype=torch.int64, device="cuda") for _ in shares_b]
            for S, x in zip(ctxs, wb2):
                S.foe_contend(f, rb.data_ptr(), x.data_ptr())
            rb2 = wb2[0] if len(wb2) == 1 else torch.maximum(wb2[0], wb2[1])
            ctxs[0].foe_resolve(rb.data_ptr(), rb2.data_ptr())
            assert torch.equal(rb, reds[0]), shares_b
        # (back to the first split for the per-peak stages: the contexts' partial state is the last partial call's)
        for S, (a, n), w, m in zip(ctxs, shares, words, meta):
            S.foe_partial(cap, f, a, n, fc, fc, fs, w.data_ptr(), m.data_ptr())
        got, seen = [], []
        for S, red in zip(ctxs, reds):
            cells, order, peaks = S.foe_finish(red.data_ptr(), meta[0].data_ptr(), f)
            got += list(zip(order.tolist(), cells))
            seen.append([(p.n_id_2, p.freq, p.reserved) for p in peaks])
            assert [(p.n_id_2, p.freq) for p in peaks] == [(p.n_id_2, p.freq) for p in exp_peaks]
        # each peak is owned (reserved == 0) by exactly one context; nobody owns anything on the empty rank
        for k in range(len(exp_peaks)):
            assert sum(1 for s in seen if s[k][2] == 0) == 1
        assert all(x[2] == 1 for x in seen[2])
        got.sort(key=lambda x: x[0])
        assert [c.n_id_cell() for _, c in got] == [c.n_id_cell() for c in exp] == [277, 271]
        for (_, a), b in zip(got, exp):
            assert (a.ind, a.n_id_1, a.n_ports, a.n_rb_dl, a.sfn) == (b.ind, b.n_id_1, b.n_ports, b.n_rb_dl, b.sfn)
            assert abs(a.pss_pow - b.pss_pow) < 1e-5 * b.pss_pow and abs(a.freq_superfine - b.freq_superfine) < 1e-3
        with pytest.raises(pkg.SearcherError):
            A.foe_finish(red.data_ptr(), meta[0].data_ptr(), f)          # no partial call pending


@pytest.mark.gpu
def test_two_rank_sweep_real_searcher_shared_gpu(tmp_path):
    """tools/sweep_cellsearch.py with two ranks sharing GPU 0 (gloo) against one rank: same table, real Searcher."""
    import json
    tool = os.path.join(ROOT, "tools", "sweep_cellsearch.py")
    args = ["-s", "738e6", "-e", "740.3e6", "--occupied-every", "5", "--json"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", GPU_MAX_HW_QUEUES="8")
    a = subprocess.run([sys.executable, tool] + args, env=env, capture_output=True, text=True, timeout=600)
    assert a.returncode == 0, a.stderr[-2000:]
    b = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", tool] + args + ["--share-gpu0", "--dist-backend", "gloo"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    ja = json.loads([l for l in a.stdout.splitlines() if l.startswith("{")][-1])
    jb = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1])
    assert ja["cells"] == jb["cells"] and ja["carriers"] == jb["carriers"] == 24 and len(ja["cells"]) >= 3


def test_bench_gpus_flag_must_match_the_launcher():
    """bench.py --gpus N is the number of ranks: a launcher that started a different number is an error, not a silent
    single-GPU run (`--gpus 8` used to time one GPU and print n_gpus 1).  No GPU needed: it exits first."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "the two must agree" in r.stderr


@pytest.mark.gpu
def test_bench_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher: bench.py re-executes itself through torch.distributed.run with two ranks
    (here sharing GPU 0 over gloo, the only way two ranks fit a one-GPU box) and prints ONE line that says n_gpus 2, carries
    both ranks' rates, and whose value is their sum over the slower rank's time."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu0", "--dist-backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--batch", "8", "--batches-per-step", "3", "--no-cpu-baseline", "--no-dense"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["verified"] is True and len(j["config"]["per_rank_buffers_per_s"]) == 2
    assert j["config"]["buffers_timed"] == 2 * 8 * 3 * 2 and len(j["config"]["devices"]) == 2
    assert j["value"] <= sum(j["config"]["per_rank_buffers_per_s"]) * 1.001
    # without --share-gpu0 two ranks on a one-GPU box are refused (each rank must own a GPU)
    import torch
    if torch.cuda.device_count() == 1:
        r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "8",
                             "--batches-per-step", "2", "--no-cpu-baseline", "--no-dense"], env=env, capture_output=True, text=True, timeout=600)
        assert r2.returncode != 0 and "GPU(s) visible" in (r2.stderr + r2.stdout)
