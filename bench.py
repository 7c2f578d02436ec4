#!/usr/bin/env python3
"""bench.py -- capture-buffers/s of the searcher hot path on MI355X (driver contract).

A "step" is one pass of the chain over --batches-per-step (100) batches of --batch (128) synthetic
153600-sample, 1.92 Msps capture buffers that are ALREADY RESIDENT IN HBM when the timed region
starts (12800 buffers per step: the driver's 20 steps time ~6 s of GPU work, long enough for a 5 s
SMI sampler to see the GPU busy).  A batch is one enqueue = one correlation launch; two contexts are kept in
flight (profiles/r03/experiments/batch_size_and_depth.txt: 128 x 2 in flight is 2.5 % faster than 64 x 3).  N=1 workload = BASELINE.json configs[2], the metric's "full
CellSearch": PSS correlation over the full +-100 ppm grid at 739 MHz (n_f = 31), peak_search and every
per-cell stage down to the decoded MIB (--stage pss stops after peak_search; --stage single is
configs[1] as written: ONE host buffer per step through lcs_search_capbuf, PCIe included).
--gpus N: when the process was not started by torch.distributed.run it starts N ranks itself
(python -m torch.distributed.run --nproc-per-node N, RCCL); every rank owns one GPU (checked: N distinct
devices) and processes its own shard of carriers (weak scaling, no data-path collective); the
detected-cell records of a step are all-gathered by ONE asynchronous collective per step that is waited
for a step later, off the critical path.  --input-host feeds the same batches from page-locked HOST
memory (lcs_batch_enqueue_host: PCIe inside the timed region).

The run verifies itself: every batch collected inside the timed, pipelined region must return the
same bytes as a sequential single-context run of the same buffers afterwards, and 16 buffers (12 of
the timed batch, 4 of the dense band) are checked against the CPU oracle ("verified" in the JSON line;
tools/parity_population.py checks every timed buffer and 512 more).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

# more streams than the default 4 hardware queues per process are in flight (3 contexts x 2 streams + the
# collective library): give each its own queue (same single-process throughput; without it two processes
# sharing ONE GPU, as in the gloo test setup, stall in the launch path)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CAP = 153600
FS = 1.92e6
FC = 739e6
PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense FP32 MFMA peak == packed FP32 vector peak
PEAK_BF16_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense BF16 MFMA peak (~2.5 PFLOP/s)
PUBLISHED_BUFFERS_PER_S = 1.0 / 6.0   # ~6 s per centre frequency at ppm 100 (dual-core i7-2640)
PEAK_I8_TOPS = 5000.0         # MI355X_MICROARCH.md: I8 MFMA "~2x bf16 rate" (no spec line; 16x16x64 micro-benchmark ceiling 3944 TOPS)


def synth_batch(pkg, n_buf, seed, fc_list, dense=False):
    """Deterministic synthetic capture buffers as raw RTL-SDR u8 I/Q bytes: a band scan (every 4th carrier holds 1-2
    cells, BASELINE.md: 0-3 per buffer) or, dense=True, a busy band (every carrier holds 2-3 cells)."""
    if dense:
        return pkg.synth.make_batch_u8(n_buf, seed, fc_list, occupied_every=1, n_distinct=8, cells_cycle=(2, 3))
    return pkg.synth.make_batch_u8(n_buf, seed, fc_list)


def cpu_baseline(pkg, host_u8, f, fcs, stage, sample=None):
    """The CPU oracle (a C port of the reference's path, oracle/lcs_oracle.c) on a bounded sample of the same workload --
    the first 8 buffers of the bench batch (two of them occupied, the band scan's own 1-in-4 ratio; 4 buffers, two occupied,
    on grids above 48 hypotheses so that the leg stays within ~30 s) -- single thread, then with OpenMP over the lags as
    the reference does, on this host.  Reported next to the GPU number; never part of `value`.
    Returns (report, the oracle's cells of buffer 0)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    if sample is None:
        sample = list(range(8)) if f.size <= 48 else [0, 1, 4, 5]
    sample = [b for b in sample if b < len(host_u8)]
    n_sample = len(sample)
    caps = []
    for b in sample:
        iq = host_u8[b].astype(np.float64)
        caps.append(((iq[0::2] - 127.0) / 128.0) + 1j * ((iq[1::2] - 127.0) / 128.0))
    fcs = [fcs[b] for b in sample]

    def one(cap, fc):
        if stage == "full":
            return O.search_capbuf(cap, f, fc, fc, FS)[0]
        r = O.xcorr_pss(cap, f, 2, fc, fc, FS)
        return O.peak_search(r["pow"], r["frq"], O.z_th1(r["sp_incoherent"], r["n_comb_xc"]), f, fc, fc, r["single"], 2)

    O.set_threads(1)
    t0 = time.perf_counter()
    res = [one(c, float(fc)) for c, fc in zip(caps, fcs)]
    dt = time.perf_counter() - t0
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:   # a container may be allowed far fewer CPUs than it can see
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            ncpu = max(1, min(ncpu, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    ncpu = min(ncpu, 64)      # one socket's worth of physical cores is where this loop stops scaling
    O.set_threads(ncpu)
    t0 = time.perf_counter()
    for c, fc in zip(caps, fcs):
        one(c, float(fc))
    dt_mt = time.perf_counter() - t0
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    cpu_baseline.results = dict(zip(sample, res))       # the oracle's cells of the sampled buffers (the verification leg reuses them)
    cpu_baseline.threads = ncpu
    return ({"value": n_sample / dt, "unit": "capture-buffers/s", "cores": 1, "kind": "port",
             "sample": f"buffers {sample} of the bench batch ({sum(1 for r in res if r)} of them with detections), n_f={f.size}, stage={stage}, "
                       f"{dt:.2f} s single-thread (C oracle, gcc -O3); {ncpu} threads (OpenMP over lags as the reference): {n_sample / dt_mt:.3f} buffers/s",
             "cpu_model": model, "nproc": os.cpu_count(), "threads_allowed": ncpu, "multi_thread_value": n_sample / dt_mt,
             "n_results": [len(r) for r in res]}, res[0])


def stream_bench(pkg, args, rank, world, local_rank, dist):
    """configs[4]: LTE-Tracker streaming mode, one carrier per GPU ("replicas", no collective in the data
    path).  A step = one 80 ms host buffer pushed through the captured graph and collected."""
    import torch
    fc = FC + 100e3 * rank
    # the tracked carrier: two cells within the pull-in range of the single hypothesis, every other buffer noise only
    occ = pkg.synth.make_capbuf(4321 + rank, fc, [dict(n_id_1=33 + rank, n_id_2=1, f_off=900.0, n_ports=2, n_rb_dl=50),
                                                   dict(n_id_1=120, n_id_2=0, f_off=-700.0, gain_db=-4.0)], 8.0)[0]
    rng = np.random.default_rng(99 + rank)
    host = [occ if i % 2 == 0 else np.clip(np.rint(rng.normal(127.0, 12.0, 2 * N_CAP)), 0, 255).astype(np.uint8) for i in range(4)]
    S = pkg.Searcher(local_rank if world > 1 else 0)
    S.stream_open(pkg.FMT_IQ_U8, N_CAP, fc, fc, FS)
    gpu_ms, n_cells = [], 0
    def step(i, last=False):
        # two buffers in flight (lcs_stream_push is double-buffered): push buffer i, then collect buffer i - 1
        nonlocal n_cells
        S.stream_push(host[i % len(host)], 0.0)
        for _ in range(2 if last else (1 if i else 0)):
            cells, _, ms = S.stream_collect()
            gpu_ms.append(ms)
            n_cells = max(n_cells, len(cells))
    # pre-conditioning (untimed, like the warm-up): a fresh process runs its first ~100 ms of pushes at 0.5 ms each before it
    # settles at the GPU time of the graph (0.35 ms); keep the stream busy for 0.6 s first, as the batch bench does
    t_pre, i = time.perf_counter(), 0
    while time.perf_counter() - t_pre < 0.6:
        step(i, last=False)
        i += 1
    S.stream_collect()
    for i in range(args.warmup):
        step(i, last=(i == args.warmup - 1))
    gpu_ms.clear()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, last=(i == args.steps - 1))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the same buffers through the eager (launch by launch) chain, for the graph-vs-launches comparison
    E = pkg.Searcher(local_rank if world > 1 else 0)
    d_one = torch.from_numpy(np.stack(host)).to(torch.device("cuda", local_rank if world > 1 else 0))
    fz = np.array([0.0])
    for i in range(3):
        E.search_batch(d_one[i % 4].data_ptr(), pkg.FMT_IQ_U8, 1, N_CAP, fz, fc, fc, FS, pkg.STAGE_FULL)
    t0 = time.perf_counter()
    for i in range(20):
        E.search_batch(d_one[i % 4].data_ptr(), pkg.FMT_IQ_U8, 1, N_CAP, fz, fc, fc, FS, pkg.STAGE_FULL)
    eager_ms = 1e3 * (time.perf_counter() - t0) / 20
    E.close()
    if rank == 0:
        value = world * args.steps / dt
        print(json.dumps({
            "metric": "capture-buffers/s (1.92 Msps, 153600-samp) streaming searcher, n_f = 1", "value": value,
            "unit": "capture-buffers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[4]: LTE-Tracker streaming mode, one carrier per GPU, one 80 ms u8 I/Q host "
                                   "buffer per step (PCIe copy included), hipGraph-captured chain, n_f = 1, two buffers in flight",
                       "gpu_ms_per_buffer": float(np.mean(gpu_ms)), "realtime_factor": 0.08 * value / world,
                       "eager_ms_per_buffer_device_resident_input": eager_ms,
                       "n_cells_in_occupied_buffers": n_cells, "parallelism": "replicas" if world > 1 else "single GPU"},
            # launch-bound chain (~30 kernels of 5-50 us in one graph): what is worth reporting is how much of the wall time
            # the GPU is active at all and the (tiny) fraction of the HBM roof the compulsory bytes reach
            "roofline": {"bound": "launch", "achieved": (1651200 + 230400) / (float(np.mean(gpu_ms)) * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": (1651200 + 230400) / (float(np.mean(gpu_ms)) * 1e-3) / 1e9 / 8000.0, "traffic": None,
                         "gpu_active_frac": float(np.mean(gpu_ms)) / (1e3 * dt / args.steps),
                         "note": "SURVEY 8(d) compulsory bytes of one buffer at n_f = 1 (1.88 MB) over the graph's GPU time (HIP events around "
                                 "the graph launch on its stream); gpu_active_frac = that time / wall time per buffer (the rest: 0.3 MB pinned "
                                 "memcpy + PCIe + launch + collect)"}}))
    S.close()
    if dist is not None:
        dist.destroy_process_group()


def single_bench(pkg, args, rank, world, local_rank, dist):
    """BASELINE configs[1] as written: xcorr_pss + peak_search (+ the per-cell chain) over the full +-100 ppm grid on ONE
    153600-sample capture buffer, handed over the way the reference's callers hold it -- complex<double> in host memory
    (searcher.h: cvec capbuf) -- through lcs_search_capbuf.  A step = one buffer, PCIe copy (2.46 MB) included; the
    library recognises the dongle's (u8 - 127) / 128 values on the device and takes the int8 kernel."""
    import torch
    dev_i = local_rank if world > 1 else 0
    f = pkg.f_search_set_for(FC, args.ppm)
    fc = FC + 100e3 * rank
    host = synth_batch(pkg, 8, 1234 + rank, np.full(8, fc))
    caps = [pkg.synth.iq_u8_to_complex(host[b]) for b in range(8)]
    S = pkg.Searcher(dev_i)
    n_cells, lat = 0, []

    def step(i):
        nonlocal n_cells
        t = time.perf_counter()
        cells, _ = S.search_capbuf(caps[i % 8], f, fc, fc, FS)
        lat.append(1e3 * (time.perf_counter() - t))
        n_cells += len(cells)
    t_pre, i = time.perf_counter(), 0
    while time.perf_counter() - t_pre < 0.6:      # pre-conditioning (untimed): clocks and queues settle, as in the batch bench
        step(i)
        i += 1
    for i in range(max(2, args.warmup)):
        step(i)
    lat.clear(); n_cells = 0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kname, _ = S.last_xcorr_info()
    # the same buffers as resident u8 bytes, one per call: what the PCIe copy and the complex<double> detour cost
    d_one = torch.from_numpy(host).to(torch.device("cuda", dev_i))
    for i in range(3):
        S.search_batch(d_one[i % 8].data_ptr(), pkg.FMT_IQ_U8, 1, N_CAP, f, fc, fc, FS, pkg.STAGE_FULL)
    t1 = time.perf_counter()
    for i in range(20):
        S.search_batch(d_one[i % 8].data_ptr(), pkg.FMT_IQ_U8, 1, N_CAP, f, fc, fc, FS, pkg.STAGE_FULL)
    res_ms = 1e3 * (time.perf_counter() - t1) / 20
    xc_ms = S.last_xcorr_ms()[0]
    # the hypothesis-split form of the same search (lcs_foe_partial + lcs_foe_finish, SURVEY 8e latency mode) on this one GPU:
    # what the split costs before any second GPU helps (the packed words make a round trip through a device buffer, the
    # peak search runs on the unpacked arrays); at world 2 each rank would correlate half the hypotheses between the two calls
    words = torch.empty(3 * 9600, dtype=torch.int64, device=torch.device("cuda", dev_i))
    meta = torch.empty(9601, dtype=torch.float64, device=torch.device("cuda", dev_i))
    for i in range(23):
        if i == 3:
            t1 = time.perf_counter()
        S.foe_partial(caps[i % 8], f, 0, f.size, fc, fc, FS, words.data_ptr(), meta.data_ptr())
        S.foe_finish(words.data_ptr(), meta.data_ptr(), f)
    foe_ms = 1e3 * (time.perf_counter() - t1) / 20
    half = []
    for i in range(23):
        if i == 3:
            t1 = time.perf_counter()
        S.foe_partial(caps[i % 8], f, 0, (f.size + 1) // 2, fc, fc, FS, words.data_ptr(), meta.data_ptr())
    foe_half_ms = 1e3 * (time.perf_counter() - t1) / 20
    if rank == 0:
        value = world * args.steps / dt
        flops = 8.0 * 137 * 3 * (N_CAP - 136) * f.size
        la = np.asarray(lat)
        print(json.dumps({
            "metric": "capture-buffers/s (1.92 Msps, 153600-samp) full CellSearch, ONE host buffer at a time", "value": value,
            "unit": "capture-buffers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": value / PUBLISHED_BUFFERS_PER_S,
            "dtype": "i8 x 3 base-256 digits of 24-bit integer templates, i32 accumulate (exact)" if kname.startswith("k_xcorr_i8") else "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: one 153600-sample capbuf per step as complex<double> in host memory through lcs_search_capbuf "
                                   "(xcorr_pss over the +-100 ppm grid, peak_search, per-cell chain), PCIe copy included",
                       "n_f": int(f.size), "latency_ms": {"min": float(la.min()), "median": float(np.median(la)), "max": float(la.max())},
                       "xcorr_kernel": kname, "cells_per_buffer": n_cells / max(1, args.steps),
                       "ms_per_buffer_resident_u8_input": res_ms, "xcorr_kernel_ms_one_buffer": xc_ms,
                       "foe_split": {"world1_partial_plus_finish_ms": foe_ms, "partial_with_half_the_hypotheses_ms": foe_half_ms,
                                     "note": "lcs_foe_partial + lcs_foe_finish on one GPU with all hypotheses (no collective), and the "
                                             "partial call alone with half of them (a rank's share at world 2; RCCL's 230 KB all-reduce and "
                                             "77 KB broadcast come on top)"},
                       "parallelism": "replicas" if world > 1 else "single GPU"},
            "roofline": {"bound": "launch/latency", "achieved": flops / (xc_ms * 1e-3) / 1e12, "peak": PEAK_I8_TOPS, "unit": "TOP/s",
                         "frac": flops / (xc_ms * 1e-3) / 1e12 / PEAK_I8_TOPS, "traffic": None,
                         "note": "one buffer is 285 workgroups on 256 CUs (~0.6 of one round): the call is bound by launch latency, the 2.46 MB "
                                 "PCIe copy and the exactness probe's readback, not by a pipe; frac = algorithmic flops of the one-buffer "
                                 "correlation launch / its HIP-event time / int8 peak"}}))
    S.close()
    if dist is not None:
        dist.destroy_process_group()


def track_bench(pkg, args, rank, world, local_rank, dist):
    """LTE-Tracker's per-symbol pipeline.  A step = one block of 980 OFDM symbols (7 frames, 70 ms of air
    time) for --batch tracked cells, time-domain symbols resident in HBM; replicas only (cells are independent)."""
    import torch
    dev_i = local_rank if world > 1 else 0
    g = np.load(os.path.join(ROOT, "tests", "golden", "capbuf_0000.npz"))
    iq = g["iq_u8"].astype(np.float64)
    cap = ((iq[0::2] - 127.0) / 128.0) + 1j * ((iq[1::2] - 127.0) / 128.0)
    fc = float(g["fc"][0])
    S = pkg.Searcher(dev_i)
    found, _ = S.search_capbuf(cap, np.array([30e3, 35e3, 40e3]), fc, fc, FS)        # the cells to track: 277 and 271
    n_sym, C = 980, args.batch
    per = []
    for c in found:
        kf = (fc - c.freq_superfine) / fc
        per.append(pkg.tracker.cut_symbols(cap, c.frame_start * (30.72e6 / 16) / (FS * kf), c.cp_type, c.freq_superfine, fc, fc, FS, n_sym))
    cells = [found[i % len(found)] for i in range(C)]
    td = np.stack([per[i % len(found)][0] for i in range(C)])
    late = np.stack([per[i % len(found)][1] for i in range(C)])
    ftv = np.stack([per[i % len(found)][2] for i in range(C)])
    fov = np.stack([per[i % len(found)][3] for i in range(C)])
    d_td = torch.from_numpy(td).to(torch.device("cuda", dev_i))
    # Two contexts, one host thread each (the shape of host/TrackCells.cpp's tracker threads): while one context's block is on
    # the GPU the other thread prepares, launches and unpacks its own -- lcs_track_block is synchronous per context, the
    # library call releases the interpreter lock.  Round 3 ran ONE context from one thread: 1.15 ms of GPU work in a 3.3 ms
    # step.  Every block is a full, independent pass (same cells, same symbols); a step = one block.  Four contexts by default.
    import threading
    depth = max(1, args.pipeline)
    ctxs = [S] + [pkg.Searcher(dev_i) for _ in range(depth - 1)]
    outs = [None] * depth
    gpu_ms, locks, lock = [], [0], threading.Lock()

    def one_block(k):
        r = ctxs[k].track_block(cells, None, fov, ftv, late, fc, fc, FS, want_syms=False, want_ce=False, td_device_ptr=d_td.data_ptr(),
                                n_sym=n_sym, out=outs[k])
        outs[k] = r
        with lock:
            gpu_ms.append(r["gpu_ms"])
            locks[0] = int(np.count_nonzero(r["mib_ok"] == 3))

    def run_blocks(n):
        todo = iter(range(n))
        def worker(k):
            while True:
                with lock:
                    i = next(todo, None)
                if i is None:
                    return
                one_block(k)
        ts = [threading.Thread(target=worker, args=(k,)) for k in range(depth)]
        for t in ts: t.start()
        for t in ts: t.join()

    # Round 5: the timed loop runs in C++ (host/TrackBench: one host thread per context calling lcs_track_block on the same block,
    # symbols resident in HBM) -- round 4's Python threads made the figure follow the box's host (91-150 M symbols/s for the
    # same 0.50 ms of GPU time per block).  The Python loop below stays as the fallback and as `python_loop_symbols_per_s`.
    cxx = cxx_bytes = None
    exe = os.path.join(ROOT, "host", "TrackBench")
    if os.path.exists(exe) and not args.lib:
        import struct, subprocess, tempfile
        tc = (pkg.capi.LcsTrackCell * C)()
        for i, c in enumerate(cells):
            for fld in ("n_id_1", "n_id_2", "cp_type", "n_ports", "n_rb_dl", "phich_duration", "phich_resource"):
                setattr(tc[i], fld, int(getattr(c, fld)))
        with tempfile.NamedTemporaryFile(suffix=".trkblock", delete=False) as fh:
            fh.write(struct.pack("<ii3d", C, n_sym, fc, fc, FS))
            fh.write(bytes(tc))
            for a in (fov, ftv, late):
                fh.write(np.ascontiguousarray(a, np.float64).tobytes())
            fh.write(np.ascontiguousarray(td, np.complex128).tobytes())
            blk_path = fh.name
    run_blocks(max(depth, args.warmup))
    gpu_ms.clear()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_blocks(args.steps)
    torch.cuda.synchronize()
    dt_py = time.perf_counter() - t0
    python_rate = C * n_sym * args.steps / dt_py
    if os.path.exists(exe) and not args.lib:
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        # (at least 120 untimed blocks first: ~50 ms of load, the time the shader clock takes to come up from idle -- with 6 the 60
        # timed blocks of the collection command measured the ramp, 95 M symbols/s instead of 155 M)
        try:
            r = subprocess.run([exe, blk_path, str(depth), str(args.steps), str(max(depth, args.warmup, 120)), str(dev_i)], capture_output=True, text=True, timeout=600)
            cxx = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            # a run that lost blocks is no measurement: its JSON counts only the blocks that succeeded
            if r.returncode != 0 or cxx.get("failed", 0) or cxx.get("blocks") != args.steps:
                sys.stderr.write("bench.py: host/TrackBench reported failed blocks (rc %d, %s): the Python loop's figure is reported\n" % (r.returncode, cxx))
                cxx = None
            # the same loop STARTING FROM THE DONGLE'S BYTES (round 6): per block lcs_track_cut on the capture resident in HBM, then the block
            # on the symbols it left there -- untimed for `value`, reported as config.cutter.cxx_from_bytes
            if cxx and hasattr(pkg.capi.load(), "lcs_track_cut") and not args.no_dense:
                with tempfile.NamedTemporaryFile(suffix=".u8", delete=False) as fh:
                    fh.write(np.ascontiguousarray(g["iq_u8"]).tobytes())
                    cap_path = fh.name
                try:
                    r2 = subprocess.run([exe, blk_path, str(depth), str(args.steps), str(max(depth, args.warmup, 120)), str(dev_i), cap_path], capture_output=True, text=True, timeout=600)
                    cxx_bytes = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
                    if r2.returncode != 0 or cxx_bytes.get("failed", 0) or cxx_bytes.get("blocks") != args.steps:
                        cxx_bytes = None
                except Exception as e:
                    cxx_bytes = None
                    sys.stderr.write("bench.py: host/TrackBench (from bytes) failed: %r\n" % (e,))
                finally:
                    os.unlink(cap_path)
        except Exception as e:
            cxx = None
            sys.stderr.write("bench.py: host/TrackBench failed (%r): the Python loop's figure is reported\n" % (e,))
        finally:
            if os.path.exists(blk_path):
                os.unlink(blk_path)
    if dist is not None:
        # one clock for every rank: the C++ loop's only if EVERY rank's C++ run was clean
        ok_t = torch.tensor([1.0 if cxx else 0.0], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        if ok_t.item() == 0.0:
            cxx = None
        dist.barrier()
    dt = cxx["seconds"] if cxx else dt_py
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    pipelined_ms = float(np.mean(gpu_ms))          # per block on its own stream; blocks of different contexts overlap here
    gpu_ms.clear()
    for _ in range(5):                             # the same block with nothing else on the GPU
        one_block(0)
    alone_ms = float(np.mean(gpu_ms[1:]))
    gpu_ms[:] = [alone_ms]
    locks = locks[0]
    # the continuous form (lcs_track_stream_block: symbols arrive block after block, the carried frames' rows stay on the
    # device): the same cells fed 980 symbols per call, host samples in (PCIe inside), measurements and MIB attempts out
    stream_form = None
    if not args.no_dense:
        reps = 6
        ctxs[0].track_stream_reset()
        ctxs[0].track_stream_block(cells, td, fov, ftv, late, fc, fc, FS, want_syms=False, want_ce=False)      # first block: no tail yet
        ctxs[0].track_stream_block(cells, td, fov, ftv, late, fc, fc, FS, want_syms=False, want_ce=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            so = ctxs[0].track_stream_block(cells, td, fov, ftv, late, fc, fc, FS, want_syms=False, want_ce=False)
        torch.cuda.synchronize()
        dts = (time.perf_counter() - t1) / reps
        # the same with the new symbols already in HBM (round 5: the call takes host, page-locked or device memory): what is left
        # of a call without the 128 MB PCIe copy from pageable memory
        ctxs[0].track_stream_reset()
        for _ in range(2):
            ctxs[0].track_stream_block(cells, None, fov, ftv, late, fc, fc, FS, want_syms=False, want_ce=False, td_device_ptr=d_td.data_ptr())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            ctxs[0].track_stream_block(cells, None, fov, ftv, late, fc, fc, FS, want_syms=False, want_ce=False, td_device_ptr=d_td.data_ptr())
        torch.cuda.synchronize()
        dtd = (time.perf_counter() - t1) / reps
        stream_form = {"symbols_per_s": C * n_sym / dts, "ms_per_call": 1e3 * dts, "symbols_per_call": n_sym, "mib_attempts_per_call": int(so["n_mib"].sum()),
                       "ms_per_call_symbols_in_hbm": 1e3 * dtd, "symbols_per_s_symbols_in_hbm": C * n_sym / dtd,
                       "note": "one context, one thread; ms_per_call: time-domain symbols handed over in pageable HOST memory (128 MB per call of 64 cells: "
                               "the PCIe copy is most of it); ms_per_call_symbols_in_hbm: the same call on symbols already resident in HBM.  The three carried "
                               "frames are not transformed again (their frequency-domain rows stay on the device) and no frame offset is decoded twice"}
        ctxs[0].track_stream_reset()
    # the producer thread's symbol extraction on the device (lcs_track_cut, round 6): the 80 ms capture as the dongle's BYTES in HBM
    # (0.3 MB), the symbols of all C tracked cells cut into [C][n_sym][128] complex<double> there -- checked against the host-cut
    # symbols the timed blocks ran on -- and the block on them
    cutter = None
    if hasattr(pkg.capi.load(), "lcs_track_cut") and not args.no_dense:
        d_iq = torch.from_numpy(np.ascontiguousarray(g["iq_u8"])).to(torch.device("cuda", dev_i))
        d_cut = torch.empty((C, n_sym, 128), dtype=torch.complex128, device=torch.device("cuda", dev_i))
        cps = [int(c.cp_type) for c in cells]
        args_cut = (d_iq.data_ptr(), pkg.FMT_IQ_U8, cap.size, cps, ftv[:, 0], fov[:, 0], fc, fc, FS, n_sym, d_cut.data_ptr())
        late_d, n_cut = ctxs[0].track_cut(*args_cut)
        same = bool(np.all(n_cut == n_sym) and np.array_equal(late_d, late) and torch.equal(d_cut, d_td))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            ctxs[0].track_cut(*args_cut)
        torch.cuda.synchronize()
        dtc_ = (time.perf_counter() - t1) / 10
        t1 = time.perf_counter()
        for _ in range(10):
            lt_, _n = ctxs[0].track_cut(*args_cut)
            ctxs[0].track_block(cells, None, fov, ftv, lt_, fc, fc, FS, want_syms=False, want_ce=False, td_device_ptr=d_cut.data_ptr(), n_sym=n_sym, out=outs[0])
        torch.cuda.synchronize()
        dte_ = (time.perf_counter() - t1) / 10
        cutter = {"ms_per_block": 1e3 * dtc_, "cut_plus_block_ms": 1e3 * dte_, "symbols_per_s_cut_plus_block": C * n_sym / dte_,
                  "identical_to_host_cut": same, "cxx_from_bytes": cxx_bytes, "capture_bytes_in_hbm": int(d_iq.numel()), "symbols_bytes_written": int(d_cut.numel() * 16),
                  "note": "lcs_track_cut: one thread per (cell, symbol) locates the capture's first sample in closed form (the host walks the "
                          "samples one by one), one wave per symbol converts and copies its 128 samples; one context, one thread, the "
                          "call synchronous (late / n_cut come back to the host)"}
        if not same:
            sys.stderr.write("bench.py: lcs_track_cut's symbols differ from the host cutter's\n")
    if rank == 0:
        value = world * C * n_sym * args.steps / dt
        out = {"metric": "OFDM symbols/s, LTE-Tracker per-symbol pipeline (get_fd + CRS channel estimate + FOE/TOE + MIB re-decode)",
               "value": value, "unit": "OFDM-symbols/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
               "data": "symbols cut from tests/golden/capbuf_0000 (cells 277, 271), tiled over the tracked cells",
               "config": {"workload": f"SURVEY 8 f4: {C} tracked cells x {n_sym} OFDM symbols (70 ms) per step, time-domain symbols resident in HBM",
                          "tracked_cells": C, "symbols_per_block": n_sym, "gpu_ms_per_block": alone_ms,
                          "gpu_ms_per_block_in_the_pipelined_run": pipelined_ms, "contexts_in_flight": depth, "stream_form": stream_form, "cutter": cutter,
                          "mib_locks_per_block": locks, "cells_in_real_time": value / world / 14000.0,
                          "timed_loop": ("host/TrackBench (C++): one host thread per context" if cxx else "Python threads (host/TrackBench not built)"),
                          "cxx_driver": cxx, "python_loop_symbols_per_s": python_rate,
                          "parallelism": "replicas" if world > 1 else "single GPU"}}
        # algorithmic bytes of a block: the time-domain symbols in (128 complex<double> each) + symbols and the two ports'
        # channel estimates out (72 complex<double> per symbol each)
        blk_bytes = C * n_sym * (128 * 16 + 72 * 16 * (1 + 2))
        g_ms = float(np.mean(gpu_ms))
        out["roofline"] = {"bound": "hbm", "achieved": blk_bytes / (g_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                           "frac": blk_bytes / (g_ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                           "gpu_active_frac": g_ms / (1e3 * dt / args.steps),
                           "note": "four launches per block (prep, FFT, channel estimate, MIB): latency- and fp64-bound far below the HBM roof; "
                                   "achieved = algorithmic bytes of a block / GPU time of the block alone (HIP events on its stream); "
                                   "gpu_active_frac = that GPU time / wall time per block of the pipelined run (> 1 would mean the overlapped "
                                   "blocks of the two contexts fill each other's idle SIMDs)"}
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as O
            t1 = time.perf_counter()
            n_done = 0
            for i in range(min(4, C)):
                c = O.Cell()
                for fld in ("n_id_1", "n_id_2", "cp_type", "n_ports", "n_rb_dl", "phich_duration", "phich_resource"):
                    setattr(c, fld, int(getattr(cells[i], fld)))
                sy, _, _ = O.trk_get_fd(c, td[i], 0, 0, fov[i], late[i], fc, fc, FS)
                r = O.trk_chan_est(c, sy, 0, 0, fov[i], ftv[i], fc, fc, FS)
                for o in range(4):
                    ii = [(o + fr) * 140 + 7 + s_ for fr in range(4) for s_ in range(4)]
                    if ii[-1] < min(r["ce_upto"][:c.n_ports]):
                        O.trk_mib(c, sy[ii], r["ce"][:c.n_ports][:, ii], r["ce_pw"][:c.n_ports][:, ii, 3])
                n_done += n_sym
            dtc = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": n_done / dtc, "unit": "OFDM-symbols/s", "cores": 1, "kind": "port",
                                   "sample": f"{n_done} symbols (4 cells x one block) through the C oracle's restatement of the same pipeline, {dtc:.2f} s"}
        print(json.dumps(out))
    for x in ctxs:
        x.close()
    if dist is not None:
        dist.destroy_process_group()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU) through torch.distributed.run."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts: RCCL needs it
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def device_identity(torch, index):
    """(host, device index) tells the GPUs of one node apart; uuid / PCI address ride along for the record (some ROCm
    builds report an all-zero uuid, so they are not what the distinct-devices check compares)."""
    import socket
    pr = torch.cuda.get_device_properties(index)
    extra = ",".join(f"{a}={getattr(pr, a)}" for a in ("uuid", "pci_bus_id", "pci_device_id") if getattr(pr, a, None) not in (None, ""))
    return f"{socket.gethostname()}/cuda:{index}" + (f" ({extra})" if extra else "")


def power_probe(run_steps, device_index, seconds=2.0):
    """Outside the timed region: keep the same pipelined steps running for ~`seconds` while rocm-smi is sampled, and report
    the shader clock and the socket power the chain holds (the correlation runs at the chip's power cap: DESIGN 3.0).
    Returns None when rocm-smi is not there or does not answer."""
    import subprocess, threading
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                p = subprocess.run(["rocm-smi", "-d", str(device_index), "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
                card = next(iter(json.loads(p.stdout).values()))
                sclk = next((v for k, v in card.items() if "sclk" in k.lower() and "mhz" in str(v).lower()), None)
                pw = next((v for k, v in card.items() if "power (w)" in k.lower()), None)
                if sclk is not None and pw is not None:
                    samples.append((float("".join(ch for ch in str(sclk) if ch.isdigit() or ch == ".")), float(pw)))
            except Exception:
                return
            stop.wait(0.25)

    th = threading.Thread(target=sampler, daemon=True)
    t0 = time.perf_counter()
    th.start()
    n = 0
    while time.perf_counter() - t0 < seconds:
        run_steps()
        n += 1
    stop.set()
    th.join(timeout=6)
    busy = samples[1:] if len(samples) > 2 else samples        # the first sample may predate the load
    if not busy:
        return None
    return {"sclk_mhz": sorted(s for s, _ in busy)[len(busy) // 2], "socket_power_w": sorted(w for _, w in busy)[len(busy) // 2],
            "samples": len(busy), "steps_run": n,
            "note": "median of rocm-smi samples taken while the same pipelined steps ran on, AFTER the timed region (MI355X: 2400 MHz peak, 1400 W cap)"}


def kernel_source_sha():
    """sha256 over the sources of the dominant kernel: roofline.traffic is only reported when the committed PMC
    summary was collected from exactly this code."""
    import hashlib
    h = hashlib.sha256()
    for name in ("pss_xcorr_i8.hip", "pss_xcorr_f16.hip", "pss_xcorr.hip", "lcs_internal.h"):
        h.update(open(os.path.join(ROOT, "lte-cell-scanner_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def digest(rec, cnt):
    """Bytes of the valid records of one collected batch (order and every field, NaNs included)."""
    return b"".join([cnt.tobytes()] + [rec[b, :cnt[b]].tobytes() for b in range(len(cnt))])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None,
                    help="capture buffers per enqueue (one correlation launch) per GPU; default 128 (--stage track: 64 tracked cells)")
    ap.add_argument("--batches-per-step", type=int, default=None,
                    help="enqueues per step: a step is batch x this many buffers per GPU; default 12800 // batch")
    ap.add_argument("--distinct", type=int, default=4, help="distinct resident batches the enqueues cycle through")
    ap.add_argument("--ppm", type=float, default=100.0)
    ap.add_argument("--fc", type=float, default=FC,
                    help="first carrier of the batch (default 739 MHz: BASELINE configs[2], n_f = 31 at --ppm 100); e.g. --fc 2.6e9 --ppm 120 "
                         "is the CLI's band-7 grid, n_f = 125 (src/CellSearch.cpp:463-465)")
    ap.add_argument("--stage", choices=["pss", "full", "single", "stream", "track"], default="full",
                    help="full = BASELINE configs[2], the whole CellSearch chain (default); pss = configs[1], xcorr_pss + "
                         "peak_search only; single = configs[1] as written, ONE 153600-sample host buffer per step through "
                         "lcs_search_capbuf (latency, PCIe included); stream = configs[4], one host buffer at a time through the hipGraph-captured "
                         "single-hypothesis chain (separate, shorter report); track = SURVEY 8 f4, LTE-Tracker's per-symbol "
                         "pipeline on blocks of OFDM symbols of --batch tracked cells")
    ap.add_argument("--input", choices=["u8", "c64"], default="u8",
                    help="resident input format: raw RTL-SDR u8 I/Q (int8 MFMA correlation kernel, default) or complex<float> "
                         "(fp16 three-product MFMA correlation kernel)")
    ap.add_argument("--c64-probe", action="store_true",
                    help="with --input c64: lcs_set_float_batch_probe -- the batches (dongle data held as complex<float>) are recognised on the "
                         "device and take the u8 / int8 route; without it --input c64 measures the fp16 kernel")
    ap.add_argument("--pipeline", type=int, default=None,
                    help="contexts (streams + workspaces) used round-robin: the latency-bound per-cell "
                         "stages of batch i overlap the PSS correlation of batch i+1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power-probe", action="store_true", help="skip the ~2 s of extra steps under rocm-smi sampling (roofline.power_probe)")
    ap.add_argument("--no-xc-timing", action="store_true", help="developer A/B runs with builds that record no kernel events")
    ap.add_argument("--synth-cache", default=None, help="developer A/B runs: directory caching the synthetic host batch between runs")
    ap.add_argument("--lib", default=None, help="developer A/B runs: load this build of liblcs_amd.so instead of the in-tree one")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (testing the multi-rank path on one GPU)")
    ap.add_argument("--share-gpu0", action="store_true", help="testing only: every rank uses GPU 0")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-rank branch with ONE rank: process group (RCCL), device-identity all-gather, the per-step "
                         "asynchronous all-gather of the cell records, the timing all-gather / MAX all-reduce -- so that the first 8-GPU "
                         "run is not the collective code's first execution")
    ap.add_argument("--input-host", action="store_true",
                    help="feed the batches from page-locked HOST memory (lcs_batch_enqueue_host): the PCIe transfer is inside the timed region")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense-band line (2-3 cells planted in every buffer) reported in config.dense_band")
    ap.add_argument("--dense-main", action="store_true",
                    help="developer / profiling runs: the MAIN workload is the dense band (2-3 cells planted in every buffer); the line says so in config.workload")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 64 if args.stage == "track" else 128
    if args.pipeline is None:
        args.pipeline = 4 if args.stage == "track" else 2      # contexts in flight (track: profiles/r04/experiments/tracker_depth.txt)
    if args.batches_per_step is None:
        args.batches_per_step = max(1, 12800 // args.batch)

    # --gpus N is the number of ranks.  Started by a launcher (WORLD_SIZE set, e.g. the driver's torch.distributed.run
    # line) the two must agree; started plainly with N > 1, start the N ranks ourselves.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s): the two must agree")

    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    if args.lib:
        pkg.capi.LIB_PATH = args.lib if os.path.isabs(args.lib) or os.path.exists(args.lib) else os.path.join(ROOT, args.lib)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    devices = None
    multi = world > 1 or args.force_dist      # the multi-rank code path (collectives live), whatever the rank count
    if multi:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.share_gpu0:
            local_rank = 0
        elif torch.cuda.device_count() < world:
            sys.exit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible (testing on one GPU: --share-gpu0 --dist-backend gloo)")
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
        # every rank must own a different GPU
        devices = [None] * world
        dist.all_gather_object(devices, device_identity(torch, local_rank))
        if len({d.split(" ")[0] for d in devices}) != world and not args.share_gpu0:
            sys.exit(f"bench.py: the {world} ranks do not sit on {world} distinct GPUs: {devices}")
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if multi else 0)
    coll_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")

    if args.stage == "stream":
        return stream_bench(pkg, args, rank, world, local_rank, dist)
    if args.stage == "track":
        return track_bench(pkg, args, rank, world, local_rank, dist)
    if args.stage == "single":
        return single_bench(pkg, args, rank, world, local_rank, dist)
    f = pkg.f_search_set_for(args.fc, args.ppm)
    stage_mask = pkg.STAGE_FULL if args.stage == "full" else pkg.STAGE_PSS
    B, K, D = args.batch, args.batches_per_step, max(1, args.distinct)
    fmt = pkg.FMT_IQ_U8 if args.input == "u8" else pkg.FMT_C64
    # rank r searches carriers FC + 100 kHz * (r*B + b): the sweep's carrier axis is the shard axis
    fcs = args.fc + 100e3 * (np.arange(B) + rank * B)
    cache = os.path.join(args.synth_cache, f"batch_{B}_{1234 + rank}{'_dense' if args.dense_main else ''}.npy") if args.synth_cache else None
    if cache and os.path.exists(cache):
        host = np.load(cache)
    else:
        host = synth_batch(pkg, B, 1234 + rank, fcs, dense=args.dense_main)
        if cache:
            os.makedirs(args.synth_cache, exist_ok=True)
            np.save(cache, host)
    base = torch.from_numpy(host).to(dev)
    # D distinct resident batches: the synthetic base batch and copies rotated in time by whole samples on the device
    # (cells move, frame timing changes, every correlation value changes).  Inputs are in HBM before timing starts.
    d_caps = [base if d == 0 else torch.roll(base, shifts=2 * 1117 * d, dims=1).contiguous() for d in range(D)]
    if fmt == pkg.FMT_C64:
        d_caps = [torch.view_as_complex(((x.to(torch.float32) - 127.0) / 128.0).view(B, N_CAP, 2).contiguous()) for x in d_caps]
    torch.cuda.synchronize()
    ctxs = [pkg.Searcher(local_rank if multi else 0) for _ in range(max(1, args.pipeline))]
    if args.c64_probe and hasattr(pkg.capi.load(), "lcs_set_float_batch_probe"):
        for x in ctxs:
            x.set_float_batch_probe(True)
    MAXC = 16
    # one fixed-size record block per step for the all-gather: [n, then n x (n_id_cell, fc, f_off, pss_pow, sfn)]
    MAXREC = max(64, B) * K
    gather_in = [torch.zeros(1 + 5 * MAXREC, dtype=torch.float64, pin_memory=(coll_dev.type == "cuda")) for _ in range(2)] if multi else None
    gather_dev = [torch.zeros(1 + 5 * MAXREC, dtype=torch.float64, device=coll_dev) for _ in range(2)] if multi else None
    gather_out = [torch.zeros((world, 1 + 5 * MAXREC), dtype=torch.float64, device=coll_dev) for _ in range(2)] if multi else None
    pending = {"work": None}

    host_t = {"enqueue": 0.0, "collect": 0.0, "n": 0, "lib_us": 0.0 if hasattr(pkg.capi.load(), "lcs_last_collect_host_us") else None}
    seen = {}            # distinct-batch index -> digest of the first collect; every later collect must match
    state = {"mismatch": 0, "collected": 0}

    # --input-host: the same distinct batches in page-locked host memory (lcs_host_alloc), copied by every enqueue
    work = {"caps": d_caps, "host": None}
    if args.input_host:
        if fmt != pkg.FMT_IQ_U8:
            sys.exit("bench.py: --input-host feeds u8 I/Q")
        work["host"] = []
        for x in d_caps:
            h = ctxs[0].host_alloc(x.numel())
            h[:] = x.cpu().numpy().reshape(-1)
            work["host"].append(h)

    def enqueue(i):
        t = time.perf_counter()
        if work["host"] is not None:
            ctxs[i % len(ctxs)].batch_enqueue_host(work["host"][i % len(work["host"])], fmt, B, N_CAP, f, fcs, fcs, FS, stage_mask)
        else:
            ctxs[i % len(ctxs)].batch_enqueue(work["caps"][i % len(work["caps"])].data_ptr(), fmt, B, N_CAP, f, fcs, fcs, FS, stage_mask)
        host_t["enqueue"] += time.perf_counter() - t

    def collect(i):
        t = time.perf_counter()
        rec, cnt = ctxs[i % len(ctxs)].batch_collect_raw(B, MAXC)
        host_t["collect"] += time.perf_counter() - t
        host_t["n"] += 1
        if host_t["lib_us"] is not None:
            host_t["lib_us"] += ctxs[i % len(ctxs)].last_collect_host_us()
        dg = digest(rec, cnt)
        if seen.setdefault(i % len(work["caps"]), dg) != dg:
            state["mismatch"] += 1
        state["collected"] += 1
        return rec, cnt

    ar_maxc = np.arange(MAXC)[None, :]

    def pack_records(rec, cnt):
        """The valid cell records of one collected batch as rows (n_id_cell, fc, f_off, pss_pow, sfn) -- vectorised, ~30 us per
        batch (round 4: a per-buffer Python loop here cost the forced-dist run 20 % of its rate)."""
        b, k = np.nonzero(ar_maxc < cnt[:, None])
        return np.stack([(rec["n_id_2"][b, k] + 3 * rec["n_id_1"][b, k]).astype(np.float64), rec["fc_requested"][b, k],
                         (rec["freq_superfine"] if stage_mask == 3 else rec["freq"])[b, k], rec["pss_pow"][b, k],
                         rec["sfn"][b, k].astype(np.float64)], axis=1)

    def gather_step(step_idx, blocks):
        """ONE asynchronous all-gather of this step's cell records; the previous step's is waited for first (it has had
        a whole step to complete), so the collective never sits on the critical path."""
        if pending["work"] is not None:
            pending["work"].wait()
        blk = np.concatenate(blocks) if blocks else np.zeros((0, 5))
        n = min(len(blk), MAXREC)
        buf = gather_in[step_idx % 2].numpy()
        buf[0] = n
        buf[1:1 + 5 * n] = blk[:n].reshape(-1)
        gather_dev[step_idx % 2].copy_(gather_in[step_idx % 2], non_blocking=True)
        pending["work"] = dist.all_gather_into_tensor(gather_out[step_idx % 2].view(-1), gather_dev[step_idx % 2], async_op=True)

    def run(n_steps, xc_ms=None, gather=True, step_ms=None, first_step=0):
        """n_steps steps of K batches each, software-pipelined over the contexts: batch i is enqueued before batch
        i-(depth-1) is collected, so nothing but the stream of the collected batch ever blocks."""
        depth = len(ctxs)
        total = n_steps * K
        recs, last = [], None
        t_prev = time.perf_counter()
        for i in range(total + depth - 1):
            if i < total:
                enqueue(i)
            j = i - (depth - 1)
            if j >= 0:
                last = collect(j)
                if multi and gather:
                    recs.append(pack_records(*last))
                if xc_ms is not None and not args.no_xc_timing:
                    xc_ms.append(ctxs[j % depth].last_xcorr_ms()[0])
                if (j + 1) % K == 0:
                    if multi and gather:
                        gather_step(first_step + j // K, recs)
                    recs = []
                    if step_ms is not None:
                        now = time.perf_counter()
                        step_ms.append(1e3 * (now - t_prev))
                        t_prev = now
        return last

    # Pre-conditioning (untimed, like the warm-up): a fresh process shows one ~50 ms stall in its
    # first few hundred milliseconds of GPU activity (clock / power-state ramp); keep the GPU
    # busy for ~0.6 s so that it does not land in the timed steps of a short run.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.6:
        run(1, gather=False)      # time-based, so no collective in here (ranks may differ in count)
    host_t.update(enqueue=0.0, collect=0.0, n=0, lib_us=(0.0 if host_t["lib_us"] is not None else None))
    if args.warmup:
        run(args.warmup, first_step=0)
    xc_ms, step_ms = [], []
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = run(args.steps, xc_ms, step_ms=step_ms, first_step=args.warmup)
    if pending["work"] is not None:
        pending["work"].wait()
        pending["work"] = None
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0
    mem_free, mem_total = torch.cuda.mem_get_info(dev)      # everything the run holds: resident batches + the contexts' workspaces
    hbm_in_use = {"after_timed_region_GB": (mem_total - mem_free) / 1e9, "resident_input_GB": sum(x.numel() * x.element_size() for x in d_caps) / 1e9,
                  "contexts": len(ctxs)}
    if multi:
        # identity check of the record all-gather (outside the clock): this rank's row of the last step's gather must be
        # what it contributed, and every rank's row must carry a plausible count
        li = (args.warmup + args.steps - 1) % 2
        got = gather_out[li].cpu()
        if not torch.equal(got[rank], gather_in[li]):
            sys.exit("bench.py: the all-gathered cell records of this rank differ from what it sent")
        state["gathered"] = [int(got[r, 0].item()) for r in range(world)]
        # carriers named in every rank's records (column 1 of a row): rank r searches FC + 100 kHz * (r B .. r B + B - 1)
        rows = [got[r, 1:1 + 5 * state["gathered"][r]].view(-1, 5) for r in range(world)]
        state["gathered_fc"] = [[float(x[:, 1].min()), float(x[:, 1].max())] if len(x) else None for x in rows]
        state["gather_shape"] = list(got.shape)
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    dt_ranks = [dt]
    if dist is not None:
        t = torch.tensor([dt_own], dtype=torch.float64, device=coll_dev)
        tl = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(tl, t)
        dt_ranks = [float(x.item()) for x in tl]        # each rank's own time to finish its K x steps batches
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    probe = None
    if rank == 0 and not multi and not args.no_power_probe:
        probe = power_probe(lambda: run(1, gather=False), 0)

    # ---- verification (outside the timed region) -------------------------------------------------------------
    # (1) every pipelined collect of a distinct batch returned identical bytes (state["mismatch"] == 0);
    # (2) a sequential, single-context, synchronous run of each distinct batch returns those bytes too;
    # (3) buffer 0 of batch 0 against the CPU oracle (done in the cpu_baseline leg below, rank 0).
    seq_ok = True
    iso_ms = []
    n_cells_per_batch = []
    for d in range(D):
        ctxs[0].batch_enqueue(d_caps[d].data_ptr(), fmt, B, N_CAP, f, fcs, fcs, FS, stage_mask)
        rec, cnt = ctxs[0].batch_collect_raw(B, MAXC)
        iso_ms.append(float('nan') if args.no_xc_timing else ctxs[0].last_xcorr_ms()[0])       # the dominant kernel alone on the GPU (nothing overlaps it here)
        n_cells_per_batch.append(int(cnt.sum()))
        seq_ok = seq_ok and (seen.get(d) == digest(rec, cnt))
        if d == 0:
            rec0, cnt0 = rec.copy(), cnt.copy()
    kname, executed_ops = ctxs[0].last_xcorr_info()
    st_main = ctxs[0].last_batch_stats() if stage_mask == pkg.STAGE_FULL and hasattr(pkg.capi.load(), "lcs_last_batch_stats") else None
    # ---- dense band (reported next to the default, never part of `value`): 2-3 cells planted in EVERY buffer, so the
    # per-cell stages carry ~8x the cells of the band-scan workload above
    dense, dense_rec, dense_host = None, None, None
    if args.stage == "full" and not args.no_dense and fmt == pkg.FMT_IQ_U8 and world == 1:
        dh = synth_batch(pkg, B, 4321, fcs, dense=True)
        dd = torch.from_numpy(dh).to(dev)
        work_saved, seen_saved, host_saved, coll_saved = dict(work), dict(seen), dict(host_t), state["collected"]
        work.update(caps=[dd, torch.roll(dd, shifts=2 * 2221, dims=1).contiguous()], host=None)
        seen.clear()
        mism0 = state["mismatch"]
        n_dense = max(1, min(K, 100))
        K_, dms, last_d = n_dense, [], None
        for rep in range(3):                      # the first pass warms up (the dense batches size the per-cell rounds)
            t1 = time.perf_counter()
            for i in range(K_ + len(ctxs) - 1):
                if i < K_:
                    enqueue(i)
                j = i - (len(ctxs) - 1)
                if j >= 0:
                    last_d = collect(j)
            torch.cuda.synchronize()
            dms.append(1e3 * (time.perf_counter() - t1) / K_)
        cells_d = int(last_d[1].sum())
        dms = dms[1:]
        ctxs[0].batch_enqueue(dd.data_ptr(), fmt, B, N_CAP, f, fcs, fcs, FS, stage_mask)       # the un-rolled dense batch once more, alone: the
        dense_rec = ctxs[0].batch_collect_raw(B, MAXC)                                        # records the oracle leg below checks
        dense_host = dh
        st_d = ctxs[0].last_batch_stats() if hasattr(ctxs[0], "last_batch_stats") and hasattr(pkg.capi.load(), "lcs_last_batch_stats") else None
        dense = {"buffers_per_s": B / (min(dms) * 1e-3), "ms_per_batch": min(dms), "cells_decoded_per_buffer": cells_d / B,
                 "cells_past_sss_per_buffer": (st_d["cells_past_sss"] / B) if st_d else None,
                 "pbch_candidates_decoded_per_cell_past_sss": (st_d["pbch_candidates_decoded"] / max(1, st_d["cells_past_sss"])) if st_d else None,
                 "cells_planted_per_buffer": 2.5, "batches_timed": K_, "pipelined_mismatches": state["mismatch"] - mism0,
                 "note": f"same chain, same grid; every one of the {B} buffers of a batch carries 2-3 synthetic cells (SNR 0-10 dB)"}
        work.clear(); work.update(work_saved)
        seen.clear(); seen.update(seen_saved)
        state["mismatch"], state["collected"] = mism0, coll_saved
        host_t.update(host_saved)
    mem_free, _ = torch.cuda.mem_get_info(dev)
    hbm_in_use["after_dense_band_GB"] = (mem_total - mem_free) / 1e9      # the dense batches grow the per-cell buffers (512 -> 1024 cells per context)
    verify = {"pipelined_collects": state["collected"], "pipelined_mismatches": state["mismatch"],
              "sequential_run_identical": bool(seq_ok), "oracle_buffer0": None, "oracle_buffers_checked": [], "oracle_buffers_differing": []}

    if rank == 0:
        n_f = f.size
        n_buffers = world * B * K * args.steps
        value = n_buffers / dt
        # dominant kernel: PSS correlation.  Algorithmic work per buffer:
        # F = 8*137*3*(N-136)*n_f real flops (reference-faithful count over all lags); the consumed subset is the
        # 15 x 9600 lags xc_combine reads.
        flops_per_buf = 8.0 * 137 * 3 * (N_CAP - 136) * n_f
        flops_consumed = 8.0 * 137 * 3 * 9600 * 15 * n_f
        k_ms = float(np.mean(xc_ms)) if xc_ms else float('nan')
        k_iso = float(np.mean(iso_ms))
        i8 = kname.startswith("k_xcorr_i8")
        f16 = kname.startswith("k_xcorr_f16")
        peak = PEAK_I8_TOPS if i8 else (PEAK_BF16_TFLOPS if f16 else PEAK_FP32_TFLOPS)
        # HBM traffic of the dominant kernel from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes,
        # corrected as MI355X_MICROARCH.md prescribes), taken from the committed summary ONLY if it was collected from
        # exactly the kernel sources that are running now.
        traffic, traffic_src = None, None
        for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
            try:
                pmj = json.load(open(os.path.join(ROOT, "profiles", tag, "pmc_summary.json")))
                if pmj.get("kernel_source_sha") != kernel_source_sha():
                    continue
                # the running correlation kernel's entry: the int8 kernel's counters come from the u8 passes, the fp16 kernel's
                # from the passes over complex<float> batches (profiles/collect.sh)
                name = next((k for k, v in pmj["kernels"].items() if k.startswith(kname.split("<")[0]) and "hbm_bytes_per_buffer" in v), None)
                if name and int(pmj["kernels"][name]["n_f"]) == int(n_f):
                    traffic, traffic_src = float(pmj["kernels"][name]["hbm_bytes_per_buffer"]) * B, f"profiles/{tag}/pmc_summary.json"
                    break
            except Exception:
                continue
        achieved = flops_per_buf * B / (k_ms * 1e-3) / 1e12
        bytes_per_buf = 1651200 + 230400 * n_f                 # SURVEY.md section 8d compulsory HBM bytes
        sm = np.asarray(step_ms)
        out = {
            "metric": "capture-buffers/s (1.92 Msps, 153600-samp) full CellSearch; HBM GB/s vs peak",
            "value": value, "unit": "capture-buffers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": value / PUBLISHED_BUFFERS_PER_S,
            "dtype": ("i8 x 3 base-256 digits of 24-bit integer templates, i32 accumulate (exact)" if i8 else
                      "f16 hi + lo parts of fp32 samples and templates (22 bits each), 3 products, f32 accumulate" if f16 else "f32"),
            "data": "synthetic", "verified": bool(seq_ok and state["mismatch"] == 0),
            "iq_samples_per_s": value * N_CAP,
            "config": {"workload": ("configs[2]: full searcher chain (PSS+SSS+FOE+TFG+MIB)" if args.stage == "full" else
                                    "configs[1]: xcorr_pss + peak_search over the full +-100 ppm foe grid") +
                                   (" -- DENSE BAND (2-3 cells planted in every buffer; not the headline workload)" if args.dense_main else "") +
                                   ("" if (args.fc == FC and args.ppm == 100.0) else f" -- NOT the headline grid: the CLI's grid for fc {args.fc / 1e6:g} MHz at {args.ppm:g} ppm, n_f = {f.size}") +
                                   f", one MI355X per rank, step = {K} batches x {B} = {K * B} 153600-sample capbufs per GPU, "
                                   f"fc {args.fc / 1e6:g} MHz + 100 kHz raster, +-{args.ppm:g} ppm, {D} distinct resident batches",
                       "n_f": int(n_f), "batch_per_gpu": B, "batches_per_step": K, "buffers_per_step_per_gpu": B * K,
                       "buffers_timed": n_buffers, "timed_region_s": dt, "stage": args.stage,
                       "float_batch_probe": bool(args.c64_probe),
                       "ingest": ("u8 I/Q in page-locked host memory, PCIe copy of every batch inside the timed region" if args.input_host else
                                  "u8 I/Q resident in HBM") if fmt == pkg.FMT_IQ_U8 else "complex<float> resident in HBM",
                       "xcorr_kernel": "mfma_i32_16x16x64_i8, three int8 digits per 24-bit integer template tap" if i8 else
                                       ("mfma_f32_16x16x32_f16, hi / lo fp16 split of samples and templates, three products" if f16 else "mfma_f32_16x16x4_f32"),
                       "pipeline_depth": len(ctxs),
                       "parallelism": (f"carrier-sweep shard x{world}, one async {'RCCL' if args.dist_backend == 'nccl' else args.dist_backend} all-gather of the "
                                       f"cell records per step") if multi else "single GPU",
                       "collectives": ({"backend": dist.get_backend(), "world": world, "gathered_records_last_step": state.get("gathered"),
                                        "gathered_fc_min_max_per_rank": state.get("gathered_fc"), "gather_out_shape": state.get("gather_shape")} if multi else None),
                       "baseline_note": "vs_baseline = value / (1 buffer per ~6 s), doc/CellSearch.html:52-54 (dual-core i7-2640, ppm 100; BASELINE.md section 1)",
                       "cells_per_distinct_batch": n_cells_per_batch,
                       "cells_per_buffer": float(np.mean(n_cells_per_batch)) / B,
                       "cells_past_sss_per_buffer": (st_main["cells_past_sss"] / B) if st_main else None,
                       "pbch_candidates_decoded_per_cell_past_sss": (st_main["pbch_candidates_decoded"] / max(1, st_main["cells_past_sss"])) if st_main else None,
                       "dense_band": dense,
                       "hbm_in_use": hbm_in_use,
                       "per_rank_buffers_per_s": [B * K * args.steps / x for x in dt_ranks],
                       "devices": devices,
                       "ms_per_batch": 1e3 * dt / (args.steps * K),
                       "step_ms": {"min": float(sm.min()), "median": float(np.median(sm)), "max": float(sm.max())} if sm.size else None,
                       "host_ms_per_batch": {"enqueue": 1e3 * host_t["enqueue"] / max(1, host_t["n"]),
                                             "collect_incl_wait": 1e3 * host_t["collect"] / max(1, host_t["n"]),
                                             "collect_excl_wait_in_library": (1e-3 * host_t["lib_us"] / max(1, host_t["n"])) if host_t["lib_us"] is not None else None}},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TOP/s" if i8 else "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "peak_source": ("MI355X_MICROARCH.md lists no int8 MFMA line in its spec table: 5000 TOP/s = 2 x its dense bf16 figure (the 8-bit "
                                         "formats' ratio); the guide's own measured ceiling for v_mfma_i32_16x16x64_i8 is 3944 TOP/s (frac_vs_measured_ceiling)"
                                         if i8 else "MI355X_MICROARCH.md spec table: dense fp16 / bf16 MFMA 2.5 PFLOP/s" if f16 else
                                         "MI355X_MICROARCH.md spec table: dense fp32 MFMA 157.3 TFLOP/s"),
                         "frac_vs_measured_ceiling": (achieved / 3944.0) if i8 else None,
                         "frac_algorithmic": flops_consumed * B / (k_ms * 1e-3) / 1e12 / peak,
                         "frac_executed": (executed_ops / (k_ms * 1e-3) / 1e12 / peak) if executed_ops else None,
                         "note": ("achieved = SURVEY 8(d) algorithmic flops of one launch (8*137*3*(N-136)*n_f per buffer x %d buffers, counted ONCE) / "
                                  "the kernel's mean duration inside the timed region (HIP events on its stream); peak = dense int8 MFMA, 2x the bf16 "
                                  "figure of MI355X_MICROARCH.md (16x16x64 issues every ~18 cycles, 32x32x32 every 32: both measure 4.3-4.4 POP/s at the "
                                  "sustained clock, tools/microbench/mfma_rate.hip).  frac_algorithmic counts only the 15 x 9600 lags that are consumed; "
                                  "frac_executed counts the MFMA work issued (x3 digits, 160 of 137 taps, 96 of 93 columns)." % B) if i8 else
                                 ("achieved = SURVEY 8(d) algorithmic flops of one launch / the kernel's mean duration in the timed region, against the dense "
                                  "fp16 MFMA peak (2.5 PFLOP/s); frac_executed counts the MFMA work issued (x3 products, 160 of 137 taps, 96 of 93 columns)") if f16 else
                                 "achieved = SURVEY 8(d) algorithmic flops / kernel time, against the fp32 MFMA peak",
                         "kernel": kname, "kernel_ms": k_ms, "kernel_ms_isolated": k_iso,
                         "frac_isolated": flops_per_buf * B / (k_iso * 1e-3) / 1e12 / peak,
                         "frac_executed_isolated": (executed_ops / (k_iso * 1e-3) / 1e12 / peak) if executed_ops else None,
                         "flops_per_launch": flops_per_buf * B, "executed_ops_per_launch": executed_ops or None,
                         "traffic_source": traffic_src,
                         "traffic_note": "HBM bytes per launch from rocprofv3 PMC (2*FETCH_SIZE+WRITE_SIZE)*1024, reported only when the committed "
                                         "summary's kernel_source_sha equals the running sources'; algorithmic bytes of this kernel per launch "
                                         "(capture buffer in, xc_incoherent_single out): %d" % int(((2 if i8 else 8) * N_CAP + 4 * 3 * 9600 * n_f) * B),
                         "kernel_source_sha": kernel_source_sha(),
                         "hbm_algorithmic_GBps": bytes_per_buf * B / (k_ms * 1e-3) / 1e9, "hbm_peak_GBps": 8000.0,
                         "hbm_frac_algorithmic": bytes_per_buf * B / (k_ms * 1e-3) / 1e9 / 8000.0,
                         "hbm_measured_GBps": (traffic / (k_ms * 1e-3) / 1e9) if traffic else None,
                         "hbm_frac_measured": (traffic / (k_ms * 1e-3) / 1e9 / 8000.0) if traffic else None,
                         "power_probe": probe},
        }
        if not args.no_cpu_baseline:
            cb, ocells = cpu_baseline(pkg, host, f, fcs, args.stage)
            out["cpu_baseline"] = cb
            # (3) the GPU's records against the oracle's, buffer by buffer: identities exact, powers 1e-5, frequencies 1e-3 Hz.
            # Round 5: 12 buffers of the timed batch (the 4 of the baseline sample + 8 more, occupied carriers among them)
            # and 4 of the dense band -- the whole population (every timed buffer, 512 synthetic ones) is
            # tools/parity_population.py's job, profiles/r05/parity_population.json its record.
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as O

            def same(got, ocells):
                ok = len(got) == len(ocells)
                for g_, o_ in zip(got, ocells):
                    ok = ok and (int(g_["n_id_2"]), int(g_["ind"]), float(g_["freq"])) == (o_.n_id_2, o_.ind, o_.freq)
                    ok = ok and abs(float(g_["pss_pow"]) - o_.pss_pow) <= 1e-5 * o_.pss_pow
                    if args.stage == "full":
                        ok = ok and (int(g_["n_id_1"]), int(g_["cp_type"]), int(g_["n_ports"]), int(g_["n_rb_dl"]), int(g_["phich_duration"]),
                                     int(g_["phich_resource"]), int(g_["sfn"])) == (o_.n_id_1, o_.cp_type, o_.n_ports, o_.n_rb_dl,
                                                                                     o_.phich_duration, o_.phich_resource, o_.sfn)
                        ok = ok and abs(float(g_["freq_superfine"]) - o_.freq_superfine) < 1e-3
                return bool(ok)

            def oracle_cells(u8, fc):
                x = u8.astype(np.float64)
                cap = ((x[0::2] - 127.0) / 128.0) + 1j * ((x[1::2] - 127.0) / 128.0)
                if args.stage == "full":
                    return O.search_capbuf(cap, f, fc, fc, FS)[0]
                r = O.xcorr_pss(cap, f, 2, fc, fc, FS)
                return O.peak_search(r["pow"], r["frq"], O.z_th1(r["sp_incoherent"], r["n_comb_xc"]), f, fc, fc, r["single"], 2)

            O.set_threads(cpu_baseline.threads)
            checked, bad = [], []
            main_ids = [b for b in (0, 1, 2, 3, 4, 8, 12, 16, 20, 24, 28, 5) if b < B]
            for b in main_ids:
                oc = cpu_baseline.results[b] if b in cpu_baseline.results else oracle_cells(host[b], float(fcs[b]))
                checked.append(f"timed/{b}")
                if not same(rec0[b, :cnt0[b]], oc):
                    bad.append(f"timed/{b}")
            if dense_rec is not None:
                for b in [b for b in (0, 1, 2, 3) if b < B]:
                    checked.append(f"dense/{b}")
                    if not same(dense_rec[0][b, :dense_rec[1][b]], oracle_cells(dense_host[b], float(fcs[b]))):
                        bad.append(f"dense/{b}")
            ok = not bad
            verify["oracle_buffer0"] = bool("timed/0" not in bad)
            verify["oracle_buffers_checked"] = checked
            verify["oracle_buffers_differing"] = bad
            out["verified"] = bool(out["verified"] and ok)
        out["verify"] = verify
        print(json.dumps(out))
    for S in ctxs:
        S.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
