#!/usr/bin/env python3
"""bench.py -- capture-buffers/s of the searcher hot path on MI355X (driver contract).

A "step" is one pass of the chain over one batch of synthetic 153600-sample, 1.92 Msps
capture buffers that are ALREADY RESIDENT IN HBM when the timed region starts.  N=1 workload =
BASELINE.json configs[2], the metric's "full CellSearch": PSS correlation over the full +-100 ppm
grid at 739 MHz (n_f = 31), peak_search and every per-cell stage down to the decoded MIB
(--stage pss stops after peak_search = configs[1]).  With --gpus N (launched
by torch.distributed.run, one rank per GPU) every rank processes its own shard of buffers
(weak scaling, no data-path collective) and the detected-cell lists are all-gathered with
RCCL once per step.

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
PEAK_I8_TOPS = 5000.0         # MI355X_MICROARCH.md: I8 MFMA "~2x bf16 rate" (no spec line; 16x16x64 micro-benchmark ceiling 3944 TOPS)
PUBLISHED_BUFFERS_PER_S = 1.0 / 6.0   # ~6 s per centre frequency at ppm 100 (dual-core i7-2640)


def synth_batch(pkg, n_buf, seed, fc_list):
    """Deterministic synthetic capture buffers as raw RTL-SDR u8 I/Q bytes."""
    synth = getattr(pkg, "synth", None)
    if synth is not None:
        return synth.make_batch_u8(n_buf, seed, fc_list)
    # interim generator: recorded golden buffer rotated by a per-buffer offset + quantised noise buffers
    g = np.load(os.path.join(ROOT, "tests", "golden", "capbuf_0000.npz"))["iq_u8"]
    rng = np.random.default_rng(seed)
    out = np.empty((n_buf, 2 * N_CAP), np.uint8)
    for b in range(n_buf):
        if b % 4 == 0:
            out[b] = np.roll(g, 2 * int(rng.integers(0, N_CAP)))
        else:
            out[b] = np.clip(np.rint(rng.normal(127.0, 12.0, 2 * N_CAP)), 0, 255).astype(np.uint8)
    return out


def cpu_baseline(pkg, iq_u8, f, fc, stage):
    """The CPU oracle (a port of the reference's C++ path) on ONE buffer of the same workload,
    single thread, on this host.  Reported next to the GPU number; never part of `value`."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    iq = iq_u8.astype(np.float64)
    cap = ((iq[0::2] - 127.0) / 128.0) + 1j * ((iq[1::2] - 127.0) / 128.0)
    O.set_threads(1)
    t0 = time.perf_counter()
    if stage == "full":
        cells, peaks = O.search_capbuf(cap, f, fc, fc, FS)
    else:
        r = O.xcorr_pss(cap, f, 2, fc, fc, FS)
        peaks = O.peak_search(r["pow"], r["frq"], O.z_th1(r["sp_incoherent"], r["n_comb_xc"]), f, fc, fc, r["single"], 2)
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
    if stage == "full":
        O.search_capbuf(cap, f, fc, fc, FS)
    else:
        O.xcorr_pss(cap, f, 2, fc, fc, FS)
    dt_mt = time.perf_counter() - t0
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": 1.0 / dt, "unit": "capture-buffers/s", "cores": 1, "kind": "port",
            "sample": f"1 synthetic buffer, n_f={f.size}, stage={stage}, {dt:.2f} s single-thread "
                      f"(C oracle, gcc -O3); {ncpu} threads (OpenMP over lags as the reference): {1.0 / dt_mt:.3f} buffers/s",
            "cpu_model": model, "n_peaks": len(peaks)}


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
    def step(i):
        nonlocal n_cells
        S.stream_push(host[i % len(host)], 0.0)
        cells, _, ms = S.stream_collect()
        gpu_ms.append(ms)
        n_cells = max(n_cells, len(cells))
    for i in range(args.warmup):
        step(i)
    gpu_ms.clear()
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
                                   "buffer per step (PCIe copy included), hipGraph-captured chain, n_f = 1",
                       "gpu_ms_per_buffer": float(np.mean(gpu_ms)), "realtime_factor": 0.08 * value / world,
                       "eager_ms_per_buffer_device_resident_input": eager_ms,
                       "n_cells_in_occupied_buffers": n_cells, "parallelism": "replicas" if world > 1 else "single GPU"}}))
    S.close()
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="capture buffers per step per GPU")
    ap.add_argument("--ppm", type=float, default=100.0)
    ap.add_argument("--stage", choices=["pss", "full", "stream"], default="full",
                    help="full = BASELINE configs[2], the whole CellSearch chain (default); pss = configs[1], xcorr_pss + "
                         "peak_search only; stream = configs[4], one host buffer at a time through the hipGraph-captured "
                         "single-hypothesis chain (separate, shorter report)")
    ap.add_argument("--variant", type=int, default=0, help="PSS correlation kernel: 0 = default (int8 three-digit MFMA kernel for the u8 input), 1 = fp32 VALU twin, 2 = one-wave fp32 MFMA kernel, 3 = 4-wave fp32 MFMA kernel, 4 = bf16 three-term MFMA kernel")
    ap.add_argument("--pipeline", type=int, default=3,
                    help="contexts (streams + workspaces) used round-robin: with 2, the latency-bound per-cell "
                         "stages of step i overlap the PSS correlation of step i+1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (testing the multi-rank path on one GPU)")
    ap.add_argument("--share-gpu0", action="store_true", help="testing only: every rank uses GPU 0")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu0:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    coll_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")

    if args.stage == "stream":
        return stream_bench(pkg, args, rank, world, local_rank, dist)
    f = pkg.f_search_set_for(FC, args.ppm)
    stage_mask = pkg.STAGE_FULL if args.stage == "full" else pkg.STAGE_PSS
    B = args.batch
    # rank r searches carriers FC + 100 kHz * (r*B + b): the sweep's carrier axis is the shard axis
    fcs = FC + 100e3 * (np.arange(B) + rank * B)
    host = synth_batch(pkg, B, 1234 + rank, fcs)
    d_cap = torch.from_numpy(host).to(dev)            # inputs resident in HBM before timing starts
    ctxs = [pkg.Searcher(local_rank if world > 1 else 0) for _ in range(max(1, args.pipeline))]
    for S in ctxs:
        S.set_xcorr_variant(args.variant)
    MAXC = 16
    gather_buf = torch.zeros((world, B, 1 + MAXC * 4), dtype=torch.float64, device=coll_dev) if world > 1 else None

    host_t = {"enqueue": 0.0, "collect": 0.0, "n": 0}

    def enqueue(i):
        t = time.perf_counter()
        ctxs[i % len(ctxs)].batch_enqueue(d_cap.data_ptr(), pkg.FMT_IQ_U8, B, N_CAP, f, fcs, fcs, FS, stage_mask)
        host_t["enqueue"] += time.perf_counter() - t

    def collect(i, gather=True):
        t = time.perf_counter()
        res = ctxs[i % len(ctxs)].batch_collect(B, MAXC)
        host_t["collect"] += time.perf_counter() - t
        host_t["n"] += 1
        host_t.setdefault("stamps", []).append(time.perf_counter())
        if world > 1 and gather:   # RCCL all-gather of the detected-cell list (fixed-size records), nothing else
            mine = torch.zeros((B, 1 + MAXC * 4), dtype=torch.float64)
            for b, cells in enumerate(res):
                mine[b, 0] = len(cells)
                for i, c in enumerate(cells[:MAXC]):
                    mine[b, 1 + 4 * i: 5 + 4 * i] = torch.tensor([c.n_id_cell(), c.fc_requested, c.freq_superfine if stage_mask == 3 else c.freq, c.pss_pow])
            dist.all_gather_into_tensor(gather_buf.view(-1), mine.to(coll_dev).view(-1))
        return res

    def run(n_steps, xc_ms=None, gather=True):
        """n_steps steps, software-pipelined over the contexts: step i is enqueued before step
        i-(depth-1) is collected, so nothing but the stream of the collected step ever blocks."""
        depth = len(ctxs)
        res = None
        for i in range(n_steps + depth - 1):
            if i < n_steps:
                enqueue(i)
            j = i - (depth - 1)
            if j >= 0:
                res = collect(j, gather)
                if xc_ms is not None:
                    xc_ms.append(ctxs[j % depth].last_xcorr_ms()[0])
        return res

    # Pre-conditioning (untimed, like the warm-up): a fresh process shows one ~50 ms stall in its
    # first few hundred milliseconds of GPU activity (clock / power-state ramp); keep the GPU
    # busy for ~0.6 s so that it does not land in the timed steps of a short run.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.6:
        run(2, gather=False)      # time-based, so no collective in here (ranks may differ in count)
    host_t.update(enqueue=0.0, collect=0.0, n=0, stamps=[])
    if args.warmup:
        res = run(args.warmup)
    xc_ms = []
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run(args.steps, xc_ms)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    n_peaks = sum(len(r) for r in res)
    # the dominant kernel once more, alone on the GPU (no overlap with the other context's kernels)
    iso_ms = []
    for _ in range(3):
        ctxs[0].batch_enqueue(d_cap.data_ptr(), pkg.FMT_IQ_U8, B, N_CAP, f, fcs, fcs, FS, pkg.STAGE_PSS)
        ctxs[0].batch_collect(B, MAXC)
        iso_ms.append(ctxs[0].last_xcorr_ms()[0])
    if rank == 0:
        n_f = f.size
        value = world * B * args.steps / dt
        # dominant kernel: PSS correlation.  Algorithmic work per buffer:
        # F = 8*137*3*(N-136)*n_f real flops (reference-faithful count over all lags).
        flops_per_buf = 8.0 * 137 * 3 * (N_CAP - 136) * n_f
        flops_consumed = 8.0 * 137 * 3 * 9600 * 15 * n_f       # the 15x9600 lags that are ever used
        k_ms = float(np.mean(xc_ms))
        # Which kernel ran: u8 sources take the int8 three-digit kernel (24-bit integer templates as three int8
        # digits, i.e. 3 MFMA MACs per algorithmic MAC) or, with LCS_NO_I8 / --variant 4, the bf16 three-term kernel
        # (a*t = a*t1 + a*t2 + a*t3 with exact bf16 factors); --variant 1..3 the fp32 kernels.
        i8 = args.variant == 0 and os.environ.get("LCS_NO_I8") is None
        bf16 = (not i8) and (args.variant == 4 or (args.variant == 0 and os.environ.get("LCS_NO_BF16") is None))
        macs_factor, peak = (3.0, PEAK_I8_TOPS) if i8 else ((3.0, PEAK_BF16_TFLOPS) if bf16 else (1.0, PEAK_FP32_TFLOPS))
        kname = "k_xcorr_i8x3" if i8 else ("k_xcorr_bf16x3_unrolled<9>" if bf16 else
                                            {0: "k_xcorr_mfma_blk<4,4,32>", 1: "k_xcorr_valu", 2: "k_xcorr_mfma", 3: "k_xcorr_mfma_blk<4,4,32>"}.get(args.variant))
        # HBM traffic of the dominant kernel from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate
        # rocprofv3 passes, corrected as MI355X_MICROARCH.md prescribes): measured per buffer at this
        # n_f with the default kernel, summary committed under profiles/ -- null when not measured.
        traffic = None
        try:
            pmj = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_summary.json")))
            pm = pmj["kernels"][pmj["dominant_kernel"]]
            if int(pm["n_f"]) == int(n_f) and pmj["dominant_kernel"].startswith(kname.split("<")[0]):
                traffic = float(pm["hbm_bytes_per_buffer"]) * B
        except Exception:
            traffic = None
        achieved = macs_factor * flops_per_buf * B / (k_ms * 1e-3) / 1e12
        bytes_per_buf = 1651200 + 230400 * n_f                 # SURVEY.md section 8d compulsory HBM bytes
        out = {
            "metric": "capture-buffers/s (1.92 Msps, 153600-samp) full CellSearch; HBM GB/s vs peak",
            "value": value, "unit": "capture-buffers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": value / PUBLISHED_BUFFERS_PER_S,
            "dtype": ("i8 x 3 base-256 digits of 24-bit integer templates, i32 accumulate (exact)" if i8 else
                      "bf16x3 products (exact), f32 accumulate" if bf16 else "f32"), "data": "synthetic",
            "iq_samples_per_s": value * N_CAP,
            "config": {"workload": ("configs[2]: full searcher chain (PSS+SSS+FOE+TFG+MIB)" if args.stage == "full" else
                                    "configs[1]: xcorr_pss + peak_search over the full +-100 ppm foe grid") +
                                   f", one MI355X per rank, {B} x 153600-sample capbufs per step, fc 739 MHz + 100 kHz raster",
                       "n_f": int(n_f), "batch_per_gpu": B, "stage": args.stage, "ingest": "u8 I/Q resident in HBM",
                       "xcorr_kernel": "mfma_i32_16x16x64_i8, three int8 digits per 24-bit integer template tap" if i8 else
                                       "mfma_f32_16x16x32_bf16, three exact bf16 terms per fp32 template tap" if bf16 else
                                       ("valu_f32" if args.variant == 1 else "mfma_f32_16x16x4_f32"),
                       "pipeline_depth": len(ctxs),
                       "parallelism": f"carrier-sweep shard x{world}, RCCL all-gather of cell list" if world > 1 else "single GPU",
                       "baseline_note": "vs_baseline = value / (1 buffer per ~6 s), doc/CellSearch.html:52-54 (dual-core i7-2640, ppm 100)",
                       "n_cells_reported_last_step": n_peaks,
                       "step_done_ms": [round(1e3 * (x - t0), 2) for x in host_t.get("stamps", [])[-args.steps:]],
                       "host_ms_per_step": {"enqueue": 1e3 * host_t["enqueue"] / max(1, host_t["n"]),
                                            "collect_incl_wait": 1e3 * host_t["collect"] / max(1, host_t["n"])}},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TOP/s" if i8 else "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "flops_note": ("achieved = 3 x the algorithmic flops of SURVEY 8d (three int8 digit MACs realise one MAC with a 24-bit "
                                        "integer template tap) / kernel time, against 2x the dense bf16 MFMA peak (the 16x16x64 i8 micro-benchmark "
                                        "ceiling is 3944 TOP/s); fp32_equivalent_tflops counts the algorithmic flops once") if i8 else
                                       ("achieved = 3 x the algorithmic flops of SURVEY 8d (three exact bf16 MACs realise one fp32 MAC) / kernel time, "
                                        "against the dense bf16 MFMA peak; fp32_equivalent_tflops counts the algorithmic flops once") if bf16 else
                                       "achieved = algorithmic flops of SURVEY 8d / kernel time, against the fp32 MFMA peak",
                         "fp32_equivalent_tflops": flops_per_buf * B / (k_ms * 1e-3) / 1e12,
                         "traffic_note": "HBM bytes per launch from rocprofv3 PMC (2*FETCH_SIZE+WRITE_SIZE)*1024, profiles/r01/pmc_summary.json; "
                                         "algorithmic bytes of this kernel per launch (capture buffer in, xc_incoherent_single out): %d"
                                         % int(((2 if i8 else 4 if bf16 else 8) * N_CAP + 4 * 3 * 9600 * n_f) * B),
                         "kernel": kname, "kernel_ms": k_ms,
                         "kernel_ms_isolated": float(np.mean(iso_ms)),
                         "frac_isolated": macs_factor * flops_per_buf * B / (float(np.mean(iso_ms)) * 1e-3) / 1e12 / peak,
                         "flops_per_launch": macs_factor * flops_per_buf * B,
                         "achieved_consumed_lags_only": macs_factor * flops_consumed * B / (k_ms * 1e-3) / 1e12,
                         "hbm_algorithmic_GBps": bytes_per_buf * B / (k_ms * 1e-3) / 1e9, "hbm_peak_GBps": 8000.0,
                         "hbm_frac_algorithmic": bytes_per_buf * B / (k_ms * 1e-3) / 1e9 / 8000.0,
                         "hbm_measured_GBps": (traffic / (k_ms * 1e-3) / 1e9) if traffic else None,
                         "hbm_frac_measured": (traffic / (k_ms * 1e-3) / 1e9 / 8000.0) if traffic else None},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, host[0], f, float(fcs[0]), args.stage)
        print(json.dumps(out))
    for S in ctxs:
        S.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
