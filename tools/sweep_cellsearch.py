#!/usr/bin/env python3
"""CellSearch band sweep on N GPUs of one node (BASELINE configs[3]: 715-768 MHz, 531 carriers).

    python tools/sweep_cellsearch.py -s 715e6 -e 768e6                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
           tools/sweep_cellsearch.py -s 715e6 -e 768e6                       # one rank per GPU

Mirrors the carrier loop of the reference's CLI (src/CellSearch.cpp:465-573): one capture buffer per
100 kHz raster point, the whole searcher chain on each, `dedup`, the final table.  The carriers are
sharded block-cyclically over the ranks (lte-cell-scanner_amd/sweep.py); every rank pushes its
carriers through the device-resident batch API in batches that fit HBM, and the detected-cell records
are all-gathered ONCE at the end (RCCL).  Capture buffers come from --load DIR (capbuf_NNNN.it files
as written by `CellSearch --record`) or are synthesised (most carriers empty, every `--occupied-every`-th
carries 1-2 cells), since a GPU node has no SDR.  Every rank's buffers are made RESIDENT IN HBM before the clock
starts (noise-only carriers are drawn on the device, occupied ones come from the numpy generator), so the reported
time is the search, not the signal generator; --write-it DIR dumps the same sweep as capbuf_NNNN.it files for the
C++ CLI (host/CellSearch -l -d DIR).
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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N_CAP, FS = 153600, 1.92e6


def freq_formatter(f):     # src/CellSearch.cpp:322-340
    for lim, div, suf in ((998.0, 1.0, "h"), (998e3, 1e3, "k"), (998e6, 1e6, "m"), (998e9, 1e9, "g")):
        if abs(f) < lim:
            return f"{f / div:5.3g}{suf}"
    return f"{f / 1e12:5.3g}t"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-s", "--freq-start", type=float, default=715e6)
    ap.add_argument("-e", "--freq-end", type=float, default=768e6)
    ap.add_argument("-p", "--ppm", type=float, default=120.0)
    ap.add_argument("-c", "--correction", type=float, default=1.0)
    ap.add_argument("-l", "--load", default=None, help="directory with capbuf_NNNN.it files (one per carrier)")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--occupied-every", type=int, default=16)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--dist-backend", default="nccl")
    ap.add_argument("--share-gpu0", action="store_true", help="testing only: every rank uses GPU 0")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the record all-gather even with ONE rank (the RCCL path on a one-GPU box)")
    ap.add_argument("--json", action="store_true", help="print a JSON summary line instead of the table")
    ap.add_argument("--write-it", default=None, help="also write every carrier's buffer as capbuf_NNNN.it into this directory")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    sw = pkg.sweep
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.share_gpu0:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        kw = dict(rank=rank, world_size=world)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)
        else:
            dist.init_process_group(args.dist_backend, **kw)

    # raster handling and the frequency grid exactly as the CLI: n_extra from freq_start only (CellSearch.cpp:463)
    fs_, fe_ = round(args.freq_start / 100e3) * 100e3, round(args.freq_end / 100e3) * 100e3
    fcs = sw.fc_search_set(fs_, fe_)
    f_set = pkg.f_search_set_for(fs_, args.ppm)
    fs_prog = FS * args.correction

    mine = sw.shard(len(fcs), rank, world)
    t_gen = time.perf_counter()
    resident = torch.empty((len(mine), 2 * N_CAP), dtype=torch.uint8, device=dev)      # this rank's carriers, in HBM
    truth = {}
    for j, ci in enumerate(mine):
        ci = int(ci)
        if args.load:
            cap = np.asarray(pkg.itfile.read_it(os.path.join(args.load, f"capbuf_{ci:04d}.it"))["capbuf"]).ravel()
            iq = np.empty(2 * N_CAP)
            iq[0::2], iq[1::2] = cap.real[:N_CAP], cap.imag[:N_CAP]
            resident[j] = torch.from_numpy(np.clip(np.rint(iq * 128.0 + 127.0), 0, 255).astype(np.uint8)).to(dev)
        elif ci % args.occupied_every == 0:     # a carrier's buffer depends on its index only: the same band whatever the sharding
            rng = np.random.default_rng(args.seed + ci)
            cells = [dict(n_id_1=int(rng.integers(0, 168)), n_id_2=int(rng.integers(0, 3)), cp_normal=bool(rng.integers(0, 8) != 0),
                          n_ports=int((1, 2, 2, 4)[rng.integers(0, 4)]), n_rb_dl=int((6, 15, 25, 50, 75, 100)[rng.integers(0, 6)]),
                          f_off=float(rng.uniform(-60e3, 60e3)), gain_db=-3.0 * k) for k in range(1 + int(rng.integers(0, 2)))]
            resident[j] = torch.from_numpy(pkg.synth.make_capbuf(args.seed + ci, float(fcs[ci]), cells, snr_db=float(rng.uniform(0, 10)))[0]).to(dev)
            truth[ci] = [c["n_id_2"] + 3 * c["n_id_1"] for c in cells]
        else:                                   # receiver noise only, drawn on the device (seeded by the carrier index)
            gen = torch.Generator(device=dev)
            gen.manual_seed(args.seed + ci)
            resident[j] = torch.clamp(torch.round(torch.randn(2 * N_CAP, device=dev, generator=gen) * 12.0 + 127.0), 0, 255).to(torch.uint8)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    if args.write_it:
        os.makedirs(args.write_it, exist_ok=True)
        for j, ci in enumerate(mine):
            iq = resident[j].cpu().numpy().astype(np.float64)
            pkg.itfile.write_it(os.path.join(args.write_it, f"capbuf_{int(ci):04d}.it"),
                                {"capbuf": ((iq[0::2] - 127.0) / 128.0) + 1j * ((iq[1::2] - 127.0) / 128.0), "fc": np.array([int(fcs[ci])], np.int32)})
    pos = {int(ci): j for j, ci in enumerate(mine)}

    def get_capbufs(idx):          # a contiguous run of this rank's resident buffers (run_sweep walks `mine` in order)
        return resident[pos[int(idx[0])]: pos[int(idx[0])] + len(idx)]

    S = pkg.Searcher(local)

    def search_fn(bufs, fc):
        S.batch_enqueue(bufs.data_ptr(), pkg.FMT_IQ_U8, len(fc), N_CAP, f_set, fc, fc, fs_prog, pkg.STAGE_FULL)
        return S.batch_collect_raw(len(fc), sw.MAXC)

    if len(mine):      # warm-up: workspace allocation, first launches
        search_fn(get_capbufs(mine[:min(len(mine), args.batch)]), fcs[mine[:min(len(mine), args.batch)]])
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    final, detected = sw.run_sweep(search_fn, get_capbufs, fcs, rank, world, dist,
                                   dev if (dist is not None and args.dist_backend == "nccl") else None, batch=args.batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        if args.json:
            print(json.dumps({"carriers": int(len(fcs)), "n_f": int(len(f_set)), "n_gpus": world, "seconds_search": dt,
                              "collective": (f"{dist.get_backend()} all-gather, world {world}" if dist is not None else None),
                              "carriers_per_s": len(fcs) / dt, "seconds_making_buffers_resident_rank0": t_gen,
                              "planted": {str(k): v for k, v in sorted(truth.items())} if world == 1 else None,
                              "cells": [(c["n_id_cell"], c["fc_requested"], c["n_rb_dl"], c["n_ports"]) for c in final]}))
        elif not final:
            print("No LTE cells were found...")
        else:
            print("Detected the following cells:")
            print("A: #antenna ports C: CP type ; P: PHICH duration ; PR: PHICH resource type")
            print("CID A      fc   foff RXPWR C nRB P  PR CrystalCorrectionFactor")
            for c in final:
                cp = {1: "N", 2: "E"}.get(c["cp_type"], "U")
                pd = {1: "N", 2: "E"}.get(c["phich_duration"], "U")
                pr = {1: "1/6", 2: "half", 3: "one", 4: "two"}.get(c["phich_resource"], "unk")
                k = (c["fc_requested"] - c["freq_superfine"]) / c["fc_programmed"]
                print(f"{c['n_id_cell']:3d}{c['n_ports']:2d} {c['fc_requested'] / 1e6:6.5g}M {freq_formatter(c['freq_superfine'])} "
                      f"{10 * np.log10(c['pss_pow']):5.3g} {cp} {c['n_rb_dl']:3d} {pd} {pr:>3s} {k * args.correction:.14g}")
    S.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
