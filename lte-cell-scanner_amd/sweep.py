"""Carrier-frequency sweep sharded over the GPUs of one node (host-side driver).

Mirrors the outer loop of the reference's CLI (src/CellSearch.cpp:465-573): every carrier on the
100 kHz raster gets one capture buffer and one pass of the searcher chain; results are merged
with the reference's `dedup` (:285-319).  The carriers are independent, so the sweep shards on
that axis: rank r of `world` processes (one per GPU, torch.distributed; backend "nccl" = RCCL on
ROCm, "gloo" in the CPU tests) takes carriers r, r+world, ... and runs the device-resident batch
API on them; the only communication is ONE all-gather of fixed-size cell records at the end.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import numpy as np

MAXC = 16            # cell records kept per carrier
FIELDS = ("fc_requested", "fc_programmed", "pss_pow", "freq", "frame_start", "freq_fine", "freq_superfine",
          "ind", "n_id_2", "n_id_1", "cp_type", "n_ports", "n_rb_dl", "phich_duration", "phich_resource", "sfn")


def fc_search_set(freq_start: float, freq_end: float) -> np.ndarray:
    """src/CellSearch.cpp:465: freq_start : 100 kHz : freq_end."""
    n = int(np.floor((freq_end - freq_start) / 100e3)) + 1
    return freq_start + 100e3 * np.arange(n)


def shard(n_carriers: int, rank: int, world: int) -> np.ndarray:
    """Block-cyclic assignment of carrier indices to ranks."""
    return np.arange(rank, n_carriers, world)


def cell_to_row(c) -> np.ndarray:
    return np.array([float(getattr(c, f)) for f in FIELDS], np.float64)


def row_to_dict(r: np.ndarray) -> dict:
    d = {f: (float(v) if i < 7 else int(v)) for i, (f, v) in enumerate(zip(FIELDS, r))}
    d["n_id_cell"] = d["n_id_2"] + 3 * d["n_id_1"] if (d["n_id_1"] >= 0 and d["n_id_2"] >= 0) else -1
    return d


def dedup(detected: Sequence[Sequence[dict]]) -> List[dict]:
    """src/CellSearch.cpp:285-319: same cell ID within 1 MHz -> keep the one with the larger pss_pow."""
    final: List[dict] = []
    for cells in detected:
        for c in cells:
            for i, f in enumerate(final):
                if c["n_id_cell"] == f["n_id_cell"] and \
                        abs((c["fc_requested"] + c["freq_superfine"]) - (f["fc_requested"] + f["freq_superfine"])) < 1e6:
                    if c["pss_pow"] > f["pss_pow"]:
                        final[i] = c
                    break
            else:
                final.append(c)
    return final


def run_sweep(search_fn: Callable[[np.ndarray, np.ndarray], List[list]], get_capbufs: Callable[[np.ndarray], np.ndarray],
              fcs: np.ndarray, rank: int = 0, world: int = 1, dist=None, device=None, batch: int = 64):
    """search_fn(bufs, fc_of_each) -> per-buffer lists of cell records (objects with the FIELDS
    attributes); get_capbufs(carrier_indices) -> the capture buffers of those carriers.
    Returns (cells_final, detected_per_carrier) on every rank (the all-gather leaves all ranks
    with the full list; only rank 0 normally prints it)."""
    import torch
    mine = shard(len(fcs), rank, world)
    n_max = int(np.ceil(len(fcs) / world))
    rows = np.zeros((n_max, 1 + MAXC * len(FIELDS)), np.float64)
    for a in range(0, len(mine), batch):
        idx = mine[a:a + batch]
        res = search_fn(get_capbufs(idx), fcs[idx])
        for j, cells in enumerate(res):
            rows[a + j, 0] = min(len(cells), MAXC)
            for k, c in enumerate(cells[:MAXC]):
                rows[a + j, 1 + k * len(FIELDS): 1 + (k + 1) * len(FIELDS)] = cell_to_row(c)
    if world > 1:
        t = torch.from_numpy(rows)
        if device is not None:
            t = t.to(device)
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)               # the sweep's only collective
        allrows = [g.cpu().numpy() for g in gathered]
    else:
        allrows = [rows]
    detected = [[] for _ in range(len(fcs))]
    for r in range(world):
        for j, ci in enumerate(shard(len(fcs), r, world)):
            n = int(allrows[r][j, 0])
            detected[ci] = [row_to_dict(allrows[r][j, 1 + k * len(FIELDS): 1 + (k + 1) * len(FIELDS)]) for k in range(n)]
    return dedup(detected), detected
