"""Carrier-frequency sweep and frequency-hypothesis split over the GPUs of one node (host-side drivers).

Two independent shard axes, both over torch.distributed (backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests), one process per GPU:

* ``run_sweep`` -- the outer loop of the reference's CLI (src/CellSearch.cpp:465-573): every carrier on the 100 kHz
  raster gets one capture buffer and one pass of the searcher chain, results are merged with the reference's
  duplicate rule (:285-319).  Carriers are independent: rank r of `world` takes carriers r, r+world, ... through the
  device-resident batch API; the only communication is ONE all-gather of the fixed-size cell records at the end
  (raw lcs_cell bytes -- a numpy structured view, no per-cell Python objects on the way).
* ``search_capbuf_foe_split`` -- ONE buffer spread over the ranks (latency mode): the frequency hypotheses are split
  into contiguous blocks, every rank correlates its block, and the blocks meet where the reference takes the maximum
  over the frequency axis (xc_peak_freq, src/searcher.cpp:369-382): a MAX all-reduce over 3 x 9600 packed words
  ``bits(pow_f32) << 32 | (0xFFFFFFFF - foi)`` -- non-negative floats order like their bit patterns, and the
  complemented index makes the LOWEST foi win a tie, as the reference's strict `>` does (:374).  peak_search then runs
  identically everywhere; the per-peak stages run on the rank that owns the winning hypothesis (it holds the
  xc_incoherent_single slice the refinement step reads, :457-465) and the decoded cells are all-gathered.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import numpy as np

MAXC = 104   # == capi.MAX_PEAKS: cell records kept per carrier: the longest list the reference can return (include/lcs.h LCS_MAX_PEAKS)
FIELDS = ("fc_requested", "fc_programmed", "pss_pow", "freq", "frame_start", "freq_fine", "freq_superfine",
          "ind", "n_id_2", "n_id_1", "cp_type", "n_ports", "n_rb_dl", "phich_duration", "phich_resource", "sfn")


def cell_dtype():
    from . import capi
    return capi.cell_dtype()


def fc_search_set(freq_start: float, freq_end: float) -> np.ndarray:
    """src/CellSearch.cpp:465: freq_start : 100 kHz : freq_end."""
    n = int(np.floor((freq_end - freq_start) / 100e3)) + 1
    return freq_start + 100e3 * np.arange(n)


def shard(n_carriers: int, rank: int, world: int) -> np.ndarray:
    """Block-cyclic assignment of carrier indices to ranks."""
    return np.arange(rank, n_carriers, world)


def cells_to_records(cells) -> np.ndarray:
    """Cell objects (LcsCell or anything with the FIELDS attributes) -> structured array with lcs_cell's layout."""
    rec = np.zeros(len(cells), cell_dtype())
    for i, c in enumerate(cells):
        for f in FIELDS:
            rec[i][f] = getattr(c, f)
    return rec


def record_to_dict(r) -> dict:
    d = {f: (float(r[f]) if r.dtype[f].kind == "f" else int(r[f])) for f in FIELDS}
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


def _all_gather_bytes(arr: np.ndarray, dist, device, world: int) -> List[np.ndarray]:
    """One all-gather of a fixed-size numpy array (viewed as bytes) -> the array of every rank."""
    import torch
    t = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy())
    if device is not None:
        t = t.to(device)
    out = torch.empty((world, t.numel()), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out.view(-1), t)
    host = out.cpu().numpy()
    return [host[r].view(arr.dtype).reshape(arr.shape) for r in range(world)]


def run_sweep(search_fn: Callable, get_capbufs: Callable[[np.ndarray], object], fcs: np.ndarray, rank: int = 0, world: int = 1,
              dist=None, device=None, batch: int = 64):
    """search_fn(bufs, fc_of_each) -> (records [n][MAXC] structured, counts [n]) as Searcher.batch_collect_raw returns
    them, or per-buffer lists of cell objects; get_capbufs(carrier_indices) -> the capture buffers of those carriers in
    whatever form search_fn takes.  Returns (cells_final, detected_per_carrier) on every rank (the all-gather leaves
    all ranks with the full list; only rank 0 normally prints it)."""
    mine = shard(len(fcs), rank, world)
    n_max = int(np.ceil(len(fcs) / world))
    rec = np.zeros((n_max, MAXC), cell_dtype())
    cnt = np.zeros(n_max, np.int32)
    for a in range(0, len(mine), batch):
        idx = mine[a:a + batch]
        res = search_fn(get_capbufs(idx), fcs[idx])
        if isinstance(res, tuple):
            r, c = res
            rec[a:a + len(idx)] = r[:, :MAXC]
            cnt[a:a + len(idx)] = np.minimum(c, MAXC)
            if np.any(np.asarray(c) > MAXC):
                import warnings
                warnings.warn(f"run_sweep: a carrier reported more than {MAXC} cells; the list was truncated", RuntimeWarning)
        else:
            for j, cells in enumerate(res):
                cnt[a + j] = min(len(cells), MAXC)
                rec[a + j, :cnt[a + j]] = cells_to_records(cells[:MAXC])
    if dist is not None:      # the sweep's only collective: counts ride in front of the records.  Also run at world 1 when
        # a process group is live: the RCCL path of an 8-GPU node exercised on one GPU (tests/test_gpu_rccl.py)
        blob = np.concatenate([cnt.view(np.uint8), rec.view(np.uint8).reshape(-1)])
        blobs = _all_gather_bytes(blob, dist, device, world)
        parts = [(b[:cnt.nbytes].view(np.int32), b[cnt.nbytes:].view(cell_dtype()).reshape(rec.shape)) for b in blobs]
    else:
        parts = [(cnt, rec)]
    detected = [[] for _ in range(len(fcs))]
    for r in range(world):
        c_r, rec_r = parts[r]
        for j, ci in enumerate(shard(len(fcs), r, world)):
            detected[ci] = [record_to_dict(rec_r[j, k]) for k in range(int(c_r[j]))]
    return dedup(detected), detected


class SearcherStages:
    """The searcher.h functions of a GPU `Searcher` in the form search_capbuf_foe_split takes."""

    def __init__(self, searcher, z_th1_fn):
        self._s, self.z_th1 = searcher, z_th1_fn
        for name in ("peak_search", "sss_detect", "pss_sss_foe", "extract_tfg", "tfoec", "decode_mib"):
            setattr(self, name, getattr(searcher, name))

    def xcorr_pss(self, capbuf, f, ds, fc_req, fc_prog, fs):
        return self._s.xcorr_pss(capbuf, f, ds, fc_req, fc_prog, fs, want_incoherent=False)


# ---------------------------------------------------------------------------------------------- foe-axis split
def foe_blocks(n_f: int, world: int) -> List[np.ndarray]:
    """Contiguous, near-equal blocks of hypothesis indices, one per rank (some empty when world > n_f)."""
    edges = np.linspace(0, n_f, world + 1).round().astype(int)
    return [np.arange(edges[r], edges[r + 1]) for r in range(world)]


def pack_pow_frq(pow_: np.ndarray, frq_global: np.ndarray) -> np.ndarray:
    """(pow as float32-exact doubles, global foi) -> int64 words whose MAX is the reference's first-maximum rule."""
    bits = pow_.astype(np.float32).view(np.uint32).astype(np.int64)
    return (bits << 32) | (0xFFFFFFFF - frq_global.astype(np.int64))


def unpack_pow_frq(words: np.ndarray):
    pow32 = (words >> 32).astype(np.uint32).view(np.float32)
    return pow32.astype(np.float64), (0xFFFFFFFF - (words & 0xFFFFFFFF)).astype(np.int32)


def search_capbuf_foe_split(stages, capbuf, f_search_set, fc_requested, fc_programmed, fs_programmed, rank=0, world=1,
                            dist=None, device=None, ds_comb_arm=2, thresh2_n_sigma=3.0):
    """One capture buffer, hypotheses split over the ranks.  `stages` offers the searcher.h functions
    (xcorr_pss, peak_search, sss_detect, pss_sss_foe, extract_tfg, tfoec, decode_mib, z_th1): a Searcher plus the
    package's z_th1, or the oracle in the CPU tests.  Returns (cells in peak order, dict of the collapsed arrays)."""
    import torch
    f = np.asarray(f_search_set, np.float64)
    blocks = foe_blocks(f.size, world)
    own = blocks[rank]
    words = np.full((3, 9600), -1, np.int64)        # smaller than any packed word (pow >= 0 packs non-negative)
    single_full = np.zeros((3, 9600, f.size), np.float32)
    sp_inc, n_comb_xc = None, 0
    if own.size:
        r = stages.xcorr_pss(capbuf, f[own], ds_comb_arm, fc_requested, fc_programmed, fs_programmed)
        words = pack_pow_frq(r["pow"], own[0] + r["frq"])
        single_full[:, :, own] = r["single"]
        sp_inc, n_comb_xc = r["sp_incoherent"], int(r["n_comb_xc"])
    if dist is not None:
        t = torch.from_numpy(words)
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)         # where the hypotheses meet: MAX with index, 230 KB
        words = t.cpu().numpy()
        # sp_incoherent does not depend on the hypotheses: a rank with an empty block takes it from rank 0
        meta = torch.zeros(9601, dtype=torch.float64)
        if rank == 0:
            meta[:9600] = torch.from_numpy(sp_inc)
            meta[9600] = n_comb_xc
        if device is not None:
            meta = meta.to(device)
        dist.broadcast(meta, src=0)
        meta = meta.cpu().numpy()
        sp_inc, n_comb_xc = meta[:9600].copy(), int(meta[9600])
    pow_, frq = unpack_pow_frq(words)
    Z = stages.z_th1(sp_inc, n_comb_xc, ds_comb_arm)
    # the greedy loop depends on the collapsed arrays only; the refinement reads `single` at the winning hypothesis
    peaks = stages.peak_search(pow_, frq, Z, f, fc_requested, fc_programmed, single_full, ds_comb_arm)
    mine = []
    for order, pk in enumerate(peaks):
        foi = int(np.flatnonzero(f == pk.freq)[0])           # the hypothesis the peak was found at (lowest index on duplicates)
        if foi not in own:
            continue
        c = stages.sss_detect(pk, capbuf, thresh2_n_sigma, fc_requested, fc_programmed, fs_programmed)
        c = c[0] if isinstance(c, tuple) else c
        if c.n_id_1 == -1:
            continue
        c = stages.pss_sss_foe(c, capbuf, fc_requested, fc_programmed, fs_programmed)
        tfg, ts = stages.extract_tfg(c, capbuf, fc_requested, fc_programmed, fs_programmed)
        c, tfg_comp, _ = stages.tfoec(c, tfg, ts, fc_requested, fc_programmed)
        c = stages.decode_mib(c, tfg_comp)
        if c.n_rb_dl == -1:
            continue
        mine.append((order, c))
    rec = np.zeros(MAXC, cell_dtype())
    order = np.full(MAXC, -1, np.int32)
    if mine:
        rec[:len(mine)] = cells_to_records([c for _, c in mine[:MAXC]])
        order[:len(mine)] = [o for o, _ in mine[:MAXC]]
    if dist is not None:
        blobs = _all_gather_bytes(np.concatenate([order.view(np.uint8), rec.view(np.uint8)]), dist, device, world)
        allc = []
        for b in blobs:
            o, rr = b[:order.nbytes].view(np.int32), b[order.nbytes:].view(cell_dtype())
            allc += [(int(o[k]), record_to_dict(rr[k])) for k in range(MAXC) if o[k] >= 0]
    else:
        allc = [(int(order[k]), record_to_dict(rec[k])) for k in range(MAXC) if order[k] >= 0]
    allc.sort(key=lambda x: x[0])
    return [c for _, c in allc], dict(pow=pow_, frq=frq, sp_incoherent=sp_inc, n_comb_xc=n_comb_xc)


def search_capbuf_foe_split_dev(searcher, capbuf, f_search_set, fc_requested, fc_programmed, fs_programmed, rank=0, world=1,
                                dist=None, device=None):
    """The same split with everything between the correlation and the peak search staying on the GPUs: `searcher`
    (a Searcher on this rank's GPU) leaves the packed (pow, ~foi) words and the power estimate in two torch tensors on
    `device` (lcs_foe_partial), torch.distributed reduces / broadcasts them in place (RCCL: one 230 KB MAX all-reduce, one
    77 KB broadcast, and a second 230 KB MAX all-reduce of the exactly recomputed near-ties, lcs_foe_contend -- no host copy of
    any array), lcs_foe_finish takes them back.  Only the decoded cell records (a few
    hundred bytes) travel through the host, in one all-gather.  Returns (cells in peak order, peak list)."""
    import torch
    f = np.asarray(f_search_set, np.float64)
    own = foe_blocks(f.size, world)[rank]
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    words = torch.empty(3 * 9600, dtype=torch.int64, device=dev)
    meta = torch.empty(9601, dtype=torch.float64, device=dev)
    searcher.foe_partial(capbuf, f, int(own[0]) if own.size else 0, int(own.size), fc_requested, fc_programmed, fs_programmed,
                         words.data_ptr(), meta.data_ptr())
    if dist is not None:
        dist.all_reduce(words, op=dist.ReduceOp.MAX)
        dist.broadcast(meta, src=0)
        torch.cuda.synchronize(dev)
    # near-ties of the arg-max (within one rank's share or across two): every contending rank recomputes its contenders and the
    # global winner in the reference's arithmetic; a second MAX all-reduce makes the exact winner everybody's (lcs.h)
    words2 = torch.empty(3 * 9600, dtype=torch.int64, device=dev)
    searcher.foe_contend(f, words.data_ptr(), words2.data_ptr())
    if dist is not None:
        dist.all_reduce(words2, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize(dev)
    searcher.foe_resolve(words.data_ptr(), words2.data_ptr())
    cells, order, peaks = searcher.foe_finish(words.data_ptr(), meta.data_ptr(), f)
    rec = np.zeros(MAXC, cell_dtype())
    ordv = np.full(MAXC, -1, np.int32)
    n = min(len(cells), MAXC)
    if n:
        rec[:n] = cells_to_records(cells[:n])
        ordv[:n] = order[:n]
    if dist is not None:
        gdev = dev if dist.get_backend() == "nccl" else None        # the record gather of a gloo test run goes through host tensors
        blobs = _all_gather_bytes(np.concatenate([ordv.view(np.uint8), rec.view(np.uint8)]), dist, gdev, world)
    else:
        blobs = [np.concatenate([ordv.view(np.uint8), rec.view(np.uint8)])]
    allc = []
    for b in blobs:
        o, rr = b[:ordv.nbytes].view(np.int32), b[ordv.nbytes:].view(cell_dtype())
        allc += [(int(o[k]), record_to_dict(rr[k])) for k in range(MAXC) if o[k] >= 0]
    allc.sort(key=lambda x: x[0])
    return [c for _, c in allc], peaks
