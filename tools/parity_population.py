#!/usr/bin/env python3
"""Population parity: the GPU chain and the CPU oracle on the SAME >= 1152 capture buffers, every decision compared.

SURVEY.md section 7 (hard part 1) asks for parity on a population and a stated disagreement rate for the chain's threshold
decisions: Z_th1 (src/CellSearch.cpp:500-503), the 3-sigma SSS test (src/searcher.cpp:751-758), the PBCH CRC (:1628-1636).
The buffers:

  synthetic  512 buffers shaped like Matlab/create_dl_sig.m:45-112 + a PBCH: 64 scenes x 8 noise realisations -- 0-3 cells,
             SNR -12 / -9 / -6 / 0 / +10 dB, both CP types, 1 / 2 / 4 ports, every bandwidth, LO errors over the whole grid,
             a third of the scenes with fc_programmed != fc_requested and fs_programmed != 1.92 MHz, grids n_f = 31 / 35 / 37
  bench      the 4 x 128 buffers bench.py times (its base batch and the three rolled copies)
  dense      the 128 buffers of bench.py's dense band (2-3 cells planted in every buffer)
  channels   (round 6) 512 buffers whose cells went through a RADIO CHANNEL, not a gain: 64 scenes x 8 noise realisations -- per
             (cell, antenna port) an independent Rayleigh tapped delay line with the EPA / EVA / ETU delay profile of 36.101 B.2
             (up to 9.6 samples of delay spread: beyond the normal CP) and Jakes Doppler 5 / 70 / 300 Hz, 1-3 cells, both CP
             types, 1 / 2 / 4 ports, SNR -9 .. +10 dB, and the front end of a zero-IF dongle: DC spike, I/Q gain and phase
             imbalance, hard clipping of the 8-bit range.  These are the stages the reference's fixtures do not pin:
             ce_interp_hex on a frequency-selective, time-varying channel (src/searcher.cpp:1223-1362), tfoec's timing estimate
             (:1013-1058), 4-port SFBC (:1582-1611).  For every cell the oracle decodes in this group the ARRAYS are compared
             too: extract_tfg's grid (1e-10), tfoec's compensated grid (1e-9), chan_est's estimate of every port (1e-9) and its
             noise power (1e-11), through the stage entry points on the oracle's own inputs.

  highband   (round 6) 96 buffers on the WIDE grids the reference's CLI builds at its default 120 ppm above 1 GHz
             (src/CellSearch.cpp:463-465): n_f = 61 (1.25 GHz), 87 (1.8 GHz, band 3), 103 (2.14 GHz, band 1), 125 (2.6 GHz, band 7),
             169 (3.5 GHz, bands 42 / 43), 141 (2.9 GHz on a 120 ppm crystal that the dongle programmes 9 ppm off) -- 12 scenes x 8
             noise realisations, 1-2 cells through the fading channels of the `channels` group, LO errors out to 0.9 of the
             grid's edge (up to 380 kHz), the stage arrays of every decoded cell compared as there.

Per buffer, GPU (lcs_batch_enqueue / lcs_batch_collect / lcs_batch_readback) against oracle (oracle/lcs_oracle.c, one
process per host core):
  xc_incoherent_collapsed_frq   every one of the 3 x 9600 indices EQUAL
  xc_incoherent_collapsed_pow   1e-5 relative (north_star)
  Z_th1                          1e-10
  peak list (peak_search)        n_id_2, ind, freq EQUAL, in order; pss_pow 1e-5
  per peak                       found / rejected by sss_detect, by decode_mib: EQUAL
  per cell                       n_id_1, cp_type, n_ports, n_rb_dl, phich_duration, phich_resource, sfn EQUAL;
                                 frame_start 1e-6 samples, freq_fine 1e-4 Hz, freq_superfine 1e-3 Hz
Every disagreement is reported with the oracle's own margin to the threshold that decides it.  Writes one JSON file.

TEST TOOLING (it imports oracle/): used by tests/test_gpu_population.py and run by hand for profiles/r05/.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

N_CAP = 153600
FS = 1.92e6
FC = 739e6
SNRS = (-12.0, -9.0, -6.0, 0.0, 10.0, -9.0, -6.0, 0.0)


def f_grid(freq_start, ppm):
    n_extra = int(np.floor((freq_start * ppm / 1e6 + 2.5e3) / 5e3))
    return np.arange(-n_extra, n_extra + 1) * 5000.0


GRIDS = ((739e6, 100.0), (715e6, 120.0), (739e6, 120.0))      # n_f = 31 (bench), 35 (configs[3]), 37 (the CLI default)


def synthetic_scene(synth, s, seed_offset=0):
    """Scene s of 64: the noise-free signal of 0-3 cells and the parameters the receiver reports."""
    rng = np.random.default_rng(50_000 + s + 1000 * seed_offset)
    fc, ppm = GRIDS[s % 3]
    f = f_grid(fc, ppm)
    fc_req = fc + 100e3 * (s % 7)
    dongle = (s % 3 == 1)                                        # the dongle's fc_programmed / fs_programmed differ from the request
    fc_prog = fc_req * (1 + 17e-6) if dongle else fc_req
    fs_prog = FS * (1 - 23e-6) if dongle else FS
    n_cells = (s // 3) % 4
    cells = []
    for j in range(n_cells):
        cells.append(dict(n_id_1=int(rng.integers(0, 168)), n_id_2=int(rng.integers(0, 3)), cp_normal=bool((s + j) % 3 != 0),
                          n_ports=int((1, 2, 4)[(s + j) % 3]), n_rb_dl=int((6, 15, 25, 50, 75, 100)[(s + 2 * j) % 6]),
                          phich_duration_ext=int((s + j) % 2), phich_res=int((s + j) % 4),
                          f_off=float(rng.uniform(-0.9, 0.9) * f[-1]), gain_db=-3.0 * j))
    sig, ref_pow, _ = synth.make_signal(rng, fc_req, cells, N_CAP, fc_prog, fs_prog)
    return dict(sig=sig, ref_pow=ref_pow, f=f, fc_req=fc_req, fc_prog=fc_prog, fs_prog=fs_prog, planted=cells)


CH_SNRS = (-9.0, -6.0, -3.0, 0.0, 3.0, 6.0, 10.0, 10.0)
CH_PROFILES = ("EPA", "EVA", "ETU")
CH_DOPPLER = (5.0, 70.0, 300.0)


def channel_scene(synth, s, seed_offset=0):
    """Scene s of 64 of the `channels` group: 1-3 cells, each antenna port of each cell through its own fading multipath channel."""
    rng = np.random.default_rng(70_000 + s + 1000 * seed_offset)
    fc, ppm = GRIDS[s % 3]
    f = f_grid(fc, ppm)
    fc_req = fc + 100e3 * (s % 5)
    dongle = (s % 4 == 3)
    fc_prog = fc_req * (1 - 11e-6) if dongle else fc_req
    fs_prog = FS * (1 + 19e-6) if dongle else FS
    n_cells = 1 + (s // 9) % 3
    cells = []
    for j in range(n_cells):
        cells.append(dict(n_id_1=int(rng.integers(0, 168)), n_id_2=int(rng.integers(0, 3)), cp_normal=bool((s + j) % 4 != 1),
                          n_ports=int((2, 1, 4, 2, 4)[(s + j) % 5]), n_rb_dl=int((6, 15, 25, 50, 75, 100)[(s + j) % 6]),
                          phich_duration_ext=int((s + j) % 2), phich_res=int((s + 3 * j) % 4),
                          f_off=float(rng.uniform(-0.9, 0.9) * f[-1]), gain_db=-2.5 * j,
                          channel=CH_PROFILES[(s + j) % 3], doppler_hz=CH_DOPPLER[((s // 3) + j) % 3]))
    sig, ref_pow, _ = synth.make_signal(rng, fc_req, cells, N_CAP, fc_prog, fs_prog)
    # the front end: none / DC spike / I-Q imbalance / both
    fe = (None, dict(dc=0.25 - 0.15j), dict(iq_gain_db=0.6, iq_phase_deg=4.0), dict(dc=-0.1 + 0.4j, iq_gain_db=-0.4, iq_phase_deg=-3.0))[s % 4]
    return dict(sig=sig, ref_pow=ref_pow, f=f, fc_req=fc_req, fc_prog=fc_prog, fs_prog=fs_prog, planted=cells, front_end=fe)


HB_GRIDS = ((1.25e9, 120.0), (1.8e9, 120.0), (2.14e9, 120.0), (2.6e9, 120.0), (3.5e9, 120.0), (2.9e9, 120.0))     # n_f = 61, 87, 103, 125, 169, 141


def highband_scene(synth, s, seed_offset=0):
    """Scene s of 12 of the `highband` group: a wide hypothesis grid, 1-2 cells through fading channels, large LO errors."""
    rng = np.random.default_rng(80_000 + s + 1000 * seed_offset)
    fc, ppm = HB_GRIDS[s % 6]
    f = f_grid(fc, ppm)
    fc_req = fc + 100e3 * (s % 3)
    dongle = (s % 6 == 5)
    fc_prog = fc_req * (1 + 9e-6) if dongle else fc_req
    fs_prog = FS * (1 - 14e-6) if dongle else FS
    cells = []
    for j in range(1 + (s // 6)):
        # the first cell of a scene sits near an EDGE of the grid (alternating sides), the second anywhere
        off = (0.9 if s % 2 else -0.9) * f[-1] + float(rng.uniform(-6e3, 6e3)) if j == 0 else float(rng.uniform(-0.9, 0.9) * f[-1])
        cells.append(dict(n_id_1=int(rng.integers(0, 168)), n_id_2=int(rng.integers(0, 3)), cp_normal=bool((s + j) % 3 != 2),
                          n_ports=int((1, 2, 4)[(s + j) % 3]), n_rb_dl=int((6, 15, 25, 50, 75, 100)[(s + 4 * j) % 6]),
                          phich_duration_ext=int((s + j) % 2), phich_res=int((s + j) % 4), f_off=off, gain_db=-3.0 * j,
                          channel=CH_PROFILES[(s + j) % 3], doppler_hz=CH_DOPPLER[(s + j) % 3]))
    sig, ref_pow, _ = synth.make_signal(rng, fc_req, cells, N_CAP, fc_prog, fs_prog)
    fe = (None, dict(dc=0.2 + 0.1j), dict(iq_gain_db=0.5, iq_phase_deg=-3.0))[s % 3]
    return dict(sig=sig, ref_pow=ref_pow, f=f, fc_req=fc_req, fc_prog=fc_prog, fs_prog=fs_prog, planted=cells, front_end=fe)


def _highband_scene_job(args):
    import __graft_entry__ as ge
    s, seed_offset = args
    return highband_scene(ge.load_package().synth, s, seed_offset)


def _channel_scene_job(args):
    import __graft_entry__ as ge
    s, seed_offset = args
    return channel_scene(ge.load_package().synth, s, seed_offset)


def build_population(pkg, groups, limit=None, dense_limit=None, seed_offset=0, pool=None):
    """-> list of dict(name, group, iq (uint8 [2 n_cap]), f, fc_req, fc_prog, fs_prog, n_planted)."""
    import bench
    synth = pkg.synth
    items = []
    if "synthetic" in groups:
        n_scenes = 64 if limit is None else max(1, min(64, limit // 8))
        for s in range(n_scenes):
            sc = synthetic_scene(synth, s, seed_offset)
            for v in range(8):
                rng = np.random.default_rng(90_000 + 8 * s + v + 10_000 * seed_offset)
                sig = np.roll(sc["sig"], int(rng.integers(0, N_CAP))) if v else sc["sig"]
                iq = synth.add_noise_and_quantise(rng, sig, sc["ref_pow"], SNRS[v], rms=float(rng.uniform(0.08, 0.22)))
                items.append(dict(name=f"synthetic/scene{s:02d}/snr{SNRS[v]:+.0f}dB/v{v}", group="synthetic", iq=iq, f=sc["f"],
                                  fc_req=sc["fc_req"], fc_prog=sc["fc_prog"], fs_prog=sc["fs_prog"], n_planted=len(sc["planted"]),
                                  snr_db=SNRS[v]))
    if "channels" in groups:
        n_scenes = 64 if limit is None else max(1, min(64, limit // 8))
        jobs = [(s, seed_offset) for s in range(n_scenes)]
        scenes = pool.map(_channel_scene_job, jobs, chunksize=1) if pool is not None else [_channel_scene_job(j) for j in jobs]
        for s, sc in enumerate(scenes):
            for v in range(8):
                rng = np.random.default_rng(95_000 + 8 * s + v + 10_000 * seed_offset)
                sig = np.roll(sc["sig"], int(rng.integers(0, N_CAP))) if v else sc["sig"]
                # realisations 6 and 7 drive the ADC into its rails (0.8 % / 5 % of the bytes at the 0 / 255 stops)
                rms = (0.5, 0.75)[v - 6] if v >= 6 else float(rng.uniform(0.08, 0.22))
                iq = synth.add_noise_and_quantise(rng, sig, sc["ref_pow"], CH_SNRS[v], rms=rms, front_end=sc["front_end"])
                items.append(dict(name=f"channels/scene{s:02d}/snr{CH_SNRS[v]:+.0f}dB/v{v}", group="channels", iq=iq, f=sc["f"],
                                  fc_req=sc["fc_req"], fc_prog=sc["fc_prog"], fs_prog=sc["fs_prog"], n_planted=len(sc["planted"]),
                                  snr_db=CH_SNRS[v], arrays=True))
    if "highband" in groups:
        n_scenes = 12 if limit is None else max(1, min(12, limit // 8))
        jobs = [(s, seed_offset) for s in range(n_scenes)]
        scenes = pool.map(_highband_scene_job, jobs, chunksize=1) if pool is not None else [_highband_scene_job(j) for j in jobs]
        for s, sc in enumerate(scenes):
            for v in range(8):
                rng = np.random.default_rng(97_000 + 8 * s + v + 10_000 * seed_offset)
                sig = np.roll(sc["sig"], int(rng.integers(0, N_CAP))) if v else sc["sig"]
                iq = synth.add_noise_and_quantise(rng, sig, sc["ref_pow"], CH_SNRS[v], rms=float(rng.uniform(0.08, 0.22)), front_end=sc["front_end"])
                items.append(dict(name=f"highband/scene{s:02d}/nf{sc['f'].size}/snr{CH_SNRS[v]:+.0f}dB/v{v}", group="highband", iq=iq, f=sc["f"],
                                  fc_req=sc["fc_req"], fc_prog=sc["fc_prog"], fs_prog=sc["fs_prog"], n_planted=len(sc["planted"]),
                                  snr_db=CH_SNRS[v], arrays=True))
    fcs = FC + 100e3 * np.arange(128)
    f31 = f_grid(FC, 100.0)
    if "bench" in groups:
        base = bench.synth_batch(pkg, 128, 1234, fcs)
        for d in range(4):
            b = base if d == 0 else np.roll(base, 2 * 1117 * d, axis=1)      # bench.py: torch.roll(base, shifts=2 * 1117 * d, dims=1)
            for k in range(128 if limit is None else min(128, max(1, limit // 4))):
                items.append(dict(name=f"bench/batch{d}/buffer{k:03d}", group="bench", iq=np.ascontiguousarray(b[k]), f=f31,
                                  fc_req=float(fcs[k]), fc_prog=float(fcs[k]), fs_prog=FS, n_planted=None, snr_db=None))
    if "dense" in groups:
        dh = bench.synth_batch(pkg, 128, 4321, fcs, dense=True)
        for k in range(128 if (limit is None and dense_limit is None) else min(128, dense_limit or limit)):
            items.append(dict(name=f"dense/buffer{k:03d}", group="dense", iq=np.ascontiguousarray(dh[k]), f=f31, fc_req=float(fcs[k]),
                              fc_prog=float(fcs[k]), fs_prog=FS, n_planted=None, snr_db=None))
    return items


# ----------------------------------------------------------------------------------------------------- oracle side
CELL_INT = ("n_id_2", "ind", "n_id_1", "cp_type", "n_ports", "n_rb_dl", "phich_duration", "phich_resource", "sfn")
CELL_FLT = ("pss_pow", "freq", "frame_start", "freq_fine", "freq_superfine")


def cell_to_dict(c):
    d = {k: int(getattr(c, k)) for k in CELL_INT}
    d.update({k: float(getattr(c, k)) for k in CELL_FLT})
    return d


def oracle_job(it):
    """The reference's per-buffer sequence (src/CellSearch.cpp:484-558) stage by stage on the oracle, keeping what the
    comparison needs: the collapsed arrays, the peak list, every peak's fate and the margins of the three threshold tests."""
    import oracle as O
    O.set_legacy(False)
    O.set_threads(1)
    t0 = time.perf_counter()
    x = it["iq"].astype(np.float64)
    cap = ((x[0::2] - 127.0) / 128.0) + 1j * ((x[1::2] - 127.0) / 128.0)
    f, fr, fp, fs = it["f"], it["fc_req"], it["fc_prog"], it["fs_prog"]
    ro = O.xcorr_pss(cap, f, 2, fr, fp, fs)
    Z = O.z_th1(ro["sp_incoherent"], ro["n_comb_xc"])
    peaks = O.peak_search(ro["pow"], ro["frq"], Z, f, fr, fp, ro["single"], 2)
    # replay of the greedy loop (src/searcher.cpp:440-505) on the oracle's own arrays: every accepted maximum's ratio to its
    # threshold, and the ratio of the first REJECTED one (where the loop stopped) -- the margins of the Z_th1 decisions
    w = ro["pow"].copy()
    z_ratio = []
    for _ in peaks:
        r, c = np.unravel_index(int(np.argmax(w)), w.shape)
        pk = float(w[r, c])
        z_ratio.append(pk / float(Z[c]))
        w[r, (c + np.arange(-274, 275)) % 9600] = 0
        w[w < pk * 10.0 ** (-12.0 / 10.0)] = 0
    r, c = np.unravel_index(int(np.argmax(w)), w.shape)
    z_rejected = float(w[r, c]) / float(Z[c])
    out_peaks = []
    for p, zr in zip(peaks, z_ratio):
        rec = dict(peak=cell_to_dict(p), z_ratio=zr)
        c1, dbg = O.sss_detect(p, cap, 3.0, fr, fp, fs)
        L = np.concatenate([dbg["ll_nrm"].T.reshape(-1), dbg["ll_ext"].T.reshape(-1)])
        mean, sd = float(L.mean()), float(np.sqrt(L.var(ddof=1)))
        ll = dbg["ll_nrm"] if dbg["ll_nrm"].max() > dbg["ll_ext"].max() else dbg["ll_ext"]
        col = 0 if ll[:, 0].max() > ll[:, 1].max() else 1
        rec["sss_sigma"] = float((ll[:, col].max() - mean) / sd)       # the test is >= 3 (src/searcher.cpp:751-758)
        rec["sss_found"] = bool(c1.n_id_1 != -1)
        rec["mib_found"] = False
        if rec["sss_found"]:
            c2 = O.pss_sss_foe(c1, cap, fr, fp, fs)
            tfg, ts = O.extract_tfg(c2, cap, fr, fp, fs)
            c3, tfgc, _ = O.tfoec(c2, tfg, ts, fr, fp)
            c4 = O.decode_mib(c3, tfgc)
            rec["mib_found"] = bool(c4.n_rb_dl != -1)
            rec["cell"] = cell_to_dict(c4)
            if it.get("arrays") and rec["mib_found"]:
                # the stage arrays of a decoded cell, for the comparison through the GPU's stage entry points (on THESE inputs)
                ce = [O.chan_est(c3, tfgc, port) for port in range(int(c4.n_ports))]
                rec["arrays"] = dict(c2=cell_to_dict(c2), c3=cell_to_dict(c3), tfg=tfg, ts=ts, tfgc=tfgc, ce=[x[0] for x in ce], np=[float(x[1]) for x in ce])
        out_peaks.append(rec)
    return dict(name=it["name"], frq=ro["frq"].astype(np.int16), pow=ro["pow"].astype(np.float32), zth=Z, peaks=out_peaks,
                z_rejected=z_rejected, seconds=time.perf_counter() - t0, frq_margin=_frq_margins(ro))


def _frq_margins(ro):
    """Per position: relative gap between the oracle's best and second-best hypothesis (float32) -- what a differing index is
    judged against."""
    inc = ro["incoherent"]
    if inc.shape[2] < 2:
        return np.ones(inc.shape[:2], np.float32)
    part = np.partition(inc, inc.shape[2] - 2, axis=2)
    best, second = part[:, :, -1], part[:, :, -2]
    return ((best - second) / best).astype(np.float32)


# -------------------------------------------------------------------------------------------------------- GPU side
def gpu_pass(pkg, items, batch=128, input_fmt="u8"):
    """Every buffer through the batched, device-resident chain; buffers that share (grid, fs_programmed) share batches.
    input_fmt "c64": the same samples handed over as complex<float> ((u8 - 127) / 128 is exact in fp32) -- LCS_FMT_C64 batches take
    the fp16 three-product kernel (k_xcorr_f16x3) and every later stage reads the float buffer in place."""
    import torch
    out = [None] * len(items)
    keys = {}
    for i, it in enumerate(items):
        keys.setdefault((it["f"].tobytes(), it["fs_prog"]), []).append(i)
    n_repairs = 0
    t_gpu = 0.0
    kernels = set()
    with pkg.Searcher(0) as S:
        for (_, fs), idx in keys.items():
            f = items[idx[0]]["f"]
            for a in range(0, len(idx), batch):
                ids = idx[a:a + batch]
                host = np.stack([items[i]["iq"] for i in ids])
                fmt = pkg.FMT_IQ_U8
                if input_fmt == "c64":
                    x = (host.astype(np.float32) - 127.0) / 128.0
                    host = np.ascontiguousarray(x).view(np.complex64)
                    fmt = pkg.FMT_C64
                d = torch.from_numpy(host).cuda()
                fr = np.array([items[i]["fc_req"] for i in ids])
                fp = np.array([items[i]["fc_prog"] for i in ids])
                t0 = time.perf_counter()
                pk = S.search_batch(d.data_ptr(), fmt, len(ids), N_CAP, f, fr, fp, fs, pkg.STAGE_PSS, max_cells_per_buf=pkg.MAX_PEAKS)
                kernels.add(S.last_xcorr_info()[0])
                n_repairs += S.last_frq_repairs()
                arr = [S.batch_readback(b, f.size) for b in range(len(ids))]
                full = S.search_batch(d.data_ptr(), fmt, len(ids), N_CAP, f, fr, fp, fs, pkg.STAGE_FULL, max_cells_per_buf=pkg.MAX_PEAKS)
                t_gpu += time.perf_counter() - t0
                assert not S.last_overflow
                for b, i in enumerate(ids):
                    out[i] = dict(frq=arr[b]["frq"], pow=arr[b]["pow"], zth=arr[b]["z_th1"], peaks=[cell_to_dict(c) for c in pk[b]],
                                  cells=[cell_to_dict(c) for c in full[b]])
    return out, n_repairs, t_gpu, sorted(kernels)


def compare_arrays(pkg, S, it, o):
    """The `channels` group: for every cell the oracle decoded, the GPU's stage entry points on the oracle's own inputs --
    extract_tfg on the oracle's post-FOE cell, tfoec on the oracle's grid, chan_est of every port on the oracle's compensated
    grid -- at the stage tests' tolerances (tests/test_gpu_cells.py).  -> (disagreements, cells compared, worst deviations)"""
    dis, n = [], 0
    worst = dict(tfg=0.0, tfg_comp=0.0, ce_tfg=0.0, np=0.0)
    x = it["iq"].astype(np.float64)
    cap = ((x[0::2] - 127.0) / 128.0) + 1j * ((x[1::2] - 127.0) / 128.0)
    fr, fp, fs = it["fc_req"], it["fc_prog"], it["fs_prog"]

    def dev(a, b):
        return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())

    for p in o["peaks"]:
        a = p.pop("arrays", None)
        if a is None:
            continue
        n += 1
        tag = f"{it['name']} peak ({p['peak']['n_id_2']}, {p['peak']['ind']})"
        c2 = pkg.new_cell(fc_requested=fr, fc_programmed=fp, **a["c2"])
        c3 = pkg.new_cell(fc_requested=fr, fc_programmed=fp, **a["c3"])
        tfg_g, ts_g = S.extract_tfg(c2, cap, fr, fp, fs)
        d = dev(tfg_g, a["tfg"])
        worst["tfg"] = max(worst["tfg"], d)
        if not (d <= 1e-10 and np.array_equal(ts_g, a["ts"])):
            dis.append(dict(buffer=tag, stage="extract_tfg", what=f"grid deviates by {d:.3e} of its largest element", margin=None, threshold="1e-10"))
        c3g, tfgc_g, _ = S.tfoec(c2, a["tfg"], a["ts"], fr, fp)
        d = dev(tfgc_g, a["tfgc"])
        worst["tfg_comp"] = max(worst["tfg_comp"], d)
        if not (d <= 1e-9 and abs(c3g.freq_superfine - a["c3"]["freq_superfine"]) < 1e-7):
            dis.append(dict(buffer=tag, stage="tfoec", what=f"tfg_comp deviates by {d:.3e}; freq_superfine {c3g.freq_superfine!r} vs {a['c3']['freq_superfine']!r}", margin=None, threshold="1e-9 / 1e-7 Hz"))
        for port, (ce_o, np_o) in enumerate(zip(a["ce"], a["np"])):
            ce_g, np_g = S.chan_est(c3, a["tfgc"], port)
            d, dn = dev(ce_g, ce_o), abs(np_g - np_o) / np_o
            worst["ce_tfg"], worst["np"] = max(worst["ce_tfg"], d), max(worst["np"], dn)
            if not (d <= 1e-9 and dn <= 1e-11):
                dis.append(dict(buffer=tag, stage="chan_est", what=f"port {port}: ce_tfg deviates by {d:.3e}, np by {dn:.3e}", margin=None, threshold="1e-9 / 1e-11"))
    return dis, n, worst


# ------------------------------------------------------------------------------------------------------ comparison
def compare(it, g, o):
    """-> (list of disagreements, counters)"""
    dis = []
    name = it["name"]
    bad = np.argwhere(g["frq"] != o["frq"])
    for t, i in bad:
        dis.append(dict(buffer=name, stage="xc_peak_freq", what=f"frq[{t},{i}] = {int(g['frq'][t, i])}, oracle {int(o['frq'][t, i])}",
                        margin=float(o["frq_margin"][t, i]), threshold="first maximum over the hypotheses (src/searcher.cpp:374)"))
    rel = np.abs(g["pow"] - o["pow"].astype(np.float64)) / o["pow"].astype(np.float64)
    if rel.max() > 1e-5:
        dis.append(dict(buffer=name, stage="xc_incoherent_collapsed_pow", what=f"max relative deviation {rel.max():.3e}", margin=None, threshold="1e-5"))
    zr = np.abs(g["zth"] - o["zth"]) / o["zth"]
    if zr.max() > 1e-10:
        dis.append(dict(buffer=name, stage="Z_th1", what=f"max relative deviation {zr.max():.3e}", margin=None, threshold="1e-10"))
    gp, op = g["peaks"], [p["peak"] for p in o["peaks"]]
    key = lambda p: (p["n_id_2"], p["ind"], p["freq"])
    if [key(p) for p in gp] != [key(p) for p in op]:
        zmin = min([abs(p["z_ratio"] - 1.0) for p in o["peaks"]] + [abs(o["z_rejected"] - 1.0)])
        dis.append(dict(buffer=name, stage="peak_search", what=f"peak lists differ: GPU {[key(p) for p in gp]} oracle {[key(p) for p in op]}",
                        margin=zmin, threshold="peak_pow >= Z_th1 (src/CellSearch.cpp:500-503, src/searcher.cpp:449); margin = smallest |maximum / Z_th1 - 1| over the "
                                               "oracle's accepted maxima and the first rejected one"))
    else:
        for a, b in zip(gp, op):
            if abs(a["pss_pow"] - b["pss_pow"]) > 1e-5 * b["pss_pow"]:
                dis.append(dict(buffer=name, stage="peak_search", what=f"pss_pow {a['pss_pow']!r} vs {b['pss_pow']!r}", margin=None, threshold="1e-5"))
    # the cells: the oracle's peaks that passed SSS and MIB, in peak order
    oc = [p for p in o["peaks"] if p["mib_found"]]
    gc = g["cells"]
    ident = lambda c: tuple(c[k] for k in CELL_INT)
    if [ident(c) for c in gc] != [ident(p["cell"]) for p in oc]:
        # which decision flipped?  match by (n_id_2, ind)
        gset = {(c["n_id_2"], c["ind"]): c for c in gc}
        for p in o["peaks"]:
            k2 = (p["peak"]["n_id_2"], p["peak"]["ind"])
            c = gset.pop(k2, None)
            if p["mib_found"] and c is None:
                dis.append(dict(buffer=name, stage="sss_detect/decode_mib", what=f"oracle decodes peak {k2} (cell {p['cell']['n_id_2'] + 3 * p['cell']['n_id_1']}), GPU drops it",
                                margin=p["sss_sigma"] - 3.0, threshold="3-sigma SSS test (src/searcher.cpp:751-758), then the PBCH CRC (:1628-1636); margin = oracle sigma - 3"))
            elif not p["mib_found"] and c is not None:
                dis.append(dict(buffer=name, stage="sss_detect/decode_mib", what=f"GPU decodes peak {k2} (cell {c['n_id_2'] + 3 * c['n_id_1']}), oracle drops it "
                                f"({'SSS' if not p['sss_found'] else 'MIB'})", margin=p["sss_sigma"] - 3.0,
                                threshold="3-sigma SSS test / PBCH CRC; margin = oracle sigma - 3"))
            elif p["mib_found"] and ident(c) != ident(p["cell"]):
                dis.append(dict(buffer=name, stage="identity", what=f"peak {k2}: GPU {ident(c)} oracle {ident(p['cell'])}", margin=p["sss_sigma"] - 3.0,
                                threshold="fields of the decoded cell"))
        for k2 in gset:
            dis.append(dict(buffer=name, stage="sss_detect/decode_mib", what=f"GPU reports a cell at {k2} the oracle has no peak for", margin=None, threshold=""))
    else:
        for a, p in zip(gc, oc):
            b = p["cell"]
            for fld, tol, rel_ in (("pss_pow", 1e-5, True), ("frame_start", 1e-6, False), ("freq_fine", 1e-4, False), ("freq_superfine", 1e-3, False)):
                d = abs(a[fld] - b[fld]) / (abs(b[fld]) if rel_ else 1.0)
                if not d <= tol:
                    dis.append(dict(buffer=name, stage="continuous", what=f"{fld}: {a[fld]!r} vs {b[fld]!r}", margin=None, threshold=str(tol)))
    return dis, dict(peaks=len(op), sss_found=sum(p["sss_found"] for p in o["peaks"]), cells=len(oc),
                     min_sss_margin=min([abs(p["sss_sigma"] - 3.0) for p in o["peaks"]], default=None),
                     min_z_ratio=min([abs(p["z_ratio"] - 1.0) for p in o["peaks"]] + [abs(o["z_rejected"] - 1.0)]),
                     frq_near_ties_1e_5=int(np.count_nonzero(o["frq_margin"] < 1e-5)), frq_near_ties_4e_6=int(np.count_nonzero(o["frq_margin"] < 4e-6)),
                     frq_near_ties_1e_6=int(np.count_nonzero(o["frq_margin"] < 1e-6)), frq_min_margin=float(o["frq_margin"].min()))


def run(groups=("synthetic", "bench", "dense"), limit=None, workers=None, out_path=None, quiet=False, dense_limit=None, seed_offset=0, input_fmt="u8", batch=128):
    t_start = time.perf_counter()
    workers = workers or max(1, min(len(os.sched_getaffinity(0)), 32))
    # the oracle's worker processes start BEFORE this process touches the GPU runtime (fork of a process with live HIP threads is unsafe)
    pool = mp.get_context("fork").Pool(workers)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    items = build_population(pkg, groups, limit, dense_limit, seed_offset, pool)
    t_built = time.perf_counter()
    res_async = pool.imap(oracle_job, items, chunksize=1)
    gpu, n_repairs, t_gpu, kernels = gpu_pass(pkg, items, batch=batch, input_fmt=input_fmt)
    # results are taken as they arrive: the stage arrays of the `channels` group (a few MB per decoded cell) are compared through
    # the GPU's stage entry points at once and dropped
    orc, arr_dis, arr_cells = [], [], 0
    arr_worst = dict(tfg=0.0, tfg_comp=0.0, ce_tfg=0.0, np=0.0)
    with pkg.Searcher(0) as S_arr:
        for it, o in zip(items, res_async):
            if it.get("arrays"):
                d_, n_, w_ = compare_arrays(pkg, S_arr, it, o)
                arr_dis += d_
                arr_cells += n_
                arr_worst = {k: max(arr_worst[k], w_[k]) for k in arr_worst}
            orc.append(o)
    pool.close()
    pool.join()
    t_done = time.perf_counter()
    all_dis, per_group = list(arr_dis), {}
    tot = dict(buffers=0, peaks=0, sss_found=0, cells=0, frq_positions=0, frq_near_ties_1e_5=0, frq_near_ties_4e_6=0, frq_near_ties_1e_6=0)
    min_sss, min_z, min_frq = None, None, None
    for it, g, o in zip(items, gpu, orc):
        dis, cnt = compare(it, g, o)
        all_dis += dis
        pg = per_group.setdefault(it["group"], dict(buffers=0, peaks=0, sss_found=0, cells=0, disagreements=0))
        pg["buffers"] += 1
        pg["disagreements"] += len(dis)
        for k in ("peaks", "sss_found", "cells"):
            pg[k] += cnt[k]
            tot[k] += cnt[k]
        tot["buffers"] += 1
        tot["frq_positions"] += 3 * 9600
        for k in ("frq_near_ties_1e_5", "frq_near_ties_4e_6", "frq_near_ties_1e_6"):
            tot[k] += cnt[k]
        if cnt["min_sss_margin"] is not None:
            min_sss = cnt["min_sss_margin"] if min_sss is None else min(min_sss, cnt["min_sss_margin"])
        if cnt["min_z_ratio"] is not None:
            min_z = cnt["min_z_ratio"] if min_z is None else min(min_z, cnt["min_z_ratio"])
        min_frq = cnt["frq_min_margin"] if min_frq is None else min(min_frq, cnt["frq_min_margin"])
    by_stage = {}
    for d in all_dis:
        by_stage[d["stage"]] = by_stage.get(d["stage"], 0) + 1
    report = dict(
        what="GPU chain (lcs_batch_enqueue / _collect / _readback) against oracle/lcs_oracle.c on the same capture buffers",
        groups=list(groups), seed_offset=seed_offset, input=input_fmt, correlation_kernels=kernels, buffers_per_batch=batch,
        totals=tot, per_group=per_group, disagreements=len(all_dis), disagreements_by_stage=by_stage,
        disagreement_rate_per_buffer=len(all_dis) / max(1, tot["buffers"]),
        gpu_frq_positions_repaired=n_repairs,
        stage_arrays=dict(cells_compared=arr_cells, disagreements=len(arr_dis), worst_relative_deviation=arr_worst,
                          tolerances=dict(tfg=1e-10, tfg_comp=1e-9, ce_tfg=1e-9, np=1e-11),
                          note="channels / highband groups: extract_tfg / tfoec / chan_est (every port) of every cell the oracle decoded, GPU stage entry points on the oracle's inputs"),
        closest_calls=dict(sss_abs_sigma_minus_3_min=min_sss, z_th1_abs_ratio_minus_1_min=min_z, frq_best_vs_second_min_rel=min_frq),
        details=all_dis[:200],
        seconds=dict(population=t_built - t_start, gpu_and_oracle=t_done - t_built, gpu_calls=t_gpu,
                     oracle_cpu_s=float(sum(o["seconds"] for o in orc)), oracle_workers=workers))
    if out_path:
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        with open(out_path, "w") as fh:
            json.dump(report, fh, indent=1)
    if not quiet:
        print(json.dumps({k: v for k, v in report.items() if k != "details"}))
        for d in all_dis[:20]:
            print("DISAGREEMENT", json.dumps(d))
    return report


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", default="synthetic,bench,dense", help="any of synthetic, bench, dense, channels, highband")
    ap.add_argument("--limit", type=int, default=None, help="quick runs: at most this many buffers per group (synthetic: whole scenes of 8; bench: a quarter of it from each of the four batches)")
    ap.add_argument("--dense-limit", type=int, default=None)
    ap.add_argument("--workers", type=int, default=None)
    ap.add_argument("--seed-offset", type=int, default=0, help="other synthetic scenes and noise realisations (the bench's and the dense band's buffers are fixed)")
    ap.add_argument("--input", default="u8", choices=("u8", "c64"), help="c64: the same samples as complex<float> batches (the fp16 three-product kernel)")
    ap.add_argument("--batch", type=int, default=128, help="buffers per lcs_batch_enqueue call (any number: 1, an odd one, more than 128)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_population.json"))
    a = ap.parse_args()
    r = run(tuple(a.groups.split(",")), a.limit, a.workers, a.out, dense_limit=a.dense_limit, seed_offset=a.seed_offset, input_fmt=a.input, batch=a.batch)
    sys.exit(0 if r["disagreements"] == 0 else 1)
