#!/usr/bin/env python3
"""Reduce gpurun_out/<tag>/ (written by profiles/collect.sh) to the small files kept under profiles/<tag>/.

    python profiles/summarize.py r01
"""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)

for name in ("parity_population.json", "parity_population_channels.json", "parity_population_highband.json", "parity_population_highband_seed1.json", "parity_population_c64.json", "parity_population_channels_seed1.json", "parity_population_channels_seed2.json", "sweep_715_768_n1.json", "cli_time.txt", "cli_sweep_table.txt"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, name))
for name in ("bench_full_n1.json", "bench_forced_dist_n1.json", "bench_full_n1_input_host.json", "bench_single_n1.json", "bench_pss_n1.json", "bench_stream_n1.json", "bench_track_n1.json", "bench_full_n1_c64_f16_kernel.json", "bench_full_n1_fc2600MHz_ppm120_nf125.json", "bench_full_n1_c64_probe_int8_route.json", "bench_full_n1_steps200.json", "pytest_gpu.log", "pytest_gpu_new_tests.log"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        lines = [l for l in open(p).read().splitlines() if l.startswith("{")] if name.endswith(".json") else open(p).read().splitlines()[-6:]
        if lines:
            open(os.path.join(dst, name), "w").write((lines[-1] if name.endswith(".json") else "\n".join(lines)) + "\n")

for d, out in (("stats_full_p1", "kernel_stats_full_chain_b64_nf31_pipeline1.csv"),
               ("stats_full_default", "kernel_stats_full_chain_b128_nf31_default_command.csv"),
               ("stats_dense_default", "kernel_stats_dense_band_b128_nf31_default_command.csv"),
               ("stats_c64_p1", "kernel_stats_full_chain_b64_nf31_c64_pipeline1.csv"),
               ("stats_track", "kernel_stats_tracker_block_64cells_980sym.csv"),
               ("stats_stream", "kernel_stats_streaming_mode_nf1.csv"),
               ("stats_cut", "kernel_stats_track_cut_64cells_980sym.csv")):
    f = sorted(glob.glob(os.path.join(src, d, "*", "*_kernel_stats.csv")), key=os.path.getmtime)     # newest collection wins
    if f and d == "stats_track":
        # two traced processes: bench.py itself (the stream form, the Python loop) and host/TrackBench, whose C++ loop is the timed
        # one -- the file with the most tracker launches
        def n_trk(path):
            return sum(int(r["Calls"]) for r in csv.DictReader(open(path)) if r["Name"].startswith("k_trk_ce"))
        f = sorted(f, key=n_trk)
    if f:
        shutil.copy(f[-1], os.path.join(dst, out))
f = sorted(glob.glob(os.path.join(src, "stats_full_p1", "*", "*_agent_info.csv")), key=os.path.getmtime)
if f:
    shutil.copy(f[-1], os.path.join(dst, "agent_info.csv"))

# PMC: average every counter per kernel over its dispatches
kern = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
        acc = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc.setdefault((k, r["Counter_Name"]), {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[(k, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for (k, c), per in acc.items():
            kern.setdefault(k, {})[c] = sum(per.values()) / len(per)
dom = next((k for k in kern if k.startswith("k_xcorr_i8x3")), None) or next((k for k in kern if k.startswith("k_xcorr_mfma_blk")), None)
sys.path.insert(0, root)
import hashlib
def kernel_source_sha():          # the same digest bench.py computes: traffic is only reported for exactly these sources
    h = hashlib.sha256()
    for name in ("pss_xcorr_i8.hip", "pss_xcorr_f16.hip", "pss_xcorr.hip", "lcs_internal.h"):
        h.update(open(os.path.join(root, "lte-cell-scanner_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]
summary = {
    "source": "profiles/collect.sh: rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python bench.py --steps 1 "
              "--warmup 0 --batch 64 --batches-per-step 4 --pipeline 1 --no-cpu-baseline --no-dense, one pass per counter group; values are per-launch averages "
              "(64 buffers per launch, n_f = 31); kernel_source_sha = sha256 of the correlation kernel sources the counters were taken from",
    "corrections": "FETCH_SIZE is reported in KiB and, on gfx950, as half of the bytes fetched; WRITE_SIZE in KiB is exact "
                   "(MI355X_MICROARCH.md, HBM / rocprofv3 section).  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
    "kernels": kern,
    "kernel_source_sha": kernel_source_sha(),
    # LDS bank-conflict cycles per LDS instruction of the kernels that transform windows in LDS (round-3 review: 3.2 / 2.4 / 1.9)
    "lds_conflict_cycles_per_lds_instruction": {k: v["SQ_LDS_BANK_CONFLICT"] / v["SQ_INSTS_LDS"] for k, v in kern.items()
                                                if v.get("SQ_INSTS_LDS") and "SQ_LDS_BANK_CONFLICT" in v and v["SQ_INSTS_LDS"] > 1000},
}
if dom:
    k = kern[dom]
    summary["dominant_kernel"] = dom          # bench.py looks the traffic up under this key
    d = {}
    if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
        d["hbm_bytes_per_launch"] = (2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in k and "GRBM_GUI_ACTIVE" in k:
        # busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs: per-SIMD busy / per-XCD active
        d["mfma_pipe_busy_frac"] = (k["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (k["GRBM_GUI_ACTIVE"] / 8.0)
    if "TCC_HIT_sum" in k and "TCC_MISS_sum" in k:
        d["l2_hit_rate"] = k["TCC_HIT_sum"] / max(1.0, k["TCC_HIT_sum"] + k["TCC_MISS_sum"])
    if "SQ_LDS_BANK_CONFLICT" in k:
        d["lds_bank_conflict_cycles"] = k["SQ_LDS_BANK_CONFLICT"]
    summary["derived"] = d
    # fields bench.py reads for roofline.traffic (per buffer, at the n_f / batch the counters were taken at)
    k["n_f"], k["buffers_per_launch"] = 31, 64
    if "hbm_bytes_per_launch" in d:
        k["hbm_bytes_per_launch"] = d["hbm_bytes_per_launch"]
        k["hbm_bytes_per_buffer"] = d["hbm_bytes_per_launch"] / 64
# the fp16 kernel's traffic (complex<float> batches): its own FETCH_SIZE / WRITE_SIZE passes (pmcf_*), so that the --input c64 line
# carries roofline.traffic as well
kf = {}
for d in sorted(glob.glob(os.path.join(src, "pmcf_*"))):
    for f in sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
        acc = {}
        for r in csv.DictReader(open(f)):
            if "k_xcorr_f16x3" not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for cname, per in acc.items():
            kf[cname] = sum(per.values()) / len(per)
if "FETCH_SIZE" in kf and "WRITE_SIZE" in kf:
    kf["n_f"], kf["buffers_per_launch"] = 31, 64
    kf["hbm_bytes_per_launch"] = (2 * kf["FETCH_SIZE"] + kf["WRITE_SIZE"]) * 1024
    kf["hbm_bytes_per_buffer"] = kf["hbm_bytes_per_launch"] / 64
    summary["kernels"]["k_xcorr_f16x3 (complex<float> batches, own passes)"] = kf
json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print("wrote", dst, "kernels:", len(kern))
