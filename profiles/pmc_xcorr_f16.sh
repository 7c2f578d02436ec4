#!/bin/bash
# PMC counters of k_xcorr_f16x3 (complex<float> batches, one context, 64 buffers per launch), one counter group per rocprofv3 pass
TAG=pmcf; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 0 --batch 64 --batches-per-step 4 --pipeline 1 --no-cpu-baseline --no-dense --no-power-probe --input c64"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$i -- $B > $OUT/pmc_$i.log 2>&1
done
python3 - <<PY
import csv, glob
acc = {}
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    per = {}
    for r in csv.DictReader(open(f)):
        if "xcorr_f16" not in r["Kernel_Name"]: continue
        per.setdefault((r["Counter_Name"], r["Dispatch_Id"]), 0.0)
        per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    by = {}
    for (c, d), v in per.items(): by.setdefault(c, []).append(v)
    for c, v in by.items(): acc[c] = sum(v) / len(v)
for c in sorted(acc): print(f"{c:28s} {acc[c]:16.0f}")
PY
