#!/bin/bash
# PMC counters of the correlation kernel alone (one context, PSS stage): where its wave cycles go.
#   bash profiles/pmc_xcorr.sh <tag>   -> gpurun_out/<tag>/pmc_*/...counter_collection.csv
TAG=${1:-pmcx}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --stage pss --pipeline 1 --steps 1 --warmup 0 --batches-per-step 4 --no-cpu-baseline --no-power-probe"
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$i -- $B > $OUT/pmc_$i.log 2>&1
done
python3 - <<PY
import csv, glob
acc = {}
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    per = {}
    for r in csv.DictReader(open(f)):
        if "xcorr_i8" not in r["Kernel_Name"]: continue
        per.setdefault((r["Counter_Name"], r["Dispatch_Id"]), 0.0)
        per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    by = {}
    for (c, d), v in per.items(): by.setdefault(c, []).append(v)
    for c, v in by.items(): acc[c] = sum(v) / len(v)
for c in sorted(acc): print(f"{c:28s} {acc[c]:16.0f}")
PY
