#!/bin/bash
# Collect the round's profiling artifacts on a GPU box (run from the repository root through gpurun):
#   bash profiles/collect.sh r01
# Kernel-trace statistics and PMC counters are taken in SEPARATE rocprofv3 runs (one counter group per pass,
# --kernel-trace only), as MI355X_MICROARCH.md prescribes.  Outputs land in gpurun_out/<tag>/; copy the
# summaries into profiles/<tag>/ afterwards (profiles/summarize.py does that).
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
cd /tmp; export TMPDIR=/tmp
B="python $REPO/bench.py"
timeout 300 $B --steps 20 --warmup 3 > "$OUT/bench_full_n1.json" 2> "$OUT/bench_full_n1.err"
timeout 300 $B --steps 20 --warmup 3 --stage pss > "$OUT/bench_pss_n1.json" 2> "$OUT/bench_pss_n1.err"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_full_p1" -- $B --steps 5 --warmup 1 --pipeline 1 --no-cpu-baseline > "$OUT/stats_full_p1.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_full_default" -- $B --steps 8 --warmup 2 --no-cpu-baseline > "$OUT/stats_full_default.log" 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc_$i" -- $B --steps 3 --warmup 1 --pipeline 1 --no-cpu-baseline > "$OUT/pmc_$i.log" 2>&1
done
echo collected > "$OUT/done"
