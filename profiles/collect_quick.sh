#!/bin/bash
# Trimmed, priority-ordered version of collect.sh for a short GPU slot: GPU tests first, then the PMC passes for
# the HBM traffic, the headline bench line, the kernel-trace statistics, the remaining PMC groups, and -- if
# time is left -- the PSS-only line and the one-context statistics.  Usage: bash profiles/collect_quick.sh r01
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
(cd "$REPO" && timeout 150 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > "$OUT/pytest_gpu.log"
cd /tmp; export TMPDIR=/tmp
B="python $REPO/bench.py"
pmc() { timeout 60 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d "$OUT/pmc_$1" -- $B --steps 3 --warmup 1 --pipeline 1 --no-cpu-baseline > "$OUT/pmc_$1.log" 2>&1; }
pmc 1 "FETCH_SIZE"
pmc 2 "WRITE_SIZE"
timeout 120 $B --steps 20 --warmup 3 > "$OUT/bench_full_n1.json" 2> "$OUT/bench_full_n1.err"
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_full_default" -- $B --steps 8 --warmup 2 --no-cpu-baseline > "$OUT/stats_full_default.log" 2>&1
pmc 3 "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
pmc 5 "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
timeout 60 $B --steps 20 --warmup 3 --stage pss --no-cpu-baseline > "$OUT/bench_pss_n1.json" 2> "$OUT/bench_pss_n1.err"
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_full_p1" -- $B --steps 5 --warmup 1 --pipeline 1 --no-cpu-baseline > "$OUT/stats_full_p1.log" 2>&1
echo collected > "$OUT/done"
