#!/bin/bash
# Kernel-trace statistics of the bench workload, one context (isolated kernel times) and the default pipeline:
#   bash profiles/kstats.sh <tag>     -> gpurun_out/<tag>/{p1,p3}/.../*_kernel_stats.csv
TAG=${1:-kstats}
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p1 -- python $R/bench.py --steps 2 --warmup 1 --batches-per-step 8 --pipeline 1 --no-cpu-baseline > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p3 -- python $R/bench.py --steps 2 --warmup 1 --batches-per-step 8 --no-cpu-baseline > $OUT/p3.log 2>&1
