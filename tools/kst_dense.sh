cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 2 --warmup 1 --batches-per-step 8 --no-cpu-baseline --no-dense --no-power-probe --dense-main"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst_dense_p1 -- $B --pipeline 1 > $R/gpurun_out/kst_dense_p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst_dense_p2 -- $B > $R/gpurun_out/kst_dense_p2.log 2>&1
find $R/gpurun_out/kst_dense_p1 $R/gpurun_out/kst_dense_p2 -name "*kernel_stats.csv" | head
# keep only the stats csv (traces are large)
find $R/gpurun_out/kst_dense_p1 $R/gpurun_out/kst_dense_p2 -type f ! -name "*kernel_stats.csv" -delete
