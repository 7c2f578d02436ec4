cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/e4_trace -- python $R/bench.py --no-cpu-baseline --synth-cache /tmp/synth --steps 3 --warmup 1 --batches-per-step 16 > $R/gpurun_out/e4_trace.log 2>&1
python $R/tools/trace_overlap.py $R/gpurun_out/e4_trace/*/*_kernel_trace.csv > $R/gpurun_out/e4_overlap.txt 2>&1
rm -f $R/gpurun_out/e4_trace/*/*_kernel_trace.csv
