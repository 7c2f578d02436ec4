set -u
O=gpurun_out/r05g; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py"
SHORT="--steps 2 --warmup 1 --batches-per-step 8 --no-cpu-baseline --no-dense --no-power-probe --synth-cache /tmp/synth"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_default -- $B $SHORT > $GRAFT_REPO_ROOT/$O/stats_default.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_default_r04 -- $B $SHORT --lib build_exp/liblcs_r04.so > $GRAFT_REPO_ROOT/$O/stats_default_r04.log 2>&1
cd $GRAFT_REPO_ROOT/$O; for d in stats_default stats_default_r04; do f=$(find $d -name '*kernel_stats.csv' | head -1); cp "$f" $d.csv; rm -rf $d; done
