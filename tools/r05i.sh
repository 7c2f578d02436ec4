set -u
O=gpurun_out/r05i; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_cells.py tests/test_gpu_configs.py tests/test_gpu_stream.py tests/test_gpu_pss.py tests/test_cli.py tests/test_gpu_properties.py -m gpu -x -q 2>&1 | tail -5) > $O/pytest.log
python tools/ab.py r05i '--steps 8 --warmup 3 --lib build_exp/liblcs_r04.so' '--steps 8 --warmup 3' '--steps 8 --warmup 3 --lib build_exp/liblcs_r04.so' '--steps 8 --warmup 3'
cd /tmp; export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py"
SHORT="--steps 2 --warmup 1 --batches-per-step 8 --no-cpu-baseline --no-dense --no-power-probe --synth-cache /tmp/synth"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_default -- $B $SHORT > $GRAFT_REPO_ROOT/$O/stats_default.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_dense -- $B $SHORT --dense-main > $GRAFT_REPO_ROOT/$O/stats_dense.log 2>&1
cd $GRAFT_REPO_ROOT/$O; for d in stats_default stats_dense; do f=$(find $d -name '*kernel_stats.csv' | head -1); cp "$f" $d.csv; rm -rf $d; done
