set -u
O=gpurun_out/r05n; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_cells.py tests/test_gpu_configs.py tests/test_gpu_stream.py tests/test_cli.py -m gpu -x -q 2>&1 | tail -4) > $O/pytest.log
python tools/ab.py r05n '--steps 8 --warmup 3 --lib build_exp/liblcs_r04.so' '--steps 8 --warmup 3' '--steps 8 --warmup 3 --lib build_exp/liblcs_r04.so' '--steps 8 --warmup 3'
cd /tmp; export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py"
SHORT="--steps 2 --warmup 1 --batches-per-step 8 --no-cpu-baseline --no-dense --no-power-probe --synth-cache /tmp/synth"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_default -- $B $SHORT > $GRAFT_REPO_ROOT/$O/stats_default.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_dense -- $B $SHORT --dense-main > $GRAFT_REPO_ROOT/$O/stats_dense.log 2>&1
for lib in "--lib build_exp/liblcs_r04.so" ""; do
  timeout 120 $B --stage stream --steps 400 --warmup 20 $lib 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stream [$lib]', round(j['value']), j['ms_per_step'], j['config']['gpu_ms_per_buffer'])" >> $GRAFT_REPO_ROOT/$O/stream_single.txt
  timeout 120 $B --stage single --steps 200 --warmup 20 --no-cpu-baseline $lib 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single [$lib]', round(j['value']), j['ms_per_step'])" >> $GRAFT_REPO_ROOT/$O/stream_single.txt
done
cd $GRAFT_REPO_ROOT/$O; for d in stats_default stats_dense; do f=$(find $d -name '*kernel_stats.csv' | head -1); cp "$f" $d.csv; rm -rf $d; done
