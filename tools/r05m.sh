set -u
O=gpurun_out/r05m; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_cells.py tests/test_gpu_configs.py tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -4) > $O/pytest.log
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py"
for lib in "--lib build_exp/liblcs_r04.so" "" "--lib build_exp/liblcs_r04.so" ""; do
  timeout 120 $B --stage stream --steps 400 --warmup 20 $lib 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stream [$lib]', round(j['value']), j['ms_per_step'], j['config']['gpu_ms_per_buffer'])" >> $GRAFT_REPO_ROOT/$O/stream_single.txt
done
for lib in "--lib build_exp/liblcs_r04.so" ""; do
  timeout 120 $B --stage single --steps 200 --warmup 20 --no-cpu-baseline $lib 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single [$lib]', round(j['value']), j['ms_per_step'])" >> $GRAFT_REPO_ROOT/$O/stream_single.txt
done
cd $GRAFT_REPO_ROOT
python tools/ab.py r05m '--steps 8 --warmup 3 --lib build_exp/liblcs_r04.so' '--steps 8 --warmup 3'
python tools/phase_ts.py build_exp/liblcs_phts.so > $O/phase_ts.txt 2>&1
