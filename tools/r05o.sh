set -u
O=gpurun_out/r05o; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_cells.py tests/test_gpu_configs.py tests/test_gpu_stream.py tests/test_gpu_pss.py tests/test_cli.py tests/test_gpu_properties.py -m gpu -x -q 2>&1 | tail -4) > $O/pytest.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > $O/smoke.log
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py"
for lib in "--lib build_exp/liblcs_r04.so" "" "--lib build_exp/liblcs_r04.so" ""; do
  timeout 120 $B --stage single --steps 200 --warmup 20 --no-cpu-baseline $lib 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single [$lib]', round(j['value']), j['ms_per_step'])" >> $GRAFT_REPO_ROOT/$O/single.txt
done
timeout 120 $B --stage single --steps 200 --warmup 20 > $GRAFT_REPO_ROOT/$O/bench_single_n1.json 2>/dev/null
