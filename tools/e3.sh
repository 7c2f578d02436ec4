cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/e3_pytest.log
S="--steps 6 --warmup 2"
python tools/ab.py e3 "$S" "$S --pipeline 1" "$S --batch 512 --batches-per-step 4 --pipeline 1"
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/e3_stats_p1 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --synth-cache /tmp/synth --steps 2 --warmup 1 --batches-per-step 8 --pipeline 1 > $GRAFT_REPO_ROOT/gpurun_out/e3_stats_p1.log 2>&1
