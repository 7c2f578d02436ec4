cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/e12_pytest.log
cd /tmp
python $GRAFT_REPO_ROOT/bench.py --stage track --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/e12_track.json 2>/dev/null
cd $GRAFT_REPO_ROOT
S="--steps 8 --warmup 2"
python tools/ab.py e12 "$S" "$S"
