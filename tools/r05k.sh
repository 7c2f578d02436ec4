set -u
O=gpurun_out/r05k; mkdir -p $O
(timeout 900 python -m pytest tests/test_tracker.py tests/test_cli.py -m gpu -x -q 2>&1 | tail -8) > $O/pytest.log
(cd /tmp && timeout 300 python $GRAFT_REPO_ROOT/bench.py --stage track --steps 60 --warmup 6 > $GRAFT_REPO_ROOT/$O/bench_track.json 2> $GRAFT_REPO_ROOT/$O/bench_track.err)
