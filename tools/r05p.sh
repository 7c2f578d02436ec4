set -u
O=gpurun_out/r05p; mkdir -p $O
(timeout 1200 python -m pytest tests/test_sweep_dist.py tests/test_gpu_rccl.py tests/test_gpu_pss.py -m gpu -x -q 2>&1 | tail -15) > $O/pytest.log
