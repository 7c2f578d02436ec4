set -u
O=gpurun_out/r05b; mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30) > $O/pytest.log
(timeout 1500 python tools/parity_population.py --out $O/parity_population.json 2>&1 | tail -40) > $O/population.log
