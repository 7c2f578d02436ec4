set -u
O=gpurun_out/r05j; mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > $O/pytest.log
(timeout 1500 python tools/parity_population.py --out $O/parity_population.json 2>&1 | tail -30) > $O/population.log
python tools/ab.py r05j '--steps 8 --warmup 3' '--steps 8 --warmup 3 --lib build_exp/liblcs_pkepi.so' '--steps 8 --warmup 3' '--steps 8 --warmup 3 --lib build_exp/liblcs_pkepi.so' '--steps 8 --warmup 3 --lib build_exp/liblcs_r04.so'
