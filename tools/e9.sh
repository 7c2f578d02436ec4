cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/e9_pytest.log
S="--steps 8 --warmup 2"
python tools/ab.py e9 "$S --lib build_exp/liblcs_base.so" "$S" "$S --lib build_exp/liblcs_base.so" "$S" "$S --stage pss --lib build_exp/liblcs_base.so" "$S --stage pss"
