cd $GRAFT_REPO_ROOT
S="--steps 8 --warmup 2"
python tools/ab.py e11 "$S --lib build_exp/liblcs_base.so" "$S --lib build_exp/liblcs_ntfill.so" "$S --lib build_exp/liblcs_base.so" "$S --lib build_exp/liblcs_ntfill.so" "$S --stage pss --lib build_exp/liblcs_base.so" "$S --stage pss --lib build_exp/liblcs_ntfill.so"
