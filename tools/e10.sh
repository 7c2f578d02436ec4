cd $GRAFT_REPO_ROOT
S="--steps 8 --warmup 2"
python tools/ab.py e10 "$S --lib build_exp/liblcs_base.so" "$S --lib build_exp/liblcs_pbnoprio.so" "$S --lib build_exp/liblcs_noprio.so" "$S --lib build_exp/liblcs_xcprio3.so" "$S --lib build_exp/liblcs_xcprio3noprio.so" "$S --lib build_exp/liblcs_base.so" "$S --lib build_exp/liblcs_pbnoprio.so" "$S --lib build_exp/liblcs_noprio.so" "$S --lib build_exp/liblcs_xcprio3.so" "$S --lib build_exp/liblcs_xcprio3noprio.so"
