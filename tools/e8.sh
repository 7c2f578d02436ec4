cd $GRAFT_REPO_ROOT
S="--steps 8 --warmup 2"
python tools/ab.py e8b "$S" "$S --lib build_exp/liblcs_skingest.so" "$S --lib build_exp/liblcs_skprep.so" "$S --lib build_exp/liblcs_skfill.so" "$S --lib build_exp/liblcs_sksp.so" "$S --lib build_exp/liblcs_skcollapse.so" "$S --lib build_exp/liblcs_skallpre.so" "$S"
