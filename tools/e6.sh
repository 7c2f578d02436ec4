cd $GRAFT_REPO_ROOT
S="--steps 8 --warmup 2"
python tools/ab.py e6b "$S" "$S --lib build_exp/liblcs_notime.so --no-xc-timing" "$S --lib build_exp/liblcs_merged.so" "$S --lib build_exp/liblcs_mergednotime.so --no-xc-timing" "$S" "$S --lib build_exp/liblcs_mergednotime.so --no-xc-timing" "$S --lib build_exp/liblcs_mergednotime.so --no-xc-timing --stage pss" "$S --lib build_exp/liblcs_mergednotime.so --no-xc-timing --pipeline 4"
