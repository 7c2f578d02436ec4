cd $GRAFT_REPO_ROOT
S="--steps 8 --warmup 2"
python tools/ab.py e5 "$S" "$S --lib build_exp/liblcs_sharedxc.so" "$S --lib build_exp/liblcs_skpbch.so" "$S --lib build_exp/liblcs_skce.so" "$S --lib build_exp/liblcs_sktfg.so" "$S --lib build_exp/liblcs_sktfoec.so" "$S --lib build_exp/liblcs_sksss.so" "$S --stage pss" "$S" "$S --lib build_exp/liblcs_sharedxc.so"
