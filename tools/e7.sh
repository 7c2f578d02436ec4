cd $GRAFT_REPO_ROOT
S="--steps 8 --warmup 2"
python tools/ab.py e7 "$S --lib build_exp/liblcs_base.so" "$S --lib build_exp/liblcs_ntsingle.so" "$S --lib build_exp/liblcs_ntboth.so" "$S --lib build_exp/liblcs_merged.so" "$S --lib build_exp/liblcs_evdev.so" "$S --lib build_exp/liblcs_base.so" "$S --lib build_exp/liblcs_ntsingle.so" "$S --lib build_exp/liblcs_ntboth.so" "$S --lib build_exp/liblcs_merged.so" "$S --lib build_exp/liblcs_evdev.so"
