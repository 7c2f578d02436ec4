cd $GRAFT_REPO_ROOT
S="--steps 6 --warmup 2"
python tools/ab.py e1 "$S" "$S --pipeline 4" "$S --pipeline 5" "$S --pipeline 6" \
 "$S --lib build_exp/liblcs_cap256.so" "$S --lib build_exp/liblcs_cap256.so --pipeline 4" "$S --lib build_exp/liblcs_cap256.so --pipeline 5" "$S --lib build_exp/liblcs_cap256.so --pipeline 6" \
 "$S --lib build_exp/liblcs_cap512.so" "$S --lib build_exp/liblcs_cap512.so --pipeline 5" \
 "$S --pipeline 1" \
 "$S --batch 512 --batches-per-step 4 --pipeline 1" "$S --batch 512 --batches-per-step 4 --pipeline 2" "$S --batch 512 --batches-per-step 4 --pipeline 3" \
 "$S --batch 256 --batches-per-step 8 --pipeline 1" "$S --batch 256 --batches-per-step 8 --pipeline 3" "$S --batch 128 --batches-per-step 16 --pipeline 3" \
 "$S --batch 1024 --batches-per-step 2 --pipeline 1"
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/e1_stats_b512 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --synth-cache /tmp/synth --steps 2 --warmup 1 --batch 512 --batches-per-step 2 --pipeline 1 > $GRAFT_REPO_ROOT/gpurun_out/e1_stats_b512.log 2>&1
