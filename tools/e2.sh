cd $GRAFT_REPO_ROOT
S="--steps 6 --warmup 2"
python tools/ab.py e2 "$S --lib build_exp/liblcs_cap256.so" "$S --lib build_exp/liblcs_cap256.so --pipeline 5" "$S --lib build_exp/liblcs_cap512.so" "$S --lib build_exp/liblcs_cap512.so --pipeline 5"
# clocks / power under sustained load: sample rocm-smi while long runs are in flight
(for i in $(seq 1 400); do echo "t=$(date +%s.%N) $(rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -n +2 | tr '\n' ' ')"; sleep 0.05; done) > gpurun_out/e2_smi.txt &
SMI=$!
sleep 2
echo "mark B1024P1 $(date +%s.%N)" >> gpurun_out/e2_marks.txt
python tools/ab.py e2 "--steps 30 --warmup 2 --batch 1024 --batches-per-step 2 --pipeline 1"
echo "mark B64P1 $(date +%s.%N)" >> gpurun_out/e2_marks.txt
python tools/ab.py e2 "--steps 30 --warmup 2 --pipeline 1"
echo "mark B64P3 $(date +%s.%N)" >> gpurun_out/e2_marks.txt
python tools/ab.py e2 "--steps 30 --warmup 2"
echo "mark end $(date +%s.%N)" >> gpurun_out/e2_marks.txt
kill $SMI
