cd $GRAFT_REPO_ROOT
S="--steps 8 --warmup 2"
python tools/ab.py e17 "$S --batch 128 --batches-per-step 16"
sleep 15
python tools/ab.py e17 "$S --batch 128 --batches-per-step 16"
sleep 15
python tools/ab.py e17 "$S --batch 128 --batches-per-step 16" "$S --batch 128 --batches-per-step 16"
sleep 15
python tools/ab.py e17 "$S --batch 160 --batches-per-step 13" "$S --batch 160 --batches-per-step 13"
sleep 15
python tools/ab.py e17 "$S" "$S"
