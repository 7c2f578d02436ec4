cd $GRAFT_REPO_ROOT
S="--steps 8 --warmup 2"
python tools/ab.py e15 "$S" "$S --batch 96 --batches-per-step 21" "$S --batch 128 --batches-per-step 16" "$S --batch 192 --batches-per-step 11" "$S --batch 128 --batches-per-step 16 --pipeline 4" "$S --batch 128 --batches-per-step 16 --pipeline 2" "$S" "$S --batch 128 --batches-per-step 16" "$S --batch 160 --batches-per-step 13"
