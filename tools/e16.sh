cd $GRAFT_REPO_ROOT
S="--steps 8 --warmup 2"
for q in 4 6 8 12 16; do
  export GPU_MAX_HW_QUEUES=$q
  python tools/ab.py e16 "$S --batch 64 --batches-per-step 32 --distinct $((4))" "$S --batch 128 --batches-per-step 16" "$S --batch 128 --batches-per-step 16"
done
