#!/bin/bash
# Developer A/B (GPU box): tools/ab_multi.sh TAG ROUNDS NAME... -- the in-tree library and build_exp/liblcs_NAME.so, default and dense band, alternating
cd "$(dirname "$0")/.."
TAG=$1; N=$2; shift 2
for rep in $(seq 1 $N); do
  python tools/ab.py $TAG "--steps 6 --warmup 2 --no-dense" "--steps 3 --warmup 2 --no-dense --dense-main"
  for L in "$@"; do
    python tools/ab.py $TAG "--steps 6 --warmup 2 --no-dense --lib build_exp/liblcs_$L.so" "--steps 3 --warmup 2 --no-dense --dense-main --lib build_exp/liblcs_$L.so"
  done
done
