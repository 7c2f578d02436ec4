#!/bin/bash
# Developer A/B (GPU box): the in-tree library against build_exp/liblcs_$1.so, default and dense band, alternating, $2 rounds.
cd "$(dirname "$0")/.."
REF=$1; N=${2:-2}; TAG=${3:-pair}
for rep in $(seq 1 $N); do
  python tools/ab.py $TAG "--steps 6 --warmup 2 --no-dense --lib build_exp/liblcs_$REF.so" "--steps 6 --warmup 2 --no-dense" \
                          "--steps 3 --warmup 2 --no-dense --dense-main --lib build_exp/liblcs_$REF.so" "--steps 3 --warmup 2 --no-dense --dense-main"
done
