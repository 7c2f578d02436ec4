#!/bin/bash
# Developer experiment (GPU box): in-chain cost of the per-cell chain, kernel by kernel -- the chain cut short after stage L
# (profiles/r06/experiments/ab_chain_truncation.patch; libraries built into build_exp/ by hand).  Two rounds, default and dense band.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for L in 0 1 2 3 4 5; do
    python tools/ab.py trunc "--steps 6 --warmup 2 --no-dense --lib build_exp/liblcs_trunc$L.so" "--steps 3 --warmup 2 --no-dense --dense-main --lib build_exp/liblcs_trunc$L.so"
  done
  python tools/ab.py trunc "--steps 6 --warmup 2 --no-dense" "--steps 3 --warmup 2 --no-dense --dense-main"
done
