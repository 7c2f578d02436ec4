#!/bin/bash
# Round 6, after profiles/collect.sh: the parity populations added late in the round and a longer tracker line.
#   bash profiles/collect_extra.sh r06        (from the repository root, through gpurun)
# Outputs land in gpurun_out/<tag>/; profiles/summarize.py copies the JSON files to profiles/<tag>/.
set -u
TAG=${1:-r06}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
P="python tools/parity_population.py"
# the tests written after the collection
# wide grids (n_f = 61 .. 169) on fading channels, two draws
timeout 900 $P --groups highband --out "$OUT/parity_population_highband.json" > "$OUT/parity_population_highband.log" 2>&1
timeout 900 $P --groups highband --seed-offset 1 --out "$OUT/parity_population_highband_seed1.json" > "$OUT/parity_population_highband_seed1.log" 2>&1
# the same synthetic + channels populations as complex<float> batches: the fp16 three-product kernel and the float source of every later stage
timeout 1500 $P --groups synthetic,channels --input c64 --out "$OUT/parity_population_c64.json" > "$OUT/parity_population_c64.log" 2>&1
# two more draws of the channels group
timeout 900 $P --groups channels --seed-offset 1 --out "$OUT/parity_population_channels_seed1.json" > "$OUT/parity_population_channels_seed1.log" 2>&1
timeout 900 $P --groups channels --seed-offset 2 --out "$OUT/parity_population_channels_seed2.json" > "$OUT/parity_population_channels_seed2.log" 2>&1
echo collected > "$OUT/done_extra"
