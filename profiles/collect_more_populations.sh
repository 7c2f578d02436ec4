#!/bin/bash
# Round 6: further draws of the parity populations on the final library (bash profiles/collect_more_populations.sh r06, through gpurun)
set -u
TAG=${1:-r06}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
P="python tools/parity_population.py"
for s in 4 5; do timeout 900 $P --groups synthetic --seed-offset $s --out "$OUT/parity_population_synthetic_seed$s.json" > "$OUT/pp_syn$s.log" 2>&1; done
for s in 3 4; do timeout 900 $P --groups channels --seed-offset $s --out "$OUT/parity_population_channels_seed$s.json" > "$OUT/pp_ch$s.log" 2>&1; done
for s in 2 3 4; do timeout 900 $P --groups highband --seed-offset $s --out "$OUT/parity_population_highband_seed$s.json" > "$OUT/pp_hb$s.log" 2>&1; done
timeout 900 $P --groups highband --input c64 --out "$OUT/parity_population_highband_c64.json" > "$OUT/pp_hbc64.log" 2>&1
echo done > "$OUT/done_more"
