#!/bin/bash
# Sample the GPU's shader clock, power and temperature every 0.5 s while a command runs: tools/clock_sampler.sh OUT.txt -- cmd ...
OUT=$1; shift; shift
( while true; do echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi -d 0 --showclocks --showpower --showtemp 2>/dev/null | grep -E 'sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|hotspot)' | sed 's/GPU\[0\][ \t]*: //' | tr '\n' ';')"; sleep 0.5; done ) > "$OUT" &
S=$!
"$@"
kill $S 2>/dev/null
