#!/bin/bash
# Developer A/B (GPU box): the latency modes -- one host buffer at a time (--stage single) and the hipGraph streaming mode -- in-tree vs build_exp/liblcs_$1.so
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for L in "" "--lib build_exp/liblcs_$1.so"; do
  for st in "single --steps 300 --warmup 30" "stream --steps 400 --warmup 40"; do
    timeout 120 python bench.py --stage $st --no-cpu-baseline $L 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s %-34s ms_per_step %.4f' % ('$st'.split()[0], '$L'[-30:] or 'in-tree', j['ms_per_step']))"
  done
done
done
