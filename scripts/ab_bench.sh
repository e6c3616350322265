#!/bin/bash
# interleaved A/B of two builds of libdcunet.so on the bench line (same box, alternating runs):
#   bash scripts/ab_bench.sh deep_calcium_amd/lib/ab/base.so deep_calcium_amd/lib/ab/new.so [rounds] [bench args]
A=$1; B=$2; R=${3:-3}; shift 3
for r in $(seq $R); do
  for v in $A $B; do
    echo "== $v"; DC_LIB_PATH=$PWD/$v python bench.py --steps 20 --warmup 5 "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step, dominant kernel frac', d['roofline']['frac'])"
  done
done
