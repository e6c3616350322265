#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_run9; mkdir -p $O
show() { grep '^{' $1 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$2', d['value'], d['ms_per_step'])"; }
for r in 1 2 3; do
  for q in 4 8; do
    GPU_MAX_HW_QUEUES=$q python bench.py --window 128 --batch 20 --steps 100 --warmup 20 --no-cpu-baseline > $O/w128_q$q.txt 2>&1; show $O/w128_q$q.txt w128_q$q
  done
done
for r in 1 2; do for q in 4 8; do
    GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b16_q$q.txt 2>&1; show $O/b16_q$q.txt b16_q$q
done; done
