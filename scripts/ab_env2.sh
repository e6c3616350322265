#!/bin/bash
# like ab_env.sh but values may be empty: bash scripts/ab_env2.sh VAR "none 0 0,1"   ("none" = unset)
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do
  for v in $VALS; do
    echo -n "$VAR=$v  "
    if [ "$v" = "none" ]; then unset $VAR; else export $VAR=$v; fi
    python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'kernel avg ms', r['avg_launch_ms'])"
  done
done
