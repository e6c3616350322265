#!/bin/bash
# A/B of one environment knob on the same box, interleaved:  bash scripts/ab_env.sh VAR "v1 v2 v3" [bench args]
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do
  for v in $VALS; do
    echo -n "$VAR=$v  "
    env $VAR=$v DC_STREAMS=${DC_STREAMS:-2} python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'kernel avg ms', r['avg_launch_ms'], 'alone TF/s', r.get('achieved_without_concurrent_wgrad_stream'))"
  done
done
