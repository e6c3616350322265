#!/bin/bash
# Tile-walk A/B on one box (DC_TILE_WALK = 0 row-major | 1 column-major): correctness of the new default, interleaved bench
# lines, and ONE FETCH_SIZE pass per setting (single stream).   bash scripts/ab_walk.sh gpurun_out/walk
set -u
OUT=${1:-gpurun_out/walk}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_fullsize_gpu.py tests/test_hip_ops.py -m gpu -q -x -k "moments or dgrad_bn or pooled or conv or wgrad" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
bash scripts/ab_env.sh DC_TILE_WALK "0 1" > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
for w in 0 1; do
  DC_TILE_WALK=$w DC_STREAMS=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch$w -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/fetch$w.log 2>&1
  python3 - $OUT/fetch$w <<'PY'
import csv, sys, collections, glob, re
f = glob.glob(sys.argv[1] + '/**/p_counter_collection.csv', recursive=True)[0]
v = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'FETCH_SIZE':
        k = re.sub(r'\(.*$', '', re.sub(r'^void ', '', r['Kernel_Name'])).replace(' ', '')
        v[k].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
print(sys.argv[1])
for k, xs in sorted(v.items(), key=lambda kv: -sum(x[0] for x in kv[1]))[:16]:
    print('  %-60s n=%3d  fetch(x2) %8.1f MB/launch  mean %7.1f us' % (k[:60], len(xs), 2 * sum(x[0] for x in xs) / len(xs) / 1024, sum(x[1] for x in xs) / len(xs) / 1e3))
PY
done
