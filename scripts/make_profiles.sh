#!/bin/bash
# Everything profiles/ cites, from one GPU box (run from the repo root):  bash scripts/make_profiles.sh gpurun_out/final
#   1. un-profiled bench lines: train (default settings: 2 streams), infer (configs[1]), tta (configs[4]) -> bench*.json
#   2. rocprofv3 --kernel-trace --stats of the train and infer commands                                  -> stats/, stats_infer/
#   3. PMC passes (FETCH_SIZE | WRITE_SIZE | MFMA busy), single stream                                   -> pmc/
set -u
OUT=${1:-gpurun_out/final}
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd "$ROOT"
python3 bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --mode infer --steps 20 --warmup 5 > $OUT/bench_infer.json 2>> $OUT/bench.err
python3 bench.py --mode tta --steps 4 --warmup 3 > $OUT/bench_tta.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_infer -o p -- python3 bench.py --mode infer --steps 10 --warmup 2 > $OUT/stats_infer.log 2>&1
bash scripts/pmc_passes.sh $OUT/pmc > $OUT/pmc.log 2>&1
tail -c 400 $OUT/bench.json
