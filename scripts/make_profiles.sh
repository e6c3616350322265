#!/bin/bash
# Everything profiles/ cites, from one GPU box (run from the repo root):  bash scripts/make_profiles.sh gpurun_out/final
#   1. PMC passes of the train step (FETCH_SIZE | WRITE_SIZE | MFMA busy, single stream) -> pmc/, and profiles/pmc_traffic.json
#      stamped from them FIRST, so that the bench line of step 2 carries roofline.traffic of THIS build
#   2. un-profiled bench lines: train (default settings: 2 streams), infer (configs[1]), tta (configs[4]) -> bench*.json
#   3. rocprofv3 --kernel-trace --stats of the train, infer and tta commands                              -> stats/, stats_infer/, stats_tta/
#   4. PMC passes of the forward-only command                                                            -> infer_pmc/
set -u
OUT=${1:-gpurun_out/final}
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd "$ROOT"
bash scripts/pmc_passes.sh $OUT/pmc > $OUT/pmc.log 2>&1
python3 scripts/pmc_table.py $OUT/pmc --json $OUT/pmc_per_kernel.json --traffic profiles/pmc_traffic.json > $OUT/pmc_table.md 2>> $OUT/pmc.log
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python3 bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --mode infer --steps 20 --warmup 5 > $OUT/bench_infer.json 2>> $OUT/bench.err
python3 bench.py --mode tta --steps 6 --warmup 3 > $OUT/bench_tta.json 2>> $OUT/bench.err
python3 bench.py --window 128 --batch 20 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_w128.json 2>> $OUT/bench.err
python3 bench.py --window 96 --batch 32 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_w96.json 2>> $OUT/bench.err
DC_MFMA=f32 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_f32.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_infer -o p -- python3 bench.py --mode infer --steps 10 --warmup 2 > $OUT/stats_infer.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_tta -o p -- python3 bench.py --mode tta --steps 4 --warmup 3 > $OUT/stats_tta.log 2>&1
bash scripts/pmc_passes.sh $OUT/infer_pmc --mode infer > $OUT/infer_pmc.log 2>&1
python3 scripts/pmc_table.py $OUT/infer_pmc --json $OUT/infer_pmc_per_kernel.json > $OUT/infer_pmc_table.md 2>> $OUT/infer_pmc.log
# keep what merges back small: the per-dispatch traces are not needed once the tables exist
find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "p_kernel_trace.csv" -path "*pmc*" -delete
tail -c 600 $OUT/bench.json
