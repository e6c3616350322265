#!/bin/bash
# PMC evidence for DESIGN.md section 5: separate passes (TCC FETCH_SIZE costs 3 of 4 slots, WRITE_SIZE 2), one train step
# each, single stream so that every kernel's counters are its own.  Run from the repo root on the GPU box:
#   bash scripts/pmc_passes.sh gpurun_out/pmc [bench.py arguments, e.g. --mode infer]
set -u
OUT=${1:-gpurun_out/pmc}
shift || true
EXTRA="$*"
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd "$ROOT"
export DC_STREAMS=1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/$tag -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/$tag.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline $EXTRA > $OUT/trace.log 2>&1
find $OUT -name "*.csv" | head -20
