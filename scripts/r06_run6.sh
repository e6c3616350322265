#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_run6; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
show() { grep '^{' $1 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$2', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d.get('comm'), d.get('allreduce_exposed_ms'))"; }
( export DC_DIST_FORCE=1 DC_DIST_BACKEND=nccl MASTER_ADDR=127.0.0.1
  $B > $O/dp_native.txt 2>&1; show $O/dp_native.txt native
  DC_COMM=torch $B > $O/dp_torch.txt 2>&1; show $O/dp_torch.txt torch
  GPU_MAX_HW_QUEUES=4 $B > $O/dp_native_q4.txt 2>&1; show $O/dp_native_q4.txt native_q4 )
$B > $O/single.txt 2>&1; show $O/single.txt single
timeout 900 python scripts/soak_fit.py 25 100 128 20 19 > $O/soak_fit.txt 2>&1; grep -E "^epoch (0|1|2|3|10|20|24)" $O/soak_fit.txt | cut -c1-260
bash scripts/pmc_passes.sh $O/infer_pmc --mode infer > $O/infer_pmc.log 2>&1
python scripts/pmc_table.py $O/infer_pmc --json $O/infer_pmc_per_kernel.json > $O/infer_pmc_table.md 2>&1; head -14 $O/infer_pmc_table.md
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tta_stats -o p -- python3 bench.py --mode tta --steps 4 --warmup 3 > $O/tta_stats.log 2>&1
cp $(find $O/tta_stats -name "p_kernel_stats.csv" | head -1) $O/tta_kernel_stats.csv; cp $(find $O/infer_pmc/trace -name "p_kernel_stats.csv" | head -1) $O/infer_kernel_stats.csv
rm -rf $O/tta_stats $O/infer_pmc/*/*/*_agent_info.csv
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
