#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_run5; mkdir -p $O
timeout 1500 python -m pytest tests/test_comm_gpu.py tests/test_dp_gpu.py tests/test_tape_gpu.py tests/test_api_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
export DC_DIST_FORCE=1 DC_DIST_BACKEND=nccl MASTER_ADDR=127.0.0.1
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
show() { grep '^{' $1 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$2', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d.get('comm'), d.get('allreduce_exposed_ms'))"; }
$B > $O/dp_native.txt 2>&1; show $O/dp_native.txt native
DC_COMM=torch $B > $O/dp_torch.txt 2>&1; show $O/dp_torch.txt torch
GPU_MAX_HW_QUEUES=8 $B > $O/dp_native_q8.txt 2>&1; show $O/dp_native_q8.txt native_q8
DC_AR_BUCKETS=1 $B > $O/dp_native_1bucket.txt 2>&1; show $O/dp_native_1bucket.txt native_1bucket
unset DC_DIST_FORCE DC_DIST_BACKEND
GPU_MAX_HW_QUEUES=8 $B > $O/single_q8.txt 2>&1; show $O/single_q8.txt single_q8
$B > $O/single.txt 2>&1; show $O/single.txt single
python bench.py --mode infer --steps 30 --warmup 5 > $O/infer.txt 2>&1; grep '^{' $O/infer.txt | cut -c1-200
python bench.py --mode tta --steps 6 --warmup 3 > $O/tta.txt 2>&1; grep '^{' $O/tta.txt | cut -c1-200
