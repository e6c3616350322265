#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_run4; mkdir -p $O
timeout 600 python scripts/wgrad_variants.py run base abl_presplit base abl_presplit 2>&1 | grep -v amdgpu.ids > $O/wgrad_presplit.txt
for v in 0 1; do for shape in "128 128 128" "64 256 256"; do
  echo "### variant $v shape $shape" >> $O/pp_presplit.txt
  timeout 600 python scripts/igemm_pp_ablate.py $shape $v 0 32 0 32 2>&1 | grep -v amdgpu.ids >> $O/pp_presplit.txt
done; done
cat $O/wgrad_presplit.txt $O/pp_presplit.txt
export DC_DIST_FORCE=1 DC_DIST_BACKEND=nccl MASTER_ADDR=127.0.0.1
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
show() { tail -1 $1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$2', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d.get('comm'), d.get('allreduce_exposed_ms'))"; }
$B > $O/dp_native.txt 2>$O/dp_native.err; show $O/dp_native.txt native
DC_COMM=torch $B > $O/dp_torch.txt 2>$O/dp_torch.err; show $O/dp_torch.txt torch
DC_COMM_FENCE=0 $B > $O/dp_native_nofence.txt 2>&1; show $O/dp_native_nofence.txt native_nofence
DC_AR_BUCKETS=1 $B > $O/dp_native_1bucket.txt 2>&1; show $O/dp_native_1bucket.txt native_1bucket
DC_TAIL_MAIN=0 $B > $O/dp_native_tail0.txt 2>&1; show $O/dp_native_tail0.txt native_tail0
DC_TAPES=0 $B > $O/dp_native_notape.txt 2>&1; show $O/dp_native_notape.txt native_notape
rocprofv3 --kernel-trace --output-format csv -d $O/trace_native -o p -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/trace_native.log 2>&1
DC_COMM=torch rocprofv3 --kernel-trace --output-format csv -d $O/trace_torch -o p -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/trace_torch.log 2>&1
for t in native torch; do f=$(find $O/trace_$t -name "p_kernel_trace.csv" | head -1); python scripts/queue_timeline.py $f 5 > $O/timeline_$t.txt 2>&1; head -3 $O/timeline_$t.txt; rm -rf $O/trace_$t; done
