#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_run3; mkdir -p $O
timeout 1500 python -m pytest tests/test_comm_gpu.py tests/test_dp_gpu.py tests/test_tape_gpu.py tests/test_sync_bn_gpu.py tests/test_abi.py -m gpu -x -q > $O/pytest_dp.txt 2>&1
tail -5 $O/pytest_dp.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.txt 2>$O/bench.err
DC_DIST_FORCE=1 DC_DIST_BACKEND=nccl MASTER_ADDR=127.0.0.1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_dp_native.txt 2>$O/bench_dp_native.err
DC_COMM=torch DC_DIST_FORCE=1 DC_DIST_BACKEND=nccl MASTER_ADDR=127.0.0.1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_dp_torch.txt 2>$O/bench_dp_torch.err
for f in bench bench_dp_native bench_dp_torch; do tail -1 $O/$f.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$f', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['launches'], d.get('comm'), d.get('allreduce_exposed_ms'), d.get('allreduce_ms'))"; done
bash scripts/ab_bench.sh deep_calcium_amd/lib/libdcunet.so deep_calcium_amd/lib/libdcunet_rw8.so 3 --no-cpu-baseline > $O/ab_rw8.txt 2>&1
cat $O/ab_rw8.txt | grep -v amdgpu
timeout 600 python scripts/wgrad_variants.py run base abl_presplit abl_nostage base 2>&1 | grep -v amdgpu.ids > $O/wgrad_presplit.txt
for v in 0 1; do for shape in "128 128 128" "64 256 256"; do
  echo "### variant $v shape $shape" >> $O/pp_presplit.txt
  timeout 600 python scripts/igemm_pp_ablate.py $shape $v 0 32 2 0 2>&1 | grep -v amdgpu.ids >> $O/pp_presplit.txt
done; done
cat $O/wgrad_presplit.txt $O/pp_presplit.txt
