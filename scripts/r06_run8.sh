#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_run8; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_dzin_gpu.py tests/test_engine_gpu.py -m gpu -x -q > $O/pytest_ops.txt 2>&1; tail -3 $O/pytest_ops.txt
timeout 300 python scripts/pp_check.py > $O/pp_check.txt 2>&1; tail -4 $O/pp_check.txt
for v in 0 3 1; do for shape in "128 128 128" "64 256 256"; do
  echo "### variant $v shape $shape" >> $O/pp_pf.txt
  for lib in ppabl0_pf0 ppabl0 ppabl0_pf0 ppabl0; do
    echo -n "$lib " >> $O/pp_pf.txt
    DC_LIB_PATH=$PWD/deep_calcium_amd/lib/libdcunet_$lib.so timeout 300 python scripts/igemm_pp_ablate.py --worker $PWD/deep_calcium_amd/lib/libdcunet_$lib.so $shape 0 $v 2>&1 | grep -v amdgpu.ids >> $O/pp_pf.txt
  done
done; done
cat $O/pp_pf.txt
bash scripts/ab_bench.sh deep_calcium_amd/lib/libdcunet_pf0.so deep_calcium_amd/lib/libdcunet.so 3 --no-cpu-baseline 2>&1 | grep -v amdgpu > $O/ab_pf.txt; cat $O/ab_pf.txt
for lib in libdcunet_pf0.so libdcunet.so libdcunet_pf0.so libdcunet.so; do DC_LIB_PATH=$PWD/deep_calcium_amd/lib/$lib python bench.py --mode infer --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib infer', d['value'], d['ms_per_step'])"; done | tee $O/ab_pf_infer.txt
