#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_run10; mkdir -p $O
timeout 300 python scripts/pp_check.py > $O/pp_check.txt 2>&1; tail -2 $O/pp_check.txt
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_dzin_gpu.py tests/test_engine_gpu.py tests/test_range_guard_gpu.py -m gpu -x -q > $O/pytest_ops.txt 2>&1; tail -3 $O/pytest_ops.txt
python scripts/level0_store_ablate.py run > $O/wstore_layers.txt 2>&1; cat $O/wstore_layers.txt
bash scripts/ab_bench.sh deep_calcium_amd/lib/libdcunet_ws0.so deep_calcium_amd/lib/libdcunet.so 3 --no-cpu-baseline 2>&1 | grep -v amdgpu > $O/ab_ws.txt; cat $O/ab_ws.txt
for lib in libdcunet_ws0.so libdcunet.so libdcunet_ws0.so libdcunet.so; do DC_LIB_PATH=$PWD/deep_calcium_amd/lib/$lib python bench.py --mode infer --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib infer', d['value'], d['ms_per_step'])"; done | tee $O/ab_ws_infer.txt
