#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_run11; mkdir -p $O
DC_LIB_PATH=$PWD/deep_calcium_amd/lib/libdcunet_order1.so timeout 300 python scripts/pp_check.py > $O/pp_check.txt 2>&1; tail -1 $O/pp_check.txt
bash scripts/ab_bench.sh deep_calcium_amd/lib/libdcunet.so deep_calcium_amd/lib/libdcunet_order1.so 4 --no-cpu-baseline 2>&1 | grep -v amdgpu > $O/ab_order.txt; cat $O/ab_order.txt
for lib in libdcunet.so libdcunet_order1.so libdcunet.so libdcunet_order1.so; do DC_LIB_PATH=$PWD/deep_calcium_amd/lib/$lib python bench.py --mode infer --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib infer', d['value'], d['ms_per_step'])"; done | tee $O/ab_order_infer.txt
