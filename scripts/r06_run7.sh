#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_run7; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_dzin_gpu.py -m gpu -x -q > $O/pytest_ops.txt 2>&1; tail -3 $O/pytest_ops.txt
timeout 600 python scripts/wgrad_variants.py run xtile0 base xtile0 base 2>&1 | grep -v amdgpu.ids > $O/wgrad_xtile.txt; cat $O/wgrad_xtile.txt
bash scripts/ab_bench.sh deep_calcium_amd/lib/libdcunet_xtile0.so deep_calcium_amd/lib/libdcunet.so 3 --no-cpu-baseline 2>&1 | grep -v amdgpu > $O/ab_xtile.txt; cat $O/ab_xtile.txt
