#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_run2; mkdir -p $O
timeout 900 python scripts/kernel_timeline.py run 2>&1 | grep -v amdgpu.ids > $O/kernel_timeline.txt
timeout 1200 python -m pytest tests/test_tape_gpu.py tests/test_hip_ops.py tests/test_abi.py -m gpu -x -q > $O/pytest_subset.txt 2>&1
cat $O/kernel_timeline.txt; tail -5 $O/pytest_subset.txt
