#!/bin/bash
# round 6, GPU call 1: weight-gradient variants + ablation, role-split conv ablation (3 instantiations x 2 shapes), bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_run1; mkdir -p $O
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
timeout 900 python scripts/wgrad_variants.py run > $O/wgrad_variants.txt 2>&1
for v in 0 3 1; do
  for shape in "128 128 128" "64 256 256"; do
    echo "### variant $v shape $shape" >> $O/pp_ablate.txt
    timeout 600 python scripts/igemm_pp_ablate.py $shape $v >> $O/pp_ablate.txt 2>&1
  done
done
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.txt 2>&1
tail -3 $O/wgrad_variants.txt; tail -3 $O/pp_ablate.txt; tail -1 $O/bench.txt | cut -c1-300
