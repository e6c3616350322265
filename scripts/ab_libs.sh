#!/bin/bash
# interleaved A/B of two builds of libdcunet.so on the per-layer table (same box, alternating runs):
#   bash scripts/ab_libs.sh deep_calcium_amd/lib/ab/base.so deep_calcium_amd/lib/ab/new.so [rounds]
A=$1; B=$2; R=${3:-3}
for r in $(seq $R); do
  for v in $A $B; do echo "== $v"; DC_LIB_PATH=$PWD/$v python scripts/layer_table.py 16 2>&1 | grep "per step"; done
done
