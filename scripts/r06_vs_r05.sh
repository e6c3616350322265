#!/bin/bash
# round-5 tree (scripts/_r05_tree: `git worktree add scripts/_r05_tree 3a6e506` + build, not committed) against this tree, same box, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_vs_r05; mkdir -p $O
show() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['value'], d['unit'], d['ms_per_step'], 'ms')"; }
for r in 1 2; do
  for t in r05 r06; do
    if [ $t = r05 ]; then D=scripts/_r05_tree; else D=.; fi
    ( cd $D; python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | show "$t train 512^2x16" )
    ( cd $D; python bench.py --window 128 --batch 20 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | show "$t train 128^2x20" )
    ( cd $D; python bench.py --mode infer --steps 30 --warmup 5 2>/dev/null | show "$t forward batch 8" )
    ( cd $D; python bench.py --mode tta --steps 6 --warmup 3 2>/dev/null | show "$t predict 8xTTA" )
    ( cd $D; DC_DIST_FORCE=1 DC_DIST_BACKEND=nccl MASTER_ADDR=127.0.0.1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | show "$t one-rank RCCL step" )
  done
done | tee $O/ab.txt
