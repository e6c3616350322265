python -m pytest tests -q -m gpu -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
DC_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/hp -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
grep -E "conv_c1" gpurun_out/hp/p_kernel_stats.csv | cut -c1-120
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
