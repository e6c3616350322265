python -m pytest tests -q -m gpu -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
DC_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/hp -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
grep -E "head_|conv_c1|maxpool" gpurun_out/hp/p_kernel_stats.csv | cut -c1-160
