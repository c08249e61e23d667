cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/staged
export LUM_LIB=$PWD/luminary_amd/lib/variants/staged44/libluminary_amd.so
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/staged/stats -- python3 bench.py --steps 2 --warmup 1 --cpu-budget 0 --secondary none > gpurun_out/staged/stats.log 2>&1
find gpurun_out/staged/stats -name "*kernel_stats.csv" | head -1 | xargs head -14
