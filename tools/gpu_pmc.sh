# PMC passes for the bench workload (run on the GPU box through gpurun; counters in separate passes, kernel trace only).
# usage: bash tools/gpu_pmc.sh <tag> [bench args...]
tag=${1:-pmc}; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
args="--steps 2 --warmup 1 --cpu-budget 0 --samples-per-pass 8 $@"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/${tag}_sq -- python3 bench.py $args > gpurun_out/${tag}_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/${tag}_sq2 -- python3 bench.py $args > gpurun_out/${tag}_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d gpurun_out/${tag}_tcc -- python3 bench.py $args > gpurun_out/${tag}_tcc.log 2>&1
ls -R gpurun_out/${tag}_* | grep csv | head -20
