# Memory-path / instruction-cache PMC passes: bash tools/gpu_pmc2.sh <tag> [bench args...]
tag=${1:-pmc2}; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
args="--steps 2 --warmup 1 --cpu-budget 0 --samples-per-pass 8 $@"
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_a -- python3 bench.py $args > gpurun_out/${tag}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_STALL --output-format csv -d gpurun_out/${tag}_b -- python3 bench.py $args > gpurun_out/${tag}_b.log 2>&1
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum --output-format csv -d gpurun_out/${tag}_c -- python3 bench.py $args > gpurun_out/${tag}_c.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum --output-format csv -d gpurun_out/${tag}_d -- python3 bench.py $args > gpurun_out/${tag}_d.log 2>&1
tail -2 gpurun_out/${tag}_*.log | cut -c1-300
