# Round-end evidence run on the GPU box (through gpurun): parity tests, default bench, rocprofv3 kernel stats of the same bench
# command, and separate PMC passes for memory-side traffic. Everything lands in gpurun_out/<tag>/ ; copy what is judged to profiles/.
#   bash tools/gpu_round_profile.sh r01
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 900 python bench.py > $out/bench_example.json 2> $out/bench_example.err; tail -c 600 $out/bench_example.json
timeout 900 python bench.py --workload hall --steps 4 --warmup 1 > $out/bench_hall.json 2> $out/bench_hall.err; tail -c 300 $out/bench_hall.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_example -- python3 bench.py --cpu-budget 0 > $out/stats_example.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_hall -- python3 bench.py --workload hall --steps 4 --warmup 1 --cpu-budget 0 > $out/stats_hall.log 2>&1
for w in example hall; do
  extra=""; [ $w = hall ] && extra="--workload hall"
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$w -- python3 bench.py --steps 2 --warmup 1 --cpu-budget 0 $extra > $out/pmc_fetch_$w.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$w -- python3 bench.py --steps 2 --warmup 1 --cpu-budget 0 $extra > $out/pmc_write_$w.log 2>&1
done
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc_lds_example -- python3 bench.py --steps 2 --warmup 1 --cpu-budget 0 > $out/pmc_lds_example.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_tcc_hall -- python3 bench.py --workload hall --steps 2 --warmup 1 --cpu-budget 0 > $out/pmc_tcc_hall.log 2>&1
find $out -name "*.csv" | head -40
# additional records: the large-scene workload, builder comparison, adaptive throughput, phase occupancy (diagnostic build last)
timeout 900 python bench.py --workload scan --steps 2 --warmup 1 --cpu-budget 0 > $out/bench_scan.json 2> $out/bench_scan.err; tail -c 300 $out/bench_scan.json
timeout 600 python tools/lbvh_bench.py hall > $out/lbvh_hall.txt 2>&1; tail -3 $out/lbvh_hall.txt
timeout 600 python tools/adaptive_bench.py example 8 2 > $out/adaptive_example.txt 2>&1; tail -5 $out/adaptive_example.txt
LUM_CXXFLAGS=-DLUM_PHASE_STATS python -m luminary_amd.build --force > /dev/null 2>&1
timeout 600 python tools/phase_stats.py example > $out/phase_example.txt 2>&1; timeout 600 python tools/phase_stats.py hall > $out/phase_hall.txt 2>&1; tail -9 $out/phase_hall.txt
python -m luminary_amd.build --force > /dev/null 2>&1
