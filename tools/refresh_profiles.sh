# Copies the judged summaries of a tools/gpu_round2.sh run from gpurun_out/<tag>/ into profiles/ under a round prefix (run here, after gpurun merged the files):
#   gpurun --timeout 3000 -- 'bash tools/gpu_round2.sh r03 tests bench stats pmc calib'  &&  bash tools/refresh_profiles.sh r03 r03
tag=${1:?tag of the gpu_round2.sh run}; round=${2:-$tag}
src=gpurun_out/$tag
[ -f $src/bench.json ] && cp $src/bench.json profiles/${round}_bench.json
stats=$(find $src/stats -name "*kernel_stats.csv" 2>/dev/null | xargs ls -t 2>/dev/null | head -1)  # the newest (a tag used twice leaves older files behind)
[ -n "$stats" ] && cp "$stats" profiles/${round}_kernel_stats_hall.csv
[ -f $src/pmc/pmc_counters.json ] && cp $src/pmc/pmc_counters.json profiles/pmc_counters.json
if [ -f $src/pytest_gpu.log ]; then { cat $src/pytest_gpu.log; echo; tail -1 $src/smoke.log 2>/dev/null; } > profiles/${round}_pytest_gpu.log; fi
ls -la profiles/${round}_* profiles/pmc_counters.json 2>/dev/null
