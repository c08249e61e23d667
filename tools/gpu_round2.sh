# Round-2 evidence run on the GPU box (through gpurun). Everything lands in gpurun_out/<tag>/; copy what is judged to profiles/.
#   bash tools/gpu_round2.sh <tag> [steps: tests pmc bench stats calib]   (run in this order)
tag=${1:-r02}; shift
steps=${@:-tests pmc bench stats}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
has() { case " $steps " in *" $1 "*) return 0;; esac; return 1; }
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $out/pytest_gpu.log 2>&1; tail -12 $out/pytest_gpu.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
fi
if has pmc; then  # before the bench line: its roofline blocks carry counter figures only for the source tree the counters were collected from
  extra=""; has calib && extra="--calib"
  timeout 3000 python tools/pmc_collect.py $out/pmc $extra --workloads ${PMC_WORKLOADS:-hall,example,scan} > $out/pmc.log 2>&1; tail -40 $out/pmc.log
  [ -f $out/pmc/pmc_counters.json ] && cp $out/pmc/pmc_counters.json profiles/pmc_counters.json
fi
if has bench; then
  timeout 1200 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 1500 $out/bench.json; tail -3 $out/bench.err
  cp profiles/bench_detail.json $out/bench_detail.json 2>/dev/null
fi
if has stats; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --cpu-budget 0 --secondary none > $out/stats.log 2>&1
  find $out/stats -name "*kernel_stats.csv" | head -1 | xargs head -12
fi
if has flavourdiag; then
  trap 'python -m luminary_amd.build --force > /dev/null 2>&1' EXIT
  for ff in "-ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -freciprocal-math|" "-ffp-contract=off -fno-fast-math|-DLUM_FAST_RSQ=0 -DLUM_FAST_SINCOS=0 -DLUM_FAST_EXPLOG=0" "-ffp-contract=fast|-DLUM_FAST_RSQ=0 -DLUM_FAST_SINCOS=0 -DLUM_FAST_EXPLOG=0" "-ffp-contract=off -fno-hip-fp32-correctly-rounded-divide-sqrt|-DLUM_FAST_RSQ=0 -DLUM_FAST_SINCOS=0 -DLUM_FAST_EXPLOG=0" "-ffp-contract=off -freciprocal-math|-DLUM_FAST_RSQ=0 -DLUM_FAST_SINCOS=0 -DLUM_FAST_EXPLOG=0" "-ffp-contract=off|-DLUM_FAST_SINCOS=0 -DLUM_FAST_EXPLOG=0" "-ffp-contract=off|-DLUM_FAST_RSQ=0 -DLUM_FAST_EXPLOG=0" "-ffp-contract=off|-DLUM_FAST_RSQ=0 -DLUM_FAST_SINCOS=0"; do
    fast_flags="${ff%%|*}"; cxx="${ff##*|}"
    echo "=== LUM_FAST_FLAGS=[$fast_flags] LUM_CXXFLAGS=[$cxx]"
    LUM_FAST_FLAGS="$fast_flags" LUM_CXXFLAGS="$cxx" python -m luminary_amd.build --force > /dev/null 2>&1
    LUM_FAST_FLAGS="$fast_flags" LUM_CXXFLAGS="$cxx" python tools/flavour_diff.py zoo 256 1024 2>&1 | tail -4
  done > $out/flavourdiag.txt 2>&1
  cat $out/flavourdiag.txt
fi
