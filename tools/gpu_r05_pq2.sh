# round 5: more phase-queue variants (A/B), the counters of the default build and of the first variant on the hall, the MFMA slab microbenchmark
out=$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
./tools/microbench/mfma_slab > $out/mfma_slab.txt 2>&1; cat $out/mfma_slab.txt
WORKLOADS="hall scan" bash tools/gpu_ab_variants.sh $out/ab2.txt pq96s4 pqb512 pqb512p192
timeout 900 python tools/pmc_collect.py $out/pmc_default --workloads hall --passes sq,sq2,ta > $out/pmc_default.log 2>&1
LUM_LIB=$PWD/luminary_amd/lib/variants/pq/libluminary_amd.so timeout 900 python tools/pmc_collect.py $out/pmc_pq --workloads hall --passes sq,sq2,ta --label pq > $out/pmc_pq.log 2>&1
tail -3 $out/pmc_default.log $out/pmc_pq.log
