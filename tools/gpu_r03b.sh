# Round-3 measurements that need no rebuild on the box: gather ceiling, counter list, fast-vs-exact on the hall at 1024 spp, phase statistics of the
# shipped diagnostic variant (built here: LUM_CXXFLAGS=-DLUM_PHASE_STATS python -m luminary_amd.build --variant phase)
out=gpurun_out/${1:-r03b}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -o /tmp/gather tools/microbench/gather.hip 2> /dev/null && timeout 600 /tmp/gather > $out/gather_1024x1.txt 2>&1
timeout 300 /tmp/gather --block 512 --blocks-per-cu 4 --steps 1000 > $out/gather_512x4.txt 2>&1
rocprofv3 -L > $out/counters_list.txt 2>&1
timeout 900 python tools/flavour_diff.py hall 64 256 1024 > $out/flavour_hall.txt 2>&1; tail -2 $out/flavour_hall.txt
for w in hall scan; do LUM_LIB=$PWD/luminary_amd/lib/variants/phase/libluminary_amd.so timeout 600 python tools/phase_stats.py $w > $out/phase_$w.txt 2>&1; cat $out/phase_$w.txt; done
cat $out/gather_1024x1.txt
