# round 6, fifteenth call: one rank's work of an 8-GPU (and 4-GPU) bench step emulated on one GPU (tools/rank_emulation.py)
out=gpurun_out/r06o; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python tools/rank_emulation.py --world 8 > $out/rank_emulation_8.json 2> $out/rank_emulation_8.err; cat $out/rank_emulation_8.err
timeout 900 python tools/rank_emulation.py --world 4 --ranks 0,2 > $out/rank_emulation_4.json 2> $out/rank_emulation_4.err; cat $out/rank_emulation_4.err
