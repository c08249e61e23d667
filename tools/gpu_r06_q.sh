# round 6: the driver's own commands once more on the final tree - the N = 1 bench line as the driver asks for it, and the N > 1 code path with the one rank a single-GPU box offers
out=gpurun_out/r06q; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_driver.json 2> $out/bench_driver.err; tail -1 $out/bench_driver.json | cut -c1-700; tail -4 $out/bench_driver.err
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --cpu-budget 0 --secondary none ) > $out/bench_torchrun1.json 2> $out/bench_torchrun1.err; tail -1 $out/bench_torchrun1.json | cut -c1-900; tail -4 $out/bench_torchrun1.err
