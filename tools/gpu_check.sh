# Parity tests + a short bench of both workloads on the GPU box: bash tools/gpu_check.sh [bench args]
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for w in example hall; do
  timeout 300 python bench.py --cpu-budget 0 --steps 4 --warmup 1 --workload $w "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],2), 'ms/step', d['config']['kernel_ms_rank0'], d['config']['per_ray_rank0'], 'frac', d['roofline']['frac'])"
done
