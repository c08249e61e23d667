# A/B of compile-time variants on the GPU box: bash tools/gpu_ab.sh "<flags 1>" "<flags 2>" ...
for flags in "$@"; do
  LUM_CXXFLAGS="$flags" python -m luminary_amd.build --force > /dev/null 2>&1
  grep -E "k_trace|k_shadow_rays" -A12 luminary_amd/lib/obj/kernel_resource_usage.txt | grep -E "VGPRs:|Scratch|Occupancy" | head -3 | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' | tr '\n' ' '
  echo
  for b in 8; do
    echo "== [$flags] batch $b"
    python bench.py --steps 2 --warmup 1 --cpu-budget 0 --samples-per-pass $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value'],1),'Mrays/s', d['config']['kernel_ms_rank0'], d['config']['per_ray_rank0'])"
  done
done
