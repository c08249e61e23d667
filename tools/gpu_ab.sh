for w in 1 2 3 4; do
  LUM_CXXFLAGS="-DLUM_SHADE_WAVES=$w" python -m luminary_amd.build --force > /dev/null 2>&1
  for b in 1 8; do
    echo "== waves $w batch $b"
    python bench.py --steps 4 --warmup 1 --cpu-budget 0 --samples-per-pass $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value'],1),'Mrays/s', d['config']['kernel_ms_rank0'])"
  done
done
