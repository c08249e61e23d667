# round 5: co-scheduling with half-size ray workgroups (2 waves per SIMD, 72 KB of LDS: a k_shade wave per SIMD fits beside them) and the single-context control at 64 ids per pass
out=$1; mkdir -p $out
show='import json,sys; d=json.load(sys.stdin); print({k:(round(v["samples_per_s"]/1e6,1) if isinstance(v,dict) else v) for k,v in d.items()})'
for w in hall example; do
  LUM_LDS_NODES=320 LUM_LIB=$PWD/luminary_amd/lib/variants/tb512/libluminary_amd.so python tools/coschedule.py --workload $w > $out/cosched_${w}_tb512.json 2>> $out/log.txt
  echo -n "tb512 $w: "; python -c "$show" < $out/cosched_${w}_tb512.json
done
for spp in 32 64; do
  echo -n "one context, hall, $spp ids per pass: "; python bench.py --steps 3 --warmup 1 --cpu-budget 0 --secondary none --workload hall --samples-per-pass $spp 2>/dev/null | python tools/ab_line.py
done
