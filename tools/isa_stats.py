#!/usr/bin/env python3
"""Static instruction mix per kernel from a gfx950 assembly dump.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -S --cuda-device-only -o /tmp/core.s luminary_amd/csrc/host/core.hip
  python tools/isa_stats.py /tmp/core.s
"""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read()
parts = re.split(r"\n(_ZN3lum\w+):[^\n]*\n", text)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
    ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = Counter(ins)
    grp = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    short = re.sub(r"^_ZN3lum\d+(exact|fast)\d+", lambda m: m.group(1)[0] + ":", name)[:20]
    print(f"{short:20s} total {len(ins):6d}  valu {grp('v_'):6d}  salu {grp('s_'):6d}  scratch {grp('scratch'):4d}  gload {grp('global_load'):4d}  "
          f"gstore {grp('global_store'):3d}  div {grp('v_div_scale_f32') // 2:4d}  sqrt {grp('v_sqrt_f32'):4d}  rcp {grp('v_rcp_f32'):4d}  "
          f"waitcnt {c.get('s_waitcnt', 0):4d}  branch {grp('s_cbranch'):4d}  pk {grp('v_pk_'):4d}")
