#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S dump.
    hipcc ... -S --cuda-device-only file.hip -o /tmp/x.s ; python tools/isa_blocks.py /tmp/x.s <mangled-substring> [min-instr]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
minins = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().endswith(":") or (l.startswith("_ZN") and key in l and "; @" in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = [(i, l) for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
print("function lines", len(body), "blocks", len(labels))
for idx, (i, l) in enumerate(labels):
    j = labels[idx + 1][0] if idx + 1 < len(labels) else len(body)
    cnt = {}
    for x in body[i:j]:
        m = re.match(r"\s+([a-z_0-9]+)", x)
        if m:
            cnt[m.group(1)] = cnt.get(m.group(1), 0) + 1
    tot = sum(cnt.values())
    if tot < minins:
        continue
    f = lambda p: sum(v for k, v in cnt.items() if p(k))
    print(f"{l:14s} instr {tot:4d} valu {f(lambda k: k.startswith('v_')):4d} pk {f(lambda k: k.startswith('v_pk')):3d} "
          f"trans {f(lambda k: k in ('v_exp_f32', 'v_log_f32', 'v_rsq_f32', 'v_rsq_f64', 'v_rcp_f32', 'v_sqrt_f32')):3d} "
          f"ds {f(lambda k: k.startswith('ds_')):3d} bperm {cnt.get('ds_bpermute_b32', 0):3d} scratch {f(lambda k: 'scratch' in k):3d} "
          f"branch {f(lambda k: 'cbranch' in k):2d} mov {f(lambda k: k.startswith('v_mov') or k.startswith('v_accvgpr')):3d} waitcnt {cnt.get('s_waitcnt', 0):3d}")
