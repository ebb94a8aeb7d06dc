#!/usr/bin/env python3
"""Per-loop census of one kernel's ISA: where scratch (spill) traffic, LDS reads and lane rotations sit.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only file.hip -o /tmp/k.s
    python tools/isa_loop_census.py /tmp/k.s <mangled-name-prefix>
"""
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    prefix = sys.argv[2]
    start = end = None
    for n, l in enumerate(lines):
        if start is None and l.startswith(prefix) and ": ; @" in l:
            start = n
        if start is not None and n > start and l.startswith(".Lfunc_end"):
            end = n
            break
    seg = lines[start:end + 1]
    labels = {}
    for n, l in enumerate(seg):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = n
    loops = []
    for n, l in enumerate(seg):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            loops.append((labels[m.group(1)], n))
    valu = re.compile(r"\s+v_")
    print("lines", len(seg), "scratch ops", sum("scratch_" in x for x in seg))
    for a, b in sorted(loops):
        s = seg[a:b + 1]
        print(f"  loop {a}-{b}: {b - a} lines, scratch {sum('scratch_' in x for x in s)}, "
              f"bpermute {sum('ds_bpermute' in x for x in s)}, valu {sum(valu.match(x) is not None for x in s)}, "
              f"ds_read {sum('ds_read' in x for x in s)}")


if __name__ == "__main__":
    main()
