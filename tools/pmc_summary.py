"""Fold the per-pass rocprofv3 --pmc CSVs into one JSON: {kernel: {counter: {avg, launches}}}.

    python tools/pmc_summary.py out.json gpurun_out/pmc8_FETCH_SIZE gpurun_out/pmc8_WRITE_SIZE ...

Each directory is one `rocprofv3 --pmc <counters> --kernel-trace --output-format csv` pass (the
MI355X guide wants FETCH_SIZE, WRITE_SIZE, the SQ_* set and GRBM_GUI_ACTIVE collected in separate
runs).  Values are kept raw: FETCH_SIZE / WRITE_SIZE are in KiB and FETCH_SIZE under-reports by 2x on
gfx950 — bench.py's pmc_traffic() applies that correction when it reads the summary.
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*\)$", "", name)


def main(out, dirs):
    acc = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = {}
            for row in csv.DictReader(open(path)):
                key = (row["Dispatch_Id"], short(row["Kernel_Name"]), row["Counter_Name"])
                per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])
            for (_, kern, ctr), val in per_dispatch.items():
                s = acc.setdefault(kern, {}).setdefault(ctr, [0.0, 0])
                s[0] += val
                s[1] += 1
    summary = {k: {c: {"avg": s[0] / s[1], "launches": s[1]} for c, s in sorted(v.items())}
               for k, v in sorted(acc.items())}
    json.dump(summary, open(out, "w"), indent=1)
    for k, v in summary.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            hbm = (2.0 * v["FETCH_SIZE"]["avg"] + v["WRITE_SIZE"]["avg"]) * 1024.0
            print(f"{k:60s} {hbm / 1e6:10.3f} MB/launch  ({v['FETCH_SIZE']['launches']} launches)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
