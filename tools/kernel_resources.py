#!/usr/bin/env python3
"""VGPR / scratch / occupancy table of the kernels of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py nbody_cosmological_simulation_amd/csrc/nb_force_sym.hip [name-filter]
"""
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
           "-I/opt/rocm/include", *sys.argv[3:], "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    for b in txt.split("Function Name: ")[1:]:
        name = b.split("\n")[0].strip().split()[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"^void \(anonymous namespace\)::", "", dem).split("(")[0]
        if flt and flt not in dem:
            continue
        g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
        scr, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        print(f"{dem:70s} VGPR {g('VGPRs'):>3s} spill {g('VGPRs Spill'):>3s} scratch {scr:>4s} occ {occ} LDS {lds}")


if __name__ == "__main__":
    main()
