"""us per step over NB_SYM_TAIL (pieces per sweep of the tail-smoothed super-rows; "none" = NB_SYM_SPLIT=1, no smoothing):
   python tools/tail_sweep.py MODE N..."""
import os, subprocess, sys
mode = sys.argv[1]; ns = sys.argv[2:]
vals = ["default", "none", "2", "4", "8"]
tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "small_n_timing.py")
print("N       " + "".join(f"{s:>10}" for s in vals))
for n in ns:
    row = []
    for g in vals:
        env = dict(os.environ, MODE=mode)
        if g == "none": env["NB_SYM_SPLIT"] = "1"
        elif g != "default": env["NB_SYM_TAIL"] = g
        out = subprocess.run([sys.executable, tool, n], env=env, capture_output=True, text=True).stdout
        row.append(out.split(":")[1].split("us")[0].strip() if "us/step" in out else "fail")
    print(f"{n:<8}" + "".join(f"{r:>10}" for r in row), flush=True)
