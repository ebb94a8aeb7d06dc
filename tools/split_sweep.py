"""us per step over NB_SYM_SPLIT (pieces per sweep of the pair-symmetric kernel) at mid sizes:
   python tools/split_sweep.py float64 5120 6144 8192 ...   (each point in its own process: the knob is read at nb_create)"""
import os, subprocess, sys
mode = sys.argv[1]; ns = sys.argv[2:]
splits = os.environ.get("SPLITS", "0 3 4 5 6 7 8 10").split()
tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "small_n_timing.py")
print("N      " + "".join(f"{'split ' + s:>10}" for s in splits))
for n in ns:
    row = []
    for sp in splits:
        env = dict(os.environ, MODE=mode)
        knob = os.environ.get("KNOB", "NB_SYM_SPLIT")        # KNOB=NB_SYM_ROWSPLIT SPLITS="d 0 1 2": "d" = knob unset
        if knob == "NB_SYM_SPLIT":
            if sp != "0": env[knob] = sp
            else: env.pop(knob, None)
        elif sp == "d": env.pop(knob, None)
        else: env[knob] = sp
        out = subprocess.run([sys.executable, tool, n], env=env, capture_output=True, text=True).stdout
        row.append(out.split(":")[1].split("us")[0].strip() if "us/step" in out else "fail")
    print(f"{n:<7}" + "".join(f"{r:>10}" for r in row), flush=True)
