#!/usr/bin/env python3
"""Run an UNMODIFIED script of the reference checkout against the MI355X engine:

    python dropin/run_script.py /path/to/reference/main.py --stars 1024 --ticks 200 --compare float64 --no-show

`import simulation / quantization / galaxy` resolve to the modules next to this file (which
re-export nbody_cosmological_simulation_amd); everything else the script imports (metrics.py,
visualization.py, ...) still comes from the reference checkout, untouched.
"""
import os
import runpy
import sys

if len(sys.argv) < 2:
    raise SystemExit(__doc__)
here = os.path.dirname(os.path.abspath(__file__))
script = os.path.abspath(sys.argv[1])
sys.path[:0] = [here, os.path.dirname(here), os.path.dirname(script)]
sys.argv = sys.argv[1:]
runpy.run_path(script, run_name="__main__")
