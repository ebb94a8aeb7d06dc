for v in "" _r2exact "" _r2exact; do
  if [ -n "$v" ]; then export NBODY_LIB=$PWD/nbody_cosmological_simulation_amd/libnbody_amd$v.so; else unset NBODY_LIB; fi
  echo "lib [$v]"
  python tools/mode_sweep.py 2>&1 | grep "int8_sim\|int4_sim\|custom" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ', d['mode'], d['ms_per_launch'])"
done
