"""Probe of the direct all-reduce kernel between virtual ranks on one GPU (nb_comm_p2p_virtual_test): wrong elements
and time per all-reduce for the whole-node single-dispatch mode.  NBODY_LIB selects an experimental build."""
import os, sys, ctypes as C, time
sys.path.insert(0, os.getcwd())
from nbody_cosmological_simulation_amd import _native as N
L = N.lib()
bad, us = C.c_int32(0), C.c_double(0.0)
tag = os.path.basename(os.environ.get("NBODY_LIB", "default"))
tot = 0
for P in (2, 4):
    for dt, count in ((N.NB_F64, 131072), (N.NB_F32, 262144), (N.NB_F64, 524288)):
        N.check(L.nb_comm_p2p_virtual_test(0, P, count, dt, 2, 4, 0.5, C.byref(bad), C.byref(us)))
        tot += bad.value
        print(tag, "node", P, "f64" if dt == N.NB_F64 else "f32", count, "bad", bad.value, "us", round(us.value, 2), flush=True)
print(tag, "TOTAL BAD", tot, flush=True)
