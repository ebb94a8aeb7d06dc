import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
from nbody_cosmological_simulation_amd import _native as N
L = N.lib()
bad, us = C.c_int32(0), C.c_double(0.0)
tag = os.path.basename(os.environ.get("NBODY_LIB", "default")) + (" +sync" if os.environ.get("NB_P2P_SYNC") else "")
tot = 0
for P in (1, 2, 4, 8):
    for count in (4099, 131072, 262144, 524288):
        N.check(L.nb_comm_p2p_virtual_test(0, P, count, N.NB_F64, 2, int(os.environ.get("ITERS", "30")), 0.5, C.byref(bad), C.byref(us)))
        tot += bad.value
        print(tag, "P", P, "count", count, "bad", bad.value, "us", round(us.value, 1), flush=True)
print(tag, "TOTAL", tot, flush=True)
