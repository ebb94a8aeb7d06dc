"""Where and when the workgroups of one force_sym_kernel launch ran (experimental build with -DNB_WG_TRACE):
   cd nbody_cosmological_simulation_amd/csrc && make exp EXPNAME=wgtrace EXPFLAGS=-DNB_WG_TRACE
   NBODY_LIB=$PWD/nbody_cosmological_simulation_amd/libnbody_amd_wgtrace.so python tools/wg_trace.py 8192 float64
Prints the launch's span, the workgroups per CU and the per-workgroup durations (100 MHz constant clock)."""
import collections, ctypes as C, os, sys
import numpy as np
if sys.argv[1] == "--load":            # analyse a dump written earlier with DUMP=file.npy (no GPU needed)
    buf = np.load(sys.argv[2]); n = sys.argv[2]; mode = "(dump)"; kernel = "?"
else:
    import torch
    sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import nbody_cosmological_simulation_amd as nb
    from nbody_cosmological_simulation_amd import galaxy, _native
    n = int(sys.argv[1]); mode = sys.argv[2]
    pos, vel, m = galaxy.create_disk_galaxy(n, seed=42, device="cpu")
    sim = nb.GalaxySimulation(pos.cuda(), vel.cuda(), m.cuda(), precision_mode=nb.get_mode_from_string(mode))
    sim.run(200); sim.synchronize()
    L = _native.lib()
    f = L.nb_debug_wg_trace; f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int]
    buf = np.zeros((16384, 8), dtype=np.uint64)
    got = f(buf.ctypes.data, 16384)
    assert got > 0
    kernel = sim.force_kernel_name()
nz = int(np.count_nonzero(buf[:, 1]))
b = buf[:nz].astype(np.int64)
t0 = b[:, 0].min()
start = (b[:, 0] - t0) * 0.01; end = (b[:, 1] - t0) * 0.01          # us
hw = b[:, 4] & 0xffffffff; xcc = b[:, 2] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 3
key = xcc * 1000 + se * 100 + sh * 20 + cu
per = collections.Counter(key.tolist())
dur = end - start
print(f"N={n} {mode}: {nz} workgroups, kernel {kernel}")
print(f"span {end.max():.2f} us; workgroup duration min/median/mean/max {dur.min():.2f}/{np.median(dur):.2f}/{dur.mean():.2f}/{dur.max():.2f} us")
print(f"CUs used {len(per)}; workgroups per CU histogram {sorted(collections.Counter(per.values()).items())}")
print(f"per XCC {sorted(collections.Counter(xcc.tolist()).items())}")
order = np.argsort(start)
print("starts (us) at deciles:", np.round(np.quantile(start, np.linspace(0, 1, 11)), 2).tolist())
print("ends   (us) at deciles:", np.round(np.quantile(end, np.linspace(0, 1, 11)), 2).tolist())
# concurrency per CU: max simultaneously resident workgroups on one CU
mx = 0
for k in per:
    idx = np.nonzero(key == k)[0]
    ev = sorted([(start[i], 1) for i in idx] + [(end[i], -1) for i in idx])
    c = 0
    for _, d in ev: c += d; mx = max(mx, c)
print("max resident workgroups on one CU:", mx)
busy = sorted(((end[key == k].max(), k, per[k]) for k in per), reverse=True)[:5]
print("last CUs to finish (end us, cu key, workgroups):", [(round(float(e), 2), int(k), c) for e, k, c in busy])
first = [(round(float(start[i]), 2), round(float(dur[i]), 2)) for i in order[:6]]
last = [(round(float(start[i]), 2), round(float(dur[i]), 2)) for i in order[-6:]]
print("first workgroups (start, duration):", first, " last:", last)
# placement of a workgroup's four waves on the SIMDs, and the waves resident per SIMD
wsimd = (b[:, 4:8] >> 4) & 3
wdur = (b[:, 4:8] >> 32) * 0.01
print("distinct SIMDs used by a workgroup's 4 waves:", sorted(collections.Counter(len(set(r)) for r in wsimd.tolist()).items()))
load = collections.Counter()
for i in range(nz):
    for w in range(4): load[(int(key[i]), int(wsimd[i, w]))] += 1
print("waves per SIMD histogram:", sorted(collections.Counter(load.values()).items()))
print("per-wave duration min/median/max:", wdur.min(), np.median(wdur), wdur.max())
cyc = b[:, 3].astype(float)
if cyc.max() > 0: print("shader clock (cycles / us) over the workgroups: min/median/max", np.round([np.min(cyc / dur), np.median(cyc / dur), np.max(cyc / dur)], 1).tolist())
if os.environ.get("DUMP"):
    np.save(os.environ["DUMP"], buf[:nz])
ts = np.linspace(0, end.max(), 24, endpoint=False)
print("resident workgroups at", round(float(ts[1]), 1), "us intervals:", [int(((start <= t) & (end > t)).sum()) for t in ts])
