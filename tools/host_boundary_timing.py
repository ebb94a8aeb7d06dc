"""PCIe-inclusive rates of the drop-in boundary at N = 65 536: what a caller pays when
it drives the engine tick by tick from Python and reads the state back to host memory every tick, next to
the device-resident run() loop that bench.py times."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import nbody_cosmological_simulation_amd as nb
from nbody_cosmological_simulation_amd import galaxy

n = 65536
pos, vel, mass = galaxy.create_disk_galaxy(n, seed=42, device="cpu")
t = time.perf_counter()
sim = nb.GalaxySimulation(pos, vel, mass, precision_mode=nb.PrecisionMode.FLOAT64)    # host tensors in
sim.synchronize()
print(f"construct from host tensors (upload + first force): {(time.perf_counter() - t) * 1e3:.2f} ms")
sim.run(60); sim.synchronize()
t = time.perf_counter(); sim.run(200); sim.synchronize(); dt_run = time.perf_counter() - t
t = time.perf_counter()
for _ in range(200):
    sim.step()
sim.synchronize(); dt_step = time.perf_counter() - t
t = time.perf_counter()
for _ in range(200):
    sim.step()
    p = sim.positions; v = sim.velocities          # host tensors: two 1 MiB device-to-host copies per tick
dt_read = time.perf_counter() - t
t = time.perf_counter()
for _ in range(200):
    sim.positions = p                                # host tensor handed back: 1 MiB host-to-device per tick
    sim.step()
    p = sim.positions
dt_rw = time.perf_counter() - t
for name, dt in (("run(200), state resident", dt_run), ("200 x step()", dt_step),
                 ("200 x (step + read pos, vel to host)", dt_read),
                 ("200 x (write pos from host + step + read pos)", dt_rw)):
    print(f"{name:48s} {dt / 200 * 1e3:7.3f} ms/step  {n * 200 / dt:.3e} particle-steps/s")
