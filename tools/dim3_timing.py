"""Step time of the D = 3 kernels (reality_glitch_tests.py runs the stock class with (N,3) tensors)."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import nbody_cosmological_simulation_amd as nb

torch.manual_seed(0)
n = 65536
pos = torch.randn(n, 3) * 5
vel = torch.randn(n, 3) * 0.05
mass = torch.ones(n)
for mode in (nb.PrecisionMode.FLOAT64, nb.PrecisionMode.FLOAT32, nb.PrecisionMode.INT8_SIM):
    sim = nb.GalaxySimulation(pos.cuda(), vel.cuda(), mass.cuda(), precision_mode=mode, profile=True)
    sim.run(40); sim.synchronize(); sim.kernel_time()      # 40 steps: past the clock ramp (profiles/r01_v8_clock_ramp.txt)
    t = time.perf_counter(); sim.run(100); sim.synchronize(); dt = time.perf_counter() - t
    ms, k = sim.kernel_time()
    flops = 19.0 * n * n        # 5D+4 per ordered pair
    peak = 78.6 if mode == nb.PrecisionMode.FLOAT64 else 157.3
    print(f"D=3 {mode.value}: {dt/100*1e3:.3f} ms/step, kernel {ms/k:.3f} ms = {flops/(ms/k*1e-3)/1e12:.1f} TFLOP/s "
          f"({flops/(ms/k*1e-3)/1e12/peak*100:.0f} % of peak), {sim.force_kernel_name()}")
