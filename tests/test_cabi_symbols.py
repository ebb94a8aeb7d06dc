"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/nbody_amd.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "nbody_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from nbody_cosmological_simulation_amd import _native
    lib = _native.lib()
    declared = header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/nbody_amd.h but not exported"
    assert sorted(_native.EXPORTS) == declared
    assert lib.nb_abi_version() == 3


def test_config_struct_layout_matches_header():
    from nbody_cosmological_simulation_amd import _native
    # 4 int32, 3 double, 4 int32 -> 16 + 24 + 16 bytes, no padding surprises
    assert C.sizeof(_native.NbConfig) == 56
    assert _native.NbConfig.G.offset == 16 and _native.NbConfig.device.offset == 40


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_gpu_means_loud_failure_not_fallback():
    import nbody_cosmological_simulation_amd as nb
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nb.GalaxySimulation(torch.randn(8, 2), torch.randn(8, 2), torch.ones(8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nb._grid_quantize_safe(torch.rand(4, 4), 16)


def test_direct_allreduce_entry_points_fail_loudly_without_a_device():
    """The setup calls of the direct all-reduce (include/nbody_amd.h, nb_comm_p2p_*) report errors -- they never
    crash and never pretend -- when there is no GPU, no exported region or no communicator."""
    import ctypes as C
    from nbody_cosmological_simulation_amd import _native as N
    L = N.lib()
    assert L.nb_comm_p2p_state() == 0 and L.nb_comm_ready() == 0
    buf = C.create_string_buffer(128)
    size = C.c_int32(128)
    if N.device_count() == 0:
        assert L.nb_comm_p2p_export(0, 0, 1, 1 << 20, buf, C.byref(size)) != 0
        assert "device" in N.last_error().lower() or "hip" in N.last_error().lower()
    assert L.nb_comm_p2p_import(buf, 1) != 0                       # nothing exported in this process
    assert L.nb_comm_p2p_selftest(1, 0.1) != 0
    x = (C.c_double * 4)()
    assert L.nb_comm_p2p_allreduce(x, 4, N.NB_F64, 0.1) != 0
    us = C.c_double(0.0)
    assert L.nb_comm_allreduce_time(None, 1, 10, C.byref(us)) != 0
    assert L.nb_comm_allreduce_time(None, 0, 10, C.byref(us)) != 0
    assert L.nb_comm_p2p_enable(1) == 0 and L.nb_comm_p2p_state() == 0   # enabling without an attached region is a no-op
    assert L.nb_comm_quiesce() == 0 and L.nb_comm_shutdown() == 0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "nbody_cosmological_simulation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower(), f"{f} mentions the oracle"


def test_mode_helpers_match_reference_tables():
    import json
    import nbody_cosmological_simulation_amd as nb
    api = json.load(open(os.path.join(ROOT, "tests", "golden", "api.json")))
    for s, v in api["mode_from_string"].items():
        assert nb.get_mode_from_string(s).value == v
    for m, text in api["describe_mode"].items():
        assert nb.describe_mode(nb.PrecisionMode(m)) == text
    assert {m.name: m.value for m in nb.PrecisionMode} == api["precision_mode_members"]


def test_shard_ranges_tile_the_sources():
    from nbody_cosmological_simulation_amd.runtime import shard_range
    for n in (1, 7, 257, 65536, 1048576):
        for world in (1, 2, 3, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))


def test_galaxy_generators_reproduce_reference_draws():
    import numpy as np
    from conftest import load_golden
    from nbody_cosmological_simulation_amd import galaxy
    g = load_golden("g6_galaxy_metrics.npz")
    for name, fn, kw in (("disk", galaxy.create_disk_galaxy, dict(num_stars=2000)),
                         ("test", galaxy.create_test_galaxy, dict(num_stars=1000)),
                         ("halo", galaxy.create_galaxy_with_halo, dict(num_stars=1500))):
        torch.manual_seed(7)
        p, v, m = fn(device="cpu", **kw)
        assert p.dtype == torch.float32
        assert np.allclose(p.numpy(), g[f"{name}/pos"], rtol=0, atol=1e-6)
        assert np.allclose(v.numpy(), g[f"{name}/vel"], rtol=0, atol=1e-6)
        assert np.array_equal(m.numpy(), g[f"{name}/mass"])
    r = torch.from_numpy(g["nfw_r"])
    assert np.allclose(galaxy.nfw_enclosed_mass(r, 5000.0, 30.0).numpy(), g["nfw"], rtol=1e-6)


def test_zero_stars_matches_reference_behaviour():
    """N = 0 needs no device: empty tensors with the reference's dtype promotion, 0.0 / -0.0 energies,
    tick counting, and the grid modes failing like torch's min() of an empty tensor."""
    import math
    import nbody_cosmological_simulation_amd as nb
    for mode, dt in ((nb.PrecisionMode.FLOAT64, torch.float64), (nb.PrecisionMode.FLOAT32, torch.float32)):
        s = nb.GalaxySimulation(torch.zeros(0, 2), torch.zeros(0, 2), torch.zeros(0), mode)
        assert s.accelerations.shape == (0, 2) and s.accelerations.dtype == dt
        s.step()
        s.run(3)
        assert s.tick == 4 and s.positions.shape == (0, 2) and s.positions.dtype == dt
        assert s.get_kinetic_energy() == 0.0
        pe = s.get_potential_energy()
        assert pe == 0.0 and math.copysign(1.0, pe) == -1.0
        assert sorted(s.get_state().keys()) == ["masses", "positions", "precision_mode", "tick", "velocities"]
    with pytest.raises(RuntimeError):
        nb.GalaxySimulation(torch.zeros(0, 2), torch.zeros(0, 2), torch.zeros(0), nb.PrecisionMode.INT4_SIM)


def test_galaxy_generators_non_default_parameters():
    """g11: radius, core fraction, halo radius, dark-matter ratio, tiny star counts -- same draws as the reference."""
    import numpy as np
    from conftest import load_golden
    from nbody_cosmological_simulation_amd import galaxy
    g = load_golden("g11_generators.npz")
    cases = [("disk_r5_c0.6", galaxy.create_disk_galaxy, dict(num_stars=777, galaxy_radius=5.0, core_mass_fraction=0.6)),
             ("disk_r40_c0", galaxy.create_disk_galaxy, dict(num_stars=1234, galaxy_radius=40.0, core_mass_fraction=0.0)),
             ("disk_n3", galaxy.create_disk_galaxy, dict(num_stars=3)),
             ("test_n17", galaxy.create_test_galaxy, dict(num_stars=17)),
             ("halo_r8_h50_dm20", galaxy.create_galaxy_with_halo,
              dict(num_stars=900, galaxy_radius=8.0, halo_radius=50.0, dm_mass_ratio=20.0)),
             ("halo_dm0", galaxy.create_galaxy_with_halo, dict(num_stars=500, dm_mass_ratio=0.0))]
    for name, fn, kw in cases:
        torch.manual_seed(11)
        p, v, m = fn(device="cpu", **kw)
        assert [str(p.dtype), str(v.dtype), str(m.dtype)] == list(g[f"{name}/dtypes"]), name
        assert p.shape == g[f"{name}/pos"].shape and v.shape == g[f"{name}/vel"].shape, name
        assert np.allclose(p.numpy(), g[f"{name}/pos"], rtol=1e-6, atol=1e-6), name
        assert np.allclose(v.numpy(), g[f"{name}/vel"], rtol=1e-5, atol=1e-6), name
        assert np.array_equal(m.numpy(), g[f"{name}/mass"]), name
