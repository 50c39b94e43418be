"""BASELINE.json's full sizes on the GPU.  gx1v7 is small enough for the oracle (two steps); tx0.1v3 is not, so
it is checked through size-independent properties of the scheme: the volume-integrated tracer content is
conserved to round-off when the surface fluxes vanish (flux-form advection and diffusion with the variable-
thickness surface layer), a halo update is idempotent, and a run is reproducible bit for bit (ordered sums)."""
import numpy as np
import pytest

from popcfg import named_config
from orclib import Oracle

pytestmark = pytest.mark.gpu


def content(m, cfg, dz, tl, n):
    """sum over ocean cells of TAREA * thickness * T (the surface layer includes the free-surface height)"""
    tarea = m.get("TAREA")[:, 2:-2, 2:-2]
    kmt = m.geti("KMT")[:, 2:-2, 2:-2]
    P = m.get("PSURF", tl)[:, 2:-2, 2:-2]
    T = m.get("TRACER", tl, n)[:, :, 2:-2, 2:-2]
    tot = 0.0
    for k in range(cfg.km):
        th = dz[k] + (P / 980.6 if k == 0 else 0.0)
        tot += float((tarea * th * (kmt > k) * T[:, k]).sum())
    return tot


@pytest.mark.parametrize("kw", [{}, {"solver_choice": 2}, {"solver_choice": 3}, {"tmix_opt": 3, "tadvect": 2}])
def test_gx1v7_full_size_matches_oracle(pkg, orclib_built, kw):
    """BASELINE configs[2] (KPP + pcg) and the CESM production choices on the same grid: ChronGear, P-CSI, and the
    Robert filter with third-order upwind advection."""
    cfg = named_config("gx1v7", **kw)
    gpu, orc = pkg.PopModel(cfg), Oracle(cfg)
    for s in range(2):
        gpu.step(); it = orc.step()
        assert gpu.solver_diagnostics()[0] == it
    for name, three_d in (("TRACER", True), ("UVEL", True), ("VVEL", True), ("RHO", True), ("PSURF", False), ("UBTROP", False)):
        a = gpu.get(name, 1, 0)
        b = (orc.f3 if three_d else orc.f2)(name, 1, 0)
        e = np.abs(a - b).max() / np.abs(b).max()
        assert e <= 1e-8, (name, e)
    gpu.close(); orc.close()


@pytest.mark.parametrize("kw", [{}, {"solver_choice": 3}])
def test_tx01v3_full_size_properties(pkg, orclib_built, kw):
    cfg = named_config("tx0.1v3", **kw)
    small = named_config("tx0.1v3", nx_global=36, ny_global=24, block_size_x=36, block_size_y=24)
    o = Oracle(small)                      # vertical grid only (same km, same generator)
    dz = o.v1("dz")[1:cfg.km + 1].copy()
    o.close()
    m = pkg.PopModel(cfg)
    c0 = [content(m, cfg, dz, 1, n) for n in (0, 1)]
    iters = []
    for _ in range(3):
        m.step(); iters.append(m.solver_diagnostics()[0])
    c1 = [content(m, cfg, dz, 1, n) for n in (0, 1)]
    for a, b in zip(c0, c1):
        assert abs(a - b) <= 1e-12 * abs(a), (a, b)
    # halo update is idempotent on an updated field
    before = m.get("UVEL", 1)[:, :4].copy()
    m.halo_update("UVEL", tl=1)
    assert np.array_equal(before, m.get("UVEL", 1)[:, :4])
    psurf = m.get("PSURF", 1).copy()
    m.close()
    # reproducible bit for bit
    m2 = pkg.PopModel(cfg)
    it2 = []
    for _ in range(3):
        m2.step(); it2.append(m2.solver_diagnostics()[0])
    assert it2 == iters
    assert np.array_equal(psurf, m2.get("PSURF", 1))
    m2.close()


def test_tx01v3_land_elimination_is_invisible(pkg, monkeypatch):
    """the benchmark configuration itself: seven steps (four write the land values, three run with the workgroups without ocean
    left out, the solver and stencil kernels on their compacted lists) against seven steps with every workgroup -- same
    iteration counts, same surface pressure and surface fields bit for bit, land and ghost cells included"""
    cfg = named_config("tx0.1v3")
    out = []
    for skip in ("0", "1"):
        monkeypatch.setenv("POP_LAND_SKIP", skip)
        m = pkg.PopModel(cfg)
        iters = []
        for _ in range(7):
            m.step(); iters.append(m.solver_diagnostics())
        assert m.dim("land_skip_active") == int(skip)
        if skip == "1":
            assert 0.3 < m.scalar("land_tile_fraction") < 0.4
        out.append((iters, m.get("PSURF", 1).copy(), m.get("UBTROP", 1).copy(), m.get("TRACER", 1, 0)[:, 0].copy(), m.get("UVEL", 1)[:, 3].copy(),
                    m.get("RHO", 1)[:, cfg.km - 1].copy(), m.get("HBLT").copy()))
        m.close()
    a, b = out
    assert a[0] == b[0]
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(x, y)
