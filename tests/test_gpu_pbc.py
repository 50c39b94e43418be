"""Partial bottom cells (grid.F90:916-1020 and the `partial_bottom_cells` branches of advection.F90,
hmix_del2.F90, hmix_del4.F90, vertical_mix.F90, baroclinic.F90, vmix_kpp.F90, sw_absorption.F90): every phase of the step
through the C ABI against the CPU oracle, phase by phase, on stepped bathymetry with a bottom-cell thickness that differs from
column to column."""
import numpy as np
import pytest

from popcfg import named_config, synthetic_grid, synthetic_dzbc
from orclib import Oracle
from test_gpu_parity import run_phases, force_kpp_case, relerr, TOL_LOCAL, TOL_SOLVE

pytestmark = pytest.mark.gpu

DEL4 = {"hmix_momentum": 4, "hmix_tracer": 4, "lvariable_hmix": 1, "am": -1.0e22, "ah": -1.0e21}


def _run(pkg, cfg, grid, nsteps, calm=False):
    gpu, orc = pkg.PopModel(cfg, grid=grid), Oracle(cfg, grid=grid)
    for n in (0, 1):
        assert np.array_equal(gpu.get("TRACER", 1, n), orc.f3("TRACER", 1, n))
    if calm:   # lw_lim across a fold: see test_gpu_parity.py::test_grid_input_step_phases_match_oracle
        z = np.zeros_like(gpu.get("SMF", n=0))
        for n in (0, 1):
            gpu.set("SMF", z, n=n); gpu.set("SMFT", z, n=n)
            orc.f2("SMF", 1, n)[...] = 0.0; orc.f2("SMFT", 1, n)[...] = 0.0
    if cfg.vmix_choice == 3:
        force_kpp_case(gpu, orc)
    tol = TOL_LOCAL
    for s in range(1, nsteps + 1):
        run_phases(gpu, orc, s, tol)
        tol = TOL_SOLVE
    out = [gpu.get(n, 1, 0).copy() for n in ("TRACER", "UVEL", "PSURF")]
    gpu.close(); orc.close()
    return out


@pytest.mark.parametrize("name,kw,nsteps", [
    ("tiny", {"stepped_bathymetry": 1}, 5),                                         # const vmix, del2, centred advection, 16 blocks
    ("tiny", {"stepped_bathymetry": 1, "block_size_x": 48, "block_size_y": 40, "solver_choice": 2}, 4),   # one block, ChronGear
    ("tiny", dict(DEL4, stepped_bathymetry=1), 5),                                  # del4 + variable mixing
    ("tiny", {"stepped_bathymetry": 1, "vmix_choice": 3, "km": 24}, 5),             # KPP
    ("tiny", dict(DEL4, stepped_bathymetry=1, vmix_choice=3, km=24, ldbl_diff=1), 5),   # the tx0.1v3 physics: del4 + KPP + double diffusion
    ("tiny", {"stepped_bathymetry": 1, "vmix_choice": 3, "km": 20, "lshort_wave": 1, "sw_absorption_type": 1, "lsw_absorb": 1, "lcheckekmo": 1}, 4),
    ("tiny", {"stepped_bathymetry": 1, "tadvect": 2}, 5),                            # upwind3
    ("tiny", {"stepped_bathymetry": 1, "lpressure_avg": 0, "tmix_opt": 1, "time_mix_freq": 3, "impcor": 0}, 4),
    ("tiny", {"stepped_bathymetry": 1, "tmix_opt": 3, "solver_choice": 3}, 4),       # Robert filter, P-CSI
    ("tiny", {"stepped_bathymetry": 1, "km": 60, "vmix_choice": 3}, 3),              # production level count
    ("tiny", {}, 3),                                                                 # flat bottom with a partial bottom level
    ("tiny", {"stepped_bathymetry": 1, "vmix_choice": 2}, 5),                        # Richardson mixing (vmix_rich.F90:266-312)
    ("tiny", dict(DEL4, stepped_bathymetry=1, vmix_choice=2, tadvect=2, km=24), 4),
    ("tiny", {"stepped_bathymetry": 1, "tadvect": 3}, 5),                            # lw_lim (advection.F90:2757-3140)
    ("tiny", dict(DEL4, stepped_bathymetry=1, tadvect=3, vmix_choice=3, km=24), 4),
    ("tiny", {"stepped_bathymetry": 1, "tadvect": 3, "block_size_x": 48, "block_size_y": 40}, 4),   # one block
    ("test", {"stepped_bathymetry": 1, "vmix_choice": 1}, 4),                        # 96 blocks
    ("gx3v7", {"stepped_bathymetry": 1}, 3),
])
def test_partial_bottom_cells_step_phases_match_oracle(pkg, orclib_built, name, kw, nsteps):
    cfg = named_config(name, partial_bottom_cells=1, **kw)
    a = _run(pkg, cfg, None, nsteps)
    if name == "tiny" and not kw.get("tmix_opt"):
        # ... and the bottom thickness matters: the same run on full cells ends somewhere else
        b = pkg.PopModel(named_config(name, **kw))
        if cfg.vmix_choice == 3:
            o = Oracle(named_config(name, **kw)); force_kpp_case(b, o); o.close()
        for _ in range(nsteps):
            b.step()
        assert np.abs(a[0] - b.get("TRACER", 1, 0)).max() > 1.0e-6
        b.close()


@pytest.mark.parametrize("kw,nsteps", [
    ({"ns_boundary": 2}, 5),                                                         # through a tripole fold
    ({"ns_boundary": 2, "vmix_choice": 3, "km": 24, "ldbl_diff": 1, "hmix_momentum": 4, "hmix_tracer": 4, "lvariable_hmix": 1, "am": -1.0e22, "ah": -1.0e21}, 4),
    ({"ns_boundary": 2, "block_size_x": 48, "block_size_y": 40, "tadvect": 2}, 4),
    ({"ns_boundary": 0, "ew_boundary": 0, "vmix_choice": 3, "km": 24}, 3),           # closed boundaries
    ({"ns_boundary": 1}, 3),
    ({"ns_boundary": 2, "tadvect": 3}, 4),                                           # lw_lim through a tripole fold
    ({"ns_boundary": 0, "ew_boundary": 0, "tadvect": 3, "vmix_choice": 2}, 4),
])
def test_partial_bottom_cells_on_a_caller_grid(pkg, orclib_built, kw, nsteps):
    """the production route: horiz_grid_file / topography_file / bottom_cell_file records (pop_create_with_grid), stepped KMT and a
    random bottom thickness in [0.2, 1] dz(KMT)"""
    cfg = named_config("tiny", partial_bottom_cells=1, **kw)
    grid = synthetic_grid(cfg)
    grid["DZBC"] = synthetic_dzbc(cfg, grid["KMT"])
    _run(pkg, cfg, grid, nsteps, calm=(kw.get("tadvect") == 3 and kw.get("ns_boundary") == 2))


@pytest.mark.parametrize("vmix,tadvect", [(1, 1), (2, 1), (1, 3)])
def test_bottom_cells_of_full_thickness_change_nothing_physical(pkg, orclib_built, vmix, tadvect):
    """DZBC = dz(KMT) everywhere is the full-cell geometry written through the partial-bottom-cell formulas: the two runs agree
    to rounding (the formulas divide by thicknesses where the full-cell ones multiply by reciprocals), not bitwise.  Constant
    vertical mixing: the partial-bottom-cell branches of KPP are not a generalisation of the full-cell ones (e.g. the bulk
    Richardson number measures depth from zt(1) instead of from half the surface-layer thickness, vmix_kpp.F90:2561-2575);
    Richardson mixing is (vmix_rich.F90:266-312: the shear over the U-cell spacing, averaged, against the difference of averages)."""
    kw = {"stepped_bathymetry": 1, "hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e22, "ah": -1.0e21, "vmix_choice": vmix, "tadvect": tadvect}
    cfg = named_config("tiny", partial_bottom_cells=1, ns_boundary=1, **kw)
    ref = named_config("tiny", ns_boundary=1, **kw)
    grid = synthetic_grid(cfg)
    o = Oracle(ref, grid=grid)
    dz = o.v1("dz")
    o.close()
    grid["DZBC"] = np.where(grid["KMT"] > 0, dz[np.clip(grid["KMT"], 0, cfg.km)], 0.0)
    a, b = pkg.PopModel(cfg, grid=grid), pkg.PopModel(ref, grid=grid)
    for _ in range(5):
        a.step(); b.step()
    for f in ("TRACER", "UVEL", "VVEL", "PSURF"):
        e = relerr(a.get(f, 1, 0), b.get(f, 1, 0))
        assert e < 1.0e-9, (f, e)
    a.close(); b.close()


def test_partial_bottom_cells_refusals(pkg):
    """a topography record without the record of bottom_cell_file is refused (every scheme of the path is built with partial bottom
    cells: Richardson mixing, lw_lim advection and the mixed-layer diagnostics since round 3)"""
    cfg = named_config("tiny", partial_bottom_cells=1)
    grid = synthetic_grid(cfg)
    with pytest.raises(pkg.PopError, match="DZBC"):
        pkg.PopModel(cfg, grid=grid, host_only=True)


@pytest.mark.parametrize("kw", [{"km": 24}, {"km": 24, "ldbl_diff": 1, "ns_boundary": 2}])
def test_mixed_layer_depth_diagnostic_with_partial_bottom_cells(pkg, orclib_built, kw):
    """HMXL with partial bottom cells (vmix_kpp.F90:1326-1356: depths and spacings from the cell thicknesses) on a caller's grid
    with a random bottom thickness: every cell against the oracle, and different from the full-cell diagnostic in the columns
    whose maximum gradient sits at the bottom cell"""
    cfg = named_config("tiny", partial_bottom_cells=1, vmix_choice=3, kpp_diagnostics=1, **kw)
    grid = synthetic_grid(cfg)
    grid["DZBC"] = synthetic_dzbc(cfg, grid["KMT"])
    gpu, orc = pkg.PopModel(cfg, grid=grid), Oracle(cfg, grid=grid)
    force_kpp_case(gpu, orc)
    tol = TOL_LOCAL
    for s in range(1, 4):
        run_phases(gpu, orc, s, tol)
        tol = TOL_SOLVE
        for f in ("HMXL", "HMXL_DR"):
            a, b = gpu.get(f), orc.f2(f)
            assert np.abs(b).max() > 1.0e3 and len(np.unique(np.round(b, 3))) > 20, f + ": trivial field"
            assert relerr(a, b) <= tol * 100, "%s step %d: %g" % (f, s, relerr(a, b))
    full = pkg.PopModel(named_config("tiny", vmix_choice=3, kpp_diagnostics=1, **kw), grid={k: v for k, v in grid.items() if k != "DZBC"})
    o2 = Oracle(named_config("tiny", vmix_choice=3, kpp_diagnostics=1, **kw), grid={k: v for k, v in grid.items() if k != "DZBC"})
    force_kpp_case(full, o2); o2.close()
    for _ in range(3):
        full.step()
    assert np.abs(full.get("HMXL") - gpu.get("HMXL")).max() > 1.0     # cm
    gpu.close(); orc.close(); full.close()


@pytest.mark.parametrize("kw", [{"tadvect": 1}, {"tadvect": 2, "hmix_tracer": 4, "hmix_momentum": 4, "am": -1.0e22, "ah": -1.0e21}, {"vmix_choice": 3, "km": 20},
                                {"tadvect": 3}, {"tadvect": 3, "vmix_choice": 2}])
def test_tracer_content_is_conserved_with_partial_bottom_cells(pkg, kw):
    """What the flux form must do on the device, whatever the thicknesses: with no surface flux the volume integral of a tracer --
    the bottom cell of every column counted with ITS thickness DZBC, the surface layer with dz(1) + eta -- is unchanged to
    rounding (the oracle twin: tests/test_oracle_fixtures.py::test_advection_conserves_tracer_volume)."""
    cfg = named_config("tiny", stepped_bathymetry=1, partial_bottom_cells=1, **kw)
    m = pkg.PopModel(cfg)
    o = Oracle(cfg)
    dz = o.v1("dz")[1:cfg.km + 1].copy()
    o.close()
    tarea, kmt, dzbc = m.get("TAREA"), m.geti("KMT"), m.get("DZBC")
    k = np.arange(1, cfg.km + 1)[None, :, None, None]
    wet = (k <= kmt[:, None])[..., 2:-2, 2:-2]

    def content(n):
        T = m.get("TRACER", 1, n)[..., 2:-2, 2:-2]
        eta = m.get("PSURF", 1)[..., 2:-2, 2:-2] / 980.6
        thick = np.broadcast_to(dz[None, :, None, None], T.shape).copy()
        thick = np.where((k == kmt[:, None])[..., 2:-2, 2:-2], dzbc[:, None, 2:-2, 2:-2], thick)
        thick[:, 0] = thick[:, 0] + eta
        return float((np.where(wet, T * thick, 0.0) * tarea[:, None, 2:-2, 2:-2]).sum())

    for _ in range(3):
        m.step()
    c0 = [content(n) for n in (0, 1)]
    for _ in range(4):
        m.step()
    assert np.abs(m.get("UVEL", 1)).max() > 1.0
    for n in (0, 1):
        assert abs(content(n) - c0[n]) <= 2e-9 * abs(c0[n]), (kw, n, content(n), c0[n])
    m.close()


def test_partial_bottom_cells_at_a_quarter_of_tx01v3(pkg, orclib_built):
    """Size-independent properties at 1800 x 1200 x 62 (a quarter of the tx0.1v3 columns; KPP + biharmonic mixing + stepped
    bathymetry, the production kernel selection for large grids), where the oracle is out of reach:
      * geometry: DZUB is the minimum over the four surrounding T columns of DZT at level KMU (grid.F90:1003-1020);
      * with no surface flux the volume integral of each tracer, every bottom cell counted with ITS thickness, is unchanged
        to rounding over leapfrog steps;
      * the boundary layer never reaches below the column's own depth zw(KMT-1) + DZBC (vmix_kpp.F90:3835-3864);
      * land stays land: velocities below KMU and tracers below KMT are exactly zero."""
    cfg = named_config("tx0.1v3", nx_global=1800, ny_global=1200, block_size_x=1800, block_size_y=1200,
                       stepped_bathymetry=1, partial_bottom_cells=1)
    o = Oracle(named_config("tiny", km=cfg.km))            # the vertical grid depends on km only (bit-compared in test_host_grid_parity)
    dz, zw = o.v1("dz")[1:cfg.km + 1].copy(), o.v1("zw")[0:cfg.km + 1].copy()
    o.close()
    m = pkg.PopModel(cfg)
    km = cfg.km
    tarea, kmt, kmu, dzbc, dzub = m.get("TAREA"), m.geti("KMT"), m.geti("KMU"), m.get("DZBC"), m.get("DZUB")
    # geometry (every U point whose four T columns lie inside the block array)
    lev = kmu[:, :-1, :-1]
    quad = [q for q in (
        np.where(lev == kmt[:, :-1, :-1], dzbc[:, :-1, :-1], dz[np.clip(lev, 1, km) - 1]),
        np.where(lev == kmt[:, :-1, 1:], dzbc[:, :-1, 1:], dz[np.clip(lev, 1, km) - 1]),
        np.where(lev == kmt[:, 1:, :-1], dzbc[:, 1:, :-1], dz[np.clip(lev, 1, km) - 1]),
        np.where(lev == kmt[:, 1:, 1:], dzbc[:, 1:, 1:], dz[np.clip(lev, 1, km) - 1]))]
    expect = np.minimum(np.minimum(quad[0], quad[1]), np.minimum(quad[2], quad[3]))
    ocean_u = lev > 0
    assert ocean_u.sum() > 1000000
    assert np.array_equal(dzub[:, :-1, :-1][ocean_u], expect[ocean_u])
    assert (dzbc[kmt > 0] > 0.2 * dz[kmt[kmt > 0] - 1]).all() and (dzbc[kmt > 0] <= dz[kmt[kmt > 0] - 1]).all()
    assert (dzbc[kmt > 0] < dz[kmt[kmt > 0] - 1]).mean() > 0.5          # most bottom cells are partial
    k = np.arange(1, km + 1)[None, :, None, None]
    inner = (slice(None), slice(None), slice(2, -2), slice(2, -2))
    wet = (k <= kmt[:, None])[inner]
    bottom = (k == kmt[:, None])[inner]

    def content(n):
        T = m.get("TRACER", 1, n)[inner]
        thick = np.where(bottom, dzbc[:, None, 2:-2, 2:-2], dz[None, :, None, None])
        thick[:, 0] = thick[:, 0] + m.get("PSURF", 1)[:, 2:-2, 2:-2] / 980.6
        return float((np.where(wet, T * thick, 0.0) * tarea[:, None, 2:-2, 2:-2]).sum(dtype=np.longdouble))

    for _ in range(3):
        m.step()
    c0 = [content(n) for n in (0, 1)]
    for _ in range(4):
        m.step()
    for n in (0, 1):
        c1 = content(n)
        assert abs(c1 - c0[n]) <= 2e-9 * abs(c0[n]), (n, c1, c0[n])
    u = m.get("UVEL", 1)
    assert np.abs(u).max() > 0.1
    assert np.abs(np.where(k <= kmu[:, None], 0.0, u)).max() == 0.0
    del u
    t = m.get("TRACER", 1, 0)
    assert np.abs(np.where(k <= kmt[:, None], 0.0, t)).max() == 0.0
    del t
    depth = zw[np.clip(kmt, 1, km) - 1] + dzbc             # zw(KMT-1) + DZBC
    hblt = m.get("HBLT")
    sel = (kmt > 0)
    sel[:, :2, :] = False; sel[:, -2:, :] = False; sel[:, :, :2] = False; sel[:, :, -2:] = False
    assert (hblt[sel] <= depth[sel] * (1.0 + 1e-14)).all() and (hblt[sel] > 0.0).all()
    its = m.solver_diagnostics()[0]
    assert 0 < its < cfg.max_iterations
    m.close()


@pytest.mark.parametrize("kw", [{"vmix_choice": 3, "km": 62, **DEL4}, {"vmix_choice": 1, "km": 60}, {"vmix_choice": 3, "km": 60, "ldbl_diff": 1}],
                         ids=["kpp-km62-del4", "const-km60", "kpp-dd-km60"])
def test_register_thomas_kernels_with_partial_bottom_cells_are_bitwise_the_generic_ones(pkg, monkeypatch, kw):
    """km = 60 / 62: the column-in-registers instantiations k_impvmixt_reg / k_impvmixt2_reg / k_impvmixu_reg<..., PBC> against the
    scratch-staged kernels (POP_PBC_GENERIC_THOMAS=1), six steps on stepped bathymetry, every prognostic field to the last bit;
    with the barotropic sum fused into the velocity solve (POP_VMIXU_DEFER=1) and without."""
    cfg = named_config("tiny", stepped_bathymetry=1, partial_bottom_cells=1, **kw)
    out = {}
    for tag, env in (("reg", {}), ("reg-defer", {"POP_VMIXU_DEFER": "1"}), ("pair", {"POP_THOMAS_PAIR": "1"}), ("generic", {"POP_PBC_GENERIC_THOMAS": "1"})):
        for k in ("POP_VMIXU_DEFER", "POP_THOMAS_PAIR", "POP_PBC_GENERIC_THOMAS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = pkg.PopModel(cfg)
        for _ in range(6):
            m.step()
        out[tag] = [m.get(n, 1, 0).copy() for n in ("UVEL", "VVEL", "TRACER", "PSURF", "RHO")] + [m.get("TRACER", 1, 1).copy(), m.get("TRACER", 2, 0).copy()]
        m.close()
    assert np.abs(out["reg"][0]).max() > 1.0
    for tag in ("reg-defer", "pair", "generic"):
        for a, b in zip(out["reg"], out[tag]):
            assert np.array_equal(a, b), tag


@pytest.mark.parametrize("kw", [{"km": 62, **DEL4}, {"km": 60, "block_size_x": 48, "block_size_y": 40}, {"km": 24, "lrich": 0}],
                         ids=["km62-del4", "km60-one-block", "km24"])
def test_kpp_column_march_kernels_with_partial_bottom_cells_are_bitwise_the_generic_ones(pkg, orclib_built, monkeypatch, kw):
    """The production KPP kernel selection for large grids (shear column kernel with the level hint, buoydiff + interior as one
    column march, boundary-layer depth with the on-demand surface-layer buoyancy, sparse boundary-layer kernel) in their PBC
    instantiations against the 3-D-parallel / scratch-staged PBC kernels (POP_PBC_GENERIC_KPP=1), on stepped bathymetry with
    a bottom thickness that differs from column to column and a boundary layer several levels deep: every KPP output and the
    prognostic fields after five steps, to the last bit."""
    monkeypatch.setenv("POP_XCD_REMAP", "0")
    monkeypatch.setenv("POP_KPP_COL", "31")
    cfg = named_config("tiny", vmix_choice=3, stepped_bathymetry=1, partial_bottom_cells=1, **kw)
    out = {}
    for tag, env in (("march", {}), ("march-streaming", {"POP_KPP_SPARSE": "0"}), ("generic", {"POP_PBC_GENERIC_KPP": "1"})):
        for k in ("POP_KPP_SPARSE", "POP_PBC_GENERIC_KPP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m, orc = pkg.PopModel(cfg), Oracle(cfg)
        force_kpp_case(m, orc)
        orc.close()
        for _ in range(5):
            m.step()
        out[tag] = [m.get("HBLT").copy(), m.get("VDC", 1, 0).copy(), m.get("VVC").copy()] + [m.get("KPP_SRC", 1, n).copy() for n in (0, 1)] + \
                   [m.get(n, 1, 0).copy() for n in ("UVEL", "VVEL", "TRACER", "PSURF", "RHO")] + [m.get("TRACER", 1, 1).copy(), m.geti("KBL").copy()]
        m.close()
    h = out["march"][0]
    assert h.max() > 3.0 * h[h > 0].min()
    for tag in ("march-streaming", "generic"):
        for a, b in zip(out["march"], out[tag]):
            assert np.array_equal(a, b), tag


@pytest.mark.parametrize("kw", [{"vmix_choice": 3, "km": 62}, {"vmix_choice": 1, "km": 16, "time_mix_freq": 4}], ids=["kpp-km62", "const-km16-averaging-steps"])
def test_fused_del4_first_laplacians_with_partial_bottom_cells_are_bitwise_the_separate_kernels(pkg, monkeypatch, kw):
    """POP_D2T_FUSE=1 (the large-grid default): the tracer / momentum kernels of step n also form the first Laplacians step n + 1
    needs, with the thickness-scaled neighbour weights of hmix_del4.F90:683-697, 964-984, against k_del4_d2t / k_del4_d2u<PBC> as
    launches of their own: eight steps (Euler, leapfrog and averaging steps), every prognostic field to the last bit."""
    cfg = named_config("tiny", stepped_bathymetry=1, partial_bottom_cells=1, **DEL4, **kw)
    out = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("POP_D2T_FUSE", fuse)
        m = pkg.PopModel(cfg)
        assert m.dim("d2t_fused") == int(fuse) and m.dim("d2u_fused") == int(fuse)
        for _ in range(8):
            m.step()
        out[fuse] = [m.get(n, tl, 0).copy() for n in ("UVEL", "VVEL", "TRACER", "PSURF") for tl in (0, 1)] + [m.get("TRACER", 1, 1).copy()]
        m.close()
    assert np.abs(out["1"][0]).max() > 1.0
    for a, b in zip(out["1"], out["0"]):
        assert np.array_equal(a, b)
