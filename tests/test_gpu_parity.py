"""GPU parity (run on the MI355X box): every phase of the step_mod.F90 sequence through the C ABI
against the CPU oracle on the same inputs.

Tolerances (fp64):
  * phases without a global reduction (tracer/momentum tendencies, Thomas solves, state, halos):
    TOL_LOCAL = 1e-13 relative to the field's max -- both sides are compiled with
    -ffp-contract=off in the reference's evaluation order, so only library-function rounding
    (sqrt, division are correctly rounded; exp/atan differ by ulps) separates them;
  * anything downstream of the barotropic solver: TOL_SOLVE = 1e-8, because the GPU sums dot
    products in a fixed tree instead of the serial (j,i) order; the PCG iteration count must be
    IDENTICAL (north_star: "PCG iteration count unchanged").
"""
import numpy as np
import pytest

from popcfg import named_config, synthetic_grid, synthetic_dzbc
from orclib import Oracle

pytestmark = pytest.mark.gpu
TOL_LOCAL = 1e-13
TOL_SOLVE = 1e-8


def interior(a):
    return a[..., 2:-2, 2:-2]


def relerr(a, b):
    d = np.abs(a - b).max()
    s = np.abs(b).max()
    return d / s if s > 0 else d


def pick(gpu, a, inner):
    """the cells a comparison looks at: the physical cells (inner) or every cell of the block arrays; with padded blocks (block
    size not dividing the domain, tests/test_gpu_padded.py sets gpu.masks) the physical cells of each block, resp. the cells that
    exist and whose neighbours exist (the padding holds nothing, and a ghost cell next to it is formed from it)"""
    m = getattr(gpu, "masks", None)
    if m is None:
        return interior(a) if inner else a
    sel = m["phys"] if inner else m["near"]
    if a.ndim == 4:
        sel = np.broadcast_to(sel[:, None], a.shape)
    return a[sel]


def check(gpu, orc, name, tol, tl=1, n=0, three_d=True, inner=True, what=""):
    a = gpu.get(name, tl, n)
    b = (orc.f3 if three_d else orc.f2)(name, tl, n)
    a, b = pick(gpu, a, inner), pick(gpu, b, inner)
    e = relerr(a, b)
    assert e <= tol, "%s %s(tl=%d,n=%d): rel err %.3e > %.1e" % (what, name, tl, n, e, tol)
    return e


def run_phases(gpu, orc, step, tol_state):
    L = orc.L
    gpu.time_manager(); L.orc_time_manager(orc.h)
    assert gpu.dim("leapfrogts") == orc.dim("leapfrogts") and gpu.dim("avg_ts") == orc.dim("avg_ts")
    gpu.dhdt(); L.orc_dhdt(orc.h)
    w = "step %d dhdt" % step
    check(gpu, orc, "DH", tol_state, three_d=False, inner=False, what=w)
    check(gpu, orc, "DHU", tol_state, three_d=False, inner=False, what=w)
    gpu.baroclinic_driver(); L.orc_baroclinic_driver(orc.h)
    w = "step %d baroclinic_driver" % step
    for n in (0, 1):
        check(gpu, orc, "TRACER", tol_state, tl=2, n=n, what=w)
    check(gpu, orc, "UVEL", tol_state, tl=2, what=w)
    check(gpu, orc, "VVEL", tol_state, tl=2, what=w)
    check(gpu, orc, "ZX", tol_state, three_d=False, what=w)
    check(gpu, orc, "ZY", tol_state, three_d=False, what=w)
    check(gpu, orc, "VVC", tol_state, what=w)
    if gpu.cfg.vmix_choice == 3:
        a, b = gpu.get("VDC", n=0), orc.vdc(0)
        assert relerr(pick(gpu, a, True), pick(gpu, b, True)) <= tol_state * 10, "%s VDC(T): %g" % (w, relerr(pick(gpu, a, True), pick(gpu, b, True)))
        a, b = gpu.get("VDC", n=1), orc.vdc(1)
        assert relerr(pick(gpu, a, True), pick(gpu, b, True)) <= tol_state * 10, "%s VDC(S)" % w
        check(gpu, orc, "HBLT", tol_state * 10, three_d=False, what=w)
        for n in (0, 1):
            check(gpu, orc, "KPP_SRC", tol_state * 100, n=n, what=w)
    gpu.barotropic_driver(); assert L.orc_barotropic_driver(orc.h) == 0
    w = "step %d barotropic_driver" % step
    it_g, rms_g = gpu.solver_diagnostics()
    assert it_g == L.orc_solver_iterations(orc.h), "%s: PCG iterations %d vs oracle %d" % (w, it_g, L.orc_solver_iterations(orc.h))
    check(gpu, orc, "RHS", max(tol_state, TOL_LOCAL * 10), three_d=False, inner=False, what=w)
    for f in ("PSURF", "GRADPX", "GRADPY", "UBTROP", "VBTROP"):
        check(gpu, orc, f, TOL_SOLVE, tl=2, three_d=False, inner=(f in ("UBTROP", "VBTROP")), what=w)
    gpu.baroclinic_correct_adjust(); L.orc_baroclinic_correct_adjust(orc.h)
    w = "step %d correct_adjust" % step
    for n in (0, 1):
        check(gpu, orc, "TRACER", TOL_SOLVE, tl=2, n=n, what=w)
    check(gpu, orc, "RHO", TOL_SOLVE, tl=2, what=w)
    gpu.step_tail(); L.orc_step_tail(orc.h)
    w = "step %d tail" % step
    for tl in (0, 1):
        for n in (0, 1):
            check(gpu, orc, "TRACER", TOL_SOLVE, tl=tl, n=n, inner=False, what=w)
        for f in ("UVEL", "VVEL", "RHO"):
            check(gpu, orc, f, TOL_SOLVE, tl=tl, inner=False, what=w)
        for f in ("PSURF", "GRADPX", "GRADPY", "UBTROP", "VBTROP"):
            check(gpu, orc, f, TOL_SOLVE, tl=tl, three_d=False, inner=False, what=w)
    check(gpu, orc, "PGUESS", TOL_SOLVE, three_d=False, inner=False, what=w)
    return it_g


def force_kpp_case(gpu, orc):
    """Surface buoyancy forcing + a weakly stratified upper ocean so the KPP boundary layer spans
    several levels (bulk-Richardson interpolation, shape functions, non-local source all active)."""
    tlat = orc.f2("TLAT")
    stf_t = -3.0e-2 * np.sin(tlat) - 1.0e-2          # degC cm/s: cooling (unstable) in the north
    stf_s = 2.0e-6 * np.cos(2.0 * tlat)
    kmt = orc.i2("KMT")
    for tl in (0, 1, 2):
        for n, slope in ((0, 2.0e-4), (1, -2.0e-9)):
            T = orc.f3("TRACER", tl, n)
            z = np.arange(T.shape[1])[None, :, None, None]
            mix = T[:, 0:1] - slope * z                   # nearly homogeneous upper ocean
            shallow = (z < 8) & (z < kmt[:, None]) & (np.sin(3 * tlat)[:, None] > 0)
            T[...] = np.where(shallow, mix, T)
            gpu.set("TRACER", T, tl=tl, n=n)
    orc.f2("STF", 1, 0)[...] = stf_t
    orc.f2("STF", 1, 1)[...] = stf_s
    gpu.set("STF", stf_t, n=0); gpu.set("STF", stf_s, n=1)
    # densities of the modified state (init_ts computes RHO for cur and old)
    import ctypes as C
    for tl in (0, 1):
        T, S, R = orc.f3("TRACER", tl, 0), orc.f3("TRACER", tl, 1), orc.f3("RHO", tl)
        P = C.POINTER(C.c_double)
        orc.L.orc_state.argtypes = [C.c_void_p, C.c_int, C.c_int, P, P, P, P, P, C.c_int]
        for k in range(orc.km):
            t = np.ascontiguousarray(T[:, k]); s_ = np.ascontiguousarray(S[:, k]); r = np.empty_like(t)
            orc.L.orc_state(orc.h, k + 1, k + 1, t.ctypes.data_as(P), s_.ctypes.data_as(P), r.ctypes.data_as(P), None, None, t.size)
            R[:, k] = r
        gpu.set("RHO", R, tl=tl)


@pytest.mark.parametrize("name,kw,nsteps", [
    ("tiny", {}, 4),                                   # 16 blocks, const vmix, avgfit (step 2 averages)
    ("tiny", {"solver_choice": 2}, 4),                 # ChronGear
    ("tiny", {"vmix_choice": 2}, 4),                   # Richardson vmix
    ("tiny", {"lpressure_avg": 0, "tmix_opt": 1, "time_mix_freq": 3}, 4),
    ("tiny", {"block_size_x": 48, "block_size_y": 40}, 3),   # one block
    ("gx3v7", {}, 3),
    ("tiny", {"hmix_momentum": 4, "hmix_tracer": 4, "lvariable_hmix": 1, "am": -1.0e22, "ah": -1.0e21}, 4),   # del4
    ("tiny", {"hmix_momentum": 4, "hmix_tracer": 2, "am": -1.0e22}, 3),                                      # mixed
    ("tiny", {"vmix_choice": 3, "km": 24}, 5),                       # KPP
    ("tiny", {"vmix_choice": 3, "km": 24, "ldbl_diff": 1}, 5),       # KPP + double diffusion (CESM default)
    ("tiny", {"tadvect": 2}, 5),                                     # third-order upwind tracer advection
    ("tiny", {"tadvect": 2, "vmix_choice": 3, "km": 24, "hmix_tracer": 4, "hmix_momentum": 4, "am": -1.0e22, "ah": -1.0e21}, 4),
    ("gx3v7", {"tadvect": 2}, 3),
    ("test", {}, 10),                                                # BASELINE configs[0]: test_domain_size, 96 blocks, 10 steps
    ("tiny", {"ew_boundary": 0}, 3),                                 # closed east-west boundary
    ("tiny", {"impcor": 0, "lbouss_correct": 1, "reset_to_freezing": 0, "tmix_opt": 0}, 3),
    ("tiny", {"solver_choice": 3}, 4),                               # P-CSI (no inner product per iteration)
    ("gx3v7", {"solver_choice": 3}, 3),
    ("tiny", {"solver_choice": 3, "block_size_x": 48, "block_size_y": 40, "vmix_choice": 3, "km": 24}, 4),
    ("tiny", {"precond_choice": 1}, 4),                              # EVP block preconditioner, pcg
    ("tiny", {"precond_choice": 1, "solver_choice": 2, "block_size_x": 16, "block_size_y": 20}, 4),   # ChronGear + EVP, 8/6/6 pieces
    ("tiny", {"precond_choice": 1, "solver_choice": 3}, 4),          # P-CSI + EVP (Lanczos through the preconditioner)
    ("gx3v7", {"precond_choice": 1}, 3),
    ("test", {"precond_choice": 1, "solver_choice": 3}, 4),
    ("tiny", {"tmix_opt": 3}, 5),                                    # Robert-Asselin-Williams filter (alpha 0.53, nu 0.2)
    ("tiny", {"tmix_opt": 3, "robert_alpha": 1.0, "vmix_choice": 3, "km": 24}, 5),   # classic Robert-Asselin: previous-step averaging
    # stepped synthetic bathymetry (KMT = 3 ... km in stairs): every k > KMT / k > KMU branch, shallow columns in the Thomas solves
    ("tiny", {"stepped_bathymetry": 1}, 4),
    ("tiny", {"stepped_bathymetry": 1, "vmix_choice": 3, "km": 24, "ldbl_diff": 1}, 5),
    ("tiny", {"stepped_bathymetry": 1, "vmix_choice": 2, "hmix_momentum": 4, "hmix_tracer": 4, "lvariable_hmix": 1, "am": -1.0e22, "ah": -1.0e21}, 4),
    ("tiny", {"stepped_bathymetry": 1, "tadvect": 2, "solver_choice": 2}, 4),
    ("tiny", {"stepped_bathymetry": 1, "km": 60, "vmix_choice": 3}, 3),                  # register Thomas kernels with shallow columns
    ("gx3v7", {"stepped_bathymetry": 1, "solver_choice": 3}, 3),
    # tadvect = 3: Lax-Wendroff advection with one-dimensional flux limiters ('lw_lim', advection.F90:2684-3280)
    ("tiny", {"tadvect": 3}, 5),
    ("tiny", {"tadvect": 3, "stepped_bathymetry": 1, "vmix_choice": 3, "km": 24}, 5),
    ("tiny", {"tadvect": 3, "hmix_momentum": 4, "hmix_tracer": 4, "lvariable_hmix": 1, "am": -1.0e22, "ah": -1.0e21, "vmix_choice": 2}, 4),
    ("tiny", {"tadvect": 3, "block_size_x": 48, "block_size_y": 40, "ew_boundary": 0}, 4),       # one block, closed east-west
    ("gx3v7", {"tadvect": 3}, 3),
    ("test", {"tadvect": 3, "stepped_bathymetry": 1}, 4),                                         # 96 blocks
])
def test_step_phases_match_oracle(pkg, orclib_built, name, kw, nsteps):
    cfg = named_config(name, **kw)
    gpu, orc = pkg.PopModel(cfg), Oracle(cfg)
    # identical initial state by construction (both build it from the same formulas)
    for n in (0, 1):
        assert np.array_equal(gpu.get("TRACER", 1, n), orc.f3("TRACER", 1, n))
    assert relerr(gpu.get("RHO", 1), orc.f3("RHO", 1)) < 1e-15
    if cfg.vmix_choice == 3:
        force_kpp_case(gpu, orc)
    tol = TOL_LOCAL
    for s in range(1, nsteps + 1):
        run_phases(gpu, orc, s, tol)
        tol = TOL_SOLVE        # later steps inherit the solver's summation-order difference
    gpu.close(); orc.close()


@pytest.mark.parametrize("kw,nsteps,env", [
    # tripole northern boundary (ns_boundary = 2) on a grid supplied by the caller: every halo update of the step carries
    # the reference's fieldLoc / fieldKind (U-grid vectors change sign beyond the fold and have their top row symmetrised)
    ({"ns_boundary": 2}, 5, {}),                                                        # 16 blocks, fused pcg (fold inside srcmap)
    ({"ns_boundary": 2}, 4, {"POP_SOLVER_UNFUSED": "1"}),                               # per-operation solver: tripole halo passes
    ({"ns_boundary": 2, "block_size_x": 48, "block_size_y": 40, "solver_choice": 2}, 4, {}),   # one block, ChronGear
    ({"ns_boundary": 2, "vmix_choice": 3, "km": 24, "ldbl_diff": 1}, 5, {}),            # KPP
    ({"ns_boundary": 2, "hmix_momentum": 4, "hmix_tracer": 4, "lvariable_hmix": 1, "am": -1.0e22, "ah": -1.0e21, "tadvect": 2}, 4, {}),
    ({"ns_boundary": 2, "solver_choice": 3}, 4, {}),                                    # P-CSI (Lanczos through the fold)
    ({"ns_boundary": 2, "precond_choice": 1, "solver_choice": 2}, 4, {}),               # EVP preconditioner
    ({"ns_boundary": 2, "tmix_opt": 3, "vmix_choice": 2}, 5, {}),                       # Robert filter
    ({"ns_boundary": 2, "km": 60, "vmix_choice": 3, "block_size_x": 24, "block_size_y": 20}, 3, {}),
    # lw_lim across the fold: UTE travels as an E-face vector, VTN takes its ghost rows as an N-face vector WITHOUT the symmetrised
    # top row (the reference forms VTN there from the mirrored velocities: the same numbers)
    # (without wind: the analytic zonal wind is not antisymmetric under the fold, the symmetrised top row then flips signs from
    # update to update, and the limiter's min / max switches turn the solver's 1e-9 summation-order difference into 1.1e-8 in ZX)
    ({"ns_boundary": 2, "tadvect": 3}, 5, {"calm": "1"}),
    ({"ns_boundary": 2, "tadvect": 3, "vmix_choice": 3, "km": 24, "block_size_x": 48, "block_size_y": 40}, 4, {"calm": "1"}),
    # the same grid arrays under ordinary boundaries (horiz_grid_opt / topography_opt = 'file' without a fold)
    ({"ns_boundary": 0}, 3, {}),
    ({"ns_boundary": 0, "ew_boundary": 0, "vmix_choice": 3, "km": 24}, 3, {}),
    ({"ns_boundary": 1, "tadvect": 3}, 3, {}),
])
def test_grid_input_step_phases_match_oracle(pkg, orclib_built, monkeypatch, kw, nsteps, env):
    env = dict(env)
    calm = env.pop("calm", None)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cfg = named_config("tiny", **kw)
    grid = synthetic_grid(cfg)
    gpu, orc = pkg.PopModel(cfg, grid=grid), Oracle(cfg, grid=grid)
    for n in (0, 1):
        assert np.array_equal(gpu.get("TRACER", 1, n), orc.f3("TRACER", 1, n))
    if calm:
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
    gpu.close(); orc.close()


@pytest.mark.parametrize("name,kw", [("tiny", {"km": 24}), ("tiny", {"km": 24, "stepped_bathymetry": 1, "ldbl_diff": 1}), ("gx3v7", {})])
def test_kpp_mixed_layer_depth_diagnostics(pkg, orclib_built, name, kw):
    """HMXL (depth of the maximum buoyancy gradient) and HMXL_DR (0.03 kg/m^3 density criterion), vmix_kpp.F90:1310-1418,
    computed every step when kpp_ml_diagnostics = 1: every cell of every block against the oracle"""
    cfg = named_config(name, vmix_choice=3, kpp_diagnostics=1, **kw)
    gpu, orc = pkg.PopModel(cfg), Oracle(cfg)
    force_kpp_case(gpu, orc)
    tol = TOL_LOCAL
    for s in range(1, 4):
        run_phases(gpu, orc, s, tol)
        tol = TOL_SOLVE
        for f in ("HMXL", "HMXL_DR"):
            a, b = gpu.get(f), orc.f2(f)
            assert np.abs(b).max() > 1.0e3 and len(np.unique(np.round(b, 3))) > 20, f + ": trivial field"
            assert relerr(a, b) <= tol * 100, "%s step %d: %g" % (f, s, relerr(a, b))
    gpu.close(); orc.close()
    off = pkg.PopModel(named_config(name, vmix_choice=3, **kw))
    off.step()
    assert not off.get("HMXL").any()          # not computed unless asked for
    off.close()


@pytest.mark.parametrize("kw", [
    {"lshort_wave": 1},                                        # sw_absorption_type 'top-layer': BFSFC = BO + BOSOL
    {"lshort_wave": 1, "sw_absorption_type": 1},                # 'jerlov', water type IB (the CESM default 3)
    {"lshort_wave": 1, "sw_absorption_type": 1, "jerlov_water_type": 5, "ldbl_diff": 1, "stepped_bathymetry": 1},
    {"lshort_wave": 1, "sw_absorption_type": 2},                # 'chlorophyll' (the CESM default): transmission table look-up, CHL varies in space
    {"lshort_wave": 1, "sw_absorption_type": 2, "lcheckekmo": 1, "stepped_bathymetry": 1},
    {"lcheckekmo": 1},                                         # Ekman / Monin-Obukhov depth limits under stable forcing
    {"lcheckekmo": 1, "lshort_wave": 1, "sw_absorption_type": 1, "block_size_x": 48, "block_size_y": 40},
])
def test_kpp_short_wave_and_depth_limits(pkg, orclib_built, kw):
    """vmix_kpp_nml lshort_wave (bldepth :2236-2256, :2387-2412, :2707-2742 with sw_absorb_frac, sw_absorption.F90:736-811) and
    lcheckekmo (:2231-2265, :2426-2455, :2676-2690): phase parity with a short-wave flux that varies with latitude; the options
    must also change the boundary layer depth (so the branches are live)"""
    cfg = named_config("tiny", vmix_choice=3, km=24, **kw)
    ref = named_config("tiny", vmix_choice=3, km=24, **{k: v for k, v in kw.items() if k in ("ldbl_diff", "stepped_bathymetry", "block_size_x", "block_size_y")})
    gpu, orc, plain = pkg.PopModel(cfg), Oracle(cfg), pkg.PopModel(ref)
    force_kpp_case(gpu, orc)
    tlat = orc.f2("TLAT")
    qsw = 5.0e-3 * (1.0 + np.cos(tlat))          # degC cm/s
    orc.f2("SHF_QSW")[...] = qsw
    gpu.set("SHF_QSW", qsw); plain.set("SHF_QSW", qsw)
    chl = 0.003 + 4.0 * np.abs(np.sin(3.0 * tlat)) ** 3      # mg/m^3: both ends of the table's range
    if kw.get("sw_absorption_type") == 2:
        assert np.all(gpu.get("CHL") == 0.25)                # the default until set
        orc.f2("CHL")[...] = chl; gpu.set("CHL", chl)
    for tl in (0, 1, 2):
        for n in (0, 1):
            plain.set("TRACER", orc.f3("TRACER", tl, n), tl=tl, n=n)
    for n in (0, 1):
        plain.set("STF", orc.f2("STF", 1, n), n=n)
    for tl in (0, 1):
        plain.set("RHO", orc.f3("RHO", tl), tl=tl)
    tol = TOL_LOCAL
    for s in range(1, 4):
        run_phases(gpu, orc, s, tol)
        tol = TOL_SOLVE
    plain.step()
    gpu2 = pkg.PopModel(cfg)     # first step of the option against the first step without it
    gpu.close(); orc.close()
    hb_plain = plain.get("HBLT")
    plain.close()
    o2 = Oracle(cfg)
    force_kpp_case(gpu2, o2)
    o2.f2("SHF_QSW")[...] = qsw; gpu2.set("SHF_QSW", qsw)
    if kw.get("sw_absorption_type") == 2:
        gpu2.set("CHL", chl)
    gpu2.step()
    assert np.abs(gpu2.get("HBLT") - hb_plain).max() > 1.0, "the option did not change the boundary layer depth"
    gpu2.close(); o2.close()


@pytest.mark.parametrize("kw", [
    {"lsw_absorb": 1, "sw_absorption_type": 0},                                  # all of it in the top level (const vmix, LDS tracer kernel)
    {"lsw_absorb": 1, "sw_absorption_type": 1, "jerlov_water_type": 2, "stepped_bathymetry": 1},
    {"lsw_absorb": 1, "sw_absorption_type": 2, "tadvect": 2},                     # chlorophyll table, direct-load tracer kernel (upwind3)
    {"lsw_absorb": 1, "sw_absorption_type": 2, "vmix_choice": 3, "km": 24, "lshort_wave": 1, "stepped_bathymetry": 1},   # the CESM combination
    {"lsw_absorb": 1, "sw_absorption_type": 1, "tadvect": 3, "lpressure_avg": 0},
])
def test_penetrating_short_wave_source(pkg, orclib_built, kw):
    """add_sw_absorb (sw_absorption.F90:818-947, called from tracer_update next to add_kpp_sources): the short-wave flux
    SHF_QSW heats level k with its absorbed share, everything that reaches the bottom level stays there"""
    cfg = named_config("tiny", **kw)
    gpu, orc = pkg.PopModel(cfg), Oracle(cfg)
    if cfg.vmix_choice == 3:
        force_kpp_case(gpu, orc)
    tlat = orc.f2("TLAT")
    qsw = 6.0e-3 * np.cos(tlat) - 1.0e-3          # degC cm/s, negative near the poles (clipped to 0 by the routine)
    orc.f2("SHF_QSW")[...] = qsw; gpu.set("SHF_QSW", qsw)
    if kw["sw_absorption_type"] == 2:
        chl = 0.003 + 4.0 * np.abs(np.sin(3.0 * tlat)) ** 3
        orc.f2("CHL")[...] = chl; gpu.set("CHL", chl)
    t0 = gpu.get("TRACER", 1, 0).copy()
    tol = TOL_LOCAL
    for s in range(1, 4):
        run_phases(gpu, orc, s, tol)
        tol = TOL_SOLVE
    off = pkg.PopModel(named_config("tiny", **{k: v for k, v in kw.items() if k != "lsw_absorb"}))
    if cfg.vmix_choice == 3:
        o2 = Oracle(cfg); force_kpp_case(off, o2); o2.close()
    off.set("SHF_QSW", qsw)
    for _ in range(3):
        off.step()
    d = gpu.get("TRACER", 1, 0)[:, 0] - off.get("TRACER", 1, 0)[:, 0]
    assert d.max() > 1.0e-3, d.max()                                         # the surface level warms where the sun shines
    if cfg.vmix_choice != 3:                                                 # (KPP mixes the extra heat down: no sign for every cell)
        assert d.min() > -1.0e-6, d.min()
    gpu.close(); orc.close(); off.close()


def test_tripole_without_grid_input_refuses_to_step(pkg):
    m = pkg.PopModel(named_config("tiny", ns_boundary=2))
    with pytest.raises(pkg.PopError, match="pop_create_with_grid"):
        m.step()
    m.close()


@pytest.mark.parametrize("kw,nsteps", [
    ({}, 3),
    ({"block_size_x": 1056}, 3),                                           # two blocks side by side
    ({"vmix_choice": 3, "hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e19, "ah": -1.0e18}, 3),   # KPP + del4
])
def test_large_grid_tile_order_matches_oracle(pkg, orclib_built, monkeypatch, kw, nsteps):
    """POP_XCD_REMAP=2 forces the tile-column workgroup order production uses above 2^19 columns
    (kernels_common.hpp) on a grid small enough for the oracle: full 8-column groups, left-over
    columns and surplus workgroups all occur (2116 = 33 x 64 + 4 = 8 x 256 + 68)."""
    monkeypatch.setenv("POP_XCD_REMAP", "2")
    monkeypatch.setenv("POP_RED_TILES", "1")
    cfg = named_config("wide", **kw)
    gpu, orc = pkg.PopModel(cfg), Oracle(cfg)
    if cfg.vmix_choice == 3:
        force_kpp_case(gpu, orc)
    tol = TOL_LOCAL
    for s in range(1, nsteps + 1):
        run_phases(gpu, orc, s, tol)
        tol = TOL_SOLVE
    gpu.close(); orc.close()


def test_state_known_answer_and_derivatives(pkg, orclib_built):
    """state_mod.F90:413-414: rho(S=35 psu, theta=20 C, p=200 bar) -- the comment quotes
    1.033213242; the reference's own coefficient set evaluates to 1.0332133866 (see
    tests/test_oracle_fixtures.py), which the GPU must reproduce to rounding."""
    cfg = named_config("tiny")
    gpu, orc = pkg.PopModel(cfg), Oracle(cfg)
    rng = np.random.default_rng(7)
    T = rng.uniform(-3.0, 32.0, 4096); S = rng.uniform(-0.001, 0.042, 4096)
    for kk in (1, cfg.km // 2, cfg.km):
        rho, dt, ds = gpu.state(kk, T, S, derivs=True)
        ro, dto, dso = np.empty_like(T), np.empty_like(T), np.empty_like(T)
        import ctypes as C
        P = C.POINTER(C.c_double)
        orc.L.orc_state.argtypes = [C.c_void_p, C.c_int, C.c_int, P, P, P, P, P, C.c_int]
        orc.L.orc_state(orc.h, kk, kk, T.ctypes.data_as(P), S.ctypes.data_as(P), ro.ctypes.data_as(P), dto.ctypes.data_as(P),
                        dso.ctypes.data_as(P), T.size)
        assert relerr(rho, ro) < 1e-15 and relerr(dt, dto) < 1e-13 and relerr(ds, dso) < 1e-13
    gpu.close(); orc.close()


def test_halo_update_rule_on_device(pkg):
    """test/unit/halo/POP.F90Dipole:134-147,277-292: fill array(i,j)=iGlobal+jGlobal on the physical
    domain, ghosts = -999, update; every ghost must equal i_glob+j_glob of its global source, or 0
    outside closed boundaries."""
    cfg = named_config("tiny")
    gpu = pkg.PopModel(cfg)
    nb, ny, nx, km = gpu.nblocks, gpu.nyb, gpu.nxb, gpu.km
    a2 = np.full((nb, ny, nx), -999.0); a3 = np.full((nb, km, ny, nx), -999.0)
    exp2 = np.zeros_like(a2)
    for b in range(nb):
        blk = gpu.get_block(b + 1)
        ig, jg = blk["i_glob"], blk["j_glob"]
        g = ig[None, :] + jg[:, None]
        valid = (ig[None, :] > 0) & (jg[:, None] > 0)
        exp2[b] = np.where(valid, g, 0.0)
        a2[b, 2:-2, 2:-2] = g[2:-2, 2:-2]
        for k in range(km):
            a3[b, k, 2:-2, 2:-2] = g[2:-2, 2:-2] + 1000.0 * k
    gpu.set("PSURF", a2, tl=2); gpu.halo_update("PSURF", tl=2)
    assert np.array_equal(gpu.get("PSURF", tl=2), exp2)
    for n in (0, 1):   # 4-D update: all tracers in one call
        gpu.set("TRACER", a3 + 7.0 * n, tl=2, n=n)
    gpu.halo_update("TRACER", tl=2, n=-1)
    for n in (0, 1):
        got = gpu.get("TRACER", tl=2, n=n)
        for k in range(km):
            e = np.where(exp2 != 0, exp2 + 1000.0 * k + 7.0 * n, 0.0)
            e[:, 2:-2, 2:-2] = exp2[:, 2:-2, 2:-2] + 1000.0 * k + 7.0 * n
            assert np.array_equal(got[:, k], e), (n, k)
    gpu.set("UVEL", a3, tl=2); gpu.halo_update("UVEL", tl=2)
    got = gpu.get("UVEL", tl=2)
    for k in range(km):
        e = np.where(exp2 != 0, exp2 + 1000.0 * k, 0.0)
        e[:, 2:-2, 2:-2] = exp2[:, 2:-2, 2:-2] + 1000.0 * k
        assert np.array_equal(got[:, k], e), k
    gpu.close()


@pytest.mark.parametrize("loc,kind", [("center", "scalar"), ("center", "vector"), ("NEcorner", "vector"), ("NEcorner", "scalar"),
                                      ("Nface", "scalar"), ("Eface", "vector")])
def test_tripole_halo_on_device(pkg, loc, kind):
    """Device halo update on a tripole decomposition against the rule of test/unit/halo/POP.F90Tripole
    (2-D and 3-D fields); time stepping on such a decomposition is refused (no tripole grid)."""
    from test_oracle_fixtures import _tripole_expected
    cfg = named_config("tiny", ns_boundary=2)
    m = pkg.PopModel(cfg)
    nx, ny, nb = cfg.nx_global, cfg.ny_global, m.nblocks
    rng = np.random.default_rng(9)
    G = rng.standard_normal((ny, nx)) * 100.0
    a2 = np.full((nb, m.nyb, m.nxb), -999.0); a3 = np.full((nb, m.km, m.nyb, m.nxb), -999.0)
    e2 = np.zeros_like(a2); e3 = np.zeros_like(a3)
    for b in range(nb):
        blk = m.get_block(b + 1)
        ig, jg = blk["i_glob"], blk["j_glob"]
        for j in range(2, m.nyb - 2):
            a2[b, j, 2:-2] = G[jg[j] - 1, ig[2:-2] - 1]
        e2[b] = _tripole_expected(G, ig, jg, nx, ny, loc, kind)
        for k in range(m.km):
            a3[b, k, 2:-2, 2:-2] = a2[b, 2:-2, 2:-2] * (k + 1)
            e3[b, k] = _tripole_expected(G * (k + 1), ig, jg, nx, ny, loc, kind)
    m.set("PSURF", a2, tl=2); m.halo_update_loc("PSURF", tl=2, loc=loc, kind=kind)
    assert np.array_equal(m.get("PSURF", tl=2), e2)
    m.set("UVEL", a3, tl=2); m.halo_update_loc("UVEL", tl=2, loc=loc, kind=kind)
    assert np.array_equal(m.get("UVEL", tl=2), e3)
    with pytest.raises(pkg.PopError, match="tripole"):
        m.step()
    m.close()


def test_global_count_and_extremes(pkg):
    """test/unit/reduction/POP.F90: global count, maxval / minval and maxloc / minloc against a serial loop over
    the physical domain, with and without a mask."""
    cfg = named_config("tiny")
    m = pkg.PopModel(cfg)
    rng = np.random.default_rng(33)
    a = rng.standard_normal((m.nblocks, m.nyb, m.nxb)) * 7.0
    a[rng.random(a.shape) < 0.3] = 0.0
    m.set("RHS", a)
    phys = interior(a)
    assert m.global_count("RHS") == int(np.count_nonzero(phys))
    mask = interior(m.get("mMask")) != 0.0
    for want_max in (True, False):
        for use_mask in (False, True):
            sel = np.where(mask, phys, -np.inf if want_max else np.inf) if use_mask else phys
            ref = sel.max() if want_max else sel.min()
            b, j, i = np.unravel_index(sel.argmax() if want_max else sel.argmin(), sel.shape)
            blk = m.get_block(b + 1)
            v, ig, jg = m.global_extreme("RHS", mask="mMask" if use_mask else None, want_max=want_max)
            assert v == ref and (ig, jg) == (blk["i_glob"][i + 2], blk["j_glob"][j + 2])
    m.close()


def test_tripole_global_sum(pkg, orclib_built):
    """mpi/POP_ReductionsMod.F90:308-341: on a tripole grid N-face / NE-corner fields count the redundant half
    of the top row once; centre / E-face fields are summed as usual."""
    import ctypes as C
    cfg = named_config("tiny", ns_boundary=2)
    m, o = pkg.PopModel(cfg), Oracle(cfg)
    rng = np.random.default_rng(21)
    a = rng.standard_normal((m.nblocks, m.nyb, m.nxb)) * 50.0
    m.set("RHS", a)
    P = C.POINTER(C.c_double)
    ac = np.ascontiguousarray(a); mk = np.ascontiguousarray(o.f2("mMask"))
    scale = np.abs(interior(a)).sum()
    for loc, li in (("NEcorner", 1), ("Nface", 2), ("center", 0)):
        ref = o.L.orc_global_sum_tripole(o.h, ac.ctypes.data_as(P), None, li)
        refm = o.L.orc_global_sum_tripole(o.h, ac.ctypes.data_as(P), mk.ctypes.data_as(P), li)
        assert abs(m.global_sum_loc("RHS", loc=loc) - ref) <= 1e-12 * scale
        assert abs(m.global_sum_loc("RHS", mask="mMask", loc=loc) - refm) <= 1e-12 * scale
    plain = o.L.orc_global_sum_tripole(o.h, ac.ctypes.data_as(P), None, 0)
    dup = o.L.orc_global_sum_tripole(o.h, ac.ctypes.data_as(P), None, 1)
    assert plain != dup                 # the redundant points really are removed
    m.close(); o.close()


def test_global_sum_matches_serial_rule(pkg, orclib_built):
    """test/unit/reduction/POP.F90: global sum of a known array (with and without mMask) against a
    serial loop over the physical domain; the GPU tree sum must agree to 1e-14 relative."""
    cfg = named_config("tiny")
    gpu, orc = pkg.PopModel(cfg), Oracle(cfg)
    rng = np.random.default_rng(3)
    a = rng.standard_normal((gpu.nblocks, gpu.nyb, gpu.nxb)) * 1e3
    gpu.set("RHS", a)
    ref_nomask = interior(a).sum()
    mask = orc.f2("mMask")
    got = gpu.global_sum("RHS"); assert abs(got - ref_nomask) <= 1e-12 * np.abs(interior(a)).sum()
    got = gpu.global_sum("RHS", mask="mMask")
    assert abs(got - (interior(a) * interior(mask)).sum()) <= 1e-12 * np.abs(interior(a)).sum()
    gpu.close(); orc.close()


@pytest.mark.parametrize("kw", [{}, {"vmix_choice": 3, "km": 24, "hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e22, "ah": -1.0e21},
                                {"block_size_x": 48, "block_size_y": 40}])
def test_momentum_kernels_agree_bitwise(pkg, monkeypatch, kw):
    """The LDS-tiled momentum and tracer right-hand-side kernels (64x8 and 64x4 tiles) and the direct-load
    kernels evaluate the same expressions in the same order: results must be identical to the last bit."""
    cfg = named_config("tiny", **kw)
    out = {}
    for rows in ("8", "4", "0"):
        monkeypatch.setenv("POP_MOMENTUM_LDS", rows)
        monkeypatch.setenv("POP_TRACER_LDS", rows)
        m = pkg.PopModel(cfg)
        for _ in range(3):
            m.step()
        out[rows] = [m.get(n, 1, 0).copy() for n in ("UVEL", "VVEL", "TRACER", "PSURF")]
        m.close()
    for rows in ("4", "0"):
        for a, b in zip(out["8"], out[rows]):
            assert np.array_equal(a, b), rows


@pytest.mark.parametrize("kw,env", [
    ({"vmix_choice": 3, "km": 62, "stepped_bathymetry": 1, "hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e22, "ah": -1.0e21}, {"POP_KPP_COL": "15", "POP_XCD_REMAP": "0"}),
    ({"vmix_choice": 3, "km": 24, "lshort_wave": 1, "sw_absorption_type": 1}, {"POP_KPP_COL": "15"}),
    ({"vmix_choice": 3, "km": 60, "block_size_x": 48, "block_size_y": 40}, {"POP_KPP_COL": "15", "POP_KPP_SIDE_STREAM": "0"}),
    ({"vmix_choice": 3, "km": 24, "stepped_bathymetry": 1}, {}),                      # small grids: the 3-D buoydiff kernel without DBSFC
    ({"vmix_choice": 3, "km": 60}, {"POP_KPP_COL": "1", "POP_XCD_REMAP": "1"}),       # the gx1v7 kernel selection
])
def test_kpp_on_demand_surface_buoyancy_is_bitwise_invisible(pkg, orclib_built, monkeypatch, kw, env):
    """k_kpp_bldepth<true>: the buoyancy difference against the surface layer is evaluated inside the bulk-Richardson march,
    only down to the level where the last column of a wave has found its boundary-layer depth, instead of at every level by
    the buoydiff kernel.  Same operations in the same order: every output of the step is equal to the last bit, on a state
    whose boundary layer spans several levels (and the minimum depth elsewhere)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cfg = named_config("tiny", **kw)
    out = {}
    # "0": full fields; "1": on demand, shear kernel limited by the previous KBL + 3 levels (the default); "1h0": on demand with the
    # full shear field; "1m": the hint undercuts the march everywhere (margin -2), so the march forms the shear itself
    variants = {"0": {"POP_KPP_LAZY": "0"}, "1": {"POP_KPP_LAZY": "1"}, "1h0": {"POP_KPP_LAZY": "1", "POP_KPP_USHEAR_HINT": "0"},
                "1m": {"POP_KPP_LAZY": "1", "POP_KPP_USHEAR_MARGIN": "-2"}}
    for lazy, venv in variants.items():
        for k in ("POP_KPP_LAZY", "POP_KPP_USHEAR_HINT", "POP_KPP_USHEAR_MARGIN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in venv.items():
            monkeypatch.setenv(k, v)
        m, orc = pkg.PopModel(cfg), Oracle(cfg)
        force_kpp_case(m, orc)
        if kw.get("lshort_wave"):
            m.set("SHF_QSW", 5.0e-3 * (1.0 + np.cos(orc.f2("TLAT"))))
        orc.close()
        for _ in range(4):
            m.step()
        out[lazy] = [m.get("HBLT").copy(), m.get("VDC", 1, 0).copy(), m.get("VVC").copy()] + [m.get("KPP_SRC", 1, n).copy() for n in (0, 1)] + \
                    [m.get(n, 1, 0).copy() for n in ("UVEL", "VVEL", "TRACER", "PSURF", "RHO")] + [m.get("TRACER", 1, 1).copy()]
        m.close()
    h = out["0"][0]
    assert h.max() > 3.0 * h[h > 0].min(), "the boundary layer has one depth everywhere: the march exits at once in every wave"
    for v in ("1", "1h0", "1m"):
        for a, b in zip(out["0"], out[v]):
            assert np.array_equal(a, b), v


@pytest.mark.parametrize("kw", [
    {"vmix_choice": 3, "km": 62, "stepped_bathymetry": 1, "hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e22, "ah": -1.0e21},
    {"vmix_choice": 3, "km": 60, "block_size_x": 48, "block_size_y": 40},
    {"vmix_choice": 3, "km": 24, "stepped_bathymetry": 1, "lrich": 0},
    {"vmix_choice": 3, "km": 21, "ns_boundary": 1},                                   # odd level count: the two-level loop ends on its tail
], ids=["km62-stepped-del4", "km60-blocks", "km24-no-shear-term", "km21-cyclic"])
def test_kpp_interior_column_march_is_bitwise_the_level_parallel_kernel(pkg, orclib_built, monkeypatch, kw):
    """k_kpp_buoy_interior_march (buoydiff + ri_iwmix as one column walk, POP_KPP_COL bit 4) against the level-parallel LDS
    kernel (bits 0-3) and against the separate buoydiff / interior kernels (bits 0-1): every output of four steps equal to
    the last bit, shallow and land columns included."""
    monkeypatch.setenv("POP_XCD_REMAP", "0")
    cfg = named_config("tiny", **kw)
    out = {}
    # "31": column march + the sparse boundary-layer kernel (convection mask, source cleared only as deep as it was written);
    # "31s": column march + the streaming boundary-layer kernel; "15" / "3": the level-parallel / separate interior kernels
    for col in ("31", "31s", "15", "3"):
        monkeypatch.setenv("POP_KPP_COL", col[:2])
        monkeypatch.setenv("POP_KPP_SPARSE", "0" if col == "31s" else "1")
        m, orc = pkg.PopModel(cfg), Oracle(cfg)
        force_kpp_case(m, orc)
        orc.close()
        assert m.tuning()["kpp_col"] == int(col[:2])
        res = []
        for step in range(7):
            if step == 4:       # a caller writes the non-local source: honoured by the next step, then cleared at every level again
                junk = np.random.default_rng(3).standard_normal(m.get("KPP_SRC", 1, 0).shape) * 1.0e-7
                m.set("KPP_SRC", junk, 1, 0)
            if step == 5:       # the boundary layer shoals: what the deeper one left in KPP_SRC must go
                m.set("STF", np.abs(m.get("STF", 1, 0)) * 4.0, 1, 0)
            m.step()
            if step in (3, 4, 6):
                res += [m.get("HBLT").copy(), m.get("VDC", 1, 0).copy(), m.get("VVC").copy()] + [m.get("KPP_SRC", 1, n).copy() for n in (0, 1)] + \
                       [m.get(n, 1, 0).copy() for n in ("UVEL", "VVEL", "TRACER", "PSURF", "RHO")] + [m.get("TRACER", 1, 1).copy()]
        out[col] = res
        m.close()
    for col in ("31s", "15", "3"):
        for a, b in zip(out["31"], out[col]):
            assert np.array_equal(a, b), col


@pytest.mark.parametrize("kw,env", [
    ({"vmix_choice": 3, "km": 24}, {}),
    ({"vmix_choice": 3, "km": 62, "stepped_bathymetry": 1}, {"POP_KPP_AHEAD": "1", "POP_KPP_COL": "15", "POP_XCD_REMAP": "0"}),   # look-ahead: KBL travels with its KPP_SRC
    ({"vmix_choice": 3, "km": 60, "tmix_opt": 3}, {"POP_KPP_AHEAD": "1", "POP_TRACER_LDS": "8"}),
])
def test_nonlocal_source_read_down_to_kbl_only_is_bitwise_invisible(pkg, orclib_built, monkeypatch, kw, env):
    """KPP's non-local source is +-0 below level KBL, and the tracer kernel starts its source sum from +0.0: not reading those
    levels (the default) leaves every field equal to the last bit to reading them all (POP_KPP_SRC_FULL=1) -- also when the
    coefficients come from the look-ahead evaluation, and after a caller has written KPP_SRC itself."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cfg = named_config("tiny", **kw)
    out = {}
    for full in ("1", ""):
        if full:
            monkeypatch.setenv("POP_KPP_SRC_FULL", full)
        else:
            monkeypatch.delenv("POP_KPP_SRC_FULL", raising=False)
        m, orc = pkg.PopModel(cfg), Oracle(cfg)
        force_kpp_case(m, orc)
        orc.close()
        for _ in range(5):
            m.step()
        src = m.get("KPP_SRC", 1, 0)
        assert np.abs(src).max() > 0.0
        m.set("KPP_SRC", np.full_like(src, 1.0e-7), n=0)     # a caller's own source, non-zero at every level: must be read in full
        m.time_manager(); m.dhdt()
        m.run_phase("hmix_tracer"); m.run_phase("tracer_rhs")
        out[full] = [m.get("TRACER", 2, 0).copy(), m.get("TRACER", 1, 0).copy(), m.get("TRACER", 1, 1).copy(), m.get("UVEL", 1, 0).copy(), src.copy()]
        m.close()
    for a, b in zip(out["1"], out[""]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("kw", [{"km": 62, "vmix_choice": 3, "stepped_bathymetry": 1}, {"km": 60, "vmix_choice": 3, "block_size_x": 48, "block_size_y": 40}])
def test_two_tracer_corrector_solve_is_bitwise_invisible(pkg, monkeypatch, kw):
    """k_impvmixt2_reg: with one diffusivity array for both tracer classes the corrector's register Thomas solve takes T and S of a
    column in one thread (the elimination coefficients are formed once).  Forced on a small grid against one tracer per thread."""
    monkeypatch.setenv("POP_REG_THOMAS_T", "1")
    cfg = named_config("tiny", **kw)
    out = {}
    for pair in ("0", "1"):
        monkeypatch.setenv("POP_THOMAS_PAIR", pair)
        m = pkg.PopModel(cfg)
        for _ in range(4):
            m.step()
        out[pair] = [m.get("TRACER", 1, n).copy() for n in (0, 1)] + [m.get(n, 1, 0).copy() for n in ("RHO", "UVEL", "PSURF")]
        m.close()
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("kw", [{"km": 62, "vmix_choice": 3, "stepped_bathymetry": 1}, {"km": 21, "tmix_opt": 3}, {"km": 20, "ns_boundary": 1}])
def test_density_kernels_agree_bitwise(pkg, monkeypatch, kw):
    """k_state3d_lv: the density of a whole 3-D array with 2 / 4 / 8 levels per thread and the level's pressure-dependent coefficients
    read from a table the device formed with mwjf_level itself, against one cell per thread with the coefficients formed in place
    (level counts that are and are not multiples of the levels per thread; the Robert filter's two evaluations included)."""
    cfg = named_config("tiny", **kw)
    out = {}
    for lv in ("1", "2", "4", "8"):
        monkeypatch.setenv("POP_STATE3D_LEVELS", lv)
        m = pkg.PopModel(cfg)
        assert m.tuning()["state3d_levels"] == int(lv)
        for _ in range(4):
            m.step()
        out[lv] = [m.get("RHO", tl, 0).copy() for tl in (0, 1, 2)] + [m.get(n, 1, 0).copy() for n in ("TRACER", "UVEL", "PSURF")]
        m.close()
    assert np.abs(out["1"][1]).max() > 1.0
    for lv in ("2", "4", "8"):
        for a, b in zip(out["1"], out[lv]):
            assert np.array_equal(a, b), lv


def test_del4_first_laplacian_patch_shapes_agree_bitwise(pkg, monkeypatch):
    """The first Laplacians of del4 (k_del4_d2t / k_del4_d2u) and KPP's viscosity average to U points (k_kpp_vvc) run over 256
    consecutive cells or over 64 x R patches (large grids: R = 4); the cell -> thread map is all that changes."""
    cfg = named_config("tiny", hmix_momentum=4, hmix_tracer=4, am=-1.0e22, ah=-1.0e21, stepped_bathymetry=1, lvariable_hmix=1, vmix_choice=3, km=24)
    out = {}
    for rows in ("0", "2", "4", "8", "16"):
        monkeypatch.setenv("POP_DEL4_TILE", rows)
        m = pkg.PopModel(cfg)
        for _ in range(3):
            m.step()
        out[rows] = [m.get(n, 1, 0).copy() for n in ("UVEL", "VVEL", "TRACER", "PSURF", "VVC")] + [m.get("TRACER", 1, 1).copy()]
        m.close()
    for rows in ("2", "4", "8", "16"):
        for a, b in zip(out["0"], out[rows]):
            assert np.array_equal(a, b), rows


@pytest.mark.parametrize("kw,env", [
    ({"hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e22, "ah": -1.0e21, "lvariable_hmix": 1, "stepped_bathymetry": 1}, {}),   # 16 blocks: the ghost ring crosses blocks
    ({"hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e22, "ah": -1.0e21, "vmix_choice": 3, "km": 62, "block_size_x": 48, "block_size_y": 40}, {"POP_TRACER_LDS": "4"}),
    ({"hmix_tracer": 4, "ah": -1.0e21, "tmix_opt": 1, "time_mix_freq": 3, "lpressure_avg": 0}, {"POP_TRACER_LDS": "8"}),          # frequent averaging steps
    ({"hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e22, "ah": -1.0e21, "tmix_opt": 3}, {}),                                      # Robert filter: never formed ahead
])
def test_first_laplacians_formed_by_the_previous_step_are_bitwise_invisible(pkg, monkeypatch, kw, env):
    """del4 on large grids: the tracer and momentum kernels also form the first Laplacian of their current fields -- the next
    leapfrog step's mix-time fields -- from the tiles they hold in LDS, and that step skips k_del4_d2t / k_del4_d2u.  Forced on small grids: every field equal to
    the last bit over Euler, averaging and leapfrog steps, and after a caller has replaced the tracers between two steps."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cfg = named_config("tiny", **kw)
    out = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("POP_D2T_FUSE", fuse)           # the velocity's first Laplacian follows the same switch (k_momentum_rhs_lds)
        m = pkg.PopModel(cfg)
        for s_ in range(7):
            m.step()
            if s_ == 3:                                   # a new state between two steps: the field formed ahead must be dropped
                u = m.get("UVEL", 1, 0)
                m.set("UVEL", u * (1.0 + 1.0e-3 * np.sin(np.arange(u.shape[-1]))), 1, 0)
                for n in (0, 1):
                    t = m.get("TRACER", 1, n)
                    # ghost cells that are NOT copies of their source cells: the library must then keep forming the first
                    # Laplacian from them (k_del4_d2t) until a halo update has made the slot consistent again
                    m.set("TRACER", t * (1.0 + 1.0e-3 * np.cos(np.arange(t.shape[-1]))), 1, n)
        out[fuse] = [m.get(f, 1, 0).copy() for f in ("TRACER", "UVEL", "VVEL", "PSURF", "RHO")] + [m.get("TRACER", 1, 1).copy(), m.get("TRACER", 0, 0).copy()]
        m.close()
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("kw", [{"km": 62}, {"km": 60, "vmix_choice": 3, "stepped_bathymetry": 1},
                                {"km": 62, "vmix_choice": 3, "hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e22, "ah": -1.0e21, "block_size_x": 48, "block_size_y": 40}])
def test_deferred_vertical_mixing_with_barotropic_sum_is_bitwise_invisible(pkg, monkeypatch, kw):
    """Large grids hold the implicit vertical mixing of U, V back until the barotropic solve has finished and add the
    barotropic velocity in its final store (k_impvmixu_reg<., ., true>).  Forced here on a small grid: equal to the last
    bit to the two-launch form, over Euler, averaging and leapfrog steps; and a caller that reads a field between the
    driver calls (which flushes the held-back launch) sees the same fields as the phase-by-phase run."""
    cfg = named_config("tiny", **kw)
    names = ("UVEL", "VVEL", "TRACER", "PSURF", "UBTROP", "RHO")
    out = {}
    for defer in ("0", "1"):
        monkeypatch.setenv("POP_VMIXU_DEFER", defer)
        m = pkg.PopModel(cfg)
        for _ in range(5):
            m.step()
        out[defer] = [m.get(n, 1, 0).copy() for n in names] + [m.get(n, 2, 0).copy() for n in names]
        m.close()
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a, b)
    # interrupted sequence: a field read after baroclinic_driver must show the mixed velocity, and the step must end the same
    monkeypatch.setenv("POP_VMIXU_DEFER", "1")
    m = pkg.PopModel(cfg)
    monkeypatch.setenv("POP_VMIXU_DEFER", "0")
    r = pkg.PopModel(cfg)
    for s_ in range(3):
        for x in (m, r):
            x.time_manager(); x.dhdt(); x.baroclinic_driver()
        for n in ("UVEL", "VVEL"):          # new time level, between the drivers
            assert np.array_equal(m.get(n, 2, 0), r.get(n, 2, 0)), (s_, n)
        for x in (m, r):
            x.barotropic_driver(); x.baroclinic_correct_adjust(); x.step_tail()
        for n in names:
            assert np.array_equal(m.get(n, 1, 0), r.get(n, 1, 0)), (s_, n)
    m.close(); r.close()


@pytest.mark.parametrize("kw,env", [({"vmix_choice": 3, "km": 24}, {}), ({"vmix_choice": 3, "km": 62, "stepped_bathymetry": 1}, {"POP_KPP_COL": "15", "POP_XCD_REMAP": "0"}),
                                    ({"vmix_choice": 3, "km": 60}, {"POP_KPP_COL": "3"})])
def test_kpp_shared_diffusivity_array_is_bitwise_invisible(pkg, monkeypatch, kw, env):
    """KPP without double diffusion gives both tracer classes the same diffusivity value for value, so the library keeps one
    array for both (POP_VDC_SHARED=0 keeps two).  Two arrays hold identical values, and the runs agree to the last bit."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cfg = named_config("tiny", **kw)
    out = {}
    for shared in ("0", "1"):
        monkeypatch.setenv("POP_VDC_SHARED", shared)
        m = pkg.PopModel(cfg)
        for _ in range(4):
            m.step()
        out[shared] = [m.get("VDC", 1, 0).copy(), m.get("VDC", 1, 1).copy()] + [m.get(n, 1, 0).copy() for n in ("UVEL", "TRACER", "PSURF", "VVC", "HBLT")] + [m.get("TRACER", 1, 1).copy(), m.get("KPP_SRC", 1, 1).copy()]
        m.close()
    assert np.array_equal(out["0"][0], out["0"][1]), "the two classes differ without double diffusion"
    assert np.abs(out["0"][0]).max() > 0.0
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("kw,env", [
    ({"km": 62}, {}),                                                    # tx0.1v3 level count: k_impvmixu_reg<62>, generic tracer solve
    ({"km": 62, "vmix_choice": 3, "hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e22, "ah": -1.0e21},
     {"POP_REG_THOMAS_T": "1", "POP_KPP_COL": "3", "POP_XCD_REMAP": "0"}),   # the tx0.1v3 kernel selection + register tracer solve
    ({"km": 60, "vmix_choice": 3}, {"POP_REG_THOMAS_T": "0", "POP_KPP_COL": "3"}),
    ({"km": 62, "vmix_choice": 3, "hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e22, "ah": -1.0e21, "stepped_bathymetry": 1},
     {"POP_REG_THOMAS_T": "0", "POP_KPP_COL": "7", "POP_XCD_REMAP": "0"}),   # round 2: buoydiff in the level-parallel LDS form
    ({"km": 60, "vmix_choice": 3, "ldbl_diff": 1}, {"POP_KPP_COL": "7"}),    # the same with the XCD-banded column order
    ({"km": 62, "vmix_choice": 3, "hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e22, "ah": -1.0e21, "stepped_bathymetry": 1},
     {"POP_REG_THOMAS_T": "0", "POP_KPP_COL": "15", "POP_XCD_REMAP": "0"}),  # buoydiff + interior coefficients in one level-parallel launch
    ({"km": 60, "vmix_choice": 3, "ldbl_diff": 1, "stepped_bathymetry": 1}, {"POP_KPP_COL": "15"}),
    ({"km": 24, "vmix_choice": 3, "num_v_smooth_Ri": 3}, {"POP_KPP_COL": "15"}),   # several smoothing passes
])
def test_production_level_counts_match_oracle(pkg, orclib_built, monkeypatch, kw, env):
    """Kernels that are compiled for the production level counts (km = 60, 62) or selected by grid size are
    forced here on a grid small enough for the oracle, so every kernel the tx0.1v3 / gx1v7 bench runs has a
    parity case."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cfg = named_config("tiny", **kw)
    gpu, orc = pkg.PopModel(cfg), Oracle(cfg)
    if cfg.vmix_choice == 3:
        force_kpp_case(gpu, orc)
    tol = TOL_LOCAL
    for s in range(1, 4):
        run_phases(gpu, orc, s, tol)
        tol = TOL_SOLVE
    gpu.close(); orc.close()


@pytest.mark.parametrize("kw", [{"vmix_choice": 3, "km": 24}, {"vmix_choice": 3, "km": 24, "ldbl_diff": 1, "block_size_x": 48, "block_size_y": 40},
                                {"vmix_choice": 3, "km": 62, "stepped_bathymetry": 1, "num_v_smooth_Ri": 2, "ldbl_diff": 1}])
def test_kpp_column_kernels_agree_bitwise(pkg, orclib_built, monkeypatch, kw):
    """buoydiff / ushear exist in a 3-D-parallel form (small grids) and a column form with the top
    reference levels in registers (large grids; POP_KPP_COL is a bit mask that forces either): same operations in the same
    order, so every output of the step must be identical to the last bit."""
    cfg = named_config("tiny", **kw)
    out = {}
    for mode in ("3", "0", "7", "15"):
        monkeypatch.setenv("POP_KPP_COL", mode)
        m, orc = pkg.PopModel(cfg), Oracle(cfg)
        force_kpp_case(m, orc)
        orc.close()
        for _ in range(3):
            m.step()
        out[mode] = [m.get(n, 1, 0).copy() for n in ("UVEL", "VVEL", "TRACER", "PSURF", "VVC", "HBLT")]
        m.close()
    for a, b in zip(out["3"], out["0"]):
        assert np.array_equal(a, b)
    for a, b in zip(out["7"], out["0"]):          # buoydiff level-parallel with the surface layer in LDS
        assert np.array_equal(a, b)
    for a, b in zip(out["15"], out["0"]):         # buoydiff + ri_iwmix + ddmix fused, level-parallel
        assert np.array_equal(a, b)


def test_robert_filter_conserves_tracer_volume(pkg, orclib_built):
    """step_RF_diag (step_mod.F90:1362-1430): <T*volume> of curtime is the same before and after the
    filter (exact when robert_newtime = 0 and no previous-step averaging applies: first filtered step)."""
    cfg = named_config("tiny", tmix_opt=3, robert_alpha=1.0)
    gpu = pkg.PopModel(cfg)
    orc = Oracle(cfg)
    dz = orc.v1("dz")[1:cfg.km + 1].copy()      # copy: the view dies with the oracle
    tarea = interior(gpu.get("TAREA")); kmt = interior(gpu.geti("KMT"))
    orc.close()

    def trvol(tl, n):
        T = gpu.get("TRACER", tl, n)[:, :, 2:-2, 2:-2]
        P = interior(gpu.get("PSURF", tl))
        thick = np.broadcast_to(dz[None, :, None, None], T.shape).copy()
        thick[:, 0] = dz[0] + P / 980.6
        mask = (np.arange(1, cfg.km + 1)[None, :, None, None] <= kmt[:, None])
        return float((tarea[:, None] * thick * mask * T).sum())

    gpu.step()                       # forward-Euler step: filter acts on old = cur
    gpu.time_manager(); gpu.dhdt(); gpu.baroclinic_driver(); gpu.barotropic_driver(); gpu.baroclinic_correct_adjust()
    pre = [trvol(1, n) for n in (0, 1)]
    gpu.step_tail()                  # filter + rotation: the filtered curtime is now oldtime
    post = [trvol(0, n) for n in (0, 1)]
    for a, b in zip(pre, post):
        assert abs(a - b) <= 1e-12 * abs(a), (a, b)
    gpu.close()


def test_global_sum_family_and_solver_diagonal(pkg, orclib_built):
    """The other members of the POP_GlobalSum interface (NFields, Prod, Scalar, 2DI4;
    mpi/POP_ReductionsMod.F90:50-64) and POP_SolversDiagonal (POP_SolversMod.F90:1110-1151)."""
    cfg = named_config("tiny")
    gpu, orc = pkg.PopModel(cfg), Oracle(cfg)
    rng = np.random.default_rng(11)
    a = rng.standard_normal((gpu.nblocks, gpu.nyb, gpu.nxb)) * 1e2
    b = rng.standard_normal((gpu.nblocks, gpu.nyb, gpu.nxb))
    gpu.set("RHS", a); gpu.set("PSURF", b, tl=2)
    mask = interior(orc.f2("mMask"))
    scale = np.abs(interior(a)).sum()
    s1, s2 = gpu.global_sum_nfields(["RHS", "TAREA"], mask="mMask")
    assert s1 == gpu.global_sum("RHS", mask="mMask") and s2 == gpu.global_sum("TAREA", mask="mMask")
    got = gpu.global_sum_prod("RHS", "PSURF", tl_b=2, mask="mMask")
    assert abs(got - (interior(a) * interior(b) * mask).sum()) <= 1e-12 * scale
    got = gpu.global_sum_prod("RHS", "PSURF", tl_b=2)
    assert abs(got - (interior(a) * interior(b)).sum()) <= 1e-12 * scale
    assert gpu.global_sum_scalar(3.25) == 3.25
    assert gpu.global_sum_i4("KMT") == int(interior(orc.i2("KMT")).sum())
    # operator diagonal of block 2 <- independent part - correction
    corr = rng.standard_normal((gpu.nyb, gpu.nxb))
    before = gpu.get("centerWgt").copy()
    gpu.solver_diagonal(2, corr)
    after = gpu.get("centerWgt")
    assert np.array_equal(after[1], gpu.get("centerWgtIndep")[1] - corr)
    assert np.array_equal(after[0], before[0]) and np.array_equal(after[2:], before[2:])
    with pytest.raises(pkg.PopError):
        gpu.solver_diagonal(gpu.nblocks + 1, corr)
    gpu.close(); orc.close()


@pytest.mark.parametrize("wave", ["3", "2", "1", "0"], ids=["wavefront-registers-loads-up-front", "wavefront-registers", "wavefront-lds", "thread-per-sub-block"])
@pytest.mark.parametrize("name,kw", [("tiny", {}), ("tiny", {"block_size_x": 16, "block_size_y": 20}), ("gx3v7", {}),
                                     ("tiny", {"nx_global": 66, "ny_global": 52, "block_size_x": 33, "block_size_y": 26, "stepped_bathymetry": 1})])
def test_evp_preconditioner_is_bitwise_the_oracle(pkg, orclib_built, monkeypatch, name, kw, wave):
    """preconditioner() on its own (POP_SolversMod.F90:2268-2369, 2618-2696): every value of the sub-block marches is formed by
    the reference's expression on both sides -- one thread per sub-block in the reference's loop order, or the anti-diagonal
    wavefront form (eight lanes per sub-block; sub-blocks of every size from 1 x 1 to 8 x 8, land sub-blocks) --, so PX agrees
    bit for bit; the eigenvalue bounds P-CSI derives through it (Lanczos on the host) are identical too."""
    monkeypatch.setenv("POP_EVP_WAVE", wave)
    cfg = named_config(name, precond_choice=1, solver_choice=3, **kw)
    gpu, orc = pkg.PopModel(cfg), Oracle(cfg)
    assert gpu.scalar("PcsiMaxEigs") == orc.scalar("PcsiMaxEigs") and gpu.scalar("PcsiMinEigs") == orc.scalar("PcsiMinEigs")
    rng = np.random.default_rng(17)
    x = rng.standard_normal((gpu.nblocks, gpu.nyb, gpu.nxb))
    gpu.set("RHS", x)
    gpu.solver_preconditioner("RHS", "DH")
    assert np.array_equal(gpu.get("DH"), orc.preconditioner(x))
    gpu.close(); orc.close()
    # diagonal choice through the same entry
    cfg = named_config(name, **kw)
    gpu, orc = pkg.PopModel(cfg), Oracle(cfg)
    gpu.step(); orc.step()
    gpu.set("RHS", x)
    gpu.solver_preconditioner("RHS", "DH")
    assert np.array_equal(interior(gpu.get("DH")), interior(orc.preconditioner(x)))
    gpu.close(); orc.close()


@pytest.mark.parametrize("name,kw", [
    ("tiny", {"solver_choice": 3, "stepped_bathymetry": 1}),                                  # P-CSI (fused: step kernel + sub-block solves)
    ("tiny", {"solver_choice": 3, "block_size_x": 24, "block_size_y": 20, "ew_boundary": 0}),
    ("tiny", {"solver_choice": 1, "block_size_x": 28, "block_size_y": 24}),                   # pcg, padded blocks: sub-blocks beyond the block's own extent
    ("tiny", {"solver_choice": 2}),                                                           # ChronGear
    ("test", {"solver_choice": 3}),                                                           # 96 blocks of 16 x 16
    ("gx1v7", {"solver_choice": 3}),                                                          # continents: whole waves of sub-blocks without an ocean cell
    ("gx1v7", {"solver_choice": 1}),
    ("tx0.1v3", {"nx_global": 1800, "ny_global": 1200, "block_size_x": 1800, "block_size_y": 1200, "km": 20, "vmix_choice": 1, "solver_choice": 3}),
])
def test_evp_without_reading_land_sub_blocks_is_bitwise(pkg, monkeypatch, name, kw):
    """k_evp_apply_wave3<true> (the solvers' applications: the operand is a residual, zero on land) does not read sub-blocks without an ocean
    cell and lets waves that hold nothing else leave at once; k_evp_apply_wave2 reads and scales every sub-block.  Iteration counts and
    fields equal (np.array_equal: a zero is a zero)."""
    cfg = named_config(name, preconditioner_choice=1, **kw)
    a = pkg.PopModel(cfg, tuning={"evp_wave": 2})
    b = pkg.PopModel(cfg, tuning={"evp_wave": 3})
    for step in range(3):
        a.step(); b.step()
        assert a.solver_diagnostics() == b.solver_diagnostics(), "step %d" % step
    for f in ("PSURF", "UBTROP", "VBTROP", "UVEL", "TRACER", "GRADPX"):
        assert np.array_equal(a.get(f), b.get(f)), f
    a.close(); b.close()


@pytest.mark.parametrize("kw", [{}, {"tadvect": 2}, {"tadvect": 3}, {"hmix_tracer": 4, "hmix_momentum": 4, "am": -1.0e22, "ah": -1.0e21},
                                {"block_size_x": 48, "block_size_y": 40}])
def test_uniform_tracers_stay_uniform_in_the_interior(pkg, kw):
    """The scheme's own invariant on the device (see the oracle twin in tests/test_oracle_fixtures.py): with
    wind-driven flow, uniform T and S stay EXACTLY uniform away from the surface layer and the sea floor --
    advection consistent with continuity, del2 / del4 of a constant = 0, vertical mixing of a constant = 0."""
    cfg = named_config("tiny", **kw)
    m = pkg.PopModel(cfg)
    kmt = m.geti("KMT")
    k = np.arange(1, m.km + 1)[None, :, None, None]
    wet = k <= kmt[:, None]
    for tl in (0, 1, 2):
        m.set("TRACER", np.where(wet, 10.0, 0.0), tl, 0)
        m.set("TRACER", np.where(wet, 0.035, 0.0), tl, 1)
    for _ in range(6):
        m.step()
    assert np.abs(m.get("UVEL", 1)).max() > 1.0
    mid = slice(7, m.km - 3)
    for n, val in ((0, 10.0), (1, 0.035)):
        f = m.get("TRACER", 1, n)
        assert np.array_equal(f[:, mid][wet[:, mid]], np.full(int(wet[:, mid].sum()), val))
        assert np.abs(np.where(wet, f - val, 0.0)).max() < 1e-4 * val
    m.close()


@pytest.mark.parametrize("kw", [{}, {"solver_choice": 2}, {"solver_choice": 3}, {"precond_choice": 1}, {"solver_choice": 3, "precond_choice": 1},
                                {"block_size_x": 48, "block_size_y": 40}])
def test_solvers_recover_a_manufactured_solution(pkg, orclib_built, kw):
    """b = A x_true (operator of the oracle, same centre weight) for a random x_true on the ocean points: the
    device solvers recover it from a zero first guess, with the oracle's iteration count."""
    cfg = named_config("tiny", convergence_criterion=1.0e-13, **kw)
    gpu, orc = pkg.PopModel(cfg), Oracle(cfg)
    gpu.step(); orc.step()
    assert np.array_equal(gpu.get("centerWgt"), orc.f2("centerWgt"))
    rng = np.random.default_rng(4)
    mask = orc.f2("mMask").copy()
    x_true = orc.halo(rng.standard_normal(mask.shape) * 1.0e3 * mask)
    b = orc.halo(orc.btrop_operator(x_true))
    rc, x_orc = orc.solver_run(np.zeros_like(x_true), b)
    assert rc == 0
    gpu.set("RHS", b)
    gpu.set("PSURF", np.zeros_like(b), 2)
    gpu.solver_run()
    x = gpu.get("PSURF", 2)
    assert gpu.solver_diagnostics()[0] == orc.L.orc_solver_iterations(orc.h)
    assert np.abs(interior(x - x_true)).max() <= 1e-7 * np.abs(x_true).max()
    assert np.abs(interior(x - x_orc)).max() <= 1e-9 * np.abs(x_true).max()
    gpu.close(); orc.close()


def test_operators_are_bitwise_the_oracle(pkg, orclib_built):
    """operators.F90 grad / div / zcurl as stand-alone entry points (pop_operator), on 2-D fields and on a level of
    3-D fields: same expressions in the same order as the oracle, so the results are identical."""
    cfg = named_config("tiny")
    gpu, orc = pkg.PopModel(cfg), Oracle(cfg)
    rng = np.random.default_rng(12)
    s2 = (gpu.nblocks, gpu.nyb, gpu.nxb)
    f, u, v = (rng.standard_normal(s2) for _ in range(3))
    gpu.set("PSURF", f); gpu.set("UBTROP", u); gpu.set("VBTROP", v)
    gpu.operator("grad", 1, "PSURF", o1="GRADPX", o2="GRADPY")
    gx, gy = orc.operator("grad", 1, f)
    assert np.array_equal(gpu.get("GRADPX"), gx) and np.array_equal(gpu.get("GRADPY"), gy)
    for op in ("div", "zcurl"):
        gpu.operator(op, 1, "UBTROP", "VBTROP", o1="DH")
        assert np.array_equal(gpu.get("DH"), orc.operator(op, 1, u, v)), op
    k = 5
    u3 = rng.standard_normal((gpu.nblocks, gpu.km, gpu.nyb, gpu.nxb)); v3 = rng.standard_normal(u3.shape)
    gpu.set("UVEL", u3); gpu.set("VVEL", v3)
    gpu.operator("zcurl", k, "UVEL", "VVEL", o1="DH")
    assert np.array_equal(gpu.get("DH"), orc.operator("zcurl", k, u3[:, k - 1], v3[:, k - 1]))
    gpu.operator("div", k, "UVEL", "VVEL", o1="VVC")             # 3-D output: its level-k slab
    assert np.array_equal(gpu.get("VVC")[:, k - 1], orc.operator("div", k, u3[:, k - 1], v3[:, k - 1]))
    # the reference's own argument lists, grad / div / zcurl(k, ..., this_block): host arrays of one block (pop_operator_host)
    for kk, (fu, fv) in ((1, (u, v)), (k, (u3[:, k - 1], v3[:, k - 1]))):
        gx, gy = orc.operator("grad", kk, f)
        dv, cu = orc.operator("div", kk, fu, fv), orc.operator("zcurl", kk, fu, fv)
        for ib in range(1, gpu.nblocks + 1):
            hx, hy = gpu.operator_host("grad", kk, ib, f[ib - 1])
            assert np.array_equal(hx, gx[ib - 1]) and np.array_equal(hy, gy[ib - 1])
            assert np.array_equal(gpu.operator_host("div", kk, ib, fu[ib - 1], fv[ib - 1]), dv[ib - 1])
            assert np.array_equal(gpu.operator_host("zcurl", kk, ib, fu[ib - 1], fv[ib - 1]), cu[ib - 1])
    with pytest.raises(pkg.PopError, match="local_id"):
        gpu.operator_host("div", 1, gpu.nblocks + 1, u[0], v[0])
    gpu.close(); orc.close()


def test_entry_points_fail_loudly(pkg, tmp_path):
    """Error convention of the C ABI (non-zero return + message, no partial work): bad operator arguments, unknown
    fields, a missing or truncated restart file."""
    m = pkg.PopModel(named_config("tiny"))
    with pytest.raises(pkg.PopError, match="op 0 grad"):
        m.operator("grad", 0, "PSURF")
    with pytest.raises(pkg.PopError, match="unknown field"):
        m.operator("div", 1, "NOPE", "VBTROP")
    with pytest.raises(pkg.PopError, match="unknown 2-D field"):
        m.solver_preconditioner("UVEL", "DH")                    # 3-D field where a 2-D one is required
    with pytest.raises(pkg.PopError, match="cannot open"):
        m.read_restart(str(tmp_path / "missing.bin"))
    m.step()
    path = str(tmp_path / "r.bin")
    m.write_restart(path)
    with open(path, "r+b") as f:
        f.truncate(1000)
    with pytest.raises(pkg.PopError, match="short read"):
        m.read_restart(path)
    hdr = open(path + ".hdr").read().replace("&SALT_OLD", "&SALT_GONE")
    open(path + ".hdr", "w").write(hdr)
    with pytest.raises(pkg.PopError, match="could not find field in binary header file: SALT_OLD"):
        m.read_restart(path)
    m.close()


@pytest.mark.parametrize("solver", [1, 2, 3])
def test_solver_error_convention(pkg, orclib_built, solver):
    """POP_SolversMod.F90:1492-1497: hitting maxIterations is an error (errorCode set, message) unless
    convergenceCriterion == 0, in which case the solver simply runs maxIterations steps."""
    cfg = named_config("tiny", solver_choice=solver, max_iterations=20, convergence_check_freq=10)
    cfg.convergence_check_start = 10                                  # PCSI: start checking early
    cfg.convergence_criterion = 1.0e-30                      # unreachable
    m = pkg.PopModel(cfg)
    with pytest.raises(pkg.PopError, match="not converged"):
        m.step()
    m.close()
    cfg.convergence_criterion = 0.0
    m, o = pkg.PopModel(cfg), Oracle(cfg)
    m.step(); o.step()
    assert m.solver_diagnostics()[0] == 20 == o.L.orc_solver_iterations(o.h)
    assert relerr(m.get("PSURF", 1), o.f2("PSURF", 1)) < TOL_SOLVE
    m.close(); o.close()


@pytest.mark.parametrize("precond", [0, 1], ids=["diagonal", "evp"])
def test_fused_pcsi_is_bitwise_the_unfused_pcsi(pkg, monkeypatch, precond):
    """P-CSI: the one-launch-per-iteration form (neighbour updates recomputed in the matvec, ping-pong state,
    hipGraph per check interval) against the operation-by-operation form with explicit halo updates.  With the EVP block
    preconditioner the fused form is two launches per iteration: the step kernel leaves the residual itself, the sub-block solves
    (k_evp_apply_wave2) turn it into r' for the next step -- no halo update, no copy."""
    cfg = named_config("tiny", block_size_x=24, block_size_y=20, solver_choice=3, precond_choice=precond)
    a = pkg.PopModel(cfg)
    monkeypatch.setenv("POP_SOLVER_UNFUSED", "1")
    b = pkg.PopModel(cfg)
    monkeypatch.delenv("POP_SOLVER_UNFUSED")
    monkeypatch.setenv("POP_SOLVER_NOGRAPH", "1")
    c = pkg.PopModel(cfg)
    monkeypatch.delenv("POP_SOLVER_NOGRAPH")
    monkeypatch.setenv("POP_PCSI_STEP2", "1")             # two cells per thread (the large-grid form)
    d = pkg.PopModel(cfg)
    for _ in range(4):
        a.step(); b.step(); c.step(); d.step()
        assert a.solver_diagnostics() == b.solver_diagnostics() == c.solver_diagnostics() == d.solver_diagnostics()
    for name in ("PSURF", "UBTROP", "VBTROP", "UVEL", "TRACER"):
        assert np.array_equal(a.get(name), b.get(name)), name
        assert np.array_equal(a.get(name), c.get(name)), name
        assert np.array_equal(a.get(name), d.get(name)), name
    a.close(); b.close(); c.close(); d.close()


@pytest.mark.parametrize("kw", [{}, {"solver_choice": 2}, {"solver_choice": 2, "convergence_check_freq": 5, "max_iterations": 203}])
def test_fused_solver_is_bitwise_the_unfused_solver(pkg, monkeypatch, kw):
    """The single-rank fused PCG / ChronGear (halo folded into the matvec, in-kernel final reduction, hipGraph
    replay) must produce exactly the bits of the plain kernel-per-operation path."""
    import os
    cfg = named_config("tiny", block_size_x=24, block_size_y=20, **kw)      # 4 blocks
    a = pkg.PopModel(cfg)
    monkeypatch.setenv("POP_SOLVER_UNFUSED", "1")
    b = pkg.PopModel(cfg)
    monkeypatch.delenv("POP_SOLVER_UNFUSED")
    monkeypatch.setenv("POP_SOLVER_NOGRAPH", "1")
    c = pkg.PopModel(cfg)
    monkeypatch.setenv("POP_SOLVER_PRESUM", "1")          # the large-grid reduction scheme (pcg: two cells per thread in step B)
    d = pkg.PopModel(cfg)
    monkeypatch.setenv("POP_FPCG_B2", "0")                # ... with one cell per thread
    e = pkg.PopModel(cfg)
    monkeypatch.delenv("POP_FPCG_B2")
    for _ in range(4):
        a.step(); b.step(); c.step(); d.step(); e.step()
        assert a.solver_diagnostics() == b.solver_diagnostics() == c.solver_diagnostics() == d.solver_diagnostics() == e.solver_diagnostics()
    for name in ("PSURF", "UBTROP", "VBTROP", "UVEL", "TRACER"):
        assert np.array_equal(a.get(name), b.get(name)), name
        assert np.array_equal(a.get(name), c.get(name)), name
        assert np.array_equal(a.get(name), d.get(name)), name
        assert np.array_equal(a.get(name), e.get(name)), name
    a.close(); b.close(); c.close(); d.close(); e.close()


@pytest.mark.parametrize("name,kw", [
    ("tiny", {"block_size_x": 48, "block_size_y": 40}),                                     # one block, cyclic east-west: rim cells of the sub-blocks read at their source cells
    ("tiny", {"block_size_x": 24, "block_size_y": 20}),                                     # 4 blocks
    ("tiny", {"block_size_x": 24, "block_size_y": 20, "ew_boundary": 0, "stepped_bathymetry": 1}),   # closed everywhere: fill values in the rims; sub-blocks with land (diagonal scaling)
    ("tiny", {"block_size_x": 28, "block_size_y": 24}),                                     # padded blocks: ragged sub-blocks
    ("tiny", {"block_size_x": 24, "block_size_y": 20, "convergence_check_freq": 7, "max_iterations": 200, "convergence_check_start": 14}),   # odd interval, a last interval without a check
    ("tiny", {"ns_boundary": 2, "block_size_x": 24, "block_size_y": 20}),                   # tripole fold: rim cells beyond it are mirrored copies (formed at their source cells)
    ("gx3v7", {}),
    ("gx1v7", {}),
    ("tx0.1v3", {"nx_global": 1800, "ny_global": 1200, "block_size_x": 1800, "block_size_y": 1200, "km": 20, "vmix_choice": 1}),
])
def test_one_launch_pcsi_evp_is_bitwise_the_two_launch_form(pkg, name, kw):
    """pop_tuning.pcsi_evp_fused: a P-CSI iteration with the EVP preconditioner as ONE launch (k_pcsi_evp_step: the wave that solves eight
    sub-blocks first forms dx, x and r = b - A x for them) against k_pcsi_step2 + k_evp_apply_wave2.  Same expressions in the same order:
    iteration counts and fields bit for bit."""
    cfg = named_config(name, solver_choice=3, preconditioner_choice=1, **kw)
    grid = synthetic_grid(cfg) if cfg.ns_boundary == 2 else None
    a = pkg.PopModel(cfg, tuning={"pcsi_evp_fused": 0}, grid=grid)
    b = pkg.PopModel(cfg, tuning={"pcsi_evp_fused": 1}, grid=grid)
    assert (a.dim("pcsi_evp_fused"), b.dim("pcsi_evp_fused")) == (0, 1)
    for step in range(4):
        a.step(); b.step()
        assert a.solver_diagnostics() == b.solver_diagnostics(), "step %d" % step
    for f in ("PSURF", "UBTROP", "VBTROP", "UVEL", "TRACER", "GRADPX"):
        assert np.array_equal(a.get(f), b.get(f)), f
    a.close(); b.close()


@pytest.mark.parametrize("name,kw", [
    ("tiny", {"block_size_x": 48, "block_size_y": 40}),                                     # 9 chunks: accumulators with one term or none, three of the four quarters empty
    ("tiny", {"block_size_x": 24, "block_size_y": 20}),                                     # 4 blocks: one workgroup per block
    ("gx3v7", {}),                                                                          # 49 chunks
    ("gx1v7", {}),                                                                          # 492 chunks: two terms per accumulator, ragged
    ("gx1v7", {"solver_choice": 2}),                                                        # ChronGear: two fields, one workgroup per (block, field)
    ("tiny", {"solver_choice": 2, "block_size_x": 24, "block_size_y": 20}),
    ("tx0.1v3", {"nx_global": 1800, "ny_global": 1200, "block_size_x": 1800, "block_size_y": 1200, "km": 20, "vmix_choice": 1}),   # 8 484 chunks: 34 terms, quarters of 9, 9, 9, 7 (+1 ragged)
    ("tx0.1v3", {"nx_global": 1800, "ny_global": 1200, "block_size_x": 1800, "block_size_y": 1200, "km": 20, "vmix_choice": 1, "solver_choice": 2}),
])
def test_relay_block_sums_are_bitwise_the_batched_block_sums(pkg, name, kw):
    """pop_tuning.block_sums_relay: the ordered block sums between the kernels of a fused pcg / ChronGear iteration by 1024 threads, four per
    accumulator (k_block_sums_relay: every term requested at once, the quarters of an accumulator's terms added in turn) against the
    256-thread form.  The same additions in the same order: iteration counts and fields bit for bit.  (2 = also where an accumulator
    has fewer than 64 terms; the default 1 applies from 64 terms on, i.e. tx0.1v3 in one block -- compared at full size by
    profiles/probes/relay_fullsize.py.)"""
    cfg = named_config(name, **kw)
    base = {"pcg_persist": 0, "solver_presum": 1}
    a = pkg.PopModel(cfg, tuning=dict(base, block_sums_relay=0))
    b = pkg.PopModel(cfg, tuning=dict(base, block_sums_relay=2))
    for step in range(3):
        a.step(); b.step()
        assert a.solver_diagnostics() == b.solver_diagnostics(), "step %d" % step
    for f in ("PSURF", "UBTROP", "VBTROP", "UVEL", "TRACER", "GRADPX"):
        assert np.array_equal(a.get(f), b.get(f)), f
    a.close(); b.close()


@pytest.mark.parametrize("name,kw", [
    ("tiny", {"block_size_x": 48, "block_size_y": 40}),                                     # one block, cyclic east-west, closed north-south
    ("tiny", {"block_size_x": 24, "block_size_y": 20}),                                     # 4 blocks: the rings of a tile cross block boundaries through the source map
    ("tiny", {"block_size_x": 24, "block_size_y": 20, "ew_boundary": 0, "stepped_bathymetry": 1}),   # closed everywhere: fill cells in the rings
    ("tiny", {"block_size_x": 28, "block_size_y": 24}),                                     # padded blocks: tiles overhang the physical domain
    ("tiny", {"block_size_x": 48, "block_size_y": 40, "convergence_check_freq": 7, "max_iterations": 200, "convergence_check_start": 14}),   # odd interval: an odd number of launches
    ("tiny", {"block_size_x": 48, "block_size_y": 40, "convergence_check_freq": 4, "max_iterations": 203}),                                    # a last interval of 3 without a check
    ("gx3v7", {}),                                                                          # 100 x 116: two tile columns, the second 36 wide
    ("wide", {}),                                                                           # 2112 columns: 33 tile columns
    # tripole fold: the ghost cells beyond it are formed as mirror images of their source cells (the tile walked the other way round)
    ("tiny", {"ns_boundary": 2, "block_size_x": 48, "block_size_y": 40}),                   # one block: the fold partner is the block itself
    ("tiny", {"ns_boundary": 2, "block_size_x": 24, "block_size_y": 20}),                   # 2 x 2 blocks: the partner is the other block of the top row
    ("tiny", {"ns_boundary": 2, "block_size_x": 12, "block_size_y": 10, "convergence_check_freq": 5}),   # 16 blocks, pairs and single steps mixed
    ("tiny", {"ns_boundary": 2, "block_size_x": 28, "block_size_y": 24, "partial_bottom_cells": 1}),   # padded blocks below the fold: tiles overhang beyond it
    ("tiny", {"ns_boundary": 2, "nx_global": 192, "block_size_x": 96, "block_size_y": 40}),   # two tile columns per block: mirrored partners in other tiles
])
def test_two_step_pcsi_is_bitwise_the_one_step_pcsi(pkg, name, kw):
    """pop_tuning.pcsi_two_step: two P-CSI iterations per pass over the state (k_pcsi_step_x2: x, dx, r' of a 64 x 8 tile and two rings
    of cells around it in LDS, the cells of the rings formed at their source cells) wherever no check follows.  Every value goes through
    the operations of k_pcsi_step / k_pcsi_step2: iteration counts and fields bit for bit."""
    cfg = named_config(name, solver_choice=3, **kw)
    grid = None
    if cfg.ns_boundary == 2:
        grid = synthetic_grid(cfg)
        if cfg.partial_bottom_cells:
            grid["DZBC"] = synthetic_dzbc(cfg, grid["KMT"])
    a = pkg.PopModel(cfg, tuning={"pcg_persist": 0, "pcsi_two_step": 0}, grid=grid)
    b = pkg.PopModel(cfg, tuning={"pcg_persist": 0, "pcsi_two_step": 1}, grid=grid)
    assert (a.dim("pcsi_two_step"), b.dim("pcsi_two_step")) == (0, 1)
    for step in range(5):
        a.step(); b.step()
        assert a.solver_diagnostics() == b.solver_diagnostics(), "step %d" % step
    for f in ("PSURF", "UBTROP", "VBTROP", "UVEL", "TRACER", "GRADPX"):
        assert np.array_equal(a.get(f), b.get(f)), f
    a.close(); b.close()


@pytest.mark.parametrize("name,kw,grid", [
    ("tiny", {"block_size_x": 24, "block_size_y": 20}, False),                      # 4 blocks: ghosts between blocks through the source map
    ("tiny", {"block_size_x": 48, "block_size_y": 40}, False),                      # one block, cyclic east-west
    ("tiny", {"block_size_x": 24, "block_size_y": 20, "ew_boundary": 0, "stepped_bathymetry": 1}, False),   # closed east-west: fill-value neighbours
    ("tiny", {"block_size_x": 28, "block_size_y": 24}, False),                      # padded blocks (2 x 2, last column / row short)
    ("tiny", {"block_size_x": 24, "block_size_y": 20, "convergence_check_freq": 4, "max_iterations": 203, "convergence_criterion": 1.0e-13}, False),
    ("tiny", {"ns_boundary": 2, "block_size_x": 24, "block_size_y": 20}, True),     # tripole fold inside the source map
    ("test", {}, False),                                                            # 96 blocks (more than a thread can collect: must fall back)
    ("gx3v7", {}, False),
    ("gx1v7", {}, False),                                                           # BASELINE configs[2]: 492 chunks, 246 workgroups of 2 chunks
    ("gx1v7", {"block_size_y": 96}, False),                                         # four j-band blocks in one view (the shape of the replicated solve)
    ("gx1v7", {"block_size_y": 48}, False),                                         # the eight bands of the 8-rank decomposition in one view: eight block totals per exchange
    # ChronGear (k_cg_persist): the iterations after the start-up pass as one resident launch
    ("tiny", {"solver_choice": 2, "block_size_x": 24, "block_size_y": 20}, False),
    ("tiny", {"solver_choice": 2, "block_size_x": 24, "block_size_y": 20, "ew_boundary": 0, "stepped_bathymetry": 1}, False),
    ("tiny", {"solver_choice": 2, "block_size_x": 28, "block_size_y": 24, "convergence_check_freq": 4, "max_iterations": 203, "convergence_criterion": 1.0e-13}, False),
    ("tiny", {"solver_choice": 2, "ns_boundary": 2, "block_size_x": 24, "block_size_y": 20}, True),
    ("gx3v7", {"solver_choice": 2}, False),
    ("gx1v7", {"solver_choice": 2}, False),
    ("gx1v7", {"solver_choice": 2, "block_size_y": 96}, False),
    # P-CSI (k_pcsi_persist): neighbour waits per iteration, grid-wide exchanges only at the checks
    ("tiny", {"solver_choice": 3, "block_size_x": 24, "block_size_y": 20}, False),
    ("tiny", {"solver_choice": 3, "block_size_x": 24, "block_size_y": 20, "ew_boundary": 0, "stepped_bathymetry": 1}, False),
    ("tiny", {"solver_choice": 3, "block_size_x": 28, "block_size_y": 24, "convergence_check_freq": 4, "max_iterations": 203, "convergence_check_start": 8}, False),
    ("tiny", {"solver_choice": 3, "ns_boundary": 2, "block_size_x": 24, "block_size_y": 20}, True),
    ("gx3v7", {"solver_choice": 3}, False),
    ("gx1v7", {"solver_choice": 3}, False),
    ("gx1v7", {"solver_choice": 3, "block_size_y": 96}, False),
])
def test_persistent_pcg_is_bitwise_the_fused_pcg(pkg, name, kw, grid):
    """pop_tuning.pcg_persist: the whole pcg solve of a small 2-D system as one resident launch (kernels_pcg_persist.hpp) -- the
    vectors in LDS / registers, partials and halo z exchanged through memory words that are their own flags.  Same chunk
    partials, same ordered totals, same cell arithmetic: iteration counts and every field bit for bit the two-launch fused form."""
    cfg = named_config(name, **kw)
    g = synthetic_grid(cfg) if grid else None
    a = pkg.PopModel(cfg, grid=g, tuning={"pcg_persist": 0})
    b = pkg.PopModel(cfg, grid=g)                              # the default where the view qualifies
    used = 0
    for step in range(5):
        a.step(); b.step()
        assert a.solver_diagnostics() == b.solver_diagnostics(), "step %d" % step
        used += b.dim("pcg_persist_used")
        assert a.dim("pcg_persist_used") == 0
    assert used == (0 if name == "test" else 5)
    if kw.get("block_size_y") == 48 and kw.get("solver_choice", 1) == 1:
        assert (b.dim("pcg_persist_workgroups"), b.dim("pcg_persist_chunks_per_workgroup")) == (144, 4)
    for f in ("PSURF", "UBTROP", "VBTROP", "UVEL", "TRACER", "GRADPX"):
        assert np.array_equal(a.get(f), b.get(f)), f
    a.close(); b.close()


@pytest.mark.parametrize("kw,env", [({"vmix_choice": 3, "km": 24}, {}),
                                    ({"vmix_choice": 3, "km": 62, "stepped_bathymetry": 1, "hmix_tracer": 4, "hmix_momentum": 4, "am": -1.0e22, "ah": -1.0e21}, {}),
                                    ({"vmix_choice": 3, "km": 24, "tmix_opt": 1, "time_mix_freq": 3}, {}), ({"vmix_choice": 3, "km": 24, "tmix_opt": 3}, {}),
                                    # the scratch-staged corrector and interior kernels share E3 with the look-ahead: the corrector must follow it
                                    ({"vmix_choice": 3, "km": 62, "stepped_bathymetry": 1}, {"POP_REG_THOMAS_T": "0", "POP_KPP_INTERIOR_GENERIC": "1", "POP_KPP_COL": "1"}),
                                    ({"vmix_choice": 3, "km": 20, "ldbl_diff": 1, "block_size_x": 48, "block_size_y": 40}, {"POP_KPP_COL": "0"}),
                                    # r4: with the mixed-layer-depth diagnostics (a second HMXL / HMXL_DR pair) and with Gent-McWilliams mixing, which adds to the
                                    # VDC swapped in and whose transition layer reads the HMXL swapped in
                                    ({"vmix_choice": 3, "km": 24, "kpp_diagnostics": 1}, {}),
                                    ({"vmix_choice": 3, "km": 24, "hmix_tracer": 3, "ah": 0.8e7, "gm_transition_layer": 1, "gm_kappa_type": 1, "gm_kappa_freq": 1}, {}),
                                    ({"vmix_choice": 3, "km": 20, "hmix_tracer": 3, "ah": 0.8e7, "ah_bolus": 0.5e7, "block_size_x": 48, "block_size_y": 40, "tmix_opt": 1, "time_mix_freq": 4}, {})])
def test_kpp_look_ahead_is_bitwise_neutral(pkg, orclib_built, monkeypatch, kw, env):
    """POP_KPP_AHEAD=1: pop_step computes the next step's KPP coefficients on a third stream beside the barotropic solver
    (inputs: this step's curtime fields) and the next step swaps them in.  Twelve steps -- first (Euler) step, leapfrog
    steps, averaging steps (no look-ahead: they rewrite curtime) -- must equal the run without look-ahead bit for bit, also
    when a field is read or set between steps (which drops the look-ahead in flight)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cfg = named_config("tiny", **kw)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("POP_KPP_AHEAD", mode)
        m, orc = pkg.PopModel(cfg), Oracle(cfg)
        force_kpp_case(m, orc)
        orc.close()
        for s in range(12):
            m.step()
            if s == 5:
                m.set("STF", m.get("STF", 1, 0) * 1.5, 1, 0)     # a caller changes the forcing between steps
            if s == 8:
                m.get("UVEL", 1, 0)
        out[mode] = [m.get(n, 1, 0).copy() for n in ("UVEL", "VVEL", "TRACER", "PSURF", "VVC", "HBLT", "KPP_SRC")] + [m.get("VDC", 0, 1).copy()]
        if cfg.kpp_ml_diagnostics or (cfg.hmix_tracer == 3 and cfg.gm_transition_layer):
            out[mode] += [m.get("HMXL").copy(), m.get("HMXL_DR").copy()]
        m.close()
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("kw", [{}, {"stepped_bathymetry": 1, "vmix_choice": 3, "km": 24}, {"km": 62, "vmix_choice": 3, "hmix_tracer": 4, "hmix_momentum": 4, "am": -1.0e22, "ah": -1.0e21},
                                {"lpressure_avg": 0, "tmix_opt": 1, "time_mix_freq": 3}])
def test_fused_forward_elimination_is_bitwise_neutral(pkg, monkeypatch, kw):
    """POP_TRACER_FWD=1: the forward elimination of the predictor's implicit vertical mixing runs inside the tracer
    right-hand-side kernel (k_tracer_rhs_lds<R, true> + k_impvmixt_back) instead of k_impvmixt after it: same operations,
    same order -- every field equal to the last bit, for both tile heights, land and shallow columns included."""
    cfg = named_config("tiny", **kw)
    out = {}
    for mode, rows in (("0", "8"), ("1", "8"), ("1", "4")):
        monkeypatch.setenv("POP_TRACER_FWD", mode)
        monkeypatch.setenv("POP_TRACER_LDS", rows)
        monkeypatch.setenv("POP_REG_THOMAS_T", "0")
        m = pkg.PopModel(cfg)
        for _ in range(5):
            m.step()
        out[(mode, rows)] = [m.get(n, 1, 0).copy() for n in ("UVEL", "VVEL", "PSURF")] + [m.get("TRACER", tl, n).copy() for tl in (0, 1) for n in (0, 1)]
        m.close()
    for key in (("1", "8"), ("1", "4")):
        for a, b in zip(out[("0", "8")], out[key]):
            assert np.array_equal(a, b), key
