"""Gent-McWilliams eddy transport + isopycnal diffusion of tracers (hmix_tracer = 3: hmix_gm.F90:1102-2226,
hmix_gm_submeso_share.F90:149-432) through the C ABI against the CPU oracle, phase by phase, and by properties that do not
involve the oracle."""
import numpy as np
import pytest

from popcfg import named_config, synthetic_grid
from orclib import Oracle
from test_gpu_parity import run_phases, force_kpp_case, relerr, pick, TOL_LOCAL, TOL_SOLVE

pytestmark = pytest.mark.gpu

GM = {"hmix_tracer": 3, "ah": 0.8e7}


def _steep(gpu, orc):
    """a front: isopycnal slopes from gentle to beyond the tapering limits, so every branch of the slope control is taken"""
    tlat = orc.f2("TLAT")
    for tl in (0, 1, 2):
        T = orc.f3("TRACER", tl, 0)
        z = np.arange(T.shape[1])[None, :, None, None]
        T[...] = T + 6.0 * np.tanh(8.0 * (tlat[:, None] - 0.3)) * np.exp(-z / 6.0) + 2.0 * np.sin(5.0 * tlat[:, None]) * np.exp(-z / 3.0)
        gpu.set("TRACER", T, tl=tl, n=0)


@pytest.mark.parametrize("name,kw,nsteps", [
    ("tiny", {}, 5),                                                                  # const vmix, 16 blocks; skew-flux terms cancel
    ("tiny", {"ah_bolus": 0.4e7}, 5),                                                 # ... do not cancel
    ("tiny", {"slm_b": 0.2, "gm_slope_control": 1, "stepped_bathymetry": 1}, 4),      # tanh tapering, different limits for the two diffusivities
    ("tiny", {"vmix_choice": 3, "km": 24, "stepped_bathymetry": 1}, 5),               # KPP: boundary-layer depth from HBLT, shared diffusivity array
    ("tiny", {"vmix_choice": 3, "km": 24, "ldbl_diff": 1, "ah_bolus": 1.2e7}, 4),     # two diffusivity arrays
    ("tiny", {"vmix_choice": 2, "tadvect": 2, "ah_bkg_srfbl": 0.3e7}, 4),             # Richardson, upwind3
    ("tiny", {"tadvect": 3, "block_size_x": 48, "block_size_y": 40}, 4),              # lw_lim, one block
    ("tiny", {"hmix_momentum": 4, "am": -1.0e22, "stepped_bathymetry": 1}, 4),        # del4 momentum beside it (side stream)
    ("tiny", {"tmix_opt": 3, "solver_choice": 2}, 4),                                 # Robert filter
    ("tiny", {"km": 60, "vmix_choice": 3}, 3),                                        # production level count (register Thomas kernels)
    # kappa type 'bfre' (buoyancy_frequency_dependent_profile): KAPPA_VERTICAL = N^2 / N_ref^2 below the surface diabatic layer
    ("tiny", {"gm_kappa_type": 1, "gm_kappa_freq": 2, "stepped_bathymetry": 1}, 5),    # 'once_a_day': the profile of the first step is kept until a day has ended
    ("tiny", {"gm_kappa_type": 1, "gm_kappa_freq": 1, "vmix_choice": 3, "km": 24, "stepped_bathymetry": 1, "ah_bolus": 0.5e7}, 5),   # every step, SDL = HBLT
    ("tiny", {"gm_kappa_type": 1, "gm_kappa_freq": 1, "km": 60, "tadvect": 2, "gm_slope_control": 1}, 3),
    ("gx3v7", {"gm_kappa_type": 1, "gm_kappa_freq": 1, "vmix_choice": 3}, 3),
    # transition layer (transition_layer, merged_streamfunction, apply_vertical_profile_to_isop_hor_diff): the diabatic depth is zw(1)
    # without KPP, the smoothed HMXL with it
    ("tiny", {"gm_transition_layer": 1, "stepped_bathymetry": 1}, 5),
    ("tiny", {"gm_transition_layer": 1, "vmix_choice": 3, "km": 24, "stepped_bathymetry": 1}, 5),
    ("tiny", {"gm_transition_layer": 1, "gm_kappa_type": 1, "gm_kappa_freq": 1, "vmix_choice": 3, "km": 24, "stepped_bathymetry": 1, "ah_bolus": 0.5e7, "slm_b": 0.2}, 5),   # the CESM set-up but for kappa_freq
    ("tiny", {"gm_transition_layer": 1, "gm_kappa_type": 1, "gm_kappa_freq": 2, "vmix_choice": 3, "km": 60, "gm_slope_control": 1, "tadvect": 2}, 3),
    ("gx3v7", {"gm_transition_layer": 1, "gm_kappa_type": 1, "gm_kappa_freq": 1, "vmix_choice": 3}, 3),
    # kappa_freq 'once_a_day': recomputed at the first step after a day has ended (two day boundaries inside the run)
    ("tiny", {"gm_kappa_type": 1, "gm_kappa_freq": 2, "steps_per_day": 6, "time_mix_freq": 4, "vmix_choice": 3, "km": 20}, 20),   # avgfit: the fit interval is the day
    ("tiny", {"gm_kappa_type": 1, "gm_kappa_freq": 2, "gm_transition_layer": 1, "steps_per_day": 8, "tmix_opt": 3}, 18),          # Robert filter: every eighth step ends a day
    # the remaining switches of hmix_gm_nml: horizontal diffusivity of the boundary layer from KAPPA_ISOP, and in the bottom half of the bottom cell
    ("tiny", {"gm_kappa_bkg_srfbl": 1, "ah_bkg_bottom": 0.2e7, "stepped_bathymetry": 1, "gm_kappa_type": 1, "gm_kappa_freq": 1}, 4),
    ("tiny", {"gm_kappa_bkg_srfbl": 1, "ah_bkg_bottom": 0.2e7, "gm_transition_layer": 1, "vmix_choice": 3, "km": 24, "stepped_bathymetry": 1}, 4),
    # slope control 'clip' and 'Gerd' (limits low enough that the front is beyond them), kappa type 'depth'
    ("tiny", {"gm_slope_control": 2, "slm_r": 1.0e-3, "slm_b": 1.0e-3, "stepped_bathymetry": 1, "ah_bolus": 0.5e7}, 4),
    ("tiny", {"gm_slope_control": 3, "slm_r": 1.0e-3, "slm_b": 2.0e-3, "vmix_choice": 3, "km": 24}, 4),
    ("tiny", {"gm_kappa_type": 2, "kappa_depth_1": 0.2, "kappa_depth_2": 0.8, "kappa_depth_scale": 1.0e5, "ah_bolus": 0.5e7, "stepped_bathymetry": 1}, 4),
    ("test", {"stepped_bathymetry": 1}, 3),                                           # 96 blocks
    ("gx3v7", {"vmix_choice": 3}, 3),
])
def test_gm_step_phases_match_oracle(pkg, orclib_built, name, kw, nsteps):
    cfg = named_config(name, **dict(GM, **kw))
    gpu, orc = pkg.PopModel(cfg), Oracle(cfg)
    if cfg.vmix_choice == 3:
        force_kpp_case(gpu, orc)
    _steep(gpu, orc)
    tol = TOL_LOCAL
    for s in range(1, nsteps + 1):
        run_phases(gpu, orc, s, tol)
        a, b = gpu.get("VDC", n=0), orc.vdc(0)          # with the isopycnal part added (hmix_gm.F90:1725-1748)
        assert relerr(pick(gpu, a, True), pick(gpu, b, True)) <= max(tol, TOL_LOCAL) * 10, "step %d VDC" % s
        tol = TOL_SOLVE
    # the front is steep enough for the tapering to act and gentle enough elsewhere for the isopycnal part to be there
    vd = orc.vdc(0).copy()
    gpu.close(); orc.close()
    if cfg.vmix_choice == 1:
        assert (vd > 10.0 * cfg.const_vdc).sum() > 50 and (vd == cfg.const_vdc).sum() > 50


@pytest.mark.parametrize("kw", [{"ns_boundary": 2}, {"ns_boundary": 2, "vmix_choice": 3, "km": 24, "ah_bolus": 0.5e7}, {"ns_boundary": 0, "ew_boundary": 0},
                                {"ns_boundary": 2, "vmix_choice": 3, "km": 24, "gm_transition_layer": 1, "gm_kappa_type": 1, "gm_kappa_freq": 1}])   # the filter over HMXL next to the fold
def test_gm_on_a_caller_grid(pkg, orclib_built, kw):
    """tripole fold and closed boundaries on a grid supplied by the caller (the scheme exchanges nothing: everything is formed from the
    mix-time tracers' own ghost cells)"""
    cfg = named_config("tiny", **dict(GM, **kw))
    grid = synthetic_grid(cfg)
    gpu, orc = pkg.PopModel(cfg, grid=grid), Oracle(cfg, grid=grid)
    if cfg.vmix_choice == 3:
        force_kpp_case(gpu, orc)
    tol = TOL_LOCAL
    for s in range(1, 5):
        run_phases(gpu, orc, s, tol)
        tol = TOL_SOLVE
    gpu.close(); orc.close()


@pytest.mark.parametrize("kw", [{}, {"ah_bolus": 0.3e7, "slm_b": 0.2}, {"vmix_choice": 3, "km": 20, "tadvect": 2}, {"gm_kappa_type": 1, "gm_kappa_freq": 1, "vmix_choice": 3, "km": 20},
                                {"gm_transition_layer": 1, "gm_kappa_type": 1, "gm_kappa_freq": 1, "vmix_choice": 3, "km": 20}])
def test_gm_conserves_tracer_content(pkg, kw):
    """every term is a flux form -- east / north fluxes shared by neighbours, the flux through the bottom face of a level is the one
    through the top face of the next, the isopycnal addition to VDC goes through the (conservative) implicit solve: with no surface
    flux the volume integral of a tracer does not change (surface layer weighted with its actual thickness)"""
    cfg = named_config("tiny", stepped_bathymetry=1, **dict(GM, **kw))
    m = pkg.PopModel(cfg)
    o = Oracle(cfg)
    dz = o.v1("dz")[1:cfg.km + 1].copy()
    o.close()
    tarea, kmt = m.get("TAREA"), m.geti("KMT")
    k = np.arange(1, cfg.km + 1)[None, :, None, None]
    wet = (k <= kmt[:, None])[..., 2:-2, 2:-2]

    def content(n):
        T = m.get("TRACER", 1, n)[..., 2:-2, 2:-2]
        eta = m.get("PSURF", 1)[..., 2:-2, 2:-2] / 980.6
        thick = np.broadcast_to(dz[None, :, None, None], T.shape).copy()
        thick[:, 0] = thick[:, 0] + eta
        return float((np.where(wet, T * thick, 0.0) * tarea[:, None, 2:-2, 2:-2]).sum())

    for _ in range(3):
        m.step()
    c0 = [content(n) for n in (0, 1)]
    for _ in range(4):
        m.step()
    for n in (0, 1):
        assert abs(content(n) - c0[n]) <= 2e-9 * abs(c0[n]), (kw, n)
    m.close()


def test_gm_leaves_level_isopycnals_alone(pkg):
    """T(z), S(z) at rest: no slope, no flux, nothing added to VDC -- the first step gives bit for bit the tracers and the
    diffusivity of del2 mixing, which has nothing to mix either"""
    out = {}
    for hm in (2, 3):
        m = pkg.PopModel(named_config("tiny", hmix_tracer=hm, ah=0.8e7, stepped_bathymetry=1, init_ts_perturbation=0.0))
        m.step()
        out[hm] = [m.get("TRACER", 1, n).copy() for n in (0, 1)] + [m.get("VDC", n=0).copy()]
        m.close()
    for a, b in zip(out[2], out[3]):
        assert np.array_equal(a, b)


def test_gm_buoyancy_frequency_profile_is_what_it_says(pkg, orclib_built):
    """kappa type 'bfre': below the surface diabatic layer the isopycnal diffusivity follows N^2 / N_ref^2, clamped to [0.1, 1]
    (hmix_gm.F90:3011-3180) -- so in the weakly stratified deep ocean of the Levitus profile the isopycnal addition to VDC must be
    several times smaller than with constant kappa, and identical in the levels above the reference level (KAPPA_VERTICAL = 1 there)"""
    out = {}
    for kt in (0, 1):
        m = pkg.PopModel(named_config("tiny", gm_kappa_type=kt, gm_kappa_freq=1, km=24, **GM))
        o = Oracle(named_config("tiny", gm_kappa_type=kt, gm_kappa_freq=1, km=24, **GM))
        _steep(m, o); o.close()
        m.step()
        out[kt] = m.get("VDC", n=0) - m.cfg.const_vdc
        m.close()
    a, b = out[0][:, 1:3], out[1][:, 1:3]                  # bottoms of levels 1, 2: they read KAPPA_VERTICAL(1..3) = 1 (reference level K_MIN = 2 below SDL = zw(1))
    assert np.abs(a).max() > 1.0 and np.array_equal(a, b)
    deep_c, deep_b = out[0][:, 15:22], out[1][:, 15:22]
    big = deep_c > 1.0
    assert big.sum() > 50 and np.median(deep_b[big] / deep_c[big]) < 0.5 and (deep_b[big] >= 0.1 * deep_c[big] * (1 - 1e-12)).all()


def test_gm_transition_layer_switches_isopycnal_diffusion_off_in_the_diabatic_layer(pkg, orclib_built):
    """With the transition layer on, KAPPA_ISOP is zero in every half cell whose middle lies within the diabatic depth
    (apply_vertical_profile_to_isop_hor_diff, hmix_gm.F90:3776-3780) -- so the isopycnal addition to VDC at the bottom of level k,
    which is made of KAPPA_ISOP of the two half cells around that interface, must be exactly zero wherever the lower of the two
    (middle at zt(k+1) - dz(k+1)/4) is within the diabatic depth, and present below the layer.  The diabatic depth is recomputed
    here from the library's own HMXL with the 1-1-4-1-1 filter of smooth_hblt (vmix_kpp.F90:3797-3845), independently."""
    kw = dict(GM, gm_transition_layer=1, vmix_choice=3, km=24, block_size_x=48, block_size_y=40)
    cfg = named_config("tiny", **kw)
    m, o = pkg.PopModel(cfg), Oracle(cfg)
    force_kpp_case(m, o); _steep(m, o)
    zt, dz = o.v1("zt").copy(), o.v1("dz").copy()
    o.close()
    # the KPP coefficients before the isopycnal part is added: the same state with del2 mixing
    ref, o2 = pkg.PopModel(named_config("tiny", **dict(kw, hmix_tracer=2, gm_transition_layer=0, kpp_diagnostics=1))), Oracle(named_config("tiny", **dict(kw, hmix_tracer=2, gm_transition_layer=0)))
    force_kpp_case(ref, o2); _steep(ref, o2); o2.close()
    for x in (m, ref):
        x.time_manager(); x.dhdt(); x.baroclinic_driver()
    add = m.get("VDC", n=0) - ref.get("VDC", n=0)
    hm, kmt = m.get("HMXL")[0], m.geti("KMT")[0]
    assert np.array_equal(hm, ref.get("HMXL")[0]) and hm.max() > 3.0 * zt[2]
    dd = hm.copy()
    ny, nx = hm.shape
    for j in range(1, ny - 1):
        for i in range(1, nx - 1):
            if kmt[j, i] == 0:
                continue
            cw = ce = cn = cs = 0.125; cc = 0.5
            if kmt[j, i - 1] == 0: cc += cw; cw = 0.0
            if kmt[j, i + 1] == 0: cc += ce; ce = 0.0
            if kmt[j - 1, i] == 0: cc += cs; cs = 0.0
            if kmt[j + 1, i] == 0: cc += cn; cn = 0.0
            dd[j, i] = cw * hm[j, i - 1] + ce * hm[j, i + 1] + cs * hm[j - 1, i] + cn * hm[j + 1, i] + cc * hm[j, i]
            dd[j, i] = min(dd[j, i], zt[kmt[j, i]])
    inner = np.zeros_like(hm, dtype=bool); inner[2:-2, 2:-2] = True
    nzero = npos = 0
    for k in range(1, cfg.km - 1):                     # VDC index k = the bottom of level k
        wet = inner & (k < kmt)
        within = wet & (zt[k + 1] - 0.25 * dz[k + 1] <= dd)
        below = wet & (zt[k] + 0.25 * dz[k] > dd + 3.0 * dz[k])
        assert (add[0, k][within] == 0.0).all(), k
        nzero += int(within.sum()); npos += int((add[0, k][below] > 0.0).sum())
    assert nzero > 200 and npos > 200
    m.close(); ref.close()


@pytest.mark.parametrize("kw", [{"ah_bolus": 0.4e7, "stepped_bathymetry": 1}, {"gm_transition_layer": 1, "gm_kappa_type": 1, "gm_kappa_freq": 1, "vmix_choice": 3, "km": 24}])
def test_gm_stored_stream_function_terms_are_bitwise_the_re_derived_ones(pkg, monkeypatch, kw):
    """k_gm_sf stores SF_SLX / SF_SLY of every half cell once (the reference's arrays); POP_GM_SF_STORED=0 re-derives them at every use
    in the flux kernel: the same function, so the same bits"""
    out = {}
    for st in ("1", "0"):
        monkeypatch.setenv("POP_GM_SF_STORED", st)
        m = pkg.PopModel(named_config("tiny", **dict(GM, **kw)))
        for _ in range(4):
            m.step()
        out[st] = [m.get("TRACER", 1, n).copy() for n in (0, 1)] + [m.get("VDC", n=0).copy(), m.get("UVEL", 1).copy()]
        m.close()
    for a, b in zip(out["1"], out["0"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("kw", [{"ah_bolus": 0.4e7, "stepped_bathymetry": 1},                                                   # with cancellation of the skew terms
                                {"gm_transition_layer": 1, "gm_kappa_type": 1, "gm_kappa_freq": 1, "vmix_choice": 3, "km": 24},       # CESM's set-up: without
                                {"ah_bolus": 0.4e7, "block_size_x": 20, "block_size_y": 16, "ldbl_diff": 1, "vmix_choice": 3, "km": 20},   # padded blocks, tiles overhang; two VDC arrays
                                {"gm_transition_layer": 1, "block_size_x": 48, "block_size_y": 40, "ew_boundary": 0, "km": 62}])
def test_gm_flux_tile_is_bitwise_the_cell_kernel(pkg, kw):
    """k_gm_flux_tile (r4: every horizontal face flux formed once per 64 x 4 tile -- own east / north face, west by lane shuffle, south
    through LDS) against k_gm_flux (pop_tuning.gm_flux_tile = 0: each face flux evaluated in both cells): the same six flux values
    combined in the same order, so tendencies and the isopycnal VDC agree to the last bit"""
    out = {}
    for tile in (1, 0):
        m = pkg.PopModel(named_config("tiny", **dict(GM, **kw)), tuning={"gm_flux_tile": tile})
        for _ in range(4):
            m.step()
        out[tile] = [m.get("TRACER", 1, n).copy() for n in (0, 1)] + [m.get("VDC", n=0).copy(), m.get("VDC", n=1).copy(), m.get("UVEL", 1).copy()]
        m.close()
    for a, b in zip(out[1], out[0]):
        assert np.array_equal(a, b)
    assert np.abs(out[1][0]).max() > 0.0


@pytest.mark.parametrize("kw", [{"ah_bolus": 0.5e7, "stepped_bathymetry": 1},
                                {"gm_transition_layer": 1, "gm_kappa_type": 1, "gm_kappa_freq": 1, "vmix_choice": 3, "km": 24, "stepped_bathymetry": 1},
                                {"block_size_x": 48, "block_size_y": 40}])
def test_gm_bolus_velocity_diagnostics(pkg, orclib_built, kw):
    """diag_gm_bolus (hmix_gm.F90:2079-2151): U_ISOP, V_ISOP, WTOP_ISOP of every level against the oracle, and what an eddy-induced
    velocity derived from a stream function that vanishes at the surface and at the bottom must do whatever the stream function is:
    its transport through every face column integrates to zero, and it is non-divergent (the vertical velocity at the top of
    a level is minus the horizontal divergence summed over the levels above ... and comes out zero below the bottom level)."""
    cfg = named_config("tiny", gm_diag_bolus=1, **dict(GM, **kw))
    gpu, orc = pkg.PopModel(cfg), Oracle(cfg)
    if cfg.vmix_choice == 3:
        force_kpp_case(gpu, orc)
    _steep(gpu, orc)
    tol = TOL_LOCAL
    for s in range(1, 4):
        run_phases(gpu, orc, s, tol)
        tol = TOL_SOLVE
        for f in ("UISOP", "VISOP", "WISOP"):
            a, b = pick(gpu, gpu.get(f), True), pick(gpu, orc.f3(f), True)
            assert np.abs(b).max() > 0.0, f
            assert relerr(a, b) <= tol * 100, "%s step %d: %g" % (f, s, relerr(a, b))
    dz = orc.v1("dz")[1:cfg.km + 1]
    u, v, wv = gpu.get("UISOP"), gpu.get("VISOP"), gpu.get("WISOP")
    kmt = gpu.geti("KMT")
    for x in (u, v):
        col = (x * dz[None, :, None, None]).sum(axis=1)
        assert np.abs(col).max() <= 1e-12 * (np.abs(x) * dz[None, :, None, None]).sum(axis=1).max()
    assert np.abs(u).max() > 1e-4 and np.abs(wv).max() > 1e-8
    k = np.arange(1, cfg.km + 1)[None, :, None, None]
    assert not wv[(k > kmt[:, None]) & np.ones_like(wv, dtype=bool)].any()      # nothing at the top of levels below the bottom
    gpu.close(); orc.close()
