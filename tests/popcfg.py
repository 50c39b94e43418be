"""Shared ctypes mirror of pop_config / orc_config and the named BASELINE configs.

Used by tests/, bench.py and __graft_entry__.py.  The struct layout is the
one declared (separately) in include/pop_amd.h and oracle/pop_oracle.h.
"""
import ctypes as C


class PopConfig(C.Structure):
    _fields_ = [
        ("struct_version", C.c_int),
        ("nx_global", C.c_int), ("ny_global", C.c_int), ("km", C.c_int), ("nt", C.c_int),
        ("block_size_x", C.c_int), ("block_size_y", C.c_int),
        ("ew_boundary", C.c_int), ("ns_boundary", C.c_int),
        ("hmix_momentum", C.c_int), ("hmix_tracer", C.c_int), ("lvariable_hmix", C.c_int),
        ("vmix_choice", C.c_int), ("tadvect", C.c_int), ("solver_choice", C.c_int),
        ("max_iterations", C.c_int), ("convergence_check_freq", C.c_int),
        ("tmix_opt", C.c_int), ("time_mix_freq", C.c_int), ("steps_per_day", C.c_int),
        ("lbouss_correct", C.c_int), ("lpressure_avg", C.c_int), ("impcor", C.c_int),
        ("reset_to_freezing", C.c_int),
        ("lrich", C.c_int), ("ldbl_diff", C.c_int), ("lshort_wave", C.c_int), ("lcheckekmo", C.c_int),
        ("num_v_smooth_Ri", C.c_int),
        ("maxlanczosstep", C.c_int), ("convergence_check_start", C.c_int), ("preconditioner_choice", C.c_int),
        ("stepped_bathymetry", C.c_int), ("distribution_type", C.c_int), ("kpp_ml_diagnostics", C.c_int),
        ("sw_absorption_type", C.c_int), ("jerlov_water_type", C.c_int), ("lsw_absorb", C.c_int),
        ("partial_bottom_cells", C.c_int),
        ("gm_slope_control", C.c_int), ("gm_kappa_type", C.c_int), ("gm_kappa_freq", C.c_int),
        ("am", C.c_double), ("ah", C.c_double),
        ("const_vvc", C.c_double), ("const_vdc", C.c_double),
        ("convect_diff", C.c_double), ("convect_visc", C.c_double), ("bottom_drag", C.c_double),
        ("aidif", C.c_double),
        ("rich_bckgrnd_vvc", C.c_double), ("rich_bckgrnd_vdc", C.c_double), ("rich_mix", C.c_double),
        ("bckgrnd_vdc1", C.c_double), ("bckgrnd_vdc2", C.c_double), ("bckgrnd_vdc_dpth", C.c_double),
        ("bckgrnd_vdc_linv", C.c_double), ("Prandtl", C.c_double), ("kpp_rich_mix", C.c_double),
        ("convergence_criterion", C.c_double),
        ("init_ts_perturbation", C.c_double), ("robert_alpha", C.c_double), ("robert_nu", C.c_double),
        ("lanczos_convergence_criterion", C.c_double),
        ("ah_bolus", C.c_double), ("ah_bkg_srfbl", C.c_double), ("slm_r", C.c_double), ("slm_b", C.c_double),
        ("gm_transition_layer", C.c_int), ("gm_diag_bolus", C.c_int), ("gm_kappa_bkg_srfbl", C.c_int), ("reserved_i", C.c_int * 1),
        ("ah_bkg_bottom", C.c_double), ("kappa_depth_1", C.c_double), ("kappa_depth_2", C.c_double), ("kappa_depth_scale", C.c_double),
    ]


def base_config(**kw):
    """Defaults = the reference's code defaults for the options this path supports
    (vertical_mix.F90:233-240, POP_SolversMod.F90:578-662, pressure_grad.F90:118-119,
    baroclinic.F90:208, vmix_rich.F90:108-110, vmix_const.F90:101-102)."""
    c = PopConfig()
    c.struct_version = 5
    c.nt = 2
    c.ew_boundary, c.ns_boundary = 1, 0
    c.hmix_momentum = c.hmix_tracer = 2
    c.lvariable_hmix = 0
    c.vmix_choice, c.tadvect, c.solver_choice = 1, 1, 1
    c.max_iterations, c.convergence_check_freq = 1000, 10
    c.tmix_opt, c.time_mix_freq = 2, 17
    c.lbouss_correct, c.lpressure_avg, c.impcor, c.reset_to_freezing = 0, 1, 1, 1
    c.lrich, c.ldbl_diff, c.lshort_wave, c.lcheckekmo, c.num_v_smooth_Ri = 1, 0, 0, 0, 1
    c.const_vvc = c.const_vdc = 0.25
    c.convect_diff = c.convect_visc = 1000.0
    c.bottom_drag, c.aidif = 1.0e-3, 1.0
    c.rich_bckgrnd_vvc, c.rich_bckgrnd_vdc, c.rich_mix = 1.0, 0.1, 50.0
    c.bckgrnd_vdc1, c.bckgrnd_vdc2, c.bckgrnd_vdc_dpth, c.bckgrnd_vdc_linv = 0.1, 0.0, 2500.0e2, 4.5e-5
    c.Prandtl, c.kpp_rich_mix = 10.0, 50.0
    c.convergence_criterion = 1.0e-12
    c.init_ts_perturbation = 1.0e-2          # init T perturbation amplitude
    _apply(c, kw)
    return c


def _apply(c, kw):
    """Set fields by name (a few historical aliases are kept: precond_choice, distribution, kpp_diagnostics)."""
    alias = {"precond_choice": "preconditioner_choice", "distribution": "distribution_type", "kpp_diagnostics": "kpp_ml_diagnostics"}
    for k, v in kw.items():
        k = alias.get(k, k)
        if k == "lsw_absorb":
            v = int(bool(v))
        if not hasattr(c, k):
            raise AttributeError("pop_config has no field %r" % k)
        setattr(c, k, v)


def named_config(name, **kw):
    """BASELINE.json configs."""
    if name == "test":      # test_domain_size 192x128x20, 16x16 blocks, Richardson vmix
        c = base_config(nx_global=192, ny_global=128, km=20, block_size_x=16, block_size_y=16,
                        vmix_choice=2, steps_per_day=24, am=1.0e8, ah=1.0e7)
    elif name == "tiny":    # small multi-block case for fast CPU parity
        c = base_config(nx_global=48, ny_global=40, km=16, block_size_x=12, block_size_y=10,
                        vmix_choice=1, steps_per_day=24, am=3.0e9, ah=1.0e7)
    elif name == "wide":    # many 64-/256-wide tile columns in i: exercises the large-grid XCD tile order
        c = base_config(nx_global=2112, ny_global=16, km=16, block_size_x=2112, block_size_y=16,
                        vmix_choice=1, steps_per_day=96, am=1.0e7, ah=1.0e6)
    elif name == "gx3v7":
        c = base_config(nx_global=100, ny_global=116, km=60, block_size_x=100, block_size_y=116,
                        vmix_choice=1, steps_per_day=12, am=3.0e9, ah=1.0e7)
    elif name == "gx1v7":
        c = base_config(nx_global=320, ny_global=384, km=60, block_size_x=320, block_size_y=384,
                        vmix_choice=3, steps_per_day=24, am=0.5e8, ah=0.6e7,
                        convergence_criterion=1.0e-13)
    elif name == "tx0.1v3":
        c = base_config(nx_global=3600, ny_global=2400, km=62, block_size_x=3600, block_size_y=2400,
                        vmix_choice=3, steps_per_day=300, hmix_momentum=4, hmix_tracer=4,
                        lvariable_hmix=1, am=-27.0e17, ah=-3.0e17, convergence_criterion=1.0e-13)
    else:
        raise KeyError(name)
    _apply(c, kw)
    return c


def synthetic_grid(c, lat_span=160.0, wobble=0.04, stepped=True, kmt=True):
    """Global arrays in the layout of the reference's horiz_grid_file / topography_file records (grid.F90:1314-1542,
    :2025-2107), shape (ny_global, nx_global): a lat-lon grid over +-lat_span/2 degrees whose spacings also vary with
    longitude (so that the mirrored column of a tripole ghost row is a different number), ocean up to the northern
    edge (the fold of a tripole decomposition runs through water) and, optionally, stepped bathymetry.  TEST DATA, not
    a physical tripole grid: the code under test only ever sees arrays."""
    import numpy as np
    nx, ny, km = c.nx_global, c.ny_global, c.km
    radius, radian = 6370.0e5, 180.0 / np.pi
    i = np.arange(1, nx + 1, dtype=np.float64)[None, :]
    j = np.arange(1, ny + 1, dtype=np.float64)[:, None]
    dlat, dlon = lat_span / ny, 360.0 / nx
    ulat = (-lat_span / 2 + j * dlat) / radian + 0.0 * i
    lon = i * dlon + 0.0 * j
    ulon = np.where(lon > 180.0, lon - 360.0, lon) / radian
    east = 1.0 + wobble * np.cos(2.0 * np.pi * i / nx)            # E faces / NE corners: symmetric under i -> nx - i
    north = 1.0 + wobble * np.cos(2.0 * np.pi * (i - 0.5) / nx)   # N faces / centres:    symmetric under i -> nx + 1 - i
    cell, cellx = dlat * radius / radian, dlon * radius / radian
    g = {"ULAT": ulat, "ULON": ulon,
         "HTN": cellx * np.cos(ulat) * north, "HTE": cell * east + 0.0 * j,
         "HUS": cellx * np.cos((-lat_span / 2 + (j - 0.5) * dlat) / radian) * east, "HUW": cell * north + 0.0 * j,
         "ANGLE": np.zeros((ny, nx))}
    if kmt:
        latd, lond = ulat * radian, np.where(ulon < 0, ulon * radian + 360.0, ulon * radian)
        k = np.full((ny, nx), km, dtype=np.int32)
        k[(latd > -35.0) & (lond > 210.0) & (lond < 250.0)] = 0
        k[(latd > 25.0) & (latd < 60.0) & (lond > 210.0) & (lond < 330.0)] = 0
        k[(latd > -60.0) & (latd < 55.0) & (lond > 110.0) & (lond < 150.0)] = 0
        k[latd < -70.0] = 0
        if stepped:
            ii, jj = np.arange(1, nx + 1)[None, :], np.arange(1, ny + 1)[:, None]
            cut = ((ii // 3) * 5 + (jj // 2) * 3) % (km // 2 + 1)
            k = np.where(k > 0, np.maximum(3, km - cut), 0).astype(np.int32)
        g["KMT"] = k
    return {n: np.ascontiguousarray(a) for n, a in g.items()}


def synthetic_dzbc(c, kmt, seed=11):
    """A bottom_cell_file record (grid.F90:2116-2186) for the KMT record `kmt` (ny_global, nx_global): the thickness of the bottom T
    cell of every ocean column, a random fraction in [0.2, 1] of dz(KMT) [cm]; 0 on land.  TEST DATA.  dz restates the
    reference's internal vertical grid (grid.F90:1565-1640) only to scale the numbers; any positive thickness would do."""
    import numpy as np
    km, zmax, dz_sfc, dz_deep, eps = c.km, 5500.0, 25.0, 400.0, 1.0e-10

    def profile(zl):
        dz, depth = [], 0.0
        for _ in range(km):
            r = depth / zl
            dz.append(dz_deep - (dz_deep - dz_sfc) * np.exp(-(r * r))); depth += dz[-1]
        return depth, dz
    zl0, zl1 = eps, zmax
    d0 = profile(zl0)[0]
    dzv = profile(zl1)[1]
    while (zl1 - zl0) / zmax > eps:
        zl = zl0 + 0.5 * (zl1 - zl0)
        d, dzv = profile(zl)
        if (d0 - zmax) * (d - zmax) < 0.0:
            zl1 = zl
        else:
            d0, zl0 = d, zl
    dz = np.concatenate([[0.0], np.array(dzv) * 100.0])
    rng = np.random.default_rng(seed)
    frac = 0.2 + 0.8 * rng.random(kmt.shape)
    return np.ascontiguousarray(np.where(kmt > 0, frac * dz[np.clip(kmt, 0, km)], 0.0))
