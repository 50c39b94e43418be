"""The independent pins of tests/pins.py against the DEVICE library through the C ABI (pop_run_phase): LAPACK for the
implicit vertical solves -- generic kernels (km = 16, 20) and the column-in-registers kernels (km = 60, 62), with shallow
and land columns --, closed forms for the pressure gradient and horizontal diffusion."""
import numpy as np
import pytest

import pins
from popcfg import named_config

pytestmark = pytest.mark.gpu

CASES = [
    ("const-km16-stepped", dict(stepped_bathymetry=1)),
    ("kpp-km20-stepped", dict(vmix_choice=3, km=20, stepped_bathymetry=1)),
    ("kpp-km60-stepped", dict(vmix_choice=3, km=60, stepped_bathymetry=1)),      # k_impvmixt_reg<60>, k_impvmixu_reg<60>
    ("kpp-dd-km60-stepped", dict(vmix_choice=3, km=60, ldbl_diff=1, stepped_bathymetry=1)),   # double diffusion: a diffusivity array per tracer class
    ("kpp-km62-del4-stepped", dict(vmix_choice=3, km=62, stepped_bathymetry=1, hmix_tracer=4, hmix_momentum=4, am=-1.0e19, ah=-1.0e19)),
    ("rich-flat", dict(vmix_choice=2)),
    ("pbc-const-km16-stepped", dict(stepped_bathymetry=1, partial_bottom_cells=1)),
    ("pbc-kpp-dd-km20-stepped", dict(vmix_choice=3, km=20, ldbl_diff=1, stepped_bathymetry=1, partial_bottom_cells=1)),
    ("pbc-kpp-km62-del4-stepped", dict(vmix_choice=3, km=62, stepped_bathymetry=1, partial_bottom_cells=1, hmix_tracer=4, hmix_momentum=4, am=-1.0e19, ah=-1.0e19)),
]


@pytest.fixture(params=CASES, ids=[c[0] for c in CASES])
def adapter(request, pkg):
    A = pins.GpuAdapter(pkg, named_config("tiny", **request.param[1]))
    yield A
    A.close()


def test_impvmixt_solves_its_tridiagonal_system(adapter):
    pins.check_impvmixt(adapter, np.random.default_rng(11))


def test_impvmixt_correct_solves_its_tridiagonal_system(adapter):
    pins.check_impvmixt_correct(adapter, np.random.default_rng(12))


def test_impvmixu_and_mean_removal_match_lapack(adapter):
    pins.check_impvmixu(adapter, np.random.default_rng(13))


@pytest.mark.parametrize("env", [{}, {"POP_GENERIC_THOMAS": "1"}], ids=["register-kernels", "generic-kernels"])
def test_thomas_kernel_variants_km60(pkg, env, monkeypatch):
    """both implementations of the km = 60 solves against LAPACK"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for check, seed in ((pins.check_impvmixt, 21), (pins.check_impvmixt_correct, 22), (pins.check_impvmixu, 23)):
        A = pins.GpuAdapter(pkg, named_config("tiny", vmix_choice=3, km=60, stepped_bathymetry=1))
        check(A, np.random.default_rng(seed))
        A.close()


@pytest.mark.parametrize("kw", [dict(), dict(stepped_bathymetry=1), dict(impcor=0), dict(hmix_momentum=4, am=-1.0e19), dict(km=60)],
                         ids=["flat", "stepped", "explicit-coriolis", "del4", "km60"])
def test_pressure_gradient_closed_forms(kw, pkg, monkeypatch):
    for lds in ("8", "0"):                       # LDS-tiled and direct-load momentum kernels
        monkeypatch.setenv("POP_MOMENTUM_LDS", lds)
        A = pins.GpuAdapter(pkg, named_config("tiny", **kw))
        pins.check_gradp(A)
        A.close()


@pytest.mark.parametrize("kw", [dict(), dict(hmix_tracer=4, ah=-1.0e19), dict(hmix_tracer=4, ah=-1.0e19, lvariable_hmix=1), dict(vmix_choice=3, km=20)],
                         ids=["del2", "del4", "del4-variable", "del2-kpp"])
def test_tracer_diffusion_of_a_quadratic_field(kw, pkg, monkeypatch):
    for lds in ("8", "0"):                       # LDS-tiled and direct-load tracer kernels
        monkeypatch.setenv("POP_TRACER_LDS", lds)
        A = pins.GpuAdapter(pkg, named_config("tiny", block_size_x=48, block_size_y=40, **kw))
        pins.check_hdifft(A)
        A.close()


# ---- KPP against Large, McWilliams & Doney (1994) and advection against exact flux divergences (tests/pins.py) ----------
KPP_PIN = dict(vmix_choice=3, lrich=0, bckgrnd_vdc2=0.0, block_size_x=48, block_size_y=40)


@pytest.mark.parametrize("km", [20, 60], ids=["km20-generic-kernels", "km60-register-kernels"])
@pytest.mark.parametrize("nu0", [0.0, 2000.0], ids=["no-interior-mixing", "uniform-interior-nu"])
@pytest.mark.parametrize("regime", ["stable", "weak", "strong"])
def test_kpp_velocity_scales_and_shape_function(regime, nu0, km, pkg):
    A = pins.GpuAdapter(pkg, named_config("tiny", km=km, bckgrnd_vdc1=nu0, **KPP_PIN))
    pins.check_kpp_scales_and_shape(A, regime, nu0)
    A.close()


@pytest.mark.parametrize("km", [20, 60])
def test_kpp_boundary_layer_depth_of_a_two_layer_column(km, pkg):
    A = pins.GpuAdapter(pkg, named_config("tiny", vmix_choice=3, km=km, block_size_x=48, block_size_y=40))
    pins.check_kpp_hblt_two_layer(A)
    A.close()


@pytest.mark.parametrize("kw", [dict(tadvect=1), dict(tadvect=2), dict(tadvect=1, vmix_choice=3, km=60), dict(tadvect=2, hmix_tracer=4)],
                         ids=["centred", "upwind3", "centred-kpp-km60", "upwind3-del4"])
def test_advection_of_a_linear_field_is_the_exact_flux_divergence(kw, pkg, monkeypatch):
    for lds in ("4", "0"):                       # LDS-tiled and direct-load tracer kernels
        monkeypatch.setenv("POP_TRACER_LDS", lds)
        A = pins.GpuAdapter(pkg, named_config("tiny", ah=0.0, block_size_x=48, block_size_y=40, **kw))
        pins.check_advt_linear(A)
        A.close()


def test_limited_advection_creates_no_new_extremum(pkg):
    A = pins.GpuAdapter(pkg, named_config("tiny", ah=0.0, tadvect=3, block_size_x=48, block_size_y=40))
    pins.check_lw_lim_monotone(A)
    A.close()


@pytest.mark.parametrize("ah_bolus", [0.0, 0.3e7])
def test_gm_fluxes_of_a_linear_field_are_the_closed_form(pkg, orclib_built, ah_bolus):
    A = pins.GpuAdapter(pkg, named_config("tiny", hmix_tracer=3, ah=0.8e7, ah_bolus=ah_bolus, km=20, block_size_x=48, block_size_y=40))
    pins.check_gm_linear(A)
    A.close()


@pytest.mark.parametrize("control,slm_b,ah_bolus", [(0, 0.0, 0.0), (0, 0.2, 0.5e7), (1, 0.0, 0.0), (1, 0.2, 0.5e7), (3, 0.0, 0.0), (3, 0.2, 0.5e7), (2, 0.0, 0.0)])
def test_gm_slope_tapers_on_a_constructed_slope_field(pkg, orclib_built, control, slm_b, ah_bolus):
    """the device kernels (k_gm_coeffs, k_gm_flux, k_gm_bolus) against the closed forms of the four slope tapers: pins.check_gm_tapers"""
    A = pins.GpuAdapter(pkg, named_config("tiny", hmix_tracer=3, ah=0.8e7, ah_bolus=ah_bolus, km=20, block_size_x=48, block_size_y=40,
                                          gm_slope_control=control, slm_b=slm_b, gm_diag_bolus=1))
    pins.check_gm_tapers(A)
    A.close()


@pytest.mark.parametrize("kw", [{}, {"gm_slope_control": 1, "slm_b": 0.2}], ids=["notanh", "tanh-diff-tapering"])
def test_gm_transition_layer_depths_and_merged_streamfunction(pkg, orclib_built, kw):
    """the device kernels (k_gm_transition_layer, k_gm_msf_column, k_gm_sf) against the walk and the structural properties of the merged
    stream function: pins.check_gm_transition_layer"""
    A = pins.GpuAdapter(pkg, named_config("tiny", hmix_tracer=3, ah=0.8e7, ah_bolus=0.5e7, km=20, block_size_x=48, block_size_y=40, gm_transition_layer=1, **kw))
    pins.check_gm_transition_layer(A)
    A.close()


def test_gm_buoyancy_frequency_profile_and_its_bounds(pkg, orclib_built):
    """k_gm_kappa_vertical against N^2 / N_ref^2 in [0.1, 1] formed from central differences of the device's own density: pins.check_gm_bfre_profile"""
    A = pins.GpuAdapter(pkg, named_config("tiny", hmix_tracer=3, ah=0.8e7, km=20, block_size_x=48, block_size_y=40, gm_kappa_type=1, gm_kappa_freq=1))
    pins.check_gm_bfre_profile(A)
    A.close()


@pytest.mark.parametrize("frac,km,kstar", [(0.4, 20, 10), (0.7, 20, 10), (0.55, 62, 30)])
def test_kpp_boundary_layer_ending_in_a_partial_bottom_cell(pkg, orclib_built, frac, km, kstar):
    """the PBC instantiations of the KPP kernels (62 levels: the column-march forms of large grids are forced by pbc_generic_kpp = 0 only above
    2^19 columns, so km = 62 here runs the 3-D-parallel forms with the register level count) against the closed form with the jump AT the partial
    bottom cell and the zero column sum of the non-local source: pins.check_kpp_hblt_two_layer_pbc"""
    cfg = named_config("tiny", vmix_choice=3, km=km, block_size_x=48, block_size_y=40, partial_bottom_cells=1, ns_boundary=0)
    A = pins.GpuAdapter(pkg, cfg, grid=pins.pbc_flat_grid(cfg, kstar, frac))
    pins.check_kpp_hblt_two_layer_pbc(A, kstar, frac)
    A.close()
