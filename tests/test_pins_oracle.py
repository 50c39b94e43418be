"""The independent pins of tests/pins.py against the CPU oracle (no GPU): LAPACK for the implicit vertical solves, closed
forms for the pressure gradient and for horizontal diffusion.  tests/test_gpu_pins.py runs the same checks on the device."""
import numpy as np
import pytest

import pins
from popcfg import named_config

CASES = [
    ("const-stepped", dict(stepped_bathymetry=1)),                              # shallow columns: KMT = 8 ... 16
    ("kpp-km20-stepped", dict(vmix_choice=3, km=20, stepped_bathymetry=1)),     # two VDC fields holding the same values
    ("kpp-dd-km20-stepped", dict(vmix_choice=3, km=20, ldbl_diff=1, stepped_bathymetry=1)),   # double diffusion: distinct values per tracer class
    ("rich-flat", dict(vmix_choice=2)),
    # partial bottom cells: the systems on the columns' own thicknesses (DZT / DZU formed in the pin from KMT and DZBC)
    ("pbc-const-stepped", dict(stepped_bathymetry=1, partial_bottom_cells=1)),
    ("pbc-kpp-dd-km20-stepped", dict(vmix_choice=3, km=20, ldbl_diff=1, stepped_bathymetry=1, partial_bottom_cells=1)),
]


@pytest.fixture(params=CASES, ids=[c[0] for c in CASES])
def adapter(request, orclib_built):
    A = pins.OracleAdapter(named_config("tiny", **request.param[1]))
    yield A
    A.close()


def test_impvmixt_solves_its_tridiagonal_system(adapter):
    pins.check_impvmixt(adapter, np.random.default_rng(11))


def test_impvmixt_correct_solves_its_tridiagonal_system(adapter):
    pins.check_impvmixt_correct(adapter, np.random.default_rng(12))


def test_impvmixu_and_mean_removal_match_lapack(adapter):
    pins.check_impvmixu(adapter, np.random.default_rng(13))


@pytest.mark.parametrize("kw", [dict(), dict(stepped_bathymetry=1), dict(impcor=0), dict(hmix_momentum=4, am=-1.0e19)],
                         ids=["flat", "stepped", "explicit-coriolis", "del4"])
def test_pressure_gradient_closed_forms(kw, orclib_built):
    A = pins.OracleAdapter(named_config("tiny", **kw))
    pins.check_gradp(A)
    A.close()


@pytest.mark.parametrize("kw", [dict(), dict(hmix_tracer=4, ah=-1.0e19), dict(hmix_tracer=4, ah=-1.0e19, lvariable_hmix=1), dict(vmix_choice=3, km=20)],
                         ids=["del2", "del4", "del4-variable", "del2-kpp"])
def test_tracer_diffusion_of_a_quadratic_field(kw, orclib_built):
    A = pins.OracleAdapter(named_config("tiny", block_size_x=48, block_size_y=40, **kw))
    pins.check_hdifft(A)
    A.close()


# ---- KPP against Large, McWilliams & Doney (1994) and advection against exact flux divergences (tests/pins.py) ----------
KPP_PIN = dict(vmix_choice=3, lrich=0, bckgrnd_vdc2=0.0, block_size_x=48, block_size_y=40)


@pytest.mark.parametrize("nu0", [0.0, 2000.0], ids=["no-interior-mixing", "uniform-interior-nu"])
@pytest.mark.parametrize("regime", ["stable", "weak", "strong"])
def test_kpp_velocity_scales_and_shape_function(regime, nu0, orclib_built):
    A = pins.OracleAdapter(named_config("tiny", km=20, bckgrnd_vdc1=nu0, **KPP_PIN))
    pins.check_kpp_scales_and_shape(A, regime, nu0)
    A.close()


def test_kpp_boundary_layer_depth_of_a_two_layer_column(orclib_built):
    A = pins.OracleAdapter(named_config("tiny", vmix_choice=3, km=20, block_size_x=48, block_size_y=40))
    pins.check_kpp_hblt_two_layer(A)
    A.close()


@pytest.mark.parametrize("kw", [dict(tadvect=1), dict(tadvect=2), dict(tadvect=1, vmix_choice=3, km=20), dict(tadvect=2, hmix_tracer=4)],
                         ids=["centred", "upwind3", "centred-kpp", "upwind3-del4"])
def test_advection_of_a_linear_field_is_the_exact_flux_divergence(kw, orclib_built):
    A = pins.OracleAdapter(named_config("tiny", ah=0.0, block_size_x=48, block_size_y=40, **kw))
    pins.check_advt_linear(A)
    A.close()


def test_limited_advection_creates_no_new_extremum(orclib_built):
    A = pins.OracleAdapter(named_config("tiny", ah=0.0, tadvect=3, block_size_x=48, block_size_y=40))
    pins.check_lw_lim_monotone(A)
    A.close()


@pytest.mark.parametrize("ah_bolus", [0.0, 0.3e7])
def test_gm_fluxes_of_a_linear_field_are_the_closed_form(orclib_built, ah_bolus):
    A = pins.OracleAdapter(named_config("tiny", hmix_tracer=3, ah=0.8e7, ah_bolus=ah_bolus, km=20, block_size_x=48, block_size_y=40))
    pins.check_gm_linear(A)
    A.close()


GM_TAPER_CASES = [(0, 0.0, 0.0), (0, 0.2, 0.5e7), (1, 0.0, 0.0), (1, 0.2, 0.5e7), (3, 0.0, 0.0), (3, 0.2, 0.5e7), (2, 0.0, 0.0)]


@pytest.mark.parametrize("control,slm_b,ah_bolus", GM_TAPER_CASES)
def test_gm_slope_tapers_on_a_constructed_slope_field(orclib_built, control, slm_b, ah_bolus):
    """slope_control 'notanh' / 'tanh' / 'clip' / 'Gerd', equal and different slope limits (diff_tapering): pins.check_gm_tapers"""
    A = pins.OracleAdapter(named_config("tiny", hmix_tracer=3, ah=0.8e7, ah_bolus=ah_bolus, km=20, block_size_x=48, block_size_y=40,
                                        gm_slope_control=control, slm_b=slm_b, gm_diag_bolus=1))
    pins.check_gm_tapers(A)
    A.close()


@pytest.mark.parametrize("kw", [{}, {"gm_slope_control": 1, "slm_b": 0.2}], ids=["notanh", "tanh-diff-tapering"])
def test_gm_transition_layer_depths_and_merged_streamfunction(orclib_built, kw):
    """transition_layer / merged_streamfunction (hmix_gm.F90:3183-3743) against what they are supposed to deliver: pins.check_gm_transition_layer"""
    A = pins.OracleAdapter(named_config("tiny", hmix_tracer=3, ah=0.8e7, ah_bolus=0.5e7, km=20, block_size_x=48, block_size_y=40, gm_transition_layer=1, **kw))
    pins.check_gm_transition_layer(A)
    A.close()


def test_gm_buoyancy_frequency_profile_and_its_bounds(orclib_built):
    """kappa type 'bfre': KAPPA_VERTICAL = N^2 / N_ref^2 in [0.1, 1] with N^2 from central differences of the model's density: pins.check_gm_bfre_profile"""
    A = pins.OracleAdapter(named_config("tiny", hmix_tracer=3, ah=0.8e7, km=20, block_size_x=48, block_size_y=40, gm_kappa_type=1, gm_kappa_freq=1))
    pins.check_gm_bfre_profile(A)
    A.close()


@pytest.mark.parametrize("frac,km,kstar", [(0.4, 20, 10), (0.7, 20, 10), (0.55, 62, 30)])
def test_kpp_boundary_layer_ending_in_a_partial_bottom_cell(orclib_built, frac, km, kstar):
    """vmix_kpp.F90:2212-2220, 2359-2366, 2561-2575, 1296-1302 with partial_bottom_cells: pins.check_kpp_hblt_two_layer_pbc"""
    cfg = named_config("tiny", vmix_choice=3, km=km, block_size_x=48, block_size_y=40, partial_bottom_cells=1, ns_boundary=0)
    A = pins.OracleAdapter(cfg, grid=pins.pbc_flat_grid(cfg, kstar, frac))
    pins.check_kpp_hblt_two_layer_pbc(A, kstar, frac)
    A.close()
