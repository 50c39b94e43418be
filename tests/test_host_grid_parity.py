"""Host logic of libpop_amd (no GPU): block decomposition, internal grid, masks, operator
coefficients and scalars must equal the CPU oracle bit for bit (SURVEY.md 8a rows a1, a2:
KMT/KMU and index maps bit-exact; all init-time fields are pure functions of the global index)."""
import numpy as np
import pytest

from popcfg import named_config
from orclib import Oracle

FIELDS = ["ULAT", "ULON", "TLAT", "HTN", "HTE", "HUS", "HUW", "DXU", "DYU", "DXT", "DYT", "DXUR", "DYUR",
          "UAREA", "TAREA", "UAREA_R", "TAREA_R", "AU0", "AUN", "AUE", "AUNE", "FCOR", "FCORT", "HU", "HUR", "HT",
          "RCALCT", "RCALCU", "AMF", "AHF", "DTN", "DTS", "DTE", "DTW", "DUC", "DUN", "DUS", "DUE", "DUW", "DMC", "DMN",
          "DMS", "DME", "DMW", "DUM", "KXU", "KYU", "btropWgtNE", "btropWgtEast", "btropWgtNorth", "centerWgtIndep",
          "mMask", "CHECKER", "CONSTNT"]
IFIELDS = ["KMT", "KMU", "KMTN", "KMTS", "KMTE", "KMTW", "KMTEE", "KMTNN"]


@pytest.mark.parametrize("name,kw", [("tiny", {}), ("test", {}), ("gx3v7", {}), ("tiny", {"lvariable_hmix": 1}),
                                     ("tiny", {"ew_boundary": 0}), ("tiny", {"tmix_opt": 1}),
                                     ("tiny", {"hmix_momentum": 4, "hmix_tracer": 4, "lvariable_hmix": 1}),
                                     ("tiny", {"hmix_momentum": 4, "hmix_tracer": 4}),
                                     ("tiny", {"tadvect": 2}), ("gx3v7", {"tadvect": 2}),
                                     ("tiny", {"tmix_opt": 3}), ("tiny", {"tmix_opt": 3, "robert_alpha": 1.0, "robert_nu": 0.1}),
                                     ("tiny", {"solver_choice": 3}), ("gx3v7", {"solver_choice": 3}), ("test", {"solver_choice": 3})])
def test_host_fields_bit_exact(pkg, orclib_built, name, kw):
    cfg = named_config(name, **kw)
    m = pkg.PopModel(cfg, host_only=True)
    o = Oracle(cfg)
    assert (m.nxb, m.nyb, m.km, m.nblocks) == (o.nxb, o.nyb, o.km, o.nblocks)
    for f in IFIELDS:
        assert np.array_equal(m.geti(f), o.i2(f)), f
    for f in FIELDS:
        a, b = m.get(f), o.f2(f)
        assert np.array_equal(a, b), "%s max diff %g" % (f, np.abs(a - b).max())
    if cfg.hmix_momentum == 4:
        for f in ("DUC", "DUN", "DUS", "DUE", "DUW", "DMC", "DMN", "DMS", "DME", "DMW", "DUM", "DTN", "DTS", "DTE", "DTW"):
            assert np.array_equal(m.get("d4" + f), o.f2("d4" + f)), "d4" + f
        assert np.array_equal(m.get("D4AMF"), o.f2("D4_AMF")) and np.array_equal(m.get("D4AHF"), o.f2("D4_AHF"))
    if cfg.tadvect == 2:     # third-order upwind weights (advection.F90:420-562)
        for f in ("TALFXP", "TBETXP", "TGAMXP", "TALFXM", "TBETXM", "TDELXM", "TALFYP", "TBETYP", "TGAMYP", "TALFYM", "TBETYM", "TDELYM"):
            assert np.array_equal(m.get(f), o.f2(f)), f
    for n in (0, 1):
        assert np.array_equal(m.get("SMF", 1, n), o.f2("SMF", 1, n))
        assert np.array_equal(m.get("SMFT", 1, n), o.f2("SMFT", 1, n))
    for s in ("dtt", "dtu", "dtp", "residualNorm", "convergenceCriterion", "rcheck", "rconst", "uarea_equator"):
        assert m.scalar(s) == o.scalar(s), s
    if cfg.solver_choice == 3:   # P-CSI preprocessing: Lanczos eigenvalue bounds (POP_SolversMod.F90:2699-2990)
        for s in ("PcsiMaxEigs", "PcsiMinEigs", "lanczos_steps"):
            assert m.scalar(s) == o.scalar(s), s
        assert 0.0 < m.scalar("PcsiMinEigs") < m.scalar("PcsiMaxEigs")
    if cfg.tmix_opt == 3:    # Robert filter coefficients, budget area and volumes (time_management.F90:897, step_mod.F90:1606)
        for s in ("robert_curtime", "robert_newtime", "bgtarea_t_1", "rf_volume_2_km", "open_ocean_volume_2_km"):
            assert m.scalar(s) == o.scalar(s), s
    assert m.dim("nsteps_per_interval") == o.dim("nsteps_per_interval")
    # block table (blocks.F90 create_blocks): ids, extents and global index maps
    ig, jg = o.ivec("i_glob", o.nxb * o.nblocks), o.ivec("j_glob", o.nyb * o.nblocks)
    for bid in range(1, m.nblocks_tot + 1):
        blk = m.get_block(bid)
        assert blk["block_id"] == bid and blk["local_id"] == bid
        assert np.array_equal(blk["i_glob"], ig[(bid - 1) * o.nxb: bid * o.nxb])
        assert np.array_equal(blk["j_glob"], jg[(bid - 1) * o.nyb: bid * o.nyb])
        assert (blk["ib"], blk["ie"], blk["jb"], blk["je"]) == (3, m.nxb - 2, 3, m.nyb - 2)
    m.close()
    o.close()


def test_host_only_context_refuses_compute(pkg):
    """The product has no CPU fallback: compute entry points must fail loudly without a GPU."""
    m = pkg.PopModel(named_config("tiny"), host_only=True)
    for fn in (m.dhdt, m.baroclinic_driver, m.barotropic_driver, m.baroclinic_correct_adjust, m.step_tail, m.step,
               m.solver_run):
        with pytest.raises(pkg.PopError, match="no CPU fallback|host-only"):
            fn()
    m.close()


def test_abi_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    names = pkg.abi_symbols()
    assert len(names) >= 35
    for s in names:
        assert hasattr(L, s), s


def test_bad_configs_are_rejected(pkg):
    with pytest.raises(pkg.PopError, match="divide"):
        pkg.PopModel(named_config("tiny", block_size_x=13), host_only=True)
    with pytest.raises(pkg.PopError, match="nt"):
        pkg.PopModel(named_config("tiny", nt=1), host_only=True)
    m = pkg.PopModel(named_config("tiny"), host_only=True)
    assert m.L.pop_get_block(m.h, 0, None, None, None) != 0          # get_block: invalid block_id
    assert m.L.pop_get_block(m.h, m.nblocks_tot + 1, None, None, None) != 0
    m.close()


@pytest.mark.parametrize("field,value", [("hmix_momentum", 3), ("hmix_tracer", 1), ("vmix_choice", 4), ("solver_choice", 0), ("tadvect", 4),
                                         ("aidif", 0.5), ("tmix_opt", 7), ("ew_boundary", 2), ("ns_boundary", 5), ("convergence_check_freq", 0),
                                         ("precond_choice", 2), ("km", 1)])
def test_unsupported_options_are_refused_at_create(pkg, field, value):
    """every option value the library does not implement fails loudly in pop_create (host-only contexts included), with a
    message that names the option"""
    cfg = named_config("tiny", **{field: value})
    with pytest.raises(pkg.PopError) as e:
        pkg.PopModel(cfg, host_only=True)
    assert "pop_create" in str(e.value)
