"""Host logic of libpop_amd (no GPU): block decomposition, internal grid, masks, operator
coefficients and scalars must equal the CPU oracle bit for bit (SURVEY.md 8a rows a1, a2:
KMT/KMU and index maps bit-exact; all init-time fields are pure functions of the global index)."""
import numpy as np
import pytest

from popcfg import named_config, synthetic_grid, synthetic_dzbc
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
    _host_fields_bit_exact(pkg, cfg, None)


@pytest.mark.parametrize("kw", [{"ns_boundary": 2}, {"ns_boundary": 0}, {"ns_boundary": 1}, {"ns_boundary": 0, "ew_boundary": 0},
                                {"ns_boundary": 2, "hmix_momentum": 4, "hmix_tracer": 4, "lvariable_hmix": 1},
                                {"ns_boundary": 2, "lvariable_hmix": 1, "tadvect": 2, "solver_choice": 3},
                                {"ns_boundary": 2, "precond_choice": 1, "block_size_x": 48, "block_size_y": 40}])
def test_host_fields_from_grid_input_bit_exact(pkg, orclib_built, kw):
    """horiz_grid_opt = 'file' / topography_opt = 'file' (pop_create_with_grid): the caller's global arrays scattered with
    the reference's field locations (tripole ghost rows mirrored), every derived field equal to the oracle's bit for bit"""
    cfg = named_config("tiny", **kw)
    _host_fields_bit_exact(pkg, cfg, synthetic_grid(cfg))


@pytest.mark.parametrize("name,kw,grid_kw", [
    ("tiny", {"stepped_bathymetry": 1}, None),                                    # internal grid: synthetic bottom thickness
    ("test", {"stepped_bathymetry": 1, "vmix_choice": 1}, None),
    ("tiny", {"ns_boundary": 2}, {"dzbc": True}),                                 # caller's DZBC record, through a tripole fold
    ("tiny", {"ns_boundary": 2, "block_size_x": 48, "block_size_y": 40}, {"dzbc": True}),
    ("tiny", {"ns_boundary": 0, "ew_boundary": 0}, {"dzbc": True}),
    ("tiny", {"ns_boundary": 1, "hmix_momentum": 4, "hmix_tracer": 4}, {"dzbc": False}),   # grid records without DZBC: synthetic
])
def test_partial_bottom_cells_host_fields_bit_exact(pkg, orclib_built, name, kw, grid_kw):
    """partial_bottom_cells = 1 (grid.F90:916-1020): DZBC, the thicknesses DZT / DZU and the depths HT, HU, HUR they change"""
    cfg = named_config(name, partial_bottom_cells=1, **kw)
    grid = None
    if grid_kw is not None:
        grid = synthetic_grid(cfg, kmt=grid_kw["dzbc"])
        if grid_kw["dzbc"]:
            grid["DZBC"] = synthetic_dzbc(cfg, grid["KMT"])
    _host_fields_bit_exact(pkg, cfg, grid)


def test_grid_input_scatter_rule(pkg):
    """scatter_global (mpi/gather_scatter.F90:929-945, 1027-1038) stated constructively: a cell with positive global
    indices holds the global value; a ghost row n beyond the fold holds global row ny + yoffset - n at column
    nx + xoffset - i (wrapped), offsets (1,1) centre, (0,0) NE corner, (0,1) E face, (1,0) N face"""
    cfg = named_config("tiny", ns_boundary=2)
    g = synthetic_grid(cfg)
    m = pkg.PopModel(cfg, host_only=True, grid=g)
    nx, ny = cfg.nx_global, cfg.ny_global
    big = lambda a: np.where(a <= 0.0, 1.0, a)
    # the mirrored spacings: DXT from HTN averaged in j (centre), DYU from HTE averaged in j with the tripole row (NE corner)
    dxt = 0.5 * (g["HTN"] + np.roll(g["HTN"], 1, axis=0))
    dyu = 0.5 * (g["HTE"] + np.roll(g["HTE"], -1, axis=0)); dyu[-1] = g["HTE"][-1]
    for name, G, xo, yo in (("ULAT", g["ULAT"], 0, 0), ("HTN", g["HTN"], 1, 0), ("HTE", g["HTE"], 0, 1), ("HUS", g["HUS"], 0, 1),
                            ("HUW", g["HUW"], 1, 0), ("DXT", dxt, 1, 1), ("DYU", dyu, 0, 0), ("KMT", g["KMT"], 1, 1)):
        A = m.geti(name) if name == "KMT" else m.get(name)
        seen_fold = 0
        for bid in range(1, m.nblocks_tot + 1):
            blk = m.get_block(bid)
            for j, jg in enumerate(blk["j_glob"]):
                for i, ig in enumerate(blk["i_glob"]):
                    if ig == 0 or jg == 0:
                        if name in ("DXT", "DYU"):
                            continue          # the closed-boundary extension overwrites those (grid.F90:587-634)
                        want = 0 if name == "KMT" else (0.0 if name == "ULAT" else 1.0)
                    elif jg > 0:
                        want = G[jg - 1, ig - 1]
                    else:
                        n = -jg - ny
                        js, is_ = ny + yo - n, nx + xo - ig
                        is_ = is_ + nx if is_ < 1 else (is_ - nx if is_ > nx else is_)
                        want = G[js - 1, is_ - 1]
                        seen_fold += 1
                    assert A[bid - 1, j, i] == want, (name, bid, j, i)
        assert seen_fold == 2 * m.nxb * (nx // cfg.block_size_x)
    m.close()


def test_grid_files_round_trip(pkg, tmp_path):
    """the reference's direct-access binary files (7 r8 records; one i4 record) read back into the same model"""
    cfg = named_config("tiny", ns_boundary=2)
    g = synthetic_grid(cfg)
    hf, tf = str(tmp_path / "horiz_grid.bin"), str(tmp_path / "topography.bin")
    np.stack([g[n] for n in ("ULAT", "ULON", "HTN", "HTE", "HUS", "HUW", "ANGLE")]).astype(np.float64).tofile(hf)
    g["KMT"].astype(np.int32).tofile(tf)
    g2 = pkg.read_grid_files(hf, tf, cfg.nx_global, cfg.ny_global)
    for n in g:
        assert np.array_equal(g[n], g2[n]), n
    with pytest.raises(pkg.PopError, match="cannot read"):
        pkg.read_grid_files(hf + ".missing", tf, cfg.nx_global, cfg.ny_global)
    a, b = pkg.PopModel(cfg, host_only=True, grid=g), pkg.PopModel(cfg, host_only=True, grid=g2)
    assert np.array_equal(a.get("TAREA"), b.get("TAREA")) and np.array_equal(a.geti("KMU"), b.geti("KMU"))
    a.close(); b.close()
    with pytest.raises(pkg.PopError, match="required"):
        pkg.PopModel(cfg, host_only=True, grid={"ULAT": g["ULAT"]})


def _host_fields_bit_exact(pkg, cfg, grid):
    m = pkg.PopModel(cfg, host_only=True, grid=grid)
    o = Oracle(cfg, grid=grid)
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
    if cfg.partial_bottom_cells:
        # the library keeps two 2-D fields (DZBC, DZUB = DZU at level KMU); the thicknesses it forms from them must be the
        # reference's 3-D DZT / DZU (grid.F90:926-1016, levels 0 and km+1 zero) on EVERY cell, ghosts included
        assert np.array_equal(m.get("DZBC"), o.f2("DZBC"))
        dz = o.v1("dz")
        k = np.arange(cfg.km + 2)[None, :, None, None]
        dzk = np.where((k >= 1) & (k <= cfg.km), dz[np.clip(k, 0, cfg.km)], 0.0)
        for name2, kb, bot in (("DZT", m.geti("KMT"), m.get("DZBC")), ("DZU", m.geti("KMU"), m.get("DZUB"))):
            mine = np.where((k == kb[:, None]) & (k >= 1), bot[:, None], dzk)
            ref = o.f3p(name2)
            assert np.array_equal(mine, ref), "%s: %d cells differ" % (name2, int((mine != ref).sum()))
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
    with pytest.raises(pkg.PopError, match="nt"):
        pkg.PopModel(named_config("tiny", nt=1), host_only=True)
    m = pkg.PopModel(named_config("tiny"), host_only=True)
    assert m.L.pop_get_block(m.h, 0, None, None, None) != 0          # get_block: invalid block_id
    assert m.L.pop_get_block(m.h, m.nblocks_tot + 1, None, None, None) != 0
    m.close()


def test_combinations_the_reference_aborts_on_are_refused(pkg, orclib_built):
    """init_gm aborts for Gent-McWilliams with partial bottom cells ('hmix_gm currently incompatible with partial bottom cells',
    hmix_gm.F90:782-785: slopes and fluxes use dz(k), the tracer budget DZT) and for the 'depth' kappa profile with kappa_depth_2 = 0
    (:724-730) and for the 'bfre' kappa with kappa_freq 'never' (:756-780).  Library and oracle refuse all three; a pop_config of another layout version is refused by both as well."""
    from orclib import Oracle
    for kw, word in (({"hmix_tracer": 3, "ah": 0.8e7, "partial_bottom_cells": 1, "stepped_bathymetry": 1}, "partial_bottom_cells"),
                     ({"hmix_tracer": 3, "ah": 0.8e7, "gm_kappa_type": 2, "kappa_depth_1": 1.0}, "kappa_depth_2"),
                     ({"hmix_tracer": 3, "ah": 0.8e7, "gm_kappa_type": 1}, "gm_kappa_freq"),
                     ({"struct_version": 4}, "struct_version")):
        cfg = named_config("tiny", **kw)
        with pytest.raises(pkg.PopError, match=word):
            pkg.PopModel(cfg, host_only=True)
        with pytest.raises(Exception):
            Oracle(cfg)


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


@pytest.mark.parametrize("kw", [{"block_size_x": 20, "block_size_y": 16, "solver_choice": 3}, {"block_size_x": 28, "block_size_y": 24, "solver_choice": 3, "precond_choice": 1},
                                {"block_size_x": 20, "block_size_y": 16, "ns_boundary": 2, "solver_choice": 3, "precond_choice": 1}])
def test_padded_blocks_solver_set_up_bit_exact(pkg, orclib_built, kw):
    """r4: padded blocks with P-CSI (Lanczos bounds, POP_SolversMod.F90:2699-2990), the EVP preconditioner (sub-block tables on every
    block's own extents, :2434-2696) and the tripole fold: the eigenvalue bounds and step count equal the oracle's bit for bit."""
    cfg = named_config("tiny", **kw)
    g = synthetic_grid(cfg) if cfg.ns_boundary == 2 else None
    m, o = pkg.PopModel(cfg, host_only=True, grid=g), Oracle(cfg, grid=g)
    for s in ("PcsiMaxEigs", "PcsiMinEigs", "lanczos_steps", "residualNorm", "convergenceCriterion"):
        assert m.scalar(s) == o.scalar(s), s
    assert 0.0 < m.scalar("PcsiMinEigs") < m.scalar("PcsiMaxEigs")
    m.close(); o.close()


@pytest.mark.parametrize("kw", [{"block_size_x": 20, "block_size_y": 16}, {"block_size_x": 36, "block_size_y": 40}, {"block_size_x": 20, "block_size_y": 16, "ew_boundary": 0},
                                {"block_size_x": 20, "block_size_y": 16, "hmix_momentum": 4, "hmix_tracer": 4, "lvariable_hmix": 1, "tadvect": 2},
                                {"block_size_x": 20, "block_size_y": 16, "stepped_bathymetry": 1, "partial_bottom_cells": 1}],
                         ids=["20x16", "36x40", "20x16-closed", "20x16-del4-upwind3", "20x16-pbc"])
def test_padded_blocks_host_fields_bit_exact(pkg, orclib_built, kw):
    """Block sizes that do not divide the 48 x 40 domain (blocks.F90:174-265): the last column / row of blocks is padded.  Block
    table (ie, je, global index maps with 0 in the padding) equal to the oracle's, and every init-time field equal bit for bit on
    the cells that exist (global index non-zero in both directions; what sits in the padding is never read)."""
    cfg = named_config("tiny", **kw)
    m = pkg.PopModel(cfg, host_only=True)
    o = Oracle(cfg)
    assert (m.nxb, m.nyb, m.km, m.nblocks) == (o.nxb, o.nyb, o.km, o.nblocks)
    ig, jg = o.ivec("i_glob", o.nxb * o.nblocks).reshape(o.nblocks, o.nxb), o.ivec("j_glob", o.nyb * o.nblocks).reshape(o.nblocks, o.nyb)
    short = 0
    for bid in range(1, m.nblocks_tot + 1):
        blk = m.get_block(bid)
        assert np.array_equal(blk["i_glob"], ig[bid - 1]) and np.array_equal(blk["j_glob"], jg[bid - 1])
        assert blk["ie"] == o.ivec("blk_ie", o.nblocks)[bid - 1] and blk["je"] == o.ivec("blk_je", o.nblocks)[bid - 1]
        short += (blk["ie"] < m.nxb - 2) or (blk["je"] < m.nyb - 2)
    assert short > 0, "no padded block in this decomposition"
    cell = (jg != 0)[:, :, None] & (ig != 0)[:, None, :]
    assert (~cell).any()
    # stencil-built coefficients (AUE = TAREA(i+1,j) / 4 UAREA ...) of a ghost cell next to the padding read the padding: compared are
    # the cells whose eight neighbours exist too -- every physical cell and the first ring of ghost cells
    exists = cell.copy()
    for dj in (-1, 0, 1):
        for di in (-1, 0, 1):
            sh = np.ones_like(cell)
            js = slice(max(dj, 0), cell.shape[1] + min(dj, 0)); jd = slice(max(-dj, 0), cell.shape[1] + min(-dj, 0))
            is_ = slice(max(di, 0), cell.shape[2] + min(di, 0)); id_ = slice(max(-di, 0), cell.shape[2] + min(-di, 0))
            sh[:, jd, id_] = cell[:, js, is_]
            exists &= sh
    for f in IFIELDS:
        assert np.array_equal(m.geti(f)[exists], o.i2(f)[exists]), f
    names = list(FIELDS)
    if cfg.hmix_momentum == 4:
        names += ["d4" + f for f in ("DUC", "DUN", "DUS", "DUE", "DUW", "DMC", "DMN", "DMS", "DME", "DMW", "DUM", "DTN", "DTS", "DTE", "DTW")]
    if cfg.tadvect == 2:
        names += ["TALFXP", "TBETXP", "TGAMXP", "TALFXM", "TBETXM", "TDELXM", "TALFYP", "TBETYP", "TGAMYP", "TALFYM", "TBETYM", "TDELYM"]
    if cfg.partial_bottom_cells:
        names += ["DZBC"]
    for f in names:
        a, b = m.get(f), o.f2(f)
        assert np.array_equal(a[exists], b[exists]), "%s: %d of %d existing cells differ" % (f, int((a[exists] != b[exists]).sum()), int(exists.sum()))
    assert np.isfinite(np.concatenate([m.get(f).ravel() for f in ("DXUR", "DYUR", "TAREA_R", "UAREA_R", "HUR")])).all()
    for s in ("residualNorm", "convergenceCriterion", "rcheck", "rconst"):
        assert m.scalar(s) == o.scalar(s), s
    m.close(); o.close()
