"""Land elimination (workgroups whose tile holds no ocean cell return at once; the reference drops land BLOCKS from its
distribution, here the unit is the tile a workgroup works on): it must be invisible -- every field, land cells included,
bit for bit what the run without it gives, and equal to the oracle on every cell."""
import numpy as np
import pytest

from popcfg import named_config, synthetic_grid
from orclib import Oracle
from test_gpu_parity import run_phases, force_kpp_case, TOL_LOCAL, TOL_SOLVE

pytestmark = pytest.mark.gpu

FIELDS = [("TRACER", 0), ("TRACER", 1), ("UVEL", 0), ("VVEL", 0), ("RHO", 0), ("PSURF", 0), ("UBTROP", 0), ("VBTROP", 0),
          ("GRADPX", 0), ("GRADPY", 0), ("PGUESS", 0)]
WORK = ["DH", "DHU", "ZX", "ZY", "VVC", "RHS"]

BIG = {"POP_XCD_REMAP": "2", "POP_RED_TILES": "1"}     # the tile orders production uses above 2^19 columns
CASES = [
    ("wide", {"vmix_choice": 3, "hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e19, "ah": -1.0e18, "time_mix_freq": 5}, BIG, 14),
    ("wide", {"vmix_choice": 3, "hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e19, "ah": -1.0e18, "time_mix_freq": 5, "solver_choice": 2}, {}, 12),
    ("wide", {"block_size_x": 1056, "tadvect": 2, "vmix_choice": 2, "tmix_opt": 1, "time_mix_freq": 4}, BIG, 12),
    ("wide", {"time_mix_freq": 5}, {}, 10),                                                      # fused pcg on the compacted chunk list
    ("wide", {"block_size_x": 1056, "solver_choice": 2}, {}, 10),                                # two blocks: lists padded to one length
    # the large-grid forms of the fused pcg on the compacted list: block sums by their own launch, two chunks per workgroup in step A
    # (k_fpcg_a_pair), two cells per thread in step B; one block and two
    ("wide", {"time_mix_freq": 5}, {"POP_SOLVER_PRESUM": "1"}, 10),
    ("wide", {"block_size_x": 1056}, {"POP_SOLVER_PRESUM": "1"}, 8),
    ("wide", {"time_mix_freq": 5}, {"POP_SOLVER_PRESUM": "1", "POP_FPCG_A_PAIR": "0"}, 8),
    ("wide", {"solver_choice": 3, "time_mix_freq": 5}, {}, 10),                                 # P-CSI: land chunks of both ping-pong halves stay 0
    ("wide", {"solver_choice": 3, "precond_choice": 1, "block_size_x": 1056}, BIG, 8),           # P-CSI + EVP, two blocks, tile order
    ("test", {"vmix_choice": 3, "stepped_bathymetry": 1, "time_mix_freq": 6}, {}, 13),          # 96 blocks, many of them land
    ("gx3v7", {"tadvect": 3, "tmix_opt": 3}, {}, 10),
    ("tiny", {"solver_choice": 3, "hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e22, "ah": -1.0e21, "lvariable_hmix": 1}, {}, 10),
    # the buffers the fused del4 first Laplacians share with other kernel choices: forward elimination inside the tracer kernel (E, F
    # of the second tracer in S3c / S3d) and the in-line order (d2t / d2u aliased onto S3a / S3b, which KPP also uses as scratch)
    ("wide", {"vmix_choice": 3, "hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e19, "ah": -1.0e18, "time_mix_freq": 5},
     {"POP_D2T_FUSE": "1", "POP_TRACER_FWD": "1", "POP_REG_THOMAS_T": "0"}, 10),
    ("wide", {"vmix_choice": 3, "hmix_momentum": 4, "hmix_tracer": 4, "am": -1.0e19, "ah": -1.0e18, "time_mix_freq": 5},
     {"POP_DEL4_SIDE": "0", "POP_D2T_FUSE": "1"}, 10),
]


def _model(pkg, monkeypatch, cfg, env, skip):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("POP_LAND_SKIP", "1" if skip else "0")
    monkeypatch.setenv("POP_LAND_FULL_STEPS", "4")
    return pkg.PopModel(cfg)


@pytest.mark.parametrize("name,kw,env,nsteps", CASES)
def test_land_elimination_is_bitwise_invisible(pkg, monkeypatch, name, kw, env, nsteps):
    cfg = named_config(name, **kw)
    a, b = _model(pkg, monkeypatch, cfg, env, False), _model(pkg, monkeypatch, cfg, env, True)
    assert b.scalar("land_tile_fraction") > 0.05, "the case has no land tiles"
    if cfg.vmix_choice == 3:   # the same forcing on both (set before the first step)
        tlat = a.get("TLAT")
        for m in (a, b):
            m.set("STF", -3.0e-2 * np.sin(tlat) - 1.0e-2, n=0); m.set("STF", 2.0e-6 * np.cos(2.0 * tlat), n=1)
    for s in range(1, nsteps + 1):
        a.step(); b.step()
        assert a.dim("land_skip_active") == 0 and b.dim("land_skip_active") == (1 if s > 4 else 0)
        assert a.solver_diagnostics() == b.solver_diagnostics(), "step %d" % s
        if s in (5, 6, nsteps):
            for f, n in FIELDS:
                for tl in (0, 1, 2):
                    assert np.array_equal(a.get(f, tl, n), b.get(f, tl, n)), "step %d %s tl %d" % (s, f, tl)
            for f in WORK:
                assert np.array_equal(a.get(f), b.get(f)), "step %d %s" % (s, f)
            if cfg.vmix_choice == 3:
                for n in (0, 1):
                    assert np.array_equal(a.get("VDC", n=n), b.get("VDC", n=n)) and np.array_equal(a.get("KPP_SRC", n=n), b.get("KPP_SRC", n=n))
                assert np.array_equal(a.get("HBLT"), b.get("HBLT"))
    a.close(); b.close()


@pytest.mark.parametrize("name,kw,env,nsteps", [CASES[0], CASES[3], CASES[7]])
def test_phases_match_oracle_with_land_elimination_active(pkg, orclib_built, monkeypatch, name, kw, env, nsteps):
    """the oracle computes every cell; steps 5.. run with land tiles skipped and are compared on every cell, ghosts included"""
    cfg = named_config(name, **kw)
    gpu, orc = _model(pkg, monkeypatch, cfg, env, True), Oracle(cfg)
    if cfg.vmix_choice == 3:
        force_kpp_case(gpu, orc)
    tol = TOL_LOCAL
    for s in range(1, 7):      # two steps with the tiles skipped; more steps only widen the solver's summation-order difference (KPP)
        run_phases(gpu, orc, s, tol)
        tol = TOL_SOLVE
    assert gpu.dim("land_skip_active") == 1
    gpu.close(); orc.close()


def test_new_state_restarts_the_full_steps(pkg, monkeypatch):
    cfg = named_config("tiny")
    m = _model(pkg, monkeypatch, cfg, {}, True)
    for _ in range(6):
        m.step()
    assert m.dim("land_skip_active") == 1
    m.set("TRACER", m.get("TRACER", 1, 0), tl=1, n=0)
    m.step()
    assert m.dim("land_skip_active") == 0
    m.set("STF", m.get("STF", n=0), n=0)     # forcing does not
    for _ in range(5):
        m.step()
    assert m.dim("land_skip_active") == 1
    m.close()
