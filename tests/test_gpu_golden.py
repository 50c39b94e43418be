"""GPU vs the committed golden vectors (tests/golden/*.npz, produced by the CPU oracle)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-8     # fp64, downstream of the solver's fixed-tree dot products (see test_gpu_parity.py)


@pytest.mark.parametrize("case", ["const", "kpp_del4", "upwind3", "robert", "pcsi_evp", "lw_lim", "pbc_kpp_del4", "padded", "gm", "gm_tlt"])
def test_gpu_matches_golden(pkg, case):
    import golden.make_golden as mg
    g = np.load(os.path.join(GOLD, "golden_%s.npz" % case))
    m = pkg.PopModel(mg.config(case))
    mg.prepare(m, case)
    iters = []
    for _ in range(int(g["nsteps"])):
        m.step()
        iters.append(m.solver_diagnostics()[0])
    assert iters == list(g["iters"]), "solver iteration counts differ from the golden run"
    for name, _ in mg.FIELDS:
        a, b = m.get(name, 1, 0), g[name]
        if case == "padded":      # the cells that exist, and not the ghost cells that touch the padding
            ex = g["exists"]
            near = ex.copy()
            near[:, 1:, :] &= ex[:, :-1, :]; near[:, :-1, :] &= ex[:, 1:, :]; near[:, :, 1:] &= ex[:, :, :-1]; near[:, :, :-1] &= ex[:, :, 1:]
            sel = np.broadcast_to(near[:, None], a.shape) if a.ndim == 4 else near
            a, b = a[sel], b[sel]
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
        assert err <= TOL, "%s rel err %.3e" % (name, err)
    m.close()
