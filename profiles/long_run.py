"""Stability check: many steps of a BASELINE workload with land elimination on; prints iteration counts and field ranges."""
import os, sys, importlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from popcfg import named_config
pkg = importlib.import_module("pop2-cesm_amd")
name, nsteps = sys.argv[1], int(sys.argv[2])
kw = eval("dict(%s)" % sys.argv[3]) if len(sys.argv) > 3 else {}     # e.g. "partial_bottom_cells=1,stepped_bathymetry=1"
m = pkg.PopModel(named_config(name, **kw))
for s in range(1, nsteps + 1):
    m.step()
    if s % (nsteps // 6) == 0 or s == nsteps:
        T = m.get("TRACER", 1, 0); U = m.get("UVEL", 1); P = m.get("PSURF", 1)
        print(name, kw, "step", s, "iters", m.solver_diagnostics()[0], "T [%.3f, %.3f] |U|max %.3f |P|max %.2f finite %s skip %d" %
              (T.min(), T.max(), np.abs(U).max(), np.abs(P).max(), bool(np.isfinite(T).all() and np.isfinite(U).all()), m.dim("land_skip_active")), flush=True)
m.close()
