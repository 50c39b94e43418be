import os, sys, importlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from popcfg import named_config, synthetic_grid
pkg = importlib.import_module("pop2-cesm_amd")
for kw in ({"block_size_x": 24, "block_size_y": 20}, {"block_size_x": 48, "block_size_y": 40}, {"block_size_x": 48, "block_size_y": 10}):
    cfg = named_config("tiny", ns_boundary=2, **kw)
    g = synthetic_grid(cfg)
    os.environ.pop("POP_SOLVER_UNFUSED", None)
    a = pkg.PopModel(cfg, grid=g)
    os.environ["POP_SOLVER_UNFUSED"] = "1"
    b = pkg.PopModel(cfg, grid=g)
    for s in range(3):
        a.step(); b.step()
        pa, pb = a.get("PSURF", 1, 0), b.get("PSURF", 1, 0)
        w = np.argwhere(pa != pb)
        print(kw, "step", s, a.solver_diagnostics()[0], b.solver_diagnostics()[0], "PSURF differs at", len(w), "cells", w[:4].tolist())
