import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import __graft_entry__ as ge
from popcfg import named_config
pkg = ge.load_package()
def run(cfg, n):
    m = pkg.PopModel(cfg); its = []
    for _ in range(n):
        m.step(); its.append(m.solver_diagnostics()[0])
    out = [m.get(f, 1, 0).copy() for f in ("TRACER", "UVEL", "PSURF")]
    m.close(); return its, out
for name, kw, n in (("gx1v7", {}, 400), ("gx1v7", {"solver_choice": 3, "tmix_opt": 3}, 200), ("tx0.1v3", {"nx_global": 1800, "ny_global": 1200, "block_size_x": 1800, "block_size_y": 1200}, 40)):
    cfg = named_config(name, **kw)
    a = run(cfg, n); b = run(cfg, n)
    same = a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))
    print(name, kw, "steps", n, "reproducible", same, "iters last", a[0][-3:], "Tmax %.6f finite %s" % (a[1][0].max(), np.isfinite(a[1][1]).all()), flush=True)
