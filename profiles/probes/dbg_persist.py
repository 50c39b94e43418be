import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import __graft_entry__ as ge
from popcfg import named_config
pkg = ge.load_package()
for name, kw in (("tiny", dict(block_size_x=48, block_size_y=5)), ("tiny", dict(block_size_x=48, block_size_y=5, distribution=1)), ("tiny", dict(block_size_x=48, block_size_y=5)), ("tiny", dict(block_size_x=48, block_size_y=5))):
    cfg = named_config(name, **kw)
    a = pkg.PopModel(cfg, tuning={"pcg_persist": 0}); b = pkg.PopModel(cfg)
    for s in range(5):
        a.step()
        err = ""
        try:
            b.step()
        except Exception as e:
            err = "ERR %s" % e
        same = np.array_equal(a.get("PSURF"), b.get("PSURF"))
        print(name, kw, "step", s, "fused", a.solver_diagnostics(), "persist", [b.scalar(n) for n in ("persist_iterations","persist_rr","persist_status","persist_checks")], "same", same, err, flush=True)
        if err: break
    a.close(); b.close()
