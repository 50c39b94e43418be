"""tx0.1v3 at full size in one block (33 844 chunk partials, 133 terms per accumulator): k_block_sums_relay (pop_tuning.block_sums_relay = 1,
the default there) against the 256-thread k_block_sums (0) -- iteration counts and PSURF bit for bit, solver time per iteration of both.
usage: python3 profiles/probes/relay_fullsize.py [pcg|chrongear]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
from popcfg import named_config
pkg = ge.load_package()
solver = {"pcg": 1, "chrongear": 2}[sys.argv[1] if len(sys.argv) > 1 else "pcg"]
out = {"solver": solver, "rows": []}
ref = None
for relay in (0, 1, 0, 1):
    m = pkg.PopModel(named_config("tx0.1v3", solver_choice=solver), tuning={"block_sums_relay": relay})
    for _ in range(6):
        m.step()
    m.sync(); m.scalar("solver_ms_reset")
    its = []
    for _ in range(5):
        m.step(); its.append(m.solver_diagnostics()[0])
    m.sync()
    ms, n = m.scalar("solver_ms_total"), m.scalar("solver_iterations_total")
    ps = m.get("PSURF").copy()
    if ref is None:
        ref = (its, ps)
    same = its == ref[0] and np.array_equal(ps, ref[1])
    out["rows"].append({"block_sums_relay": relay, "iterations": its, "solver_us_per_iteration": round(1e3 * ms / n, 2), "bitwise_equal_first_run": bool(same)})
    print(out["rows"][-1], flush=True)
    m.close()
print(json.dumps(out))
