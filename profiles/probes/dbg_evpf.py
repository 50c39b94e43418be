import os, sys
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
import numpy as np
import __graft_entry__ as ge
from popcfg import named_config
pkg = ge.load_package()
def cfg(): return named_config("tiny", solver_choice=3, preconditioner_choice=1, block_size_x=28, block_size_y=24)
# sequential
res = []
for fused in (0, 1, 1, 0):
    m = pkg.PopModel(cfg(), tuning={"pcsi_evp_fused": fused}); m.step(); m.sync()
    res.append((fused, m.get("PSURF").copy(), m.solver_diagnostics())); m.close()
for r in res[1:]:
    print("sequential", res[0][0], r[0], int((res[0][1] != r[1]).sum()), res[0][2], r[2])
# concurrent
a = pkg.PopModel(cfg(), tuning={"pcsi_evp_fused": 0}); b = pkg.PopModel(cfg(), tuning={"pcsi_evp_fused": 1}); c = pkg.PopModel(cfg(), tuning={"pcsi_evp_fused": 0})
a.step(); b.step(); c.step()
pa, pb, pc = a.get("PSURF"), b.get("PSURF"), c.get("PSURF")
print("concurrent a-b", int((pa != pb).sum()), "a-c", int((pa != pc).sum()), "a-seq0", int((pa != res[0][1]).sum()), "b-seq1", int((pb != res[1][1]).sum()), a.solver_diagnostics(), b.solver_diagnostics())
