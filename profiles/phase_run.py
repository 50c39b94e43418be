"""Runs one named phase of the tx0.1v3 (or another) workload a few times -- the target of per-kernel rocprofv3 --pmc passes.
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU ... -d out -- python3 profiles/phase_run.py vmix [workload] [reps]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from popcfg import named_config

phase = sys.argv[1] if len(sys.argv) > 1 else "vmix"
wl = sys.argv[2] if len(sys.argv) > 2 else "tx0.1v3"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pkg = ge.load_package()
m = pkg.PopModel(named_config(wl))
m.step(); m.step()
m.time_manager()
print(phase, wl, "ms", m.time_phase(phase, reps=reps))
m.close()
