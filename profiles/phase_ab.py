"""A/B of kernel variants on one box: per-phase HIP-event times and ms per step of a workload, one line per configuration.
    python3 profiles/phase_ab.py tx0.1v3 "POP_MOMENTUM_LDS=8" "POP_MOMENTUM_LDS=4" "POP_AMD_LIB=pop2-cesm_amd/libpop_amd_trc3.so"
Each configuration runs in its own process (environment variables are read at pop_create / library load)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, time, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import __graft_entry__ as ge
from popcfg import named_config
pkg = ge.load_package()
kw = json.loads(sys.argv[3])
m = pkg.PopModel(named_config(sys.argv[2], **kw))
for _ in range(6):
    m.step()
m.sync()
m.scalar("solver_ms_reset")
t0 = time.perf_counter()
n = 8
for _ in range(n):
    m.step()
m.sync()
ms = 1e3 * (time.perf_counter() - t0) / n
out = {"ms_per_step": round(ms, 3), "iters": m.solver_diagnostics()[0]}
try:
    out["solver_us_per_iter"] = round(1e3 * m.scalar("solver_ms_total") / max(m.scalar("solver_iterations_total"), 1.0), 2)
    out["solver_ms"] = round(m.scalar("solver_ms_total") / max(m.scalar("solver_calls_total"), 1.0), 3)
except Exception:
    pass
m.time_manager()
for ph in ("vmix", "tracer_rhs", "impvmixt", "state", "momentum_rhs", "impvmixu", "correct"):
    try:
        out[ph] = round(m.time_phase(ph, reps=10), 3)
    except Exception as e:
        out[ph] = None
print("AB " + json.dumps(out), flush=True)
m.close()
"""


def main():
    wl = sys.argv[1]
    kw = "{}"
    cfgs = sys.argv[2:]
    if cfgs and cfgs[0].startswith("{"):
        kw, cfgs = cfgs[0], cfgs[1:]
    for cfg in cfgs or [""]:
        env = dict(os.environ)
        for kv in cfg.split():
            k, v = kv.split("=", 1)
            if k == "POP_AMD_LIB" and not os.path.isabs(v):
                v = os.path.join(ROOT, v)
            env[k] = v
        p = subprocess.run([sys.executable, "-c", CHILD, ROOT, wl, kw], env=env, capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("AB ")]
        print("%-60s %s" % (cfg or "(default)", line[0][3:] if line else "FAILED rc %d: %s" % (p.returncode, p.stderr[-600:])), flush=True)


if __name__ == "__main__":
    main()
