#!/usr/bin/env python3
"""Print ms_per_step and selected per-phase timings from a bench.py log (last JSON line)."""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
ph = d["roofline"]["phases"]
names = sys.argv[2:] or list(ph)
print(d["config"]["workload"], "ms/step", d["ms_per_step"], " ".join("%s=%.4f" % (n, ph[n]["ms"]) for n in names if n in ph))
