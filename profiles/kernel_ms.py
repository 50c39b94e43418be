#!/usr/bin/env python3
"""Average duration per kernel from a rocprofv3 --kernel-trace output directory (any depth)."""
import collections
import csv
import glob
import sys

fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
acc = collections.OrderedDict()
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pop::", "")
    a = acc.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
filt = sys.argv[2:] 
for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if filt and not any(f in k for f in filt):
        continue
    print("%-44s n=%5d avg %10.3f us  total %9.3f ms" % (k[:44], a[0], a[1] / a[0] / 1e3, a[1] / 1e6))
