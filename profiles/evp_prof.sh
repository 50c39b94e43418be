R=${GRAFT_REPO_ROOT:-$(pwd)}
P=$R/gpurun_out/evp; mkdir -p $P; cd /tmp && export TMPDIR=/tmp
export POP_LAND_FULL_STEPS=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 $R/bench.py --workload tx0.1v3 --steps 2 --warmup 1 --no-cpu-baseline --solver pcsi --precond evp > $P/stats.log 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$P/stats/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-60s %6s %10.1f us avg  %8.2f ms total" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
