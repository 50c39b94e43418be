#!/bin/bash
# Collects the round-1 rocprofv3 evidence on the GPU box (run from the repo root through gpurun):
#   kernel-trace stats and two PMC passes (FETCH_SIZE, WRITE_SIZE -- separate passes, no tracing domains
#   combined with --pmc) for the gx1v7 and tx0.1v3 workloads.  Raw output lands in gpurun_out/prof/;
#   profiles/summarize_r01.py turns it into the committed summaries.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
for wl in gx1v7 tx0.1v3; do
  if [ $wl = gx1v7 ]; then S=20; else S=3; fi
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/${wl}_stats -- python3 $R/bench.py --workload $wl --steps $S --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof/${wl}_stats.log 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 500 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/prof/${wl}_$ctr -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/${wl}_$ctr.log 2>&1
  done
  echo "$wl done"
done
