#!/bin/bash
# kernel-trace statistics with land elimination active from the first step (POP_LAND_FULL_STEPS=0: timing only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
P=$R/gpurun_out/land
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
export POP_LAND_FULL_STEPS=0
export POP_RED_BAND=${POP_RED_BAND:-0}
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats2 -- python3 $R/bench.py --workload tx0.1v3 --steps 4 --warmup 2 --no-cpu-baseline > $P/stats2.log 2>&1
echo stats done
f=$(ls $P/stats2/*/*kernel_stats.csv | head -1); head -40 $f | cut -d, -f1-4 | cut -c1-60,100- > $P/top2.txt
