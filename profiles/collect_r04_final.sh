#!/bin/bash
# End-of-round refresh of the round-4 evidence at HEAD (run from the repo root through gpurun; profiles/collect_r04.sh is the full collection incl. the PMC passes):
# the bench lines and the headline kernel trace of tx0.1v3.  profiles/summarize_r04.py then rebuilds the committed summaries.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
P=$R/gpurun_out/prof_r04
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $P/tx0.1v3_bench.json 2> $P/tx0.1v3_bench.err
echo "tx bench done"
python3 $R/bench.py --workload gx1v7 --steps 50 --warmup 10 > $P/gx1v7_bench.json 2> $P/gx1v7_bench.err
echo "gx bench done"
python3 $R/bench.py --solver pcsi --steps 10 --warmup 5 --no-cpu-baseline > $P/tx0.1v3_pcsi_bench.json 2> $P/tx0.1v3_pcsi_bench.err
echo "tx pcsi bench done"
python3 $R/bench.py --solver chrongear --steps 10 --warmup 5 --no-cpu-baseline > $P/tx0.1v3_chrongear_bench.json 2> $P/tx0.1v3_chrongear_bench.err
echo "tx chrongear bench done"
export POP_LAND_FULL_STEPS=0
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/tx0.1v3_stats -- python3 $R/bench.py --workload tx0.1v3 --steps 3 --warmup 2 --profile-run headline > $P/tx0.1v3_stats.log 2>&1
echo "tx stats done"
