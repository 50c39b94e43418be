#!/bin/bash
# A/B of run-time switches on one box: ms per step of tx0.1v3 (8 timed steps after 6)
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { ( for kv in $1; do export $kv; done; python3 $R/bench.py --steps 8 --warmup 6 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%-40s %8.3f ms' % ('$1', d['ms_per_step']))" ); }
for cfg in "$@"; do run "$cfg"; done
