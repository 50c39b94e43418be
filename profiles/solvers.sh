#!/bin/bash
# the three solvers (and the EVP preconditioner, wavefront kernel and thread-per-sub-block kernel) on one box: ms per step of tx0.1v3
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() {   # $1 = environment assignment (or "-"), rest = bench arguments
  local e=$1; shift
  if [ "$e" = "-" ]; then e=""; fi
  env $e python3 $R/bench.py --steps 8 --warmup 6 --no-cpu-baseline --no-deep-state "$@" 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);s=d['roofline'].get('solver',{});print('%-46s %8.3f ms  %6.1f iterations/step  %7.1f us/iteration' % ('$e $*', d['ms_per_step'], d['config']['pcg_iters_per_step'], s.get('us_per_iteration', float('nan'))))"
}
run - --solver pcg
run - --solver chrongear
run - --solver pcsi
run - --solver pcsi --precond evp
run POP_EVP_WAVE=0 --solver pcsi --precond evp
run - --solver pcg --precond evp
run POP_EVP_WAVE=0 --solver pcg --precond evp
