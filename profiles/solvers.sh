#!/bin/bash
# the three solvers (and the EVP preconditioner) on one box: ms per step of tx0.1v3
R=${GRAFT_REPO_ROOT:-$(pwd)}
for args in "--solver pcg" "--solver chrongear" "--solver pcsi" "--solver pcsi --precond evp" "--solver pcg --precond evp"; do
  python3 $R/bench.py --steps 8 --warmup 6 --no-cpu-baseline $args 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%-32s %8.3f ms  %6.1f iterations/step' % ('$args', d['ms_per_step'], d['config']['pcg_iters_per_step']))"
done
