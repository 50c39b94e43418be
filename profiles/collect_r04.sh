#!/bin/bash
# Collects the round-4 rocprofv3 evidence on the GPU box (run from the repo root through gpurun):
#   * bench lines (tx0.1v3 default incl. cpu_baseline, gx1v7) -> gpurun_out/prof_r04/*_bench.json
#   * kernel-trace stats for tx0.1v3 and gx1v7
#   * PMC passes for tx0.1v3: FETCH_SIZE, WRITE_SIZE (separate passes; FETCH_SIZE doubled on gfx950 by the summariser)
#     and two SQ passes (waves / VALU / wait), no tracing domain combined with --pmc
#   * the N > 1 rehearsal on one GPU: gx1v7 on 2 and 4 ranks through the stand-in librccl (native transport)
# profiles/summarize_r04.py turns the raw output into the committed summaries.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
P=$R/gpurun_out/prof_r04
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $P/tx0.1v3_bench.json 2> $P/tx0.1v3_bench.err
echo "tx bench done"
python3 $R/bench.py --workload gx1v7 --steps 50 --warmup 10 > $P/gx1v7_bench.json 2> $P/gx1v7_bench.err
echo "gx bench done"
# Profiled runs (VERDICT r3 weak #9 / ADVICE: the r3 collection traced bench.py itself, so every per-kernel average mixed the headline steps with the
# deep-boundary-layer steps and the 10 repetitions per phase of pop_time_phase): `--profile-run headline` launches nothing but warm-up + step calls of
# the headline state, `--profile-run deep` nothing but the deep-state steps; each gets its own trace.  Land elimination is active from the first step
# (POP_LAND_FULL_STEPS=0: timing / traffic only) so that two or three steps suffice.
export POP_LAND_FULL_STEPS=0
for wl in gx1v7 tx0.1v3; do
  if [ $wl = gx1v7 ]; then S=20; else S=3; fi
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/${wl}_stats -- python3 $R/bench.py --workload $wl --steps $S --warmup 2 --profile-run headline > $P/${wl}_stats.log 2>&1
  echo "$wl stats done"
done
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/tx0.1v3_deep_stats -- python3 $R/bench.py --workload tx0.1v3 --steps 3 --warmup 6 --profile-run deep > $P/tx0.1v3_deep_stats.log 2>&1
echo "deep stats done"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $ctr --output-format csv -d $P/tx0.1v3_$ctr -- python3 $R/bench.py --workload tx0.1v3 --steps 2 --warmup 1 --profile-run headline > $P/tx0.1v3_$ctr.log 2>&1
  echo "$ctr done"
done
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $P/tx0.1v3_SQ1 -- python3 $R/bench.py --workload tx0.1v3 --steps 2 --warmup 1 --profile-run headline > $P/tx0.1v3_SQ1.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $P/tx0.1v3_SQ2 -- python3 $R/bench.py --workload tx0.1v3 --steps 2 --warmup 1 --profile-run headline > $P/tx0.1v3_SQ2.log 2>&1
echo "SQ done"
unset POP_LAND_FULL_STEPS
for n in 2 4; do
  POP_BENCH_BACKEND=gloo POP_RCCL_LIB=$R/tests/rccl_stub/librccl_stub.so POP_RCCL_STUB_BOX_MB=32 POP_RCCL_STUB_SLOT_MB=16 timeout -k 10 300 \
    python3 -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node $n $R/bench.py --gpus $n --steps 5 --warmup 2 --workload gx1v7 > $P/gx1v7_stub_n$n.json 2> $P/gx1v7_stub_n$n.err
  echo "stub n=$n rc=$?"
done
