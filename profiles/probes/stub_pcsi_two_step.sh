R=$GRAFT_REPO_ROOT; P=$R/gpurun_out
for ts in 0 1; do
POP_PCSI_TWO_STEP=$ts POP_BENCH_BACKEND=gloo POP_RCCL_LIB=$R/tests/rccl_stub/librccl_stub.so POP_RCCL_STUB_BOX_MB=32 POP_RCCL_STUB_SLOT_MB=16 timeout -k 10 400 \
  python3 -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node 2 $R/bench.py --gpus 2 --steps 3 --warmup 2 --workload tx0.1v3 --solver pcsi > $P/tx_pcsi_stub_n2_ts$ts.json 2> $P/tx_pcsi_stub_n2_ts$ts.err || exit 1
echo "ts=$ts done"
done
