cd $GRAFT_REPO_ROOT
run() { echo "== $1 | $2 | $3"; env $3 python3 -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node 2 tests/mr_gpu_check.py --config tiny --steps 2 --grid $1 --kw "$2" 2>&1 | grep -E "differs|MR_GPU_CHECK|iterations|Error" | head -6; }
run 1 "ns_boundary=0" "A=1"
run 1 "ns_boundary=2" "POP_SOLVER_UNFUSED=1"
run 1 "ns_boundary=2,block_size_x=48,block_size_y=10" "A=1"
run 1 "ns_boundary=2,block_size_x=48,block_size_y=20" "A=1"
