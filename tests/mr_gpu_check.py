"""Multi-rank run on ONE GPU with a CPU-staged gloo transport (test harness, not a product path).

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
        tests/mr_gpu_check.py --config tiny --steps 4

Every rank drives libpop_amd on cuda:0 with its own blocks; halo messages and the block-sum vector
go device -> host -> gloo -> device.  Rank 0 also runs the same configuration single-rank and all
ranks compare their blocks against it bit for bit (same block decomposition => same arithmetic,
b4b sums).  Exercises pack/unpack kernels, peer plans, callbacks and the unfused solver path that
the N>1 bench uses with RCCL."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="tiny")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--kw", default="")
    ap.add_argument("--no-restart", action="store_true", help="skip the restart round trip (two more models per rank: memory of the large cases)")
    ap.add_argument("--grid", type=int, default=0, help="1: the caller-supplied synthetic grid (pop_create_with_grid), e.g. for ns_boundary=2")
    ap.add_argument("--transport", default="staged", choices=["staged", "native"],
                    help="staged: callback transport over gloo; native: the library's own RCCL binding "
                         "(POP_RCCL_LIB = tests/rccl_stub/librccl_stub.so lets several ranks share one GPU)")
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    import __graft_entry__ as ge
    from popcfg import named_config
    import bench
    pkg = ge.load_package()
    kw = eval("dict(%s)" % args.kw)
    cfg = named_config(args.config, **kw)
    from popcfg import synthetic_grid
    grid = synthetic_grid(cfg) if args.grid else None
    keep = []

    def attach(model):
        """install the transport under test on a multi-rank model"""
        if args.transport == "native":
            box = [pkg.PopModel.rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            model.comm_init_rccl(box[0])
        else:
            keep.append(bench.TorchComm(pkg, model, rank, world, staged=True))

    def calm(model):
        """tripole: the analytic zonal wind has the same value at both copies of a point of the degenerate top row, so the
        symmetrising halo update is not idempotent on what it drives and a restart (which updates halos again) cannot be
        exact -- see tests/test_gpu_restart.py; these runs go without wind"""
        if cfg.ns_boundary == 2:
            z = np.zeros_like(model.get("SMF", n=0))
            for n in (0, 1):
                model.set("SMF", z, n=n); model.set("SMFT", z, n=n)
        return model

    m = calm(pkg.PopModel(cfg, rank=rank, nranks=world, grid=grid))
    attach(m)
    m.comm_selftest()                                          # all-reduce of known values + a self message
    ref = calm(pkg.PopModel(cfg, grid=grid))                  # every rank keeps a single-rank twin
    ids = m.local_block_ids()
    ok = True
    for s in range(args.steps):
        for who, model in (("multi-rank model", m), ("single-rank twin", ref)):
            try:
                model.step()
            except Exception as e:
                raise RuntimeError("%s, step %d: %s" % (who, s, e))
        if m.solver_diagnostics()[0] != ref.solver_diagnostics()[0]:
            print("rank %d step %d: iterations %s vs %s" % (rank, s, m.solver_diagnostics(), ref.solver_diagnostics())); ok = False
        for name in ("TRACER", "UVEL", "VVEL", "PSURF", "UBTROP", "RHO"):
            a = m.get(name, 1, 0); b = ref.get(name, 1, 0)[[i - 1 for i in ids]]
            if not np.array_equal(a, b):
                print("rank %d step %d: %s differs, max %g" % (rank, s, name, np.abs(a - b).max())); ok = False
    if rank == 0:   # stream operations the distributed solver enqueued per iteration in the last solve (0: replicated / unfused path)
        print("MR_SOLVER_OPS %.3f" % (m.dim("solver_stream_ops") / max(m.dim("solver_iterations_enqueued"), 1)))
    # global reductions across ranks: same value / location / count as the single-rank twin
    for want_max in (True, False):
        if m.global_extreme("PSURF", 1, 0, want_max=want_max) != ref.global_extreme("PSURF", 1, 0, want_max=want_max):
            print("rank %d: global_extreme(max=%s) differs" % (rank, want_max)); ok = False
    if m.global_count("UBTROP", 1, 0) != ref.global_count("UBTROP", 1, 0) or m.global_sum("PSURF", 1, 0) != ref.global_sum("PSURF", 1, 0):
        print("rank %d: global count / sum differs" % rank); ok = False
    if args.no_restart:
        t = torch.tensor([1 if ok else 0]); dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if rank == 0:
            print("MR_GPU_CHECK", "OK" if int(t.item()) == 1 else "FAILED", "world", world, "config", args.config, args.kw, "transport", args.transport)
        m.close(); ref.close()
        dist.destroy_process_group()
        sys.exit(0 if int(t.item()) == 1 else 1)
    # restart file written by all ranks together (each its own rows), read back by all ranks and by the single-rank twin
    import tempfile
    path = os.path.join(tempfile.gettempdir(), "mr_restart_%s.bin" % os.environ.get("MASTER_PORT", "0"))
    m.write_restart(path)
    dist.barrier()
    m2 = calm(pkg.PopModel(cfg, rank=rank, nranks=world, grid=grid))
    attach(m2)
    m2.read_restart(path)
    ref2 = calm(pkg.PopModel(cfg, grid=grid)); ref2.read_restart(path)
    for s in range(2):
        m.step(); m2.step(); ref2.step()
    for name in ("TRACER", "UVEL", "PSURF", "UBTROP"):
        a = m.get(name, 1, 0)
        b2, r2 = m2.get(name, 1, 0), ref2.get(name, 1, 0)[[i - 1 for i in ids]]
        if not (np.array_equal(a, b2) and np.array_equal(a, r2)):
            print("rank %d: %s differs after the restart round trip (continued vs re-read on the same ranks: max %g; vs re-read single rank: max %g)"
                  % (rank, name, np.abs(a - b2).max(), np.abs(a - r2).max())); ok = False
    dist.barrier()
    if rank == 0:
        for f in (path, path + ".hdr"):
            if os.path.exists(f):
                os.remove(f)
    m2.close(); ref2.close()
    t = torch.tensor([1 if ok else 0]); dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("MR_GPU_CHECK", "OK" if int(t.item()) == 1 else "FAILED", "world", world, "config", args.config, args.kw, "transport", args.transport)
    m.close(); ref.close()
    dist.destroy_process_group()
    sys.exit(0 if int(t.item()) == 1 else 1)


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        # the launcher keeps only the tail of stderr of the rank it reports: say on stdout which rank failed first and why
        import traceback
        print("MR_GPU_CHECK EXCEPTION rank %s\n%s" % (os.environ.get("RANK"), traceback.format_exc()), flush=True)
        raise
