"""Several ranks sharing the one GPU of the test box (CPU-staged gloo transport): the multi-rank
paths of libpop_amd (peer halo messages, block-sum all-reduce, replicated barotropic solve) must
reproduce the single-rank run bit for bit.  See tests/mr_gpu_check.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


STUB = os.path.join(ROOT, "tests", "rccl_stub", "librccl_stub.so")
_RDZV_ERRORS = ("address already in use", "EADDRINUSE", "RendezvousConnectionError", "RendezvousTimeoutError", "DistNetworkError",
                "failed to bind", "The server socket has failed")


def _run_check(args, timeout, env=None, transport="staged"):
    """Launch tests/mr_gpu_check.py under torch.distributed.run.  torchrun itself binds a free rendezvous port
    (--standalone), so there is no window between choosing and binding it.  The launch is repeated ONCE, and only
    when there is no verdict AND stderr names a rendezvous / bind error; a run that ends without a verdict for any
    other reason (a rank that aborted or faulted) fails with its output, and a FAILED verdict is never retried."""
    env = dict(os.environ, **(env or {}))
    if transport == "native":
        if not os.path.exists(STUB):
            subprocess.check_call(["make", "-s", "-C", os.path.dirname(STUB)])
        env["POP_RCCL_LIB"] = STUB
        args = args + ["--transport", "native"]
    out = None
    for attempt in range(2):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1"] + args
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        if "MR_GPU_CHECK" in out.stdout or not any(e in out.stderr for e in _RDZV_ERRORS):
            break
    assert "MR_GPU_CHECK OK" in out.stdout, out.stdout[-6000:] + out.stderr[-3000:]
    return out.stdout


def test_gx1v7_two_ranks_equal_single_rank():
    """BASELINE configs[3] rehearsed on one GPU: gx1v7 (KPP, pcg) in two j-bands -- the replicated barotropic solve
    and the 3-D halo exchanges at production size -- is bit for bit the single-rank run."""
    _run_check(["--nproc-per-node", "2", os.path.join(ROOT, "tests", "mr_gpu_check.py"), "--config", "gx1v7", "--steps", "2",
                "--kw", "block_size_y=192"], 600)


def test_large_grid_two_ranks_equal_single_rank():
    """The large-grid kernel selection of tx0.1v3 (del4 + KPP, 62 levels; presummed block sums, two-cell solver
    kernels, column KPP, 64x4 tracer tiles, fused distributed pcg) on a quarter-size domain in two j-bands: bit for
    bit the single-rank run."""
    _run_check(["--nproc-per-node", "2", os.path.join(ROOT, "tests", "mr_gpu_check.py"), "--config", "tx0.1v3", "--steps", "2",
                "--kw", "nx_global=1800,ny_global=1200,block_size_x=1800,block_size_y=600"], 900)


def test_large_grid_pcsi_two_ranks_equal_single_rank():
    """... and with P-CSI, where the large-grid selection is two iterations per launch and per halo exchange on both sides of the
    comparison (k_pcsi_step_x2: single rank with the rings read at their source cells, two ranks with x, dx, r' exchanged two rings wide)."""
    _run_check(["--nproc-per-node", "2", os.path.join(ROOT, "tests", "mr_gpu_check.py"), "--config", "tx0.1v3", "--steps", "2",
                "--kw", "nx_global=1800,ny_global=1200,block_size_x=1800,block_size_y=600,solver_choice=3"], 900)


@pytest.mark.parametrize("nranks,kw,env", [
    (2, "block_size_x=48,block_size_y=20", {}),                       # replicated barotropic solve
    (2, "", {}),                                                      # 16 blocks: fused distributed pcg (one z halo per iteration)
    (2, "", {"POP_SOLVER_UNFUSED": "1"}),                             # kernel-per-operation distributed pcg
    (3, "block_size_x=24,block_size_y=20,vmix_choice=3,km=24", {}),   # uneven block ownership, KPP
    (4, "block_size_x=24,block_size_y=20", {"POP_SOLVER_DISTRIBUTED": "1"}),   # one block per rank, E-W and N-S peers
    (2, "solver_choice=3", {}),                                       # P-CSI fused: one r' halo + one launch per iteration, no collective
    (2, "solver_choice=3", {"POP_SOLVER_UNFUSED": "1"}),              # P-CSI operation by operation
    (4, "solver_choice=3,block_size_x=24,block_size_y=20", {}),       # P-CSI, one block per rank
    (2, "solver_choice=3", {"POP_PCSI_TWO_STEP": "1"}),               # P-CSI two iterations per launch across ranks: x, dx, r' two rings wide once per pair
    (4, "solver_choice=3,block_size_x=24,block_size_y=20,convergence_check_freq=5", {"POP_PCSI_TWO_STEP": "1"}),   # ... one block per rank, pairs and single steps mixed
    (3, "solver_choice=3,block_size_x=20,block_size_y=16", {"POP_PCSI_TWO_STEP": "1"}),   # ... padded blocks, uneven ownership
    (2, "solver_choice=2", {}),                                       # ChronGear, fused distributed form (one all-reduce per iteration)
    (2, "solver_choice=2", {"POP_SOLVER_UNFUSED": "1"}),              # ChronGear operation by operation
    (4, "solver_choice=2,block_size_x=24,block_size_y=20", {}),       # ChronGear fused, one block per rank: corner cells go to three peers
    (2, "", {"POP_HALO_SEPARATE": "1"}),                              # one message per field instead of the batched halo updates
    (2, "precond_choice=1", {}),                                      # EVP preconditioner, pcg: extra z halo per iteration
    (3, "precond_choice=1,solver_choice=3,block_size_x=24,block_size_y=20", {}),   # P-CSI + EVP, uneven ownership
    (2, "tmix_opt=3,tadvect=2", {}),                                  # Robert filter sums + upwind3 across ranks
    (2, "tadvect=3", {}),                                             # lw_lim: halo update of the flux-velocity fields across ranks
    (2, "block_size_x=20,block_size_y=16", {}),                       # padded blocks (3 x 3, last column / row short): fused distributed pcg
    (3, "block_size_x=20,block_size_y=16,solver_choice=2,vmix_choice=3,km=24", {}),   # padded blocks, ChronGear, KPP, one row of blocks per rank
    (2, "hmix_momentum=4,hmix_tracer=4,am=-1.0e22,ah=-1.0e21,lvariable_hmix=1", {"POP_D2T_FUSE": "1"}),   # del4 first Laplacians formed by the previous step's kernels: their ghost ring crosses ranks
])
def test_multirank_equals_single_rank(nranks, kw, env):
    _run_check(["--nproc-per-node", str(nranks), os.path.join(ROOT, "tests", "mr_gpu_check.py"), "--config", "tiny", "--steps", "3",
                "--kw", kw], 300, env)


# ---- the library's own RCCL binding (rccl_transport.hpp: ncclSend/ncclRecv groups + ncclAllReduce on the launch
# stream) with several ranks on the one GPU, through the stand-in librccl of tests/rccl_stub (POP_RCCL_LIB)
@pytest.mark.parametrize("nranks,kw,env", [
    (2, "", {}),                                                      # fused distributed pcg
    (2, "block_size_x=48,block_size_y=20", {}),                       # replicated barotropic solve (one all-reduce gathers RHS + guess)
    (4, "block_size_x=24,block_size_y=20", {"POP_SOLVER_DISTRIBUTED": "1"}),   # one block per rank: E-W, N-S and corner peers
    (2, "solver_choice=2", {}),                                       # ChronGear
    (4, "solver_choice=2,block_size_x=24,block_size_y=20", {}),       # ChronGear, one block per rank
    (2, "", {"POP_SOLVER_OVERLAP_OFF": "1"}),                         # z exchange in line instead of on the side stream
    (2, "", {"POP_RCCL_OVERLAP": "0"}),                               # one communicator only
    (2, "ny_global=80,block_size_x=48,block_size_y=40", {"POP_SOLVER_DISTRIBUTED": "1"}),   # two tall j-band blocks: T,S halo beside the interior tiles
    (2, "ny_global=80,block_size_x=48,block_size_y=40,vmix_choice=3,km=24", {"POP_HALO_OVERLAP_OFF": "1", "POP_SOLVER_DISTRIBUTED": "1"}),
    (3, "ny_global=120,block_size_x=24,block_size_y=40,vmix_choice=3,km=24", {"POP_SOLVER_DISTRIBUTED": "1"}),   # two blocks per rank side by side, three bands
    (4, "solver_choice=3,block_size_x=24,block_size_y=20", {}),       # P-CSI
    (4, "solver_choice=3,block_size_x=24,block_size_y=20,convergence_check_freq=5", {"POP_PCSI_TWO_STEP": "1"}),   # P-CSI, two iterations per launch
    (3, "block_size_x=24,block_size_y=20,vmix_choice=3,km=24", {}),   # uneven ownership, KPP
    (2, "tmix_opt=3,tadvect=2", {}),                                  # Robert filter sums + upwind3
    (4, "tadvect=3,block_size_x=24,block_size_y=20", {}),             # lw_lim, one block per rank
    (2, "hmix_tracer=3,ah=0.8e7,gm_transition_layer=1,gm_kappa_type=1,gm_kappa_freq=1,vmix_choice=3,km=24", {}),   # ... with the transition layer (the filter over HMXL reads ghost columns) and 'bfre' kappa
    (3, "hmix_tracer=3,ah=0.8e7,ah_bolus=0.5e7,vmix_choice=3,km=24,block_size_x=24,block_size_y=20", {}),   # Gent-McWilliams (no exchange of its own: ghost cells of the mix-time tracers), uneven ownership
    (2, "km=62,vmix_choice=3,ny_global=80,block_size_x=48,block_size_y=40", {"POP_VMIXU_DEFER": "1", "POP_SOLVER_DISTRIBUTED": "1"}),   # U,V vertical mixing held back past the distributed solve
    (2, "hmix_momentum=4,hmix_tracer=4,am=-1.0e22,ah=-1.0e21,ny_global=80,block_size_x=48,block_size_y=40", {"POP_D2T_FUSE": "1", "POP_SOLVER_DISTRIBUTED": "1"}),   # del4 first Laplacians formed ahead; the momentum kernel runs in three pieces beside the T,S halo
    (3, "hmix_momentum=4,hmix_tracer=4,am=-1.0e22,ah=-1.0e21,block_size_x=24,block_size_y=20,vmix_choice=3,km=24", {"POP_D2T_FUSE": "1"}),   # the same with uneven ownership and KPP
])
def test_native_transport_equals_single_rank(nranks, kw, env):
    _run_check(["--nproc-per-node", str(nranks), os.path.join(ROOT, "tests", "mr_gpu_check.py"), "--config", "tiny", "--steps", "3",
                "--kw", kw], 300, env, transport="native")


def test_native_transport_large_grid_two_ranks():
    """quarter-size tx0.1v3 in two j-bands over the native transport: the large-grid kernel selection with the
    distributed solver (3-D halo messages of 62 levels through ncclSend/ncclRecv groups)"""
    _run_check(["--nproc-per-node", "2", os.path.join(ROOT, "tests", "mr_gpu_check.py"), "--config", "tx0.1v3", "--steps", "2",
                "--kw", "nx_global=1800,ny_global=1200,block_size_x=1800,block_size_y=600"], 900,
               {"POP_RCCL_STUB_BOX_MB": "16", "POP_SOLVER_DISTRIBUTED": "1"}, transport="native")


def _ops(stdout):
    return float([l for l in stdout.splitlines() if l.startswith("MR_SOLVER_OPS")][0].split()[1])


def test_stream_operations_per_distributed_iteration():
    """What the distributed solvers enqueue per iteration, counted by the library (pop_get_dim solver_stream_ops) and
    averaged over the last solve including its convergence checks (one per 10 iterations: x update, residual, block
    sums, all-reduce, check = 5 more).  pcg: k_fpcg_a(+pack) | block sums | all-reduce || exchange z on the side
    stream, k_fpcg_b(reads the receive buffer) | block sums | all-reduce = 7 enqueued, 6 on the critical path (round 1:
    9, all serial).  ChronGear: exchange | k_fcg_a | block sums | ONE all-reduce | k_fcg_b(+pack) = 5."""
    args = ["--nproc-per-node", "2", os.path.join(ROOT, "tests", "mr_gpu_check.py"), "--config", "tiny", "--steps", "2", "--kw"]
    assert _ops(_run_check(args + [""], 300, transport="native")) <= 7.0 + 0.5
    assert _ops(_run_check(args + ["solver_choice=2"], 300, transport="native")) <= 5.0 + 0.5


# ---- land elimination across ranks (POP_LAND_FULL_STEPS=0: tiles without ocean are skipped from the first step, in the
# multi-rank run and in its single-rank twin alike): the distributed solvers work for their neighbours near the edge of a
# rank's blocks (packing z, advancing ghost cells), so chunks there are never skipped; the compacted chunk lists of the
# fused pcg must agree with that.  'wide' has a continent several 64-column tiles wide.
_SKIP = {"POP_LAND_FULL_STEPS": "0"}


@pytest.mark.parametrize("nranks,kw,env,transport", [
    (2, "block_size_x=528", dict(_SKIP, POP_SOLVER_DISTRIBUTED="1"), "staged"),                   # fused distributed pcg, compacted launches
    (4, "block_size_x=528,solver_choice=2", dict(_SKIP, POP_SOLVER_DISTRIBUTED="1"), "staged"),    # ChronGear, one block per rank
    (2, "block_size_x=528,vmix_choice=3,hmix_momentum=4,hmix_tracer=4,am=-1.0e19,ah=-1.0e18", _SKIP, "staged"),   # replicated solve
    (2, "block_size_x=528", dict(_SKIP, POP_SOLVER_DISTRIBUTED="1"), "native"),
    (4, "block_size_x=528,vmix_choice=3", dict(_SKIP, POP_SOLVER_DISTRIBUTED="1"), "native"),
    (2, "block_size_x=1056,solver_choice=2", dict(_SKIP, POP_SOLVER_DISTRIBUTED="1"), "native"),
    (2, "block_size_x=528,solver_choice=3", _SKIP, "native"),                                       # P-CSI: r' of skipped chunks stays 0 in the messages
])
def test_land_elimination_across_ranks(nranks, kw, env, transport):
    """(see the comment above the parameter list)"""
    _run_check(["--nproc-per-node", str(nranks), os.path.join(ROOT, "tests", "mr_gpu_check.py"), "--config", "wide", "--steps", "4",
                "--kw", kw], 300, env, transport=transport)


# ---- load-balanced distribution (distribution_type = 1): contiguous runs of blocks with equal ocean columns, so ranks own
# different numbers of blocks (the polar ranks more); same numbers as the single-rank run
@pytest.mark.parametrize("nranks,kw,env,transport", [
    (3, "distribution=1,block_size_x=48,block_size_y=5", {}, "staged"),                                   # 8 j-bands over 3 ranks
    (2, "distribution=1,block_size_x=48,block_size_y=4,vmix_choice=3,km=24", {"POP_SOLVER_DISTRIBUTED": "1"}, "staged"),
    (3, "distribution=1,block_size_x=48,block_size_y=5,solver_choice=2", {"POP_SOLVER_DISTRIBUTED": "1"}, "native"),
    (4, "distribution=1,block_size_x=48,block_size_y=4", dict(_SKIP, POP_SOLVER_DISTRIBUTED="1"), "native"),
])
def test_balanced_distribution_equals_single_rank(nranks, kw, env, transport):
    _run_check(["--nproc-per-node", str(nranks), os.path.join(ROOT, "tests", "mr_gpu_check.py"), "--config", "tiny", "--steps", "3",
                "--kw", kw], 300, env, transport=transport)


# ---- tripole fold across ranks: the fold pairs cells of the top row of blocks only, so with that row on one rank (full-width
# or not: here 4 blocks wide, all on the last rank) it is a rank-local pass after the ordinary exchange
@pytest.mark.parametrize("nranks,kw,env,transport", [
    (2, "ns_boundary=2", {}, "staged"),                                                   # 16 blocks, replicated / fused paths as chosen
    (4, "ns_boundary=2,vmix_choice=3,km=24", {"POP_SOLVER_DISTRIBUTED": "1"}, "staged"),  # fused distributed pcg: fold inside srcmap on the top rank
    (3, "ns_boundary=2,solver_choice=2,hmix_momentum=4,hmix_tracer=4,am=-1.0e22,ah=-1.0e21", {"POP_SOLVER_DISTRIBUTED": "1"}, "native"),
    (2, "ns_boundary=2,block_size_x=48,block_size_y=10", {"POP_SOLVER_UNFUSED": "1", "POP_SOLVER_DISTRIBUTED": "1"}, "native"),   # full-width j-bands
    (2, "ns_boundary=2,solver_choice=3", {}, "native"),
    (2, "ns_boundary=2,solver_choice=3", {"POP_PCSI_TWO_STEP": "1"}, "staged"),          # P-CSI two iterations per launch: mirrored ring cells beyond the fold, x, dx, r' two rings wide across the ranks
    (2, "ns_boundary=2,solver_choice=3,block_size_x=48,block_size_y=10,convergence_check_freq=5", {"POP_PCSI_TWO_STEP": "1"}, "native"),   # ... full-width j-bands, pairs and single steps mixed
    (2, "ns_boundary=2,tadvect=3,block_size_x=48,block_size_y=10", {"POP_SOLVER_DISTRIBUTED": "1"}, "native"),   # lw_lim: flux-velocity fields across the fold and the ranks
])
def test_tripole_across_ranks_equals_single_rank(nranks, kw, env, transport):
    _run_check(["--nproc-per-node", str(nranks), os.path.join(ROOT, "tests", "mr_gpu_check.py"), "--config", "tiny", "--steps", "3",
                "--grid", "1", "--kw", kw], 300, env, transport=transport)


@pytest.mark.parametrize("nranks,kw,env", [
    (2, "ny_global=80,block_size_x=48,block_size_y=40", dict(_SKIP, POP_SOLVER_DISTRIBUTED="1")),                    # T,S halo beside the interior tile rows
    (3, "ny_global=120,block_size_x=24,block_size_y=40,vmix_choice=3,km=24", dict(_SKIP, POP_SOLVER_DISTRIBUTED="1")),
])
def test_land_elimination_with_overlapped_halo(nranks, kw, env):
    """tall j-band blocks: the mid-step exchange of the new tracers runs beside the momentum kernel on a window of tile rows
    (windows classify their tiles one by one, the whole launch uses the compacted list)"""
    _run_check(["--nproc-per-node", str(nranks), os.path.join(ROOT, "tests", "mr_gpu_check.py"), "--config", "tiny", "--steps", "3",
                "--kw", kw], 300, env, transport="native")


@pytest.mark.parametrize("nranks", [2, 4])
def test_bench_starts_its_own_ranks(nranks):
    """`python bench.py --gpus N` with NO launcher around it (the shape of the driver's N = 1 command): bench.py starts the N ranks
    itself as a child process before touching the GPU and relays the one JSON line.  Rehearsed on the one GPU through the stand-in
    librccl (torch's own group is gloo, the library binds the stub): the line must carry the evidence that the library's RCCL
    transport spanned N ranks on both communicators."""
    import json
    if not os.path.exists(STUB):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(STUB)])
    env = dict(os.environ, POP_BENCH_BACKEND="gloo", POP_RCCL_LIB=STUB)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--steps", "2", "--warmup", "1", "--workload", "gx1v7"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == nranks and rec["value"] > 0 and rec["config"]["workload"] == "gx1v7"
    mg = rec["multi_gpu"]
    assert mg["transport"] == "rccl-native" and mg["launcher"].startswith("self")
    assert mg["ncclCommCount"] == [nranks] * nranks and mg["ncclCommCount_second_communicator"] == [nranks] * nranks
    assert len(mg["ocean_columns_per_rank"]) == nranks and len(mg["rank_ms_per_step"]) == nranks
    assert rec["roofline"]["phases"]["momentum_rhs"]["alg_words"] == 10 and rec["roofline"]["phases"]["tracer_rhs"]["alg_words"] == 12
    # every rank says which code path it ran (gx1v7: replicated fused solve, register Thomas kernels at km = 60), first attempt
    assert mg["supervised"] and mg["attempt"] == 1 and mg["earlier_attempts"] == []
    assert [pr["solver_path"] for pr in mg["per_rank"]] == ["replicated fused solve on every rank"] * nranks
    assert all(pr["thomas_velocity"] == "register" and len(pr["blocks"]) == 1 for pr in mg["per_rank"])


def _bench_env():
    if not os.path.exists(STUB):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(STUB)])
    env = dict(os.environ, POP_BENCH_BACKEND="gloo", POP_RCCL_LIB=STUB)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)
    return env


def _one_line(out):
    import json
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_recovers_when_the_native_transport_stalls(launcher):
    """The watchdog of `bench.py --gpus N` (VERDICT r3 #1c, #8).  Rank 1 of the first attempt stalls right after the process group is
    up (POP_BENCH_TEST_HANG = "0:1": what a wedged communicator looks like from outside -- every other rank blocks in its first
    collective).  The supervisors must kill the workers' process groups at the limit and run ONE more attempt in fresh processes
    with the torch.distributed transport; the line must come from that attempt and say so.  Both ways of starting the ranks: by
    bench.py itself (the driver's N = 1 command form) and by an external torch.distributed.run (the driver's N > 1 form)."""
    env = dict(_bench_env(), POP_BENCH_TEST_HANG="0:1", POP_BENCH_ATTEMPT_TIMEOUT="75")
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "gx1v7"]
    if launcher == "self":
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", "2"] + args
    rec = _one_line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env))
    mg = rec["multi_gpu"]
    assert rec["n_gpus"] == 2 and rec["value"] > 0
    assert mg["attempt"] == 2 and mg["transport"].startswith("torch.distributed")
    assert len(mg["earlier_attempts"]) == 1 and mg["earlier_attempts"][0]["transport"] == "rccl" and "stalled" in mg["earlier_attempts"][0]["outcome"]


def test_bench_under_an_external_launcher():
    """the driver's N > 1 command form: torch.distributed.run around bench.py; the ranks it starts supervise their workers"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", "2",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "gx1v7"]
    rec = _one_line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_bench_env()))
    mg = rec["multi_gpu"]
    assert mg["transport"] == "rccl-native" and mg["launcher"] == "external" and mg["supervised"] and mg["attempt"] == 1
    assert mg["ncclCommCount"] == [2, 2]


# ---- the band height of `bench.py --gpus 8 --workload tx0.1v3` (VERDICT r3, next #1b): 2400 rows in thirty-two 75-row blocks.  75 is
# not a multiple of the 4-row LDS tile (18 x 4 + 3: the last tile row of every block overhangs), and a rank owns three to five blocks.
_Q75 = "nx_global=1800,ny_global=1200,block_size_x=1800,block_size_y=75,distribution=1"


def test_75_row_bands_equal_the_one_block_run(pkg):
    """A quarter of tx0.1v3 (1800 x 1200 x 62: the large-grid kernel selection -- LDS tiles, register Thomas kernels, column KPP, compacted
    launches) cut into sixteen 75-row blocks on ONE rank against the same domain as one block, land elimination active from the first
    step.  Everything before the first elliptic solve has no reduction in it and must agree BIT FOR BIT on the physical cells; after the
    solve (block sums of the dot products are formed per block) to the tolerance of the solve, with the same iteration counts."""
    import numpy as np
    from popcfg import named_config
    out = {}
    for rows in (1200, 75):
        cfg = named_config("tx0.1v3", nx_global=1800, ny_global=1200, block_size_x=1800, block_size_y=rows)
        m = pkg.PopModel(cfg, tuning={"land_full_steps": 0})

        def glob(name, tl, n=0):
            a = m.get(name, tl, n)
            return np.concatenate([a[b][..., 2:-2, 2:-2] for b in range(m.nblocks)], axis=-2)     # j-bands in block order = global rows

        kmt = np.concatenate([k[2:-2, 2:-2] for k in m.geti("KMT")], axis=-2)
        m.time_manager(); m.dhdt(); m.baroclinic_driver()
        pre = {"T": glob("TRACER", 2, 0), "S": glob("TRACER", 2, 1), "U": glob("UVEL", 2), "V": glob("VVEL", 2),
               "ZX": glob("ZX", 1), "ZY": glob("ZY", 1), "VVC": glob("VVC", 1),
               # land elimination from the FIRST step (land_full_steps = 0): a tile without ocean is never computed, and which cells share
               # a tile depends on the block boundaries, so the boundary-layer depth on land is whatever the buffer held: ocean cells only
               "HBLT": np.where(kmt > 0, glob("HBLT", 1), 0.0)}
        m.barotropic_driver(); m.baroclinic_correct_adjust(); m.step_tail()
        its = [m.solver_diagnostics()[0]]
        for _ in range(3):
            m.step(); its.append(m.solver_diagnostics()[0])
        assert m.dim("land_skip_active") == 1
        post = {"T": glob("TRACER", 1, 0), "U": glob("UVEL", 1), "P": glob("PSURF", 1), "RHO": glob("RHO", 1)}
        out[rows] = (pre, its, post)
        m.close()
    (pa, ia, qa), (pb, ib, qb) = out[1200], out[75]
    for k in pa:
        assert np.array_equal(pa[k], pb[k]), "%s differs before the first solve: max %g" % (k, np.abs(pa[k] - pb[k]).max())
    assert ia == ib, (ia, ib)
    for k in qa:
        e = np.abs(qa[k] - qb[k]).max() / np.abs(qa[k]).max()
        assert e < 1e-8, (k, e)
    assert np.abs(qa["U"]).max() > 0.1


def test_75_row_bands_on_four_ranks_equal_single_rank():
    """the same band height over the native transport on four ranks (balanced by ocean columns: uneven block counts per rank), a
    narrower domain so that four ranks and their single-rank twins fit the one GPU of the test box: 900 x 1200 x 62 is still above
    the 2^19-column threshold of the large-grid kernel selection"""
    _run_check(["--nproc-per-node", "4", os.path.join(ROOT, "tests", "mr_gpu_check.py"), "--config", "tx0.1v3", "--steps", "3", "--no-restart",
                "--kw", "nx_global=900,ny_global=1200,block_size_x=900,block_size_y=75,distribution=1"], 900,
               {"POP_RCCL_STUB_BOX_MB": "16", "POP_SOLVER_DISTRIBUTED": "1", "POP_LAND_FULL_STEPS": "0"}, transport="native")
