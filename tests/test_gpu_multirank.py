"""Several ranks sharing the one GPU of the test box (CPU-staged gloo transport): the multi-rank
paths of libpop_amd (peer halo messages, block-sum all-reduce, replicated barotropic solve) must
reproduce the single-rank run bit for bit.  See tests/mr_gpu_check.py."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("nranks,kw,env", [
    (2, "block_size_x=48,block_size_y=20", {}),                       # replicated barotropic solve
    (2, "", {}),                                                      # 16 blocks: fused distributed pcg (one z halo per iteration)
    (2, "", {"POP_SOLVER_UNFUSED": "1"}),                             # kernel-per-operation distributed pcg
    (3, "block_size_x=24,block_size_y=20,vmix_choice=3,km=24", {}),   # uneven block ownership, KPP
    (4, "block_size_x=24,block_size_y=20", {"POP_SOLVER_DISTRIBUTED": "1"}),   # one block per rank, E-W and N-S peers
    (2, "solver_choice=3", {}),                                       # P-CSI fused: one r' halo + one launch per iteration, no collective
    (2, "solver_choice=3", {"POP_SOLVER_UNFUSED": "1"}),              # P-CSI operation by operation
    (4, "solver_choice=3,block_size_x=24,block_size_y=20", {}),       # P-CSI, one block per rank
    (2, "solver_choice=2", {}),                                       # ChronGear
    (2, "precond_choice=1", {}),                                      # EVP preconditioner, pcg: extra z halo per iteration
    (3, "precond_choice=1,solver_choice=3,block_size_x=24,block_size_y=20", {}),   # P-CSI + EVP, uneven ownership
    (2, "tmix_opt=3,tadvect=2", {}),                                  # Robert filter sums + upwind3 across ranks
])
def test_multirank_equals_single_rank(nranks, kw, env):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(_port()),
           os.path.join(ROOT, "tests", "mr_gpu_check.py"), "--config", "tiny", "--steps", "3", "--kw", kw]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
    assert "MR_GPU_CHECK OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
