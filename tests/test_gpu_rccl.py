"""In-library RCCL transport (include/pop_amd.h pop_rccl_unique_id / pop_comm_init_rccl): the binding
is exercised on one rank -- communicator creation, a stream-ordered all-reduce and a grouped
send/recv through librccl -- and must leave the model's results untouched.  N > 1 message logic
(pack / peer lists / unpack / block-sum vectors) is the same code the callback transport runs in
test_gpu_multirank.py; RCCL itself refuses several ranks on one device, so that half runs there."""
import numpy as np
import pytest

from popcfg import named_config

pytestmark = pytest.mark.gpu


def test_rccl_transport_single_rank(pkg):
    cfg = named_config("tiny")
    a = pkg.PopModel(cfg)
    b = pkg.PopModel(cfg)
    uid = pkg.PopModel.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    a.comm_init_rccl(uid)
    a.comm_selftest()
    with pytest.raises(pkg.PopError):
        a.comm_init_rccl(uid)          # a second communicator on the same context is refused
    for _ in range(3):
        a.step(); b.step()
    assert a.solver_diagnostics()[0] == b.solver_diagnostics()[0]
    for name in ("TRACER", "UVEL", "PSURF"):
        assert np.array_equal(a.get(name, 1, 0), b.get(name, 1, 0))
    a.comm_selftest()
    a.close(); b.close()


def test_second_communicator_on_real_rccl(pkg, monkeypatch):
    """POP_RCCL_OVERLAP=2: the second communicator (exchanges on the side stream beside an all-reduce on the first) is
    created on a single rank too -- its id travels through an all-reduce on the first communicator -- and the self-test
    runs a send/recv group on it concurrently with an all-reduce: the real librccl accepts the whole set-up."""
    monkeypatch.setenv("POP_RCCL_OVERLAP", "2")
    m = pkg.PopModel(named_config("tiny"))
    m.comm_init_rccl(pkg.PopModel.rccl_unique_id())
    m.comm_selftest()
    for _ in range(2):
        m.step()
    m.comm_selftest()
    m.close()


def test_selftest_without_transport_fails(pkg):
    m = pkg.PopModel(named_config("tiny"))
    with pytest.raises(pkg.PopError):
        m.comm_selftest()
    m.close()
