"""ctypes binding of oracle/libpop_oracle.so (TEST INFRASTRUCTURE: the checker)."""
import ctypes as C
import os
import subprocess
import numpy as np
from popcfg import PopConfig

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# POP_ORACLE_LIB: another build of the same source (bench.py's cpu_baseline times the -O3 build; parity uses the default)
_SO = os.environ.get("POP_ORACLE_LIB") or os.path.join(_ROOT, "oracle", "libpop_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


def load():
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [C.POINTER(PopConfig)]
    L.orc_create_with_grid.restype = C.c_void_p
    L.orc_create_with_grid.argtypes = [C.POINTER(PopConfig), C.c_void_p]
    L.orc_halo.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int]
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_field.restype = C.POINTER(C.c_double)
    L.orc_field.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    L.orc_ifield.restype = C.POINTER(C.c_int)
    L.orc_ifield.argtypes = [C.c_void_p, C.c_char_p]
    L.orc_vfield.restype = C.POINTER(C.c_double)
    L.orc_vfield.argtypes = [C.c_void_p, C.c_char_p]
    L.orc_dim.argtypes = [C.c_void_p, C.c_char_p]
    L.orc_scalar.restype = C.c_double
    L.orc_scalar.argtypes = [C.c_void_p, C.c_char_p]
    for f in ("orc_time_manager", "orc_dhdt", "orc_baroclinic_correct_adjust", "orc_step_tail"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = None
    for f in ("orc_baroclinic_driver", "orc_barotropic_driver", "orc_step", "orc_solver_iterations"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.c_int
    L.orc_baroclinic_stages.argtypes = [C.c_void_p, C.c_int]
    L.orc_baroclinic_stages.restype = C.c_int
    L.orc_solver_rms.argtypes = [C.c_void_p]
    L.orc_solver_rms.restype = C.c_double
    L.orc_state_point.restype = C.c_double
    L.orc_state_point.argtypes = [C.c_double] * 3
    L.orc_halo_update.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int]
    L.orc_halo_update_int.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.orc_global_sum_tripole.restype = C.c_double
    L.orc_global_sum_tripole.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
    L.orc_halo_update_tripole.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int]
    L.orc_halo_update_tripole_int.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int]
    L.orc_preconditioner.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.orc_preconditioner.restype = None
    L.orc_evp_info.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_btrop_operator.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.orc_btrop_operator.restype = None
    L.orc_solver_run.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.orc_operator.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.POINTER(C.c_double)] * 4
    L.orc_operator.restype = None
    L.orc_global_sum.restype = C.c_double
    L.orc_global_sum.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return L


class OrcGridInput(C.Structure):
    """oracle/pop_oracle.h orc_grid_input"""
    _fields_ = [(n, C.POINTER(C.c_double)) for n in ("ULAT", "ULON", "HTN", "HTE", "HUS", "HUW", "ANGLE")] + [("KMT", C.POINTER(C.c_int)),
                                                                                                               ("DZBC", C.POINTER(C.c_double))]


class Oracle:
    """Thin object wrapper; arrays come back as numpy views in Fortran index order
    reversed, i.e. shape (nblocks, [km,] ny_block, nx_block)."""

    def __init__(self, cfg, grid=None):
        self.L = load()
        self.cfg = cfg
        if grid is None:
            self.h = self.L.orc_create(C.byref(cfg))
        else:   # orc_grid_input has pop_grid_input's layout (declared separately on purpose, like the config)
            gin, keep = OrcGridInput(), []
            for n, ty, ct in [(n, np.float64, C.c_double) for n in ("ULAT", "ULON", "HTN", "HTE", "HUS", "HUW", "ANGLE", "DZBC")] + [("KMT", np.int32, C.c_int)]:
                if grid.get(n) is not None:
                    a = np.ascontiguousarray(grid[n], dtype=ty)
                    assert a.shape == (cfg.ny_global, cfg.nx_global), n
                    keep.append(a)
                    setattr(gin, n, a.ctypes.data_as(C.POINTER(ct)))
            self.h = self.L.orc_create_with_grid(C.byref(cfg), C.cast(C.byref(gin), C.c_void_p))
            del keep
        if not self.h:
            raise RuntimeError("orc_create failed")
        d = lambda n: self.L.orc_dim(self.h, n.encode())
        self.nxb, self.nyb, self.km, self.nt, self.nblocks = (d("nx_block"), d("ny_block"), d("km"),
                                                              d("nt"), d("nblocks"))

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def dim(self, n):
        return self.L.orc_dim(self.h, n.encode())

    def scalar(self, n):
        return self.L.orc_scalar(self.h, n.encode())

    def f2(self, name, tl=1, n=0):
        p = self.L.orc_field(self.h, name.encode(), tl, n)
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(self.nblocks, self.nyb, self.nxb))

    def f3(self, name, tl=1, n=0):
        p = self.L.orc_field(self.h, name.encode(), tl, n)
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(self.nblocks, self.km, self.nyb, self.nxb))

    def vdc(self, n=0):
        p = self.L.orc_field(self.h, b"VDC", 0, n)
        return np.ctypeslib.as_array(p, shape=(self.nblocks, self.km + 2, self.nyb, self.nxb))

    def f3p(self, name):
        """(nblocks, km + 2, ny, nx) arrays with levels 0 .. km+1 (DZT, DZU)"""
        p = self.L.orc_field(self.h, name.encode(), 0, 0)
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(self.nblocks, self.km + 2, self.nyb, self.nxb))

    def i2(self, name):
        p = self.L.orc_ifield(self.h, name.encode())
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(self.nblocks, self.nyb, self.nxb))

    def ivec(self, name, n):
        p = self.L.orc_ifield(self.h, name.encode())
        return np.ctypeslib.as_array(p, shape=(n,))

    def v1(self, name):
        p = self.L.orc_vfield(self.h, name.encode())
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(self.km + 3,))

    def preconditioner(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        px = np.empty_like(x)
        self.L.orc_preconditioner(self.h, x.ctypes.data_as(C.POINTER(C.c_double)), px.ctypes.data_as(C.POINTER(C.c_double)))
        return px

    def _dp(self, a):
        return a.ctypes.data_as(C.POINTER(C.c_double))

    def halo(self, a, nz=1):
        a = np.ascontiguousarray(a, dtype=np.float64)
        self.L.orc_halo_update(self.h, self._dp(a), nz, 0)
        return a

    def global_sum(self, a, mask=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.float64)
        return self.L.orc_global_sum(self.h, self._dp(a), None if mk is None else self._dp(mk))

    def btrop_operator(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        ax = np.empty_like(x)
        self.L.orc_btrop_operator(self.h, self._dp(x), self._dp(ax))
        return ax

    def operator(self, op, k, a, b=None):
        """op: 'grad' -> (GX, GY); 'div' / 'zcurl' -> field at T points"""
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = a if b is None else np.ascontiguousarray(b, dtype=np.float64)
        o1, o2 = np.empty_like(a), np.empty_like(a)
        self.L.orc_operator(self.h, {'grad': 0, 'div': 1, 'zcurl': 2}[op], k, self._dp(a), self._dp(b), self._dp(o1), self._dp(o2))
        return (o1, o2) if op == 'grad' else o1

    def solver_run(self, x, b):
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        b = np.ascontiguousarray(b, dtype=np.float64)
        rc = self.L.orc_solver_run(self.h, self._dp(x), self._dp(b))
        return rc, x

    def evp_info(self, what, idx=0):
        return self.L.orc_evp_info(self.h, what, idx)

    STAGE = {"tracer_rhs": 1, "impvmixt": 2, "state": 4, "momentum_rhs": 8, "impvmixu": 16}

    def run_phase(self, phase):
        """one stage of baroclinic_driver on its own (the oracle's vmix_coeffs + tracer_update form one stage)"""
        assert self.L.orc_baroclinic_stages(self.h, self.STAGE[phase]) == 0

    def step(self):
        e = self.L.orc_step(self.h)
        if e:
            raise RuntimeError("oracle solver did not converge")
        return self.L.orc_solver_iterations(self.h)
