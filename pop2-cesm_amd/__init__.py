"""pop2-cesm_amd: host-side mirror of the reference's step interface over libpop_amd.so.

The product is the HIP library (csrc/ -> libpop_amd.so, C ABI in include/pop_amd.h).  This
module is plumbing: a ctypes binding whose method names follow the reference routines
(step_mod.F90 `step`, baroclinic.F90 `baroclinic_driver`, barotropic.F90 `barotropic_driver`,
POP_HaloMod `POP_HaloUpdate`, POP_ReductionsMod `POP_GlobalSum`, POP_SolversMod
`POP_SolversRun` / `POP_SolversGetDiagnostics`, blocks.F90 `get_block`), used by bench.py and
the tests.  There is no CPU fallback: a missing library or GPU raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("POP_AMD_LIB") or os.path.join(_HERE, "libpop_amd.so")   # POP_AMD_LIB: build-variant experiments
POP_CREATE_HOST_ONLY = 1
POP_CREATE_PLAN_ONLY = 2

XCHG_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_longlong),
                      C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong))
ALLRED_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_longlong, C.c_longlong)


def build(verbose=False):
    """Compile libpop_amd.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc")]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise RuntimeError("libpop_amd.so is missing: run __graft_entry__.build() (no CPU fallback exists)")
    L = C.CDLL(_SO)
    vp, ci, cd, cs, ll = C.c_void_p, C.c_int, C.c_double, C.c_char_p, C.c_longlong
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int)
    sig = {
        "pop_create": (ci, [vp, ci, ci, ci, C.POINTER(vp)]), "pop_destroy": (ci, [vp]),
        "pop_create_with_grid": (ci, [vp, vp, ci, ci, ci, C.POINTER(vp)]),
        "pop_create_tuned": (ci, [vp, vp, vp, ci, ci, ci, C.POINTER(vp)]),
        "pop_tuning_init": (None, [vp]), "pop_get_tuning": (ci, [vp, vp]),
        "pop_read_grid_files": (ci, [cs, cs, ci, ci, pd, pi]),
        "pop_last_error": (cs, [vp]), "pop_get_dim": (ci, [vp, cs]), "pop_get_scalar": (cd, [vp, cs]),
        "pop_get_block": (ci, [vp, ci, pi, pi, pi]), "pop_local_block_ids": (ci, [vp, pi]),
        "pop_get_field": (ci, [vp, cs, ci, ci, pd, ll]), "pop_set_field": (ci, [vp, cs, ci, ci, pd, ll]),
        "pop_get_ifield": (ci, [vp, cs, pi, ll]), "pop_field_count": (ll, [vp, cs]),
        "pop_field_device_ptr": (vp, [vp, cs, ci, ci]),
        "pop_time_manager": (ci, [vp]), "pop_dhdt": (ci, [vp]), "pop_baroclinic_driver": (ci, [vp]),
        "pop_barotropic_driver": (ci, [vp]), "pop_barotropic_driver_updated": (ci, [vp]), "pop_baroclinic_correct_adjust": (ci, [vp]),
        "pop_step_tail": (ci, [vp]), "pop_step": (ci, [vp]),
        "pop_halo_update": (ci, [vp, cs, ci, ci]),
        "pop_halo_update_host_r8": (ci, [vp, pd, ci, cd]), "pop_halo_update_host_i4": (ci, [vp, pi, ci, ci]),
        "pop_halo_update_loc": (ci, [vp, cs, ci, ci, ci, ci]),
        "pop_halo_update_host_r8_loc": (ci, [vp, pd, ci, cd, ci, ci]), "pop_halo_update_host_i4_loc": (ci, [vp, pi, ci, ci, ci, ci]),
        "pop_global_sum": (ci, [vp, cs, ci, ci, cs, pd]), "pop_solver_run": (ci, [vp]),
        "pop_global_count": (ci, [vp, cs, ci, ci, ci, C.POINTER(ll)]),
        "pop_global_extreme": (ci, [vp, cs, ci, ci, cs, ci, pd, pi, pi]),
        "pop_global_sum_loc": (ci, [vp, cs, ci, ci, cs, ci, pd]),
        "pop_global_sum_host": (ci, [vp, pd, pd, ci, pd]),
        "pop_global_sum_nfields": (ci, [vp, ci, C.POINTER(cs), pi, pi, cs, pd]),
        "pop_global_sum_prod": (ci, [vp, cs, ci, ci, cs, ci, ci, cs, pd]),
        "pop_global_sum_scalar": (ci, [vp, cd, pd]), "pop_global_sum_i4": (ci, [vp, cs, C.POINTER(ll)]),
        "pop_write_restart": (ci, [vp, cs]), "pop_read_restart": (ci, [vp, cs, ci]),
        "pop_solver_diagonal": (ci, [vp, ci, pd]),
        "pop_solver_preconditioner": (ci, [vp, cs, ci, cs, ci]),
        "pop_operator": (ci, [vp, ci, ci, cs, cs, ci, cs, cs]),
        "pop_operator_host": (ci, [vp, ci, ci, ci, vp, vp, vp, vp]),
        "pop_solver_get_diagnostics": (ci, [vp, pi, pd]),
        "pop_state_host": (ci, [vp, ci, pd, pd, pd, pd, pd, ll]),
        "pop_set_comm": (ci, [vp, vp, vp, vp, ll, XCHG_FN, ALLRED_FN, vp]),
        "pop_comm_buffer_doubles": (ll, [vp]), "pop_set_stream": (ci, [vp, vp]),
        "pop_reduce_buffer_doubles": (ll, [vp]), "pop_set_reduce_buffer": (ci, [vp, vp, ll]),
        "pop_rccl_unique_id": (ci, [C.c_char_p]), "pop_comm_init_rccl": (ci, [vp, C.c_char_p]),
        "pop_comm_selftest": (ci, [vp]), "pop_comm_info": (ci, [vp, pi, C.c_char_p, ci]),
        "pop_halo_plan_counts": (ci, [vp, pi, pi, pi]), "pop_halo_plan_peer": (ci, [vp, ci, pi, pi, pi]),
        "pop_halo_plan_lists": (ci, [vp, ci, pi, pi]), "pop_halo_plan_local": (ci, [vp, pi, pi, pi]),
        "pop_timers_reset": (ci, [vp]), "pop_timer_ms": (ci, [vp, cs, pd, pi]),
        "pop_time_phase": (ci, [vp, cs, ci, pd]), "pop_device_sync": (ci, [vp]), "pop_run_phase": (ci, [vp, cs]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)       # AttributeError here = the library does not export the ABI
        f.restype, f.argtypes = res, args
    _lib = L
    return L


ABI_SYMBOLS = None


def abi_symbols():
    """Names declared in include/pop_amd.h (parsed), for the export check."""
    import re
    hdr = open(os.path.join(os.path.dirname(_HERE), "include", "pop_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(pop_[a-z0-9_]+)\s*\(", hdr)) - {"pop_exchange_fn", "pop_allreduce_fn"})


class PopError(RuntimeError):
    pass


def tuning_fields():
    """field names of include/pop_amd.h pop_tuning, in declaration order (all int)"""
    import re
    hdr = open(os.path.join(os.path.dirname(_HERE), "include", "pop_amd.h")).read()
    body = hdr[hdr.index("typedef struct pop_tuning {"):hdr.index("} pop_tuning;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"\bint\s+([^;]+);", body):
        names += [n.strip() for n in decl.split(",")]
    return names


def make_tuning(**kw):
    """pop_tuning with every field unset (pop_tuning_init) except the ones given"""
    names = tuning_fields()
    t = (C.c_int * len(names))()
    lib().pop_tuning_init(C.cast(t, C.c_void_p))
    for k, v in kw.items():
        t[names.index(k)] = int(v)
    return t


class PopGridInput(C.Structure):
    """include/pop_amd.h pop_grid_input"""
    _fields_ = [(n, C.POINTER(C.c_double)) for n in ("ULAT", "ULON", "HTN", "HTE", "HUS", "HUW", "ANGLE")] + [("KMT", C.POINTER(C.c_int)),
                                                                                                               ("DZBC", C.POINTER(C.c_double))]


def grid_input(grid, nx, ny):
    """dict of (ny, nx) arrays -> (PopGridInput, the contiguous arrays it points into)"""
    gin, keep = PopGridInput(), []
    for n in ("ULAT", "ULON", "HTN", "HTE", "HUS", "HUW", "ANGLE", "DZBC"):
        if grid.get(n) is None:
            continue
        a = np.ascontiguousarray(grid[n], dtype=np.float64)
        if a.shape != (ny, nx):
            raise ValueError("grid[%s]: shape %s, expected (ny_global, nx_global) = %s" % (n, a.shape, (ny, nx)))
        keep.append(a)
        setattr(gin, n, a.ctypes.data_as(C.POINTER(C.c_double)))
    if grid.get("KMT") is not None:
        a = np.ascontiguousarray(grid["KMT"], dtype=np.int32)
        if a.shape != (ny, nx):
            raise ValueError("grid[KMT]: shape %s, expected %s" % (a.shape, (ny, nx)))
        keep.append(a)
        gin.KMT = a.ctypes.data_as(C.POINTER(C.c_int))
    return gin, keep


def read_grid_files(horiz_grid_file, topography_file, nx, ny):
    """the reference's direct-access binary files -> the dict PopModel(grid=...) takes"""
    rec = np.empty((7, ny, nx), dtype=np.float64)
    kmt = np.empty((ny, nx), dtype=np.int32)
    e = lib().pop_read_grid_files(horiz_grid_file.encode(), topography_file.encode() if topography_file else None, nx, ny,
                                  rec.ctypes.data_as(C.POINTER(C.c_double)), kmt.ctypes.data_as(C.POINTER(C.c_int)))
    if e:
        raise PopError("pop_read_grid_files: cannot read %s" % (horiz_grid_file if e == 1 else topography_file))
    g = {n: rec[i] for i, n in enumerate(("ULAT", "ULON", "HTN", "HTE", "HUS", "HUW", "ANGLE"))}
    if topography_file:
        g["KMT"] = kmt
    return g


class PopModel:
    """One rank's model instance.  Array views are numpy arrays shaped
    (nblocks_local, [km,] ny_block, nx_block) = the reference layout read in C order."""

    def __init__(self, cfg, rank=0, nranks=1, host_only=False, grid=None, tuning=None, plan_only=False):
        """grid: None (the internal lat-lon grid) or a dict of global (ny_global, nx_global) arrays ULAT, ULON, HTN,
        HTE, HUS, HUW [, ANGLE] [, KMT] -- the records of horiz_grid_file / topography_file (pop_create_with_grid).
        plan_only: block table, distribution and halo plan of this rank only (POP_CREATE_PLAN_ONLY; no fields, no GPU)."""
        self.L = lib()
        self.cfg = cfg
        self.h = C.c_void_p()
        flags = (POP_CREATE_HOST_ONLY if host_only else 0) | (POP_CREATE_PLAN_ONLY if plan_only else 0)
        tun = C.cast(make_tuning(**tuning), C.c_void_p) if tuning else None    # dict of pop_tuning fields (pop_create_tuned)
        if grid is None:
            e = self.L.pop_create_tuned(C.byref(cfg), None, tun, rank, nranks, flags, C.byref(self.h))
        else:
            gin, keep = grid_input(grid, cfg.nx_global, cfg.ny_global)
            e = self.L.pop_create_tuned(C.byref(cfg), C.byref(gin), tun, rank, nranks, flags, C.byref(self.h))
            del keep
        if e:
            msg = self.L.pop_last_error(self.h).decode() if self.h else "pop_create failed"
            raise PopError(msg)
        d = self.dim
        self.nxb, self.nyb, self.km, self.nt = d("nx_block"), d("ny_block"), d("km"), d("nt")
        self.nblocks, self.nblocks_tot = d("nblocks"), d("nblocks_tot")
        self._cb = None

    def close(self):
        if self.h:
            self.L.pop_destroy(self.h)
            self.h = C.c_void_p()

    def _chk(self, e):
        if e:
            raise PopError(self.L.pop_last_error(self.h).decode())

    def dim(self, name):
        return self.L.pop_get_dim(self.h, name.encode())

    def tuning(self):
        """the resolved pop_tuning as a dict (None = the library's size rule applied)"""
        names = tuning_fields()
        t = (C.c_int * len(names))()
        self._chk(self.L.pop_get_tuning(self.h, C.cast(t, C.c_void_p)))
        return {n: (None if t[i] == -2147483648 else t[i]) for i, n in enumerate(names)}

    def scalar(self, name):
        return self.L.pop_get_scalar(self.h, name.encode())

    # ---- blocks.F90 get_block
    def get_block(self, block_id):
        out = (C.c_int * 8)()
        ig, jg = (C.c_int * self.nxb)(), (C.c_int * self.nyb)()
        self._chk(self.L.pop_get_block(self.h, block_id, out, ig, jg))
        keys = ("block_id", "local_id", "ib", "ie", "jb", "je", "iblock", "jblock")
        blk = dict(zip(keys, list(out)))
        blk["i_glob"], blk["j_glob"] = np.array(ig), np.array(jg)
        return blk

    def local_block_ids(self):
        ids = (C.c_int * self.nblocks)()
        self.L.pop_local_block_ids(self.h, ids)
        return list(ids)

    # ---- fields
    def _shape(self, name):
        cnt = self.L.pop_field_count(self.h, name.encode())
        n2 = self.nxb * self.nyb * self.nblocks
        nz = cnt // n2
        return (self.nblocks, self.nyb, self.nxb) if nz == 1 else (self.nblocks, nz, self.nyb, self.nxb)

    def get(self, name, tl=1, n=0):
        shp = self._shape(name)
        a = np.empty(shp, dtype=np.float64)
        self._chk(self.L.pop_get_field(self.h, name.encode(), tl, n, a.ctypes.data_as(C.POINTER(C.c_double)), a.size))
        return a

    def set(self, name, arr, tl=1, n=0):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        self._chk(self.L.pop_set_field(self.h, name.encode(), tl, n, a.ctypes.data_as(C.POINTER(C.c_double)), a.size))

    def geti(self, name):
        a = np.empty((self.nblocks, self.nyb, self.nxb), dtype=np.int32)
        self._chk(self.L.pop_get_ifield(self.h, name.encode(), a.ctypes.data_as(C.POINTER(C.c_int)), a.size))
        return a

    # ---- step_mod.F90 sequence
    def time_manager(self):
        self._chk(self.L.pop_time_manager(self.h))

    def dhdt(self):
        self._chk(self.L.pop_dhdt(self.h))

    def baroclinic_driver(self):
        self._chk(self.L.pop_baroclinic_driver(self.h))

    def barotropic_driver(self, zx_zy_updated=False):
        """zx_zy_updated: the caller has already updated the halos of ZX, ZY (step_mod.F90:405-423)"""
        self._chk((self.L.pop_barotropic_driver_updated if zx_zy_updated else self.L.pop_barotropic_driver)(self.h))

    def baroclinic_correct_adjust(self):
        self._chk(self.L.pop_baroclinic_correct_adjust(self.h))

    def step_tail(self):
        self._chk(self.L.pop_step_tail(self.h))

    def step(self):
        self._chk(self.L.pop_step(self.h))

    def sync(self):
        self._chk(self.L.pop_device_sync(self.h))

    # ---- POP_HaloUpdate / POP_GlobalSum / POP_Solvers*
    def halo_update(self, name, tl=1, n=0):
        self._chk(self.L.pop_halo_update(self.h, name.encode(), tl, n))

    LOC = {"center": 0, "NEcorner": 1, "Nface": 2, "Eface": 3}
    KIND = {"scalar": 0, "vector": 1, "angle": 2}

    def halo_update_loc(self, name, tl=1, n=0, loc="center", kind="scalar"):
        self._chk(self.L.pop_halo_update_loc(self.h, name.encode(), tl, n, self.LOC[loc], self.KIND[kind]))

    def halo_update_host_loc(self, arr, fill=0, loc="center", kind="scalar"):
        a = arr
        nz = a.size // (self.nxb * self.nyb * self.nblocks_tot)
        if a.dtype == np.int32:
            self._chk(self.L.pop_halo_update_host_i4_loc(self.h, a.ctypes.data_as(C.POINTER(C.c_int)), nz, int(fill), self.LOC[loc], self.KIND[kind]))
        else:
            assert a.dtype == np.float64
            self._chk(self.L.pop_halo_update_host_r8_loc(self.h, a.ctypes.data_as(C.POINTER(C.c_double)), nz, float(fill), self.LOC[loc], self.KIND[kind]))

    def halo_update_host(self, arr, fill=0):
        a = arr
        nz = a.size // (self.nxb * self.nyb * self.nblocks_tot)
        if a.dtype == np.int32:
            self._chk(self.L.pop_halo_update_host_i4(self.h, a.ctypes.data_as(C.POINTER(C.c_int)), nz, int(fill)))
        else:
            self._chk(self.L.pop_halo_update_host_r8(self.h, a.ctypes.data_as(C.POINTER(C.c_double)), nz, float(fill)))

    def global_sum(self, name, tl=1, n=0, mask=None):
        r = C.c_double()
        self._chk(self.L.pop_global_sum(self.h, name.encode(), tl, n, mask.encode() if mask else None, C.byref(r)))
        return r.value

    def global_count(self, name, tl=1, n=0, loc="center"):
        r = C.c_longlong()
        self._chk(self.L.pop_global_count(self.h, name.encode(), tl, n, self.LOC[loc], C.byref(r)))
        return r.value

    def global_extreme(self, name, tl=1, n=0, mask=None, want_max=True):
        """(value, iGlobal, jGlobal) of the global maximum / minimum"""
        v, i, j = C.c_double(), C.c_int(), C.c_int()
        self._chk(self.L.pop_global_extreme(self.h, name.encode(), tl, n, mask.encode() if mask else None, 1 if want_max else 0,
                                            C.byref(v), C.byref(i), C.byref(j)))
        return v.value, i.value, j.value

    def global_sum_loc(self, name, tl=1, n=0, mask=None, loc="center"):
        r = C.c_double()
        self._chk(self.L.pop_global_sum_loc(self.h, name.encode(), tl, n, mask.encode() if mask else None, self.LOC[loc], C.byref(r)))
        return r.value

    def global_sum_host(self, arr, mask=None, loc="center"):
        """POP_GlobalSum of a host array of the local blocks (optional multiplicative mask of the same shape)"""
        a = np.ascontiguousarray(arr, dtype=np.float64)
        mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.float64)
        r = C.c_double()
        P = C.POINTER(C.c_double)
        self._chk(self.L.pop_global_sum_host(self.h, a.ctypes.data_as(P), mk.ctypes.data_as(P) if mk is not None else None,
                                             self.LOC[loc], C.byref(r)))
        return r.value

    def global_sum_nfields(self, names, tl=1, n=0, mask=None):
        nf = len(names)
        arr = (C.c_char_p * nf)(*[x.encode() for x in names])
        tls, ns = (C.c_int * nf)(*([tl] * nf)), (C.c_int * nf)(*([n] * nf))
        out = (C.c_double * nf)()
        self._chk(self.L.pop_global_sum_nfields(self.h, nf, arr, tls, ns, mask.encode() if mask else None, out))
        return list(out)

    def global_sum_prod(self, a, b, tl_a=1, n_a=0, tl_b=1, n_b=0, mask=None):
        r = C.c_double()
        self._chk(self.L.pop_global_sum_prod(self.h, a.encode(), tl_a, n_a, b.encode(), tl_b, n_b,
                                             mask.encode() if mask else None, C.byref(r)))
        return r.value

    def global_sum_scalar(self, x):
        r = C.c_double()
        self._chk(self.L.pop_global_sum_scalar(self.h, float(x), C.byref(r)))
        return r.value

    def global_sum_i4(self, name):
        r = C.c_longlong()
        self._chk(self.L.pop_global_sum_i4(self.h, name.encode(), C.byref(r)))
        return r.value

    def solver_diagonal(self, block_local, corr):
        a = np.ascontiguousarray(corr, dtype=np.float64)
        assert a.size == self.nxb * self.nyb
        self._chk(self.L.pop_solver_diagonal(self.h, block_local, a.ctypes.data_as(C.POINTER(C.c_double))))

    def write_restart(self, path):
        """POP binary restart (<path> + <path>.hdr), restart.F90:1095-1715"""
        self._chk(self.L.pop_write_restart(self.h, os.fsencode(path)))

    def read_restart(self, path, byteswap=False):
        self._chk(self.L.pop_read_restart(self.h, os.fsencode(path), 1 if byteswap else 0))

    def solver_preconditioner(self, x_name, px_name, x_tl=1, px_tl=1):
        """PX = M^-1 X on the physical cells (EVP sub-block solves when preconditioner_choice = 1, else the diagonal)"""
        self._chk(self.L.pop_solver_preconditioner(self.h, x_name.encode(), x_tl, px_name.encode(), px_tl))

    def operator(self, op, k, a, b=None, o1="DH", o2="DHU", tl=1):
        """operators.F90 grad / div / zcurl at level k on named device fields (results in o1 [, o2])"""
        self._chk(self.L.pop_operator(self.h, {"grad": 0, "div": 1, "zcurl": 2}[op], k, a.encode(), (b or a).encode(), tl,
                                      o1.encode(), o2.encode()))

    def operator_host(self, op, k, block_local, a, b=None):
        """the reference's argument lists (operators.F90:49, 126, 199) on host arrays (ny_block, nx_block) of ONE local block,
        block_local = this_block%local_id (1-based): grad -> (GRADX, GRADY), div / zcurl -> one array"""
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == (self.nyb, self.nxb)
        bb = np.ascontiguousarray(b if b is not None else a, dtype=np.float64)
        o1, o2 = np.empty_like(a), np.empty_like(a)
        code = {"grad": 0, "div": 1, "zcurl": 2}[op]
        self._chk(self.L.pop_operator_host(self.h, code, k, block_local, a.ctypes.data, bb.ctypes.data if code else None,
                                           o1.ctypes.data, o2.ctypes.data if code == 0 else None))
        return (o1, o2) if code == 0 else o1

    def solver_run(self):
        self._chk(self.L.pop_solver_run(self.h))

    def solver_diagnostics(self):
        it, rms = C.c_int(), C.c_double()
        self.L.pop_solver_get_diagnostics(self.h, C.byref(it), C.byref(rms))
        return it.value, rms.value

    def state(self, kk, T, S, derivs=False):
        T = np.ascontiguousarray(T, dtype=np.float64)
        S = np.ascontiguousarray(S, dtype=np.float64)
        rho = np.empty_like(T)
        dt_ = np.empty_like(T) if derivs else None
        ds_ = np.empty_like(T) if derivs else None
        P = C.POINTER(C.c_double)
        self._chk(self.L.pop_state_host(self.h, kk, T.ctypes.data_as(P), S.ctypes.data_as(P), rho.ctypes.data_as(P),
                                        dt_.ctypes.data_as(P) if derivs else None,
                                        ds_.ctypes.data_as(P) if derivs else None, T.size))
        return (rho, dt_, ds_) if derivs else rho

    # ---- timers / bench support
    def timers_reset(self):
        self.L.pop_timers_reset(self.h)

    def timer(self, name):
        ms, calls = C.c_double(), C.c_int()
        self.L.pop_timer_ms(self.h, name.encode(), C.byref(ms), C.byref(calls))
        return ms.value, calls.value

    def run_phase(self, phase):
        """one phase of baroclinic_driver / baroclinic_correct_adjust on its own (include/pop_amd.h pop_run_phase)"""
        self._chk(self.L.pop_run_phase(self.h, phase.encode()))

    def time_phase(self, phase, reps=10):
        ms = C.c_double()
        self._chk(self.L.pop_time_phase(self.h, phase.encode(), reps, C.byref(ms)))
        return ms.value

    # ---- halo plan introspection
    # ---- in-library RCCL transport (include/pop_amd.h) ----
    @staticmethod
    def rccl_unique_id():
        """128-byte id made on rank 0; broadcast it, then every rank calls comm_init_rccl(id)."""
        buf = C.create_string_buffer(128)
        if lib().pop_rccl_unique_id(buf):
            raise PopError("pop_rccl_unique_id failed (librccl not loadable?)")
        return buf.raw

    def comm_init_rccl(self, id128):
        assert len(id128) == 128
        self._chk(self.L.pop_comm_init_rccl(self.h, C.c_char_p(id128)))

    def comm_selftest(self):
        self._chk(self.L.pop_comm_selftest(self.h))

    def comm_info(self):
        """what the installed transport is (pop_comm_info), as a dict for a run's record"""
        out, path = (C.c_int * 6)(), C.create_string_buffer(256)
        self._chk(self.L.pop_comm_info(self.h, out, path, 256))
        return {"kind": ["none", "host-callbacks", "rccl-native"][out[0]], "ncclCommCount": out[1], "ncclCommUserRank": out[2],
                "ncclCommCount_second_communicator": out[3], "halo_neighbour_ranks": out[4], "midstep_halo_overlap": bool(out[5]),
                "librccl": path.value.decode()}

    def halo_plan(self):
        nl, nf, npeer = C.c_int(), C.c_int(), C.c_int()
        self.L.pop_halo_plan_counts(self.h, C.byref(nl), C.byref(nf), C.byref(npeer))
        dst, src, fill = (C.c_int * max(nl.value, 1))(), (C.c_int * max(nl.value, 1))(), (C.c_int * max(nf.value, 1))()
        self.L.pop_halo_plan_local(self.h, dst, src, fill)
        ia = lambda x: np.array(x, dtype=np.int64)
        plan = {"copy_dst": ia(dst[:nl.value]), "copy_src": ia(src[:nl.value]),
                "fill_dst": ia(fill[:nf.value]), "peers": []}
        for ip in range(npeer.value):
            r, ns, nr = C.c_int(), C.c_int(), C.c_int()
            self.L.pop_halo_plan_peer(self.h, ip, C.byref(r), C.byref(ns), C.byref(nr))
            s, d = (C.c_int * max(ns.value, 1))(), (C.c_int * max(nr.value, 1))()
            self.L.pop_halo_plan_lists(self.h, ip, s, d)
            plan["peers"].append({"rank": r.value, "send_src": ia(s[:ns.value]), "recv_dst": ia(d[:nr.value])})
        return plan
