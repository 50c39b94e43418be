"""Independent pins for the parts of the hot path the reference holds no fixtures for: checks that are
NOT a third restatement of the reference's loops but what the routines are supposed to compute, formed with numpy / scipy:

  * the implicit vertical-mixing solves (vertical_mix.F90:1240-1368 impvmixt, :1563-1658 impvmixt_correct,
    :1750-1868 impvmixu + baroclinic.F90:1077-1129): the tridiagonal system is assembled from VDC / VVC, afac, hfac, H1,
    KMT / KMU and solved by LAPACK's banded solver (scipy.linalg.solve_banded, partial pivoting); the model's solution must
    satisfy the system (residual) and equal LAPACK's;
  * the hydrostatic pressure gradient (pressure_grad.F90:258-301): exactly zero for a horizontally uniform density, and the
    closed form g * a * bouss-weighted trapezoid sum / DXU for a density linear in the i index;
  * horizontal tracer diffusion (hmix_del2.F90:1040-1100, hmix_del4.F90:988-1059): for a field quadratic in the i index the
    Laplacian on the uniform interior of the lat-lon grid is 2a * (east coefficient), a function of j only, so the
    biharmonic operator reduces to a three-point formula in j.

The same functions run against the CPU oracle (tests/test_pins_oracle.py, no GPU) and against the device library through the
C ABI (tests/test_gpu_pins.py): `A` is an adapter with get / set / run_phase (see OracleAdapter, GpuAdapter)."""
import numpy as np
from scipy.linalg import solve_banded

GRAV = 980.6           # pop_constants.F90:235 (non-CCSMCOUPLED)
THREE_D = ("TRACER", "UVEL", "VVEL", "RHO", "VVC", "KPP_SRC", "UISOP", "VISOP", "WISOP", "GM_SF_SLX", "GM_SF_SLY")


class OracleAdapter:
    def __init__(self, cfg, grid=None):
        from orclib import Oracle
        self.o = Oracle(cfg, grid=grid)
        self.cfg = cfg
        self.km, self.nblocks = self.o.km, self.o.nblocks

    def _arr(self, name, tl, n):
        if name == "VDC":
            return self.o.vdc(n)
        name = {"D4AHF": "D4_AHF"}.get(name, name)
        return (self.o.f3 if name in THREE_D else self.o.f2)(name, tl, n)

    def get(self, name, tl=1, n=0):
        return self._arr(name, tl, n).copy()

    def set(self, name, arr, tl=1, n=0):
        self._arr(name, tl, n)[...] = arr

    def geti(self, name):
        return self.o.i2(name).copy()

    def vert(self, name):
        return self.o.v1(name).copy()       # 1-based with slot 0

    def scalar(self, name):
        return self.o.scalar(name)

    def step(self):
        self.o.step()

    def time_manager(self):
        self.o.L.orc_time_manager(self.o.h)

    def dhdt(self):
        self.o.L.orc_dhdt(self.o.h)

    def run_phase(self, phase):
        if phase in ("vmix", "hmix_tracer", "hmix_momentum"):
            return                            # the oracle forms them inside tracer_rhs / momentum_rhs
        self.o.run_phase(phase)

    def correct(self):
        self.o.L.orc_baroclinic_correct_adjust(self.o.h)

    def leapfrog(self):
        return self.o.dim("leapfrogts")

    def close(self):
        self.o.close()


class GpuAdapter:
    """the device library through the C ABI; vertical grid arrays come from a host-only oracle of the same configuration
    (they are init-time data, compared bit for bit in tests/test_host_grid_parity.py)"""

    def __init__(self, pkg, cfg, grid=None):
        from orclib import Oracle
        self.m = pkg.PopModel(cfg, grid=grid)
        self.cfg = cfg
        self.km, self.nblocks = self.m.km, self.m.nblocks
        o = Oracle(cfg, grid=grid)
        self._vert = {k: o.v1(k).copy() for k in ("dz", "dzw", "dzwr", "bouss", "afac_t", "afac_u", "zt", "zw")}
        o.close()

    def get(self, name, tl=1, n=0):
        return self.m.get({"d4DTE": "d4DTE"}.get(name, name), tl, n)

    def set(self, name, arr, tl=1, n=0):
        self.m.set(name, arr, tl, n)

    def geti(self, name):
        return self.m.geti(name)

    def vert(self, name):
        return self._vert[name]

    def scalar(self, name):
        return self.m.scalar(name)

    def step(self):
        self.m.step()

    def time_manager(self):
        self.m.time_manager()

    def dhdt(self):
        self.m.dhdt()

    def run_phase(self, phase):
        self.m.run_phase(phase)

    def correct(self):
        self.m.baroclinic_correct_adjust()
        self.m.sync()

    def leapfrog(self):
        return self.m.dim("leapfrogts")

    def close(self):
        self.m.close()


def interior(a):
    return a[..., 2:-2, 2:-2]


def _prepare(A, nsteps=2):
    """a developed state, then the step parameters of a leapfrog step"""
    for _ in range(nsteps):
        A.step()
    A.time_manager()
    assert A.leapfrog() == 1
    return 2.0 * A.scalar("dtt")          # c2dtt = c2dtu (step_mod.F90:302-320; dt(k) = dtt, dtu = dtt)


def _tridiag_solve(h, a_face, kbot, rhs_h):
    """LAPACK solution of the columns' systems.  Row k (1-based, k <= kbot):
         -A(k-1) x(k-1) + (h(k) + A(k-1) + A(k)) x(k) - A(k) x(k+1) = rhs_h(k),   A(0) = A(kbot) = 0
    (what the recurrences of vertical_mix.F90:1263-1368 eliminate); rows below the bottom: x = 0.
    h, a_face, rhs_h: (km, ncol); kbot: (ncol,).  Returns x (km, ncol) and the residual of a given solution."""
    km, ncol = h.shape
    x = np.zeros((km, ncol))
    for c in range(ncol):
        kb = int(kbot[c])
        if kb < 1:
            continue
        A = a_face[:kb, c].copy()
        A[kb - 1] = 0.0
        Am = np.concatenate(([0.0], A[:-1]))
        ab = np.zeros((3, kb))
        ab[0, 1:] = -A[:-1]
        ab[1, :] = h[:kb, c] + Am + A
        ab[2, :-1] = -A[:-1]
        x[:kb, c] = solve_banded((1, 1), ab, rhs_h[:kb, c])
    return x


def _tridiag_residual(h, a_face, kbot, rhs_h, x):
    km, ncol = h.shape
    k = np.arange(1, km + 1)[:, None]
    A = np.where(k < kbot[None, :], a_face, 0.0)            # no flux through the bottom face or below
    Am = np.vstack([np.zeros((1, ncol)), A[:-1]])
    xp = np.vstack([x[1:], np.zeros((1, ncol))])
    xm = np.vstack([np.zeros((1, ncol)), x[:-1]])
    r = -Am * xm + (h + Am + A) * x - A * xp - rhs_h
    r = np.where(k <= kbot[None, :], r, x)                  # below the bottom the solution itself must vanish
    return np.where(kbot[None, :] > 0, r, 0.0)              # land columns: the reference leaves don't-care values at k = 1


def _thickness(A, kind):
    """level thicknesses per interior column, (km, ncol): dz(k), or with partial bottom cells (grid.F90:926-951) DZT = DZBC at
    the bottom level of a T column, and DZU = the minimum of the four surrounding DZT at a U column -- formed here from KMT and
    DZBC alone, independently of what either implementation stores"""
    km = A.km
    dz = A.vert("dz")[1:km + 1]
    kmt = A.geti("KMT")
    if not A.cfg.partial_bottom_cells:
        ncol = interior(kmt).size
        return np.repeat(dz[:, None], ncol, axis=1)
    dzbc = A.get("DZBC")
    k = np.arange(1, km + 1)[None, :, None, None]
    dzt = np.where(k == kmt[:, None], dzbc[:, None], dz[None, :, None, None])          # (nblocks, km, ny, nx), ghosts included
    if kind == "T":
        return _cols(dzt)
    dzu = dzt.copy()
    dzu[..., :-1, :-1] = np.minimum(np.minimum(dzt[..., :-1, :-1], dzt[..., :-1, 1:]), np.minimum(dzt[..., 1:, :-1], dzt[..., 1:, 1:]))
    return _cols(dzu)


def _faces(dzc, coef):
    """A(k) = coefficient(k) / (distance between the centres of levels k and k + 1) for per-column thicknesses dzc (km, ncol)"""
    below = np.vstack([dzc[1:], np.zeros((1, dzc.shape[1]))])
    return coef / (0.5 * (dzc + below))


def _cols(a3):
    """(nblocks, km, ny, nx) interior -> (km, ncol)"""
    b = interior(a3)
    return np.moveaxis(b, 1, 0).reshape(b.shape[1], -1)


def check_impvmixt(A, rng, tol_res=1e-13, tol_sol=1e-12):
    """impvmixt (predictor, PSFC = PSURF(cur)): T(new) = T(old) + x, x from the tridiagonal system with right-hand side
    hfac_t(k) * rhs(k) and H1 = hfac_t(1) + PSFC / (grav c2dtt(1)) on the diagonal of the first row"""
    c2dt = _prepare(A)
    km = A.km
    dz, afac = A.vert("dz")[1:km + 1], A.vert("afac_t")[1:km + 1]
    kmt = A.geti("KMT")
    # KPP keeps one diffusivity per tracer class; without double diffusion the two classes hold the same values (the library then
    # keeps ONE array for both), so distinct values per class are pinned with ldbl_diff only
    nset = 2 if A.cfg.vmix_choice == 3 else 1
    nvdc = 2 if (A.cfg.vmix_choice == 3 and A.cfg.ldbl_diff) else 1
    shp = A.get("TRACER", 2, 0).shape
    ps = 40.0 * GRAV * (rng.random(A.get("PSURF", 1).shape) - 0.5)       # +- 20 cm of sea level
    A.set("PSURF", ps, 1)
    vdc = [0.05 + 50.0 * rng.random(A.get("VDC", 0, n).shape) ** 3 for n in range(2)]
    for n in range(nset):
        A.set("VDC", vdc[n if nvdc == 2 else 0], 0, n)
    rhs = [10.0 * (rng.random(shp) - 0.5) for _ in range(2)]
    told = [A.get("TRACER", 0, n) for n in range(2)]
    for n in range(2):
        A.set("TRACER", rhs[n], 2, n)
    A.run_phase("impvmixt")
    worst = 0.0
    for n in range(2):
        out = A.get("TRACER", 2, n)
        v = vdc[n if nvdc == 2 else 0]
        a_face = afac[:, None] * _cols(v[:, 1:km + 1])
        h = np.repeat((dz / c2dt)[:, None], a_face.shape[1], axis=1)
        if A.cfg.partial_bottom_cells:      # the same system on the columns' own thicknesses
            dzc = _thickness(A, "T")
            a_face, h = _faces(dzc, _cols(v[:, 1:km + 1])), dzc / c2dt
        rhs_h = h * _cols(rhs[n])
        h[0] = h[0] + interior(ps).reshape(-1) / (GRAV * c2dt)
        kb = interior(kmt).reshape(-1)
        x = _cols(out) - _cols(told[n])
        scale = np.abs(rhs_h).max()
        res = np.abs(_tridiag_residual(h, a_face, kb, rhs_h, x)).max() / scale
        ref = _tridiag_solve(h, a_face, kb, rhs_h)
        err = np.abs(np.where(kb[None, :] > 0, x - ref, 0.0)).max() / np.abs(ref).max()
        assert res <= tol_res, "impvmixt tracer %d: residual %.3e" % (n, res)
        assert err <= tol_sol, "impvmixt tracer %d: differs from LAPACK by %.3e" % (n, err)
        worst = max(worst, res, err)
    assert (interior(kmt) == 0).any() and (interior(kmt) > 0).any()     # land and ocean columns were both checked
    return worst


def check_impvmixt_correct(A, rng, tol_res=1e-13, tol_sol=1e-12):
    """impvmixt_correct (baroclinic.F90:1261-1330 + vertical_mix.F90:1563-1658), pressure-averaging leapfrog step:
    T(new) += x with the same matrix (PSFC = PSURF(new)) and the right-hand side hfac_t(1) * RHS1 in the first row only,
    RHS1 = ((2 T(cur) - T(old)) (P(cur) - P(old)) - T(new) (P(new) - P(cur))) / (grav dz(1)) at the surface level"""
    c2dt = _prepare(A)
    assert A.cfg.lpressure_avg == 1
    km = A.km
    dz, afac = A.vert("dz")[1:km + 1], A.vert("afac_t")[1:km + 1]
    kmt = A.geti("KMT")
    # KPP keeps one diffusivity per tracer class; without double diffusion the two classes hold the same values (the library then
    # keeps ONE array for both), so distinct values per class are pinned with ldbl_diff only
    nset = 2 if A.cfg.vmix_choice == 3 else 1
    nvdc = 2 if (A.cfg.vmix_choice == 3 and A.cfg.ldbl_diff) else 1
    shp2 = A.get("PSURF", 1).shape
    P = [40.0 * GRAV * (rng.random(shp2) - 0.5) for _ in range(3)]
    for tl in range(3):
        A.set("PSURF", P[tl], tl)
    vdc = [0.05 + 50.0 * rng.random(A.get("VDC", 0, n).shape) ** 3 for n in range(2)]
    for n in range(nset):
        A.set("VDC", vdc[n if nvdc == 2 else 0], 0, n)
    tn = [A.get("TRACER", 1, n) * (1.0 + 0.05 * (rng.random(A.get("TRACER", 1, n).shape) - 0.5)) for n in range(2)]
    tc = [A.get("TRACER", 1, n) for n in range(2)]
    to = [A.get("TRACER", 0, n) for n in range(2)]
    for n in range(2):
        A.set("TRACER", tn[n], 2, n)
    A.correct()
    worst = 0.0
    for n in range(2):
        out = A.get("TRACER", 2, n)
        v = vdc[n if nvdc == 2 else 0]
        a_face = afac[:, None] * _cols(v[:, 1:km + 1])
        h = np.repeat((dz / c2dt)[:, None], a_face.shape[1], axis=1)
        if A.cfg.partial_bottom_cells:
            dzc = _thickness(A, "T")
            a_face, h = _faces(dzc, _cols(v[:, 1:km + 1])), dzc / c2dt
        kb = interior(kmt).reshape(-1)
        rhs1 = ((2.0 * tc[n][:, 0] - to[n][:, 0]) * (P[1] - P[0]) - tn[n][:, 0] * (P[2] - P[1])) / (GRAV * dz[0])
        rhs_h = np.zeros_like(h)
        rhs_h[0] = h[0] * np.where(kb > 0, interior(rhs1).reshape(-1), 0.0)
        h[0] = h[0] + interior(P[2]).reshape(-1) / (GRAV * c2dt)
        x = _cols(out) - _cols(tn[n])
        scale = np.abs(rhs_h).max()
        res = np.abs(_tridiag_residual(h, a_face, kb, rhs_h, x)).max() / scale
        ref = _tridiag_solve(h, a_face, kb, rhs_h)
        # x is a difference of O(20) numbers: its rounding floor is eps * |T| / |x|
        floor = 4.0e-16 * np.abs(_cols(tn[n])).max() / np.abs(ref).max()
        err = np.abs(np.where(kb[None, :] > 0, x - ref, 0.0)).max() / np.abs(ref).max()
        assert res <= tol_res + 10 * floor, "impvmixt_correct tracer %d: residual %.3e" % (n, res)
        assert err <= tol_sol + 10 * floor, "impvmixt_correct tracer %d: differs from LAPACK by %.3e" % (n, err)
        worst = max(worst, res, err)
    return worst


def check_impvmixu(A, rng, tol_sol=1e-12):
    """impvmixu + the tail of baroclinic_driver: x from the tridiagonal system (hfac_u, afac_u VVC, bottom at KMU), then
    U(new) = U(old) + x minus its depth mean (sum over k of U dz / HU) on k <= KMU, 0 below"""
    c2dt = _prepare(A)
    km = A.km
    dz, afac = A.vert("dz")[1:km + 1], A.vert("afac_u")[1:km + 1]
    kmu = A.geti("KMU")
    vvc = 0.5 + 100.0 * rng.random(A.get("VVC").shape) ** 3
    A.set("VVC", vvc)
    shp = vvc.shape
    rhs = [5.0 * (rng.random(shp) - 0.5) for _ in range(2)]
    old = [A.get(f, 0) for f in ("UVEL", "VVEL")]
    hur = A.get("HUR")
    for f, r in zip(("UVEL", "VVEL"), rhs):
        A.set(f, r, 2)
    A.run_phase("impvmixu")
    worst = 0.0
    kb = interior(kmu).reshape(-1)
    klev = np.arange(1, km + 1)[:, None]
    for f, r, o in zip(("UVEL", "VVEL"), rhs, old):
        out = _cols(A.get(f, 2))
        a_face = afac[:, None] * _cols(vvc)
        h = np.repeat((dz / c2dt)[:, None], a_face.shape[1], axis=1)
        dzc = np.repeat(dz[:, None], a_face.shape[1], axis=1)
        if A.cfg.partial_bottom_cells:
            dzc = _thickness(A, "U")
            a_face, h = _faces(dzc, _cols(vvc)), dzc / c2dt
            # the depth the mean divides by is the sum of the column's thicknesses (grid.F90:1011-1014)
            hu = np.where(klev <= kb[None, :], dzc, 0.0).sum(axis=0)
            assert np.allclose(np.where(kb > 0, hu * interior(hur).reshape(-1), 1.0), 1.0, rtol=1e-13)
        x = _tridiag_solve(h, a_face, kb, h * _cols(r))
        unew = _cols(o) + x
        mean = (unew * dzc).sum(axis=0) * interior(hur).reshape(-1)
        expect = np.where(klev <= kb[None, :], unew - mean[None, :], 0.0)
        err = np.abs(out - expect).max() / np.abs(expect).max()
        assert err <= tol_sol, "%s after impvmixu: differs from the LAPACK-based value by %.3e" % (f, err)
        worst = max(worst, err)
    assert (kb == 0).any() and (kb > 0).any()
    return worst


def _quiet(A):
    """no flow, flat sea surface, no wind: only the pressure-gradient / diffusion terms under test remain"""
    z3 = np.zeros(A.get("UVEL", 1).shape)
    for tl in range(3):
        A.set("UVEL", z3, tl)
        A.set("VVEL", z3, tl)
        A.set("PSURF", np.zeros(A.get("PSURF", 1).shape), tl)
    for n in range(2):
        A.set("SMF", np.zeros(A.get("SMF", 1, n).shape), 1, n)


def _forces(A, c2dt):
    """FX, FY of clinic from U, V(new) = (FX + w FY, FY - w FX) c2dtu / (1 + w^2), w = c2dtu beta FCOR (baroclinic.F90:1013-1045)"""
    u, v = A.get("UVEL", 2), A.get("VVEL", 2)
    if not A.cfg.impcor:
        return u / c2dt, v / c2dt
    w = (c2dt * (1.0 / 3.0) * A.get("FCOR"))[:, None]
    return (u - w * v) / c2dt, (v + w * u) / c2dt


def check_gradp(A, tol=1e-12):
    c2dt = _prepare(A)
    km = A.km
    _quiet(A)
    kmu = A.geti("KMU")
    shp = A.get("RHO", 1).shape
    prof = 1.0 + np.round(40.0 * np.arange(km) / km) * 2.0 ** -10    # rho(z), g/cm^3, exactly representable
    # (1) horizontally uniform density: no pressure force anywhere, exactly
    for tl in range(3):
        A.set("RHO", np.broadcast_to(prof[None, :, None, None], shp).copy(), tl)
    A.run_phase("hmix_momentum")
    A.run_phase("momentum_rhs")
    for f in ("UVEL", "VVEL"):
        assert np.abs(interior(A.get(f, 2))).max() == 0.0, "gradp of rho(z) is not exactly zero (%s)" % f
    assert np.abs(interior(A.get("ZX"))).max() == 0.0 and np.abs(interior(A.get("ZY"))).max() == 0.0
    # (2) rho = rho(z) + a * (i index within the block): grad_x = a bouss(k) / DXU exactly, grad_y = 0;
    # hydrostatic sum (pressure_grad.F90:280-296): SUMX(k) = SUMX(k-1) + dzw(k-1) grav 0.5 (gx(k) + gx(k-1)), gx(0) = gx(1)
    a = 2.0 ** -7                                                   # large on purpose: the check is arithmetic, and a difference of
                                                                    # densities must stay far above their rounding (4e-16)
    ii = np.arange(shp[-1], dtype=np.float64)[None, None, None, :]
    for tl in range(3):
        A.set("RHO", prof[None, :, None, None] + a * ii + np.zeros(shp), tl)
    A.run_phase("hmix_momentum")
    A.run_phase("momentum_rhs")
    fx, fy = _forces(A, c2dt)
    bouss, dzw = A.vert("bouss")[1:km + 1], A.vert("dzw")[0:km]
    gx = a * bouss[None, :, None, None] / A.get("DXU")[:, None]
    gxm = np.concatenate([gx[:, :1], gx[:, :-1]], axis=1)
    sumx = np.cumsum(dzw[None, :, None, None] * GRAV * 0.5 * (gx + gxm), axis=1)
    ocean = np.arange(1, km + 1)[None, :, None, None] <= kmu[:, None]
    expect = np.where(ocean, -sumx, 0.0)
    # the i index is linear inside a block only: leave out the two columns next to a block's eastern edge
    sel = (slice(None), slice(None), slice(2, -2), slice(2, -4))
    err = np.abs(fx[sel] - expect[sel]).max() / np.abs(expect[sel]).max()
    assert err <= tol, "gradp of a density linear in x: FX differs from the closed form by %.3e" % err
    assert np.abs(fy[sel]).max() <= tol * np.abs(expect[sel]).max(), "gradp of a density linear in x: FY is not zero"
    assert ocean[sel].any() and (~ocean[sel]).any()
    return err


def check_hdifft(A, tol=None):
    """tracer_update with nothing but horizontal diffusion acting (no flow, level-independent tracers, no fluxes):
    T(new) = c2dtt * HDTK.  T = a i^2 (i = index within the block) away from land and block edges:
      del2:  HDTK = ah * 2a * DTE(j)                                    (DTE = DTW on the lat-lon grid, hmix_del2.F90:619-634)
      del4:  D2 = AHF * 2a * DTE(j);  HDTK = ah * (DTN (D2(j+1) - D2(j)) + DTS (D2(j-1) - D2(j)))   (hmix_del4.F90:1021-1059)"""
    c2dt = _prepare(A)
    km = A.km
    _quiet(A)
    del4 = A.cfg.hmix_tracer == 4
    # rounding floor: the first Laplacian cancels 2i of 2i + 1 parts (i up to ~50), the second differences first
    # Laplacians that agree to ~1e-3 between neighbouring rows: 1e-16 * 1e2 (* 1e3) -- a wrong coefficient shows at O(1)
    tol = tol or (1e-9 if del4 else 1e-12)
    pre = "d4" if del4 else ""
    dte, dtw, dtn, dts = (A.get(pre + n) for n in ("DTE", "DTW", "DTN", "DTS"))
    kmt = A.geti("KMT")
    shp = A.get("TRACER", 1, 0).shape
    a = 2.0 ** -6
    ii = np.arange(shp[-1], dtype=np.float64)[None, None, None, :]
    field = [a * ii * ii + np.zeros(shp), 0.5 * a * ii * ii + 0.01 + np.zeros(shp)]
    for n in range(2):
        for tl in range(3):
            A.set("TRACER", field[n], tl, n)
        A.set("STF", np.zeros(kmt.shape), 1, n)
    if A.cfg.vmix_choice == 3:
        for n in range(2):
            A.set("KPP_SRC", np.zeros(shp), 1, n)
    A.dhdt()
    A.run_phase("hmix_tracer")
    A.run_phase("tracer_rhs")
    # cells whose whole stencil (radius 2 for del4 incl. the land test) is full-depth ocean inside one block
    full = (kmt == km)
    ok = full.copy()
    r = 3
    for dj in range(-r, r + 1):
        for di in range(-r, r + 1):
            ok &= np.roll(np.roll(full, dj, axis=1), di, axis=2)
    ok[:, :r + 2, :] = False; ok[:, -(r + 2):, :] = False; ok[:, :, :r + 2] = False; ok[:, :, -(r + 2):] = False
    assert ok.any(), "no open-ocean patch in this configuration"
    ah = A.cfg.ah
    worst = 0.0
    for n, amp in ((0, a), (1, 0.5 * a)):
        out = A.get("TRACER", 2, n) / c2dt
        if del4:
            ahf = A.get("D4AHF") if A.cfg.lvariable_hmix else np.ones(dte.shape)
            d2 = ahf * 2.0 * amp * dte
            expect = ah * (dtn * (np.roll(d2, -1, axis=1) - d2) + dts * (np.roll(d2, 1, axis=1) - d2))
        else:
            expect = ah * 2.0 * amp * dte
        assert np.abs(dte - dtw)[ok].max() <= 1e-15 * np.abs(dte)[ok].max()
        for k in range(km):
            e = np.abs(out[:, k] - expect)[ok].max() / np.abs(expect)[ok].max()
            worst = max(worst, e)
    assert worst <= tol, "hdifft of a quadratic field differs from the closed form by %.3e" % worst
    return worst


# ------------------------------------------------------------------------------------------------------------------
# KPP (vmix_kpp.F90) against Large, McWilliams & Doney (1994): the published similarity functions, the boundary-layer
# shape function and the bulk-Richardson-number depth on a two-layer column.  Nothing below evaluates the model's own
# formulas: the expected values are written from the paper (appendix B for phi_m, phi_s and their constants) and from the
# geometry of the column.
# ------------------------------------------------------------------------------------------------------------------
VONKAR, EPSSFC, RICR = 0.4, 0.1, 0.3           # von Karman constant, surface-layer extent, critical bulk Ri (LMD94 2, 21)
ZETA_M, ZETA_S, A_M, C_M, A_S, C_S = -0.2, -1.0, 1.26, 8.38, -28.86, 98.96    # LMD94 (B1), (B2)
CSTAR = 10.0                                   # non-local transport, LMD94 (20)
KPP_EPS = 1.0e-10                              # vmix_kpp.F90:104


def phi_m(zeta):
    """LMD94 (B1): 1 + 5 zeta | (1 - 16 zeta)^-1/4 | (a_m - c_m zeta)^-1/3"""
    z = np.asarray(zeta, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        return np.where(z >= 0.0, 1.0 + 5.0 * z, np.where(z >= ZETA_M, (1.0 - 16.0 * z) ** -0.25, (A_M - C_M * z) ** (-1.0 / 3.0)))


def phi_s(zeta):
    """LMD94 (B1): 1 + 5 zeta | (1 - 16 zeta)^-1/2 | (a_s - c_s zeta)^-1/3"""
    z = np.asarray(zeta, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        return np.where(z >= 0.0, 1.0 + 5.0 * z, np.where(z >= ZETA_S, (1.0 - 16.0 * z) ** -0.5, (A_S - C_S * z) ** (-1.0 / 3.0)))


def _patch(kmt, km, r=3):
    """columns whose neighbourhood of radius r is full-depth ocean inside one block"""
    full = (kmt == km)
    ok = full.copy()
    for dj in range(-r, r + 1):
        for di in range(-r, r + 1):
            ok &= np.roll(np.roll(full, dj, axis=1), di, axis=2)
    ok[:, :r + 2, :] = False; ok[:, -(r + 2):, :] = False; ok[:, :, :r + 2] = False; ok[:, :, -(r + 2):] = False
    assert ok.any(), "no open-ocean patch in this configuration"
    return ok


def _two_layer(A, kstar, t1, t2, salt=35.0, u1=0.0, u2=0.0):
    """levels 1 .. kstar-1: temperature t1, velocity (u1, 0); levels kstar .. km: t2, (u2, 0); uniform salinity (model units)"""
    shp = A.get("TRACER", 1, 0).shape
    km = A.km
    lev = np.arange(1, km + 1)[None, :, None, None]
    T = np.where(lev < kstar, t1, t2) + np.zeros(shp)
    U = np.where(lev < kstar, u1, u2) + np.zeros(shp)
    for tl in range(3):
        A.set("TRACER", T, tl, 0)
        A.set("TRACER", np.full(shp, salt), tl, 1)
        A.set("UVEL", U, tl)
        A.set("VVEL", np.zeros(shp), tl)
    return T


def _rho_of_uniform_column(A, temp, salt):
    """in-situ density of (temp, salt) at the pressure of every level: the model's own `state` phase on a uniform column.
    (The equation of state is compared with the oracle level by level in the phase-parity tests; here it only supplies
    the buoyancy jump and the surface expansion coefficient the closed forms need.)"""
    shp = A.get("TRACER", 1, 0).shape
    keep = [(A.get("TRACER", tl, n), tl, n) for tl in range(3) for n in range(2)]
    for tl in range(3):
        A.set("TRACER", np.full(shp, temp), tl, 0)
        A.set("TRACER", np.full(shp, salt), tl, 1)
    A.run_phase("state")
    tl_new = 2
    rho = A.get("RHO", tl_new)
    for arr, tl, n in keep:
        A.set("TRACER", arr, tl, n)
    return rho


def _surface_buoyancy_flux(A, t1, salt, stf_t, ok):
    """Bo = -g (d rho / d T) STF_T / rho at the surface level (LMD94 (A3b), no salt flux): the derivative by central differences
    of the density the model's `state` phase returns"""
    d = 2.0 ** -6
    rp, rm, r0 = (_rho_of_uniform_column(A, t, salt)[:, 0][ok] for t in (t1 + d, t1 - d, t1))
    drdt = (rp - rm) / (2.0 * d)
    return float(np.median(-GRAV * drdt * stf_t / r0)), float(np.abs(drdt - np.median(drdt)).max())


def _kpp_setup(A, kstar, t1, t2, ustar, stf_t, u1=0.0, u2=0.0, salt=35.0):
    _prepare(A)
    for tl in range(3):
        A.set("PSURF", np.zeros(A.get("PSURF", 1).shape), tl)
    _two_layer(A, kstar, t1, t2, salt, u1, u2)
    kmt = A.geti("KMT")
    z2 = np.zeros(kmt.shape)
    for name in ("SMF", "SMFT"):                 # |tau| / rho_0 = ustar^2 (vmix_kpp.F90:2177: USTAR = sqrt(|SMFT|); the stress at U and at T points
        A.set(name, z2 + ustar * ustar, 1, 0)    # are separate forcing fields, forcing_ws.F90:307)
        A.set(name, z2, 1, 1)
    A.set("STF", z2 + stf_t, 1, 0)
    A.set("STF", z2, 1, 1)
    return kmt


def _kpp_run(A):
    A.dhdt()
    A.run_phase("vmix")
    A.run_phase("hmix_tracer")
    A.run_phase("tracer_rhs")


def check_kpp_scales_and_shape(A, regime, nu0=0.0, tol=1e-7):
    """Inside the boundary layer the diffusivity at the interface of depth d = zw(k) is K_x = h w_x(sigma) G(sigma), sigma = d / h
    (LMD94 (10)), w_x = kappa ustar / phi_x(zeta), zeta = min(sigma, eps) h kappa Bf / ustar^3 (LMD94 (13); the reference clips
    sigma at eps in every regime, vmix_kpp.F90:3085), G(sigma) = sigma (1 + a2 sigma + a3 sigma^2) with G(1) = nu(h) / (h w_x(1)),
    dG(1) = 0 for a depth-independent interior nu (LMD94 (17), (18), (D5)): a2 = -2 + 3 G(1), a3 = 1 - 2 G(1).
    Non-local term of the tracer equation (LMD94 (20)): K_s gamma_s = C* kappa (c_s kappa eps)^1/3 G(sigma) * surface flux in
    unstable forcing, none in stable forcing.  `regime`: 'stable' | 'weak' | 'strong' picks the forcing; the similarity regime
    reached (the branch of phi) is asserted.  Interior: no shear-instability term (lrich = 0), uniform background nu0."""
    km = A.km
    kstar = 9
    ustar, stf = {"stable": (0.5, 1.0e-4), "weak": (2.0, -1.0e-3), "strong": (0.6, -6.0e-3)}[regime]
    t1, t2, salt = 18.0, 8.0, 0.035
    kmt = _kpp_setup(A, kstar, t1, t2, ustar, stf, salt=salt)
    ok = _patch(kmt, km)
    bo, spread = _surface_buoyancy_flux(A, t1, salt, stf, ok)
    assert spread <= 1e-12
    if bo >= 0.0:
        bo = bo + 2.0 * KPP_EPS                             # the reference's regularisation of stable forcing, applied in the level march of
                                                            # bldepth and again after it (vmix_kpp.F90:2418, 2755; no short wave)
    u3 = ustar ** 3 + KPP_EPS                               # and of zeta (:3297)
    _kpp_run(A)
    hblt, kbl = A.get("HBLT"), A.geti("KBL")
    zw, zt, dz = A.vert("zw"), A.vert("zt"), A.vert("dz")
    h = hblt[ok]                                            # every column of the patch with its own depth
    assert np.abs(h - np.median(h)).max() <= 1e-6 * np.median(h) and (kbl[ok] == kbl[ok].flat[0]).all()
    KB = int(kbl[ok].flat[0])
    assert KB >= 5, "boundary layer too shallow for the check (KBL = %d)" % KB
    zeta_sl = EPSSFC * h * VONKAR * bo / u3
    if regime == "stable":
        assert zeta_sl.min() > 0.0
    elif regime == "weak":
        assert ZETA_M < zeta_sl.min() and zeta_sl.max() < 0.0     # both phi on their (1 - 16 zeta) branch
    else:
        assert zeta_sl.max() < ZETA_S                       # both phi on their convective (a - c zeta)^1/3 branch
    worst = 0.0
    cg = CSTAR * VONKAR * (C_S * VONKAR * EPSSFC) ** (1.0 / 3.0)
    vdc = [A.get("VDC", 1, n) for n in range(2)]
    vvc = A.get("VVC")
    src = A.get("KPP_SRC", 1, 0)
    w1 = {"m": VONKAR * ustar / phi_m(zeta_sl), "s": VONKAR * ustar / phi_s(zeta_sl)}
    nu = {"m": nu0 * A.cfg.Prandtl, "s": nu0}
    # the viscosity reaches U points as the area-weighted mean of the four surrounding T cells (grid.F90 tgrid_to_ugrid):
    # a horizontally uniform value is multiplied by the sum of the weights
    ta, ua = A.get("TAREA"), A.get("UAREA")
    t2u = (0.25 * (ta + np.roll(ta, -1, axis=2) + np.roll(ta, -1, axis=1) + np.roll(np.roll(ta, -1, axis=1), -1, axis=2)) / ua)[ok]
    g_prev = 0.0
    for k in range(1, KB - 1):                              # interfaces 1 .. KBL-2: pure boundary-layer values (LMD94 appendix D changes KBL-1)
        sig = zw[k] / h
        zeta = np.minimum(sig, EPSSFC) * h * VONKAR * bo / u3
        G = {}
        for x, phi in (("m", phi_m), ("s", phi_s)):
            wx = VONKAR * ustar / phi(zeta)
            g1 = nu[x] / (h * w1[x])
            G[x] = sig * (1.0 + sig * ((-2.0 + 3.0 * g1) + (1.0 - 2.0 * g1) * sig))
            expect = h * wx * G[x] * (t2u if x == "m" else 1.0)
            got = (vvc[:, k - 1] if x == "m" else vdc[0][:, k])[ok]          # VDC carries the levels 0 .. km+1, VVC 1 .. km
            worst = max(worst, float((np.abs(got - expect) / expect).max()))
        assert np.array_equal(vdc[0][:, k][ok], vdc[1][:, k][ok])
        # non-local source of level k: STF / dz(k) * (K gamma (top) - K gamma (bottom)), K gamma = cg G_s(sigma) when unstable
        gam = cg * G["s"] if bo < 0.0 else 0.0
        expect = stf / dz[k] * (g_prev - gam)
        got = src[:, k - 1][ok]
        if bo < 0.0:
            worst = max(worst, float(np.abs(got - expect).max() / abs(stf / dz[k] * cg)))
        else:
            assert np.abs(got).max() == 0.0
        g_prev = gam
    assert worst <= tol, "KPP boundary-layer coefficients differ from the published forms by %.3e (%s, nu0 = %g)" % (worst, regime, nu0)
    return worst


def check_kpp_hblt_two_layer(A, tol=2e-7):
    """Boundary-layer depth of a two-layer column (LMD94 (21)): the bulk Richardson number is 0 at every level of the upper
    layer and Ri* = (d - eps d / 2) db / |dV|^2 at the first level of the lower layer (d = zt(k*), db the buoyancy jump at that
    level's pressure, dV the velocity jump; no local stratification below the jump, so no unresolved-shear term).  The
    reference interpolates Ri_b(z) by the parabola through the last three levels (vmix_kpp.F90:2600-2640): with Ri_b = 0
    at the two upper ones its root is  h = zt(k*-1) + (zt(k*) - zt(k*-1)) sqrt(Ri_c / Ri*)."""
    km = A.km
    kstar, t1, t2, salt = 8, 16.0, 15.0, 0.035
    du = 14.0
    kmt = _kpp_setup(A, kstar, t1, t2, 1.0, -1.0e-4, u1=du, u2=0.0, salt=salt)
    ok = _patch(kmt, km)
    ra = _rho_of_uniform_column(A, t1, salt)[:, kstar - 1][ok]
    rb = _rho_of_uniform_column(A, t2, salt)[:, kstar - 1][ok]
    db = float(np.median(GRAV * (1.0 - ra / rb)))
    _kpp_run(A)
    zt = A.vert("zt")
    d = zt[kstar]
    ri = (d - 0.5 * EPSSFC * d) * db / (du * du)
    assert ri > RICR
    expect = zt[kstar - 1] + (zt[kstar] - zt[kstar - 1]) * np.sqrt(RICR / ri)
    hblt, kbl = A.get("HBLT")[ok], A.geti("KBL")[ok]
    err = float(np.abs(hblt - expect).max() / expect)
    assert err <= tol, "HBLT of the two-layer column differs from the closed form by %.3e (%.6f vs %.6f cm)" % (err, hblt.flat[0], expect)
    assert (kbl == kstar).all()
    return err


# ------------------------------------------------------------------------------------------------------------------
# advection of tracers (advection.F90): exact flux divergence of a linear field, monotonicity of the limited scheme
# ------------------------------------------------------------------------------------------------------------------
def _adv_setup(A, u0):
    c2dt = _prepare(A)
    _quiet(A)
    shp = A.get("UVEL", 1).shape
    for tl in range(3):
        A.set("UVEL", np.full(shp, u0), tl)
    kmt = A.geti("KMT")
    for n in range(2):
        A.set("STF", np.zeros(kmt.shape), 1, n)
    if A.cfg.vmix_choice == 3:
        for n in range(2):
            A.set("KPP_SRC", np.zeros(shp), 1, n)
    return c2dt, kmt, shp


def check_advt_linear(A, tol=1e-12):
    """Uniform zonal velocity u0 at every U point, V = 0, T = a i + b j (indices within the block), no diffusion (ah = 0):
    the east-face transport is u0 (DYU(i,j) + DYU(i,j-1)) / 2 (advection.F90:1555-1566), independent of i on the lat-lon
    grid, there is no vertical velocity, and the flux divergence is exactly  u0 a HTE_eff(j) / TAREA(j)  for the centred
    scheme (advection.F90:1667-1729) and for the third-order upwind scheme (a linear profile is interpolated exactly on a
    grid uniform in i, :2313-2676)."""
    assert A.cfg.ah == 0.0
    u0, a, b = 3.0, 2.0 ** -5, 2.0 ** -4
    c2dt, kmt, shp = _adv_setup(A, u0)
    km = A.km
    ii = np.arange(shp[-1], dtype=np.float64)[None, None, None, :]
    jj = np.arange(shp[-2], dtype=np.float64)[None, None, :, None]
    field = [a * ii + b * jj + np.zeros(shp), 0.5 * a * ii - 0.25 * b * jj + 1.0 + np.zeros(shp)]
    for n in range(2):
        for tl in range(3):
            A.set("TRACER", field[n], tl, n)
    A.dhdt()
    A.run_phase("vmix")
    A.run_phase("hmix_tracer")
    A.run_phase("tracer_rhs")
    ok = _patch(kmt, km, r=4)
    dyu, tarea = A.get("DYU"), A.get("TAREA")
    ute = u0 * 0.5 * (dyu + np.roll(dyu, 1, axis=1))
    worst = 0.0
    for n, amp in ((0, a), (1, 0.5 * a)):
        expect = -(ute * amp / tarea)
        out = A.get("TRACER", 2, n) / c2dt
        for k in range(km):
            worst = max(worst, float(np.abs(out[:, k] - expect)[ok].max() / np.abs(expect)[ok].max()))
    assert worst <= tol, "advection of a linear field differs from the exact flux divergence by %.3e" % worst
    return worst


def check_lw_lim_monotone(A):
    """One-dimensional limited Lax-Wendroff advection (advection.F90:2684-3280) of a step in i by a uniform zonal flow at
    Courant number ~0.4: the advected profile X - dt div(F) has no new extremum (stays inside [lo, hi] and monotone in i across
    the step), and what the rows lose is what the flow carried across the step: sum_i TAREA div(F) = UTE (X_right - X_left)."""
    assert A.cfg.tadvect == 3 and A.cfg.ah == 0.0
    c2dt0 = 2.0 * A.scalar("dtt")
    dxt = A.get("DXT")
    u0 = 0.4 * float(np.median(dxt)) / c2dt0
    c2dt, kmt, shp = _adv_setup(A, u0)
    km = A.km
    ok = _patch(kmt, km, r=4)
    # the longest run of patch columns in a row of block 0
    b, j = 0, int(np.argmax(ok[0].sum(axis=1)))
    cols = np.flatnonzero(ok[b, j])
    runs = np.split(cols, np.flatnonzero(np.diff(cols) > 1) + 1)
    run = max(runs, key=len)
    assert len(run) >= 12
    i0, i1 = int(run[0]), int(run[-1])
    istep = (i0 + i1) // 2
    lo, hi = 2.0, 3.0
    ii = np.arange(shp[-1])[None, None, None, :]
    step = np.where(ii <= istep, lo, hi) + np.zeros(shp)
    for n, f in ((0, step), (1, 5.0 - step)):
        for tl in range(3):
            A.set("TRACER", f, tl, n)
    A.dhdt()
    A.run_phase("vmix")
    A.run_phase("hmix_tracer")
    A.run_phase("tracer_rhs")
    dyu, tarea = A.get("DYU"), A.get("TAREA")
    ute = u0 * 0.5 * (dyu + np.roll(dyu, 1, axis=1))
    sel = slice(i0 + 2, i1 - 1)
    for n, f, jump in ((0, step, hi - lo), (1, 5.0 - step, lo - hi)):
        ft = A.get("TRACER", 2, n) / c2dt                    # = -div(F)
        for k in (0, km // 2, km - 1):
            new = (f[b, k, j] + c2dt * ft[b, k, j])[sel]
            assert new.min() >= lo - 1e-12 and new.max() <= hi + 1e-12, "limited advection created a new extremum: [%r, %r]" % (new.min(), new.max())
            d = np.diff(new) * np.sign(jump)
            assert d.min() >= -1e-12, "limited advection is not monotone across the step"
            assert np.abs(new - f[b, k, j][sel]).max() > 0.05            # the step did move
            lost = float((tarea[b, j][sel] * -ft[b, k, j][sel]).sum())
            expect = float(ute[b, j, istep] * jump)
            assert abs(lost - expect) <= 1e-11 * abs(expect), "transport across the step: %r vs %r" % (lost, expect)
    return True


# ------------------------------------------------------------------------------------------------------------------
# Gent-McWilliams / isopycnal mixing (hmix_gm.F90): a field linear in the grid indices has closed-form fluxes
# ------------------------------------------------------------------------------------------------------------------
def check_gm_linear(A, tol=2e-12):
    """T = T0 + a i + c j + b k (b < 0: stably stratified; uniform salinity; at rest), constant kappa, well below the boundary layer
    and inside the slope limits, open ocean away from coasts, on the lat-lon grid (metrics uniform in i, varying with j).
    Then the isopycnal slopes are Sx = a / (-b), Sy = c / (-b) everywhere (the expansion coefficient cancels), every taper is 1 and,
    with HYX = HTE / HUS, HXY = HTN / HUW, _w / _s the west / south neighbour's value, kappa_b = ah_bolus:
      * the east-face fluxes are equal at i and i - 1; the north-face flux is FY = dz HXY c kappa_b: dz (HXY / 4) c (4 kappa) from
        :1827-1828 (the eight-term sum holds four KAPPA_ISOP and four HOR_DIFF = 0), minus, without cancellation, the skew terms
        (HXY / 4) 4 (kappa - kappa_b) Sy dz (-b) of :1870-1896; so the horizontal part of the tendency is kappa_b c (HXY - HXY_s) / TAREA;
      * the flux through the bottom face is fz(k) = - (kappa + kappa_b) / 4 Q (dz(k) + dz(k+1)) with
        Q = Sx a (HYX + HYX_w) + Sy c (HXY + HXY_s) (:1923-2050), so the vertical part is
        (kappa + kappa_b) / 4 Q (dz(k+1) - dz(k-1)) / (dz(k) TAREA);
      * the isopycnal part added to the vertical diffusivity at the bottom of level k (:1725-1748) is
        kappa dzw(k)^2 [Sx^2 (HYX + HYX_w) + Sy^2 (HXY + HXY_s)] / (2 TAREA)  (= kappa times the squared true slope, as it should be).
    The second tracer, uniform, has no tendency at all."""
    assert A.cfg.hmix_tracer == 3 and A.cfg.vmix_choice == 1 and A.cfg.gm_transition_layer == 0
    c2dt, kmt, shp = _adv_setup(A, 0.0)
    km = A.km
    a, b, c, t0 = 2.0 ** -5, -(2.0 ** -2), 2.0 ** -6, 20.0
    ii = np.arange(shp[-1], dtype=np.float64)[None, None, None, :]
    jj = np.arange(shp[-2], dtype=np.float64)[None, None, :, None]
    kk = np.arange(km, dtype=np.float64)[None, :, None, None]
    field = [t0 + a * ii + c * jj + b * kk + np.zeros(shp), 0.035 + np.zeros(shp)]
    for n in range(2):
        for tl in range(3):
            A.set("TRACER", field[n], tl, n)
    A.dhdt()
    A.run_phase("vmix")
    A.run_phase("hmix_tracer")
    A.run_phase("tracer_rhs")
    ok = _patch(kmt, km, r=4)
    kappa, Sx, Sy = A.cfg.ah, a / (-b), c / (-b)
    kappa_b = A.cfg.ah_bolus if A.cfg.ah_bolus != 0.0 else kappa
    dz, dzw = A.vert("dz"), A.vert("dzw")
    hyx, hxy = A.get("HTE") / A.get("HUS"), A.get("HTN") / A.get("HUW")
    hyx_w, hxy_s = np.roll(hyx, 1, axis=2), np.roll(hxy, 1, axis=1)
    tarea = A.get("TAREA")
    Q = Sx * a * (hyx + hyx_w) + Sy * c * (hxy + hxy_s)
    assert np.abs(hxy - hxy_s)[ok].max() > 1e-4 * hxy[ok].max()                           # ... and varying with j: the horizontal part is there
    assert np.abs(hyx - np.roll(hyx, 1, axis=2))[ok].max() <= 1e-14 * hyx[ok].max()      # the premise: metrics uniform in i
    vdc_all = A.get("VDC", 1, 0)
    vdc = vdc_all - A.cfg.const_vdc
    gt = [A.get("TRACER", 2, n) / c2dt for n in range(2)]
    dzr, dzwr = 1.0 / dz[1:km + 1], A.vert("dzwr")
    worst_g = worst_v = 0.0
    levels = list(range(4, km - 1))
    nsig = 0
    for k in levels:
        exp_v = kappa * dzw[k] * dzw[k] * (Sx * Sx * (hyx + hyx_w) + Sy * Sy * (hxy + hxy_s)) / (2.0 * tarea)
        worst_v = max(worst_v, float(np.abs(vdc[:, k] - exp_v)[ok].max() / np.abs(exp_v)[ok].max()))
        # the right-hand side also holds the explicit vertical diffusion of the old tracer with the (now larger) diffusivity
        # (vdifft, vertical_mix.F90:795-806): VDTK(k) = (VDC(k-1) (T(k-1) - T(k)) dzwr(k-1) - VDC(k) (T(k) - T(k+1)) dzwr(k)) dzr(k)
        vdtk = (vdc_all[:, k - 1] * (-b) * dzwr[k - 1] - vdc_all[:, k] * (-b) * dzwr[k]) * dzr[k - 1]
        exp_g = kappa_b * c * (hxy - hxy_s) / tarea + 0.25 * (kappa + kappa_b) * Q * (dz[k + 1] - dz[k - 1]) / (dz[k] * tarea)
        rem = gt[0][:, k - 1] - vdtk
        floor = 4.0e-16 * np.abs(vdtk)[ok].max()                                           # what the subtraction leaves of rounding
        if np.abs(exp_g)[ok].max() > 1.0e4 * floor:
            nsig += 1
            worst_g = max(worst_g, float((np.abs(rem - exp_g)[ok].max() - floor) / np.abs(exp_g)[ok].max()))
        assert np.abs(gt[1][:, k - 1])[ok].max() == 0.0
    assert nsig >= 4
    assert worst_g <= 1.0e-9, "GM tendency of a linear field differs from the closed form by %.3e" % worst_g     # what is left after the subtraction
    assert worst_v <= tol, "isopycnal part of VDC differs from its closed form by %.3e" % worst_v
    return worst_g, worst_v


def _taper_slope(control, x):
    """the slope tapers of hmix_gm.F90:1480-1594 as functions of x = SLA / slm, written from the papers the reference cites:
    'tanh' Danabasoglu & McWilliams (1995): 1/2 (1 - tanh(10 x - 4)) below x = 1, 0 above; 'notanh', its piecewise-parabolic
    stand-in: 1 up to x = 0.2, 0 from x = 0.6, in between 1/2 (1 - (2.5 x - 1)(4 - |10 x - 4|)); 'Gerd' (Gerdes et al. 1991): 1 up to
    x = 1, x^-2 above; 'clip': no taper (the slope itself is limited)."""
    x = np.asarray(x, dtype=np.float64)
    if control == 1:
        return np.where(x < 1.0, 0.5 * (1.0 - np.tanh(10.0 * x - 4.0)), 0.0)
    if control == 0:
        mid = 0.5 * (1.0 - (2.5 * x - 1.0) * (4.0 - np.abs(10.0 * x - 4.0)))
        return np.where(x <= 0.2, 1.0, np.where(x >= 0.6, 0.0, mid))
    if control == 3:
        return np.where(x > 1.0, 1.0 / (x * x), 1.0)
    return np.ones_like(x)


def check_gm_tapers(A, tol=5e-12):
    """The slope tapers of Gent-McWilliams mixing on a CONSTRUCTED slope field (until round 4 they rested on two
    restatements by one author).  T = T0 + a i + b k at rest with uniform salinity gives the isopycnal slope Sx = a / (-b) in index
    units at every quarter cell, hence the true slope magnitude of hmix_gm.F90:1431-1436

        SLA(j, k) = dzw(k) |Sx| / DXT(j) + eps

    at both half cells next to the interface below level k -- a function of latitude and depth that sweeps the break points of every
    taper.  Below the second level the near-surface taper is 1 (no KPP: the boundary layer is the first level), so with
    tau_r = taper(SLA / slm_r), tau_b = taper(SLA / slm_b) (`diff_tapering`; tau_b = tau_r when the limits are equal):

      * the isopycnal part added to the vertical diffusivity (:1725-1748) is tau_r kappa dzw(k)^2 Sx^2 (HYX + HYX_w) / (2 TAREA);
      * the stream function of the thickness diffusion at that interface, east face (:1684-1690, 2091-2097), is
        UIB(k) = tau_b kappa_b Sx HYX dzw(k), so the eddy-induced velocity of diag_gm_bolus (:2112) is
        U_ISOP(k) = (UIB(k-1) - UIB(k)) / (dz(k) HTE);
      * 'clip' limits the slope itself instead: |Sx| -> min(|Sx|, slm_r HUS / dzw(k)) (:1541-1573), both formulas with tau = 1.

    Exercised on the oracle and on the device in all four slope_control choices, with equal and with different slope limits."""
    cfg = A.cfg
    assert cfg.hmix_tracer == 3 and cfg.vmix_choice == 1 and cfg.gm_transition_layer == 0 and cfg.gm_kappa_type == 0 and cfg.gm_diag_bolus == 1
    c2dt, kmt, shp = _adv_setup(A, 0.0)
    km = A.km
    a, b, t0 = 0.5, -(2.0 ** (-12 if cfg.gm_slope_control == 3 else -10)), 2.0      # (Gerdes' taper only acts above the slope limit: steeper)
    ii = np.arange(shp[-1], dtype=np.float64)[None, None, None, :]
    kk = np.arange(km, dtype=np.float64)[None, :, None, None]
    field = [t0 + a * ii + b * kk + np.zeros(shp), 0.035 + np.zeros(shp)]
    for n in range(2):
        for tl in range(3):
            A.set("TRACER", field[n], tl, n)
    A.dhdt()
    A.run_phase("vmix")
    A.run_phase("hmix_tracer")
    A.run_phase("tracer_rhs")
    ok = _patch(kmt, km, r=4)
    control = cfg.gm_slope_control
    slm_r = cfg.slm_r if cfg.slm_r != 0.0 else 0.3
    slm_b = cfg.slm_b if cfg.slm_b != 0.0 else 0.3
    kappa = cfg.ah
    kappa_b = cfg.ah_bolus if cfg.ah_bolus != 0.0 else kappa
    dz, dzw = A.vert("dz"), A.vert("dzw")
    hyx = A.get("HTE") / A.get("HUS")
    hyx_w = np.roll(hyx, 1, axis=2)
    tarea, dxt, hus, hte = A.get("TAREA"), A.get("DXT"), A.get("HUS"), A.get("HTE")
    Sx = a / (-b)
    vdc = A.get("VDC", 1, 0) - cfg.const_vdc
    uis = A.get("UISOP")

    def sx_eff(k):       # |slope| in index units after 'clip'
        return np.minimum(Sx, slm_r * hus / dzw[k]) if control == 2 else Sx + 0.0 * hus

    def sla(k):
        return dzw[k] * Sx / dxt + 1.0e-10

    seen_r, seen_b = [], []
    worst_v = worst_u = 0.0
    uib_prev, sign = None, None
    for k in range(3, km - 1):
        tr, tb = _taper_slope(control, sla(k) / slm_r), _taper_slope(control, sla(k) / slm_b)
        se = sx_eff(k)
        exp_v = tr * kappa * dzw[k] * dzw[k] * se * se * (hyx + hyx_w) / (2.0 * tarea)
        scale = (kappa * dzw[k] * dzw[k] * se * se * (hyx + hyx_w) / (2.0 * tarea))[ok].max()
        worst_v = max(worst_v, float(np.abs(vdc[:, k] - exp_v)[ok].max() / scale))
        uib = tb * kappa_b * se * hyx * dzw[k]
        if uib_prev is not None:
            exp_u = (uib_prev - uib) / (dz[k] * hte)
            got = uis[:, k - 1]
            big = np.abs(exp_u) > 1.0e-3 * np.abs(exp_u)[ok].max()
            if sign is None:     # the orientation of the slope (sign of SLX for dT/dx > 0 in stable water) is one sign for the whole field
                sel = ok & big
                sign = 1.0 if float((got[sel] * exp_u[sel]).sum()) > 0.0 else -1.0
            uscale = max(np.abs(uib_prev / (dz[k] * hte))[ok].max(), 1e-300)       # the two terms cancel partly: error relative to one of them
            worst_u = max(worst_u, float(np.abs(got - sign * exp_u)[ok].max() / uscale))
        uib_prev = uib
        seen_r.append((sla(k) / slm_r)[ok]); seen_b.append((sla(k) / slm_b)[ok])
    xr, xb = np.concatenate(seen_r), np.concatenate(seen_b)
    # the constructed field must actually visit every branch of the function under test
    if control == 0:
        for x in (xr, xb):
            assert (x < 0.2).any() and ((x > 0.25) & (x < 0.55)).any() and (x > 0.6).any(), "slopes miss a branch of the 'notanh' taper"
    elif control == 1:
        for x in (xr, xb):
            assert (x < 0.3).any() and ((x > 0.3) & (x < 0.6)).any() and (x > 1.0).any(), "slopes miss a branch of the 'tanh' taper"
    elif control == 3:
        for x in (xr, xb):
            assert (x < 1.0).any() and (x > 1.5).any(), "slopes miss a branch of Gerdes' taper"
    else:
        clipped = np.concatenate([(slm_r * hus / dzw[k] < Sx)[ok] for k in range(3, km - 1)])
        assert clipped.any() and (~clipped).any(), "slopes miss a branch of the clipping"
    assert worst_v <= tol, "tapered isopycnal part of VDC differs from its closed form by %.3e" % worst_v
    assert worst_u <= tol, "eddy-induced velocity differs from the tapered stream function's by %.3e" % worst_u
    return worst_v, worst_u


def check_gm_transition_layer(A, tol=2e-11):
    """The transition layer of Gent-McWilliams mixing (hmix_gm.F90:3183-3440 transition_layer, :3441-3743 merged_streamfunction) on the
    constructed slope field of check_gm_tapers (T = T0 + a i + b k at rest, no KPP, so the diabatic depth D is zw(1)), against what the
    two routines are supposed to deliver -- stated independently of their 250-line state machine and interpolation code:

      * INTERIOR_DEPTH I is the deepest of the grid depths zt(2), zw(2), zt(3), zw(3), ... that can be reached from zt(2) one by one
        while the isopycnal displacement over one deformation radius still reaches the diabatic layer, D >= d - R |S|(d), with
        |S| = SLA at that depth and R the Rossby radius bounded to [15, 100] km; THICKNESS = I - D.  (Closed form of SLA as in
        check_gm_tapers; the walk is the statement of Danabasoglu et al. 2008, eq. 6.)
      * the merged stream function psi(z) sampled at the half-cell centres z = zt(k) -+ dz(k) / 4 is LINEAR through the origin above D;
        in the layer it leaves that line QUADRATICALLY, psi = line - c (z - D)^2 with ONE c per column; extended to I it meets the
        first interior sample (continuity) and its slope there is the one-sided interior difference of smaller magnitude.
    Only the library's own outputs enter the second part (GM_SF_SLX, TLT_*)."""
    cfg = A.cfg
    assert cfg.hmix_tracer == 3 and cfg.vmix_choice == 1 and cfg.gm_transition_layer == 1 and cfg.gm_kappa_type == 0
    c2dt, kmt, shp = _adv_setup(A, 0.0)
    km = A.km
    a, b, t0 = 0.5, -(2.0 ** -5), 2.0          # R |S| of the order of the level depths at low latitudes, well below them at high ones
    ii = np.arange(shp[-1], dtype=np.float64)[None, None, None, :]
    kk = np.arange(km, dtype=np.float64)[None, :, None, None]
    field = [t0 + a * ii + b * kk + np.zeros(shp), 0.035 + np.zeros(shp)]
    for n in range(2):
        for tl in range(3):
            A.set("TRACER", field[n], tl, n)
    A.dhdt()
    A.run_phase("vmix")
    A.run_phase("hmix_tracer")
    A.run_phase("tracer_rhs")
    ok = _patch(kmt, km, r=4)
    dz, dzw, dzwr, zt, zw = A.vert("dz"), A.vert("dzw"), A.vert("dzwr"), A.vert("zt"), A.vert("zw")
    dxt, fcort = A.get("DXT"), A.get("FCORT")
    Sx = a / (-b)
    rb = 1.0 / np.maximum(np.minimum(np.abs(fcort) / 200.0, 1.0 / 1.5e6), 1.0e-7)
    D, TH, ID = A.get("TLT_DIABATIC_DEPTH"), A.get("TLT_THICKNESS"), A.get("TLT_INTERIOR_DEPTH")
    assert np.all(D[ok] == zw[1])

    def sla(k):
        return dzw[k] * Sx / dxt + 1.0e-10

    # ---- the walk
    exp_i = np.full(D.shape, zt[2])
    going = np.ones(D.shape, dtype=bool)
    cands = [(zw[2], 2)] + [c for k in range(3, km - 1) for c in ((zt[k], k), (zw[k], k))]
    for d, k in cands:
        reach = going & (D >= d - rb * sla(k))
        exp_i = np.where(reach, d, exp_i)
        going = reach
    assert np.array_equal(ID[ok], exp_i[ok]), "INTERIOR_DEPTH differs from the walk in %d columns" % int((ID != exp_i)[ok].sum())
    assert np.array_equal(TH[ok], (exp_i - D)[ok])
    assert len(np.unique(ID[ok])) >= 4, "the constructed slopes give a trivial transition layer"
    assert not going[ok].any()
    # ---- the merged stream function, east face
    top, bot = A.get("GM_SF_SLX", 1, 0), A.get("GM_SF_SLX", 1, 1)              # (nblocks, km, ny, nx): levels 1 .. km
    zs = np.empty(2 * km); ps = np.empty((2 * km,) + D.shape)
    for k in range(1, km + 1):
        zs[2 * (k - 1)] = zt[k] - 0.25 * dz[k]; zs[2 * (k - 1) + 1] = zt[k] + 0.25 * dz[k]
        ps[2 * (k - 1)] = top[:, k - 1]; ps[2 * (k - 1) + 1] = bot[:, k - 1]
    assert zs[1] <= zw[1] < zs[2]
    slope = ps[0] / zs[0]
    scale = np.abs(ps[:, ok]).max()
    assert scale > 0.0
    worst = {"linear": float(np.abs(ps[1] - slope * zs[1])[ok].max() / scale)}
    nb, ny, nx = D.shape
    w_quad = w_cont = w_slope = 0.0
    n_quad = 0
    for bi, j, i in zip(*np.nonzero(ok)):
        d0, i0 = D[bi, j, i], ID[bi, j, i]
        p = ps[:, bi, j, i]
        inl = np.nonzero((zs > d0) & (zs <= i0))[0]
        first_int = int(np.nonzero(zs > i0)[0][0])
        if len(inl) == 0:
            continue
        cq = -(p[inl] - slope[bi, j, i] * zs[inl]) / (zs[inl] - d0) ** 2
        c0 = cq[-1]                                                      # the deepest sample carries the most signal
        if len(inl) >= 2:
            n_quad += 1
            w_quad = max(w_quad, float(np.abs((cq - c0) * (zs[inl] - d0) ** 2).max() / scale))
        T = i0 - d0
        w_cont = max(w_cont, abs(slope[bi, j, i] * i0 - c0 * T * T - p[first_int]) / scale)
        # one-sided interior differences at I: between the first and the second interior sample, and between the second and the third
        k1, h1 = first_int // 2 + 1, first_int % 2
        if h1 == 1:      # I = zt(k1): psi_I = SF(kbt, k1)
            e1 = 2.0 * dzwr[k1] * (p[first_int] - p[first_int + 1])
            e2 = 2.0 * (p[first_int + 1] - p[first_int + 2]) / dz[k1 + 1]
        else:            # I = zw(k1 - 1): psi_I = SF(ktp, k1)
            e1 = 2.0 * (p[first_int] - p[first_int + 1]) / dz[k1]
            e2 = 2.0 * dzwr[k1] * (p[first_int + 1] - p[first_int + 2])
        dpsi = e2 if abs(e2) < abs(e1) else e1
        w_slope = max(w_slope, abs((slope[bi, j, i] - 2.0 * c0 * T) + dpsi) / (scale / zw[2]))
    worst.update(quadratic=w_quad, continuity=w_cont, slope=w_slope)
    assert n_quad >= 20, "too few columns with two samples inside the transition layer"
    for what, v in worst.items():
        assert v <= tol, "merged stream function: %s violated by %.3e" % (what, v)
    return worst


def _rho_of_field(A, temp, salt):
    """in-situ density of an arbitrary (temp, salt) field: the model's own `state` phase (see _rho_of_uniform_column)"""
    keep = [(A.get("TRACER", tl, n), tl, n) for tl in range(3) for n in range(2)]
    for tl in range(3):
        A.set("TRACER", temp, tl, 0)
        A.set("TRACER", salt, tl, 1)
    A.run_phase("state")
    rho = A.get("RHO", 2)
    for arr, tl, n in keep:
        A.set("TRACER", arr, tl, n)
    return rho


def check_gm_bfre_profile(A, tol=2e-6):
    """kappa type 'bfre' (buoyancy_frequency_dependent_profile, hmix_gm.F90:3011-3180): below the surface diabatic layer the
    diffusivities are scaled by KAPPA_VERTICAL(k) = N^2(k-1) / N^2(K_MIN) bounded to [0.1, 1], N^2 at the interface below level k
    = -g (d rho / d T)(T_k, S, p_{k+1}) (T_k - T_{k+1}) / dzw(k) (uniform salinity), K_MIN the first interface below the layer.

    Constructed column: T = T(k) + a i with three linear pieces in k -- moderate, four times steeper (N^2 above the reference value:
    upper bound), then sixteen times weaker (N^2 below a tenth of it: lower bound).  The expansion coefficient comes from CENTRAL
    DIFFERENCES of the model's own density (state phase on columns of uniform temperature T_k +- delta read at level k + 1), not from
    the routine under test.  Both half cells next to an interface have the same slope a / (T_k - T_{k+1}) and the same taper, so the
    isopycnal part added to VDC there is the constant-kappa closed form of check_gm_tapers times the thickness-weighted mean
    (dz(k) KV(k) + dz(k+1) KV(k+1)) / (dz(k) + dz(k+1))."""
    cfg = A.cfg
    assert cfg.hmix_tracer == 3 and cfg.vmix_choice == 1 and cfg.gm_transition_layer == 0 and cfg.gm_kappa_type == 1 and cfg.gm_slope_control == 0
    c2dt, kmt, shp = _adv_setup(A, 0.0)
    km = A.km
    nb, _, ny, nx = shp
    a, t0, salt = 2.0 ** -6, 24.0, 0.035
    b1 = -(2.0 ** -4)
    k1, k2 = 4, 9                                               # (0-based levels) the steep piece starts at k1, the weak one at k2
    tk = np.empty(km)
    tk[0] = t0
    for k in range(1, km):
        tk[k] = tk[k - 1] + (b1 if k <= k1 else 4.0 * b1 if k <= k2 else b1 / 16.0)
    ii = np.arange(nx, dtype=np.float64)[None, None, None, :]
    T = tk[None, :, None, None] + a * ii + np.zeros(shp)
    S = salt + np.zeros(shp)
    for tl in range(3):
        A.set("TRACER", T, tl, 0); A.set("TRACER", S, tl, 1)
    A.dhdt()
    A.run_phase("vmix")
    A.run_phase("hmix_tracer")
    A.run_phase("tracer_rhs")
    vdc = A.get("VDC", 1, 0) - cfg.const_vdc
    ok = _patch(kmt, km, r=4)
    dz, dzw = A.vert("dz"), A.vert("dzw")
    # ---- N^2 at the interfaces k = 1 .. km-1 (1-based), expansion coefficient by central differences of the model's density
    d = 2.0 ** -6
    n2 = np.zeros((km + 1,) + (nb, ny, nx))
    for k in range(1, km):                                      # interface below level k; T_k = T[:, k-1], density read at level k+1 = index k
        col = np.broadcast_to(T[:, k - 1:k], shp)
        rp, rm = _rho_of_field(A, col + d, S)[:, k], _rho_of_field(A, col - d, S)[:, k]
        n2[k] = np.maximum(0.0, -GRAV / dzw[k] * ((rp - rm) / (2.0 * d)) * (T[:, k - 1] - T[:, k]))
    kmin = 2                                                    # zw(2) is the first interface depth below the diabatic layer zw(1); N^2 > 0 there
    assert np.all(n2[kmin][ok] > 0.0)
    norm = np.ones_like(n2)
    for k in range(kmin, km):
        norm[k] = np.minimum(np.maximum(n2[k] / n2[kmin], 0.1), 1.0)
    norm[km] = norm[km - 1]
    kv = np.ones_like(n2)                                       # KAPPA_VERTICAL(k), 1-based
    for k in range(kmin + 1, km + 1):
        kv[k] = norm[k - 1]
    assert (kv[:, ok] == 0.1).any() and ((kv[:, ok] > 0.1) & (kv[:, ok] < 1.0)).any() and (kv[kmin + 2:, ok] == 1.0).any(), "the column misses a branch of the bounds"
    # ---- the isopycnal part of VDC
    kappa = cfg.ah
    slm_r = cfg.slm_r if cfg.slm_r != 0.0 else 0.3
    hyx = A.get("HTE") / A.get("HUS")
    hyx_w = np.roll(hyx, 1, axis=2)
    tarea, dxt = A.get("TAREA"), A.get("DXT")
    worst = 0.0
    for k in range(3, km - 1):
        sx = a / (tk[k - 1] - tk[k])
        tap = _taper_slope(0, (dzw[k] * sx / dxt + 1.0e-10) / slm_r)
        base = tap * kappa * dzw[k] * dzw[k] * sx * sx * (hyx + hyx_w) / (2.0 * tarea)
        mean = (dz[k] * kv[k] + dz[k + 1] * kv[k + 1]) / (dz[k] + dz[k + 1])
        worst = max(worst, float((np.abs(vdc[:, k] - base * mean) / np.abs(base))[ok].max()))
    assert worst <= tol, "isopycnal part of VDC with the N^2 profile differs from its closed form by %.3e" % worst
    return worst


def pbc_flat_grid(cfg, kstar, frac):
    """caller grid for check_kpp_hblt_two_layer_pbc: the synthetic lat-lon grid with a FLAT bottom at level kstar whose bottom cell has the
    same partial thickness frac * dz(kstar) everywhere (a record of bottom_cell_file), so that the boundary-layer depth is the same in
    every column and the horizontal filter of smooth_hblt leaves it alone"""
    from popcfg import synthetic_grid
    from orclib import Oracle
    small = type(cfg).from_buffer_copy(bytes(cfg))
    small.partial_bottom_cells = 0
    o = Oracle(small)
    dz = o.v1("dz").copy()
    o.close()
    g = synthetic_grid(cfg, stepped=False)
    g["KMT"] = np.where(g["KMT"] > 0, kstar, 0).astype(np.int32)
    g["DZBC"] = np.where(g["KMT"] > 0, frac * dz[kstar], 0.0)
    return g


def check_kpp_hblt_two_layer_pbc(A, kstar, frac, tol=2e-7):
    """check_kpp_hblt_two_layer with the velocity and density jump at the PARTIAL BOTTOM CELL itself (KBL = KMT = kstar, bottom thickness
    frac * dz; grid of pbc_flat_grid).  With partial bottom cells the reference does not use LMD94's (d - eps d / 2) db / |dV|^2 but the
    gradient form between the first T point and T point kl (vmix_kpp.F90:2359-2366, 2561-2575):
        Ri* = (db / h_T) / (|dV|^2 / h_U^2),   h_T = zt(k*-1) + (dz(k*-1) + DZT(k*) - dz(1)) / 2   (= h_U: the bottom thickness is uniform)
    and the T point of the bottom cell sits at ZKL = zt(k*-1) + (dz(k*-1) + DZT(k*)) / 2 (:2212-2220), so the root of the parabola is
        h = zt(k*-1) + (ZKL - zt(k*-1)) sqrt(Ri_c / Ri*).
    Also the non-local source (:1296-1302), which reaches the bottom cell here: its thickness-weighted column sum vanishes (the transport only
    redistributes) WITH the partial thickness -- with dz(k*) in its place it is off by the per cent the cell is thinner."""
    km = A.km
    t1, t2, salt, du = 16.0, 15.99, 0.035, 24.0        # a weak jump: Ri* ~ 0.45, so the boundary layer ends INSIDE the bottom cell, below its top face
    kmt = _kpp_setup(A, kstar, t1, t2, 1.0, -1.0e-4, u1=du, u2=0.0, salt=salt)
    # A shear of 1e-3 du across the upper layer (level 1 keeps du, so |dV|^2 at the jump is unchanged).  With NO shear above the jump the
    # bulk Richardson number there is (roundoff of the density difference of equal water) / (the eps regularisation of the denominator)
    # -- noise of either sign and of any size up to O(0.1), which the parabola takes its slope at z(k*-1) from (the reference's formula
    # has the same property; seen at km = 62, k* = 30: -0.053 at level 28 moved HBLT by 2e-4).  With it that number is O(1e-8).
    shp = A.get("UVEL", 1).shape
    lev = np.arange(1, km + 1)[None, :, None, None]
    U = np.where(lev < kstar, du * (1.0 + 1.0e-3 * (lev - 1.0) / kstar), 0.0) + np.zeros(shp)
    for tl in range(3):
        A.set("UVEL", U, tl)
    wet = kmt == kstar
    ok = wet.copy()
    for dj in range(-3, 4):
        for di in range(-3, 4):
            ok &= np.roll(np.roll(wet, dj, axis=1), di, axis=2)
    ok[:, :5, :] = False; ok[:, -5:, :] = False; ok[:, :, :5] = False; ok[:, :, -5:] = False
    assert ok.sum() > 50
    dz, zt = A.vert("dz"), A.vert("zt")
    dzbc = A.get("DZBC")
    assert np.all(np.abs(dzbc[ok] - frac * dz[kstar]) <= 1e-9 * dz[kstar])
    ra = _rho_of_uniform_column(A, t1, salt)[:, kstar - 1][ok]
    rb = _rho_of_uniform_column(A, t2, salt)[:, kstar - 1][ok]
    db = float(np.median(GRAV * (1.0 - ra / rb)))
    _kpp_run(A)
    dzb = frac * dz[kstar]
    h_t = zt[kstar - 1] + 0.5 * (dz[kstar - 1] + dzb - dz[1])
    zkl = zt[kstar - 1] + 0.5 * (dz[kstar - 1] + dzb)
    # the reference velocity is the mean over the surface layer eps zt(k*) (vmix_kpp.F90:2324-2349; full-cell depths also with partial cells)
    zw = np.concatenate([[0.0], np.cumsum(dz[1:])])
    surf = EPSSFC * zt[kstar]
    kref = next(k for k in range(1, kstar + 1) if zw[k] >= surf)
    ulev = U[0, :, 0, 0]
    uref = (ulev[kref - 1] * (surf - zw[kref - 1]) + sum(dz[k] * ulev[k - 1] for k in range(1, kref))) / surf
    ri = (db / h_t) / (uref * uref / (h_t * h_t))
    assert ri > RICR
    expect = zt[kstar - 1] + (zkl - zt[kstar - 1]) * np.sqrt(RICR / ri)
    hblt, kbl = A.get("HBLT")[ok], A.geti("KBL")[ok]
    err = float(np.abs(hblt - expect).max() / expect)
    assert (kbl == kstar).all()
    assert err <= tol, "HBLT with the jump at a partial bottom cell differs from the closed form by %.3e (%.6f vs %.6f cm)" % (err, hblt.flat[0], expect)
    full = zt[kstar - 1] + (zt[kstar] - zt[kstar - 1]) * np.sqrt(RICR / ((zt[kstar] - 0.5 * EPSSFC * zt[kstar]) * db / (du * du)))
    assert abs(full - expect) > 1.0e-3 * expect, "the case does not distinguish the partial-cell form from the full-cell one"
    # non-local source: sum_k SRC(k) DZT(k) = 0 with DZT(k*) = the partial thickness
    src = A.get("KPP_SRC", 1, 0)
    thick = np.array([dz[k] for k in range(1, kstar)] + [dzb])
    col = np.stack([src[:, k][ok] for k in range(kstar)])          # (kstar, ncol)
    s1 = np.abs((col * thick[:, None]).sum(axis=0)).max()
    s0 = (np.abs(col) * thick[:, None]).sum(axis=0).max()
    wrong = np.abs((col * np.array([dz[k] for k in range(1, kstar + 1)])[:, None]).sum(axis=0)).max()
    assert s0 > 0.0 and np.abs(col[kstar - 1]).max() > 0.0, "the non-local source does not reach the bottom cell"
    assert s1 <= 1e-13 * s0, "thickness-weighted non-local source does not sum to zero: %.3e" % (s1 / s0)
    assert wrong > 1e-4 * s0
    return err, s1 / s0
