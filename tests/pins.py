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
THREE_D = ("TRACER", "UVEL", "VVEL", "RHO", "VVC", "KPP_SRC")


class OracleAdapter:
    def __init__(self, cfg):
        from orclib import Oracle
        self.o = Oracle(cfg)
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

    def __init__(self, pkg, cfg):
        from orclib import Oracle
        self.m = pkg.PopModel(cfg)
        self.cfg = cfg
        self.km, self.nblocks = self.m.km, self.m.nblocks
        o = Oracle(cfg)
        self._vert = {k: o.v1(k).copy() for k in ("dz", "dzw", "dzwr", "bouss", "afac_t", "afac_u")}
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
