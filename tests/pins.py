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
