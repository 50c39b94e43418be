"""Distribution of the KPP boundary-layer depth in the tx0.1v3 bench state: how many levels does the bulk-Richardson march of
bldepth need?  (Decides whether evaluating the surface-layer buoyancy difference on demand pays.)"""
import os, sys, importlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from popcfg import named_config
pkg = importlib.import_module("pop2-cesm_amd")
name, nsteps = sys.argv[1], int(sys.argv[2])
cfg = named_config(name)
m = pkg.PopModel(cfg)
def vert_dz(km, zmax=5500.0, dz_sfc=25.0, dz_deep=400.0, eps=1.0e-10):     # host_setup.cpp make_vertical (vert_grid compute_dz)
    def profile(zl):
        dz, depth = [], 0.0
        for _ in range(km):
            r = depth / zl
            dz.append(dz_deep - (dz_deep - dz_sfc) * np.exp(-(r * r))); depth += dz[-1]
        return depth, dz
    zl0, zl1 = eps, zmax
    d0, d1 = profile(zl0)[0], profile(zl1)[0]
    while (zl1 - zl0) / zmax > eps:
        zl = zl0 + 0.5 * (zl1 - zl0)
        d, dz = profile(zl)
        if (d0 - zmax) * (d - zmax) < 0.0: d1, zl1 = d, zl
        else: d0, zl0 = d, zl
    return np.array(dz) * 100.0
zw = np.cumsum(vert_dz(cfg.km))                       # bottom depth of level k (cm)
for s in range(1, nsteps + 1):
    m.step()
    if s in (2, nsteps // 2, nsteps):
        H = m.get("HBLT")
        kmt = m.geti("KMT")
        oc = kmt > 0
        kbl = np.searchsorted(zw, H[oc]) + 1
        q = np.percentile(kbl, [50, 90, 99, 100])
        print(name, "step", s, "HBLT m: median %.1f p90 %.1f max %.1f | level of HBLT: median %d p90 %d p99 %d max %d | mean %.1f of km %d" %
              (np.median(H[oc]) / 100, np.percentile(H[oc], 90) / 100, H[oc].max() / 100, q[0], q[1], q[2], q[3], kbl.mean(), len(zw)), flush=True)
        # per 64-column run (one wave): the deepest column decides
        kfull = np.zeros(H.shape, dtype=np.int64); kfull[oc] = kbl
        flat = kfull.reshape(-1)
        n = (flat.size // 64) * 64
        wmax = flat[:n].reshape(-1, 64).max(axis=1)
        wmax = wmax[wmax > 0]
        print("   per 64-column run: mean of max level %.1f, median %d, p90 %d" % (wmax.mean(), np.median(wmax), np.percentile(wmax, 90)), flush=True)
m.close()
