"""Generates the committed golden vectors with the CPU oracle (oracle/libpop_oracle.so).

The reference holds no fixtures for advection / hmix / vmix / the solver, so these
vectors pin the path against the CPU restatement of the reference algorithm: inputs are fully
determined by the configuration (synthetic grid, Levitus profile + analytic perturbation, analytic
wind) plus the surface fluxes set in prepare(); outputs are the prognostic fields after N steps.

    python tests/golden/make_golden.py        # rewrites tests/golden/golden_*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from popcfg import named_config  # noqa: E402

FIELDS = [("TRACER", True), ("UVEL", True), ("VVEL", True), ("RHO", True), ("PSURF", False), ("UBTROP", False)]
NSTEPS = {"const": 4, "kpp_del4": 4, "upwind3": 5, "robert": 6, "pcsi_evp": 4, "lw_lim": 5, "pbc_kpp_del4": 5, "padded": 5, "gm": 5, "gm_tlt": 5}


def config(case):
    small = dict(nx_global=24, ny_global=20, km=16, block_size_x=12, block_size_y=10)
    if case == "const":
        return named_config("tiny", **small)
    if case == "kpp_del4":
        return named_config("tiny", vmix_choice=3, ldbl_diff=1, hmix_momentum=4, hmix_tracer=4, lvariable_hmix=1,
                            am=-1.0e23, ah=-1.0e22, solver_choice=2, **small)
    if case == "pbc_kpp_del4":   # partial bottom cells on stepped bathymetry with the tx0.1v3 physics (del4 + variable mixing + KPP + double diffusion)
        return named_config("tiny", vmix_choice=3, ldbl_diff=1, hmix_momentum=4, hmix_tracer=4, lvariable_hmix=1,
                            am=-1.0e23, ah=-1.0e22, stepped_bathymetry=1, partial_bottom_cells=1, **small)
    if case == "padded":      # block size 10 x 8 on the 24 x 20 domain: 3 x 3 blocks, the last column / row 4 wide / 4 high (blocks.F90:174-265)
        return named_config("tiny", vmix_choice=3, hmix_momentum=4, hmix_tracer=4, lvariable_hmix=1, am=-1.0e23, ah=-1.0e22,
                            stepped_bathymetry=1, nx_global=24, ny_global=20, km=16, block_size_x=10, block_size_y=8)
    if case == "gm":          # Gent-McWilliams + isopycnal tracer mixing (no cancellation of the skew-flux terms), KPP, stepped bathymetry
        return named_config("tiny", hmix_tracer=3, ah=0.8e7, ah_bolus=0.5e7, slm_b=0.2, vmix_choice=3, stepped_bathymetry=1, **small)
    if case == "gm_tlt":      # ... with the transition layer and the buoyancy-frequency-dependent kappa recomputed every step (the CESM set-up but for kappa_freq)
        return named_config("tiny", hmix_tracer=3, ah=0.8e7, ah_bolus=0.5e7, gm_transition_layer=1, gm_kappa_type=1, gm_kappa_freq=1,
                            vmix_choice=3, stepped_bathymetry=1, **small)
    if case == "upwind3":     # third-order upwind tracer advection + Richardson vmix
        return named_config("tiny", tadvect=2, vmix_choice=2, **small)
    if case == "lw_lim":      # Lax-Wendroff advection with one-dimensional flux limiters + KPP
        return named_config("tiny", tadvect=3, vmix_choice=3, stepped_bathymetry=1, **small)
    if case == "robert":      # Robert-Asselin-Williams time filter (step_RF) instead of averaging steps
        return named_config("tiny", tmix_opt=3, **small)
    if case == "pcsi_evp":    # CESM's production solver pair: P-CSI with the EVP block preconditioner (one 24x20 block: 8/8/8 x 8/6/6 pieces)
        return named_config("tiny", solver_choice=3, precond_choice=1, nx_global=24, ny_global=20, km=16, block_size_x=24, block_size_y=20)
    raise KeyError(case)


def surface_fluxes(tlat):
    return -2.0e-2 * np.sin(tlat) - 5.0e-3, 2.0e-6 * np.cos(2.0 * tlat)


def prepare(model, case):
    """Set the surface tracer fluxes (the KPP case needs buoyancy forcing).  `model` is an Oracle or a
    PopModel-like object with f2()/set()."""
    if case not in ("kpp_del4", "pbc_kpp_del4", "padded", "gm", "gm_tlt"):
        return
    tlat = model.f2("TLAT") if hasattr(model, "f2") else model.get("TLAT")
    st, ss = surface_fluxes(tlat)
    if hasattr(model, "f2"):
        model.f2("STF", 1, 0)[...] = st
        model.f2("STF", 1, 1)[...] = ss
    else:
        model.set("STF", st, n=0)
        model.set("STF", ss, n=1)


def main():
    from orclib import Oracle
    only = sys.argv[1:]
    for case in NSTEPS:
        if only and case not in only:
            continue
        o = Oracle(config(case))
        prepare(o, case)
        iters = [o.step() for _ in range(NSTEPS[case])]
        out = {"nsteps": NSTEPS[case], "iters": np.array(iters)}
        # the cells a comparison may look at: global index non-zero in both directions (padded blocks hold nothing beyond)
        ig = o.ivec("i_glob", o.nxb * o.nblocks).reshape(o.nblocks, o.nxb)
        jg = o.ivec("j_glob", o.nyb * o.nblocks).reshape(o.nblocks, o.nyb)
        out["exists"] = (jg != 0)[:, :, None] & (ig != 0)[:, None, :]
        for name, three_d in FIELDS:
            a = (o.f3 if three_d else o.f2)(name, 1, 0).copy()
            out[name] = np.where(out["exists"][:, None] if three_d else out["exists"], a, 0.0) if case == "padded" else a
        np.savez_compressed(os.path.join(HERE, "golden_%s.npz" % case), **out)
        print(case, "iters", iters, "Tmax", out["TRACER"].max())
        o.close()


if __name__ == "__main__":
    main()
