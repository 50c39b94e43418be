/*
 * pop_amd.h -- C ABI of the MI355X-native POP2 dynamics core (libpop_amd.so).
 *
 * This is the drop-in boundary for the reference's per-timestep hot path.
 * The reference (ESCOMP/POP2-CESM) has no FFI: its seam is the Fortran
 * module-procedure surface that source/step_mod.F90 calls.  Each entry point
 * below names the reference routine it replaces (file:line under
 * /root/reference/); pop2-cesm_amd/fortran/ holds ISO_C_BINDING modules with
 * the reference's names and argument lists that forward here, and
 * INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions:
 *   - plain pointers and sizes only; no torch / HIP types in signatures;
 *   - every function returns int: 0 = POP_Success, nonzero = error, message
 *     via pop_last_error() (mirrors POP_ErrorSet/return, POP_ErrorMod.F90:82);
 *   - one host thread per GPU/rank; all entry points are collective across
 *     ranks (same order on every rank), like the reference's MPI tasks;
 *   - host arrays are Fortran order (nx_block, ny_block[, km], nblocks_local),
 *     i fastest, exactly source/prognostic.F90:47-66 per time level;
 *   - the three time levels are rotated by index, never copied
 *     (step_mod.F90:827-830): tl = 0 old, 1 cur, 2 new (logical).
 */
#ifndef POP_AMD_H
#define POP_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

/* Run-time configuration: the namelist subset this path reads
 * (domain_nml, grid_nml, time_manager_nml, hmix_*_nml, vertical_mix_nml,
 *  vmix_*_nml, advect_nml, pressure_grad_nml, baroclinic_nml, &solvers). */
#define POP_CONFIG_VERSION 5   /* layout of pop_config below; pop_create refuses any other struct_version (4: gm_transition_layer appended; 5: gm_diag_bolus and
                                 * gm_kappa_bkg_srfbl out of reserved_i, ah_bkg_bottom and kappa_depth_* appended -- added late in round 3 without a new number) */
typedef struct pop_config {
  int struct_version;         /* = POP_CONFIG_VERSION (round 3: every option has its own named field) */
  int nx_global, ny_global, km, nt;   /* domain_size.F90 */
  int block_size_x, block_size_y;     /* domain_size.F90 */
  int ew_boundary;            /* 0 closed, 1 cyclic   (domain.F90 ew_boundary_type) */
  int ns_boundary;            /* 0 closed, 1 cyclic, 2 tripole (time stepping: with pop_create_with_grid) */
  int hmix_momentum;          /* 2 del2, 4 del4       (horizontal_mix.F90:427-472) */
  int hmix_tracer;            /* 2 del2, 4 del4, 3 gm: Gent-McWilliams eddy transport + isopycnal (Redi) diffusion with ah as the isopycnal
                               * diffusivity (hmix_gm.F90:1102-2226 in the code-default hmix_gm_nml: constant kappa, kappa_freq 'never',
                               * no transition layer; see gm_slope_control, ah_bolus ... below) */
  int lvariable_hmix;         /* hmix_del2.F90:223, hmix_del4.F90:200 */
  int vmix_choice;            /* 1 const, 2 rich, 3 kpp (vertical_mix.F90:280-296) */
  int tadvect;                /* 1 centered, 2 upwind3, 3 lw_lim (advection.F90:1667-1729, 2313-2676, 2684-3280) */
  int solver_choice;          /* 1 pcg, 2 ChronGear, 3 PCSI with Lanczos eigenvalue bounds
                               * (POP_SolversMod.F90:442-472, 1510-1835, 2699-2990) */
  int max_iterations;
  int convergence_check_freq;
  int tmix_opt;               /* 0 none, 1 avg, 2 avgfit (time_management.F90:2170-2213), 3 robert
                               * (Robert-Asselin-Williams filter, step_mod.F90:919-1350) */
  int time_mix_freq;
  int steps_per_day;          /* dt_option='steps_per_day', dt_count */
  int lbouss_correct, lpressure_avg, impcor, reset_to_freezing;
  int lrich, ldbl_diff, lshort_wave, lcheckekmo, num_v_smooth_Ri; /* vmix_kpp_nml (lshort_wave reads SHF_QSW: pop_set_field) */
  int maxlanczosstep;         /* solvers_nml, PCSI (POP_SolversMod.F90:626): 0 = 20 */
  int convergence_check_start;/* solvers_nml convergenceCheckStart, PCSI (:636): 0 = 60 */
  int preconditioner_choice;  /* solvers_nml preconditionerChoice (:124, :252-290, :2434-2696): 0 'diagonal', 1 'evp'; any solver_choice */
  int stepped_bathymetry;     /* synthetic topography: 0 the reference's internal flat bottom (grid.F90:880-884, 1957-1985),
                               * 1 stepped bathymetry KMT = 3 ... km (TEST EXTENSION, not in the reference) */
  int distribution_type;      /* domain_nml clinic_distribution_type (domain.F90): 0 contiguous runs of equal block counts
                               * ('cartesian' for one column of blocks), 1 contiguous runs of equal ocean columns
                               * (load-balanced, in the spirit of 'rake' / 'spacecurve', distribution.F90) */
  int kpp_ml_diagnostics;     /* 1: the diagnostic mixed-layer depths of vmix_coeffs_kpp every step (HMXL, HMXL_DR,
                               * vmix_kpp.F90:1310-1418; fields "HMXL", "HMXL_DR"); 0: not computed (nothing on the path reads them) */
  int sw_absorption_type;     /* sw_absorption_nml: 0 'top-layer', 1 'jerlov' (sw_absorption.F90:736-811),
                               * 2 'chlorophyll' (:467-728, 951-1047; the field "CHL", mg/m^3, is 0.25 until set with pop_set_field) */
  int jerlov_water_type;      /* 1..5 (0 = 3, the CESM default) */
  int lsw_absorb;             /* != 0: the penetrating short wave SHF_QSW heats the levels below the first
                               * (add_sw_absorb, sw_absorption.F90:818-947, in tracer_update) with sw_absorption_type */
  int partial_bottom_cells;   /* grid_nml partial_bottom_cells (grid.F90:916-1020): 1 = the bottom T cell of every column has the
                               * thickness DZBC (pop_grid_input.DZBC = the record of bottom_cell_file; NULL with the internal
                               * topography: a synthetic DZBC in (0.25, 1] dz(KMT), TEST EXTENSION), DZT / DZU as the reference forms them */
  int gm_slope_control;       /* hmix_gm_nml slope_control_choice: 0 'notanh' (the default, hmix_gm.F90:1508-1539), 1 'tanh' (:1490-1506),
                               * 2 'clip' (:1541-1573: the slopes themselves are limited), 3 'Gerd' (:1575-1594) */
  int gm_kappa_type;          /* hmix_gm_nml kappa_isop_choice = kappa_thic_choice: 0 'constant', 2 'depth' (kappa_depth_1 + kappa_depth_2 exp(-zt / kappa_depth_scale),
                               * hmix_gm.F90:850-874), 1 'bfre' (buoyancy_frequency_dependent_profile,
                               * hmix_gm.F90:3011-3180: KAPPA_VERTICAL = N^2 / N_ref^2 in [0.1, 1] below the surface diabatic layer; kappa_*_deep = 0.1) */
  int gm_kappa_freq;          /* kappa_freq_choice with gm_kappa_type = 1: 0 'never' (refused with 'bfre' as in the reference, hmix_gm.F90:756-780),
                               * 1 'every_time_step', 2 'once_a_day' (the first step after a day has ended, eod_last; for runs that start at midnight:
                               * with avgfit the fit interval is the day, without averaging steps every steps_per_day-th step ends one) */
  double am, ah;              /* del2 [cm^2/s] or del4 [cm^4/s] */
  double const_vvc, const_vdc;
  double convect_diff, convect_visc, bottom_drag, aidif;
  double rich_bckgrnd_vvc, rich_bckgrnd_vdc, rich_mix;
  double bckgrnd_vdc1, bckgrnd_vdc2, bckgrnd_vdc_dpth, bckgrnd_vdc_linv;
  double Prandtl, kpp_rich_mix;
  double convergence_criterion;
  double init_ts_perturbation;           /* amplitude of the synthetic initial T perturbation (1e-2) */
  double robert_alpha, robert_nu;        /* tmix_opt = 3 (0 = defaults 0.53, 0.20, time_management.F90:461-462) */
  double lanczos_convergence_criterion;  /* PCSI LanczosconvergenceCriterion (0 = 0.1) */
  /* hmix_gm_nml (hmix_gm.F90:364-428); each 0 = the value that makes the default set-up: */
  double ah_bolus;                       /* thickness (bolus) diffusivity, 0 = ah (then the skew-flux terms cancel, 'cancellation_occurs' :970-983) */
  double ah_bkg_srfbl;                   /* horizontal diffusivity inside the surface boundary layer, 0 = ah */
  double slm_r, slm_b;                   /* maximum slope for isopycnal / thickness diffusion, 0 = 0.3 */
  int gm_transition_layer;               /* hmix_gm_nml transition_layer_on (hmix_gm.F90:3183-3848: transition_layer, merged_streamfunction,
                                          * apply_vertical_profile_to_isop_hor_diff); with KPP the diabatic depth is the smoothed HMXL */
  int gm_diag_bolus;                     /* hmix_gm_nml diag_gm_bolus: 1 = the eddy-induced (bolus) velocity of hdifft_gm (hmix_gm.F90:2079-2151) every step:
                                          * fields "UISOP", "VISOP" (east / north face of the T cell) and "WISOP" (top of the T cell); nothing on the path reads them */
  int gm_kappa_bkg_srfbl;                /* 1: hmix_gm_nml use_const_ah_bkg_srfbl = .false. -- the horizontal diffusivity of the surface boundary layer follows
                                          * KAPPA_ISOP instead of ah_bkg_srfbl (hmix_gm.F90:1369-1370, 1602-1631) */
  int reserved_i[1];                     /* must be 0 */
  double ah_bkg_bottom;                  /* hmix_gm_nml ah_bkg_bottom: horizontal diffusivity in the bottom half of the bottom cell (:1757-1761), 0 = none */
  double kappa_depth_1, kappa_depth_2, kappa_depth_scale;   /* gm_kappa_type = 2; scale 0 = 150000 cm */
} pop_config;

typedef struct pop_ctx pop_ctx;

/* flags for pop_create */
#define POP_CREATE_HOST_ONLY 1   /* build blocks/grid/plans on the host, touch no GPU */
#define POP_CREATE_PLAN_ONLY 2   /* implies HOST_ONLY: the block table (create_blocks, blocks.F90:100-270), the distribution (domain.F90:379-543)
                                  * and the halo plan (POP_HaloCreate, mpi/POP_HaloMod.F90:142-1640) of this rank only -- no grid fields.  For
                                  * checking a decomposition of any size in milliseconds: pop_get_block, pop_local_block_ids, pop_halo_plan_*,
                                  * pop_halo_update_host_*, pop_get_dim("ocean_columns_local" | "ocean_columns_total") work; field access does not */

/* ---- tuning: kernel-form and schedule choices.  NONE changes a result: every alternative is tested bitwise equal to the
 *      default (tests/test_gpu_parity.py, test_gpu_land.py, test_gpu_multirank.py); they exist for measurement and for those
 *      tests.  Resolved ONCE, at pop_create*: the library's size rules, overridden by the caller's pop_tuning (fields left at
 *      POP_TUNING_UNSET keep the rule), overridden by the environment variable POP_<FIELD NAME IN UPPER CASE> (read at create
 *      only; nothing in the step reads the environment).  pop_get_tuning returns what was resolved. ---------------------- */
#define POP_TUNING_UNSET (-2147483647 - 1)
typedef struct pop_tuning {
  int struct_bytes;        /* sizeof(pop_tuning) of the caller */
  int land_skip;           /* 0: every workgroup runs (no land elimination at tile granularity) */
  int land_full_steps;     /* steps after set-up / restart / a new state that run every workgroup (default 4) */
  int xcd_remap;           /* column kernels: workgroup order 0 linear, 1 XCD bands, 2 XCD-strided 2-D tiles */
  int red_tiles, red_band; /* 2-D reduction / solver kernels: 64x4 tiles | XCD-banded chunk order */
  int lds_order;           /* LDS-tiled stencil kernels: 1 = XCD patch order */
  int momentum_lds;        /* momentum right-hand side: LDS tile rows 8 | 4, 0 = direct-load kernel */
  int tracer_lds;          /* tracer right-hand side (centred advection): LDS tile rows 8 | 4, 0 = direct-load kernel */
  int generic_thomas;      /* 1: scratch-staged Thomas kernels for U, V even at km = 60 / 62 */
  int reg_thomas_t;        /* tracers: 1 register Thomas kernels (km = 60 / 62), 0 generic */
  int thomas_pair;         /* corrector: both tracers in one thread when they share the diffusivity array */
  int tracer_fwd;          /* predictor: forward elimination inside the tracer right-hand-side kernel */
  int vdc_shared;          /* 0: two diffusivity arrays even without double diffusion */
  int side_stream;         /* 0: no side stream at all (every launch in the reference's order on one stream) */
  int del4_side;           /* 0: del4 first Laplacians in line instead of beside the vertical-mixing coefficients */
  int del4_tile;           /* del4 first Laplacians: patch rows 0 | 2 | 4 | 8 | 16 */
  int d2t_fuse, d2u_fuse;  /* del4: the previous step's tracer / momentum launch forms the first Laplacian */
  int vmixu_defer;         /* implicit vertical mixing of U, V launched after the solve with the barotropic sum */
  int vmixu_inline;        /* 1: implicit vertical mixing of U, V on the launch stream */
  int btrop_inline;        /* 1: barotropic velocity added in the step tail (the reference's order) */
  int kpp_ahead;           /* KPP of the next step beside the barotropic solver */
  int kpp_col;             /* KPP kernel forms, bit mask: 1 ushear column, 2 buoydiff column, 4 buoydiff LDS, 8 buoydiff + interior fused, 16 the two as one column march */
  int kpp_lazy;            /* 0: surface-layer buoyancy difference at every level */
  int kpp_ushear_hint;     /* 0: shear kernel forms every level */
  int kpp_ushear_margin;   /* levels beyond the previous boundary-layer level (default 3; may be negative) */
  int kpp_side_stream;     /* 0: KPP kernels on one stream */
  int kpp_buoy_waves;      /* column buoydiff: waves per SIMD 1 | 2 */
  int kpp_interior_generic;/* 1: scratch-staged interior kernel even at km = 60 / 62 */
  int kpp_src_full;        /* 1: the tracer kernel reads KPP_SRC at every level */
  int solver_unfused;      /* 1: one kernel per solver operation */
  int solver_nograph;      /* 1: no hipGraph replay of the check intervals */
  int solver_presum;       /* 1: block sums by their own launch even on small grids */
  int solver_distributed;  /* 1: never the replicated barotropic solve */
  int solver_overlap_off;  /* 1: solver halo exchange in line with the all-reduce */
  int fpcg_b2;             /* 0: one cell per thread in step B of the fused pcg */
  int pcsi_step2;          /* fused P-CSI step with two cells per thread */
  int halo_separate;       /* 1: one message per field instead of one per neighbour */
  int halo_overlap_off;    /* 1: mid-step tracer halo in line instead of beside interior tiles */
  int rccl_overlap;        /* second communicator: 0 none, 1 (default) own stream, 2 ... */
  int evp_wave;            /* EVP block preconditioner: 0 one thread per sub-block, 1 anti-diagonal wavefronts (8 lanes per sub-block) with the
                            * operands in LDS, 2 with the operands in registers and the coefficients read from their 2-D fields, 3 (default) = 2 with
                            * every load of the front end issued up front (unconditional at clamped addresses; the conditions select values) */
  int fpcg_a_pair;         /* 0: one chunk per workgroup in step A of the fused pcg even on compacted launches */
  int kpp_sparse;          /* 0: KPP boundary-layer kernel streams every level even when the interior kernel formed the convection mask */
  int pbc_generic_thomas;  /* 1: partial bottom cells with the scratch-staged Thomas kernels even at km = 60 / 62 */
  int pbc_generic_kpp;     /* 1: partial bottom cells with the 3-D-parallel / scratch-staged KPP kernels on large grids too */
  int stream_priority;     /* 1: the launch stream above the gap-filling side streams (default 0: measured slower) */
  int gm_sf_stored;        /* 0: Gent-McWilliams without cancellation: the stream-function terms SF_SLX / SF_SLY re-derived in the flux kernel instead of stored */
  int state3d_levels;      /* density of a whole 3-D array: levels per thread, 4 (default) | 2 | 8 with the per-level EOS coefficients read from
                            * a table, 1 = one cell per thread with the coefficients formed in place */
  int pcg_persist;         /* pcg, ChronGear and P-CSI with the diagonal preconditioner on small 2-D systems (<= 2000 chunks of 256 cells in <= 8 blocks, single rank or the replicated solve):
                            * 1 (the default where it applies) = the whole solve as ONE resident launch, vectors in LDS / registers, workgroups exchanging partials and halo z through
                            * tagged 16-byte memory words (kernels_pcg_persist.hpp); 0 = the two-launch fused iteration; 2 | 4 | 8: measurement only, that many chunks per workgroup */
  int gm_flux_tile;        /* 0: Gent-McWilliams fluxes cell by cell (every horizontal face flux evaluated in both cells that share it) instead of once per face in 64 x 4 tiles */
  int pcsi_two_step;       /* fused P-CSI (diagonal preconditioner; one rank or blocks spread over ranks; through a tripole fold whose partners are on the rank): 1 = two iterations per pass over the state where no check follows
                            * (k_pcsi_step_x2; the default on large grids), 0 = one launch per iteration */
  int block_sums_relay;    /* ordered block sums of more than 64 x 256 chunk partials (the fused pcg / ChronGear of large grids): 0 = 256 threads, each adding
                            * its accumulator's terms in batches (two or three memory round trips); default 1 = 1024 threads, four per accumulator, every term
                            * requested at once and the quarters added in turn (k_block_sums_relay: the same additions in the same order); 2 = also below 64 terms (cross-check) */
  int pcsi_evp_fused;      /* fused P-CSI with the EVP preconditioner: 1 = the iteration (dx, x, r = b - A x) and the sub-block solves r' = M^-1 r in ONE
                            * launch (k_pcsi_evp_step; bitwise, but measured slower: its loads are issued by the few waves of the sub-block solve);
                            * default 0 = two launches per iteration (k_pcsi_step2, k_evp_apply_wave2) */
} pop_tuning;
void pop_tuning_init(pop_tuning *t);   /* struct_bytes = sizeof, every field POP_TUNING_UNSET */
int pop_get_tuning(const pop_ctx *ctx, pop_tuning *resolved);   /* fields still POP_TUNING_UNSET: the size rule applied */

/* ---- lifecycle (no reference counterpart: the reference's state is module
 *      global, initial.F90:133-699 builds it) -------------------------------- */
int pop_create(const pop_config *cfg, int rank, int nranks, int flags, pop_ctx **out);
/* horiz_grid_opt = 'file' / topography_opt = 'file' (grid.F90:1314-1542 read_horiz_grid, :2025-2107 read_topography):
 * the seven records of horiz_grid_file and the KMT record of topography_file as global (nx_global, ny_global)
 * arrays, i fastest, scattered with the reference's field locations (mpi/gather_scatter.F90:862-1161, tripole ghost
 * rows mirrored).  KMT = NULL: topography_internal on the supplied ULAT / ULON.  ANGLE is accepted for file-format
 * parity; nothing on this path reads it (the analytic wind stress is not rotated).  The arrays are read during the
 * call only.  A tripole decomposition (ns_boundary = 2) steps only on a grid supplied this way: the internal grid is
 * lat-lon and has no values beyond the fold.  pop_read_grid_files fills the arrays from the reference's direct-access
 * binary files (records of nx_global*ny_global r8 / one record of i4, native byte order). */
typedef struct pop_grid_input {
  const double *ULAT, *ULON, *HTN, *HTE, *HUS, *HUW, *ANGLE;
  const int *KMT;
  const double *DZBC;   /* partial_bottom_cells: the record of bottom_cell_file (read_bottom_cell, grid.F90:2116-2186): thickness of
                         * the bottom T cell of every column [cm], global (nx_global, ny_global), scattered as a centre scalar.
                         * NULL: see pop_config.partial_bottom_cells */
} pop_grid_input;
int pop_create_with_grid(const pop_config *cfg, const pop_grid_input *grid, int rank, int nranks, int flags, pop_ctx **out);
/* the same with the caller's tuning (NULL: none); grid may be NULL */
int pop_create_tuned(const pop_config *cfg, const pop_grid_input *grid, const pop_tuning *tuning, int rank, int nranks, int flags, pop_ctx **out);
int pop_read_grid_files(const char *horiz_grid_file, const char *topography_file, int nx_global, int ny_global,
                        double *seven_records /* 7*nx*ny, or NULL */, int *kmt /* nx*ny, or NULL */);
int pop_destroy(pop_ctx *ctx);
const char *pop_last_error(const pop_ctx *ctx);

/* ---- blocks.F90:43-63 / get_block (blocks.F90:282-320) ---------------------- */
int pop_get_dim(const pop_ctx *ctx, const char *name);         /* nx_block, ny_block, km, nt,
                                                                  nblocks (local), nblocks_tot,
                                                                  nblocks_x, nblocks_y, ... */
double pop_get_scalar(const pop_ctx *ctx, const char *name);   /* dtt, dtu, dtp, residualNorm ... */
/* block_id is the global 1-based id; out[8] = {block_id, local_id, ib, ie, jb, je, iblock, jblock};
 * i_glob (nx_block) / j_glob (ny_block) may be NULL */
int pop_get_block(const pop_ctx *ctx, int block_id, int *out8, int *i_glob, int *j_glob);
int pop_local_block_ids(const pop_ctx *ctx, int *ids /* nblocks local */);

/* ---- state transfer: host arrays in the reference layout -------------------- */
/* name = the reference's variable name (TRACER, UVEL, VVEL, RHO, PSURF, GRADPX, GRADPY, UBTROP,
 * VBTROP, PGUESS, FW, FW_OLD, SMF, SMFT, STF, TFW, SHF_QSW, ZX, ZY, DH, DHU, VDC, VVC, RHS, grid
 * fields DXU ... ); tl = logical time level; n = tracer / vector component.
 * VDC (levels 0 .. km+1): n = 0 / 1 are KPP's two tracer classes.  Without double diffusion (ldbl_diff = 0) the reference fills
 * both with the same values (vmix_kpp.F90 ri_iwmix / blmix) and the library keeps ONE array for them: n = 0 and n = 1 read and
 * write the same storage.  KPP_SRC: a value written with pop_set_field is used at every level until KPP has run again. */
int pop_get_field(pop_ctx *ctx, const char *name, int tl, int n, double *host, long long count);
int pop_set_field(pop_ctx *ctx, const char *name, int tl, int n, const double *host, long long count);
int pop_get_ifield(pop_ctx *ctx, const char *name, int *host, long long count);
long long pop_field_count(const pop_ctx *ctx, const char *name);
/* device pointer of a field (for zero-copy host frameworks); 0 if unknown.  The call is the library's only notice that the caller
 * may WRITE the field: it drops work computed ahead from the old values (KPP look-ahead, del4 first Laplacians), marks the ghost
 * cells as possibly inconsistent and restarts the full (no land elimination) steps, exactly as pop_set_field does -- once.  A
 * caller that writes through a cached pointer must therefore fetch the pointer again before EVERY write (one call per field per
 * step; the call is cheap: no copy, no synchronisation beyond joining the library's side streams); the time levels rotate by
 * index, so the pointer of (name, tl) changes from step to step anyway. */
void *pop_field_device_ptr(pop_ctx *ctx, const char *name, int tl, int n);

/* ---- restart files: write_restart (restart.F90:1095-1715) / read_restart (:184-1088) in the reference's 'bin'
 * format (io_binary.F90): <path> = direct-access records of nx_global*ny_global r8 (native byte order, a 3-D field
 * = km records), <path>.hdr = text header with the scalars as "&GLOBAL" attributes and one "&NAME ... id:int:<first
 * record> ... /" section per field.  Fields: {UBTROP,VBTROP,PSURF,GRADPX,GRADPY}_{CUR,OLD}, PGUESS, FW_OLD,
 * FW_FREEZE (zeros), {UVEL,VVEL,TEMP,SALT}_{CUR,OLD}.  Every rank writes / reads the rows of its own blocks
 * (shared file system), rank 0 writes the header.  Reading applies the land masks and halo updates of
 * read_restart :881-1040, recomputes RHO at both time levels (initial.F90:1665-1681) and continues with leapfrog
 * steps (first_step = .false., initial.F90:1088).  flags bit 0: the data file is byte-swapped. */
int pop_write_restart(pop_ctx *ctx, const char *path);
int pop_read_restart(pop_ctx *ctx, const char *path, int flags);

/* ---- the step_mod.F90:126-911 call sequence --------------------------------- */
int pop_time_manager(pop_ctx *ctx);              /* time_management.F90:1823-1847, 2139-2234 */
int pop_dhdt(pop_ctx *ctx);                      /* surface_hgt.F90:131   dhdt(DH,DHU) */
int pop_baroclinic_driver(pop_ctx *ctx);         /* baroclinic.F90:578    baroclinic_driver(ZX,ZY,DH,DHU,err) */
int pop_barotropic_driver(pop_ctx *ctx);         /* barotropic.F90:267    barotropic_driver(ZX,ZY,err)
                                                    (includes the ZX,ZY halo of step_mod.F90:405-423) */
int pop_barotropic_driver_updated(pop_ctx *ctx); /* barotropic.F90:267 alone: ZX, ZY already halo-updated by the caller
                                                    (as step_mod.F90:405-423 does before the call) */
int pop_baroclinic_correct_adjust(pop_ctx *ctx); /* baroclinic.F90:1217 */
int pop_step_tail(pop_ctx *ctx);                 /* step_mod.F90:467-832  halos, +barotropic, PGUESS,
                                                    averaging step / time-level rotation */
int pop_step(pop_ctx *ctx);                      /* step_mod.F90:126      step(errorCode) */

/* ---- POP_HaloMod / POP_ReductionsMod / POP_SolversMod surface --------------- */
/* POP_HaloUpdate(array, halo, fieldLoc, fieldKind, errorCode, fillValue)
 * mpi/POP_HaloMod.F90:1732-1773 (2-D), :2766-3211 (3-D), :4122-4585 (4-D: n = -1 updates every
 * tracer of a field with a tracer dimension); on a device-resident field */
int pop_halo_update(pop_ctx *ctx, const char *name, int tl, int n);
/* the same with the reference's fieldLoc / fieldKind arguments (POP_GridHorzMod / POP_FieldMod constants):
 * field_loc 0 centre, 1 NE corner, 2 N face, 3 E face; field_kind 0 scalar, 1 vector, 2 angle.  They matter
 * on a tripole northern boundary only (mpi/POP_HaloMod.F90:1936-2050: mirrored copy with offsets and sign,
 * symmetrised degenerate top row for NE-corner / N-face fields; the top row of blocks on one rank). */
int pop_halo_update_loc(pop_ctx *ctx, const char *name, int tl, int n, int field_loc, int field_kind);
int pop_halo_update_host_r8_loc(pop_ctx *ctx, double *array, int nz, double fill, int field_loc, int field_kind);
int pop_halo_update_host_i4_loc(pop_ctx *ctx, int *array, int nz, int fill, int field_loc, int field_kind);
/* host-array variants used at init time and by the unit-test rule of
 * test/unit/halo/POP.F90Dipole (nz = product of trailing dims).  pop_halo_update_host_r8_loc also serves decompositions
 * over several ranks (the array is then staged through a device work field; 1 <= nz <= km) */
int pop_halo_update_host_r8(pop_ctx *ctx, double *array, int nz, double fill);
int pop_halo_update_host_i4(pop_ctx *ctx, int *array, int nz, int fill);
/* POP_GlobalSum(array, dist, fieldLoc, errorCode, mMask) mpi/POP_ReductionsMod.F90:144-389
 * (b4b formulation :348-383); mask_name may be NULL */
int pop_global_sum(pop_ctx *ctx, const char *name, int tl, int n, const char *mask_name, double *result);
/* the other members of the generic interface (mpi/POP_ReductionsMod.F90:50-64), same b4b rule:
 *   POP_GlobalSumNfields2DR8 :823-1084, POP_GlobalSumProd2DR8 :1395-1618,
 *   POP_GlobalSumScalarR8 :1091-1191 (one value per task), POP_GlobalSum2DI4 :621-816 (integer field) */
/* with fieldLoc: on a tripole grid fields on north faces (2) / NE corners (1) count the redundant half of the
 * top row once (:308-341); otherwise identical to pop_global_sum */
int pop_global_sum_loc(pop_ctx *ctx, const char *name, int tl, int n, const char *mask_name, int field_loc, double *result);
/* POP_GlobalCount :2062-2207 (non-zero physical cells), POP_GlobalMaxval/Minval :2670-3223 and
 * POP_GlobalMaxloc/Minloc :4002-4400 (value and global (i,j) of the first cell attaining it; the optional mask
 * selects cells whose mask value is non-zero; iloc / jloc may be NULL) */
int pop_global_count(pop_ctx *ctx, const char *name, int tl, int n, int field_loc, long long *count);
int pop_global_extreme(pop_ctx *ctx, const char *name, int tl, int n, const char *mask_name, int want_max,
                       double *value, int *iloc, int *jloc);
/* POP_GlobalSum on HOST arrays of the local blocks, array(nx_block, ny_block, nblocks) and an optional multiplicative mask
 * of the same shape (NULL: none) -- the reference's own argument list; staged through device work fields, same b4b rule */
int pop_global_sum_host(pop_ctx *ctx, const double *array, const double *mask, int field_loc, double *result);
int pop_global_sum_nfields(pop_ctx *ctx, int nfields, const char *const *names, const int *tl, const int *n,
                           const char *mask_name, double *results);
int pop_global_sum_prod(pop_ctx *ctx, const char *name_a, int tl_a, int n_a, const char *name_b, int tl_b, int n_b,
                        const char *mask_name, double *result);
int pop_global_sum_scalar(pop_ctx *ctx, double local_value, double *result);
int pop_global_sum_i4(pop_ctx *ctx, const char *int_field_name, long long *result);
/* POP_SolversRun(sfcPressure, rhsClinic, errorCode) POP_SolversMod.F90:327;
 * operates on PSURF(newtime) and RHS in place */
int pop_solver_run(pop_ctx *ctx);
/* POP_SolversDiagonal(diagonalCorrection(nx_block,ny_block), blockIndx, errorCode) :1110-1151:
 * centre weight of local block blockIndx (1-based) = time-independent part - correction (host array) */
int pop_solver_diagonal(pop_ctx *ctx, int block_local, const double *diagonal_correction);
/* preconditioner(PX, X, bid) :2268-2369 on every local block: PX = M^-1 X on the physical cells.  M is the EVP
 * block preconditioner (8x8 sub-block solves, :2434-2696; preconditionerChoice = 'evp') when preconditioner_choice = 1,
 * else the diagonal (current centre weight).  Private in the reference; exported so that parity tests can pin the
 * preconditioner on its own.  x_name / px_name: 2-D device fields (time level tl where it applies). */
int pop_solver_preconditioner(pop_ctx *ctx, const char *x_name, int x_tl, const char *px_name, int px_tl);
/* operators.F90 as stand-alone calls on device fields at level k (the time step has them inlined in its kernels):
 * op 0  grad(k, GRADX, GRADY, F)   :126-192   F at T points  -> o1, o2 at U points (0 where k > KMU)
 * op 1  div(k, DIV, UX, UY)        :49-119    a, b at U points -> o1 at T points, times the cell area (0 where k > KMT)
 * op 2  zcurl(k, CURL, UX, UY)     :199-272   a, b at U points -> o1 at T points, times the cell area
 * A 3-D field name selects its level-k slab; tl applies to names with time levels. */
int pop_operator(pop_ctx *ctx, int op, int k, const char *a_name, const char *b_name, int tl, const char *o1_name, const char *o2_name);
/* The same three operators with the reference's own argument lists -- div(k, DIV_OUT, UX, UY, this_block) operators.F90:49,
 * grad(k, GRADX, GRADY, F, this_block) :126, zcurl(k, CURL, UX, UY, this_block) :199 --: host arrays (nx_block, ny_block) of ONE
 * block, block_local = this_block%local_id (1-based).  a (and b for div / zcurl) in, o1 (and o2 for grad) out; the arrays are
 * staged through the GPU (the kernel is the one pop_operator launches). */
int pop_operator_host(pop_ctx *ctx, int op, int k, int block_local, const double *a, const double *b, double *o1, double *o2);
/* POP_SolversGetDiagnostics(iterationCount, residual, errorCode) :1158 */
int pop_solver_get_diagnostics(const pop_ctx *ctx, int *iterations, double *rms_residual);
/* state(k,kk,TEMPK,SALTK,this_block,RHOOUT,...) state_mod.F90:258 on n device-resident or
 * host points (host variant stages through the GPU) */
int pop_state_host(pop_ctx *ctx, int kk, const double *T, const double *S, double *rho,
                   double *drhodt, double *drhods, long long n);

/* ---- multi-rank transport: the library packs/unpacks on the GPU and asks the
 *      host to move bytes (RCCL via torch.distributed in bench.py, MPI/RCCL from
 *      Fortran).  Buffers are device pointers owned by the host framework. ---- */
typedef int (*pop_exchange_fn)(void *user, int nmsg, const int *peer, const long long *send_off,
                               const long long *send_cnt, const long long *recv_off,
                               const long long *recv_cnt);   /* offsets/counts in doubles */
typedef int (*pop_allreduce_fn)(void *user, long long off, long long cnt);
int pop_set_comm(pop_ctx *ctx, void *dev_sendbuf, void *dev_recvbuf, void *dev_redbuf,
                 long long buf_doubles, pop_exchange_fn xchg, pop_allreduce_fn allred, void *user);
long long pop_comm_buffer_doubles(const pop_ctx *ctx);   /* size the host must provide */
/* reduce buffer (block-sum vectors; in replicated-barotropic mode also the gathered RHS + guess) */
long long pop_reduce_buffer_doubles(const pop_ctx *ctx);
int pop_set_reduce_buffer(pop_ctx *ctx, void *dev_redbuf, long long doubles);
/* in-library RCCL transport (replaces the communicator set-up of mpi/POP_CommMod.F90:70-135 and the
 * MPI calls of mpi/POP_HaloMod.F90:1865-1960, mpi/POP_ReductionsMod.F90:348-383): rank 0 obtains a
 * 128-byte id, the host broadcasts it by whatever means it has (MPI_Bcast, torch.distributed), every
 * rank calls pop_comm_init_rccl.  Halo messages and block-sum all-reduces are then enqueued on the
 * context's stream with no host round trip.  librccl is opened at run time. */
int pop_rccl_unique_id(unsigned char *id128);
int pop_comm_init_rccl(pop_ctx *ctx, const unsigned char *id128);
/* what the installed transport is, for a run's record: out6 = {kind (0 none, 1 host callbacks of pop_set_comm, 2 in-library RCCL),
 * ncclCommCount of the first communicator, its ncclCommUserRank, ncclCommCount of the second communicator (0: none),
 * number of neighbour ranks in the halo plan, 1 if the mid-step halo runs beside interior tiles}; -1 where RCCL does not say.
 * lib_path (may be NULL) receives the name of the librccl that was bound. */
int pop_comm_info(const pop_ctx *ctx, int *out6, char *lib_path, int lib_path_bytes);
/* checks the installed transport (either kind): an all-reduce of known values + a self message */
int pop_comm_selftest(pop_ctx *ctx);
/* halo plan introspection (host logic, testable without a GPU) */
int pop_halo_plan_counts(const pop_ctx *ctx, int *n_local_copies, int *n_fill, int *n_peers);
int pop_halo_plan_peer(const pop_ctx *ctx, int ipeer, int *peer_rank, int *n_send, int *n_recv);
int pop_halo_plan_lists(const pop_ctx *ctx, int ipeer, int *send_src /* n_send */, int *recv_dst /* n_recv */);
int pop_halo_plan_local(const pop_ctx *ctx, int *dst, int *src /* n_local_copies */, int *fill_dst /* n_fill */);

/* ---- timers (timers.F90 phase names STEP/BAROCLINIC/BAROTROPIC/3D-UPDATE) --- */
int pop_timers_reset(pop_ctx *ctx);
int pop_timer_ms(pop_ctx *ctx, const char *name, double *total_ms, int *calls);
/* bench support: time `reps` launches of one named kernel phase with HIP events on the
 * stream it runs on; returns average ms */
int pop_time_phase(pop_ctx *ctx, const char *phase, int reps, double *avg_ms);
/* one phase of baroclinic_driver / baroclinic_correct_adjust on its own (the public routines the reference's drivers
 * call one after the other): "vmix" vmix_coeffs vertical_mix.F90:518, "hmix_tracer" / "hmix_momentum" first Laplacians of
 * hmix_del4.F90:1021 / :730 (no-ops for del2), "tracer_rhs" tracer_update baroclinic.F90:1902, "impvmixt"
 * vertical_mix.F90:1164, "state" state_mod.F90:258 on the new tracers, "momentum_rhs" clinic baroclinic.F90:1635,
 * "impvmixu" vertical_mix.F90:1679 + baroclinic.F90:1077-1129, "correct" impvmixt_correct :1460 with the surface
 * terms of baroclinic.F90:1261-1475, "add_btrop" step_mod.F90:572-600.  Uses the step parameters of the last
 * pop_time_manager call. */
int pop_run_phase(pop_ctx *ctx, const char *phase);
int pop_device_sync(pop_ctx *ctx);
/* run all launches on the host framework's HIP stream (e.g. torch.cuda.current_stream().cuda_stream) */
int pop_set_stream(pop_ctx *ctx, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
