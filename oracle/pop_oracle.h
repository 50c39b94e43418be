/*
 * pop_oracle.h -- CPU restatement of the POP2 per-timestep dynamics hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / reported CPU baseline.
 *
 * It follows the reference (ESCOMP/POP2-CESM, read-only under /root/reference)
 * statement for statement, in the reference's own (i,j,k,block) i-fastest
 * layout and 1-based index conventions; every function cites the file:line it
 * restates.  The reference itself cannot be compiled here, so
 * this restatement is pinned by the reference's own fixtures only:
 *   - MWJF known answer rho(S=35,theta=20,p=200bar)=1.033213242
 *     (source/state_mod.F90:413-414)
 *   - the constructive halo rule of test/unit/halo/POP.F90Dipole:134-147
 *   - the serial-sum rule of test/unit/reduction/POP.F90
 * For advection / hmix / vmix / solvers the reference holds no fixtures:
 * parity for those is "unpinned by reference tests" (see DESIGN.md).
 */
#ifndef POP_ORACLE_H
#define POP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order as include/pop_amd.h's pop_config so one ctypes
 * Structure serves both libraries (declared separately on purpose). */
#define ORC_CONFIG_VERSION 5   /* = POP_CONFIG_VERSION of include/pop_amd.h */
typedef struct {
  int struct_version;         /* = ORC_CONFIG_VERSION = include/pop_amd.h POP_CONFIG_VERSION; orc_create returns NULL for any other */
  int nx_global, ny_global, km, nt;
  int block_size_x, block_size_y;
  int ew_boundary;            /* 0 closed, 1 cyclic */
  int ns_boundary;            /* 0 closed, 1 cyclic, 2 tripole (time stepping needs orc_create_with_grid) */
  int hmix_momentum;          /* 2 del2, 4 del4 */
  int hmix_tracer;            /* 2 del2, 4 del4 */
  int lvariable_hmix;         /* variable hmix coefficients */
  int vmix_choice;            /* 1 const, 2 rich, 3 kpp */
  int tadvect;                /* 1 centered, 2 upwind3, 3 lw_lim */
  int solver_choice;          /* 1 pcg, 2 ChronGear, 3 PCSI */
  int max_iterations;
  int convergence_check_freq;
  int tmix_opt;               /* 0 none, 1 avg, 2 avgfit, 3 robert */
  int time_mix_freq;
  int steps_per_day;
  int lbouss_correct, lpressure_avg, impcor, reset_to_freezing;
  int lrich, ldbl_diff, lshort_wave, lcheckekmo, num_v_smooth_Ri; /* kpp */
  int maxlanczosstep, convergence_check_start, preconditioner_choice;   /* solvers_nml */
  int stepped_bathymetry;     /* test extension */
  int distribution_type;
  int kpp_ml_diagnostics;
  int sw_absorption_type, jerlov_water_type, lsw_absorb;
  int partial_bottom_cells;   /* grid.F90:916-1020 */
  int gm_slope_control;       /* hmix_tracer = 3 (gm): 0 notanh, 1 tanh */
  int gm_kappa_type, gm_kappa_freq;   /* 0 constant | 1 bfre; 0 never | 1 every_time_step */
  double am, ah;              /* del2 or del4 coefficients */
  double const_vvc, const_vdc;
  double convect_diff, convect_visc, bottom_drag, aidif;
  double rich_bckgrnd_vvc, rich_bckgrnd_vdc, rich_mix;
  double bckgrnd_vdc1, bckgrnd_vdc2, bckgrnd_vdc_dpth, bckgrnd_vdc_linv;
  double Prandtl, kpp_rich_mix;
  double convergence_criterion;
  double init_ts_perturbation, robert_alpha, robert_nu, lanczos_convergence_criterion;
  double ah_bolus, ah_bkg_srfbl, slm_r, slm_b;   /* hmix_gm_nml; 0 = ah, ah, 0.3, 0.3 */
  int gm_transition_layer;
  int gm_diag_bolus;
  int gm_kappa_bkg_srfbl;
  int reserved_i[1];
  double ah_bkg_bottom;
  double kappa_depth_1, kappa_depth_2, kappa_depth_scale;
} orc_config;

typedef struct orc_model orc_model;

/* horiz_grid_opt = 'file' / topography_opt = 'file' (grid.F90:1314-1542 read_horiz_grid, :2025-2107 read_topography):
 * the records of horiz_grid_file (ULAT, ULON, HTN, HTE, HUS, HUW, ANGLE) and of topography_file (KMT) as global
 * (nx_global, ny_global) arrays, i fastest.  KMT = NULL: topography_internal on the supplied ULAT/ULON. */
typedef struct {
  const double *ULAT, *ULON, *HTN, *HTE, *HUS, *HUW, *ANGLE;
  const int *KMT;
  const double *DZBC;   /* partial_bottom_cells: record of bottom_cell_file (grid.F90:2116-2186), or NULL */
} orc_grid_input;

orc_model *orc_create(const orc_config *cfg);
orc_model *orc_create_with_grid(const orc_config *cfg, const orc_grid_input *grid);   /* grid = NULL: orc_create */
void       orc_destroy(orc_model *m);

/* array access by the reference's variable name; tl = 0 old,1 cur,2 new
 * (logical time level, resolved through the rotating indices); n = tracer.
 * Returns pointer to the (nx_block,ny_block[,km],nblocks) array, or NULL. */
double *orc_field(orc_model *m, const char *name, int tl, int n);
int    *orc_ifield(orc_model *m, const char *name);
double *orc_vfield(orc_model *m, const char *name);  /* vertical 1-D arrays */
int     orc_dim(orc_model *m, const char *name);
double  orc_scalar(orc_model *m, const char *name);

/* the step_mod.F90 call sequence, one entry per reference routine */
void orc_time_manager(orc_model *m);      /* set step flags for next step */
void orc_dhdt(orc_model *m);
int  orc_baroclinic_driver(orc_model *m);
/* the same in stages (bit mask): 1 vmix_coeffs + tracer_update, 2 impvmixt + halo, 4 state(new), 8 clinic, 16 impvmixu + finish */
int  orc_baroclinic_stages(orc_model *m, int stages);
int  orc_barotropic_driver(orc_model *m);
void orc_baroclinic_correct_adjust(orc_model *m);
void orc_step_tail(orc_model *m);
int  orc_step(orc_model *m);              /* all of the above */

/* pieces exposed for unit parity */
void   orc_state(orc_model *m, int k, int kk, const double *T, const double *S,
                 double *rho, double *drhodt, double *drhods, int npts);
double orc_state_point(double T, double S_msu, double p_bar);
void   orc_halo_update(orc_model *m, double *a, int nz, int fieldloc_unused);
void   orc_halo_update_int(orc_model *m, int *a);
/* POP_HaloUpdate(array, halo, fieldLoc, fieldKind): the ordinary update, plus the tripole pass when ns_boundary = 2 */
enum { ORC_CENTER = 0, ORC_NECORNER = 1, ORC_NFACE = 2, ORC_EFACE = 3 };
enum { ORC_SCALAR = 0, ORC_VECTOR = 1, ORC_ANGLE = 2 };
void   orc_halo(orc_model *m, double *a, int nz, int loc, int kind);
void   orc_halo_int(orc_model *m, int *a, int loc, int kind);
/* tripole northern boundary (mpi/POP_HaloMod.F90:1936-2050); loc 0 centre, 1 NE corner, 2 N face, 3 E face; kind 0 scalar, 1 vector, 2 angle */
void   orc_halo_update_tripole(orc_model *m, double *a, int nz, int loc, int kind);
void   orc_halo_update_tripole_int(orc_model *m, int *a, int loc, int kind);
double orc_global_sum(orc_model *m, const double *a, const double *mask);
double orc_global_sum_tripole(orc_model *m, const double *a, const double *mask, int loc);
int    orc_solver_iterations(orc_model *m);
double orc_solver_rms(orc_model *m);
/* POP_SolversMod.F90:2268-2369 preconditioner on whole arrays (EVP when preconditioner_choice = 1), :2992 partition */
void orc_preconditioner(orc_model *m, const double *X, double *PX);
int orc_evp_info(orc_model *m, int what, int idx);
void orc_btrop_operator(orc_model *m, const double *X, double *AX);   /* POP_SolversMod.F90:2414-2426 */
int orc_solver_run(orc_model *m, double *X, const double *B);
/* operators.F90: 0 grad :126-192, 1 div :49-119, 2 zcurl :199-272 on whole 2-D fields at level k */
void orc_operator(orc_model *m, int op, int k, const double *A, const double *B, double *O1, double *O2);         /* :327-417 */

#ifdef __cplusplus
}
#endif
#endif
