/* orc_internal.h -- model state of the CPU restatement (TEST INFRASTRUCTURE). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pop_oracle.h"

struct orc_model {
  orc_config c;
  const orc_grid_input *gin;  /* only during construction */
  int nxb, nyb, km, nt, nblocks, nbx, nby;
  size_t n2, n3;
  int *i_glob, *j_glob, *blk_ib, *blk_ie, *blk_jb, *blk_je;
  /* vertical grid, 1-based (index 0 used by dzw, dzwr, zgrid) */
  double *dz, *dzw, *zt, *zw, *c2dz, *dzr, *dz2r, *dzwr, *pressz, *bouss, *dt, *c2dtt;
  double *afac_t, *afac_u, *hfac;
  /* horizontal grid */
  double *ULAT, *ULON, *TLAT, *HTN, *HTE, *HUS, *HUW, *DXU, *DYU, *DXT, *DYT;
  double *DXUR, *DYUR, *DXTR, *DYTR, *UAREA, *TAREA, *UAREA_R, *TAREA_R;
  double *AU0, *AUN, *AUE, *AUNE, *FCOR, *FCORT, *HU, *HUR, *HT, *RCALCT, *RCALCU;
  int *KMT, *KMU, *KMTN, *KMTS, *KMTE, *KMTW, *KMTEE, *KMTNN;
  /* partial bottom cells (grid.F90:916-1020): DZBC(nx,ny,blk) bottom T-cell thickness; DZT, DZU (nx,ny,0:km+1,blk), levels 0 and
   * km+1 stay 0 as in the reference; NULL without partial_bottom_cells */
  double *DZBC, *DZT, *DZU;
  double uarea_equator;
  /* hmix */
  double *AMF, *AHF, *DTN, *DTS, *DTE, *DTW;
  double *DUC, *DUN, *DUS, *DUE, *DUW, *DMC, *DMN, *DMS, *DME, *DMW, *DUM, *KXU, *KYU;
  /* del4 */
  double *D4_AMF, *D4_AHF;
  /* solver */
  double *btropWgtNE, *btropWgtEast, *btropWgtNorth, *centerWgtIndep, *centerWgt, *mMask;
  double *CHECKER, *CONSTNT;
  double residualNorm, convergenceCriterion, rcheck, rconst, rmsResidual;
  int numIterations;
  /* prognostic state (3 physical slots, rotated by index) */
  double *TRACER[8][3], *UVEL[3], *VVEL[3], *RHO[3];
  double *PSURF[3], *GRADPX[3], *GRADPY[3], *UBTROP[3], *VBTROP[3];
  double *PGUESS, *FW, *FW_OLD;
  int oldtime, curtime, newtime, mixtime;
  /* forcing */
  double *SMF[2], *SMFT[2], *STF[8], *TFW[8], *SHF_QSW, *CHL;   /* CHL: chlorophyll, mg/m^3 (sw_absorption_type 'chlorophyll') */
  /* work fields */
  double *DH, *DHU, *ZX, *ZY, *UH, *VH, *RHS;
  double *VDC[2], *VVC, *KPP_SRC[8];   /* VDC: (nx,ny,0:km+1) per tracer class */
  double *HBLT, *HMXL, *HMXL_DR; int *KBL;
  /* time stepping */
  double dtt, dtu, dtp, c2dtu, c2dtp, beta;
  int first_step, leapfrogts, f_euler_ts, avg_ts, nsteps_total, nsteps_this_interval, nsteps_per_interval;
  int eod, eod_last;          /* the step ends a day / the previous one did (time_management.F90:1809, 3586-3592), for runs that start at midnight */
  /* kpp */
  void *kpp;
  void *del4;
  void *upw3;
  void *lwlim;                /* orc_lwlim.inc (tadvect = 3) */
  void *gm;                   /* orc_gm.inc (hmix_tracer = 3) */
  void *sw;                   /* short-wave absorption tables (orc_kpp.inc: orc_sw) */
  void *rf;
  void *pcsi;
  void *evp;
};

extern const double orc_grav, orc_omega, orc_radius;
#endif
