/*
 * pop_oracle.c -- CPU restatement of the POP2 dynamics hot path (TEST
 * INFRASTRUCTURE ONLY -- see pop_oracle.h).  Plain C, fp64, compiled with
 * -ffp-contract=off so the operation order below IS the arithmetic.
 *
 * Conventions: 1-based (i,j,k) as in the Fortran; arrays are i-fastest
 * (nx_block, ny_block[, km], nblocks).  All citations are file:line under
 * /root/reference/.
 *
 * Parity status: pinned by the reference's own fixtures for state (MWJF
 * known answer), halo rule and global-sum rule; unpinned by reference tests
 * for advection/hmix/vmix/solver (the reference has none).
 */
#include "pop_oracle.h"
#include "orc_internal.h"

/* ------------------------------------------------------------------ */
/* constants: source/pop_constants.F90:40-56, 234-266 (non-CCSMCOUPLED) */
/* ------------------------------------------------------------------ */
const double orc_grav = 980.6, orc_omega = 7.292123625e-5, orc_radius = 6370.0e5;

static double *dalloc(size_t n) {
  double *p = (double *)calloc(n ? n : 1, sizeof(double));
  if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
  return p;
}
static int *ialloc(size_t n) {
  int *p = (int *)calloc(n ? n : 1, sizeof(int));
  if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
  return p;
}

/* eoshift of a 2-D block array with zero fill: value of A at (i+di, j+dj)
 * or 0 outside the block (Fortran eoshift default boundary). */
static inline double esh(const double *A, int nxb, int nyb, int i, int j) {
  if (i < 1 || i > nxb || j < 1 || j > nyb) return 0.0;
  return A[(size_t)(j - 1) * nxb + (i - 1)];
}
static inline int ieshf(const int *A, int nxb, int nyb, int i, int j) {
  if (i < 1 || i > nxb || j < 1 || j > nyb) return 0;
  return A[(size_t)(j - 1) * nxb + (i - 1)];
}

/* ------------------------------------------------------------------ */
/* blocks: source/blocks.F90:90-275 (create_blocks)                    */
/* ------------------------------------------------------------------ */
static void create_blocks(orc_model *m) {
  const orc_config *c = &m->c;
  int nghost = 2;
  m->nbx = (c->nx_global - 1) / c->block_size_x + 1;
  m->nby = (c->ny_global - 1) / c->block_size_y + 1;
  m->nblocks = m->nbx * m->nby;
  m->nxb = c->block_size_x + 2 * nghost;
  m->nyb = c->block_size_y + 2 * nghost;
  m->i_glob = ialloc((size_t)m->nxb * m->nblocks);
  m->j_glob = ialloc((size_t)m->nyb * m->nblocks);
  m->blk_ib = ialloc(m->nblocks); m->blk_ie = ialloc(m->nblocks);
  m->blk_jb = ialloc(m->nblocks); m->blk_je = ialloc(m->nblocks);
  int n = 0;
  for (int jblock = 1; jblock <= m->nby; jblock++) {
    int js = (jblock - 1) * c->block_size_y + 1;
    for (int iblock = 1; iblock <= m->nbx; iblock++) {
      int is = (iblock - 1) * c->block_size_x + 1;
      int *ig = m->i_glob + (size_t)n * m->nxb, *jg = m->j_glob + (size_t)n * m->nyb;
      m->blk_ib[n] = nghost + 1; m->blk_jb[n] = nghost + 1;
      m->blk_ie[n] = m->nxb - nghost; m->blk_je[n] = m->nyb - nghost;
      for (int j = 1; j <= m->nyb; j++) {
        int g = js - nghost + j - 1;
        if (g < 1) g = (c->ns_boundary == 1) ? g + c->ny_global : 0;
        if (g > c->ny_global + nghost) g = 0;
        else if (g > c->ny_global) g = (c->ns_boundary == 1) ? g - c->ny_global : (c->ns_boundary == 2) ? -g : 0;   /* tripole: blocks.F90:205-208 */
        else if (g == c->ny_global && j > m->blk_jb[n]) m->blk_je[n] = j;
        jg[j - 1] = g;
      }
      for (int i = 1; i <= m->nxb; i++) {
        int g = is - nghost + i - 1;
        if (g < 1) g = (c->ew_boundary == 1) ? g + c->nx_global : 0;
        if (g > c->nx_global + nghost) g = 0;
        else if (g > c->nx_global) g = (c->ew_boundary == 1) ? g - c->nx_global : 0;
        else if (g == c->nx_global && i > m->blk_ib[n]) m->blk_ie[n] = i;
        ig[i - 1] = g;
      }
      n++;
    }
  }
  m->n2 = (size_t)m->nxb * m->nyb;
  m->n3 = m->n2 * m->km;
}

/* ------------------------------------------------------------------ */
/* halo update (non-tripole): rule of test/unit/halo/POP.F90Dipole:134-147,
 * 277-292 -- ghost = field at (iGlobal,jGlobal); fill value (0) where the
 * global index is 0 (closed boundary).  mpi/POP_HaloMod.F90:1895-1914.   */
/* ------------------------------------------------------------------ */
void orc_halo_update(orc_model *m, double *a, int nz, int unused) {
  (void)unused;
  const orc_config *c = &m->c;
  int nxb = m->nxb, nyb = m->nyb;
  size_t n2 = m->n2;
  for (int b = 0; b < m->nblocks; b++) {
    const int *ig = m->i_glob + (size_t)b * nxb, *jg = m->j_glob + (size_t)b * nyb;
    int ib = m->blk_ib[b], ie = m->blk_ie[b], jb = m->blk_jb[b], je = m->blk_je[b];
    for (int j = 1; j <= nyb; j++)
      for (int i = 1; i <= nxb; i++) {
        if (i >= ib && i <= ie && j >= jb && j <= je) continue;
        int gi = ig[i - 1], gj = jg[j - 1];
        for (int k = 0; k < nz; k++) {
          double v = 0.0;
          if (gi > 0 && gj > 0) {
            int sbx = (gi - 1) / c->block_size_x, sby = (gj - 1) / c->block_size_y;
            int sb = sby * m->nbx + sbx;
            int si = gi - sbx * c->block_size_x + 2, sj = gj - sby * c->block_size_y + 2;
            v = a[((size_t)sb * nz + k) * n2 + (size_t)(sj - 1) * nxb + (si - 1)];
          }
          a[((size_t)b * nz + k) * n2 + (size_t)(j - 1) * nxb + (i - 1)] = v;
        }
      }
  }
}
void orc_halo_update_int(orc_model *m, int *a) {
  const orc_config *c = &m->c;
  int nxb = m->nxb, nyb = m->nyb;
  size_t n2 = m->n2;
  for (int b = 0; b < m->nblocks; b++) {
    const int *ig = m->i_glob + (size_t)b * nxb, *jg = m->j_glob + (size_t)b * nyb;
    int ib = m->blk_ib[b], ie = m->blk_ie[b], jb = m->blk_jb[b], je = m->blk_je[b];
    for (int j = 1; j <= nyb; j++)
      for (int i = 1; i <= nxb; i++) {
        if (i >= ib && i <= ie && j >= jb && j <= je) continue;
        int gi = ig[i - 1], gj = jg[j - 1], v = 0;
        if (gi > 0 && gj > 0) {
          int sbx = (gi - 1) / c->block_size_x, sby = (gj - 1) / c->block_size_y;
          int sb = sby * m->nbx + sbx;
          int si = gi - sbx * c->block_size_x + 2, sj = gj - sby * c->block_size_y + 2;
          v = a[(size_t)sb * n2 + (size_t)(sj - 1) * nxb + (si - 1)];
        }
        a[(size_t)b * n2 + (size_t)(j - 1) * nxb + (i - 1)] = v;
      }
  }
}

/* ------------------------------------------------------------------ */
/* Tripole northern boundary of a halo update, as the reference does it (mpi/POP_HaloMod.F90:1936-2050 for
 * 2-D r8; :2280-2394 i4; same in the 3-D/4-D variants): the top haloWidth+1 physical rows of the whole
 * domain are gathered into a global buffer (POP_HaloMsgCreate :5830-5862: buffer row j <- block row
 * je-1-haloWidth+j), the degenerate top row is symmetrised for NE-corner and N-face fields, and every cell
 * of rows je..je+haloWidth of the northern blocks (E-W ghost columns included) is copied out of the mirrored
 * buffer address (:5864-5882: iSrc = nxGlobal-iGlobal(i)+1, jSrc = haloWidth+3-j) shifted by the location's
 * offsets, times the sign of the field kind.  loc: 0 centre, 1 NE corner, 2 N face, 3 E face;
 * kind: 0 scalar, 1 vector, 2 angle.  Call after the ordinary update (north ghosts of the top blocks carry
 * j_glob < 0 there and receive the fill value first).                                              */
#define ORC_TRIPOLE_BODY(TYPE, ABSF, AVG)                                                                       \
  const orc_config *c = &m->c;                                                                                \
  const int nxb = m->nxb, nyb = m->nyb, nx = c->nx_global, hw = 2;                                            \
  const size_t n2 = m->n2;                                                                                    \
  const int isign = (kind == 0) ? 1 : -1;                                                                     \
  int ioffset = 0, joffset = 0;                                                                               \
  if (loc == 1) { ioffset = 1; joffset = 1; } else if (loc == 3) { ioffset = 1; } else if (loc == 2) { joffset = 1; } \
  TYPE *buf = (TYPE *)calloc((size_t)(nx + 1) * (hw + 2), sizeof(TYPE));                                      \
  for (int k = 0; k < nz; k++) {                                                                              \
    for (int b = 0; b < m->nblocks; b++) {        /* copy in: northern blocks */                              \
      const int *ig = m->i_glob + (size_t)b * nxb, *jg = m->j_glob + (size_t)b * nyb;                         \
      const int ib = m->blk_ib[b], ie = m->blk_ie[b], je = m->blk_je[b];                                      \
      if (!(jg[je] < 0)) continue;                                                                            \
      for (int j = 1; j <= hw + 1; j++) for (int i = ib; i <= ie; i++)                                        \
        buf[(size_t)j * (nx + 1) + ig[i - 1]] = a[((size_t)b * nz + k) * n2 + (size_t)(je - 1 - hw + j - 1) * nxb + (i - 1)]; \
    }                                                                                                         \
    TYPE *top = buf + (size_t)(hw + 1) * (nx + 1);                                                            \
    if (loc == 1) {                                                                                           \
      for (int i = 1; i <= nx / 2; i++) {                                                                     \
        const int iDst = nx - i;                                                                              \
        const TYPE x1 = top[i], x2 = top[iDst];                                                               \
        const TYPE xavg = AVG(ABSF(x1), ABSF(x2));                                                            \
        top[i] = isign * ((x2 < 0) ? -xavg : xavg);                                                           \
        top[iDst] = isign * ((x1 < 0) ? -xavg : xavg);                                                        \
      }                                                                                                       \
      top[nx] = isign * top[nx];                                                                              \
    } else if (loc == 2) {                                                                                    \
      for (int i = 1; i <= nx / 2; i++) {                                                                     \
        const int iDst = nx + 1 - i;                                                                          \
        const TYPE x1 = top[i], x2 = top[iDst];                                                               \
        const TYPE xavg = AVG(ABSF(x1), ABSF(x2));                                                            \
        top[i] = isign * ((x2 < 0) ? -xavg : xavg);                                                           \
        top[iDst] = isign * ((x1 < 0) ? -xavg : xavg);                                                        \
      }                                                                                                       \
    }                                                                                                         \
    for (int b = 0; b < m->nblocks; b++) {        /* copy out */                                              \
      const int *ig = m->i_glob + (size_t)b * nxb, *jg = m->j_glob + (size_t)b * nyb;                         \
      const int ie = m->blk_ie[b], je = m->blk_je[b];                                                         \
      if (!(jg[je] < 0)) continue;                                                                            \
      for (int j = 1; j <= hw + 1; j++) for (int i = 1; i <= ie + hw; i++) {                                  \
        int iSrc = nx - ig[i - 1] + 1, jSrc = hw + 3 - j;                                                     \
        iSrc = iSrc - ioffset; jSrc = jSrc - joffset;                                                         \
        if (iSrc == 0) iSrc = nx;                                                                             \
        if (jSrc <= hw + 1)                                                                                   \
          a[((size_t)b * nz + k) * n2 + (size_t)(je + j - 1 - 1) * nxb + (i - 1)] = isign * buf[(size_t)jSrc * (nx + 1) + iSrc]; \
      }                                                                                                       \
    }                                                                                                         \
  }                                                                                                           \
  free(buf);
#define ORC_AVG_R8(x, y) (0.5 * ((x) + (y)))
#define ORC_AVG_I4(x, y) ((int)lround(0.5 * ((double)(x) + (double)(y))))
void orc_halo_update_tripole(orc_model *m, double *a, int nz, int loc, int kind) {
  orc_halo_update(m, a, nz, 0);
  ORC_TRIPOLE_BODY(double, fabs, ORC_AVG_R8)
}
void orc_halo_update_tripole_int(orc_model *m, int *a, int loc, int kind) {
  const int nz = 1;
  orc_halo_update_int(m, a);
  ORC_TRIPOLE_BODY(int, abs, ORC_AVG_I4)
}
/* POP_HaloUpdate with its fieldLoc / fieldKind arguments (mpi/POP_HaloMod.F90:1732-1773) */
void orc_halo(orc_model *m, double *a, int nz, int loc, int kind) {
  if (m->c.ns_boundary == 2) orc_halo_update_tripole(m, a, nz, loc, kind); else orc_halo_update(m, a, nz, 0);
}
void orc_halo_int(orc_model *m, int *a, int loc, int kind) {
  if (m->c.ns_boundary == 2) orc_halo_update_tripole_int(m, a, loc, kind); else orc_halo_update_int(m, a);
}

/* scatter_global (mpi/gather_scatter.F90:862-1161 r8, int alike): local (ghost cells included) <- global array;
 * cells with a zero global index keep 0; ghost rows beyond a tripole boundary (j_glob < 0) read the mirrored
 * address isrc = nx_global + xoffset - i_glob, jsrc = ny_global + yoffset + (j_glob + ny_global) with the
 * offsets of the field location (:929-945).  No sign factor is applied there (:1036, :1117).           */
#define ORC_SCATTER_BODY                                                                              \
  const orc_config *c = &m->c;                                                                        \
  const int nxb = m->nxb, nyb = m->nyb, nx = c->nx_global, ny = c->ny_global;                         \
  int xoffset = 1, yoffset = 1;                                                                       \
  if (loc == ORC_NECORNER) { xoffset = 0; yoffset = 0; }                                              \
  else if (loc == ORC_EFACE) { xoffset = 0; yoffset = 1; }                                            \
  else if (loc == ORC_NFACE) { xoffset = 1; yoffset = 0; }                                            \
  for (int b = 0; b < m->nblocks; b++) {                                                              \
    const int *ig = m->i_glob + (size_t)b * nxb, *jg = m->j_glob + (size_t)b * nyb;                   \
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++) {                                   \
      size_t p = (size_t)b * m->n2 + (size_t)(j - 1) * nxb + (i - 1);                                 \
      A[p] = 0;                                                                                       \
      if (ig[i - 1] == 0 || jg[j - 1] == 0) continue;                                                 \
      if (jg[j - 1] > 0) A[p] = G[(size_t)(jg[j - 1] - 1) * nx + ig[i - 1] - 1];                      \
      else {                                                                                          \
        int jsrc = ny + yoffset + (jg[j - 1] + ny), isrc = nx + xoffset - ig[i - 1];                  \
        if (isrc < 1) isrc = isrc + nx;                                                               \
        if (isrc > nx) isrc = isrc - nx;                                                              \
        A[p] = G[(size_t)(jsrc - 1) * nx + isrc - 1];                                                 \
      }                                                                                               \
    }                                                                                                 \
  }
static void scatter_global_r8(orc_model *m, double *A, const double *G, int loc) { ORC_SCATTER_BODY }
static void scatter_global_i4(orc_model *m, int *A, const int *G, int loc) { ORC_SCATTER_BODY }

/* ------------------------------------------------------------------ */
/* global sum: serial/POP_ReductionsMod.F90:200-300, serial/global_reductions.F90
 * :237-262 -- per block, j outer / i inner over the physical domain, block
 * sums added in block order.                                           */
/* ------------------------------------------------------------------ */
double orc_global_sum(orc_model *m, const double *a, const double *mask) {
  double g = 0.0;
  int nxb = m->nxb;
  for (int b = 0; b < m->nblocks; b++) {
    double s = 0.0;
    const double *A = a + (size_t)b * m->n2;
    const double *M = mask ? mask + (size_t)b * m->n2 : NULL;
    for (int j = m->blk_jb[b]; j <= m->blk_je[b]; j++)
      for (int i = m->blk_ib[b]; i <= m->blk_ie[b]; i++) {
        size_t p = (size_t)(j - 1) * nxb + (i - 1);
        if (M) s = s + A[p] * M[p]; else s = s + A[p];
      }
    g = g + s;
  }
  return g;
}

/* POP_GlobalSum on a tripole grid (mpi/POP_ReductionsMod.F90:308-341): for fields on north faces or NE
 * corners the top row holds every point twice; the points with iGlobal > nxGlobal/2 are taken out of the
 * block sum again (added first, then subtracted, as the reference does).  loc: 1 NE corner, 2 N face.   */
double orc_global_sum_tripole(orc_model *m, const double *a, const double *mask, int loc) {
  double g = 0.0;
  const int nxb = m->nxb, nyb = m->nyb;
  for (int b = 0; b < m->nblocks; b++) {
    double s = 0.0;
    const double *A = a + (size_t)b * m->n2;
    const double *M = mask ? mask + (size_t)b * m->n2 : NULL;
    for (int j = m->blk_jb[b]; j <= m->blk_je[b]; j++)
      for (int i = m->blk_ib[b]; i <= m->blk_ie[b]; i++) {
        size_t p = (size_t)(j - 1) * nxb + (i - 1);
        if (M) s = s + A[p] * M[p]; else s = s + A[p];
      }
    const int *ig = m->i_glob + (size_t)b * nxb, *jg = m->j_glob + (size_t)b * nyb;
    if (jg[m->blk_je[b]] < 0 && (loc == 1 || loc == 2)) {
      const int j = m->blk_je[b];
      for (int i = m->blk_ib[b]; i <= m->blk_ie[b]; i++)
        if (ig[i - 1] > m->c.nx_global / 2) {
          size_t p = (size_t)(j - 1) * nxb + (i - 1);
          if (M) s = s - A[p] * M[p]; else s = s - A[p];
        }
    }
    g = g + s;
  }
  return g;
}

/* ------------------------------------------------------------------ */
/* vertical grid: source/grid.F90:1549-1709 (vert_grid_internal,
 * compute_dz) and :786-803 (derived quantities)                        */
/* ------------------------------------------------------------------ */
static void compute_dz(orc_model *m, double *depth, double zlength, double dz_sfc, double dz_deep) {
  *depth = 0.0;
  for (int k = 1; k <= m->km; k++) {
    double r = *depth / zlength;
    m->dz[k] = dz_deep - (dz_deep - dz_sfc) * exp(-(r * r));
    *depth = *depth + m->dz[k];
  }
}
static void vert_grid(orc_model *m) {
  const double zmax = 5500.0, dz_sfc = 25.0, dz_deep = 400.0, eps = 1.0e-10;
  int km = m->km;
  double zl0 = eps, zl1 = zmax, dzl = zl1 - zl0, d0, d1, depth = 0, zlength;
  compute_dz(m, &d0, zl0, dz_sfc, dz_deep);
  compute_dz(m, &d1, zl1, dz_sfc, dz_deep);
  if ((d0 - zmax) * (d1 - zmax) > 0.0) { fprintf(stderr, "oracle vert_grid: no zero\n"); abort(); }
  while ((dzl / zmax) > eps) {
    zlength = zl0 + 0.5 * dzl;
    compute_dz(m, &depth, zlength, dz_sfc, dz_deep);
    if ((d0 - zmax) * (depth - zmax) < 0.0) { d1 = depth; zl1 = zlength; }
    else if ((d1 - zmax) * (depth - zmax) < 0.0) { d0 = depth; zl0 = zlength; }
    else { fprintf(stderr, "oracle vert_grid: not in interval\n"); abort(); }
    dzl = zl1 - zl0;
  }
  for (int k = 1; k <= km; k++) m->dz[k] = m->dz[k] * 100.0; /* cmperm */
  /* grid.F90:786-803 */
  m->dzw[0] = 0.5 * m->dz[1];
  m->dzw[km] = 0.5 * m->dz[km];
  m->dzwr[0] = 1.0 / m->dzw[0];
  m->zw[1] = m->dz[1];
  m->zt[1] = m->dzw[0];
  for (int k = 1; k <= km - 1; k++) {
    m->dzw[k] = 0.5 * (m->dz[k] + m->dz[k + 1]);
    m->zw[k + 1] = m->zw[k] + m->dz[k + 1];
    m->zt[k + 1] = m->zt[k] + m->dzw[k];
  }
  for (int k = 1; k <= km; k++) {
    m->c2dz[k] = 2.0 * m->dz[k];
    m->dzr[k] = 1.0 / m->dz[k];
    m->dz2r[k] = 1.0 / m->c2dz[k];
    m->dzwr[k] = 1.0 / m->dzw[k];
  }
}

/* ------------------------------------------------------------------ */
/* horizontal grid: source/grid.F90:1226-1297 (horiz_grid_internal),
 * :587-647 (closed-boundary extension + derived metrics), :2908-2928
 * (cf_area_avg), :1158-1159 (FCOR), topography :1957-2016, KMU :978-995,
 * HU/HUR :1024-1043, landmasks :2555-2591, calc_tpoints :2984-3062.    */
/* ------------------------------------------------------------------ */
static void horiz_grid(orc_model *m) {
  const orc_config *c = &m->c;
  int nxb = m->nxb, nyb = m->nyb;
  double pi = 4.0 * atan(1.0), radian = 180.0 / pi;
  int nxg = c->nx_global, nyg = c->ny_global;
  double dlon = 360.0 / (double)nxg, dlat = 180.0 / (double)nyg;
  double *ULAT_G = dalloc((size_t)nxg * nyg), *ULON_G = dalloc((size_t)nxg * nyg);
  int *KMT_G = ialloc((size_t)nxg * nyg);
  const orc_grid_input *gin = m->gin;
  if (gin) {
    memcpy(ULAT_G, gin->ULAT, (size_t)nxg * nyg * sizeof(double));
    memcpy(ULON_G, gin->ULON, (size_t)nxg * nyg * sizeof(double));
  } else {
  for (int i = 1; i <= nxg; i++) {
    double xdeg = i * dlon;
    if (xdeg > 180.0) xdeg = xdeg - 360.0;
    for (int j = 1; j <= nyg; j++) ULON_G[(size_t)(j - 1) * nxg + i - 1] = xdeg / radian;
  }
  for (int j = 1; j <= nyg; j++)
    for (int i = 1; i <= nxg; i++) ULAT_G[(size_t)(j - 1) * nxg + i - 1] = (-90.0 + j * dlat) / radian;
  }
  /* topography_internal (kmt_global branch) grid.F90:1957-1985 */
  for (int j = 1; j <= nyg; j++)
    for (int i = 1; i <= nxg; i++) {
      size_t p = (size_t)(j - 1) * nxg + i - 1;
      double latd = ULAT_G[p] * radian, lond = ULON_G[p] * radian;
      if (lond < 0.0) lond = lond + 360.0;
      int kmt = m->km;
      if (latd > -35.0 && lond > 210.0 && lond < 250.0) kmt = 0;
      if (latd > 25.0 && lond > 210.0 && lond < 330.0) kmt = 0;
      if (latd > 60.0 && lond > 210.0 && lond < 150.0) kmt = 0;
      if (latd > -60.0 && lond > 110.0 && lond < 150.0) kmt = 0;
      if (fabs(latd) > 75.0) kmt = 0;
      /* stepped_bathymetry = 1: stepped synthetic bathymetry (test extension; the reference's internal topography is flat) */
      if (kmt > 0 && m->c.stepped_bathymetry == 1) {
        int cut = ((i / 3) * 5 + (j / 2) * 3) % (m->km / 2 + 1);
        kmt = m->km - cut;
        if (kmt < 3) kmt = 3;
      }
      KMT_G[p] = kmt;
    }
  if (gin && gin->KMT) memcpy(KMT_G, gin->KMT, (size_t)nxg * nyg * sizeof(int));   /* read_topography :2062-2090 */
  if (gin) {
    /* read_horiz_grid grid.F90:1422-1536: records scattered with their field locations, the U/T spacings from
     * averaged HTN / HTE, non-positive lengths (closed-boundary ghosts) replaced by 1 */
    size_t ng = (size_t)nxg * nyg;
    double *W = dalloc(ng);
#define G2(A, i, j) (A)[(size_t)((j)-1) * nxg + (i)-1]
    scatter_global_r8(m, m->ULAT, ULAT_G, ORC_NECORNER);
    scatter_global_r8(m, m->ULON, ULON_G, ORC_NECORNER);
    scatter_global_r8(m, m->HTN, gin->HTN, ORC_NFACE);
    for (int j = 1; j <= nyg; j++) for (int i = 1; i <= nxg; i++) {
      int ip1 = i + 1; if (i == nxg) ip1 = 1;
      G2(W, i, j) = 0.5 * (G2(gin->HTN, i, j) + G2(gin->HTN, ip1, j));
    }
    scatter_global_r8(m, m->DXU, W, ORC_NECORNER);
    for (int j = 1; j <= nyg; j++) {
      int jm1 = j - 1; if (j == 1) jm1 = nyg;
      for (int i = 1; i <= nxg; i++) G2(W, i, j) = 0.5 * (G2(gin->HTN, i, j) + G2(gin->HTN, i, jm1));
    }
    scatter_global_r8(m, m->DXT, W, ORC_CENTER);
    scatter_global_r8(m, m->HTE, gin->HTE, ORC_EFACE);
    for (int j = 1; j <= nyg; j++) for (int i = 1; i <= nxg; i++) {
      int im1 = i - 1; if (i == 1) im1 = nxg;
      G2(W, i, j) = 0.5 * (G2(gin->HTE, i, j) + G2(gin->HTE, im1, j));
    }
    scatter_global_r8(m, m->DYT, W, ORC_CENTER);
    for (int j = 1; j <= nyg; j++) {
      int jp1 = j + 1; if (j == nyg) jp1 = 1;
      for (int i = 1; i <= nxg; i++) G2(W, i, j) = 0.5 * (G2(gin->HTE, i, j) + G2(gin->HTE, i, jp1));
    }
    if (c->ns_boundary == 2)   /* tripole-grid correction :1495-1500 */
      for (int i = 1; i <= nxg; i++) G2(W, i, nyg) = G2(gin->HTE, i, nyg);
    scatter_global_r8(m, m->DYU, W, ORC_NECORNER);
    scatter_global_r8(m, m->HUS, gin->HUS, ORC_EFACE);
    scatter_global_r8(m, m->HUW, gin->HUW, ORC_NFACE);
#undef G2
    free(W);
    double *pos[8] = {m->HTN, m->HTE, m->HUS, m->HUW, m->DXU, m->DYU, m->DXT, m->DYT};
    for (int q = 0; q < 8; q++)
      for (size_t p = 0; p < m->n2 * m->nblocks; p++) if (pos[q][p] <= 0.0) pos[q][p] = 1.0;
    scatter_global_i4(m, m->KMT, KMT_G, ORC_CENTER);
  }
  for (int b = 0; b < m->nblocks; b++) {
    const int *ig = m->i_glob + (size_t)b * nxb, *jgl = m->j_glob + (size_t)b * nyb;
    size_t o = (size_t)b * m->n2;
    double *HTN = m->HTN + o, *HTE = m->HTE + o, *HUS = m->HUS + o, *HUW = m->HUW + o;
    double *DXU = m->DXU + o, *DYU = m->DYU + o, *DXT = m->DXT + o, *DYT = m->DYT + o;
    double *ULAT = m->ULAT + o, *ULON = m->ULON + o;
    int *KMT = m->KMT + o;
    for (int j = 1; j <= nyb && !gin; j++) {
      int jg = jgl[j - 1], jm1 = jg - 1;
      if (jm1 < 1) jm1 = nyg;
      for (int i = 1; i <= nxb; i++) {
        size_t p = (size_t)(j - 1) * nxb + i - 1;
        HTN[p] = dlon * orc_radius / radian;
        HTE[p] = dlat * orc_radius / radian;
        HUS[p] = dlon * orc_radius / radian;
        HUW[p] = dlat * orc_radius / radian;
        DYT[p] = dlat * orc_radius / radian;
        DYU[p] = dlat * orc_radius / radian;
        int igl = ig[i - 1];
        if (igl > 0 && jg > 0) {
          ULON[p] = ULON_G[(size_t)(jg - 1) * nxg + igl - 1];
          ULAT[p] = ULAT_G[(size_t)(jg - 1) * nxg + igl - 1];
          HTN[p] = HTN[p] * cos(ULAT[p]);
          DXU[p] = HTN[p];
          double lathalf = (-90.0 + (jg - 0.5) * dlat) / radian;
          HUS[p] = HUS[p] * cos(lathalf);
          DXT[p] = dlon * orc_radius / radian * 0.5 *
                   (cos(ULAT_G[(size_t)(jg - 1) * nxg + igl - 1]) + cos(ULAT_G[(size_t)(jm1 - 1) * nxg + igl - 1]));
        } else {
          ULON[p] = 0.0; ULAT[p] = 0.0;
          HTN[p] = 1.0; HUS[p] = 1.0; DXU[p] = 1.0;
          /* DXT not set here in the reference (fixed up by the closed-boundary
             extension below); keep the allocation's zero. */
        }
        /* local KMT: grid.F90:1996-2011 */
        if (jg > 0 && igl != 0) KMT[p] = KMT_G[(size_t)(jg - 1) * nxg + igl - 1]; else KMT[p] = 0;
      }
    }
    /* closed boundary extension grid.F90:587-634 */
    int ib = m->blk_ib[b], ie = m->blk_ie[b], jb = m->blk_jb[b], je = m->blk_je[b];
#define P2(i, j) ((size_t)((j)-1) * nxb + (i)-1)
    if (ig[0] == 0)
      for (int j = 1; j <= nyb; j++) for (int i = 1; i <= ib - 1; i++) {
        DXU[P2(i, j)] = DXU[P2(ib, j)]; DYU[P2(i, j)] = DYU[P2(ib, j)];
        DXT[P2(i, j)] = DXT[P2(ib, j)]; DYT[P2(i, j)] = DYT[P2(ib, j)];
      }
    if (ig[ie + 1 - 1] == 0)
      for (int j = 1; j <= nyb; j++) for (int i = ie + 1; i <= nxb; i++) {
        DXU[P2(i, j)] = DXU[P2(ie, j)]; DYU[P2(i, j)] = DYU[P2(ie, j)];
        DXT[P2(i, j)] = DXT[P2(ie, j)]; DYT[P2(i, j)] = DYT[P2(ie, j)];
      }
    if (jgl[0] == 0)
      for (int j = 1; j <= jb - 1; j++) for (int i = 1; i <= nxb; i++) {
        DXU[P2(i, j)] = DXU[P2(i, jb)]; DYU[P2(i, j)] = DYU[P2(i, jb)];
        DXT[P2(i, j)] = DXT[P2(i, jb)]; DYT[P2(i, j)] = DYT[P2(i, jb)];
      }
    if (jgl[je + 1 - 1] == 0)
      for (int j = je + 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++) {
        DXU[P2(i, j)] = DXU[P2(i, je)]; DYU[P2(i, j)] = DYU[P2(i, je)];
        DXT[P2(i, j)] = DXT[P2(i, je)]; DYT[P2(i, j)] = DYT[P2(i, je)];
      }
    for (size_t p = 0; p < m->n2; p++) {
      m->DXUR[o + p] = 1.0 / DXU[p]; m->DYUR[o + p] = 1.0 / DYU[p];
      m->UAREA[o + p] = DXU[p] * DYU[p]; m->UAREA_R[o + p] = 1.0 / m->UAREA[o + p];
      m->DXTR[o + p] = 1.0 / DXT[p]; m->DYTR[o + p] = 1.0 / DYT[p];
      m->TAREA[o + p] = DXT[p] * DYT[p]; m->TAREA_R[o + p] = 1.0 / m->TAREA[o + p];
    }
    /* cf_area_avg grid.F90:2908-2928 */
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++) {
      size_t p = P2(i, j);
      const double *TA = m->TAREA + o;
      double au0 = TA[p], aun = esh(TA, nxb, nyb, i, j + 1), aue = esh(TA, nxb, nyb, i + 1, j);
      double aune = esh(TA, nxb, nyb, i + 1, j + 1);   /* eoshift(AUE,dim=2,+1) */
      m->AU0[o + p] = au0 * 0.25 * m->UAREA_R[o + p];
      m->AUN[o + p] = aun * 0.25 * m->UAREA_R[o + p];
      m->AUE[o + p] = aue * 0.25 * m->UAREA_R[o + p];
      m->AUNE[o + p] = aune * 0.25 * m->UAREA_R[o + p];
    }
    /* calc_tpoints grid.F90:2984-3062 (TLAT only needed for forcing/FCORT) */
    double *TLAT = m->TLAT + o;
    for (int j = 2; j <= nyb; j++) for (int i = 2; i <= nxb; i++) {
      double zsw = cos(ULAT[P2(i-1,j-1)]), xsw = cos(ULON[P2(i-1,j-1)]) * zsw, ysw = sin(ULON[P2(i-1,j-1)]) * zsw;
      zsw = sin(ULAT[P2(i-1,j-1)]);
      double zs = cos(ULAT[P2(i,j-1)]), xs = cos(ULON[P2(i,j-1)]) * zs, ys = sin(ULON[P2(i,j-1)]) * zs;
      zs = sin(ULAT[P2(i,j-1)]);
      double zw = cos(ULAT[P2(i-1,j)]), xw = cos(ULON[P2(i-1,j)]) * zw, yw = sin(ULON[P2(i-1,j)]) * zw;
      zw = sin(ULAT[P2(i-1,j)]);
      double zc = cos(ULAT[P2(i,j)]), xc = cos(ULON[P2(i,j)]) * zc, yc = sin(ULON[P2(i,j)]) * zc;
      zc = sin(ULAT[P2(i,j)]);
      double tx = 0.25 * (xc + xs + xw + xsw), ty = 0.25 * (yc + ys + yw + ysw), tz = 0.25 * (zc + zs + zw + zsw);
      double da = sqrt(tx * tx + ty * ty + tz * tz);
      TLAT[P2(i, j)] = asin(tz / da);
    }
    if (jgl[jb - 1] == 1)
      for (int i = ib; i <= ie; i++) TLAT[P2(i, jb)] = 2.0 * TLAT[P2(i, jb + 1)] - TLAT[P2(i, jb + 2)];
  }
  orc_halo(m, m->TLAT, 1, ORC_CENTER, ORC_SCALAR);
  /* partial bottom cells: read_bottom_cell (grid.F90:2116-2186) = the caller's record scattered as a centre scalar.  Without a
   * record (the reference always reads a file): a synthetic thickness in (0.25, 1] dz(KMT) -- TEST EXTENSION, same integer rule
   * as the library (host_setup.cpp) */
  if (c->partial_bottom_cells) {
    double *DZBC_G = dalloc((size_t)nxg * nyg);
    for (int j = 1; j <= nyg; j++)
      for (int i = 1; i <= nxg; i++) {
        size_t p = (size_t)(j - 1) * nxg + i - 1;
        if (gin && gin->DZBC) DZBC_G[p] = gin->DZBC[p];
        else DZBC_G[p] = KMT_G[p] > 0 ? (0.25 + 0.75 * (double)((i * 7 + j * 13) % 16 + 1) / 16.0) * m->dz[KMT_G[p]] : 0.0;
      }
    m->DZBC = dalloc(m->n2 * m->nblocks);
    scatter_global_r8(m, m->DZBC, DZBC_G, ORC_CENTER);
    free(DZBC_G);
  }
  free(ULAT_G); free(ULON_G); free(KMT_G);

  /* DZT, DZU (grid.F90:926-965) */
  if (c->partial_bottom_cells) {
    const int km = m->km;
    const size_t n2 = m->n2;
    m->DZT = dalloc((size_t)(km + 2) * n2 * m->nblocks);
    m->DZU = dalloc((size_t)(km + 2) * n2 * m->nblocks);
    for (int b = 0; b < m->nblocks; b++) {
      const int *KMT = m->KMT + (size_t)b * n2;
      const double *DZBC = m->DZBC + (size_t)b * n2;
      double *DZT = m->DZT + (size_t)b * (km + 2) * n2, *DZU = m->DZU + (size_t)b * (km + 2) * n2;
      for (int k = 1; k <= km; k++) {
        for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++) {
          size_t p = (size_t)(j - 1) * nxb + i - 1;
          DZT[(size_t)k * n2 + p] = (KMT[p] == k) ? DZBC[p] : m->dz[k];
        }
        for (int j = 1; j <= nyb - 1; j++) for (int i = 1; i <= nxb - 1; i++) {
          size_t p = (size_t)(j - 1) * nxb + i - 1;
          const double *D = DZT + (size_t)k * n2;
          double v = D[p];
          if (D[p + 1] < v) v = D[p + 1];
          if (D[p + nxb] < v) v = D[p + nxb];
          if (D[p + nxb + 1] < v) v = D[p + nxb + 1];
          DZU[(size_t)k * n2 + p] = v;
        }
      }
    }
    /* POP_HaloUpdate(DZU, ..., NECorner, scalar, fillValue = 0): all km+2 levels of every block (levels 0, km+1 are 0) */
    for (int b = 0; b < m->nblocks; b++) (void)b;
    {
      /* the halo routine works on (nx,ny,nz,blk) arrays with nz levels contiguous per block */
      orc_halo(m, m->DZU, km + 2, ORC_NECORNER, ORC_SCALAR);
    }
  }

  /* flat bottom: where (KMT /= 0) KMT = km (grid.F90:880-884) -- already km */
  /* KMU grid.F90:978-995 */
  for (int b = 0; b < m->nblocks; b++) {
    size_t o = (size_t)b * m->n2;
    int *KMT = m->KMT + o, *KMU = m->KMU + o;
    for (int j = 1; j <= nyb - 1; j++) for (int i = 1; i <= nxb - 1; i++) {
      int a = KMT[P2(i, j)], bb = KMT[P2(i + 1, j)], cc = KMT[P2(i, j + 1)], d = KMT[P2(i + 1, j + 1)];
      int mn = a < bb ? a : bb; if (cc < mn) mn = cc; if (d < mn) mn = d;
      KMU[P2(i, j)] = mn;
    }
  }
  orc_halo_int(m, m->KMU, ORC_NECORNER, ORC_SCALAR);
  for (int b = 0; b < m->nblocks; b++) {
    size_t o = (size_t)b * m->n2;
    int *KMT = m->KMT + o, *KMU = m->KMU + o;
    /* HT, HU, HUR grid.F90:1024-1043; with partial bottom cells :1001-1020 */
    if (c->partial_bottom_cells) {
      const size_t n2 = m->n2;
      const double *DZT = m->DZT + (size_t)b * (m->km + 2) * n2;
      double *DZU = m->DZU + (size_t)b * (m->km + 2) * n2;
      for (int k = 1; k <= m->km; k++)
        for (size_t p = 0; p < n2; p++) {
          if (k == KMT[p]) m->HT[o + p] = m->zw[k - 1] + DZT[(size_t)k * n2 + p];
          if (k == KMU[p]) { m->HU[o + p] = m->zw[k - 1] + DZU[(size_t)k * n2 + p]; m->HUR[o + p] = 1.0 / m->HU[o + p]; }
          else if (k > KMU[p]) DZU[(size_t)k * n2 + p] = m->dz[k];   /* to prevent divide by zero */
        }
    } else
    for (int k = 1; k <= m->km; k++)
      for (size_t p = 0; p < m->n2; p++) {
        if (k == KMT[p]) m->HT[o + p] = m->zw[k];
        if (k == KMU[p]) { m->HU[o + p] = m->zw[k]; m->HUR[o + p] = 1.0 / m->zw[k]; }
      }
    /* landmasks grid.F90:2555-2591 */
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++) {
      size_t p = P2(i, j);
      m->RCALCT[o + p] = (KMT[p] >= 1) ? 1.0 : 0.0;
      m->RCALCU[o + p] = (KMU[p] >= 1) ? 1.0 : 0.0;
      m->KMTN[o + p] = ieshf(KMT, nxb, nyb, i, j + 1);
      m->KMTS[o + p] = ieshf(KMT, nxb, nyb, i, j - 1);
      m->KMTE[o + p] = ieshf(KMT, nxb, nyb, i + 1, j);
      m->KMTW[o + p] = ieshf(KMT, nxb, nyb, i - 1, j);
      m->KMTEE[o + p] = ieshf(KMT, nxb, nyb, i + 2, j);
      m->KMTNN[o + p] = ieshf(KMT, nxb, nyb, i, j + 2);
    }
    /* FCOR grid.F90:1158-1159 */
    for (size_t p = 0; p < m->n2; p++) {
      m->FCOR[o + p] = 2.0 * orc_omega * sin(m->ULAT[o + p]);
      m->FCORT[o + p] = 2.0 * orc_omega * sin(m->TLAT[o + p]);
    }
  }
  /* uarea_equator grid.F90:1123-1135 (min UAREA at min |ULAT| over ocean U pts) */
  {
    double wmin = 1.0e300;
    for (int b = 0; b < m->nblocks; b++) {
      size_t o = (size_t)b * m->n2;
      for (int j = m->blk_jb[b]; j <= m->blk_je[b]; j++) for (int i = m->blk_ib[b]; i <= m->blk_ie[b]; i++) {
        size_t p = P2(i, j);
        if (m->KMU[o + p] >= 1 && fabs(m->ULAT[o + p]) < wmin) wmin = fabs(m->ULAT[o + p]);
      }
    }
    double amin = 1.0e300;
    for (int b = 0; b < m->nblocks; b++) {
      size_t o = (size_t)b * m->n2;
      for (int j = m->blk_jb[b]; j <= m->blk_je[b]; j++) for (int i = m->blk_ib[b]; i <= m->blk_ie[b]; i++) {
        size_t p = P2(i, j);
        if (m->KMU[o + p] >= 1) {
          double w = (fabs(m->ULAT[o + p]) == wmin) ? m->UAREA[o + p] : 1.e+20;
          if (w < amin) amin = w;
        }
      }
    }
    m->uarea_equator = amin;
  }
#undef P2
}

/* ------------------------------------------------------------------ */
/* equation of state (MWJF): source/state_mod.F90:394-498; pressure()
 * :1724-1771; ranges :1040-1062 (state_range_opt='enforce').           */
/* ------------------------------------------------------------------ */
static const double
  mwjfnp0s0t0 = 9.99843699e+2 * 0.001, mwjfnp0s0t1 = 7.35212840e+0 * 0.001,
  mwjfnp0s0t2 = -5.45928211e-2 * 0.001, mwjfnp0s0t3 = 3.98476704e-4 * 0.001,
  mwjfnp0s1t0 = 2.96938239e+0 * 0.001, mwjfnp0s1t1 = -7.23268813e-3 * 0.001,
  mwjfnp0s2t0 = 2.12382341e-3 * 0.001, mwjfnp1s0t0 = 1.04004591e-2 * 0.001,
  mwjfnp1s0t2 = 1.03970529e-7 * 0.001, mwjfnp1s1t0 = 5.18761880e-6 * 0.001,
  mwjfnp2s0t0 = -3.24041825e-8 * 0.001, mwjfnp2s0t2 = -1.23869360e-11 * 0.001;
static const double
  mwjfdp0s0t0 = 1.0e+0, mwjfdp0s0t1 = 7.28606739e-3, mwjfdp0s0t2 = -4.60835542e-5,
  mwjfdp0s0t3 = 3.68390573e-7, mwjfdp0s0t4 = 1.80809186e-10, mwjfdp0s1t0 = 2.14691708e-3,
  mwjfdp0s1t1 = -9.27062484e-6, mwjfdp0s1t3 = -1.78343643e-10, mwjfdp0sqt0 = 4.76534122e-6,
  mwjfdp0sqt2 = 1.63410736e-9, mwjfdp1s0t0 = 5.30848875e-6, mwjfdp2s0t3 = -3.03175128e-16,
  mwjfdp3s0t1 = -1.27934137e-17;

static inline void mwjf_point(double TK, double SK, double pbar, double tmin, double tmax,
                              double smin, double smax, double *rho, double *drdt, double *drds) {
  double TQ = TK < tmax ? TK : tmax; TQ = TQ > tmin ? TQ : tmin;
  double SQ = SK < smax ? SK : smax; SQ = SQ > smin ? SQ : smin;
  double p = 10.0 * pbar;
  SQ = 1000.0 * SQ;
  double SQR = sqrt(SQ);
  double n0 = mwjfnp0s0t0 + p * (mwjfnp1s0t0 + p * mwjfnp2s0t0);
  double n1 = mwjfnp0s0t1;
  double n2 = mwjfnp0s0t2 + p * (mwjfnp1s0t2 + p * mwjfnp2s0t2);
  double n3 = mwjfnp0s0t3;
  double ns1t0 = mwjfnp0s1t0 + p * mwjfnp1s1t0, ns1t1 = mwjfnp0s1t1, ns2t0 = mwjfnp0s2t0;
  double W1 = n0 + TQ * (n1 + TQ * (n2 + n3 * TQ)) + SQ * (ns1t0 + ns1t1 * TQ + ns2t0 * SQ);
  double d0 = mwjfdp0s0t0 + p * mwjfdp1s0t0;
  double d1 = mwjfdp0s0t1 + (p * p * p) * mwjfdp3s0t1;
  double d2 = mwjfdp0s0t2;
  double d3 = mwjfdp0s0t3 + (p * p) * mwjfdp2s0t3;
  double d4 = mwjfdp0s0t4;
  double ds1t0 = mwjfdp0s1t0, ds1t1 = mwjfdp0s1t1, ds1t3 = mwjfdp0s1t3;
  double dsqt0 = mwjfdp0sqt0, dsqt2 = mwjfdp0sqt2;
  double W2 = d0 + TQ * (d1 + TQ * (d2 + TQ * (d3 + d4 * TQ))) +
              SQ * (ds1t0 + TQ * (ds1t1 + TQ * TQ * ds1t3) + SQR * (dsqt0 + TQ * TQ * dsqt2));
  double DENOMK = 1.0 / W2;
  if (rho) *rho = W1 * DENOMK;
  if (drdt) {
    double W3 = n1 + TQ * (2.0 * n2 + 3.0 * n3 * TQ) + ns1t1 * SQ;
    double W4 = d1 + SQ * ds1t1 + TQ * (2.0 * (d2 + SQ * SQR * dsqt2) +
                TQ * (3.0 * (d3 + SQ * ds1t3) + TQ * 4.0 * d4));
    *drdt = (W3 - W1 * DENOMK * W4) * DENOMK;
  }
  if (drds) {
    double W3 = ns1t0 + ns1t1 * TQ + 2.0 * ns2t0 * SQ;
    double W4 = ds1t0 + TQ * (ds1t1 + TQ * TQ * ds1t3) + 1.5 * SQR * (dsqt0 + TQ * TQ * dsqt2);
    *drds = (W3 - W1 * DENOMK * W4) * DENOMK * 1000.0;
  }
}
static double pressure_fn(double depth) { /* state_mod.F90:1764-1765 */
  return 0.059808 * (exp(-0.025 * depth) - 1.0) + 0.100766 * depth + 2.28405e-7 * (depth * depth);
}
double orc_state_point(double T, double S_msu, double p_bar) {
  double r; mwjf_point(T, S_msu, p_bar, -2.0, 999.0, 0.0, 0.999, &r, NULL, NULL); return r;
}
/* state(k,kk,...) over npts points: density of water from level k displaced to kk */
void orc_state(orc_model *m, int k, int kk, const double *T, const double *S,
               double *rho, double *drhodt, double *drhods, int npts) {
  (void)k;
  double pz = m->pressz[kk];
  for (int p = 0; p < npts; p++)
    mwjf_point(T[p], S[p], pz, -2.0, 999.0, 0.0, 0.999, rho ? rho + p : NULL,
               drhodt ? drhodt + p : NULL, drhods ? drhods + p : NULL);
}

/* ------------------------------------------------------------------ */
/* horizontal mixing coefficients                                      */
/* del2: source/hmix_del2.F90:287-404 (init_del2u), :619-634 (init_del2t)
 * variable hmix: :223-262 (AMF), :560-590 (AHF)                        */
/* ------------------------------------------------------------------ */
static void init_del2(orc_model *m) {
  const orc_config *c = &m->c;
  int nxb = m->nxb, nyb = m->nyb;
  double pi = 4.0 * atan(1.0);
  size_t n2 = m->n2;
  double *W1 = dalloc(n2), *W2 = dalloc(n2), *KXU = dalloc(n2), *KYU = dalloc(n2);
  double *DXKX = dalloc(n2), *DYKY = dalloc(n2), *DXKY = dalloc(n2), *DYKX = dalloc(n2);
  /* AMF / AHF */
  for (size_t p = 0; p < n2 * m->nblocks; p++) { m->AMF[p] = 1.0; m->AHF[p] = 1.0; }
  if (c->lvariable_hmix && (c->hmix_momentum == 2 || c->hmix_tracer == 2)) {
    double ref = (2.0 * pi * orc_radius / c->nx_global); ref = ref * ref;
    for (size_t p = 0; p < n2 * m->nblocks; p++) {
      m->AMF[p] = sqrt(m->UAREA[p] / ref);
      m->AHF[p] = sqrt(m->TAREA[p] / ref);
    }
    orc_halo(m, m->AMF, 1, ORC_NECORNER, ORC_SCALAR);
    orc_halo(m, m->AHF, 1, ORC_CENTER, ORC_SCALAR);
  }
#define P2(i, j) ((size_t)((j)-1) * nxb + (i)-1)
#define E(A, i, j) esh(A, nxb, nyb, i, j)
  for (int b = 0; b < m->nblocks; b++) {
    size_t o = (size_t)b * n2;
    const double *HUS = m->HUS + o, *HUW = m->HUW + o, *HTN = m->HTN + o, *HTE = m->HTE + o;
    const double *AMF = m->AMF + o, *AHF = m->AHF + o, *UAR = m->UAREA_R + o, *TAR = m->TAREA_R + o;
    const double *DXUR = m->DXUR + o, *DYUR = m->DYUR + o;
    double *DUS = m->DUS + o, *DUN = m->DUN + o, *DUW = m->DUW + o, *DUE = m->DUE + o;
    /* momentum laplacian weights */
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      W1[P2(i, j)] = (HUS[P2(i, j)] / HTE[P2(i, j)]) * 0.5 * (AMF[P2(i, j)] + E(AMF, i, j - 1));
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++) {
      DUS[P2(i, j)] = W1[P2(i, j)] * UAR[P2(i, j)];
      DUN[P2(i, j)] = E(W1, i, j + 1) * UAR[P2(i, j)];
    }
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      W1[P2(i, j)] = (HUW[P2(i, j)] / HTN[P2(i, j)]) * 0.5 * (AMF[P2(i, j)] + E(AMF, i - 1, j));
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++) {
      DUW[P2(i, j)] = W1[P2(i, j)] * UAR[P2(i, j)];
      DUE[P2(i, j)] = E(W1, i + 1, j) * UAR[P2(i, j)];
    }
    /* metric terms */
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++) {
      KXU[P2(i, j)] = (E(HUW, i + 1, j) - HUW[P2(i, j)]) * UAR[P2(i, j)];
      KYU[P2(i, j)] = (E(HUS, i, j + 1) - HUS[P2(i, j)]) * UAR[P2(i, j)];
      W1[P2(i, j)] = (HTE[P2(i, j)] - E(HTE, i - 1, j)) * TAR[P2(i, j)]; /* KXT */
    }
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      W2[P2(i, j)] = 0.5 * (W1[P2(i, j)] + E(W1, i, j + 1)) * 0.5 * (E(AMF, i - 1, j) + AMF[P2(i, j)]);
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      DXKX[P2(i, j)] = (E(W2, i + 1, j) - W2[P2(i, j)]) * DXUR[P2(i, j)];
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      W2[P2(i, j)] = 0.5 * (W1[P2(i, j)] + E(W1, i + 1, j)) * 0.5 * (E(AMF, i, j - 1) + AMF[P2(i, j)]);
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      DYKX[P2(i, j)] = (E(W2, i, j + 1) - W2[P2(i, j)]) * DYUR[P2(i, j)];
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      W1[P2(i, j)] = (HTN[P2(i, j)] - E(HTN, i, j - 1)) * TAR[P2(i, j)]; /* KYT */
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      W2[P2(i, j)] = 0.5 * (W1[P2(i, j)] + E(W1, i + 1, j)) * 0.5 * (E(AMF, i, j - 1) + AMF[P2(i, j)]);
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      DYKY[P2(i, j)] = (E(W2, i, j + 1) - W2[P2(i, j)]) * DYUR[P2(i, j)];
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      W2[P2(i, j)] = 0.5 * (W1[P2(i, j)] + E(W1, i, j + 1)) * 0.5 * (E(AMF, i - 1, j) + AMF[P2(i, j)]);
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      DXKY[P2(i, j)] = (E(W2, i + 1, j) - W2[P2(i, j)]) * DXUR[P2(i, j)];
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++) {
      size_t p = P2(i, j);
      m->DUM[o + p] = -(DXKX[p] + DYKY[p] + 2.0 * AMF[p] * (KXU[p] * KXU[p] + KYU[p] * KYU[p]));
      m->DMC[o + p] = DXKY[p] - DYKX[p];
    }
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      W1[P2(i, j)] = (E(AMF, i, j + 1) - E(AMF, i, j - 1)) / (HTE[P2(i, j)] + E(HTE, i, j + 1));
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      m->DME[o + P2(i, j)] = (2.0 * AMF[P2(i, j)] * KYU[P2(i, j)] + W1[P2(i, j)]) / (HTN[P2(i, j)] + E(HTN, i + 1, j));
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      W1[P2(i, j)] = (E(AMF, i + 1, j) - E(AMF, i - 1, j)) / (HTN[P2(i, j)] + E(HTN, i + 1, j));
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      m->DMN[o + P2(i, j)] = -(2.0 * AMF[P2(i, j)] * KXU[P2(i, j)] + W1[P2(i, j)]) / (HTE[P2(i, j)] + E(HTE, i, j + 1));
    for (size_t p = 0; p < n2; p++) {
      m->DUC[o + p] = -(DUN[p] + DUS[p] + DUE[p] + DUW[p]);
      m->DMW[o + p] = -m->DME[o + p];
      m->DMS[o + p] = -m->DMN[o + p];
    }
    /* tracer laplacian weights hmix_del2.F90:619-634 */
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      W1[P2(i, j)] = (HTN[P2(i, j)] / HUW[P2(i, j)]) * 0.5 * (AHF[P2(i, j)] + E(AHF, i, j + 1));
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++) {
      m->DTN[o + P2(i, j)] = W1[P2(i, j)] * TAR[P2(i, j)];
      m->DTS[o + P2(i, j)] = E(W1, i, j - 1) * TAR[P2(i, j)];
    }
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++)
      W1[P2(i, j)] = (HTE[P2(i, j)] / HUS[P2(i, j)]) * 0.5 * (AHF[P2(i, j)] + E(AHF, i + 1, j));
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++) {
      m->DTE[o + P2(i, j)] = W1[P2(i, j)] * TAR[P2(i, j)];
      m->DTW[o + P2(i, j)] = E(W1, i - 1, j) * TAR[P2(i, j)];
    }
    /* advection metric coefficients advection.F90:387-396 */
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++) {
      m->KXU[o + P2(i, j)] = (E(HUW, i + 1, j) - HUW[P2(i, j)]) * UAR[P2(i, j)];
      m->KYU[o + P2(i, j)] = (E(HUS, i, j + 1) - HUS[P2(i, j)]) * UAR[P2(i, j)];
    }
  }
#undef E
#undef P2
  free(W1); free(W2); free(KXU); free(KYU); free(DXKX); free(DYKY); free(DXKY); free(DYKX);
}

/* ------------------------------------------------------------------ */
/* solver operator coefficients: source/POP_SolversMod.F90:771-822,
 * residual norm :895-906; checkerboard null space barotropic.F90:150-205 */
/* ------------------------------------------------------------------ */
static void init_solver(orc_model *m) {
  int nxb = m->nxb, nyb = m->nyb;
  size_t n2 = m->n2;
  double *work0 = dalloc(n2 * m->nblocks);
#define P2(i, j) ((size_t)((j)-1) * nxb + (i)-1)
  for (int b = 0; b < m->nblocks; b++) {
    size_t o = (size_t)b * n2;
    const double *HU = m->HU + o, *DXUR = m->DXUR + o, *DYUR = m->DYUR + o, *DXU = m->DXU + o, *DYU = m->DYU + o;
    for (int j = 2; j <= nyb; j++) for (int i = 2; i <= nxb; i++) {
      double xne = 0.25 * HU[P2(i, j)] * DXUR[P2(i, j)] * DYU[P2(i, j)];
      double xse = 0.25 * HU[P2(i, j - 1)] * DXUR[P2(i, j - 1)] * DYU[P2(i, j - 1)];
      double xnw = 0.25 * HU[P2(i - 1, j)] * DXUR[P2(i - 1, j)] * DYU[P2(i - 1, j)];
      double xsw = 0.25 * HU[P2(i - 1, j - 1)] * DXUR[P2(i - 1, j - 1)] * DYU[P2(i - 1, j - 1)];
      double yne = 0.25 * HU[P2(i, j)] * DYUR[P2(i, j)] * DXU[P2(i, j)];
      double yse = 0.25 * HU[P2(i, j - 1)] * DYUR[P2(i, j - 1)] * DXU[P2(i, j - 1)];
      double ynw = 0.25 * HU[P2(i - 1, j)] * DYUR[P2(i - 1, j)] * DXU[P2(i - 1, j)];
      double ysw = 0.25 * HU[P2(i - 1, j - 1)] * DYUR[P2(i - 1, j - 1)] * DXU[P2(i - 1, j - 1)];
      size_t p = o + P2(i, j);
      m->btropWgtNE[p] = xne + yne;
      double ase = xse + yse, anw = xnw + ynw, asw = xsw + ysw;
      m->btropWgtEast[p] = xne + xse - yne - yse;
      m->btropWgtNorth[p] = yne + ynw - xne - xnw;
      m->centerWgtIndep[p] = -(m->btropWgtNE[p] + ase + anw + asw);
      work0[p] = m->TAREA[p] * m->TAREA[p];
      m->mMask[p] = m->RCALCT[p];
    }
  }
  m->residualNorm = 1.0 / orc_global_sum(m, work0, m->mMask);
  m->convergenceCriterion = (m->c.convergence_criterion * m->c.convergence_criterion) / m->residualNorm;
  free(work0);
  /* init_barotropic: barotropic.F90:150-205 */
  double *CA = dalloc(n2 * m->nblocks), *KA = dalloc(n2 * m->nblocks);
  for (int b = 0; b < m->nblocks; b++) {
    size_t o = (size_t)b * n2;
    const int *ig = m->i_glob + (size_t)b * nxb, *jg = m->j_glob + (size_t)b * nyb;
    for (int j = 1; j <= nyb; j++) for (int i = 1; i <= nxb; i++) {
      size_t p = o + P2(i, j);
      int n = ig[i - 1] + abs(jg[j - 1]);
      m->CHECKER[p] = (double)(2 * (n % 2) - 1);
      CA[p] = m->CHECKER[p] * m->TAREA[p];
      if (m->KMT[p] > 0) { m->CONSTNT[p] = 1.0; KA[p] = m->TAREA[p]; }
      else { m->CHECKER[p] = 0.0; m->CONSTNT[p] = 0.0; CA[p] = 0.0; KA[p] = 0.0; }
    }
  }
  double sum_check = orc_global_sum(m, m->CHECKER, NULL);
  double sum_const = orc_global_sum(m, m->CONSTNT, NULL);
  double acheck = orc_global_sum(m, CA, NULL) / orc_global_sum(m, KA, NULL);
  m->rcheck = acheck / (sum_const - acheck * sum_check);
  m->rconst = 1.0 / (sum_const - acheck * sum_check);
  free(CA); free(KA);
#undef P2
}

/* ------------------------------------------------------------------ */
/* time stepping parameters: source/time_management.F90:753-790, 800-858,
 * 950-1005                                                            */
/* ------------------------------------------------------------------ */
static void init_time(orc_model *m) {
  const orc_config *c = &m->c;
  double seconds_in_day = 86400.0;
  double steps_per_day = (double)c->steps_per_day;
  m->dtt = seconds_in_day / steps_per_day;
  m->nsteps_per_interval = c->steps_per_day;
  if (c->tmix_opt == 2) { /* avgfit, fit_freq = 1 */
    int tmf = c->time_mix_freq;
    int full = c->steps_per_day / 1;
    if (full < 1) full = 1;
    int half = (tmf + full) / (tmf - 1);
    int nsteps = full + half;
    if (nsteps % tmf == 0) { full = full + 1; half = (tmf + full) / (tmf - 1); nsteps = full + half; }
    if (full == 1 && half == 1) { full = full + 1; nsteps = full + half; }
    m->nsteps_per_interval = nsteps;
    m->dtt = seconds_in_day / (full + 0.5 * half);
  }
  m->dtu = m->dtt; m->dtp = m->dtt;
  for (int k = 1; k <= m->km; k++) m->dt[k] = m->dtt * 1.0;
  m->first_step = 1; m->nsteps_total = 0; m->nsteps_this_interval = 0;
  m->oldtime = 0; m->curtime = 1; m->newtime = 2; m->mixtime = 1;
}

/* time_manager + set_switches: time_management.F90:1823-1847, 2118-2225 */
void orc_time_manager(orc_model *m) {
  const orc_config *c = &m->c;
  m->leapfrogts = 1; m->f_euler_ts = 0; m->avg_ts = 0;
  m->nsteps_total = m->nsteps_total + 1;
  if (c->tmix_opt == 2) {
    m->nsteps_this_interval = m->nsteps_this_interval + 1;
    if (m->nsteps_this_interval > m->nsteps_per_interval) m->nsteps_this_interval = 1;
  }
  /* end of day: with avgfit the fit interval is one day; without averaging steps every steps_per_day-th step ends at midnight */
  m->eod_last = m->eod;
  m->eod = (c->tmix_opt == 2) ? (m->nsteps_this_interval == m->nsteps_per_interval) : (m->nsteps_total % c->steps_per_day == 0);
  if (m->first_step) { m->leapfrogts = 0; m->f_euler_ts = 1; m->first_step = 0; }
  if (c->tmix_opt == 1) {
    if (m->nsteps_total % c->time_mix_freq == 0) m->avg_ts = 1;
  }
  if (c->tmix_opt == 2) {
    int n = m->nsteps_this_interval;
    if (n == 1) { }
    else if (n == 2) m->avg_ts = 1;
    else if ((n + 1) % c->time_mix_freq == 0) { }
    else if (n % c->time_mix_freq == 0) m->avg_ts = 1;
  }
}

#include "orc_dyn.inc"
#include "orc_api.inc"
