// pop_internal.hpp -- internal types of libpop_amd (MI355X-native POP2 dynamics core).
//
// Data layout on the device (DESIGN.md "Layout"): every field is stored per rank as
// (i, j, k, local_block) with i fastest -- the reference's own block layout
// (source/prognostic.F90:47-66) -- so one wavefront reads 64 consecutive i of one level
// (512 B, fully coalesced) and a thread owns a water column and marches k in registers.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "../../include/pop_amd.h"

namespace pop {

constexpr int NGHOST = 2;          // blocks.F90:51-56
constexpr double GRAV = 980.6;     // pop_constants.F90:235 (non-CCSMCOUPLED)
constexpr double OMEGA = 7.292123625e-5;
constexpr double RADIUS = 6370.0e5;
constexpr int MAXNT = 8;

struct BlockInfo {                 // blocks.F90:30-39 type(block)
  int block_id, local_id, ib, ie, jb, je, iblock, jblock;
  std::vector<int> i_glob, j_glob;
};

// ---- halo plan (mpi/POP_HaloMod.F90:142-1640 POP_HaloCreate, restated as flat index lists)
struct PeerPlan {
  int rank;
  std::vector<int> send_src;   // local 2-D cell index (block*n2 + j*nxb + i) to pack, in message order
  std::vector<int> recv_dst;   // local 2-D cell index to unpack into, in message order
};
// Tripole northern boundary (mpi/POP_HaloMod.F90:1936-2050), per field location: every cell of rows je..je+nghost
// of the northern blocks is a function of one or two physical cells of the global top rows.
//   b < 0 : dst = isign * F[a]                                   (mirrored copy; isign = -1 for vector / angle kinds)
//   b >= 0: dst = sign(0.5*(|F[a]| + |F[b]|), F[a])              (symmetrised degenerate top row, NE-corner / N-face fields)
// a, b are cells of this rank (single-rank decompositions only).
struct TripolePlan { std::vector<int> dst, a, b; };
struct HaloPlan {
  std::vector<int> copy_dst, copy_src;  // ghost <- interior copies between blocks of this rank
  std::vector<int> fill_dst;            // ghosts outside closed boundaries / padding: fill value
  std::vector<PeerPlan> peers;
  long long max_msg_cells = 0;          // sum over peers of max(send, recv) cells
  TripolePlan tripole[5];               // by location: 0 centre, 1 NE corner, 2 N face, 3 E face (ns_boundary = 2); cells numbered over the rank's blocks.
                                        // [4]: N face, ghost rows only (no symmetrised top row): fields the reference forms locally in the ghost rows
  TripolePlan tripole_g[5];             // the same with cells numbered over ALL blocks (host set-up arrays)
  bool tripole_split = false;           // the top row of blocks has more than one owner: no plan
};

// ---- EVP block preconditioner (preconditioner_choice = 1; POP_SolversMod.F90:252-290, 2434-2696): sub-blocks of at most
//      EVP_BS x EVP_BS cells with a one-cell rim, for every block of the decomposition
constexpr int EVP_BS = 8, EVP_LD = EVP_BS + 2, EVP_LE = 2 * EVP_BS - 1;
struct EvpHost {
  int xnb = 0, ynb = 0;                     // sub-blocks per block in x, y
  std::vector<int> xidx, yidx;              // 1-based start indices xidx[1..xnb+1] (EvpBlockPartition)
  // sub-block s = (block*ynb + j-1)*xnb + i-1; (a,c) of the sub-block with rim at (a-1) + EVP_LD*(c-1)
  std::vector<double> cc, ne, icc, ine;     // centre / NE weight and their inverses
  std::vector<double> rinv;                 // inverse influence matrix, (k,j) at (k-1) + EVP_LE*(j-1)
  std::vector<int> land;                    // 1: diagonal scaling instead of the EVP solve
};

// ---- host-side model: everything init-time (restates grid.F90, hmix_del*.F90 init,
//      POP_SolversInit, init_barotropic, init_ts) on the local blocks of this rank
// short-wave absorption tables on the device (sw_absorption.F90): sw_absorb(0:km) of 'top-layer' / 'jerlov' (:355-370); for
// 'chlorophyll' the transmission table Tr over the levels ztr and 401 chlorophyll amounts (:525-728) and the table column of
// every cell (set_chl :500-512).  Shared by KPP (lshort_wave) and the temperature source add_sw_absorb.
struct SwTab {
  double *swabs = nullptr, *Tr = nullptr, *ztr = nullptr;
  int *CHLI = nullptr;
  int ksol = 0;
  double chlmin = 0, chlmax = 0, dlogchl = 0;
};
// tuning switches (include/pop_amd.h pop_tuning): resolved once at pop_create -- size rule < caller's struct < POP_* environment
inline bool tun_set(int v) { return v != POP_TUNING_UNSET; }
inline bool tun_on(int v) { return v != POP_TUNING_UNSET && v != 0; }
inline bool tun_off(int v) { return v == 0; }
inline int tun_or(int v, int dflt) { return v != POP_TUNING_UNSET ? v : dflt; }
struct HostModel {
  SwTab sw;
  pop_config c{};
  pop_tuning tun{};
  const pop_grid_input *gin = nullptr;     // caller's grid (pop_create_with_grid); read during host_build only
  int rank = 0, nranks = 1;
  bool plan_only = false;                  // POP_CREATE_PLAN_ONLY: blocks, distribution and halo plan, no fields
  long long ocean_cols_local = -1, ocean_cols_total = -1;   // ocean columns of the physical domain (global KMT rule / record), make_blocks
  int nxb = 0, nyb = 0, km = 0, nt = 2;
  int nbx = 0, nby = 0, nblocks_tot = 0, nblocks = 0;   // nblocks = local
  size_t n2 = 0, n3 = 0;
  std::vector<BlockInfo> all_blocks;      // every block of the decomposition (1-based id = index+1)
  std::vector<int> local_ids;             // global ids (1-based) of this rank's blocks
  std::vector<int> block_owner;           // rank owning block id-1
  std::vector<int> block_local;           // local index of block id-1 on its owner
  // vertical grid, 1-based with slot 0 (dzw(0), dzwr(0))
  std::vector<double> dz, dzw, zt, zw, c2dz, dzr, dz2r, dzwr, pressz, bouss, dt, afac_t, afac_u;
  std::vector<double> upw_z[6];           // upwind3 vertical weights talfzp,tbetzp,tgamzp,talfzm,tbetzm,tdelzm (1..km)
  // named 2-D / 3-D fields on local blocks
  std::map<std::string, std::vector<double>> f2;   // (nxb,nyb,nblocks)
  std::map<std::string, std::vector<int>> i2;
  std::map<std::string, std::vector<double>> f3;   // (nxb,nyb,km,nblocks), initial state only
  double uarea_equator = 0, residualNorm = 0, convergenceCriterion = 0, rcheck = 0, rconst = 0;
  double dtt = 0, dtu = 0, dtp = 0;
  int nsteps_per_interval = 0;
  // Robert filter (tmix_opt = 3): time_management.F90:897-945, step_mod.F90:1577-1615
  double robert_curtime = 0, robert_newtime = 0, bgtarea_t_1 = 0, rf_volume_2_km = 0, open_ocean_volume_2_km = 0;
  int rf_nonzero_newtime = 0;
  // P-CSI (solver_choice = 3): Lanczos eigenvalue bounds from host_pcsi_prep (POP_SolversMod.F90:181-320)
  double pcsi_max_eig = 0, pcsi_min_eig = 0;
  int pcsi_lanczos_steps = 0;
  EvpHost evp;
  HaloPlan halo;
  std::string err;

  std::vector<double> &F(const std::string &n) { return f2[n]; }
  std::vector<int> &I(const std::string &n) { return i2[n]; }
};

int host_build(HostModel &h);            // host_setup.cpp
// ---- POP binary restart files (host_restart.cpp; restart.F90, io_binary.F90)
struct RestartField {
  std::string name, dev; int tl, n, ndims, id;       // file name; device field, time level (0 old, 1 cur), tracer; first record
  std::string long_name, units, grid_loc;
  int mask;                                          // 1 CALCU, 2 CALCT, 3 k > KMU, 4 k > KMT (read_restart :881-935)
};
struct RestartAttr { std::string name, type, value; };
std::vector<RestartField> restart_fields(const HostModel &h);
int restart_write_header(const HostModel &h, const std::string &path, const std::vector<RestartAttr> &attrs,
                         const std::vector<RestartField> &fields, std::string &err);
int restart_parse_header(const std::string &path, std::map<std::string, std::map<std::string, std::string>> &sec, std::string &err);
int restart_slab_io(const HostModel &h, int fd, long long rec, double *loc, size_t blk_stride, bool write, bool swap, std::string &err);
void restart_mask(const HostModel &h, const RestartField &f, double *loc, const std::vector<int> &KMT, const std::vector<int> &KMU);
std::vector<double> host_center_init(HostModel &h);                      // centre weight POP_SolversPrep sees (host_pcsi.cpp)
int host_pcsi_prep(HostModel &h, const std::vector<double> &C);         // host_pcsi.cpp
int host_evp_prep(HostModel &h, const std::vector<double> &C);          // host_evp.cpp
void host_evp_apply(const HostModel &h, double *PX, const double *X);   // all blocks
inline bool use_evp(const pop_config &c) { return c.preconditioner_choice == 1; }
void build_halo_plan(HostModel &h);      // halo_plan.cpp
void host_halo_r8(const HostModel &h, double *a, int nz, double fill);   // single-rank host halo
void host_halo_i4(const HostModel &h, int *a, int nz, int fill);
// with field location / kind: the ordinary update, then the tripole pass when ns_boundary = 2
void host_halo_r8_loc(const HostModel &h, double *a, int nz, double fill, int loc, int kind);
void host_halo_i4_loc(const HostModel &h, int *a, int nz, int fill, int loc, int kind);
double host_global_sum(const HostModel &h, const double *a, const double *mask);
double host_global_sum_loc(const HostModel &h, const double *a, const double *mask, int loc);
std::vector<int> global_srcmap(const HostModel &h);   // all blocks: ghost -> source cell, -1 = fill

}  // namespace pop
