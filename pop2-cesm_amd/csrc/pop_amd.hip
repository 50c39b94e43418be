// pop_amd.hip -- context, launch orchestration (the step_mod.F90 call sequence) and the C ABI of
// libpop_amd.so.  gfx950 only.  See include/pop_amd.h for the boundary and DESIGN.md for the
// kernel list.  The product path has no CPU fallback: every compute entry point fails loudly
// when the context was created host-only or no HIP device is usable.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <unordered_map>
#include "kernels_baroclinic.hpp"
#include "kernels_barotropic.hpp"
#include "kernels_mix.hpp"
#include "kernels_thomas_reg.hpp"
#include "kernels_momentum_lds.hpp"
#include "kernels_tracer_lds.hpp"
#include "kernels_rf.hpp"
#include "kernels_pcsi.hpp"
#include "kernels_evp.hpp"
#include "kernels_lwlim.hpp"
#include "kernels_gm.hpp"
#include "kernels_pcg_persist.hpp"
#include <fcntl.h>
#include <unistd.h>
#include "rccl_transport.hpp"

using namespace pop;

#define HIPCHK(ctx, call)                                                                       \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
      return 1;                                                                                 \
    }                                                                                           \
  } while (0)

struct PhaseTimer { double ms = 0; int calls = 0; };

struct DevPeer { int rank; int *send_src = nullptr, *recv_dst = nullptr; int nsend = 0, nrecv = 0; };

struct pop_ctx {
  HostModel h;
  bool host_only = true;
  std::string err;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  DevGrid g{};
  std::map<std::string, double *> d2;     // device 2-D fields (local blocks)
  std::map<std::string, int *> di2;
  std::vector<void *> allocs;
  // prognostic state, physical slots; logical levels via oldt/curt/newt
  double *TR[MAXNT][3] = {}, *U[3] = {}, *V[3] = {}, *RHO[3] = {};
  double *PS[3] = {}, *GX[3] = {}, *GY[3] = {}, *UB[3] = {}, *VB[3] = {};
  double *PGUESS = nullptr, *FW = nullptr, *FW_OLD = nullptr, *SHF_QSW = nullptr, *CHL = nullptr;
  double *STF[MAXNT] = {}, *TFW[MAXNT] = {}, *KPP_SRC[MAXNT] = {}, *VDC[2] = {}, *VVC = nullptr;
  double *DH = nullptr, *DHU = nullptr, *ZX = nullptr, *ZY = nullptr, *UH = nullptr, *VH = nullptr;
  double *W3 = nullptr, *W4 = nullptr, *RHS = nullptr, *centerWgt = nullptr;
  double *E3 = nullptr, *F3 = nullptr, *S3a = nullptr, *S3b = nullptr, *S3c = nullptr, *S3d = nullptr;
  // del4: the first Laplacians need only the mix-time fields, so they run on a side stream beside the vertical-mixing
  // coefficients (own output buffers d2t / d2u instead of the shared scratch; POP_DEL4_SIDE=0: in line, scratch reused)
  double *d2t[2] = {nullptr, nullptr}, *d2u[2] = {nullptr, nullptr};
  // del4: the tracer kernel of a step also forms the first Laplacian of its CURRENT tracers -- the mix-time field of the next
  // (leapfrog) step -- from the tile it has in LDS; d2t_next receives it, d2t_next_slot is the time slot it belongs to
  double *d2t_next[2] = {nullptr, nullptr};
  bool d2t_next_valid = false; int d2t_next_slot = -1;
  // the ghost cells of the tracers in a time slot are copies of their source cells (set-up and every halo update leave them so;
  // a caller's pop_set_field / a restart file may not): only then is the halo update of the field formed ahead the same arithmetic
  // as k_del4_d2t on the ghost ring
  bool tr_ghosts_ok[3] = {true, true, true};
  // the same for the velocity (k_momentum_rhs_lds forms k_del4_d2u's field for the next step)
  double *d2u_next[2] = {nullptr, nullptr};
  bool d2u_next_valid = false; int d2u_next_slot = -1;
  bool uv_ghosts_ok[3] = {true, true, true};
  bool d2t_last_formed = false, d2u_last_formed = false;   // did the last tracer / momentum launch write the next step's field (bench accounting)
  hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_d2t = nullptr, ev_d2u = nullptr, ev_vmixu = nullptr;
  bool side_del4 = false, vmixu_pending = false, btrop_added = false, vmixu_deferred = false;   // implicit vertical mixing of U,V in flight on the side stream
  double *HBLT = nullptr, *HMXL = nullptr, *HMXL_DR = nullptr;
  MixDev mix{};
  // KPP look-ahead: the vertical-mixing coefficients of the NEXT step depend only on this step's curtime fields (its
  // mixtime on a leapfrog step), so pop_step computes them on a third stream beside the barotropic solver (VALU-bound
  // work beside bandwidth-bound work) into a second set of output fields; the next step swaps the sets in.
  bool vdc_shared = false;
  bool kpp_src_user = false;   // the caller wrote KPP_SRC (pop_set_field): read it at every level until KPP has run again
  bool src_dirty = true, src_dirty_alt = true;   // KPP_SRC / KPPa may hold non-zeros below the KBL stored with them: the next evaluation into that set clears every level
  double *VDCa[2] = {nullptr, nullptr}, *VVCa = nullptr, *KPPa[MAXNT] = {}, *HBLTa = nullptr, *HMXLa = nullptr, *HMXL_DRa = nullptr;
  int *KBL = nullptr, *KBLa = nullptr;   // KBL that belongs to KPP_SRC / KPPa (the tracer kernel reads KPP_SRC down to it)
  hipStream_t ahead = nullptr; hipEvent_t ev_ahead_fork = nullptr, ev_ahead = nullptr;
  bool ahead_enabled = false, ahead_valid = false; int ahead_slot = -1;
  // solver
  double *R = nullptr, *S0 = nullptr, *S1 = nullptr, *Q = nullptr, *Z = nullptr, *AZ = nullptr;
  double *partial = nullptr, *blocksum = nullptr;
  SolverScalars *sc = nullptr;
  int *gid = nullptr, *srcmap = nullptr, *iota = nullptr, *loc_of_gid = nullptr;
  std::vector<int> opre_host;
  int *red_act = nullptr, *red_cnt = nullptr; int red_nact = 0;                // chunks with an ocean cell (fused solver launches, land elimination)
  SolverScalars *host_sc = nullptr;                       // pinned
  double *host_rr = nullptr;                              // pinned ring of (r,r) check results (k_rr_total)
  hipEvent_t chk_ev[4] = {};                              // one event per check interval in flight
  std::vector<std::pair<double *, hipGraphExec_t>> graphs;  // fused-solver interval graphs, keyed by solution array
  bool no_graph = false, fused_ok = false, evp_fused_ok = false, replicated = false, grid_from_input = false;
  // land elimination: the first land_full_steps steps after set-up / a restart / a new state run every workgroup (they write
  // the state-independent values of the land tiles), later steps skip workgroups without an ocean cell (DevGrid::skip)
  bool land_skip = true; int land_full_steps = 4, full_left = 4, full_seen = 0; double land_fraction = 0.0;
  bool fpcg_one_cell = false;   // POP_FPCG_B2=0: one cell per thread in step B of the fused pcg even on large grids
  bool force_presum = false;
  // the resident pcg of small grids (kernels_pcg_persist.hpp; pop_tuning.pcg_persist): one plan per solver view
  struct PersistPlan {
    const void *key = nullptr; bool ok = false; std::string why;
    int CP = 0, nwg = 0, nwin_max = 0, nslots = 0;
    int *own_q = nullptr, *halo_off = nullptr, *halo_q = nullptr; unsigned short *nbr = nullptr;
    PWord *W = nullptr;                                    // [2][nslots] partial words + [2][ncell] z words
    double *X0 = nullptr;                                  // copy of the first guess (a solve that gave up is repeated by the two-launch form)
  };
  unsigned long long persist_epoch = 0;                    // high half of the tags of the next resident solve (never repeats)
  std::vector<PersistPlan> persist;
  std::vector<int> h_srcmap;                               // host copy of srcmap (local view)
  double *persist_out = nullptr;                           // pinned: iterations, (r,r), status, checks
  int persist_used = 0;                                    // the last pcg solve ran as the resident launch
  int persist_gave_up = 0;                                 // resident solves that gave up a wait (then never used again in this model)
  int red_active_total = 0;                                // fused solver kernels: chunks that have work, summed over the local blocks
  int persist_nwg = 0, persist_cp = 0;                     // shape of the last resident launch
  bool pcsi_two_cell = false;   // fused P-CSI step with two cells per thread (large grids, even row pitch; POP_PCSI_STEP2=0|1)
  bool pcsi_two_step = false;   // ... and two iterations per launch (k_pcsi_step_x2; pop_tuning.pcsi_two_step)
  bool pcsi_two_step_dist = false;   // ... with blocks spread over ranks
  double *pcsi_raw = nullptr;   // the residual of the pair before a check (k_pcsi_step_x2<true> -> k_pcsi_rr_chunks)
  int *pcsi_jfold = nullptr;    // tripole: per local block, the first array row beyond the fold (PcsiArgs::jfold)
  bool pcsi_evp_fused = false;  // P-CSI + EVP: iteration and sub-block solves in one launch (k_pcsi_evp_step; pop_tuning.pcsi_evp_fused)
  bool reg_thomas_t = true;
  int trc_lds_rows = 4;                                    // tracer RHS (centred advection): LDS tile rows, 0 = direct loads
  int mom_lds_rows = 4;                                    // momentum RHS: LDS tile rows (0 = direct-load kernel)
  bool reg_thomas = true;                                  // column-in-registers Thomas kernels (km = 60, 62)
  SolveView gv{};                                         // replicated barotropic mode: all blocks
  double *gTAREA = nullptr; int *gKMT = nullptr;
  int nchunk = 0, numIterations = 0;
  double rmsResidual = 0.0;
  // halo plan on device
  int *copy_dst = nullptr, *copy_src = nullptr, *fill_dst = nullptr;
  int ncopy = 0, nfill = 0;
  std::vector<DevPeer> peers;
  // all peers concatenated (one pack / unpack launch per halo update)
  int *sa_src = nullptr, *sa_start = nullptr, *sa_cnt = nullptr, *ra_dst = nullptr, *ra_start = nullptr, *ra_cnt = nullptr;
  int nsend_all = 0, nrecv_all = 0;
  // fused distributed solvers: per-cell send entries / receive slots (FusedArgs::sendmap, rmap), nz = 1 message order
  int *sendmap = nullptr, *send_off = nullptr, *send_slot = nullptr, *rmap = nullptr;
  int max_blocks_per_rank = 0;                             // over all ranks: choices between collective code paths must not depend on the rank
  bool halo_ns_only = false;                               // every ghost cell owned by another rank lies in a ghost ROW (j-band shards)
  pop_exchange_fn xchg_side = nullptr;                     // the same exchange on the communication stream (own communicator), or null
  hipStream_t comm_side = nullptr;                         // stream of those exchanges (not `side`: impvmixu runs there beside the solver)
  hipEvent_t ev_sa = nullptr, ev_sx = nullptr;            // solver: z packed (launch stream) / z received (side stream)
  long long solver_ops = 0, solver_enq = 0;               // stream operations / iterations enqueued by the last distributed solve (incl. look-ahead)
  // tripole northern boundary, per field location (single rank)
  int *tp_dst[5] = {}, *tp_a[5] = {}, *tp_b[5] = {}; int tp_n[5] = {}; double *tp_buf = nullptr;
  // comm hooks
  double *sendbuf = nullptr, *recvbuf = nullptr, *redbuf = nullptr;
  long long comm_doubles = 0, red_doubles = 0;
  pop_exchange_fn xchg = nullptr;
  pop_allreduce_fn allred = nullptr;
  void *comm_user = nullptr;
  EvpDev evp{}; bool use_evp = false;                                                // EVP block preconditioner (preconditioner_choice = 1)
  double *pcsi_omega = nullptr; int *pcsi_base = nullptr; double pcsi_csy = 0;        // P-CSI: omega_k table, interval base
  std::vector<std::pair<std::pair<double *, int>, hipGraphExec_t>> pcsi_graphs;   // keyed by (solution array, variant)
  double rf_S[MAXNT] = {}, rf_S_prev[MAXNT] = {}; bool rf_S_prev_valid[MAXNT] = {};   // Robert filter
  Upw3Dev upw3{};                                          // tadvect = 2
  LwDev lw{};                                              // tadvect = 3 (lw_lim): flux-velocity and work fields
  GmDev gm{};                                              // hmix_tracer = 3 (gm): slopes, tapered diffusivities, GTK
  RcclTransport *rccl_tr = nullptr;                       // in-library RCCL transport (pop_comm_init_rccl)
  // time stepping
  int oldt = 0, curt = 1, newt = 2, mixt = 1;
  int first_step = 1, leapfrogts = 1, f_euler_ts = 0, avg_ts = 0, nsteps_total = 0, nsteps_this_interval = 0;
  int eod = 0, eod_last = 0;                               // the step ends a day / the previous one did (time_management.F90:1809, 3586-3592; runs that start at midnight)
  double c2dtt = 0, c2dtu = 0, c2dtp = 0, beta = 0;
  std::map<std::string, PhaseTimer> timers;
  // the barotropic solve bracketed by two events on the launch stream, read back one step later (no synchronisation inside the
  // step): totals since the last pop_timers_reset / "solver_ms_reset" for the bench's per-iteration figure
  hipEvent_t ev_solve[2] = {nullptr, nullptr}; bool solve_pending = false; int solve_iters_pending = 0;
  double solver_ms_total = 0.0; long long solver_iters_total = 0, solver_calls_total = 0;
  bool timing = false;
  bool phase_timing = false;   // inside pop_time_phase: kernels only
  double *op_scratch = nullptr;   // pop_operator_host: four block-sized 2-D arrays
  bool prio_on = false; int prio_least = 0;   // stream priorities in use; the lowest one
};

namespace {

template <class T>
int dev_alloc(pop_ctx *c, T **p, size_t n, bool zero = true) {
  void *v = nullptr;
  HIPCHK(c, hipMalloc(&v, std::max<size_t>(n, 1) * sizeof(T)));
  c->allocs.push_back(v);
  if (zero) HIPCHK(c, hipMemset(v, 0, std::max<size_t>(n, 1) * sizeof(T)));
  *p = (T *)v;
  return 0;
}
template <class T>
int dev_upload(pop_ctx *c, T **p, const T *src, size_t n) {
  if (dev_alloc(c, p, n, false)) return 1;
  HIPCHK(c, hipMemcpy(*p, src, n * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}
// extract the local blocks of an all-blocks host field
template <class T>
std::vector<T> local_part(const HostModel &h, const std::vector<T> &all) {
  std::vector<T> out(h.n2 * h.nblocks);
  for (int lb = 0; lb < h.nblocks; ++lb)
    std::copy(all.begin() + (size_t)(h.local_ids[lb] - 1) * h.n2, all.begin() + (size_t)h.local_ids[lb] * h.n2, out.begin() + (size_t)lb * h.n2);
  return out;
}

// column kernels: one wave per workgroup; tile order per kernels_common.hpp col_setup
dim3 grid_cols(const pop_ctx *c) {
  if (c->g.xcd_remap == 2) return dim3(tile_grid_x(c->g.nxb, c->g.nyb, POP_COL_THREADS, 1), c->g.nblocks);
  return dim3(col_grid(c->g, POP_COL_THREADS), c->g.nblocks);
}
dim3 block_stencil() { return dim3(POP_COL_THREADS, 1); }
dim3 grid_stencil(const pop_ctx *c) { return grid_cols(c); }
dim3 grid_2d(const pop_ctx *c) { return dim3(red_grid_x(c->g), c->g.nblocks); }
dim3 grid_3d(const pop_ctx *c) { return dim3((c->g.n2 + 255) / 256, c->g.km, c->g.nblocks); }

StepParams step_params(const pop_ctx *c) {
  const pop_config &cf = c->h.c;
  StepParams s{};
  s.c2dtu = c->c2dtu; s.c2dtp = c->c2dtp; s.beta = c->beta; s.gamma = 1.0 - 2.0 * (1.0 / 3.0);
  s.dtp = c->h.dtp; s.grav = GRAV;
  s.am = cf.am; s.ah = cf.ah; s.bottom_drag = cf.bottom_drag;
  s.const_vvc = cf.const_vvc; s.const_vdc = cf.const_vdc; s.convect_diff = cf.convect_diff; s.convect_visc = cf.convect_visc;
  s.aidif = cf.aidif;
  s.rich_bckgrnd_vvc = cf.rich_bckgrnd_vvc; s.rich_bckgrnd_vdc = cf.rich_bckgrnd_vdc; s.rich_mix = cf.rich_mix;
  s.leapfrogts = c->leapfrogts; s.pavg = (cf.lpressure_avg && c->leapfrogts) ? 1 : 0;
  s.impcor = cf.impcor; s.reset_to_freezing = cf.reset_to_freezing;
  s.nvdc = (cf.vmix_choice == 3) ? 2 : 1;
  return s;
}

struct ScopedPhase {   // optional HIP-event timing of a phase on the launch stream
  pop_ctx *c; const char *name; hipEvent_t e0 = nullptr, e1 = nullptr;
  ScopedPhase(pop_ctx *c_, const char *n) : c(c_), name(n) {
    if (c->timing) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, c->stream); }
  }
  ~ScopedPhase() {
    if (c->timing) {
      hipEventRecord(e1, c->stream); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      auto &t = c->timers[name]; t.ms += ms; t.calls += 1;
      hipEventDestroy(e0); hipEventDestroy(e1);
    }
  }
};

// ---------------------------------------------------------------------------------------------
// halo update of a device-resident field with nz levels (mpi/POP_HaloMod.F90:1732-2071 2-D,
// :2766-3211 3-D): local ghost copies + fills in one launch, then one packed message per peer
// ---------------------------------------------------------------------------------------------
// what the transport said about a failed exchange / all-reduce
std::string tr_err(const pop_ctx *c) { return c->rccl_tr ? ": " + c->rccl_tr->err : std::string(" in the host callback"); }
// remote part only: one pack launch, the exchange, one unpack launch
int halo_remote(pop_ctx *c, double *F, int nz) {
  if (c->peers.empty()) return 0;
  const int n2 = c->g.n2;
  if (!c->xchg || !c->sendbuf) { c->err = "halo_update: multi-rank run without pop_set_comm"; return 1; }
  std::vector<int> peer; std::vector<long long> soff, scnt, roff, rcnt;
  long long so = 0, ro = 0;
  for (auto &p : c->peers) {
    peer.push_back(p.rank); soff.push_back(so); scnt.push_back((long long)p.nsend * nz); roff.push_back(ro); rcnt.push_back((long long)p.nrecv * nz);
    so += (long long)p.nsend * nz; ro += (long long)p.nrecv * nz;
  }
  if (so > c->comm_doubles || ro > c->comm_doubles) { c->err = "halo_update: comm buffer too small"; return 1; }
  if (c->nsend_all) hipLaunchKernelGGL(k_halo_pack_all, dim3((c->nsend_all + 255) / 256, nz), dim3(256), 0, c->stream, (const double *)F, c->sa_src, c->sa_start, c->sa_cnt, c->nsend_all, c->sendbuf, nz, n2);
  if (c->xchg(c->comm_user, (int)peer.size(), peer.data(), soff.data(), scnt.data(), roff.data(), rcnt.data())) {
    c->err = "halo_update: exchange failed" + tr_err(c); return 1;
  }
  if (c->nrecv_all) hipLaunchKernelGGL(k_halo_unpack_all, dim3((c->nrecv_all + 255) / 256, nz), dim3(256), 0, c->stream, F, c->ra_dst, c->ra_start, c->ra_cnt, c->nrecv_all, (const double *)c->recvbuf, nz, n2);
  return 0;
}
int halo_update(pop_ctx *c, double *F, int nz, double fill = 0.0, int loc = 0, int kind = 0) {
  const int n2 = c->g.n2;
  if (halo_remote(c, F, nz)) return 1;
  const int nloc = c->ncopy + c->nfill;
  if (nloc) hipLaunchKernelGGL(k_halo_local, dim3((nloc + 255) / 256, nz), dim3(256), 0, c->stream, F, c->copy_dst, c->copy_src, c->ncopy, c->fill_dst, c->nfill, fill, nz, n2);
  if (c->h.c.ns_boundary == 2 && c->tp_n[loc]) {   // tripole northern boundary (mpi/POP_HaloMod.F90:1936-2050)
    const int n = c->tp_n[loc];
    if (nz > c->h.km + 2) { c->err = "halo_update: too many levels for the tripole buffer"; return 1; }
    hipLaunchKernelGGL(k_tripole_eval, dim3((n + 255) / 256, nz), dim3(256), 0, c->stream, (const double *)F, c->tp_a[loc], c->tp_b[loc], n, c->tp_buf,
                       kind == 0 ? 1.0 : -1.0, nz, n2);
    hipLaunchKernelGGL(k_tripole_store, dim3((n + 255) / 256, nz), dim3(256), 0, c->stream, F, c->tp_dst[loc], n, (const double *)c->tp_buf, nz, n2);
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}

// Several fields, one halo update: ONE message per neighbour rank carrying all of them (pack, exchange, unpack = three
// stream operations whatever the number of fields) and one launch for the ghost copies / fills inside the rank.
// Field by field the result is the one halo_update gives (same cells, same values).  fill value 0.
// loc / kind: POP_HaloUpdate's fieldLoc / fieldKind (0 centre, 1 NE corner, 2 N face, 3 E face; 0 scalar, 1 vector); they matter on a tripole boundary only
struct HaloItem { double *F; int nz; int loc = 0, kind = 0; };
int halo_update_many(pop_ctx *c, const std::vector<HaloItem> &items) {
  if (items.size() == 1 || items.size() > 8 || c->h.c.ns_boundary == 2 || tun_on(c->h.tun.halo_separate)) {
    for (const HaloItem &it : items) if (halo_update(c, it.F, it.nz, 0.0, it.loc, it.kind)) return 1;
    return 0;
  }
  HaloFields H{};
  H.nf = (int)items.size();
  int tot = 0;
  for (int f = 0; f < H.nf; ++f) { H.F[f] = items[f].F; H.nz[f] = items[f].nz; H.lev0[f] = tot; tot += items[f].nz; }
  H.nztot = tot;
  const int n2 = c->g.n2;
  if (!c->peers.empty()) {
    if (!c->xchg || !c->sendbuf) { c->err = "halo_update: multi-rank run without pop_set_comm"; return 1; }
    std::vector<int> peer; std::vector<long long> soff, scnt, roff, rcnt;
    long long so = 0, ro = 0;
    for (auto &p : c->peers) {
      peer.push_back(p.rank); soff.push_back(so); scnt.push_back((long long)p.nsend * tot); roff.push_back(ro); rcnt.push_back((long long)p.nrecv * tot);
      so += (long long)p.nsend * tot; ro += (long long)p.nrecv * tot;
    }
    if (so > c->comm_doubles || ro > c->comm_doubles) {   // buffers of an older host framework: field by field
      for (const HaloItem &it : items) if (halo_update(c, it.F, it.nz)) return 1;
      return 0;
    }
    if (c->nsend_all) hipLaunchKernelGGL(k_halo_pack_many, dim3((c->nsend_all + 255) / 256, tot), dim3(256), 0, c->stream, H, c->sa_src, c->sa_start, c->sa_cnt, c->nsend_all, c->sendbuf, n2);
    if (c->xchg(c->comm_user, (int)peer.size(), peer.data(), soff.data(), scnt.data(), roff.data(), rcnt.data())) { c->err = "halo_update: exchange failed" + tr_err(c); return 1; }
    if (c->nrecv_all) hipLaunchKernelGGL(k_halo_unpack_many, dim3((c->nrecv_all + 255) / 256, tot), dim3(256), 0, c->stream, H, c->ra_dst, c->ra_start, c->ra_cnt, c->nrecv_all, (const double *)c->recvbuf, n2);
  }
  const int nloc = c->ncopy + c->nfill;
  if (nloc) hipLaunchKernelGGL(k_halo_local_many, dim3((nloc + 255) / 256, tot), dim3(256), 0, c->stream, H, c->copy_dst, c->copy_src, c->ncopy, c->fill_dst, c->nfill, 0.0, n2);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// The same update in two halves around work that does not need the ghost cells of other ranks (north_star: "halo updates
// ... overlapped with interior stencil work on a second HIP stream").  begin: ghost copies inside the rank, pack, and the
// exchange on the communication stream (second communicator); end: unpack and the ghost copies again (corner ghosts
// take values that just arrived; the others are rewritten with the same values).  Between the two the launch stream
// may run anything that reads only cells this rank owns or ghosts with a source on this rank.
bool halo_async_ok(const pop_ctx *c) {
  return !c->peers.empty() && c->halo_ns_only && c->xchg_side && c->comm_side && c->h.c.ns_boundary != 2 && !tun_on(c->h.tun.halo_overlap_off);
}
struct HaloAsync { HaloFields H; int tot; };
int halo_many_begin(pop_ctx *c, const std::vector<HaloItem> &items, HaloAsync &A) {
  HaloFields &H = A.H;
  H = HaloFields{};
  H.nf = (int)items.size();
  int tot = 0;
  for (int f = 0; f < H.nf; ++f) { H.F[f] = items[f].F; H.nz[f] = items[f].nz; H.lev0[f] = tot; tot += items[f].nz; }
  H.nztot = tot; A.tot = tot;
  const int n2 = c->g.n2, nloc = c->ncopy + c->nfill;
  std::vector<int> peer; std::vector<long long> soff, scnt, roff, rcnt;
  long long so = 0, ro = 0;
  for (auto &p : c->peers) {
    peer.push_back(p.rank); soff.push_back(so); scnt.push_back((long long)p.nsend * tot); roff.push_back(ro); rcnt.push_back((long long)p.nrecv * tot);
    so += (long long)p.nsend * tot; ro += (long long)p.nrecv * tot;
  }
  if (so > c->comm_doubles || ro > c->comm_doubles) { c->err = "halo_update: comm buffer too small"; return 1; }
  if (nloc) hipLaunchKernelGGL(k_halo_local_many, dim3((nloc + 255) / 256, tot), dim3(256), 0, c->stream, H, c->copy_dst, c->copy_src, c->ncopy, c->fill_dst, c->nfill, 0.0, n2);
  if (c->nsend_all) hipLaunchKernelGGL(k_halo_pack_many, dim3((c->nsend_all + 255) / 256, tot), dim3(256), 0, c->stream, H, c->sa_src, c->sa_start, c->sa_cnt, c->nsend_all, c->sendbuf, n2);
  HIPCHK(c, hipEventRecord(c->ev_sa, c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->comm_side, c->ev_sa, 0));
  if (c->xchg_side(c->comm_user, (int)peer.size(), peer.data(), soff.data(), scnt.data(), roff.data(), rcnt.data())) { c->err = "halo_update: exchange failed" + tr_err(c); return 1; }
  HIPCHK(c, hipEventRecord(c->ev_sx, c->comm_side));
  return 0;
}
int halo_many_end(pop_ctx *c, const HaloAsync &A) {
  const int n2 = c->g.n2, nloc = c->ncopy + c->nfill;
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_sx, 0));
  if (c->nrecv_all) hipLaunchKernelGGL(k_halo_unpack_many, dim3((c->nrecv_all + 255) / 256, A.tot), dim3(256), 0, c->stream, A.H, c->ra_dst, c->ra_start, c->ra_cnt, c->nrecv_all, (const double *)c->recvbuf, n2);
  if (nloc) hipLaunchKernelGGL(k_halo_local_many, dim3((nloc + 255) / 256, A.tot), dim3(256), 0, c->stream, A.H, c->copy_dst, c->copy_src, c->ncopy, c->fill_dst, c->nfill, 0.0, n2);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// stage 2+3 of a reduction whose partials are already in c->partial
template <int NF>
int reduce_finish(pop_ctx *c, int mode) {
  double *bs = c->blocksum;
  if (c->h.nranks > 1) {
    if (!c->allred || !c->redbuf) { c->err = "global sum: multi-rank run without pop_set_comm"; return 1; }
    bs = c->redbuf;
    hipLaunchKernelGGL(k_block_sums_global<NF>, dim3(c->h.nblocks_tot), dim3(POP_RED_THREADS), 0, c->stream, c->partial, c->nchunk, c->loc_of_gid, bs);
  } else hipLaunchKernelGGL(k_block_sums<NF>, dim3(c->g.nblocks), dim3(POP_RED_THREADS), 0, c->stream, c->partial, c->nchunk, c->gid, bs);
  if (c->h.nranks > 1 && c->allred(c->comm_user, 0, (long long)NF * c->h.nblocks_tot)) { c->err = "global sum: allreduce callback failed" + tr_err(c); return 1; }
  hipLaunchKernelGGL(k_finalize<NF>, dim3(1), dim3(1), 0, c->stream, bs, c->h.nblocks_tot, c->sc, mode);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int read_scalars(pop_ctx *c, SolverScalars *out) {
  HIPCHK(c, hipMemcpyAsync(out, c->sc, sizeof(SolverScalars), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

SolverArgs solver_args(pop_ctx *c) {
  SolverArgs a{};
  a.X = c->PS[c->newt]; a.R = c->R; a.S0 = c->S0; a.S1 = c->S1; a.Q = c->Q; a.Z = c->Z; a.AZ = c->AZ;
  a.Bv = c->RHS; a.C = c->centerWgt; a.partial = c->partial; a.sc = c->sc;
  return a;
}

// preconditioner() with preconditionerChoice = 'evp' (:2331-2366): PX <- sub-block solves of X on the physical cells
// residual: X is a residual of a solver (zero on land): sub-blocks without an ocean cell are not read (k_evp_apply_wave3<true>)
int evp_apply(pop_ctx *c, const double *X, double *PX, bool residual = true) {
  const int wave = tun_or(c->h.tun.evp_wave, 3);   // 3 (default): wavefronts, operands in registers, every load up front; 2: the same with the loads behind their conditions; 1: wavefronts, operands in LDS; 0: a thread per sub-block
  if (wave == 3 && c->evp.C0 && residual)
    hipLaunchKernelGGL(k_evp_apply_wave3<true>, dim3((unsigned)((c->evp.S + POP_EVP_SB - 1) / POP_EVP_SB)), dim3(64), 0, c->stream, c->evp, c->g.nxb, X, PX);
  else if (wave == 3 && c->evp.C0)
    hipLaunchKernelGGL(k_evp_apply_wave3<false>, dim3((unsigned)((c->evp.S + POP_EVP_SB - 1) / POP_EVP_SB)), dim3(64), 0, c->stream, c->evp, c->g.nxb, X, PX);
  else if (wave == 2 && c->evp.C0)
    hipLaunchKernelGGL(k_evp_apply_wave2, dim3((unsigned)((c->evp.S + POP_EVP_SB - 1) / POP_EVP_SB)), dim3(64), 0, c->stream, c->evp, c->g.nxb, X, PX);
  else if (wave != 0)   // anti-diagonal wavefronts: eight lanes per sub-block, eight sub-blocks per wave
    hipLaunchKernelGGL(k_evp_apply_wave, dim3((unsigned)((c->evp.S + POP_EVP_SB - 1) / POP_EVP_SB)), dim3(64), 0, c->stream, c->evp, c->g.nxb, X, PX);
  else
    hipLaunchKernelGGL(k_evp_apply, dim3((unsigned)((c->evp.S + POP_EVP_THREADS - 1) / POP_EVP_THREADS)), dim3(POP_EVP_THREADS), 0, c->stream,
                       c->evp, c->g.nxb, X, PX);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// POP_SolversRun -> pcg (POP_SolversMod.F90:1255-1503), diagonal or EVP preconditioner
int solver_pcg(pop_ctx *c) {
  const pop_config &cf = c->h.c;
  const dim3 G = grid_2d(c), B(POP_RED_THREADS);
  SolverScalars init{}; init.eta0 = 1.0;
  HIPCHK(c, hipMemcpyAsync(c->sc, &init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(c->S0, 0, sizeof(double) * c->g.n2 * c->g.nblocks, c->stream));
  SolverArgs a = solver_args(c);
  hipLaunchKernelGGL(k_residual<false>, grid_2d(c), B, 0, c->stream, c->g, a);
  if (halo_update(c, c->R, 1)) return 1;
  c->numIterations = cf.max_iterations;
  double rr = 0.0;
  bool pending = false;   // x,r update of the previous iteration not yet applied
  for (int m = 1; m <= cf.max_iterations; ++m) {
    a = solver_args(c);
    if (c->use_evp) {   // :1322-1362: z = M^-1 r by sub-block solves, (r,z), halo of z
      if (pending) hipLaunchKernelGGL(k_pcg_xr, G, B, 0, c->stream, c->g, a);
      if (evp_apply(c, c->R, c->Z)) return 1;
      hipLaunchKernelGGL(k_dot_partial, G, B, 0, c->stream, c->g, (const double *)c->R, (const double *)c->Z, c->g.mMask, c->partial);
      if (halo_update(c, c->Z, 1)) return 1;
    } else if (pending) hipLaunchKernelGGL(k_pcg_a<true>, G, B, 0, c->stream, c->g, a);
    else hipLaunchKernelGGL(k_pcg_a<false>, G, B, 0, c->stream, c->g, a);
    if (reduce_finish<1>(c, FIN_PCG_RZ)) return 1;
    hipLaunchKernelGGL(k_pcg_b, G, B, 0, c->stream, c->g, a);
    std::swap(c->S0, c->S1);
    if (halo_update(c, c->Q, 1)) return 1;
    if (reduce_finish<1>(c, FIN_PCG_SQ)) return 1;
    pending = true;
    if (m % cf.convergence_check_freq == 0) {
      a = solver_args(c);
      hipLaunchKernelGGL(k_pcg_xr, G, B, 0, c->stream, c->g, a);
      pending = false;
      hipLaunchKernelGGL(k_residual<true>, grid_2d(c), B, 0, c->stream, c->g, a);
      if (halo_update(c, c->R, 1)) return 1;
      if (reduce_finish<1>(c, FIN_RR)) return 1;
      SolverScalars s;
      if (read_scalars(c, &s)) return 1;
      rr = s.rr;
      if (rr < c->h.convergenceCriterion) { c->numIterations = m; break; }
    }
  }
  if (pending) { a = solver_args(c); hipLaunchKernelGGL(k_pcg_xr, G, B, 0, c->stream, c->g, a); }
  c->rmsResidual = std::sqrt(rr * c->h.residualNorm);
  HIPCHK(c, hipGetLastError());
  if (c->numIterations == cf.max_iterations && c->h.convergenceCriterion != 0.0) { c->err = "POP_SolversPCG: solver not converged"; return 2; }
  return 0;
}

// pcg, fused form: two launches per iteration, halo folded into the matvec through srcmap, final
// reduction stage recomputed by the consumer kernel, and one hipGraph replay per
// convergenceCheckFreq iterations (same arithmetic and summation order as solver_pcg).  It runs on a
// SolveView: the rank's own blocks (single rank), or -- replicated barotropic mode -- every block
// of the decomposition on every rank.
// which forms of step A / step B the fused pcg launches (launch_fpcg_a / launch_fpcg_b)
static bool fpcg_pair_ok(const pop_ctx *c, const SolveView &v, const FusedArgs &a) {
  return v.g.red_act && a.presummed && !a.sendmap && (v.g.red_nact % 16) == 0 && !c->fpcg_one_cell && !tun_off(c->h.tun.fpcg_a_pair);
}
static bool fpcg_two_ok(const pop_ctx *c, const SolveView &v, const FusedArgs &a) {
  return a.presummed && (v.g.nxb & 1) == 0 && !v.g.red_tiles && !c->fpcg_one_cell;
}
FusedArgs fused_args(pop_ctx *c, const SolveView &v) {
  FusedArgs a{};
  a.X = v.X; a.R = v.R; a.Z = v.Z; a.S0 = v.S0; a.S1 = v.S1; a.Q = v.Q;
  a.Bv = v.RHS; a.C = v.C; a.partA = v.partial; a.partB = v.partial + (size_t)v.nchunk * v.g.nblocks;
  a.sc = c->sc; a.srcmap = v.srcmap; a.nchunk = v.nchunk; a.nblocks = v.g.nblocks;
  a.bsA = v.blocksum + 2 * v.nblocks_tot; a.bsB = v.blocksum + 3 * v.nblocks_tot;
  a.presummed = ((long long)v.nchunk * v.g.nblocks > 2048 || c->force_presum) ? 1 : 0;
  return a;
}
// large grids: ordered block sums of a partial array between solver kernels (view-local block order)
// (more than 64 terms per accumulator: the four-threads-per-accumulator form, one memory round trip instead of two or three)
constexpr int POP_RELAY_LMAX = 36;
static bool presum_relay(const pop_ctx *c, const SolveView &v) {
  const int terms = (v.nchunk + POP_RED_THREADS - 1) / POP_RED_THREADS;
  const int t = tun_or(c->h.tun.block_sums_relay, 1);   // 2: wherever it can run (the cross-check on small grids)
  return (terms > 64 || t == 2) && (terms + 3) / 4 <= POP_RELAY_LMAX && t != 0;
}
void presum(pop_ctx *c, const SolveView &v, const double *partial, double *bs) {
  if (presum_relay(c, v)) hipLaunchKernelGGL((k_block_sums_relay<1, POP_RELAY_LMAX>), dim3(v.g.nblocks), dim3(1024), 0, c->stream, partial, v.nchunk, (const int *)c->iota, bs);
  else hipLaunchKernelGGL(k_block_sums<1>, dim3(v.g.nblocks), dim3(POP_RED_THREADS), 0, c->stream, partial, v.nchunk, c->iota, bs);
}
dim3 view_grid(const SolveView &v) { return dim3(red_grid_x(v.g), v.g.nblocks); }
// r = b - A x (+ partial (r,r)) of the fused solvers: two cells per thread on large grids, else one
template <bool WITH_RR>
void launch_fresidual(pop_ctx *c, const SolveView &v, const FusedArgs &a) {
  const dim3 G = view_grid(v);
  if (a.presummed && (v.g.nxb & 1) == 0 && !v.g.red_tiles && !c->fpcg_one_cell)
    hipLaunchKernelGGL(k_fresidual2<WITH_RR>, G, dim3(POP_RED_THREADS / 2), 0, c->stream, v.g, a);
  else hipLaunchKernelGGL(k_fresidual<WITH_RR>, G, dim3(POP_RED_THREADS), 0, c->stream, v.g, a);
}
// step A of the fused pcg: two chunks per workgroup on compacted launches (single rank), else one
void launch_fpcg_a(pop_ctx *c, const SolveView &v, const FusedArgs &a, bool update) {
  const dim3 G = view_grid(v), B(POP_RED_THREADS);
  const bool pair = fpcg_pair_ok(c, v, a);
  if (pair) {
    const dim3 GP(G.x / 2, G.y);
    if (update) hipLaunchKernelGGL(k_fpcg_a_pair<true>, GP, B, 0, c->stream, v.g, a);
    else hipLaunchKernelGGL(k_fpcg_a_pair<false>, GP, B, 0, c->stream, v.g, a);
  } else if (update) hipLaunchKernelGGL(k_fpcg_a<true>, G, B, 0, c->stream, v.g, a);
  else hipLaunchKernelGGL(k_fpcg_a<false>, G, B, 0, c->stream, v.g, a);
}
// step B of the fused pcg: two cells per thread on large grids (presummed block sums, even row pitch), else one
// xupd: the pending x += alpha s of the previous iteration is applied here (k_fpcg_a<true> ran before and published alpha)
void launch_fpcg_b(pop_ctx *c, const SolveView &v, const FusedArgs &a, bool xupd) {
  const dim3 G = view_grid(v);
  const bool two = fpcg_two_ok(c, v, a);
  // (occupancy probe, profiles/r03_ab_b2_occupancy.txt: with dynamic LDS holding the kernel to 3 / 2 waves per SIMD instead of its 4
  // the step costs +2.2 / +7.1 ms; the two-cell form needs 108 VGPRs, a 96- or 80-register budget spills 84 / 140 B)
  if (two && xupd) hipLaunchKernelGGL(k_fpcg_b2<true>, G, dim3(POP_RED_THREADS / 2), 0, c->stream, v.g, a);
  else if (two) hipLaunchKernelGGL(k_fpcg_b2<false>, G, dim3(POP_RED_THREADS / 2), 0, c->stream, v.g, a);
  else if (xupd) hipLaunchKernelGGL(k_fpcg_b<true>, G, dim3(POP_RED_THREADS), 0, c->stream, v.g, a);
  else hipLaunchKernelGGL(k_fpcg_b<false>, G, dim3(POP_RED_THREADS), 0, c->stream, v.g, a);
}
// one check interval: freq iterations, pending update, residual + (r,r) -> host
int fused_interval(pop_ctx *c, SolveView &v, int freq, bool first_has_pending) {
  const dim3 G = view_grid(v), B(POP_RED_THREADS);
  bool pending = first_has_pending;
  for (int it = 0; it < freq; ++it) {
    FusedArgs a = fused_args(c, v);
    launch_fpcg_a(c, v, a, pending);
    if (a.presummed) presum(c, v, a.partA, (double *)a.bsA);
    launch_fpcg_b(c, v, a, pending);
    if (a.presummed) presum(c, v, a.partB, (double *)a.bsB);
    std::swap(v.S0, v.S1);
    pending = true;
  }
  FusedArgs a = fused_args(c, v);
  hipLaunchKernelGGL(k_fpcg_xr, G, B, 0, c->stream, v.g, a);
  launch_fresidual<true>(c, v, a);
  // the view holds every block it sums (single rank or replicated), in block-id order
  hipLaunchKernelGGL(k_rr_total, dim3(1), dim3(POP_RED_THREADS), 0, c->stream, (const double *)v.partial, v.nchunk, v.g.nblocks, c->sc, c->host_rr, c->h.convergenceCriterion);
  return 0;
}
// Check intervals with one interval of look-ahead.  `enqueue(i)` puts interval i on the stream (a hipGraph replay or
// plain launches) and returns whether it ends with a convergence check (k_rr_total).  The host keeps at most two checked
// intervals in flight and examines them in order; the check that meets the criterion raises the device stop flag, so the
// interval already enqueued behind it does nothing, and the GPU never idles while the host looks at a residual.
// Returns the index of the converged interval or -1; rr = last residual seen.
template <class Enqueue>
int run_intervals(pop_ctx *c, int nint, Enqueue enqueue, double &rr, int &err) {
  int next = 0, ring = 0, head = 0;          // ring: checks enqueued; head: checks examined
  int idx[8] = {};
  err = 0;
  auto fill = [&]() {
    while (next < nint && ring - head < 2) {
      const int chk = enqueue(next);
      if (chk < 0) { err = 1; return; }
      if (chk) {
        if (hipEventRecord(c->chk_ev[ring & 3], c->stream) != hipSuccess) { err = 1; return; }
        idx[ring & 7] = next; ++ring;
      }
      ++next;
    }
  };
  fill();
  while (!err && head < ring) {
    if (hipEventSynchronize(c->chk_ev[head & 3]) != hipSuccess) { err = 1; break; }
    rr = c->host_rr[head & 7];
    const int i = idx[head & 7];
    ++head;
    if (rr < c->h.convergenceCriterion) return i;
    fill();
  }
  return -1;
}

// ---- the resident pcg of small grids (kernels_pcg_persist.hpp) ------------------------------------------------------------------
// Plan of one view: which chunks a workgroup owns, the window index of every stencil neighbour, the halo cells.  Built on the host
// from the view's source map (ghost -> source cell, -1 = fill), so block boundaries inside the view, the cyclic wrap, closed
// boundaries and a tripole fold need no case of their own.  Returns nullptr (with the reason kept) when the view does not qualify.
static pop_ctx::PersistPlan *persist_plan(pop_ctx *c, const SolveView &v) {
  for (auto &p : c->persist) if (p.key == (const void *)v.srcmap) return p.ok ? &p : nullptr;
  c->persist.emplace_back();
  pop_ctx::PersistPlan &pl = c->persist.back();
  pl.key = (const void *)v.srcmap;
  const HostModel &h = c->h;
  const int n2 = (int)h.n2, nxb = h.nxb, nchunk = v.nchunk, nb = v.g.nblocks, nslots = nchunk * nb;
  const bool global = v.srcmap != c->srcmap;                // the replicated view holds every block of the decomposition
  auto refuse = [&](const char *why) -> pop_ctx::PersistPlan * { pl.why = why; return nullptr; };
  if (nb * ((nchunk + POP_RED_THREADS - 1) / POP_RED_THREADS) > POP_PERSIST_MAXP) return refuse("too many partial slots per thread");
  // chunks per workgroup: the smallest of 1 / 2 / 4 / 8 that needs at most 250 workgroups (one per CU: the waits need every workgroup
  // resident).  gx1v7 in one block: 246 x 2 (10.7 us per iteration; 123 x 4: 12.0, 62 x 8: 14.1 on the same box, profiles/r04_ab_persist_shape.txt);
  // gx1v7 in the eight 48-row bands of the 8-rank decomposition (replicated solve): 144 x 4.
  // measurement only: pop_tuning.pcg_persist = 2 | 4 | 8 forces that many chunks per workgroup
  std::vector<int> cand;
  if (c->h.tun.pcg_persist == 2 || c->h.tun.pcg_persist == 4 || c->h.tun.pcg_persist == 8) { if ((nslots + c->h.tun.pcg_persist - 1) / c->h.tun.pcg_persist <= 250) cand.push_back(c->h.tun.pcg_persist); }
  if (cand.empty())
    for (int cp : {1, 2, 4, 8}) if ((nslots + cp - 1) / cp <= 250) { cand.push_back(cp); break; }
  if (cand.empty()) return refuse("more than 2000 chunks");
  const std::vector<int> sm = global ? global_srcmap(h) : c->h_srcmap;
  if ((long long)sm.size() != (long long)n2 * nb) return refuse("source map size");
  const int off[8] = {nxb, -nxb, 1, -1, nxb + 1, -nxb + 1, nxb - 1, -nxb - 1};
  int CP = 0, nwg = 0, nwin_max = 0;
  std::vector<int> own, hoff, hq;
  std::vector<unsigned short> nbr;
  const char *why = "";
  for (int cp : cand) {
    why = "";
    nwg = (nslots + cp - 1) / cp; nwin_max = 0;
    const int NOWN = cp * POP_RED_THREADS;
    own.assign((size_t)nwg * NOWN, -1); hoff.assign(nwg + 1, 0); hq.clear();
    nbr.assign((size_t)nwg * NOWN * 8, 0);
    for (int w = 0; w < nwg && !*why; ++w) {
      std::unordered_map<int, int> where;                    // cell -> window index
      for (int u = 0; u < cp; ++u) {
        const int slot = w * cp + u;
        if (slot >= nslots) break;
        const int b = slot / nchunk, ch = slot % nchunk;
        const BlockInfo &B = h.all_blocks[global ? b : h.local_ids[b] - 1];
        for (int t = 0; t < POP_RED_THREADS; ++t) {
          const int p2 = ch * POP_RED_THREADS + t;
          if (p2 >= n2) break;
          const int i = p2 % nxb + 1, j = p2 / nxb + 1;
          if (i < B.ib || i > B.ie || j < B.jb || j > B.je) continue;
          own[(size_t)w * NOWN + u * POP_RED_THREADS + t] = b * n2 + p2;
          where[b * n2 + p2] = u * POP_RED_THREADS + t;
        }
      }
      const size_t h0 = hq.size();
      for (int L = 0; L < NOWN; ++L) {
        const int q = own[(size_t)w * NOWN + L];
        if (q < 0) continue;
        for (int n = 0; n < 8; ++n) {
          const int m = sm[q + off[n]];
          int idx;
          if (m < 0) idx = -1;
          else {
            auto it = where.find(m);
            if (it != where.end()) idx = it->second;
            else { idx = NOWN + (int)(hq.size() - h0); where[m] = idx; hq.push_back(m); }
          }
          nbr[((size_t)w * NOWN + L) * 8 + n] = (unsigned short)(idx < 0 ? 0xFFFF : idx);
        }
      }
      const int nhalo = (int)(hq.size() - h0), nwin = NOWN + nhalo + 1;
      if ((nhalo + POP_RED_THREADS - 1) / POP_RED_THREADS > POP_PERSIST_MAXH) { why = "halo of a workgroup too large"; break; }
      if (nwin >= 0xFFFF) { why = "window too large"; break; }
      for (int L = 0; L < NOWN; ++L) for (int n = 0; n < 8; ++n) {
        unsigned short &x = nbr[((size_t)w * NOWN + L) * 8 + n];
        if (x == 0xFFFF) x = (unsigned short)(nwin - 1);     // the cell of zeros (fill value of closed boundaries)
      }
      hoff[w + 1] = (int)hq.size();
      nwin_max = std::max(nwin_max, nwin);
    }
    if (!*why && (size_t)3 * nwin_max * sizeof(double) > 60000) why = "window does not fit the LDS budget";
    if (!*why) { CP = cp; break; }
  }
  if (!CP) return refuse(why);
  if (hq.empty()) hq.push_back(0);
  if (dev_upload(c, &pl.own_q, own.data(), own.size()) || dev_upload(c, &pl.nbr, nbr.data(), nbr.size()) ||
      dev_upload(c, &pl.halo_off, hoff.data(), hoff.size()) || dev_upload(c, &pl.halo_q, hq.data(), hq.size())) return nullptr;
  const size_t nwords = 4 * (size_t)nslots + 2 * (size_t)n2 * nb;      // partials [2 buffers][2 fields (ChronGear)][nslots], then z [2][ncell]
  if (nwords * sizeof(PWord) >= (1ULL << 32)) return refuse("exchange buffer beyond 32-bit offsets");
  double *p = nullptr;
  if (dev_alloc(c, &p, 2 * nwords)) return nullptr;          // zero-filled: tag 0 is never waited for (epochs start at 1)
  pl.W = reinterpret_cast<PWord *>(p);
  if (dev_alloc(c, &pl.X0, (size_t)n2 * nb)) return nullptr;
  pl.CP = CP; pl.nwg = nwg; pl.nwin_max = nwin_max; pl.nslots = nslots; pl.ok = true;
  return &pl;
}
int solver_pcg_persist(pop_ctx *c, SolveView &v, const pop_ctx::PersistPlan &pl) {
  const pop_config &cf = c->h.c;
  const long long ncell = (long long)v.g.n2 * v.g.nblocks;
  PersistArgs a{};
  a.X = v.X; a.Bv = v.RHS; a.C = v.C; a.WNo = v.g.WNo; a.WEa = v.g.WEa; a.WNE = v.g.WNE; a.mMask8 = v.g.mMask8;
  a.nxb = v.g.nxb; a.nchunk = v.nchunk; a.nblocks = v.g.nblocks; a.nslots = pl.nslots; a.ncell = ncell;
  a.own_q = pl.own_q; a.nbr = pl.nbr; a.halo_off = pl.halo_off; a.halo_q = pl.halo_q; a.W = pl.W;
  a.epoch = (++c->persist_epoch) << 32;                    // tags of this solve: no word of an earlier solve can carry one of them
  a.max_iter = cf.max_iterations; a.freq = cf.convergence_check_freq; a.criterion = c->h.convergenceCriterion; a.out = c->persist_out;
  a.wait_ticks = 200000000ULL;                             // 2 s
  HIPCHK(c, hipMemcpyAsync(pl.X0, v.X, sizeof(double) * ncell, hipMemcpyDeviceToDevice, c->stream));   // the first guess, should the solve have to be repeated
  c->persist_out[0] = -1.0; c->persist_out[1] = 0.0; c->persist_out[2] = 0.0; c->persist_out[3] = 0.0;
  const size_t lds = (size_t)3 * pl.nwin_max * sizeof(double);
  const dim3 G(pl.nwg), B(POP_RED_THREADS);
  switch (pl.CP) {
    case 1: hipLaunchKernelGGL(k_pcg_persist<1>, G, B, lds, c->stream, a); break;
    case 2: hipLaunchKernelGGL(k_pcg_persist<2>, G, B, lds, c->stream, a); break;
    case 4: hipLaunchKernelGGL(k_pcg_persist<4>, G, B, lds, c->stream, a); break;
    default: hipLaunchKernelGGL(k_pcg_persist<8>, G, B, lds, c->stream, a); break;
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->chk_ev[0], c->stream));
  HIPCHK(c, hipEventSynchronize(c->chk_ev[0]));
  if (c->persist_out[2] != 0.0 || c->persist_out[0] < 0.0) {
    char b[200];
    snprintf(b, sizeof b, " [iterations %g, status %g, checks %g, %d workgroups x %d chunks, %d blocks]", c->persist_out[0], c->persist_out[2], c->persist_out[3], pl.nwg, pl.CP, v.g.nblocks);
    c->err = std::string(c->persist_out[0] < 0.0 ? "resident pcg: the launch left no result" : "resident pcg: a wait for another workgroup's data gave up (kernels_pcg_persist.hpp)") + b;
    return 3;
  }
  c->numIterations = (int)c->persist_out[0];
  c->rmsResidual = std::sqrt(c->persist_out[1] * c->h.residualNorm);
  c->persist_used = 1; c->persist_nwg = pl.nwg; c->persist_cp = pl.CP;
  hipLaunchKernelGGL(k_halo_srcmap, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, c->stream, v.X, v.srcmap, ncell);
  HIPCHK(c, hipGetLastError());
  const bool conv = c->persist_out[3] > 0.0 && c->persist_out[1] < c->h.convergenceCriterion;
  if (!conv && c->h.convergenceCriterion != 0.0) { c->err = "POP_SolversPCG: solver not converged"; return 2; }
  return 0;
}

// the iterations of the fused ChronGear as one resident launch (k_cg_persist), after the start-up pass of solver_chrongear_fused
int solver_cg_persist(pop_ctx *c, SolveView &v, const pop_ctx::PersistPlan &pl, const FusedArgs &fa) {
  const pop_config &cf = c->h.c;
  const long long ncell = (long long)v.g.n2 * v.g.nblocks;
  CgPersistArgs ca{};
  PersistArgs &a = ca.p;
  a.X = v.X; a.Bv = v.RHS; a.C = v.C; a.WNo = v.g.WNo; a.WEa = v.g.WEa; a.WNE = v.g.WNE; a.mMask8 = v.g.mMask8;
  a.nxb = v.g.nxb; a.nchunk = v.nchunk; a.nblocks = v.g.nblocks; a.nslots = pl.nslots; a.ncell = ncell;
  a.own_q = pl.own_q; a.nbr = pl.nbr; a.halo_off = pl.halo_off; a.halo_q = pl.halo_q; a.W = pl.W;
  a.epoch = (++c->persist_epoch) << 32;
  a.max_iter = cf.max_iterations; a.freq = cf.convergence_check_freq; a.criterion = c->h.convergenceCriterion; a.out = c->persist_out;
  a.wait_ticks = 200000000ULL;
  ca.R = fa.R; ca.S = fa.S0; ca.Q = fa.Q; ca.A0R = fa.A0R; ca.sc = c->sc;
  c->persist_out[0] = -1.0; c->persist_out[1] = 0.0; c->persist_out[2] = 0.0; c->persist_out[3] = 0.0;
  HIPCHK(c, hipMemcpyAsync(pl.X0, v.X, sizeof(double) * ncell, hipMemcpyDeviceToDevice, c->stream));   // x after the start-up pass, should the iterations have to be repeated
  const size_t lds = (size_t)3 * pl.nwin_max * sizeof(double);
  const dim3 G(pl.nwg), B(POP_RED_THREADS);
  switch (pl.CP) {
    case 1: hipLaunchKernelGGL(k_cg_persist<1>, G, B, lds, c->stream, ca); break;
    case 2: hipLaunchKernelGGL(k_cg_persist<2>, G, B, lds, c->stream, ca); break;
    case 4: hipLaunchKernelGGL(k_cg_persist<4>, G, B, lds, c->stream, ca); break;
    default: hipLaunchKernelGGL(k_cg_persist<8>, G, B, lds, c->stream, ca); break;
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->chk_ev[0], c->stream));
  HIPCHK(c, hipEventSynchronize(c->chk_ev[0]));
  if (c->persist_out[2] != 0.0 || c->persist_out[0] < 0.0) {
    char b[200];
    snprintf(b, sizeof b, " [iterations %g, status %g, checks %g, %d workgroups x %d chunks, %d blocks]", c->persist_out[0], c->persist_out[2], c->persist_out[3], pl.nwg, pl.CP, v.g.nblocks);
    c->err = std::string("resident ChronGear: a wait for another workgroup's data gave up (kernels_pcg_persist.hpp)") + b;
    return 3;
  }
  c->numIterations = (int)c->persist_out[0];
  c->rmsResidual = std::sqrt(c->persist_out[1] * c->h.residualNorm);
  c->persist_used = 1; c->persist_nwg = pl.nwg; c->persist_cp = pl.CP;
  hipLaunchKernelGGL(k_halo_srcmap, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, c->stream, v.X, v.srcmap, ncell);
  HIPCHK(c, hipGetLastError());
  const bool conv = c->persist_out[3] > 0.0 && c->persist_out[1] < c->h.convergenceCriterion;
  if (!conv && c->h.convergenceCriterion != 0.0) { c->err = "POP_SolversChronGear: solver not converged"; return 2; }
  return 0;
}

int solver_pcg_fused(pop_ctx *c, SolveView &v) {
  const pop_config &cf = c->h.c;
  c->persist_used = 0;
  if (!tun_off(c->h.tun.pcg_persist) && !fused_args(c, v).presummed) {   // the rule: wherever the plan qualifies (small views); pop_tuning.pcg_persist = 0 switches it off
    const pop_ctx::PersistPlan *pl = c->persist_gave_up ? nullptr : persist_plan(c, v);
    if (pl) {
      const int e = solver_pcg_persist(c, v, *pl);
      if (e != 3) return e;
      // the resident launch did not complete its exchanges (its header: several processes on one GPU): not again in this model; this
      // solve is repeated from the same first guess with the two launches per iteration -- the same numbers
      c->persist_gave_up += 1;
      fprintf(stderr, "libpop_amd: %s -- continuing with the two-launch pcg\n", c->err.c_str());
      c->err.clear();
      HIPCHK(c, hipMemcpyAsync(v.X, pl->X0, sizeof(double) * v.g.n2 * v.g.nblocks, hipMemcpyDeviceToDevice, c->stream));
    }
  }
  const dim3 G = view_grid(v), B(POP_RED_THREADS);
  const int freq = cf.convergence_check_freq;
  SolverScalars init{}; init.eta0 = 1.0;
  HIPCHK(c, hipMemcpyAsync(c->sc, &init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(v.S0, 0, sizeof(double) * v.g.n2 * v.g.nblocks, c->stream));
  launch_fresidual<false>(c, v, fused_args(c, v));
  c->numIterations = cf.max_iterations;
  double rr = 0.0;
  const bool use_graph = (freq % 2 == 0) && !c->no_graph;
  int m = 0, lerr = 0;
  const int nint = cf.max_iterations / freq;
  const int conv = run_intervals(c, nint, [&](int) -> int {
    if (use_graph) {
      // the graph is keyed by the solution array (the time-level rotation cycles three of them)
      hipGraphExec_t exec = nullptr;
      for (auto &g : c->graphs) if (g.first == v.X) exec = g.second;
      if (!exec) {
        hipGraph_t graph;
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return -1;
        const int e = fused_interval(c, v, freq, false);
        hipError_t ce = hipStreamEndCapture(c->stream, &graph);
        if (e || ce != hipSuccess) return -1;
        if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) return -1;
        hipGraphDestroy(graph);
        c->graphs.push_back({v.X, exec});
      }
      if (hipGraphLaunch(exec, c->stream) != hipSuccess) return -1;
    } else if (fused_interval(c, v, freq, false)) return -1;
    return 1;
  }, rr, lerr);
  if (lerr) { c->err = "fused pcg: interval launch failed"; return 1; }
  if (conv >= 0) c->numIterations = (conv + 1) * freq;
  m = nint * freq;
  if (c->numIterations == cf.max_iterations && m < cf.max_iterations) {   // remainder without a check
    bool pending = false;
    for (; m < cf.max_iterations; ++m) {
      FusedArgs a = fused_args(c, v);
      launch_fpcg_a(c, v, a, pending);
      if (a.presummed) presum(c, v, a.partA, (double *)a.bsA);
      launch_fpcg_b(c, v, a, pending);
      if (a.presummed) presum(c, v, a.partB, (double *)a.bsB);
      std::swap(v.S0, v.S1);
      pending = true;
    }
    if (pending) hipLaunchKernelGGL(k_fpcg_xr, G, B, 0, c->stream, v.g, fused_args(c, v));
  }
  // ghosts of the solution as POP_SolversRun leaves them (every ghost has a source inside the view)
  const long long ncell = (long long)v.g.n2 * v.g.nblocks;
  hipLaunchKernelGGL(k_halo_srcmap, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, c->stream, v.X, v.srcmap, ncell);
  c->rmsResidual = std::sqrt(rr * c->h.residualNorm);
  HIPCHK(c, hipGetLastError());
  if (c->numIterations == cf.max_iterations && c->h.convergenceCriterion != 0.0) { c->err = "POP_SolversPCG: solver not converged"; return 2; }
  return 0;
}
SolveView local_view(pop_ctx *c) {
  SolveView v{};
  v.g = c->g; v.X = c->PS[c->newt]; v.R = c->R; v.Z = c->Z; v.S0 = c->S0; v.S1 = c->S1; v.Q = c->Q;
  v.RHS = c->RHS; v.C = c->centerWgt; v.partial = c->partial; v.blocksum = c->blocksum;
  v.srcmap = c->srcmap; v.gid = c->gid; v.nchunk = c->nchunk; v.nblocks_tot = c->h.nblocks_tot;
  return v;
}
// the same view for the fused pcg / ChronGear kernels: with land elimination active their launches cover only the chunks
// that hold an ocean cell (DevGrid::red_act); the partials of the chunks left out are zeroed once per solve
SolveView fused_view(pop_ctx *c) {
  SolveView v = local_view(c);
  if (c->g.skip && c->red_act) {
    v.g.red_act = c->red_act; v.g.red_cnt = c->red_cnt; v.g.red_nact = c->red_nact;
    hipMemsetAsync(v.partial, 0, (size_t)v.nchunk * v.g.nblocks * 2 * sizeof(double), c->stream);
  }
  return v;
}
// Replicated barotropic mode (small 2-D problems on several GPUs): the tropic distribution of the
// reference (domain.F90:433-543, POP_RedistributeBlocks around the solve, POP_SolversMod.F90:390-417,
// 481) taken to its limit -- every rank gathers RHS and the first guess of ALL blocks with one
// all-reduce of disjoint contributions, runs the fused solver on the whole 2-D domain with no
// per-iteration communication, and keeps its own blocks.  Arithmetic and iteration count equal the
// single-rank run (same blocks, same b4b sums).
// ---- fused solvers for blocks spread over ranks -------------------------------------------------------------------
// Ghosts with a source on this rank are read there (srcmap); ghosts owned by another rank need ONE exchange per
// iteration (z: the search direction and the solution at those ghosts are then advanced locally with the owner's
// arithmetic, so they never travel).  The message is packed by the kernel that produces z (FusedArgs::sendmap) and read
// in place from the receive buffer by the kernel that consumes it (rmap): no pack / unpack launches.  The dot products
// go through the b4b block-sum vector (own blocks' ordered sums, zeros elsewhere) and an all-reduce.  Forming the
// block sums in the producing kernel (last workgroup by atomic ticket) was measured and rejected: the agent-scope
// release every workgroup needs costs 20 ns per workgroup (profiles/probes/ticket_probe.hip: 77-88 us against 10 us
// for the two launches at 4 224 workgroups).
//   pcg       : k_fpcg_a(+pack) | block sums | all-reduce (launch stream)  ||  exchange z (side stream, own communicator)
//               k_fpcg_b(reads rbuf) | block sums | all-reduce            = 7 operations, 6 on the critical path
//   ChronGear : exchange z | k_fcg_a(reads rbuf) | block sums<2> | ONE all-reduce | k_fcg_b(+pack)   = 5 operations
// Convergence checks keep one interval of look-ahead (run_intervals): the residual lands in pinned host memory, the
// check that converges raises the device stop flag, and -- the all-reduced sums being the same bits on every rank --
// all ranks stop at the same check.  Bitwise the same results as the single-rank run.
struct DistSolve {
  pop_ctx *c; SolveView v; int nbt;
  FusedArgs args() const {
    FusedArgs a = fused_args(c, v);
    a.presummed = 1; a.nblocks = nbt; a.bsA = c->redbuf; a.bsB = c->redbuf + 2 * nbt;
    a.sendmap = c->sendmap; a.send_off = c->send_off; a.send_slot = c->send_slot; a.sendbuf = c->sendbuf;
    a.rmap = c->rmap; a.rbuf = c->recvbuf;
    return a;
  }
  // ordered block sums of every rank -> all ranks; NF interleaved fields at redbuf + off
  template <int NF> int allsum(const double *partial, long long off) {
    hipLaunchKernelGGL(k_block_sums_global<NF>, dim3(nbt), dim3(POP_RED_THREADS), 0, c->stream, partial, v.nchunk, c->loc_of_gid, c->redbuf + off);
    if (c->allred(c->comm_user, off, (long long)NF * nbt)) { c->err = "distributed solver: allreduce failed" + tr_err(c); return 1; }
    c->solver_ops += 2;
    return 0;
  }
  // the one-level exchange of the buffers the kernels packed: on the side stream beside the all-reduce when the
  // transport has a second communicator, else in line.  fork: the packed data is complete on the launch stream now.
  bool overlap = true;
  bool side() const { return overlap && c->xchg_side && c->comm_side && !tun_on(c->h.tun.solver_overlap_off); }
  int xchg_begin() {
    std::vector<int> peer; std::vector<long long> so, sc, ro, rc;
    long long s0 = 0, r0 = 0;
    for (auto &p : c->peers) { peer.push_back(p.rank); so.push_back(s0); sc.push_back(p.nsend); ro.push_back(r0); rc.push_back(p.nrecv); s0 += p.nsend; r0 += p.nrecv; }
    c->solver_ops += 1;
    if (side()) {
      if (hipEventRecord(c->ev_sa, c->stream) != hipSuccess || hipStreamWaitEvent(c->comm_side, c->ev_sa, 0) != hipSuccess) { c->err = "distributed solver: event failed"; return 1; }
      if (c->xchg_side(c->comm_user, (int)peer.size(), peer.data(), so.data(), sc.data(), ro.data(), rc.data())) { c->err = "distributed solver: exchange failed" + tr_err(c); return 1; }
      if (hipEventRecord(c->ev_sx, c->comm_side) != hipSuccess) { c->err = "distributed solver: event failed"; return 1; }
      return 0;
    }
    if (c->xchg(c->comm_user, (int)peer.size(), peer.data(), so.data(), sc.data(), ro.data(), rc.data())) { c->err = "distributed solver: exchange failed" + tr_err(c); return 1; }
    return 0;
  }
  int xchg_end() {   // the launch stream may read the receive buffer after this
    if (side() && hipStreamWaitEvent(c->stream, c->ev_sx, 0) != hipSuccess) { c->err = "distributed solver: event failed"; return 1; }
    return 0;
  }
  // residual + (r,r) of all ranks -> device scalars, pinned host ring, stop flag (the check of run_intervals)
  int check() {
    launch_fresidual<true>(c, v, args());
    if (allsum<1>(args().partA, 0)) return 1;
    hipLaunchKernelGGL(k_rr_blocks, dim3(1), dim3(1), 0, c->stream, (const double *)c->redbuf, nbt, c->sc, c->host_rr, c->h.convergenceCriterion);
    c->solver_ops += 2;
    return 0;
  }
};

int solver_pcg_fused_dist(pop_ctx *c) {
  const pop_config &cf = c->h.c;
  DistSolve D{c, fused_view(c), c->h.nblocks_tot};
  SolveView &v = D.v;
  const dim3 G = view_grid(v), B(POP_RED_THREADS);
  const int nbt = D.nbt, freq = cf.convergence_check_freq;
  if (!c->allred || !c->xchg || !c->redbuf || !c->sendbuf || c->red_doubles < 4LL * nbt) { c->err = "distributed pcg: no transport / reduce buffer"; return 1; }
  SolverScalars init{}; init.eta0 = 1.0;
  HIPCHK(c, hipMemcpyAsync(c->sc, &init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(v.S0, 0, sizeof(double) * v.g.n2 * v.g.nblocks, c->stream));
  launch_fresidual<false>(c, v, D.args());
  c->numIterations = cf.max_iterations;
  c->solver_ops = 0; c->solver_enq = 0;
  auto iterations = [&](int n, bool pending) -> int {
    for (int it = 0; it < n; ++it) {
      c->solver_enq += 1;
      FusedArgs a = D.args();
      if (pending) hipLaunchKernelGGL(k_fpcg_a<true>, G, B, 0, c->stream, v.g, a);
      else hipLaunchKernelGGL(k_fpcg_a<false>, G, B, 0, c->stream, v.g, a);
      c->solver_ops += 1;
      if (D.xchg_begin() || D.allsum<1>(a.partA, 0) || D.xchg_end()) return 1;
      launch_fpcg_b(c, v, a, pending);
      c->solver_ops += 1;
      if (D.allsum<1>(a.partB, 2 * nbt)) return 1;
      std::swap(v.S0, v.S1);
      pending = true;
    }
    return 0;
  };
  double rr = 0.0;
  int lerr = 0;
  const int nint = cf.max_iterations / freq;
  const int conv = run_intervals(c, nint, [&](int) -> int {
    if (iterations(freq, false)) return -1;
    hipLaunchKernelGGL(k_fpcg_xr, G, B, 0, c->stream, v.g, D.args());
    c->solver_ops += 1;
    if (D.check()) return -1;
    return 1;
  }, rr, lerr);
  if (lerr) { if (c->err.empty()) c->err = "distributed pcg: interval launch failed"; return 1; }
  if (conv >= 0) c->numIterations = (conv + 1) * freq;
  if (c->numIterations == cf.max_iterations && nint * freq < cf.max_iterations) {   // remainder without a check
    if (iterations(cf.max_iterations - nint * freq, false)) return 1;
    hipLaunchKernelGGL(k_fpcg_xr, G, B, 0, c->stream, v.g, D.args());
  }
  // ghosts of the solution as POP_SolversRun leaves them: remote ones were advanced with their owners'
  // arithmetic, the ones with a source on this rank are copied now (srcmap is the identity on remote ghosts)
  const long long ncell = (long long)v.g.n2 * v.g.nblocks;
  hipLaunchKernelGGL(k_halo_srcmap, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, c->stream, v.X, v.srcmap, ncell);
  c->S0 = v.S0; c->S1 = v.S1;
  c->rmsResidual = std::sqrt(rr * c->h.residualNorm);
  HIPCHK(c, hipGetLastError());
  if (c->numIterations == cf.max_iterations && c->h.convergenceCriterion != 0.0) { c->err = "POP_SolversPCG: solver not converged"; return 2; }
  return 0;
}

int solver_pcg_replicated(pop_ctx *c) {
  SolveView &v = c->gv;
  const size_t n2 = c->g.n2, NG = n2 * c->h.nblocks_tot;
  double *PN = c->PS[c->newt];
  HIPCHK(c, hipMemsetAsync(c->redbuf, 0, sizeof(double) * 2 * NG, c->stream));
  for (int lb = 0; lb < c->g.nblocks; ++lb) {
    const size_t go = (size_t)(c->h.local_ids[lb] - 1) * n2;
    HIPCHK(c, hipMemcpyAsync(c->redbuf + go, c->RHS + lb * n2, n2 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->redbuf + NG + go, PN + lb * n2, n2 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }
  if (c->allred(c->comm_user, 0, (long long)(2 * NG))) { c->err = "replicated solve: allreduce callback failed" + tr_err(c); return 1; }
  HIPCHK(c, hipMemcpyAsync(v.RHS, c->redbuf, NG * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(v.X, c->redbuf + NG, NG * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  hipLaunchKernelGGL(k_center_all, dim3((unsigned)((NG + 255) / 256)), dim3(256), 0, c->stream, v.g, step_params(c), c->gTAREA, c->gKMT, v.C, (long long)NG);
  const int e = solver_pcg_fused(c, v);
  if (e) return e;
  for (int lb = 0; lb < c->g.nblocks; ++lb) {
    const size_t go = (size_t)(c->h.local_ids[lb] - 1) * n2;
    HIPCHK(c, hipMemcpyAsync(PN + lb * n2, v.X + go, n2 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }
  return 0;
}

// ChronGear (POP_SolversMod.F90:1960-2266), diagonal or EVP preconditioner
int solver_chrongear(pop_ctx *c) {
  const pop_config &cf = c->h.c;
  const dim3 G = grid_2d(c), B(POP_RED_THREADS);
  SolverScalars init{};
  HIPCHK(c, hipMemcpyAsync(c->sc, &init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  SolverArgs a = solver_args(c);
  hipLaunchKernelGGL(k_residual<false>, grid_2d(c), B, 0, c->stream, c->g, a);
  if (halo_update(c, c->R, 1)) return 1;
  if (c->use_evp) {   // :2009-2032
    if (evp_apply(c, c->R, c->Z) || halo_update(c, c->Z, 1)) return 1;
    hipLaunchKernelGGL(k_cg_init<true>, grid_2d(c), B, 0, c->stream, c->g, a);
  } else hipLaunchKernelGGL(k_cg_init<false>, grid_2d(c), B, 0, c->stream, c->g, a);
  if (halo_update(c, c->Q, 1)) return 1;
  if (reduce_finish<2>(c, FIN_CG_INIT)) return 1;
  hipLaunchKernelGGL(k_cg_update<true>, grid_2d(c), B, 0, c->stream, c->g, a);
  c->numIterations = cf.max_iterations;
  double rr = 0.0;
  for (int m = 1; m <= cf.max_iterations; ++m) {
    if (c->use_evp) { if (evp_apply(c, c->R, c->Z)) return 1; }
    else hipLaunchKernelGGL(k_cg_z, G, B, 0, c->stream, c->g, a);
    if (halo_update(c, c->Z, 1)) return 1;
    hipLaunchKernelGGL(k_cg_az, G, B, 0, c->stream, c->g, a);
    if (reduce_finish<2>(c, FIN_CG_ITER)) return 1;
    hipLaunchKernelGGL(k_cg_update<false>, grid_2d(c), B, 0, c->stream, c->g, a);
    if (m % cf.convergence_check_freq == 0) {
      hipLaunchKernelGGL(k_residual<true>, grid_2d(c), B, 0, c->stream, c->g, a);
      if (halo_update(c, c->R, 1)) return 1;
      if (reduce_finish<1>(c, FIN_RR)) return 1;
      SolverScalars s;
      if (read_scalars(c, &s)) return 1;
      rr = s.rr;
      if (rr < c->h.convergenceCriterion) { c->numIterations = m; break; }
    }
  }
  c->rmsResidual = std::sqrt(rr * c->h.residualNorm);
  HIPCHK(c, hipGetLastError());
  if (c->numIterations == cf.max_iterations && c->h.convergenceCriterion != 0.0) { c->err = "POP_SolversChronGear: solver not converged"; return 2; }
  return 0;
}

// ChronGear, fused form for one rank: the start-up pass as in solver_chrongear, then two launches per iteration
// (k_fcg_a, k_fcg_b: the z halo folded into the matvec through srcmap, scalar recurrences recomputed by every
// workgroup from the ordered totals) and one hipGraph replay per check interval.  Same arithmetic and summation
// order as solver_chrongear: bitwise the same solution and iteration count.
// compacted launches (DevGrid::red_act): the iterations store pairs of partials, the checks single ones, in the same slots;
// chunks that are not launched cannot zero theirs, so the slots are cleared whenever the layout changes
static void cg_clear_partials(pop_ctx *c, const SolveView &v) {
  if (v.g.red_act) hipMemsetAsync(v.partial, 0, (size_t)v.nchunk * v.g.nblocks * 2 * sizeof(double), c->stream);
}
static int cg_fused_iterations(pop_ctx *c, SolveView &v, int n, int &par) {
  const dim3 G = view_grid(v), B(POP_RED_THREADS);
  for (int it = 0; it < n; ++it) {
    FusedArgs a = fused_args(c, v);
    a.AZ = c->AZ; a.A0R = v.S1;
    if (a.presummed && (v.g.nxb & 1) == 0 && !v.g.red_tiles && !c->fpcg_one_cell) hipLaunchKernelGGL(k_fcg_a2, G, dim3(POP_RED_THREADS / 2), 0, c->stream, v.g, a);
    else hipLaunchKernelGGL(k_fcg_a, G, B, 0, c->stream, v.g, a);
    if (a.presummed && presum_relay(c, v)) hipLaunchKernelGGL((k_block_sums_relay<2, POP_RELAY_LMAX>), dim3(v.g.nblocks, 2), dim3(1024), 0, c->stream, (const double *)a.partA, v.nchunk, (const int *)c->iota, (double *)a.bsA);
    else if (a.presummed) hipLaunchKernelGGL(k_block_sums<2>, dim3(v.g.nblocks), dim3(POP_RED_THREADS), 0, c->stream, (const double *)a.partA, v.nchunk, (const int *)c->iota, (double *)a.bsA);
    hipLaunchKernelGGL(k_fcg_b, G, B, 0, c->stream, v.g, a, par);
    par = 1 - par;
  }
  return 0;
}
static int cg_fused_interval(pop_ctx *c, SolveView &v, int freq) {
  int par = 0;
  cg_fused_iterations(c, v, freq, par);
  cg_clear_partials(c, v);
  launch_fresidual<true>(c, v, fused_args(c, v));
  hipLaunchKernelGGL(k_rr_total, dim3(1), dim3(POP_RED_THREADS), 0, c->stream, (const double *)v.partial, v.nchunk, v.g.nblocks, c->sc, c->host_rr, c->h.convergenceCriterion);
  cg_clear_partials(c, v);
  return 0;
}
int solver_chrongear_fused(pop_ctx *c) {
  const pop_config &cf = c->h.c;
  SolveView v = fused_view(c);
  const dim3 B(POP_RED_THREADS);
  const int freq = cf.convergence_check_freq;
  const long long a2 = (long long)c->g.n2 * c->g.nblocks;
  SolverScalars init{};
  HIPCHK(c, hipMemcpyAsync(c->sc, &init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  SolverArgs a = solver_args(c);
  hipLaunchKernelGGL(k_residual<false>, grid_2d(c), B, 0, c->stream, c->g, a);
  if (halo_update(c, c->R, 1)) return 1;
  hipLaunchKernelGGL(k_cg_init<false>, grid_2d(c), B, 0, c->stream, c->g, a);
  if (halo_update(c, c->Q, 1)) return 1;
  if (reduce_finish<2>(c, FIN_CG_INIT)) return 1;
  hipLaunchKernelGGL(k_cg_update<true>, grid_2d(c), B, 0, c->stream, c->g, a);
  hipLaunchKernelGGL(k_pcsi_a0r, dim3((unsigned)((a2 + 255) / 256)), dim3(256), 0, c->stream, (const double *)c->centerWgt, v.S1, a2);
  c->persist_used = 0;
  {   // small views: the iterations as one resident launch (k_cg_persist); pop_tuning.pcg_persist = 0 switches it off
    FusedArgs fa = fused_args(c, v);
    fa.AZ = c->AZ; fa.A0R = v.S1;
    if (!tun_off(c->h.tun.pcg_persist) && !fa.presummed && !c->persist_gave_up &&
        v.g.nblocks * ((v.nchunk + POP_RED_THREADS - 1) / POP_RED_THREADS) <= POP_CGP_MAXP) {
      if (const pop_ctx::PersistPlan *pl = persist_plan(c, v)) {
        const int e = solver_cg_persist(c, v, *pl, fa);
        if (e != 3) return e;
        // the resident launch did not complete its exchanges: not again in this model.  It changed x only at its very end, if at all: x of the
        // start-up pass is restored; r, s, q and the scalars were only read
        c->persist_gave_up += 1;
        fprintf(stderr, "libpop_amd: %s -- continuing with the two-launch ChronGear\n", c->err.c_str());
        c->err.clear();
        HIPCHK(c, hipMemcpyAsync(v.X, pl->X0, sizeof(double) * a2, hipMemcpyDeviceToDevice, c->stream));
      }
    }
  }
  c->numIterations = cf.max_iterations;
  double rr = 0.0;
  const bool use_graph = (freq % 2 == 0) && !c->no_graph;   // even: the (rho, sigma) ping-pong ends where it started
  int m = 0, lerr = 0;
  const int nint = cf.max_iterations / freq;
  const int conv = run_intervals(c, nint, [&](int i) -> int {
    if (use_graph) {
      hipGraphExec_t exec = nullptr;
      for (auto &g : c->graphs) if (g.first == v.X) exec = g.second;
      if (!exec) {
        hipGraph_t graph;
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return -1;
        const int e = cg_fused_interval(c, v, freq);
        hipError_t ce = hipStreamEndCapture(c->stream, &graph);
        if (e || ce != hipSuccess) return -1;
        if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) return -1;
        hipGraphDestroy(graph);
        c->graphs.push_back({v.X, exec});
      }
      if (hipGraphLaunch(exec, c->stream) != hipSuccess) return -1;
    } else {
      int par = (i * freq) & 1;   // odd freq: the ping-pong slot carries over between intervals
      cg_fused_iterations(c, v, freq, par);
      cg_clear_partials(c, v);
      launch_fresidual<true>(c, v, fused_args(c, v));
      hipLaunchKernelGGL(k_rr_total, dim3(1), dim3(POP_RED_THREADS), 0, c->stream, (const double *)v.partial, v.nchunk, v.g.nblocks, c->sc, c->host_rr, c->h.convergenceCriterion);
      cg_clear_partials(c, v);
    }
    return 1;
  }, rr, lerr);
  if (lerr) { c->err = "fused ChronGear: interval launch failed"; return 1; }
  if (conv >= 0) c->numIterations = (conv + 1) * freq;
  m = nint * freq;
  if (c->numIterations == cf.max_iterations && m < cf.max_iterations) {   // remainder without a check
    int par = m & 1;
    cg_fused_iterations(c, v, cf.max_iterations - m, par);
  }
  hipLaunchKernelGGL(k_halo_srcmap, dim3((unsigned)((a2 + 255) / 256)), dim3(256), 0, c->stream, v.X, v.srcmap, a2);
  c->rmsResidual = std::sqrt(rr * c->h.residualNorm);
  HIPCHK(c, hipGetLastError());
  if (c->numIterations == cf.max_iterations && c->h.convergenceCriterion != 0.0) { c->err = "POP_SolversChronGear: solver not converged"; return 2; }
  return 0;
}

// ChronGear for blocks spread over ranks (see DistSolve): start-up pass as in solver_chrongear, then per iteration
// exchange z | k_fcg_a | block sums of (r,z), (az,z) | ONE all-reduce | k_fcg_b, which also packs the next z
int solver_chrongear_fused_dist(pop_ctx *c) {
  const pop_config &cf = c->h.c;
  DistSolve D{c, local_view(c), c->h.nblocks_tot};   // whole launches: the (r,r) partials of the checks and the pairs of the iterations share slots
  D.overlap = false;   // nothing runs beside the exchange here: the next kernel needs it
  SolveView &v = D.v;
  const dim3 G = view_grid(v), B(POP_RED_THREADS);
  const int nbt = D.nbt, freq = cf.convergence_check_freq;
  const long long a2 = (long long)c->g.n2 * c->g.nblocks;
  if (!c->allred || !c->xchg || !c->redbuf || !c->sendbuf || c->red_doubles < 4LL * nbt) { c->err = "distributed ChronGear: no transport / reduce buffer"; return 1; }
  SolverScalars init{};
  HIPCHK(c, hipMemcpyAsync(c->sc, &init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  SolverArgs sa = solver_args(c);
  hipLaunchKernelGGL(k_residual<false>, grid_2d(c), B, 0, c->stream, c->g, sa);
  if (halo_update(c, c->R, 1)) return 1;
  hipLaunchKernelGGL(k_cg_init<false>, grid_2d(c), B, 0, c->stream, c->g, sa);
  if (halo_update(c, c->Q, 1)) return 1;
  if (reduce_finish<2>(c, FIN_CG_INIT)) return 1;
  hipLaunchKernelGGL(k_cg_update<true>, grid_2d(c), B, 0, c->stream, c->g, sa);
  hipLaunchKernelGGL(k_pcsi_a0r, dim3((unsigned)((a2 + 255) / 256)), dim3(256), 0, c->stream, (const double *)c->centerWgt, v.S1, a2);
  // z of the first iteration at the neighbours' ghosts: z = r*A0R on the whole array, packed and exchanged once
  hipLaunchKernelGGL(k_cg_z, grid_2d(c), B, 0, c->stream, c->g, sa);
  if (c->nsend_all) hipLaunchKernelGGL(k_halo_pack_all, dim3((c->nsend_all + 255) / 256, 1), dim3(256), 0, c->stream, (const double *)c->Z, c->sa_src, c->sa_start, c->sa_cnt, c->nsend_all, c->sendbuf, 1, c->g.n2);
  if (D.xchg_begin()) return 1;
  c->numIterations = cf.max_iterations;
  c->solver_ops = 0; c->solver_enq = 0;
  auto args = [&]() { FusedArgs a = D.args(); a.AZ = c->AZ; a.A0R = v.S1; return a; };
  auto iterations = [&](int n, int &par) -> int {
    for (int it = 0; it < n; ++it) {
      c->solver_enq += 1;
      FusedArgs a = args();
      if ((v.g.nxb & 1) == 0 && !v.g.red_tiles && !c->fpcg_one_cell && (long long)v.nchunk * v.g.nblocks > 2048) hipLaunchKernelGGL(k_fcg_a2, G, dim3(POP_RED_THREADS / 2), 0, c->stream, v.g, a);
      else hipLaunchKernelGGL(k_fcg_a, G, B, 0, c->stream, v.g, a);
      c->solver_ops += 1;
      if (D.allsum<2>(a.partA, 0)) return 1;
      hipLaunchKernelGGL(k_fcg_b, G, B, 0, c->stream, v.g, a, par);
      c->solver_ops += 1;
      if (D.xchg_begin()) return 1;
      par = 1 - par;
    }
    return 0;
  };
  double rr = 0.0;
  int lerr = 0;
  const int nint = cf.max_iterations / freq;
  const int conv = run_intervals(c, nint, [&](int i) -> int {
    int par = (i * freq) & 1;
    if (iterations(freq, par)) return -1;
    // r = b - A x; its z = r*A0R is packed by the residual kernel and exchanged for the next interval
    {
      FusedArgs a = args();
      const dim3 GG = view_grid(v);
      if ((v.g.nxb & 1) == 0 && !v.g.red_tiles && !c->fpcg_one_cell && (long long)v.nchunk * v.g.nblocks > 2048) hipLaunchKernelGGL(k_fresidual2<true>, GG, dim3(POP_RED_THREADS / 2), 0, c->stream, v.g, a);
      else hipLaunchKernelGGL(k_fresidual<true>, GG, dim3(POP_RED_THREADS), 0, c->stream, v.g, a);
      if (D.allsum<1>(a.partA, 0)) return -1;
      hipLaunchKernelGGL(k_rr_blocks, dim3(1), dim3(1), 0, c->stream, (const double *)c->redbuf, nbt, c->sc, c->host_rr, c->h.convergenceCriterion);
      if (D.xchg_begin()) return -1;
      c->solver_ops += 2;
    }
    return 1;
  }, rr, lerr);
  if (lerr) { if (c->err.empty()) c->err = "distributed ChronGear: interval launch failed"; return 1; }
  if (conv >= 0) c->numIterations = (conv + 1) * freq;
  if (c->numIterations == cf.max_iterations && nint * freq < cf.max_iterations) {   // remainder without a check
    int par = (nint * freq) & 1;
    if (iterations(cf.max_iterations - nint * freq, par)) return 1;
  }
  hipLaunchKernelGGL(k_halo_srcmap, dim3((unsigned)((a2 + 255) / 256)), dim3(256), 0, c->stream, v.X, v.srcmap, a2);
  c->rmsResidual = std::sqrt(rr * c->h.residualNorm);
  HIPCHK(c, hipGetLastError());
  if (c->numIterations == cf.max_iterations && c->h.convergenceCriterion != 0.0) { c->err = "POP_SolversChronGear: solver not converged"; return 2; }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// P-CSI (POP_SolversMod.F90:1510-1835), diagonal or EVP preconditioner (EVP: operation-by-operation form only).  kernels_pcsi.hpp describes the
// fused one-launch-per-iteration form; solver_pcsi is the operation-by-operation form that also
// serves multi-rank runs (one halo update per iteration, no collective except at the checks).
// ---------------------------------------------------------------------------------------------
__global__ void k_set_int(int *p, int v) { *p = v; }

int pcsi_check_start(const pop_ctx *c) { return c->h.c.convergence_check_start > 0 ? c->h.c.convergence_check_start : 60; }   // convergenceCheckStart :636

int solver_pcsi(pop_ctx *c) {
  const pop_config &cf = c->h.c;
  const dim3 B(POP_RED_THREADS);
  const long long a2 = (long long)c->g.n2 * c->g.nblocks;
  const dim3 G1((unsigned)((a2 + 255) / 256)), B1(256);
  SolverScalars init{};
  HIPCHK(c, hipMemcpyAsync(c->sc, &init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  SolverArgs a = solver_args(c);
  hipLaunchKernelGGL(k_residual<false>, grid_2d(c), B, 0, c->stream, c->g, a);
  auto precond = [&]() -> int {   // r' = M^-1 r in place (:1646-1660, :1738-1752)
    if (c->use_evp) {
      if (evp_apply(c, c->R, c->Z)) return 1;
      HIPCHK(c, hipMemcpyAsync(c->R, c->Z, sizeof(double) * a2, hipMemcpyDeviceToDevice, c->stream));
    } else hipLaunchKernelGGL(k_pcsi_precond, G1, B1, 0, c->stream, c->g, c->R, (const double *)c->centerWgt, a2);
    return 0;
  };
  if (precond()) return 1;
  if (halo_update(c, c->R, 1)) return 1;
  hipLaunchKernelGGL(k_pcsi_update<true>, G1, B1, 0, c->stream, (const double *)c->R, c->Q, a.X, a2, (const double *)c->pcsi_omega,
                     (const int *)c->pcsi_base, 0, c->pcsi_csy);
  hipLaunchKernelGGL(k_residual<false>, grid_2d(c), B, 0, c->stream, c->g, a);
  c->numIterations = cf.max_iterations;
  double rr = 0.0;
  const int start = pcsi_check_start(c);
  for (int m = 1; m <= cf.max_iterations; ++m) {
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, c->stream, c->pcsi_base, m - 1);
    if (precond()) return 1;
    if (halo_update(c, c->R, 1)) return 1;
    hipLaunchKernelGGL(k_pcsi_update<false>, G1, B1, 0, c->stream, (const double *)c->R, c->Q, a.X, a2, (const double *)c->pcsi_omega,
                       (const int *)c->pcsi_base, 1, c->pcsi_csy);
    const bool check = (m % cf.convergence_check_freq == 0) && m >= start;
    if (check) hipLaunchKernelGGL(k_residual<true>, grid_2d(c), B, 0, c->stream, c->g, a);
    else hipLaunchKernelGGL(k_residual<false>, grid_2d(c), B, 0, c->stream, c->g, a);
    if (check) {
      if (reduce_finish<1>(c, FIN_RR)) return 1;
      SolverScalars s;
      if (read_scalars(c, &s)) return 1;
      rr = s.rr;
      if (rr < c->h.convergenceCriterion) { c->numIterations = m; break; }
    }
  }
  c->rmsResidual = std::sqrt(rr * c->h.residualNorm);
  HIPCHK(c, hipGetLastError());
  if (c->numIterations == cf.max_iterations && c->h.convergenceCriterion != 0.0) { c->err = "POP_SolversPCSI: solver not converged"; return 2; }
  return 0;
}

// fused form; state ping-pongs between (X, R, Q) and (Z, AZ, S1)
struct PcsiBufs { double *X[2], *R[2], *Q[2]; };
static PcsiArgs pcsi_args(pop_ctx *c, const PcsiBufs &bf, int in, int j) {
  PcsiArgs a{};
  a.Xi = bf.X[in]; a.Ri = bf.R[in]; a.Qi = bf.Q[in]; a.Xo = bf.X[1 - in]; a.Ro = bf.R[1 - in]; a.Qo = bf.Q[1 - in];
  a.Bv = c->RHS; a.C = c->centerWgt; a.A0R = c->S0; a.omega = c->pcsi_omega; a.base = c->pcsi_base; a.srcmap = c->srcmap; a.partial = c->partial; a.sc = c->sc;
  a.csy = c->pcsi_csy; a.j = j; a.nchunk = c->nchunk;
  if (c->use_evp) { a.raw_r = 1; a.Ri = c->R; a.Ro = c->AZ; }   // r' = M^-1 r in R (read), the residual itself to AZ (written): evp_apply(AZ -> R) follows every step
  return a;
}
// DevGrid of the single-rank fused P-CSI launches: with land elimination active, the compacted chunk list (DevGrid::red_act)
static DevGrid pcsi_grid(const pop_ctx *c) {
  DevGrid g = c->g;
  if (c->g.skip && c->red_act && c->peers.empty()) { g.red_act = c->red_act; g.red_cnt = c->red_cnt; g.red_nact = c->red_nact; }
  return g;
}
// `freq` steps starting from buffer `in`; the last one also forms (r,r) -> host when with_rr
// two iterations per launch (k_pcsi_step_x2): how many of the n iterations of an interval go in pairs -- the last two stay single (the check
// needs the chunk partials of (r, r) of k_pcsi_step2, and a single step before it keeps the pairs aligned for every n)
// An interval of an even number of iterations goes in pairs throughout: the pair before a check leaves the residual itself in a scratch field
// and k_pcsi_rr_chunks forms the chunk partials of (r, r) from it.  An odd interval: pairs, then one single step (which carries the check).
static int pcsi_pairs(const pop_ctx *c, int n) { return c->pcsi_two_step ? n / 2 : 0; }
// one pair; with_raw: the residual itself to pcsi_raw as well, and the chunk partials of (r, r) from it
static void pcsi_launch_pair(pop_ctx *c, const DevGrid &gg, PcsiArgs a, bool with_raw) {
  const int tiles_i = (gg.nxb - 2 * NGHOST + 63) / 64, tiles_j = (gg.nyb - 2 * NGHOST + 7) / 8;
  const dim3 GT(lds_launch_x<8>(gg, tiles_i, tiles_j), gg.nblocks), TB(64, 8);
  a.jfold = c->pcsi_jfold;
  double *raw = with_raw ? c->pcsi_raw : nullptr;
  if (a.jfold) {
    if (with_raw) hipLaunchKernelGGL((k_pcsi_step_x2<true, true>), GT, TB, 0, c->stream, gg, a, raw);
    else hipLaunchKernelGGL((k_pcsi_step_x2<false, true>), GT, TB, 0, c->stream, gg, a, raw);
  } else if (with_raw) hipLaunchKernelGGL((k_pcsi_step_x2<true, false>), GT, TB, 0, c->stream, gg, a, raw);
  else hipLaunchKernelGGL((k_pcsi_step_x2<false, false>), GT, TB, 0, c->stream, gg, a, raw);
  if (with_raw) hipLaunchKernelGGL(k_pcsi_rr_chunks, dim3(red_grid_x(gg), gg.nblocks), dim3(POP_RED_THREADS), 0, c->stream, gg, a, (const double *)raw);
}
static int pcsi_launches(const pop_ctx *c, int n) { return n - pcsi_pairs(c, n); }
static void pcsi_interval(pop_ctx *c, const PcsiBufs &bf, int in, int freq, bool with_rr) {
  const DevGrid gg = pcsi_grid(c);
  const dim3 G(red_grid_x(gg), gg.nblocks), B(POP_RED_THREADS);
  int j0 = 1;
  if (c->pcsi_evp_fused) {   // EVP: one launch per iteration (the step and the sub-block solves); r' ping-pongs with x and dx
    const dim3 GE((unsigned)((c->evp.S + POP_EVP_SB - 1) / POP_EVP_SB));
    for (int j = 1; j <= freq; ++j) {
      PcsiArgs a = pcsi_args(c, bf, in, j);
      a.raw_r = 0; a.Ri = bf.R[in]; a.Ro = bf.R[1 - in];
      if (j == freq && with_rr) {
        hipLaunchKernelGGL(k_pcsi_evp_step<true>, GE, dim3(64), 0, c->stream, c->evp, gg, a, c->pcsi_raw);
        hipLaunchKernelGGL(k_pcsi_rr_chunks, G, B, 0, c->stream, gg, a, (const double *)c->pcsi_raw);
      } else hipLaunchKernelGGL(k_pcsi_evp_step<false>, GE, dim3(64), 0, c->stream, c->evp, gg, a, (double *)nullptr);
      in = 1 - in;
    }
    if (with_rr) hipLaunchKernelGGL(k_rr_total, dim3(1), dim3(POP_RED_THREADS), 0, c->stream, (const double *)c->partial, c->nchunk, c->g.nblocks, c->sc, c->host_rr, c->h.convergenceCriterion);
    return;
  }
  const int npairs = pcsi_pairs(c, freq);
  for (int p = 0; p < npairs; ++p, j0 += 2) {
    pcsi_launch_pair(c, gg, pcsi_args(c, bf, in, j0), with_rr && j0 + 1 == freq);   // (the last pair of an even interval that ends in a check)
    in = 1 - in;
  }
  for (int j = j0; j <= freq; ++j) {
    const PcsiArgs a = pcsi_args(c, bf, in, j);
    if (c->pcsi_two_cell) {
      const bool rr = j == freq && with_rr;
      if (rr && c->use_evp) hipLaunchKernelGGL((k_pcsi_step2<true, true>), G, dim3(POP_RED_THREADS / 2), 0, c->stream, gg, a);
      else if (c->use_evp) hipLaunchKernelGGL((k_pcsi_step2<false, true>), G, dim3(POP_RED_THREADS / 2), 0, c->stream, gg, a);
      else if (rr) hipLaunchKernelGGL((k_pcsi_step2<true>), G, dim3(POP_RED_THREADS / 2), 0, c->stream, gg, a);
      else hipLaunchKernelGGL((k_pcsi_step2<false>), G, dim3(POP_RED_THREADS / 2), 0, c->stream, gg, a);
    } else if (j == freq && with_rr) hipLaunchKernelGGL((k_pcsi_step<false, true>), G, B, 0, c->stream, gg, a);
    else hipLaunchKernelGGL((k_pcsi_step<false, false>), G, B, 0, c->stream, gg, a);
    if (c->use_evp) evp_apply(c, c->AZ, c->R);
    in = 1 - in;
  }
  if (with_rr) {
    hipLaunchKernelGGL(k_rr_total, dim3(1), dim3(POP_RED_THREADS), 0, c->stream, (const double *)c->partial, c->nchunk, c->g.nblocks, c->sc, c->host_rr, c->h.convergenceCriterion);
  }
}
int solver_pcsi_fused(pop_ctx *c) {
  const pop_config &cf = c->h.c;
  const dim3 G = grid_2d(c), B(POP_RED_THREADS);
  const int freq = cf.convergence_check_freq, start = pcsi_check_start(c);
  PcsiBufs bf{{c->PS[c->newt], c->Z}, {c->R, c->AZ}, {c->Q, c->S1}};   // (with EVP the residual pair is fixed: pcsi_args)
  if (c->pcsi_evp_fused) std::swap(bf.R[0], bf.R[1]);   // ... except in the one-launch form: the start-up step leaves r' in R, which is then the half the first iteration reads
  SolverScalars init{};
  HIPCHK(c, hipMemcpyAsync(c->sc, &init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  // r0 = b - A x0 (ghosts of x0 read at their sources), then the start-up step x1 = x0 + r0'/gamma, r1 = b - A x1
  {
    SolveView v = local_view(c);
    const long long a2 = (long long)c->g.n2 * c->g.nblocks;
    hipLaunchKernelGGL(k_pcsi_a0r, dim3((unsigned)((a2 + 255) / 256)), dim3(256), 0, c->stream, (const double *)c->centerWgt, c->S0, a2);
    if (c->use_evp) {   // r0 into AZ, r0' = M^-1 r0 by the sub-block solves into R; both start from zero (cells no kernel writes are never read)
      HIPCHK(c, hipMemsetAsync(c->AZ, 0, sizeof(double) * a2, c->stream));
      HIPCHK(c, hipMemsetAsync(c->R, 0, sizeof(double) * a2, c->stream));
      v.R = c->AZ;
      hipLaunchKernelGGL(k_fresidual<false>, G, B, 0, c->stream, c->g, fused_args(c, v));
      if (evp_apply(c, c->AZ, c->R)) return 1;
    } else {
      hipLaunchKernelGGL(k_fresidual<false>, G, B, 0, c->stream, c->g, fused_args(c, v));
      hipLaunchKernelGGL(k_pcsi_scale, dim3((c->g.n2 + 255) / 256, c->g.nblocks), dim3(256), 0, c->stream, c->g, c->R, (const double *)c->S0);
    }
  }
  hipLaunchKernelGGL((k_pcsi_step<true, false>), G, B, 0, c->stream, c->g, pcsi_args(c, bf, 0, 0));
  if (c->use_evp && evp_apply(c, c->AZ, c->R)) return 1;
  c->persist_used = 0;
  if (!c->use_evp && !tun_off(c->h.tun.pcg_persist) && !c->persist_gave_up) {
    // small views: the iterations as one resident launch (k_pcsi_persist: neighbour waits only, grid-wide exchanges at the checks)
    SolveView v = local_view(c);
    const FusedArgs fa = fused_args(c, v);
    const pop_ctx::PersistPlan *pl = fa.presummed ? nullptr : persist_plan(c, v);
    if (pl) {
      const long long ncell = (long long)v.g.n2 * v.g.nblocks;
      PcsiPersistArgs pa{};
      PersistArgs &a = pa.p;
      a.X = bf.X[0]; a.Bv = c->RHS; a.C = c->centerWgt; a.WNo = v.g.WNo; a.WEa = v.g.WEa; a.WNE = v.g.WNE; a.mMask8 = v.g.mMask8;
      a.nxb = v.g.nxb; a.nchunk = v.nchunk; a.nblocks = v.g.nblocks; a.nslots = pl->nslots; a.ncell = ncell;
      a.own_q = pl->own_q; a.nbr = pl->nbr; a.halo_off = pl->halo_off; a.halo_q = pl->halo_q; a.W = pl->W;
      a.epoch = (++c->persist_epoch) << 32;
      a.max_iter = cf.max_iterations; a.freq = freq; a.criterion = c->h.convergenceCriterion; a.out = c->persist_out; a.wait_ticks = 200000000ULL;
      pa.Xin = bf.X[1]; pa.Rin = bf.R[1]; pa.Qin = bf.Q[1]; pa.A0R = c->S0; pa.omega = c->pcsi_omega; pa.csy = c->pcsi_csy; pa.start = start;
      c->persist_out[0] = -1.0; c->persist_out[1] = 0.0; c->persist_out[2] = 0.0; c->persist_out[3] = 0.0;
      const size_t lds = (size_t)3 * pl->nwin_max * sizeof(double);
      const dim3 GP(pl->nwg);
      switch (pl->CP) {
        case 1: hipLaunchKernelGGL(k_pcsi_persist<1>, GP, B, lds, c->stream, pa); break;
        case 2: hipLaunchKernelGGL(k_pcsi_persist<2>, GP, B, lds, c->stream, pa); break;
        case 4: hipLaunchKernelGGL(k_pcsi_persist<4>, GP, B, lds, c->stream, pa); break;
        default: hipLaunchKernelGGL(k_pcsi_persist<8>, GP, B, lds, c->stream, pa); break;
      }
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipEventRecord(c->chk_ev[0], c->stream));
      HIPCHK(c, hipEventSynchronize(c->chk_ev[0]));
      if (c->persist_out[2] == 0.0 && c->persist_out[0] >= 0.0) {
        c->numIterations = (int)c->persist_out[0];
        c->rmsResidual = std::sqrt(c->persist_out[1] * c->h.residualNorm);
        c->persist_used = 1; c->persist_nwg = pl->nwg; c->persist_cp = pl->CP;
        hipLaunchKernelGGL(k_halo_srcmap, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, c->stream, bf.X[0], c->srcmap, ncell);
        HIPCHK(c, hipGetLastError());
        const bool conv = c->persist_out[3] > 0.0 && c->persist_out[1] < c->h.convergenceCriterion;
        if (!conv && c->h.convergenceCriterion != 0.0) { c->err = "POP_SolversPCSI: solver not converged"; return 2; }
        return 0;
      }
      // a wait gave up: the launch wrote its solution array only at its end (x of the start-up step is still in the other half of the pair,
      // r', dx were only read): the iterations are repeated by the launches below, and the resident form is not used again in this model
      c->persist_gave_up += 1;
      fprintf(stderr, "libpop_amd: resident P-CSI: a wait for another workgroup's data gave up -- continuing with one launch per iteration\n");
    }
  }
  if (pcsi_grid(c).red_act)   // compacted launches from here on: the partials of the chunks that are left out must read as zero
    HIPCHK(c, hipMemsetAsync(c->partial, 0, (size_t)c->nchunk * c->g.nblocks * 2 * sizeof(double), c->stream));
  int in = 1;
  c->numIterations = cf.max_iterations;
  double rr = 0.0;
  // intervals of `freq` steps (the last one may be shorter); those that end on a multiple of freq at or after
  // convergenceCheckStart carry a check.  in_before[i]: ping-pong half interval i starts from
  const int nint = (cf.max_iterations + freq - 1) / freq;
  std::vector<int> in_after(nint + 1, in);
  int lerr = 0;
  const int conv = run_intervals(c, nint, [&](int i) -> int {
    const int m = i * freq, n = std::min(freq, cf.max_iterations - m);
    const bool with_rr = (n == freq) && (m + n >= start);
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, c->stream, c->pcsi_base, m);
    if (!c->no_graph && n == freq) {
      const int variant = in * 2 + (with_rr ? 1 : 0);
      hipGraphExec_t exec = nullptr;
      for (auto &gk : c->pcsi_graphs) if (gk.first.first == bf.X[0] && gk.first.second == variant) exec = gk.second;
      if (!exec) {
        hipGraph_t graph;
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return -1;
        pcsi_interval(c, bf, in, n, with_rr);
        if (hipStreamEndCapture(c->stream, &graph) != hipSuccess) return -1;
        if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) return -1;
        hipGraphDestroy(graph);
        c->pcsi_graphs.push_back({{bf.X[0], variant}, exec});
      }
      if (hipGraphLaunch(exec, c->stream) != hipSuccess) return -1;
    } else pcsi_interval(c, bf, in, n, with_rr);
    if (pcsi_launches(c, n) % 2) in = 1 - in;
    in_after[i] = in;
    return with_rr ? 1 : 0;
  }, rr, lerr);
  if (lerr) { c->err = "fused P-CSI: interval launch failed"; return 1; }
  if (conv >= 0) { c->numIterations = (conv + 1) * freq; in = in_after[conv]; }   // later intervals did nothing on the device
  const long long ncell = (long long)c->g.n2 * c->g.nblocks;
  if (in == 1) HIPCHK(c, hipMemcpyAsync(bf.X[0], bf.X[1], sizeof(double) * ncell, hipMemcpyDeviceToDevice, c->stream));
  hipLaunchKernelGGL(k_halo_srcmap, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, c->stream, bf.X[0], c->srcmap, ncell);
  c->rmsResidual = std::sqrt(rr * c->h.residualNorm);
  HIPCHK(c, hipGetLastError());
  if (c->numIterations == cf.max_iterations && c->h.convergenceCriterion != 0.0) { c->err = "POP_SolversPCSI: solver not converged"; return 2; }
  return 0;
}

// fused P-CSI with blocks spread over ranks: one halo exchange (r') and one launch per iteration, a block-sum
// all-reduce only at the convergence checks
int solver_pcsi_fused_dist(pop_ctx *c) {
  const pop_config &cf = c->h.c;
  const dim3 G = grid_2d(c), B(POP_RED_THREADS);
  const int freq = cf.convergence_check_freq, start = pcsi_check_start(c), nbt = c->h.nblocks_tot;
  if (!c->allred || !c->redbuf || c->red_doubles < nbt) { c->err = "distributed P-CSI: no transport / reduce buffer"; return 1; }
  PcsiBufs bf{{c->PS[c->newt], c->Z}, {c->R, c->AZ}, {c->Q, c->S1}};
  SolverScalars init{};
  HIPCHK(c, hipMemcpyAsync(c->sc, &init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  const long long a2 = (long long)c->g.n2 * c->g.nblocks;
  {
    SolveView v = local_view(c);
    hipLaunchKernelGGL(k_pcsi_a0r, dim3((unsigned)((a2 + 255) / 256)), dim3(256), 0, c->stream, (const double *)c->centerWgt, c->S0, a2);
    hipLaunchKernelGGL(k_fresidual<false>, G, B, 0, c->stream, c->g, fused_args(c, v));
    hipLaunchKernelGGL(k_pcsi_scale, dim3((c->g.n2 + 255) / 256, c->g.nblocks), dim3(256), 0, c->stream, c->g, c->R, (const double *)c->S0);
  }
  auto step = [&](int in, int j, bool first, bool rr) -> int {
    if (halo_remote(c, bf.R[in], 1)) return 1;
    PcsiArgs a = pcsi_args(c, bf, in, j);
    a.remote_ghosts = 1;
    if (first) hipLaunchKernelGGL((k_pcsi_step<true, false>), G, B, 0, c->stream, c->g, a);
    else if (c->pcsi_two_cell && rr) hipLaunchKernelGGL((k_pcsi_step2<true>), G, dim3(POP_RED_THREADS / 2), 0, c->stream, c->g, a);
    else if (c->pcsi_two_cell) hipLaunchKernelGGL((k_pcsi_step2<false>), G, dim3(POP_RED_THREADS / 2), 0, c->stream, c->g, a);
    else if (rr) hipLaunchKernelGGL((k_pcsi_step<false, true>), G, B, 0, c->stream, c->g, a);
    else hipLaunchKernelGGL((k_pcsi_step<false, false>), G, B, 0, c->stream, c->g, a);
    return 0;
  };
  if (step(0, 0, true, false)) return 1;
  int in = 1;
  c->numIterations = cf.max_iterations;
  double rr = 0.0;
  if (c->pcsi_two_step_dist && halo_update_many(c, {{c->RHS, 1}})) return 1;   // the pairs form r at the first ring of ghost cells
  // two iterations per launch across ranks (k_pcsi_step_x2): x, dx and r' travel two rings wide once per PAIR instead of r' once per
  // iteration -- half the messages per iteration
  const bool pairs = c->pcsi_two_step_dist;
  const DevGrid gg = c->g;
  for (int m = 1; m <= cf.max_iterations; ++m) {
    bool check = (m % freq == 0) && m >= start;
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, c->stream, c->pcsi_base, m - 1);
    if (pairs && !check && m + 1 <= cf.max_iterations) {
      if (halo_update_many(c, {{bf.X[in], 1}, {bf.Q[in], 1}, {bf.R[in], 1}})) return 1;
      const bool check2 = ((m + 1) % freq == 0) && m + 1 >= start;
      pcsi_launch_pair(c, gg, pcsi_args(c, bf, in, 1), check2);
      ++m; check = check2;
    } else {
      if (pairs && halo_update_many(c, {{bf.X[in], 1}, {bf.Q[in], 1}})) return 1;   // (the pairs do not advance x, dx at the ghosts of other ranks)
      if (step(in, 1, false, check)) return 1;
    }
    in = 1 - in;
    if (check) {
      hipLaunchKernelGGL(k_block_sums_global<1>, dim3(nbt), dim3(POP_RED_THREADS), 0, c->stream, (const double *)c->partial, c->nchunk, c->loc_of_gid, c->redbuf);
      if (c->allred(c->comm_user, 0, nbt)) { c->err = "distributed P-CSI: allreduce failed" + tr_err(c); return 1; }
      hipLaunchKernelGGL(k_finalize<1>, dim3(1), dim3(1), 0, c->stream, c->redbuf, nbt, c->sc, (int)FIN_RR);
      SolverScalars s;
      if (read_scalars(c, &s)) return 1;
      rr = s.rr;
      if (rr < c->h.convergenceCriterion) { c->numIterations = m; break; }
    }
  }
  if (in == 1) HIPCHK(c, hipMemcpyAsync(bf.X[0], bf.X[1], sizeof(double) * a2, hipMemcpyDeviceToDevice, c->stream));
  if (pairs && halo_remote(c, bf.X[0], 1)) return 1;   // (the single steps keep x current at the ghosts of other ranks; the pairs do not)
  hipLaunchKernelGGL(k_halo_srcmap, dim3((unsigned)((a2 + 255) / 256)), dim3(256), 0, c->stream, bf.X[0], c->srcmap, a2);
  c->rmsResidual = std::sqrt(rr * c->h.residualNorm);
  HIPCHK(c, hipGetLastError());
  if (c->numIterations == cf.max_iterations && c->h.convergenceCriterion != 0.0) { c->err = "POP_SolversPCSI: solver not converged"; return 2; }
  return 0;
}

// elapsed time of the last bracketed solve into the totals (waits for its closing event: at the next solve that is long past)
void solve_collect(pop_ctx *c) {
  if (!c->solve_pending) return;
  c->solve_pending = false;
  float ms = 0;
  if (hipEventSynchronize(c->ev_solve[1]) == hipSuccess && hipEventElapsedTime(&ms, c->ev_solve[0], c->ev_solve[1]) == hipSuccess) {
    c->solver_ms_total += ms; c->solver_iters_total += c->solve_iters_pending; c->solver_calls_total += 1;
  }
}
int need_device(pop_ctx *c) {
  if (!c) return 1;
  if (c->host_only) { c->err = "context was created host-only: no GPU path available (there is no CPU fallback)"; return 1; }
  return 0;
}

// resolve a named field: device pointer, element count (local), whether tl / n apply
int resolve(pop_ctx *c, const std::string &name, int tl, int n, double **ptr, long long *count) {
  const long long a2 = (long long)c->g.n2 * c->g.nblocks, a3 = (long long)c->g.n3 * c->g.nblocks;
  const int t = tl == 0 ? c->oldt : tl == 1 ? c->curt : c->newt;
  auto ok = [&](double *p, long long cnt) { *ptr = p; *count = cnt; return p ? 0 : 1; };
  if (name == "TRACER") return (n >= 0 && n < c->h.nt) ? ok(c->TR[n][t], a3) : 1;
  if (name == "UVEL") return ok(c->U[t], a3);
  if (name == "VVEL") return ok(c->V[t], a3);
  if (name == "RHO") return ok(c->RHO[t], a3);
  if (name == "PSURF") return ok(c->PS[t], a2);
  if (name == "GRADPX") return ok(c->GX[t], a2);
  if (name == "GRADPY") return ok(c->GY[t], a2);
  if (name == "UBTROP") return ok(c->UB[t], a2);
  if (name == "VBTROP") return ok(c->VB[t], a2);
  if (name == "PGUESS") return ok(c->PGUESS, a2);
  if (name == "FW") return ok(c->FW, a2);
  if (name == "FW_OLD") return ok(c->FW_OLD, a2);
  if (name == "SHF_QSW") return ok(c->SHF_QSW, a2);
  if (name == "CHL") return ok(c->CHL, a2);                   // chlorophyll, mg/m^3 (sw_absorption_type 'chlorophyll')
  if (name == "STF") return (n >= 0 && n < c->h.nt) ? ok(c->STF[n], a2) : 1;
  if (name == "TFW") return (n >= 0 && n < c->h.nt) ? ok(c->TFW[n], a2) : 1;
  if (name == "KPP_SRC") return (n >= 0 && n < c->h.nt) ? ok(c->KPP_SRC[n], a3) : 1;
  if (name == "VDC") return (n == 0 || n == 1) ? ok(c->VDC[n], (long long)c->g.n2 * (c->g.km + 2) * c->g.nblocks) : 1;
  if (name == "VVC") return ok(c->VVC, a3);
  if (name == "DH") return ok(c->DH, a2);
  if (name == "DHU") return ok(c->DHU, a2);
  if (name == "ZX") return ok(c->ZX, a2);
  if (name == "ZY") return ok(c->ZY, a2);
  if (name == "UH") return ok(c->UH, a2);
  if (name == "VH") return ok(c->VH, a2);
  if (name == "RHS") return ok(c->RHS, a2);
  if (name == "centerWgt") return ok(c->centerWgt, a2);
  if (name == "HBLT") return ok(c->HBLT, a2);
  if (name == "HMXL") return ok(c->HMXL, a2);
  if (name == "HMXL_DR") return ok(c->HMXL_DR, a2);
  if (name == "UISOP") return ok(c->gm.UISOP, a3);           // diag_gm_bolus (hmix_tracer = 3): eddy-induced velocity, east / north face and top of the T cell
  if (name == "VISOP") return ok(c->gm.VISOP, a3);
  if (name == "WISOP") return ok(c->gm.WISOP, a3);
  // work fields of Gent-McWilliams mixing, for the pins of tests/pins.py: the merged stream function of the half cells (stored in the branch
  // without cancellation: n = 2 face + half, face 0 east / north, half 0 top) and the transition layer's depths
  if (name == "GM_SF_SLX") return (n >= 0 && n < 4) ? ok(c->gm.SF[n], a3) : 1;
  if (name == "GM_SF_SLY") return (n >= 0 && n < 4) ? ok(c->gm.SF[4 + n], a3) : 1;
  if (name == "TLT_DIABATIC_DEPTH") return ok(c->gm.DD, a2);
  if (name == "TLT_THICKNESS") return ok(c->gm.TH, a2);
  if (name == "TLT_INTERIOR_DEPTH") return ok(c->gm.ID, a2);
  if (name == "SMF") return ok(c->d2[n == 0 ? "SMF1" : "SMF2"], a2);
  if (name == "SMFT") return ok(c->d2[n == 0 ? "SMFT1" : "SMFT2"], a2);
  auto it = c->d2.find(name);
  if (it != c->d2.end()) return ok(it->second, a2);
  return 1;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int pop_create(const pop_config *cfg, int rank, int nranks, int flags, pop_ctx **out) { return pop_create_tuned(cfg, nullptr, nullptr, rank, nranks, flags, out); }
int pop_create_with_grid(const pop_config *cfg, const pop_grid_input *grid, int rank, int nranks, int flags, pop_ctx **out) {
  return pop_create_tuned(cfg, grid, nullptr, rank, nranks, flags, out);
}

// ---- tuning switches (include/pop_amd.h pop_tuning): one table of (field, environment variable)
#define POP_TUNING_FIELDS(X)                                                                                                            \
  X(land_skip, "POP_LAND_SKIP") X(land_full_steps, "POP_LAND_FULL_STEPS") X(xcd_remap, "POP_XCD_REMAP") X(red_tiles, "POP_RED_TILES")   \
  X(red_band, "POP_RED_BAND") X(lds_order, "POP_LDS_ORDER") X(momentum_lds, "POP_MOMENTUM_LDS") X(tracer_lds, "POP_TRACER_LDS")         \
  X(generic_thomas, "POP_GENERIC_THOMAS") X(reg_thomas_t, "POP_REG_THOMAS_T") X(thomas_pair, "POP_THOMAS_PAIR")                         \
  X(tracer_fwd, "POP_TRACER_FWD") X(vdc_shared, "POP_VDC_SHARED") X(side_stream, "POP_SIDE_STREAM") X(del4_side, "POP_DEL4_SIDE")       \
  X(del4_tile, "POP_DEL4_TILE") X(d2t_fuse, "POP_D2T_FUSE") X(d2u_fuse, "POP_D2U_FUSE") X(vmixu_defer, "POP_VMIXU_DEFER")               \
  X(vmixu_inline, "POP_VMIXU_INLINE") X(btrop_inline, "POP_BTROP_INLINE") X(kpp_ahead, "POP_KPP_AHEAD") X(kpp_col, "POP_KPP_COL")       \
  X(kpp_lazy, "POP_KPP_LAZY") X(kpp_ushear_hint, "POP_KPP_USHEAR_HINT") X(kpp_ushear_margin, "POP_KPP_USHEAR_MARGIN")                   \
  X(kpp_side_stream, "POP_KPP_SIDE_STREAM") X(kpp_buoy_waves, "POP_KPP_BUOY_WAVES") X(kpp_interior_generic, "POP_KPP_INTERIOR_GENERIC") \
  X(kpp_src_full, "POP_KPP_SRC_FULL") X(solver_unfused, "POP_SOLVER_UNFUSED") X(solver_nograph, "POP_SOLVER_NOGRAPH")                   \
  X(solver_presum, "POP_SOLVER_PRESUM") X(solver_distributed, "POP_SOLVER_DISTRIBUTED") X(solver_overlap_off, "POP_SOLVER_OVERLAP_OFF") \
  X(fpcg_b2, "POP_FPCG_B2") X(pcsi_step2, "POP_PCSI_STEP2") X(halo_separate, "POP_HALO_SEPARATE")                                       \
  X(halo_overlap_off, "POP_HALO_OVERLAP_OFF") X(rccl_overlap, "POP_RCCL_OVERLAP") X(evp_wave, "POP_EVP_WAVE") X(fpcg_a_pair, "POP_FPCG_A_PAIR") X(stream_priority, "POP_STREAM_PRIORITY") X(kpp_sparse, "POP_KPP_SPARSE") X(pbc_generic_thomas, "POP_PBC_GENERIC_THOMAS") X(pbc_generic_kpp, "POP_PBC_GENERIC_KPP") X(state3d_levels, "POP_STATE3D_LEVELS") X(gm_sf_stored, "POP_GM_SF_STORED") X(pcg_persist, "POP_PCG_PERSIST") X(gm_flux_tile, "POP_GM_FLUX_TILE") X(pcsi_two_step, "POP_PCSI_TWO_STEP") X(block_sums_relay, "POP_BLOCK_SUMS_RELAY") X(pcsi_evp_fused, "POP_PCSI_EVP_FUSED")
void pop_tuning_init(pop_tuning *t) {
  if (!t) return;
  t->struct_bytes = (int)sizeof(pop_tuning);
#define X(f, e) t->f = POP_TUNING_UNSET;
  POP_TUNING_FIELDS(X)
#undef X
}
// caller's struct, then the environment (a variable that is set but not a number counts as 1, as the older on/off switches did)
static void tuning_resolve(pop_tuning &t, const pop_tuning *user) {
  pop_tuning_init(&t);
  if (user && user->struct_bytes == (int)sizeof(pop_tuning)) t = *user;
  auto env = [](const char *name, int &v) {
    const char *e = getenv(name);
    if (!e) return;
    char *end = nullptr;
    const long x = strtol(e, &end, 10);
    v = (end == e) ? 1 : (int)x;
  };
#define X(f, e) env(e, t.f);
  POP_TUNING_FIELDS(X)
#undef X
}
int pop_get_tuning(const pop_ctx *c, pop_tuning *resolved) {
  if (!c || !resolved) return 1;
  *resolved = c->h.tun;
  return 0;
}

// the reference's direct-access binary grid files (grid.F90:1362-1380, :2066-2085): whole records, native byte order
int pop_read_grid_files(const char *horiz_grid_file, const char *topography_file, int nx_global, int ny_global, double *seven_records, int *kmt) {
  const size_t n = (size_t)nx_global * ny_global;
  auto slurp = [&](const char *path, void *dst, size_t bytes) {
    FILE *f = fopen(path, "rb");
    if (!f) return 1;
    const size_t got = fread(dst, 1, bytes, f);
    fclose(f);
    return got == bytes ? 0 : 1;
  };
  if (horiz_grid_file && seven_records && slurp(horiz_grid_file, seven_records, 7 * n * sizeof(double))) return 1;
  if (topography_file && kmt && slurp(topography_file, kmt, n * sizeof(int))) return 2;
  return 0;
}

int pop_create_tuned(const pop_config *cfg, const pop_grid_input *grid, const pop_tuning *tuning, int rank, int nranks, int flags, pop_ctx **out) {
  if (!cfg || !out || nranks < 1 || rank < 0 || rank >= nranks) return 1;
  pop_ctx *c = new pop_ctx();
  *out = c;
  c->h.c = *cfg; c->h.rank = rank; c->h.nranks = nranks;
  if (cfg->hmix_tracer == 3 && cfg->gm_transition_layer == 1 && cfg->vmix_choice == 3) c->h.c.kpp_ml_diagnostics = 1;   // the diabatic depth of the transition layer is the smoothed HMXL (hmix_gm.F90:1226-1228)
  if (tuning && tuning->struct_bytes != (int)sizeof(pop_tuning)) { c->err = "pop_create_tuned: pop_tuning.struct_bytes does not match this library (use pop_tuning_init)"; return 1; }
  tuning_resolve(c->h.tun, tuning);
  c->grid_from_input = grid != nullptr;
  if (grid && !(grid->ULAT && grid->ULON && grid->HTN && grid->HTE && grid->HUS && grid->HUW)) { c->err = "pop_create_with_grid: ULAT, ULON, HTN, HTE, HUS, HUW are required"; return 1; }
  c->h.gin = grid;
  {   // every option this library does not implement is refused here, before anything is built (the reference aborts
      // in the init routine of the option's own module, e.g. vertical_mix.F90:280-296, POP_SolversMod.F90:442-472)
    auto bad = [&](const std::string &m) { c->err = "pop_create: " + m; return 1; };
    if (cfg->struct_version != POP_CONFIG_VERSION) return bad("pop_config.struct_version is " + std::to_string(cfg->struct_version) + ", this library was built for " + std::to_string(POP_CONFIG_VERSION) + " (include/pop_amd.h)");
    if (cfg->reserved_i[0] != 0) return bad("pop_config.reserved_i must be 0");
    if (cfg->gm_kappa_bkg_srfbl != 0 && cfg->gm_kappa_bkg_srfbl != 1) return bad("gm_kappa_bkg_srfbl: 0 or 1");
    if (cfg->ah_bkg_bottom < 0.0) return bad("ah_bkg_bottom: >= 0");
    if (cfg->gm_diag_bolus != 0 && cfg->gm_diag_bolus != 1) return bad("gm_diag_bolus: 0 or 1");
    if (cfg->gm_diag_bolus && cfg->hmix_tracer != 3) return bad("gm_diag_bolus needs hmix_tracer = 3");
    if (cfg->gm_transition_layer != 0 && cfg->gm_transition_layer != 1) return bad("gm_transition_layer: 0 or 1");
    if (cfg->gm_transition_layer && cfg->hmix_tracer != 3) return bad("gm_transition_layer needs hmix_tracer = 3");
    // the reference aborts in init_gm: 'hmix_gm currently incompatible with partial bottom cells' (hmix_gm.F90:782-785) -- its slopes, stream-function
    // terms and flux divergence use dz(k), the tracer budget DZT (accepted until round 3 with dz(k) throughout, tracer content not conserved)
    if (cfg->hmix_tracer == 3 && cfg->partial_bottom_cells) return bad("hmix_tracer = 3 (Gent-McWilliams) with partial_bottom_cells: the reference refuses the combination (hmix_gm.F90:782-785), so does this library");
    // hmix_gm.F90:724-730: kappa_depth_2 = 0 with the 'depth' profile aborts in init_gm
    if (cfg->hmix_tracer == 3 && cfg->gm_kappa_type == 2 && cfg->kappa_depth_2 == 0.0) return bad("gm_kappa_type = 2 ('depth') needs kappa_depth_2 /= 0 (hmix_gm.F90:724-730)");
    if (cfg->hmix_tracer == 3 && cfg->gm_kappa_type == 1 && cfg->gm_kappa_freq == 0) return bad("gm_kappa_type = 1 ('bfre') needs gm_kappa_freq = 1 | 2: 'kappa_freq should not be set to never when model fields dependent kappa types are chosen' (hmix_gm.F90:756-780; no N^2 file here)");
    if (cfg->gm_kappa_type < 0 || cfg->gm_kappa_type > 2) return bad("gm_kappa_type: 0 constant, 1 bfre, 2 depth (the other kappa choices of hmix_gm_nml are not built)");
    if (cfg->gm_kappa_freq < 0 || cfg->gm_kappa_freq > 2) return bad("gm_kappa_freq: 0 never, 1 every_time_step, 2 once_a_day");
    if (cfg->gm_kappa_freq == 2 && cfg->tmix_opt == 1) return bad("gm_kappa_freq = once_a_day with time_mix_opt 'avg' (half steps that do not fit the day: the end-of-day test of time_management.F90:3586-3592 on the calendar) is not built: avgfit, robert or none");
    if (cfg->gm_slope_control < 0 || cfg->gm_slope_control > 3) return bad("gm_slope_control: 0 notanh, 1 tanh, 2 clip, 3 Gerd");
    if (cfg->gm_slope_control == 2 && cfg->gm_transition_layer) return bad("gm_slope_control = clip with the transition layer (SLA_SAVE is formed before the slopes are clipped, hmix_gm.F90:1234-1245) is not built");
    if (cfg->ah_bolus < 0.0 || cfg->ah_bkg_srfbl < 0.0 || cfg->slm_r < 0.0 || cfg->slm_b < 0.0) return bad("ah_bolus, ah_bkg_srfbl, slm_r, slm_b: >= 0 (0 = ah, ah, 0.3, 0.3)");
    if (cfg->partial_bottom_cells != 0 && cfg->partial_bottom_cells != 1) return bad("partial_bottom_cells: 0 or 1");
    if (cfg->partial_bottom_cells && grid && grid->DZBC == nullptr && grid->KMT != nullptr) return bad("partial_bottom_cells with a topography record needs pop_grid_input.DZBC (the record of bottom_cell_file)");
    if (cfg->lsw_absorb != 0 && cfg->lsw_absorb != 1) return bad("lsw_absorb: 0 or 1");
    if (cfg->nx_global < 1 || cfg->ny_global < 1 || cfg->km < 2 || cfg->block_size_x < 1 || cfg->block_size_y < 1) return bad("domain / block sizes must be positive (km >= 2)");
    if (cfg->ew_boundary != 0 && cfg->ew_boundary != 1) return bad("ew_boundary: 0 closed, 1 cyclic");
    if (cfg->ns_boundary < 0 || cfg->ns_boundary > 2) return bad("ns_boundary: 0 closed, 1 cyclic, 2 tripole");
    if (cfg->hmix_momentum != 2 && cfg->hmix_momentum != 4) return bad("hmix_momentum: 2 (del2) or 4 (del4); the anisotropic viscosity is not built");
    if (cfg->hmix_tracer != 2 && cfg->hmix_tracer != 4 && cfg->hmix_tracer != 3) return bad("hmix_tracer: 2 (del2), 4 (del4) or 3 (gm)");
    if (cfg->vmix_choice < 1 || cfg->vmix_choice > 3) return bad("vmix_choice: 1 const, 2 rich, 3 kpp");
    if (cfg->tadvect < 1 || cfg->tadvect > 3) return bad("tadvect: 1 centered, 2 upwind3, 3 lw_lim");
    if (cfg->solver_choice < 1 || cfg->solver_choice > 3) return bad("solver_choice: 1 pcg, 2 ChronGear, 3 PCSI");
    if (cfg->preconditioner_choice != 0 && cfg->preconditioner_choice != 1) return bad("preconditioner_choice: 0 diagonal, 1 evp");
    if (cfg->stepped_bathymetry != 0 && cfg->stepped_bathymetry != 1) return bad("stepped_bathymetry: 0 flat, 1 stepped");
    if (cfg->distribution_type != 0 && cfg->distribution_type != 1) return bad("distribution_type: 0 equal block counts, 1 balanced by ocean columns");
    if (cfg->kpp_ml_diagnostics != 0 && cfg->kpp_ml_diagnostics != 1) return bad("kpp_ml_diagnostics: 0 off, 1 HMXL and HMXL_DR every step");
    if (cfg->max_iterations < 1 || cfg->convergence_check_freq < 1) return bad("max_iterations and convergence_check_freq must be >= 1");
    if (cfg->tmix_opt < 0 || cfg->tmix_opt > 3) return bad("tmix_opt: 0 none, 1 avg, 2 avgfit, 3 robert");
    if ((cfg->tmix_opt == 1 || cfg->tmix_opt == 2) && cfg->time_mix_freq < 1) return bad("time_mix_freq must be >= 1");
    if (cfg->steps_per_day < 1) return bad("steps_per_day must be >= 1");
    if (cfg->aidif != 1.0) return bad("aidif: only the fully implicit vertical mixing (aidif = 1) is built");
    if (cfg->vmix_choice == 3 && cfg->lshort_wave && (cfg->sw_absorption_type < 0 || cfg->sw_absorption_type > 2)) return bad("KPP lshort_wave: sw_absorption_type 0 top-layer, 1 jerlov, 2 chlorophyll");
    if (cfg->jerlov_water_type < 0 || cfg->jerlov_water_type > 5) return bad("jerlov_water_type: 1..5 (0 = 3)");
    if (cfg->lsw_absorb != 0 && (cfg->sw_absorption_type < 0 || cfg->sw_absorption_type > 2)) return bad("lsw_absorb: sw_absorption_type 0 top-layer, 1 jerlov, 2 chlorophyll");
    if (cfg->vmix_choice == 3 && cfg->num_v_smooth_Ri < 1) return bad("KPP: num_v_smooth_Ri must be >= 1 (the reference leaves FRI unset otherwise)");
    if (!(cfg->convergence_criterion >= 0.0)) return bad("convergence_criterion must be >= 0");
  }
  c->h.plan_only = (flags & POP_CREATE_PLAN_ONLY) != 0;
  const int hb = host_build(c->h);
  c->h.gin = nullptr;
  if (hb) { c->err = c->h.err; return 1; }
  c->host_only = (flags & (POP_CREATE_HOST_ONLY | POP_CREATE_PLAN_ONLY)) != 0;
  if (c->host_only) return 0;
  if (cfg->nt != 2) { c->err = "device kernels are built for nt = 2 (T,S) in this round"; return 1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { c->err = "no HIP device available; libpop_amd has no CPU fallback"; return 1; }
  // stream priorities (r3, measured and left OFF: 84.7 ms against 83.2 without, profiles/r03_ab_stream_priority.txt): the launch stream above
  // the streams that only fill its gaps (KPP of the next step), so that the
  // one-workgroup block sums between the solver's kernels are dispatched ahead of KPP workgroups waiting in the other queues
  int prio_least = 0, prio_greatest = 0;
  const bool prio = tun_on(c->h.tun.stream_priority) && hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) == hipSuccess && prio_least != prio_greatest;
  if (prio) HIPCHK(c, hipStreamCreateWithPriority(&c->stream, hipStreamDefault, prio_greatest));
  else HIPCHK(c, hipStreamCreate(&c->stream));
  c->prio_least = prio ? prio_least : 0; c->prio_on = prio;
  c->own_stream = true;
  HostModel &h = c->h;
  DevGrid &g = c->g;
  g.nxb = h.nxb; g.nyb = h.nyb; g.km = h.km; g.nt = h.nt; g.nblocks = h.nblocks;
  g.n2 = (int)h.n2; g.n3 = (long long)h.n3;
  // workgroup order (kernels_common.hpp).  Measured on MI355X: the XCD-banded order cuts fabric re-fetch of
  // the stencil kernels from 3.2x to 1.3x of the algorithmic reads at gx1v7 (-6% time) but costs +20% at
  // tx0.1v3; row-aligned tile columns (mode 2 / POP_RED_TILES) lose 128-B alignment of the rows
  // (nx_block*8 B is not a multiple of 128) and cost +10% (columns) / +2% (solver) at tx0.1v3, where
  // the 256 MB infinity cache already absorbs the re-fetch.  Large grids therefore keep the natural
  // linear order; the other orders stay selectable (POP_XCD_REMAP=0|1|2, POP_RED_TILES=0|1) and tested.
  g.xcd_remap = (h.n2 * h.nblocks <= (1u << 19)) ? 1 : 0;
  g.xcd_remap = tun_or(h.tun.xcd_remap, g.xcd_remap);
  g.red_tiles = tun_or(h.tun.red_tiles, 0);
  // 2-D solver / reduction kernels: XCD-banded chunk order (same chunks, same partial order, only the workgroup ->
  // chunk assignment changes, so results are bitwise unchanged): gx1v7 4.31 -> 4.01 ms per step (the 9-point
  // matvec finds its j+-1 rows in the XCD's own L2), tx0.1v3 177.1 -> 175.2.  POP_RED_BAND=0 restores the natural order.
  g.lds_order = tun_or(h.tun.lds_order, 1);
  g.red_band = tun_or(h.tun.red_band, 1);
  g.ib = NGHOST + 1; g.ie = h.nxb - NGHOST; g.jb = NGHOST + 1; g.je = h.nyb - NGHOST;
  {   // padded blocks (blocks.F90:174-265): per-block last physical column / row, only when some local block ends early
    std::vector<int> ieb(h.nblocks), jeb(h.nblocks);
    bool short_block = false;
    for (int lb = 0; lb < h.nblocks; ++lb) {
      const BlockInfo &B = h.all_blocks[h.local_ids[lb] - 1];
      ieb[lb] = B.ie; jeb[lb] = B.je;
      if (B.ie != g.ie || B.je != g.je) short_block = true;
    }
    g.ieb = nullptr; g.jeb = nullptr;
    if (short_block) {
      int *d0, *d1;
      if (dev_upload(c, &d0, ieb.data(), ieb.size()) || dev_upload(c, &d1, jeb.data(), jeb.size())) return 1;
      g.ieb = d0; g.jeb = d1;
    }
  }
  // vertical arrays
  {
    struct { CArr *dst; std::vector<double> *src; } V[] = {
      {&g.dz, &h.dz}, {&g.dzw, &h.dzw}, {&g.zt, &h.zt}, {&g.zw, &h.zw}, {&g.c2dz, &h.c2dz}, {&g.dzr, &h.dzr}, {&g.dz2r, &h.dz2r},
      {&g.dzwr, &h.dzwr}, {&g.pressz, &h.pressz}, {&g.bouss, &h.bouss}, {&g.afac_t, &h.afac_t}, {&g.afac_u, &h.afac_u}};
    for (auto &v : V) { double *p; if (dev_upload(c, &p, v.src->data(), v.src->size())) return 1; *v.dst = p; }
    // per-level MWJF coefficients (k_state3d_lv): formed on the device
    if (g.km + 1 <= 1024) {
      double *p = nullptr;
      std::vector<double> z((size_t)6 * (g.km + 2), 0.0);
      if (dev_upload(c, &p, z.data(), z.size())) return 1;
      hipLaunchKernelGGL(k_eos_level_table, dim3(1), dim3(g.km + 1), 0, 0, g, p);
      HIPCHK(c, hipDeviceSynchronize());
      g.eosP = p;
    }
    g.state_lv = tun_or(h.tun.state3d_levels, 4);
  }
  // 2-D fields: upload the local blocks of every host field
  for (auto &kv : h.f2) {
    std::vector<double> all;
    const std::vector<double> *src = &kv.second;
    if (cfg->ns_boundary == 2 && kv.first == "centerWgtIndep") {
      // Beyond a tripole fold the reference forms z = r / centerWgt in the ghost rows with the ghost cells' OWN centerWgt, whose four
      // terms were added in the mirrored order: equal to the owner's value up to rounding only.  The fused solver kernels (and
      // the exchange of z between ranks) give a ghost cell its owner's z instead; so that every solver path of this library
      // agrees bit for bit, the device copy of the state-independent part of centerWgt takes the owners' values in those rows
      // (the area and the mask in the other part are mirrored copies already).  The host copy keeps the reference's values.
      all = kv.second;
      host_halo_r8_loc(h, all.data(), 1, 0.0, 0, 0);
      src = &all;
    }
    const int wgt = kv.first == "btropWgtNE" ? 1 : kv.first == "btropWgtEast" ? 2 : kv.first == "btropWgtNorth" ? 3 : 0;
    if (wgt) {
      // The reference forms the off-centre weights from i = 2, j = 2 of the block array on (POP_SolversMod.F90:795-815); the
      // first row and column stay 0 and nothing reads them.  The two-iterations-per-launch P-CSI across ranks forms the operator at the
      // first ring of ghost cells, whose west / south weights live in that row and column: the device copy holds them, made of the
      // same two U-point terms in the same order.  The host copy keeps the reference's values.
      all = kv.second;
      const std::vector<double> &XW = h.f2.at("btropXW"), &YW = h.f2.at("btropYW");
      const int nxb = h.nxb, nyb = h.nyb;
      const size_t n2 = (size_t)nxb * nyb;
      for (size_t b = 0; b < all.size() / n2; ++b)
        for (int j = 0; j < nyb; ++j) for (int i = 0; i < nxb; ++i) {
          if (i > 0 && j > 0) continue;
          const size_t q = b * n2 + (size_t)j * nxb + i;
          if (wgt == 1) all[q] = XW[q] + YW[q];
          else if (wgt == 2 && j > 0) all[q] = XW[q] + XW[q - nxb] - YW[q] - YW[q - nxb];
          else if (wgt == 3 && i > 0) all[q] = YW[q] + YW[q - 1] - XW[q] - XW[q - 1];
        }
      src = &all;
    }
    auto loc = local_part(h, *src); double *p; if (dev_upload(c, &p, loc.data(), loc.size())) return 1; c->d2[kv.first] = p;
  }
  for (auto &kv : h.i2) { auto loc = local_part(h, kv.second); int *p; if (dev_upload(c, &p, loc.data(), loc.size())) return 1; c->di2[kv.first] = p; }
#define G2(f) g.f = c->d2[#f]
  G2(DXU); G2(DYU); G2(DXUR); G2(DYUR); G2(UAREA_R); G2(TAREA_R); G2(TAREA); G2(FCOR); G2(FCORT); G2(HU); G2(HUR);
  G2(AU0); G2(AUN); G2(AUE); G2(AUNE); G2(RCALCT); G2(DTN); G2(DTS); G2(DTE); G2(DTW);
  G2(DUC); G2(DUN); G2(DUS); G2(DUE); G2(DUW); G2(DMC); G2(DMN); G2(DMS); G2(DME); G2(DMW); G2(DUM); G2(KXU); G2(KYU);
  {   // byte copy of the solver mask (RCALCT: exactly 0 or 1, POP_SolversMod.F90:886)
    const std::vector<double> mm = local_part(h, h.f2["mMask"]);
    std::vector<unsigned char> m8(mm.size());
    for (size_t p = 0; p < mm.size(); ++p) {
      if (mm[p] != 0.0 && mm[p] != 1.0) { c->err = "solver mask is not 0/1"; return 1; }
      m8[p] = mm[p] != 0.0;
    }
    unsigned char *d8;
    if (dev_upload(c, &d8, m8.data(), m8.size())) return 1;
    g.mMask8 = d8;
  }
  G2(mMask); G2(CHECKER); G2(CONSTNT); G2(SMF1); G2(SMF2); G2(SMFT1); G2(SMFT2);
  g.pbc = cfg->partial_bottom_cells ? 1 : 0;
  if (g.pbc) { G2(DZBC); G2(DZUB); }
#undef G2
  if (dev_alloc(c, &g.dump, 8192)) return 1;
  { double *z; if (dev_alloc(c, &z, 64)) return 1; g.zero = z; }
  {   // land elimination (DevGrid::opre): prefix count of the cells that have an ocean T cell within two cells in either
      // direction.  The margin is what makes the skipped values independent of the state: every field the full kernels write
      // on a cell further than one stencil from any ocean cell (tgrid_to_ugrid averages, gradients at land U points, ...) is a
      // constant.  Ghost cells count with the KMT their halo update gave them.
    const std::vector<int> kmt = local_part(h, h.i2["KMT"]);
    std::vector<int> pre((size_t)(h.n2 + 1) * h.nblocks);
    long long tiles = 0, land = 0;
    for (int b = 0; b < h.nblocks; ++b) {
      const int *K = kmt.data() + (size_t)b * h.n2;
      int *P = pre.data() + (size_t)b * (h.n2 + 1);
      const BlockInfo &B = h.all_blocks[h.local_ids[b] - 1];
      P[0] = 0;
      for (int j = 0; j < h.nyb; ++j)
        for (int i = 0; i < h.nxb; ++i) {
          // ghost cells beyond a closed boundary are rewritten with the fill value by every halo update and recomputed by the
          // whole-block kernels of averaging steps: their values do depend on the step, so they count as cells to compute
          int near = (B.i_glob[i] == 0 || B.j_glob[j] == 0) ? 1 : 0;
          for (int dj = -2; dj <= 2 && !near; ++dj)
            for (int di = -2; di <= 2; ++di) {
              const int ii = i + di, jj = j + dj;
              if (ii >= 0 && ii < h.nxb && jj >= 0 && jj < h.nyb && K[(size_t)jj * h.nxb + ii] > 0) { near = 1; break; }
            }
          P[(size_t)j * h.nxb + i + 1] = P[(size_t)j * h.nxb + i] + near;
        }
      for (int j = 0; j < h.nyb; ++j)
        for (int i0 = 0; i0 < h.nxb; i0 += 64) {
          const int i1 = std::min(i0 + 64, h.nxb);
          ++tiles;
          if (P[(size_t)j * h.nxb + i1] == P[(size_t)j * h.nxb + i0]) ++land;
        }
    }
    int *d;
    if (dev_upload(c, &d, pre.data(), pre.size())) return 1;
    g.opre = d;
    g.skip = 0;
    c->opre_host = pre;
    for (int R : {4, 8}) {   // DevGrid::lds_act4 / lds_act8
      const int ntx = (h.nxb - 2 * NGHOST + POP_COL_THREADS - 1) / POP_COL_THREADS, nty = (h.nyb - 2 * NGHOST + R - 1) / R;
      if (ntx > 0xffff || nty > 0x7fff) continue;
      // tiles in the numbering the launch order is built on (patch by patch for lds_order 1, row-major otherwise)
      std::vector<std::vector<int>> act(h.nblocks);
      size_t longest = 0;
      for (int b = 0; b < h.nblocks; ++b) {
        const int *P = pre.data() + (size_t)b * (h.n2 + 1);
        for (int gi = 0; gi < ntx * nty; ++gi) {
          int ti, tj;
          if (g.lds_order) { if (!lds_tile_from_gi(gi, ntx, nty, ti, tj)) continue; }
          else { ti = gi % ntx; tj = gi / ntx; }
          const int i0 = NGHOST + ti * POP_COL_THREADS, i1 = std::min(i0 + POP_COL_THREADS, h.nxb);
          int n = 0;
          for (int j = NGHOST + tj * R; j < std::min(NGHOST + tj * R + R, h.nyb); ++j) n += P[(size_t)j * h.nxb + i1] - P[(size_t)j * h.nxb + i0];
          if (n) act[b].push_back(tj << 16 | ti);
        }
        longest = std::max(longest, act[b].size());
      }
      if (longest == 0 || longest == (size_t)ntx * nty) continue;
      // lds_order 1: the a-th tile WITH OCEAN goes where the a-th tile of the full numbering would: runs of 32 such tiles -- a
      // compact patch -- to one XCD, so the halo rows neighbouring tiles share are still fetched into one L2 (a plain packed list
      // would deal consecutive tiles to eight different XCDs: measured 1.34 x the algorithmic traffic for the momentum kernel
      // against 1.15 x)
      if (g.lds_order) longest = 256 * ((longest + 255) / 256);
      std::vector<int> list((size_t)longest * h.nblocks, -1);
      for (int b = 0; b < h.nblocks; ++b)
        for (size_t a = 0; a < act[b].size(); ++a) list[(size_t)b * longest + (g.lds_order ? (size_t)lds_slot_of((int)a) : a)] = act[b][a];
      int *dl;
      if (dev_upload(c, &dl, list.data(), list.size())) return 1;
      if (R == 4) { g.lds_act4 = dl; g.lds_n4 = (int)longest; } else { g.lds_act8 = dl; g.lds_n8 = (int)longest; }
    }
    c->land_fraction = tiles ? (double)land / (double)tiles : 0.0;
    c->land_skip = !tun_off(h.tun.land_skip);
    c->land_full_steps = tun_or(h.tun.land_full_steps, 4);
    c->full_left = c->land_full_steps;
  }
  g.WNE = c->d2["btropWgtNE"]; g.WEa = c->d2["btropWgtEast"]; g.WNo = c->d2["btropWgtNorth"]; g.WC0 = c->d2["centerWgtIndep"];
  g.XW = c->d2["btropXW"]; g.YW = c->d2["btropYW"];
  if (cfg->hmix_tracer == 4) { g.DTN = c->d2["d4DTN"]; g.DTS = c->d2["d4DTS"]; g.DTE = c->d2["d4DTE"]; g.DTW = c->d2["d4DTW"]; }
  if (cfg->hmix_momentum == 4) {
    g.DUC = c->d2["d4DUC"]; g.DUN = c->d2["d4DUN"]; g.DUS = c->d2["d4DUS"]; g.DUE = c->d2["d4DUE"]; g.DUW = c->d2["d4DUW"];
    g.DMC = c->d2["d4DMC"]; g.DMN = c->d2["d4DMN"]; g.DMS = c->d2["d4DMS"]; g.DME = c->d2["d4DME"]; g.DMW = c->d2["d4DMW"]; g.DUM = c->d2["d4DUM"];
  }
  if (cfg->hmix_tracer == 4 || cfg->hmix_momentum == 4) { c->mix.D4AMF = c->d2["D4AMF"]; c->mix.D4AHF = c->d2["D4AHF"]; }
  if (cfg->tadvect == 2) {
    const char *nx[6] = {"TALFXP", "TBETXP", "TGAMXP", "TALFXM", "TBETXM", "TDELXM"};
    const char *ny[6] = {"TALFYP", "TBETYP", "TGAMYP", "TALFYM", "TBETYM", "TDELYM"};
    for (int t = 0; t < 6; ++t) {
      c->upw3.cx[t] = c->d2[nx[t]]; c->upw3.cy[t] = c->d2[ny[t]];
      double *p; if (dev_upload(c, &p, h.upw_z[t].data(), h.upw_z[t].size())) return 1;
      c->upw3.cz[t] = p;
    }
  }
  if (cfg->tadvect == 3) {   // lw_lim: three flux-velocity fields, three work fields per tracer
    const size_t a3l = h.n3 * h.nblocks;
    if (dev_alloc(c, &c->lw.UTE, a3l) || dev_alloc(c, &c->lw.VTN, a3l) || dev_alloc(c, &c->lw.WTKB, a3l)) return 1;
    for (int n = 0; n < 2; ++n) if (dev_alloc(c, &c->lw.XOUT[n], a3l) || dev_alloc(c, &c->lw.XSTAR[n], a3l) || dev_alloc(c, &c->lw.XSTAR2[n], a3l)) return 1;
    c->lw.HTE = c->d2["HTE"]; c->lw.HTN = c->d2["HTN"]; c->lw.DXT = c->d2["DXT"]; c->lw.DYT = c->d2["DYT"];
    if (!c->lw.HTE || !c->lw.HTN || !c->lw.DXT || !c->lw.DYT) { c->err = "lw_lim: grid fields HTE / HTN / DXT / DYT missing"; return 1; }
  }
  if (cfg->hmix_tracer == 3) {   // Gent-McWilliams: 14 coefficient fields + the tendency of each tracer
    const size_t a3g = h.n3 * h.nblocks;
    GmDev &G = c->gm;
    for (int t = 0; t < 4; ++t) if (dev_alloc(c, &G.SLX[t], a3g) || dev_alloc(c, &G.SLY[t], a3g)) return 1;
    for (int t = 0; t < 2; ++t) if (dev_alloc(c, &G.KI[t], a3g) || dev_alloc(c, &G.KT[t], a3g) || dev_alloc(c, &G.HD[t], a3g) || dev_alloc(c, &G.GTK[t], a3g)) return 1;
    if (cfg->gm_kappa_type == 1) {   // KAPPA_VERTICAL: module state, 1 until the profile is computed (hmix_gm.F90:860)
      if (dev_alloc(c, &G.KV, a3g, false)) return 1;
      std::vector<double> ones(a3g, 1.0);
      HIPCHK(c, hipMemcpy(G.KV, ones.data(), a3g * sizeof(double), hipMemcpyHostToDevice));
    }
    G.tlt = cfg->gm_transition_layer == 1;
    if (G.tlt) {   // transition layer: SLA_SAVE of both halves, the layer's 2-D fields, the column terms of the merged stream function
      const size_t a2g = h.n2 * h.nblocks;
      for (int t = 0; t < 2; ++t) if (dev_alloc(c, &G.SLA[t], a3g)) return 1;
      if (dev_alloc(c, &G.DD, a2g) || dev_alloc(c, &G.TH, a2g) || dev_alloc(c, &G.ID, a2g) || dev_alloc(c, &G.KL, a2g) || dev_alloc(c, &G.ZTW, a2g)) return 1;
      for (int t = 0; t < 8; ++t) if (dev_alloc(c, &G.MW[t], a2g)) return 1;
    }
    if (cfg->gm_diag_bolus) {
      if (dev_alloc(c, &G.UISOP, a3g) || dev_alloc(c, &G.VISOP, a3g) || dev_alloc(c, &G.WISOP, a3g)) return 1;
      G.HTE = c->d2["HTE"]; G.HTN = c->d2["HTN"];
      if (!G.HTE || !G.HTN) { c->err = "gm: HTE / HTN missing"; return 1; }
    }
    G.HYX = c->d2["gmHYX"]; G.HXY = c->d2["gmHXY"]; G.RBR = c->d2["gmRBR"]; G.DXT = c->d2["DXT"]; G.DYT = c->d2["DYT"];
    if (!G.HYX || !G.HXY || !G.RBR || !G.DXT || !G.DYT) { c->err = "gm: grid fields missing"; return 1; }
    // hmix_gm_nml (hmix_gm.F90:364-428); 0 = the value of the default set-up
    G.ah = cfg->ah;
    G.ah_bolus = (cfg->ah_bolus != 0.0) ? cfg->ah_bolus : cfg->ah;
    G.ah_bkg_srfbl = (cfg->ah_bkg_srfbl != 0.0) ? cfg->ah_bkg_srfbl : cfg->ah;
    G.slm_r = (cfg->slm_r != 0.0) ? cfg->slm_r : 0.3;
    G.slm_b = (cfg->slm_b != 0.0) ? cfg->slm_b : 0.3;
    G.slope_tanh = cfg->gm_slope_control == 1; G.slope_ctl = cfg->gm_slope_control;
    G.HUS = c->d2["HUS"]; G.HUW = c->d2["HUW"];
    if (cfg->gm_kappa_type == 2) {   // KAPPA_VERTICAL(k) = kappa_depth(k) (:850-874): a 1-D profile, uploaded as the level table the coefficient kernel reads
      std::vector<double> kd(h.km + 2, 1.0);
      const double sc = (cfg->kappa_depth_scale != 0.0) ? cfg->kappa_depth_scale : 150000.0;
      for (int k = 1; k <= h.km; ++k) kd[k] = cfg->kappa_depth_1 + cfg->kappa_depth_2 * std::exp(-h.zt[k] / sc);
      double *p = nullptr;
      if (dev_upload(c, &p, kd.data(), kd.size())) return 1;
      G.kdepth = p;
    }
    G.kappa_bkg = cfg->gm_kappa_bkg_srfbl == 1; G.ah_bkg_bottom = cfg->ah_bkg_bottom;
    G.diff_tapering = G.slm_r != G.slm_b;                              // :964-968
    G.cancellation = !(G.diff_tapering || G.ah != G.ah_bolus) && !G.tlt;   // :970-987 (both kappa types equal)
    if (!G.cancellation && !tun_off(h.tun.gm_sf_stored)) for (int t = 0; t < 8; ++t) if (dev_alloc(c, &G.SF[t], a3g)) return 1;
  }
#define GI(f) g.f = c->di2[#f]
  GI(KMT); GI(KMU); GI(KMTN); GI(KMTS); GI(KMTE); GI(KMTW); GI(KMTEE); GI(KMTNN);
#undef GI
  const size_t a2 = h.n2 * h.nblocks, a3 = h.n3 * h.nblocks;
  for (int t = 0; t < 3; ++t) {
    for (int n = 0; n < h.nt; ++n) if (dev_alloc(c, &c->TR[n][t], a3)) return 1;
    if (dev_alloc(c, &c->U[t], a3) || dev_alloc(c, &c->V[t], a3) || dev_alloc(c, &c->RHO[t], a3)) return 1;
    if (dev_alloc(c, &c->PS[t], a2) || dev_alloc(c, &c->GX[t], a2) || dev_alloc(c, &c->GY[t], a2) || dev_alloc(c, &c->UB[t], a2) || dev_alloc(c, &c->VB[t], a2)) return 1;
  }
  for (int n = 0; n < h.nt; ++n)
    if (dev_alloc(c, &c->STF[n], a2) || dev_alloc(c, &c->TFW[n], a2) || dev_alloc(c, &c->KPP_SRC[n], a3)) return 1;
  // KPP without double diffusion gives both tracer classes the same diffusivity, value for value (vmix_kpp.F90 ri_iwmix: VDC(:,:,k,2) =
  // VDC(:,:,k,1); blmix applies the same shape function to both): one array then serves both, so the KPP kernels write it once and
  // the tracer kernels find the second read in cache.  pop_get_field("VDC", n) returns it for n = 0 and 1.
  c->vdc_shared = cfg->vmix_choice == 3 && !cfg->ldbl_diff && !tun_off(h.tun.vdc_shared);
  if (dev_alloc(c, &c->VDC[0], (size_t)(h.km + 2) * a2)) return 1;
  if (c->vdc_shared) c->VDC[1] = c->VDC[0];
  else if (dev_alloc(c, &c->VDC[1], (size_t)(h.km + 2) * a2)) return 1;
  double **two[] = {&c->PGUESS, &c->FW, &c->FW_OLD, &c->SHF_QSW, &c->CHL, &c->DH, &c->DHU, &c->ZX, &c->ZY, &c->UH, &c->VH, &c->W3, &c->W4, &c->RHS,
                    &c->R, &c->S0, &c->S1, &c->Q, &c->Z, &c->AZ, &c->HBLT, &c->HMXL, &c->HMXL_DR};
  for (auto p : two) if (dev_alloc(c, p, a2)) return 1;
  c->centerWgt = c->d2["centerWgt"];
  double **three[] = {&c->VVC, &c->E3, &c->F3, &c->S3a, &c->S3b, &c->S3c, &c->S3d};
  for (auto p : three) if (dev_alloc(c, p, a3)) return 1;
  c->d2t[0] = c->S3a; c->d2t[1] = c->S3b; c->d2u[0] = c->S3a; c->d2u[1] = c->S3b;
  if (!tun_off(h.tun.side_stream)) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_d2t, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_d2u, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_vmixu, hipEventDisableTiming));
    if ((cfg->hmix_tracer == 4 || cfg->hmix_momentum == 4) && !tun_off(h.tun.del4_side)) {
      if (dev_alloc(c, &c->d2t[0], a3) || dev_alloc(c, &c->d2t[1], a3) || dev_alloc(c, &c->d2u[0], a3) || dev_alloc(c, &c->d2u[1], a3)) return 1;
      c->side_del4 = true;
      // no tripole fold (the ghost ring of the field comes from a halo update, which is the same arithmetic only where ghost cells are
      // plain copies), centred advection through the LDS kernel, bandwidth-bound grids; POP_D2T_FUSE=0|1 overrides the size rule
      const int fuse_env = tun_or(h.tun.d2t_fuse, -1);
      if (cfg->hmix_tracer == 4 && cfg->ns_boundary != 2 && cfg->tadvect == 1 &&
          (fuse_env >= 0 ? fuse_env != 0 : (long long)h.n2 * h.nblocks > (1 << 19)))
        if (dev_alloc(c, &c->d2t_next[0], a3) || dev_alloc(c, &c->d2t_next[1], a3)) return 1;
      if (cfg->hmix_momentum == 4 && cfg->ns_boundary != 2 && !tun_off(h.tun.d2u_fuse) &&
          (fuse_env >= 0 ? fuse_env != 0 : (long long)h.n2 * h.nblocks > (1 << 19)))
        if (dev_alloc(c, &c->d2u_next[0], a3) || dev_alloc(c, &c->d2u_next[1], a3)) return 1;
    }
  }
  c->nchunk = red_grid_x(g);
  if (!g.red_tiles) {   // DevGrid::red_act: the chunks (256 consecutive cells) the fused solver kernels have to visit
    const int nc = (int)((h.n2 + 255) / 256);
    const bool multi = h.nranks > 1;
    std::vector<std::vector<int>> act(h.nblocks), land(h.nblocks);
    size_t longest = 0;
    for (int b = 0; b < h.nblocks; ++b) {
      const int *P = c->opre_host.data() + (size_t)b * (h.n2 + 1);
      for (int k = 0; k < nc; ++k) {
        const long long p0 = 256LL * k, p1 = std::min<long long>(p0 + 256, (long long)h.n2);
        bool is_land = P[p1] == P[p0];
        if (is_land && multi) {   // as red_land(g, deep): chunks within NGHOST of the edge of the physical domain work for other ranks
          const int j0 = (int)(p0 / h.nxb), j1 = (int)((p1 - 1) / h.nxb);
          const int i0 = (j0 == j1) ? (int)(p0 % h.nxb) : 0, i1 = (j0 == j1) ? (int)((p1 - 1) % h.nxb) : h.nxb - 1;
          const BlockInfo &Bk = h.all_blocks[h.local_ids[b] - 1];
          is_land = i0 >= g.ib - 1 + NGHOST && i1 <= Bk.ie - 1 - NGHOST && j0 >= g.jb - 1 + NGHOST && j1 <= Bk.je - 1 - NGHOST;
        }
        (is_land && !(b == 0 && k == 0) ? land[b] : act[b]).push_back(k);
      }
      longest = std::max(longest, act[b].size());
      c->red_active_total += (int)act[b].size();
    }
    // launch order: workgroup w runs on XCD w % 8; XCD x takes one contiguous band of the chunks with ocean (as red_band does
    // for the full launch), so the rows j +- 1 of the 9-point matvec are in the L2 that fetched row j.  The publishing workgroup
    // (0,0) stays chunk 0.  Padding (land chunks) fills each band up to the common length.
    longest = 16 * ((longest + 15) / 16);   // multiple of 16: k_fpcg_a_pair takes entries e and e + 8
    std::vector<int> list((size_t)longest * h.nblocks), cnt_pad(h.nblocks);
    bool can_pad = true;
    for (int b = 0; b < h.nblocks; ++b) if (act[b].size() + land[b].size() < longest) can_pad = false;
    for (int b = 0; b < h.nblocks && can_pad; ++b) {
      std::vector<int> seq = act[b];
      for (size_t w = act[b].size(); w < longest; ++w) seq.push_back(land[b][w - act[b].size()]);   // shorter list => it has land chunks to pad with
      const size_t per = longest / 8;
      for (size_t w = 0; w < longest; ++w) list[(size_t)b * longest + (w % per) * 8 + w / per] = seq[w];
    }
    if (!can_pad) longest = (size_t)nc;   // no list
    if (longest < (size_t)nc) {
      std::vector<int> cnt(h.nblocks);
      for (int b = 0; b < h.nblocks; ++b) cnt[b] = (int)act[b].size();
      if (dev_upload(c, &c->red_act, list.data(), list.size()) || dev_upload(c, &c->red_cnt, cnt.data(), cnt.size())) return 1;
      c->red_nact = (int)longest;
    }
  }
  if (dev_alloc(c, &c->partial, (size_t)c->nchunk * h.nblocks * 2 + POP_RELAY_SLACK) || dev_alloc(c, &c->blocksum, (size_t)h.nblocks_tot * 4)) return 1;
  { std::vector<int> io(h.nblocks_tot); for (int b = 0; b < h.nblocks_tot; ++b) io[b] = b; if (dev_upload(c, &c->iota, io.data(), io.size())) return 1; }
  if (dev_alloc(c, &c->sc, 1)) return 1;
  { std::vector<int> gid(h.nblocks); for (int lb = 0; lb < h.nblocks; ++lb) gid[lb] = h.local_ids[lb] - 1; if (dev_upload(c, &c->gid, gid.data(), gid.size())) return 1; }
  { std::vector<int> log(h.nblocks_tot, -1); for (int lb = 0; lb < h.nblocks; ++lb) log[h.local_ids[lb] - 1] = lb; if (dev_upload(c, &c->loc_of_gid, log.data(), log.size())) return 1; }
  // halo plan lists
  c->ncopy = (int)h.halo.copy_dst.size(); c->nfill = (int)h.halo.fill_dst.size();
  if (c->ncopy && (dev_upload(c, &c->copy_dst, h.halo.copy_dst.data(), c->ncopy) || dev_upload(c, &c->copy_src, h.halo.copy_src.data(), c->ncopy))) return 1;
  if (c->nfill && dev_upload(c, &c->fill_dst, h.halo.fill_dst.data(), c->nfill)) return 1;
  for (auto &pp : h.halo.peers) {
    DevPeer d; d.rank = pp.rank; d.nsend = (int)pp.send_src.size(); d.nrecv = (int)pp.recv_dst.size();
    if (d.nsend && dev_upload(c, &d.send_src, pp.send_src.data(), d.nsend)) return 1;
    if (d.nrecv && dev_upload(c, &d.recv_dst, pp.recv_dst.data(), d.nrecv)) return 1;
    c->peers.push_back(d);
  }
  if (cfg->ns_boundary == 2) {   // tripole plan (non-empty on the rank that owns the top row of blocks) + evaluation buffer
    size_t nmax = 1;
    for (int loc = 0; loc < 5; ++loc) {
      const TripolePlan &T = h.halo.tripole[loc];
      c->tp_n[loc] = (int)T.dst.size();
      nmax = std::max(nmax, T.dst.size());
      if (c->tp_n[loc] && (dev_upload(c, &c->tp_dst[loc], T.dst.data(), T.dst.size()) || dev_upload(c, &c->tp_a[loc], T.a.data(), T.a.size()) ||
                           dev_upload(c, &c->tp_b[loc], T.b.data(), T.b.size()))) return 1;
    }
    if (dev_alloc(c, &c->tp_buf, nmax * (size_t)(h.km + 2))) return 1;
  }
  {   // concatenated peer lists
    std::vector<int> ss, st, sc, rd, rt, rc;
    for (auto &pp : h.halo.peers) {
      const int s0 = (int)ss.size(), r0 = (int)rd.size();
      for (int v : pp.send_src) { ss.push_back(v); st.push_back(s0); sc.push_back((int)pp.send_src.size()); }
      for (int v : pp.recv_dst) { rd.push_back(v); rt.push_back(r0); rc.push_back((int)pp.recv_dst.size()); }
    }
    c->nsend_all = (int)ss.size(); c->nrecv_all = (int)rd.size();
    if (!h.halo.peers.empty()) {   // per-cell maps of the same lists for the kernels that pack / read a one-level message themselves
      std::vector<int> smap(a2, -1), rmp(a2, -1), off(1, 0), slot;
      std::map<int, std::vector<int>> by_cell;
      for (size_t t = 0; t < ss.size(); ++t) by_cell[ss[t]].push_back((int)t);
      for (auto &kv : by_cell) { smap[kv.first] = (int)off.size() - 1; for (int t : kv.second) slot.push_back(t); off.push_back((int)slot.size()); }
      for (size_t t = 0; t < rd.size(); ++t) rmp[rd[t]] = (int)t;
      // the same answer on every rank (the overlapped update uses the second communicator: all ranks or none): every
      // block's east neighbour -- the cyclic one included -- is owned by the block's own rank, i.e. the shards are j-bands
      c->halo_ns_only = true;
      for (const BlockInfo &B : h.all_blocks) {
        int ie = B.iblock + 1;
        if (ie > h.nbx) { if (h.c.ew_boundary != 1) continue; ie = 1; }
        for (const BlockInfo &E : h.all_blocks)
          if (E.jblock == B.jblock && E.iblock == ie && h.block_owner[E.block_id - 1] != h.block_owner[B.block_id - 1]) c->halo_ns_only = false;
      }
      if (dev_upload(c, &c->sendmap, smap.data(), smap.size()) || dev_upload(c, &c->send_off, off.data(), off.size()) ||
          dev_upload(c, &c->send_slot, slot.data(), std::max<size_t>(slot.size(), 1)) || dev_upload(c, &c->rmap, rmp.data(), rmp.size())) return 1;
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_sa, hipEventDisableTiming));
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_sx, hipEventDisableTiming));
    }
    if (c->nsend_all && (dev_upload(c, &c->sa_src, ss.data(), ss.size()) || dev_upload(c, &c->sa_start, st.data(), st.size()) || dev_upload(c, &c->sa_cnt, sc.data(), sc.size()))) return 1;
    if (c->nrecv_all && (dev_upload(c, &c->ra_dst, rd.data(), rd.size()) || dev_upload(c, &c->ra_start, rt.data(), rt.size()) || dev_upload(c, &c->ra_cnt, rc.data(), rc.size()))) return 1;
  }
  {   // source map for the fused solver path: ghost cell -> local source cell, -1 = fill value
    std::vector<int> sm(a2);
    for (size_t p = 0; p < a2; ++p) sm[p] = (int)p;
    for (size_t i = 0; i < h.halo.copy_dst.size(); ++i) sm[h.halo.copy_dst[i]] = h.halo.copy_src[i];
    for (int d : h.halo.fill_dst) sm[d] = -1;
    if (cfg->ns_boundary == 2) {   // centre scalars beyond the fold are plain mirrored copies of physical cells (TripolePlan loc 0: no symmetrised row)
      const TripolePlan &T = h.halo.tripole[0];
      for (size_t e = 0; e < T.dst.size(); ++e) sm[T.dst[e]] = T.a[e];
    }
    if (dev_upload(c, &c->srcmap, sm.data(), sm.size())) return 1;
    c->h_srcmap = sm;
    HIPCHK(c, hipHostMalloc((void **)&c->persist_out, 4 * sizeof(double)));
    HIPCHK(c, hipHostMalloc((void **)&c->host_sc, sizeof(SolverScalars)));
    HIPCHK(c, hipHostMalloc((void **)&c->host_rr, 8 * sizeof(double)));
    for (auto &e : c->chk_ev) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    {
      std::vector<int> per(h.nranks, 0);
      for (int o : h.block_owner) if (o >= 0 && o < h.nranks) per[o]++;
      c->max_blocks_per_rank = *std::max_element(per.begin(), per.end());
    }
    c->fused_ok = h.halo.peers.empty() && h.nblocks <= 8 && !tun_on(h.tun.solver_unfused);
    if (cfg->solver_choice == 3) {   // omega_k of P-CSI (POP_SolversMod.F90:1617-1620, 1695): a function of the eigenvalue bounds only
      const double csalpha = 2.0 / (h.pcsi_max_eig - h.pcsi_min_eig);
      const double csbeta = (h.pcsi_max_eig + h.pcsi_min_eig) / (h.pcsi_max_eig - h.pcsi_min_eig);
      const double csy = csbeta / csalpha;
      std::vector<double> om(cfg->max_iterations + 2);
      om[0] = 1.0 / csy;
      double csomga = 2.0 / csy;
      for (int m = 1; m <= cfg->max_iterations; ++m) { csomga = 1.0 / (csy - csomga / (4.0 * csalpha * csalpha)); om[m] = csomga; }
      c->pcsi_csy = csy;
      if (dev_upload(c, &c->pcsi_omega, om.data(), om.size()) || dev_alloc(c, &c->pcsi_base, 1)) return 1;
    }
    if (use_evp(*cfg)) {   // EVP sub-blocks of the local blocks, coefficient arrays transposed to [cell][sub-block]
      const EvpHost &E = h.evp;
      const size_t nsb = (size_t)E.xnb * E.ynb, S = nsb * h.nblocks;
      constexpr int NC = EVP_LD * EVP_LD, NR = EVP_LE * EVP_LE;
      std::vector<int4> meta(S);
      std::vector<double> cc(S * NC), ne(S * NC), icc(S * NC), ine(S * NC), rinv(S * NR);
      for (int lb = 0; lb < h.nblocks; ++lb) {
        const size_t gb = (size_t)h.local_ids[lb] - 1;
        for (int j = 1; j <= E.ynb; ++j) for (int i = 1; i <= E.xnb; ++i) {
          const size_t sl = lb * nsb + (size_t)(j - 1) * E.xnb + (i - 1), sg = gb * nsb + (size_t)(j - 1) * E.xnb + (i - 1);
          const int is = E.xidx[i], ie = E.xidx[i + 1] + 1, js = E.yidx[j], je = E.yidx[j + 1] + 1;
          int ocean = 0;   // w: no ocean cell in the interior of the sub-block (cells of no block at all count as land)
          {
            const std::vector<int> &KMT = h.i2.at("KMT");
            for (int cj = js + 1; cj <= je - 1; ++cj) for (int ci = is + 1; ci <= ie - 1; ++ci)
              if (KMT[gb * h.n2 + (size_t)(cj - 1) * h.nxb + (ci - 1)] > 0) ocean = 1;
          }
          meta[sl] = make_int4((int)(lb * h.n2 + (size_t)(js - 1) * h.nxb + (is - 1)), (ie - is + 1) | ((je - js + 1) << 8), E.land[sg], (E.land[sg] && !ocean) ? 1 : 0);
          for (int q = 0; q < NC; ++q) {
            cc[q * S + sl] = E.cc[sg * NC + q]; ne[q * S + sl] = E.ne[sg * NC + q];
            icc[q * S + sl] = E.icc[sg * NC + q]; ine[q * S + sl] = E.ine[sg * NC + q];
          }
          for (int q = 0; q < NR; ++q) rinv[q * S + sl] = E.rinv[sg * NR + q];
        }
      }
      int4 *dm; double *d0, *d1, *d2, *d3, *d4;
      if (dev_upload(c, &dm, meta.data(), S) || dev_upload(c, &d0, cc.data(), cc.size()) || dev_upload(c, &d1, ne.data(), ne.size()) ||
          dev_upload(c, &d2, icc.data(), icc.size()) || dev_upload(c, &d3, ine.data(), ine.size()) || dev_upload(c, &d4, rinv.data(), rinv.size())) return 1;
      c->evp = EvpDev{(long long)S, dm, d0, d1, d2, d3, d4, c->d2["evpC0"], c->d2["btropWgtNE"]};
      c->use_evp = true;
      c->evp_fused_ok = c->fused_ok && cfg->solver_choice == 3;   // P-CSI + EVP: step kernel + sub-block solves, two launches per iteration (r3)
      c->fused_ok = false;
    }
    c->no_graph = tun_on(h.tun.solver_nograph);
    c->mom_lds_rows = tun_or(h.tun.momentum_lds, c->mom_lds_rows);
    // tracer RHS through LDS tiles (kernels_tracer_lds.hpp).  Measured against the direct-load kernel: tx0.1v3 15.0 ms ->
    // 12.8 (64x4 tiles) / 13.6 (64x8); gx1v7 0.260 ms -> 0.208 (64x4) / 0.192 (64x8).  POP_TRACER_LDS=0|4|8 overrides.
    c->trc_lds_rows = 4;   // round 3 (branch-free kernels, two waves per SIMD): 64 x 4 tiles at every size (gx1v7 0.149 vs 0.156 ms, tx0.1v3 7.6 vs 8.0)
    c->trc_lds_rows = tun_or(h.tun.tracer_lds, c->trc_lds_rows);
    c->reg_thomas = !tun_on(h.tun.generic_thomas);
    // tracer solve: the register kernel keeps the elimination coefficients of a column in VGPRs instead of writing them to
    // scratch fields and reading them back (the generic corrector moves 46 GB at the L2 for 19 GB of algorithmic traffic).
    // gx1v7: 0.19 vs 0.30 ms.  tx0.1v3: round 1 measured the generic march faster for the predictor (9.9 vs 11.6 ms, whole
    // grid); with land elimination the register form wins for both (same box, A/B twice: corrector 6.3 vs 8.1-8.6 ms,
    // predictor 7.2 vs 7.5-8.1, step -1.2 .. -1.6 ms).  The velocity solve is faster in registers at both sizes.
    c->reg_thomas_t = c->reg_thomas;
    if (tun_set(h.tun.reg_thomas_t)) c->reg_thomas_t = h.tun.reg_thomas_t != 0;
    if (cfg->partial_bottom_cells && tun_on(h.tun.pbc_generic_thomas)) {
      // partial bottom cells (round 3): every Thomas kernel form carries the PBC branches; pop_tuning.pbc_generic_thomas = 1 keeps
      // the scratch-staged ones (the cross-check of the register instantiations)
      c->reg_thomas = false; c->reg_thomas_t = false;
    }
    c->force_presum = tun_on(h.tun.solver_presum);
    c->fpcg_one_cell = tun_off(h.tun.fpcg_b2);
    c->pcsi_two_cell = (h.nxb & 1) == 0 && !g.red_tiles && (long long)c->nchunk * h.nblocks > 2048;
    // two iterations per launch with a tripole fold: the ghost cells beyond the fold are formed as mirror images of their source cells, which
    // must be cells of this rank (true of any decomposition into bands of whole rows, and of one rank)
    bool fold_local = true;
    std::vector<int> jfold(h.nblocks, h.nyb);
    if (cfg->ns_boundary == 2)
      for (int lb = 0; lb < h.nblocks; ++lb) {
        const BlockInfo &B = h.all_blocks[h.local_ids[lb] - 1];
        if (!(B.j_glob[B.je] < 0)) continue;
        jfold[lb] = B.je;
        for (int j = B.je; j < h.nyb; ++j) for (int i = 0; i < h.nxb; ++i) {
          const int cell = lb * (int)h.n2 + j * h.nxb + i;
          if (c->h_srcmap[cell] == cell) fold_local = false;
        }
      }
    const bool two_ok = !use_evp(*cfg) && !g.red_tiles && fold_local, two_on = tun_set(h.tun.pcsi_two_step) ? h.tun.pcsi_two_step != 0 : c->pcsi_two_cell;
    c->pcsi_two_step = two_ok && two_on && h.halo.peers.empty();
    c->pcsi_two_step_dist = two_ok && two_on && !h.halo.peers.empty();
    if ((c->pcsi_two_step || c->pcsi_two_step_dist) && dev_alloc(c, &c->pcsi_raw, a2)) return 1;
    c->pcsi_evp_fused = c->evp_fused_ok && c->evp.C0 && tun_or(h.tun.evp_wave, 3) >= 2 && tun_on(h.tun.pcsi_evp_fused);   // measured slower: off unless asked for
    if (c->pcsi_evp_fused && !c->pcsi_raw && dev_alloc(c, &c->pcsi_raw, a2)) return 1;
    if ((c->pcsi_two_step || c->pcsi_two_step_dist) && cfg->ns_boundary == 2 && dev_upload(c, &c->pcsi_jfold, jfold.data(), jfold.size())) return 1;
    if (tun_set(h.tun.pcsi_step2)) c->pcsi_two_cell = (h.nxb & 1) == 0 && !g.red_tiles && h.tun.pcsi_step2 != 0;
    c->replicated = !h.halo.peers.empty() && cfg->solver_choice == 1 && !use_evp(*cfg) && h.nblocks_tot <= 8 &&
                    (long long)h.n2 * h.nblocks_tot <= (4LL << 20) && !tun_on(h.tun.solver_distributed);
    if (c->replicated) {
      const size_t NG = h.n2 * h.nblocks_tot;
      SolveView &v = c->gv;
      v.g = c->g; v.g.nblocks = h.nblocks_tot;
      double *p;
      if (dev_upload(c, &p, h.f2["btropWgtNE"].data(), NG)) return 1; v.g.WNE = p;
      if (dev_upload(c, &p, h.f2["btropWgtEast"].data(), NG)) return 1; v.g.WEa = p;
      if (dev_upload(c, &p, h.f2["btropWgtNorth"].data(), NG)) return 1; v.g.WNo = p;
      if (dev_upload(c, &p, h.f2["btropXW"].data(), NG)) return 1; v.g.XW = p;
      if (dev_upload(c, &p, h.f2["btropYW"].data(), NG)) return 1; v.g.YW = p;
      if (dev_upload(c, &p, h.f2["centerWgtIndep"].data(), NG)) return 1; v.g.WC0 = p;
      if (dev_upload(c, &p, h.f2["mMask"].data(), NG)) return 1; v.g.mMask = p;
      { std::vector<unsigned char> m8(NG); for (size_t q = 0; q < NG; ++q) m8[q] = h.f2["mMask"][q] != 0.0;
        unsigned char *d8; if (dev_upload(c, &d8, m8.data(), NG)) return 1; v.g.mMask8 = d8; }
      if (dev_upload(c, &c->gTAREA, h.f2["TAREA"].data(), NG)) return 1;
      if (dev_upload(c, &c->gKMT, h.i2["KMT"].data(), NG)) return 1;
      double **vecs[] = {&v.X, &v.R, &v.Z, &v.S0, &v.S1, &v.Q, &v.RHS, &v.C};
      for (auto q : vecs) if (dev_alloc(c, q, NG)) return 1;
      if (dev_alloc(c, &v.partial, (size_t)c->nchunk * h.nblocks_tot * 2 + POP_RELAY_SLACK) || dev_alloc(c, &v.blocksum, (size_t)h.nblocks_tot * 4)) return 1;
      std::vector<int> gsm = global_srcmap(h), gid(h.nblocks_tot);
      for (int b = 0; b < h.nblocks_tot; ++b) gid[b] = b;
      if (dev_upload(c, &v.srcmap, gsm.data(), gsm.size()) || dev_upload(c, &v.gid, gid.data(), gid.size())) return 1;
      v.nchunk = c->nchunk; v.nblocks_tot = h.nblocks_tot;
    }
  }
  // vmix_const: constant coefficients for all time (vmix_const.F90:121-122)
  if (cfg->vmix_choice == 1) {
    std::vector<double> v((size_t)(h.km + 2) * a2, cfg->const_vdc), w(a3, cfg->const_vvc);
    HIPCHK(c, hipMemcpy(c->VDC[0], v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->VVC, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  if (mix_create(c->h, c->g, c->mix, c->allocs, c->err)) return 1;
  c->KBL = const_cast<int *>(mix_kpp_kbl(c->mix));
  if (cfg->lsw_absorb != 0 && sw_tables_create(c->h, c->allocs, c->err)) return 1;   // lsw_absorb without KPP's lshort_wave
  if (c->h.sw.CHLI) {   // the default chlorophyll amount the table index was built for
    std::vector<double> chl((size_t)c->g.n2 * c->g.nblocks, 0.25);
    HIPCHK(c, hipMemcpy(c->CHL, chl.data(), chl.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  // KPP look-ahead (bandwidth-bound grids; POP_KPP_AHEAD=0|1 overrides): second set of KPP outputs, own stream
  // ... and (round 4) small grids whose pcg solve is the one resident launch: that launch keeps at most one workgroup per CU busy waiting
  // on memory for 2 ms, the KPP kernels of the next step fill the rest (gx1v7: 3.26 -> 3.09, 3.36 -> 3.18 ms per step, A/B on one box)
  // (pcg and P-CSI -- 1.85 -> 1.81 ms there; beside the resident ChronGear the look-ahead costs more than it hides: 3.08 -> 3.34 ms per step)
  const bool resident_solve = (cfg->solver_choice == 1 || cfg->solver_choice == 3) && !use_evp(*cfg) && !tun_off(h.tun.pcg_persist) && h.nranks == 1 && h.nblocks <= 8 &&
                              h.n2 * h.nblocks <= 250u * 8u * 256u;
  c->ahead_enabled = cfg->vmix_choice == 3 && c->side && (h.n2 * h.nblocks > (1u << 19) || resident_solve);
  if (tun_set(h.tun.kpp_ahead)) c->ahead_enabled = cfg->vmix_choice == 3 && c->side && h.tun.kpp_ahead != 0;
  // (Gent-McWilliams adds its isopycnal part to VDC after vmix_coeffs -- to the arrays swapped in, at the step they belong to; with the
  // mixed-layer-depth diagnostics the look-ahead writes a second pair HMXL / HMXL_DR that is swapped in with the rest)
  if (c->ahead_enabled) {
    for (int n = 0; n < 2; ++n) if (dev_alloc(c, &c->KPPa[n], a3)) return 1;
    if (dev_alloc(c, &c->VDCa[0], (size_t)(h.km + 2) * a2)) return 1;
    if (c->vdc_shared) c->VDCa[1] = c->VDCa[0];
    else if (dev_alloc(c, &c->VDCa[1], (size_t)(h.km + 2) * a2)) return 1;
    if (dev_alloc(c, &c->VVCa, a3) || dev_alloc(c, &c->HBLTa, a2) || dev_alloc(c, &c->KBLa, a2)) return 1;
    if (c->h.c.kpp_ml_diagnostics == 1 && (dev_alloc(c, &c->HMXLa, a2) || dev_alloc(c, &c->HMXL_DRa, a2))) return 1;
    if (c->prio_on) HIPCHK(c, hipStreamCreateWithPriority(&c->ahead, hipStreamNonBlocking, c->prio_least));
    else HIPCHK(c, hipStreamCreateWithFlags(&c->ahead, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_ahead_fork, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_ahead, hipEventDisableTiming));
  }
  // initial state (initial.F90:1389-1427, 1660-1676): T,S on all three levels, RHO(cur), RHO(old)
  c->oldt = 0; c->curt = 1; c->newt = 2; c->mixt = 1;
  for (int t = 0; t < 3; ++t) {
    HIPCHK(c, hipMemcpy(c->TR[0][t], h.f3["TEMP0"].data(), a3 * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->TR[1][t], h.f3["SALT0"].data(), a3 * sizeof(double), hipMemcpyHostToDevice));
  }
  launch_state3d(c->g, c->TR[0][c->curt], c->TR[1][c->curt], c->RHO[c->curt], c->stream);
  launch_state3d(c->g, c->TR[0][c->oldt], c->TR[1][c->oldt], c->RHO[c->oldt], c->stream);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  h.f3.clear();
  return 0;
}

int pop_destroy(pop_ctx *c) {
  if (c) for (hipEvent_t &e : c->ev_solve) if (e) { hipEventDestroy(e); e = nullptr; }
  if (!c) return 0;
  for (auto &g : c->graphs) hipGraphExecDestroy(g.second);
  for (auto &g : c->pcsi_graphs) hipGraphExecDestroy(g.second);
  if (c->host_sc) hipHostFree(c->host_sc);
  if (c->host_rr) hipHostFree(c->host_rr);
  if (c->persist_out) hipHostFree(c->persist_out);
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  if (c->ev_d2t) hipEventDestroy(c->ev_d2t);
  if (c->ev_d2u) hipEventDestroy(c->ev_d2u);
  if (c->ev_vmixu) hipEventDestroy(c->ev_vmixu);
  if (c->ahead) { hipStreamSynchronize(c->ahead); hipStreamDestroy(c->ahead); }
  if (c->ev_ahead_fork) hipEventDestroy(c->ev_ahead_fork);
  if (c->ev_ahead) hipEventDestroy(c->ev_ahead);
  if (c->ev_sa) hipEventDestroy(c->ev_sa);
  if (c->ev_sx) hipEventDestroy(c->ev_sx);
  for (auto &e : c->chk_ev) if (e) hipEventDestroy(e);
  if (c->rccl_tr) {
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->comm_side) hipStreamSynchronize(c->comm_side);
    if (c->rccl_tr->comm2) rccl().CommDestroy(c->rccl_tr->comm2);
    if (c->rccl_tr->comm) rccl().CommDestroy(c->rccl_tr->comm);
    delete c->rccl_tr;
  }
  if (c->comm_side) hipStreamDestroy(c->comm_side);
  if (c->side) { hipStreamSynchronize(c->side); hipStreamDestroy(c->side); }
  kpp_destroy(c->mix);
  for (void *p : c->allocs) hipFree(p);
  if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
  delete c;
  return 0;
}
const char *pop_last_error(const pop_ctx *c) { return c ? c->err.c_str() : "null context"; }

static int solver_path_code(const pop_ctx *c);
int pop_get_dim(const pop_ctx *c, const char *name) {
  const std::string n(name);
  if (n == "nx_block") return c->h.nxb;
  if (n == "ny_block") return c->h.nyb;
  if (n == "km") return c->h.km;
  if (n == "nt") return c->h.nt;
  if (n == "nblocks") return c->h.nblocks;
  if (n == "nblocks_tot") return c->h.nblocks_tot;
  if (n == "nblocks_x") return c->h.nbx;
  if (n == "nblocks_y") return c->h.nby;
  if (n == "nghost") return NGHOST;
  if (n == "oldtime") return c->oldt;
  if (n == "curtime") return c->curt;
  if (n == "newtime") return c->newt;
  if (n == "leapfrogts") return c->leapfrogts;
  if (n == "land_skip_active") return c->g.skip;
  if (n == "d2t_fused") return c->d2t_next[0] != nullptr;   // the tracer / momentum kernels also form the next step's del4 first Laplacian
  if (n == "d2u_fused") return c->d2u_next[0] != nullptr;
  if (n == "d2t_last_formed") return c->d2t_last_formed;   // ... and whether the last such launch did (not on averaging steps)
  if (n == "d2u_last_formed") return c->d2u_last_formed;
  if (n == "avg_ts") return c->avg_ts;
  if (n == "nsteps_total") return c->nsteps_total;
  if (n == "nsteps_per_interval") return c->h.nsteps_per_interval;
  if (n == "solver_stream_ops") return (int)c->solver_ops;
  if (n == "solver_iterations_enqueued") return (int)c->solver_enq;
  if (n == "rank") return c->h.rank;
  if (n == "nranks") return c->h.nranks;
  if (n == "solver_chunks_per_block") return c->nchunk;                 // 256-cell chunks of a block (the full launch of the fused solver kernels)
  if (n == "solver_chunks_listed") return c->red_act ? c->red_nact : -1;  // compacted launch: workgroups per block (-1: no list, every chunk is launched)
  if (n == "solver_chunks_active") return c->red_active_total;          // chunks with an ocean cell (or work for another rank), all local blocks
  if (n == "pcg_persist_used") return c->persist_used;
  if (n == "pcg_persist_gave_up") return c->persist_gave_up;
  if (n == "pcsi_evp_fused") return c->pcsi_evp_fused ? 1 : 0;   // P-CSI + EVP: one launch per iteration (k_pcsi_evp_step) in use
  if (n == "pcsi_two_step") return (c->pcsi_two_step || c->pcsi_two_step_dist) ? 1 : 0;   // P-CSI: two iterations per launch (k_pcsi_step_x2) in use
  if (n == "pcg_persist_workgroups") return c->persist_nwg;
  if (n == "pcg_persist_chunks_per_workgroup") return c->persist_cp;
  if (n == "solver_path") return c->host_only ? 0 : solver_path_code(c);   // 1 per operation, 2 fused, 3 fused distributed, 4 replicated fused
  if (n == "thomas_register_tracers") return c->reg_thomas_t && (c->g.km == 60 || c->g.km == 62);    // column-in-registers Thomas kernels in use
  if (n == "thomas_register_velocity") return c->reg_thomas && (c->g.km == 60 || c->g.km == 62);
  if (n == "max_blocks_per_rank") return c->max_blocks_per_rank;
  if (n == "ocean_columns_local") return (int)c->h.ocean_cols_local;     // POP_CREATE_PLAN_ONLY contexts (-1 otherwise)
  if (n == "ocean_columns_total") return (int)c->h.ocean_cols_total;
  return -1;
}
double pop_get_scalar(const pop_ctx *c, const char *name) {
  const std::string n(name);
  if (n == "dtt") return c->h.dtt;
  if (n == "dtu") return c->h.dtu;
  if (n == "dtp") return c->h.dtp;
  if (n == "residualNorm") return c->h.residualNorm;
  if (n == "convergenceCriterion") return c->h.convergenceCriterion;
  if (n == "rcheck") return c->h.rcheck;
  if (n == "rconst") return c->h.rconst;
  if (n == "uarea_equator") return c->h.uarea_equator;
  if (n == "rmsResidual") return c->rmsResidual;
  if (n == "land_tile_fraction") return c->land_fraction;     // 64-column row segments without an ocean cell nearby (land elimination)
  if (n == "PcsiMaxEigs") return c->h.pcsi_max_eig;
  if (n == "PcsiMinEigs") return c->h.pcsi_min_eig;
  if (n == "lanczos_steps") return (double)c->h.pcsi_lanczos_steps;
  if (n == "robert_curtime") return c->h.robert_curtime;
  if (n == "robert_newtime") return c->h.robert_newtime;
  if (n == "rf_volume_2_km") return c->h.rf_volume_2_km;
  if (n == "open_ocean_volume_2_km") return c->h.open_ocean_volume_2_km;
  if (n == "bgtarea_t_1") return c->h.bgtarea_t_1;
  if (n == "persist_iterations") return c->persist_out ? c->persist_out[0] : NAN;   // what the last resident pcg launch reported (kernels_pcg_persist.hpp)
  if (n == "persist_rr") return c->persist_out ? c->persist_out[1] : NAN;
  if (n == "persist_status") return c->persist_out ? c->persist_out[2] : NAN;
  if (n == "persist_checks") return c->persist_out ? c->persist_out[3] : NAN;
  if (n == "rf_S1") return c->rf_S[0];
  if (n == "rf_S2") return c->rf_S[1];
  // HIP-event time of the barotropic solves (POP_SolversRun) since "solver_ms_reset", the iterations they took and their number
  if (n == "solver_ms_total") { solve_collect(const_cast<pop_ctx *>(c)); return c->solver_ms_total; }
  if (n == "solver_iterations_total") { solve_collect(const_cast<pop_ctx *>(c)); return (double)c->solver_iters_total; }
  if (n == "solver_calls_total") { solve_collect(const_cast<pop_ctx *>(c)); return (double)c->solver_calls_total; }
  if (n == "solver_ms_reset") { pop_ctx *m = const_cast<pop_ctx *>(c); solve_collect(m); m->solver_ms_total = 0; m->solver_iters_total = 0; m->solver_calls_total = 0; return 0.0; }
  return NAN;
}
int pop_get_block(const pop_ctx *c, int block_id, int *out8, int *i_glob, int *j_glob) {
  if (block_id < 1 || block_id > c->h.nblocks_tot) return 1;   // get_block: invalid block_id (blocks.F90:309-311)
  const BlockInfo &B = c->h.all_blocks[block_id - 1];
  if (out8) { int v[8] = {B.block_id, B.local_id, B.ib, B.ie, B.jb, B.je, B.iblock, B.jblock}; std::copy(v, v + 8, out8); }
  if (i_glob) std::copy(B.i_glob.begin(), B.i_glob.end(), i_glob);
  if (j_glob) std::copy(B.j_glob.begin(), B.j_glob.end(), j_glob);
  return 0;
}
int pop_local_block_ids(const pop_ctx *c, int *ids) { std::copy(c->h.local_ids.begin(), c->h.local_ids.end(), ids); return 0; }

long long pop_field_count(const pop_ctx *c, const char *name) {
  const std::string n(name);
  const long long a2 = (long long)c->h.n2 * c->h.nblocks, a3 = (long long)c->h.n3 * c->h.nblocks;
  for (const char *s : {"TRACER", "UVEL", "VVEL", "RHO", "KPP_SRC", "VVC", "UISOP", "VISOP", "WISOP", "GM_SF_SLX", "GM_SF_SLY"}) if (n == s) return a3;
  if (n == "VDC") return (long long)c->h.n2 * (c->h.km + 2) * c->h.nblocks;
  return a2;
}
// wait for side-stream work whose results the launch stream (or the host) is about to use
// drop a KPP look-ahead in flight: whatever follows on the launch stream is ordered after it, and the next step computes
// its coefficients itself.  Every entry point through which a caller may read or change fields (join_side) does this, so
// only an uninterrupted sequence of pop_step calls uses the look-ahead.
static int ahead_cancel(pop_ctx *c) {
  if (c->ahead_valid) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_ahead, 0)); c->ahead_valid = false; }
  c->d2t_next_valid = false;   // same rule for the first Laplacians the tracer / momentum kernels formed for the next step
  c->d2u_next_valid = false;
  return 0;
}
static int phase_impvmixu(pop_ctx *c, hipStream_t st);
static int join_side(pop_ctx *c, bool keep_ahead = false) {
  // implicit vertical mixing of U, V held back for the fused form of baroclinic_correct_adjust: a caller that looks at (or changes)
  // fields in between gets the plain kernel now, and the barotropic velocity is added by its own launch later
  if (c->vmixu_deferred) { c->vmixu_deferred = false; if (phase_impvmixu(c, nullptr)) return 1; }
  if (c->vmixu_pending) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_vmixu, 0)); c->vmixu_pending = false; }
  if (!keep_ahead && ahead_cancel(c)) return 1;
  return 0;
}
int pop_get_field(pop_ctx *c, const char *name, int tl, int n, double *host, long long count) {
  const std::string nm(name);
  if (c->host_only) {   // host-only contexts expose the init-time 2-D fields of the local blocks
    std::string key = nm;
    if (nm == "SMF") key = n == 0 ? "SMF1" : "SMF2";
    if (nm == "SMFT") key = n == 0 ? "SMFT1" : "SMFT2";
    auto it = c->h.f2.find(key);
    if (it == c->h.f2.end()) { c->err = "unknown field " + nm; return 1; }
    auto loc = local_part(c->h, it->second);
    if ((long long)loc.size() != count) { c->err = "count mismatch for " + nm; return 1; }
    std::copy(loc.begin(), loc.end(), host);
    return 0;
  }
  double *p; long long cnt;
  if (resolve(c, nm, tl, n, &p, &cnt)) { c->err = "unknown field " + nm; return 1; }
  if (cnt != count) { c->err = "count mismatch for " + nm; return 1; }
  if (join_side(c)) return 1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(host, p, cnt * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}
int pop_set_field(pop_ctx *c, const char *name, int tl, int n, const double *host, long long count) {
  if (need_device(c)) return 1;
  double *p; long long cnt;
  if (resolve(c, name, tl, n, &p, &cnt)) { c->err = std::string("unknown field ") + name; return 1; }
  if (cnt != count) { c->err = std::string("count mismatch for ") + name; return 1; }
  if (join_side(c)) return 1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(p, host, cnt * sizeof(double), hipMemcpyHostToDevice));
  if (!strcmp(name, "CHL") && c->h.sw.CHLI) {   // set_chl (sw_absorption.F90:500-512): the column of the transmission table per cell
    std::vector<int> idx((size_t)cnt);
    for (long long q = 0; q < cnt; ++q) idx[q] = sw_chl_index(c->h.sw, host[q]);
    HIPCHK(c, hipMemcpy(c->h.sw.CHLI, idx.data(), (size_t)cnt * sizeof(int), hipMemcpyHostToDevice));
  }
  if (!strcmp(name, "KPP_SRC")) { c->kpp_src_user = true; c->src_dirty = true; c->src_dirty_alt = true; }
  if (!strcmp(name, "TRACER")) c->tr_ghosts_ok[tl == 0 ? c->oldt : tl == 1 ? c->curt : c->newt] = false;
  if (!strcmp(name, "UVEL") || !strcmp(name, "VVEL")) c->uv_ghosts_ok[tl == 0 ? c->oldt : tl == 1 ? c->curt : c->newt] = false;
  // a new prognostic state may carry other values on land: the next steps run every workgroup again (land elimination)
  for (const char *f : {"TRACER", "UVEL", "VVEL", "RHO", "PSURF", "GRADPX", "GRADPY", "UBTROP", "VBTROP", "PGUESS", "FW_OLD"})
    if (!strcmp(name, f)) c->full_left = c->land_full_steps;
  return 0;
}
int pop_get_ifield(pop_ctx *c, const char *name, int *host, long long count) {
  if (!strcmp(name, "KBL")) {   // KPP: level of the boundary-layer depth that belongs to the current KPP_SRC (device-resident)
    if (need_device(c) || join_side(c)) return 1;
    if (!c->KBL) { c->err = "KBL exists with vmix_choice = 3 only"; return 1; }
    if (count != (long long)c->g.n2 * c->g.nblocks) { c->err = "count mismatch"; return 1; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(host, c->KBL, (size_t)count * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
  }
  auto it = c->h.i2.find(name);
  if (it == c->h.i2.end()) { c->err = std::string("unknown integer field ") + name; return 1; }
  auto loc = local_part(c->h, it->second);
  if ((long long)loc.size() != count) { c->err = "count mismatch"; return 1; }
  std::copy(loc.begin(), loc.end(), host);
  return 0;
}
void *pop_field_device_ptr(pop_ctx *c, const char *name, int tl, int n) {
  if (c->host_only) return nullptr;
  join_side(c);   // the caller may read the field on the launch stream
  if (!strcmp(name, "TRACER")) c->tr_ghosts_ok[tl == 0 ? c->oldt : tl == 1 ? c->curt : c->newt] = false;   // ... or write it
  if (!strcmp(name, "UVEL") || !strcmp(name, "VVEL")) c->uv_ghosts_ok[tl == 0 ? c->oldt : tl == 1 ? c->curt : c->newt] = false;
  if (!strcmp(name, "KPP_SRC")) { c->kpp_src_user = true; c->src_dirty = true; c->src_dirty_alt = true; }
  // ... with other values on land: the next steps run every workgroup again, as after pop_set_field
  for (const char *f : {"TRACER", "UVEL", "VVEL", "RHO", "PSURF", "GRADPX", "GRADPY", "UBTROP", "VBTROP", "PGUESS", "FW_OLD"})
    if (!strcmp(name, f)) c->full_left = c->land_full_steps;
  double *p; long long cnt;
  return resolve(c, name, tl, n, &p, &cnt) ? nullptr : (void *)p;
}

// ---- restart files (restart.F90 write_restart :1095-1715, read_restart :184-1088; 'bin' format of io_binary.F90)
static std::string fmt_r8(double v) { char b[40]; snprintf(b, sizeof b, "%.17g", v); return b; }
int pop_write_restart(pop_ctx *c, const char *path) {
  if (need_device(c) || join_side(c)) return 1;
  const HostModel &h = c->h;
  const std::vector<RestartField> fields = restart_fields(h);
  if (h.rank == 0) {
    // calendar of a run that starts 0001-01-01 00:00 with 365-day years (time_management.F90 defaults)
    const double secs = (double)c->nsteps_total * h.dtt;
    const long long day = (long long)(secs / 86400.0);
    const int sod = (int)(secs - 86400.0 * (double)day);
    static const int mdays[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    int doy = (int)(day % 365), month = 1;
    while (doy >= mdays[month - 1]) { doy -= mdays[month - 1]; ++month; }
    std::vector<RestartAttr> at = {
      {"title", "char", "POP restart"}, {"history", "char", "written by libpop_amd"}, {"conventions", "char", "POP binary restart"},
      {"runid", "char", "pop_amd"}, {"iyear", "int", std::to_string(1 + day / 365)}, {"imonth", "int", std::to_string(month)},
      {"iday", "int", std::to_string(doy + 1)}, {"ihour", "int", std::to_string(sod / 3600)}, {"iminute", "int", std::to_string(sod % 3600 / 60)},
      {"isecond", "int", std::to_string(sod % 60)}, {"iyear0", "int", "1"}, {"imonth0", "int", "1"}, {"iday0", "int", "1"},
      {"ihour0", "int", "0"}, {"iminute0", "int", "0"}, {"isecond0", "int", "0"}, {"dtt", "r8", fmt_r8(h.dtt)},
      {"elapsed_days", "int", std::to_string(day)}, {"seconds_this_day", "r8", fmt_r8((double)sod)},
      {"nsteps_total", "int", std::to_string(c->nsteps_total)},
      {"nsteps_this_interval", "int", std::to_string(c->nsteps_this_interval)},   // extension: exact restart inside an averaging interval
      {"eod_last", "log", c->eod ? "T" : "F"},   // restart.F90:346: the step that has just finished ended a day (what the next step's eod_last will be, time_management.F90:1809)
    };
    if (h.c.tmix_opt == 3) {
      static const char *tn[2] = {"TEMP", "SALT"};
      for (int n = 0; n < 2; ++n) if (c->rf_S_prev_valid[n]) at.push_back({std::string("rf_S_prev_") + tn[n], "r8", fmt_r8(c->rf_S_prev[n])});
    }
    if (restart_write_header(h, path, at, fields, c->err)) return 1;
  }
  const int fd = open(path, O_WRONLY | O_CREAT, 0644);
  if (fd < 0) { c->err = std::string("cannot open ") + path + " for writing"; return 1; }
  const size_t a2 = (size_t)c->g.n2 * c->g.nblocks, a3 = (size_t)c->g.n3 * c->g.nblocks;
  std::vector<double> buf(a3);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int rc = 0;
  for (const RestartField &f : fields) {
    const size_t cnt = f.ndims == 3 ? a3 : a2;
    if (f.dev.empty()) std::fill(buf.begin(), buf.begin() + cnt, 0.0);
    else {
      double *p; long long n;
      if (resolve(c, f.dev, f.tl, f.n, &p, &n) || (size_t)n != cnt) { c->err = "restart: unknown field " + f.dev; rc = 1; break; }
      if (hipMemcpy(buf.data(), p, cnt * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { c->err = "restart: device copy failed"; rc = 1; break; }
    }
    const int nz = f.ndims == 3 ? h.km : 1;
    for (int k = 0; k < nz && !rc; ++k)
      rc = restart_slab_io(h, fd, f.id + k, buf.data() + (size_t)k * h.n2, f.ndims == 3 ? h.n3 : h.n2, true, false, c->err);
    if (rc) break;
  }
  if (close(fd) != 0 && !rc) { c->err = "restart: close failed"; rc = 1; }
  return rc;
}

int pop_read_restart(pop_ctx *c, const char *path, int flags) {
  if (need_device(c) || join_side(c)) return 1;
  const HostModel &h = c->h;
  std::map<std::string, std::map<std::string, std::string>> sec;
  if (restart_parse_header(path, sec, c->err)) return 1;
  std::vector<RestartField> fields = restart_fields(h);
  for (RestartField &f : fields) {   // record of each field from the header (define_field_binary :1075-1080), not from our own order
    auto it = sec.find(f.name);
    if (it == sec.end() || !it->second.count("id")) {
      if (f.dev.empty()) { f.id = -1; continue; }                    // FW_FREEZE is not used here
      c->err = "could not find field in binary header file: " + f.name; return 1;
    }
    f.id = atoi(it->second["id"].c_str());
    if (it->second.count("nfield_dims") && atoi(it->second["nfield_dims"].c_str()) != f.ndims) { c->err = "restart: wrong rank for " + f.name; return 1; }
  }
  const int fd = open(path, O_RDONLY);
  if (fd < 0) { c->err = std::string("cannot open ") + path; return 1; }
  const std::vector<int> KMT = local_part(h, c->h.i2["KMT"]), KMU = local_part(h, c->h.i2["KMU"]);
  const size_t a2 = (size_t)c->g.n2 * c->g.nblocks, a3 = (size_t)c->g.n3 * c->g.nblocks;
  std::vector<double> buf(a3);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int rc = 0;
  for (const RestartField &f : fields) {
    if (f.dev.empty() || f.id < 0) continue;
    const size_t cnt = f.ndims == 3 ? a3 : a2;
    std::fill(buf.begin(), buf.begin() + cnt, 0.0);
    const int nz = f.ndims == 3 ? h.km : 1;
    for (int k = 0; k < nz && !rc; ++k)
      rc = restart_slab_io(h, fd, f.id + k, buf.data() + (size_t)k * h.n2, f.ndims == 3 ? h.n3 : h.n2, false, (flags & 1) != 0, c->err);
    if (rc) break;
    restart_mask(h, f, buf.data(), KMT, KMU);
    double *p; long long n;
    if (resolve(c, f.dev, f.tl, f.n, &p, &n) || (size_t)n != cnt) { c->err = "restart: unknown field " + f.dev; rc = 1; break; }
    if (hipMemcpy(p, buf.data(), cnt * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { c->err = "restart: device copy failed"; rc = 1; break; }
    const int uv = (f.mask == 1 || f.mask == 3) ? 1 : 0;             // U-grid fields: NE corner, vector (restart.F90:969-1014)
    if (halo_update(c, p, nz, 0.0, uv, uv)) { rc = 1; break; }       // read_restart :960-1040 (fillValue 0)
  }
  close(fd);
  if (rc) return rc;
  c->full_left = c->land_full_steps;   // land elimination: the state just read is new
  for (bool &b : c->tr_ghosts_ok) b = false;
  for (bool &b : c->uv_ghosts_ok) b = false;
  // init_ts :1665-1681: density of both time levels from the tracers just read
  launch_state3d(c->g, c->TR[0][c->curt], c->TR[1][c->curt], c->RHO[c->curt], c->stream);
  launch_state3d(c->g, c->TR[0][c->oldt], c->TR[1][c->oldt], c->RHO[c->oldt], c->stream);
  HIPCHK(c, hipGetLastError());
  // scalars: initial.F90:1088 first_step = .false.; time_management.F90:1426 nsteps_this_interval = 0 unless the file says otherwise
  std::map<std::string, std::string> &g = sec["GLOBAL"];
  if (!g.count("nsteps_total")) { c->err = "restart: header has no nsteps_total"; return 1; }
  c->nsteps_total = atoi(g["nsteps_total"].c_str());
  c->nsteps_this_interval = g.count("nsteps_this_interval") ? atoi(g["nsteps_this_interval"].c_str()) : 0;
  c->first_step = 0;
  // restart.F90:468: eod_last comes from the file (list-directed logical); time_manager of the next step copies eod into eod_last
  // (time_management.F90:1809), so the value read is parked in eod.  A file without it (another writer): not the end of a day.
  c->eod = 0; c->eod_last = 0;
  if (g.count("eod_last")) { std::string v = g["eod_last"]; size_t k = v.find_first_not_of(" ."); c->eod = (k != std::string::npos && (v[k] == 'T' || v[k] == 't')) ? 1 : 0; }
  // module state of hmix_gm that no restart file carries: KAPPA_VERTICAL is 1 until compute_kappa runs again (init_gm, hmix_gm.F90:860) -- with
  // 'once_a_day' that is the first step of a restart written at the end of a day (eod_last), with 'never' it stays 1 (the reference's behaviour too)
  if (c->gm.KV) {
    std::vector<double> ones((size_t)c->g.n3 * c->g.nblocks, 1.0);
    HIPCHK(c, hipMemcpy(c->gm.KV, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  if (h.c.tmix_opt == 3) {
    static const char *tn[2] = {"TEMP", "SALT"};
    for (int n = 0; n < 2; ++n) {
      const std::string key = std::string("rf_S_prev_") + tn[n];
      c->rf_S_prev_valid[n] = g.count(key) != 0;                     // extract_attrib_file(..., from_file=rf_S_prev_valid) :516-519
      if (c->rf_S_prev_valid[n]) c->rf_S_prev[n] = strtod(g[key].c_str(), nullptr);
    }
  }
  return 0;
}

// ---- time_manager + set_switches (time_management.F90:1823-1847, 2139-2234) ----------------
// DevGrid::skip for the step that starts now; cached solver graphs hold the flag by value
static void land_skip_for_step(pop_ctx *c) {
  // every one of the three rotating time levels must have been the `new` one in a step that ran every workgroup (averaging
  // steps do not rotate: with a short averaging period four steps may not get round), and POP_LAND_FULL_STEPS steps at least
  if (c->full_left == c->land_full_steps) c->full_seen = 0;      // a reset (set-up, restart, new state) starts the count again
  const int want = (c->land_skip && c->full_left == 0 && (c->full_seen == 7 || c->land_full_steps == 0)) ? 1 : 0;
  if (!want) c->full_seen |= 1 << c->newt;
  if (c->full_left > 0) --c->full_left;
  if (want == c->g.skip) return;
  c->g.skip = want;
  if (c->stream) hipStreamSynchronize(c->stream);
  for (auto &g : c->graphs) hipGraphExecDestroy(g.second);
  for (auto &g : c->pcsi_graphs) hipGraphExecDestroy(g.second);
  c->graphs.clear(); c->pcsi_graphs.clear();
}
int pop_time_manager(pop_ctx *c) {
  const pop_config &cf = c->h.c;
  if (!c->host_only) land_skip_for_step(c);
  c->leapfrogts = 1; c->f_euler_ts = 0; c->avg_ts = 0;
  c->nsteps_total += 1;
  if (cf.tmix_opt == 2) { c->nsteps_this_interval += 1; if (c->nsteps_this_interval > c->h.nsteps_per_interval) c->nsteps_this_interval = 1; }
  c->eod_last = c->eod;
  c->eod = (cf.tmix_opt == 2) ? (c->nsteps_this_interval == c->h.nsteps_per_interval) : (c->nsteps_total % cf.steps_per_day == 0);
  if (c->first_step) { c->leapfrogts = 0; c->f_euler_ts = 1; c->first_step = 0; }
  if (cf.tmix_opt == 1 && c->nsteps_total % cf.time_mix_freq == 0) c->avg_ts = 1;
  if (cf.tmix_opt == 2) {
    const int n = c->nsteps_this_interval;
    if (n == 2) c->avg_ts = 1;
    else if (n != 1 && (n + 1) % cf.time_mix_freq != 0 && n % cf.time_mix_freq == 0) c->avg_ts = 1;
  }
  // step_mod.F90:302-320
  if (c->leapfrogts) { c->mixt = c->oldt; c->beta = 1.0 / 3.0; c->c2dtt = 2.0 * c->h.dt[1]; c->c2dtu = 2.0 * c->h.dtu; c->c2dtp = 2.0 * c->h.dtp; }
  else { c->mixt = c->curt; c->beta = 0.5; c->c2dtt = c->h.dt[1]; c->c2dtu = c->h.dtu; c->c2dtp = c->h.dtp; }
  return 0;
}

int pop_dhdt(pop_ctx *c) {
  if (need_device(c)) return 1;
  ScopedPhase ph(c, "DHDT");
  hipLaunchKernelGGL(k_dhdt, dim3(col_grid(c->g, 256), c->g.nblocks), dim3(256), 0, c->stream, c->g, step_params(c),
                     c->PS[c->curt], c->PS[c->oldt], c->FW_OLD, c->DH, c->DHU);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// launchers of the individual baroclinic phases (also used by pop_time_phase)
static MixState kpp_mix_state(pop_ctx *c, int slot, bool into_alt) {
  MixState ms{};
  for (int n = 0; n < 2; ++n) {
    ms.TMIX[n] = c->TR[n][slot]; ms.STF[n] = c->STF[n];
    ms.KPP_SRC[n] = into_alt ? c->KPPa[n] : c->KPP_SRC[n]; ms.VDC[n] = into_alt ? c->VDCa[n] : c->VDC[n];
  }
  ms.UMIX = c->U[slot]; ms.VMIX = c->V[slot]; ms.UCUR = c->U[c->curt]; ms.VCUR = c->V[c->curt]; ms.RHOMIX = c->RHO[slot];
  ms.VVC = into_alt ? c->VVCa : c->VVC; ms.SHF_QSW = c->SHF_QSW; ms.HBLT = into_alt ? c->HBLTa : c->HBLT;
  ms.HMXL = (into_alt && c->HMXLa) ? c->HMXLa : c->HMXL; ms.HMXL_DR = (into_alt && c->HMXL_DRa) ? c->HMXL_DRa : c->HMXL_DR;
  ms.KBL = into_alt ? c->KBLa : c->KBL;
  ms.src_clear_all = (into_alt ? c->src_dirty_alt : c->src_dirty) ? 1 : 0;
  (into_alt ? c->src_dirty_alt : c->src_dirty) = false;   // the evaluation that follows clears the set
  ms.S3a = c->S3a; ms.S3b = c->S3b; ms.S3c = c->S3c; ms.S3d = c->S3d; ms.E3 = c->E3; ms.F3 = c->F3;
  return ms;
}
static int phase_vmix(pop_ctx *c) {
  const pop_config &cf = c->h.c;
  const StepParams sp = step_params(c);
  if (cf.vmix_choice == 1)
    hipLaunchKernelGGL(k_vmix_const, grid_3d(c), dim3(256), 0, c->stream, c->g, sp, c->TR[0][c->mixt], c->TR[1][c->mixt], c->VDC[0], c->VVC);
  else {
    if (c->ahead_valid && c->ahead_slot == c->mixt) {   // computed beside the previous step's solver: swap the output sets in
      HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_ahead, 0));
      c->ahead_valid = false; c->kpp_src_user = false;
      for (int n = 0; n < 2; ++n) { std::swap(c->VDC[n], c->VDCa[n]); std::swap(c->KPP_SRC[n], c->KPPa[n]); }
      std::swap(c->VVC, c->VVCa); std::swap(c->HBLT, c->HBLTa); std::swap(c->KBL, c->KBLa); std::swap(c->src_dirty, c->src_dirty_alt);
      if (c->HMXLa) { std::swap(c->HMXL, c->HMXLa); std::swap(c->HMXL_DR, c->HMXL_DRa); }
      return 0;
    }
    if (ahead_cancel(c)) return 1;
    const MixState ms = kpp_mix_state(c, c->mixt, false);
    if (mix_vmix_coeffs(c->h, c->g, sp, c->mix, ms, c->stream, c->err)) return 1;
    c->kpp_src_user = false;
  }
  return 0;
}
// KPP of the next step on the look-ahead stream.  Valid when the next step is a leapfrog step (mixtime = its oldtime = this
// step's curtime) and nothing rewrites the curtime fields before then (no averaging step, no Robert filter).  Inputs:
// T, S, U, V at curtime and the surface fluxes; scratch: the 3-D work fields, idle between baroclinic_driver and the
// next step; outputs: the second set of VDC / VVC / KPP_SRC / HBLT.
static int kpp_look_ahead(pop_ctx *c) {
  // an averaging step rewrites oldtime and curtime in its tail and does not rotate; the Robert filter rewrites curtime
  // (with the mixed-layer-depth diagnostics on, HMXL / HMXL_DR of the step that just ran stay where they are: the look-ahead writes the second pair)
  if (!c->ahead_enabled || c->h.c.vmix_choice != 3 || c->avg_ts || c->h.c.tmix_opt == 3 || (c->h.c.kpp_ml_diagnostics == 1 && !c->HMXLa)) return 0;
  HIPCHK(c, hipEventRecord(c->ev_ahead_fork, c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->ahead, c->ev_ahead_fork, 0));
  const MixState ms = kpp_mix_state(c, c->curt, true);
  if (mix_vmix_coeffs(c->h, c->g, step_params(c), c->mix, ms, c->ahead, c->err)) return 1;
  HIPCHK(c, hipEventRecord(c->ev_ahead, c->ahead));
  c->ahead_valid = true; c->ahead_slot = c->curt;
  return 0;
}
// hmix_tracer = 3 (horizontal_mix.F90:549-554, hmix_gm.F90:1102-2226): slopes and tapered diffusivities of the mix-time tracers, the
// isopycnal part added to VDC (after vmix_coeffs, before the tracer right-hand side reads it), and the tendency GTK of both tracers
static int phase_hmix_gm(pop_ctx *c) {
  const double *T = c->TR[0][c->mixt], *S = c->TR[1][c->mixt];
  GmDev G = c->gm;
  G.HBLT = (c->h.c.vmix_choice == 3) ? c->HBLT : nullptr;            // BL_DEPTH = KPP_HBLT | zw(1) (:1210-1212)
  const dim3 G3((c->g.n2 + 255) / 256, c->g.km, c->g.nblocks);
  const dim3 G2(G3.x, c->g.nblocks);
  const bool kappa_now = G.KV && (c->h.c.gm_kappa_freq == 1 || c->nsteps_total == 1 || (c->h.c.gm_kappa_freq == 2 && c->eod_last));   // compute_kappa (:1258-1332): the first step of the run, or every step
  if (G.tlt) {   // :1222-1250, then the tapering with the layer's rules, then merged_streamfunction / apply_vertical_profile (:1668-1674)
    G.HMXL = (c->h.c.vmix_choice == 3) ? c->HMXL : nullptr;
    hipLaunchKernelGGL(k_gm_diabatic_depth, G2, dim3(256), 0, c->stream, c->g, G);
    hipLaunchKernelGGL(k_gm_coeffs<1>, G3, dim3(256), 0, c->stream, c->g, G, T, S);
    hipLaunchKernelGGL(k_gm_transition_layer, G2, dim3(256), 0, c->stream, c->g, G);
    if (kappa_now) hipLaunchKernelGGL(k_gm_kappa_vertical, G2, dim3(256), 0, c->stream, c->g, G, T, S, step_params(c).grav);
    hipLaunchKernelGGL(k_gm_coeffs<2>, G3, dim3(256), 0, c->stream, c->g, G, T, S);
    hipLaunchKernelGGL(k_gm_msf_column, G2, dim3(256), 0, c->stream, c->g, G);
  } else {
    if (kappa_now) hipLaunchKernelGGL(k_gm_kappa_vertical, G2, dim3(256), 0, c->stream, c->g, G, T, S, step_params(c).grav);
    hipLaunchKernelGGL(k_gm_coeffs<0>, G3, dim3(256), 0, c->stream, c->g, G, T, S);
  }
  const StepParams sp = step_params(c);
  double *v1 = (sp.nvdc == 2 && c->VDC[1] != c->VDC[0]) ? c->VDC[1] : nullptr;   // one shared array is added to once
  if (G.SF[0]) hipLaunchKernelGGL(k_gm_sf, G3, dim3(256), 0, c->stream, c->g, G);   // without cancellation: SF_SLX, SF_SLY once per half cell
  if (G.UISOP) hipLaunchKernelGGL(k_gm_bolus, G2, dim3(256), 0, c->stream, c->g, G);   // diag_gm_bolus
  if (!tun_off(c->h.tun.gm_flux_tile) && (G.cancellation || G.SF[0])) {
    // straight-line flux functions, every horizontal face flux formed once (64 x 4 patches computing 63 x 3 cells); pop_tuning.gm_flux_tile = 0,
    // or the stream-function terms not stored (gm_sf_stored = 0): the cell-by-cell kernel
    const int R = (c->h.tun.gm_flux_tile == 4) ? 4 : 8;     // rows of the patch: 64 x 8 computing 63 x 7 cells (gm_flux_tile = 4: 64 x 4, measured 2 % of the step slower)
    const dim3 GT(((c->g.nxb + 62) / 63) * ((c->g.nyb + R - 2) / (R - 1)), (c->g.km + POP_GM_KC - 1) / POP_GM_KC, G3.z);
    if (R == 8) {
      if (G.cancellation) hipLaunchKernelGGL((k_gm_flux_tile<8, true>), GT, dim3(64, 8), 0, c->stream, c->g, G, T, S, c->VDC[0], v1);
      else hipLaunchKernelGGL((k_gm_flux_tile<8, false>), GT, dim3(64, 8), 0, c->stream, c->g, G, T, S, c->VDC[0], v1);
    } else {
      if (G.cancellation) hipLaunchKernelGGL((k_gm_flux_tile<4, true>), GT, dim3(64, 4), 0, c->stream, c->g, G, T, S, c->VDC[0], v1);
      else hipLaunchKernelGGL((k_gm_flux_tile<4, false>), GT, dim3(64, 4), 0, c->stream, c->g, G, T, S, c->VDC[0], v1);
    }
  } else
  hipLaunchKernelGGL(k_gm_flux, dim3(G3.x, (c->g.km + POP_GM_KC - 1) / POP_GM_KC, G3.z), dim3(256), 0, c->stream, c->g, G, T, S, c->VDC[0], v1);
  HIPCHK(c, hipGetLastError());
  return 0;
}
static int phase_hmix_tracer(pop_ctx *c, hipStream_t st = nullptr) {   // del4: first Laplacian of the tracers into d2t; gm: the whole tendency
  if (c->h.c.hmix_tracer == 3) return phase_hmix_gm(c);
  if (c->h.c.hmix_tracer != 4) return 0;
  if (c->d2t_next_valid && c->d2t_next_slot == c->mixt) {   // formed by the previous step's tracer kernel (ghost ring already updated)
    c->d2t_next_valid = false;
    std::swap(c->d2t[0], c->d2t_next[0]); std::swap(c->d2t[1], c->d2t_next[1]);
    return 0;
  }
  c->d2t_next_valid = false;
  return mix_hdifft_del4(c->h, c->g, step_params(c), c->mix, c->TR[0][c->mixt], c->TR[1][c->mixt], c->d2t[0], c->d2t[1], c->S3c, c->S3d, st ? st : c->stream, c->err);
}
// advt with tadvect = 3 (advection.F90:1667-1708, 2684-3280; comp_flux_vel_ghost :1014-1120): L(T) of both tracers into lw.XOUT
static int phase_advt_lw_lim(pop_ctx *c) {
  const int km = c->g.km;
  const double *X0 = c->TR[0][c->mixt], *X1 = c->TR[1][c->mixt];
  if (c->g.pbc) hipLaunchKernelGGL(k_lw_flux<true>, grid_cols(c), dim3(POP_COL_THREADS), 0, c->stream, c->g, c->lw, (const double *)c->U[c->curt], (const double *)c->V[c->curt], (const double *)c->DH);
  else hipLaunchKernelGGL(k_lw_flux<false>, grid_cols(c), dim3(POP_COL_THREADS), 0, c->stream, c->g, c->lw, (const double *)c->U[c->curt], (const double *)c->V[c->curt], (const double *)c->DH);
  // UTE: E face, vector; WTKB: centre (comp_flux_vel_ghost :1080-1100).  VTN is not exchanged by the reference: it forms it in
  // the ghost rows from the ghost velocities.  Beyond a tripole fold those are the mirrored velocities with the sign of a
  // vector, so the N-face mirror of VTN with that sign is the same number (the two products are added in the other order);
  // the degenerate top row stays as computed (location 4: N face, ghost rows only).
  if (halo_update_many(c, {{c->lw.UTE, km, 3, 1}, {c->lw.VTN, km, 4, 1}, {c->lw.WTKB, km}})) return 1;
  const dim3 G3((c->g.n2 + 255) / 256, km, c->g.nblocks * 2);
  if (c->g.pbc) {
    hipLaunchKernelGGL(k_lw_z<true>, G3, dim3(256), 0, c->stream, c->g, c->lw, X0, X1, c->c2dtt);
    hipLaunchKernelGGL(k_lw_x<true>, G3, dim3(256), 0, c->stream, c->g, c->lw, X0, X1, c->c2dtt);
    hipLaunchKernelGGL(k_lw_y<true>, G3, dim3(256), 0, c->stream, c->g, c->lw, X0, X1, c->c2dtt);
  } else {
    hipLaunchKernelGGL(k_lw_z<false>, G3, dim3(256), 0, c->stream, c->g, c->lw, X0, X1, c->c2dtt);
    hipLaunchKernelGGL(k_lw_x<false>, G3, dim3(256), 0, c->stream, c->g, c->lw, X0, X1, c->c2dtt);
    hipLaunchKernelGGL(k_lw_y<false>, G3, dim3(256), 0, c->stream, c->g, c->lw, X0, X1, c->c2dtt);
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}
static int phase_tracer_rhs(pop_ctx *c, bool fwd = false) {
  const StepParams sp = step_params(c);
  TracerRhsArgs a{};
  if (c->h.c.tadvect == 3) {
    if (phase_advt_lw_lim(c)) return 1;
    a.LTK[0] = c->lw.XOUT[0]; a.LTK[1] = c->lw.XOUT[1];
  }
  a.E[0] = c->E3; a.F[0] = c->F3; a.E[1] = c->S3c; a.F[1] = c->S3d;
  for (int n = 0; n < 2; ++n) {
    a.TCUR[n] = c->TR[n][c->curt]; a.TOLD[n] = c->TR[n][c->oldt]; a.TMIX[n] = c->TR[n][c->mixt]; a.TNEW[n] = c->TR[n][c->newt];
    a.VDC[n] = c->VDC[sp.nvdc == 2 ? n : 0]; a.KPP_SRC[n] = c->KPP_SRC[n]; a.STF[n] = c->STF[n]; a.TFW[n] = c->TFW[n];
  }
  if (c->h.c.hmix_tracer == 4) { a.TMIX[0] = c->d2t[0]; a.TMIX[1] = c->d2t[1]; }   // del4: second Laplacian acts on D2T
  a.UCUR = c->U[c->curt]; a.VCUR = c->V[c->curt]; a.DH = c->DH; a.PCUR = c->PS[c->curt]; a.POLD = c->PS[c->oldt];
  a.c2dtt = c->c2dtt; a.use_kpp_src = (c->h.c.vmix_choice == 3);
  if (a.use_kpp_src && !c->kpp_src_user && !tun_on(c->h.tun.kpp_src_full)) a.KBL = c->KBL;
  if (c->h.c.lsw_absorb != 0) {   // lsw_absorb: penetrating short wave (add_sw_absorb)
    a.sw_on = 1; a.sw_type = c->h.c.sw_absorption_type; a.sw_ksol = c->h.sw.ksol;
    a.QSW = c->SHF_QSW; a.swabs = c->h.sw.swabs; a.swTr = c->h.sw.Tr; a.swCHLI = c->h.sw.CHLI;
  }
  // the next step's first Laplacian: valid when that step is a leapfrog step whose mix time is this step's current time and nothing
  // rewrites the current tracers before then (no averaging step, no Robert filter) -- the rule of the KPP look-ahead
  // Gent-McWilliams (r4): the LDS kernel with the mixing tendency given (k_tracer_rhs_lds<., ., false, true>); pop_tuning.gm_flux_tile = 0: the generic kernel
  const bool gm_lds = c->h.c.hmix_tracer == 3 && !c->g.pbc && !tun_off(c->h.tun.gm_flux_tile);
  const bool lds_kernel = c->h.c.tadvect == 1 && (c->h.c.hmix_tracer != 3 || gm_lds) && (c->trc_lds_rows == 8 || c->trc_lds_rows == 4);
  if (lds_kernel && gm_lds) { a.HDT[0] = c->gm.GTK[0]; a.HDT[1] = c->gm.GTK[1]; }
  const bool form_next = lds_kernel && !gm_lds && c->d2t_next[0] && !c->avg_ts && c->h.c.tmix_opt != 3 && c->tr_ghosts_ok[c->curt];
  if (form_next) { a.D2N[0] = c->d2t_next[0]; a.D2N[1] = c->d2t_next[1]; a.AHF = c->mix.D4AHF; }
  c->d2t_last_formed = form_next;
  if (lds_kernel) {
    if (c->trc_lds_rows == 8) launch_tracer_lds<8>(c->g, sp, a, c->stream, fwd); else launch_tracer_lds<4>(c->g, sp, a, c->stream, fwd);
    if (form_next && !c->phase_timing) {
      if (halo_update_many(c, {{c->d2t_next[0], c->g.km}, {c->d2t_next[1], c->g.km}})) return 1;
      c->d2t_next_valid = true; c->d2t_next_slot = c->curt;
    }
    return 0;
  }
  if (fwd) { c->err = "fused forward elimination needs the LDS tracer kernel"; return 1; }
  if (c->h.c.hmix_tracer == 3) {   // Gent-McWilliams: the tendency formed by phase_hmix_gm in place of del2 mixing
    a.HDT[0] = c->gm.GTK[0]; a.HDT[1] = c->gm.GTK[1];
    if (c->h.c.tadvect == 2) {
      a.up = c->upw3;
      if (c->g.pbc) hipLaunchKernelGGL((k_tracer_rhs<true, true, true>), grid_stencil(c), block_stencil(), 0, c->stream, c->g, sp, a);
      else hipLaunchKernelGGL((k_tracer_rhs<true, true>), grid_stencil(c), block_stencil(), 0, c->stream, c->g, sp, a);
    } else if (c->g.pbc) hipLaunchKernelGGL((k_tracer_rhs<true, false, true>), grid_stencil(c), block_stencil(), 0, c->stream, c->g, sp, a);
    else hipLaunchKernelGGL((k_tracer_rhs<true, false>), grid_stencil(c), block_stencil(), 0, c->stream, c->g, sp, a);
    return 0;
  }
  if (c->h.c.tadvect == 2) {
    a.up = c->upw3;
    if (c->g.pbc) hipLaunchKernelGGL((k_tracer_rhs<false, true, true>), grid_stencil(c), block_stencil(), 0, c->stream, c->g, sp, a);
    else hipLaunchKernelGGL((k_tracer_rhs<false, true>), grid_stencil(c), block_stencil(), 0, c->stream, c->g, sp, a);
  } else if (c->g.pbc) hipLaunchKernelGGL((k_tracer_rhs<false, false, true>), grid_stencil(c), block_stencil(), 0, c->stream, c->g, sp, a);
  else hipLaunchKernelGGL((k_tracer_rhs<false, false>), grid_stencil(c), block_stencil(), 0, c->stream, c->g, sp, a);
  return 0;
}
static ImpvmixtArgs impvmixt_args(pop_ctx *c, const double *psfc) {
  const StepParams sp = step_params(c);
  ImpvmixtArgs a{};
  for (int n = 0; n < 2; ++n) { a.TNEW[n] = c->TR[n][c->newt]; a.TOLD[n] = c->TR[n][c->oldt]; a.TCUR[n] = c->TR[n][c->curt]; a.VDC[n] = c->VDC[sp.nvdc == 2 ? n : 0]; }
  a.PSFC = psfc; a.POLD = c->PS[c->oldt]; a.PCUR = c->PS[c->curt]; a.PNEW = c->PS[c->newt]; a.PMIX = c->PS[c->mixt];
  a.E = c->E3; a.F = c->F3; a.RHO = c->RHO[c->newt]; a.c2dtt = c->c2dtt; a.nfirst = 1; a.nlast = 2;
  return a;
}
// predictor with the forward elimination inside the right-hand-side kernel: bandwidth-bound grids, centred advection
// through the LDS kernel, generic (scratch-staged) solve -- POP_TRACER_FWD=0|1 overrides
static bool tracer_fwd_fused(const pop_ctx *c) {
  const bool can = c->h.c.tadvect == 1 && c->h.c.hmix_tracer != 3 && (c->trc_lds_rows == 4 || c->trc_lds_rows == 8) && c->h.c.lpressure_avg && c->leapfrogts;
  if (tun_set(c->h.tun.tracer_fwd)) return can && c->h.tun.tracer_fwd != 0;
  return can && !c->reg_thomas_t;
}
static int phase_impvmixt_back(pop_ctx *c) {
  ImpvmixtBackArgs b{};
  for (int n = 0; n < 2; ++n) { b.TNEW[n] = c->TR[n][c->newt]; b.TOLD[n] = c->TR[n][c->oldt]; }
  b.E[0] = c->E3; b.F[0] = c->F3; b.E[1] = c->S3c; b.F[1] = c->S3d;
  const dim3 G = grid_cols(c);
  hipLaunchKernelGGL(k_impvmixt_back, dim3(G.x, G.y, 2), dim3(POP_COL_THREADS), 0, c->stream, c->g, b);
  return 0;
}
static int phase_impvmixt_pred(pop_ctx *c) {
  launch_impvmixt<0, false, false>(c->g, step_params(c), impvmixt_args(c, c->PS[c->curt]), grid_cols(c), c->stream, c->reg_thomas_t, c->h.tun.thomas_pair);
  return 0;
}
static int phase_state_new(pop_ctx *c) {
  launch_state3d(c->g, c->TR[0][c->newt], c->TR[1][c->newt], c->RHO[c->newt], c->stream);
  return 0;
}
// density of the new tracers on the rows j_first <= j < j_end (0-based) of every block
static void state_new_rows(pop_ctx *c, int j_first, int j_end) {
  const int p0 = j_first * c->g.nxb, p1 = j_end * c->g.nxb;
  if (p1 <= p0) return;
  hipLaunchKernelGGL(k_state3d_rows, dim3((p1 - p0 + 255) / 256, c->g.km, c->g.nblocks), dim3(256), 0, c->stream, c->g,
                     (const double *)c->TR[0][c->newt], (const double *)c->TR[1][c->newt], c->RHO[c->newt], p0, p1);
}
static int phase_hmix_momentum(pop_ctx *c, hipStream_t st = nullptr) {   // del4 only: first Laplacian of the velocity into d2u
  if (c->h.c.hmix_momentum != 4) return 0;
  if (c->d2u_next_valid && c->d2u_next_slot == c->mixt) {   // formed by the previous step's momentum kernel (ghost ring already updated)
    c->d2u_next_valid = false;
    std::swap(c->d2u[0], c->d2u_next[0]); std::swap(c->d2u[1], c->d2u_next[1]);
    return 0;
  }
  c->d2u_next_valid = false;
  return mix_hdiffu_del4(c->h, c->g, step_params(c), c->mix, c->U[c->mixt], c->V[c->mixt], c->d2u[0], c->d2u[1], c->S3c, c->S3d, st ? st : c->stream, c->err);
}
static int phase_momentum_rhs(pop_ctx *c, int tj_first = 0, int tj_count = -1, bool last_piece = true) {
  MomentumRhsArgs a{};
  a.UCUR = c->U[c->curt]; a.VCUR = c->V[c->curt]; a.UOLD = c->U[c->oldt]; a.VOLD = c->V[c->oldt]; a.UMIX = c->U[c->mixt]; a.VMIX = c->V[c->mixt];
  a.RHOOLD = c->RHO[c->oldt]; a.RHOCUR = c->RHO[c->curt]; a.RHONEW = c->RHO[c->newt]; a.VVC = c->VVC; a.DHU = c->DHU;
  if (c->h.c.hmix_momentum == 4) { a.UMIX = c->d2u[0]; a.VMIX = c->d2u[1]; }   // del4: second Laplacian acts on D2U, D2V
  a.UNEW = c->U[c->newt]; a.VNEW = c->V[c->newt]; a.ZX = c->ZX; a.ZY = c->ZY;
  // 3x3 stencils staged through LDS (kernels_momentum_lds.hpp): 64x8 tiles measured -11 % (tx0.1v3) / -12 % (gx1v7)
  // against the direct-load kernel, 64x4 +8 %; POP_MOMENTUM_LDS=0|4|8 selects (read at pop_create)
  // the next step's first Laplacian of the velocity (see phase_tracer_rhs; whole-domain launches of the LDS kernel only)
  // (several ranks launch the kernel in three pieces -- interior tile rows, then the rim rows: every piece writes its tiles, the halo
  // update follows the last one)
  const bool form_next = (c->mom_lds_rows == 8 || c->mom_lds_rows == 4) && c->d2u_next[0] && !c->avg_ts &&
                         c->h.c.tmix_opt != 3 && c->uv_ghosts_ok[c->curt];
  if (form_next) { a.D2N[0] = c->d2u_next[0]; a.D2N[1] = c->d2u_next[1]; a.AMF = c->mix.D4AMF; }
  c->d2u_last_formed = form_next;
  if (c->mom_lds_rows == 8) launch_momentum_lds<8>(c->g, step_params(c), a, c->stream, tj_first, tj_count);
  else if (c->mom_lds_rows == 4) launch_momentum_lds<4>(c->g, step_params(c), a, c->stream, tj_first, tj_count);
  else if (c->g.pbc) hipLaunchKernelGGL((k_momentum_rhs<false, true>), grid_stencil(c), block_stencil(), 0, c->stream, c->g, step_params(c), a);
  else hipLaunchKernelGGL((k_momentum_rhs<false, false>), grid_stencil(c), block_stencil(), 0, c->stream, c->g, step_params(c), a);
  if (form_next && last_piece && !c->phase_timing) {
    if (halo_update_many(c, {{c->d2u_next[0], c->g.km, 1, 1}, {c->d2u_next[1], c->g.km, 1, 1}})) return 1;
    c->d2u_next_valid = true; c->d2u_next_slot = c->curt;
  }
  return 0;
}
static int phase_impvmixu(pop_ctx *c, hipStream_t st) {
  ImpvmixuArgs a{c->U[c->newt], c->V[c->newt], c->E3, c->U[c->oldt], c->V[c->oldt], c->VVC};
  launch_impvmixu(c->g, step_params(c), a, grid_cols(c), st ? st : c->stream, c->reg_thomas);
  return 0;
}

static int phase_correct(pop_ctx *c) {
  const StepParams sp = step_params(c);
  // the generic Thomas kernel stages E, F through the shared 3-D scratch (E3, F3), which a KPP look-ahead in flight on its own
  // stream also uses (E3 = the Richardson column of the generic k_kpp_interior): the corrector then follows the look-ahead.
  // The register kernels (km = 60 / 62) touch no scratch and run beside it.
  const bool reg_kernel = c->reg_thomas_t && (c->g.km == 60 || c->g.km == 62);
  if (!reg_kernel && c->ahead_valid) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_ahead, 0));
  if (sp.pavg) launch_impvmixt<1, false, true>(c->g, sp, impvmixt_args(c, c->PS[c->newt]), grid_cols(c), c->stream, c->reg_thomas_t, c->h.tun.thomas_pair);
  else launch_impvmixt<0, true, true>(c->g, sp, impvmixt_args(c, c->PS[c->newt]), grid_cols(c), c->stream, c->reg_thomas_t, c->h.tun.thomas_pair);
  return 0;
}
static int phase_add_btrop(pop_ctx *c, hipStream_t st = nullptr) {
  hipLaunchKernelGGL(k_add_barotropic, grid_3d(c), dim3(256), 0, st ? st : c->stream, c->g, c->U[c->newt], c->V[c->newt], c->UB[c->newt], c->VB[c->newt]);
  return 0;
}

int pop_baroclinic_driver(pop_ctx *c) {
  if (need_device(c)) return 1;
  ScopedPhase ph(c, "BAROCLINIC");
  if (c->vmixu_deferred) { c->vmixu_deferred = false; if (phase_impvmixu(c, nullptr)) return 1; }   // a driver call that was never followed by its correct_adjust
  const StepParams sp = step_params(c);
  const bool fork = c->side_del4;
  if (fork) {   // del4 first Laplacians beside the vertical-mixing coefficients
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
    if (c->h.c.hmix_tracer != 3 && phase_hmix_tracer(c, c->side)) return 1;
    HIPCHK(c, hipEventRecord(c->ev_d2t, c->side));
    if (phase_hmix_momentum(c, c->side)) return 1;
    HIPCHK(c, hipEventRecord(c->ev_d2u, c->side));
  }
  if (phase_vmix(c)) return 1;
  if (fork) {
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_d2t, 0));
    if (c->h.c.hmix_tracer == 3 && phase_hmix_tracer(c)) return 1;   // Gent-McWilliams reads and adds to the coefficients vmix just formed
  } else if (phase_hmix_tracer(c)) return 1;
  const bool fwd = sp.pavg && tracer_fwd_fused(c);
  if (phase_tracer_rhs(c, fwd)) return 1;
  // several ranks: the exchange of the new tracers' ghost rows runs on the communication stream while the launch stream
  // forms the density and the momentum right-hand side of every tile that reads no ghost row of another rank; the first
  // and last tile rows follow once the rows have arrived (same kernels on disjoint tiles: bitwise the serial result)
  const int mrows = c->mom_lds_rows;
  const bool overlap = sp.pavg && halo_async_ok(c) && (mrows == 4 || mrows == 8) && !c->h.c.ns_boundary;
  const int mtiles_j = overlap ? (c->g.nyb - 2 * NGHOST + mrows - 1) / mrows : 0;
  if (sp.pavg) {
    if (fwd ? phase_impvmixt_back(c) : phase_impvmixt_pred(c)) return 1;
    if (overlap && mtiles_j >= 3) {
      HaloAsync HA;
      if (halo_many_begin(c, {{c->TR[0][c->newt], c->g.km}, {c->TR[1][c->newt], c->g.km}}, HA)) return 1;
      state_new_rows(c, NGHOST, c->g.nyb - NGHOST);                      // physical rows: own cells + ghosts copied inside the rank
      if (fork) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_d2u, 0));
      else if (phase_hmix_momentum(c)) return 1;
      if (phase_momentum_rhs(c, 1, mtiles_j - 2, false)) return 1;      // interior tile rows
      if (halo_many_end(c, HA)) return 1;
      state_new_rows(c, 0, NGHOST); state_new_rows(c, c->g.nyb - NGHOST, c->g.nyb);   // ghost rows
      if (phase_momentum_rhs(c, 0, 1, false) || phase_momentum_rhs(c, mtiles_j - 1, 1, true)) return 1;   // rim tile rows
    } else {
      if (halo_update_many(c, {{c->TR[0][c->newt], c->g.km}, {c->TR[1][c->newt], c->g.km}})) return 1;
      if (phase_state_new(c)) return 1;
      if (fork) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_d2u, 0));
      else if (phase_hmix_momentum(c)) return 1;
      if (phase_momentum_rhs(c)) return 1;
    }
  } else {
    if (fork) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_d2u, 0));
    else if (phase_hmix_momentum(c)) return 1;
    if (phase_momentum_rhs(c)) return 1;
  }
  // the implicit vertical mixing of U, V is not needed before the step tail: with the register kernel (no shared scratch)
  // it runs on the side stream beside the barotropic solver, whose one-workgroup reduction kernels leave the GPU idle
  // bandwidth-bound grids: held back until the barotropic solve has finished and launched then with the barotropic velocity added on
  // the way out (k_impvmixu_reg<., ., true>): the separate k_add_barotropic pass over U, V(new) is gone.  Nothing between here and
  // baroclinic_correct_adjust reads U, V(new); a caller that does (any field access: join_side) gets the plain kernel first.
  // Not across a tripole fold (the sum follows the halo update there).  POP_VMIXU_DEFER=0|1 overrides the size rule.
  const int defer_env = tun_or(c->h.tun.vmixu_defer, -1);
  const bool defer = c->side && impvmixu_add_available(c->g, c->reg_thomas) && c->h.c.ns_boundary != 2 && !tun_on(c->h.tun.btrop_inline) &&
                     !tun_on(c->h.tun.vmixu_inline) && (defer_env >= 0 ? defer_env != 0 : (long long)c->g.n2 * c->g.nblocks > (1 << 19));
  if (defer) c->vmixu_deferred = true;
  else if (c->side && c->reg_thomas && (c->g.km == 60 || c->g.km == 62) && !tun_on(c->h.tun.vmixu_inline)) {
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
    if (phase_impvmixu(c, c->side)) return 1;
    HIPCHK(c, hipEventRecord(c->ev_vmixu, c->side));
    c->vmixu_pending = true;
  } else if (phase_impvmixu(c, nullptr)) return 1;
  HIPCHK(c, hipGetLastError());
  return 0;
}

// which form of the solver pop_solver_run dispatches to (same tests, same order): 1 operation by operation, 2 fused on one rank,
// 3 fused with the blocks spread over ranks, 4 replicated fused solve on every rank -- reported per rank by bench.py
static int solver_path_code(const pop_ctx *c) {
  const bool unf = tun_on(c->h.tun.solver_unfused);
  if (c->h.c.solver_choice == 2) {
    if (c->fused_ok && !c->use_evp) return 2;
    if (c->h.nranks > 1 && c->max_blocks_per_rank <= 16 && !c->use_evp && !unf) return 3;
    return 1;
  }
  if (c->h.c.solver_choice == 3) {
    if (c->use_evp) return c->evp_fused_ok ? 2 : 1;
    if (c->fused_ok) return 2;
    if (c->h.nranks > 1 && !unf) return 3;
    return 1;
  }
  if (c->use_evp) return 1;
  if (c->replicated) return 4;
  if (c->fused_ok) return 2;
  if (c->h.nranks > 1 && c->max_blocks_per_rank <= 16 && !unf) return 3;
  return 1;
}
int pop_solver_run(pop_ctx *c) {
  if (need_device(c)) return 1;
  if (c->h.c.solver_choice == 2) {
    if (c->fused_ok && !c->use_evp) return solver_chrongear_fused(c);
    if (c->h.nranks > 1 && c->max_blocks_per_rank <= 16 && !c->use_evp && !tun_on(c->h.tun.solver_unfused)) return solver_chrongear_fused_dist(c);
    return solver_chrongear(c);
  }
  if (c->h.c.solver_choice == 3) {
    if (c->use_evp && c->evp_fused_ok) return solver_pcsi_fused(c);
    if (c->use_evp) return solver_pcsi(c);
    if (c->fused_ok) return solver_pcsi_fused(c);
    if (c->h.nranks > 1 && !tun_on(c->h.tun.solver_unfused)) return solver_pcsi_fused_dist(c);
    return solver_pcsi(c);
  }
  if (c->use_evp) return solver_pcg(c);
  if (c->replicated) {
    if (!c->allred || !c->redbuf || c->red_doubles < 2LL * c->g.n2 * c->h.nblocks_tot) { c->err = "replicated solve needs pop_set_comm with a reduce buffer of pop_reduce_buffer_doubles()"; return 1; }
    return solver_pcg_replicated(c);
  }
  if (c->fused_ok) { SolveView v = fused_view(c); const int e = solver_pcg_fused(c, v); c->S0 = v.S0; c->S1 = v.S1; return e; }
  if (c->h.nranks > 1 && c->max_blocks_per_rank <= 16 && !tun_on(c->h.tun.solver_unfused)) return solver_pcg_fused_dist(c);
  return solver_pcg(c);
}
int pop_solver_preconditioner(pop_ctx *c, const char *x_name, int x_tl, const char *px_name, int px_tl) {
  if (need_device(c)) return 1;
  if (join_side(c)) return 1;
  double *x, *px; long long cnt, a2 = (long long)c->g.n2 * c->g.nblocks;
  if (resolve(c, x_name, x_tl, 0, &x, &cnt) || cnt != a2) { c->err = std::string("unknown 2-D field ") + x_name; return 1; }
  if (resolve(c, px_name, px_tl, 0, &px, &cnt) || cnt != a2 || px == x) { c->err = std::string("unknown 2-D field ") + px_name; return 1; }
  HIPCHK(c, hipMemsetAsync(px, 0, sizeof(double) * a2, c->stream));
  if (c->use_evp) return evp_apply(c, x, px, false);   // (any field the caller names: no assumption about its land values)
  HIPCHK(c, hipMemcpyAsync(px, x, sizeof(double) * a2, hipMemcpyDeviceToDevice, c->stream));
  hipLaunchKernelGGL(k_pcsi_precond, dim3((unsigned)((a2 + 255) / 256)), dim3(256), 0, c->stream, c->g, px, (const double *)c->centerWgt, a2);
  HIPCHK(c, hipGetLastError());
  return 0;
}
// grad / div / zcurl of operators.F90 on device fields.  A 3-D field name selects its level-k slab.
int pop_operator(pop_ctx *c, int op, int k, const char *a_name, const char *b_name, int tl, const char *o1_name, const char *o2_name) {
  if (need_device(c)) return 1;
  if (join_side(c)) return 1;
  if (op < 0 || op > 2 || k < 1 || k > c->g.km) { c->err = "pop_operator: op 0 grad, 1 div, 2 zcurl; 1 <= k <= km"; return 1; }
  const long long a2 = (long long)c->g.n2 * c->g.nblocks, a3 = (long long)c->g.n3 * c->g.nblocks;
  auto slab = [&](const char *name, int t, double **p, long long *stride) -> int {
    long long cnt;
    if (!name || resolve(c, name, t, 0, p, &cnt) || (cnt != a2 && cnt != a3)) { c->err = std::string("pop_operator: unknown field ") + (name ? name : "(null)"); return 1; }
    *stride = cnt == a3 ? c->g.n3 : c->g.n2;
    if (cnt == a3) *p += (long long)(k - 1) * c->g.n2;
    return 0;
  };
  double *A, *B, *O1, *O2 = nullptr; long long sa, sb, so1, so2 = 0;
  if (slab(a_name, tl, &A, &sa)) return 1;
  B = A; sb = sa;
  if (op != 0 && slab(b_name, tl, &B, &sb)) return 1;
  if (slab(o1_name, tl, &O1, &so1)) return 1;
  if (op == 0 && slab(o2_name, tl, &O2, &so2)) return 1;
  if (sb != sa || (op == 0 && so2 != so1)) { c->err = "pop_operator: the two inputs (outputs) must have the same rank"; return 1; }
  hipLaunchKernelGGL(k_operator, dim3((c->g.n2 + 255) / 256, c->g.nblocks), dim3(256), 0, c->stream, c->g, op, k,
                     (const double *)A, (const double *)B, O1, O2 ? O2 : O1, sa, so1, 0);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int pop_operator_host(pop_ctx *c, int op, int k, int block_local, const double *a, const double *b, double *o1, double *o2) {
  if (need_device(c)) return 1;
  if (join_side(c)) return 1;
  if (op < 0 || op > 2 || k < 1 || k > c->g.km) { c->err = "pop_operator_host: op 0 grad, 1 div, 2 zcurl; 1 <= k <= km"; return 1; }
  if (block_local < 1 || block_local > c->g.nblocks) { c->err = "pop_operator_host: block_local is this_block%local_id, 1 .. nblocks"; return 1; }
  if (!a || !o1 || (op != 0 && !b) || (op == 0 && !o2)) { c->err = "pop_operator_host: grad(F -> GRADX, GRADY), div / zcurl(UX, UY -> one field)"; return 1; }
  const size_t n2 = (size_t)c->g.n2;
  if (!c->op_scratch && dev_alloc(c, &c->op_scratch, 4 * n2)) return 1;
  double *A = c->op_scratch, *B = A + n2, *O1 = B + n2, *O2 = O1 + n2;
  HIPCHK(c, hipMemcpyAsync(A, a, n2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (op != 0) HIPCHK(c, hipMemcpyAsync(B, b, n2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_operator, dim3((c->g.n2 + 255) / 256, 1), dim3(256), 0, c->stream, c->g, op, k,
                     (const double *)A, (const double *)(op != 0 ? B : A), O1, O2, 0LL, 0LL, block_local - 1);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(o1, O1, n2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (op == 0) HIPCHK(c, hipMemcpyAsync(o2, O2, n2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int pop_solver_get_diagnostics(const pop_ctx *c, int *it, double *rms) {
  if (it) *it = c->numIterations;
  if (rms) *rms = c->rmsResidual;
  return 0;
}

static int barotropic_driver(pop_ctx *c, bool update_zx_zy);
int pop_barotropic_driver(pop_ctx *c) { return barotropic_driver(c, true); }
// barotropic.F90:267 on its own: the caller has already updated the halos of ZX, ZY (step_mod.F90:405-423).  Beyond a tripole
// fold a second update would not be a no-op (the symmetrised top row takes each sign from the partner point).
int pop_barotropic_driver_updated(pop_ctx *c) { return barotropic_driver(c, false); }
static int barotropic_driver(pop_ctx *c, bool update_zx_zy) {
  if (need_device(c)) return 1;
  ScopedPhase ph(c, "BAROTROPIC");
  const StepParams sp = step_params(c);
  if (update_zx_zy && halo_update_many(c, {{c->ZX, 1, 1, 1}, {c->ZY, 1, 1, 1}})) return 1;   // NE corner, vector (step_mod.F90:405-423)
  BtropArgs a{};
  a.ZX = c->ZX; a.ZY = c->ZY; a.GXC = c->GX[c->curt]; a.GXO = c->GX[c->oldt]; a.GYC = c->GY[c->curt]; a.GYO = c->GY[c->oldt];
  a.UBO = c->UB[c->oldt]; a.VBO = c->VB[c->oldt]; a.PCUR = c->PS[c->curt]; a.FW = c->FW; a.PGUESS = c->PGUESS;
  a.UH = c->UH; a.VH = c->VH; a.W3 = c->W3; a.W4 = c->W4; a.RHS = c->RHS; a.centerWgt = c->centerWgt; a.PNEW = c->PS[c->newt];
  a.GXN = c->GX[c->newt]; a.GYN = c->GY[c->newt]; a.UBN = c->UB[c->newt]; a.VBN = c->VB[c->newt];
  a.GXR = c->leapfrogts ? c->GX[c->oldt] : c->GX[c->curt]; a.GYR = c->leapfrogts ? c->GY[c->oldt] : c->GY[c->curt];
  a.scal = &c->sc->xcheck; a.rcheck = c->h.rcheck; a.rconst = c->h.rconst;
  const dim3 G(col_grid(c->g, 256), c->g.nblocks), B(256);
  hipLaunchKernelGGL(k_btrop_rhs1, G, B, 0, c->stream, c->g, sp, a);
  hipLaunchKernelGGL(k_btrop_rhs2, G, B, 0, c->stream, c->g, sp, a);
  if (halo_update(c, c->RHS, 1)) return 1;
  solve_collect(c);
  if (!c->ev_solve[0]) { HIPCHK(c, hipEventCreate(&c->ev_solve[0])); HIPCHK(c, hipEventCreate(&c->ev_solve[1])); }
  HIPCHK(c, hipEventRecord(c->ev_solve[0], c->stream));
  const int e = pop_solver_run(c);
  if (e) return e;
  HIPCHK(c, hipEventRecord(c->ev_solve[1], c->stream));
  c->solve_pending = true; c->solve_iters_pending = c->numIterations;
  hipLaunchKernelGGL(k_dot_partial, grid_2d(c), dim3(POP_RED_THREADS), 0, c->stream, c->g, c->PS[c->newt], c->g.CHECKER, (const double *)nullptr, c->partial);
  if (reduce_finish<1>(c, FIN_XCHECK)) return 1;
  hipLaunchKernelGGL(k_btrop_fin1, G, B, 0, c->stream, c->g, a);
  hipLaunchKernelGGL(k_btrop_fin2, G, B, 0, c->stream, c->g, sp, a);
  if (halo_update_many(c, {{c->PS[c->newt], 1}, {c->GX[c->newt], 1, 1, 1}, {c->GY[c->newt], 1, 1, 1}})) return 1;   // barotropic.F90:699-729
  HIPCHK(c, hipGetLastError());
  return 0;
}

int pop_baroclinic_correct_adjust(pop_ctx *c) {
  if (need_device(c)) return 1;
  ScopedPhase ph(c, "CORRECT_ADJUST");
  // the barotropic velocity is added to U, V(new) (step_mod.F90:572-600) on the side stream while the tracer corrector runs:
  // same sum at every cell; the halo update of U, V in the step tail then carries it to the ghost cells.  Not beyond a
  // tripole fold: there the update symmetrises |U| of the degenerate top row, which does not commute with the sum, so the
  // reference's order (halo updates first, step_mod.F90:467-513, then the sum over whole blocks) is kept.
  if (c->vmixu_deferred) {   // implicit vertical mixing of U, V and the sum in one launch (see pop_baroclinic_driver)
    c->vmixu_deferred = false;
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
    ImpvmixuArgs a{c->U[c->newt], c->V[c->newt], c->E3, c->U[c->oldt], c->V[c->oldt], c->VVC, c->UB[c->newt], c->VB[c->newt]};
    launch_impvmixu_add(c->g, step_params(c), a, grid_cols(c), c->side);
    HIPCHK(c, hipEventRecord(c->ev_vmixu, c->side));
    c->vmixu_pending = true; c->btrop_added = true;
  } else if (c->side && !tun_on(c->h.tun.btrop_inline) && c->h.c.ns_boundary != 2) {
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
    if (phase_add_btrop(c, c->side)) return 1;
    HIPCHK(c, hipEventRecord(c->ev_vmixu, c->side));
    c->vmixu_pending = true; c->btrop_added = true;
  }
  if (phase_correct(c)) return 1;
  HIPCHK(c, hipGetLastError());
  return 0;
}

// b4b global sum of a device array (physical domain) times an optional mask; result on the host
static int global_sum_dev(pop_ctx *c, const double *p, const double *mask, double *result) {
  hipLaunchKernelGGL(k_dot_partial, grid_2d(c), dim3(POP_RED_THREADS), 0, c->stream, c->g, p, (const double *)nullptr, mask, c->partial);
  if (reduce_finish<1>(c, FIN_PLAIN)) return 1;
  SolverScalars s;
  if (read_scalars(c, &s)) return 1;
  *result = s.sum0;
  return 0;
}

// step_RF (step_mod.F90:919-1350): Robert-Asselin-Williams filter of curtime (and newtime) with the
// volume-conserving adjustment of PSURF and the tracers, then the leapfrog index rotation
static int step_rf(pop_ctx *c) {
  const HostModel &h = c->h;
  const int o = c->oldt, cu = c->curt, nw = c->newt, nt = h.nt;
  const long long a2 = (long long)c->g.n2 * c->g.nblocks, a3 = (long long)c->g.n3 * c->g.nblocks;
  RfParams p{h.robert_newtime, h.robert_curtime, h.rf_nonzero_newtime, h.dz[1], GRAV};
  auto filt = [&](double *const F[3], long long n) {
    hipLaunchKernelGGL(k_rf_filter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, (const double *)F[o], F[cu], F[nw], p);
  };
  filt(c->UB, a2); filt(c->VB, a2); filt(c->GX, a2); filt(c->GY, a2); filt(c->U, a3); filt(c->V, a3);
  double *WORKN[2] = {c->W3, c->W4}, *WB = c->UH;          // 2-D work arrays free at this point of the step
  RfTracerArgs ta{};
  RfSurfArgs sa{};
  for (int n = 0; n < 2; ++n) {
    ta.TO[n] = c->TR[n][o]; ta.TC[n] = c->TR[n][cu]; ta.TN[n] = c->TR[n][nw]; ta.WORKN[n] = WORKN[n];
    sa.TO[n] = c->TR[n][o]; sa.TC[n] = c->TR[n][cu]; sa.TN[n] = c->TR[n][nw]; sa.WORKN[n] = WORKN[n];
  }
  sa.PO = c->PS[o]; sa.PC = c->PS[cu]; sa.PN = c->PS[nw];
  const dim3 GC = grid_cols(c);
  hipLaunchKernelGGL(k_rf_tracer_interior, dim3(GC.x, GC.y, nt), dim3(POP_COL_THREADS), 0, c->stream, c->g, p, ta);
  double svol[MAXNT] = {};
  for (int n = 0; n < nt; ++n) if (global_sum_dev(c, WORKN[n], nullptr, &svol[n])) return 1;
  hipLaunchKernelGGL(k_rf_surface, dim3((unsigned)((a2 + 255) / 256), nt), dim3(256), 0, c->stream, c->g, p, sa);
  for (int n = 0; n < nt; ++n) { double s1; if (global_sum_dev(c, WORKN[n], nullptr, &s1)) return 1; svol[n] = svol[n] + s1; }
  hipLaunchKernelGGL(k_rf_psurf, dim3((unsigned)((a2 + 255) / 256)), dim3(256), 0, c->stream, c->g, p, (const double *)c->PS[o], c->PS[cu], c->PS[nw], WB);
  double rf_sump;
  if (global_sum_dev(c, WB, c->g.CONSTNT, &rf_sump)) return 1;      // MASK_TRBUDGET(:,:,1) = (KMT >= 1) = CONSTNT
  rf_sump = rf_sump / h.bgtarea_t_1;
  hipLaunchKernelGGL(k_rf_psurf_adjust, dim3((unsigned)((a2 + 255) / 256)), dim3(256), 0, c->stream, c->g, p, rf_sump, c->PS[cu], c->PS[nw],
                     c->TR[0][cu], c->TR[0][nw], c->TR[1][cu], c->TR[1][nw], WB);
  double vsurf, vsurf_oo;
  if (global_sum_dev(c, WB, c->g.CONSTNT, &vsurf) || global_sum_dev(c, WB, c->g.RCALCT, &vsurf_oo)) return 1;
  const double rf_ocean_norm = h.open_ocean_volume_2_km + vsurf_oo;   // fully coupled normalisation (:1166-1172)
  (void)vsurf;
  for (int n = 0; n < nt; ++n) {
    c->rf_S[n] = svol[n] / rf_ocean_norm;
    const double factor = (!c->rf_S_prev_valid[n] || h.rf_nonzero_newtime) ? c->rf_S[n] : 0.5 * (c->rf_S[n] + c->rf_S_prev[n]);
    hipLaunchKernelGGL(k_rf_conserve, grid_3d(c), dim3(256), 0, c->stream, c->g, p, factor * h.robert_newtime, factor * h.robert_curtime,
                       c->TR[n][cu], c->TR[n][nw]);
  }
  HIPCHK(c, hipMemcpyAsync(c->FW_OLD, c->FW, a2 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  launch_state3d(c->g, (const double *)c->TR[0][cu], (const double *)c->TR[1][cu], c->RHO[cu], c->stream);
  launch_state3d(c->g, (const double *)c->TR[0][nw], (const double *)c->TR[1][nw], c->RHO[nw], c->stream);
  hipLaunchKernelGGL(k_pguess, dim3((unsigned)((a2 + 255) / 256)), dim3(256), 0, c->stream, a2, c->PGUESS, c->PS[nw], c->PS[cu], c->PS[o]);
  c->oldt = cu; c->curt = nw; c->newt = o;                              // step_mod.F90:1318-1322
  if (!h.rf_nonzero_newtime) for (int n = 0; n < nt; ++n) { c->rf_S_prev[n] = c->rf_S[n]; c->rf_S_prev_valid[n] = true; }
  HIPCHK(c, hipGetLastError());
  return 0;
}

int pop_step_tail(pop_ctx *c) {
  if (need_device(c)) return 1;
  ScopedPhase ph(c, "3D-UPDATE");
  const int km = c->g.km;
  if (join_side(c, true)) return 1;
  {   // the seven updates of step_mod.F90:467-560 as one message per neighbour
    std::vector<HaloItem> items = {{c->UB[c->newt], 1, 1, 1}, {c->VB[c->newt], 1, 1, 1}, {c->U[c->newt], km, 1, 1}, {c->V[c->newt], km, 1, 1}, {c->RHO[c->newt], km}};
    for (int n = 0; n < c->h.nt; ++n) items.push_back({c->TR[n][c->newt], km});
    if (halo_update_many(c, items)) return 1;
    c->tr_ghosts_ok[c->newt] = true; c->uv_ghosts_ok[c->newt] = true;
  }
  if (!c->btrop_added && phase_add_btrop(c)) return 1;
  c->btrop_added = false;
  const long long a2 = (long long)c->g.n2 * c->g.nblocks;
  hipLaunchKernelGGL(k_pguess, dim3((a2 + 255) / 256), dim3(256), 0, c->stream, a2, c->PGUESS, c->PS[c->newt], c->PS[c->curt], c->PS[c->oldt]);
  if (c->avg_ts) {
    const int o = c->oldt, cu = c->curt, nw = c->newt;
    Avg2dArgs a{};
    a.UBO = c->UB[o]; a.UBC = c->UB[cu]; a.VBO = c->VB[o]; a.VBC = c->VB[cu]; a.GXO = c->GX[o]; a.GXC = c->GX[cu]; a.GYO = c->GY[o]; a.GYC = c->GY[cu];
    a.PO = c->PS[o]; a.PC = c->PS[cu]; a.PG = c->PGUESS; a.FW_OLD = c->FW_OLD;
    a.UBN = c->UB[nw]; a.VBN = c->VB[nw]; a.GXN = c->GX[nw]; a.GYN = c->GY[nw]; a.PN = c->PS[nw]; a.FW = c->FW;
    for (int n = 0; n < 2; ++n) { a.T1O[n] = c->TR[n][o]; a.T1C[n] = c->TR[n][cu]; a.T1N[n] = c->TR[n][nw]; }
    a.dz1 = c->h.dz[1]; a.grav = GRAV;
    hipLaunchKernelGGL(k_avg2d, dim3(col_grid(c->g, 256), c->g.nblocks), dim3(256), 0, c->stream, c->g, a);
    Avg3dArgs b{};
    b.UO = c->U[o]; b.UC = c->U[cu]; b.VO = c->V[o]; b.VC = c->V[cu]; b.RO = c->RHO[o]; b.RC = c->RHO[cu]; b.UN = c->U[nw]; b.VN = c->V[nw];
    for (int n = 0; n < 2; ++n) { b.TO[n] = c->TR[n][o]; b.TC[n] = c->TR[n][cu]; b.TN[n] = c->TR[n][nw]; }
    hipLaunchKernelGGL(k_avg3d, grid_3d(c), dim3(256), 0, c->stream, c->g, b);
  } else if (c->h.c.tmix_opt == 3) {                                   // step_mod.F90:798-802
    if (step_rf(c)) return 1;
  } else {
    HIPCHK(c, hipMemcpyAsync(c->FW_OLD, c->FW, a2 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    const int tmp = c->oldt; c->oldt = c->curt; c->curt = c->newt; c->newt = tmp;   // step_mod.F90:827-830
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}

int pop_step(pop_ctx *c) {
  if (need_device(c)) return 1;
  if (c->h.c.ns_boundary == 2 && !c->grid_from_input) { c->err = "time stepping on a tripole decomposition needs the caller's grid (pop_create_with_grid): the internal lat-lon grid has no values beyond the fold"; return 1; }
  ScopedPhase ph(c, "STEP");
  int e;
  if ((e = pop_time_manager(c)) || (e = pop_dhdt(c)) || (e = pop_baroclinic_driver(c)) || (e = kpp_look_ahead(c)) ||
      (e = pop_barotropic_driver(c)) || (e = pop_baroclinic_correct_adjust(c)) || (e = pop_step_tail(c))) return e;
  return 0;
}

int pop_halo_update(pop_ctx *c, const char *name, int tl, int n) {
  if (need_device(c)) return 1;
  if (join_side(c)) return 1;
  double *p; long long cnt;
  if (n < 0) {   // POP_HaloUpdate4DR8 (mpi/POP_HaloMod.F90:4122-4585): every tracer of a (nx,ny,km,nt,block) field
    if (std::string(name) != "TRACER" && std::string(name) != "KPP_SRC" && std::string(name) != "STF" && std::string(name) != "TFW") {
      c->err = std::string("pop_halo_update: field has no tracer dimension: ") + name; return 1;
    }
    for (int m = 0; m < c->h.nt; ++m) if (pop_halo_update(c, name, tl, m)) return 1;
    return 0;
  }
  if (resolve(c, name, tl, n, &p, &cnt)) { c->err = std::string("unknown field ") + name; return 1; }
  const int nz = (int)(cnt / ((long long)c->g.n2 * c->g.nblocks));
  return halo_update(c, p, nz);
}
int pop_halo_update_loc(pop_ctx *c, const char *name, int tl, int n, int field_loc, int field_kind) {
  if (need_device(c)) return 1;
  if (join_side(c)) return 1;
  if (field_loc < 0 || field_loc > 3 || field_kind < 0 || field_kind > 2) { c->err = "pop_halo_update_loc: unknown field location / kind"; return 1; }
  double *p; long long cnt;
  if (resolve(c, name, tl, n < 0 ? 0 : n, &p, &cnt)) { c->err = std::string("unknown field ") + name; return 1; }
  const int nz = (int)(cnt / ((long long)c->g.n2 * c->g.nblocks));
  return halo_update(c, p, nz, 0.0, field_loc, field_kind);
}
// host array (nx_block, ny_block, nz, local blocks) staged through a device work field: the update then runs through
// the same plan, kernels and transport as a device-resident field, so it also serves decompositions over several ranks
static int halo_host_staged(pop_ctx *c, double *array, int nz, double fill, int field_loc, int field_kind) {
  if (need_device(c) || join_side(c)) return 1;
  if (nz < 1 || nz > c->g.km) { c->err = "host halo update over several ranks: 1 <= nz <= km (update a 4-D field tracer by tracer)"; return 1; }
  const size_t cnt = (size_t)c->g.n2 * nz * c->g.nblocks;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->S3a, array, cnt * sizeof(double), hipMemcpyHostToDevice));
  if (halo_update(c, c->S3a, nz, fill, field_loc, field_kind)) return 1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(array, c->S3a, cnt * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}
int pop_halo_update_host_r8_loc(pop_ctx *c, double *array, int nz, double fill, int field_loc, int field_kind) {
  if (field_loc < 0 || field_loc > 3 || field_kind < 0 || field_kind > 2) { c->err = "unknown field location / kind"; return 1; }
  if (c->h.nranks != 1) return halo_host_staged(c, array, nz, fill, field_loc, field_kind);
  host_halo_r8_loc(c->h, array, nz, fill, field_loc, field_kind);
  return 0;
}
// POP_GlobalSum(array, dist, fieldLoc, errorCode, mMask) on HOST arrays of the local blocks (mpi/POP_ReductionsMod.F90:144-389):
// array and the optional multiplicative mask are staged into 2-D device work fields and summed by the same b4b kernels
// as a device-resident field (field_loc as in pop_global_sum_loc)
int pop_global_sum_host(pop_ctx *c, const double *array, const double *mask, int field_loc, double *result) {
  if (need_device(c) || join_side(c)) return 1;
  if (!array || !result) { c->err = "pop_global_sum_host: null argument"; return 1; }
  const size_t a2 = (size_t)c->g.n2 * c->g.nblocks;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->W3, array, a2 * sizeof(double), hipMemcpyHostToDevice));
  if (mask) HIPCHK(c, hipMemcpy(c->W4, mask, a2 * sizeof(double), hipMemcpyHostToDevice));
  const double *mk = mask ? c->W4 : nullptr;
  if (c->h.c.ns_boundary == 2 && (field_loc == 1 || field_loc == 2)) {
    hipLaunchKernelGGL(k_dot_partial_dup, grid_2d(c), dim3(POP_RED_THREADS), 0, c->stream, c->g, (const double *)c->W3, mk, (const double *)c->d2["TRIPOLE_DUP"], c->partial);
    if (reduce_finish<2>(c, FIN_TRIPOLE)) return 1;
  } else {
    hipLaunchKernelGGL(k_dot_partial, grid_2d(c), dim3(POP_RED_THREADS), 0, c->stream, c->g, (const double *)c->W3, (const double *)nullptr, mk, c->partial);
    if (reduce_finish<1>(c, FIN_PLAIN)) return 1;
  }
  SolverScalars sres;
  if (read_scalars(c, &sres)) return 1;
  *result = sres.sum0;
  return 0;
}
int pop_halo_update_host_i4_loc(pop_ctx *c, int *array, int nz, int fill, int field_loc, int field_kind) {
  if (c->h.nranks != 1) { c->err = "host halo update needs all blocks on one rank"; return 1; }
  if (field_loc < 0 || field_loc > 3 || field_kind < 0 || field_kind > 2) { c->err = "unknown field location / kind"; return 1; }
  host_halo_i4_loc(c->h, array, nz, fill, field_loc, field_kind);
  return 0;
}
// host-array halo: valid for single-rank decompositions (all blocks local), used at init time
int pop_halo_update_host_r8(pop_ctx *c, double *array, int nz, double fill) {
  if (c->h.nranks != 1) { c->err = "host halo update needs all blocks on one rank"; return 1; }
  host_halo_r8(c->h, array, nz, fill);
  return 0;
}
int pop_halo_update_host_i4(pop_ctx *c, int *array, int nz, int fill) {
  if (c->h.nranks != 1) { c->err = "host halo update needs all blocks on one rank"; return 1; }
  host_halo_i4(c->h, array, nz, fill);
  return 0;
}
int pop_global_sum(pop_ctx *c, const char *name, int tl, int n, const char *mask_name, double *result) {
  if (need_device(c)) return 1;
  if (join_side(c)) return 1;
  double *p, *mk = nullptr; long long cnt;
  if (resolve(c, name, tl, n, &p, &cnt)) { c->err = std::string("unknown field ") + name; return 1; }
  if (mask_name && resolve(c, mask_name, 0, 0, &mk, &cnt)) { c->err = std::string("unknown mask ") + mask_name; return 1; }
  hipLaunchKernelGGL(k_dot_partial, grid_2d(c), dim3(POP_RED_THREADS), 0, c->stream, c->g, p, (const double *)nullptr, mk, c->partial);
  if (reduce_finish<1>(c, FIN_PLAIN)) return 1;
  SolverScalars s;
  if (read_scalars(c, &s)) return 1;
  *result = s.sum0;
  return 0;
}
// POP_GlobalSum with fieldLoc on a tripole grid (mpi/POP_ReductionsMod.F90:308-341): N-face / NE-corner fields count
// the redundant half of the top row once
int pop_global_sum_loc(pop_ctx *c, const char *name, int tl, int n, const char *mask_name, int field_loc, double *result) {
  if (need_device(c)) return 1;
  if (join_side(c)) return 1;
  if (c->h.c.ns_boundary != 2 || (field_loc != 1 && field_loc != 2)) return pop_global_sum(c, name, tl, n, mask_name, result);
  double *p, *mk = nullptr; long long cnt;
  if (resolve(c, name, tl, n, &p, &cnt)) { c->err = std::string("unknown field ") + name; return 1; }
  if (mask_name && resolve(c, mask_name, 0, 0, &mk, &cnt)) { c->err = std::string("unknown mask ") + mask_name; return 1; }
  hipLaunchKernelGGL(k_dot_partial_dup, grid_2d(c), dim3(POP_RED_THREADS), 0, c->stream, c->g, (const double *)p, (const double *)mk,
                     (const double *)c->d2["TRIPOLE_DUP"], c->partial);
  if (reduce_finish<2>(c, FIN_TRIPOLE)) return 1;
  SolverScalars s;
  if (read_scalars(c, &s)) return 1;
  *result = s.sum0;
  return 0;
}
// every rank's `nv` values side by side (slot vector + sum all-reduce; exact because the other slots are zero)
static int gather_slots(pop_ctx *c, const double *local, int nv, std::vector<double> &all) {
  const int nr = c->h.nranks;
  all.assign((size_t)nv * nr, 0.0);
  if (nr == 1) { std::copy(local, local + nv, all.begin()); return 0; }
  if (!c->allred || !c->redbuf || c->red_doubles < (long long)nv * nr) { c->err = "global reduction: multi-rank run without a transport"; return 1; }
  std::copy(local, local + nv, all.begin() + (size_t)nv * c->h.rank);
  HIPCHK(c, hipMemcpyAsync(c->redbuf, all.data(), sizeof(double) * all.size(), hipMemcpyHostToDevice, c->stream));
  if (c->allred(c->comm_user, 0, (long long)all.size())) { c->err = "global reduction: allreduce failed" + tr_err(c); return 1; }
  HIPCHK(c, hipMemcpyAsync(all.data(), c->redbuf, sizeof(double) * all.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
// POP_GlobalMaxval/Minval (:2670-3223) and Maxloc/Minloc (:4002-4400): value, and the global (i,j) of the first cell
// (block order, then j, then i) that attains it; mask_name selects cells with a non-zero mask value
int pop_global_extreme(pop_ctx *c, const char *name, int tl, int n, const char *mask_name, int want_max, double *value, int *iloc, int *jloc) {
  if (need_device(c)) return 1;
  if (join_side(c)) return 1;
  double *p, *mk = nullptr; long long cnt;
  if (resolve(c, name, tl, n, &p, &cnt)) { c->err = std::string("unknown field ") + name; return 1; }
  if (mask_name && resolve(c, mask_name, 0, 0, &mk, &cnt)) { c->err = std::string("unknown mask ") + mask_name; return 1; }
  const dim3 G = grid_2d(c);
  if (want_max) hipLaunchKernelGGL(k_extreme_partial<true>, G, dim3(POP_RED_THREADS), 0, c->stream, c->g, (const double *)p, (const double *)mk, c->partial);
  else hipLaunchKernelGGL(k_extreme_partial<false>, G, dim3(POP_RED_THREADS), 0, c->stream, c->g, (const double *)p, (const double *)mk, c->partial);
  const size_t np = (size_t)G.x * G.y;
  std::vector<double> part(2 * np);
  HIPCHK(c, hipMemcpyAsync(part.data(), c->partial, sizeof(double) * part.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double best = 0.0, bidx = -1.0;
  for (size_t s = 0; s < np; ++s) {
    const double v = part[2 * s], ix = part[2 * s + 1];
    if (ix < 0.0) continue;
    if (bidx < 0.0 || (want_max ? v > best : v < best) || (v == best && ix < bidx)) { best = v; bidx = ix; }
  }
  // local cell -> global block id, global (i,j)
  double loc3[4] = {0.0, -1.0, 0.0, 0.0};   // value, global block id (ordering key), iGlobal, jGlobal
  if (bidx >= 0.0) {
    const long long q = (long long)bidx;
    const int lb = (int)(q / c->g.n2), p2 = (int)(q % c->g.n2), gb = c->h.local_ids[lb] - 1;
    const BlockInfo &B = c->h.all_blocks[gb];
    loc3[0] = best; loc3[1] = gb; loc3[2] = B.i_glob[p2 % c->g.nxb]; loc3[3] = B.j_glob[p2 / c->g.nxb];
  }
  std::vector<double> all;
  if (gather_slots(c, loc3, 4, all)) return 1;
  int win = -1;
  for (int r = 0; r < c->h.nranks; ++r) {
    if (all[4 * r + 1] < 0.0) continue;
    if (win < 0 || (want_max ? all[4 * r] > all[4 * win] : all[4 * r] < all[4 * win]) || (all[4 * r] == all[4 * win] && all[4 * r + 1] < all[4 * win + 1])) win = r;
  }
  if (win < 0) { c->err = "pop_global_extreme: the mask selects no physical cell"; return 1; }
  if (value) *value = all[4 * win];
  if (iloc) *iloc = (int)all[4 * win + 2];
  if (jloc) *jloc = (int)all[4 * win + 3];
  return 0;
}
// POP_GlobalCount (:2062-2207): non-zero cells of the physical domain (tripole: redundant top-row points of N-face /
// NE-corner fields counted once)
int pop_global_count(pop_ctx *c, const char *name, int tl, int n, int field_loc, long long *count) {
  if (need_device(c)) return 1;
  if (join_side(c)) return 1;
  double *p; long long cnt;
  if (resolve(c, name, tl, n, &p, &cnt)) { c->err = std::string("unknown field ") + name; return 1; }
  const double *dup = (c->h.c.ns_boundary == 2 && (field_loc == 1 || field_loc == 2)) ? c->d2["TRIPOLE_DUP"] : nullptr;
  hipLaunchKernelGGL(k_count_partial, grid_2d(c), dim3(POP_RED_THREADS), 0, c->stream, c->g, (const double *)p, dup, c->partial);
  if (reduce_finish<1>(c, FIN_PLAIN)) return 1;
  SolverScalars s;
  if (read_scalars(c, &s)) return 1;
  *count = (long long)s.sum0;
  return 0;
}
// POP_GlobalSumProd2DR8 (mpi/POP_ReductionsMod.F90:1395-1618): sum of A*B[*mask] over the physical domain
int pop_global_sum_prod(pop_ctx *c, const char *name_a, int tl_a, int n_a, const char *name_b, int tl_b, int n_b,
                        const char *mask_name, double *result) {
  if (need_device(c)) return 1;
  if (join_side(c)) return 1;
  double *pa, *pb, *mk = nullptr; long long cnt;
  if (resolve(c, name_a, tl_a, n_a, &pa, &cnt)) { c->err = std::string("unknown field ") + name_a; return 1; }
  if (resolve(c, name_b, tl_b, n_b, &pb, &cnt)) { c->err = std::string("unknown field ") + name_b; return 1; }
  if (mask_name && resolve(c, mask_name, 0, 0, &mk, &cnt)) { c->err = std::string("unknown mask ") + mask_name; return 1; }
  hipLaunchKernelGGL(k_dot_partial, grid_2d(c), dim3(POP_RED_THREADS), 0, c->stream, c->g, pa, (const double *)pb, mk, c->partial);
  if (reduce_finish<1>(c, FIN_PLAIN)) return 1;
  SolverScalars s;
  if (read_scalars(c, &s)) return 1;
  *result = s.sum0;
  return 0;
}
// POP_GlobalSumNfields2DR8 (mpi/POP_ReductionsMod.F90:823-1084): several fields, one result each
int pop_global_sum_nfields(pop_ctx *c, int nf, const char *const *names, const int *tl, const int *n, const char *mask_name, double *results) {
  for (int f = 0; f < nf; ++f)
    if (pop_global_sum(c, names[f], tl ? tl[f] : 1, n ? n[f] : 0, mask_name, results + f)) return 1;
  return 0;
}
// POP_GlobalSumScalarR8 (mpi/POP_ReductionsMod.F90:1091-1191): every rank's value lands in
// its own slot of a zeroed vector, the vector is all-reduced, and the slots are added in rank order
int pop_global_sum_scalar(pop_ctx *c, double local, double *result) {
  const int nr = c->h.nranks;
  if (nr == 1) { *result = local; return 0; }
  if (need_device(c)) return 1;
  if (!c->allred || !c->redbuf || c->red_doubles < nr) { c->err = "pop_global_sum_scalar: multi-rank run without a transport"; return 1; }
  std::vector<double> v(nr, 0.0);
  v[c->h.rank] = local;
  HIPCHK(c, hipMemcpyAsync(c->redbuf, v.data(), sizeof(double) * nr, hipMemcpyHostToDevice, c->stream));
  if (c->allred(c->comm_user, 0, nr)) { c->err = "pop_global_sum_scalar: allreduce failed" + tr_err(c); return 1; }
  HIPCHK(c, hipMemcpyAsync(v.data(), c->redbuf, sizeof(double) * nr, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double t = 0.0;
  for (int r = 0; r < nr; ++r) t = t + v[r];
  *result = t;
  return 0;
}
// POP_GlobalSum2DI4 (mpi/POP_ReductionsMod.F90:621-816): init-time integer fields live
// on the host; integer addition is exact in any order, ranks are combined through the scalar sum
int pop_global_sum_i4(pop_ctx *c, const char *name, long long *result) {
  auto it = c->h.i2.find(name);
  if (it == c->h.i2.end()) { c->err = std::string("unknown integer field ") + name; return 1; }
  const HostModel &h = c->h;
  long long s = 0;
  for (int lb = 0; lb < h.nblocks; ++lb) {
    const int *a = it->second.data() + (size_t)(h.local_ids[lb] - 1) * h.n2;
    const BlockInfo &B = h.all_blocks[h.local_ids[lb] - 1];
    for (int j = B.jb - 1; j < B.je; ++j) for (int i = B.ib - 1; i < B.ie; ++i) s += a[(size_t)j * h.nxb + i];
  }
  double tot = 0.0;
  if (pop_global_sum_scalar(c, (double)s, &tot)) return 1;
  *result = (long long)tot;
  return 0;
}
// POP_SolversDiagonal(diagonalCorrection, blockIndx, errorCode) POP_SolversMod.F90:1110-1151
int pop_solver_diagonal(pop_ctx *c, int block_local, const double *diagonal_correction) {
  if (need_device(c)) return 1;
  if (block_local < 1 || block_local > c->h.nblocks) { c->err = "pop_solver_diagonal: block index out of range"; return 1; }
  const int n = (int)c->h.n2;
  const size_t off = (size_t)(block_local - 1) * n;
  HIPCHK(c, hipMemcpyAsync(c->Q + off, diagonal_correction, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));   // Q: solver work array
  hipLaunchKernelGGL(k_solver_diagonal, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->g.WC0 + off, c->Q + off, c->centerWgt + off, n);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int pop_state_host(pop_ctx *c, int kk, const double *T, const double *S, double *rho, double *drhodt, double *drhods, long long n) {
  if (need_device(c)) return 1;
  return mix_state_host(c->h, c->g, kk, T, S, rho, drhodt, drhods, n, c->stream, c->err);
}

int pop_set_comm(pop_ctx *c, void *sb, void *rb, void *red, long long buf_doubles, pop_exchange_fn x, pop_allreduce_fn ar, void *user) {
  c->sendbuf = (double *)sb; c->recvbuf = (double *)rb; c->comm_doubles = buf_doubles;
  if (red) { c->redbuf = (double *)red; if (c->red_doubles == 0) c->red_doubles = 4LL * c->h.nblocks_tot; }
  c->xchg = x; c->allred = ar; c->comm_user = user;
  return 0;
}
// ---- in-library RCCL transport ------------------------------------------------------------------
int pop_rccl_unique_id(unsigned char *id128) {
  std::string err;
  if (rccl().load(err)) return 1;
  RcclApi::UniqueId id;
  if (rccl().GetUniqueId(&id)) return 1;
  std::memcpy(id128, id.internal, 128);
  return 0;
}
int pop_comm_init_rccl(pop_ctx *c, const unsigned char *id128) {
  if (need_device(c)) return 1;
  if (c->rccl_tr) { c->err = "pop_comm_init_rccl: transport already initialised"; return 1; }
  if (rccl().load(c->err)) return 1;
  RcclApi::UniqueId id;
  std::memcpy(id.internal, id128, 128);
  RcclTransport *t = new RcclTransport();
  const int rc = rccl().CommInitRank(&t->comm, c->h.nranks, id, c->h.rank);
  if (rc) { c->err = "pop_comm_init_rccl: ncclCommInitRank: " + rccl().what(rc); delete t; return 1; }
  c->rccl_tr = t;
  t->stream = &c->stream;
  const long long nb = pop_comm_buffer_doubles(c), nr = std::max<long long>(pop_reduce_buffer_doubles(c), 128);
  if (dev_alloc(c, &t->send, (size_t)nb) || dev_alloc(c, &t->recv, (size_t)nb) || dev_alloc(c, &t->red, (size_t)nr)) return 1;
  c->sendbuf = t->send; c->recvbuf = t->recv; c->comm_doubles = nb;
  c->redbuf = t->red; c->red_doubles = nr;
  c->xchg = rccl_exchange; c->allred = rccl_allreduce; c->comm_user = t;
  // second communicator for exchanges on the side stream (beside an all-reduce on the first).  Its id is made by
  // rank 0 and travels over the first communicator, one byte per double (exact under the sum with the other ranks'
  // zeros), so the host has nothing more to broadcast.  POP_RCCL_OVERLAP=0 keeps everything on one communicator.
  // (POP_RCCL_OVERLAP=2 also creates it on a single rank, so that the whole set-up can be checked against the real librccl)
  const int want2 = tun_or(c->h.tun.rccl_overlap, 1);
  if (c->side && want2 != 0 && (c->h.nranks > 1 || want2 == 2)) {
    std::vector<double> enc(128, 0.0);
    RcclApi::UniqueId id2;
    if (c->h.rank == 0) {
      if (rccl().GetUniqueId(&id2)) { c->err = "pop_comm_init_rccl: ncclGetUniqueId for the second communicator failed"; return 1; }
      for (int b = 0; b < 128; ++b) enc[b] = (double)(unsigned char)id2.internal[b];
    }
    HIPCHK(c, hipMemcpyAsync(t->red, enc.data(), 128 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (rccl_allreduce(t, 0, 128)) { c->err = "pop_comm_init_rccl: broadcasting the second id failed: " + t->err; return 1; }
    HIPCHK(c, hipMemcpyAsync(enc.data(), t->red, 128 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int b = 0; b < 128; ++b) id2.internal[b] = (char)(unsigned char)enc[b];
    const int rc2 = rccl().CommInitRank(&t->comm2, c->h.nranks, id2, c->h.rank);
    if (rc2) { c->err = "pop_comm_init_rccl: second communicator: " + rccl().what(rc2); return 1; }
    HIPCHK(c, hipStreamCreateWithFlags(&c->comm_side, hipStreamNonBlocking));
    t->side = &c->comm_side;
    c->xchg_side = rccl_exchange_side;
  }
  return 0;
}
int pop_comm_info(const pop_ctx *c, int *out6, char *lib_path, int lib_path_bytes) {
  if (!c || !out6) return 1;
  for (int i = 0; i < 6; ++i) out6[i] = 0;
  if (lib_path && lib_path_bytes > 0) lib_path[0] = 0;
  out6[4] = (int)c->peers.size();
  out6[5] = halo_async_ok(c) ? 1 : 0;
  if (c->rccl_tr) {
    out6[0] = 2; out6[1] = out6[2] = -1; out6[3] = c->rccl_tr->comm2 ? -1 : 0;
    RcclApi &a = rccl();
    if (a.CommCount) { a.CommCount(c->rccl_tr->comm, &out6[1]); if (c->rccl_tr->comm2) a.CommCount(c->rccl_tr->comm2, &out6[3]); }
    if (a.CommUserRank) a.CommUserRank(c->rccl_tr->comm, &out6[2]);
    if (lib_path && lib_path_bytes > 0) snprintf(lib_path, (size_t)lib_path_bytes, "%s", a.path.c_str());
  } else if (c->xchg) out6[0] = 1;
  return 0;
}
// transport self-test (also the single-rank check of the RCCL binding): every rank contributes
// rank+1 in slot `rank` of the reduce buffer and passes one value round the ring of ranks
int pop_comm_selftest(pop_ctx *c) {
  if (need_device(c)) return 1;
  if (!c->xchg || !c->allred || !c->redbuf || !c->sendbuf) { c->err = "pop_comm_selftest: no transport installed"; return 1; }
  const int nr = c->h.nranks;
  if (c->red_doubles < nr) { c->err = "pop_comm_selftest: reduce buffer too small"; return 1; }
  std::vector<double> v(nr, 0.0);
  v[c->h.rank] = c->h.rank + 1.0;
  // pageable host memory: blocking copies bracketed by stream synchronisation (an asynchronous copy from / to a std::vector is
  // only ordered with the stream once its staging has happened)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->redbuf, v.data(), sizeof(double) * nr, hipMemcpyHostToDevice));
  if (c->allred(c->comm_user, 0, nr)) { c->err = "pop_comm_selftest: allreduce failed" + (c->rccl_tr ? ": " + c->rccl_tr->err : std::string()); return 1; }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(v.data(), c->redbuf, sizeof(double) * nr, hipMemcpyDeviceToHost));
  const double probe = 1000.0 + c->h.rank;
  double back = 0.0;
  HIPCHK(c, hipMemcpy(c->sendbuf, &probe, sizeof(double), hipMemcpyHostToDevice));
  // ring: one value to rank+1, one from rank-1 (a self message on one rank)
  const int nxt = (c->h.rank + 1) % nr, prv = (c->h.rank + nr - 1) % nr;
  const int peer[2] = {nxt, prv};
  const long long z2[2] = {0, 0}, sc1[2] = {1, 0}, rc1[2] = {nxt == prv ? 1 : 0, 1};
  if (c->xchg(c->comm_user, nxt == prv ? 1 : 2, peer, z2, sc1, z2, rc1)) { c->err = "pop_comm_selftest: exchange failed" + (c->rccl_tr ? ": " + c->rccl_tr->err : std::string()); return 1; }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(&back, c->recvbuf, sizeof(double), hipMemcpyDeviceToHost));
  for (int r = 0; r < nr; ++r) if (v[r] != r + 1.0) {
    char msg[160]; snprintf(msg, sizeof msg, "pop_comm_selftest: all-reduce returned a wrong sum (slot %d holds %.17g, expected %d)", r, v[r], r + 1);
    c->err = msg; return 1;
  }
  if (back != 1000.0 + prv) { c->err = "pop_comm_selftest: ring send/recv returned a wrong value"; return 1; }
  if (c->xchg_side && c->comm_side) {   // the same ring through the second communicator on its own stream, beside an all-reduce on the first
    const double probe2 = 2000.0 + c->h.rank;
    double back2 = 0.0;
    HIPCHK(c, hipMemcpyAsync(c->sendbuf, &probe2, sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->comm_side, c->ev_fork, 0));
    if (c->xchg_side(c->comm_user, nxt == prv ? 1 : 2, peer, z2, sc1, z2, rc1)) { c->err = "pop_comm_selftest: side-stream exchange failed" + tr_err(c); return 1; }
    if (c->allred(c->comm_user, 0, nr)) { c->err = "pop_comm_selftest: allreduce beside the side-stream exchange failed" + tr_err(c); return 1; }
    HIPCHK(c, hipStreamSynchronize(c->comm_side));
    HIPCHK(c, hipMemcpyAsync(&back2, c->recvbuf, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (back2 != 2000.0 + prv) { c->err = "pop_comm_selftest: side-stream ring send/recv returned a wrong value"; return 1; }
  }
  return 0;
}

int pop_set_reduce_buffer(pop_ctx *c, void *dev_redbuf, long long doubles) { c->redbuf = (double *)dev_redbuf; c->red_doubles = doubles; return 0; }
long long pop_reduce_buffer_doubles(const pop_ctx *c) {
  long long n = 4LL * c->h.nblocks_tot;
  if (c->replicated) n = std::max(n, 2LL * (long long)c->h.n2 * c->h.nblocks_tot);
  return n;
}
int pop_set_stream(pop_ctx *c, void *hip_stream) {   // run on the host framework's stream (e.g. torch's current stream)
  if (need_device(c)) return 1;
  if (c->own_stream && c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
  c->stream = (hipStream_t)hip_stream; c->own_stream = false;
  return 0;
}
long long pop_comm_buffer_doubles(const pop_ctx *c) {
  long long cells = 0;
  for (auto &p : c->h.halo.peers) cells += (long long)std::max(p.send_src.size(), p.recv_dst.size());
  // the largest message: the end-of-step update of UBTROP, VBTROP, U, V, RHO and the tracers in one piece
  return std::max<long long>(cells * (long long)(2 + (3 + c->h.nt) * std::max(c->h.km, 1)), 4LL * c->h.nblocks_tot);
}
int pop_halo_plan_counts(const pop_ctx *c, int *nl, int *nf, int *np) {
  if (nl) *nl = (int)c->h.halo.copy_dst.size();
  if (nf) *nf = (int)c->h.halo.fill_dst.size();
  if (np) *np = (int)c->h.halo.peers.size();
  return 0;
}
int pop_halo_plan_peer(const pop_ctx *c, int ip, int *rank, int *ns, int *nr) {
  if (ip < 0 || ip >= (int)c->h.halo.peers.size()) return 1;
  const PeerPlan &p = c->h.halo.peers[ip];
  if (rank) *rank = p.rank;
  if (ns) *ns = (int)p.send_src.size();
  if (nr) *nr = (int)p.recv_dst.size();
  return 0;
}
int pop_halo_plan_lists(const pop_ctx *c, int ip, int *send_src, int *recv_dst) {
  if (ip < 0 || ip >= (int)c->h.halo.peers.size()) return 1;
  const PeerPlan &p = c->h.halo.peers[ip];
  if (send_src) std::copy(p.send_src.begin(), p.send_src.end(), send_src);
  if (recv_dst) std::copy(p.recv_dst.begin(), p.recv_dst.end(), recv_dst);
  return 0;
}
int pop_halo_plan_local(const pop_ctx *c, int *dst, int *src, int *fill_dst) {
  const HaloPlan &P = c->h.halo;
  if (dst) std::copy(P.copy_dst.begin(), P.copy_dst.end(), dst);
  if (src) std::copy(P.copy_src.begin(), P.copy_src.end(), src);
  if (fill_dst) std::copy(P.fill_dst.begin(), P.fill_dst.end(), fill_dst);
  return 0;
}

int pop_timers_reset(pop_ctx *c) { c->timers.clear(); c->timing = true; return 0; }
int pop_timer_ms(pop_ctx *c, const char *name, double *ms, int *calls) {
  auto it = c->timers.find(name);
  if (it == c->timers.end()) { if (ms) *ms = 0; if (calls) *calls = 0; return 1; }
  if (ms) *ms = it->second.ms;
  if (calls) *calls = it->second.calls;
  return 0;
}
int pop_device_sync(pop_ctx *c) {
  if (need_device(c)) return 1;
  if (join_side(c)) return 1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
// time `reps` launches of one kernel phase with HIP events on the launch stream (state is left as
// the phase leaves it; callers use it between steps for roofline measurement only)
typedef int (*phase_fn_t)(pop_ctx *);
static phase_fn_t phase_by_name(const std::string &p) {
  if (p == "vmix") return phase_vmix;
  if (p == "tracer_rhs") return [](pop_ctx *x) { return phase_tracer_rhs(x); };
  if (p == "tracer_rhs_fwd") return [](pop_ctx *x) { return phase_tracer_rhs(x, true); };
  if (p == "impvmixt_back") return phase_impvmixt_back;
  if (p == "impvmixt") return phase_impvmixt_pred;
  if (p == "state") return phase_state_new;
  if (p == "momentum_rhs") return [](pop_ctx *x) { return phase_momentum_rhs(x); };
  if (p == "impvmixu") return [](pop_ctx *x) { return phase_impvmixu(x, nullptr); };
  if (p == "correct") return phase_correct;
  if (p == "add_btrop") return [](pop_ctx *x) { return phase_add_btrop(x); };
  if (p == "hmix_tracer") return [](pop_ctx *x) { return phase_hmix_tracer(x); };
  if (p == "hmix_momentum") return [](pop_ctx *x) { return phase_hmix_momentum(x); };
  // the ordered block sums between the two kernels of a fused pcg iteration (k_block_sums<1>, one workgroup per block), as the
  // solver launches it: timing only (the partial array holds whatever the last solve left)
  if (p == "block_sums") return [](pop_ctx *x) { SolveView v = local_view(x); presum(x, v, v.partial, v.blocksum + 2 * v.nblocks_tot); return 0; };
  return nullptr;
}
// one phase of baroclinic_driver / baroclinic_correct_adjust on its own, once, on the launch stream with the step
// parameters pop_time_manager set: the public routines the reference's drivers call (vmix_coeffs vertical_mix.F90:518,
// tracer_update baroclinic.F90:1902 [tracer_rhs; hmix_tracer = the first Laplacian of hdifft_del4], impvmixt
// vertical_mix.F90:1164, state state_mod.F90:258 on the new tracers, clinic baroclinic.F90:1635 [momentum_rhs;
// hmix_momentum = first Laplacian of hdiffu_del4], impvmixu vertical_mix.F90:1679 + baroclinic.F90:1077-1129,
// impvmixt_correct :1460 [correct]).  Lets a caller -- and the parity tests -- drive and check the phases one by one.
int pop_run_phase(pop_ctx *c, const char *phase) {
  if (need_device(c) || join_side(c)) return 1;
  phase_fn_t fn = phase_by_name(phase ? phase : "");
  if (!fn) { c->err = std::string("unknown phase ") + (phase ? phase : "(null)"); return 1; }
  if (fn(c)) return 1;
  HIPCHK(c, hipGetLastError());
  return 0;
}
int pop_time_phase(pop_ctx *c, const char *phase, int reps, double *avg_ms) {
  if (need_device(c) || join_side(c)) return 1;
  const std::string p(phase);
  phase_fn_t fn = phase_by_name(p);
  if (!fn) { c->err = "unknown phase " + p; return 1; }
  struct Events {   // destroyed on every return path
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Events() { if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); }
  } ev;
  HIPCHK(c, hipEventCreate(&ev.e0)); HIPCHK(c, hipEventCreate(&ev.e1));
  // the launches of the phase only: the halo update of a first Laplacian formed for the next step (del4, large grids) belongs to
  // the step, not to the kernel, and the fields formed here are not kept
  struct Flag { bool &f; Flag(bool &x) : f(x) { f = true; } ~Flag() { f = false; } } timing(c->phase_timing);
  if (fn(c)) return 1;   // warm
  HIPCHK(c, hipEventRecord(ev.e0, c->stream));
  for (int r = 0; r < reps; ++r) if (fn(c)) return 1;
  HIPCHK(c, hipEventRecord(ev.e1, c->stream));
  HIPCHK(c, hipEventSynchronize(ev.e1));
  float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, ev.e0, ev.e1));
  c->d2t_next_valid = false; c->d2u_next_valid = false;
  *avg_ms = ms / std::max(reps, 1);
  return 0;
}

}  // extern "C"
