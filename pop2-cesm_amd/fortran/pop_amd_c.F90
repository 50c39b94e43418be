!|||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||
! pop_amd_c -- ISO_C_BINDING interfaces of libpop_amd.so (include/pop_amd.h).
! The Fortran host owns namelists and the step sequence; all model state
! lives on the GPU behind the opaque handle `pop_ctx`.
!|||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||
 module pop_amd_c

   use, intrinsic :: iso_c_binding
   implicit none
   public

   integer (c_int), parameter :: POP_CREATE_HOST_ONLY = 1

   ! mirrors `struct pop_config` field for field
   integer (c_int), parameter :: POP_CONFIG_VERSION = 5

   type, bind(C) :: pop_config
      integer (c_int) :: struct_version = POP_CONFIG_VERSION
      integer (c_int) :: nx_global, ny_global, km, nt
      integer (c_int) :: block_size_x, block_size_y
      integer (c_int) :: ew_boundary, ns_boundary
      integer (c_int) :: hmix_momentum, hmix_tracer, lvariable_hmix
      integer (c_int) :: vmix_choice, tadvect, solver_choice
      integer (c_int) :: max_iterations, convergence_check_freq
      integer (c_int) :: tmix_opt, time_mix_freq, steps_per_day
      integer (c_int) :: lbouss_correct, lpressure_avg, impcor, reset_to_freezing
      integer (c_int) :: lrich, ldbl_diff, lshort_wave, lcheckekmo, num_v_smooth_Ri
      integer (c_int) :: maxlanczosstep = 0, convergence_check_start = 0   ! solvers_nml (PCSI); 0 = the reference's defaults
      integer (c_int) :: preconditioner_choice = 0                         ! solvers_nml preconditionerChoice: 0 'diagonal', 1 'evp'
      integer (c_int) :: stepped_bathymetry = 0                            ! test extension of the internal topography
      integer (c_int) :: distribution_type = 0                             ! domain_nml: 0 equal block counts, 1 equal ocean columns
      integer (c_int) :: kpp_ml_diagnostics = 0                            ! HMXL, HMXL_DR every step
      integer (c_int) :: sw_absorption_type = 0, jerlov_water_type = 0, lsw_absorb = 0   ! sw_absorption_nml
      integer (c_int) :: partial_bottom_cells = 0                          ! grid_nml
      integer (c_int) :: gm_slope_control = 0                              ! hmix_gm_nml slope_control_choice: 0 'notanh', 1 'tanh', 2 'clip', 3 'Gerd' (hmix_tracer = 3)
      integer (c_int) :: gm_kappa_type = 0, gm_kappa_freq = 0              ! hmix_gm_nml kappa_*_choice: 0 constant, 1 bfre, 2 depth; kappa_freq_choice: 0 never, 1 every_time_step, 2 once_a_day
      real (c_double) :: am, ah
      real (c_double) :: const_vvc, const_vdc
      real (c_double) :: convect_diff, convect_visc, bottom_drag, aidif
      real (c_double) :: rich_bckgrnd_vvc, rich_bckgrnd_vdc, rich_mix
      real (c_double) :: bckgrnd_vdc1, bckgrnd_vdc2, bckgrnd_vdc_dpth, bckgrnd_vdc_linv
      real (c_double) :: Prandtl, kpp_rich_mix
      real (c_double) :: convergence_criterion
      real (c_double) :: init_ts_perturbation = 0.0_c_double
      real (c_double) :: robert_alpha = 0.0_c_double, robert_nu = 0.0_c_double
      real (c_double) :: lanczos_convergence_criterion = 0.0_c_double
      real (c_double) :: ah_bolus = 0.0_c_double, ah_bkg_srfbl = 0.0_c_double   ! hmix_gm_nml; 0 = ah
      real (c_double) :: slm_r = 0.0_c_double, slm_b = 0.0_c_double             ! hmix_gm_nml; 0 = 0.3
      integer (c_int) :: gm_transition_layer = 0                                ! hmix_gm_nml transition_layer_on
      integer (c_int) :: gm_diag_bolus = 0                                      ! hmix_gm_nml diag_gm_bolus: UISOP, VISOP, WISOP every step
      integer (c_int) :: gm_kappa_bkg_srfbl = 0                                 ! hmix_gm_nml: 1 = use_const_ah_bkg_srfbl .false.
      integer (c_int) :: reserved_i(1) = 0
      real (c_double) :: ah_bkg_bottom = 0.0_c_double                           ! hmix_gm_nml ah_bkg_bottom
      real (c_double) :: kappa_depth_1 = 0.0_c_double, kappa_depth_2 = 0.0_c_double, kappa_depth_scale = 0.0_c_double   ! gm_kappa_type = 2
   end type pop_config

   ! mirrors `struct pop_grid_input`: the records of horiz_grid_file / topography_file (grid.F90:1314-1542, 2025-2107)
   ! as global (nx_global,ny_global) arrays; c_loc of the host arrays, c_null_ptr for an absent ANGLE / KMT
   type, bind(C) :: pop_grid_input
      type (c_ptr) :: ULAT, ULON, HTN, HTE, HUS, HUW, ANGLE
      type (c_ptr) :: KMT
      type (c_ptr) :: DZBC = c_null_ptr   ! partial_bottom_cells: record of bottom_cell_file (grid.F90:2116-2186)
   end type pop_grid_input
   ! pop_tuning (include/pop_amd.h): kernel-form and schedule choices, none changes a result; fill with pop_tuning_init, then set fields
   type, bind(C) :: pop_tuning
      integer (c_int) :: struct_bytes, land_skip, land_full_steps, xcd_remap, red_tiles, red_band
      integer (c_int) :: lds_order, momentum_lds, tracer_lds, generic_thomas, reg_thomas_t, thomas_pair
      integer (c_int) :: tracer_fwd, vdc_shared, side_stream, del4_side, del4_tile, d2t_fuse
      integer (c_int) :: d2u_fuse, vmixu_defer, vmixu_inline, btrop_inline, kpp_ahead, kpp_col
      integer (c_int) :: kpp_lazy, kpp_ushear_hint, kpp_ushear_margin, kpp_side_stream, kpp_buoy_waves, kpp_interior_generic
      integer (c_int) :: kpp_src_full, solver_unfused, solver_nograph, solver_presum, solver_distributed, solver_overlap_off
      integer (c_int) :: fpcg_b2, pcsi_step2, halo_separate, halo_overlap_off, rccl_overlap, evp_wave
      integer (c_int) :: fpcg_a_pair, kpp_sparse, pbc_generic_thomas, pbc_generic_kpp, stream_priority, gm_sf_stored, state3d_levels
      integer (c_int) :: pcg_persist
      integer (c_int) :: gm_flux_tile
      integer (c_int) :: pcsi_two_step
      integer (c_int) :: block_sums_relay
      integer (c_int) :: pcsi_evp_fused
   end type pop_tuning

   type (c_ptr), save :: pop_ctx = c_null_ptr   ! the one model instance of this task

   interface
      integer (c_int) function pop_create(cfg, rank, nranks, flags, ctx) bind(C, name='pop_create')
         import :: c_int, c_ptr, pop_config
         type (pop_config), intent(in) :: cfg
         integer (c_int), value :: rank, nranks, flags
         type (c_ptr), intent(out) :: ctx
      end function
      integer (c_int) function pop_create_with_grid(cfg, grid, rank, nranks, flags, ctx) bind(C, name='pop_create_with_grid')
         import :: c_int, c_ptr, pop_config, pop_grid_input
         type (pop_config), intent(in) :: cfg
         type (pop_grid_input), intent(in) :: grid
         integer (c_int), value :: rank, nranks, flags
         type (c_ptr), intent(out) :: ctx
      end function
      integer (c_int) function pop_read_grid_files(horiz_grid_file, topography_file, nx_global, ny_global, seven_records, kmt) &
                                bind(C, name='pop_read_grid_files')
         import :: c_int, c_char, c_double
         character (kind=c_char), dimension(*), intent(in) :: horiz_grid_file, topography_file
         integer (c_int), value :: nx_global, ny_global
         real (c_double), dimension(*), intent(out) :: seven_records
         integer (c_int), dimension(*), intent(out) :: kmt
      end function
      integer (c_int) function pop_destroy(ctx) bind(C, name='pop_destroy')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
      end function
      type (c_ptr) function pop_last_error(ctx) bind(C, name='pop_last_error')
         import :: c_ptr
         type (c_ptr), value :: ctx
      end function
      integer (c_int) function pop_get_dim(ctx, name) bind(C, name='pop_get_dim')
         import :: c_int, c_ptr, c_char
         type (c_ptr), value :: ctx
         character (kind=c_char), intent(in) :: name(*)
      end function
      real (c_double) function pop_get_scalar(ctx, name) bind(C, name='pop_get_scalar')
         import :: c_double, c_ptr, c_char
         type (c_ptr), value :: ctx
         character (kind=c_char), intent(in) :: name(*)
      end function
      integer (c_int) function pop_get_block(ctx, block_id, out8, i_glob, j_glob) bind(C, name='pop_get_block')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
         integer (c_int), value :: block_id
         integer (c_int), intent(out) :: out8(8), i_glob(*), j_glob(*)
      end function
      integer (c_int) function pop_local_block_ids(ctx, ids) bind(C, name='pop_local_block_ids')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
         integer (c_int), intent(out) :: ids(*)
      end function
      integer (c_int) function pop_get_field(ctx, name, tl, n, host, count) bind(C, name='pop_get_field')
         import :: c_int, c_ptr, c_char, c_double, c_long_long
         type (c_ptr), value :: ctx
         character (kind=c_char), intent(in) :: name(*)
         integer (c_int), value :: tl, n
         real (c_double), intent(out) :: host(*)
         integer (c_long_long), value :: count
      end function
      integer (c_int) function pop_set_field(ctx, name, tl, n, host, count) bind(C, name='pop_set_field')
         import :: c_int, c_ptr, c_char, c_double, c_long_long
         type (c_ptr), value :: ctx
         character (kind=c_char), intent(in) :: name(*)
         integer (c_int), value :: tl, n
         real (c_double), intent(in) :: host(*)
         integer (c_long_long), value :: count
      end function
      integer (c_int) function pop_get_ifield(ctx, name, host, count) bind(C, name='pop_get_ifield')
         import :: c_int, c_ptr, c_char, c_long_long
         type (c_ptr), value :: ctx
         character (kind=c_char), intent(in) :: name(*)
         integer (c_int), intent(out) :: host(*)
         integer (c_long_long), value :: count
      end function
      integer (c_int) function pop_time_manager(ctx) bind(C, name='pop_time_manager')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
      end function
      integer (c_int) function pop_dhdt(ctx) bind(C, name='pop_dhdt')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
      end function
      integer (c_int) function pop_baroclinic_driver(ctx) bind(C, name='pop_baroclinic_driver')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
      end function
      integer (c_int) function pop_barotropic_driver(ctx) bind(C, name='pop_barotropic_driver')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
      end function
      integer (c_int) function pop_barotropic_driver_updated(ctx) bind(C, name='pop_barotropic_driver_updated')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
      end function
      integer (c_int) function pop_baroclinic_correct_adjust(ctx) bind(C, name='pop_baroclinic_correct_adjust')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
      end function
      integer (c_int) function pop_step_tail(ctx) bind(C, name='pop_step_tail')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
      end function
      integer (c_int) function pop_step(ctx) bind(C, name='pop_step')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
      end function
      integer (c_int) function pop_halo_update(ctx, name, tl, n) bind(C, name='pop_halo_update')
         import :: c_int, c_ptr, c_char
         type (c_ptr), value :: ctx
         character (kind=c_char), intent(in) :: name(*)
         integer (c_int), value :: tl, n
      end function
      integer (c_int) function pop_halo_update_host_r8(ctx, array, nz, fill) bind(C, name='pop_halo_update_host_r8')
         import :: c_int, c_ptr, c_double
         type (c_ptr), value :: ctx
         real (c_double), intent(inout) :: array(*)
         integer (c_int), value :: nz
         real (c_double), value :: fill
      end function
      integer (c_int) function pop_halo_update_host_i4(ctx, array, nz, fill) bind(C, name='pop_halo_update_host_i4')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
         integer (c_int), intent(inout) :: array(*)
         integer (c_int), value :: nz, fill
      end function
      integer (c_int) function pop_global_sum(ctx, name, tl, n, mask_name, res) bind(C, name='pop_global_sum')
         import :: c_int, c_ptr, c_char, c_double
         type (c_ptr), value :: ctx
         character (kind=c_char), intent(in) :: name(*)
         integer (c_int), value :: tl, n
         type (c_ptr), value :: mask_name
         real (c_double), intent(out) :: res
      end function
      integer (c_int) function pop_global_sum_host(ctx, array, mask, field_loc, res) bind(C, name='pop_global_sum_host')
         import :: c_int, c_ptr, c_double
         type (c_ptr), value :: ctx
         real (c_double), intent(in) :: array(*)
         type (c_ptr), value :: mask                 ! c_loc of a host array of the same shape, or c_null_ptr
         integer (c_int), value :: field_loc
         real (c_double), intent(out) :: res
      end function
      integer (c_int) function pop_solver_run(ctx) bind(C, name='pop_solver_run')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
      end function
      integer (c_int) function pop_solver_get_diagnostics(ctx, iters, rms) bind(C, name='pop_solver_get_diagnostics')
         import :: c_int, c_ptr, c_double
         type (c_ptr), value :: ctx
         integer (c_int), intent(out) :: iters
         real (c_double), intent(out) :: rms
      end function
      integer (c_int) function pop_device_sync(ctx) bind(C, name='pop_device_sync')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
      end function
      integer (c_int) function pop_global_sum_prod(ctx, na, tla, nna, nb, tlb, nnb, mask_name, res) bind(C, name='pop_global_sum_prod')
         import :: c_int, c_ptr, c_char, c_double
         type (c_ptr), value :: ctx
         character (kind=c_char), intent(in) :: na(*), nb(*)
         integer (c_int), value :: tla, nna, tlb, nnb
         type (c_ptr), value :: mask_name
         real (c_double), intent(out) :: res
      end function
      integer (c_int) function pop_global_sum_scalar(ctx, x, res) bind(C, name='pop_global_sum_scalar')
         import :: c_int, c_ptr, c_double
         type (c_ptr), value :: ctx
         real (c_double), value :: x
         real (c_double), intent(out) :: res
      end function
      integer (c_int) function pop_global_count(ctx, name, tl, n, field_loc, res) bind(C, name='pop_global_count')
         import :: c_int, c_ptr, c_char, c_long_long
         type (c_ptr), value :: ctx
         character (kind=c_char), intent(in) :: name(*)
         integer (c_int), value :: tl, n, field_loc
         integer (c_long_long), intent(out) :: res
      end function
      integer (c_int) function pop_global_extreme(ctx, name, tl, n, mask_name, want_max, val, iloc, jloc) &
                                                  bind(C, name='pop_global_extreme')
         import :: c_int, c_ptr, c_char, c_double
         type (c_ptr), value :: ctx
         character (kind=c_char), intent(in) :: name(*)
         integer (c_int), value :: tl, n, want_max
         type (c_ptr), value :: mask_name
         real (c_double), intent(out) :: val
         integer (c_int), intent(out) :: iloc, jloc
      end function
      integer (c_int) function pop_global_sum_i4(ctx, name, res) bind(C, name='pop_global_sum_i4')
         import :: c_int, c_ptr, c_char, c_long_long
         type (c_ptr), value :: ctx
         character (kind=c_char), intent(in) :: name(*)
         integer (c_long_long), intent(out) :: res
      end function
      integer (c_int) function pop_solver_diagonal(ctx, block_local, corr) bind(C, name='pop_solver_diagonal')
         import :: c_int, c_ptr, c_double
         type (c_ptr), value :: ctx
         integer (c_int), value :: block_local
         real (c_double), intent(in) :: corr(*)
      end function
      integer (c_int) function pop_write_restart(ctx, path) bind(C, name='pop_write_restart')
         import :: c_int, c_ptr, c_char
         type (c_ptr), value :: ctx
         character (kind=c_char), intent(in) :: path(*)
      end function
      integer (c_int) function pop_read_restart(ctx, path, flags) bind(C, name='pop_read_restart')
         import :: c_int, c_ptr, c_char
         type (c_ptr), value :: ctx
         character (kind=c_char), intent(in) :: path(*)
         integer (c_int), value :: flags
      end function
      integer (c_int) function pop_operator(ctx, op, k, aname, bname, tl, o1name, o2name) bind(C, name='pop_operator')
         import :: c_int, c_ptr, c_char
         type (c_ptr), value :: ctx
         integer (c_int), value :: op, k, tl
         character (kind=c_char), intent(in) :: aname(*), bname(*), o1name(*), o2name(*)
      end function
      subroutine pop_tuning_init(t) bind(C, name='pop_tuning_init')
         import :: pop_tuning
         type (pop_tuning), intent(out) :: t
      end subroutine
      integer (c_int) function pop_get_tuning(ctx, resolved) bind(C, name='pop_get_tuning')
         import :: c_int, c_ptr, pop_tuning
         type (c_ptr), value :: ctx
         type (pop_tuning), intent(out) :: resolved
      end function
      integer (c_int) function pop_create_tuned(cfg, grid, tuning, rank, nranks, flags, ctx) bind(C, name='pop_create_tuned')
         import :: c_int, c_ptr, pop_config, pop_tuning
         type (pop_config), intent(in) :: cfg
         type (c_ptr), value :: grid             ! c_loc of a pop_grid_input, or c_null_ptr
         type (pop_tuning), intent(in) :: tuning
         integer (c_int), value :: rank, nranks, flags
         type (c_ptr), intent(out) :: ctx
      end function
      integer (c_int) function pop_operator_host(ctx, op, k, block_local, a, b, o1, o2) bind(C, name='pop_operator_host')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
         integer (c_int), value :: op, k, block_local
         type (c_ptr), value :: a, b, o1, o2
      end function
      integer (c_int) function pop_solver_preconditioner(ctx, xname, xtl, pxname, pxtl) bind(C, name='pop_solver_preconditioner')
         import :: c_int, c_ptr, c_char
         type (c_ptr), value :: ctx
         character (kind=c_char), intent(in) :: xname(*), pxname(*)
         integer (c_int), value :: xtl, pxtl
      end function
      integer (c_int) function pop_halo_update_loc(ctx, name, tl, n, loc, kind) bind(C, name='pop_halo_update_loc')
         import :: c_int, c_ptr, c_char
         type (c_ptr), value :: ctx
         character (kind=c_char), intent(in) :: name(*)
         integer (c_int), value :: tl, n, loc, kind
      end function
      integer (c_int) function pop_halo_update_host_r8_loc(ctx, array, nz, fill, loc, kind) bind(C, name='pop_halo_update_host_r8_loc')
         import :: c_int, c_ptr, c_double
         type (c_ptr), value :: ctx
         real (c_double), intent(inout) :: array(*)
         integer (c_int), value :: nz, loc, kind
         real (c_double), value :: fill
      end function
      ! in-library RCCL transport (include/pop_amd.h)
      integer (c_int) function pop_halo_update_host_i4_loc(ctx, array, nz, fill, loc, kind) bind(C, name='pop_halo_update_host_i4_loc')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
         integer (c_int), intent(inout) :: array(*)
         integer (c_int), value :: nz, fill, loc, kind
      end function
      integer (c_int) function pop_rccl_unique_id(id128) bind(C, name='pop_rccl_unique_id')
         import :: c_int, c_signed_char
         integer (c_signed_char), intent(out) :: id128(128)
      end function
      integer (c_int) function pop_comm_init_rccl(ctx, id128) bind(C, name='pop_comm_init_rccl')
         import :: c_int, c_ptr, c_signed_char
         type (c_ptr), value :: ctx
         integer (c_signed_char), intent(in) :: id128(128)
      end function
      integer (c_int) function pop_comm_selftest(ctx) bind(C, name='pop_comm_selftest')
         import :: c_int, c_ptr
         type (c_ptr), value :: ctx
      end function
   end interface

 contains

   ! C string helper: trimmed Fortran string + NUL
   function cstr(s) result(c)
      character (*), intent(in) :: s
      character (kind=c_char) :: c(len_trim(s)+1)
      integer :: i
      do i = 1, len_trim(s)
         c(i) = s(i:i)
      end do
      c(len_trim(s)+1) = c_null_char
   end function cstr

   ! error text of the last failing call (POP_ErrorSet analogue)
   subroutine pop_amd_error_message(msg)
      character (*), intent(out) :: msg
      type (c_ptr) :: p
      character (kind=c_char), pointer :: f(:)
      integer :: i
      msg = ' '
      p = pop_last_error(pop_ctx)
      if (.not. c_associated(p)) return
      call c_f_pointer(p, f, [len(msg)])
      do i = 1, len(msg)
         if (f(i) == c_null_char) exit
         msg(i:i) = f(i)
      end do
   end subroutine pop_amd_error_message

 end module pop_amd_c
