!|||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||
! Minimal Fortran main replacing the coupler cap (drivers/mct/ocn_comp_mct.F90:627-678):
! reads a few settings from the command line, builds the model through the C ABI and runs the
! reference's `step` sequence from Fortran.  Prints global means and the solver diagnostics so
! the test suite can compare them with the Python-driven run.
!   pop_driver <nx> <ny> <km> <bx> <by> <vmix> <nsteps> [hostonly]
!|||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||
 program pop_driver

   use kinds_mod
   use pop_amd_c
   use blocks
   use step_mod, only: step
   use POP_SolversMod, only: POP_SolversGetDiagnostics
   use POP_ReductionsMod, only: POP_GlobalSum
   implicit none

   type (pop_config) :: cfg
   type (block) :: this_block
   integer (POP_i4) :: errorCode, nsteps, n, iters, flags, nocean
   integer (POP_i4), allocatable :: KMT(:,:,:)
   real (POP_r8) :: rms, tsum, psum
   character (char_len) :: arg, msg

   cfg = default_config()
   call geti(1, cfg%nx_global); call geti(2, cfg%ny_global); call geti(3, cfg%km)
   call geti(4, cfg%block_size_x); call geti(5, cfg%block_size_y); call geti(6, cfg%vmix_choice)
   call geti(7, nsteps)
   flags = 0
   if (command_argument_count() >= 8) flags = POP_CREATE_HOST_ONLY

   errorCode = pop_create(cfg, 0, 1, flags, pop_ctx)
   if (errorCode /= POP_Success) then
      call pop_amd_error_message(msg)
      write(*,*) 'pop_create failed: ', trim(msg)
      stop 1
   endif
   call init_blocks_from_ctx
   this_block = get_block(1, 1)
   write(*,'(a,4i6)') 'blocks: nx_block ny_block nblocks_tot nblocks_clinic', nx_block, ny_block, nblocks_tot, nblocks_clinic
   write(*,'(a,4i6)') 'block 1: ib ie jb je', this_block%ib, this_block%ie, this_block%jb, this_block%je
   allocate(KMT(nx_block, ny_block, nblocks_clinic))
   errorCode = pop_get_ifield(pop_ctx, cstr('KMT'), KMT, int(size(KMT), c_long_long))
   nocean = count(KMT > 0)
   write(*,'(a,i10)') 'ocean points (with ghosts): ', nocean
   if (flags == POP_CREATE_HOST_ONLY) then
      errorCode = pop_destroy(pop_ctx)
      stop
   endif

   do n = 1, nsteps
      call step(errorCode)
      if (errorCode /= POP_Success) then
         call pop_amd_error_message(msg)
         write(*,*) 'step failed: ', trim(msg)
         stop 2
      endif
      call POP_SolversGetDiagnostics(iters, rms, errorCode)
      tsum = POP_GlobalSum('TRACER', 1, 0, errorCode)     ! level-1 sums are enough for a checksum
      psum = POP_GlobalSum('PSURF', 1, 0, errorCode, mMask='mMask')
      write(*,'(a,i4,a,i5,a,es23.15,a,es23.15)') 'step ', n, ' iters ', iters, ' sumT1 ', tsum, ' sumP ', psum
   end do
   errorCode = pop_destroy(pop_ctx)

 contains

   subroutine geti(i, v)
      integer, intent(in) :: i
      integer (c_int), intent(inout) :: v
      if (command_argument_count() >= i) then
         call get_command_argument(i, arg)
         read(arg, *) v
      endif
   end subroutine

   ! the reference's code defaults for the supported options (tests/popcfg.py base_config)
   function default_config() result(c)
      type (pop_config) :: c
      c%nx_global = 48; c%ny_global = 40; c%km = 16; c%nt = 2
      c%block_size_x = 12; c%block_size_y = 10
      c%ew_boundary = 1; c%ns_boundary = 0
      c%hmix_momentum = 2; c%hmix_tracer = 2; c%lvariable_hmix = 0
      c%vmix_choice = 1; c%tadvect = 1; c%solver_choice = 1
      c%max_iterations = 1000; c%convergence_check_freq = 10
      c%tmix_opt = 2; c%time_mix_freq = 17; c%steps_per_day = 24
      c%lbouss_correct = 0; c%lpressure_avg = 1; c%impcor = 1; c%reset_to_freezing = 1
      c%lrich = 1; c%ldbl_diff = 0; c%lshort_wave = 0; c%lcheckekmo = 0; c%num_v_smooth_Ri = 1
      c%am = 3.0e9_c_double; c%ah = 1.0e7_c_double
      c%const_vvc = 0.25_c_double; c%const_vdc = 0.25_c_double
      c%convect_diff = 1000.0_c_double; c%convect_visc = 1000.0_c_double
      c%bottom_drag = 1.0e-3_c_double; c%aidif = 1.0_c_double
      c%rich_bckgrnd_vvc = 1.0_c_double; c%rich_bckgrnd_vdc = 0.1_c_double; c%rich_mix = 50.0_c_double
      c%bckgrnd_vdc1 = 0.1_c_double; c%bckgrnd_vdc2 = 0.0_c_double
      c%bckgrnd_vdc_dpth = 2500.0e2_c_double; c%bckgrnd_vdc_linv = 4.5e-5_c_double
      c%Prandtl = 10.0_c_double; c%kpp_rich_mix = 50.0_c_double
      c%convergence_criterion = 1.0e-12_c_double
      c%init_ts_perturbation = 1.0e-2_c_double
   end function default_config

 end program pop_driver
