!|||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||
! A caller written against the REFERENCE's argument lists (the call style of source/step_mod.F90:361-560 and of
! the routines it calls), compiled against pop_amd_mods.F90 to show that code written for the reference's
! module-procedure surface compiles and runs unchanged in shape:
!    call dhdt(DH, DHU)
!    call baroclinic_driver(ZX, ZY, DH, DHU, errorCode)
!    call POP_HaloUpdate(ZX, POP_haloClinic, POP_gridHorzLocNECorner, POP_fieldKindVector, errorCode, fillValue = 0.0_POP_r8)
!    call barotropic_driver(ZX, ZY, errorCode)
!    call baroclinic_correct_adjust
! followed by the library's step tail, a global sum with the reference's POP_GlobalSum(array, dist, fieldLoc,
! errorCode, mMask | lMask), the legacy global_sum(X, dist, field_loc, MASK), grad / div / zcurl(k, ..., this_block) and -- once -- a stand-alone POP_SolversRun(sfcPressure, rhsClinic, errorCode) that must reproduce
! the pressure the step just computed.  Prints the same checksums as pop_driver.F90 (named-field forms), so the
! test suite can compare both with the Python-driven run.
!   pop_driver_ref <nx> <ny> <km> <bx> <by> <vmix> <nsteps> [<ns_boundary> <horiz_grid_file> <topography_file>]
!|||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||
 program pop_driver_ref

   use kinds_mod
   use pop_amd_c
   use blocks
   use POP_GridHorzMod
   use POP_FieldMod
   use POP_DomainMod, only: POP_haloClinic, POP_distrbClinic
   use POP_HaloMod, only: POP_HaloUpdate, POP_HaloCreate
   use POP_ReductionsMod, only: POP_GlobalSum
   use global_reductions, only: global_sum, field_loc_center
   use operators, only: grad, div, zcurl
   use POP_SolversMod, only: POP_SolversRun, POP_SolversGetDiagnostics, POP_SolversInit, POP_SolversPrep
   use surface_hgt, only: dhdt
   use baroclinic, only: baroclinic_driver, baroclinic_correct_adjust
   use barotropic, only: barotropic_driver
   implicit none

   type (pop_config) :: cfg
   integer (POP_i4) :: errorCode, nsteps, n, iters, iters2
   real (POP_r8), allocatable, dimension(:,:,:) :: DH, DHU, ZX, ZY, PSURF, RHS, GUESS, MASK, T1
   real (POP_r8) :: rms, tsum, psum, dmax
   real (POP_r8), allocatable, dimension(:,:) :: GX, GY, DV, CU
   integer (c_int), allocatable :: ids(:)
   real (POP_r4), allocatable :: P4(:,:,:), P43(:,:,:,:), P44(:,:,:,:,:)
   integer (POP_i4), allocatable :: I3(:,:,:,:), I4(:,:,:,:,:), K2(:,:,:)
   real (POP_r8), allocatable :: NF(:,:,:,:), nfsum(:)
   type (block) :: this_block
   integer (POP_i4) :: iblock
   character (char_len) :: arg, msg, horiz_grid_file, topography_file
   type (pop_grid_input) :: grid
   real (POP_r8), allocatable, target, dimension(:,:,:) :: GRID_G
   integer (POP_i4), allocatable, target, dimension(:,:) :: KMT_G

   cfg = default_config()
   call geti(1, cfg%nx_global); call geti(2, cfg%ny_global); call geti(3, cfg%km)
   call geti(4, cfg%block_size_x); call geti(5, cfg%block_size_y); call geti(6, cfg%vmix_choice)
   call geti(7, nsteps)
   if (command_argument_count() >= 10) then
      ! horiz_grid_opt = 'file', topography_opt = 'file' (grid.F90:441-445, 467-470): the reference's direct-access
      ! binary files, read whole and handed over as global arrays; <ns_boundary> 2 = tripole
      call geti(8, cfg%ns_boundary)
      call get_command_argument(9, horiz_grid_file)
      call get_command_argument(10, topography_file)
      allocate(GRID_G(cfg%nx_global, cfg%ny_global, 7), KMT_G(cfg%nx_global, cfg%ny_global))
      errorCode = pop_read_grid_files(cstr(trim(horiz_grid_file)), cstr(trim(topography_file)), cfg%nx_global, cfg%ny_global, &
                                      GRID_G, KMT_G)
      if (errorCode /= POP_Success) call die('pop_read_grid_files')
      grid%ULAT = c_loc(GRID_G(1,1,1)); grid%ULON = c_loc(GRID_G(1,1,2)); grid%HTN = c_loc(GRID_G(1,1,3))
      grid%HTE = c_loc(GRID_G(1,1,4)); grid%HUS = c_loc(GRID_G(1,1,5)); grid%HUW = c_loc(GRID_G(1,1,6))
      grid%ANGLE = c_loc(GRID_G(1,1,7)); grid%KMT = c_loc(KMT_G)
      errorCode = pop_create_with_grid(cfg, grid, 0, 1, 0, pop_ctx)
      deallocate(GRID_G, KMT_G)
   else
      errorCode = pop_create(cfg, 0, 1, 0, pop_ctx)
   endif
   if (errorCode /= POP_Success) call die('pop_create')
   call init_blocks_from_ctx
   POP_haloClinic = POP_HaloCreate(errorCode)
   call POP_SolversInit(errorCode)                                           ! initial.F90 calls both; set-up happened in pop_create
   if (errorCode /= POP_Success) call die('POP_SolversInit')
   call POP_SolversPrep(errorCode)
   if (errorCode /= POP_Success) call die('POP_SolversPrep')

   allocate(DH(nx_block,ny_block,nblocks_clinic), DHU(nx_block,ny_block,nblocks_clinic), ZX(nx_block,ny_block,nblocks_clinic), &
            ZY(nx_block,ny_block,nblocks_clinic), PSURF(nx_block,ny_block,nblocks_clinic), RHS(nx_block,ny_block,nblocks_clinic), &
            GUESS(nx_block,ny_block,nblocks_clinic), MASK(nx_block,ny_block,nblocks_clinic), T1(nx_block,ny_block,nblocks_clinic))
   errorCode = pop_get_field(pop_ctx, cstr('mMask'), 1, 0, MASK, int(size(MASK), c_long_long))

   do n = 1, nsteps
      errorCode = pop_time_manager(pop_ctx)                                  ! time_manager / set_switches
      if (errorCode /= POP_Success) call die('time_manager')
      call dhdt(DH, DHU)                                                     ! step_mod.F90:361
      call baroclinic_driver(ZX, ZY, DH, DHU, errorCode)                     ! :369
      if (errorCode /= POP_Success) call die('baroclinic_driver')
      call POP_HaloUpdate(ZX, POP_haloClinic, POP_gridHorzLocNECorner, &     ! :405-423
                          POP_fieldKindVector, errorCode, fillValue = 0.0_POP_r8)
      if (errorCode /= POP_Success) call die('POP_HaloUpdate(ZX)')
      call POP_HaloUpdate(ZY, POP_haloClinic, POP_gridHorzLocNECorner, &
                          POP_fieldKindVector, errorCode, fillValue = 0.0_POP_r8)
      if (errorCode /= POP_Success) call die('POP_HaloUpdate(ZY)')
      call barotropic_driver(ZX, ZY, errorCode)                              ! :431
      if (errorCode /= POP_Success) call die('barotropic_driver')
      call POP_SolversGetDiagnostics(iters, rms, errorCode)
      if (n == nsteps) then
         ! the elliptic solve on its own, reference argument list: same right-hand side, same first guess (PGUESS) -> the
         ! same iteration count; the pressure barotropic_driver left (null space removed) is put back afterwards
         errorCode = pop_get_field(pop_ctx, cstr('RHS'), 1, 0, RHS, int(size(RHS), c_long_long))
         errorCode = pop_get_field(pop_ctx, cstr('PGUESS'), 1, 0, GUESS, int(size(GUESS), c_long_long))
         errorCode = pop_get_field(pop_ctx, cstr('PSURF'), 2, 0, PSURF, int(size(PSURF), c_long_long))
         call POP_SolversRun(GUESS, RHS, errorCode)
         if (errorCode /= POP_Success) call die('POP_SolversRun')
         call POP_SolversGetDiagnostics(iters2, rms, errorCode)
         write(*,'(a,i5,a,i5)') 'solver rerun iters ', iters2, ' of ', iters
         errorCode = pop_set_field(pop_ctx, cstr('PSURF'), 2, 0, PSURF, int(size(PSURF), c_long_long))
      endif
      call baroclinic_correct_adjust                                         ! :458
      errorCode = pop_step_tail(pop_ctx)                                     ! :467-832
      if (errorCode /= POP_Success) call die('step tail')
      errorCode = pop_get_field(pop_ctx, cstr('PSURF'), 1, 0, PSURF, int(size(PSURF), c_long_long))
      errorCode = pop_get_field(pop_ctx, cstr('TRACER'), 1, 0, T1, int(size(T1), c_long_long))   ! first nx*ny*nblocks values only
      psum = POP_GlobalSum(PSURF, POP_distrbClinic, POP_gridHorzLocCenter, errorCode, mMask = MASK)
      if (errorCode /= POP_Success) call die('POP_GlobalSum')
      write(*,'(a,i4,a,i5,a,es23.15)') 'step ', n, ' iters ', iters, ' sumP ', psum
   end do
   write(*,'(a,es12.4)') 'halo: max |ZX| ', maxval(abs(ZX))
   ! the legacy reduction and the logical mask, reference lists (global_reductions.F90:383, POP_ReductionsMod.F90:144)
   write(*,'(a,es23.15)') 'legacy sumP ', global_sum(PSURF, POP_distrbClinic, field_loc_center, MASK)
   write(*,'(a,es23.15)') 'lmask sumP ', POP_GlobalSum(PSURF, POP_distrbClinic, POP_gridHorzLocCenter, errorCode, lMask = (MASK > 0.5_POP_r8))
   ! the remaining specifics of the two generic names (mpi/POP_HaloMod.F90:79-89, mpi/POP_ReductionsMod.F90:50-58), reference lists:
   ! single-precision and integer halo updates of 2-, 3- and 4-D arrays built from the surface pressure -- the ghost cells are cleared
   ! first, so what comes back was put there by the update -- and the scalar / integer / several-field sums
   allocate(P4(nx_block,ny_block,nblocks_clinic), P43(nx_block,ny_block,2,nblocks_clinic), P44(nx_block,ny_block,2,2,nblocks_clinic), &
            I3(nx_block,ny_block,2,nblocks_clinic), I4(nx_block,ny_block,2,2,nblocks_clinic), K2(nx_block,ny_block,nblocks_clinic), &
            NF(nx_block,ny_block,2,nblocks_clinic), nfsum(2))
   P4 = real(PSURF, POP_r4); call clear_ghosts_r4(P4)
   call POP_HaloUpdate(P4, POP_haloClinic, POP_gridHorzLocCenter, POP_fieldKindScalar, errorCode, fillValue = 0.0_POP_r4)
   if (errorCode /= POP_Success) call die('POP_HaloUpdate2DR4')
   write(*,'(a,es23.15)') 'halo r4 2d ', sum(abs(real(P4, POP_r8)))
   P43(:,:,1,:) = P4; P43(:,:,2,:) = 2.0_POP_r4 * P4
   do n = 1, 2
      call clear_ghosts_r4(P43(:,:,n,:))
   end do
   call POP_HaloUpdate(P43, POP_haloClinic, POP_gridHorzLocCenter, POP_fieldKindScalar, errorCode)
   if (errorCode /= POP_Success) call die('POP_HaloUpdate3DR4')
   write(*,'(a,es23.15)') 'halo r4 3d ', sum(abs(real(P43, POP_r8)))
   P44(:,:,:,1,:) = P43; P44(:,:,:,2,:) = -P43
   call POP_HaloUpdate(P44, POP_haloClinic, POP_gridHorzLocCenter, POP_fieldKindScalar, errorCode)
   if (errorCode /= POP_Success) call die('POP_HaloUpdate4DR4')
   write(*,'(a,es23.15)') 'halo r4 4d ', sum(abs(real(P44, POP_r8)))
   K2 = nint(1.0e3_POP_r8 * PSURF / max(maxval(abs(PSURF)), 1.0e-30_POP_r8))
   I3(:,:,1,:) = K2; I3(:,:,2,:) = K2 + 7
   I3(1:2,:,:,:) = -5000; I3(nx_block-1:nx_block,:,:,:) = -5000; I3(:,1:2,:,:) = -5000; I3(:,ny_block-1:ny_block,:,:) = -5000
   call POP_HaloUpdate(I3, POP_haloClinic, POP_gridHorzLocCenter, POP_fieldKindScalar, errorCode, fillValue = 0)
   if (errorCode /= POP_Success) call die('POP_HaloUpdate3DI4')
   write(*,'(a,i12)') 'halo i4 3d ', sum(abs(I3))
   I4(:,:,:,1,:) = I3; I4(:,:,:,2,:) = 2 * I3
   call POP_HaloUpdate(I4, POP_haloClinic, POP_gridHorzLocCenter, POP_fieldKindScalar, errorCode)
   if (errorCode /= POP_Success) call die('POP_HaloUpdate4DI4')
   write(*,'(a,i12)') 'halo i4 4d ', sum(abs(I4))
   write(*,'(a,es23.15)') 'scalar r8 ', POP_GlobalSum(2.5_POP_r8, POP_distrbClinic, errorCode)
   write(*,'(a,es23.15)') 'scalar r4 ', real(POP_GlobalSum(1.25_POP_r4, POP_distrbClinic, errorCode), POP_r8)
   write(*,'(a,i12)') 'scalar i4 ', POP_GlobalSum(3, POP_distrbClinic, errorCode)
   write(*,'(a,i12)') 'sum i4 2d ', POP_GlobalSum(K2, POP_distrbClinic, POP_gridHorzLocCenter, errorCode, lMask = (MASK > 0.5_POP_r8))
   write(*,'(a,es23.15)') 'sum r4 2d ', real(POP_GlobalSum(real(PSURF, POP_r4), POP_distrbClinic, POP_gridHorzLocCenter, errorCode, mMask = real(MASK, POP_r4)), POP_r8)
   NF(:,:,1,:) = PSURF; NF(:,:,2,:) = 3.0_POP_r8 * PSURF
   nfsum = POP_GlobalSum(NF, POP_distrbClinic, POP_gridHorzLocCenter, errorCode, mMask = MASK)
   write(*,'(a,2es23.15)') 'sum nfields ', nfsum
   ! operators.F90 with its own lists, block by block: grad of the surface pressure, then div and zcurl of that gradient
   allocate(ids(nblocks_clinic), GX(nx_block,ny_block), GY(nx_block,ny_block), DV(nx_block,ny_block), CU(nx_block,ny_block))
   errorCode = pop_local_block_ids(pop_ctx, ids)
   do iblock = 1, nblocks_clinic
      this_block = get_block(ids(iblock), iblock)
      call grad(1, GX, GY, PSURF(:,:,iblock), this_block)
      call div(1, DV, GX, GY, this_block)
      call zcurl(1, CU, GX, GY, this_block)
      write(*,'(a,i4,4es23.15)') 'ops block ', iblock, sum(abs(GX)), sum(abs(GY)), sum(abs(DV)), sum(abs(CU))
   end do
   errorCode = pop_destroy(pop_ctx)

 contains

   subroutine die(what)
      character (*), intent(in) :: what
      call pop_amd_error_message(msg)
      write(*,*) trim(what), ' failed: ', trim(msg)
      stop 2
   end subroutine

   subroutine clear_ghosts_r4(A)
      real (POP_r4), intent(inout) :: A(:,:,:)
      A(1:2,:,:) = -99.0_POP_r4; A(size(A,1)-1:,:,:) = -99.0_POP_r4; A(:,1:2,:) = -99.0_POP_r4; A(:,size(A,2)-1:,:) = -99.0_POP_r4
   end subroutine

   subroutine geti(i, v)
      integer, intent(in) :: i
      integer (c_int), intent(inout) :: v
      if (command_argument_count() >= i) then
         call get_command_argument(i, arg)
         read(arg, *) v
      endif
   end subroutine

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

 end program pop_driver_ref
