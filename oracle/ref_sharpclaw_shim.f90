! oracle/ref_sharpclaw_shim.f90 -- TEST INFRASTRUCTURE ONLY.
! A bind(C) driver around the REFERENCE's SharpClaw Fortran modules (compiled unchanged from
! /root/reference by oracle/Makefile).  It does what SharpClawSolver.set_fortran_parameters does
! through f2py (src/pyclaw/sharpclaw.py:257-283): set the ClawParams scalars, allocate the module
! arrays, fill dx/mthlim -- module allocatables cannot be reached through ctypes directly.
subroutine sc_setup(ndim_in, meqn, mwaves_in, mbc, maxnx, lim_type_in, weno_order_in, &
                    char_decomp_in, mcapa_in, dx_in, mthlim_in) bind(C, name="sc_setup")
    use iso_c_binding
    use ClawParams
    use workspace
    use reconstruct
    implicit none
    integer(c_int), value :: ndim_in, meqn, mwaves_in, mbc, maxnx, lim_type_in, weno_order_in
    integer(c_int), value :: char_decomp_in, mcapa_in
    real(c_double) :: dx_in(ndim_in)
    integer(c_int) :: mthlim_in(mwaves_in)
    integer :: i

    if (allocated(dx)) call dealloc_clawparams()
    ! (workspace's own dealloc routine declares its argument implicitly REAL: free by hand)
    if (allocated(amdq)) deallocate(amdq, apdq, amdq2, apdq2, ql, qr, wave, s, dtdx)
    if (allocated(dq1m)) call dealloc_recon_workspace(lim_type, char_decomp)
    ndim = ndim_in
    lim_type = lim_type_in
    weno_order = weno_order_in
    char_decomp = char_decomp_in
    tfluct_solver = .false.
    fwave = .false.
    mcapa = mcapa_in
    mwaves = mwaves_in
    call alloc_clawparams()
    do i = 1, ndim
        xlower(i) = 0.d0
        xupper(i) = 1.d0
        dx(i) = dx_in(i)
    end do
    do i = 1, mwaves
        mthlim(i) = mthlim_in(i)
    end do
    call alloc_workspace(maxnx, mbc, meqn, mwaves, char_decomp)
    call alloc_recon_workspace(maxnx, mbc, meqn, mwaves, lim_type, char_decomp)
end subroutine sc_setup

! The wave-based reconstructions of 1d/sharpclaw/reconstruct.f90 (char_decomp = 1: tvd2_wave :728-806, weno5_wave
! :393-478, weno5_fwave :481-565) called directly on arrays the test supplies -- the 1-D flux1.f90 that calls them needs
! an rp1 that is not in the reference tree.  kind 1 = tvd2_wave, 2 = weno5_wave, 3 = weno5_fwave (divides `wave` by `s`
! in place, as the Fortran does).  tvd2_wave keeps its limiter values in the private module array uu: allocated here
! once, through the module's own routine (call this entry BEFORE any sc_setup of the process: the module cannot be
! asked whether uu is allocated).
subroutine sc_recon_wave(kind, meqn, mwaves_in, n, q, ql, qr, wave, s, mthlim_in) bind(C, name="sc_recon_wave")
    use iso_c_binding
    use reconstruct
    implicit none
    integer(c_int), value :: kind, meqn, mwaves_in, n
    real(c_double) :: q(meqn, n), ql(meqn, n), qr(meqn, n), wave(meqn, mwaves_in, n), s(mwaves_in, n)
    integer(c_int) :: mthlim_in(mwaves_in)
    logical, save :: uu_ready = .false.

    select case (kind)
    case (1)
        if (.not. uu_ready) then
            call alloc_recon_workspace(4096, 3, meqn, 16, 1, 1)
            uu_ready = .true.
        end if
        call tvd2_wave(q, ql, qr, wave, s, mthlim_in)
    case (2)
        call weno5_wave(q, ql, qr, wave)
    case (3)
        call weno5_fwave(q, ql, qr, wave, s)
    end select
end subroutine sc_recon_wave
