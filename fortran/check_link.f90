!! Links fortran/mqc_hip_c.f90 against libmqc_hip.so and calls the lifecycle entry points.
program check_link
   use, intrinsic :: iso_c_binding
   use mqc_hip_c
   implicit none
   type(mqc_hip_scf_options_t) :: opts
   type(c_ptr) :: ctx
   integer :: rc
   print "(a,i0)", "abi version ", mqc_hip_abi_version()
   print "(a,i0)", "backend available ", mqc_hip_backend_available()
   call mqc_hip_default_options(opts)
   print "(a,i0,a,es9.2,a,i0)", "defaults: max_iter ", opts%max_iter, " energy_tol ", opts%energy_tol, &
      " diis_size ", opts%diis_size
   rc = mqc_hip_context_get(0_c_int32_t, ctx)
   print "(a,i0)", "context_get status ", rc
   rc = mqc_hip_finalize()
end program check_link
