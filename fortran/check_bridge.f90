!! Drives the drop-in bridge once: a hydrogen molecule with an inline two-shell basis handed over by the stub reader
!! is not possible (the stub reader returns no shells), so the call must come back with a validation error --
!! which proves the whole call chain (settings -> flat_basis -> error_t) links and runs.  With density_fitting
!! set, the auxiliary basis takes the same path (the round-1 bridge passed c_null_ptr for it).
program check_bridge
   use mqc_cuest_bridge, only: run_cuest_scf, cuest_backend_available
   use mqc_cuest_iface, only: cuest_scf_settings_t
   use mqc_physical_fragment, only: physical_fragment_t
   use mqc_result_types, only: calculation_result_t
   implicit none
   type(cuest_scf_settings_t) :: settings
   type(physical_fragment_t) :: frag
   type(calculation_result_t) :: res
   frag%n_atoms = 2
   allocate (frag%element_numbers(2), frag%coordinates(3, 2))
   frag%element_numbers = 1
   frag%coordinates = 0.0d0; frag%coordinates(3, 2) = 1.4d0
   frag%nelec = 2
   settings%density_fitting = .true.
   print "(a,l1)", "backend available ", cuest_backend_available()
   call run_cuest_scf(settings, frag, res, want_gradient=.true.)
   print "(a,l1,a,l1)", "has_error ", res%has_error, " has_energy ", res%has_energy
   print "(a,a)", "message: ", res%error%get_message()
end program check_bridge
