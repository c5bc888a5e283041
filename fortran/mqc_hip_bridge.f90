!! HIP bridge, compiled by metalquicha's CMake INSTEAD of backends/cuest/backend/mqc_cuest_bridge.f90
!! (or of src/methods/mqc_cuest_bridge_stub.f90) when the MI355X backend is enabled -- the same
!! one-of-{real, stub} selection src/methods/CMakeLists.txt:24-43 already makes.
!!
!! Exports exactly what hf_run / dft_run import (src/methods/mqc_method_hf.F90:197-216,
!! mqc_method_dft.F90:223):  run_cuest_scf(settings, fragment, result, want_gradient) and
!! cuest_backend_available().  Basis-file parsing and error_t stay on this side; the engine gets
!! plain arrays through fortran/mqc_hip_c.f90.  This file uses metalquicha's own modules and is
!! therefore built inside metalquicha's tree, not here (see INTEGRATION.md).
module mqc_cuest_bridge
   use, intrinsic :: iso_c_binding
   use pic_types, only: dp, int64
   use mqc_cuest_iface, only: cuest_scf_settings_t
   use mqc_physical_fragment, only: physical_fragment_t
   use mqc_result_types, only: calculation_result_t, SCF_CONVERGED, SCF_NOT_CONVERGED
   use mqc_cgto, only: molecular_basis_type
   use mqc_basis_file_reader, only: find_basis_file
   use mqc_json_basis_reader, only: build_molecular_basis_json
   use mqc_elements, only: element_number_to_symbol
   use mqc_error, only: error_t, ERROR_VALIDATION, ERROR_GENERIC
   use mqc_hip_c
   implicit none
   private

   public :: run_cuest_scf
   public :: cuest_backend_available

contains

   logical function cuest_backend_available() result(available)
      !! .true. when a HIP device is visible to this process
      available = mqc_hip_backend_available() /= 0
   end function cuest_backend_available

   subroutine run_cuest_scf(settings, fragment, result, want_gradient)
      type(cuest_scf_settings_t), intent(in) :: settings
      type(physical_fragment_t), intent(in) :: fragment
      type(calculation_result_t), intent(inout) :: result
      logical, intent(in), optional :: want_gradient

      type(molecular_basis_type) :: basis
      type(error_t) :: error
      character(len=:), allocatable :: path
      character(len=2), allocatable :: symbols(:)
      type(c_ptr) :: ctx
      type(mqc_hip_molecule_t) :: mol
      type(mqc_hip_basis_t) :: orb
      type(mqc_hip_scf_options_t) :: opts
      type(mqc_hip_scf_result_t) :: res
      integer(c_int32_t), allocatable, target :: z(:), shell_l(:), shell_nprim(:)
      integer(c_int64_t), allocatable, target :: nshell_per_atom(:)
      integer(c_int8_t), allocatable, target :: ghost(:)
      real(c_double), allocatable, target :: xyz(:), exps(:), coefs(:), eps(:)
      integer :: iatom, ish, nsh, nprim, off, rc, i

      ! ---- basis: same loader, same refusal of Cartesian sets (mqc_cuest_driver.f90:299-344)
      allocate (symbols(fragment%n_atoms))
      do iatom = 1, fragment%n_atoms
         symbols(iatom) = element_number_to_symbol(fragment%element_numbers(iatom))
      end do
      call find_basis_file(settings%basis_set, path, error)
      if (.not. error%has_error()) call build_molecular_basis_json(path, symbols, basis, error)
      if (error%has_error()) then
         call fail(result, ERROR_VALIDATION, error%get_message()); return
      end if
      if (basis%is_cartesian()) then
         call fail(result, ERROR_VALIDATION, "the basis set '"//trim(settings%basis_set)// &
                   "' is Cartesian; the HIP backend builds spherical shells only"); return
      end if

      ! ---- flatten molecular_basis_type into the PODs of include/mqc_hip.h
      nsh = 0; nprim = 0
      do iatom = 1, fragment%n_atoms
         nsh = nsh + basis%elements(iatom)%nshells
         do ish = 1, basis%elements(iatom)%nshells
            nprim = nprim + basis%elements(iatom)%shells(ish)%nfunc
         end do
      end do
      allocate (nshell_per_atom(fragment%n_atoms), shell_l(nsh), shell_nprim(nsh), exps(nprim), coefs(nprim))
      nsh = 0; off = 0
      do iatom = 1, fragment%n_atoms
         nshell_per_atom(iatom) = basis%elements(iatom)%nshells
         do ish = 1, basis%elements(iatom)%nshells
            nsh = nsh + 1
            shell_l(nsh) = basis%elements(iatom)%shells(ish)%ang_mom
            shell_nprim(nsh) = basis%elements(iatom)%shells(ish)%nfunc
            exps(off + 1:off + shell_nprim(nsh)) = basis%elements(iatom)%shells(ish)%exponents
            coefs(off + 1:off + shell_nprim(nsh)) = basis%elements(iatom)%shells(ish)%coefficients   ! RAW
            off = off + shell_nprim(nsh)
         end do
      end do
      orb%spherical = 1; orb%n_atoms = fragment%n_atoms; orb%n_shells = nsh
      orb%nshell_per_atom = c_loc(nshell_per_atom); orb%shell_l = c_loc(shell_l)
      orb%shell_nprim = c_loc(shell_nprim); orb%exponents = c_loc(exps); orb%coefficients = c_loc(coefs)

      allocate (z(fragment%n_atoms), xyz(3*fragment%n_atoms))
      z = fragment%element_numbers
      xyz = reshape(fragment%coordinates, [3*fragment%n_atoms])        ! (3,n) column-major == atom-major
      mol%n_atoms = fragment%n_atoms; mol%atomic_numbers = c_loc(z); mol%xyz = c_loc(xyz)
      mol%charge = fragment%charge; mol%multiplicity = fragment%multiplicity; mol%nelec = fragment%nelec
      mol%ghost = c_null_ptr
      if (allocated(fragment%is_ghost)) then
         allocate (ghost(fragment%n_atoms))
         ghost = merge(1_c_int8_t, 0_c_int8_t, fragment%is_ghost)
         mol%ghost = c_loc(ghost)
      end if

      call mqc_hip_default_options(opts)
      do i = 1, min(31, len_trim(settings%functional))
         opts%functional(i) = settings%functional(i:i)
      end do
      opts%density_fitting = merge(1, 0, settings%density_fitting)
      opts%grid_level = settings%grid_level
      opts%max_iter = settings%max_iter
      opts%energy_tol = settings%energy_tol; opts%density_tol = settings%density_tol
      opts%use_diis = merge(1, 0, settings%use_diis); opts%diis_size = settings%diis_size
      select case (trim(settings%guess))
      case ("core"); opts%guess = MQC_HIP_GUESS_CORE
      case ("gwh"); opts%guess = MQC_HIP_GUESS_GWH
      case ("auto"); opts%guess = MQC_HIP_GUESS_AUTO
      case default
         call fail(result, ERROR_VALIDATION, "initial guess '"//trim(settings%guess)// &
                   "' is not available on the HIP backend"); return
      end select
      opts%unrestricted = merge(1, 0, settings%unrestricted)
      opts%want_gradient = 0
      if (present(want_gradient)) opts%want_gradient = merge(1, 0, want_gradient)
      opts%allow_crap_scf = merge(1, 0, settings%allow_crap_scf)
      opts%verbose = merge(1, 0, settings%verbose)

      ! ---- context (process-wide singleton; device = device_rank mod device_count) and the run
      rc = mqc_hip_context_get(int(settings%device_rank, c_int32_t), ctx)
      if (rc /= MQC_HIP_OK) then
         call fail(result, ERROR_GENERIC, c_message(mqc_hip_last_error())); return
      end if
      allocate (eps(sum(2*shell_l + 1)))
      res%orbital_energies = c_loc(eps); res%density = c_null_ptr
      rc = mqc_hip_scf_run(ctx, mol, orb, c_null_ptr, opts, res)

      result%scf_iterations = res%iterations
      if (res%scf_status == MQC_HIP_SCF_CONVERGED) result%scf_status = SCF_CONVERGED
      if (res%scf_status == MQC_HIP_SCF_NOT_CONVERGED) result%scf_status = SCF_NOT_CONVERGED
      if (rc /= MQC_HIP_OK .or. res%has_error /= 0) then
         call fail(result, ERROR_GENERIC, trim(transfer(res%message, repeat(" ", 256)))); return
      end if
      result%energy%scf = res%e_total
      result%has_energy = .true.
      result%homo = res%homo; result%lumo = res%lumo
      result%has_orbitals = res%has_orbitals /= 0
   end subroutine run_cuest_scf

   subroutine fail(result, code, message)
      !! result%error%set(...), has_error, has_energy = .false.  (mqc_cuest_driver.f90:385-393)
      type(calculation_result_t), intent(inout) :: result
      integer, intent(in) :: code
      character(len=*), intent(in) :: message
      call result%error%set(code, message)
      result%has_error = .true.
      result%has_energy = .false.
   end subroutine fail

   function c_message(p) result(s)
      type(c_ptr), intent(in) :: p
      character(len=:), allocatable :: s
      character(kind=c_char), pointer :: buf(:)
      integer :: n
      s = ""
      if (.not. c_associated(p)) return
      call c_f_pointer(p, buf, [512])
      n = 0
      do while (n < 512)
         if (buf(n + 1) == c_null_char) exit
         n = n + 1
      end do
      allocate (character(len=n) :: s)
      s = transfer(buf(1:n), s)
   end function c_message

end module mqc_cuest_bridge
