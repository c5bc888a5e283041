!! HIP bridge, compiled by metalquicha's CMake INSTEAD of backends/cuest/backend/mqc_cuest_bridge.f90
!! (or of src/methods/mqc_cuest_bridge_stub.f90) when the MI355X backend is enabled -- the same
!! one-of-{real, stub} selection src/methods/CMakeLists.txt:24-43 already makes.
!!
!! Exports exactly what hf_run / dft_run import (src/methods/mqc_method_hf.F90:197-216,
!! mqc_method_dft.F90:223):  run_cuest_scf(settings, fragment, result, want_gradient) and
!! cuest_backend_available().  Basis-file parsing and error_t stay on this side; the engine gets
!! plain arrays through fortran/mqc_hip_c.f90.  It uses metalquicha's own modules and is built inside
!! metalquicha's tree (INTEGRATION.md); fortran/check_bridge.sh compiles it here against interface
!! stubs of those modules (fortran/stubs/, declarations only) so that it is type-checked in this repo.
!!
!! What it does, step by step, is run_cuest_scf's own sequence (backends/cuest/backend/mqc_cuest_driver.f90:37-276):
!! element symbols -> load_basis(orbital) [+ load_basis(auxiliary) when density_fitting] -> context ->
!! guess selection -> SCF -> result fields (energy%scf, scf_status, scf_iterations, homo/lumo, dipole, gradient).
!! The flattened shells are cached per (basis name, element sequence): the reference re-builds
!! molecular_basis_type for every fragment from a cached JSON tree (:82-88); an MBE job sees two or three
!! distinct element sequences, so after the first monomer and the first dimer nothing is parsed again.
module mqc_cuest_bridge
   use, intrinsic :: iso_c_binding
   use pic_types, only: dp
   use mqc_cuest_iface, only: cuest_scf_settings_t
   use mqc_physical_fragment, only: physical_fragment_t
   use mqc_result_types, only: calculation_result_t, SCF_CONVERGED, SCF_NOT_CONVERGED
   use mqc_cgto, only: molecular_basis_type
   use mqc_basis_utils, only: find_basis_file
   use mqc_json_basis_reader, only: build_molecular_basis_json
   use mqc_elements, only: element_number_to_symbol
   use mqc_error, only: error_t, ERROR_VALIDATION, ERROR_GENERIC
   use mqc_hip_c
   implicit none
   private

   public :: run_cuest_scf
   public :: cuest_backend_available

   !> One flattened basis: the arrays an mqc_hip_basis_t points into
   type :: flat_basis_t
      character(len=32) :: name = ""
      integer, allocatable :: z(:)                                   !! element sequence it was built for
      integer(c_int64_t), allocatable :: nshell_per_atom(:)
      integer(c_int32_t), allocatable :: shell_l(:), shell_nprim(:)
      real(c_double), allocatable :: exps(:), coefs(:)               !! RAW Basis-Set-Exchange coefficients
      integer :: n_ao = 0
   end type flat_basis_t

   integer, parameter :: CACHE_SLOTS = 8
   type(flat_basis_t), target, save :: cache(CACHE_SLOTS)
   integer, save :: cache_next = 1

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

      type(error_t) :: error
      type(flat_basis_t), pointer :: orb_flat, aux_flat
      type(c_ptr) :: ctx
      type(mqc_hip_molecule_t) :: mol
      type(mqc_hip_basis_t), target :: orb, aux
      type(mqc_hip_scf_options_t) :: opts
      type(mqc_hip_scf_result_t) :: res
      integer(c_int32_t), allocatable, target :: z(:)
      integer(c_int8_t), allocatable, target :: ghost(:)
      real(c_double), allocatable, target :: xyz(:), eps(:), grad(:)
      logical :: need_gradient
      integer :: rc, i

      need_gradient = .false.
      if (present(want_gradient)) need_gradient = want_gradient

      ! ---- basis sets: same loader, same refusal of Cartesian sets (mqc_cuest_driver.f90:299-344)
      call flat_basis(settings%basis_set, fragment, "orbital", orb_flat, error)
      if (error%has_error()) then
         call fail(result, ERROR_VALIDATION, error%get_message()); return
      end if
      call point_at(orb_flat, fragment%n_atoms, orb)
      aux_flat => null()
      if (settings%density_fitting) then
         call flat_basis(settings%aux_basis_set, fragment, "auxiliary", aux_flat, error)
         if (error%has_error()) then
            call fail(result, ERROR_VALIDATION, error%get_message()); return
         end if
         call point_at(aux_flat, fragment%n_atoms, aux)
      end if

      allocate (z(fragment%n_atoms), xyz(3*fragment%n_atoms))
      z = fragment%element_numbers
      xyz = reshape(fragment%coordinates, [3*fragment%n_atoms])        ! (3,n) column-major == atom-major
      mol%n_atoms = fragment%n_atoms; mol%atomic_numbers = c_loc(z); mol%xyz = c_loc(xyz)
      mol%charge = fragment%charge; mol%multiplicity = fragment%multiplicity; mol%nelec = fragment%nelec
      mol%ghost = c_null_ptr
      mol%n_point_charges = 0; mol%point_charge_xyz = c_null_ptr; mol%point_charges = c_null_ptr
      mol%h_extra = c_null_ptr
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
      case ("sad"); opts%guess = MQC_HIP_GUESS_SAD
      case ("sac"); opts%guess = MQC_HIP_GUESS_SAC
      case default
         ! refused rather than replaced by another guess, as the cuEST driver does (:104-121)
         call fail(result, ERROR_VALIDATION, "initial guess '"//trim(settings%guess)// &
                   "' is not available on the HIP backend"); return
      end select
      opts%unrestricted = merge(1, 0, settings%unrestricted)
      opts%want_gradient = merge(1, 0, need_gradient)
      opts%allow_crap_scf = merge(1, 0, settings%allow_crap_scf)
      opts%verbose = merge(1, 0, settings%verbose)

      ! ---- context (process-wide singleton; device = device_rank mod device_count) and the run
      rc = mqc_hip_context_get(int(settings%device_rank, c_int32_t), ctx)
      if (rc /= MQC_HIP_OK) then
         call fail(result, ERROR_GENERIC, c_message(mqc_hip_last_error())); return
      end if
      allocate (eps(max(orb_flat%n_ao, 1)))
      res%orbital_energies = c_loc(eps); res%density = c_null_ptr
      res%orbital_energies_beta = c_null_ptr
      res%gradient = c_null_ptr
      res%embedding_matrix = c_null_ptr; res%mulliken_charges = c_null_ptr
      if (need_gradient) then
         allocate (grad(3*fragment%n_atoms))
         grad = 0.0_c_double
         res%gradient = c_loc(grad)
      end if
      if (settings%density_fitting) then
         rc = mqc_hip_scf_run(ctx, mol, orb, c_loc(aux), opts, res)
      else
         rc = mqc_hip_scf_run(ctx, mol, orb, c_null_ptr, opts, res)
      end if

      result%scf_iterations = res%iterations
      if (res%scf_status == MQC_HIP_SCF_CONVERGED) result%scf_status = SCF_CONVERGED
      if (res%scf_status == MQC_HIP_SCF_NOT_CONVERGED) result%scf_status = SCF_NOT_CONVERGED
      if (rc /= MQC_HIP_OK .or. res%has_error /= 0) then
         if (res%has_error /= 0) then
            call fail(result, ERROR_GENERIC, c_chars(res%message))
         else
            call fail(result, ERROR_GENERIC, c_message(mqc_hip_last_error()))
         end if
         return
      end if
      result%energy%scf = res%e_total
      result%has_energy = .true.
      result%homo = res%homo; result%lumo = res%lumo
      result%has_orbitals = res%has_orbitals /= 0
      ! The dipole is what IR intensities are built from (mqc_cuest_driver.f90:264-270)
      if (res%has_dipole /= 0) then
         if (allocated(result%dipole)) deallocate (result%dipole)
         allocate (result%dipole(3))
         result%dipole = res%dipole
         result%has_dipole = .true.
      end if
      if (need_gradient .and. res%has_gradient /= 0) then
         if (allocated(result%gradient)) deallocate (result%gradient)
         allocate (result%gradient(3, fragment%n_atoms))
         result%gradient = reshape(grad, [3, fragment%n_atoms])
         result%has_gradient = .true.
      end if
   end subroutine run_cuest_scf

   subroutine flat_basis(basis_name, fragment, role, flat, error)
      !! load_basis (mqc_cuest_driver.f90:299-344) + flattening, cached per (name, element sequence)
      character(len=*), intent(in) :: basis_name
      type(physical_fragment_t), intent(in) :: fragment
      character(len=*), intent(in) :: role
      type(flat_basis_t), pointer, intent(out) :: flat
      type(error_t), intent(out) :: error

      type(molecular_basis_type) :: basis
      character(len=:), allocatable :: path
      character(len=2), allocatable :: symbols(:)
      integer :: slot, iatom, ish, nsh, nprim, off

      flat => null()
      if (len_trim(basis_name) == 0) then
         call error%set(ERROR_VALIDATION, "No basis set specified")
         return
      end if
      do slot = 1, CACHE_SLOTS
         if (.not. allocated(cache(slot)%z)) cycle
         if (trim(cache(slot)%name) /= trim(basis_name)) cycle
         if (size(cache(slot)%z) /= fragment%n_atoms) cycle
         if (any(cache(slot)%z /= fragment%element_numbers(1:fragment%n_atoms))) cycle
         flat => cache(slot)
         return
      end do

      allocate (symbols(fragment%n_atoms))
      do iatom = 1, fragment%n_atoms
         symbols(iatom) = element_number_to_symbol(fragment%element_numbers(iatom))
      end do
      call find_basis_file(basis_name, path, error)
      if (error%has_error()) return
      call build_molecular_basis_json(path, symbols, basis, error)
      if (error%has_error()) return
      if (basis%is_cartesian()) then
         call error%set(ERROR_VALIDATION, "the "//trim(role)//" basis set '"//trim(basis_name)// &
                        "' is Cartesian; the HIP backend builds spherical shells only")
         return
      end if

      nsh = 0; nprim = 0
      do iatom = 1, fragment%n_atoms
         if (basis%elements(iatom)%nshells == 0) then
            ! check_basis_covers_atoms (mqc_cuest_driver.f90:346-383): an element the file does not define
            call error%set(ERROR_VALIDATION, "the "//trim(role)//" basis set '"//trim(basis_name)// &
                           "' has no entry for element "//trim(symbols(iatom)))
            return
         end if
         nsh = nsh + basis%elements(iatom)%nshells
         do ish = 1, basis%elements(iatom)%nshells
            nprim = nprim + basis%elements(iatom)%shells(ish)%nfunc
         end do
      end do

      slot = cache_next
      cache_next = mod(cache_next, CACHE_SLOTS) + 1
      flat => cache(slot)
      if (allocated(flat%z)) deallocate (flat%z, flat%nshell_per_atom, flat%shell_l, flat%shell_nprim, flat%exps, flat%coefs)
      flat%name = basis_name
      allocate (flat%z(fragment%n_atoms), flat%nshell_per_atom(fragment%n_atoms), flat%shell_l(nsh), &
                flat%shell_nprim(nsh), flat%exps(nprim), flat%coefs(nprim))
      flat%z = fragment%element_numbers(1:fragment%n_atoms)
      nsh = 0; off = 0; flat%n_ao = 0
      do iatom = 1, fragment%n_atoms
         flat%nshell_per_atom(iatom) = basis%elements(iatom)%nshells
         do ish = 1, basis%elements(iatom)%nshells
            nsh = nsh + 1
            flat%shell_l(nsh) = basis%elements(iatom)%shells(ish)%ang_mom
            flat%shell_nprim(nsh) = basis%elements(iatom)%shells(ish)%nfunc
            flat%exps(off + 1:off + flat%shell_nprim(nsh)) = basis%elements(iatom)%shells(ish)%exponents
            flat%coefs(off + 1:off + flat%shell_nprim(nsh)) = basis%elements(iatom)%shells(ish)%coefficients   ! RAW
            off = off + flat%shell_nprim(nsh)
            flat%n_ao = flat%n_ao + 2*flat%shell_l(nsh) + 1
         end do
      end do
      call basis%destroy()
   end subroutine flat_basis

   subroutine point_at(flat, n_atoms, pod)
      !! the POD of include/mqc_hip.h over a cached flattened basis
      type(flat_basis_t), pointer, intent(in) :: flat
      integer, intent(in) :: n_atoms
      type(mqc_hip_basis_t), intent(out) :: pod
      pod%spherical = 1; pod%n_atoms = n_atoms; pod%n_shells = size(flat%shell_l)
      pod%nshell_per_atom = c_loc(flat%nshell_per_atom); pod%shell_l = c_loc(flat%shell_l)
      pod%shell_nprim = c_loc(flat%shell_nprim); pod%exponents = c_loc(flat%exps); pod%coefficients = c_loc(flat%coefs)
   end subroutine point_at

   subroutine fail(result, code, message)
      !! result%error%set(...), has_error, has_energy = .false.  (mqc_cuest_driver.f90:385-393)
      type(calculation_result_t), intent(inout) :: result
      integer, intent(in) :: code
      character(len=*), intent(in) :: message
      call result%error%set(code, message)
      result%has_error = .true.
      result%has_energy = .false.
   end subroutine fail

   function c_chars(buf) result(s)
      !! a NUL-terminated C character array as a Fortran string, cut at the first NUL
      character(kind=c_char), intent(in) :: buf(:)
      character(len=:), allocatable :: s
      integer :: n, i
      n = 0
      do while (n < size(buf))
         if (buf(n + 1) == c_null_char) exit
         n = n + 1
      end do
      allocate (character(len=n) :: s)
      do i = 1, n
         s(i:i) = buf(i)
      end do
   end function c_chars

   function c_message(p) result(s)
      type(c_ptr), intent(in) :: p
      character(len=:), allocatable :: s
      character(kind=c_char), pointer :: buf(:)
      s = ""
      if (.not. c_associated(p)) return
      call c_f_pointer(p, buf, [512])
      s = c_chars(buf)
   end function c_message

end module mqc_cuest_bridge
