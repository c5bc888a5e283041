!! HIP bridge, compiled by metalquicha's CMake INSTEAD of backends/cuest/backend/mqc_cuest_bridge.f90
!! (or of src/methods/mqc_cuest_bridge_stub.f90) when the MI355X backend is enabled -- the same
!! one-of-{real, stub} selection src/methods/CMakeLists.txt:24-43 already makes.
!!
!! Exports exactly what hf_run / dft_run import (src/methods/mqc_method_hf.F90:197-216,
!! mqc_method_dft.F90:223):  run_cuest_scf(settings, fragment, result, want_gradient) and
!! cuest_backend_available() -- plus run_cuest_scf_batch(settings, fragments(:), results(:), want_gradient),
!! the same call for MANY fragments at once (SURVEY.md section 8f item 4; used by the batched worker loop of
!! fortran/mqc_hip_node_worker.f90).  Basis-file parsing and error_t stay on this side; the engine gets
!! plain arrays through fortran/mqc_hip_c.f90.  It uses metalquicha's own modules and is built inside
!! metalquicha's tree (INTEGRATION.md); fortran/check_bridge.sh compiles it here against stand-ins
!! of those modules (fortran/stubs/) and RUNS it: fortran/check_bridge.f90 reproduces the reference's
!! check_rhf energy through this file on a GPU box.
!!
!! What it does, step by step, is run_cuest_scf's own sequence (backends/cuest/backend/mqc_cuest_driver.f90:37-276):
!! element symbols -> load_basis(orbital) [+ load_basis(auxiliary) when density_fitting] -> context ->
!! guess selection -> SCF -> result fields (energy%scf, scf_status, scf_iterations, homo/lumo, dipole, gradient).
!! The flattened shells are cached per (basis name, element sequence): the reference re-builds
!! molecular_basis_type for every fragment from a cached JSON tree (:82-88); an MBE job sees two or three
!! distinct element sequences, so after the first monomer and the first dimer nothing is parsed again.
!! The cache is least-recently-used over CACHE_SLOTS entries and NEVER evicts an entry the call in flight points
!! at (each call pins what it uses; a batch that needs more distinct entries than there are slots is cut into
!! sub-batches).
module mqc_cuest_bridge
   use, intrinsic :: iso_c_binding
   use, intrinsic :: iso_fortran_env, only: int64
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
   public :: run_cuest_scf_batch
   public :: cuest_backend_available
   public :: hip_device_visible
   public :: bridge_basis_cache_misses

   !> One flattened basis: the arrays an mqc_hip_basis_t points into
   type :: flat_basis_t
      character(len=32) :: name = ""
      integer, allocatable :: z(:)                                   !! element sequence it was built for
      integer(c_int64_t), allocatable :: nshell_per_atom(:)
      integer(c_int32_t), allocatable :: shell_l(:), shell_nprim(:)
      real(c_double), allocatable :: exps(:), coefs(:)               !! RAW Basis-Set-Exchange coefficients
      integer :: n_ao = 0
   end type flat_basis_t

   integer, parameter :: CACHE_SLOTS = 32
   type(flat_basis_t), target, save :: cache(CACHE_SLOTS)
   integer(int64), save :: last_used(CACHE_SLOTS) = 0_int64      !! LRU clock value of the slot's last hit
   logical, save :: pinned(CACHE_SLOTS) = .false.                !! in use by the call in flight: not evictable
   integer(int64), save :: lru_clock = 0_int64
   integer, save :: cache_misses = 0

contains

   pure function cuest_backend_available() result(available)
      !! .true. -- this build has the GPU backend.  Same signature and meaning as the reference's
      !! (backends/cuest/backend/mqc_cuest_bridge.f90:20-30: "a property of the binary that linked"): it is asked by
      !! the deck reader on every rank, coordinator included (src/io/mqc_json_config_reader.f90:1149), so it must not
      !! depend on a device being visible to the asking process.  A rank without a device gets the engine's
      !! "no HIP device" error from run_cuest_scf itself.
      logical :: available
      available = .true.
   end function cuest_backend_available

   logical function hip_device_visible() result(visible)
      !! Whether a HIP device is visible to THIS process right now (not part of the reference interface)
      visible = mqc_hip_backend_available() /= 0
   end function hip_device_visible

   integer function bridge_basis_cache_misses() result(n)
      !! How many times a basis file had to be parsed and flattened (diagnostics and fortran/check_bridge.f90)
      n = cache_misses
   end function bridge_basis_cache_misses

   subroutine run_cuest_scf(settings, fragment, result, want_gradient)
      type(cuest_scf_settings_t), intent(in) :: settings
      type(physical_fragment_t), intent(in) :: fragment
      type(calculation_result_t), intent(inout) :: result
      logical, intent(in), optional :: want_gradient

      type(error_t) :: error
      type(c_ptr) :: ctx
      type(mqc_hip_molecule_t) :: mol
      type(mqc_hip_basis_t), target :: orb, aux
      type(mqc_hip_scf_options_t) :: opts
      type(mqc_hip_scf_result_t) :: res
      integer(c_int32_t), allocatable, target :: z(:)
      integer(c_int8_t), allocatable, target :: ghost(:)
      real(c_double), allocatable, target :: xyz(:), eps(:), grad(:)
      logical :: need_gradient
      integer :: rc, orb_slot, aux_slot

      need_gradient = .false.
      if (present(want_gradient)) need_gradient = want_gradient

      pinned = .false.
      ! ---- basis sets: same loader, same refusal of Cartesian sets (mqc_cuest_driver.f90:299-344)
      call flat_basis(settings%basis_set, fragment, "orbital", orb_slot, error)
      if (error%has_error()) then
         call fail(result, ERROR_VALIDATION, error%get_message()); return
      end if
      call point_at(cache(orb_slot), fragment%n_atoms, orb)
      if (settings%density_fitting) then
         call flat_basis(settings%aux_basis_set, fragment, "auxiliary", aux_slot, error)     ! cannot evict orb_slot: pinned
         if (error%has_error()) then
            pinned = .false.
            call fail(result, ERROR_VALIDATION, error%get_message()); return
         end if
         call point_at(cache(aux_slot), fragment%n_atoms, aux)
      end if

      allocate (z(fragment%n_atoms), xyz(3*fragment%n_atoms), ghost(fragment%n_atoms))
      call flatten_fragment(fragment, z, xyz, ghost, mol)

      call map_options(settings, need_gradient, opts, error)
      if (error%has_error()) then
         pinned = .false.
         call fail(result, ERROR_VALIDATION, error%get_message()); return
      end if

      ! ---- context (process-wide singleton; device = device_rank mod device_count) and the run
      rc = mqc_hip_context_get(int(settings%device_rank, c_int32_t), ctx)
      if (rc /= MQC_HIP_OK) then
         pinned = .false.
         call fail(result, ERROR_GENERIC, c_message(mqc_hip_last_error())); return
      end if
      allocate (eps(max(cache(orb_slot)%n_ao, 1)))
      call blank_result(res)
      res%orbital_energies = c_loc(eps)
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
      pinned = .false.
      if (need_gradient) then
         call unpack_result(rc, res, fragment%n_atoms, need_gradient, result, grad)
      else
         call unpack_result(rc, res, fragment%n_atoms, need_gradient, result)
      end if
   end subroutine run_cuest_scf

   subroutine run_cuest_scf_batch(settings, fragments, results, want_gradient)
      !! run_cuest_scf for many fragments in ONE engine call (mqc_hip_scf_run_batch): the engine groups them by
      !! topology and advances each group through every SCF stage in single launches.  A fragment that fails is
      !! reported in its own result (has_error, error) and does not fail the others -- what a worker gets today
      !! from calling do_fragment_work once per fragment
      !! (src/fragmentation/mbe/mqc_mbe_mpi_fragment_distribution_scheme.F90:156-238).
      type(cuest_scf_settings_t), intent(in) :: settings
      type(physical_fragment_t), intent(in) :: fragments(:)
      type(calculation_result_t), intent(inout) :: results(:)
      logical, intent(in), optional :: want_gradient

      logical :: need_gradient
      integer :: n, first, last, i
      integer, allocatable :: orb_slot(:), aux_slot(:)
      type(error_t) :: error
      integer :: needed

      need_gradient = .false.
      if (present(want_gradient)) need_gradient = want_gradient
      n = size(fragments)
      if (size(results) < n) then
         do i = 1, size(results)
            call fail(results(i), ERROR_VALIDATION, "run_cuest_scf_batch: fewer results than fragments")
         end do
         return
      end if
      allocate (orb_slot(n), aux_slot(n))
      orb_slot = 0; aux_slot = 0

      ! Sub-batches: as many consecutive fragments as the cache can pin at once (each distinct (basis, element
      ! sequence) takes one slot; an MBE list needs two or three, so in practice this loop runs once)
      first = 1
      do while (first <= n)
         pinned = .false.
         last = first - 1
         do i = first, n
            needed = merge(2, 1, settings%density_fitting)
            if (count(.not. pinned) < needed) exit          ! might need new slots and none could be evicted
            call flat_basis(settings%basis_set, fragments(i), "orbital", orb_slot(i), error)
            if (.not. error%has_error() .and. settings%density_fitting) &
               call flat_basis(settings%aux_basis_set, fragments(i), "auxiliary", aux_slot(i), error)
            if (error%has_error()) then
               call fail(results(i), ERROR_VALIDATION, error%get_message())
               orb_slot(i) = 0                              ! left out of the engine call
            end if
            last = i
         end do
         if (last < first) then
            call fail(results(first), ERROR_GENERIC, "run_cuest_scf_batch: basis cache exhausted")
            last = first
         else
            call run_sub_batch(settings, fragments(first:last), results(first:last), orb_slot(first:last), &
                               aux_slot(first:last), need_gradient)
         end if
         first = last + 1
      end do
      pinned = .false.
   end subroutine run_cuest_scf_batch

   subroutine run_sub_batch(settings, fragments, results, orb_slot, aux_slot, need_gradient)
      type(cuest_scf_settings_t), intent(in) :: settings
      type(physical_fragment_t), intent(in) :: fragments(:)
      type(calculation_result_t), intent(inout) :: results(:)
      integer, intent(in) :: orb_slot(:), aux_slot(:)
      logical, intent(in) :: need_gradient

      type(c_ptr) :: ctx
      type(mqc_hip_scf_options_t) :: opts
      type(error_t) :: error
      type(mqc_hip_molecule_t), allocatable :: mols(:)
      type(mqc_hip_basis_t), allocatable, target :: orbs(:), auxes(:)
      type(mqc_hip_scf_result_t), allocatable :: res(:)
      integer(c_int32_t), allocatable, target :: z(:)
      integer(c_int8_t), allocatable, target :: ghost(:)
      real(c_double), allocatable, target :: xyz(:), eps(:), grad(:)
      integer, allocatable :: which(:), atom_off(:), eps_off(:)
      integer :: n, m, i, k, rc, natoms_total, eps_total

      n = size(fragments)
      allocate (which(n))
      m = 0
      do i = 1, n
         if (orb_slot(i) > 0) then
            m = m + 1; which(m) = i
         end if
      end do
      if (m == 0) return

      call map_options(settings, need_gradient, opts, error)
      if (error%has_error()) then
         do k = 1, m
            call fail(results(which(k)), ERROR_VALIDATION, error%get_message())
         end do
         return
      end if
      rc = mqc_hip_context_get(int(settings%device_rank, c_int32_t), ctx)
      if (rc /= MQC_HIP_OK) then
         do k = 1, m
            call fail(results(which(k)), ERROR_GENERIC, c_message(mqc_hip_last_error()))
         end do
         return
      end if

      allocate (atom_off(m + 1), eps_off(m + 1))
      atom_off(1) = 0; eps_off(1) = 0
      do k = 1, m
         atom_off(k + 1) = atom_off(k) + fragments(which(k))%n_atoms
         eps_off(k + 1) = eps_off(k) + max(cache(orb_slot(which(k)))%n_ao, 1)
      end do
      natoms_total = atom_off(m + 1); eps_total = eps_off(m + 1)
      allocate (mols(m), orbs(m), auxes(m), res(m))
      allocate (z(natoms_total), xyz(3*natoms_total), ghost(natoms_total), eps(eps_total))
      if (need_gradient) then
         allocate (grad(3*natoms_total))
         grad = 0.0_c_double
      end if
      do k = 1, m
         i = which(k)
         call flatten_fragment(fragments(i), z(atom_off(k) + 1:atom_off(k + 1)), xyz(3*atom_off(k) + 1:3*atom_off(k + 1)), &
                               ghost(atom_off(k) + 1:atom_off(k + 1)), mols(k))
         call point_at(cache(orb_slot(i)), fragments(i)%n_atoms, orbs(k))
         if (settings%density_fitting) call point_at(cache(aux_slot(i)), fragments(i)%n_atoms, auxes(k))
         call blank_result(res(k))
         res(k)%orbital_energies = c_loc(eps(eps_off(k) + 1))
         if (need_gradient) res(k)%gradient = c_loc(grad(3*atom_off(k) + 1))
      end do

      if (settings%density_fitting) then
         rc = mqc_hip_scf_run_batch(ctx, int(m, c_int64_t), mols, orbs, c_loc(auxes), opts, res)
      else
         rc = mqc_hip_scf_run_batch(ctx, int(m, c_int64_t), mols, orbs, c_null_ptr, opts, res)
      end if
      ! per-fragment failures are in res(k); a call-level failure that left a fragment without its own message
      ! (device lost, out of memory) is that fragment's error too
      do k = 1, m
         i = which(k)
         if (need_gradient) then
            call unpack_result(merge(rc, MQC_HIP_OK, res(k)%has_error == 0 .and. res(k)%scf_status == MQC_HIP_SCF_NOT_RUN), &
                               res(k), fragments(i)%n_atoms, need_gradient, results(i), &
                               grad(3*atom_off(k) + 1:3*atom_off(k + 1)))
         else
            call unpack_result(merge(rc, MQC_HIP_OK, res(k)%has_error == 0 .and. res(k)%scf_status == MQC_HIP_SCF_NOT_RUN), &
                               res(k), fragments(i)%n_atoms, need_gradient, results(i))
         end if
      end do
   end subroutine run_sub_batch

   subroutine flatten_fragment(fragment, z, xyz, ghost, mol)
      !! physical_fragment_t -> mqc_hip_molecule_t over caller-owned arrays (which must outlive the engine call)
      type(physical_fragment_t), intent(in) :: fragment
      integer(c_int32_t), intent(out), target, contiguous :: z(:)
      real(c_double), intent(out), target, contiguous :: xyz(:)
      integer(c_int8_t), intent(out), target, contiguous :: ghost(:)
      type(mqc_hip_molecule_t), intent(out) :: mol
      z = fragment%element_numbers(1:fragment%n_atoms)
      xyz = reshape(fragment%coordinates(:, 1:fragment%n_atoms), [3*fragment%n_atoms])   ! (3,n) column-major == atom-major
      mol%n_atoms = fragment%n_atoms; mol%atomic_numbers = c_loc(z); mol%xyz = c_loc(xyz)
      mol%charge = fragment%charge; mol%multiplicity = fragment%multiplicity; mol%nelec = fragment%nelec
      mol%ghost = c_null_ptr
      mol%n_point_charges = 0; mol%point_charge_xyz = c_null_ptr; mol%point_charges = c_null_ptr
      mol%h_extra = c_null_ptr
      ghost = 0_c_int8_t
      if (allocated(fragment%is_ghost)) then
         ghost = merge(1_c_int8_t, 0_c_int8_t, fragment%is_ghost(1:fragment%n_atoms))
         mol%ghost = c_loc(ghost)
      end if
   end subroutine flatten_fragment

   subroutine map_options(settings, need_gradient, opts, error)
      !! cuest_scf_settings_t -> mqc_hip_scf_options_t (the fields run_cuest_scf acts on, mqc_cuest_driver.f90:93-135)
      type(cuest_scf_settings_t), intent(in) :: settings
      logical, intent(in) :: need_gradient
      type(mqc_hip_scf_options_t), intent(out) :: opts
      type(error_t), intent(out) :: error
      integer :: i
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
         call error%set(ERROR_VALIDATION, "initial guess '"//trim(settings%guess)// &
                        "' is not available on the HIP backend")
         return
      end select
      opts%unrestricted = merge(1, 0, settings%unrestricted)
      opts%want_gradient = merge(1, 0, need_gradient)
      opts%allow_crap_scf = merge(1, 0, settings%allow_crap_scf)
      opts%verbose = merge(1, 0, settings%verbose)
   end subroutine map_options

   subroutine blank_result(res)
      type(mqc_hip_scf_result_t), intent(out) :: res
      res%orbital_energies = c_null_ptr; res%density = c_null_ptr
      res%orbital_energies_beta = c_null_ptr
      res%gradient = c_null_ptr
      res%embedding_matrix = c_null_ptr; res%mulliken_charges = c_null_ptr
      res%has_error = 0; res%scf_status = MQC_HIP_SCF_NOT_RUN; res%iterations = 0
      res%has_dipole = 0; res%has_gradient = 0; res%has_orbitals = 0
      res%message = c_null_char
   end subroutine blank_result

   subroutine unpack_result(rc, res, n_atoms, need_gradient, result, grad)
      !! what run_cuest_scf writes into calculation_result_t (mqc_cuest_driver.f90:211-275)
      integer, intent(in) :: rc
      type(mqc_hip_scf_result_t), intent(in) :: res
      integer, intent(in) :: n_atoms
      logical, intent(in) :: need_gradient
      type(calculation_result_t), intent(inout) :: result
      real(c_double), intent(in), optional :: grad(:)

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
      if (need_gradient .and. res%has_gradient /= 0 .and. present(grad)) then
         if (allocated(result%gradient)) deallocate (result%gradient)
         allocate (result%gradient(3, n_atoms))
         result%gradient = reshape(grad(1:3*n_atoms), [3, n_atoms])
         result%has_gradient = .true.
      end if
   end subroutine unpack_result

   subroutine flat_basis(basis_name, fragment, role, slot, error)
      !! load_basis (mqc_cuest_driver.f90:299-344) + flattening, cached per (name, element sequence).  The slot
      !! returned is PINNED until the caller clears `pinned`; a miss takes the least recently used unpinned slot.
      character(len=*), intent(in) :: basis_name
      type(physical_fragment_t), intent(in) :: fragment
      character(len=*), intent(in) :: role
      integer, intent(out) :: slot
      type(error_t), intent(out) :: error

      type(molecular_basis_type) :: basis
      type(flat_basis_t), pointer :: flat
      character(len=:), allocatable :: path
      character(len=2), allocatable :: symbols(:)
      integer :: s, iatom, ish, nsh, nprim, off

      slot = 0
      if (len_trim(basis_name) == 0) then
         call error%set(ERROR_VALIDATION, "No basis set specified")
         return
      end if
      lru_clock = lru_clock + 1_int64
      do s = 1, CACHE_SLOTS
         if (.not. allocated(cache(s)%z)) cycle
         if (trim(cache(s)%name) /= trim(basis_name)) cycle
         if (size(cache(s)%z) /= fragment%n_atoms) cycle
         if (any(cache(s)%z /= fragment%element_numbers(1:fragment%n_atoms))) cycle
         slot = s
         last_used(s) = lru_clock
         pinned(s) = .true.
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

      ! victim: an empty slot, else the least recently used one that the call in flight does not point at
      do s = 1, CACHE_SLOTS
         if (pinned(s)) cycle
         if (.not. allocated(cache(s)%z)) then
            slot = s; exit
         end if
         if (slot == 0) then
            slot = s
         else if (last_used(s) < last_used(slot)) then
            slot = s
         end if
      end do
      if (slot == 0) then
         call error%set(ERROR_GENERIC, "basis cache: every slot is in use by the call in flight")
         return
      end if
      cache_misses = cache_misses + 1
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
      last_used(slot) = lru_clock
      pinned(slot) = .true.
   end subroutine flat_basis

   subroutine point_at(flat, n_atoms, pod)
      !! the POD of include/mqc_hip.h over a cached flattened basis
      type(flat_basis_t), intent(in), target :: flat
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
