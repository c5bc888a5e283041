!! ISO_C_BINDING view of include/mqc_hip.h -- the thin C ABI of the MI355X SCF engine.
!!
!! This file depends on nothing but iso_c_binding, so it compiles on its own
!! (fortran/check_link.sh builds it with flang and calls the lifecycle entry points).
!! The bridge module that metalquicha would compile in place of
!! backends/cuest/backend/mqc_cuest_bridge.f90 is fortran/mqc_hip_bridge.f90.
module mqc_hip_c
   use, intrinsic :: iso_c_binding
   implicit none
   private

   integer(c_int), parameter, public :: MQC_HIP_OK = 0
   integer(c_int), parameter, public :: MQC_HIP_ERR_VALIDATION = 1, MQC_HIP_ERR_GENERIC = 2, &
                                        MQC_HIP_ERR_NO_DEVICE = 3, MQC_HIP_ERR_UNSUPPORTED = 4, &
                                        MQC_HIP_ERR_DEVICE = 5
   integer(c_int), parameter, public :: MQC_HIP_SCF_NOT_RUN = 0, MQC_HIP_SCF_CONVERGED = 1, &
                                        MQC_HIP_SCF_NOT_CONVERGED = 2
   integer(c_int), parameter, public :: MQC_HIP_GUESS_AUTO = 0, MQC_HIP_GUESS_CORE = 1, MQC_HIP_GUESS_GWH = 2, &
                                        MQC_HIP_GUESS_SAD = 3, MQC_HIP_GUESS_SAC = 4

   type, bind(C), public :: mqc_hip_molecule_t
      integer(c_int32_t) :: n_atoms
      type(c_ptr) :: atomic_numbers      !! int32 [n_atoms]
      type(c_ptr) :: xyz                 !! double [3*n_atoms], atom-major, Bohr
      type(c_ptr) :: ghost               !! uint8 [n_atoms] or c_null_ptr
      integer(c_int32_t) :: charge
      integer(c_int32_t) :: multiplicity
      integer(c_int32_t) :: nelec
      ! ABI 3: external point charges (embedding field of the FMO / EE-MBE callers)
      integer(c_int32_t) :: n_point_charges = 0
      type(c_ptr) :: point_charge_xyz = c_null_ptr   !! double [3*n_point_charges], Bohr
      type(c_ptr) :: point_charges = c_null_ptr      !! double [n_point_charges]
      type(c_ptr) :: h_extra = c_null_ptr            !! double [n_ao*n_ao] or c_null_ptr: run_libcint_rhf's h_extra
   end type

   type, bind(C), public :: mqc_hip_basis_t
      integer(c_int32_t) :: spherical
      integer(c_int32_t) :: n_atoms
      type(c_ptr) :: nshell_per_atom     !! int64 [n_atoms]
      integer(c_int32_t) :: n_shells
      type(c_ptr) :: shell_l             !! int32 [n_shells]
      type(c_ptr) :: shell_nprim         !! int32 [n_shells]
      type(c_ptr) :: exponents           !! double [sum nprim]
      type(c_ptr) :: coefficients        !! double [sum nprim], RAW BSE values
   end type

   type, bind(C), public :: mqc_hip_scf_options_t
      character(kind=c_char) :: functional(32)
      integer(c_int32_t) :: density_fitting
      integer(c_int32_t) :: grid_level
      integer(c_int32_t) :: radial_points
      integer(c_int32_t) :: angular_points
      integer(c_int32_t) :: max_iter
      real(c_double) :: energy_tol
      real(c_double) :: density_tol
      integer(c_int32_t) :: use_diis
      integer(c_int32_t) :: diis_size
      integer(c_int32_t) :: guess
      integer(c_int32_t) :: unrestricted
      integer(c_int32_t) :: want_gradient
      integer(c_int32_t) :: allow_crap_scf
      integer(c_int32_t) :: verbose
      integer(c_int32_t) :: eri_mode
      real(c_double) :: schwarz_tol
   end type

   type, bind(C), public :: mqc_hip_scf_result_t
      real(c_double) :: e_total
      real(c_double) :: e_electronic
      real(c_double) :: e_nuclear
      real(c_double) :: e_xc
      integer(c_int32_t) :: scf_status
      integer(c_int32_t) :: iterations
      integer(c_int32_t) :: n_ao
      integer(c_int32_t) :: n_mo
      integer(c_int32_t) :: n_occ
      real(c_double) :: homo
      real(c_double) :: lumo
      integer(c_int32_t) :: has_orbitals
      type(c_ptr) :: orbital_energies
      type(c_ptr) :: density
      integer(c_int32_t) :: has_error
      character(kind=c_char) :: message(256)
      ! ABI 2
      real(c_double) :: dipole(3)              !! electron-Bohr, origin = centre of nuclear charge
      integer(c_int32_t) :: has_dipole
      type(c_ptr) :: gradient                  !! optional out, double [3*n_atoms] atom-major == gradient(3,n_atoms)
      integer(c_int32_t) :: has_gradient
      type(c_ptr) :: orbital_energies_beta
      integer(c_int32_t) :: n_alpha
      integer(c_int32_t) :: n_beta
      real(c_double) :: s_squared
      ! ABI 3
      real(c_double) :: e_embedding            !! tr(D u) with the point-charge field u
      type(c_ptr) :: embedding_matrix          !! optional out, double [n_ao*n_ao]
      type(c_ptr) :: mulliken_charges          !! optional out, double [n_atoms]
   end type

   public :: mqc_hip_backend_available, mqc_hip_context_get, mqc_hip_finalize, mqc_hip_last_error, &
             mqc_hip_abi_version, mqc_hip_default_options, mqc_hip_scf_run, mqc_hip_scf_run_batch, &
             mqc_hip_coulomb_batch

   interface
      function mqc_hip_backend_available() bind(C, name="mqc_hip_backend_available") result(r)
         import :: c_int
         integer(c_int) :: r
      end function
      function mqc_hip_abi_version() bind(C, name="mqc_hip_abi_version") result(r)
         import :: c_int
         integer(c_int) :: r
      end function
      function mqc_hip_context_get(local_rank, ctx) bind(C, name="mqc_hip_context_get") result(r)
         import :: c_int, c_int32_t, c_ptr
         integer(c_int32_t), value :: local_rank
         type(c_ptr), intent(out) :: ctx
         integer(c_int) :: r
      end function
      function mqc_hip_finalize() bind(C, name="mqc_hip_finalize") result(r)
         import :: c_int
         integer(c_int) :: r
      end function
      function mqc_hip_last_error() bind(C, name="mqc_hip_last_error") result(msg)
         import :: c_ptr
         type(c_ptr) :: msg
      end function
      subroutine mqc_hip_default_options(opts) bind(C, name="mqc_hip_default_options")
         import :: mqc_hip_scf_options_t
         type(mqc_hip_scf_options_t), intent(out) :: opts
      end subroutine
      function mqc_hip_scf_run(ctx, mol, orbital, aux, opts, res) bind(C, name="mqc_hip_scf_run") result(r)
         import :: c_int, c_ptr, mqc_hip_molecule_t, mqc_hip_basis_t, mqc_hip_scf_options_t, mqc_hip_scf_result_t
         type(c_ptr), value :: ctx
         type(mqc_hip_molecule_t), intent(in) :: mol
         type(mqc_hip_basis_t), intent(in) :: orbital
         type(c_ptr), value :: aux            !! address of an mqc_hip_basis_t, or c_null_ptr
         type(mqc_hip_scf_options_t), intent(in) :: opts
         type(mqc_hip_scf_result_t), intent(inout) :: res
         integer(c_int) :: r
      end function
      function mqc_hip_scf_run_batch(ctx, n, mols, orbitals, auxes, opts, res) &
         bind(C, name="mqc_hip_scf_run_batch") result(r)
         import :: c_int, c_int64_t, c_ptr, mqc_hip_molecule_t, mqc_hip_basis_t, mqc_hip_scf_options_t, &
            mqc_hip_scf_result_t
         type(c_ptr), value :: ctx
         integer(c_int64_t), value :: n
         type(mqc_hip_molecule_t), intent(in) :: mols(*)
         type(mqc_hip_basis_t), intent(in) :: orbitals(*)
         type(c_ptr), value :: auxes
         type(mqc_hip_scf_options_t), intent(in) :: opts
         type(mqc_hip_scf_result_t), intent(inout) :: res(*)
         integer(c_int) :: r
      end function
      !! J[D] for many fragments of one topology; n_source_atoms > 0: only the (leading | source) block (local_coulomb)
      function mqc_hip_coulomb_batch(ctx, n, mols, orbital, n_source_atoms, d, j) &
         bind(C, name="mqc_hip_coulomb_batch") result(r)
         import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double, mqc_hip_molecule_t, mqc_hip_basis_t
         type(c_ptr), value :: ctx
         integer(c_int64_t), value :: n
         type(mqc_hip_molecule_t), intent(in) :: mols(*)
         type(mqc_hip_basis_t), intent(in) :: orbital
         integer(c_int32_t), value :: n_source_atoms
         real(c_double), intent(in) :: d(*)
         real(c_double), intent(inout) :: j(*)
         integer(c_int) :: r
      end function
   end interface
end module mqc_hip_c
