!! Stand-ins of the metalquicha modules fortran/mqc_hip_bridge.f90 uses, so that the bridge can be compiled with flang
!! AND RUN inside this repository (fortran/check_bridge.sh).  They restate the names, kinds and argument lists of the
!! reference's public entities and nothing of their implementation; the one stand-in that does work is the basis
!! reader, which reads this repository's own flat text files (fortran/make_flat_basis.py writes them from
!! metalquicha_amd/basis_data/*.json: one shell per contraction row, SP shells split, raw coefficients -- what
!! build_molecular_basis_json delivers) instead of Basis Set Exchange JSON.
!! Inside metalquicha's tree the bridge compiles against the real modules and these files are not used.
!!   pic_types                dp                                   (pic library, fpm.toml:29)
!!   mqc_error                error_t, ERROR_*                     src/utils/mqc_error.f90:12-44
!!   mqc_cgto                 cgto_type, atomic_basis_type, molecular_basis_type   src/basis/mqc_cgto.f90:24-74
!!   mqc_basis_utils          find_basis_file                      src/basis/mqc_basis_utils.F90:148
!!   mqc_json_basis_reader    build_molecular_basis_json           src/basis/mqc_json_basis_reader.f90:332
!!   mqc_elements             element_number_to_symbol             src/core/mqc_elements.f90:92
!!   mqc_physical_fragment    physical_fragment_t                  src/fragmentation/common/mqc_physical_fragment.f90:45-94
!!   mqc_result_types         calculation_result_t, energy_t, SCF_* src/core/mqc_result_types.f90:92-196
!!   mqc_cuest_iface          cuest_scf_settings_t                 src/methods/mqc_cuest_iface.f90:35-142
module pic_types
   use, intrinsic :: iso_fortran_env, only: real64, int64
   implicit none
   integer, parameter :: dp = real64
end module pic_types

module mqc_error
   implicit none
   integer, parameter :: SUCCESS = 0, ERROR_GENERIC = 1, ERROR_IO = 2, ERROR_PARSE = 3, ERROR_VALIDATION = 4
   type :: error_t
      integer :: code = SUCCESS
      character(len=:), allocatable :: message
   contains
      procedure :: has_error => error_has_error
      procedure :: set => error_set
      procedure :: get_message => error_get_message
   end type error_t
contains
   logical function error_has_error(this)
      class(error_t), intent(in) :: this
      error_has_error = this%code /= SUCCESS
   end function
   subroutine error_set(this, code, message)
      class(error_t), intent(inout) :: this
      integer, intent(in) :: code
      character(len=*), intent(in) :: message
      this%code = code; this%message = message
   end subroutine
   function error_get_message(this) result(msg)
      class(error_t), intent(in) :: this
      character(len=:), allocatable :: msg
      msg = ""
      if (allocated(this%message)) msg = this%message
   end function
end module mqc_error

module mqc_cgto
   use pic_types, only: dp
   implicit none
   type :: cgto_type
      integer :: ang_mom
      integer :: nfunc
      real(dp), allocatable :: exponents(:)
      real(dp), allocatable :: coefficients(:)
   end type cgto_type
   type :: atomic_basis_type
      character(len=:), allocatable :: element
      type(cgto_type), allocatable :: shells(:)
      integer :: nshells = 0
      integer :: angular_form = 0
   end type atomic_basis_type
   type :: molecular_basis_type
      type(atomic_basis_type), allocatable :: elements(:)
      integer :: nelements = 0
      integer :: angular_form = 0
   contains
      procedure :: destroy => basis_set_destroy
      procedure :: is_cartesian => molecular_basis_is_cartesian
   end type molecular_basis_type
contains
   subroutine basis_set_destroy(self)
      class(molecular_basis_type), intent(inout) :: self
      if (allocated(self%elements)) deallocate (self%elements)
   end subroutine
   pure logical function molecular_basis_is_cartesian(self)
      class(molecular_basis_type), intent(in) :: self
      molecular_basis_is_cartesian = self%angular_form == 2
   end function
end module mqc_cgto

module mqc_basis_utils
   use mqc_error, only: error_t, ERROR_IO
   implicit none
contains
   subroutine find_basis_file(basis_name, filename, error)
      !! <name>.flat in $MQC_FLAT_BASIS_PATH (default: fortran/_build/basis next to the working directory)
      character(len=*), intent(in) :: basis_name
      character(len=:), allocatable, intent(out) :: filename
      type(error_t), intent(out) :: error
      character(len=1024) :: dir
      integer :: length, status
      logical :: there
      call get_environment_variable("MQC_FLAT_BASIS_PATH", dir, length, status)
      if (status /= 0 .or. length == 0) then
         dir = "_build/basis"; length = len_trim(dir)
      end if
      filename = dir(1:length)//"/"//trim(basis_name)//".flat"
      inquire (file=filename, exist=there)
      if (.not. there) call error%set(ERROR_IO, "basis set file not found: "//filename)
   end subroutine
end module mqc_basis_utils

module mqc_json_basis_reader
   use pic_types, only: dp
   use mqc_error, only: error_t, ERROR_PARSE
   use mqc_cgto, only: molecular_basis_type, atomic_basis_type
   implicit none
   integer, save :: stub_reader_calls = 0        !! how many times a file was read (fortran/check_bridge.f90 watches it)
contains
   subroutine build_molecular_basis_json(json_path, element_symbols, mol_basis, error)
      !! Flat-file stand-in of the reference's reader (src/basis/mqc_json_basis_reader.f90:332): same arguments, same
      !! result type.  File: "nelements", then per element "symbol nshells", per shell "l nprim" and nprim lines
      !! "exponent coefficient".  An element the file lacks comes back with nshells = 0, as the reference does.
      character(len=*), intent(in) :: json_path
      character(len=*), intent(in) :: element_symbols(:)
      type(molecular_basis_type), intent(out) :: mol_basis
      type(error_t), intent(out) :: error
      type(atomic_basis_type), allocatable :: table(:)
      integer :: unit, ios, nel, ie, ish, ip, ia
      character(len=2) :: sym
      stub_reader_calls = stub_reader_calls + 1
      open (newunit=unit, file=json_path, status="old", action="read", iostat=ios)
      if (ios /= 0) then
         call error%set(ERROR_PARSE, "cannot open "//json_path); return
      end if
      read (unit, *, iostat=ios) nel
      if (ios /= 0) then
         call error%set(ERROR_PARSE, "bad header in "//json_path); close (unit); return
      end if
      allocate (table(nel))
      do ie = 1, nel
         read (unit, *, iostat=ios) sym, table(ie)%nshells
         if (ios /= 0) then
            call error%set(ERROR_PARSE, "bad element record in "//json_path); close (unit); return
         end if
         table(ie)%element = trim(sym)
         allocate (table(ie)%shells(table(ie)%nshells))
         do ish = 1, table(ie)%nshells
            read (unit, *, iostat=ios) table(ie)%shells(ish)%ang_mom, table(ie)%shells(ish)%nfunc
            if (ios /= 0) then
               call error%set(ERROR_PARSE, "bad shell record in "//json_path); close (unit); return
            end if
            allocate (table(ie)%shells(ish)%exponents(table(ie)%shells(ish)%nfunc), &
                      table(ie)%shells(ish)%coefficients(table(ie)%shells(ish)%nfunc))
            do ip = 1, table(ie)%shells(ish)%nfunc
               read (unit, *, iostat=ios) table(ie)%shells(ish)%exponents(ip), table(ie)%shells(ish)%coefficients(ip)
               if (ios /= 0) then
                  call error%set(ERROR_PARSE, "bad primitive record in "//json_path); close (unit); return
               end if
            end do
         end do
      end do
      close (unit)
      allocate (mol_basis%elements(size(element_symbols)))
      mol_basis%nelements = size(element_symbols)
      do ia = 1, size(element_symbols)
         do ie = 1, nel
            if (trim(table(ie)%element) == trim(element_symbols(ia))) then
               mol_basis%elements(ia) = table(ie)
               exit
            end if
         end do
      end do
   end subroutine
end module mqc_json_basis_reader

module mqc_elements
   implicit none
contains
   pure function element_number_to_symbol(atomic_number) result(symbol)
      integer, intent(in) :: atomic_number
      character(len=2) :: symbol
      character(len=2), parameter :: table(10) = ["H ", "He", "Li", "Be", "B ", "C ", "N ", "O ", "F ", "Ne"]
      symbol = "X "
      if (atomic_number >= 1 .and. atomic_number <= 10) symbol = table(atomic_number)
   end function
end module mqc_elements

module mqc_physical_fragment
   use pic_types, only: dp
   implicit none
   type :: physical_fragment_t
      integer :: n_atoms
      integer, allocatable :: element_numbers(:)
      real(dp), allocatable :: coordinates(:, :)
      integer :: charge = 0
      integer :: multiplicity = 1
      integer :: nelec = 0
      logical, allocatable :: is_ghost(:)
      real(dp) :: distance = 0.0_dp
   end type physical_fragment_t
end module mqc_physical_fragment

module mqc_result_types
   use pic_types, only: dp
   use mqc_error, only: error_t
   implicit none
   integer, parameter :: SCF_NOT_RUN = 0, SCF_CONVERGED = 1, SCF_NOT_CONVERGED = 2
   type :: energy_t
      real(dp) :: scf = 0.0_dp
   end type energy_t
   type :: calculation_result_t
      type(energy_t) :: energy
      real(dp), allocatable :: gradient(:, :)
      real(dp), allocatable :: dipole(:)
      type(error_t) :: error
      logical :: has_energy = .false., has_gradient = .false., has_dipole = .false., has_error = .false.
      logical :: has_orbitals = .false.
      integer :: scf_status = SCF_NOT_RUN
      integer :: scf_iterations = 0
      real(dp) :: homo = 0.0_dp, lumo = 0.0_dp
      real(dp) :: distance = 0.0_dp
   end type calculation_result_t
end module mqc_result_types

module mqc_cuest_iface
   use pic_types, only: dp
   implicit none
   type :: cuest_scf_settings_t
      character(len=32) :: basis_set = "sto-3g"
      character(len=32) :: aux_basis_set = "def2-universal-jkfit"
      logical :: density_fitting = .false.
      integer :: grid_level = 3
      character(len=32) :: functional = ""
      logical :: spherical = .true.
      logical :: verbose = .false.
      character(len=32) :: guess = "auto"
      integer :: device_rank = 0
      logical :: unrestricted = .false.
      logical :: allow_crap_scf = .false.
      integer :: max_iter = 100
      real(dp) :: energy_tol = 1.0e-8_dp
      real(dp) :: density_tol = 1.0e-6_dp
      logical :: use_diis = .true.
      integer :: diis_size = 8
      integer :: radial_points = 75
      integer :: angular_points = 302
   end type cuest_scf_settings_t
end module mqc_cuest_iface
