!! Runs the drop-in bridge (fortran/mqc_hip_bridge.f90, module mqc_cuest_bridge) from the Fortran side, the way
!! hf_run / dft_run call it, and checks what comes back:
!!   1. H2O / STO-3G (the inline table of check_rhf.f90:149-177) at the geometry of the reference's validation/check_rhf.f90:112-116 -> -74.9658162796 (+- 1e-9),
!!      the golden of validation/check_rhf.f90:142;
!!   2. the same call again: served from the bridge's shell cache (the basis reader is not called, no cache miss),
!!      same energy;
!!   3. want_gradient: a 3 x n_atoms gradient whose rows sum to zero (translational invariance);
!!   4. density fitting: the auxiliary basis takes the same path (cc-pVDZ + this repository's even-tempered set);
!!   5. cache stress: 18 distinct element sequences with density fitting on (36 cache entries > 32 slots), cycled
!!      twice -- evictions happen, never of an entry the call in flight uses, second-cycle energies equal the first's;
!!   6. run_cuest_scf_batch: several fragments of mixed topology in one engine call == the single calls, and a
!!      fragment that cannot run (an element the basis file lacks) fails alone.
!! Without a HIP device the first call must come back with the engine's "no HIP device" error (there is no CPU
!! fallback); the program reports that and stops with status 0.  Any failed check ends in `error stop 1`.
program check_bridge
   use, intrinsic :: iso_fortran_env, only: real64
   use mqc_cuest_bridge, only: run_cuest_scf, run_cuest_scf_batch, cuest_backend_available, hip_device_visible, &
                               bridge_basis_cache_misses
   use mqc_json_basis_reader, only: stub_reader_calls
   use mqc_cuest_iface, only: cuest_scf_settings_t
   use mqc_physical_fragment, only: physical_fragment_t
   use mqc_result_types, only: calculation_result_t, SCF_CONVERGED
   implicit none
   integer, parameter :: dp = real64
   real(dp), parameter :: GOLDEN_H2O_STO3G = -74.9658162796_dp      ! validation/check_rhf.f90:142
   type(cuest_scf_settings_t) :: settings
   type(physical_fragment_t) :: water
   type(calculation_result_t) :: res, res2
   integer :: n_fail, reads_before, misses_before
   real(dp) :: e_first

   n_fail = 0
   print "(a,l1)", "backend available (property of the binary) ", cuest_backend_available()
   print "(a,l1)", "device visible ", hip_device_visible()
   call make_water(water, 0.0_dp, [0.0_dp, 0.0_dp, 0.0_dp])
   settings%basis_set = "sto-3g-check_rhf"     ! the eight-digit STO-3G written inline in validation/check_rhf.f90:149-177
   settings%guess = "gwh"
   settings%energy_tol = 1.0e-10_dp; settings%density_tol = 1.0e-8_dp

   call run_cuest_scf(settings, water, res)
   if (.not. hip_device_visible()) then
      call check("no device: the call fails loudly", res%has_error .and. .not. res%has_energy, res%error%get_message())
      call check("no device: message names the device", index(res%error%get_message(), "HIP device") > 0, "")
      print "(a,i0)", "SUMMARY no_device failures ", n_fail
      if (n_fail > 0) error stop 1
      stop
   end if

   ! ---- 1. the reference's check_rhf golden through the Fortran bridge
   call check("h2o sto-3g runs", res%has_energy .and. .not. res%has_error, res%error%get_message())
   print "(a,f20.12,a,i0)", "ENERGY h2o_sto3g ", res%energy%scf, " iterations ", res%scf_iterations
   call check("h2o sto-3g == check_rhf golden (1e-9)", abs(res%energy%scf - GOLDEN_H2O_STO3G) < 1.0e-9_dp, "")
   call check("h2o sto-3g converged", res%scf_status == SCF_CONVERGED, "")
   call check("h2o sto-3g dipole delivered", res%has_dipole, "")
   e_first = res%energy%scf

   ! ---- 2. second call: from the shell cache
   reads_before = stub_reader_calls; misses_before = bridge_basis_cache_misses()
   call run_cuest_scf(settings, water, res2)
   call check("second call: basis reader not called", stub_reader_calls == reads_before, "")
   call check("second call: no cache miss", bridge_basis_cache_misses() == misses_before, "")
   call check("second call: same energy", res2%has_energy .and. abs(res2%energy%scf - e_first) < 1.0e-12_dp, "")

   call gradient_check()
   call density_fitting_check()
   call cache_stress()
   call batch_check()

   print "(a,i0)", "SUMMARY failures ", n_fail
   if (n_fail > 0) error stop 1

contains

   subroutine check(what, ok, detail)
      character(len=*), intent(in) :: what, detail
      logical, intent(in) :: ok
      if (ok) then
         print "(a,a)", "CHECK PASS ", what
      else
         print "(a,a,a,a)", "CHECK FAIL ", what, " -- ", trim(detail)
         n_fail = n_fail + 1
      end if
   end subroutine check

   subroutine make_water(frag, angle, shift)
      !! check_rhf's water (Bohr), turned about z by `angle` and moved by `shift`
      type(physical_fragment_t), intent(out) :: frag
      real(dp), intent(in) :: angle, shift(3)
      real(dp) :: base(3, 3), c, s
      integer :: a
      base(:, 1) = [0.0_dp, 0.0_dp, -0.1364652_dp]
      base(:, 2) = [0.0_dp, 1.4304924_dp, 1.0826636_dp]
      base(:, 3) = [0.0_dp, -1.4304924_dp, 1.0826636_dp]
      c = cos(angle); s = sin(angle)
      frag%n_atoms = 3
      allocate (frag%element_numbers(3), frag%coordinates(3, 3))
      frag%element_numbers = [8, 1, 1]
      do a = 1, 3
         frag%coordinates(1, a) = c*base(1, a) - s*base(2, a) + shift(1)
         frag%coordinates(2, a) = s*base(1, a) + c*base(2, a) + shift(2)
         frag%coordinates(3, a) = base(3, a) + shift(3)
      end do
      frag%nelec = 10
   end subroutine make_water

   subroutine make_h_chain(frag, n)
      !! n hydrogens on a line, bond lengths alternating 1.4 / 2.6 Bohr (a dimerised chain: a gap, a quick SCF)
      type(physical_fragment_t), intent(out) :: frag
      integer, intent(in) :: n
      integer :: a
      real(dp) :: x
      frag%n_atoms = n
      allocate (frag%element_numbers(n), frag%coordinates(3, n))
      frag%element_numbers = 1
      frag%coordinates = 0.0_dp
      x = 0.0_dp
      do a = 1, n
         frag%coordinates(3, a) = x
         x = x + merge(1.4_dp, 2.6_dp, mod(a, 2) == 1)
      end do
      frag%nelec = n
   end subroutine make_h_chain

   subroutine gradient_check()
      type(calculation_result_t) :: r
      call run_cuest_scf(settings, water, r, want_gradient=.true.)
      call check("gradient delivered", r%has_gradient .and. allocated(r%gradient), r%error%get_message())
      if (.not. allocated(r%gradient)) return
      call check("gradient shape (3, n_atoms)", size(r%gradient, 1) == 3 .and. size(r%gradient, 2) == 3, "")
      call check("gradient rows sum to zero (1e-8)", maxval(abs(sum(r%gradient, dim=2))) < 1.0e-8_dp, "")
      call check("gradient is not zero", maxval(abs(r%gradient)) > 1.0e-3_dp, "")
      print "(a,3es16.8)", "GRADIENT O ", r%gradient(:, 1)
   end subroutine gradient_check

   subroutine density_fitting_check()
      type(cuest_scf_settings_t) :: st
      type(calculation_result_t) :: exact, fitted
      st = settings
      st%basis_set = "cc-pvdz"
      call run_cuest_scf(st, water, exact)
      st%density_fitting = .true.
      st%aux_basis_set = "mqc-even-tempered-jkfit"
      call run_cuest_scf(st, water, fitted)
      call check("cc-pvdz exact runs", exact%has_energy, exact%error%get_message())
      call check("cc-pvdz density-fitted runs (auxiliary basis through the bridge)", fitted%has_energy, fitted%error%get_message())
      if (.not. (exact%has_energy .and. fitted%has_energy)) return
      print "(a,f20.12)", "ENERGY h2o_ccpvdz ", exact%energy%scf
      print "(a,f20.12)", "ENERGY h2o_ccpvdz_df ", fitted%energy%scf
      ! check_df.f90:55-61: exact -76.0220988827 at this geometry; the fit moves it by micro-Hartrees, not more
      call check("cc-pvdz exact == check_df's exact golden (1e-9)", abs(exact%energy%scf - (-76.0220988827_dp)) < 1.0e-9_dp, "")
      call check("fitting error is small but not zero", abs(fitted%energy%scf - exact%energy%scf) < 1.0e-3_dp .and. &
                 abs(fitted%energy%scf - exact%energy%scf) > 1.0e-8_dp, "")
      st%aux_basis_set = "no-such-set"
      call run_cuest_scf(st, water, fitted)
      call check("a missing auxiliary file is a validation error, not a crash", fitted%has_error .and. .not. fitted%has_energy, "")
   end subroutine density_fitting_check

   subroutine cache_stress()
      integer, parameter :: NSEQ = 18
      type(cuest_scf_settings_t) :: st
      type(physical_fragment_t) :: chain
      type(calculation_result_t) :: r
      real(dp) :: e(NSEQ, 2)
      integer :: cycle_no, k, misses0
      logical :: all_ran
      st = settings
      st%basis_set = "sto-3g"
      st%density_fitting = .true.
      st%aux_basis_set = "mqc-even-tempered-jkfit"
      st%energy_tol = 1.0e-9_dp; st%density_tol = 1.0e-7_dp
      misses0 = bridge_basis_cache_misses()
      all_ran = .true.
      e = 0.0_dp
      do cycle_no = 1, 2
         do k = 1, NSEQ
            call make_h_chain(chain, 2*k)
            call run_cuest_scf(st, chain, r)
            if (.not. r%has_energy) then
               all_ran = .false.
               print "(a,i0,a,a)", "  chain ", 2*k, " failed: ", r%error%get_message()
            else
               e(k, cycle_no) = r%energy%scf
            end if
         end do
      end do
      call check("cache stress: every chain ran in both cycles", all_ran, "")
      call check("cache stress: evictions happened (more entries than slots)", bridge_basis_cache_misses() - misses0 > 36, "")
      call check("cache stress: second cycle == first cycle (1e-10)", maxval(abs(e(:, 1) - e(:, 2))) < 1.0e-10_dp, "")
      print "(a,f20.12,a,f20.12)", "ENERGY h2_sto3g_df ", e(1, 1), "  h36 ", e(NSEQ, 1)
   end subroutine cache_stress

   subroutine batch_check()
      integer, parameter :: NB = 7
      type(physical_fragment_t) :: frags(NB)
      type(calculation_result_t) :: batch(NB), single
      real(dp) :: worst
      integer :: k
      logical :: ok
      call make_water(frags(1), 0.3_dp, [0.0_dp, 0.0_dp, 0.0_dp])
      call make_h_chain(frags(2), 2)
      call make_water(frags(3), 1.1_dp, [3.0_dp, -1.0_dp, 0.5_dp])
      call make_h_chain(frags(4), 4)
      call make_water(frags(5), 2.0_dp, [-2.0_dp, 0.0_dp, 9.0_dp])
      ! a fragment that cannot run: neon is not in the basis files written for this check
      frags(6)%n_atoms = 1
      allocate (frags(6)%element_numbers(1), frags(6)%coordinates(3, 1))
      frags(6)%element_numbers = 10; frags(6)%coordinates = 0.0_dp; frags(6)%nelec = 10
      call make_water(frags(7), 0.0_dp, [0.0_dp, 0.0_dp, 0.0_dp])
      call run_cuest_scf_batch(settings, frags, batch)
      worst = 0.0_dp
      ok = .true.
      do k = 1, NB
         if (k == 6) cycle
         call run_cuest_scf(settings, frags(k), single)
         if (.not. (batch(k)%has_energy .and. single%has_energy)) then
            ok = .false.
            print "(a,i0,a,a)", "  batch fragment ", k, ": ", batch(k)%error%get_message()
            cycle
         end if
         worst = max(worst, abs(batch(k)%energy%scf - single%energy%scf))
         if (batch(k)%scf_iterations /= single%scf_iterations) ok = .false.
      end do
      call check("batch: every valid fragment ran with the single call's iteration count", ok, "")
      call check("batch: energies == single calls (1e-10)", worst < 1.0e-10_dp, "")
      call check("batch: the last water is the golden again", abs(batch(7)%energy%scf - GOLDEN_H2O_STO3G) < 1.0e-9_dp, "")
      call check("batch: the fragment without a basis fails alone", batch(6)%has_error .and. .not. batch(6)%has_energy, "")
      call run_cuest_scf_batch(settings, frags, batch, want_gradient=.true.)
      ok = batch(1)%has_gradient .and. batch(4)%has_gradient
      if (ok) ok = size(batch(4)%gradient, 2) == 4 .and. maxval(abs(sum(batch(4)%gradient, dim=2))) < 1.0e-8_dp
      call check("batch: gradients delivered per fragment", ok, "")
   end subroutine batch_check

end program check_bridge
