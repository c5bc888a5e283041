!! The batched worker protocol of fortran/patches/node_worker_batch.patch, run in one process: a coordinator that hands
!! out tasks from a size-sorted queue and remembers what each worker holds in an outstanding_fifo_t, and two workers that
!! collect up to `capacity` tasks in a fragment_batch_t, flush them through run_cuest_scf_batch and send the results
!! back in order.  Checks: every task gets exactly one result, results are matched to the right task (each task's
!! geometry is unique, so its energy is), and with a GPU the energies equal single run_cuest_scf calls.
program check_node_worker
   use, intrinsic :: iso_fortran_env, only: real64, int64
   use mqc_hip_node_worker, only: outstanding_fifo_t, fragment_batch_t
   use mqc_cuest_bridge, only: run_cuest_scf, hip_device_visible
   use mqc_cuest_iface, only: cuest_scf_settings_t
   use mqc_physical_fragment, only: physical_fragment_t
   use mqc_result_types, only: calculation_result_t
   implicit none
   integer, parameter :: dp = real64, NTASK = 23, NWORKER = 2, CAPACITY = 5
   type(cuest_scf_settings_t) :: settings
   type(outstanding_fifo_t) :: held(NWORKER)
   type(fragment_batch_t) :: batch(NWORKER)
   type(physical_fragment_t) :: frag
   type(calculation_result_t), allocatable :: flushed(:)
   type(calculation_result_t) :: single
   type(calculation_result_t) :: results(NTASK)
   logical :: got(NTASK), gpu
   integer :: next_task, w, k, n_done, n_fail, rounds
   integer(int64) :: task
   real(dp) :: worst

   gpu = hip_device_visible()
   settings%basis_set = "sto-3g"; settings%guess = "gwh"
   settings%energy_tol = 1.0e-10_dp; settings%density_tol = 1.0e-8_dp
   got = .false.; n_fail = 0; next_task = 1; rounds = 0
   do w = 1, NWORKER
      call batch(w)%init(CAPACITY)
   end do
   do while (next_task <= NTASK .or. held(1)%length() + held(2)%length() > 0)
      rounds = rounds + 1
      do w = 1, NWORKER
         ! the worker asks until its batch is full or the queue is dry (TAG_WORKER_REQUEST / TAG_WORKER_FRAGMENT | FINISH)
         do while (.not. batch(w)%is_full() .and. next_task <= NTASK)
            call make_task(next_task, frag)
            call batch(w)%add(int(next_task, int64), frag)
            call held(w)%push(int(next_task, int64))              ! coordinator: handed out, result pending
            next_task = next_task + 1
         end do
         call batch(w)%flush(settings, .false., flushed, n_done)
         do k = 1, n_done                                          ! results go back in the order of collection
            task = held(w)%pop()
            if (task < 1 .or. task > NTASK) then
               n_fail = n_fail + 1; cycle
            end if
            if (got(task)) n_fail = n_fail + 1
            got(task) = .true.
            results(task) = flushed(k)
         end do
      end do
   end do
   call check("every task has exactly one result", all(got) .and. n_fail == 0)
   call check("nothing is left outstanding", held(1)%length() == 0 .and. held(2)%length() == 0)
   call check("the batches were used (fewer rounds than tasks)", rounds <= (NTASK + NWORKER*CAPACITY - 1)/(NWORKER*CAPACITY) + 1)
   if (gpu) then
      worst = 0.0_dp
      do k = 1, NTASK
         call make_task(k, frag)
         call run_cuest_scf(settings, frag, single)
         if (.not. (single%has_energy .and. results(k)%has_energy)) then
            n_fail = n_fail + 1
         else
            worst = max(worst, abs(single%energy%scf - results(k)%energy%scf))
         end if
      end do
      call check("batched results == single calls, task by task (1e-10)", worst < 1.0e-10_dp)
      print "(a,f20.12)", "ENERGY task1 ", results(1)%energy%scf
   else
      call check("no device: every task failed loudly", all([(results(k)%has_error .and. .not. results(k)%has_energy, k=1, NTASK)]))
   end if
   print "(a,i0)", "SUMMARY failures ", n_fail
   if (n_fail > 0) error stop 1

contains

   subroutine check(what, ok)
      character(len=*), intent(in) :: what
      logical, intent(in) :: ok
      if (ok) then
         print "(a,a)", "CHECK PASS ", what
      else
         print "(a,a)", "CHECK FAIL ", what
         n_fail = n_fail + 1
      end if
   end subroutine check

   subroutine make_task(k, f)
      !! task k: a water (odd k) or a hydrogen molecule (even k) whose geometry depends on k
      integer, intent(in) :: k
      type(physical_fragment_t), intent(out) :: f
      real(dp) :: stretch
      stretch = 1.0_dp + 0.01_dp*k
      if (mod(k, 2) == 1) then
         f%n_atoms = 3
         allocate (f%element_numbers(3), f%coordinates(3, 3))
         f%element_numbers = [8, 1, 1]
         f%coordinates(:, 1) = [0.0_dp, 0.0_dp, -0.1364652_dp]
         f%coordinates(:, 2) = [0.0_dp, 1.4304924_dp*stretch, 1.0826636_dp]
         f%coordinates(:, 3) = [0.0_dp, -1.4304924_dp*stretch, 1.0826636_dp]
         f%nelec = 10
      else
         f%n_atoms = 2
         allocate (f%element_numbers(2), f%coordinates(3, 2))
         f%element_numbers = 1
         f%coordinates = 0.0_dp
         f%coordinates(3, 2) = 1.4_dp*stretch
         f%nelec = 2
      end if
   end subroutine make_task

end program check_node_worker
