!! Worker-side and coordinator-side pieces of BATCHED fragment submission, for the patch
!! fortran/patches/node_worker_batch.patch against
!! src/fragmentation/mbe/mqc_mbe_mpi_fragment_distribution_scheme.F90 (node_worker_impl :1272-1377,
!! handle_local_worker_requests_group :794-830, handle_local_worker_results :876-928, node_coordinator_impl :1117-1254).
!!
!! Today a worker asks for ONE fragment, computes it (do_fragment_work :156-238) and sends ONE result, and the
!! coordinator remembers ONE outstanding fragment per worker (worker_fragment_map(worker), :439,:824,:919).  The engine
!! is fastest when it sees many fragments at once (mqc_hip_scf_run_batch), so the patch lets a worker ask again
!! before it computes -- up to `capacity` tasks in flight -- and flush them through run_cuest_scf_batch.  Nothing here
!! touches MPI: the two types below are the bookkeeping the patch needs on either side, and fortran/check_node_worker.f90
!! runs them against each other (with a GPU: real energies through the bridge).
!!
!!   outstanding_fifo_t  coordinator: the fragments a worker holds, in the order they were handed out.  Results of one
!!                       worker arrive in that order (same source, same tag: MPI messages do not overtake), so
!!                       push on send, pop on receive replaces the scalar map entry.
!!   fragment_batch_t    worker: collects (task index, physical_fragment_t) until full or the queue runs dry, then
!!                       flush() = one run_cuest_scf_batch call; results come back in the order of collection.
module mqc_hip_node_worker
   use, intrinsic :: iso_fortran_env, only: int64
   use mqc_cuest_iface, only: cuest_scf_settings_t
   use mqc_physical_fragment, only: physical_fragment_t
   use mqc_result_types, only: calculation_result_t
   use mqc_cuest_bridge, only: run_cuest_scf_batch
   implicit none
   private

   public :: outstanding_fifo_t, fragment_batch_t

   type :: outstanding_fifo_t
      integer(int64), allocatable :: items(:)
      integer :: head = 1, count = 0
   contains
      procedure :: push => fifo_push
      procedure :: pop => fifo_pop
      procedure :: peek => fifo_peek
      procedure :: length => fifo_length
   end type outstanding_fifo_t

   type :: fragment_batch_t
      integer :: capacity = 64
      integer :: count = 0
      integer(int64), allocatable :: task_index(:)
      type(physical_fragment_t), allocatable :: fragments(:)
   contains
      procedure :: init => batch_init
      procedure :: add => batch_add
      procedure :: is_full => batch_is_full
      procedure :: flush => batch_flush
   end type fragment_batch_t

contains

   subroutine fifo_push(this, item)
      class(outstanding_fifo_t), intent(inout) :: this
      integer(int64), intent(in) :: item
      integer(int64), allocatable :: grown(:)
      integer :: k
      if (.not. allocated(this%items)) allocate (this%items(8))
      if (this%count == size(this%items)) then
         allocate (grown(2*size(this%items)))
         do k = 1, this%count
            grown(k) = this%items(mod(this%head + k - 2, size(this%items)) + 1)
         end do
         call move_alloc(grown, this%items)
         this%head = 1
      end if
      this%items(mod(this%head + this%count - 1, size(this%items)) + 1) = item
      this%count = this%count + 1
   end subroutine fifo_push

   function fifo_pop(this) result(item)
      !! the oldest outstanding fragment; 0 when there is none (the reference's "no fragment assigned", :899-903)
      class(outstanding_fifo_t), intent(inout) :: this
      integer(int64) :: item
      item = 0_int64
      if (this%count == 0) return
      item = this%items(this%head)
      this%head = mod(this%head, size(this%items)) + 1
      this%count = this%count - 1
   end function fifo_pop

   pure function fifo_peek(this) result(item)
      !! the oldest outstanding fragment without removing it; 0 when there is none
      class(outstanding_fifo_t), intent(in) :: this
      integer(int64) :: item
      item = 0_int64
      if (this%count > 0) item = this%items(this%head)
   end function fifo_peek

   pure integer function fifo_length(this) result(n)
      class(outstanding_fifo_t), intent(in) :: this
      n = this%count
   end function fifo_length

   subroutine batch_init(this, capacity)
      class(fragment_batch_t), intent(inout) :: this
      integer, intent(in) :: capacity
      this%capacity = max(1, capacity)
      this%count = 0
      if (allocated(this%fragments)) deallocate (this%fragments)
      if (allocated(this%task_index)) deallocate (this%task_index)
      allocate (this%fragments(this%capacity), this%task_index(this%capacity))
   end subroutine batch_init

   subroutine batch_add(this, task_idx, fragment)
      class(fragment_batch_t), intent(inout) :: this
      integer(int64), intent(in) :: task_idx
      type(physical_fragment_t), intent(in) :: fragment
      if (.not. allocated(this%fragments)) call this%init(this%capacity)
      if (this%count >= this%capacity) return          ! the caller flushes when is_full()
      this%count = this%count + 1
      this%task_index(this%count) = task_idx
      this%fragments(this%count) = fragment
   end subroutine batch_add

   pure logical function batch_is_full(this) result(full)
      class(fragment_batch_t), intent(in) :: this
      full = this%count >= this%capacity
   end function batch_is_full

   subroutine batch_flush(this, settings, want_gradient, results, n_done)
      !! One engine call for everything collected; results(k) belongs to task_index(k), k = 1..n_done.  A fragment
      !! that failed carries its own error, like a do_fragment_work that returned with result%has_error.
      class(fragment_batch_t), intent(inout) :: this
      type(cuest_scf_settings_t), intent(in) :: settings
      logical, intent(in) :: want_gradient
      type(calculation_result_t), allocatable, intent(out) :: results(:)
      integer, intent(out) :: n_done
      integer :: k
      n_done = this%count
      allocate (results(max(n_done, 1)))
      if (n_done == 0) return
      call run_cuest_scf_batch(settings, this%fragments(1:n_done), results(1:n_done), want_gradient)
      do k = 1, n_done
         results(k)%distance = this%fragments(k)%distance       ! do_fragment_work :227-228
      end do
      this%count = 0
   end subroutine batch_flush

end module mqc_hip_node_worker
