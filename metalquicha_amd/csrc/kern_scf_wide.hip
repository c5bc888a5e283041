// kern_scf_wide.hip -- the SCF-step kernels of kern_scf.hip built with 512 threads per fragment (see the note at the top of
// that file): launch_orthogonalizer_wide, launch_guess_wide, launch_scf_step_wide for batches that leave CUs idle.
#define MQC_SCF_NT 512
#include "kern_scf.hip"
