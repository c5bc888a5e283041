/*
 * mqc_hip.h -- C ABI of libmqc_hip.so, the MI355X (gfx950) SCF engine that drops in
 * behind metalquicha's qc_method_t where the cuEST backend sits today.
 *
 * Boundary being replaced (all paths relative to the reference tree):
 *   run_cuest_scf(settings, fragment, result, want_gradient)
 *       backends/cuest/backend/mqc_cuest_bridge.f90:32-39  (module-level Fortran ABI)
 *       backends/cuest/backend/mqc_cuest_driver.f90:37-276 (what it does)
 *   cuest_backend_available()
 *       backends/cuest/backend/mqc_cuest_bridge.f90:20-30
 *   get_cuest_context / context_create / context_destroy
 *       backends/cuest/backend/mqc_cuest_context.f90:166-306
 * The Fortran shim that binds these (fortran/mqc_hip_bridge.f90, shown in
 * INTEGRATION.md) keeps basis-file parsing and error_t on the Fortran side and hands
 * the engine plain arrays: nothing but pointers, sizes and PODs crosses this header.
 *
 * Conventions
 *   - all floating point is IEEE double; geometry in Bohr, energies in Hartree
 *   - every entry point returns an int status: 0 = ok, never aborts or throws;
 *     mqc_hip_last_error() gives the message for the calling thread's last failure
 *     (the shim turns it into result%error%set(...), mqc_cuest_driver.f90:385-393)
 *   - the caller owns every host array for the duration of the call; the engine owns
 *     all device memory (grow-only pools, released by mqc_hip_finalize), like
 *     device_pool_t in mqc_cuest_context.f90:40-53,142-156
 *   - one calling thread per process (the context is a process-wide singleton,
 *     mqc_cuest_context.f90:134-138); no re-entrancy
 *   - the engine FAILS (MQC_HIP_ERR_NO_DEVICE) when no HIP device is present: there is
 *     no CPU fallback behind this ABI.
 */
#ifndef MQC_HIP_H
#define MQC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MQC_HIP_ABI_VERSION 3

/* status codes (0 = ok).  VALIDATION mirrors ERROR_VALIDATION, GENERIC mirrors ERROR_GENERIC
 * (src/utils/mqc_error.f90:22-44). */
enum {
    MQC_HIP_OK = 0,
    MQC_HIP_ERR_VALIDATION = 1,
    MQC_HIP_ERR_GENERIC = 2,
    MQC_HIP_ERR_NO_DEVICE = 3,
    MQC_HIP_ERR_UNSUPPORTED = 4,
    MQC_HIP_ERR_DEVICE = 5
};

/* scf_status values, same meaning as the reference's tri-state
 * (backends/cuest/backend/mqc_cuest_driver.f90:219-224). */
enum { MQC_HIP_SCF_NOT_RUN = 0, MQC_HIP_SCF_CONVERGED = 1, MQC_HIP_SCF_NOT_CONVERGED = 2 };

/* initial guess (cuest_scf_settings_t%guess, src/methods/mqc_cuest_iface.f90:104-121).
 * AUTO resolves to GWH, as the cuEST backend does. */
enum { MQC_HIP_GUESS_AUTO = 0, MQC_HIP_GUESS_CORE = 1, MQC_HIP_GUESS_GWH = 2, MQC_HIP_GUESS_SAD = 3,
       MQC_HIP_GUESS_SAC = 4 };   /* AUTO = GWH, as run_cuest_scf's default (mqc_cuest_driver.f90:107-114); SAD / SAC =
                                     superposed free-atom densities, spherically averaged / as converged
                                     (mqc_libcint_atomic_guess.f90:295-378; SAC restricted only) */

/* two-electron path.  AUTO = in-core packed ERIs in HBM when they fit the per-fragment
 * budget, else the direct build (the reference's choice at mqc_libcint_bridge.f90:819-892,
 * with the 2 GB host limit replaced by an HBM budget). */
enum { MQC_HIP_ERI_AUTO = 0, MQC_HIP_ERI_INCORE = 1, MQC_HIP_ERI_DIRECT = 2 };

typedef struct mqc_hip_context mqc_hip_context;   /* opaque, process-wide singleton */

/* physical_fragment_t flattened (src/fragmentation/common/mqc_physical_fragment.f90:45-94) */
typedef struct {
    int32_t n_atoms;
    const int32_t *atomic_numbers;   /* [n_atoms] */
    const double *xyz;               /* [3*n_atoms], atom-major, Bohr */
    const uint8_t *ghost;            /* [n_atoms] or NULL; ghost = basis functions, Z = 0 */
    int32_t charge;
    int32_t multiplicity;
    int32_t nelec;                   /* excludes ghosts */
    /* ABI 3: external point charges -- the embedding field of the FMO / EE-MBE callers
     * (embedding_operator with esp = "ptc", backends/libcint/mqc_libcint_fmo.f90:1077-1160): the operator
     * u = -sum_g q_g / |r - R_g| is added to the one-electron Hamiltonian, exactly as run_libcint_rhf's h_extra
     * (mqc_libcint_rhf.f90:479-484); the charges do NOT enter the nuclear repulsion. */
    int32_t n_point_charges;         /* 0 = none */
    const double *point_charge_xyz;  /* [3*n_point_charges], Bohr */
    const double *point_charges;     /* [n_point_charges] */
    const double *h_extra;           /* [n_ao*n_ao] row-major symmetric, or NULL: any further one-electron operator added to
                                        H as it stands (run_libcint_rhf's h_extra) -- the exact Coulomb field of near
                                        fragments' electrons in FMO (local_coulomb, mqc_libcint_fmo.f90:1337-1406).
                                        Counted in e_embedding and embedding_matrix together with the charges' part. */
} mqc_hip_molecule_t;

/* molecular_basis_type flattened (src/basis/mqc_cgto.f90); RAW Basis-Set-Exchange
 * coefficients of unnormalised primitives -- the engine normalises exactly as
 * backends/libcint/mqc_libcint_integrals.F90:519-555 does. */
typedef struct {
    int32_t spherical;               /* must be 1: Cartesian sets are refused
                                        (mqc_cuest_driver.f90:331-341) */
    int32_t n_atoms;
    const int64_t *nshell_per_atom;  /* [n_atoms] */
    int32_t n_shells;                /* = sum nshell_per_atom */
    const int32_t *shell_l;          /* [n_shells] */
    const int32_t *shell_nprim;      /* [n_shells] */
    const double *exponents;         /* [sum nprim] */
    const double *coefficients;      /* [sum nprim] */
} mqc_hip_basis_t;

/* the subset of cuest_scf_settings_t the engine acts on (mqc_cuest_iface.f90:35-142) */
typedef struct {
    char functional[32];             /* "" = Hartree-Fock */
    int32_t density_fitting;
    int32_t grid_level;              /* 1..5, default 3 */
    int32_t radial_points;           /* 0 = from grid_level */
    int32_t angular_points;          /* 0 = from grid_level */
    int32_t max_iter;                /* default 100 */
    double energy_tol;               /* default 1e-8 */
    double density_tol;              /* default 1e-6 */
    int32_t use_diis;
    int32_t diis_size;               /* default 8 */
    int32_t guess;                   /* MQC_HIP_GUESS_* */
    int32_t unrestricted;            /* force UHF / UKS (open shells take it by themselves, mqc_cuest_driver.f90:127) */
    int32_t want_gradient;           /* analytic gradient into result->gradient: HF and Kohn-Sham, exact ERIs, s-d shells */
    int32_t allow_crap_scf;
    int32_t verbose;
    int32_t eri_mode;                /* MQC_HIP_ERI_* */
    double schwarz_tol;              /* 0 = default 1e-11 (mqc_libcint_direct.f90:61) */
} mqc_hip_scf_options_t;

/* what run_cuest_scf writes into calculation_result_t (mqc_cuest_driver.f90:211-275) */
typedef struct {
    double e_total;                  /* energy%scf */
    double e_electronic;
    double e_nuclear;
    double e_xc;
    int32_t scf_status;              /* MQC_HIP_SCF_* */
    int32_t iterations;
    int32_t n_ao;
    int32_t n_mo;
    int32_t n_occ;
    double homo;                     /* orbital energies, Hartree */
    double lumo;
    int32_t has_orbitals;
    double *orbital_energies;        /* optional out [n_mo capacity >= n_ao] or NULL (alpha spin when unrestricted) */
    double *density;                 /* optional out [n_ao*n_ao] row-major or NULL (total density) */
    int32_t has_error;
    char message[256];
    /* ABI 2 */
    double dipole[3];                /* result%dipole, electron-Bohr, origin = centre of nuclear charge
                                        (system_compute_dipole, mqc_cuest_integrals.f90:1443-1521) */
    int32_t has_dipole;
    double *gradient;                /* optional out [3*n_atoms] atom-major, Hartree/Bohr: result%gradient(3,n_atoms);
                                        filled when opts->want_gradient and the pointer is not NULL */
    int32_t has_gradient;
    double *orbital_energies_beta;   /* optional out [n_ao] or NULL; written by unrestricted runs */
    int32_t n_alpha, n_beta;         /* occupied orbitals per spin (n_occ = n_alpha) */
    double s_squared;                /* <S^2> of the unrestricted determinant, 0 for restricted */
    /* ABI 3 */
    double e_embedding;              /* tr(D u): the part of e_total that is the interaction with the point charges
                                        (inner_scf subtracts it, mqc_libcint_fmo.f90:1992-1997); 0 without charges */
    double *embedding_matrix;        /* optional out [n_ao*n_ao] or NULL: u itself (nmer_term needs tr(D_split u), :1266-1272) */
    double *mulliken_charges;        /* optional out [n_atoms] or NULL: q_A = Z_A - sum_{mu on A} (D S)_mu,mu
                                        (fragment_charges, mqc_libcint_fmo.f90:2001-2021; ghosts carry Z = 0) */
} mqc_hip_scf_result_t;

/* ---- lifecycle -------------------------------------------------------------------- */
/* cuest_backend_available(): 1 if a HIP device is visible, else 0 */
int mqc_hip_backend_available(void);
/* get_cuest_context(): lazy singleton; device = local_rank mod device_count
 * (mqc_cuest_context.f90:188) */
int mqc_hip_context_get(int32_t local_rank, mqc_hip_context **ctx);
int mqc_hip_finalize(void);
const char *mqc_hip_last_error(void);
int mqc_hip_abi_version(void);
void mqc_hip_default_options(mqc_hip_scf_options_t *opts);

/* ---- the hot path ----------------------------------------------------------------- */
/* run_cuest_scf for ONE fragment */
int mqc_hip_scf_run(mqc_hip_context *ctx, const mqc_hip_molecule_t *mol,
                    const mqc_hip_basis_t *orbital, const mqc_hip_basis_t *aux /* NULL unless DF */,
                    const mqc_hip_scf_options_t *opts, mqc_hip_scf_result_t *result);

/* The same for MANY fragments at once (batch-submit API).
 * Fragments are independent (do_fragment_work has no cross-fragment state,
 * src/fragmentation/mbe/mqc_mbe_mpi_fragment_distribution_scheme.F90:156-238); the engine
 * groups them by topology and advances whole groups through each SCF stage in single
 * launches.  Per-fragment failures are reported in results[i] and do not fail the call. */
int mqc_hip_scf_run_batch(mqc_hip_context *ctx, int64_t n_fragments,
                          const mqc_hip_molecule_t *mols, const mqc_hip_basis_t *orbitals,
                          const mqc_hip_basis_t *auxes /* NULL unless DF */,
                          const mqc_hip_scf_options_t *opts, mqc_hip_scf_result_t *results);

/* ---- stage-level entry points (same kernels, used by the parity tests) ------------- */
/* S, T, V (n_ao x n_ao, row-major):  compute_overlap/kinetic/potential,
 * mqc_cuest_integrals.f90:1525-1634 <-> one_electron, mqc_libcint_integrals.F90:843-911 */
int mqc_hip_int1e(mqc_hip_context *ctx, const mqc_hip_molecule_t *mol, const mqc_hip_basis_t *orbital,
                  double *S, double *T, double *V);
/* packed in-core ERI matrix M[pair(i,j)][pair(k,l)], pair(i,j) = i(i+1)/2 + j, i >= j
 * (molecule_eris, mqc_libcint_integrals.F90:1449) */
int mqc_hip_eri_packed(mqc_hip_context *ctx, const mqc_hip_molecule_t *mol, const mqc_hip_basis_t *orbital,
                       double schwarz_tol, double *M /* [npair*npair] */);
/* J[D], K[D] from the in-core tensor (build_fock, mqc_libcint_rhf.f90:1491-1574) */
int mqc_hip_jk_incore(mqc_hip_context *ctx, const mqc_hip_molecule_t *mol, const mqc_hip_basis_t *orbital,
                      const double *D, double *J, double *K);
/* symmetric eigen-decomposition by the engine's LDS Jacobi kernel: A (n x n) -> w ascending,
 * V columns = eigenvectors, row-major (diagonalize_fock_device, mqc_cuest_scf.f90:1132-1221) */
/* J[D] for MANY fragments of ONE topology (same elements and basis, n geometries, n densities) in single launches:
 * the batched form of local_coulomb (backends/libcint/mqc_libcint_fmo.f90:1337-1406), where the FMO driver needs the
 * Coulomb operator of a neighbour's density over the supersystem fragment + neighbour for every (fragment, neighbour)
 * pair of a pass.  D and J are [n][n_ao*n_ao] row-major, contiguous; in-core exact ERIs (n_ao <= 116) for the full matrix.
 * n_source_atoms > 0 says that only the LAST n_source_atoms atoms carry density and only the block of J over the
 * other (leading) atoms is wanted -- exactly local_coulomb's use -- so only the shell quartets (leading pair | source
 * pair) are formed -- contracted with the density on the fly by the direct-path digest kernels (Schwarz bound 1e-12,
 * no integral tensor, n_ao <= 256); the rest of J is then not meaningful.  0 = the full Coulomb matrix of the full density. */
int mqc_hip_coulomb_batch(mqc_hip_context *ctx, int64_t n_fragments, const mqc_hip_molecule_t *mols,
                          const mqc_hip_basis_t *orbital, int32_t n_source_atoms, const double *D, double *J);

int mqc_hip_syev(mqc_hip_context *ctx, int32_t n, const double *A, double *w, double *V);
/* DIIS coefficients from an age-ordered overlap matrix, the device routine's algorithm
 * (diis_coefficients/solve_diis, src/methods/mqc_diis.f90:164-273) */
int mqc_hip_diis_coefficients(mqc_hip_context *ctx, int32_t n_stored, const double *overlap /* [n*n] */,
                              double *coefficients /* [n_stored] */, int32_t *ok);

/* ---- introspection ----------------------------------------------------------------- */
typedef struct {
    double t_setup, t_int1e, t_eri, t_fock, t_scf_step, t_total;   /* seconds, last batch call */
    int64_t fock_launches, eri_quartets, scf_iterations_total;
    double fock_kernel_seconds;   /* HIP-event time of the in-core J/K kernel, last batch */
    double fock_bytes;            /* algorithmic bytes it streamed */
    double eri_kernel_seconds;
    double xc_kernel_seconds;     /* HIP-event time of the XC quadrature kernel, last batch */
    double xc_points;             /* grid points it integrated (fragments x points, summed over launches) */
    /* the same three J/K figures restricted to launches that streamed >= 1 GiB (the dominant kernel of a large batch) */
    int64_t fock_big_launches;
    double fock_big_seconds, fock_big_bytes;
    /* ABI 2 */
    double xc_flops;              /* 8 P n^2 per GGA launch (4 P n^2 LDA), summed: SURVEY.md section 8d's algorithmic count */
    double scf_step_seconds;      /* HIP-event time of the per-iteration SCF-step kernel */
    int64_t eri_survivors;        /* shell quartets the integral stage actually formed (Schwarz survivors, shared blocks once) */
    double df_flops, df_bytes;    /* DF J/K: 4 n^2 A (1 + o) flop and 8 n^2 A bytes per fragment-iteration, summed */
} mqc_hip_stats_t;
int mqc_hip_get_stats(mqc_hip_context *ctx, mqc_hip_stats_t *stats);
int mqc_hip_device_name(mqc_hip_context *ctx, char *buf, int32_t len);

#ifdef __cplusplus
}
#endif
#endif /* MQC_HIP_H */
