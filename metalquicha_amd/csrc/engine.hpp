// engine.hpp -- internal types of the MI355X SCF engine (host side, C++17).
//
// Design in one paragraph: the unit of work is a BATCH of fragments that share a topology
// (same elements in the same order, same basis, same electron count) -- exactly what an
// MBE/GMBE fragment list is made of (all monomers alike, all dimers alike).  Every stage
// of the SCF (one-electron integrals, in-core ERI formation, J/K contraction, the linear
// algebra of an iteration) is ONE kernel launch over the whole batch, so a 2080-fragment
// (H2O)64 MBE-2 run keeps all 256 CUs busy instead of issuing thousands of
// latency-bound launches per fragment (the regime mqc_cuest_scf.f90:9-23 fights).
// A single fragment (the run_cuest_scf drop-in) is simply a batch of one.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>
#include <map>
#include <memory>
#include <mutex>
#include "../../include/mqc_hip.h"

namespace mqc {

constexpr int KERNEL_LMAX = 3;      // highest AO angular momentum: s, p, d by per-class register kernels, f by the general LDS kernel
constexpr int CLASS_LMAX = 2;       // highest l the per-class ERI / digest / 3-centre kernels are instantiated for
constexpr int AUX_LMAX = 3;         // highest angular momentum of an auxiliary (fitting) shell
constexpr int DIIS_MAX = 8;         // subspace size the device ring buffers are laid out for

// fragment state machine driven by the scf_step kernel
enum : int { ST_ITER = 0, ST_FINAL = 1, ST_DONE = 2 };

struct HostShell {
    int l, nprim, poff, atom, aoff;
};

// Everything that is identical for all fragments of a batch.
struct Topology {
    std::string key;
    int natoms = 0;
    std::vector<int> Z;
    std::vector<double> zeff;        // ghost -> 0
    int nelec = 0, charge = 0, multiplicity = 1;
    std::vector<HostShell> shells;
    std::vector<double> exps, coefs; // normalised radial coefficients (s,p angular factor folded in)
    int nao = 0, npair = 0, lmax = 0;
    // shell-quartet task lists, one per (la,lb,lc,ld) class, canonical order
    struct ClassList {
        int la, lb, lc, ld;
        std::vector<int> quartets;       // 4 ints each: every canonical quartet of the class
        // twin-shell cut of the same class (in-core ERI build, md_integrals.hpp TwinCoefs):
        //   twin_entries -- 4 ints each, FIRST member shells, bit 16 set on twin positions;
        //   rest         -- the quartets no twin entry covers.  Both empty when the class has no twin block.
        std::vector<int> twin_entries, rest;
        // per entry of the three lists: index into Topology::atom_sets (the atoms its four shells sit on)
        std::vector<int> set_quartets, set_twin, set_rest;
    };
    std::vector<std::vector<int>> atom_sets;   // distinct sorted atom sets (1..4 atoms) met by the quartets
    std::vector<int> twin_first;     // per shell: 1 when the shell and its successor form a twin s pair
    std::vector<ClassList> classes;
    std::vector<int> pairs;          // (A,B) with A>=B, 2 ints each, for the 1e kernel (la>=lb ordering)
    int64_t n_quartets = 0;
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

// grow-only device pool, the engine's device_pool_t (mqc_cuest_context.f90:40-53,142-156)
class DevicePool {
public:
    DevicePool();
    ~DevicePool();
    DevicePool(const DevicePool&) = delete;
    DevicePool& operator=(const DevicePool&) = delete;
    void* ensure(size_t bytes);
    void release();
    size_t capacity() const { return cap_; }
private:
    void* ptr_ = nullptr;
    size_t cap_ = 0;
};

// exchange-correlation components (libxc-equivalent ids, unpolarised)
enum : int { XC_LDA_X = 1, XC_LDA_C_VWN, XC_LDA_C_VWN_RPA, XC_GGA_X_B88, XC_GGA_C_LYP, XC_GGA_X_PBE, XC_GGA_C_PBE, XC_MGGA_X_TPSS, XC_MGGA_C_TPSS };
struct XcSpec {
    int ncomp = 0;
    int id[6] = {0, 0, 0, 0, 0, 0};
    double w[6] = {0, 0, 0, 0, 0, 0};
    double exx = 1.0;     // exact-exchange fraction
    int gga = 0;          // 1: needs grad rho; 2: meta-GGA, needs grad rho and tau
};

// quadrature grid of a batch: per-topology point list (atom, template point) + per-fragment weights
struct GridDev {
    int npts = 0;
    const int* pt_atom = nullptr;       // [npts]
    const int* pt_tmpl = nullptr;       // [npts] index into tmpl_xyz / tmpl_w
    const double* tmpl_xyz = nullptr;   // [ntmpl][3] relative to the nucleus
    const double* tmpl_w = nullptr;     // [ntmpl] 4 pi r^2 dr w_leb
    const double* sqrt_bragg = nullptr; // [natoms] sqrt(Bragg radius), Treutler size adjustment
    double* weights = nullptr;          // [nfrag][npts] product weight x Becke cell weight
    // radial cache (optional): per fragment, per tile of rad_pt points, per shell: rad_pt radial values then rad_pt
    // radial derivative factors -- R_s(r) = sum_k c_k exp(-a_k r^2) and sum_k -2 a_k c_k exp(-a_k r^2) of every
    // contracted shell at every grid point, formed ONCE per SCF (the geometry is fixed) instead of in every iteration
    double* rad = nullptr;              // [nfrag][ntiles][nshell][2][rad_pt] or nullptr
    int rad_pt = 0;
    // four numbers per (padded) grid point between the kernels of the split quadrature (kern_xc.hip): rho and grad rho / 2
    // out of the density kernel, overwritten by w v_rho / 2 and 2 w v_sigma grad rho in the functional kernel
    double* pt4 = nullptr;              // [nfrag][ntiles * rad_pt][4] or nullptr
};

struct TopologyDev {
    int *sh_l, *sh_nprim, *sh_poff, *sh_atom, *sh_aoff;
    double *exps, *coefs, *zeff;
    int nshell, nao, npair, natoms;
    int lmax;                           // highest angular momentum among the shells
    // radial groups for the AO evaluation on the grid (kern_xc.hip): consecutive shells of one atom and one l whose
    // primitives all belong to the first shell's list (general contraction: cc-pVDZ oxygen 1s/2s/3s share nine
    // exponents) evaluate their exponentials once.  gcoefs holds grp_count rows of grp_nprim coefficients (zeros
    // where a shell does not use a primitive).
    int *grp_first, *grp_count, *grp_nprim, *grp_poff, *grp_coff;
    double *gexps, *gcoefs;
    int ngroup;
    int gprim_total, gcoef_total;     // lengths of gexps / gcoefs (the tiled XC kernel keeps the tables in LDS)
};
constexpr int XC_GROUP_MAX = 3;      // shells per radial group

struct BatchView {      // plain pointers handed to kernels
    TopologyDev topo;
    int nfrag;
    int n, npair;                 // n = nao
    int nocc;
    double exx;                   // exact-exchange fraction (1 for HF)
    double e_tol, d_tol;
    int max_iter, diis_size;
    const double* boys;           // Boys Taylor table
    const double* c2s;            // cart->sph tables, l = 0..LMAX_AO, offsets in c2s_off
    double* xyz;                  // [nfrag][natoms][3]
    double *S, *H, *X, *F, *D, *C, *J, *K, *W;   // [nfrag][n*n]; W = [nfrag][6][n*n] workspace
    double* Vprev;                // [nfrag][n*n] last eigenvectors in the orthogonal basis (Jacobi warm start)
    double* eps;                  // [nfrag][n]
    double* eri;                  // [nfrag][eri_stride]: the pair matrix, square [npair][npair] or (eri_tri) its lower triangle
    // eri_tri = 1: only the elements col <= row are stored, in BLOCKS of eri_tri_pb doubles, one per pair of rows
    // (kern_fock.hip, jk_tri_kernel): block t = [row npair-1-t, zeros to the end of its last shell row | row t, zeros
    // likewise]; eri_tri_sb[t] = where row t starts inside its block; the diagonal elements (ij|ij) are stored HALVED
    int eri_tri, eri_tri_pb;
    const int* eri_tri_sb;        // [(npair + 1) / 2] short-row starts, then [npair] the shell row i of every pair row (ij)
    size_t eri_stride;            // doubles per fragment: npair^2, or (npair + 1) / 2 blocks
    // Schwarz bounds of the screened in-core build, for the J/K kernel: a pair row (ij) whose bound times the fragment's
    // largest bound is below the threshold was never written by any class kernel -- it is all zeros, and jk_tri_kernel
    // neither loads nor contracts it (three quarters of the dimers of an MBE list are far apart: half of their rows)
    const double* jk_q;           // [nfrag][nshell][nshell] or nullptr
    double jk_qthresh;
    int* jk_loaded;               // [nfrag] chunks of 128 doubles the last jk_tri_kernel launch read for each fragment (byte accounting)
    double *diis_f, *diis_e, *diis_b;   // [nfrag][8][n*n], [nfrag][8][n*n], [nfrag][8*8]
    int* diis_state;              // [nfrag][2] = n_stored, newest
    double* scal;                 // [nfrag][8]: e_elec, e_old, de, drms, e_final, E_xc, N_electrons, -
    int* istate;                  // [nfrag][4]: state, iterations, nmo, converged
    int* counters;                // [4]: n_not_done, ...
    unsigned long long* eri_count;   // [1]: shell quartets the integral kernels formed (Schwarz survivors), or nullptr
    double* dip;                  // [nfrag][4]: tr(D x), tr(D y), tr(D z) about the origin, -
    int npc;                      // external point charges per fragment (0 = none)
    const double* pc;             // [npc][4 = x, y, z, q][nfrag], fragment fastest (coalesced over the lanes of a wave)
    double* U;                    // [nfrag][n*n] embedding operator -sum_g q_g/|r - R_g| + h_extra (part of H), or nullptr
    const double* Hx;             // [nfrag][n*n] the caller's h_extra matrices, or nullptr
    // far field of the point charges (kern_int1e.hip, set by launch_int1e for large fields only): per atom the squared
    // distance beyond which every primitive pair on that atom sees a charge through the asymptotic Boys function, and
    // per (fragment, atom) the sums  sum_far q d^{tuv}(1/|A - C|), t + u + v <= 4, in hidx order
    const double* pc_far_r2;      // [natoms] or nullptr
    const double* pc_far_tab;     // [nfrag][natoms][35] or nullptr
    int slot;                     // 0/1: which pipeline slot (stream, pools, launcher scratch) this batch view lives in
    XcSpec xc;                    // ncomp == 0: no XC term
    GridDev grid;
    double* Vxc;                  // [nfrag][n*n] un-symmetrised accumulator A (V_xc = A + A^T), or nullptr
    // density fitting (naux == 0: exact-ERI path)
    TopologyDev aux;
    int naux;
    const double* unit;           // {0.0, 1.0}: exponent and coefficient of the unit shell
    double *df_a3, *df_b;         // [nfrag][naux][npair]: (P|mu nu) and the fitted tensor
    double *df_metric, *df_linv;  // [nfrag][naux][naux]
    double* df_work;              // [nfrag][naux][naux] scratch of the fit (saved diagonal; V^T of the eigen path)
    // unrestricted SCF (uhf != 0): D, C, F, J, K, Vprev, eps and the DIIS histories above are the ALPHA spin's,
    // these are the BETA spin's; densities are D_s = C_s,occ C_s,occ^T (no factor 2)
    int uhf, nalpha, nbeta;
    double *Db, *Cb, *Fb, *Jb, *Kb, *Vprevb, *epsb, *diis_fb, *diis_eb;
};

// Where an integral goes in the pair matrix of one fragment: both mirror elements of the square, or the one stored
// element of the triangular block layout (diagonal elements halved, see BatchView::eri_tri).
struct PairStore {
    double* M;
    size_t np;
    int tri, pb, first_long;
    const int* sb;
#if defined(__HIPCC__)
    __device__ __forceinline__ size_t row_start(size_t r) const
    {
        return ((int)r >= first_long) ? (np - 1 - r) * (size_t)pb : r * (size_t)pb + sb[r];
    }
    __device__ __forceinline__ void put(size_t row, size_t col, double v) const
    {
#if defined(MQC_ERI_NO_STORE)
        // MEASUREMENT build only (scripts/build_variant.sh): the arithmetic stays alive, nothing is written
        if (v == 1.2345e300) M[0] = v;
        return;
#endif
        if (tri) {
            const size_t hi = row > col ? row : col, lo = row > col ? col : row;
            M[row_start(hi) + lo] = (hi == lo) ? 0.5 * v : v;
        } else {
            M[row * np + col] = v;
            M[col * np + row] = v;
        }
    }
#endif
};
#if defined(__HIPCC__)
__device__ __forceinline__ PairStore make_pair_store(const BatchView& bv, int f)
{
    const int np = bv.npair;
    return PairStore{bv.eri + (size_t)f * bv.eri_stride, (size_t)np, bv.eri_tri, bv.eri_tri_pb, np - (np + 1) / 2, bv.eri_tri_sb};
}
#endif

struct Stats {
    double t_setup = 0, t_int1e = 0, t_eri = 0, t_fock = 0, t_scf_step = 0, t_total = 0;
    int64_t fock_launches = 0, eri_quartets = 0, scf_iterations_total = 0;
    double fock_kernel_seconds = 0, fock_bytes = 0, eri_kernel_seconds = 0, xc_kernel_seconds = 0, xc_points = 0;
    int64_t fock_big_launches = 0;
    double fock_big_seconds = 0, fock_big_bytes = 0;
    double xc_flops = 0, scf_step_seconds = 0, df_flops = 0, df_bytes = 0;
    int64_t eri_survivors = 0;
};

}  // namespace mqc

struct mqc_hip_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop;
    double* d_boys = nullptr;
    double* d_c2s = nullptr;
    std::vector<double> h_c2s;                 // packed l = 0..LMAX_AO, (2l+1) x ncart(l)
    int c2s_off[8];
    mqc::DevicePool pool_main, pool_eri, pool_topo, pool_misc, pool_grid, pool_gridw, pool_aux, pool_df;
    // second pipeline slot: the next chunk's integrals are formed while the current chunk iterates
    mqc::DevicePool pool_main2, pool_eri2, pool_misc2, pool_gridw2, pool_df2, pool_topo2, pool_aux2, pool_grid2;
    hipStream_t stream2 = nullptr;
    hipStream_t side[2][3] = {};        // per lane: side streams of the ERI stage, created right after the lane's main stream
    hipEvent_t evo[2][2] = {};          // per lane: one-electron stage done / orthogonaliser + guess done
    hipEvent_t evb0 = nullptr, evb1 = nullptr, evb2 = nullptr, evb3 = nullptr;
    hipEvent_t evq0 = nullptr, evq1 = nullptr, evq2 = nullptr, evq3 = nullptr;   // integral-stage timing per slot
    hipEvent_t evs[2][2] = {};          // per slot: SCF-step kernel timing
    int pipeline_chunks = 4;            // chunks a large batch is cut into
    int pipeline_min_fragments = 256;   // batches below this run as one chunk
    double* d_unit = nullptr;
    mqc::Stats stats;
    std::mutex stats_mutex;
    // topologies of recent calls (shell tables, quartet class lists): an MBE driver sends the same few again and again
    std::map<std::string, std::shared_ptr<mqc::Topology>> topo_cache;
    // converged free atoms of the superposed-atom guesses: total density (n_atom_ao^2, row-major) per element + shells
    std::map<std::string, std::shared_ptr<std::vector<double>>> atom_cache;
    int concurrent_groups = 1;          // 1: topology groups of one batch call run two at a time (one per slot)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    size_t hbm_budget_bytes = 0;
};

namespace mqc {

// every pool (context members and the launchers' function-static ones) is registered; mqc_hip_finalize frees them all
void release_all_pools();
// launcher state that holds streams/events of the context's device (kern_eri.hip); reset by mqc_hip_finalize
void eri_reset_state();

// true when the in-core J/K of a batch of this shape runs from the triangular tensor (kern_fock.hip); block length and
// the per-block start of the short row
bool jk_tri_layout(int n, int npair, int nfrag, bool uhf);
int jk_tri_block(int npair, std::vector<int>* short_row_start = nullptr);
// host-side pieces (basis_norm.cpp, boys_table.cpp, batch.cpp)
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
void build_boys_table(std::vector<double>& table);
void build_c2s_tables(std::vector<double>& packed, int* offsets);
int build_topology(const mqc_hip_molecule_t& mol, const mqc_hip_basis_t& bas, Topology& topo, std::string& err,
                   int max_l = KERNEL_LMAX, bool with_quartets = true);
std::string topology_key(const mqc_hip_molecule_t& mol, const mqc_hip_basis_t& bas);
double nuclear_repulsion(const Topology& topo, const double* xyz);

// lane < 0: the batch owns both slots (chunks alternate, next chunk prepared ahead); lane 0/1: it runs on that
// slot only, so that two topology groups can be driven by two host threads at once
int run_batch(mqc_hip_context* ctx, const Topology& topo, const Topology* aux, const std::vector<const double*>& xyz,
              const mqc_hip_scf_options_t& opts, std::vector<mqc_hip_scf_result_t*>& results, int lane = -1);

// kernel launchers (kern_*.hip)
void launch_int1e(const BatchView& bv, const Topology& topo, hipStream_t s);
void launch_dipole(const BatchView& bv, const Topology& topo, hipStream_t s, bool accumulate = false);
// host_xyz (optional, [nfrag][natoms][3] as uploaded): enables block sharing between fragments with identical atoms
void launch_eri(const BatchView& bv, const Topology& topo, double schwarz_tol, hipStream_t s, const double* host_xyz = nullptr);
// optional head start of the screened build (bounds + zero fill on side streams); launch_eri joins it
void launch_eri_bounds(const BatchView& bv, const Topology& topo, double schwarz_tol, hipStream_t s);
// wave-cooperative kernel for classes with an f (or g) shell (kern_eri_general.hip); false: class too large for LDS
bool launch_eri_general(const BatchView& bv, int la, int lb, int lc, int ld, const int* d_list, int nq, const int* d_tasks, int ntasks,
                        const double* Q, double thresh, hipStream_t s);
bool launch_df3c_general(const BatchView& bv, int la, int lb, int lp, const int* d_list, int nq, hipStream_t s);
bool launch_schwarz_general(const BatchView& bv, int la, int lb, const int* d_pairs, int npairs, double* Qout, hipStream_t s);
void eri_set_side_streams(int slot, const hipStream_t* streams, int count);
void launch_jk_incore(const BatchView& bv, bool only_active, hipStream_t s);
void launch_direct_setup(const BatchView& bv, const Topology& topo, hipStream_t s);
void launch_jk_direct(const BatchView& bv, const Topology& topo, double thresh, bool only_active, hipStream_t s);
void launch_orthogonalizer(const BatchView& bv, hipStream_t s);
void launch_guess(const BatchView& bv, int guess_kind, hipStream_t s);
void launch_broadcast(double* dst, const double* src, size_t count, int nfrag, hipStream_t s);   // dst[f][i] = src[i]
void launch_scf_step(const BatchView& bv, hipStream_t s);
void launch_syev(int n, double* dA, double* dw, double* dV, hipStream_t s);
void launch_diis_coeff(int n_stored, const double* d_overlap, double* d_coef, int* d_ok, hipStream_t s);
void launch_becke_weights(const BatchView& bv, hipStream_t s);
void launch_xc_radial_cache(const BatchView& bv, hipStream_t s);
// points per tile of the quadrature kernel for n basis functions (kern_xc.hip) = granularity of the radial cache
inline int xc_tile_points(int n) { return ((n + 15) / 16 <= 6) ? 32 : 16; }     // n <= 96: 32-point tiles (kern_xc.hip, xc_tile_dispatch)
void launch_df_build(const BatchView& bv, const Topology& topo, const Topology& aux, hipStream_t s);
void launch_df_jk(const BatchView& bv, bool only_active, hipStream_t s);
void launch_xc(const BatchView& bv, bool only_active, hipStream_t s);
bool build_atom_template(int z, int level, int n_radial, int n_angular, std::vector<double>& xyz, std::vector<double>& w, std::string& err);
bool parse_functional(const char* name, XcSpec& spec, std::string& err);
double bragg_radius_bohr(int z);
size_t scf_lds_bytes(int n);

}  // namespace mqc

#define HIP_CHECK_RET(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return mqc::fail(MQC_HIP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
