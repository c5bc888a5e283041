// engine.cpp -- context, batch driver and the extern "C" entry points of libmqc_hip.so.
//
// Host control flow of one batch (the device-resident SCF loop of
// backends/cuest/backend/mqc_cuest_scf.f90:281-611, re-cut for whole batches):
//   upload geometry -> int1e -> orthogonaliser -> ERI tensor -> guess ->
//   repeat { J/K stream ; scf_step ; read ONE int (fragments still running) } -> fetch results
// Only that one integer crosses the bus per iteration.
#include "engine.hpp"
#include "md_integrals.hpp"
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <atomic>
#include <thread>

namespace mqc {
const std::string& last_error_string();
void eri_plan_lists(const BatchView& bv, const Topology& topo, hipStream_t s, const double* host_xyz);   // kern_eri.hip
bool launch_gradient(const BatchView& bv, const Topology& topo, const Topology* aux, double* d_grad, double* work, int* d_lists,
                     size_t list_capacity_ints, hipStream_t s, std::string& err);                                                      // kern_grad.hip
void launch_scale(double* p, size_t count, double f, hipStream_t s);      // kern_df.hip
void int1e_reset_state();                                                 // kern_int1e.hip: fan-out streams of small batches
void eri_schwarz_view(int slot, const double** q, double* thresh);        // kern_eri.hip
void launch_jk_direct_incremental(const BatchView& bv, const Topology& topo, double thresh, bool only_active, hipStream_t s);   // kern_eri.hip
static DevicePool g_grad_pool[2];

static int stage_check(const char* stage)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MQC_HIP_ERR_DEVICE, std::string("HIP error after stage '") + stage + "': " + hipGetErrorString(e));
    return MQC_HIP_OK;
}

static double now_s()
{
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

// Scratch (private segment).  The first time a kernel with a private segment runs on a hardware queue, ROCm reserves
// that kernel's bytes per lane x 64 lanes x every wave slot of the device (CUs x 32) FOR THAT QUEUE, and aborts the
// process (HSA_STATUS_ERROR_OUT_OF_RESOURCES, recorded in rounds 1 and 2) when the memory is not there.  No kernel the
// dispatchers launch carries more than SCRATCH_BOUND_PER_LANE (scripts/scratch_report.py lists them from the code
// objects; tests/test_host_logic.py asserts the bound on every build; the classes above it -- (dd|dp), (dd|dd), the (dd|
// Schwarz bounds -- go through the LDS kernel, which has none), so the worst case is the bound on every hardware queue
// the process may use, and the pools' HBM budget leaves that much alone.
constexpr size_t SCRATCH_BOUND_PER_LANE = 8192;
static size_t scratch_reservation_bytes(const mqc_hip_context* ctx)
{
    int queues = 4;                                         // the runtime's default
    if (const char* e = std::getenv("GPU_MAX_HW_QUEUES")) queues = std::max(1, std::atoi(e));
    const size_t waves = (size_t)std::max(1, ctx->prop.multiProcessorCount) * 32;
    return (size_t)queues * SCRATCH_BOUND_PER_LANE * 64 * waves;
}

struct TopoDevHolder {
    TopologyDev dev;
};

static int upload_topology(mqc_hip_context* ctx, const Topology& topo, TopologyDev& td, DevicePool* pool = nullptr, hipStream_t stream = nullptr)
{
    if (!pool) pool = &ctx->pool_topo;
    const int ns = (int)topo.shells.size();
    std::vector<int> l(ns), np(ns), po(ns), at(ns), ao(ns);
    for (int s = 0; s < ns; ++s) {
        l[s] = topo.shells[s].l; np[s] = topo.shells[s].nprim; po[s] = topo.shells[s].poff;
        at[s] = topo.shells[s].atom; ao[s] = topo.shells[s].aoff;
    }
    const size_t ib = sizeof(int) * (size_t)ns;
    const size_t nprim = topo.exps.size();
    // radial groups (see TopologyDev)
    std::vector<int> gfirst, gcount, gnprim, gpoff, gcoff;
    std::vector<double> gex, gco;
    for (int sidx = 0; sidx < ns; ++sidx) {
        const auto& sh = topo.shells[sidx];
        bool joined = false;
        if (!gfirst.empty()) {
            const int g = (int)gfirst.size() - 1;
            const auto& lead = topo.shells[gfirst[g]];
            if (gfirst[g] + gcount[g] == sidx && gcount[g] < XC_GROUP_MAX && lead.atom == sh.atom && lead.l == sh.l) {
                std::vector<int> where(sh.nprim, -1);
                bool all = true;
                for (int i = 0; i < sh.nprim && all; ++i) {
                    for (int k = 0; k < lead.nprim; ++k)
                        if (topo.exps[lead.poff + k] == topo.exps[sh.poff + i]) { where[i] = k; break; }
                    all = where[i] >= 0;
                }
                if (all) {
                    gco.resize(gco.size() + lead.nprim, 0.0);
                    double* row = gco.data() + gcoff[g] + (size_t)gcount[g] * lead.nprim;
                    for (int i = 0; i < sh.nprim; ++i) row[where[i]] += topo.coefs[sh.poff + i];
                    gcount[g] += 1;
                    joined = true;
                }
            }
        }
        if (!joined) {
            gfirst.push_back(sidx); gcount.push_back(1); gnprim.push_back(sh.nprim);
            gpoff.push_back((int)gex.size()); gcoff.push_back((int)gco.size());
            for (int i = 0; i < sh.nprim; ++i) { gex.push_back(topo.exps[sh.poff + i]); gco.push_back(topo.coefs[sh.poff + i]); }
        }
    }
    const int ng = (int)gfirst.size();
    {
        // deepest groups first: the lanes of a wave take consecutive groups (16 points each), so groups of equal
        // primitive count side by side keep the primitive loop's trip count uniform inside a wave
        std::vector<int> order(ng);
        for (int g = 0; g < ng; ++g) order[g] = g;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
            if (gnprim[a] != gnprim[b]) return gnprim[a] > gnprim[b];
            return gcount[a] > gcount[b];
        });
        auto permute = [&](std::vector<int>& v) { std::vector<int> t(ng); for (int g = 0; g < ng; ++g) t[g] = v[order[g]]; v.swap(t); };
        permute(gfirst); permute(gcount); permute(gnprim); permute(gpoff); permute(gcoff);
    }
    const size_t gib = sizeof(int) * (size_t)ng;
    const size_t bytes = 5 * ((ib + 255) & ~size_t(255)) + 3 * ((sizeof(double) * (nprim + topo.natoms) + 255) & ~size_t(255)) + 1024
                       + 5 * ((gib + 255) & ~size_t(255)) + ((sizeof(double) * gex.size() + 255) & ~size_t(255))
                       + ((sizeof(double) * gco.size() + 255) & ~size_t(255));
    char* base = (char*)pool->ensure(bytes);
    if (!base) return fail(MQC_HIP_ERR_DEVICE, "out of device memory (topology)");
    auto take = [&base](size_t b) { char* p = base; base += (b + 255) & ~size_t(255); return p; };
    td.sh_l = (int*)take(ib); td.sh_nprim = (int*)take(ib); td.sh_poff = (int*)take(ib);
    td.sh_atom = (int*)take(ib); td.sh_aoff = (int*)take(ib);
    td.exps = (double*)take(sizeof(double) * nprim); td.coefs = (double*)take(sizeof(double) * nprim);
    td.zeff = (double*)take(sizeof(double) * topo.natoms);
    td.grp_first = (int*)take(gib); td.grp_count = (int*)take(gib); td.grp_nprim = (int*)take(gib);
    td.grp_poff = (int*)take(gib); td.grp_coff = (int*)take(gib);
    td.gexps = (double*)take(sizeof(double) * gex.size()); td.gcoefs = (double*)take(sizeof(double) * gco.size());
    td.ngroup = ng;
    td.gprim_total = (int)gex.size(); td.gcoef_total = (int)gco.size();
    td.nshell = ns; td.nao = topo.nao; td.npair = topo.npair; td.natoms = topo.natoms; td.lmax = topo.lmax;
    hipStream_t s = stream ? stream : ctx->stream;
    HIP_CHECK_RET(hipMemcpyAsync(td.sh_l, l.data(), ib, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(td.sh_nprim, np.data(), ib, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(td.sh_poff, po.data(), ib, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(td.sh_atom, at.data(), ib, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(td.sh_aoff, ao.data(), ib, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(td.exps, topo.exps.data(), sizeof(double) * nprim, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(td.coefs, topo.coefs.data(), sizeof(double) * nprim, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(td.zeff, topo.zeff.data(), sizeof(double) * topo.natoms, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(td.grp_first, gfirst.data(), gib, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(td.grp_count, gcount.data(), gib, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(td.grp_nprim, gnprim.data(), gib, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(td.grp_poff, gpoff.data(), gib, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(td.grp_coff, gcoff.data(), gib, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(td.gexps, gex.data(), sizeof(double) * gex.size(), hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(td.gcoefs, gco.data(), sizeof(double) * gco.size(), hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipStreamSynchronize(s));   // the host vectors go out of scope
    return MQC_HIP_OK;
}

static size_t per_fragment_main_doubles(int n, int natoms, bool uhf = false, int npc = 0, bool hx = false)
{
    const size_t nn = (size_t)n * n;
    return ((npc > 0 || hx) ? nn + 4 * (size_t)npc + 64 : 0)   // embedding operator U and the point charges (x, y, z, q)
           + (hx ? nn + 32 : 0)                                  // the caller's h_extra
           + (uhf ? 6 * nn + 2 * DIIS_MAX * nn + n : 0)   // beta spin: D C F J K Vprev, DIIS histories, eps
           + 9 * nn    // S H X F D C J K Vprev
           + 6 * nn    // W
           + 2 * DIIS_MAX * nn   // DIIS histories
           + DIIS_MAX * DIIS_MAX + n + 8 + 3 * (size_t)natoms + 8 + 4;   // diis_b, eps, scal, xyz, ints (padded), dipole
}

// One pipeline slot: a stream with its own pools and events.  While the SCF loop of chunk k runs on
// one slot, the integrals of chunk k+1 are formed on the other (compute-bound ERI kernels fill the
// gaps the HBM-bound J/K stream and the per-iteration host round trip leave).
struct Slot {
    int id;
    hipStream_t s;
    DevicePool *main, *eri, *misc, *gridw, *df;
    hipEvent_t e0, e1, e2, e3, q0, q1;
    int* h_counter;
    hipEvent_t s0 = nullptr, s1 = nullptr;
};

// carve one chunk's arrays out of the slot's pools
static int carve_batch(mqc_hip_context* ctx, Slot& sl, const Topology& topo, const TopologyDev& td, int nfrag, bool with_eri, BatchView& bv,
                       bool uhf = false, int npc = 0, bool hx = false)
{
    const int n = topo.nao;
    const size_t nn = (size_t)n * n, nf = (size_t)nfrag;
    const size_t main_bytes = sizeof(double) * nf * per_fragment_main_doubles(n, topo.natoms, uhf, npc, hx) + 16384;
    char* base = (char*)sl.main->ensure(main_bytes);
    if (!base) return fail(MQC_HIP_ERR_DEVICE, "out of device memory (SCF matrices)");
    auto take = [&base](size_t b) { char* p = base; base += (b + 255) & ~size_t(255); return p; };
    bv.topo = td; bv.nfrag = nfrag; bv.n = n; bv.npair = topo.npair; bv.slot = sl.id;
    bv.boys = ctx->d_boys; bv.c2s = ctx->d_c2s;
    bv.xyz = (double*)take(sizeof(double) * nf * topo.natoms * 3);
    bv.S = (double*)take(sizeof(double) * nf * nn); bv.H = (double*)take(sizeof(double) * nf * nn);
    bv.X = (double*)take(sizeof(double) * nf * nn); bv.F = (double*)take(sizeof(double) * nf * nn);
    bv.D = (double*)take(sizeof(double) * nf * nn); bv.C = (double*)take(sizeof(double) * nf * nn);
    bv.J = (double*)take(sizeof(double) * nf * nn); bv.K = (double*)take(sizeof(double) * nf * nn);
    bv.W = (double*)take(sizeof(double) * nf * 6 * nn);
    bv.Vprev = (double*)take(sizeof(double) * nf * nn);
    bv.diis_f = (double*)take(sizeof(double) * nf * DIIS_MAX * nn);
    bv.diis_e = (double*)take(sizeof(double) * nf * DIIS_MAX * nn);
    bv.diis_b = (double*)take(sizeof(double) * nf * DIIS_MAX * DIIS_MAX);
    bv.eps = (double*)take(sizeof(double) * nf * n);
    bv.scal = (double*)take(sizeof(double) * nf * 8);
    bv.dip = (double*)take(sizeof(double) * nf * 4);
    bv.uhf = uhf ? 1 : 0; bv.nalpha = 0; bv.nbeta = 0;
    bv.Db = bv.Cb = bv.Fb = bv.Jb = bv.Kb = bv.Vprevb = bv.epsb = bv.diis_fb = bv.diis_eb = nullptr;
    if (uhf) {
        bv.Db = (double*)take(sizeof(double) * nf * nn); bv.Cb = (double*)take(sizeof(double) * nf * nn);
        bv.Fb = (double*)take(sizeof(double) * nf * nn); bv.Jb = (double*)take(sizeof(double) * nf * nn);
        bv.Kb = (double*)take(sizeof(double) * nf * nn); bv.Vprevb = (double*)take(sizeof(double) * nf * nn);
        bv.diis_fb = (double*)take(sizeof(double) * nf * DIIS_MAX * nn);
        bv.diis_eb = (double*)take(sizeof(double) * nf * DIIS_MAX * nn);
        bv.epsb = (double*)take(sizeof(double) * nf * n);
    }
    bv.diis_state = (int*)take(sizeof(int) * nf * 2);
    bv.istate = (int*)take(sizeof(int) * nf * 4);
    bv.npc = npc; bv.pc = nullptr; bv.U = nullptr; bv.Hx = nullptr;
    if (npc > 0) bv.pc = (double*)take(sizeof(double) * nf * npc * 4);
    if (npc > 0 || hx) bv.U = (double*)take(sizeof(double) * nf * nn);
    if (hx) bv.Hx = (double*)take(sizeof(double) * nf * nn);
    bv.counters = (int*)sl.misc->ensure(256 + sizeof(int) * (2 * (size_t)topo.npair + 8 + (size_t)nfrag));      // + the block tables of a triangular tensor + its per-fragment read counters
    if (!bv.counters) return fail(MQC_HIP_ERR_DEVICE, "out of device memory (counters)");
    bv.eri_count = (unsigned long long*)(bv.counters + 16);
    bv.eri = nullptr; bv.eri_tri = 0; bv.eri_tri_pb = 0; bv.eri_tri_sb = nullptr; bv.eri_stride = 0; bv.jk_q = nullptr; bv.jk_qthresh = 0.0; bv.jk_loaded = nullptr;
    if (with_eri) {
        const size_t np = (size_t)topo.npair;
        bv.eri_tri = jk_tri_layout(n, topo.npair, nfrag, uhf) ? 1 : 0;
        bv.eri_stride = np * np;
        if (bv.eri_tri) {
            // triangular block layout (kern_fock.hip): block length and the start of the short row of every block
            static std::vector<int> sb_host[2];
            std::vector<int>& sbh = sb_host[sl.id & 1];
            bv.eri_tri_pb = jk_tri_block(topo.npair, &sbh);
            bv.eri_stride = (size_t)((topo.npair + 1) / 2) * (size_t)bv.eri_tri_pb;
            int* d_sb = (int*)((char*)bv.counters + 256);
            if (hipMemcpyAsync(d_sb, sbh.data(), sizeof(int) * sbh.size(), hipMemcpyHostToDevice, sl.s) != hipSuccess)
                return fail(MQC_HIP_ERR_DEVICE, "upload of the tensor block table failed");
            bv.eri_tri_sb = d_sb;
            bv.jk_loaded = d_sb + 2 * (size_t)topo.npair + 8;
        }
        bv.eri = (double*)sl.eri->ensure(sizeof(double) * nf * bv.eri_stride);
        if (!bv.eri) return fail(MQC_HIP_ERR_DEVICE, "out of device memory (ERI tensor)");
    }
    return MQC_HIP_OK;
}

static bool incore_supported(int n)
{
    // the J/K kernel stages whole packed rows through LDS: 3 row-sized buffers must fit 160 KB
    const size_t np = (size_t)n * (n + 1) / 2;
    return n <= 256 && (sizeof(double) * 3 * np <= 160 * 1024 - 512);
}

static int validate_options(const mqc_hip_scf_options_t& o, const Topology& topo, std::string& msg)
{
    {
        XcSpec tmp; std::string e;
        if (!parse_functional(o.functional, tmp, e)) { msg = e; return MQC_HIP_ERR_UNSUPPORTED; }
        if (tmp.gga == 2 && topo.nao > 140) { msg = "meta-GGA functionals are available up to n_ao = 140"; return MQC_HIP_ERR_UNSUPPORTED; }
        if (tmp.ncomp > 0 && topo.natoms > 64) { msg = "XC grid: fragments above 64 atoms are not supported yet"; return MQC_HIP_ERR_UNSUPPORTED; }
    }
    if (o.want_gradient) {
        XcSpec tg; std::string eg;
        parse_functional(o.functional, tg, eg);
        if (tg.gga == 2) { msg = "analytic gradients of meta-GGA functionals are not built (energies only)"; return MQC_HIP_ERR_UNSUPPORTED; }
        if (topo.lmax > 3) { msg = "analytic gradients cover s, p, d and f shells"; return MQC_HIP_ERR_UNSUPPORTED; }
    }
    // restricted iff multiplicity 1, even electron count and not forced (mqc_cuest_driver.f90:127)
    const bool uhf = o.unrestricted || topo.multiplicity != 1 || (topo.nelec % 2) != 0;
    if (uhf) {
        // the occupation checks of run_libcint_uhf (mqc_libcint_rhf.f90:768-793)
        if (topo.multiplicity < 1) { msg = "UHF: multiplicity must be at least 1"; return MQC_HIP_ERR_VALIDATION; }
        if ((topo.nelec + topo.multiplicity - 1) % 2 != 0) { msg = "UHF: an electron count and multiplicity that cannot be paired -- their parities disagree"; return MQC_HIP_ERR_VALIDATION; }
        const int na = (topo.nelec + topo.multiplicity - 1) / 2, nb = topo.nelec - na;
        if (nb < 0 || na < 0) { msg = "UHF: multiplicity asks for more unpaired electrons than the system has"; return MQC_HIP_ERR_VALIDATION; }
        if (na < 1) { msg = "UHF: no electrons to place"; return MQC_HIP_ERR_VALIDATION; }
        {
            XcSpec tu; std::string eu;
            parse_functional(o.functional, tu, eu);
            if (tu.ncomp > 0 && topo.nao > 140) { msg = "unrestricted Kohn-Sham is available up to n_ao = 140"; return MQC_HIP_ERR_UNSUPPORTED; }
        }
    } else if (topo.nelec < 2) { msg = "RHF: no electrons to place"; return MQC_HIP_ERR_VALIDATION; }
    if (o.guess < MQC_HIP_GUESS_AUTO || o.guess > MQC_HIP_GUESS_SAC) { msg = "unknown initial guess"; return MQC_HIP_ERR_VALIDATION; }
    if (o.guess == MQC_HIP_GUESS_SAC && uhf) { msg = "the SAC guess (free atoms' own spin densities) is available for restricted runs; unrestricted runs take sad, gwh or core"; return MQC_HIP_ERR_UNSUPPORTED; }
    if (o.max_iter < 1) { msg = "max_iter must be positive"; return MQC_HIP_ERR_VALIDATION; }
    if (o.use_diis && (o.diis_size < 0 || o.diis_size > DIIS_MAX)) { msg = "diis_size must be within 0..8"; return MQC_HIP_ERR_VALIDATION; }
    if (!o.density_fitting && o.eri_mode == MQC_HIP_ERI_INCORE && !incore_supported(topo.nao)) { msg = "fragment too large for the in-core exact-ERI path (n_ao <= 116); use eri_mode auto/direct or density fitting"; return MQC_HIP_ERR_UNSUPPORTED; }
    if (topo.lmax > CLASS_LMAX) {
        const bool direct = !o.density_fitting && (o.eri_mode == MQC_HIP_ERI_DIRECT || (o.eri_mode == MQC_HIP_ERI_AUTO && !incore_supported(topo.nao)));
        (void)direct;       // f classes are digested by the LDS kernel: the direct build covers them
        if (o.density_fitting && topo.lmax > 3) { msg = "density fitting covers orbital shells up to f"; return MQC_HIP_ERR_UNSUPPORTED; }
    }
    // n_ao <= 140: the Fock matrix is diagonalised in LDS; up to 256 it is rotated in global memory (L2), exact-ERI
    // direct path and the quadrature's z-split; density fitting keeps the 140 limit (its J/K kernels tile n in LDS)
    if (topo.nao > 256) { msg = "fragment too large for the eigen-solver (n_ao <= 256)"; return MQC_HIP_ERR_UNSUPPORTED; }
    if (topo.nao > 140 && o.density_fitting) { msg = "density fitting is available up to n_ao = 140; larger fragments run on the direct exact-ERI path"; return MQC_HIP_ERR_UNSUPPORTED; }
    if (topo.nao > 140 && o.want_gradient) { msg = "analytic gradients are available up to n_ao = 140"; return MQC_HIP_ERR_UNSUPPORTED; }
    return MQC_HIP_OK;
}

static void fill_error(mqc_hip_scf_result_t* r, const std::string& msg)
{
    r->has_error = 1;
    std::snprintf(r->message, sizeof(r->message), "%s", msg.c_str());
}

// Starting density of the superposed-atom guesses (one per topology: it does not depend on the geometry) and, for
// the density-fitted exchange, its pseudo-orbitals v_i sqrt(n_i / 2) (density_pseudo_orbitals, mqc_libcint_rhf.f90:1413-1462)
struct AtomicGuess {
    std::vector<double> D0;      // [n*n] total density, block-diagonal over the atoms
    std::vector<double> Cp;      // [n*n] row-major, nmodes columns used
    int nmodes = 0;
};
static DevicePool g_guess_pool[2];

int run_batch(mqc_hip_context* ctx, const Topology& topo, const Topology* aux, const std::vector<const double*>& xyz_in,
              const mqc_hip_scf_options_t& opts, std::vector<mqc_hip_scf_result_t*>& results_in, int lane,
              const AtomicGuess* atomic_guess = nullptr, const std::vector<const mqc_hip_molecule_t*>* mols_in = nullptr)
{
    // statistics are gathered locally and merged at the end (two lanes may run at once)
    struct StatsCtx { Stats stats; } local;
    StatsCtx* const sx = &local;
    const bool second = lane == 1;
    hipStream_t const lane_stream = second ? ctx->stream2 : ctx->stream;
    const double t_begin = now_s();
    const int ntot = (int)xyz_in.size();
    // Order the batch by compactness (nuclear repulsion, most compact first).  Lanes of a wave are
    // consecutive fragments: with similar geometries side by side, the primitive-pair screening and
    // the Schwarz ballot drop the same work in every lane, so whole waves skip it.
    std::vector<const double*> xyz(ntot);
    std::vector<mqc_hip_scf_result_t*> results(ntot);
    // external point charges (FMO / EE-MBE embedding): the same count in every fragment of the group (part of its key)
    const int npc = (mols_in && ntot > 0) ? (*mols_in)[0]->n_point_charges : 0;
    const bool hx = mols_in && ntot > 0 && (*mols_in)[0]->h_extra != nullptr;     // all or none within a group (its key says so)
    const bool embedded = npc > 0 || hx;
    std::vector<const mqc_hip_molecule_t*> pcmol(embedded ? ntot : 0);
    {
        std::vector<std::pair<double, int>> key(ntot);
        for (int i = 0; i < ntot; ++i) key[i] = {-nuclear_repulsion(topo, xyz_in[i]), i};
        std::stable_sort(key.begin(), key.end());
        for (int k = 0; k < ntot; ++k) { xyz[k] = xyz_in[key[k].second]; results[k] = results_in[key[k].second]; }
        if (embedded) for (int k = 0; k < ntot; ++k) pcmol[k] = (*mols_in)[key[k].second];
    }
    std::string msg;
    int rc = validate_options(opts, topo, msg);
    if (rc == MQC_HIP_OK && embedded && opts.want_gradient) {
        msg = "analytic gradients of a fragment embedded in point charges or an extra one-electron operator are not built (the field's own derivative is missing)";
        rc = MQC_HIP_ERR_UNSUPPORTED;
    }
    if (rc == MQC_HIP_OK && npc > 0)
        for (int k = 0; k < ntot && rc == MQC_HIP_OK; ++k)
            if (!pcmol[k]->point_charge_xyz || !pcmol[k]->point_charges) { msg = "point charges announced but their arrays are NULL"; rc = MQC_HIP_ERR_VALIDATION; }
    if (rc != MQC_HIP_OK) {
        for (auto* r : results) { fill_error(r, msg); r->scf_status = MQC_HIP_SCF_NOT_RUN; }
        return fail(rc, msg);
    }
    TopologyDev td;
    rc = upload_topology(ctx, topo, td, second ? &ctx->pool_topo2 : &ctx->pool_topo, lane_stream);
    if (rc != MQC_HIP_OK) return rc;
    const bool use_df = opts.density_fitting != 0;
    // exact-ERI path selection (mqc_libcint_bridge.f90:819-892 with an HBM budget instead of 2 GB of host memory)
    const bool use_direct = !use_df && (opts.eri_mode == MQC_HIP_ERI_DIRECT ||
                                        (opts.eri_mode == MQC_HIP_ERI_AUTO && !incore_supported(topo.nao)));
    const double direct_tol = opts.schwarz_tol > 0.0 ? opts.schwarz_tol : 1.0e-11;   // mqc_libcint_direct.f90:61
    TopologyDev tdx{};
    int naux = 0;
    if (use_df) {
        if (!aux) {
            const std::string m = "density fitting needs an auxiliary basis";
            for (auto* r : results) { fill_error(r, m); r->scf_status = MQC_HIP_SCF_NOT_RUN; }
            return fail(MQC_HIP_ERR_VALIDATION, m);
        }
        rc = upload_topology(ctx, *aux, tdx, second ? &ctx->pool_aux2 : &ctx->pool_aux, lane_stream);
        if (rc != MQC_HIP_OK) return rc;
        naux = aux->nao;
    }

    // ---- exchange-correlation: functional, per-element grid templates, per-topology point list
    XcSpec xc;
    {
        std::string e;
        parse_functional(opts.functional, xc, e);
    }
    GridDev grid;
    if (xc.ncomp > 0) {
        std::map<int, std::pair<int, int>> tmpl_of_z;     // Z -> (offset, count) in the packed template arrays
        std::vector<double> txyz, tw, sb(topo.natoms);
        std::vector<int> pt_atom, pt_tmpl;
        for (int a = 0; a < topo.natoms; ++a) {
            // a ghost centre enters the grid builder with Z = 0 (numbers = nint(mol%charges), mqc_libcint_xc.F90:181-183):
            // period-1 sizes, xi(0) = 1, Bragg radius(0) = 2 Angstrom -- it still owns grid points
            const int z = (topo.zeff[a] == 0.0) ? 0 : topo.Z[a];
            sb[a] = std::sqrt(bragg_radius_bohr(z)) + 1e-200;
            if (!tmpl_of_z.count(z)) {
                std::vector<double> x, w; std::string e;
                if (!build_atom_template(z, opts.grid_level, opts.radial_points, opts.angular_points, x, w, e)) {
                    for (auto* r : results) { fill_error(r, e); r->scf_status = MQC_HIP_SCF_NOT_RUN; }
                    return fail(MQC_HIP_ERR_UNSUPPORTED, e);
                }
                tmpl_of_z[z] = {(int)tw.size(), (int)w.size()};
                txyz.insert(txyz.end(), x.begin(), x.end());
                tw.insert(tw.end(), w.begin(), w.end());
            }
            const auto oc = tmpl_of_z[z];
            for (int k = 0; k < oc.second; ++k) { pt_atom.push_back(a); pt_tmpl.push_back(oc.first + k); }
        }
        grid.npts = (int)pt_atom.size();
        const size_t b_int = (sizeof(int) * pt_atom.size() + 255) & ~size_t(255);
        const size_t b_xyz = (sizeof(double) * txyz.size() + 255) & ~size_t(255);
        const size_t b_w = (sizeof(double) * tw.size() + 255) & ~size_t(255);
        const size_t b_sb = (sizeof(double) * sb.size() + 255) & ~size_t(255);
        char* gb = (char*)(second ? ctx->pool_grid2 : ctx->pool_grid).ensure(2 * b_int + b_xyz + b_w + b_sb + 1024);
        if (!gb) return fail(MQC_HIP_ERR_DEVICE, "out of device memory (grid templates)");
        int* d_pa = (int*)gb; int* d_pt = (int*)(gb + b_int);
        double* d_x = (double*)(gb + 2 * b_int); double* d_w = (double*)(gb + 2 * b_int + b_xyz);
        double* d_sb = (double*)(gb + 2 * b_int + b_xyz + b_w);
        HIP_CHECK_RET(hipMemcpyAsync(d_pa, pt_atom.data(), sizeof(int) * pt_atom.size(), hipMemcpyHostToDevice, lane_stream));
        HIP_CHECK_RET(hipMemcpyAsync(d_pt, pt_tmpl.data(), sizeof(int) * pt_tmpl.size(), hipMemcpyHostToDevice, lane_stream));
        HIP_CHECK_RET(hipMemcpyAsync(d_x, txyz.data(), sizeof(double) * txyz.size(), hipMemcpyHostToDevice, lane_stream));
        HIP_CHECK_RET(hipMemcpyAsync(d_w, tw.data(), sizeof(double) * tw.size(), hipMemcpyHostToDevice, lane_stream));
        HIP_CHECK_RET(hipMemcpyAsync(d_sb, sb.data(), sizeof(double) * sb.size(), hipMemcpyHostToDevice, lane_stream));
        HIP_CHECK_RET(hipStreamSynchronize(lane_stream));     // the host vectors go out of scope
        grid.pt_atom = d_pa; grid.pt_tmpl = d_pt; grid.tmpl_xyz = d_x; grid.tmpl_w = d_w; grid.sqrt_bragg = d_sb;
    }

    const int n = topo.nao;
    const size_t np = (size_t)topo.npair;
    const bool uhf_mem = opts.unrestricted || topo.multiplicity != 1 || (topo.nelec % 2) != 0;
    // in-core tensor: the square, or -- where the batch takes the triangular block layout (kern_fock.hip) -- 0.53 of it
    const size_t tensor = jk_tri_layout(n, topo.npair, ntot, uhf_mem) ? (size_t)((topo.npair + 1) / 2) * (size_t)jk_tri_block(topo.npair) : np * np;
    const size_t two_e = use_df ? (2 * (size_t)naux * np + 3 * (size_t)naux * naux) : (use_direct ? 2 * (size_t)n * n : tensor);
    // radial cache of the quadrature (MQC_HIP_XC_RADIAL_CACHE=0 turns it off): 2 doubles per shell and (padded) grid point
    static const bool rad_cache_on = [] { const char* e = std::getenv("MQC_HIP_XC_RADIAL_CACHE"); return !(e && e[0] == '0'); }();
    const int rad_pt = xc_tile_points(n);
    const size_t rad_tiles = xc.ncomp > 0 ? ((size_t)grid.npts + rad_pt - 1) / rad_pt : 0;
    const size_t rad_doubles = (xc.ncomp > 0 && rad_cache_on && !uhf_mem) ? rad_tiles * topo.shells.size() * 2 * rad_pt : 0;
    // point buffer of the split quadrature (n <= 96, s-f shells, 32-point tiles): 4 doubles per padded grid point
    const size_t pt4_doubles = (rad_doubles && n <= 96 && topo.lmax <= 3 && rad_pt == 32) ? rad_tiles * rad_pt * 4 : 0;
    const size_t per_frag = sizeof(double) * (per_fragment_main_doubles(n, topo.natoms, uhf_mem, npc, hx) + two_e + (xc.ncomp > 0 ? (size_t)n * n * (uhf_mem ? 2 : 1) + grid.npts : 0) + rad_doubles + pt4_doubles);
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    free_b += ctx->pool_main.capacity() + ctx->pool_eri.capacity() + ctx->pool_df.capacity() + ctx->pool_gridw.capacity()
            + ctx->pool_main2.capacity() + ctx->pool_eri2.capacity() + ctx->pool_df2.capacity() + ctx->pool_gridw2.capacity();
    // a lane shares the card with the other lane's batch: 40 % each instead of 80 %
    size_t budget = (size_t)((lane < 0 ? 0.80 : 0.40) * (double)free_b);
    {
        // ... and the pools never take what the runtime's scratch reservation may need (see scratch_reservation_bytes)
        const size_t res = scratch_reservation_bytes(ctx);
        const size_t room = free_b > res ? free_b - res : free_b / 8;
        const size_t cap = lane < 0 ? room : room / 2;
        if (cap < budget) budget = cap;
    }
    if (ctx->hbm_budget_bytes && ctx->hbm_budget_bytes < budget) budget = ctx->hbm_budget_bytes;
    // Chunking.  Small batches run as one chunk on slot 0.  Large ones are cut into >= 4 chunks that
    // alternate between the two slots (each slot may hold half of the budget).
    const bool pipelined = lane < 0 && ctx->pipeline_chunks > 1 && ntot >= ctx->pipeline_min_fragments;
    long chunk = (long)(budget / per_frag);
    if (lane < 0 && (pipelined || chunk < ntot)) {
        // more than one chunk: two are resident at a time
        chunk = (long)(budget / 2 / per_frag);
        const long want = pipelined ? (ntot + ctx->pipeline_chunks - 1) / ctx->pipeline_chunks : ntot;
        if (chunk > want) chunk = want;
    }
    if (chunk < 1) return fail(MQC_HIP_ERR_DEVICE, "not enough device memory for one fragment");
    if (chunk > ntot) chunk = ntot;
    if (chunk > 60000) chunk = 60000;    // grid.y limit of the J/K kernel

    const bool uhf = opts.unrestricted || topo.multiplicity != 1 || (topo.nelec % 2) != 0;
    const int nalpha = uhf ? (topo.nelec + topo.multiplicity - 1) / 2 : topo.nelec / 2;
    const int nbeta = uhf ? topo.nelec - nalpha : topo.nelec / 2;
    const int nocc = uhf ? nalpha : topo.nelec / 2;
    int* h_counter = nullptr;
    HIP_CHECK_RET(hipHostMalloc((void**)&h_counter, 256));
    Slot slots[2] = {
        {0, ctx->stream, &ctx->pool_main, &ctx->pool_eri, &ctx->pool_misc, &ctx->pool_gridw, &ctx->pool_df,
         ctx->ev0, ctx->ev1, ctx->ev2, ctx->ev3, ctx->evq0, ctx->evq1, h_counter, ctx->evs[0][0], ctx->evs[0][1]},
        {1, ctx->stream2, &ctx->pool_main2, &ctx->pool_eri2, &ctx->pool_misc2, &ctx->pool_gridw2, &ctx->pool_df2,
         ctx->evb0, ctx->evb1, ctx->evb2, ctx->evb3, ctx->evq2, ctx->evq3, h_counter + 32, ctx->evs[1][0], ctx->evs[1][1]}};

    struct Job { int start = 0, nf = 0; BatchView bv{}; std::vector<double> hx, hpc; };
    std::vector<Job> jobs;
    for (int start = 0; start < ntot; start += (int)chunk) {
        Job j; j.start = start; j.nf = (int)std::min<long>(chunk, ntot - start);
        jobs.push_back(std::move(j));
    }
    // Schwarz screening of the in-core build pays for itself through the quartets it drops; for a handful of fragments
    // the bounds (one thread per shell pair walking its primitive quartets: 1.6 ms for one cc-pVDZ water dimer) cost
    // more than the screened quartets save, and the unscreened tensor needs no zero fill.  Energies move by < 1e-11 Eh
    // either way (that is what the threshold means); MQC_HIP_SCHWARZ_MIN_FRAGMENTS=0 screens always.
    static const int schwarz_min = [] { const char* e = std::getenv("MQC_HIP_SCHWARZ_MIN_FRAGMENTS"); return e ? std::atoi(e) : 9; }();
    const double stol = (opts.schwarz_tol > 0.0 && ntot >= schwarz_min) ? opts.schwarz_tol : 0.0;

    // drains both streams before an error return hands the pools back
    auto bail = [&](int code) {
        if (lane != 1) (void)hipStreamSynchronize(ctx->stream);
        if (lane != 0) (void)hipStreamSynchronize(ctx->stream2);
        (void)hipHostFree(h_counter);
        return code;
    };

    // ---- stage 1 of a chunk: everything up to (and including) the guess, enqueued on the slot's stream
    auto prepare = [&](Slot& sl, Job& job) -> int {
        const int nf = job.nf;
        hipStream_t s = sl.s;
        const double t0 = now_s();
        BatchView& bv = job.bv;
        int rc = carve_batch(ctx, sl, topo, td, nf, !use_df && !use_direct, bv, uhf, npc, hx);
        if (rc != MQC_HIP_OK) return rc;
        bv.nalpha = nalpha; bv.nbeta = nbeta;
        bv.naux = naux; bv.aux = tdx; bv.unit = ctx->d_unit;
        if (use_df) {
            const size_t a3 = (size_t)nf * naux * np, mm = (size_t)nf * naux * naux;
            double* base = (double*)sl.df->ensure(sizeof(double) * (2 * a3 + 3 * mm) + 1024);
            if (!base) return fail(MQC_HIP_ERR_DEVICE, "out of device memory (fitted tensor)");
            bv.df_a3 = base; bv.df_b = base + a3; bv.df_metric = base + 2 * a3; bv.df_linv = base + 2 * a3 + mm;
            bv.df_work = base + 2 * a3 + 2 * mm;
            HIP_CHECK_RET(hipMemsetAsync(bv.scal, 0, sizeof(double) * (size_t)nf * 8, s));
        }
        bv.nocc = nocc; bv.exx = xc.exx; bv.e_tol = opts.energy_tol; bv.d_tol = opts.density_tol;
        bv.xc = xc; bv.grid = grid; bv.Vxc = nullptr;
        if (xc.ncomp > 0) {
            char* gw = (char*)sl.gridw->ensure(sizeof(double) * (size_t)nf * ((size_t)grid.npts + (size_t)n * n * (uhf ? 2 : 1) + rad_doubles + pt4_doubles) + 2048);
            if (!gw) return fail(MQC_HIP_ERR_DEVICE, "out of device memory (grid weights)");
            bv.grid.weights = (double*)gw;
            bv.Vxc = (double*)(gw + ((sizeof(double) * (size_t)nf * grid.npts + 255) & ~size_t(255)));
            bv.grid.rad = nullptr; bv.grid.rad_pt = rad_pt;
            if (rad_doubles) bv.grid.rad = (double*)((char*)bv.Vxc + ((sizeof(double) * (size_t)nf * n * n * (uhf ? 2 : 1) + 255) & ~size_t(255)));
            bv.grid.pt4 = nullptr;
            if (pt4_doubles) bv.grid.pt4 = (double*)((char*)bv.grid.rad + ((sizeof(double) * (size_t)nf * rad_doubles + 255) & ~size_t(255)));
        }
        bv.max_iter = opts.max_iter; bv.diis_size = opts.use_diis ? opts.diis_size : 0;
        job.hx.resize((size_t)nf * topo.natoms * 3);
        for (int f = 0; f < nf; ++f) std::memcpy(&job.hx[(size_t)f * topo.natoms * 3], xyz[job.start + f], sizeof(double) * topo.natoms * 3);
        HIP_CHECK_RET(hipMemcpyAsync(bv.xyz, job.hx.data(), sizeof(double) * job.hx.size(), hipMemcpyHostToDevice, s));
        if (npc > 0) {
            // layout [charge][x, y, z, q][fragment], FRAGMENT FASTEST: the lanes of a wave are consecutive fragments, so a
            // wave reads a charge's component as one contiguous 512-byte line -- with [fragment][charge][4] every lane
            // streamed its own 49 KB array and a wave-load touched 64 cache lines (int1e: 40 % of a 512-fragment FMO run)
            job.hpc.resize((size_t)nf * npc * 4);
            for (int f = 0; f < nf; ++f) {
                const mqc_hip_molecule_t* m = pcmol[job.start + f];
                for (int g = 0; g < npc; ++g) {
                    double* q = &job.hpc[(size_t)g * 4 * nf + f];
                    q[0] = m->point_charge_xyz[3 * g]; q[(size_t)nf] = m->point_charge_xyz[3 * g + 1]; q[2 * (size_t)nf] = m->point_charge_xyz[3 * g + 2];
                    q[3 * (size_t)nf] = m->point_charges[g];
                }
            }
            HIP_CHECK_RET(hipMemcpyAsync((void*)bv.pc, job.hpc.data(), sizeof(double) * job.hpc.size(), hipMemcpyHostToDevice, s));
        }
        if (hx)
            for (int f = 0; f < nf; ++f)
                HIP_CHECK_RET(hipMemcpyAsync((void*)(bv.Hx + (size_t)f * n * n), pcmol[job.start + f]->h_extra, sizeof(double) * n * n, hipMemcpyHostToDevice, s));
        HIP_CHECK_RET(hipMemsetAsync(bv.istate, 0, sizeof(int) * (size_t)nf * 4, s));
        HIP_CHECK_RET(hipMemsetAsync(bv.eri_count, 0, sizeof(unsigned long long), s));
        const double t1 = now_s();
        sx->stats.t_setup += t1 - t0;

        if (!use_df && !use_direct) launch_eri_bounds(bv, topo, stol, s);     // screened build: bounds run next to the 1e stage
        // The one-electron stage, the orthogonaliser and the starting guess need the geometry (S, H) only and are
        // latency-bound (six class launches; one workgroup per fragment, Jacobi sweeps): they run on a side stream next
        // to the compute-bound two-electron stage, which does not wait for them (0.7 ms of a single-fragment call)
        hipStream_t so = ctx->side[sl.id & 1][2];
        HIP_CHECK_RET(hipEventRecord(ctx->evo[sl.id & 1][0], s));             // uploads and resets above are in
        HIP_CHECK_RET(hipStreamWaitEvent(so, ctx->evo[sl.id & 1][0], 0));
        launch_int1e(bv, topo, so);
        // block-sharing plan and class lists of the integral stage: host work that depends on the geometry only, done
        // here while the bounds and one-electron kernels run
        if (!use_df && !use_direct) eri_plan_lists(bv, topo, s, job.hx.data());
        if ((rc = stage_check("int1e")) != MQC_HIP_OK) return rc;
        launch_orthogonalizer(bv, so);
        if ((rc = stage_check("orthogonalizer")) != MQC_HIP_OK) return rc;
        if (!atomic_guess) launch_guess(bv, opts.guess == MQC_HIP_GUESS_CORE ? MQC_HIP_GUESS_CORE : MQC_HIP_GUESS_GWH, so);
        if ((rc = stage_check("guess")) != MQC_HIP_OK) return rc;
        HIP_CHECK_RET(hipEventRecord(ctx->evo[sl.id & 1][1], so));
        if (xc.ncomp > 0) { launch_becke_weights(bv, s); if (bv.grid.rad) launch_xc_radial_cache(bv, s); }
        if ((rc = stage_check("grid weights")) != MQC_HIP_OK) return rc;
        const double t2 = now_s();
        sx->stats.t_int1e += t2 - t1;

        HIP_CHECK_RET(hipEventRecord(sl.q0, s));
        if (use_df) launch_df_build(bv, topo, *aux, s);
        else if (use_direct) launch_direct_setup(bv, topo, s);
        else {
            launch_eri(bv, topo, stol, s, job.hx.data());
            // the bounds of a screened build tell the J/K kernel which pair rows are all zeros (triangular tensor only)
            if (stol > 0.0 && bv.eri_tri) eri_schwarz_view(bv.slot, &bv.jk_q, &bv.jk_qthresh);
        }
        HIP_CHECK_RET(hipEventRecord(sl.q1, s));
        if ((rc = stage_check("two-electron setup")) != MQC_HIP_OK) return rc;
        sx->stats.eri_quartets += topo.n_quartets * nf;
        HIP_CHECK_RET(hipStreamWaitEvent(s, ctx->evo[sl.id & 1][1], 0));      // join: X, C, D of the guess are ready
        if (atomic_guess) {
            // superposed atoms (build_restricted_guess / atomic_guess_fock, mqc_libcint_atomic_guess.f90:168-212,
            // mqc_libcint_rhf.f90:1382-1411): the same block-diagonal density in every fragment of the topology, its
            // Hartree-Fock Fock matrix from the two-electron stage just built (full exchange), then the usual
            // diagonalisation and occupation -- both spins of an unrestricted run start from it
            const size_t nn = (size_t)n * n;
            double* d0 = (double*)g_guess_pool[sl.id & 1].ensure(sizeof(double) * 2 * nn + 256);
            if (!d0) return fail(MQC_HIP_ERR_DEVICE, "out of device memory (guess density)");
            HIP_CHECK_RET(hipMemcpyAsync(d0, atomic_guess->D0.data(), sizeof(double) * nn, hipMemcpyHostToDevice, s));
            launch_broadcast(bv.D, d0, nn, nf, s);
            BatchView vg = bv;
            vg.exx = 1.0; vg.uhf = 0;
            if (use_df) {
                HIP_CHECK_RET(hipMemcpyAsync(d0 + nn, atomic_guess->Cp.data(), sizeof(double) * nn, hipMemcpyHostToDevice, s));
                launch_broadcast(bv.C, d0 + nn, nn, nf, s);
                vg.nocc = atomic_guess->nmodes;
                launch_df_jk(vg, false, s);
            } else if (use_direct) launch_jk_direct(vg, topo, direct_tol, false, s);
            else launch_jk_incore(vg, false, s);
            launch_guess(bv, MQC_HIP_GUESS_SAD, s);
            if ((rc = stage_check("atomic guess")) != MQC_HIP_OK) return rc;
        }
        sx->stats.t_eri += now_s() - t2;      // host time to enqueue; the kernels are timed by q0/q1
        return MQC_HIP_OK;
    };

    // ---- stage 2: the SCF loop (one int back per iteration) and the result fetch
    auto iterate_and_fetch = [&](Slot& sl, Job& job) -> int {
        const int nf = job.nf;
        hipStream_t s = sl.s;
        BatchView& bv = job.bv;
        int rc;
        const double t3 = now_s();
        HIP_CHECK_RET(hipStreamSynchronize(s));
        if ((rc = stage_check("integrals")) != MQC_HIP_OK) return rc;
        {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, sl.q0, sl.q1);
            sx->stats.eri_kernel_seconds += ms * 1e-3;
        }
        int remaining = nf;
        int guard = 0;
        double tri_doubles_per_fragment = -1.0;      // read back after the first J/K launch of the chunk
        // Kohn-Sham: the quadrature needs the density only, like J/K -- it runs on the lane's side stream next to the
        // J/K build and joins before the SCF step (the event pair that synchronised orthogonaliser and guess is free now)
        hipStream_t sxc = ctx->side[sl.id & 1][2];
        // OFF by default: measured 1206 against 1230 ms per B3LYP evaluation (-2 %), but the J/K launches then wait inside
        // the quadrature's shadow and their HIP-event times (the bench's per-stage figures) stop meaning anything
        static const bool xc_side = [] { const char* e = std::getenv("MQC_HIP_XC_SIDE_STREAM"); return e && e[0] == '1'; }();
        const bool xc_on_side = xc.ncomp > 0 && xc_side;
        // Small batches are latency-bound: the host round trip after every iteration (one int back, ~50 us) costs as
        // much as the kernels.  They run BLOCKS of iterations between two reads of the counter -- finished fragments
        // drop out of every kernel by themselves (state machine), so an iteration enqueued past convergence is three
        // empty launches.  Large batches keep one read per iteration (MQC_HIP_SCF_SYNC_BLOCK overrides: 1 = always).
        static const int sync_block_env = [] { const char* e = std::getenv("MQC_HIP_SCF_SYNC_BLOCK"); return e ? std::atoi(e) : 0; }();
        static const int block_max_frag = [] { const char* e = std::getenv("MQC_HIP_SCF_BLOCK_MAX_FRAGMENTS"); return e ? std::atoi(e) : 256; }();
        const bool blocked = sync_block_env != 1 && nf <= block_max_frag;
        int blocks_done = 0;
        while (remaining > 0 && guard < opts.max_iter + 2) {
          int block_len = 1;
          if (blocked) block_len = sync_block_env > 1 ? sync_block_env : (blocks_done == 0 ? 8 : 4);
          if (block_len > opts.max_iter + 2 - guard) block_len = opts.max_iter + 2 - guard;
          ++blocks_done;
          for (int bi = 0; bi < block_len; ++bi) {
            const bool last_of_block = bi + 1 == block_len;
            if (xc_on_side) {
                HIP_CHECK_RET(hipEventRecord(ctx->evo[sl.id & 1][0], s));
                HIP_CHECK_RET(hipStreamWaitEvent(sxc, ctx->evo[sl.id & 1][0], 0));
                HIP_CHECK_RET(hipEventRecord(sl.e2, sxc));
                launch_xc(bv, true, sxc);
                HIP_CHECK_RET(hipEventRecord(sl.e3, sxc));
                HIP_CHECK_RET(hipEventRecord(ctx->evo[sl.id & 1][1], sxc));
            }
            HIP_CHECK_RET(hipEventRecord(sl.e0, s));
            if (use_df) {
                launch_df_jk(bv, true, s);
                if (uhf) {
                    // unrestricted density fitting, as run_uks_scf of the cuEST path (mqc_cuest_scf.f90:637-1009): J from each
                    // spin density, K_s = sum_P (B_P C_s)(B_P C_s)^T; the kernels return the closed-shell 2 W W^T
                    const size_t tot = (size_t)nf * n * n;
                    launch_scale(bv.K, tot, 0.5, s);
                    if (nbeta > 0) {
                        BatchView vb = bv;
                        vb.D = bv.Db; vb.J = bv.Jb; vb.K = bv.Kb; vb.C = bv.Cb; vb.nocc = nbeta;
                        launch_df_jk(vb, true, s);
                        launch_scale(bv.Kb, tot, 0.5, s);
                    } else {
                        HIP_CHECK_RET(hipMemsetAsync(bv.Jb, 0, sizeof(double) * tot, s));
                        HIP_CHECK_RET(hipMemsetAsync(bv.Kb, 0, sizeof(double) * tot, s));
                    }
                }
            } else if (use_direct) {
                launch_jk_direct_incremental(bv, topo, direct_tol, true, s);        // restricted: G_ref += G(D - D_ref)
                if (uhf) {
                    // the integrals are formed again for the beta density: J[D_b], K[D_b] (twice the direct work)
                    BatchView vb = bv;
                    vb.D = bv.Db; vb.J = bv.Jb; vb.K = bv.Kb;
                    launch_jk_direct(vb, topo, direct_tol, true, s);
                }
            } else {
                launch_jk_incore(bv, true, s);
                if (uhf) {
                    // the same stream over the tensor with the beta density: J[D_b], K[D_b]
                    BatchView vb = bv;
                    vb.D = bv.Db; vb.J = bv.Jb; vb.K = bv.Kb;
                    launch_jk_incore(vb, true, s);
                }
            }
            HIP_CHECK_RET(hipEventRecord(sl.e1, s));
            if (guard == 0 && (rc = stage_check("J/K build")) != MQC_HIP_OK) return rc;
            if (xc_on_side) HIP_CHECK_RET(hipStreamWaitEvent(s, ctx->evo[sl.id & 1][1], 0));
            else if (xc.ncomp > 0) {
                HIP_CHECK_RET(hipEventRecord(sl.e2, s));
                launch_xc(bv, true, s);
                HIP_CHECK_RET(hipEventRecord(sl.e3, s));
            }
            HIP_CHECK_RET(hipEventRecord(sl.s0, s));
            launch_scf_step(bv, s);
            HIP_CHECK_RET(hipEventRecord(sl.s1, s));
            if (!last_of_block) { ++guard; continue; }
            HIP_CHECK_RET(hipMemcpyAsync(sl.h_counter, bv.counters, sizeof(int), hipMemcpyDeviceToHost, s));
            HIP_CHECK_RET(hipStreamSynchronize(s));
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, sl.s0, sl.s1);
            sx->stats.scf_step_seconds += ms * 1e-3 * block_len;
            (void)hipEventElapsedTime(&ms, sl.e0, sl.e1);
            if (use_df) {
                sx->stats.df_bytes += (double)remaining * 8.0 * (double)naux * (double)n * (double)n * block_len;
                sx->stats.df_flops += (double)remaining * 4.0 * (double)naux * (double)n * (double)n * (1.0 + (xc.exx != 0.0 ? (double)nocc : 0.0)) * block_len;
            }
            // density fitting: the J/K kernel reads the fitted tensor once, and it is stored packed [naux][npair] -- about
            // half of SURVEY 8d's 8 n^2 A (df_bytes keeps that figure; fock_bytes is what the kernel really streams)
            // in-core: the tensor as stored -- the square, or its lower triangle (BatchView::eri_tri) -- once per spin density;
            // the triangular kernel skips rows the Schwarz bounds prove zero and reports the chunks it did read
            if (bv.eri_tri && bv.jk_loaded && tri_doubles_per_fragment < 0.0) {
                std::vector<int> ld(nf);
                if (hipMemcpy(ld.data(), bv.jk_loaded, sizeof(int) * (size_t)nf, hipMemcpyDeviceToHost) == hipSuccess) {
                    double sum = 0.0;
                    for (int v : ld) sum += v;
                    tri_doubles_per_fragment = 128.0 * sum / (double)nf;
                } else tri_doubles_per_fragment = (double)bv.eri_stride;
            }
            const double tensor_doubles = (bv.eri_tri && tri_doubles_per_fragment >= 0.0) ? tri_doubles_per_fragment
                                                                                         : (bv.eri_stride ? (double)bv.eri_stride : (double)np * (double)np);
            const double launch_bytes = use_df ? (double)remaining * (double)naux * (double)np * 8.0
                                               : (double)remaining * tensor_doubles * 8.0 * (uhf ? 2.0 : 1.0);
            sx->stats.fock_kernel_seconds += ms * 1e-3 * block_len;
            sx->stats.fock_bytes += launch_bytes * block_len;
            sx->stats.fock_launches += block_len;
            if (launch_bytes >= 1073741824.0) {
                sx->stats.fock_big_launches += 1; sx->stats.fock_big_seconds += ms * 1e-3; sx->stats.fock_big_bytes += launch_bytes;
            }
            if (xc.ncomp > 0) {
                float mx = 0.f;
                (void)hipEventElapsedTime(&mx, sl.e2, sl.e3);
                sx->stats.xc_kernel_seconds += mx * 1e-3 * block_len;
                sx->stats.xc_points += (double)remaining * grid.npts * block_len;
                sx->stats.xc_flops += (double)remaining * grid.npts * (xc.gga ? 8.0 : 4.0) * (double)n * (double)n * block_len;
            }
            remaining = sl.h_counter[0];
            ++guard;
          }
        }
        if ((rc = stage_check("SCF loop")) != MQC_HIP_OK) return rc;
        const double t4 = now_s();
        sx->stats.t_fock += t4 - t3;

        std::vector<double> scal((size_t)nf * 8), eps((size_t)nf * n), dip((size_t)nf * 4);
        std::vector<int> ist((size_t)nf * 4);
        unsigned long long formed = 0;
        // ---- analytic gradient (compute_scf_gradient, mqc_cuest_gradient.f90:91-175): device terms here, E_nuc' on the host
        std::vector<double> hgrad;
        if (opts.want_gradient) {
            size_t lint = topo.pairs.size() + 64;
            for (auto& cl : topo.classes) lint += cl.quartets.size();
            const size_t nnh = (size_t)n * n;
            const size_t bytes = sizeof(double) * ((size_t)nf * topo.natoms * 3 + 2 * (size_t)nf * nnh + 8) + sizeof(int) * (lint + 64);
            char* gb = (char*)g_grad_pool[sl.id & 1].ensure(bytes);
            if (!gb) return fail(MQC_HIP_ERR_DEVICE, "out of device memory (gradient)");
            double* d_grad = (double*)gb;
            double* gwork = d_grad + (((size_t)nf * topo.natoms * 3 + 7) & ~size_t(7));
            int* glists = (int*)(gwork + 2 * (size_t)nf * nnh);
            std::string gerr;
            if (!launch_gradient(bv, topo, aux, d_grad, gwork, glists, lint, s, gerr)) return fail(MQC_HIP_ERR_UNSUPPORTED, gerr);
            hgrad.resize((size_t)nf * topo.natoms * 3);
            HIP_CHECK_RET(hipMemcpyAsync(hgrad.data(), d_grad, sizeof(double) * hgrad.size(), hipMemcpyDeviceToHost, s));
        }
        launch_dipole(bv, topo, s);
        std::vector<double> epsb;
        if (uhf) {
            BatchView vb = bv;
            vb.D = bv.Db;
            launch_dipole(vb, topo, s, true);      // total density = D_a + D_b
            epsb.resize((size_t)nf * n);
            HIP_CHECK_RET(hipMemcpyAsync(epsb.data(), bv.epsb, sizeof(double) * epsb.size(), hipMemcpyDeviceToHost, s));
        }
        HIP_CHECK_RET(hipMemcpyAsync(dip.data(), bv.dip, sizeof(double) * dip.size(), hipMemcpyDeviceToHost, s));
        HIP_CHECK_RET(hipMemcpyAsync(&formed, bv.eri_count, sizeof(formed), hipMemcpyDeviceToHost, s));
        HIP_CHECK_RET(hipMemcpyAsync(scal.data(), bv.scal, sizeof(double) * scal.size(), hipMemcpyDeviceToHost, s));
        HIP_CHECK_RET(hipMemcpyAsync(eps.data(), bv.eps, sizeof(double) * eps.size(), hipMemcpyDeviceToHost, s));
        HIP_CHECK_RET(hipMemcpyAsync(ist.data(), bv.istate, sizeof(int) * ist.size(), hipMemcpyDeviceToHost, s));
        // matrices the callers asked for (total density, embedding operator, overlap): the whole chunk in one copy each
        std::vector<double> allD, allU, allS;
        {
            bool want_d = embedded, want_s = false;
            for (int f = 0; f < nf; ++f) {
                const mqc_hip_scf_result_t* r = results[job.start + f];
                if (r->mulliken_charges) { want_d = true; want_s = true; }
                if (r->density) want_d = true;
            }
            const size_t tot = (size_t)nf * n * n;
            if (want_d) {
                allD.resize(tot);
                HIP_CHECK_RET(hipMemcpyAsync(allD.data(), bv.D, sizeof(double) * tot, hipMemcpyDeviceToHost, s));
            }
            if (embedded) {
                allU.resize(tot);
                HIP_CHECK_RET(hipMemcpyAsync(allU.data(), bv.U, sizeof(double) * tot, hipMemcpyDeviceToHost, s));
            }
            if (want_s) {
                allS.resize(tot);
                HIP_CHECK_RET(hipMemcpyAsync(allS.data(), bv.S, sizeof(double) * tot, hipMemcpyDeviceToHost, s));
            }
            if (want_d && uhf) {
                std::vector<double> db(tot);
                HIP_CHECK_RET(hipMemcpyAsync(db.data(), bv.Db, sizeof(double) * tot, hipMemcpyDeviceToHost, s));
                HIP_CHECK_RET(hipStreamSynchronize(s));
                for (size_t k = 0; k < tot; ++k) allD[k] += db[k];          // total density = alpha + beta
            }
        }
        HIP_CHECK_RET(hipStreamSynchronize(s));
        sx->stats.eri_survivors += (int64_t)formed;
        for (int f = 0; f < nf; ++f) {
            mqc_hip_scf_result_t* r = results[job.start + f];
            const int nmo = ist[4 * f + 2];
            r->n_ao = n; r->n_mo = nmo; r->n_occ = nocc;
            if (nocc > nmo) {
                // nmo check: more occupied orbitals than the basis supports after dropping near-null modes
                fill_error(r, uhf ? "UHF: more alpha electrons than the basis supports after near-null modes were dropped"
                                  : "RHF: more occupied orbitals than the basis supports after near-null modes were dropped");
                r->scf_status = MQC_HIP_SCF_NOT_RUN;
                continue;
            }
            r->e_nuclear = nuclear_repulsion(topo, xyz[job.start + f]);
            r->e_electronic = scal[8 * f + 4];
            r->e_total = r->e_electronic + r->e_nuclear;
            r->e_xc = xc.ncomp > 0 ? scal[8 * f + 5] : 0.0;
            r->iterations = ist[4 * f + 1];
            const bool conv = ist[4 * f + 3] != 0 && ist[4 * f] == ST_DONE;
            r->scf_status = conv ? MQC_HIP_SCF_CONVERGED : MQC_HIP_SCF_NOT_CONVERGED;
            r->homo = eps[(size_t)f * n + nocc - 1];
            r->lumo = nocc < nmo ? eps[(size_t)f * n + nocc] : 0.0;
            r->has_orbitals = 1;
            r->n_alpha = nalpha; r->n_beta = nbeta; r->s_squared = 0.0;
            if (opts.want_gradient && r->gradient) {
                // + nuclear repulsion: dE_nuc/dR_A = -sum_B Z_A Z_B (R_A - R_B)/|R_AB|^3 (ghosts carry no charge)
                const double* x = xyz[job.start + f];
                for (int a = 0; a < topo.natoms; ++a) {
                    double g3[3] = {hgrad[((size_t)f * topo.natoms + a) * 3], hgrad[((size_t)f * topo.natoms + a) * 3 + 1], hgrad[((size_t)f * topo.natoms + a) * 3 + 2]};
                    for (int b = 0; b < topo.natoms; ++b) {
                        if (b == a || topo.zeff[a] == 0.0 || topo.zeff[b] == 0.0) continue;
                        const double dx = x[3 * a] - x[3 * b], dy = x[3 * a + 1] - x[3 * b + 1], dz = x[3 * a + 2] - x[3 * b + 2];
                        const double r2 = dx * dx + dy * dy + dz * dz, r3 = r2 * std::sqrt(r2);
                        const double zz = topo.zeff[a] * topo.zeff[b] / r3;
                        g3[0] -= zz * dx; g3[1] -= zz * dy; g3[2] -= zz * dz;
                    }
                    r->gradient[3 * a] = g3[0]; r->gradient[3 * a + 1] = g3[1]; r->gradient[3 * a + 2] = g3[2];
                }
                r->has_gradient = 1;
            }
            if (uhf) {
                // <S^2> = S_z (S_z + 1) + n_beta - sum_ij |<a_i|S|b_j>|^2  (spin_contamination, mqc_libcint_rhf.f90)
                const size_t nnh = (size_t)n * n;
                std::vector<double> Ca(nnh), Cb(nnh), Sm(nnh);
                HIP_CHECK_RET(hipMemcpyAsync(Ca.data(), bv.C + (size_t)f * nnh, sizeof(double) * nnh, hipMemcpyDeviceToHost, s));
                HIP_CHECK_RET(hipMemcpyAsync(Cb.data(), bv.Cb + (size_t)f * nnh, sizeof(double) * nnh, hipMemcpyDeviceToHost, s));
                HIP_CHECK_RET(hipMemcpyAsync(Sm.data(), bv.S + (size_t)f * nnh, sizeof(double) * nnh, hipMemcpyDeviceToHost, s));
                HIP_CHECK_RET(hipStreamSynchronize(s));
                double ov2 = 0.0;
                std::vector<double> SCb((size_t)n * std::max(nbeta, 1));
                for (int mu = 0; mu < n; ++mu)
                    for (int j = 0; j < nbeta; ++j) {
                        double t = 0.0;
                        for (int nu = 0; nu < n; ++nu) t += Sm[(size_t)mu * n + nu] * Cb[(size_t)nu * n + j];
                        SCb[(size_t)mu * nbeta + j] = t;
                    }
                for (int i = 0; i < nalpha; ++i)
                    for (int j = 0; j < nbeta; ++j) {
                        double t = 0.0;
                        for (int mu = 0; mu < n; ++mu) t += Ca[(size_t)mu * n + i] * SCb[(size_t)mu * nbeta + j];
                        ov2 += t * t;
                    }
                const double sz = 0.5 * (nalpha - nbeta);
                r->s_squared = sz * (sz + 1.0) + nbeta - ov2;
                if (r->orbital_energies_beta) std::memcpy(r->orbital_energies_beta, &epsb[(size_t)f * n], sizeof(double) * nmo);
            }
            {
                // mu = sum_A Z_A (R_A - O) - [tr(D r) - O tr(D S)], O = centre of nuclear charge, tr(D S) = N_electrons
                // (system_compute_dipole, mqc_cuest_integrals.f90:1443-1521)
                const double* x = xyz[job.start + f];
                double ztot = 0.0, o[3] = {0, 0, 0}, mu[3] = {0, 0, 0};
                for (int a = 0; a < topo.natoms; ++a) { ztot += topo.zeff[a]; for (int c = 0; c < 3; ++c) o[c] += topo.zeff[a] * x[3 * a + c]; }
                if (ztot > 0.0) for (int c = 0; c < 3; ++c) o[c] /= ztot; else for (int c = 0; c < 3; ++c) o[c] = 0.0;
                for (int a = 0; a < topo.natoms; ++a) for (int c = 0; c < 3; ++c) mu[c] += topo.zeff[a] * (x[3 * a + c] - o[c]);
                for (int c = 0; c < 3; ++c) r->dipole[c] = mu[c] - (dip[4 * f + c] - o[c] * (double)topo.nelec);
                r->has_dipole = 1;
            }
            if (r->orbital_energies) std::memcpy(r->orbital_energies, &eps[(size_t)f * n], sizeof(double) * nmo);
            if (embedded || r->mulliken_charges) {
                // what the embedded callers read besides the energy: tr(D u), u itself, Mulliken populations
                // (inner_scf / fragment_charges, mqc_libcint_fmo.f90:1992-2021) -- n^2 host work on the matrices of the
                // whole chunk, copied back ONCE above (one synchronous copy per matrix and fragment cost seconds of
                // a 130 000-pair batch)
                const size_t nn = (size_t)n * n;
                const double* hd = allD.data() + (size_t)f * nn;
                if (embedded) {
                    const double* hu = allU.data() + (size_t)f * nn;
                    double e = 0.0;
                    for (size_t k = 0; k < nn; ++k) e += hd[k] * hu[k];
                    r->e_embedding = e;
                    if (r->embedding_matrix) std::memcpy(r->embedding_matrix, hu, sizeof(double) * nn);
                }
                if (r->mulliken_charges) {
                    const double* hm = allS.data() + (size_t)f * nn;
                    for (int a = 0; a < topo.natoms; ++a) r->mulliken_charges[a] = topo.zeff[a];
                    for (size_t sh = 0; sh < topo.shells.size(); ++sh) {
                        const int a = topo.shells[sh].atom, o0 = topo.shells[sh].aoff, nf_sh = 2 * topo.shells[sh].l + 1;
                        for (int mu = o0; mu < o0 + nf_sh; ++mu) {
                            double pop = 0.0;
                            for (int nu = 0; nu < n; ++nu) pop += hd[(size_t)mu * n + nu] * hm[(size_t)nu * n + mu];
                            r->mulliken_charges[a] -= pop;
                        }
                    }
                }
            }
            if (r->density) std::memcpy(r->density, allD.data() + (size_t)f * n * n, sizeof(double) * n * n);
            r->has_error = 0; r->message[0] = '\0';
            if (use_df && scal[8 * f + 7] == 1.0)
                fill_error(r, "density fitting: the auxiliary metric (P|Q) could not be factorised or diagonalised");
            else if (!std::isfinite(r->e_total)) fill_error(r, "SCF produced a non-finite energy");
            else if (!conv && !opts.allow_crap_scf)
                fill_error(r, "SCF did not converge in " + std::to_string(r->iterations) + " iterations");
            sx->stats.scf_iterations_total += r->iterations;
        }
        sx->stats.t_scf_step += now_s() - t4;
        return MQC_HIP_OK;
    };

    const int njobs = (int)jobs.size();
    if (lane >= 0) {
        // one slot only: chunks strictly one after the other
        Slot& sl = slots[lane & 1];
        for (int k = 0; k < njobs; ++k) {
            if ((rc = prepare(sl, jobs[k])) != MQC_HIP_OK) return bail(rc);
            if ((rc = iterate_and_fetch(sl, jobs[k])) != MQC_HIP_OK) return bail(rc);
            std::vector<double>().swap(jobs[k].hx);
        }
    } else {
        if ((rc = prepare(slots[0], jobs[0])) != MQC_HIP_OK) return bail(rc);
        for (int k = 0; k < njobs; ++k) {
            // chunk k+1's integrals go onto the other stream before the host starts iterating chunk k
            if (k + 1 < njobs && (rc = prepare(slots[(k + 1) & 1], jobs[k + 1])) != MQC_HIP_OK) return bail(rc);
            if ((rc = iterate_and_fetch(slots[k & 1], jobs[k])) != MQC_HIP_OK) return bail(rc);
            std::vector<double>().swap(jobs[k].hx);
        }
    }
    (void)hipHostFree(h_counter);
    sx->stats.t_total += now_s() - t_begin;
    {
        std::lock_guard<std::mutex> lock(ctx->stats_mutex);
        Stats& g = ctx->stats;
        const Stats& l = sx->stats;
        g.t_setup += l.t_setup; g.t_int1e += l.t_int1e; g.t_eri += l.t_eri; g.t_fock += l.t_fock; g.t_scf_step += l.t_scf_step;
        g.t_total += l.t_total; g.fock_launches += l.fock_launches; g.eri_quartets += l.eri_quartets;
        g.scf_iterations_total += l.scf_iterations_total; g.fock_kernel_seconds += l.fock_kernel_seconds;
        g.fock_bytes += l.fock_bytes; g.eri_kernel_seconds += l.eri_kernel_seconds;
        g.xc_kernel_seconds += l.xc_kernel_seconds; g.xc_points += l.xc_points;
        g.fock_big_launches += l.fock_big_launches; g.fock_big_seconds += l.fock_big_seconds; g.fock_big_bytes += l.fock_big_bytes;
        g.xc_flops += l.xc_flops; g.scf_step_seconds += l.scf_step_seconds; g.eri_survivors += l.eri_survivors;
        g.df_flops += l.df_flops; g.df_bytes += l.df_bytes;
    }
    return MQC_HIP_OK;
}

}  // namespace mqc

// =========================================================================================
using namespace mqc;

// ---- superposed free atoms (mqc_libcint_atomic_guess.f90) ---------------------------------------------------------
// Ground-state multiplicity of a free atom by Hund's first rule over the Madelung filling
// (hund_multiplicity, src/core/mqc_atomic_guess_common.f90:19-54).
static int hund_multiplicity(int z)
{
    static const int cap[16] = {2, 2, 6, 2, 6, 2, 10, 6, 2, 10, 6, 2, 14, 10, 6, 2};
    int remaining = z, unpaired = 0;
    for (int i = 0; i < 16 && remaining > 0; ++i) {
        const int deg = cap[i] / 2, in_shell = std::min(remaining, cap[i]);
        remaining -= in_shell;
        unpaired = in_shell <= deg ? in_shell : 2 * deg - in_shell;
    }
    return unpaired + 1;
}

// eigen-decomposition of a small symmetric matrix (cyclic Jacobi, host): a[n*n] -> eigenvalues w, vectors v (columns)
static void host_jacobi(int n, std::vector<double>& a, std::vector<double>& w, std::vector<double>& v)
{
    v.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) v[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0.0;
        for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) off += a[(size_t)i * n + j] * a[(size_t)i * n + j];
        if (off < 1e-30) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a[(size_t)p * n + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double th = (a[(size_t)q * n + q] - a[(size_t)p * n + p]) / (2.0 * apq);
                const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
                    a[(size_t)k * n + p] = c * akp - sn * akq; a[(size_t)k * n + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
                    a[(size_t)p * n + k] = c * apk - sn * aqk; a[(size_t)q * n + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = v[(size_t)k * n + p], vkq = v[(size_t)k * n + q];
                    v[(size_t)k * n + p] = c * vkp - sn * vkq; v[(size_t)k * n + q] = sn * vkp + c * vkq;
                }
            }
    }
    w.resize(n);
    for (int i = 0; i < n; ++i) w[i] = a[(size_t)i * n + i];
}

// One free atom per distinct (element, shells): unrestricted Hartree-Fock at Hund's multiplicity in exactly the basis
// functions the atom contributes, GWH start, 1e-8 / 1e-6, 200 cycles (solve_free_atom :380-429) -- run through this
// same engine as a one-atom fragment -- cached for the life of the context; the blocks are dropped on the diagonal
// (build_atomic_guess :295-378; ghosts carry nothing).  SAD: the spherical average of the total density
// (spherical_average :235-293; spherical bases only reach this engine); SAC: the total as converged.
static int build_atomic_guess(mqc_hip_context* ctx, const mqc_hip_molecule_t& mol, const mqc_hip_basis_t& bas, const Topology& topo,
                              int kind, bool with_pseudo_orbitals, AtomicGuess& out, std::string& err)
{
    const int n = topo.nao;
    out.D0.assign((size_t)n * n, 0.0);
    size_t sh0 = 0, pr0 = 0;
    int ao0 = 0;
    for (int a = 0; a < mol.n_atoms; ++a) {
        const int ns = (int)bas.nshell_per_atom[a];
        size_t npr = 0;
        int nao_a = 0;
        for (int k = 0; k < ns; ++k) { npr += bas.shell_nprim[sh0 + k]; nao_a += 2 * bas.shell_l[sh0 + k] + 1; }
        const int z = mol.atomic_numbers[a];
        const bool ghost = mol.ghost && mol.ghost[a];
        if (z > 0 && !ghost && ns > 0) {
            // the key is the element and the very shells of this atom
            std::ostringstream ks;
            ks << z << ':';
            uint64_t h = 1469598103934665603ull;
            auto mix = [&h](const void* p, size_t nb) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < nb; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
            for (int k = 0; k < ns; ++k) ks << bas.shell_l[sh0 + k] << '.' << bas.shell_nprim[sh0 + k] << ',';
            mix(bas.exponents + pr0, sizeof(double) * npr); mix(bas.coefficients + pr0, sizeof(double) * npr);
            ks << '#' << h;
            auto it = ctx->atom_cache.find(ks.str());
            std::shared_ptr<std::vector<double>> dens;
            if (it != ctx->atom_cache.end()) dens = it->second;
            else {
                const int32_t zz = z; const double origin[3] = {0.0, 0.0, 0.0}; const int64_t nsa = ns;
                mqc_hip_molecule_t am{}; am.n_atoms = 1; am.atomic_numbers = &zz; am.xyz = origin; am.ghost = nullptr;
                am.charge = 0; am.multiplicity = hund_multiplicity(z); am.nelec = z;
                mqc_hip_basis_t ab{}; ab.spherical = 1; ab.n_atoms = 1; ab.nshell_per_atom = &nsa; ab.n_shells = ns;
                ab.shell_l = bas.shell_l + sh0; ab.shell_nprim = bas.shell_nprim + sh0; ab.exponents = bas.exponents + pr0; ab.coefficients = bas.coefficients + pr0;
                mqc_hip_scf_options_t ao; mqc_hip_default_options(&ao);
                ao.unrestricted = 1; ao.guess = MQC_HIP_GUESS_GWH; ao.energy_tol = 1.0e-8; ao.density_tol = 1.0e-6; ao.max_iter = 200;
                ao.use_diis = 1; ao.diis_size = 8; ao.density_fitting = 0; ao.functional[0] = 0; ao.want_gradient = 0;
                dens = std::make_shared<std::vector<double>>((size_t)nao_a * nao_a, 0.0);
                mqc_hip_scf_result_t ar; std::memset(&ar, 0, sizeof(ar));
                ar.density = dens->data();
                const int rc = mqc_hip_scf_run_batch(ctx, 1, &am, &ab, nullptr, &ao, &ar);
                if (rc != MQC_HIP_OK || ar.has_error || ar.scf_status != MQC_HIP_SCF_CONVERGED) {
                    err = "atomic guess: the free atom Z=" + std::to_string(z) + " did not converge (" + std::string(ar.message) + ")";
                    return MQC_HIP_ERR_VALIDATION;
                }
                if (ctx->atom_cache.size() >= 64) { err = "atomic guess: solution cache exhausted"; return MQC_HIP_ERR_VALIDATION; }
                ctx->atom_cache[ks.str()] = dens;
            }
            // this atom's block: as converged (SAC) or averaged over m within each pair of subshells of equal l (SAD)
            std::vector<int> first(ns), ang(ns);
            { int o = 0; for (int k = 0; k < ns; ++k) { first[k] = o; ang[k] = bas.shell_l[sh0 + k]; o += 2 * ang[k] + 1; } }
            const std::vector<double>& d = *dens;
            if (kind == MQC_HIP_GUESS_SAC) {
                for (int i = 0; i < nao_a; ++i) for (int j = 0; j < nao_a; ++j) out.D0[(size_t)(ao0 + i) * n + ao0 + j] = d[(size_t)i * nao_a + j];
            } else {
                for (int ka = 0; ka < ns; ++ka)
                    for (int kb = 0; kb < ns; ++kb) {
                        if (ang[ka] != ang[kb]) continue;
                        const int nc = 2 * ang[ka] + 1;
                        double mean = 0.0;
                        for (int m = 0; m < nc; ++m) mean += d[(size_t)(first[ka] + m) * nao_a + first[kb] + m];
                        mean /= nc;
                        for (int m = 0; m < nc; ++m) out.D0[(size_t)(ao0 + first[ka] + m) * n + ao0 + first[kb] + m] = mean;
                    }
            }
        }
        sh0 += ns; pr0 += npr; ao0 += nao_a;
    }
    if (ao0 != n) { err = "atomic guess: the atoms' basis functions do not add up to the molecule's"; return MQC_HIP_ERR_VALIDATION; }
    out.nmodes = 0;
    out.Cp.assign((size_t)n * n, 0.0);
    if (with_pseudo_orbitals) {
        std::vector<double> a = out.D0, w, v;
        host_jacobi(n, a, w, v);
        for (int i = 0; i < n; ++i) {
            if (!(w[i] > 1.0e-12)) continue;                  // OCCUPATION_FLOOR
            const double sc = std::sqrt(0.5 * w[i]);
            for (int mu = 0; mu < n; ++mu) out.Cp[(size_t)mu * n + out.nmodes] = v[(size_t)mu * n + i] * sc;
            out.nmodes += 1;
        }
        if (out.nmodes == 0) { err = "atomic guess: the guess density carries no occupation"; return MQC_HIP_ERR_VALIDATION; }
    }
    return MQC_HIP_OK;
}

static mqc_hip_context* g_ctx = nullptr;

extern "C" {

int mqc_hip_abi_version(void) { return MQC_HIP_ABI_VERSION; }

const char* mqc_hip_last_error(void) { return last_error_string().c_str(); }

// The engine keeps two lanes of 1 + 3 (or 1 + 7) streams busy at once; with the HIP runtime's default of four hardware
// queues per process they share queues and serialise (measured on one box: 138.4 ms per evaluation at 4 queues, 132.8 at
// 8, 128.2 at 16).  The runtime reads GPU_MAX_HW_QUEUES when it initialises, so the library asks for 16 before its
// first HIP call unless the variable is already set; a host that initialised HIP earlier exports it in the job script.
static void ask_for_hardware_queues()
{
    static const bool done = [] { (void)setenv("GPU_MAX_HW_QUEUES", "16", 0); return true; }();
    (void)done;
}

int mqc_hip_backend_available(void)
{
    ask_for_hardware_queues();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n > 0 ? 1 : 0;
}

void mqc_hip_default_options(mqc_hip_scf_options_t* o)
{
    std::memset(o, 0, sizeof(*o));
    o->grid_level = 3;
    o->max_iter = 100;            // src/methods/mqc_method_config.f90:24-29
    o->energy_tol = 1.0e-8;
    o->density_tol = 1.0e-6;
    o->use_diis = 1;
    o->diis_size = 8;
    o->guess = MQC_HIP_GUESS_AUTO;
    o->eri_mode = MQC_HIP_ERI_AUTO;
    o->schwarz_tol = 0.0;
}

int mqc_hip_context_get(int32_t local_rank, mqc_hip_context** out)
{
    if (!out) return fail(MQC_HIP_ERR_VALIDATION, "null context pointer");
    if (g_ctx) { *out = g_ctx; return MQC_HIP_OK; }
    ask_for_hardware_queues();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MQC_HIP_ERR_NO_DEVICE, "no HIP device is visible: the MI355X backend has no CPU fallback");
    auto* ctx = new mqc_hip_context();
    ctx->device = ((local_rank % ndev) + ndev) % ndev;    // mqc_cuest_context.f90:188
    HIP_CHECK_RET(hipSetDevice(ctx->device));
    HIP_CHECK_RET(hipGetDeviceProperties(&ctx->prop, ctx->device));
    // creation order matters: streams take the 4 hardware queues round-robin, so each lane's main stream and its
    // three side streams land on four different queues
    HIP_CHECK_RET(hipStreamCreate(&ctx->stream));
    for (int k = 0; k < 3; ++k) HIP_CHECK_RET(hipStreamCreateWithFlags(&ctx->side[0][k], hipStreamNonBlocking));
    HIP_CHECK_RET(hipStreamCreate(&ctx->stream2));
    for (int k = 0; k < 3; ++k) HIP_CHECK_RET(hipStreamCreateWithFlags(&ctx->side[1][k], hipStreamNonBlocking));
    for (int l = 0; l < 2; ++l) for (int k = 0; k < 2; ++k) HIP_CHECK_RET(hipEventCreateWithFlags(&ctx->evo[l][k], hipEventDisableTiming));
    for (int l = 0; l < 2; ++l) for (int k = 0; k < 2; ++k) HIP_CHECK_RET(hipEventCreate(&ctx->evs[l][k]));
    eri_set_side_streams(0, ctx->side[0], 3);
    eri_set_side_streams(1, ctx->side[1], 3);
    for (hipEvent_t* e : {&ctx->evb0, &ctx->evb1, &ctx->evb2, &ctx->evb3, &ctx->evq0, &ctx->evq1, &ctx->evq2, &ctx->evq3})
        HIP_CHECK_RET(hipEventCreate(e));
    HIP_CHECK_RET(hipEventCreate(&ctx->ev0));
    HIP_CHECK_RET(hipEventCreate(&ctx->ev1));
    HIP_CHECK_RET(hipEventCreate(&ctx->ev2));
    HIP_CHECK_RET(hipEventCreate(&ctx->ev3));
    std::vector<double> boys;
    build_boys_table(boys);
    HIP_CHECK_RET(hipMalloc((void**)&ctx->d_boys, sizeof(double) * boys.size()));
    HIP_CHECK_RET(hipMemcpy(ctx->d_boys, boys.data(), sizeof(double) * boys.size(), hipMemcpyHostToDevice));
    {
        const double unit[2] = {0.0, 1.0};
        HIP_CHECK_RET(hipMalloc((void**)&ctx->d_unit, sizeof(unit)));
        HIP_CHECK_RET(hipMemcpy(ctx->d_unit, unit, sizeof(unit), hipMemcpyHostToDevice));
    }
    build_c2s_tables(ctx->h_c2s, ctx->c2s_off);
    HIP_CHECK_RET(hipMalloc((void**)&ctx->d_c2s, sizeof(double) * ctx->h_c2s.size()));
    HIP_CHECK_RET(hipMemcpy(ctx->d_c2s, ctx->h_c2s.data(), sizeof(double) * ctx->h_c2s.size(), hipMemcpyHostToDevice));
    const char* env = std::getenv("MQC_HIP_HBM_BUDGET_GB");
    if (env) ctx->hbm_budget_bytes = (size_t)(std::atof(env) * 1024.0 * 1024.0 * 1024.0);
    // Two-stream chunk pipeline: measured on (H2O)64 MBE-2 RHF/cc-pVDZ it does not pay (1 chunk 261 ms,
    // 2: 251 ms, 4: 281 ms, 8: 340 ms per evaluation -- the ERI kernels and the J/K stream each fill the
    // chip, co-running them only slows both), so it is off unless MQC_HIP_PIPELINE_CHUNKS asks for it.
    // Batches that exceed the HBM budget still alternate between the two slots.
    ctx->pipeline_chunks = 1;
    // MQC_HIP_CONCURRENT_GROUPS=0: topology groups of a batch call run one after the other
    if (const char* cg = std::getenv("MQC_HIP_CONCURRENT_GROUPS")) ctx->concurrent_groups = std::atoi(cg) != 0;
    if (const char* pc = std::getenv("MQC_HIP_PIPELINE_CHUNKS")) ctx->pipeline_chunks = std::max(1, std::atoi(pc));
    if (const char* pm = std::getenv("MQC_HIP_PIPELINE_MIN_FRAGMENTS")) ctx->pipeline_min_fragments = std::max(2, std::atoi(pm));
    g_ctx = ctx;
    *out = ctx;
    return MQC_HIP_OK;
}

int mqc_hip_finalize(void)
{
    if (!g_ctx) return MQC_HIP_OK;
    (void)hipSetDevice(g_ctx->device);
    (void)hipStreamSynchronize(g_ctx->stream);
    (void)hipStreamSynchronize(g_ctx->stream2);
    for (int l = 0; l < 2; ++l) for (int k = 0; k < 3; ++k) if (g_ctx->side[l][k]) (void)hipStreamSynchronize(g_ctx->side[l][k]);
    // launcher state bound to this device (side streams, fork/join events, list caches), then every pool:
    // the context's own and the launchers' function-static ones -- a later context_get starts from nothing
    eri_reset_state();
    int1e_reset_state();
    release_all_pools();
    for (int l = 0; l < 2; ++l) {
        for (int k = 0; k < 3; ++k) if (g_ctx->side[l][k]) (void)hipStreamDestroy(g_ctx->side[l][k]);
        for (int k = 0; k < 2; ++k) { if (g_ctx->evo[l][k]) (void)hipEventDestroy(g_ctx->evo[l][k]); if (g_ctx->evs[l][k]) (void)hipEventDestroy(g_ctx->evs[l][k]); }
    }
    if (g_ctx->d_unit) (void)hipFree(g_ctx->d_unit);
    if (g_ctx->d_boys) (void)hipFree(g_ctx->d_boys);
    if (g_ctx->d_c2s) (void)hipFree(g_ctx->d_c2s);
    for (hipEvent_t e : {g_ctx->ev0, g_ctx->ev1, g_ctx->ev2, g_ctx->ev3, g_ctx->evb0, g_ctx->evb1, g_ctx->evb2, g_ctx->evb3,
                         g_ctx->evq0, g_ctx->evq1, g_ctx->evq2, g_ctx->evq3})
        (void)hipEventDestroy(e);
    (void)hipStreamDestroy(g_ctx->stream);
    (void)hipStreamDestroy(g_ctx->stream2);
    delete g_ctx;
    g_ctx = nullptr;
    return MQC_HIP_OK;
}

int mqc_hip_device_name(mqc_hip_context* ctx, char* buf, int32_t len)
{
    if (!ctx || !buf || len <= 0) return fail(MQC_HIP_ERR_VALIDATION, "bad arguments");
    std::snprintf(buf, (size_t)len, "%s (%s, %d CUs)", ctx->prop.name, ctx->prop.gcnArchName, ctx->prop.multiProcessorCount);
    return MQC_HIP_OK;
}

int mqc_hip_get_stats(mqc_hip_context* ctx, mqc_hip_stats_t* st)
{
    if (!ctx || !st) return fail(MQC_HIP_ERR_VALIDATION, "bad arguments");
    std::lock_guard<std::mutex> lock(ctx->stats_mutex);
    st->t_setup = ctx->stats.t_setup; st->t_int1e = ctx->stats.t_int1e; st->t_eri = ctx->stats.t_eri;
    st->t_fock = ctx->stats.t_fock; st->t_scf_step = ctx->stats.t_scf_step; st->t_total = ctx->stats.t_total;
    st->fock_launches = ctx->stats.fock_launches; st->eri_quartets = ctx->stats.eri_quartets;
    st->scf_iterations_total = ctx->stats.scf_iterations_total;
    st->fock_kernel_seconds = ctx->stats.fock_kernel_seconds; st->fock_bytes = ctx->stats.fock_bytes;
    st->eri_kernel_seconds = ctx->stats.eri_kernel_seconds;
    st->xc_kernel_seconds = ctx->stats.xc_kernel_seconds; st->xc_points = ctx->stats.xc_points;
    st->fock_big_launches = ctx->stats.fock_big_launches; st->fock_big_seconds = ctx->stats.fock_big_seconds;
    st->fock_big_bytes = ctx->stats.fock_big_bytes;
    st->xc_flops = ctx->stats.xc_flops; st->scf_step_seconds = ctx->stats.scf_step_seconds;
    st->eri_survivors = ctx->stats.eri_survivors; st->df_flops = ctx->stats.df_flops; st->df_bytes = ctx->stats.df_bytes;
    ctx->stats = Stats();
    return MQC_HIP_OK;
}

static void init_result(mqc_hip_scf_result_t* r)
{
    double* oe = r->orbital_energies; double* dn = r->density; double* gr = r->gradient; double* ob = r->orbital_energies_beta;
    double* em = r->embedding_matrix; double* mq = r->mulliken_charges;
    std::memset(r, 0, sizeof(*r));
    r->orbital_energies = oe; r->density = dn; r->gradient = gr; r->orbital_energies_beta = ob;
    r->embedding_matrix = em; r->mulliken_charges = mq;
    r->scf_status = MQC_HIP_SCF_NOT_RUN;
}

int mqc_hip_scf_run_batch(mqc_hip_context* ctx, int64_t nfrag, const mqc_hip_molecule_t* mols,
                          const mqc_hip_basis_t* orbitals, const mqc_hip_basis_t* auxes,
                          const mqc_hip_scf_options_t* opts, mqc_hip_scf_result_t* results)
{
    if (!ctx) return fail(MQC_HIP_ERR_VALIDATION, "null context (call mqc_hip_context_get first)");
    if (nfrag < 0 || (nfrag > 0 && (!mols || !orbitals || !opts || !results)))
        return fail(MQC_HIP_ERR_VALIDATION, "null argument");
    HIP_CHECK_RET(hipSetDevice(ctx->device));
    for (int64_t i = 0; i < nfrag; ++i) init_result(&results[i]);
    // group by topology
    std::map<std::string, std::vector<int64_t>> groups;
    int64_t last = -1;
    std::string last_key;
    std::vector<int64_t>* last_group = nullptr;
    for (int64_t i = 0; i < nfrag; ++i) {
        if (!mols[i].atomic_numbers || !mols[i].xyz || mols[i].n_atoms <= 0 || !orbitals[i].shell_l ||
            !orbitals[i].nshell_per_atom || !orbitals[i].shell_nprim || !orbitals[i].exponents || !orbitals[i].coefficients) {
            results[i].has_error = 1;
            std::snprintf(results[i].message, sizeof(results[i].message), "fragment has no geometry or basis");
            continue;
        }
        // consecutive fragments that point at the very same element and basis arrays share their key
        const bool same_as_last = last >= 0 && mols[i].n_atoms == mols[last].n_atoms && mols[i].atomic_numbers == mols[last].atomic_numbers &&
                                  mols[i].ghost == mols[last].ghost && mols[i].nelec == mols[last].nelec && mols[i].charge == mols[last].charge &&
                                  mols[i].multiplicity == mols[last].multiplicity && mols[i].n_point_charges == mols[last].n_point_charges && (mols[i].h_extra != nullptr) == (mols[last].h_extra != nullptr) &&
                                  std::memcmp(&orbitals[i], &orbitals[last], sizeof(mqc_hip_basis_t)) == 0 &&
                                  (!(opts->density_fitting && auxes) || std::memcmp(&auxes[i], &auxes[last], sizeof(mqc_hip_basis_t)) == 0);
        if (!same_as_last) {
            last_key = topology_key(mols[i], orbitals[i]);
            if (opts->density_fitting && auxes) last_key += "//" + topology_key(mols[i], auxes[i]);
            if (mols[i].n_point_charges > 0) last_key += "//pc" + std::to_string(mols[i].n_point_charges);
            if (mols[i].h_extra) last_key += "//hx";
            last_group = &groups[last_key];
        }
        last = i;
        last_group->push_back(i);
    }
    int worst = MQC_HIP_OK;
    // ---- per topology group: topology from the cache (or built now), then the batch.  With several groups
    // (monomers and dimers of an MBE list) two run at a time, each on its own slot: the small group's
    // latency-bound stages hide behind the large one.
    struct Work {
        const std::vector<int64_t>* idx;
        std::shared_ptr<Topology> topo, aux;
        std::vector<const double*> xyz;
        std::vector<const mqc_hip_molecule_t*> mol;
        std::vector<mqc_hip_scf_result_t*> res;
        std::shared_ptr<AtomicGuess> guess;
        int rc = MQC_HIP_OK;
        std::string msg;
    };
    std::vector<Work> work;
    auto cached_topology = [&](const std::string& key, const mqc_hip_molecule_t& mol, const mqc_hip_basis_t& bas, int max_l,
                               bool quartets, std::shared_ptr<Topology>& out, std::string& err) -> int {
        const std::string k = key + (quartets ? "|q" : "|n") + std::to_string(max_l);
        auto it = ctx->topo_cache.find(k);
        if (it != ctx->topo_cache.end()) { out = it->second; return MQC_HIP_OK; }
        auto t = std::make_shared<Topology>();
        const int rc = build_topology(mol, bas, *t, err, max_l, quartets);
        if (rc != MQC_HIP_OK) return rc;
        if (ctx->topo_cache.size() >= 16) ctx->topo_cache.clear();
        ctx->topo_cache[k] = t;
        out = t;
        return MQC_HIP_OK;
    };
    for (auto& kv : groups) {
        const auto& idx = kv.second;
        Work w;
        w.idx = &idx;
        std::string err;
        const bool need_quartets = !(opts->density_fitting && auxes);
        int rc = cached_topology(topology_key(mols[idx[0]], orbitals[idx[0]]), mols[idx[0]], orbitals[idx[0]], KERNEL_LMAX, need_quartets, w.topo, err);
        if (rc != MQC_HIP_OK) {
            for (auto i : idx) { results[i].has_error = 1; std::snprintf(results[i].message, sizeof(results[i].message), "%s", err.c_str()); }
            set_error(err);
            worst = rc;
            continue;
        }
        if (opts->density_fitting && auxes) {
            rc = cached_topology(topology_key(mols[idx[0]], auxes[idx[0]]), mols[idx[0]], auxes[idx[0]], AUX_LMAX, false, w.aux, err);
            if (rc != MQC_HIP_OK) {
                for (auto i : idx) { results[i].has_error = 1; std::snprintf(results[i].message, sizeof(results[i].message), "auxiliary basis: %s", err.c_str()); }
                set_error(err);
                worst = rc;
                continue;
            }
        }
        if (opts->guess == MQC_HIP_GUESS_SAD || opts->guess == MQC_HIP_GUESS_SAC) {
            // the free atoms are solved here, before any group of this call holds the pools (nested one-atom calls)
            w.guess = std::make_shared<AtomicGuess>();
            rc = build_atomic_guess(ctx, mols[idx[0]], orbitals[idx[0]], *w.topo, opts->guess, opts->density_fitting != 0, *w.guess, err);
            if (rc != MQC_HIP_OK) {
                // "a guess that will not build is a reason to start elsewhere, not to fail the run" (:175-178): GWH, loudly
                std::fprintf(stderr, "mqc_hip: initial guess: %s -- falling back to gwh\n", err.c_str());
                w.guess.reset();
            }
        }
        for (auto i : idx) { w.xyz.push_back(mols[i].xyz); w.mol.push_back(&mols[i]); w.res.push_back(&results[i]); }
        work.push_back(std::move(w));
    }
    auto run_one = [&](Work& w, int lane) {
        (void)hipSetDevice(ctx->device);
        w.rc = run_batch(ctx, *w.topo, w.aux.get(), w.xyz, *opts, w.res, lane, w.guess.get(), &w.mol);
        if (w.rc != MQC_HIP_OK) w.msg = mqc_hip_last_error();      // the error text is thread-local
    };
    if (work.size() >= 2 && ctx->concurrent_groups) {
        // largest group first on lane 0; the others queue up on lane 1, then lane 0 helps with what is left
        std::sort(work.begin(), work.end(), [](const Work& a, const Work& b) { return a.xyz.size() > b.xyz.size(); });
        std::atomic<size_t> next{0};
        auto worker = [&](int lane) {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= work.size()) return;
                run_one(work[k], lane);
            }
        };
        std::thread t1(worker, 1);
        worker(0);
        t1.join();
    } else {
        for (auto& w : work) run_one(w, -1);
    }
    for (auto& w : work) {
        if (w.rc == MQC_HIP_OK) continue;
        worst = w.rc;
        set_error(w.msg);
        for (auto* r : w.res)
            if (!r->has_error) { r->has_error = 1; std::snprintf(r->message, sizeof(r->message), "%s", w.msg.c_str()); }
    }
    // single-fragment calls report the fragment's failure as the call's status, like run_cuest_scf
    if (nfrag == 1 && worst != MQC_HIP_OK) return worst;
    return (nfrag == 1) ? MQC_HIP_OK : worst;
}

int mqc_hip_scf_run(mqc_hip_context* ctx, const mqc_hip_molecule_t* mol, const mqc_hip_basis_t* orbital,
                    const mqc_hip_basis_t* aux, const mqc_hip_scf_options_t* opts, mqc_hip_scf_result_t* result)
{
    return mqc_hip_scf_run_batch(ctx, 1, mol, orbital, aux, opts, result);
}

// ---- stage-level entry points -----------------------------------------------------------
struct StageBatch {
    Topology topo;
    TopologyDev td;
    BatchView bv{};
};

static int stage_setup(mqc_hip_context* ctx, const mqc_hip_molecule_t* mol, const mqc_hip_basis_t* bas, bool with_eri, StageBatch& sb)
{
    if (!ctx || !mol || !bas) return fail(MQC_HIP_ERR_VALIDATION, "null argument");
    HIP_CHECK_RET(hipSetDevice(ctx->device));
    std::string err;
    int rc = build_topology(*mol, *bas, sb.topo, err);
    if (rc != MQC_HIP_OK) return fail(rc, err);
    if (with_eri && !incore_supported(sb.topo.nao)) return fail(MQC_HIP_ERR_UNSUPPORTED, "fragment too large for the in-core ERI path");
    rc = upload_topology(ctx, sb.topo, sb.td);
    if (rc != MQC_HIP_OK) return rc;
    Slot sl0{0, ctx->stream, &ctx->pool_main, &ctx->pool_eri, &ctx->pool_misc, &ctx->pool_gridw, &ctx->pool_df,
             ctx->ev0, ctx->ev1, ctx->ev2, ctx->ev3, ctx->evq0, ctx->evq1, nullptr, ctx->evs[0][0], ctx->evs[0][1]};
    rc = carve_batch(ctx, sl0, sb.topo, sb.td, 1, with_eri, sb.bv);
    if (rc != MQC_HIP_OK) return rc;
    sb.bv.nocc = std::max(1, sb.topo.nelec / 2); sb.bv.exx = 1.0; sb.bv.Vxc = nullptr; sb.bv.xc = XcSpec(); sb.bv.xc.ncomp = 0;
    sb.bv.naux = 0; sb.bv.unit = ctx->d_unit;
    HIP_CHECK_RET(hipMemcpy(sb.bv.xyz, mol->xyz, sizeof(double) * 3 * mol->n_atoms, hipMemcpyHostToDevice));
    HIP_CHECK_RET(hipMemset(sb.bv.istate, 0, sizeof(int) * 4));
    return MQC_HIP_OK;
}

int mqc_hip_int1e(mqc_hip_context* ctx, const mqc_hip_molecule_t* mol, const mqc_hip_basis_t* bas, double* S, double* T, double* V)
{
    StageBatch sb;
    int rc = stage_setup(ctx, mol, bas, false, sb);
    if (rc != MQC_HIP_OK) return rc;
    launch_int1e(sb.bv, sb.topo, ctx->stream);
    HIP_CHECK_RET(hipStreamSynchronize(ctx->stream));
    HIP_CHECK_RET(hipGetLastError());
    const size_t nn = (size_t)sb.topo.nao * sb.topo.nao;
    if (S) HIP_CHECK_RET(hipMemcpy(S, sb.bv.S, sizeof(double) * nn, hipMemcpyDeviceToHost));
    if (T) HIP_CHECK_RET(hipMemcpy(T, sb.bv.W, sizeof(double) * nn, hipMemcpyDeviceToHost));
    if (V) HIP_CHECK_RET(hipMemcpy(V, sb.bv.W + nn, sizeof(double) * nn, hipMemcpyDeviceToHost));
    return MQC_HIP_OK;
}

int mqc_hip_eri_packed(mqc_hip_context* ctx, const mqc_hip_molecule_t* mol, const mqc_hip_basis_t* bas, double schwarz_tol, double* M)
{
    StageBatch sb;
    int rc = stage_setup(ctx, mol, bas, true, sb);
    if (rc != MQC_HIP_OK) return rc;
    // stage-level check of the tensor: poison it first, so that an element no class list covers shows up as NaN
    HIP_CHECK_RET(hipMemsetAsync(sb.bv.eri, 0xFF, sizeof(double) * (size_t)sb.topo.npair * sb.topo.npair, ctx->stream));
    launch_eri(sb.bv, sb.topo, schwarz_tol, ctx->stream);
    HIP_CHECK_RET(hipStreamSynchronize(ctx->stream));
    HIP_CHECK_RET(hipGetLastError());
    const size_t np = (size_t)sb.topo.npair;
    HIP_CHECK_RET(hipMemcpy(M, sb.bv.eri, sizeof(double) * np * np, hipMemcpyDeviceToHost));
    return MQC_HIP_OK;
}

int mqc_hip_jk_incore(mqc_hip_context* ctx, const mqc_hip_molecule_t* mol, const mqc_hip_basis_t* bas, const double* D, double* J, double* K)
{
    StageBatch sb;
    int rc = stage_setup(ctx, mol, bas, true, sb);
    if (rc != MQC_HIP_OK) return rc;
    const size_t nn = (size_t)sb.topo.nao * sb.topo.nao;
    launch_eri(sb.bv, sb.topo, 0.0, ctx->stream);
    HIP_CHECK_RET(hipMemcpyAsync(sb.bv.D, D, sizeof(double) * nn, hipMemcpyHostToDevice, ctx->stream));
    launch_jk_incore(sb.bv, false, ctx->stream);
    HIP_CHECK_RET(hipStreamSynchronize(ctx->stream));
    HIP_CHECK_RET(hipGetLastError());
    HIP_CHECK_RET(hipMemcpy(J, sb.bv.J, sizeof(double) * nn, hipMemcpyDeviceToHost));
    HIP_CHECK_RET(hipMemcpy(K, sb.bv.K, sizeof(double) * nn, hipMemcpyDeviceToHost));
    return MQC_HIP_OK;
}

int mqc_hip_coulomb_batch(mqc_hip_context* ctx, int64_t nfrag, const mqc_hip_molecule_t* mols, const mqc_hip_basis_t* bas,
                          int32_t n_source_atoms, const double* D, double* J)
{
    if (!ctx || nfrag < 0 || (nfrag > 0 && (!mols || !bas || !D || !J))) return fail(MQC_HIP_ERR_VALIDATION, "null argument");
    if (nfrag == 0) return MQC_HIP_OK;
    HIP_CHECK_RET(hipSetDevice(ctx->device));
    for (int64_t i = 0; i < nfrag; ++i) {
        if (!mols[i].atomic_numbers || !mols[i].xyz || mols[i].n_atoms != mols[0].n_atoms ||
            std::memcmp(mols[i].atomic_numbers, mols[0].atomic_numbers, sizeof(int32_t) * mols[0].n_atoms) != 0 || mols[i].ghost != mols[0].ghost)
            return fail(MQC_HIP_ERR_VALIDATION, "coulomb batch: every fragment must have the elements of the first (one topology per call)");
    }
    Topology topo;
    std::string err;
    int rc = build_topology(mols[0], *bas, topo, err);
    if (rc != MQC_HIP_OK) return fail(rc, err);
    if (n_source_atoms < 0 || n_source_atoms >= topo.natoms) return fail(MQC_HIP_ERR_VALIDATION, "coulomb batch: n_source_atoms must be within 0 .. n_atoms - 1");
    const bool cross = n_source_atoms > 0;
    if (cross) {
        // keep the quartets that join a pair on the leading atoms with a pair on the source atoms (either side may be
        // the bra: the class fixes the order); twin entries are dropped with the rest, the plain lists cover everything
        const int first_src = topo.natoms - n_source_atoms;
        auto on_src = [&](int sh) { return topo.shells[sh].atom >= first_src; };
        for (auto& cl : topo.classes) {
            std::vector<int> q, sets;
            for (size_t e = 0; 4 * e + 3 < cl.quartets.size(); ++e) {
                const int* s4 = &cl.quartets[4 * e];
                const bool bra_src = on_src(s4[0]) && on_src(s4[1]), bra_lead = !on_src(s4[0]) && !on_src(s4[1]);
                const bool ket_src = on_src(s4[2]) && on_src(s4[3]), ket_lead = !on_src(s4[2]) && !on_src(s4[3]);
                if (!((bra_src && ket_lead) || (bra_lead && ket_src))) continue;
                q.insert(q.end(), s4, s4 + 4);
                if (e < cl.set_quartets.size()) sets.push_back(cl.set_quartets[e]);
            }
            cl.quartets.swap(q); cl.set_quartets.swap(sets);
            cl.twin_entries.clear(); cl.rest.clear(); cl.set_twin.clear(); cl.set_rest.clear();
        }
        topo.key += "|cross" + std::to_string(n_source_atoms);
    }
    TopologyDev td;
    rc = upload_topology(ctx, topo, td);
    if (rc != MQC_HIP_OK) return rc;
    const int n = topo.nao;
    const size_t nn = (size_t)n * n, np = (size_t)topo.npair;
    // full mode: in-core (the packed tensor of every fragment of a chunk resident at once, chunks sized to the free HBM).
    // cross mode: the direct digest over the filtered lists -- no tensor at all, the few (leading | source) quartets
    // are contracted with the density as they are formed (Schwarz screening at 1e-12)
    static const bool cross_incore = [] { const char* e = std::getenv("MQC_HIP_COULOMB_CROSS_INCORE"); return e && e[0] == '1'; }();
    const bool direct = cross && !cross_incore;
    if (!direct && !incore_supported(topo.nao)) return fail(MQC_HIP_ERR_UNSUPPORTED, "fragment too large for the in-core ERI path");
    if (topo.nao > 256) return fail(MQC_HIP_ERR_UNSUPPORTED, "fragment too large (n_ao <= 256)");
    const size_t per_frag = sizeof(double) * (per_fragment_main_doubles(n, topo.natoms) + (direct ? 4 * nn : np * np));
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    free_b += ctx->pool_main.capacity() + ctx->pool_eri.capacity();
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nfrag, 16384), (int64_t)((double)free_b * 0.7 / (double)per_frag)));
    Slot sl0{0, ctx->stream, &ctx->pool_main, &ctx->pool_eri, &ctx->pool_misc, &ctx->pool_gridw, &ctx->pool_df,
             ctx->ev0, ctx->ev1, ctx->ev2, ctx->ev3, ctx->evq0, ctx->evq1, nullptr, ctx->evs[0][0], ctx->evs[0][1]};
    std::vector<double> hx;
    for (int64_t start = 0; start < nfrag; start += chunk) {
        const int nf = (int)std::min<int64_t>(chunk, nfrag - start);
        BatchView bv{};
        rc = carve_batch(ctx, sl0, topo, td, nf, !direct, bv);
        if (rc != MQC_HIP_OK) return rc;
        // only J is read back: the direct digest skips its exchange updates when exx = 0; the in-core kernels
        // form K alongside J in the same pass over the tensor either way
        bv.nocc = std::max(1, topo.nelec / 2); bv.exx = direct ? 0.0 : 1.0; bv.Vxc = nullptr; bv.xc = XcSpec(); bv.xc.ncomp = 0;
        bv.naux = 0; bv.unit = ctx->d_unit;
        hx.resize((size_t)nf * topo.natoms * 3);
        for (int f = 0; f < nf; ++f) std::memcpy(&hx[(size_t)f * topo.natoms * 3], mols[start + f].xyz, sizeof(double) * topo.natoms * 3);
        HIP_CHECK_RET(hipMemcpyAsync(bv.xyz, hx.data(), sizeof(double) * hx.size(), hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK_RET(hipMemsetAsync(bv.istate, 0, sizeof(int) * (size_t)nf * 4, ctx->stream));
        HIP_CHECK_RET(hipMemsetAsync(bv.eri_count, 0, sizeof(unsigned long long), ctx->stream));
        HIP_CHECK_RET(hipMemcpyAsync(bv.D, D + (size_t)start * nn, sizeof(double) * nn * nf, hipMemcpyHostToDevice, ctx->stream));
        if (direct) {
            launch_direct_setup(bv, topo, ctx->stream);
            launch_jk_direct(bv, topo, 1.0e-12, false, ctx->stream);
        } else {
            // a restricted list leaves most of the tensor untouched: those blocks must read as zero
            if (cross) HIP_CHECK_RET(hipMemsetAsync(bv.eri, 0, sizeof(double) * (size_t)nf * bv.eri_stride, ctx->stream));
            launch_eri(bv, topo, 0.0, ctx->stream, hx.data());
            launch_jk_incore(bv, false, ctx->stream);
        }
        HIP_CHECK_RET(hipStreamSynchronize(ctx->stream));
        HIP_CHECK_RET(hipGetLastError());
        HIP_CHECK_RET(hipMemcpy(J + (size_t)start * nn, bv.J, sizeof(double) * nn * nf, hipMemcpyDeviceToHost));
    }
    return MQC_HIP_OK;
}

int mqc_hip_syev(mqc_hip_context* ctx, int32_t n, const double* A, double* w, double* V)
{
    if (!ctx || !A || !w || !V || n <= 0) return fail(MQC_HIP_ERR_VALIDATION, "bad arguments");
    if (n > 256) return fail(MQC_HIP_ERR_UNSUPPORTED, "matrix too large for the Jacobi solver (n <= 256)");
    HIP_CHECK_RET(hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    double* d = (double*)ctx->pool_main.ensure(sizeof(double) * (2 * nn + n) + 1024);
    if (!d) return fail(MQC_HIP_ERR_DEVICE, "out of device memory");
    HIP_CHECK_RET(hipMemcpy(d, A, sizeof(double) * nn, hipMemcpyHostToDevice));
    launch_syev(n, d, d + 2 * nn, d + nn, ctx->stream);
    HIP_CHECK_RET(hipStreamSynchronize(ctx->stream));
    HIP_CHECK_RET(hipGetLastError());
    HIP_CHECK_RET(hipMemcpy(w, d + 2 * nn, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_CHECK_RET(hipMemcpy(V, d + nn, sizeof(double) * nn, hipMemcpyDeviceToHost));
    return MQC_HIP_OK;
}

int mqc_hip_diis_coefficients(mqc_hip_context* ctx, int32_t n_stored, const double* overlap, double* coefficients, int32_t* ok)
{
    if (!ctx || !overlap || !coefficients || !ok) return fail(MQC_HIP_ERR_VALIDATION, "bad arguments");
    if (n_stored < 1 || n_stored > DIIS_MAX) return fail(MQC_HIP_ERR_VALIDATION, "n_stored must be within 1..8");
    HIP_CHECK_RET(hipSetDevice(ctx->device));
    double* d = (double*)ctx->pool_misc.ensure(4096);
    if (!d) return fail(MQC_HIP_ERR_DEVICE, "out of device memory");
    HIP_CHECK_RET(hipMemcpy(d + 32, overlap, sizeof(double) * n_stored * n_stored, hipMemcpyHostToDevice));
    launch_diis_coeff(n_stored, d + 32, d + 128, (int*)d, ctx->stream);
    HIP_CHECK_RET(hipStreamSynchronize(ctx->stream));
    HIP_CHECK_RET(hipMemcpy(coefficients, d + 128, sizeof(double) * n_stored, hipMemcpyDeviceToHost));
    int okv = 0;
    HIP_CHECK_RET(hipMemcpy(&okv, d, sizeof(int), hipMemcpyDeviceToHost));
    *ok = okv;
    return MQC_HIP_OK;
}

}  // extern "C"
