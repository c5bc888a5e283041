// kern_eri.hip -- dispatcher of the four-centre ERI formation (kernels: eri_kernels.hpp,
// instantiated per class group in kern_eri_inst.hip).
#include "eri_kernels.hpp"
#include <array>
#include <cstring>
#include <map>
#include <unordered_map>
#include <functional>
#include <string>

namespace mqc {

bool launch_digest_general(const BatchView& bv, int la, int lb, int lc, int ld, const int* d_list, int nq, const double* Q,
                           const double* Dmax, double thresh, double* Jt, double* Kt, int only_active, hipStream_t s);   // kern_eri_general.hip

// Which launcher forms a class.  The per-class register kernels cover s, p, d; two of them keep whole blocks in scratch
// memory -- the one-shot / digest kernels of (dd|dp) and (dd|dd) hold a 648- or 1296-number block per LANE (5-23 KB of
// private segment), and the runtime reserves scratch per hardware queue for a full device of such waves (12 GB for the
// (dd|dd) digest): with the pools holding most of HBM that reservation fails and the process aborts
// (HSA_STATUS_ERROR_OUT_OF_RESOURCES, recorded in round 1 under GPU_MAX_HW_QUEUES=8 and in round 2 inside the test
// suite).  Those classes, the (dd| Schwarz bounds (11.5 KB) and everything with an f shell go through the
// wave-cooperative LDS kernel, which has no private segment at all.
static bool class_is_general(int la, int lb, int lc, int ld, int gen_from)
{
    if (la > CLASS_LMAX) return true;
    return eri_uses_passes(la, lb, lc, ld) && la + lb + lc + ld >= gen_from;
}

#define ERI_DECL(a, b, c, d) \
    extern template void launch_eri_class<a, b, c, d>(const BatchView&, const int*, int, const int*, int, const double*, double, hipStream_t);
#define SCHWARZ_DECL(a, b) \
    extern template void launch_schwarz_class<a, b>(const BatchView&, const int*, int, int*, double*, hipStream_t);
ERI_DECL(0, 0, 0, 0) ERI_DECL(1, 0, 0, 0) ERI_DECL(1, 0, 1, 0) ERI_DECL(1, 1, 0, 0) ERI_DECL(1, 1, 1, 0) ERI_DECL(1, 1, 1, 1)
ERI_DECL(2, 0, 0, 0) ERI_DECL(2, 0, 1, 0) ERI_DECL(2, 0, 1, 1) ERI_DECL(2, 0, 2, 0)
ERI_DECL(2, 1, 0, 0) ERI_DECL(2, 1, 1, 0) ERI_DECL(2, 1, 1, 1) ERI_DECL(2, 1, 2, 0) ERI_DECL(2, 1, 2, 1)
ERI_DECL(2, 2, 0, 0) ERI_DECL(2, 2, 1, 0) ERI_DECL(2, 2, 1, 1) ERI_DECL(2, 2, 2, 0) ERI_DECL(2, 2, 2, 1) ERI_DECL(2, 2, 2, 2)
#define TWIN_DECL(a, b, c, d) \
    extern template void launch_eri_twin_class<a, b, c, d>(const BatchView&, const int*, int, const int*, int, const double*, double, hipStream_t);
TWIN_DECL(0, 0, 0, 0) TWIN_DECL(1, 0, 0, 0) TWIN_DECL(1, 0, 1, 0) TWIN_DECL(1, 1, 0, 0)
TWIN_DECL(1, 1, 1, 0) TWIN_DECL(2, 0, 0, 0) TWIN_DECL(2, 0, 1, 0) TWIN_DECL(2, 1, 0, 0)
#define TWINW_DECL(a, b, c, d) \
    extern template void launch_eri_twin_wave_class<a, b, c, d>(const BatchView&, const int*, int, const double*, double, hipStream_t);
TWINW_DECL(0, 0, 0, 0) TWINW_DECL(1, 0, 0, 0)
SCHWARZ_DECL(0, 0) SCHWARZ_DECL(1, 0) SCHWARZ_DECL(1, 1) SCHWARZ_DECL(2, 0) SCHWARZ_DECL(2, 1) SCHWARZ_DECL(2, 2)

#define DIG_DECL(a, b, c, d) \
    extern template void launch_eri_digest_class<a, b, c, d>(const BatchView&, const int*, int, const double*, const double*, double, double*, double*, int, hipStream_t);
DIG_DECL(0, 0, 0, 0) DIG_DECL(1, 0, 0, 0) DIG_DECL(1, 0, 1, 0) DIG_DECL(1, 1, 0, 0) DIG_DECL(1, 1, 1, 0) DIG_DECL(1, 1, 1, 1)
DIG_DECL(2, 0, 0, 0) DIG_DECL(2, 0, 1, 0) DIG_DECL(2, 0, 1, 1) DIG_DECL(2, 0, 2, 0)
DIG_DECL(2, 1, 0, 0) DIG_DECL(2, 1, 1, 0) DIG_DECL(2, 1, 1, 1) DIG_DECL(2, 1, 2, 0) DIG_DECL(2, 1, 2, 1)
DIG_DECL(2, 2, 0, 0) DIG_DECL(2, 2, 1, 0) DIG_DECL(2, 2, 1, 1) DIG_DECL(2, 2, 2, 0) DIG_DECL(2, 2, 2, 1) DIG_DECL(2, 2, 2, 2)

// MQC_HIP_NO_TWIN_BLOCKS=1 forces the segmented treatment everywhere (A/B measurements, tests)
static bool twin_blocks_disabled()
{
    static const bool off = [] { const char* e = std::getenv("MQC_HIP_NO_TWIN_BLOCKS"); return e && e[0] == '1'; }();
    return off;
}

// MQC_HIP_NO_BLOCK_SHARING=1: every fragment forms all of its own integral blocks
static bool block_sharing_disabled()
{
    static const bool off = [] { const char* e = std::getenv("MQC_HIP_NO_BLOCK_SHARING"); return e && e[0] == '1'; }();
    return off;
}

// ---------------------------------------------------------------------------------------
// Block sharing.  Fragments of one batch often contain the same atoms at the same coordinates (every
// dimer of an MBE job repeats its two monomers bit for bit).  An integral block depends only on the atoms
// its four shells sit on, so entries whose atom set is repeated are formed for ONE representative
// fragment per distinct geometry of that set (task-list launch) and copied to the other fragments by
// this kernel.  It walks tensor rows: element (r, c) belongs to the atom set (atoms of AO pair r) U
// (atoms of AO pair c); pp_row[apair(r)][apair(c)] is that set's row in the representative table or -1.
// Threads run along c, so the reads and writes are contiguous; rows that can never be shared exit early.
__global__ void __launch_bounds__(256) eri_broadcast_kernel(BatchView bv, const int* __restrict__ pair_apair /*[npair]*/,
                                                            const int* __restrict__ pp_row /*[nap][nap]*/, int nap,
                                                            const unsigned char* __restrict__ apair_any /*[nap]*/,
                                                            const int* __restrict__ rep_table)
{
    const int f = blockIdx.y, r = blockIdx.x;
    const int np = bv.npair;
    const int ar = pair_apair[r];
    if (!apair_any[ar]) return;
    const int* __restrict__ prow = pp_row + (size_t)ar * nap;
    // square: the whole row; triangular blocks: the stored part of the row, columns 0..r
    const size_t roff = bv.eri_tri ? make_pair_store(bv, 0).row_start((size_t)r) : (size_t)r * np;
    const int ncol = bv.eri_tri ? r + 1 : np;
    double* __restrict__ dst = bv.eri + (size_t)f * bv.eri_stride + roff;
    for (int c = threadIdx.x; c < ncol; c += blockDim.x) {
        const int row = prow[pair_apair[c]];
        if (row < 0) continue;
        const int rep = rep_table[(size_t)row * bv.nfrag + f];
        if (rep != f) dst[c] = bv.eri[(size_t)rep * bv.eri_stride + roff + c];
    }
}

namespace {
struct CoordKey {
    uint64_t w[3];
    bool operator==(const CoordKey& o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2]; }
};
struct CoordHash {
    size_t operator()(const CoordKey& k) const { return (size_t)(k.w[0] * 0x9E3779B97F4A7C15ull ^ (k.w[1] + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full ^ k.w[2] * 0x165667B19E3779F9ull); }
};
struct SharePlan {
    bool on = false;
    std::vector<int> shared_row;             // per atom set: row in rep_table, or -1 when the set is not shared
    std::vector<std::vector<int>> uniq;      // per atom set: representative fragments
    std::vector<int> rep_table;              // [rows][nfrag]
    std::vector<int> pp_row;                 // [nap][nap]: shared row of (atom pair of the bra) U (atom pair of the ket)
    std::vector<int> apair_any;              // [nap] packed as ints: 1 when some ket makes a shared set with this bra pair
    int nap = 0;
};
}  // namespace

static inline int apair_index(int a, int b) { return a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }

// which atom sets repeat across the batch (exact, bitwise coordinate equality)
static void plan_sharing(const Topology& topo, const double* host_xyz, int nfrag, SharePlan& plan)
{
    const int na = topo.natoms, nsets = (int)topo.atom_sets.size();
    if (!host_xyz || nfrag < 16 || nfrag > 65535 || na > 12 || nsets == 0) return;
    // per atom slot: small integer id of the coordinate triple (first fragment that shows it)
    std::vector<uint16_t> aid((size_t)nfrag * na);
    {
        std::vector<std::pair<CoordKey, int>> v(nfrag);
        for (int a = 0; a < na; ++a) {
            for (int f = 0; f < nfrag; ++f) {
                std::memcpy(v[f].first.w, host_xyz + ((size_t)f * na + a) * 3, sizeof(CoordKey));
                v[f].second = f;
            }
            std::sort(v.begin(), v.end(), [](const auto& x, const auto& y) {
                if (x.first.w[0] != y.first.w[0]) return x.first.w[0] < y.first.w[0];
                if (x.first.w[1] != y.first.w[1]) return x.first.w[1] < y.first.w[1];
                if (x.first.w[2] != y.first.w[2]) return x.first.w[2] < y.first.w[2];
                return x.second < y.second;
            });
            int head = 0;
            for (int k = 0; k < nfrag; ++k) {
                if (k > 0 && !(v[k].first == v[k - 1].first)) head = k;
                aid[(size_t)v[k].second * na + a] = (uint16_t)v[head].second;     // run head = smallest fragment index
            }
        }
    }
    plan.shared_row.assign(nsets, -1);
    plan.uniq.assign(nsets, {});
    int rows = 0;
    std::vector<std::pair<uint64_t, int>> keys(nfrag);
    std::vector<int> rep(nfrag);
    for (int sidx = 0; sidx < nsets; ++sidx) {
        const auto& S = topo.atom_sets[sidx];
        for (int f = 0; f < nfrag; ++f) {
            uint64_t k = 0;
            for (size_t i = 0; i < S.size(); ++i) k = (k << 16) | aid[(size_t)f * na + S[i]];
            keys[f] = {k, f};
        }
        std::sort(keys.begin(), keys.end());
        int nu = 0, head = 0;
        for (int k = 0; k < nfrag; ++k) {
            if (k == 0 || keys[k].first != keys[k - 1].first) { head = k; ++nu; }
            rep[keys[k].second] = keys[head].second;
        }
        // too few repeats do not pay: the task-list launches are latency-bound (one wave per entry and distinct
        // geometry) and become the critical path of a small batch -- measured on one rank's share of the (H2O)64
        // job: 8-way split (252 dimers, 4 repeats) 41.7 ms with sharing, 35.9 ms without; 4-way split (8 repeats)
        // 56.7 against 58.4 ms
        if ((size_t)nu * 6 > (size_t)nfrag) continue;
        plan.shared_row[sidx] = rows++;
        auto& uq = plan.uniq[sidx];
        for (int f = 0; f < nfrag; ++f) if (rep[f] == f) uq.push_back(f);
        plan.rep_table.insert(plan.rep_table.end(), rep.begin(), rep.end());
    }
    plan.on = rows > 0;
    if (!plan.on) return;
    // (bra atom pair, ket atom pair) -> shared row
    plan.nap = na * (na + 1) / 2;
    plan.pp_row.assign((size_t)plan.nap * plan.nap, -1);
    plan.apair_any.assign(plan.nap, 0);
    std::map<std::vector<int>, int> set_id;
    for (int sidx = 0; sidx < nsets; ++sidx) set_id[topo.atom_sets[sidx]] = sidx;
    for (int a = 0; a < na; ++a) for (int b = 0; b <= a; ++b)
        for (int c = 0; c < na; ++c) for (int d = 0; d <= c; ++d) {
            std::vector<int> u{a, b, c, d};
            std::sort(u.begin(), u.end());
            u.erase(std::unique(u.begin(), u.end()), u.end());
            auto it = set_id.find(u);
            if (it == set_id.end()) continue;
            const int row = plan.shared_row[it->second];
            if (row < 0) continue;
            plan.pp_row[(size_t)apair_index(a, b) * plan.nap + apair_index(c, d)] = row;
            plan.apair_any[apair_index(a, b)] = 1;
        }
}

namespace {
struct EriLaunch { int cls; bool twin; size_t dense_off; int dense_n; size_t sh_off; int sh_n; size_t task_off; int ntasks; };
// Everything one launch_eri call uploads, kept per (topology, geometry set): an MBE driver that re-evaluates
// the same fragments (geometry optimisation steps aside, every SCF restart does) skips planning and upload.
struct EriListCache {
    uint64_t key = 0;
    bool valid = false;
    // the full key, compared on a hash hit (a 64-bit collision must not hand another batch's share plan out)
    std::string topo_key;
    std::vector<double> xyz;
    int nfrag = 0, flags = 0;
    DevicePool pool;
    std::vector<int> host;
    std::vector<EriLaunch> launches;
    bool shared = false;
    size_t pair_off = 0, pp_off = 0, any_off = 0, rep_off = 0;
    int nap = 0;
};
constexpr int ERI_CACHE_WAYS = 4;
constexpr int ERI_SIDE_MAX = 7;        // task-list launches of different classes are independent: round-robin.
// Side streams in use: 3 with the HIP runtime's default of four hardware queues per process, 7 when the process runs
// with GPU_MAX_HW_QUEUES >= 8 (more streams than hardware queues only serialise); MQC_HIP_ERI_STREAMS overrides.
static int eri_side_streams()
{
    static const int n = [] {
        if (const char* e = std::getenv("MQC_HIP_ERI_STREAMS")) { const int v = std::atoi(e); return v < 1 ? 1 : (v > ERI_SIDE_MAX ? ERI_SIDE_MAX : v); }
        const char* q = std::getenv("GPU_MAX_HW_QUEUES");
        return (q && std::atoi(q) >= 8) ? ERI_SIDE_MAX : 3;
    }();
    return n;
}
#define ERI_SIDE_STREAMS eri_side_streams()
// Three, not more: ROCm maps streams onto 4 hardware queues in creation order, and a lane's main stream plus its
// three side streams (created back to back in mqc_hip_context_get) then sit on four different queues; a fourth
// side stream shares the main stream's queue and its kernels wait behind the orthogonaliser.
struct EriSlotState {
    EriListCache cache[ERI_CACHE_WAYS];
    int next = 0;
    // Schwarz bounds + zero fill started ahead of the one-electron stage (launch_eri_bounds), joined by launch_eri
    bool bounds_pending = false;
    double* Q = nullptr;
    double bounds_tol = 0.0;
    std::vector<int> bucket[KERNEL_LMAX + 1][KERNEL_LMAX + 1];   // shell pairs by class, kept alive for the async upload
    std::vector<int> bucket1[KERNEL_LMAX + 1][KERNEL_LMAX + 1];  // ... pairs whose two shells sit on ONE atom (see launch_eri_bounds)
    std::vector<int> same_atom;                                  // all of them, for the copy to the other fragments
    int* d_same = nullptr; int n_same = 0;
    // one-centre bounds depend on the basis only: kept per topology across calls (q1_vals[ns][ns], valid for q1_key)
    std::string q1_key;
    DevicePool q1pool;
    double* q1_vals = nullptr;
    bool q1_cached = false, q1_save = false;
    DevicePool qpool, pairs;
    hipStream_t side[ERI_SIDE_MAX] = {};
    hipEvent_t fork = nullptr, join[ERI_SIDE_MAX] = {};
};
}  // namespace

static uint64_t hash_words(const void* p, size_t bytes, uint64_t h)
{
    const uint64_t* w = (const uint64_t*)p;
    for (size_t i = 0; i < bytes / 8; ++i) { h ^= w[i]; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29; }
    return h;
}

static hipStream_t g_preset_side[2][ERI_SIDE_MAX] = {};

// side streams created by the context right after the lane's main stream (hardware-queue placement, see above)
void eri_set_side_streams(int slot, const hipStream_t* streams, int count)
{
    for (int k = 0; k < ERI_SIDE_STREAMS && k < count; ++k) g_preset_side[slot & 1][k] = streams[k];
}

static EriSlotState g_eri_state_slot[2];

// mqc_hip_finalize: the events and side streams belong to the device of the context that is going away
void eri_reset_state()
{
    for (int sl = 0; sl < 2; ++sl) {
        EriSlotState& st = g_eri_state_slot[sl];
        if (st.fork) (void)hipEventDestroy(st.fork);
        st.fork = nullptr;
        for (int k = 0; k < ERI_SIDE_MAX; ++k) {
            if (st.join[k]) (void)hipEventDestroy(st.join[k]);
            st.join[k] = nullptr;
            if (st.side[k] && st.side[k] != g_preset_side[sl][k]) (void)hipStreamDestroy(st.side[k]);   // preset ones: the context's
            st.side[k] = nullptr;
            g_preset_side[sl][k] = nullptr;
        }
        for (auto& c : st.cache) { c.valid = false; c.key = 0; c.host.clear(); c.launches.clear(); }
        st.bounds_pending = false; st.Q = nullptr; st.next = 0; st.q1_key.clear(); st.q1_vals = nullptr;
    }
}

static EriSlotState& eri_slot_state(int slot)
{
    EriSlotState& st = g_eri_state_slot[slot & 1];
    if (!st.fork) {
        (void)hipEventCreateWithFlags(&st.fork, hipEventDisableTiming);
        for (int k = 0; k < ERI_SIDE_STREAMS; ++k) {
            if (g_preset_side[slot & 1][k]) st.side[k] = g_preset_side[slot & 1][k];
            else (void)hipStreamCreateWithFlags(&st.side[k], hipStreamNonBlocking);
            (void)hipEventCreateWithFlags(&st.join[k], hipEventDisableTiming);
        }
    }
    return st;
}

// Screened build, part 1: Schwarz bounds of every shell pair and the zero fill of the tensor, put on the side
// streams as soon as the geometry is on the device (call it right after the upload, before the one-electron
// stage): the six class kernels are latency-bound -- (dd|dd) alone is 6 k threads for 4.7 ms -- and run next to
// each other and next to int1e / the orthogonaliser.  launch_eri joins them.
void launch_eri_bounds(const BatchView& bv, const Topology& topo, double schwarz_tol, hipStream_t s)
{
    EriSlotState& st = eri_slot_state(bv.slot);
    st.bounds_pending = false;
    if (!(schwarz_tol > 0.0)) return;
    const size_t np = (size_t)bv.npair;
    st.Q = (double*)st.qpool.ensure(sizeof(double) * (size_t)bv.nfrag * topo.shells.size() * topo.shells.size());
    st.bounds_tol = schwarz_tol;
    int* d_pairs = (int*)st.pairs.ensure((2 * topo.pairs.size() + 32) * sizeof(int));
    for (auto& row : st.bucket) for (auto& b : row) b.clear();
    for (auto& row : st.bucket1) for (auto& b : row) b.clear();
    st.same_atom.clear();
    // One-centre pairs.  (ab|ab) of two shells on the SAME atom involves no other position: the bound is the same
    // number in every fragment of the batch -- and these are the pairs without any primitive screening (all 8^4
    // primitive quartets of an oxygen (1s 1s|1s 1s) survive), i.e. the threads the whole launch waited for (the
    // (ss| launch of 2016 dimers: 6.8 ms, ahead of every class kernel).  They are formed for fragment 0 only and
    // copied (schwarz_copy_kernel, first thing in launch_eri).  MQC_HIP_SCHWARZ_ONE_CENTRE=0 turns that off.
    static const bool one_centre = [] { const char* e = std::getenv("MQC_HIP_SCHWARZ_ONE_CENTRE"); return !(e && e[0] == '0'); }();
    const bool split = one_centre && bv.nfrag >= 4;
    for (size_t k = 0; k + 1 < topo.pairs.size(); k += 2) {
        int A = topo.pairs[k], B = topo.pairs[k + 1];
        if (topo.shells[A].l < topo.shells[B].l) std::swap(A, B);
        const bool same = split && topo.shells[A].atom == topo.shells[B].atom && topo.shells[A].l <= CLASS_LMAX;
        auto& bk = (same ? st.bucket1 : st.bucket)[topo.shells[A].l][topo.shells[B].l];
        bk.push_back(A); bk.push_back(B);
        if (same) { st.same_atom.push_back(A); st.same_atom.push_back(B); }
    }
    (void)hipEventRecord(st.fork, s);
    for (int k = 0; k < ERI_SIDE_STREAMS; ++k) (void)hipStreamWaitEvent(st.side[k], st.fork, 0);
    size_t off = 0;
    int rr = 0;
    double* Q = st.Q;
    // the zero fill first, on the last stream, which it keeps to itself when there are streams to spare: with the
    // one-centre bounds out of the way it is the longest item here (12 GB for 2016 dimers, 2.8 ms), the class kernels wait
    // for it, and a bound launch queued behind it made them wait for both
    if (bv.eri) (void)hipMemsetAsync(bv.eri, 0, sizeof(double) * bv.eri_stride * bv.nfrag, st.side[ERI_SIDE_STREAMS - 1]);
    const int nb_streams = (bv.eri && ERI_SIDE_STREAMS >= 4) ? ERI_SIDE_STREAMS - 1 : ERI_SIDE_STREAMS;
    // most expensive class first
#define SCHWARZ_CASE(a, b)                                                                                                      \
    launch_schwarz_class<a, b>(bv, st.bucket[a][b].data(), (int)st.bucket[a][b].size() / 2, d_pairs + off, Q, st.side[rr++ % nb_streams]); \
    off += st.bucket[a][b].size();
    SCHWARZ_CASE(0, 0) SCHWARZ_CASE(2, 1) SCHWARZ_CASE(1, 1)
    SCHWARZ_CASE(1, 0) SCHWARZ_CASE(2, 0)
#undef SCHWARZ_CASE
    st.n_same = 0;
    st.q1_cached = false; st.q1_save = false;
    if (!st.same_atom.empty()) {
        const size_t nsq = topo.shells.size() * topo.shells.size();
        st.q1_cached = st.q1_vals != nullptr && st.q1_key == topo.key;
        if (!st.q1_cached) {
            st.q1_vals = (double*)st.q1pool.ensure(sizeof(double) * nsq);
            st.q1_key.clear();
            st.q1_save = st.q1_vals != nullptr;
        }
    }
    if (!st.same_atom.empty() && st.q1_cached) {
        // the numbers are there from an earlier call with this topology: only the pair list goes up (launch_eri copies)
        st.d_same = d_pairs + off;
        st.n_same = (int)st.same_atom.size() / 2;
        (void)hipMemcpyAsync(st.d_same, st.same_atom.data(), st.same_atom.size() * sizeof(int), hipMemcpyHostToDevice, st.side[rr++ % nb_streams]);
        off += st.same_atom.size();
    } else if (!st.same_atom.empty()) {
        BatchView b1 = bv;
        b1.nfrag = 1;
#define SCHWARZ_ONE(a, b)                                                                                                       \
    launch_schwarz_class<a, b>(b1, st.bucket1[a][b].data(), (int)st.bucket1[a][b].size() / 2, d_pairs + off, Q, st.side[rr++ % nb_streams]); \
    off += st.bucket1[a][b].size();
        SCHWARZ_ONE(0, 0) SCHWARZ_ONE(1, 1) SCHWARZ_ONE(1, 0) SCHWARZ_ONE(2, 1) SCHWARZ_ONE(2, 0)
#undef SCHWARZ_ONE
        if (!st.bucket1[2][2].empty()) {
            hipStream_t ss = st.side[rr++ % nb_streams];
            (void)hipMemcpyAsync(d_pairs + off, st.bucket1[2][2].data(), st.bucket1[2][2].size() * sizeof(int), hipMemcpyHostToDevice, ss);
            launch_schwarz_general(b1, 2, 2, d_pairs + off, (int)st.bucket1[2][2].size() / 2, Q, ss);
            off += st.bucket1[2][2].size();
        }
        st.d_same = d_pairs + off;
        st.n_same = (int)st.same_atom.size() / 2;
        (void)hipMemcpyAsync(st.d_same, st.same_atom.data(), st.same_atom.size() * sizeof(int), hipMemcpyHostToDevice, st.side[rr++ % nb_streams]);
        off += st.same_atom.size();
    }
    // (dd| bounds (the pass kernel carries 11.5 KB of scratch per lane) and the f classes: LDS kernel
    for (int a = CLASS_LMAX; a <= KERNEL_LMAX; ++a)
        for (int b = (a == CLASS_LMAX ? CLASS_LMAX : 0); b <= a; ++b) {
            auto& bk = st.bucket[a][b];
            if (bk.empty()) continue;
            hipStream_t ss = st.side[rr++ % nb_streams];
            (void)hipMemcpyAsync(d_pairs + off, bk.data(), bk.size() * sizeof(int), hipMemcpyHostToDevice, ss);
            launch_schwarz_general(bv, a, b, d_pairs + off, (int)bk.size() / 2, Q, ss);
            off += bk.size();
        }
    for (int k = 0; k < ERI_SIDE_STREAMS; ++k) (void)hipEventRecord(st.join[k], st.side[k]);
    st.bounds_pending = true;
}

// Host side of the integral stage: block-sharing plan, class / task lists, ONE upload.  Depends on the geometry only,
// so the engine calls it right after the geometry upload (eri_plan_lists) and the planning runs on the host while the
// Schwarz-bound and one-electron kernels run on the device; launch_eri then finds the entry in the cache.
static EriListCache* eri_lists(const BatchView& bv, const Topology& topo, hipStream_t s, const double* host_xyz)
{
    EriSlotState& st = eri_slot_state(bv.slot);
    // twin-shell cut and block sharing combine with Schwarz screening (a twin entry is kept when any member
    // combination passes; shared entries are tested with their representative's bounds)
    const bool twins = !twin_blocks_disabled();
    const bool may_share = !block_sharing_disabled() && host_xyz != nullptr;

    // ---- list cache lookup: key = topology, fragment count, switches, every coordinate bit
    uint64_t key = (uint64_t)std::hash<std::string>{}(topo.key) ^ 0x243F6A8885A308D3ull;
    key = hash_words(&key, 8, (uint64_t)bv.nfrag * 0x100000001B3ull + (twins ? 1 : 0) + (may_share ? 2 : 0) + topo.key.size() * 8 + topo.nao * 131071ull);
    if (may_share) key = hash_words(host_xyz, sizeof(double) * (size_t)bv.nfrag * topo.natoms * 3, key);
    EriListCache* cc = nullptr;
    const int key_flags = (twins ? 1 : 0) + (may_share ? 2 : 0);
    const size_t xyz_count = may_share ? (size_t)bv.nfrag * topo.natoms * 3 : 0;
    for (auto& c : st.cache)
        if (c.valid && c.key == key && c.nfrag == bv.nfrag && c.flags == key_flags && c.topo_key == topo.key && c.xyz.size() == xyz_count &&
            (xyz_count == 0 || std::memcmp(c.xyz.data(), host_xyz, sizeof(double) * xyz_count) == 0)) cc = &c;
    if (!cc) {
        cc = &st.cache[st.next];
        st.next = (st.next + 1) % ERI_CACHE_WAYS;
        cc->valid = false;
        SharePlan plan;
        if (may_share) plan_sharing(topo, host_xyz, bv.nfrag, plan);
        // stage every list of this call in ONE host buffer: per launch (dense entries | shared entries + tasks)
        std::vector<int>& hb = cc->host;
        hb.clear();
        cc->launches.clear();
        auto stage = [&](int ci, bool twin, const std::vector<int>& ents, const std::vector<int>& sets) {
            EriLaunch L{ci, twin, 0, 0, 0, 0, 0, 0};
            const size_t nent = ents.size() / 4;
            if (nent == 0) return;
            if (!plan.on) {
                L.dense_off = hb.size(); L.dense_n = (int)nent;
                hb.insert(hb.end(), ents.begin(), ents.end());
                cc->launches.push_back(L);
                return;
            }
            // two passes over the entries -- count, then write straight into the staging buffer (its capacity survives
            // from call to call): the (entry, representative) pairs of an MBE dimer list are ~700 k, and growing three
            // vectors by push_back was 2 ms of host time between the bounds and the first class launch
            size_t ndense = 0, nsh = 0, ntask = 0;
            for (size_t e = 0; e < nent; ++e) {
                if (plan.shared_row[sets[e]] < 0) ++ndense;
                else { ++nsh; ntask += plan.uniq[sets[e]].size(); }
            }
            L.dense_off = hb.size(); L.dense_n = (int)ndense;
            L.sh_off = L.dense_off + 4 * ndense; L.sh_n = (int)nsh;
            L.task_off = L.sh_off + 4 * nsh; L.ntasks = (int)ntask;
            hb.resize(L.task_off + 2 * ntask);
            int* pd = hb.data() + L.dense_off;
            int* ps = hb.data() + L.sh_off;
            int* pt = hb.data() + L.task_off;
            int local = 0;
            for (size_t e = 0; e < nent; ++e) {
                const int* q4 = ents.data() + 4 * e;
                if (plan.shared_row[sets[e]] < 0) { pd[0] = q4[0]; pd[1] = q4[1]; pd[2] = q4[2]; pd[3] = q4[3]; pd += 4; continue; }
                ps[0] = q4[0]; ps[1] = q4[1]; ps[2] = q4[2]; ps[3] = q4[3]; ps += 4;
                for (int u : plan.uniq[sets[e]]) { *pt++ = local; *pt++ = u; }
                ++local;
            }
            cc->launches.push_back(L);
        };
        for (size_t ci = 0; ci < topo.classes.size(); ++ci) {
            const auto& cl = topo.classes[ci];
            if (twins && !cl.twin_entries.empty()) {
                stage((int)ci, true, cl.twin_entries, cl.set_twin);
                stage((int)ci, false, cl.rest, cl.set_rest);
            } else {
                stage((int)ci, false, cl.quartets, cl.set_quartets);
            }
        }
        cc->shared = plan.on;
        if (plan.on) {
            // AO pair -> atom pair, (atom pair, atom pair) -> shared row, representative table
            std::vector<int> ao_atom(topo.nao);
            for (auto& sh : topo.shells) for (int m = 0; m < 2 * sh.l + 1; ++m) ao_atom[sh.aoff + m] = sh.atom;
            cc->pair_off = hb.size();
            for (int i = 0; i < topo.nao; ++i) for (int j = 0; j <= i; ++j) hb.push_back(apair_index(ao_atom[i], ao_atom[j]));
            cc->pp_off = hb.size();
            hb.insert(hb.end(), plan.pp_row.begin(), plan.pp_row.end());
            cc->any_off = hb.size();
            hb.resize(hb.size() + (plan.nap + 3) / 4, 0);
            for (int a = 0; a < plan.nap; ++a) ((unsigned char*)&hb[cc->any_off])[a] = (unsigned char)plan.apair_any[a];
            cc->rep_off = hb.size();
            hb.insert(hb.end(), plan.rep_table.begin(), plan.rep_table.end());
            cc->nap = plan.nap;
        }
        int* dnew = (int*)cc->pool.ensure((hb.size() + 16) * sizeof(int));
        (void)hipMemcpyAsync(dnew, hb.data(), hb.size() * sizeof(int), hipMemcpyHostToDevice, s);
        cc->key = key;
        cc->topo_key = topo.key; cc->nfrag = bv.nfrag; cc->flags = key_flags;
        if (xyz_count) cc->xyz.assign(host_xyz, host_xyz + xyz_count); else cc->xyz.clear();
        cc->valid = true;
    }
    return cc;
}

// the Schwarz bounds and threshold the slot's last screened build used (engine.cpp hands them to the J/K kernel)
void eri_schwarz_view(int slot, const double** q, double* thresh)
{
    EriSlotState& st = eri_slot_state(slot);
    *q = st.Q; *thresh = st.bounds_tol;
}

void eri_plan_lists(const BatchView& bv, const Topology& topo, hipStream_t s, const double* host_xyz)
{
    (void)eri_lists(bv, topo, s, host_xyz);
}

// bounds of one-centre pairs, formed for fragment 0 (launch_eri_bounds), to the other fragments
// src: [ns][ns] holding the one-centre values (fragment 0's block of Q, or the per-topology cache); fragments f0 .. nfrag-1
__global__ void schwarz_copy_kernel(double* __restrict__ Q, const double* __restrict__ src, const int* __restrict__ pairs, int npairs,
                                    int f0, int nfrag, int ns)
{
    const int nf = nfrag - f0;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long)npairs * nf) return;
    const int ip = (int)(tid / nf), f = f0 + (int)(tid % nf);
    const int A = pairs[2 * ip], B = pairs[2 * ip + 1];
    const double v = src[A * ns + B];
    double* q = Q + (size_t)f * ns * ns;
    q[A * ns + B] = v;
    q[B * ns + A] = v;
}
// the one-centre values just formed for fragment 0 into the per-topology cache
__global__ void schwarz_save_kernel(const double* __restrict__ Q, double* __restrict__ dst, const int* __restrict__ pairs, int npairs, int ns)
{
    const int ip = blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= npairs) return;
    const int A = pairs[2 * ip], B = pairs[2 * ip + 1];
    dst[A * ns + B] = Q[A * ns + B];
    dst[B * ns + A] = Q[A * ns + B];
}

void launch_eri(const BatchView& bv, const Topology& topo, double schwarz_tol, hipStream_t s, const double* host_xyz)
{
    EriSlotState& st = eri_slot_state(bv.slot);
    const size_t np = (size_t)bv.npair;
    // The class lists cover every element of the pair matrix, so the unscreened build overwrites the whole
    // tensor and needs no zero fill (22 GB for the (H2O)64 dimers); a screened build leaves skipped blocks at zero.
    double* Q = nullptr;
    double thresh = 0.0;
    if (schwarz_tol > 0.0) {
        if (!st.bounds_pending || st.bounds_tol != schwarz_tol) launch_eri_bounds(bv, topo, schwarz_tol, s);
        for (int k = 0; k < ERI_SIDE_STREAMS; ++k) (void)hipStreamWaitEvent(s, st.join[k], 0);
        st.bounds_pending = false;
        Q = st.Q;
        thresh = schwarz_tol;
        if (st.n_same > 0) {
            const int ns = (int)topo.shells.size();
            if (st.q1_save) {
                hipLaunchKernelGGL(schwarz_save_kernel, dim3((unsigned)((st.n_same + 255) / 256)), dim3(256), 0, s, Q, st.q1_vals, st.d_same, st.n_same, ns);
                st.q1_key = topo.key;
                st.q1_save = false;
            }
            const int f0 = st.q1_cached ? 0 : 1;
            const long total = (long)st.n_same * (bv.nfrag - f0);
            if (total > 0)
                hipLaunchKernelGGL(schwarz_copy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, Q,
                                   st.q1_cached ? (const double*)st.q1_vals : (const double*)Q, st.d_same, st.n_same, f0, bv.nfrag, ns);
            st.n_same = 0;
        }
    }
    else if (bv.eri_tri) {
        // the triangular blocks carry zero padding that no class kernel writes: an unscreened build fills too
        (void)hipMemsetAsync(bv.eri, 0, sizeof(double) * bv.eri_stride * bv.nfrag, s);
    }
    EriListCache* cc = eri_lists(bv, topo, s, host_xyz);
    const int* d = (const int*)cc->pool.ensure(0);

    // ---- launches: dense lists on the caller's stream; the task-list launches of shared entries (few, long
    // threads: one wave per entry and distinct geometry) on a side stream so that they fill gaps instead of
    // serialising; both join before the copy kernel
    // The class launches are spread over the caller's stream and the side streams: small batches give every launch
    // only a handful of long-running waves, and even a full batch has a tail per launch; with the launches placed
    // longest-first their critical paths overlap instead of adding up.
    // (measured with the longest-first assignment below: 2016 dimers 127.3 -> 124 ms, 1008: 75.3 -> 70.8, 504: 55.0 ->
    // 49.5, so every batch is spread; MQC_HIP_ERI_SPREAD_MAX=n keeps batches above n fragments on one stream)
    static const int spread_max = [] { const char* e = std::getenv("MQC_HIP_ERI_SPREAD_MAX"); return e ? std::atoi(e) : (1 << 30); }();
    const bool spread = bv.nfrag <= spread_max;
    const bool forked = cc->shared || spread;
    if (forked) {
        (void)hipEventRecord(st.fork, s);
        for (int k = 0; k < ERI_SIDE_STREAMS; ++k) (void)hipStreamWaitEvent(st.side[k], st.fork, 0);
    }
    int rr = 0;
    // spread mode: longest-processing-time-first assignment of the class launches to the 1 + ERI_SIDE_STREAMS
    // streams.  A small-batch launch lasts as long as its heaviest thread: primitive quartets of the first (deepest)
    // entry x work per primitive quartet x passes.
    std::vector<int> lane_of(cc->launches.size(), 0), issue_order;
    if (spread) {
        std::vector<std::pair<double, int>> cost(cc->launches.size());
        for (size_t k = 0; k < cc->launches.size(); ++k) {
            const EriLaunch& L = cc->launches[k];
            const auto& cl = topo.classes[L.cls];
            double c = 0.0;
            if (L.dense_n > 0) {
                const int* e = cc->host.data() + L.dense_off;
                double prims = 1.0;
                for (int q = 0; q < 4; ++q) prims *= topo.shells[e[q] & 0xffff].nprim;
                const int nc = ncart(cl.la) * ncart(cl.lb) * ncart(cl.lc) * ncart(cl.ld);
                const int passes = eri_uses_passes(cl.la, cl.lb, cl.lc, cl.ld)
                                       ? (nsph(cl.lc) * nsph(cl.ld) + eri_pass_chunk(cl.la, cl.lb, cl.lc, cl.ld) - 1) / eri_pass_chunk(cl.la, cl.lb, cl.lc, cl.ld)
                                       : 1;
                c = prims * (nc + 8.0 * nherm(cl.la + cl.lb + cl.lc + cl.ld)) * passes * (L.twin ? 1.5 : 1.0);
            }
            cost[k] = {c, (int)k};
        }
        std::sort(cost.begin(), cost.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
        for (auto& ck : cost) issue_order.push_back(ck.second);
        double load[ERI_SIDE_MAX + 1] = {};
        // side stream 2 carries the one-electron chain of this chunk (int1e classes, orthogonaliser, guess: engine.cpp) ahead of
        // anything queued here -- 1.9 ms for a single fragment, as much as the heaviest class launch: in a batch of one,
        // three class launches placed on it waited for the guess and the whole stage with them (1.3 of 3.2 ms)
        if (ERI_SIDE_STREAMS >= 3 && !cost.empty()) load[3] = 1.5 * cost.front().first;
        for (auto& ck : cost) {
            int best = 0;
            for (int q = 1; q <= ERI_SIDE_STREAMS; ++q) if (load[q] < load[best]) best = q;
            load[best] += ck.first;
            lane_of[ck.second] = best;
        }
    }
    int li = 0;
    auto dense_stream = [&]() { if (!spread) return s; const int k = lane_of[li]; return k == 0 ? s : st.side[k - 1]; };
    // MQC_HIP_ERI_GENERAL=k: classes whose total angular momentum is >= k AND that would take the pass kernels go
    // through the wave-cooperative LDS kernel instead (no scratch, one wave per quartet and fragment); default: off
    // default 7: (dd|dp) and (dd|dd); MQC_HIP_ERI_GENERAL=k moves the border (4 = every pass class, 99 = f shells only)
    static const int gen_from = [] { const char* e = std::getenv("MQC_HIP_ERI_GENERAL"); return e ? std::atoi(e) : 7; }();
    auto to_general = [&](const Topology::ClassList& c) { return class_is_general(c.la, c.lb, c.lc, c.ld, gen_from); };
#define ERI_CASE(a, b, c, d_)                                                                                         \
    if (!general && cl.la == a && cl.lb == b && cl.lc == c && cl.ld == d_) {                                          \
        launch_eri_class<a, b, c, d_>(bv, d + L.dense_off, L.dense_n, nullptr, 0, Q, thresh, dense_stream());         \
        launch_eri_class<a, b, c, d_>(bv, d + L.sh_off, L.sh_n, d + L.task_off, L.ntasks, Q, thresh, st.side[rr++ % ERI_SIDE_STREAMS]); \
    }
#define TWIN_CASE(a, b, c, d_)                                                                                        \
    if (cl.la == a && cl.lb == b && cl.lc == c && cl.ld == d_) {                                                      \
        if (twin_wave && (a) <= 1 && (b) == 0 && (c) == 0 && (d_) == 0)                                               \
            launch_twin_wave(a, d + L.dense_off, L.dense_n, dense_stream());                                          \
        else                                                                                                          \
        launch_eri_twin_class<a, b, c, d_>(bv, d + L.dense_off, L.dense_n, nullptr, 0, Q, thresh, dense_stream());    \
        launch_eri_twin_class<a, b, c, d_>(bv, d + L.sh_off, L.sh_n, d + L.task_off, L.ntasks, Q, thresh, st.side[rr++ % ERI_SIDE_STREAMS]); \
    }
    // small batches: the twin (ss|ss) and (ps|ss) entries one wave per (entry, fragment) (eri_twin_wave_kernel)
    static const int twin_wave_max = [] { const char* e = std::getenv("MQC_HIP_TWIN_WAVE_MAX"); return e ? std::atoi(e) : ERI_TWIN_WAVE_MAX_FRAGMENTS; }();
    const bool twin_wave = bv.nfrag <= twin_wave_max;
    auto launch_twin_wave = [&](int la, const int* list, int nq, hipStream_t st_) {
        if (la == 0) launch_eri_twin_wave_class<0, 0, 0, 0>(bv, list, nq, Q, thresh, st_);
        else launch_eri_twin_wave_class<1, 0, 0, 0>(bv, list, nq, Q, thresh, st_);
    };
    // in spread mode the launches are issued heaviest first (stream order = issue order)
    std::vector<int> order(cc->launches.size());
    for (size_t k = 0; k < order.size(); ++k) order[k] = (int)k;
    if (spread) order = issue_order;
    for (size_t oi = 0; oi < order.size(); ++oi) {
        li = order[oi];
        const EriLaunch& L = cc->launches[li];
        const auto& cl = topo.classes[L.cls];
        const bool general = to_general(cl);
        if (L.twin) {
            TWIN_CASE(0, 0, 0, 0) TWIN_CASE(1, 0, 0, 0) TWIN_CASE(1, 0, 1, 0) TWIN_CASE(1, 1, 0, 0)
            TWIN_CASE(1, 1, 1, 0) TWIN_CASE(2, 0, 0, 0) TWIN_CASE(2, 0, 1, 0) TWIN_CASE(2, 1, 0, 0)
            continue;
        }
        ERI_CASE(0, 0, 0, 0)
        ERI_CASE(1, 0, 0, 0) ERI_CASE(1, 0, 1, 0)
        ERI_CASE(1, 1, 0, 0) ERI_CASE(1, 1, 1, 0) ERI_CASE(1, 1, 1, 1)
        ERI_CASE(2, 0, 0, 0) ERI_CASE(2, 0, 1, 0) ERI_CASE(2, 0, 1, 1) ERI_CASE(2, 0, 2, 0)
        ERI_CASE(2, 1, 0, 0) ERI_CASE(2, 1, 1, 0) ERI_CASE(2, 1, 1, 1) ERI_CASE(2, 1, 2, 0) ERI_CASE(2, 1, 2, 1)
        ERI_CASE(2, 2, 0, 0) ERI_CASE(2, 2, 1, 0) ERI_CASE(2, 2, 1, 1) ERI_CASE(2, 2, 2, 0) ERI_CASE(2, 2, 2, 1)
        ERI_CASE(2, 2, 2, 2)
        if (general) {
            // a class with an f shell (or a d-heavy class routed here): the wave-cooperative LDS kernel
            launch_eri_general(bv, cl.la, cl.lb, cl.lc, cl.ld, d + L.dense_off, L.dense_n, nullptr, 0, Q, thresh, dense_stream());
            launch_eri_general(bv, cl.la, cl.lb, cl.lc, cl.ld, d + L.sh_off, L.sh_n, d + L.task_off, L.ntasks, Q, thresh, st.side[rr++ % ERI_SIDE_STREAMS]);
        }
    }
#undef ERI_CASE
#undef TWIN_CASE
    if (forked) {
        for (int k = 0; k < ERI_SIDE_STREAMS; ++k) {
            (void)hipEventRecord(st.join[k], st.side[k]);
            (void)hipStreamWaitEvent(s, st.join[k], 0);
        }
    }
    if (cc->shared) {
        hipLaunchKernelGGL(eri_broadcast_kernel, dim3((unsigned)np, (unsigned)bv.nfrag), dim3(256), 0, s, bv,
                           d + cc->pair_off, d + cc->pp_off, cc->nap, (const unsigned char*)(d + cc->any_off), d + cc->rep_off);
    }
}


// ---------------------------------------------------------------------------------------
// Direct path: Schwarz bounds once per batch, then every iteration: block maxima of D, zero J~/K~,
// digest kernels per class, symmetrise into J and K.
__global__ void density_block_max_kernel(BatchView bv, double* __restrict__ Dmax)
{
    const int f = blockIdx.y, ns = bv.topo.nshell, n = bv.n;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ns * ns) return;
    const int A = idx / ns, B = idx - A * ns;
    const int oa = bv.topo.sh_aoff[A], ob = bv.topo.sh_aoff[B];
    const int na = 2 * bv.topo.sh_l[A] + 1, nb = 2 * bv.topo.sh_l[B] + 1;
    const double* D = bv.D + (size_t)f * n * n;
    double m = 0.0;
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) m = fmax(m, fabs(D[(oa + i) * n + ob + j]));
    Dmax[(size_t)f * ns * ns + idx] = m;
}

__global__ void symmetrise_jk_kernel(BatchView bv, const double* __restrict__ Jt, const double* __restrict__ Kt)
{
    const int f = blockIdx.y, n = bv.n;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    const int i = idx / n, j = idx - i * n;
    const size_t o = (size_t)f * n * n;
    bv.J[o + idx] = 0.5 * (Jt[o + idx] + Jt[o + j * n + i]);
    bv.K[o + idx] = 0.5 * (Kt[o + idx] + Kt[o + j * n + i]);
}

static DevicePool g_direct_lists_slot[2], g_direct_q_slot[2], g_direct_work_slot[2];

// Uploads the class lists and computes the Schwarz bounds; call once per batch before the SCF loop.
// Incremental direct Fock build (mqc_libcint_rhf.f90:1110-1174): G_ref += G(D - D_ref) with the density screen acting on
// the difference, a full build every INCREMENTAL_RESET iterations.  Per pipeline slot: reference density, J and K, and
// the difference / correction buffers.
constexpr int INCREMENTAL_RESET = 16;       // mqc_libcint_rhf.f90:150
struct DirectIncrState {
    bool active = false;
    int since_reset = 0;
};
static DirectIncrState g_direct_incr[2];
static DevicePool g_direct_incr_pool[2];

void launch_direct_setup(const BatchView& bv, const Topology& topo, hipStream_t s)
{
    g_direct_incr[bv.slot & 1] = DirectIncrState();      // a new batch starts from a full build
    DevicePool& g_direct_lists = g_direct_lists_slot[bv.slot & 1];
    DevicePool& g_direct_q = g_direct_q_slot[bv.slot & 1];
    size_t total_ints = topo.pairs.size();
    for (auto& cl : topo.classes) total_ints += cl.quartets.size();
    int* d_list = (int*)g_direct_lists.ensure((total_ints + 16) * sizeof(int));
    const size_t nss = (size_t)bv.nfrag * topo.shells.size() * topo.shells.size();
    double* Q = (double*)g_direct_q.ensure(sizeof(double) * 2 * nss);
    size_t off = 0;
    for (auto& cl : topo.classes) {
        (void)hipMemcpyAsync(d_list + off, cl.quartets.data(), cl.quartets.size() * sizeof(int), hipMemcpyHostToDevice, s);
        off += cl.quartets.size();
    }
    std::vector<int> bucket[KERNEL_LMAX + 1][KERNEL_LMAX + 1];
    for (size_t k = 0; k + 1 < topo.pairs.size(); k += 2) {
        int A = topo.pairs[k], B = topo.pairs[k + 1];
        if (topo.shells[A].l < topo.shells[B].l) std::swap(A, B);
        auto& bk = bucket[topo.shells[A].l][topo.shells[B].l];
        bk.push_back(A); bk.push_back(B);
    }
#define SCHWARZ_CASE(a, b)                                                                                   \
    launch_schwarz_class<a, b>(bv, bucket[a][b].data(), (int)bucket[a][b].size() / 2, d_list + off, Q, s);   \
    off += bucket[a][b].size();
    SCHWARZ_CASE(0, 0) SCHWARZ_CASE(1, 0) SCHWARZ_CASE(1, 1)
    SCHWARZ_CASE(2, 0) SCHWARZ_CASE(2, 1)
#undef SCHWARZ_CASE
    for (int a = CLASS_LMAX; a <= KERNEL_LMAX; ++a)
        for (int b = (a == CLASS_LMAX ? CLASS_LMAX : 0); b <= a; ++b) {
            auto& bk = bucket[a][b];
            if (bk.empty()) continue;
            (void)hipMemcpyAsync(d_list + off, bk.data(), bk.size() * sizeof(int), hipMemcpyHostToDevice, s);
            launch_schwarz_general(bv, a, b, d_list + off, (int)bk.size() / 2, Q, s);
            off += bk.size();
        }
    (void)hipStreamSynchronize(s);
}

#define DIG_CASE(a, b, c, d)                                                                          \
    if (!general && cl.la == a && cl.lb == b && cl.lc == c && cl.ld == d)                             \
        launch_eri_digest_class<a, b, c, d>(bv, d_list + off, (int)cl.quartets.size() / 4, Q, Dmax, thresh, Jt, Kt, oa, s);

void launch_jk_direct(const BatchView& bv, const Topology& topo, double thresh, bool only_active, hipStream_t s)
{
    const int n = bv.n, ns = (int)topo.shells.size(), oa = only_active ? 1 : 0;
    const size_t nn = (size_t)bv.nfrag * n * n, nss = (size_t)bv.nfrag * ns * ns;
    DevicePool& g_direct_lists = g_direct_lists_slot[bv.slot & 1];
    DevicePool& g_direct_q = g_direct_q_slot[bv.slot & 1];
    DevicePool& g_direct_work = g_direct_work_slot[bv.slot & 1];
    const int* d_list = (const int*)g_direct_lists.ensure(0);
    double* Q = (double*)g_direct_q.ensure(0);
    double* Dmax = Q + nss;
    double* Jt = (double*)g_direct_work.ensure(sizeof(double) * 2 * nn);
    double* Kt = Jt + nn;
    (void)hipMemsetAsync(Jt, 0, sizeof(double) * 2 * nn, s);
    hipLaunchKernelGGL(density_block_max_kernel, dim3((ns * ns + 255) / 256, bv.nfrag), dim3(256), 0, s, bv, Dmax);
    size_t off = 0;
    for (auto& cl : topo.classes) {
        // classes whose digest kernel would hold the whole block in scratch (more than ERI_UNROLL_LIMIT numbers per lane),
        // and the f classes: LDS kernel
        const bool general = cl.la > CLASS_LMAX || ncart(cl.la) * ncart(cl.lb) * ncart(cl.lc) * ncart(cl.ld) > ERI_UNROLL_LIMIT;
        if (general) launch_digest_general(bv, cl.la, cl.lb, cl.lc, cl.ld, d_list + off, (int)cl.quartets.size() / 4, Q, Dmax, thresh, Jt, Kt, oa, s);
        DIG_CASE(0, 0, 0, 0)
        DIG_CASE(1, 0, 0, 0) DIG_CASE(1, 0, 1, 0)
        DIG_CASE(1, 1, 0, 0) DIG_CASE(1, 1, 1, 0) DIG_CASE(1, 1, 1, 1)
        DIG_CASE(2, 0, 0, 0) DIG_CASE(2, 0, 1, 0) DIG_CASE(2, 0, 1, 1) DIG_CASE(2, 0, 2, 0)
        DIG_CASE(2, 1, 0, 0) DIG_CASE(2, 1, 1, 0) DIG_CASE(2, 1, 1, 1) DIG_CASE(2, 1, 2, 0) DIG_CASE(2, 1, 2, 1)
        DIG_CASE(2, 2, 0, 0) DIG_CASE(2, 2, 1, 0) DIG_CASE(2, 2, 1, 1) DIG_CASE(2, 2, 2, 0) DIG_CASE(2, 2, 2, 1)
        DIG_CASE(2, 2, 2, 2)
        off += cl.quartets.size();
    }
    hipLaunchKernelGGL(symmetrise_jk_kernel, dim3((n * n + 255) / 256, bv.nfrag), dim3(256), 0, s, bv, Jt, Kt);
}


__global__ void direct_delta_density_kernel(BatchView bv, const double* __restrict__ Dref, double* __restrict__ Dd, int only_active)
{
    const int f = blockIdx.y;
    if (only_active && bv.istate[4 * f] == ST_DONE) return;
    const size_t nn = (size_t)bv.n * bv.n, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nn) Dd[f * nn + i] = bv.D[f * nn + i] - Dref[f * nn + i];
}

// full: references <- (D, J, K) just built;  else: J_ref += J_delta, K_ref += K_delta, D_ref <- D, and J, K <- references
__global__ void direct_incr_update_kernel(BatchView bv, double* __restrict__ Dref, double* __restrict__ Jref, double* __restrict__ Kref,
                                          const double* __restrict__ Jd, const double* __restrict__ Kd, int full, int only_active)
{
    const int f = blockIdx.y;
    if (only_active && bv.istate[4 * f] == ST_DONE) return;
    const size_t nn = (size_t)bv.n * bv.n, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn) return;
    const size_t o = f * nn + i;
    if (full) { Jref[o] = bv.J[o]; Kref[o] = bv.K[o]; }
    else {
        const double j = Jref[o] + Jd[o], k = Kref[o] + Kd[o];
        Jref[o] = j; Kref[o] = k;
        bv.J[o] = j; bv.K[o] = k;
    }
    Dref[o] = bv.D[o];
}

void launch_jk_direct_incremental(const BatchView& bv, const Topology& topo, double thresh, bool only_active, hipStream_t s)
{
    // OFF by default: measured on (H2O)8 / cc-pVDZ (n = 192, scripts/direct_incremental_probe.py) the corrections dropped
    // by the density screen cost three more SCF cycles at 1e-10 / 1e-8 (17 against 14) for the same energy and no gain
    // in wall time (1.23 s against 1.16 s); MQC_HIP_DIRECT_INCREMENTAL=1 turns it on
    static const bool on = [] { const char* e = std::getenv("MQC_HIP_DIRECT_INCREMENTAL"); return e && e[0] == '1'; }();
    if (!on || bv.uhf) { launch_jk_direct(bv, topo, thresh, only_active, s); return; }
    DirectIncrState& st = g_direct_incr[bv.slot & 1];
    const int n = bv.n, oa = only_active ? 1 : 0;
    const size_t nn = (size_t)n * n, tot = (size_t)bv.nfrag * nn;
    double* base = (double*)g_direct_incr_pool[bv.slot & 1].ensure(sizeof(double) * 6 * tot + 256);
    if (!base) { launch_jk_direct(bv, topo, thresh, only_active, s); return; }
    double *Dref = base, *Jref = base + tot, *Kref = base + 2 * tot, *Dd = base + 3 * tot, *Jd = base + 4 * tot, *Kd = base + 5 * tot;
    const dim3 grid((unsigned)((nn + 255) / 256), bv.nfrag);
    const bool full = !st.active || st.since_reset >= INCREMENTAL_RESET;
    if (full) {
        launch_jk_direct(bv, topo, thresh, only_active, s);
        hipLaunchKernelGGL(direct_incr_update_kernel, grid, dim3(256), 0, s, bv, Dref, Jref, Kref, Jd, Kd, 1, oa);
        st.active = true; st.since_reset = 0;
    } else {
        hipLaunchKernelGGL(direct_delta_density_kernel, grid, dim3(256), 0, s, bv, Dref, Dd, oa);
        BatchView vd = bv;
        vd.D = Dd; vd.J = Jd; vd.K = Kd;
        launch_jk_direct(vd, topo, thresh, only_active, s);
        hipLaunchKernelGGL(direct_incr_update_kernel, grid, dim3(256), 0, s, bv, Dref, Jref, Kref, Jd, Kd, 0, oa);
        st.since_reset += 1;
    }
}

}  // namespace mqc
