// host_setup.cpp -- host-side preparation below the C ABI: contraction normalisation,
// topology (shell, pair and quartet-class lists), the Boys table and cart->sph tables.
// Nothing here computes integrals or any part of an SCF; that is all on the device.
#include "engine.hpp"
#include "md_integrals.hpp"
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <sstream>

namespace mqc {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) { g_last_error = msg; return code; }
const std::string& last_error_string() { return g_last_error; }

// ---- device pool ---------------------------------------------------------------------
namespace {
struct PoolRegistry {
    std::mutex m;
    std::vector<DevicePool*> pools;
};
PoolRegistry& pool_registry() { static PoolRegistry* r = new PoolRegistry(); return *r; }   // never destroyed: pools may outlive statics
}  // namespace

DevicePool::DevicePool()
{
    auto& r = pool_registry();
    std::lock_guard<std::mutex> lock(r.m);
    r.pools.push_back(this);
}

DevicePool::~DevicePool()
{
    auto& r = pool_registry();
    std::lock_guard<std::mutex> lock(r.m);
    r.pools.erase(std::remove(r.pools.begin(), r.pools.end(), this), r.pools.end());
    // device memory is released by mqc_hip_finalize (release_all_pools); at process exit the runtime reclaims it
}

void release_all_pools()
{
    auto& r = pool_registry();
    std::lock_guard<std::mutex> lock(r.m);
    for (DevicePool* p : r.pools) p->release();
}

void* DevicePool::ensure(size_t bytes)
{
    if (bytes <= cap_ && ptr_) return ptr_;
    if (ptr_) { (void)hipFree(ptr_); ptr_ = nullptr; cap_ = 0; }
    size_t want = bytes + bytes / 8 + 256;     // a little headroom so near-equal batches do not realloc
    if (hipMalloc(&ptr_, want) != hipSuccess) {
        ptr_ = nullptr; cap_ = 0;
        if (hipMalloc(&ptr_, bytes) != hipSuccess) { ptr_ = nullptr; return nullptr; }
        want = bytes;
    }
    cap_ = want;
    return ptr_;
}

void DevicePool::release()
{
    if (ptr_) (void)hipFree(ptr_);
    ptr_ = nullptr; cap_ = 0;
}

// ---- Boys table: F_0..F_{BOYS_COLS-1} on the grid T0 = r * BOYS_STEP --------------------
static void boys_series(int nmax, double T, double* F)
{
    if (T < 1e-13) { for (int n = 0; n <= nmax; ++n) F[n] = 1.0 / (2 * n + 1) - T / (2 * n + 3); return; }
    const double et = std::exp(-T);
    long double term = 1.0L / (2 * nmax + 1), sum = term;
    for (int k = 1; k < 600; ++k) {
        term *= 2.0L * T / (2 * nmax + 2 * k + 1);
        sum += term;
        if (term < 1e-20L * sum) break;
    }
    F[nmax] = (double)(et * sum);
    for (int n = nmax; n > 0; --n) F[n - 1] = (2.0 * T * F[n] + et) / (2 * n - 1);
}

void build_boys_table(std::vector<double>& table)
{
    table.assign((size_t)BOYS_ROWS * BOYS_COLS, 0.0);
    std::vector<double> F(BOYS_COLS + 1);
    for (int r = 0; r < BOYS_ROWS; ++r) {
        boys_series(BOYS_COLS - 1, r * BOYS_STEP, F.data());
        for (int c = 0; c < BOYS_COLS; ++c) table[(size_t)r * BOYS_COLS + c] = F[c];
    }
}

// ---- real solid harmonics r^l Y_lm in libcint's order (m = -l..l; p = x,y,z) ------------
static double binom(int n, int k)
{
    if (k < 0 || k > n) return 0.0;
    double r = 1.0;
    for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return r;
}
static double fact(int n) { double r = 1.0; for (int i = 2; i <= n; ++i) r *= i; return r; }

void build_c2s_tables(std::vector<double>& packed, int* offsets)
{
    packed.clear();
    for (int l = 0; l <= LMAX_AO; ++l) {
        offsets[l] = (int)packed.size();
        const int nc = ncart(l), ns = nsph(l);
        std::vector<double> tab((size_t)ns * nc, 0.0);
        std::vector<int> cx, cy, cz;
        for (int lx = l; lx >= 0; --lx)
            for (int ly = l - lx; ly >= 0; --ly) { cx.push_back(lx); cy.push_back(ly); cz.push_back(l - lx - ly); }
        const double to_y = std::sqrt((2.0 * l + 1.0) / (4.0 * M_PI));
        for (int m = -l; m <= l; ++m) {
            const int am = std::abs(m);
            const double nlm = 1.0 / (std::pow(2.0, am) * fact(l)) *
                               std::sqrt(2.0 * fact(l + am) * fact(l - am) / (m == 0 ? 2.0 : 1.0));
            const int twov0 = (m >= 0) ? 0 : 1;
            for (int t = 0; t <= (l - am) / 2; ++t)
                for (int u = 0; u <= t; ++u)
                    for (int twov = twov0; twov <= am; twov += 2) {
                        const int sp = t + (twov - twov0) / 2;
                        const double c = ((sp & 1) ? -1.0 : 1.0) * std::pow(0.25, t) * binom(l, t) *
                                         binom(l - t, am + t) * binom(t, u) * binom(am, twov);
                        const int ex = 2 * t + am - 2 * u - twov, ey = 2 * u + twov, ez = l - 2 * t - am;
                        if (ex < 0 || ey < 0 || ez < 0) continue;
                        for (int k = 0; k < nc; ++k)
                            if (cx[k] == ex && cy[k] == ey && cz[k] == ez) tab[(size_t)(m + l) * nc + k] += nlm * c * to_y;
                    }
        }
        if (l <= 1) {
            // s and p: the constant is folded into the contraction coefficients, identity here
            std::fill(tab.begin(), tab.end(), 0.0);
            for (int k = 0; k < nc; ++k) tab[(size_t)k * nc + k] = 1.0;
        }
        packed.insert(packed.end(), tab.begin(), tab.end());
    }
}

// ---- contraction normalisation (mqc_libcint_integrals.F90:519-555) ----------------------
static double gto_norm(int l, double a)
{
    const int n = 2 * l + 2;
    const double gint = std::tgamma((n + 1) / 2.0) / (2.0 * std::pow(2.0 * a, (n + 1) / 2.0));
    return 1.0 / std::sqrt(gint);
}

std::string topology_key(const mqc_hip_molecule_t& mol, const mqc_hip_basis_t& bas)
{
    std::ostringstream os;
    os << mol.n_atoms << ':' << mol.nelec << ':' << mol.charge << ':' << mol.multiplicity << ':';
    for (int i = 0; i < mol.n_atoms; ++i)
        os << mol.atomic_numbers[i] << (mol.ghost && mol.ghost[i] ? 'g' : 'r') << bas.nshell_per_atom[i] << ',';
    os << '|' << bas.n_shells << '|';
    size_t off = 0;
    // the basis content: hash of the raw bytes so that two different sets with equal shapes differ
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](const void* p, size_t nbytes) {
        const unsigned char* b = (const unsigned char*)p;
        for (size_t i = 0; i < nbytes; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    };
    for (int s = 0; s < bas.n_shells; ++s) {
        os << bas.shell_l[s] << '.' << bas.shell_nprim[s] << ',';
        mix(bas.exponents + off, sizeof(double) * bas.shell_nprim[s]);
        mix(bas.coefficients + off, sizeof(double) * bas.shell_nprim[s]);
        off += bas.shell_nprim[s];
    }
    os << '#' << h;
    return os.str();
}

int build_topology(const mqc_hip_molecule_t& mol, const mqc_hip_basis_t& bas, Topology& topo, std::string& err, int max_l,
                   bool with_quartets)
{
    if (mol.n_atoms <= 0) { err = "fragment has no atoms"; return MQC_HIP_ERR_VALIDATION; }
    if (bas.n_atoms != mol.n_atoms) { err = "the basis covers a different number of atoms than the geometry has"; return MQC_HIP_ERR_VALIDATION; }
    if (!bas.spherical) {
        err = "Cartesian basis sets are not supported by the HIP backend (spherical only)";
        return MQC_HIP_ERR_UNSUPPORTED;
    }
    topo = Topology();
    topo.key = topology_key(mol, bas);
    topo.natoms = mol.n_atoms;
    topo.nelec = mol.nelec; topo.charge = mol.charge; topo.multiplicity = mol.multiplicity;
    topo.Z.assign(mol.atomic_numbers, mol.atomic_numbers + mol.n_atoms);
    topo.zeff.resize(mol.n_atoms);
    for (int i = 0; i < mol.n_atoms; ++i) topo.zeff[i] = (mol.ghost && mol.ghost[i]) ? 0.0 : (double)mol.atomic_numbers[i];

    constexpr double S_FACTOR = 0.282094791773878143, P_FACTOR = 0.488602511902919921;
    size_t poff_in = 0;
    int sh = 0, aoff = 0;
    for (int at = 0; at < mol.n_atoms; ++at) {
        for (int64_t k = 0; k < bas.nshell_per_atom[at]; ++k, ++sh) {
            if (sh >= bas.n_shells) { err = "nshell_per_atom does not sum to n_shells"; return MQC_HIP_ERR_VALIDATION; }
            const int l = bas.shell_l[sh], np = bas.shell_nprim[sh];
            if (l < 0 || l > LMAX_AO) { err = "angular momentum above g is not supported"; return MQC_HIP_ERR_UNSUPPORTED; }
            if (l > max_l) {
                err = "this build of the HIP backend covers angular momentum up to l = " + std::to_string(max_l) +
                      (max_l == KERNEL_LMAX ? " for orbital shells" : " for auxiliary shells");
                return MQC_HIP_ERR_UNSUPPORTED;
            }
            if (np <= 0) { err = "shell without primitives"; return MQC_HIP_ERR_VALIDATION; }
            const double* a = bas.exponents + poff_in;
            const double* c = bas.coefficients + poff_in;
            std::vector<double> cn(np);
            for (int i = 0; i < np; ++i) {
                if (!(a[i] > 0.0)) { err = "non-positive exponent"; return MQC_HIP_ERR_VALIDATION; }
                cn[i] = c[i] * gto_norm(l, a[i]);
            }
            double norm2 = 0.0;
            for (int i = 0; i < np; ++i)
                for (int j = 0; j < np; ++j)
                    norm2 += cn[i] * cn[j] / (gto_norm(l, a[i]) * gto_norm(l, a[j])) *
                             std::pow(2.0 * std::sqrt(a[i] * a[j]) / (a[i] + a[j]), l + 1.5);
            const double scale = norm2 > 0.0 ? 1.0 / std::sqrt(norm2) : 1.0;
            const double ang = (l == 0) ? S_FACTOR : (l == 1 ? P_FACTOR : 1.0);
            HostShell hs;
            hs.l = l; hs.atom = at; hs.aoff = aoff; hs.poff = (int)topo.exps.size(); hs.nprim = 0;
            for (int i = 0; i < np; ++i) {
                if (c[i] == 0.0) continue;     // a zero row entry of a general contraction contributes nothing
                topo.exps.push_back(a[i]);
                topo.coefs.push_back(cn[i] * scale * ang);
                hs.nprim++;
            }
            if (hs.nprim == 0) { err = "shell with all-zero coefficients"; return MQC_HIP_ERR_VALIDATION; }
            topo.shells.push_back(hs);
            topo.lmax = std::max(topo.lmax, l);
            aoff += 2 * l + 1;
            poff_in += np;
        }
    }
    if (sh != bas.n_shells) { err = "nshell_per_atom does not sum to n_shells"; return MQC_HIP_ERR_VALIDATION; }
    topo.nao = aoff;
    topo.npair = aoff * (aoff + 1) / 2;

    // shell pairs A >= B
    const int ns = (int)topo.shells.size();
    for (int A = 0; A < ns; ++A)
        for (int B = 0; B <= A; ++B) { topo.pairs.push_back(A); topo.pairs.push_back(B); }

    if (!with_quartets) return MQC_HIP_OK;     // auxiliary bases and density-fitted runs need no quartet lists
    // canonical quartets: pair index ab >= cd; inside a pair the higher l first; bra class >= ket class
    struct P { int a, b, la, lb, pc, np; };
    std::vector<P> pl;
    for (int A = 0; A < ns; ++A)
        for (int B = 0; B <= A; ++B) {
            int a = A, b = B;
            if (topo.shells[a].l < topo.shells[b].l) std::swap(a, b);
            const int la = topo.shells[a].l, lb = topo.shells[b].l;
            pl.push_back({a, b, la, lb, la * (la + 1) / 2 + lb, topo.shells[a].nprim * topo.shells[b].nprim});
        }
    std::map<int, Topology::ClassList> cls;
    std::map<int, std::vector<std::pair<int, std::array<int, 4>>>> tmp;   // class -> (cost, quartet)
    for (size_t ij = 0; ij < pl.size(); ++ij)
        for (size_t kl = 0; kl <= ij; ++kl) {
            P bra = pl[ij], ket = pl[kl];
            if (bra.pc < ket.pc) std::swap(bra, ket);
            const int id = ((bra.la * 8 + bra.lb) * 8 + ket.la) * 8 + ket.lb;
            tmp[id].push_back({bra.np * ket.np, {bra.a, bra.b, ket.a, ket.b}});
            topo.n_quartets++;
        }
    // ---- twin s shells: consecutive s shells of one atom contracted over identical primitives
    topo.twin_first.assign(ns, 0);
    std::vector<int> member_of_twin(ns, 0);
    for (int A = 0; A + 1 < ns; ++A) {
        const HostShell &x = topo.shells[A], &y = topo.shells[A + 1];
        if (member_of_twin[A] || x.l != 0 || y.l != 0 || x.atom != y.atom || x.nprim != y.nprim || x.nprim < 2) continue;
        bool same = true;
        for (int k = 0; k < x.nprim; ++k) same = same && (topo.exps[x.poff + k] == topo.exps[y.poff + k]);
        if (!same) continue;
        topo.twin_first[A] = 1;
        member_of_twin[A] = member_of_twin[A + 1] = 1;
    }
    // super-shells (a twin pair counts once), their canonical pairs and quartets -> twin entries per class
    std::map<int, std::vector<std::pair<int, std::array<int, 4>>>> ttmp;
    {
        std::vector<int> sup;     // first member of every super-shell
        for (int A = 0; A < ns; ++A) if (!(member_of_twin[A] && !topo.twin_first[A])) sup.push_back(A);
        struct SP { int a, b, la, lb, pc, np; };
        std::vector<SP> spl;
        for (size_t X = 0; X < sup.size(); ++X)
            for (size_t Y = 0; Y <= X; ++Y) {
                int a = sup[X], b = sup[Y];
                if (topo.shells[a].l < topo.shells[b].l) std::swap(a, b);
                const int la = topo.shells[a].l, lb = topo.shells[b].l;
                spl.push_back({a, b, la, lb, la * (la + 1) / 2 + lb, topo.shells[a].nprim * topo.shells[b].nprim});
            }
        for (size_t ij = 0; ij < spl.size(); ++ij)
            for (size_t kl = 0; kl <= ij; ++kl) {
                SP bra = spl[ij], ket = spl[kl];
                if (bra.pc < ket.pc) std::swap(bra, ket);
                if (!eri_has_twin_block(bra.la, bra.lb, ket.la, ket.lb)) continue;
                const int sh4[4] = {bra.a, bra.b, ket.a, ket.b};
                if (!(topo.twin_first[sh4[0]] || topo.twin_first[sh4[1]] || topo.twin_first[sh4[2]] || topo.twin_first[sh4[3]])) continue;
                std::array<int, 4> e;
                for (int k = 0; k < 4; ++k) e[k] = sh4[k] | (topo.twin_first[sh4[k]] ? (1 << 16) : 0);
                const int id = ((bra.la * 8 + bra.lb) * 8 + ket.la) * 8 + ket.lb;
                ttmp[id].push_back({bra.np * ket.np, e});
            }
    }
    for (auto& kv : tmp) {
        auto& v = kv.second;
        // deepest contractions first: lanes of a wave then share a trip count even in a batch of one
        std::stable_sort(v.begin(), v.end(), [](const auto& x, const auto& y) { return x.first > y.first; });
        Topology::ClassList cl;
        const int id = kv.first;
        cl.ld = id % 8; cl.lc = (id / 8) % 8; cl.lb = (id / 64) % 8; cl.la = id / 512;
        for (auto& q : v) for (int k = 0; k < 4; ++k) cl.quartets.push_back(q.second[k]);
        if (eri_has_twin_block(cl.la, cl.lb, cl.lc, cl.ld) && ttmp.count(id)) {
            auto& tv = ttmp[id];
            std::stable_sort(tv.begin(), tv.end(), [](const auto& x, const auto& y) { return x.first > y.first; });
            for (auto& q : tv) for (int k = 0; k < 4; ++k) cl.twin_entries.push_back(q.second[k]);
            for (auto& q : v) {
                const auto& e = q.second;
                if (member_of_twin[e[0]] || member_of_twin[e[1]] || member_of_twin[e[2]] || member_of_twin[e[3]]) continue;
                for (int k = 0; k < 4; ++k) cl.rest.push_back(e[k]);
            }
        }
        topo.classes.push_back(std::move(cl));
    }
    // atom set of every list entry (block sharing between fragments of a batch, kern_eri.hip)
    {
        std::map<std::vector<int>, int> ids;
        auto set_of = [&](const int* e) {
            std::vector<int> a(4);
            for (int k = 0; k < 4; ++k) a[k] = topo.shells[e[k] & 0xffff].atom;
            std::sort(a.begin(), a.end());
            a.erase(std::unique(a.begin(), a.end()), a.end());
            auto it = ids.find(a);
            if (it != ids.end()) return it->second;
            const int id = (int)topo.atom_sets.size();
            ids.emplace(a, id);
            topo.atom_sets.push_back(a);
            return id;
        };
        for (auto& cl : topo.classes) {
            for (size_t q = 0; q + 3 < cl.quartets.size(); q += 4) cl.set_quartets.push_back(set_of(&cl.quartets[q]));
            for (size_t q = 0; q + 3 < cl.twin_entries.size(); q += 4) cl.set_twin.push_back(set_of(&cl.twin_entries[q]));
            for (size_t q = 0; q + 3 < cl.rest.size(); q += 4) cl.set_rest.push_back(set_of(&cl.rest[q]));
        }
    }
    return MQC_HIP_OK;
}

double nuclear_repulsion(const Topology& topo, const double* xyz)
{
    double e = 0.0;
    for (int i = 0; i < topo.natoms; ++i)
        for (int j = 0; j < i; ++j) {
            if (topo.zeff[i] == 0.0 || topo.zeff[j] == 0.0) continue;
            const double dx = xyz[3 * i] - xyz[3 * j], dy = xyz[3 * i + 1] - xyz[3 * j + 1], dz = xyz[3 * i + 2] - xyz[3 * j + 2];
            e += topo.zeff[i] * topo.zeff[j] / std::sqrt(dx * dx + dy * dy + dz * dz);
        }
    return e;
}

}  // namespace mqc
