// check_device_math.cpp -- TEST-ONLY harness.  Compiles the `__host__ __device__` integral
// templates of metalquicha_amd/csrc/md_integrals.hpp for the HOST so that tests/ can compare
// the arithmetic the gfx950 kernels instantiate against the oracle here, without a GPU.
// It is never linked into libmqc_hip.so and is not a fallback path of the product.
#include "../../metalquicha_amd/csrc/md_integrals.hpp"
#include "../../metalquicha_amd/csrc/engine.hpp"
#include <vector>
#include <set>
#include <array>
#include <algorithm>
#include <cstring>

namespace mqc {
void build_boys_table(std::vector<double>& table);
void build_c2s_tables(std::vector<double>& packed, int* offsets);
}
using namespace mqc;

static std::vector<double> g_boys, g_c2s;
static int g_off[8];

static void ensure_tables()
{
    if (g_boys.empty()) { build_boys_table(g_boys); build_c2s_tables(g_c2s, g_off); }
}

template <int LA, int LB, int LC, int LD>
static void run_class(const ShellRef* sh, double* out_sph)
{
    constexpr int NCA = ncart(LA), NCB = ncart(LB), NCC = ncart(LC), NCD = ncart(LD);
    std::vector<double> cart(NCA * NCB * NCC * NCD), tmp(NCA * NCB * NCC * NCD);
    eri_cart_block<LA, LB, LC, LD>(sh[0], sh[1], sh[2], sh[3], g_boys.data(), cart.data());
    const int L[4] = {LA, LB, LC, LD};
    int dims[4] = {NCA, NCB, NCC, NCD};
    for (int ax = 0; ax < 4; ++ax) {
        const int l = L[ax], nc = ncart(l), ns = nsph(l);
        int pre = 1, post = 1;
        for (int k = 0; k < ax; ++k) pre *= dims[k];
        for (int k = ax + 1; k < 4; ++k) post *= dims[k];
        const double* T = g_c2s.data() + g_off[l];
        for (int a = 0; a < pre; ++a)
            for (int s = 0; s < ns; ++s)
                for (int r = 0; r < post; ++r) {
                    double v = 0.0;
                    for (int c = 0; c < nc; ++c) v += T[s * nc + c] * cart[(a * nc + c) * post + r];
                    tmp[(a * ns + s) * post + r] = v;
                }
        dims[ax] = ns;
        cart.assign(tmp.begin(), tmp.begin() + pre * ns * post);
        cart.resize(NCA * NCB * NCC * NCD);
    }
    std::memcpy(out_sph, cart.data(), sizeof(double) * dims[0] * dims[1] * dims[2] * dims[3]);
}

struct HostSink {
    double* out; int nsb, nsc, nsd;
    void operator()(int i, int j, int k, int l, double v) { out[((i * nsb + j) * nsc + k) * nsd + l] = v; }
};

template <int LA, int LB, int LC, int LD>
static void run_class_passes(const ShellRef* sh, double* out_sph)
{
    constexpr int CH = eri_pass_chunk(LA, LB, LC, LD);
    std::vector<double> acc(ncart(LA) * ncart(LB) * CH);
    HostSink sink{out_sph, nsph(LB), nsph(LC), nsph(LD)};
    eri_passes_from<LA, LB, LC, LD, CH, 0>(sh[0], sh[1], sh[2], sh[3], g_boys.data(), g_c2s.data(), acc.data(), 1, sink);
}

// twin block vs the plain block of every member combination (Cartesian, before c2s): returns max |diff|
template <int LA, int LB, int LC, int LD>
static double run_twin(const ShellRef* sh, const double* const* second, const bool* is_twin)
{
    constexpr int NC = ncart(LA) * ncart(LB) * ncart(LC) * ncart(LD);
    constexpr int MA = twin_mult(true, LA), MB = twin_mult(true, LB), MC = twin_mult(true, LC), MD = twin_mult(true, LD);
    std::vector<double> acc(MA * MB * MC * MD * NC), ref(NC);
    TwinCoefs tw;
    tw.ca[0] = sh[0].coefs; tw.ca[1] = second[0]; tw.fa = is_twin[0] ? 1.0 : 0.0;
    tw.cb[0] = sh[1].coefs; tw.cb[1] = second[1]; tw.fb = is_twin[1] ? 1.0 : 0.0;
    tw.cc[0] = sh[2].coefs; tw.cc[1] = second[2]; tw.fc = is_twin[2] ? 1.0 : 0.0;
    tw.cd[0] = sh[3].coefs; tw.cd[1] = second[3]; tw.fd = is_twin[3] ? 1.0 : 0.0;
    eri_cart_block_twin<LA, LB, LC, LD>(sh[0], sh[1], sh[2], sh[3], tw, g_boys.data(), acc.data());
    double worst = 0.0;
    {
        // the partial blocks of eri_twin_wave_kernel (bra primitive pairs first, first + 3, ...) add up to the whole block
        std::vector<double> part(acc.size(), 0.0), tmp(acc.size());
        for (int first = 0; first < 3; ++first) {
            eri_cart_block_twin<LA, LB, LC, LD>(sh[0], sh[1], sh[2], sh[3], tw, g_boys.data(), tmp.data(), first, 3);
            for (size_t i = 0; i < acc.size(); ++i) part[i] += tmp[i];
        }
        double scale = 1.0;
        for (double v : acc) scale = std::max(scale, std::fabs(v));
        for (size_t i = 0; i < acc.size(); ++i) worst = std::max(worst, std::fabs(part[i] - acc[i]) / scale);
    }
    const int M[4] = {MA, MB, MC, MD};
    for (int ma = 0; ma < MA; ++ma) for (int mb = 0; mb < MB; ++mb) for (int mc = 0; mc < MC; ++mc) for (int md = 0; md < MD; ++md) {
        const int m[4] = {ma, mb, mc, md};
        ShellRef t[4];
        bool absent = false;
        for (int k = 0; k < 4; ++k) {
            t[k] = sh[k];
            if (m[k]) { if (!is_twin[k]) absent = true; else t[k].coefs = second[k]; }
        }
        (void)M;
        const double* got = acc.data() + (((ma * MB + mb) * MC + mc) * MD + md) * NC;
        if (absent) { for (int i = 0; i < NC; ++i) worst = std::max(worst, std::fabs(got[i])); continue; }
        eri_cart_block<LA, LB, LC, LD>(t[0], t[1], t[2], t[3], g_boys.data(), ref.data());
        double scale = 1.0;
        for (int i = 0; i < NC; ++i) scale = std::max(scale, std::fabs(ref[i]));
        for (int i = 0; i < NC; ++i) worst = std::max(worst, std::fabs(got[i] - ref[i]) / scale);
    }
    return worst;
}

extern "C" {

// shells: for each of the 4 shells: nprim, then pointers are passed flat.
int hostcheck_eri_block(const int* l, const int* nprim, const double* exps, const double* coefs /* normalised, s/p factor folded */,
                        const double* xyz /* 4x3 */, double* out_sph)
{
    ensure_tables();
    ShellRef sh[4];
    int off = 0;
    for (int k = 0; k < 4; ++k) {
        sh[k].nprim = nprim[k]; sh[k].exps = exps + off; sh[k].coefs = coefs + off;
        sh[k].x = xyz[3 * k]; sh[k].y = xyz[3 * k + 1]; sh[k].z = xyz[3 * k + 2];
        off += nprim[k];
    }
    const int id = ((l[0] * 8 + l[1]) * 8 + l[2]) * 8 + l[3];
#define CASE(a, b, c, d) case (((a * 8 + b) * 8 + c) * 8 + d): run_class<a, b, c, d>(sh, out_sph); return 0;
    switch (id) {
        CASE(0, 0, 0, 0)
        CASE(1, 0, 0, 0) CASE(1, 0, 1, 0)
        CASE(1, 1, 0, 0) CASE(1, 1, 1, 0) CASE(1, 1, 1, 1)
        CASE(2, 0, 0, 0) CASE(2, 0, 1, 0) CASE(2, 0, 1, 1) CASE(2, 0, 2, 0)
        CASE(2, 1, 0, 0) CASE(2, 1, 1, 0) CASE(2, 1, 1, 1) CASE(2, 1, 2, 0) CASE(2, 1, 2, 1)
        CASE(2, 2, 0, 0) CASE(2, 2, 1, 0) CASE(2, 2, 1, 1) CASE(2, 2, 2, 0) CASE(2, 2, 2, 1) CASE(2, 2, 2, 2)
    }
    return 1;
}

int hostcheck_eri_block_passes(const int* l, const int* nprim, const double* exps, const double* coefs, const double* xyz, double* out_sph)
{
    ensure_tables();
    ShellRef sh[4];
    int off = 0;
    for (int k = 0; k < 4; ++k) {
        sh[k].nprim = nprim[k]; sh[k].exps = exps + off; sh[k].coefs = coefs + off;
        sh[k].x = xyz[3 * k]; sh[k].y = xyz[3 * k + 1]; sh[k].z = xyz[3 * k + 2];
        off += nprim[k];
    }
    const int id = ((l[0] * 8 + l[1]) * 8 + l[2]) * 8 + l[3];
#define PCASE(a, b, c, d) case (((a * 8 + b) * 8 + c) * 8 + d): run_class_passes<a, b, c, d>(sh, out_sph); return 0;
    switch (id) {
        PCASE(0, 0, 0, 0) PCASE(1, 0, 1, 0) PCASE(1, 1, 1, 1)
        PCASE(2, 0, 1, 1) PCASE(2, 0, 2, 0) PCASE(2, 1, 1, 0) PCASE(2, 1, 1, 1) PCASE(2, 1, 2, 0) PCASE(2, 1, 2, 1)
        PCASE(2, 2, 0, 0) PCASE(2, 2, 1, 0) PCASE(2, 2, 1, 1) PCASE(2, 2, 2, 0) PCASE(2, 2, 2, 1) PCASE(2, 2, 2, 2)
    }
    return 1;
}

// coefs2: second coefficient column for each shell (same layout as coefs); twin[k] != 0 marks twin positions
int hostcheck_eri_twin(const int* l, const int* nprim, const double* exps, const double* coefs, const double* coefs2,
                       const int* twin, const double* xyz, double* worst)
{
    ensure_tables();
    ShellRef sh[4];
    const double* second[4];
    bool tw[4];
    int off = 0;
    for (int k = 0; k < 4; ++k) {
        sh[k].nprim = nprim[k]; sh[k].exps = exps + off; sh[k].coefs = coefs + off; second[k] = coefs2 + off;
        sh[k].x = xyz[3 * k]; sh[k].y = xyz[3 * k + 1]; sh[k].z = xyz[3 * k + 2];
        tw[k] = twin[k] != 0 && l[k] == 0;
        off += nprim[k];
    }
    const int id = ((l[0] * 8 + l[1]) * 8 + l[2]) * 8 + l[3];
#define TCASE(a, b, c, d) case (((a * 8 + b) * 8 + c) * 8 + d): *worst = run_twin<a, b, c, d>(sh, second, tw); return 0;
    switch (id) {
        TCASE(0, 0, 0, 0) TCASE(1, 0, 0, 0) TCASE(1, 0, 1, 0) TCASE(1, 1, 0, 0)
        TCASE(1, 1, 1, 0) TCASE(2, 0, 0, 0) TCASE(2, 0, 1, 0) TCASE(2, 1, 0, 0)
    }
    return 1;
}

// Twin cut of the topology: entries expanded over their members + rest == the canonical quartet set, per class.
int hostcheck_twin_cut(const mqc_hip_molecule_t* mol, const mqc_hip_basis_t* bas, int* n_twin_first, int* n_twin_entries, int* n_bad)
{
    Topology topo;
    std::string err;
    if (build_topology(*mol, *bas, topo, err, KERNEL_LMAX, true) != MQC_HIP_OK) return 1;
    auto key = [](int a, int b, int c, int d) {
        if (a < b) std::swap(a, b);
        if (c < d) std::swap(c, d);
        if (std::make_pair(a, b) < std::make_pair(c, d)) { std::swap(a, c); std::swap(b, d); }
        return std::array<int, 4>{a, b, c, d};
    };
    *n_twin_first = 0; *n_twin_entries = 0; *n_bad = 0;
    for (int t : topo.twin_first) *n_twin_first += t;
    for (auto& cl : topo.classes) {
        std::set<std::array<int, 4>> full, cut;
        for (size_t q = 0; q + 3 < cl.quartets.size(); q += 4) full.insert(key(cl.quartets[q], cl.quartets[q + 1], cl.quartets[q + 2], cl.quartets[q + 3]));
        if (cl.twin_entries.empty()) { if (!cl.rest.empty()) ++*n_bad; continue; }
        for (size_t q = 0; q + 3 < cl.rest.size(); q += 4) {
            if (!cut.insert(key(cl.rest[q], cl.rest[q + 1], cl.rest[q + 2], cl.rest[q + 3])).second) ++*n_bad;   // duplicate
        }
        for (size_t q = 0; q + 3 < cl.twin_entries.size(); q += 4) {
            ++*n_twin_entries;
            int sh[4], m[4];
            for (int k = 0; k < 4; ++k) { sh[k] = cl.twin_entries[q + k] & 0xffff; m[k] = (cl.twin_entries[q + k] >> 16) ? 2 : 1; }
            std::set<std::array<int, 4>> mine;
            for (int a = 0; a < m[0]; ++a) for (int b = 0; b < m[1]; ++b) for (int c = 0; c < m[2]; ++c) for (int d = 0; d < m[3]; ++d)
                mine.insert(key(sh[0] + a, sh[1] + b, sh[2] + c, sh[3] + d));
            for (auto& k4 : mine) if (!cut.insert(k4).second) ++*n_bad;     // covered twice
        }
        if (cut != full) ++*n_bad;
    }
    return 0;
}

void hostcheck_boys(int L, double T, double* F)
{
    ensure_tables();
    switch (L) {
        case 0: boys<0>(T, g_boys.data(), F); break;
        case 2: boys<2>(T, g_boys.data(), F); break;
        case 4: boys<4>(T, g_boys.data(), F); break;
        case 8: boys<8>(T, g_boys.data(), F); break;
        case 12: boys<12>(T, g_boys.data(), F); break;
        case 16: boys<16>(T, g_boys.data(), F); break;
    }
}

void hostcheck_c2s(int l, double* out)
{
    ensure_tables();
    std::memcpy(out, g_c2s.data() + g_off[l], sizeof(double) * nsph(l) * ncart(l));
}

}
