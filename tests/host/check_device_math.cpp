// check_device_math.cpp -- TEST-ONLY harness.  Compiles the `__host__ __device__` integral
// templates of metalquicha_amd/csrc/md_integrals.hpp for the HOST so that tests/ can compare
// the arithmetic the gfx950 kernels instantiate against the oracle here, without a GPU.
// It is never linked into libmqc_hip.so and is not a fallback path of the product.
#include "../../metalquicha_amd/csrc/md_integrals.hpp"
#include <vector>
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
