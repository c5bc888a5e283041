// kern_eri.hip -- dispatcher of the four-centre ERI formation (kernels: eri_kernels.hpp,
// instantiated per class group in kern_eri_inst.hip).
#include "eri_kernels.hpp"

namespace mqc {

#define ERI_DECL(a, b, c, d) \
    extern template void launch_eri_class<a, b, c, d>(const BatchView&, const int*, int, int*, const double*, double, hipStream_t);
#define SCHWARZ_DECL(a, b) \
    extern template void launch_schwarz_class<a, b>(const BatchView&, const int*, int, int*, double*, hipStream_t);
ERI_DECL(0, 0, 0, 0) ERI_DECL(1, 0, 0, 0) ERI_DECL(1, 0, 1, 0) ERI_DECL(1, 1, 0, 0) ERI_DECL(1, 1, 1, 0) ERI_DECL(1, 1, 1, 1)
ERI_DECL(2, 0, 0, 0) ERI_DECL(2, 0, 1, 0) ERI_DECL(2, 0, 1, 1) ERI_DECL(2, 0, 2, 0)
ERI_DECL(2, 1, 0, 0) ERI_DECL(2, 1, 1, 0) ERI_DECL(2, 1, 1, 1) ERI_DECL(2, 1, 2, 0) ERI_DECL(2, 1, 2, 1)
ERI_DECL(2, 2, 0, 0) ERI_DECL(2, 2, 1, 0) ERI_DECL(2, 2, 1, 1) ERI_DECL(2, 2, 2, 0) ERI_DECL(2, 2, 2, 1) ERI_DECL(2, 2, 2, 2)
#define TWIN_DECL(a, b, c, d) \
    extern template void launch_eri_twin_class<a, b, c, d>(const BatchView&, const int*, int, int*, hipStream_t);
TWIN_DECL(0, 0, 0, 0) TWIN_DECL(1, 0, 0, 0) TWIN_DECL(1, 0, 1, 0) TWIN_DECL(1, 1, 0, 0)
TWIN_DECL(1, 1, 1, 0) TWIN_DECL(2, 0, 0, 0) TWIN_DECL(2, 0, 1, 0) TWIN_DECL(2, 1, 0, 0)
SCHWARZ_DECL(0, 0) SCHWARZ_DECL(1, 0) SCHWARZ_DECL(1, 1) SCHWARZ_DECL(2, 0) SCHWARZ_DECL(2, 1) SCHWARZ_DECL(2, 2)

#define DIG_DECL(a, b, c, d) \
    extern template void launch_eri_digest_class<a, b, c, d>(const BatchView&, const int*, int, const double*, const double*, double, double*, double*, int, hipStream_t);
DIG_DECL(0, 0, 0, 0) DIG_DECL(1, 0, 0, 0) DIG_DECL(1, 0, 1, 0) DIG_DECL(1, 1, 0, 0) DIG_DECL(1, 1, 1, 0) DIG_DECL(1, 1, 1, 1)
DIG_DECL(2, 0, 0, 0) DIG_DECL(2, 0, 1, 0) DIG_DECL(2, 0, 1, 1) DIG_DECL(2, 0, 2, 0)
DIG_DECL(2, 1, 0, 0) DIG_DECL(2, 1, 1, 0) DIG_DECL(2, 1, 1, 1) DIG_DECL(2, 1, 2, 0) DIG_DECL(2, 1, 2, 1)
DIG_DECL(2, 2, 0, 0) DIG_DECL(2, 2, 1, 0) DIG_DECL(2, 2, 1, 1) DIG_DECL(2, 2, 2, 0) DIG_DECL(2, 2, 2, 1) DIG_DECL(2, 2, 2, 2)

#define ERI_CASE(a, b, c, d)                                                                          \
    if (cl.la == a && cl.lb == b && cl.lc == c && cl.ld == d)                                         \
        launch_eri_class<a, b, c, d>(bv, cl.quartets.data(), (int)cl.quartets.size() / 4, d_list + off, Q, thresh, s);

// MQC_HIP_NO_TWIN_BLOCKS=1 forces the segmented treatment everywhere (A/B measurements, tests)
static bool twin_blocks_disabled()
{
    static const bool off = [] { const char* e = std::getenv("MQC_HIP_NO_TWIN_BLOCKS"); return e && e[0] == '1'; }();
    return off;
}

void launch_eri(const BatchView& bv, const Topology& topo, double schwarz_tol, hipStream_t s)
{
    static DevicePool lists_slot[2], qpool_slot[2];
    DevicePool& lists = lists_slot[bv.slot & 1];
    DevicePool& qpool = qpool_slot[bv.slot & 1];
    // one device buffer for all class lists so that launches need no intermediate sync
    size_t total_ints = 0;
    for (auto& cl : topo.classes) total_ints += cl.quartets.size();
    total_ints += topo.pairs.size();
    int* d_list = (int*)lists.ensure((total_ints + 16) * sizeof(int));
    const size_t np = (size_t)bv.npair;
    (void)hipMemsetAsync(bv.eri, 0, sizeof(double) * np * np * bv.nfrag, s);

    double* Q = nullptr;
    double thresh = 0.0;
    size_t off = 0;
    if (schwarz_tol > 0.0) {
        Q = (double*)qpool.ensure(sizeof(double) * (size_t)bv.nfrag * topo.shells.size() * topo.shells.size());
        thresh = schwarz_tol;
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
        SCHWARZ_CASE(2, 0) SCHWARZ_CASE(2, 1) SCHWARZ_CASE(2, 2)
#undef SCHWARZ_CASE
        (void)hipStreamSynchronize(s);     // the bucket vectors go out of scope
    }
    // twin-shell cut (exact ERIs without Schwarz screening): twin entries first, then the uncovered rest
    const bool twins = (schwarz_tol <= 0.0) && !twin_blocks_disabled();
    for (auto& cl : topo.classes) {
        if (twins && !cl.twin_entries.empty()) {
            const int nt = (int)cl.twin_entries.size() / 4, nr = (int)cl.rest.size() / 4;
#define TWIN_CASE(a, b, c, d)                                                                          \
    if (cl.la == a && cl.lb == b && cl.lc == c && cl.ld == d) {                                         \
        launch_eri_twin_class<a, b, c, d>(bv, cl.twin_entries.data(), nt, d_list + off, s);             \
        launch_eri_class<a, b, c, d>(bv, cl.rest.data(), nr, d_list + off + cl.twin_entries.size(), nullptr, 0.0, s); \
    }
            TWIN_CASE(0, 0, 0, 0) TWIN_CASE(1, 0, 0, 0) TWIN_CASE(1, 0, 1, 0) TWIN_CASE(1, 1, 0, 0)
            TWIN_CASE(1, 1, 1, 0) TWIN_CASE(2, 0, 0, 0) TWIN_CASE(2, 0, 1, 0) TWIN_CASE(2, 1, 0, 0)
#undef TWIN_CASE
            off += cl.quartets.size();      // twin entries + rest never exceed the full list
            continue;
        }
        ERI_CASE(0, 0, 0, 0)
        ERI_CASE(1, 0, 0, 0) ERI_CASE(1, 0, 1, 0)
        ERI_CASE(1, 1, 0, 0) ERI_CASE(1, 1, 1, 0) ERI_CASE(1, 1, 1, 1)
        ERI_CASE(2, 0, 0, 0) ERI_CASE(2, 0, 1, 0) ERI_CASE(2, 0, 1, 1) ERI_CASE(2, 0, 2, 0)
        ERI_CASE(2, 1, 0, 0) ERI_CASE(2, 1, 1, 0) ERI_CASE(2, 1, 1, 1) ERI_CASE(2, 1, 2, 0) ERI_CASE(2, 1, 2, 1)
        ERI_CASE(2, 2, 0, 0) ERI_CASE(2, 2, 1, 0) ERI_CASE(2, 2, 1, 1) ERI_CASE(2, 2, 2, 0) ERI_CASE(2, 2, 2, 1)
        ERI_CASE(2, 2, 2, 2)
        off += cl.quartets.size();
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
void launch_direct_setup(const BatchView& bv, const Topology& topo, hipStream_t s)
{
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
    SCHWARZ_CASE(2, 0) SCHWARZ_CASE(2, 1) SCHWARZ_CASE(2, 2)
#undef SCHWARZ_CASE
    (void)hipStreamSynchronize(s);
}

#define DIG_CASE(a, b, c, d)                                                                          \
    if (cl.la == a && cl.lb == b && cl.lc == c && cl.ld == d)                                         \
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

}  // namespace mqc
