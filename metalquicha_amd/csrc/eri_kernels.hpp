// kern_eri.hip -- four-centre ERI formation into the HBM-resident packed tensor.
//
// Replaces the reference's in-core integral pass (molecule_eris,
// backends/libcint/mqc_libcint_integrals.F90:1449, chosen by mqc_libcint_bridge.f90:819-892
// whenever the tensor fits) and supplies the Schwarz bounds of
// backends/libcint/mqc_libcint_direct.f90:105-153.
//
// MI355X mapping: the tensor is kept as a symmetric pair matrix M[f][pair(i,j)][pair(k,l)]
// (pair(i,j) = i(i+1)/2 + j, i >= j) in HBM -- 11 MB for a cc-pVDZ water dimer, so a few
// thousand fragments fit in 288 GB -- and every SCF iteration then streams it once
// (kern_fock.hip).  One thread owns one contracted shell quartet of one fragment; threads
// are ordered (quartet, fragment) with the fragment fastest, so the 64 lanes of a wave run
// the SAME quartet class with the SAME contraction depth on 64 different geometries: no
// divergence, exponents and coefficients come through scalar loads.  Each (la lb|lc ld)
// class is its own template instantiation with every inner loop unrolled into registers.
//
// Wavefront-level Schwarz screening: a lane whose bound Q_ab * Q_cd is below the threshold
// skips its primitive loops; when the whole wave is below (one ballot), the wave exits
// before touching any contraction data.
#pragma once
#include "engine.hpp"
#include "md_integrals.hpp"

namespace mqc {

__device__ __forceinline__ ShellRef make_shell(const TopologyDev& tp, const double* xyz, int s)
{
    ShellRef r;
    r.nprim = tp.sh_nprim[s];
    r.exps = tp.exps + tp.sh_poff[s];
    r.coefs = tp.coefs + tp.sh_poff[s];
    const int at = tp.sh_atom[s];
    r.x = xyz[3 * at]; r.y = xyz[3 * at + 1]; r.z = xyz[3 * at + 2];
    return r;
}

// shell quartets a wave goes on to form (Schwarz survivors): one atomic per wave on the batch's counter
// (the algorithmic unit of the integral stage, counted as direct_stats_t does, mqc_libcint_direct.f90:606-610)
__device__ __forceinline__ void count_formed(unsigned long long* ctr, unsigned long long ballot)
{
#if defined(MQC_NO_ERI_COUNT)
    return;       // MEASUREMENT build (scripts/build_variant.sh): no counter traffic
#endif
    if (ctr && (threadIdx.x & 63) == 0) atomicAdd(ctr, (unsigned long long)__popcll(ballot));
}
__device__ __forceinline__ void count_formed_sum(unsigned long long* ctr, int mine)
{
#if defined(MQC_NO_ERI_COUNT)
    return;
#endif
    int v = mine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (ctr && (threadIdx.x & 63) == 0) atomicAdd(ctr, (unsigned long long)v);
}

__device__ __forceinline__ size_t pair_index(int i, int j)
{
    return i >= j ? (size_t)i * (i + 1) / 2 + j : (size_t)j * (j + 1) / 2 + i;
}

// cart -> sph on ONE index of a block: in[pre][NC][post] -> out[pre][NS][post].
// UNR: fully unrolled (register-resident small classes) or rolled (large classes, scratch).
template <int L, int PRE, int POST, bool UNR>
__device__ __forceinline__ void c2s_one_index(const double* c2s, const double* in, double* out)
{
    constexpr int NC = ncart(L), NS = nsph(L);
    if constexpr (L < 2) {
        if constexpr (UNR) {
#pragma unroll
            for (int i = 0; i < PRE * NC * POST; ++i) out[i] = in[i];
        } else {
            for (int i = 0; i < PRE * NC * POST; ++i) out[i] = in[i];
        }
    } else if constexpr (UNR) {
#pragma unroll
        for (int a = 0; a < PRE; ++a)
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int r = 0; r < POST; ++r) {
                    double v = 0.0;
#pragma unroll
                    for (int c = 0; c < NC; ++c) v += c2s_coef<L>(c2s, s, c) * in[(a * NC + c) * POST + r];
                    out[(a * NS + s) * POST + r] = v;
                }
    } else {
#pragma unroll 1
        for (int a = 0; a < PRE; ++a)
#pragma unroll 1
            for (int s = 0; s < NS; ++s)
#pragma unroll 1
                for (int r = 0; r < POST; ++r) {
                    double v = 0.0;
                    for (int c = 0; c < NC; ++c) v += c2s_coef<L>(c2s, s, c) * in[(a * NC + c) * POST + r];
                    out[(a * NS + s) * POST + r] = v;
                }
    }
}

// all four indices; `cart` is clobbered (used as the ping-pong buffer)
template <int LA, int LB, int LC, int LD>
__device__ __forceinline__ void block_to_spherical(const double* c2s, double* cart, double* sph)
{
    constexpr int NCA = ncart(LA), NCB = ncart(LB), NCC = ncart(LC), NCD = ncart(LD);
    constexpr int NSA = nsph(LA), NSB = nsph(LB), NSC = nsph(LC), NSD = nsph(LD);
    constexpr bool UNR = (NCA * NCB * NCC * NCD <= ERI_UNROLL_LIMIT);
    (void)NSD;
    c2s_one_index<LA, 1, NCB * NCC * NCD, UNR>(c2s, cart, sph);
    c2s_one_index<LB, NSA, NCC * NCD, UNR>(c2s, sph, cart);
    c2s_one_index<LC, NSA * NSB, NCD, UNR>(c2s, cart, sph);
    c2s_one_index<LD, NSA * NSB * NSC, 1, UNR>(c2s, sph, cart);
    constexpr int NOUT = NSA * NSB * NSC * NSD;
    if constexpr (UNR) {
#pragma unroll
        for (int i = 0; i < NOUT; ++i) sph[i] = cart[i];
    } else {
        for (int i = 0; i < NOUT; ++i) sph[i] = cart[i];
    }
}

// ---------------------------------------------------------------------------------------
// Schwarz bounds Q[f][A][B] = sqrt(max |(ab|ab)|) over the spherical block.
template <int LA, int LB>
__global__ void __launch_bounds__(64) schwarz_kernel(BatchView bv, const int* __restrict__ pairs, int npairs,
                                                     double* __restrict__ Q)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long)npairs * bv.nfrag) return;
    const int ip = (int)(tid / bv.nfrag), f = (int)(tid % bv.nfrag);
    const int A = pairs[2 * ip], B = pairs[2 * ip + 1];
    const TopologyDev& tp = bv.topo;
    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    ShellRef a = make_shell(tp, xyz, A), b = make_shell(tp, xyz, B);
    constexpr int NC = ncart(LA) * ncart(LB);
    constexpr int NS = nsph(LA) * nsph(LB);
    double cart[NC * NC], sph[NC * NC];
    eri_cart_block<LA, LB, LA, LB>(a, b, a, b, bv.boys, cart);
    block_to_spherical<LA, LB, LA, LB>(bv.c2s, cart, sph);
    double m = 0.0;
    for (int i = 0; i < NS * NS; ++i) m = fmax(m, fabs(sph[i]));
    const int ns = tp.nshell;
    double* q = Q + (size_t)f * ns * ns;
    const double v = sqrt(m);
    q[A * ns + B] = v;
    q[B * ns + A] = v;
}

// ---------------------------------------------------------------------------------------
template <int LA, int LB, int LC, int LD>
__global__ void __launch_bounds__(64) eri_kernel(BatchView bv, const int* __restrict__ quartets, int nquart,
                                                 const int* __restrict__ tasks, int ntasks,
                                                 const double* __restrict__ Q, double thresh)
{
    // thread -> (list entry, fragment): the dense product, or an explicit task list (entries whose blocks
    // are shared by several fragments of the batch are formed for one representative only)
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = tasks ? (long)ntasks : (long)nquart * bv.nfrag;
    const bool live = tid < total;
    const long t = live ? tid : total - 1;
    const int iq = tasks ? tasks[2 * t] : (int)(t / bv.nfrag), f = tasks ? tasks[2 * t + 1] : (int)(t % bv.nfrag);
    const int A = quartets[4 * iq], B = quartets[4 * iq + 1], C = quartets[4 * iq + 2], D = quartets[4 * iq + 3];
    const TopologyDev& tp = bv.topo;
    const int ns = tp.nshell;
    bool keep = live;
    if (Q != nullptr) {
        const double* q = Q + (size_t)f * ns * ns;
        keep = live && (q[A * ns + B] * q[C * ns + D] >= thresh);
    }
    // wavefront-level early exit: one ballot decides for all 64 lanes
    const unsigned long long alive = __ballot(keep);
    if (alive == 0ull) return;
    count_formed(bv.eri_count, alive);
    if (!keep) return;   // the tensor was zero-filled, a skipped quartet stays zero

    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const PairFly bra(make_shell(tp, xyz, A), make_shell(tp, xyz, B)), ket(make_shell(tp, xyz, C), make_shell(tp, xyz, D));
    constexpr int NC = ncart(LA) * ncart(LB) * ncart(LC) * ncart(LD);
    constexpr int NSA = nsph(LA), NSB = nsph(LB), NSC = nsph(LC), NSD = nsph(LD);
    double cart[NC], sph[NC];
    eri_cart_block_src<LA, LB, LC, LD>(bra, ket, bv.boys, cart);
    block_to_spherical<LA, LB, LC, LD>(bv.c2s, cart, sph);

    const int oa = tp.sh_aoff[A], ob = tp.sh_aoff[B], oc = tp.sh_aoff[C], od = tp.sh_aoff[D];
    const size_t np = (size_t)bv.npair;
    const PairStore M = make_pair_store(bv, f);
#define MQC_ERI_STORE                                                              \
    {                                                                              \
        if (!(A == B && j > i)) {                                                  \
            const size_t row = pair_index(oa + i, ob + j);                         \
            MQC_INNER_PRAGMA                                                       \
            for (int k = 0; k < NSC; ++k) {                                        \
                MQC_INNER_PRAGMA                                                   \
                for (int l = 0; l < NSD; ++l) {                                    \
                    if (!(C == D && l > k)) {                                      \
                        const size_t col = pair_index(oc + k, od + l);             \
                        const double v = sph[((i * NSB + j) * NSC + k) * NSD + l]; \
                        M.put(row, col, v);                                        \
                    }                                                              \
                }                                                                  \
            }                                                                      \
        }                                                                          \
    }
    if constexpr (NC <= ERI_UNROLL_LIMIT) {
#define MQC_INNER_PRAGMA _Pragma("unroll")
#pragma unroll
        for (int i = 0; i < NSA; ++i) {
#pragma unroll
            for (int j = 0; j < NSB; ++j) MQC_ERI_STORE
        }
#undef MQC_INNER_PRAGMA
    } else {
#define MQC_INNER_PRAGMA _Pragma("unroll 1")
#pragma unroll 1
        for (int i = 0; i < NSA; ++i) {
#pragma unroll 1
            for (int j = 0; j < NSB; ++j) MQC_ERI_STORE
        }
#undef MQC_INNER_PRAGMA
    }
#undef MQC_ERI_STORE
}

// ---------------------------------------------------------------------------------------
// Twin-shell kernel (md_integrals.hpp, TwinCoefs): list entries are the FIRST member shells with bit 16
// set where the position is a twin; a thread forms the primitive integrals of its (entry, fragment)
// once and stores one spherical block per member combination.  Same thread mapping as eri_kernel.
constexpr int TWIN_FLAG = 1 << 16;

template <int LA, int LB, int LC, int LD>
__global__ void __launch_bounds__(64) eri_twin_kernel(BatchView bv, const int* __restrict__ quartets, int nquart,
                                                      const int* __restrict__ tasks, int ntasks,
                                                      const double* __restrict__ Q, double thresh)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = tasks ? (long)ntasks : (long)nquart * bv.nfrag;
    const bool live = tid < total;
    const long t = live ? tid : total - 1;
    const int iq = tasks ? tasks[2 * t] : (int)(t / bv.nfrag), f = tasks ? tasks[2 * t + 1] : (int)(t % bv.nfrag);
    const int eA = quartets[4 * iq], eB = quartets[4 * iq + 1], eC = quartets[4 * iq + 2], eD = quartets[4 * iq + 3];
    const int A = eA & (TWIN_FLAG - 1), B = eB & (TWIN_FLAG - 1), C = eC & (TWIN_FLAG - 1), D = eD & (TWIN_FLAG - 1);
    const bool tA = (eA & TWIN_FLAG) != 0, tB = (eB & TWIN_FLAG) != 0, tC = (eC & TWIN_FLAG) != 0, tD = (eD & TWIN_FLAG) != 0;
    const TopologyDev& tp = bv.topo;
    bool keep = live;
    if (Q != nullptr) {
        // Schwarz bound of the entry = the largest bound of its member combinations
        const int ns = tp.nshell;
        const double* q = Q + (size_t)f * ns * ns;
        double qab = 0.0, qcd = 0.0;
        for (int ma = 0; ma <= (tA ? 1 : 0); ++ma)
            for (int mb = 0; mb <= (tB ? 1 : 0); ++mb) qab = fmax(qab, q[(A + ma) * ns + B + mb]);
        for (int mc = 0; mc <= (tC ? 1 : 0); ++mc)
            for (int md = 0; md <= (tD ? 1 : 0); ++md) qcd = fmax(qcd, q[(C + mc) * ns + D + md]);
        keep = live && (qab * qcd >= thresh);
    }
    if (__ballot(keep) == 0ull) return;
    count_formed_sum(bv.eri_count, keep ? (tA ? 2 : 1) * (tB ? 2 : 1) * (tC ? 2 : 1) * (tD ? 2 : 1) : 0);
    if (!keep) return;
    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const ShellRef sa = make_shell(tp, xyz, A), sb = make_shell(tp, xyz, B), sc = make_shell(tp, xyz, C), sd = make_shell(tp, xyz, D);
    TwinCoefs tw;
    tw.ca[0] = sa.coefs; tw.ca[1] = tA ? tp.coefs + tp.sh_poff[A + 1] : sa.coefs; tw.fa = tA ? 1.0 : 0.0;
    tw.cb[0] = sb.coefs; tw.cb[1] = tB ? tp.coefs + tp.sh_poff[B + 1] : sb.coefs; tw.fb = tB ? 1.0 : 0.0;
    tw.cc[0] = sc.coefs; tw.cc[1] = tC ? tp.coefs + tp.sh_poff[C + 1] : sc.coefs; tw.fc = tC ? 1.0 : 0.0;
    tw.cd[0] = sd.coefs; tw.cd[1] = tD ? tp.coefs + tp.sh_poff[D + 1] : sd.coefs; tw.fd = tD ? 1.0 : 0.0;

    constexpr int NC = ncart(LA) * ncart(LB) * ncart(LC) * ncart(LD);
    constexpr int NSA = nsph(LA), NSB = nsph(LB), NSC = nsph(LC), NSD = nsph(LD);
    constexpr int MA = twin_mult(true, LA), MB = twin_mult(true, LB), MC = twin_mult(true, LC), MD = twin_mult(true, LD);
    double acc[MA * MB * MC * MD * NC];
    eri_cart_block_twin<LA, LB, LC, LD>(sa, sb, sc, sd, tw, bv.boys, acc);

    const size_t np = (size_t)bv.npair;
    const PairStore M = make_pair_store(bv, f);
#pragma unroll
    for (int ma = 0; ma < MA; ++ma) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int mc = 0; mc < MC; ++mc) {
#pragma unroll
                for (int md = 0; md < MD; ++md) {
                    if ((ma && !tA) || (mb && !tB) || (mc && !tC) || (md && !tD)) continue;
                    const int a = A + ma, b = B + mb, c = C + mc, d = D + md;
                    double sph[NC];
                    block_to_spherical<LA, LB, LC, LD>(bv.c2s, acc + (((ma * MB + mb) * MC + mc) * MD + md) * NC, sph);
                    const int oa = tp.sh_aoff[a], ob = tp.sh_aoff[b], oc = tp.sh_aoff[c], od = tp.sh_aoff[d];
#pragma unroll
                    for (int i = 0; i < NSA; ++i) {
#pragma unroll
                        for (int j = 0; j < NSB; ++j) {
                            if (a == b && j > i) continue;
                            const size_t row = pair_index(oa + i, ob + j);
#pragma unroll
                            for (int k = 0; k < NSC; ++k) {
#pragma unroll
                                for (int l = 0; l < NSD; ++l) {
                                    if (c == d && l > k) continue;
                                    const size_t col = pair_index(oc + k, od + l);
                                    const double v = sph[((i * NSB + j) * NSC + k) * NSD + l];
                                    M.put(row, col, v);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Twin entries of a SMALL batch (the literal drop-in: one fragment per call).  With lane = (entry, fragment) a batch of
// one leaves every entry to a single thread, and an oxygen (1s 2s|1s 2s) entry walks 8^4 primitive quartets: 2.5 ms of a
// 3.3 ms integral stage waited for that thread.  Here ONE wave takes an (entry, fragment): its 64 lanes share the
// entry's bra primitive pairs (lane, lane + 64, ...; eri_cart_block_twin's partial form), the partial blocks are summed
// across the wave, lane 0 transforms and stores.
template <int LA, int LB, int LC, int LD>
__global__ void __launch_bounds__(64) eri_twin_wave_kernel(BatchView bv, const int* __restrict__ quartets, int nquart,
                                                           const double* __restrict__ Q, double thresh)
{
    const int w = blockIdx.x, lane = threadIdx.x;
    const int iq = w / bv.nfrag, f = w - iq * bv.nfrag;
    if (iq >= nquart) return;
    const int eA = quartets[4 * iq], eB = quartets[4 * iq + 1], eC = quartets[4 * iq + 2], eD = quartets[4 * iq + 3];
    const int A = eA & (TWIN_FLAG - 1), B = eB & (TWIN_FLAG - 1), C = eC & (TWIN_FLAG - 1), D = eD & (TWIN_FLAG - 1);
    const bool tA = (eA & TWIN_FLAG) != 0, tB = (eB & TWIN_FLAG) != 0, tC = (eC & TWIN_FLAG) != 0, tD = (eD & TWIN_FLAG) != 0;
    const TopologyDev& tp = bv.topo;
    if (Q != nullptr) {
        const int ns = tp.nshell;
        const double* q = Q + (size_t)f * ns * ns;
        double qab = 0.0, qcd = 0.0;
        for (int ma = 0; ma <= (tA ? 1 : 0); ++ma)
            for (int mb = 0; mb <= (tB ? 1 : 0); ++mb) qab = fmax(qab, q[(A + ma) * ns + B + mb]);
        for (int mc = 0; mc <= (tC ? 1 : 0); ++mc)
            for (int md = 0; md <= (tD ? 1 : 0); ++md) qcd = fmax(qcd, q[(C + mc) * ns + D + md]);
        if (!(qab * qcd >= thresh)) return;         // wave-uniform
    }
    if (bv.eri_count && lane == 0) atomicAdd(bv.eri_count, (unsigned long long)((tA ? 2 : 1) * (tB ? 2 : 1) * (tC ? 2 : 1) * (tD ? 2 : 1)));
    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const ShellRef sa = make_shell(tp, xyz, A), sb = make_shell(tp, xyz, B), sc = make_shell(tp, xyz, C), sd = make_shell(tp, xyz, D);
    TwinCoefs tw;
    tw.ca[0] = sa.coefs; tw.ca[1] = tA ? tp.coefs + tp.sh_poff[A + 1] : sa.coefs; tw.fa = tA ? 1.0 : 0.0;
    tw.cb[0] = sb.coefs; tw.cb[1] = tB ? tp.coefs + tp.sh_poff[B + 1] : sb.coefs; tw.fb = tB ? 1.0 : 0.0;
    tw.cc[0] = sc.coefs; tw.cc[1] = tC ? tp.coefs + tp.sh_poff[C + 1] : sc.coefs; tw.fc = tC ? 1.0 : 0.0;
    tw.cd[0] = sd.coefs; tw.cd[1] = tD ? tp.coefs + tp.sh_poff[D + 1] : sd.coefs; tw.fd = tD ? 1.0 : 0.0;
    constexpr int NC = ncart(LA) * ncart(LB) * ncart(LC) * ncart(LD);
    constexpr int NSA = nsph(LA), NSB = nsph(LB), NSC = nsph(LC), NSD = nsph(LD);
    constexpr int MA = twin_mult(true, LA), MB = twin_mult(true, LB), MC = twin_mult(true, LC), MD = twin_mult(true, LD);
    double acc[MA * MB * MC * MD * NC];
    eri_cart_block_twin<LA, LB, LC, LD>(sa, sb, sc, sd, tw, bv.boys, acc, lane, 64);
#pragma unroll
    for (int i = 0; i < MA * MB * MC * MD * NC; ++i) {
        double v = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        acc[i] = v;
    }
    if (lane != 0) return;
    const PairStore M = make_pair_store(bv, f);
#pragma unroll
    for (int ma = 0; ma < MA; ++ma) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int mc = 0; mc < MC; ++mc) {
#pragma unroll
                for (int md = 0; md < MD; ++md) {
                    if ((ma && !tA) || (mb && !tB) || (mc && !tC) || (md && !tD)) continue;
                    const int a = A + ma, b = B + mb, c = C + mc, d = D + md;
                    double sph[NC];
                    block_to_spherical<LA, LB, LC, LD>(bv.c2s, acc + (((ma * MB + mb) * MC + mc) * MD + md) * NC, sph);
                    const int oa = tp.sh_aoff[a], ob = tp.sh_aoff[b], oc = tp.sh_aoff[c], od = tp.sh_aoff[d];
#pragma unroll
                    for (int i = 0; i < NSA; ++i) {
#pragma unroll
                        for (int j = 0; j < NSB; ++j) {
                            if (a == b && j > i) continue;
                            const size_t row = pair_index(oa + i, ob + j);
#pragma unroll
                            for (int k = 0; k < NSC; ++k) {
#pragma unroll
                                for (int l = 0; l < NSD; ++l) {
                                    if (c == d && l > k) continue;
                                    M.put(row, pair_index(oc + k, od + l), sph[((i * NSB + j) * NSC + k) * NSD + l]);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Pass kernel for the classes whose accumulators exceed the register file (md_integrals.hpp,
// eri_pass): same (quartet, fragment) thread mapping; the accumulators of a wave sit in its own
// LDS slab acc[entry][lane].
struct TensorSink {
    PairStore M;
    int oa, ob, oc, od;
    bool ab_same, cd_same;
    __device__ __forceinline__ void operator()(int i, int j, int k, int l, double v) const
    {
        if (ab_same && j > i) return;
        if (cd_same && l > k) return;
        const size_t row = pair_index(oa + i, ob + j), col = pair_index(oc + k, od + l);
        M.put(row, col, v);
    }
};

template <int LA, int LB, int LC, int LD>
__global__ void __launch_bounds__(64) eri_pass_kernel(BatchView bv, const int* __restrict__ quartets, int nquart,
                                                      const int* __restrict__ tasks, int ntasks,
                                                      const double* __restrict__ Q, double thresh)
{
    extern __shared__ double lds[];
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = tasks ? (long)ntasks : (long)nquart * bv.nfrag;
    const bool live = tid < total;
    const long t = live ? tid : total - 1;
    const int iq = tasks ? tasks[2 * t] : (int)(t / bv.nfrag), f = tasks ? tasks[2 * t + 1] : (int)(t % bv.nfrag);
    const int A = quartets[4 * iq], B = quartets[4 * iq + 1], C = quartets[4 * iq + 2], D = quartets[4 * iq + 3];
    const TopologyDev& tp = bv.topo;
    const int ns = tp.nshell;
    bool keep = live;
    if (Q != nullptr) {
        const double* q = Q + (size_t)f * ns * ns;
        keep = live && (q[A * ns + B] * q[C * ns + D] >= thresh);
    }
    const unsigned long long alive = __ballot(keep);
    if (alive == 0ull) return;
    count_formed(bv.eri_count, alive);
    if (!keep) return;
    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const PairFly bra(make_shell(tp, xyz, A), make_shell(tp, xyz, B)), ket(make_shell(tp, xyz, C), make_shell(tp, xyz, D));
    const size_t np = (size_t)bv.npair;
    TensorSink sink{make_pair_store(bv, f), tp.sh_aoff[A], tp.sh_aoff[B], tp.sh_aoff[C], tp.sh_aoff[D], A == B, C == D};
    constexpr int CH = eri_pass_chunk(LA, LB, LC, LD);
    eri_passes_src<LA, LB, LC, LD, CH, 0>(bra, ket, bv.boys, bv.c2s, lds + threadIdx.x, 64, sink);
}

struct MaxSink {
    double m;
    __device__ __forceinline__ void operator()(int, int, int, int, double v) { m = fmax(m, fabs(v)); }
};

template <int LA, int LB>
__global__ void __launch_bounds__(64) schwarz_pass_kernel(BatchView bv, const int* __restrict__ pairs, int npairs,
                                                          double* __restrict__ Q)
{
    extern __shared__ double lds[];
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long)npairs * bv.nfrag) return;
    const int ip = (int)(tid / bv.nfrag), f = (int)(tid % bv.nfrag);
    const int A = pairs[2 * ip], B = pairs[2 * ip + 1];
    const TopologyDev& tp = bv.topo;
    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    ShellRef a = make_shell(tp, xyz, A), b = make_shell(tp, xyz, B);
    MaxSink sink{0.0};
    constexpr int CH = eri_pass_chunk(LA, LB, LA, LB);
    eri_passes_from<LA, LB, LA, LB, CH, 0>(a, b, a, b, bv.boys, bv.c2s, lds + threadIdx.x, 64, sink);
    const int ns = tp.nshell;
    double* q = Q + (size_t)f * ns * ns;
    const double v = sqrt(sink.m);
    q[A * ns + B] = v;
    q[B * ns + A] = v;
}

// ---------------------------------------------------------------------------------------
// Direct (integral-recomputing) Fock build: the same quartet arithmetic, but the block is digested
// into J~ and K~ straight away instead of being stored -- build_fock_direct,
// backends/libcint/mqc_libcint_direct.f90:306-620 (Huang/Sherrill/Chow JCP 152, 024122 Alg. 1).
// Unique quartets only (ab >= cd); per block six pre-contracted scatter updates with the shell
// degeneracy weight w = deg/8:
//     J~_ij += 4 w sum_kl v D_kl      J~_kl += 4 w sum_ij v D_ij
//     K~_ik += 2 w sum_jl v D_jl      K~_jk += 2 w sum_il v D_il
//     K~_il += 2 w sum_jk v D_jk      K~_jl += 2 w sum_ik v D_ik
// and J = (J~ + J~^T)/2, K = (K~ + K~^T)/2 afterwards (symmetrise_jk_kernel).
// Screening (mqc_libcint_direct.f90:266-288,533-551): keep iff
//     Q_ab Q_cd deg max( max(Dab, Dcd)/2, (k/8) max(Dac, Dad, Dbc, Dbd) ) >= tol,
// D.. = max |D| over the shell block, k = exact-exchange fraction; one ballot lets a whole
// wavefront leave before it touches any contraction data.
template <int LA, int LB, int LC, int LD>
__global__ void __launch_bounds__(64) eri_digest_kernel(BatchView bv, const int* __restrict__ quartets, int nquart,
                                                        const double* __restrict__ Q, const double* __restrict__ Dmax,
                                                        double thresh, double* __restrict__ Jt, double* __restrict__ Kt,
                                                        int only_active)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)nquart * bv.nfrag;
    const bool live = tid < total;
    const long t = live ? tid : total - 1;
    const int iq = (int)(t / bv.nfrag), f = (int)(t % bv.nfrag);
    const int A = quartets[4 * iq], B = quartets[4 * iq + 1], C = quartets[4 * iq + 2], D = quartets[4 * iq + 3];
    const TopologyDev& tp = bv.topo;
    const int ns = tp.nshell;
    const double sab = (A == B) ? 1.0 : 2.0, scd = (C == D) ? 1.0 : 2.0;
    const bool same = (A == C && B == D) || (A == D && B == C);
    const double deg = sab * scd * (same ? 1.0 : 2.0);
    bool keep = live && !(only_active && bv.istate[4 * f] == ST_DONE);
    if (keep) {
        const double* q = Q + (size_t)f * ns * ns;
        const double* dm = Dmax + (size_t)f * ns * ns;
        const double dj = 0.5 * fmax(dm[A * ns + B], dm[C * ns + D]);
        const double dk = 0.125 * bv.exx * fmax(fmax(dm[A * ns + C], dm[A * ns + D]), fmax(dm[B * ns + C], dm[B * ns + D]));
        keep = q[A * ns + B] * q[C * ns + D] * deg * fmax(dj, dk) >= thresh;
    }
    const unsigned long long alive = __ballot(keep);
    if (alive == 0ull) return;
    count_formed(bv.eri_count, alive);
    if (!keep) return;

    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const PairFly bra(make_shell(tp, xyz, A), make_shell(tp, xyz, B)), ket(make_shell(tp, xyz, C), make_shell(tp, xyz, D));
    constexpr int NC = ncart(LA) * ncart(LB) * ncart(LC) * ncart(LD);
    constexpr int NSA = nsph(LA), NSB = nsph(LB), NSC = nsph(LC), NSD = nsph(LD);
    double cart[NC], sph[NC];
    eri_cart_block_src<LA, LB, LC, LD>(bra, ket, bv.boys, cart);
    block_to_spherical<LA, LB, LC, LD>(bv.c2s, cart, sph);

    const int n = bv.n;
    const int oa = tp.sh_aoff[A], ob = tp.sh_aoff[B], oc = tp.sh_aoff[C], od = tp.sh_aoff[D];
    const double* __restrict__ Dm = bv.D + (size_t)f * n * n;
    double* __restrict__ J = Jt + (size_t)f * n * n;
    double* __restrict__ K = Kt + (size_t)f * n * n;
    const double wj = 4.0 * deg / 8.0, wk = 2.0 * deg / 8.0;
#define V(i, j, k, l) sph[(((i) * NSB + (j)) * NSC + (k)) * NSD + (l)]
    // J~_ij and J~_kl
#pragma unroll 1
    for (int i = 0; i < NSA; ++i)
#pragma unroll 1
        for (int j = 0; j < NSB; ++j) {
            double s = 0.0;
            for (int k = 0; k < NSC; ++k)
                for (int l = 0; l < NSD; ++l) s += V(i, j, k, l) * Dm[(oc + k) * n + od + l];
            atomicAdd(&J[(oa + i) * n + ob + j], wj * s);
        }
#pragma unroll 1
    for (int k = 0; k < NSC; ++k)
#pragma unroll 1
        for (int l = 0; l < NSD; ++l) {
            double s = 0.0;
            for (int i = 0; i < NSA; ++i)
                for (int j = 0; j < NSB; ++j) s += V(i, j, k, l) * Dm[(oa + i) * n + ob + j];
            atomicAdd(&J[(oc + k) * n + od + l], wj * s);
        }
    // K~_ik, K~_il, K~_jk, K~_jl -- unless no exact exchange is asked for (pure functionals; the Coulomb-only requests of
    // mqc_hip_coulomb_batch): the exchange digest is two thirds of the scatter work (wave-uniform branch)
    if (bv.exx == 0.0) return;
#pragma unroll 1
    for (int i = 0; i < NSA; ++i)
#pragma unroll 1
        for (int k = 0; k < NSC; ++k) {
            double s = 0.0;
            for (int j = 0; j < NSB; ++j)
                for (int l = 0; l < NSD; ++l) s += V(i, j, k, l) * Dm[(ob + j) * n + od + l];
            atomicAdd(&K[(oa + i) * n + oc + k], wk * s);
        }
#pragma unroll 1
    for (int i = 0; i < NSA; ++i)
#pragma unroll 1
        for (int l = 0; l < NSD; ++l) {
            double s = 0.0;
            for (int j = 0; j < NSB; ++j)
                for (int k = 0; k < NSC; ++k) s += V(i, j, k, l) * Dm[(ob + j) * n + oc + k];
            atomicAdd(&K[(oa + i) * n + od + l], wk * s);
        }
#pragma unroll 1
    for (int j = 0; j < NSB; ++j)
#pragma unroll 1
        for (int k = 0; k < NSC; ++k) {
            double s = 0.0;
            for (int i = 0; i < NSA; ++i)
                for (int l = 0; l < NSD; ++l) s += V(i, j, k, l) * Dm[(oa + i) * n + od + l];
            atomicAdd(&K[(ob + j) * n + oc + k], wk * s);
        }
#pragma unroll 1
    for (int j = 0; j < NSB; ++j)
#pragma unroll 1
        for (int l = 0; l < NSD; ++l) {
            double s = 0.0;
            for (int i = 0; i < NSA; ++i)
                for (int k = 0; k < NSC; ++k) s += V(i, j, k, l) * Dm[(oa + i) * n + oc + k];
            atomicAdd(&K[(ob + j) * n + od + l], wk * s);
        }
#undef V
}

template <int LA, int LB, int LC, int LD>
void launch_eri_digest_class(const BatchView& bv, const int* d_list, int nq, const double* Q, const double* Dmax,
                             double thresh, double* Jt, double* Kt, int only_active, hipStream_t s)
{
    if (nq == 0) return;
    const long total = (long)nq * bv.nfrag;
    const int blocks = (int)((total + 63) / 64);
    hipLaunchKernelGGL((eri_digest_kernel<LA, LB, LC, LD>), dim3(blocks), dim3(64), 0, s, bv, d_list, nq, Q, Dmax, thresh, Jt, Kt, only_active);
}

// ---------------------------------------------------------------------------------------
// Launchers: one explicit instantiation per class, spread over several translation units
// (kern_eri_inst.hip compiled with -DERI_GROUP=k) so that the classes compile in parallel.
// d_list: entries already on the device; d_tasks (optional): (entry, fragment) pairs replacing the dense product
template <int LA, int LB, int LC, int LD>
void launch_eri_class(const BatchView& bv, const int* d_list, int nq, const int* d_tasks, int ntasks,
                      const double* Q, double thresh, hipStream_t s)
{
    const long total = d_tasks ? (long)ntasks : (long)nq * bv.nfrag;
    if (nq == 0 || total == 0) return;
    const int blocks = (int)((total + 63) / 64);
    if constexpr (eri_uses_passes(LA, LB, LC, LD)) {
        constexpr int CH = eri_pass_chunk(LA, LB, LC, LD);
        const size_t lds = sizeof(double) * 64 * ncart(LA) * ncart(LB) * CH;
        auto kern = eri_pass_kernel<LA, LB, LC, LD>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, s, bv, d_list, nq, d_tasks, ntasks, Q, thresh);
    } else {
        hipLaunchKernelGGL((eri_kernel<LA, LB, LC, LD>), dim3(blocks), dim3(64), 0, s, bv, d_list, nq, d_tasks, ntasks, Q, thresh);
    }
}

template <int LA, int LB, int LC, int LD>
void launch_eri_twin_class(const BatchView& bv, const int* d_list, int nq, const int* d_tasks, int ntasks,
                           const double* Q, double thresh, hipStream_t s)
{
    const long total = d_tasks ? (long)ntasks : (long)nq * bv.nfrag;
    if (nq == 0 || total == 0) return;
    hipLaunchKernelGGL((eri_twin_kernel<LA, LB, LC, LD>), dim3((int)((total + 63) / 64)), dim3(64), 0, s, bv, d_list, nq, d_tasks, ntasks, Q, thresh);
}

// batches of at most this many fragments form their twin entries one WAVE per (entry, fragment)
constexpr int ERI_TWIN_WAVE_MAX_FRAGMENTS = 16;
template <int LA, int LB, int LC, int LD>
void launch_eri_twin_wave_class(const BatchView& bv, const int* d_list, int nq, const double* Q, double thresh, hipStream_t s)
{
    if (nq == 0 || bv.nfrag == 0) return;
    hipLaunchKernelGGL((eri_twin_wave_kernel<LA, LB, LC, LD>), dim3((unsigned)nq * (unsigned)bv.nfrag), dim3(64), 0, s, bv, d_list, nq, Q, thresh);
}

template <int LA, int LB>
void launch_schwarz_class(const BatchView& bv, const int* pairs_host, int np, int* d_list, double* Q, hipStream_t s)
{
    if (np == 0) return;
    (void)hipMemcpyAsync(d_list, pairs_host, (size_t)np * 2 * sizeof(int), hipMemcpyHostToDevice, s);
    const long total = (long)np * bv.nfrag;
    if constexpr (schwarz_uses_passes(LA, LB)) {
        constexpr int CH = eri_pass_chunk(LA, LB, LA, LB);
        const size_t lds = sizeof(double) * 64 * ncart(LA) * ncart(LB) * CH;
        auto kern = schwarz_pass_kernel<LA, LB>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((int)((total + 63) / 64)), dim3(64), lds, s, bv, d_list, np, Q);
    } else {
        hipLaunchKernelGGL((schwarz_kernel<LA, LB>), dim3((int)((total + 63) / 64)), dim3(64), 0, s, bv, d_list, np, Q);
    }
}

}  // namespace mqc
