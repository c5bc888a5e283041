// kern_int1e.hip -- overlap, kinetic and nuclear-attraction matrices for a whole batch.
//
// Replaces cuestOverlapCompute / KineticCompute / PotentialCompute as called from
// backends/cuest/backend/mqc_cuest_integrals.f90:1525-1634 (CPU twin: one_electron,
// backends/libcint/mqc_libcint_integrals.F90:843-911).  One thread per (shell pair,
// fragment), fragment fastest, so a wave works on one shell pair of 64 different
// fragments: identical control flow, contraction data fetched through the scalar cache.
// Cost is O(n_pairs * n_atoms) per fragment and is paid once; the kernel is not a hot spot.
#include "engine.hpp"
#include "md_integrals.hpp"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace mqc {

// Far field of an embedding with many point charges (FMO / EE-MBE: ~1500 per fragment at 512 fragments).  For a shell
// pair on ONE atom every primitive product sits on that atom, and a charge at distance R with p_min R^2 > 42 is in the
// asymptotic branch of the Boys function for every primitive pair: R_tuv(p, A - C) = (1/2) sqrt(pi/p) d^{tuv}(1/|A - C|)
// to exp(-42).  The p-independent sums over those charges are formed ONCE per (fragment, atom) by pc_far_table_kernel;
// int1e_block then loops over the near charges only for such pairs -- it used to evaluate one Boys function per
// (primitive pair, charge): 81 x 1530 for an oxygen s-s pair (int1e_kernel<0,0>: 388 ms per launch at 512 fragments,
// profiles/r02_af...).  Pairs on two atoms keep the direct sum.
constexpr int PC_FAR_LMAX = 4;                 // s, p, d pairs (f pairs keep the direct sum)
constexpr int PC_FAR_NT = nherm(PC_FAR_LMAX);  // 35
constexpr int PC_FAR_MIN = 64;                 // fields smaller than this are summed directly

__global__ void __launch_bounds__(64) pc_far_table_kernel(BatchView bv, const double* __restrict__ far_r2, double* __restrict__ tab)
{
    // thread = fragment (the charge array is fragment-fastest: coalesced), block column = atom
    const int a = blockIdx.y, f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= bv.nfrag) return;
    const TopologyDev& tp = bv.topo;
    const double* xyz = bv.xyz + ((size_t)f * tp.natoms + a) * 3;
    const double ax = xyz[0], ay = xyz[1], az = xyz[2], r2far = far_r2[a];
    const size_t nf = (size_t)bv.nfrag;
    const double* pc = bv.pc + f;
    double acc[PC_FAR_NT];
#pragma unroll
    for (int k = 0; k < PC_FAR_NT; ++k) acc[k] = 0.0;
    for (int g = 0; g < bv.npc; ++g) {
        const double* c = pc + (size_t)g * 4 * nf;
        const double X = ax - c[0], Y = ay - c[nf], Z = az - c[2 * nf], q = c[3 * nf];
        const double r2 = X * X + Y * Y + Z * Z;
        if (!(r2 > r2far) || q == 0.0) continue;
        // seeds of the Hermite recursion in the far limit: (-1)^n (2n - 1)!! / R^(2n+1)
        const double inv = 1.0 / sqrt(r2), inv2 = inv * inv;
        double F[PC_FAR_LMAX + 1];
        F[0] = inv;
#pragma unroll
        for (int n = 1; n <= PC_FAR_LMAX; ++n) F[n] = -(2 * n - 1) * F[n - 1] * inv2;
        double G[PC_FAR_NT];
        HermiteLevel<PC_FAR_LMAX, 0>::run(F, X, Y, Z, G);
#pragma unroll
        for (int k = 0; k < PC_FAR_NT; ++k) acc[k] += q * G[k];
    }
    double* out = tab + ((size_t)f * tp.natoms + a) * PC_FAR_NT;
#pragma unroll
    for (int k = 0; k < PC_FAR_NT; ++k) out[k] = acc[k];
}

template <int LA, int LB>
__device__ void int1e_block(const BatchView& bv, int f, int A, int B)
{
    constexpr int NCA = ncart(LA), NCB = ncart(LB);
    constexpr int L = LA + LB;
    const TopologyDev& tp = bv.topo;
    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const int atA = tp.sh_atom[A], atB = tp.sh_atom[B];
    const double ax = xyz[3 * atA], ay = xyz[3 * atA + 1], az = xyz[3 * atA + 2];
    const double bx = xyz[3 * atB], by = xyz[3 * atB + 1], bz = xyz[3 * atB + 2];
    const double ab2 = (ax - bx) * (ax - bx) + (ay - by) * (ay - by) + (az - bz) * (az - bz);
    double sc[NCA * NCB], tc[NCA * NCB], vc[NCA * NCB], uc[NCA * NCB];
#pragma unroll
    for (int i = 0; i < NCA * NCB; ++i) { sc[i] = 0.0; tc[i] = 0.0; vc[i] = 0.0; uc[i] = 0.0; }
    const int npc = bv.npc;
    const size_t nfp = (size_t)bv.nfrag;
    const double* pc = bv.pc + f;                 // [charge][x, y, z, q][fragment]
    const bool use_far = L <= PC_FAR_LMAX && atA == atB && npc > 0 && bv.pc_far_tab != nullptr;
    const double far_r2 = use_far ? bv.pc_far_r2[atA] : 0.0;
    const double* far_tab = use_far ? bv.pc_far_tab + ((size_t)f * tp.natoms + atA) * PC_FAR_NT : nullptr;
    const int npa = tp.sh_nprim[A], npb = tp.sh_nprim[B];
    const double* ea = tp.exps + tp.sh_poff[A]; const double* ca = tp.coefs + tp.sh_poff[A];
    const double* eb = tp.exps + tp.sh_poff[B]; const double* cb = tp.coefs + tp.sh_poff[B];
    for (int ip = 0; ip < npa; ++ip)
        for (int jp = 0; jp < npb; ++jp) {
            const double a = ea[ip], b = eb[jp], p = a + b, ip_ = 1.0 / p;
            const double kab = exp(-a * b * ip_ * ab2) * ca[ip] * cb[jp];
            const double px = (a * ax + b * bx) * ip_, py = (a * ay + b * by) * ip_, pz = (a * az + b * bz) * ip_;
            E1D<LA, LB + 2> ex, ey, ez;
            ex.build(px - ax, px - bx, 0.5 * ip_);
            ey.build(py - ay, py - by, 0.5 * ip_);
            ez.build(pz - az, pz - bz, 0.5 * ip_);
            const double s3 = kab * M_PI * ip_ * sqrt(M_PI * ip_);
            int iab = 0;
#pragma unroll
            for (int i0 = LA; i0 >= 0; --i0)
#pragma unroll
                for (int i1 = LA - i0; i1 >= 0; --i1) {
                    const int i2 = LA - i0 - i1;
#pragma unroll
                    for (int j0 = LB; j0 >= 0; --j0)
#pragma unroll
                        for (int j1 = LB - j0; j1 >= 0; --j1) {
                            const int j2 = LB - j0 - j1;
                            const double sx = ex.get(i0, j0, 0), sy = ey.get(i1, j1, 0), sz = ez.get(i2, j2, 0);
                            double tx = -2.0 * b * (2 * j0 + 1) * sx + 4.0 * b * b * ex.get(i0, j0 + 2, 0);
                            if (j0 >= 2) tx += j0 * (j0 - 1) * ex.get(i0, j0 - 2, 0);
                            double ty = -2.0 * b * (2 * j1 + 1) * sy + 4.0 * b * b * ey.get(i1, j1 + 2, 0);
                            if (j1 >= 2) ty += j1 * (j1 - 1) * ey.get(i1, j1 - 2, 0);
                            double tz = -2.0 * b * (2 * j2 + 1) * sz + 4.0 * b * b * ez.get(i2, j2 + 2, 0);
                            if (j2 >= 2) tz += j2 * (j2 - 1) * ez.get(i2, j2 - 2, 0);
                            sc[iab] += s3 * sx * sy * sz;
                            tc[iab] += -0.5 * s3 * (tx * sy * sz + sx * ty * sz + sx * sy * tz);
                            ++iab;
                        }
                }
            // nuclear attraction: sum over (non-ghost) centres, then the external point charges of an embedded
            // fragment (embedding_operator, mqc_libcint_fmo.f90:1143-1151), kept apart in uc
            for (int at = 0; at < tp.natoms + npc; ++at) {
                const bool ext = at >= tp.natoms;
                double cx, cy, cz, zq;
                if (ext) {
                    const double* c = pc + (size_t)(at - tp.natoms) * 4 * nfp;
                    cx = c[0]; cy = c[nfp]; cz = c[2 * nfp]; zq = c[3 * nfp];
                } else {
                    cx = xyz[3 * at]; cy = xyz[3 * at + 1]; cz = xyz[3 * at + 2]; zq = tp.zeff[at];
                }
                if (zq == 0.0) continue;
                if (ext && use_far) {
                    // same-centre pair (P = A for every primitive pair): a charge beyond the atom's far radius is in the
                    // asymptotic branch of the Boys function for ALL of them and sits in the per-atom table instead
                    const double fx = ax - cx, fy = ay - cy, fz = az - cz;
                    if (fx * fx + fy * fy + fz * fz > far_r2) continue;
                }
                double R[nherm(L)];
                hermite_r<L>(p, px - cx, py - cy, pz - cz, bv.boys, R);
                const double pref = -zq * 2.0 * M_PI * ip_ * kab;
                int k = 0;
#pragma unroll
                for (int i0 = LA; i0 >= 0; --i0)
#pragma unroll
                    for (int i1 = LA - i0; i1 >= 0; --i1) {
                        const int i2 = LA - i0 - i1;
#pragma unroll
                        for (int j0 = LB; j0 >= 0; --j0)
#pragma unroll
                            for (int j1 = LB - j0; j1 >= 0; --j1) {
                                const int j2 = LB - j0 - j1;
                                double v = 0.0;
#pragma unroll
                                for (int t = 0; t <= i0 + j0; ++t)
#pragma unroll
                                    for (int u = 0; u <= i1 + j1; ++u)
#pragma unroll
                                        for (int w = 0; w <= i2 + j2; ++w)
                                            v += ex.get(i0, j0, t) * ey.get(i1, j1, u) * ez.get(i2, j2, w) * R[hidx(t, u, w)];
                                vc[k] += ext ? 0.0 : pref * v;
                                uc[k] += ext ? pref * v : 0.0;
                                ++k;
                            }
                    }
            }
            if constexpr (L <= PC_FAR_LMAX) {
                if (use_far) {
                    // the far charges of this centre, all at once: R_tuv -> (1/2) sqrt(pi/p) d^{tuv}(1/|A - C|) for
                    // p R^2 > 42 (error below exp(-42)), and the sum over the charges does not depend on p
                    const double pref = -2.0 * M_PI * ip_ * kab * 0.886226925452758014 * sqrt(ip_);
                    int k = 0;
#pragma unroll
                    for (int i0 = LA; i0 >= 0; --i0)
#pragma unroll
                        for (int i1 = LA - i0; i1 >= 0; --i1) {
                            const int i2 = LA - i0 - i1;
#pragma unroll
                            for (int j0 = LB; j0 >= 0; --j0)
#pragma unroll
                                for (int j1 = LB - j0; j1 >= 0; --j1) {
                                    const int j2 = LB - j0 - j1;
                                    double v = 0.0;
#pragma unroll
                                    for (int t = 0; t <= i0 + j0; ++t)
#pragma unroll
                                        for (int u = 0; u <= i1 + j1; ++u)
#pragma unroll
                                            for (int w = 0; w <= i2 + j2; ++w)
                                                v += ex.get(i0, j0, t) * ey.get(i1, j1, u) * ez.get(i2, j2, w) * far_tab[hidx(t, u, w)];
                                    uc[k] += pref * v;
                                    ++k;
                                }
                        }
                }
            }
        }
    // cart -> sph (s and p carry their factor in the coefficients; l >= 2 uses the table)
    constexpr int NSA = nsph(LA), NSB = nsph(LB);
    const int n = bv.n;
    double* S = bv.S + (size_t)f * n * n;
    double* H = bv.H + (size_t)f * n * n;
    double* T = bv.W + (size_t)f * 6 * n * n;   // W[0] receives T, W[1] receives V (for the stage-level API)
    double* V = T + (size_t)n * n;
    const int oa = tp.sh_aoff[A], ob = tp.sh_aoff[B];
#pragma unroll
    for (int i = 0; i < NSA; ++i)
#pragma unroll
        for (int j = 0; j < NSB; ++j) {
            double s = 0.0, t = 0.0, v = 0.0, u = 0.0;
#pragma unroll
            for (int ia = 0; ia < NCA; ++ia) {
                double wa = 1.0;
                if constexpr (LA >= 2) wa = c2s_coef<LA>(bv.c2s, i, ia); else wa = (i == ia) ? 1.0 : 0.0;
                if (wa == 0.0) continue;
#pragma unroll
                for (int ib = 0; ib < NCB; ++ib) {
                    double wb = 1.0;
                    if constexpr (LB >= 2) wb = c2s_coef<LB>(bv.c2s, j, ib); else wb = (j == ib) ? 1.0 : 0.0;
                    const double w = wa * wb;
                    s += w * sc[ia * NCB + ib];
                    t += w * tc[ia * NCB + ib];
                    v += w * vc[ia * NCB + ib];
                    u += w * uc[ia * NCB + ib];
                }
            }
            const size_t ij = (size_t)(oa + i) * n + ob + j, ji = (size_t)(ob + j) * n + oa + i;
            S[ij] = s; S[ji] = s;
            T[ij] = t; T[ji] = t;
            V[ij] = v; V[ji] = v;
            double uij = u, uji = u;
            if (bv.Hx) { const double* hx = bv.Hx + (size_t)f * n * n; uij += hx[ij]; uji += hx[ji]; }
            H[ij] = t + v + uij; H[ji] = t + v + uji;
            if (bv.U) { double* U = bv.U + (size_t)f * n * n; U[ij] = uij; U[ji] = uji; }
        }
}

template <int LA, int LB>
__global__ void __launch_bounds__(64) int1e_kernel(BatchView bv, const int* __restrict__ pairs, int npairs)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)npairs * bv.nfrag;
    if (tid >= total) return;
    const int ip = (int)(tid / bv.nfrag), f = (int)(tid % bv.nfrag);
    int1e_block<LA, LB>(bv, f, pairs[2 * ip], pairs[2 * ip + 1]);
}

template <int LA, int LB>
static void launch_class(const BatchView& bv, const std::vector<int>& list, int* d, hipStream_t s)
{
    if (list.empty()) return;
    int npairs = (int)list.size() / 2;
    (void)hipMemcpyAsync(d, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice, s);
    long total = (long)npairs * bv.nfrag;
    int blocks = (int)((total + 63) / 64);
    hipLaunchKernelGGL((int1e_kernel<LA, LB>), dim3(blocks), dim3(64), 0, s, bv, d, npairs);
}

// ---------------------------------------------------------------------------------------
// Electronic dipole  tr(D x), tr(D y), tr(D z)  about the coordinate origin, accumulated into bv.dip[f][0..2]
// (system_compute_dipole, mqc_cuest_integrals.f90:1443-1521: mu = sum_A Z_A (R_A - O) - sum_uv D_uv <u|r - O|v>;
// the shift to the centre of nuclear charge O is applied on the host with tr(D S) = N_electrons).
// First moments from the same Hermite tables: <a|x|b> = (E_1^{ij} + P_x E_0^{ij}) sqrt(pi/p).
template <int LA, int LB>
__global__ void __launch_bounds__(64) dipole_kernel(BatchView bv, const int* __restrict__ pairs, int npairs)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)npairs * bv.nfrag;
    if (tid >= total) return;
    const int ip = (int)(tid / bv.nfrag), f = (int)(tid % bv.nfrag);
    const int A = pairs[2 * ip], B = pairs[2 * ip + 1];
    constexpr int NCA = ncart(LA), NCB = ncart(LB);
    const TopologyDev& tp = bv.topo;
    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const int atA = tp.sh_atom[A], atB = tp.sh_atom[B];
    const double ax = xyz[3 * atA], ay = xyz[3 * atA + 1], az = xyz[3 * atA + 2];
    const double bx = xyz[3 * atB], by = xyz[3 * atB + 1], bz = xyz[3 * atB + 2];
    const double ab2 = (ax - bx) * (ax - bx) + (ay - by) * (ay - by) + (az - bz) * (az - bz);
    double mx[NCA * NCB], my[NCA * NCB], mz[NCA * NCB];
#pragma unroll
    for (int i = 0; i < NCA * NCB; ++i) { mx[i] = 0.0; my[i] = 0.0; mz[i] = 0.0; }
    const int npa = tp.sh_nprim[A], npb = tp.sh_nprim[B];
    const double* ea = tp.exps + tp.sh_poff[A]; const double* ca = tp.coefs + tp.sh_poff[A];
    const double* eb = tp.exps + tp.sh_poff[B]; const double* cb = tp.coefs + tp.sh_poff[B];
    for (int ipa = 0; ipa < npa; ++ipa)
        for (int jp = 0; jp < npb; ++jp) {
            const double a = ea[ipa], b = eb[jp], p = a + b, ip_ = 1.0 / p;
            const double kab = exp(-a * b * ip_ * ab2) * ca[ipa] * cb[jp];
            const double px = (a * ax + b * bx) * ip_, py = (a * ay + b * by) * ip_, pz = (a * az + b * bz) * ip_;
            E1D<LA, LB> ex, ey, ez;
            ex.build(px - ax, px - bx, 0.5 * ip_);
            ey.build(py - ay, py - by, 0.5 * ip_);
            ez.build(pz - az, pz - bz, 0.5 * ip_);
            const double s3 = kab * M_PI * ip_ * sqrt(M_PI * ip_);
            int iab = 0;
#pragma unroll
            for (int i0 = LA; i0 >= 0; --i0)
#pragma unroll
                for (int i1 = LA - i0; i1 >= 0; --i1) {
                    const int i2 = LA - i0 - i1;
#pragma unroll
                    for (int j0 = LB; j0 >= 0; --j0)
#pragma unroll
                        for (int j1 = LB - j0; j1 >= 0; --j1) {
                            const int j2 = LB - j0 - j1;
                            const double sx = ex.get(i0, j0, 0), sy = ey.get(i1, j1, 0), sz = ez.get(i2, j2, 0);
                            const double dx = ((i0 + j0 >= 1) ? ex.get(i0, j0, 1) : 0.0) + px * sx;
                            const double dy = ((i1 + j1 >= 1) ? ey.get(i1, j1, 1) : 0.0) + py * sy;
                            const double dz = ((i2 + j2 >= 1) ? ez.get(i2, j2, 1) : 0.0) + pz * sz;
                            mx[iab] += s3 * dx * sy * sz;
                            my[iab] += s3 * sx * dy * sz;
                            mz[iab] += s3 * sx * sy * dz;
                            ++iab;
                        }
                }
        }
    constexpr int NSA = nsph(LA), NSB = nsph(LB);
    const int n = bv.n;
    const double* D = bv.D + (size_t)f * n * n;
    const int oa = tp.sh_aoff[A], ob = tp.sh_aoff[B];
    double tx = 0.0, ty = 0.0, tz = 0.0;
#pragma unroll
    for (int i = 0; i < NSA; ++i)
#pragma unroll
        for (int j = 0; j < NSB; ++j) {
            double vx = 0.0, vy = 0.0, vz = 0.0;
#pragma unroll
            for (int ia = 0; ia < NCA; ++ia) {
                double wa = 1.0;
                if constexpr (LA >= 2) wa = c2s_coef<LA>(bv.c2s, i, ia); else wa = (i == ia) ? 1.0 : 0.0;
                if (wa == 0.0) continue;
#pragma unroll
                for (int ib = 0; ib < NCB; ++ib) {
                    double wb = 1.0;
                    if constexpr (LB >= 2) wb = c2s_coef<LB>(bv.c2s, j, ib); else wb = (j == ib) ? 1.0 : 0.0;
                    const double w = wa * wb;
                    vx += w * mx[ia * NCB + ib]; vy += w * my[ia * NCB + ib]; vz += w * mz[ia * NCB + ib];
                }
            }
            const double d = D[(size_t)(oa + i) * n + ob + j];
            tx += d * vx; ty += d * vy; tz += d * vz;
        }
    const double w = (A == B) ? 1.0 : 2.0;      // the pair list holds A >= B once
    double* dip = bv.dip + (size_t)f * 4;
    atomicAdd(&dip[0], w * tx); atomicAdd(&dip[1], w * ty); atomicAdd(&dip[2], w * tz);
}

template <int LA, int LB>
static void launch_dipole_class(const BatchView& bv, const std::vector<int>& list, int* d, hipStream_t s)
{
    if (list.empty()) return;
    const int npairs = (int)list.size() / 2;
    (void)hipMemcpyAsync(d, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice, s);
    const long total = (long)npairs * bv.nfrag;
    hipLaunchKernelGGL((dipole_kernel<LA, LB>), dim3((int)((total + 63) / 64)), dim3(64), 0, s, bv, d, npairs);
}

void launch_dipole(const BatchView& bv, const Topology& topo, hipStream_t s, bool accumulate)
{
    static DevicePool scratch_slot[2];
    static std::vector<int> bucket_slot[2][LMAX_AO + 1][LMAX_AO + 1];      // kept alive for the async uploads
    auto& bucket = bucket_slot[bv.slot & 1];
    for (auto& row : bucket) for (auto& b : row) b.clear();
    for (size_t k = 0; k + 1 < topo.pairs.size(); k += 2) {
        int A = topo.pairs[k], B = topo.pairs[k + 1];
        int la = topo.shells[A].l, lb = topo.shells[B].l;
        if (la < lb) { std::swap(A, B); std::swap(la, lb); }
        bucket[la][lb].push_back(A);
        bucket[la][lb].push_back(B);
    }
    int* d = (int*)scratch_slot[bv.slot & 1].ensure((topo.pairs.size() + 16) * sizeof(int));
    if (!accumulate) (void)hipMemsetAsync(bv.dip, 0, sizeof(double) * 4 * (size_t)bv.nfrag, s);
    size_t off = 0;
#define DIP_CASE(a, b) launch_dipole_class<a, b>(bv, bucket[a][b], d + off, s); off += bucket[a][b].size();
    DIP_CASE(0, 0) DIP_CASE(1, 0) DIP_CASE(1, 1) DIP_CASE(2, 0) DIP_CASE(2, 1) DIP_CASE(2, 2)
    DIP_CASE(3, 0) DIP_CASE(3, 1) DIP_CASE(3, 2) DIP_CASE(3, 3)
#undef DIP_CASE
}

// Small batches: the class launches of the one-electron stage are latency-bound (0.05-0.26 ms each for one fragment,
// 0.83 ms one after the other -- and the orthogonaliser, the guess and with them the SCF loop wait for the last one):
// they are independent, so they fan out over three more streams and join before the caller's next launch.
namespace {
struct Int1eFan {
    hipStream_t x[3] = {nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[3] = {nullptr, nullptr, nullptr};
    bool ready = false;
};
Int1eFan g_int1e_fan[2];
constexpr int INT1E_FAN_MAX_FRAGMENTS = 16;
}  // namespace

void int1e_reset_state()
{
    for (auto& f : g_int1e_fan) {
        if (!f.ready) continue;
        for (int k = 0; k < 3; ++k) { if (f.x[k]) (void)hipStreamDestroy(f.x[k]); if (f.join[k]) (void)hipEventDestroy(f.join[k]); f.x[k] = nullptr; f.join[k] = nullptr; }
        if (f.fork) (void)hipEventDestroy(f.fork);
        f.fork = nullptr; f.ready = false;
    }
}

void launch_int1e(const BatchView& bv_in, const Topology& topo, hipStream_t s)
{
    static DevicePool scratch_slot[2];
    static DevicePool far_slot[2];
    BatchView bv = bv_in;
    bv.pc_far_r2 = nullptr; bv.pc_far_tab = nullptr;
    static const bool far_on = [] { const char* e = std::getenv("MQC_HIP_PC_FAR"); return !(e && e[0] == '0'); }();
    if (far_on && bv.npc >= PC_FAR_MIN && bv.pc) {
        // per atom: beyond R^2 = 42 / p_min (p_min = twice the most diffuse exponent on the atom) a charge is far for
        // every primitive pair on it
        static std::vector<double> hr2[2];
        std::vector<double>& r2 = hr2[bv.slot & 1];
        r2.assign(topo.natoms, 0.0);
        std::vector<double> amin(topo.natoms, 1.0e300);
        for (const auto& sh : topo.shells)
            for (int k = 0; k < sh.nprim; ++k) amin[sh.atom] = std::min(amin[sh.atom], topo.exps[sh.poff + k]);
        for (int a = 0; a < topo.natoms; ++a) r2[a] = amin[a] < 1.0e299 ? BOYS_TMAX / (2.0 * amin[a]) : 1.0e300;
        const size_t tab_doubles = (size_t)bv.nfrag * topo.natoms * PC_FAR_NT;
        double* base = (double*)far_slot[bv.slot & 1].ensure(sizeof(double) * (tab_doubles + topo.natoms + 32));
        if (base) {
            (void)hipMemcpyAsync(base, r2.data(), sizeof(double) * topo.natoms, hipMemcpyHostToDevice, s);
            double* tab = base + ((topo.natoms + 31) & ~31);
            hipLaunchKernelGGL(pc_far_table_kernel, dim3((bv.nfrag + 63) / 64, topo.natoms), dim3(64), 0, s, bv, base, tab);
            bv.pc_far_r2 = base; bv.pc_far_tab = tab;
        }
    }
    // one device list with an offset per class and host buckets that outlive the asynchronous uploads: no host
    // synchronisation between the class launches (the batch view of the NEXT call reuses them only after the stream
    // has been drained by the SCF loop)
    static std::vector<int> bucket_slot[2][LMAX_AO + 1][LMAX_AO + 1];
    auto& bucket = bucket_slot[bv.slot & 1];
    for (auto& row : bucket) for (auto& b : row) b.clear();
    for (size_t k = 0; k + 1 < topo.pairs.size(); k += 2) {
        int A = topo.pairs[k], B = topo.pairs[k + 1];
        int la = topo.shells[A].l, lb = topo.shells[B].l;
        if (la < lb) { std::swap(A, B); std::swap(la, lb); }
        bucket[la][lb].push_back(A);
        bucket[la][lb].push_back(B);
    }
    int* d = (int*)scratch_slot[bv.slot & 1].ensure((topo.pairs.size() + 16) * sizeof(int));
    size_t off = 0;
    Int1eFan& fan = g_int1e_fan[bv.slot & 1];
    bool fanned = bv.nfrag <= INT1E_FAN_MAX_FRAGMENTS;
    if (fanned && !fan.ready) {
        bool ok = hipEventCreateWithFlags(&fan.fork, hipEventDisableTiming) == hipSuccess;
        for (int k = 0; k < 3 && ok; ++k)
            ok = hipStreamCreateWithFlags(&fan.x[k], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&fan.join[k], hipEventDisableTiming) == hipSuccess;
        fan.ready = ok;
        if (!ok) fanned = false;
    }
    if (fanned) {
        (void)hipEventRecord(fan.fork, s);
        for (int k = 0; k < 3; ++k) (void)hipStreamWaitEvent(fan.x[k], fan.fork, 0);
    }
    // stream of a class: the caller's for (ss|, the others dealt so that the four chains are about as long
    auto st = [&](int k) { return (fanned && k > 0) ? fan.x[k - 1] : s; };
#define I1_CASE(a, b, k) launch_class<a, b>(bv, bucket[a][b], d + off, st(k)); off += bucket[a][b].size();
    I1_CASE(0, 0, 0) I1_CASE(1, 0, 2) I1_CASE(1, 1, 3) I1_CASE(2, 0, 1) I1_CASE(2, 1, 2) I1_CASE(2, 2, 1)
    I1_CASE(3, 0, 3) I1_CASE(3, 1, 3) I1_CASE(3, 2, 2) I1_CASE(3, 3, 1)
#undef I1_CASE
    if (fanned)
        for (int k = 0; k < 3; ++k) {
            (void)hipEventRecord(fan.join[k], fan.x[k]);
            (void)hipStreamWaitEvent(s, fan.join[k], 0);
        }
}

}  // namespace mqc
