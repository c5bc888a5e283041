// kern_eri_general.hip -- wave-cooperative McMurchie-Davidson ERI kernel for ANY angular momentum (l <= 4).
//
// The class kernels of eri_kernels.hpp give one thread a whole contracted quartet, with every table in registers;
// that design stops at d shells (a (dd|dd) thread already needs all 512 registers).  Quartets that involve an f or g
// shell -- def2-TZVP puts one f shell on every heavy atom (BASELINE.json configs[3]); the reference's c2s table
// goes to l = 4 (backends/libcint/mqc_libcint_ao_data.f90:28-125) -- are formed here instead:
//
//   one WAVE per (shell quartet, fragment); run-time angular momenta; every table lives in the wave's LDS:
//     E^{ab}_t, E^{cd}_t   one-dimensional Hermite expansion coefficients, built by six lanes (one per axis and side)
//     R_{tuv}              Hermite Coulomb integrals by the level recursion, all entries of a level in parallel
//     G[tuv][cd]           ket-contracted intermediates, lanes over (bra Hermite index, ket Cartesian component)
//     OUT[ab][cd]          Cartesian accumulators, lanes over components
//   after the primitive loops the four indices are transformed to real solid harmonics (libcint order) in LDS and
//   the block goes to the packed tensor M[pair(ij)][pair(kl)] (both triangles), or -- Schwarz mode -- its largest
//   magnitude becomes Q_AB = sqrt(max |(ab|ab)|) (mqc_libcint_direct.f90:105-153).
//
// Same screening, task lists (block sharing) and counters as the class kernels, so the dispatcher in kern_eri.hip
// treats it as one more launcher.  No scratch memory, 40-odd registers: it also cannot trip the per-queue scratch
// reservation that the d-heavy class kernels can.
#include "eri_kernels.hpp"

namespace mqc {

namespace {

__device__ __forceinline__ int g_hidx(int t, int u, int v) { return hidx(t, u, v); }

// run-time cart -> sph coefficient, same tables as c2s_coef<L>
__device__ __forceinline__ double g_c2s(const double* __restrict__ table, int l, int s, int c)
{
    if (l < 2) return s == c ? 1.0 : 0.0;
    if (l == 2) return c2s_coef<2>(nullptr, s, c);
    return table[c2s_table_offset(l) + s * ncart(l) + c];
}

// one axis of E^{ij}_t, t <= i + j:  e[(i*(lb+1)+j)*(la+lb+1)+t]   (E1D::build with run-time bounds)
__device__ void g_build_e(int la, int lb, double xpa, double xpb, double hp, double* __restrict__ e)
{
    const int nt = la + lb + 1;
    for (int k = 0; k < (la + 1) * (lb + 1) * nt; ++k) e[k] = 0.0;
    e[0] = 1.0;
    for (int i = 0; i <= la; ++i) {
        if (i > 0) {
            const double* prev = e + ((i - 1) * (lb + 1)) * nt;
            double* cur = e + (i * (lb + 1)) * nt;
            for (int t = 0; t <= i; ++t) {
                double v = xpa * prev[t];
                if (t > 0) v += hp * prev[t - 1];
                if (t + 1 <= i - 1) v += (t + 1) * prev[t + 1];
                cur[t] = v;
            }
        }
        for (int j = 1; j <= lb; ++j) {
            const double* prev = e + (i * (lb + 1) + j - 1) * nt;
            double* cur = e + (i * (lb + 1) + j) * nt;
            for (int t = 0; t <= i + j; ++t) {
                double v = xpb * prev[t];
                if (t > 0) v += hp * prev[t - 1];
                if (t + 1 <= i + j - 1) v += (t + 1) * prev[t + 1];
                cur[t] = v;
            }
        }
    }
}

// one index of a block to solid harmonics: in[pre][nc][post] -> out[pre][ns][post], lanes over outputs
__device__ void g_c2s_stage(const double* __restrict__ table, int l, int pre, int post, const double* __restrict__ in,
                            double* __restrict__ out, int lane)
{
    const int nc = ncart(l), ns = nsph(l);
    for (int idx = lane; idx < pre * ns * post; idx += 64) {
        const int a = idx / (ns * post), rem = idx - a * (ns * post);
        const int s = rem / post, r = rem - s * post;
        double v = 0.0;
        for (int c = 0; c < nc; ++c) {
            const double w = g_c2s(table, l, s, c);
            if (w != 0.0) v += w * in[(a * nc + c) * post + r];
        }
        out[idx] = v;
    }
}

struct GenLayout {      // LDS offsets in doubles, computed by the host for the class
    int e_ab, e_cd, r0, r1, tuv, fb, g, out, chunk, total;
};

}  // namespace

enum { GEN_MODE_STORE = 0, GEN_MODE_SCHWARZ = 1, GEN_MODE_DIGEST = 2, GEN_MODE_DF3C = 3 };

// GEN_MODE_DF3C: the three-centre integrals of density fitting for orbital classes the register kernels of kern_df.hip
// do not cover (an f shell in the bra): (ab|P) = (ab|P 1) with the "function" 1 as a one-primitive s shell of exponent 0
// in the fourth slot; list entries are (A, B, P, -), the block goes to df_a3[P][pair(ab)].
__device__ const double g_unit_shell_3c[2] = {0.0, 1.0};

// what the direct (integral-recomputing) Fock build hands the digest mode
struct GenDigest {
    const double* Dmax;     // [nfrag][nshell][nshell] block maxima of |D|
    double* Jt;             // [nfrag][n][n] accumulators (symmetrised afterwards)
    double* Kt;
    int only_active;
};

template <int MODE>
__global__ void __launch_bounds__(64) eri_general_kernel(BatchView bv, int la, int lb, int lc, int ld, GenLayout lay,
                                                         const int* __restrict__ list, int nq,
                                                         const int* __restrict__ tasks, int ntasks,
                                                         const double* __restrict__ Q, double thresh, double* __restrict__ Qout,
                                                         GenDigest dg)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const long t = blockIdx.x;
    const int iq = tasks ? tasks[2 * t] : (int)(t / bv.nfrag), f = tasks ? tasks[2 * t + 1] : (int)(t % bv.nfrag);
    int A, B, C, D;
    if (MODE == GEN_MODE_SCHWARZ) { A = list[2 * iq]; B = list[2 * iq + 1]; C = A; D = B; }
    else { A = list[4 * iq]; B = list[4 * iq + 1]; C = list[4 * iq + 2]; D = list[4 * iq + 3]; }
    const TopologyDev& tp = bv.topo;
    const int ns = tp.nshell;
    if (MODE == GEN_MODE_STORE && Q != nullptr) {
        const double* q = Q + (size_t)f * ns * ns;
        if (!(q[A * ns + B] * q[C * ns + D] >= thresh)) return;        // wave-uniform: the whole wave leaves
    }
    double deg = 1.0;
    if (MODE == GEN_MODE_DIGEST) {
        // screening of build_fock_direct (mqc_libcint_direct.f90:266-288,533-551), wave-uniform
        if (dg.only_active && bv.istate[4 * f] == ST_DONE) return;
        const double sab = (A == B) ? 1.0 : 2.0, scd = (C == D) ? 1.0 : 2.0;
        const bool same = (A == C && B == D) || (A == D && B == C);
        deg = sab * scd * (same ? 1.0 : 2.0);
        const double* q = Q + (size_t)f * ns * ns;
        const double* dm = dg.Dmax + (size_t)f * ns * ns;
        const double dj = 0.5 * fmax(dm[A * ns + B], dm[C * ns + D]);
        const double dk = 0.125 * bv.exx * fmax(fmax(dm[A * ns + C], dm[A * ns + D]), fmax(dm[B * ns + C], dm[B * ns + D]));
        if (!(q[A * ns + B] * q[C * ns + D] * deg * fmax(dj, dk) >= thresh)) return;
    }
    if (MODE != GEN_MODE_SCHWARZ && MODE != GEN_MODE_DF3C && lane == 0 && bv.eri_count) atomicAdd(bv.eri_count, 1ull);

    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const ShellRef sa = make_shell(tp, xyz, A), sb = make_shell(tp, xyz, B);
    const ShellRef sc = MODE == GEN_MODE_DF3C ? make_shell(bv.aux, xyz, C) : make_shell(tp, xyz, C);
    ShellRef sd;
    if (MODE == GEN_MODE_DF3C) { sd.nprim = 1; sd.exps = &g_unit_shell_3c[0]; sd.coefs = &g_unit_shell_3c[1]; sd.x = sc.x; sd.y = sc.y; sd.z = sc.z; }
    else sd = make_shell(tp, xyz, D);
    const int lab = la + lb, lcd = lc + ld, L = lab + lcd;
    const int nca = ncart(la), ncb = ncart(lb), ncc = ncart(lc), ncd = ncart(ld);
    const int ncab = nca * ncb, nccd = ncc * ncd;
    const int nhab = nherm(lab), nhl = nherm(L);
    const int ne_ab = (la + 1) * (lb + 1) * (lab + 1), ne_cd = (lc + 1) * (ld + 1) * (lcd + 1);
    double* Eab = lds + lay.e_ab;       // [3][ne_ab]
    double* Ecd = lds + lay.e_cd;       // [3][ne_cd]
    double* R0 = lds + lay.r0;
    double* R1 = lds + lay.r1;
    int* tuv = (int*)(lds + lay.tuv);   // [nhl] packed (t, u, v)
    double* Fb = lds + lay.fb;          // [L + 1] Boys values of the current primitive quartet
    double* G = lds + lay.g;            // [nhab][chunk]  (later: staging buffer of the solid-harmonic transform)
    double* OUT = lds + lay.out;        // [ncab][nccd]
    const int chunk = lay.chunk;

    // (t,u,v) of every packed Hermite index (lanes take the triples round-robin), accumulators to zero
    {
        int cnt = 0;
        for (int N = 0; N <= L; ++N)
            for (int tt = N; tt >= 0; --tt)
                for (int uu = N - tt; uu >= 0; --uu, ++cnt)
                    if ((cnt & 63) == lane) tuv[g_hidx(tt, uu, N - tt - uu)] = tt | (uu << 8) | ((N - tt - uu) << 16);
    }
    for (int idx = lane; idx < ncab * nccd; idx += 64) OUT[idx] = 0.0;
    __syncthreads();

    const double dabx = sa.x - sb.x, daby = sa.y - sb.y, dabz = sa.z - sb.z;
    const double ab2 = dabx * dabx + daby * daby + dabz * dabz;
    const double dcdx = sc.x - sd.x, dcdy = sc.y - sd.y, dcdz = sc.z - sd.z;
    const double cd2 = dcdx * dcdx + dcdy * dcdy + dcdz * dcdz;
    constexpr double TWO_PI_25 = 34.986836655249725693;

    for (int ip = 0; ip < sa.nprim; ++ip)
        for (int jp = 0; jp < sb.nprim; ++jp) {
            const double a = sa.exps[ip], b = sb.exps[jp], p = a + b, rp = 1.0 / p;
            const double argab = a * b * rp * ab2;
            if (!(argab < PRIM_EXP_CUTOFF)) continue;
            const double kab = exp(-argab) * sa.coefs[ip] * sb.coefs[jp] * rp;
            const double px = (a * sa.x + b * sb.x) * rp, py = (a * sa.y + b * sb.y) * rp, pz = (a * sa.z + b * sb.z) * rp;
            if (lane < 3) {
                const double pa = lane == 0 ? px - sa.x : (lane == 1 ? py - sa.y : pz - sa.z);
                const double pb = lane == 0 ? px - sb.x : (lane == 1 ? py - sb.y : pz - sb.z);
                g_build_e(la, lb, pa, pb, 0.5 * rp, Eab + lane * ne_ab);
            }
            for (int kp = 0; kp < sc.nprim; ++kp)
                for (int lp = 0; lp < sd.nprim; ++lp) {
                    const double c = sc.exps[kp], d = sd.exps[lp], q = c + d, rq = 1.0 / q;
                    const double argcd = c * d * rq * cd2;
                    if (!(argcd < PRIM_EXP_CUTOFF)) continue;
                    const double kcd = exp(-argcd) * sc.coefs[kp] * sd.coefs[lp] * rq;
                    const double qx = (c * sc.x + d * sd.x) * rq, qy = (c * sc.y + d * sd.y) * rq, qz = (c * sc.z + d * sd.z) * rq;
                    __syncthreads();      // the previous primitive quartet has finished with Ecd / R / G
                    if (lane >= 3 && lane < 6) {
                        const int ax = lane - 3;
                        const double qc = ax == 0 ? qx - sc.x : (ax == 1 ? qy - sc.y : qz - sc.z);
                        const double qd = ax == 0 ? qx - sd.x : (ax == 1 ? qy - sd.y : qz - sd.z);
                        g_build_e(lc, ld, qc, qd, 0.5 * rq, Ecd + ax * ne_cd);
                    }
                    const double rs = 1.0 / sqrt(p + q);
                    const double alpha = p * q * rs * rs;
                    const double pref = TWO_PI_25 * rs * kab * kcd;
                    const double X = px - qx, Y = py - qy, Z = pz - qz;
                    // Boys values F_n(T) (-2 alpha)^n into LDS: lane n takes order n (Taylor table), the asymptotic
                    // branch runs its upward recurrence on lane 0
                    {
                        const double T = alpha * (X * X + Y * Y + Z * Z);
                        if (T < BOYS_TMAX) {
                            if (lane <= L) {
                                const int r = (int)(T * (1.0 / BOYS_STEP) + 0.5);
                                const double dt = r * BOYS_STEP - T;
                                const double* cc = bv.boys + r * BOYS_COLS + lane;
                                double acc = cc[7] * (1.0 / 5040.0);
                                acc = acc * dt + cc[6] * (1.0 / 720.0);
                                acc = acc * dt + cc[5] * (1.0 / 120.0);
                                acc = acc * dt + cc[4] * (1.0 / 24.0);
                                acc = acc * dt + cc[3] * (1.0 / 6.0);
                                acc = acc * dt + cc[2] * 0.5;
                                acc = acc * dt + cc[1];
                                acc = acc * dt + cc[0];
                                double sc_ = 1.0;
                                for (int n = 0; n < lane; ++n) sc_ *= -2.0 * alpha;
                                Fb[lane] = acc * sc_;
                            }
                        } else if (lane == 0) {
                            const double inv = 1.0 / T;
                            const double et = exp(-T);
                            double fn = 0.886226925452758014 * sqrt(inv), sc_ = 1.0;
                            for (int n = 0; n <= L; ++n) {
                                Fb[n] = fn * sc_;
                                fn = ((2 * n + 1) * fn - et) * (0.5 * inv);
                                sc_ *= -2.0 * alpha;
                            }
                        }
                    }
                    __syncthreads();
                    // R: level n = L .. 0; level n holds degrees 0 .. L-n (HermiteLevel of md_integrals.hpp, in parallel)
                    double* prev = R0;
                    double* cur = R1;
                    if (lane == 0) prev[0] = Fb[L];
                    for (int n = L - 1; n >= 0; --n) {
                        __syncthreads();
                        const int cnt = nherm(L - n);
                        for (int h = lane; h < cnt; h += 64) {
                            const int pk = tuv[h];
                            const int tt = pk & 0xff, uu = (pk >> 8) & 0xff, vv = pk >> 16;
                            double val;
                            if (h == 0) val = Fb[n];
                            else if (tt > 0) {
                                val = X * prev[g_hidx(tt - 1, uu, vv)];
                                if (tt > 1) val += (tt - 1) * prev[g_hidx(tt - 2, uu, vv)];
                            } else if (uu > 0) {
                                val = Y * prev[g_hidx(tt, uu - 1, vv)];
                                if (uu > 1) val += (uu - 1) * prev[g_hidx(tt, uu - 2, vv)];
                            } else {
                                val = Z * prev[g_hidx(tt, uu, vv - 1)];
                                if (vv > 1) val += (vv - 1) * prev[g_hidx(tt, uu, vv - 2)];
                            }
                            cur[h] = val;
                        }
                        double* sw = prev; prev = cur; cur = sw;
                    }
                    __syncthreads();
                    const double* __restrict__ R = prev;      // level 0
                    const double* __restrict__ Fx = Ecd, * __restrict__ Fy = Ecd + ne_cd, * __restrict__ Fz = Ecd + 2 * ne_cd;
                    const double* __restrict__ Ex = Eab, * __restrict__ Ey = Eab + ne_ab, * __restrict__ Ez = Eab + 2 * ne_ab;
                    const int ntc = lcd + 1, ntb = lab + 1;
                    for (int c0 = 0; c0 < nccd; c0 += chunk) {
                        const int cw = min(chunk, nccd - c0);
                        // G[h][k] = sum_{tau,nu,phi} (-1)^{tau+nu+phi} Ecd R_{t+tau,u+nu,v+phi}
                        for (int idx = lane; idx < nhab * cw; idx += 64) {
                            const int h = idx / cw, k = idx - h * cw;
                            const int icd = c0 + k;
                            const int ic = icd / ncd, id = icd - ic * ncd;
                            int cx, cy, cz, dx, dy, dz;
                            cart_lmn(lc, ic, cx, cy, cz);
                            cart_lmn(ld, id, dx, dy, dz);
                            const int pk = tuv[h];
                            const int t0 = pk & 0xff, u0 = (pk >> 8) & 0xff, v0 = pk >> 16;
                            const double* fx = Fx + (cx * (ld + 1) + dx) * ntc;
                            const double* fy = Fy + (cy * (ld + 1) + dy) * ntc;
                            const double* fz = Fz + (cz * (ld + 1) + dz) * ntc;
                            double g = 0.0;
                            for (int tt = 0; tt <= cx + dx; ++tt)
                                for (int uu = 0; uu <= cy + dy; ++uu) {
                                    const double fxy = fx[tt] * fy[uu];
                                    for (int ww = 0; ww <= cz + dz; ++ww) {
                                        const double term = fxy * fz[ww] * R[g_hidx(t0 + tt, u0 + uu, v0 + ww)];
                                        g += ((tt + uu + ww) & 1) ? -term : term;
                                    }
                                }
                            G[idx] = g;
                        }
                        __syncthreads();
                        for (int idx = lane; idx < ncab * cw; idx += 64) {
                            const int iab = idx / cw, k = idx - iab * cw;
                            const int ia = iab / ncb, ib = iab - ia * ncb;
                            int ax, ay, az, bx, by, bz;
                            cart_lmn(la, ia, ax, ay, az);
                            cart_lmn(lb, ib, bx, by, bz);
                            const double* ex = Ex + (ax * (lb + 1) + bx) * ntb;
                            const double* ey = Ey + (ay * (lb + 1) + by) * ntb;
                            const double* ez = Ez + (az * (lb + 1) + bz) * ntb;
                            double s = 0.0;
                            for (int tt = 0; tt <= ax + bx; ++tt)
                                for (int uu = 0; uu <= ay + by; ++uu) {
                                    const double exy = ex[tt] * ey[uu];
                                    for (int vv = 0; vv <= az + bz; ++vv) s += exy * ez[vv] * G[g_hidx(tt, uu, vv) * cw + k];
                                }
                            OUT[iab * nccd + c0 + k] += pref * s;
                        }
                        __syncthreads();
                    }
                }
            __syncthreads();      // Eab is rebuilt for the next bra primitive pair
        }
    __syncthreads();
    // four indices to real solid harmonics, ping-pong OUT <-> G
    const int nsa = nsph(la), nsb = nsph(lb), nsc = nsph(lc), nsd = nsph(ld);
    g_c2s_stage(bv.c2s, la, 1, ncb * ncc * ncd, OUT, G, lane);
    __syncthreads();
    g_c2s_stage(bv.c2s, lb, nsa, ncc * ncd, G, OUT, lane);
    __syncthreads();
    g_c2s_stage(bv.c2s, lc, nsa * nsb, ncd, OUT, G, lane);
    __syncthreads();
    g_c2s_stage(bv.c2s, ld, nsa * nsb * nsc, 1, G, OUT, lane);
    __syncthreads();
    const int nout = nsa * nsb * nsc * nsd;
    if (MODE == GEN_MODE_SCHWARZ) {
        double m = 0.0;
        for (int idx = lane; idx < nout; idx += 64) m = fmax(m, fabs(OUT[idx]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
        if (lane == 0) {
            double* q = Qout + (size_t)f * ns * ns;
            const double v = sqrt(m);
            q[A * ns + B] = v; q[B * ns + A] = v;
        }
        return;
    }
    if (MODE == GEN_MODE_DF3C) {
        const int oa3 = tp.sh_aoff[A], ob3 = tp.sh_aoff[B], op3 = bv.aux.sh_aoff[C];
        double* A3 = bv.df_a3 + (size_t)f * bv.naux * (size_t)bv.npair;
        for (int idx = lane; idx < nout; idx += 64) {          // nsd = 1
            int r = idx;
            const int k = r % nsc; r /= nsc;
            const int j = r % nsb;
            const int i = r / nsb;
            if (A == B && j > i) continue;
            A3[(size_t)(op3 + k) * bv.npair + pair_index(oa3 + i, ob3 + j)] = OUT[idx];
        }
        return;
    }
    const int oa = tp.sh_aoff[A], ob = tp.sh_aoff[B], oc = tp.sh_aoff[C], od = tp.sh_aoff[D];
    if (MODE == GEN_MODE_DIGEST) {
        // the six pre-contracted scatter updates of eri_digest_kernel, lanes over the output pair of each
        const int n = bv.n;
        const double* __restrict__ Dm = bv.D + (size_t)f * n * n;
        double* __restrict__ J = dg.Jt + (size_t)f * n * n;
        double* __restrict__ K = dg.Kt + (size_t)f * n * n;
        const double wj = 4.0 * deg / 8.0, wk = 2.0 * deg / 8.0;
#define GV(i, j, k, l) OUT[(((i) * nsb + (j)) * nsc + (k)) * nsd + (l)]
        for (int idx = lane; idx < nsa * nsb; idx += 64) {
            const int i = idx / nsb, j = idx - i * nsb;
            double sm = 0.0;
            for (int k = 0; k < nsc; ++k) for (int l = 0; l < nsd; ++l) sm += GV(i, j, k, l) * Dm[(oc + k) * n + od + l];
            atomicAdd(&J[(oa + i) * n + ob + j], wj * sm);
        }
        for (int idx = lane; idx < nsc * nsd; idx += 64) {
            const int k = idx / nsd, l = idx - k * nsd;
            double sm = 0.0;
            for (int i = 0; i < nsa; ++i) for (int j = 0; j < nsb; ++j) sm += GV(i, j, k, l) * Dm[(oa + i) * n + ob + j];
            atomicAdd(&J[(oc + k) * n + od + l], wj * sm);
        }
        if (bv.exx == 0.0) return;      // no exact exchange asked for (pure functionals, Coulomb-only requests): J is all
        for (int idx = lane; idx < nsa * nsc; idx += 64) {
            const int i = idx / nsc, k = idx - i * nsc;
            double sm = 0.0;
            for (int j = 0; j < nsb; ++j) for (int l = 0; l < nsd; ++l) sm += GV(i, j, k, l) * Dm[(ob + j) * n + od + l];
            atomicAdd(&K[(oa + i) * n + oc + k], wk * sm);
        }
        for (int idx = lane; idx < nsa * nsd; idx += 64) {
            const int i = idx / nsd, l = idx - i * nsd;
            double sm = 0.0;
            for (int j = 0; j < nsb; ++j) for (int k = 0; k < nsc; ++k) sm += GV(i, j, k, l) * Dm[(ob + j) * n + oc + k];
            atomicAdd(&K[(oa + i) * n + od + l], wk * sm);
        }
        for (int idx = lane; idx < nsb * nsc; idx += 64) {
            const int j = idx / nsc, k = idx - j * nsc;
            double sm = 0.0;
            for (int i = 0; i < nsa; ++i) for (int l = 0; l < nsd; ++l) sm += GV(i, j, k, l) * Dm[(oa + i) * n + od + l];
            atomicAdd(&K[(ob + j) * n + oc + k], wk * sm);
        }
        for (int idx = lane; idx < nsb * nsd; idx += 64) {
            const int j = idx / nsd, l = idx - j * nsd;
            double sm = 0.0;
            for (int i = 0; i < nsa; ++i) for (int k = 0; k < nsc; ++k) sm += GV(i, j, k, l) * Dm[(oa + i) * n + oc + k];
            atomicAdd(&K[(ob + j) * n + od + l], wk * sm);
        }
#undef GV
        return;
    }
    const size_t np = (size_t)bv.npair;
    const PairStore M = make_pair_store(bv, f);
    for (int idx = lane; idx < nout; idx += 64) {
        int r = idx;
        const int l = r % nsd; r /= nsd;
        const int k = r % nsc; r /= nsc;
        const int j = r % nsb;
        const int i = r / nsb;
        if (A == B && j > i) continue;
        if (C == D && l > k) continue;
        const size_t row = pair_index(oa + i, ob + j), col = pair_index(oc + k, od + l);
        // a diagonal quartet (AB|AB) holds (ij|kl) and (kl|ij) in the same block, equal up to rounding and owned by
        // different lanes: only the lower triangle writes, so that M stays exactly symmetric
        if (A == C && B == D && row < col) continue;
        const double v = OUT[idx];
        M.put(row, col, v);
    }
}

static GenLayout gen_layout(int la, int lb, int lc, int ld)
{
    GenLayout g;
    const int lab = la + lb, lcd = lc + ld, L = lab + lcd;
    const int ne_ab = (la + 1) * (lb + 1) * (lab + 1), ne_cd = (lc + 1) * (ld + 1) * (lcd + 1);
    const int nhab = nherm(lab), nhl = nherm(L);
    const int nca = ncart(la), ncb = ncart(lb), ncc = ncart(lc), ncd = ncart(ld);
    const int nc4 = nca * ncb * ncc * ncd, nccd = ncc * ncd;
    const int s1 = nsph(la) * ncb * ncc * ncd, s3 = nsph(la) * nsph(lb) * nsph(lc) * ncd;
    int chunk = nccd;
    while (chunk > 1 && nhab * chunk > 4096) chunk = (chunk + 1) / 2;
    int gsize = nhab * chunk;
    if (s1 > gsize) gsize = s1;
    if (s3 > gsize) gsize = s3;
    int off = 0;
    auto take = [&off](int n) { const int o = off; off += (n + 1) & ~1; return o; };
    g.e_ab = take(3 * ne_ab); g.e_cd = take(3 * ne_cd);
    g.r0 = take(nhl); g.r1 = take(nhl);
    g.tuv = take((nhl + 1) / 2 + 1);
    g.fb = take(L + 2);
    g.g = take(gsize); g.out = take(nc4);
    g.chunk = chunk; g.total = off;
    return g;
}

// Same contract as launch_eri_class<>: dense (entry x fragment) product or an explicit task list.
bool launch_eri_general(const BatchView& bv, int la, int lb, int lc, int ld, const int* d_list, int nq,
                        const int* d_tasks, int ntasks, const double* Q, double thresh, hipStream_t s)
{
    const long total = d_tasks ? (long)ntasks : (long)nq * bv.nfrag;
    if (nq == 0 || total == 0) return true;
    const GenLayout lay = gen_layout(la, lb, lc, ld);
    const size_t lds = sizeof(double) * (size_t)lay.total;
    if (lds > 160 * 1024 || total > 0x7fffffffL) return false;
    auto kern = eri_general_kernel<GEN_MODE_STORE>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(64), lds, s, bv, la, lb, lc, ld, lay, d_list, nq, d_tasks, ntasks, Q, thresh,
                       (double*)nullptr, GenDigest{});
    return true;
}

// Three-centre integrals (ab|P) of one (la, lb, lp) class into df_a3; d_list holds (A, B, P, -) per entry
bool launch_df3c_general(const BatchView& bv, int la, int lb, int lp, const int* d_list, int nq, hipStream_t s)
{
    const long total = (long)nq * bv.nfrag;
    if (total == 0) return true;
    const GenLayout lay = gen_layout(la, lb, lp, 0);
    const size_t lds = sizeof(double) * (size_t)lay.total;
    if (lds > 160 * 1024 || total > 0x7fffffffL) return false;
    auto kern = eri_general_kernel<GEN_MODE_DF3C>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(64), lds, s, bv, la, lb, lp, 0, lay, d_list, nq, (const int*)nullptr, 0,
                       (const double*)nullptr, 0.0, (double*)nullptr, GenDigest{});
    return true;
}

// Direct Fock build: the quartets of one class digested into J~ / K~ (same contract as launch_eri_digest_class<>)
bool launch_digest_general(const BatchView& bv, int la, int lb, int lc, int ld, const int* d_list, int nq, const double* Q,
                           const double* Dmax, double thresh, double* Jt, double* Kt, int only_active, hipStream_t s)
{
    const long total = (long)nq * bv.nfrag;
    if (total == 0) return true;
    const GenLayout lay = gen_layout(la, lb, lc, ld);
    const size_t lds = sizeof(double) * (size_t)lay.total;
    if (lds > 160 * 1024 || total > 0x7fffffffL) return false;
    auto kern = eri_general_kernel<GEN_MODE_DIGEST>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(64), lds, s, bv, la, lb, lc, ld, lay, d_list, nq, (const int*)nullptr, 0, Q, thresh,
                       (double*)nullptr, GenDigest{Dmax, Jt, Kt, only_active});
    return true;
}

// Schwarz bounds of the shell pairs of one (la, lb) class: d_pairs holds (A, B) with l_A >= l_B
bool launch_schwarz_general(const BatchView& bv, int la, int lb, const int* d_pairs, int npairs, double* Qout, hipStream_t s)
{
    const long total = (long)npairs * bv.nfrag;
    if (total == 0) return true;
    const GenLayout lay = gen_layout(la, lb, la, lb);
    const size_t lds = sizeof(double) * (size_t)lay.total;
    if (lds > 160 * 1024 || total > 0x7fffffffL) return false;
    auto kern = eri_general_kernel<GEN_MODE_SCHWARZ>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(64), lds, s, bv, la, lb, la, lb, lay, d_pairs, npairs, (const int*)nullptr, 0,
                       (const double*)nullptr, 0.0, Qout, GenDigest{});
    return true;
}

}  // namespace mqc
