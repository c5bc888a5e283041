// kern_grad.hip -- analytic nuclear gradient of a converged Hartree-Fock SCF (restricted or unrestricted), exact ERIs.
//
// Terms and signs of compute_scf_gradient (backends/cuest/backend/mqc_cuest_gradient.f90:91-175) and of the CPU path
// the goldens come from (libcint_scf_gradient, backends/libcint/mqc_libcint_gradient.f90:105-329):
//     g = dE_nuc/dR                                            host arithmetic
//       + sum D (dT/dR + dV/dR)                                grad1e_kernel   (basis-function and Hellmann-Feynman parts)
//       - sum W dS/dR,   W = occ * sum_i eps_i C_i C_i^T       grad1e_kernel   (Pulay term, energy-weighted density)
//       + sum_{unique quartets} Gamma . d(ab|cd)/dR            eri_grad_kernel
// with, per unique shell quartet of degeneracy deg (1, 2, 4, 8 as in build_fock_direct),
//     Gamma_ijkl = deg/8 [ 4 Dt_ij Dt_kl - 2 exx sum_spin (Ds_ik Ds_jl + Ds_il Ds_jk) ]     (Ds = D/2 when restricted).
// Nothing here solves a response equation: the SCF energy is stationary in the orbitals.
//
// Kernel design = kern_eri_general.hip's: ONE WAVE per (shell pair | shell quartet, fragment), run-time angular momenta,
// every table in the wave's LDS, lanes over Cartesian components.  A derivative with respect to the centre of a Cartesian
// Gaussian raises and lowers its angular momentum, d/dA_x x_A^i e^{-a r_A^2} = 2a x_A^{i+1} (..) - i x_A^{i-1} (..), so the
// Hermite tables are built one unit higher and the density factor (Gamma, D, W) is brought to the CARTESIAN basis
// once per block; the derivative integrals are then contracted on the fly -- nine numbers per quartet (centres A, B, C;
// D by translational invariance), never stored.  Shells up to f; the two largest f classes pass their ket columns in chunks.
#include "eri_kernels.hpp"

namespace mqc {

namespace {

__device__ __forceinline__ int gr_hidx(int t, int u, int v) { return hidx(t, u, v); }

// index of the Cartesian component (lx, ly, lz) inside shell l (inverse of cart_lmn)
__device__ __forceinline__ int cart_index(int l, int lx, int lz)
{
    const int r = l - lx;
    return r * (r + 1) / 2 + lz;
}

__device__ __forceinline__ double gr_c2s(const double* __restrict__ table, int l, int s, int c)
{
    if (l < 2) return s == c ? 1.0 : 0.0;
    if (l == 2) return c2s_coef<2>(nullptr, s, c);
    return table[c2s_table_offset(l) + s * ncart(l) + c];
}

// one axis of E^{ij}_t for i <= la, j <= lb: e[(i*(lb+1)+j)*(la+lb+1)+t]
__device__ void gr_build_e(int la, int lb, double xpa, double xpb, double hp, double* __restrict__ e)
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

// (t,u,v) table of the packed Hermite indices up to degree L, lanes round-robin
__device__ void gr_fill_tuv(int L, int* tuv, int lane)
{
    int cnt = 0;
    for (int N = 0; N <= L; ++N)
        for (int tt = N; tt >= 0; --tt)
            for (int uu = N - tt; uu >= 0; --uu, ++cnt)
                if ((cnt & 63) == lane) tuv[gr_hidx(tt, uu, N - tt - uu)] = tt | (uu << 8) | ((N - tt - uu) << 16);
}

// Boys values F_n(T) (-2 alpha)^n, n = 0..L, into LDS (lane n takes order n; asymptotic branch on lane 0)
__device__ void gr_boys(const double* __restrict__ table, int L, double alpha, double T, double* Fb, int lane)
{
    if (T < BOYS_TMAX) {
        if (lane <= L) {
            const int r = (int)(T * (1.0 / BOYS_STEP) + 0.5);
            const double dt = r * BOYS_STEP - T;
            const double* cc = table + r * BOYS_COLS + lane;
            double acc = cc[7] * (1.0 / 5040.0);
            acc = acc * dt + cc[6] * (1.0 / 720.0);
            acc = acc * dt + cc[5] * (1.0 / 120.0);
            acc = acc * dt + cc[4] * (1.0 / 24.0);
            acc = acc * dt + cc[3] * (1.0 / 6.0);
            acc = acc * dt + cc[2] * 0.5;
            acc = acc * dt + cc[1];
            acc = acc * dt + cc[0];
            double sc = 1.0;
            for (int n = 0; n < lane; ++n) sc *= -2.0 * alpha;
            Fb[lane] = acc * sc;
        }
    } else if (lane == 0) {
        const double inv = 1.0 / T;
        const double et = exp(-T);
        double fn = 0.886226925452758014 * sqrt(inv), sc = 1.0;
        for (int n = 0; n <= L; ++n) {
            Fb[n] = fn * sc;
            fn = ((2 * n + 1) * fn - et) * (0.5 * inv);
            sc *= -2.0 * alpha;
        }
    }
}

// R_{tuv} for t+u+v <= L by the level recursion; returns the buffer that holds level 0 (call with all lanes, Fb ready)
__device__ const double* gr_hermite_r(int L, double X, double Y, double Z, const double* Fb, const int* tuv, double* R0, double* R1, int lane)
{
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
                val = X * prev[gr_hidx(tt - 1, uu, vv)];
                if (tt > 1) val += (tt - 1) * prev[gr_hidx(tt - 2, uu, vv)];
            } else if (uu > 0) {
                val = Y * prev[gr_hidx(tt, uu - 1, vv)];
                if (uu > 1) val += (uu - 1) * prev[gr_hidx(tt, uu - 2, vv)];
            } else {
                val = Z * prev[gr_hidx(tt, uu, vv - 1)];
                if (vv > 1) val += (vv - 1) * prev[gr_hidx(tt, uu, vv - 2)];
            }
            cur[h] = val;
        }
        double* sw = prev; prev = cur; cur = sw;
    }
    __syncthreads();
    return prev;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace

// ---------------------------------------------------------------------------------------
// W = occ sum_{i occupied} eps_i C_i C_i^T  (energy_weighted_density, mqc_cuest_gradient.f90:58-88); thread per element
__global__ void grad_weighted_density_kernel(BatchView bv, double* __restrict__ Wout)
{
    const int f = blockIdx.y, n = bv.n;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    const int i = idx / n, j = idx - i * n;
    const size_t nn = (size_t)n * n;
    double s = 0.0;
    const double* C = bv.C + f * nn; const double* eps = bv.eps + (size_t)f * n;
    if (bv.uhf) {
        for (int o = 0; o < bv.nalpha; ++o) s += eps[o] * C[i * n + o] * C[j * n + o];
        const double* Cb = bv.Cb + f * nn; const double* eb = bv.epsb + (size_t)f * n;
        for (int o = 0; o < bv.nbeta; ++o) s += eb[o] * Cb[i * n + o] * Cb[j * n + o];
    } else {
        for (int o = 0; o < bv.nocc; ++o) s += eps[o] * C[i * n + o] * C[j * n + o];
        s *= 2.0;
    }
    Wout[f * nn + idx] = s;
}

__global__ void grad_total_density_kernel(BatchView bv, double* __restrict__ Dtot)
{
    const size_t total = (size_t)bv.nfrag * bv.n * bv.n;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) Dtot[idx] = bv.D[idx] + bv.Db[idx];
}

// ---------------------------------------------------------------------------------------
// One-electron terms: wave = (shell pair A >= B, fragment).  Dt = total density, W = energy-weighted density.
//   g_A += w sum_ab [ Dt_ab (dT_ab/dA + dV_ab/dA) - W_ab dS_ab/dA ],   g_B -= (the S and T parts), g_B += w Dt dV/dB,
//   g_C += w Dt_ab dV_ab/dC for every nucleus C,   dV/dB = -(dV/dA + dV/dC) per nucleus;   w = 1 (A == B) or 2.
struct Grad1eLayout { int e, r0, r1, tuv, fb, dc, wc, total; };

__global__ void __launch_bounds__(64) grad1e_kernel(BatchView bv, int la, int lb, Grad1eLayout lay, const int* __restrict__ pairs, int npairs,
                                                    const double* __restrict__ Dtot, const double* __restrict__ Wmat,
                                                    double* __restrict__ grad /* [nfrag][natoms][3] */)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const long t = blockIdx.x;
    const int ip_ = (int)(t / bv.nfrag), f = (int)(t % bv.nfrag);
    const int A = pairs[2 * ip_], B = pairs[2 * ip_ + 1];
    const TopologyDev& tp = bv.topo;
    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const ShellRef sa = make_shell(tp, xyz, A), sb = make_shell(tp, xyz, B);
    const int atA = tp.sh_atom[A], atB = tp.sh_atom[B];
    const int nca = ncart(la), ncb = ncart(lb), nsa = nsph(la), nsb = nsph(lb);
    const int LA1 = la + 1, LB2 = lb + 2;               // table bounds: bra raised once, ket twice (kinetic)
    const int nt = LA1 + LB2 + 1;
    const int ne = (LA1 + 1) * (LB2 + 1) * nt;
    const int LR = la + lb + 2;                          // Hermite order: raised bra + Hellmann-Feynman shift
    double* E = lds + lay.e;                             // [3][ne]
    double* R0 = lds + lay.r0; double* R1 = lds + lay.r1;
    int* tuv = (int*)(lds + lay.tuv);
    double* Fb = lds + lay.fb;
    double* Dc = lds + lay.dc;                           // [nca][ncb] Cartesian density block
    double* Wc = lds + lay.wc;
    const int n = bv.n;
    const size_t nn = (size_t)n * n;
    const double* Dm = Dtot + f * nn; const double* Wm = Wmat + f * nn;
    const int oa = tp.sh_aoff[A], ob = tp.sh_aoff[B];

    gr_fill_tuv(LR, tuv, lane);
    // density blocks to the Cartesian basis: Dc[ia][ib] = sum_ij c2s_a[i][ia] D[i][j] c2s_b[j][ib]
    for (int idx = lane; idx < nca * ncb; idx += 64) {
        const int ia = idx / ncb, ib = idx - ia * ncb;
        double d = 0.0, w = 0.0;
        for (int i = 0; i < nsa; ++i) {
            const double ca = gr_c2s(bv.c2s, la, i, ia);
            if (ca == 0.0) continue;
            for (int j = 0; j < nsb; ++j) {
                const double cb = gr_c2s(bv.c2s, lb, j, ib);
                if (cb == 0.0) continue;
                d += ca * cb * Dm[(size_t)(oa + i) * n + ob + j];
                w += ca * cb * Wm[(size_t)(oa + i) * n + ob + j];
            }
        }
        Dc[idx] = d; Wc[idx] = w;
    }
    __syncthreads();
    const double wsym = (A == B) ? 1.0 : 2.0;
    const double dabx = sa.x - sb.x, daby = sa.y - sb.y, dabz = sa.z - sb.z;
    const double ab2 = dabx * dabx + daby * daby + dabz * dabz;
    double gA[3] = {0.0, 0.0, 0.0}, gB[3] = {0.0, 0.0, 0.0};      // per-lane partial sums
    double* gf = grad + (size_t)f * tp.natoms * 3;

    for (int ipa = 0; ipa < sa.nprim; ++ipa)
        for (int jpb = 0; jpb < sb.nprim; ++jpb) {
            const double a = sa.exps[ipa], b = sb.exps[jpb], p = a + b, rp = 1.0 / p;
            const double arg = a * b * rp * ab2;
            if (!(arg < PRIM_EXP_CUTOFF)) continue;
            const double kab = exp(-arg) * sa.coefs[ipa] * sb.coefs[jpb];
            const double px = (a * sa.x + b * sb.x) * rp, py = (a * sa.y + b * sb.y) * rp, pz = (a * sa.z + b * sb.z) * rp;
            __syncthreads();
            if (lane < 3) {
                const double pa = lane == 0 ? px - sa.x : (lane == 1 ? py - sa.y : pz - sa.z);
                const double pb = lane == 0 ? px - sb.x : (lane == 1 ? py - sb.y : pz - sb.z);
                gr_build_e(LA1, LB2, pa, pb, 0.5 * rp, E + lane * ne);
            }
            __syncthreads();
            const double* Ex = E; const double* Ey = E + ne; const double* Ez = E + 2 * ne;
            auto e0 = [&](const double* Eax, int i, int j) { return (i < 0 || j < 0) ? 0.0 : Eax[(i * (LB2 + 1) + j) * nt]; };
            // 1-D kinetic factor t(i,j) = -2b(2j+1) s(i,j) + 4b^2 s(i,j+2) + j(j-1) s(i,j-2)
            auto t1 = [&](const double* Eax, int i, int j) {
                if (i < 0) return 0.0;
                double v = -2.0 * b * (2 * j + 1) * e0(Eax, i, j) + 4.0 * b * b * e0(Eax, i, j + 2);
                if (j >= 2) v += j * (j - 1) * e0(Eax, i, j - 2);
                return v;
            };
            const double s3 = kab * M_PI * rp * sqrt(M_PI * rp);
            // ---- overlap and kinetic derivatives with respect to A
            for (int idx = lane; idx < nca * ncb; idx += 64) {
                const int ia = idx / ncb, ib = idx - ia * ncb;
                int ax, ay, az, bx, by, bz;
                cart_lmn(la, ia, ax, ay, az);
                cart_lmn(lb, ib, bx, by, bz);
                const double sx = e0(Ex, ax, bx), sy = e0(Ey, ay, by), sz = e0(Ez, az, bz);
                const double tx = t1(Ex, ax, bx), ty = t1(Ey, ay, by), tz = t1(Ez, az, bz);
                // d/dA_x: f(ax) -> 2a f(ax+1) - ax f(ax-1) on the x factors only
                const double dsx = 2.0 * a * e0(Ex, ax + 1, bx) - ax * e0(Ex, ax - 1, bx);
                const double dsy = 2.0 * a * e0(Ey, ay + 1, by) - ay * e0(Ey, ay - 1, by);
                const double dsz = 2.0 * a * e0(Ez, az + 1, bz) - az * e0(Ez, az - 1, bz);
                const double dtx = 2.0 * a * t1(Ex, ax + 1, bx) - ax * t1(Ex, ax - 1, bx);
                const double dty = 2.0 * a * t1(Ey, ay + 1, by) - ay * t1(Ey, ay - 1, by);
                const double dtz = 2.0 * a * t1(Ez, az + 1, bz) - az * t1(Ez, az - 1, bz);
                const double dS[3] = {dsx * sy * sz, sx * dsy * sz, sx * sy * dsz};
                const double dT[3] = {-0.5 * (dtx * sy * sz + dsx * ty * sz + dsx * sy * tz),
                                      -0.5 * (tx * dsy * sz + sx * dty * sz + sx * dsy * tz),
                                      -0.5 * (tx * sy * dsz + sx * ty * dsz + sx * sy * dtz)};
                const double dcw = Dc[idx], wcw = Wc[idx];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double v = wsym * s3 * (dcw * dT[c] - wcw * dS[c]);
                    gA[c] += v; gB[c] -= v;              // two-centre integrals: d/dB = -d/dA
                }
            }
            // ---- nuclear attraction: basis-function part (A), operator part (nucleus C), B by invariance
            const double pref0 = 2.0 * M_PI * rp * kab;
            for (int at = 0; at < tp.natoms; ++at) {
                const double zq = tp.zeff[at];
                if (zq == 0.0) continue;
                const double X = px - xyz[3 * at], Y = py - xyz[3 * at + 1], Z = pz - xyz[3 * at + 2];
                __syncthreads();
                gr_boys(bv.boys, LR, p, p * (X * X + Y * Y + Z * Z), Fb, lane);
                __syncthreads();
                const double* R = gr_hermite_r(LR, X, Y, Z, Fb, tuv, R0, R1, lane);
                double vA[3] = {0.0, 0.0, 0.0}, vC[3] = {0.0, 0.0, 0.0};
                for (int idx = lane; idx < nca * ncb; idx += 64) {
                    const int ia = idx / ncb, ib = idx - ia * ncb;
                    int ax, ay, az, bx, by, bz;
                    cart_lmn(la, ia, ax, ay, az);
                    cart_lmn(lb, ib, bx, by, bz);
                    // V(a', b; shift) = sum_tuv Ex[a'x][bx][t] Ey[..][u] Ez[..][v] R[t+sx, u+sy, v+sz]
                    auto vint = [&](int a0, int a1, int a2, int s0, int s1, int s2) {
                        if (a0 < 0 || a1 < 0 || a2 < 0) return 0.0;
                        const double* ex = Ex + (a0 * (LB2 + 1) + bx) * nt;
                        const double* ey = Ey + (a1 * (LB2 + 1) + by) * nt;
                        const double* ez = Ez + (a2 * (LB2 + 1) + bz) * nt;
                        double s = 0.0;
                        for (int tt = 0; tt <= a0 + bx; ++tt)
                            for (int uu = 0; uu <= a1 + by; ++uu) {
                                const double exy = ex[tt] * ey[uu];
                                for (int vv = 0; vv <= a2 + bz; ++vv) s += exy * ez[vv] * R[gr_hidx(tt + s0, uu + s1, vv + s2)];
                            }
                        return s;
                    };
                    const double dcw = wsym * Dc[idx] * (-zq) * pref0;
                    // basis function on A moves
                    vA[0] += dcw * (2.0 * a * vint(ax + 1, ay, az, 0, 0, 0) - ax * vint(ax - 1, ay, az, 0, 0, 0));
                    vA[1] += dcw * (2.0 * a * vint(ax, ay + 1, az, 0, 0, 0) - ay * vint(ax, ay - 1, az, 0, 0, 0));
                    vA[2] += dcw * (2.0 * a * vint(ax, ay, az + 1, 0, 0, 0) - az * vint(ax, ay, az - 1, 0, 0, 0));
                    // the nucleus moves: d/dC_x R_tuv(P - C) = -R_{t+1,u,v}
                    vC[0] -= dcw * vint(ax, ay, az, 1, 0, 0);
                    vC[1] -= dcw * vint(ax, ay, az, 0, 1, 0);
                    vC[2] -= dcw * vint(ax, ay, az, 0, 0, 1);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    gA[c] += vA[c];
                    gB[c] -= vA[c] + vC[c];              // translational invariance of the three-centre integral
                    const double tot = wave_sum(vC[c]);
                    if (lane == 0 && tot != 0.0) atomicAdd(&gf[3 * at + c], tot);
                }
            }
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double ta = wave_sum(gA[c]), tb = wave_sum(gB[c]);
        if (lane == 0) {
            if (ta != 0.0) atomicAdd(&gf[3 * atA + c], ta);
            if (tb != 0.0) atomicAdd(&gf[3 * atB + c], tb);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Two-electron term: wave = (unique shell quartet, fragment).
struct Grad2eLayout { int e_ab, e_cd, r0, r1, tuv, fb, gc, tmp, g0, gp, gm, total, dchunk; };

// The "function" 1 as a one-primitive s shell: exponent 0, coefficient 1.  With it in the fourth (or second and fourth)
// slot the quartet routine below differentiates the three- and two-centre Coulomb integrals of density fitting:
// (ab|P) = (ab|P 1), (P|Q) = (P 1|Q 1); the unit shell sits on its partner's atom, so the force the quartet routine
// hands it by translational invariance lands where it belongs.
__device__ const double g_unit_shell[2] = {0.0, 1.0};
__device__ __forceinline__ ShellRef unit_shell_at(const ShellRef& at)
{
    ShellRef r;
    r.nprim = 1; r.exps = &g_unit_shell[0]; r.coefs = &g_unit_shell[1];
    r.x = at.x; r.y = at.y; r.z = at.z;
    return r;
}

// mode GRAD_EXACT: list = unique orbital quartets, Gamma from the densities (header of this file).
// mode GRAD_DF3C:  list = (A >= B orbital, P auxiliary, -): Gamma = w gam3[P][ab], w = 1 (A == B) or 2.
// mode GRAD_DF2C:  list = (P >= Q auxiliary as A and C):     Gamma = w gam2[P][Q],  w = 1 (P == Q) or 2.
enum : int { GRAD_EXACT = 0, GRAD_DF3C = 1, GRAD_DF2C = 2 };

__global__ void __launch_bounds__(64) eri_grad_kernel(BatchView bv, int la, int lb, int lc, int ld, Grad2eLayout lay,
                                                      const int* __restrict__ list, int nq, const double* __restrict__ Dtot,
                                                      const double* __restrict__ Dbeta /* nullptr: restricted */,
                                                      double* __restrict__ grad, int mode = GRAD_EXACT,
                                                      const double* __restrict__ gam = nullptr /* [nfrag][naux][n*n] or [nfrag][naux][naux] */)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const long t = blockIdx.x;
    const int iq = (int)(t / bv.nfrag), f = (int)(t % bv.nfrag);
    const int A = list[4 * iq], B = list[4 * iq + 1], C = list[4 * iq + 2], D = list[4 * iq + 3];
    const TopologyDev& tp = bv.topo;
    const TopologyDev& tx = bv.aux;
    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const ShellRef sa = mode == GRAD_DF2C ? make_shell(tx, xyz, A) : make_shell(tp, xyz, A);
    const ShellRef sc = mode == GRAD_EXACT ? make_shell(tp, xyz, C) : make_shell(tx, xyz, C);
    const ShellRef sb = mode == GRAD_DF2C ? unit_shell_at(sa) : make_shell(tp, xyz, B);
    const ShellRef sd = mode == GRAD_EXACT ? make_shell(tp, xyz, D) : unit_shell_at(sc);
    const int atA = mode == GRAD_DF2C ? tx.sh_atom[A] : tp.sh_atom[A];
    const int atC = mode == GRAD_EXACT ? tp.sh_atom[C] : tx.sh_atom[C];
    const int atB = mode == GRAD_DF2C ? atA : tp.sh_atom[B];
    const int atD = mode == GRAD_EXACT ? tp.sh_atom[D] : atC;
    if (atA == atB && atB == atC && atC == atD) return;          // one centre: translational invariance, no force
    const int nca = ncart(la), ncb = ncart(lb), ncc = ncart(lc), ncd = ncart(ld);
    const int nsa = nsph(la), nsb = nsph(lb), nsc = nsph(lc), nsd = nsph(ld);
    const int ncab = nca * ncb, nccd = ncc * ncd, nc4 = ncab * nccd;
    // tables one unit higher on A, B and C
    const int LA1 = la + 1, LB1 = lb + 1, LC1 = lc + 1;
    const int ntb = LA1 + LB1 + 1, ntc = LC1 + ld + 1;
    const int ne_ab = (LA1 + 1) * (LB1 + 1) * ntb, ne_cd = (LC1 + 1) * (ld + 1) * ntc;
    const int lab = la + lb, lcd = lc + ld;
    const int LR = lab + lcd + 1;
    const int nh1 = nherm(lab + 1), nh0 = nherm(lab);
    const int ncp = ncart(lc + 1) * ncd, ncm = (lc > 0 ? ncart(lc - 1) : 0) * ncd;
    double* Eab = lds + lay.e_ab; double* Ecd = lds + lay.e_cd;
    double* R0 = lds + lay.r0; double* R1 = lds + lay.r1;
    int* tuv = (int*)(lds + lay.tuv);
    double* Fb = lds + lay.fb;
    double* GC = lds + lay.gc;          // [nc4] Gamma in the Cartesian basis
    double* TMP = lds + lay.tmp;        // staging of the transform
    double* G0 = lds + lay.g0;          // [nh1][nccd]  ket (lc, ld)
    double* GP = lds + lay.gp;          // [nh0][ncp]   ket (lc+1, ld)
    double* GM = lds + lay.gm;          // [nh0][ncm]   ket (lc-1, ld)
    const int n = bv.n;
    const size_t nn = (size_t)n * n;
    const double* Dt = Dtot + f * nn;
    const double* Db = Dbeta ? Dbeta + f * nn : nullptr;
    const int oa = mode == GRAD_DF2C ? tx.sh_aoff[A] : tp.sh_aoff[A], ob = mode == GRAD_DF2C ? 0 : tp.sh_aoff[B];
    const int oc = mode == GRAD_EXACT ? tp.sh_aoff[C] : tx.sh_aoff[C], od = mode == GRAD_EXACT ? tp.sh_aoff[D] : 0;

    gr_fill_tuv(LR, tuv, lane);
    // ---- Gamma block in the spherical basis (into TMP), then four index transforms to Cartesian (-> GC)
    if (mode != GRAD_EXACT) {
        const int na = bv.naux;
        const double w = (mode == GRAD_DF3C ? A == B : A == C) ? 1.0 : 2.0;
        const double* g3 = gam + (size_t)f * na * (mode == GRAD_DF3C ? nn : (size_t)na);
        for (int idx = lane; idx < nsa * nsb * nsc; idx += 64) {          // nsd = 1
            int r = idx;
            const int k = r % nsc; r /= nsc;
            const int j = r % nsb;
            const int i = r / nsb;
            GC[idx] = w * (mode == GRAD_DF3C ? g3[(size_t)(oc + k) * nn + (size_t)(oa + i) * n + ob + j]
                                             : g3[(size_t)(oa + i) * na + oc + k]);
        }
        __syncthreads();
    } else {
        const double sab = (A == B) ? 1.0 : 2.0, scd = (C == D) ? 1.0 : 2.0;
        const bool same = (A == C && B == D) || (A == D && B == C);
        const double w = sab * scd * (same ? 1.0 : 2.0) / 8.0;
        const double exx = bv.exx;
        for (int idx = lane; idx < nsa * nsb * nsc * nsd; idx += 64) {
            int r = idx;
            const int l = r % nsd; r /= nsd;
            const int k = r % nsc; r /= nsc;
            const int j = r % nsb;
            const int i = r / nsb;
            const int I = oa + i, J = ob + j, K = oc + k, Lx = od + l;
            double g = 4.0 * Dt[(size_t)I * n + J] * Dt[(size_t)K * n + Lx];
            if (Db) {
                const double aik = Dt[(size_t)I * n + K] - Db[(size_t)I * n + K], ajl = Dt[(size_t)J * n + Lx] - Db[(size_t)J * n + Lx];
                const double ail = Dt[(size_t)I * n + Lx] - Db[(size_t)I * n + Lx], ajk = Dt[(size_t)J * n + K] - Db[(size_t)J * n + K];
                g -= 2.0 * exx * (aik * ajl + ail * ajk + Db[(size_t)I * n + K] * Db[(size_t)J * n + Lx] + Db[(size_t)I * n + Lx] * Db[(size_t)J * n + K]);
            } else {
                g -= exx * (Dt[(size_t)I * n + K] * Dt[(size_t)J * n + Lx] + Dt[(size_t)I * n + Lx] * Dt[(size_t)J * n + K]);
            }
            GC[idx] = w * g;
        }
        __syncthreads();
    }
    {
        // sph -> cart on one index: out[pre][nc][post] = sum_s coef(s, c) in[pre][ns][post]
        auto stage = [&](int l, int pre, int post, const double* in, double* out) {
            const int nc = ncart(l), ns = nsph(l);
            for (int idx = lane; idx < pre * nc * post; idx += 64) {
                const int a = idx / (nc * post), rem = idx - a * (nc * post);
                const int c = rem / post, r = rem - c * post;
                double v = 0.0;
                for (int s = 0; s < ns; ++s) {
                    const double w2 = gr_c2s(bv.c2s, l, s, c);
                    if (w2 != 0.0) v += w2 * in[(a * ns + s) * post + r];
                }
                out[idx] = v;
            }
            __syncthreads();
        };
        stage(ld, nsa * nsb * nsc, 1, GC, TMP);
        stage(lc, nsa * nsb, ncd, TMP, GC);
        stage(lb, nsa, ncc * ncd, GC, TMP);
        stage(la, 1, ncb * ncc * ncd, TMP, GC);
    }
    const double dabx = sa.x - sb.x, daby = sa.y - sb.y, dabz = sa.z - sb.z;
    const double ab2 = dabx * dabx + daby * daby + dabz * dabz;
    const double dcdx = sc.x - sd.x, dcdy = sc.y - sd.y, dcdz = sc.z - sd.z;
    const double cd2 = dcdx * dcdx + dcdy * dcdy + dcdz * dcdz;
    constexpr double TWO_PI_25 = 34.986836655249725693;
    double gA[3] = {0.0, 0.0, 0.0}, gB[3] = {0.0, 0.0, 0.0}, gCc[3] = {0.0, 0.0, 0.0};

    for (int ip = 0; ip < sa.nprim; ++ip)
        for (int jp = 0; jp < sb.nprim; ++jp) {
            const double a = sa.exps[ip], b = sb.exps[jp], p = a + b, rp = 1.0 / p;
            const double argab = a * b * rp * ab2;
            if (!(argab < PRIM_EXP_CUTOFF)) continue;
            const double kab = exp(-argab) * sa.coefs[ip] * sb.coefs[jp] * rp;
            const double px = (a * sa.x + b * sb.x) * rp, py = (a * sa.y + b * sb.y) * rp, pz = (a * sa.z + b * sb.z) * rp;
            __syncthreads();
            if (lane < 3) {
                const double pa = lane == 0 ? px - sa.x : (lane == 1 ? py - sa.y : pz - sa.z);
                const double pb = lane == 0 ? px - sb.x : (lane == 1 ? py - sb.y : pz - sb.z);
                gr_build_e(LA1, LB1, pa, pb, 0.5 * rp, Eab + lane * ne_ab);
            }
            for (int kp = 0; kp < sc.nprim; ++kp)
                for (int lp = 0; lp < sd.nprim; ++lp) {
                    const double c = sc.exps[kp], d = sd.exps[lp], q = c + d, rq = 1.0 / q;
                    const double argcd = c * d * rq * cd2;
                    if (!(argcd < PRIM_EXP_CUTOFF)) continue;
                    const double kcd = exp(-argcd) * sc.coefs[kp] * sd.coefs[lp] * rq;
                    const double qx = (c * sc.x + d * sd.x) * rq, qy = (c * sc.y + d * sd.y) * rq, qz = (c * sc.z + d * sd.z) * rq;
                    __syncthreads();
                    if (lane >= 3 && lane < 6) {
                        const int ax = lane - 3;
                        const double qc = ax == 0 ? qx - sc.x : (ax == 1 ? qy - sc.y : qz - sc.z);
                        const double qd = ax == 0 ? qx - sd.x : (ax == 1 ? qy - sd.y : qz - sd.z);
                        gr_build_e(LC1, ld, qc, qd, 0.5 * rq, Ecd + ax * ne_cd);
                    }
                    const double rs = 1.0 / sqrt(p + q);
                    const double alpha = p * q * rs * rs;
                    const double pref = TWO_PI_25 * rs * kab * kcd;
                    const double X = px - qx, Y = py - qy, Z = pz - qz;
                    gr_boys(bv.boys, LR, alpha, alpha * (X * X + Y * Y + Z * Z), Fb, lane);
                    __syncthreads();
                    const double* R = gr_hermite_r(LR, X, Y, Z, Fb, tuv, R0, R1, lane);
                    const double* Fx = Ecd; const double* Fy = Ecd + ne_cd; const double* Fz = Ecd + 2 * ne_cd;
                    // ket-contracted intermediates: G[h][k] = sum (-1)^{tau+nu+phi} F F F R_{t+tau,u+nu,v+phi}
                    // (f f|f d) and (f f|f f) do not fit the LDS with every ket column at once: the columns go through
                    // in chunks of dn Cartesian components of shell D (dn = ncd everywhere else: one pass, as before)
                    const int dnmax = lay.dchunk;
                    for (int d0 = 0; d0 < ncd; d0 += dnmax) {
                    const int dn = min(dnmax, ncd - d0);
                    const int kcd = ncc * dn, kcp = ncart(lc + 1) * dn, kcm = (lc > 0 ? ncart(lc - 1) : 0) * dn;
                    auto ket = [&](int lcx, int nh, int nk, double* G) {
                        for (int idx = lane; idx < nh * nk; idx += 64) {
                            const int h = idx / nk, k = idx - h * nk;
                            const int ic = k / dn, id = d0 + (k - ic * dn);
                            int cx, cy, cz, dx, dy, dz;
                            cart_lmn(lcx, ic, cx, cy, cz);
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
                                        const double term = fxy * fz[ww] * R[gr_hidx(t0 + tt, u0 + uu, v0 + ww)];
                                        g += ((tt + uu + ww) & 1) ? -term : term;
                                    }
                                }
                            G[idx] = g;
                        }
                    };
                    ket(lc, nh1, kcd, G0);
                    ket(lc + 1, nh0, kcp, GP);
                    if (lc > 0) ket(lc - 1, nh0, kcm, GM);
                    __syncthreads();
                    const double* Ex = Eab; const double* Ey = Eab + ne_ab; const double* Ez = Eab + 2 * ne_ab;
                    // bra contraction: sum_tuv E[a'][b'] G[hidx(t,u,v)][col], stride = columns of that G
                    auto bra = [&](int a0, int a1, int a2, int b0, int b1, int b2, const double* G, int stride, int col) {
                        if (a0 < 0 || a1 < 0 || a2 < 0 || b0 < 0 || b1 < 0 || b2 < 0) return 0.0;
                        const double* ex = Ex + (a0 * (LB1 + 1) + b0) * ntb;
                        const double* ey = Ey + (a1 * (LB1 + 1) + b1) * ntb;
                        const double* ez = Ez + (a2 * (LB1 + 1) + b2) * ntb;
                        double s = 0.0;
                        for (int tt = 0; tt <= a0 + b0; ++tt)
                            for (int uu = 0; uu <= a1 + b1; ++uu) {
                                const double exy = ex[tt] * ey[uu];
                                for (int vv = 0; vv <= a2 + b2; ++vv) s += exy * ez[vv] * G[gr_hidx(tt, uu, vv) * stride + col];
                            }
                        return s;
                    };
                    for (int idx = lane; idx < nc4; idx += 64) {
                        const double gam = GC[idx];
                        if (gam == 0.0) continue;
                        const int iab = idx / nccd, icd = idx - iab * nccd;
                        const int ia = iab / ncb, ib = iab - ia * ncb;
                        const int ic = icd / ncd, idf = icd - ic * ncd;
                        if (idf < d0 || idf >= d0 + dn) continue;
                        const int id = idf - d0, kc = ic * dn + id;
                        int ax, ay, az, bx, by, bz, cx, cy, cz;
                        cart_lmn(la, ia, ax, ay, az);
                        cart_lmn(lb, ib, bx, by, bz);
                        cart_lmn(lc, ic, cx, cy, cz);
                        const double pg = pref * gam;
                        // centre A
                        gA[0] += pg * (2.0 * a * bra(ax + 1, ay, az, bx, by, bz, G0, kcd, kc) - ax * bra(ax - 1, ay, az, bx, by, bz, G0, kcd, kc));
                        gA[1] += pg * (2.0 * a * bra(ax, ay + 1, az, bx, by, bz, G0, kcd, kc) - ay * bra(ax, ay - 1, az, bx, by, bz, G0, kcd, kc));
                        gA[2] += pg * (2.0 * a * bra(ax, ay, az + 1, bx, by, bz, G0, kcd, kc) - az * bra(ax, ay, az - 1, bx, by, bz, G0, kcd, kc));
                        // centre B
                        gB[0] += pg * (2.0 * b * bra(ax, ay, az, bx + 1, by, bz, G0, kcd, kc) - bx * bra(ax, ay, az, bx - 1, by, bz, G0, kcd, kc));
                        gB[1] += pg * (2.0 * b * bra(ax, ay, az, bx, by + 1, bz, G0, kcd, kc) - by * bra(ax, ay, az, bx, by - 1, bz, G0, kcd, kc));
                        gB[2] += pg * (2.0 * b * bra(ax, ay, az, bx, by, bz + 1, G0, kcd, kc) - bz * bra(ax, ay, az, bx, by, bz - 1, G0, kcd, kc));
                        // centre C: ket raised / lowered on c
                        {
                            const int kxp = cart_index(lc + 1, cx + 1, cz) * dn + id;
                            const int kyp = cart_index(lc + 1, cx, cz) * dn + id;
                            const int kzp = cart_index(lc + 1, cx, cz + 1) * dn + id;
                            double vx = 2.0 * c * bra(ax, ay, az, bx, by, bz, GP, kcp, kxp);
                            double vy = 2.0 * c * bra(ax, ay, az, bx, by, bz, GP, kcp, kyp);
                            double vz = 2.0 * c * bra(ax, ay, az, bx, by, bz, GP, kcp, kzp);
                            if (cx > 0) vx -= cx * bra(ax, ay, az, bx, by, bz, GM, kcm, cart_index(lc - 1, cx - 1, cz) * dn + id);
                            if (cy > 0) vy -= cy * bra(ax, ay, az, bx, by, bz, GM, kcm, cart_index(lc - 1, cx, cz) * dn + id);
                            if (cz > 0) vz -= cz * bra(ax, ay, az, bx, by, bz, GM, kcm, cart_index(lc - 1, cx, cz - 1) * dn + id);
                            gCc[0] += pg * vx; gCc[1] += pg * vy; gCc[2] += pg * vz;
                        }
                    }
                    if (d0 + dn < ncd) __syncthreads();      // the next chunk overwrites G0 / GP / GM
                    }
                }
            __syncthreads();
        }
    double* gf = grad + (size_t)f * tp.natoms * 3;
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
        const double ta = wave_sum(gA[cc]), tb = wave_sum(gB[cc]), tc = wave_sum(gCc[cc]);
        if (lane == 0) {
            atomicAdd(&gf[3 * atA + cc], ta);
            atomicAdd(&gf[3 * atB + cc], tb);
            atomicAdd(&gf[3 * atC + cc], tc);
            atomicAdd(&gf[3 * atD + cc], -(ta + tb + tc));      // translational invariance
        }
    }
}


// ---------------------------------------------------------------------------------------
// Density-fitted two-electron gradient (the reference GPU backend's own gradient is density-fitted:
// compute_scf_gradient, backends/cuest/backend/mqc_cuest_gradient.f90:91-175; the CPU formulation it is checked
// against is df_gradient of backends/libcint/mqc_libcint_gradient.f90).  With the fit B = W A3 (W = L^{-1} or
// J^{-1/2}; W^T W = J^{-1}), fitted coefficients C^P = sum_Q W_QP B^Q and c = W^T (B . D):
//     E_2e' = sum_{P, mu nu} Gam^P_{mu nu} (mu nu|P)'  -  1/2 sum_{PQ} gam_PQ (P|Q)'
//     Gam^P = Dt c_P - x Z^P,      Z^P = sum_Q W_QP Y^Q,   Y^Q = sum_spin D_s B^Q D_s
//     gam   = c c^T  - x sum_{mu nu} C^P_{mu nu} Z^Q_{mu nu}
// x = exx (unrestricted, spin densities) or exx / 2 with D_s := D (restricted: two spins of D/2 each).
// Parity-first arithmetic on the vector units: a gradient is one evaluation per geometry, not per SCF iteration.
__device__ __forceinline__ double dfg_b(const double* __restrict__ Bq, int i, int j)
{
    return i >= j ? Bq[i * (i + 1) / 2 + j] : Bq[j * (j + 1) / 2 + i];
}

// c_P = sum_Q W[Q][P] g_Q,  g_Q = sum_pairs B[Q][pair] (2 - delta) Dt[pair]; one workgroup per fragment
__global__ void __launch_bounds__(256) dfg_coef_kernel(BatchView bv, const double* __restrict__ Dtot, double* __restrict__ cvec)
{
    extern __shared__ double g[];            // [na]
    const int f = blockIdx.x, n = bv.n, na = bv.naux, np = bv.npair, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const double* __restrict__ Dt = Dtot + (size_t)f * n * n;
    const double* __restrict__ B = bv.df_b + (size_t)f * na * (size_t)np;
    const double* __restrict__ W = bv.df_linv + (size_t)f * na * na;
    const bool full = bv.scal[(size_t)f * 8 + 7] == 2.0;
    for (int q = wave; q < na; q += 4) {
        double sum = 0.0;
        for (int pr = lane; pr < np; pr += 64) {
            int k = (int)((sqrt(8.0 * pr + 1.0) - 1.0) * 0.5);
            while ((k + 1) * (k + 2) / 2 <= pr) ++k;
            while (k * (k + 1) / 2 > pr) --k;
            const int l = pr - k * (k + 1) / 2;
            sum += B[(size_t)q * np + pr] * (k == l ? 1.0 : 2.0) * Dt[(size_t)k * n + l];
        }
        sum = wave_sum(sum);
        if (lane == 0) g[q] = sum;
    }
    __syncthreads();
    for (int p = tid; p < na; p += 256) {
        double sum = 0.0;
        for (int q = full ? 0 : p; q < na; ++q) sum += W[(size_t)q * na + p] * g[q];
        cvec[(size_t)f * na + p] = sum;
    }
}

// T^Q = B^Q D_s (pass 0) and Y^Q (+)= D_s T^Q (pass 1); thread per matrix element, grid (n*n/256, na, nfrag)
__global__ void __launch_bounds__(256) dfg_half_kernel(BatchView bv, const double* __restrict__ Ds, double* __restrict__ T)
{
    const int f = blockIdx.z, q = blockIdx.y, n = bv.n, na = bv.naux, np = bv.npair;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const int i = idx / n, j = idx - i * n;
    const double* __restrict__ Bq = bv.df_b + ((size_t)f * na + q) * (size_t)np;
    const double* __restrict__ D = Ds + (size_t)f * n * n;
    double sum = 0.0;
    for (int k = 0; k < n; ++k) sum += dfg_b(Bq, i, k) * D[(size_t)k * n + j];
    T[((size_t)f * na + q) * n * n + idx] = sum;
}
__global__ void __launch_bounds__(256) dfg_full_kernel(BatchView bv, const double* __restrict__ Ds, const double* __restrict__ T,
                                                       double* __restrict__ Y, int accumulate)
{
    const int f = blockIdx.z, q = blockIdx.y, n = bv.n, na = bv.naux;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const int i = idx / n, j = idx - i * n;
    const double* __restrict__ D = Ds + (size_t)f * n * n;
    const double* __restrict__ Tq = T + ((size_t)f * na + q) * n * n;
    double sum = 0.0;
    for (int k = 0; k < n; ++k) sum += D[(size_t)i * n + k] * Tq[(size_t)k * n + j];
    double* y = Y + ((size_t)f * na + q) * n * n + idx;
    *y = accumulate ? *y + sum : sum;
}

// Z^P = sum_Q W[Q][P] Y^Q;  Gam^P = Dt c_P - x Z^P.  grid (n*n/256, na, nfrag)
__global__ void __launch_bounds__(256) dfg_gamma_kernel(BatchView bv, const double* __restrict__ Dtot, const double* __restrict__ cvec,
                                                        const double* __restrict__ Y, double x, double* __restrict__ Z,
                                                        double* __restrict__ Gam)
{
    const int f = blockIdx.z, p = blockIdx.y, n = bv.n, na = bv.naux;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const double* __restrict__ W = bv.df_linv + (size_t)f * na * na;
    const bool full = bv.scal[(size_t)f * 8 + 7] == 2.0;
    const size_t nn = (size_t)n * n;
    double z = 0.0;
    for (int q = full ? 0 : p; q < na; ++q) z += W[(size_t)q * na + p] * Y[((size_t)f * na + q) * nn + idx];
    Z[((size_t)f * na + p) * nn + idx] = z;
    Gam[((size_t)f * na + p) * nn + idx] = Dtot[(size_t)f * nn + idx] * cvec[(size_t)f * na + p] - x * z;
}

// gam[P][Q] = c_P c_Q - x sum_{mu nu} C^P_{mu nu} Z^Q_{mu nu},  C^P = sum_R W[R][P] B^R; one wave per (P, Q), grid (na*na, nfrag)
__global__ void __launch_bounds__(64) dfg_gamma2_kernel(BatchView bv, const double* __restrict__ cvec, const double* __restrict__ Cfit,
                                                        const double* __restrict__ Z, double x, double* __restrict__ gam2)
{
    const int f = blockIdx.y, n = bv.n, na = bv.naux, np = bv.npair, lane = threadIdx.x;
    const int p = blockIdx.x / na, q = blockIdx.x - p * na;
    const double* __restrict__ Cp = Cfit + ((size_t)f * na + p) * (size_t)np;
    const double* __restrict__ Zq = Z + ((size_t)f * na + q) * (size_t)n * n;
    double sum = 0.0;
    for (int idx = lane; idx < n * n; idx += 64) {
        const int i = idx / n, j = idx - i * n;
        sum += dfg_b(Cp, i, j) * Zq[idx];
    }
    sum = wave_sum(sum);
    if (lane == 0) gam2[((size_t)f * na + p) * na + q] = -0.5 * (cvec[(size_t)f * na + p] * cvec[(size_t)f * na + q] - x * sum);
}

// C^P[pair] = sum_R W[R][P] B^R[pair]; grid (npair/256, na, nfrag)
__global__ void __launch_bounds__(256) dfg_cfit_kernel(BatchView bv, double* __restrict__ Cfit)
{
    const int f = blockIdx.z, p = blockIdx.y, na = bv.naux, np = bv.npair;
    const int pr = blockIdx.x * 256 + threadIdx.x;
    if (pr >= np) return;
    const double* __restrict__ W = bv.df_linv + (size_t)f * na * na;
    const double* __restrict__ B = bv.df_b + (size_t)f * na * (size_t)np;
    const bool full = bv.scal[(size_t)f * 8 + 7] == 2.0;
    double sum = 0.0;
    for (int r = full ? 0 : p; r < na; ++r) sum += W[(size_t)r * na + p] * B[(size_t)r * np + pr];
    Cfit[((size_t)f * na + p) * (size_t)np + pr] = sum;
}

// ---------------------------------------------------------------------------------------
static Grad1eLayout grad1e_layout(int la, int lb)
{
    Grad1eLayout g;
    const int LA1 = la + 1, LB2 = lb + 2, nt = LA1 + LB2 + 1, ne = (LA1 + 1) * (LB2 + 1) * nt, LR = la + lb + 2;
    int off = 0;
    auto take = [&off](int n) { const int o = off; off += (n + 1) & ~1; return o; };
    g.e = take(3 * ne); g.r0 = take(nherm(LR)); g.r1 = take(nherm(LR)); g.tuv = take((nherm(LR) + 1) / 2 + 1);
    g.fb = take(LR + 2); g.dc = take(ncart(la) * ncart(lb)); g.wc = take(ncart(la) * ncart(lb));
    g.total = off;
    return g;
}

static Grad2eLayout grad2e_layout(int la, int lb, int lc, int ld)
{
    Grad2eLayout g;
    const int LA1 = la + 1, LB1 = lb + 1, LC1 = lc + 1;
    const int ne_ab = (LA1 + 1) * (LB1 + 1) * (LA1 + LB1 + 1), ne_cd = (LC1 + 1) * (ld + 1) * (LC1 + ld + 1);
    const int LR = la + lb + lc + ld + 1;
    const int nc4 = ncart(la) * ncart(lb) * ncart(lc) * ncart(ld);
    // staging buffer of the sph -> cart transform: the largest intermediate
    int tmp = nsph(la) * nsph(lb) * nsph(lc) * ncart(ld);
    const int t3 = nsph(la) * ncart(lb) * ncart(lc) * ncart(ld);
    if (t3 > tmp) tmp = t3;
    int off = 0;
    auto take = [&off](int n) { const int o = off; off += (n + 1) & ~1; return o; };
    g.e_ab = take(3 * ne_ab); g.e_cd = take(3 * ne_cd);
    g.r0 = take(nherm(LR)); g.r1 = take(nherm(LR)); g.tuv = take((nherm(LR) + 1) / 2 + 1); g.fb = take(LR + 2);
    int gcsz = nc4;
    const int t2 = nsph(la) * nsph(lb) * ncart(lc) * ncart(ld), t0 = nsph(la) * nsph(lb) * nsph(lc) * nsph(ld);
    if (t2 > gcsz) gcsz = t2;
    if (t0 > gcsz) gcsz = t0;
    g.gc = take(gcsz); g.tmp = take(tmp);
    g.dchunk = ncart(ld);
    const int before = off;
    auto take_g = [&](int dn) {
        g.g0 = take(nherm(la + lb + 1) * ncart(lc) * dn);
        g.gp = take(nherm(la + lb) * ncart(lc + 1) * dn);
        g.gm = take(lc > 0 ? nherm(la + lb) * ncart(lc - 1) * dn : 2);
    };
    take_g(g.dchunk);
    if (sizeof(double) * (size_t)off > 160 * 1024) {
        // (f f|f d), (f f|f f): one Cartesian component of shell D per pass, and the intermediates take over the
        // staging buffer of the solid-harmonic transform (finished before the primitive loops start)
        g.dchunk = 1;
        off = g.tmp;
        take_g(1);
        if (off < before) off = before;
    }
    g.total = off;
    return g;
}

// Gradient of the whole batch into d_grad [nfrag][natoms][3] (zeroed here); Dtot = D (restricted) or D_a + D_b.
// work: device scratch of at least 2 * nfrag * n * n doubles (energy-weighted density, total density of a UHF run).
bool launch_xc_gradient(const BatchView& bv, double* d_grad, hipStream_t s, std::string& err);      // kern_xc.hip

static bool launch_df_gradient(const BatchView& bv, const Topology& topo, const Topology& aux, const double* Dtot, double* d_grad,
                               hipStream_t s, std::string& err)
{
    const int n = bv.n, nf = bv.nfrag, na = bv.naux, np = bv.npair;
    const size_t nn = (size_t)n * n;
    if (aux.lmax > 3) { err = "density-fitted gradients cover auxiliary shells up to f"; return false; }
    static DevicePool pool_slot[2], list_slot[2];
    // [c | T | Y | Z | Gam | Cfit | gam2]
    const size_t big = (size_t)nf * na * nn;
    const size_t doubles = (size_t)nf * na + 4 * big + (size_t)nf * na * np + (size_t)nf * na * na + 64;
    double* w = (double*)pool_slot[bv.slot & 1].ensure(sizeof(double) * doubles);
    if (!w) { err = "out of device memory (density-fitted gradient)"; return false; }
    double* cvec = w; double* T = cvec + (((size_t)nf * na + 7) & ~size_t(7));
    double* Y = T + big; double* Z = Y + big; double* Gam = Z + big; double* Cfit = Gam + big; double* gam2 = Cfit + (size_t)nf * na * np;
    const double x = bv.uhf ? bv.exx : 0.5 * bv.exx;
    hipLaunchKernelGGL(dfg_coef_kernel, dim3(nf), dim3(256), sizeof(double) * (size_t)(na + 8), s, bv, Dtot, cvec);
    const dim3 ge((unsigned)((nn + 255) / 256), na, nf);
    if (bv.exx != 0.0) {
        hipLaunchKernelGGL(dfg_half_kernel, ge, dim3(256), 0, s, bv, (const double*)bv.D, T);
        hipLaunchKernelGGL(dfg_full_kernel, ge, dim3(256), 0, s, bv, (const double*)bv.D, (const double*)T, Y, 0);
        if (bv.uhf) {
            hipLaunchKernelGGL(dfg_half_kernel, ge, dim3(256), 0, s, bv, (const double*)bv.Db, T);
            hipLaunchKernelGGL(dfg_full_kernel, ge, dim3(256), 0, s, bv, (const double*)bv.Db, (const double*)T, Y, 1);
        }
    } else {
        (void)hipMemsetAsync(Y, 0, sizeof(double) * big, s);
    }
    hipLaunchKernelGGL(dfg_gamma_kernel, ge, dim3(256), 0, s, bv, Dtot, (const double*)cvec, (const double*)Y, x, Z, Gam);
    hipLaunchKernelGGL(dfg_cfit_kernel, dim3((np + 255) / 256, na, nf), dim3(256), 0, s, bv, Cfit);
    hipLaunchKernelGGL(dfg_gamma2_kernel, dim3(na * na, nf), dim3(64), 0, s, bv, (const double*)cvec, (const double*)Cfit, (const double*)Z, x, gam2);
    // ---- task lists: (A >= B, P) by (la, lb, lp); (P >= Q) by (lp, lq)
    const int ns = (int)topo.shells.size(), nx = (int)aux.shells.size();
    std::vector<int> t3[4][4][4], t2[4][4];
    for (int A = 0; A < ns; ++A)
        for (int B = 0; B <= A; ++B)
            for (int P = 0; P < nx; ++P) {
                auto& v = t3[topo.shells[A].l][topo.shells[B].l][aux.shells[P].l];
                v.push_back(A); v.push_back(B); v.push_back(P); v.push_back(0);
            }
    for (int P = 0; P < nx; ++P)
        for (int Q = 0; Q <= P; ++Q) {
            auto& v = t2[aux.shells[P].l][aux.shells[Q].l];
            v.push_back(P); v.push_back(0); v.push_back(Q); v.push_back(0);
        }
    size_t tot = 0;
    for (auto& a : t3) for (auto& b : a) for (auto& c : b) tot += c.size();
    for (auto& a : t2) for (auto& b : a) tot += b.size();
    int* d = (int*)list_slot[bv.slot & 1].ensure(sizeof(int) * (tot + 64));
    if (!d) { err = "out of device memory (density-fitted gradient lists)"; return false; }
    size_t off = 0;
    auto run = [&](std::vector<int>& v, int la, int lb, int lc, int mode, const double* g) -> bool {
        if (v.empty()) return true;
        (void)hipMemcpyAsync(d + off, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, s);
        const Grad2eLayout lay = grad2e_layout(la, lb, lc, 0);
        if (sizeof(double) * (size_t)lay.total > 160 * 1024) { err = "gradient: class too large for LDS"; return false; }
        const long total = (long)(v.size() / 4) * nf;
        hipLaunchKernelGGL(eri_grad_kernel, dim3((unsigned)total), dim3(64), sizeof(double) * (size_t)lay.total, s, bv, la, lb, lc, 0, lay,
                           d + off, (int)(v.size() / 4), (const double*)nullptr, (const double*)nullptr, d_grad, mode, g);
        off += v.size();
        return true;
    };
    for (int la = 0; la < 4; ++la)
        for (int lb = 0; lb < 4; ++lb)
            for (int lp = 0; lp < 4; ++lp)
                if (!run(t3[la][lb][lp], la, lb, lp, GRAD_DF3C, Gam)) return false;
    for (int lp = 0; lp < 4; ++lp)
        for (int lq = 0; lq < 4; ++lq)
            if (!run(t2[lp][lq], lp, 0, lq, GRAD_DF2C, gam2)) return false;
    // the host vectors above die with this frame while their copies may still be in flight
    (void)hipStreamSynchronize(s);
    return true;
}

bool launch_gradient(const BatchView& bv, const Topology& topo, const Topology* aux, double* d_grad, double* work, int* d_lists,
                     size_t list_capacity_ints, hipStream_t s, std::string& err)
{
    const int n = bv.n, nf = bv.nfrag;
    const size_t nn = (size_t)n * n;
    if (topo.lmax > 3) { err = "analytic gradients cover s, p, d and f shells"; return false; }
    double* Wm = work;
    double* Dtot = bv.D;
    (void)hipMemsetAsync(d_grad, 0, sizeof(double) * (size_t)nf * topo.natoms * 3, s);
    hipLaunchKernelGGL(grad_weighted_density_kernel, dim3((unsigned)((nn + 255) / 256), nf), dim3(256), 0, s, bv, Wm);
    if (bv.uhf) {
        Dtot = work + (size_t)nf * nn;
        const size_t tot = (size_t)nf * nn;
        hipLaunchKernelGGL(grad_total_density_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, bv, Dtot);
    }
    // ---- one-electron: pairs bucketed by (la >= lb)
    static std::vector<int> bucket[2][4][4];
    auto& bk = bucket[bv.slot & 1];
    for (auto& row : bk) for (auto& b : row) b.clear();
    for (size_t k = 0; k + 1 < topo.pairs.size(); k += 2) {
        int A = topo.pairs[k], B = topo.pairs[k + 1];
        int la = topo.shells[A].l, lb = topo.shells[B].l;
        if (la < lb) { std::swap(A, B); std::swap(la, lb); }
        bk[la][lb].push_back(A); bk[la][lb].push_back(B);
    }
    size_t need = topo.pairs.size();
    if (bv.naux == 0) for (auto& cl : topo.classes) need += cl.quartets.size();
    if (need + 64 > list_capacity_ints) { err = "gradient: list buffer too small"; return false; }
    size_t off = 0;
    (void)hipFuncSetAttribute((const void*)grad1e_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)eri_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int la = 0; la <= 3; ++la)
        for (int lb = 0; lb <= la; ++lb) {
            auto& v = bk[la][lb];
            if (v.empty()) continue;
            (void)hipMemcpyAsync(d_lists + off, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, s);
            const Grad1eLayout lay = grad1e_layout(la, lb);
            const long total = (long)(v.size() / 2) * nf;
            hipLaunchKernelGGL(grad1e_kernel, dim3((unsigned)total), dim3(64), sizeof(double) * (size_t)lay.total, s, bv, la, lb, lay,
                               d_lists + off, (int)(v.size() / 2), Dtot, Wm, d_grad);
            off += v.size();
        }
    // ---- two-electron: density-fitted, or every canonical class of the exact integrals
    if (bv.naux > 0) {
        if (!aux) { err = "gradient: the auxiliary basis is missing"; return false; }
        if (!launch_df_gradient(bv, topo, *aux, Dtot, d_grad, s, err)) return false;
    } else
    for (auto& cl : topo.classes) {
        if (cl.quartets.empty()) continue;
        (void)hipMemcpyAsync(d_lists + off, cl.quartets.data(), cl.quartets.size() * sizeof(int), hipMemcpyHostToDevice, s);
        const Grad2eLayout lay = grad2e_layout(cl.la, cl.lb, cl.lc, cl.ld);
        if (sizeof(double) * (size_t)lay.total > 160 * 1024) { err = "gradient: class too large for LDS"; return false; }
        const long total = (long)(cl.quartets.size() / 4) * nf;
        hipLaunchKernelGGL(eri_grad_kernel, dim3((unsigned)total), dim3(64), sizeof(double) * (size_t)lay.total, s, bv, cl.la, cl.lb, cl.lc, cl.ld, lay,
                           d_lists + off, (int)(cl.quartets.size() / 4), Dtot, bv.uhf ? bv.Db : (const double*)nullptr, d_grad);
        off += cl.quartets.size();
    }
    // ---- exchange-correlation (Kohn-Sham): moving functions, moving points, moving partition
    if (bv.xc.ncomp > 0 && !launch_xc_gradient(bv, d_grad, s, err)) return false;
    return true;
}

}  // namespace mqc
