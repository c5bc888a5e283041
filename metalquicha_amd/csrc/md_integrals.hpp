// md_integrals.hpp -- McMurchie-Davidson integral arithmetic for the gfx950 kernels.
//
// Everything here is a `__host__ __device__` inline template so that (a) the HIP kernels
// in kern_eri.hip / kern_int1e.hip instantiate it per angular-momentum class with all loop
// bounds known at compile time (arrays become VGPRs after full unrolling), and (b) the
// test-only harness tests/host/check_device_math.cpp can run the very same arithmetic on the
// CPU against the oracle.  libmqc_hip.so itself contains no host execution path for these.
//
// Conventions restated from libcint (the reference's integral dependency, see
// backends/libcint/mqc_libcint_ao_data.f90:28-125 and mqc_libcint_ao.f90:17-19,166-193):
//   Cartesian component order of shell l:  x^(l-i) y^(i-j) z^j, i = 0..l, j = 0..i
//   real solid harmonics r^l Y_lm, m = -l..l, except p which is (x, y, z)
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace mqc {

#define MQC_HD __host__ __device__ __forceinline__

// 1/x and 1/sqrt(x) from the hardware seed plus two Newton steps (<= 1 ulp for the normal-range
// arguments met here: exponent sums, T >= 42).  The IEEE-exact expansions cost 12 and ~25 instructions
// per call and sat in the innermost primitive loop; the host build keeps the plain forms.
MQC_HD double fast_rcp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}
MQC_HD double fast_rsqrt(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    y = fma(y, fma(-hx * y, y, 0.5), y);
    return y;
#else
    return 1.0 / sqrt(x);
#endif
}

// Makes a register value opaque to the optimiser at this point (no instruction is emitted).  Used on
// the bra Hermite tables inside the ket primitive loops: without it LLVM's loop-invariant code motion
// hoists every ex*ey*ez product out of those loops and keeps ~100 extra doubles alive, which is what
// pushed the d classes into scratch memory.
#if defined(__HIP_DEVICE_COMPILE__)
#define MQC_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define MQC_OPAQUE(x) ((void)0)
#endif

constexpr int LMAX_AO = 4;                     // g functions, as the reference's c2s table
constexpr int BOYS_MAX_ORDER = 4 * LMAX_AO;    // highest Boys order an ERI can ask for
constexpr int BOYS_TAYLOR = 8;                 // terms k = 0..7 of the Taylor expansion
constexpr int BOYS_COLS = BOYS_MAX_ORDER + BOYS_TAYLOR;   // table columns F_0..F_{max+7}
constexpr double BOYS_STEP = 0.1;
constexpr double BOYS_TMAX = 42.0;
constexpr int BOYS_ROWS = 421;                 // T0 = 0, 0.1, ..., 42.0

MQC_HD constexpr int ncart(int l) { return (l + 1) * (l + 2) / 2; }
MQC_HD constexpr int nsph(int l) { return 2 * l + 1; }
MQC_HD constexpr int nherm(int L) { return (L + 1) * (L + 2) * (L + 3) / 6; }

// index of (t,u,v) inside the tetrahedron t+u+v <= L, ordered by total degree N, then t, then u
MQC_HD constexpr int hidx(int t, int u, int v)
{
    int N = t + u + v;
    int base = N * (N + 1) * (N + 2) / 6;     // entries of degree < N
    // within degree N: t descending from N..0, u descending
    int tt = N - t;                             // 0..N
    return base + tt * (tt + 1) / 2 + (tt - u); // u runs N-t .. 0
}

// ---------------------------------------------------------------------------------------
// Boys function F_0..F_L(T) from the pre-tabulated Taylor grid (table built on the host by
// boys_table.cpp with a convergent series; row r holds F_0..F_{BOYS_COLS-1}(r*BOYS_STEP)).
template <int L>
MQC_HD void boys(double T, const double* __restrict__ table, double* F)
{
    if (T < BOYS_TMAX) {
        int r = (int)(T * (1.0 / BOYS_STEP) + 0.5);
        double dt = r * BOYS_STEP - T;            // |dt| <= 0.05
        const double* row = table + r * BOYS_COLS;
        // F_n(T) = sum_k F_{n+k}(T0) dt^k / k!
        auto taylor = [&](int n) {
            const double* c = row + n;
            double acc = c[7] * (1.0 / 5040.0);
            acc = acc * dt + c[6] * (1.0 / 720.0);
            acc = acc * dt + c[5] * (1.0 / 120.0);
            acc = acc * dt + c[4] * (1.0 / 24.0);
            acc = acc * dt + c[3] * (1.0 / 6.0);
            acc = acc * dt + c[2] * 0.5;
            acc = acc * dt + c[1];
            acc = acc * dt + c[0];
            return acc;
        };
        F[L] = taylor(L);
        if constexpr (L > 0 && L <= 2) {
            // low orders: a second/third Taylor sum over the same table row is cheaper than exp(-T)
#pragma unroll
            for (int n = L - 1; n >= 0; --n) F[n] = taylor(n);
        } else if constexpr (L > 2) {
            double et = exp(-T);
#pragma unroll
            for (int n = L; n > 0; --n) F[n - 1] = (2.0 * T * F[n] + et) * (1.0 / (2 * n - 1));
        }
    } else {
        // asymptotic form; exp(-T) < 6e-19 is below one ulp of F_0..F_2 and only kept for higher orders
        const double rs = fast_rsqrt(T);
        const double inv = rs * rs;
        F[0] = 0.886226925452758014 * rs;   // sqrt(pi)/2 / sqrt(T); erfc(sqrt(42)) ~ 1e-20
        double et = 0.0;
        if constexpr (L > 2) et = exp(-T);
#pragma unroll
        for (int n = 0; n < L; ++n) F[n + 1] = ((2 * n + 1) * F[n] - et) * (0.5 * inv);
    }
}

// ---------------------------------------------------------------------------------------
// Hermite Coulomb integrals R_{tuv}(alpha, PQ) for t+u+v <= L, packed with hidx().
// Level recursion R^(n)_{t+1,u,v} = t R^(n+1)_{t-1,u,v} + X R^(n+1)_{t,u,v}: level n holds all (tuv)
// of degree <= L-n.  Written as a template recursion over n so that EVERY array index is a
// compile-time constant (a run-time-bounded copy loop here kept the work arrays in scratch memory).
template <int L, int N>
struct HermiteLevel {
    // out: R^(N), degrees 0..L-N
    MQC_HD static void run(const double* F, double X, double Y, double Z, double* out)
    {
        if constexpr (N == L) {
            out[0] = F[L];
        } else {
            double prev[nherm(L - N - 1)];
            HermiteLevel<L, N + 1>::run(F, X, Y, Z, prev);
            out[0] = F[N];
            constexpr int deg = L - N;
#pragma unroll
            for (int D = 1; D <= deg; ++D) {
#pragma unroll
                for (int t = D; t >= 0; --t) {
#pragma unroll
                    for (int u = D - t; u >= 0; --u) {
                        const int v = D - t - u;
                        double val;
                        if (t > 0) {
                            val = X * prev[hidx(t - 1, u, v)];
                            if (t > 1) val += (t - 1) * prev[hidx(t - 2, u, v)];
                        } else if (u > 0) {
                            val = Y * prev[hidx(t, u - 1, v)];
                            if (u > 1) val += (u - 1) * prev[hidx(t, u - 2, v)];
                        } else {
                            val = Z * prev[hidx(t, u, v - 1)];
                            if (v > 1) val += (v - 1) * prev[hidx(t, u, v - 2)];
                        }
                        out[hidx(t, u, v)] = val;
                    }
                }
            }
        }
    }
};

template <int L>
MQC_HD void hermite_r(double alpha, double X, double Y, double Z, const double* __restrict__ table, double* R, double scale = 1.0)
{
    double F[L + 1];
    boys<L>(alpha * (X * X + Y * Y + Z * Z), table, F);
    // scale: R^n_000 = (-2 alpha)^n F_n (times the caller's prefactor, which then rides through the whole recursion)
    double s = scale;
#pragma unroll
    for (int n = 0; n <= L; ++n) { F[n] *= s; s *= -2.0 * alpha; }
    HermiteLevel<L, 0>::run(F, X, Y, Z, R);
}

// ---------------------------------------------------------------------------------------
// 1-D Hermite expansion coefficients with E^{00}_0 = 1 (the Gaussian product factor is kept
// as a scalar by the caller).  E[(i*(LB+1)+j)*(LA+LB+1)+t]
template <int LA, int LB>
struct E1D {
    static constexpr int NT = LA + LB + 1;
    double e[(LA + 1) * (LB + 1) * NT];
    MQC_HD double& at(int i, int j, int t) { return e[(i * (LB + 1) + j) * NT + t]; }
    MQC_HD double get(int i, int j, int t) const { return e[(i * (LB + 1) + j) * NT + t]; }
    MQC_HD void pin()
    {
#pragma unroll
        for (int i = 0; i <= LA; ++i)
#pragma unroll
            for (int j = 0; j <= LB; ++j)
#pragma unroll
                for (int t = 0; t <= i + j; ++t) MQC_OPAQUE(e[(i * (LB + 1) + j) * NT + t]);
    }
    MQC_HD void build(double xpa, double xpb, double hp /* 1/(2p) */)
    {
#pragma unroll
        for (int k = 0; k < (LA + 1) * (LB + 1) * NT; ++k) e[k] = 0.0;
        at(0, 0, 0) = 1.0;
#pragma unroll
        for (int i = 0; i <= LA; ++i) {
            if (i > 0) {
#pragma unroll
                for (int t = 0; t <= i; ++t) {
                    double v = xpa * get(i - 1, 0, t);
                    if (t > 0) v += hp * get(i - 1, 0, t - 1);
                    if (t + 1 <= i - 1) v += (t + 1) * get(i - 1, 0, t + 1);
                    at(i, 0, t) = v;
                }
            }
#pragma unroll
            for (int j = 1; j <= LB; ++j) {
#pragma unroll
                for (int t = 0; t <= i + j; ++t) {
                    double v = xpb * get(i, j - 1, t);
                    if (t > 0) v += hp * get(i, j - 1, t - 1);
                    if (t + 1 <= i + j - 1) v += (t + 1) * get(i, j - 1, t + 1);
                    at(i, j, t) = v;
                }
            }
        }
    }
};

// ---------------------------------------------------------------------------------------
// Cartesian -> real solid harmonic coefficients.  s and p shells carry their constant
// (0.28209479..., 0.48860251...) inside the contraction coefficients (basis_norm.cpp), so
// their transform is the identity; d uses libcint's table (mqc_libcint_ao_data.f90:43-52),
// compile-time so that the zeros fold away; l >= 3 reads the packed table built on the host.
MQC_HD constexpr int c2s_table_offset(int l)
{
    int off = 0;
    for (int k = 0; k < l; ++k) off += nsph(k) * ncart(k);
    return off;
}

template <int L>
MQC_HD double c2s_coef(const double* __restrict__ table, int s, int c)
{
    if constexpr (L < 2) {
        return s == c ? 1.0 : 0.0;
    } else if constexpr (L == 2) {
        constexpr double XY = 1.0925484305920792, Z2A = -0.31539156525252005, Z2B = 0.6307831305050401,
                         X2Y2 = 0.5462742152960396;
        constexpr double T[5][6] = {{0, XY, 0, 0, 0, 0}, {0, 0, 0, 0, XY, 0}, {Z2A, 0, 0, Z2A, 0, Z2B},
                                    {0, 0, XY, 0, 0, 0}, {X2Y2, 0, 0, -X2Y2, 0, 0}};
        return T[s][c];
    } else {
        return table[c2s_table_offset(L) + s * ncart(L) + c];
    }
}

struct ShellRef {
    int nprim;
    const double* exps;     // [nprim]
    const double* coefs;    // [nprim] normalised radial coefficients
    double x, y, z;
};

// One primitive pair of a shell pair: exponent sum p, product centre P, K' = exp(-ab/p |AB|^2) ca cb / p
// and 1/(2p).  The ERI routines take their bra and ket primitive pairs from a SOURCE; PairFly builds
// them from two shells on the fly and screens pairs whose Gaussian product factor underflows 1e-20.
// (A table of precomputed pairs in HBM was measured and was no faster: the quartet loops are bound by
// FP64 issue, not by the exp(), and the table loads cost as much as they saved.)
struct PrimPair {
    double p, px, py, pz, kp, hp;
};
constexpr int PAIR_REC = 6;
constexpr double PRIM_EXP_CUTOFF = 46.0;

template <bool COEF>
struct PairFlyT {
    ShellRef A, B;
    double ab2;
    MQC_HD PairFlyT(const ShellRef& a, const ShellRef& b) : A(a), B(b)
    {
        const double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
        ab2 = dx * dx + dy * dy + dz * dz;
    }
    MQC_HD int npa() const { return A.nprim; }
    MQC_HD int npb() const { return B.nprim; }
    MQC_HD double ax() const { return A.x; }
    MQC_HD double ay() const { return A.y; }
    MQC_HD double az() const { return A.z; }
    MQC_HD double bx() const { return B.x; }
    MQC_HD double by() const { return B.y; }
    MQC_HD double bz() const { return B.z; }
    MQC_HD int npairs() const { return A.nprim * B.nprim; }
    MQC_HD PrimPair get_flat(int k) const { const int ip = k / B.nprim; return get(ip, k - ip * B.nprim); }
    MQC_HD PrimPair get(int ip, int jp) const
    {
        const double a = A.exps[ip], b = B.exps[jp];
        const double p = a + b, ip_ = fast_rcp(p);
        PrimPair r;
        r.p = p; r.hp = 0.5 * ip_;
        r.px = (a * A.x + b * B.x) * ip_; r.py = (a * A.y + b * B.y) * ip_; r.pz = (a * A.z + b * B.z) * ip_;
        // primitive screening: exp(-46) = 1e-20 -- a pair this far apart contributes nothing representable
        const double arg = a * b * ip_ * ab2;
        if constexpr (COEF) r.kp = (arg < PRIM_EXP_CUTOFF) ? exp(-arg) * A.coefs[ip] * B.coefs[jp] * ip_ : 0.0;
        else r.kp = (arg < PRIM_EXP_CUTOFF) ? exp(-arg) * ip_ : 0.0;
        return r;
    }
};
using PairFly = PairFlyT<true>;       // contraction coefficients folded into K'
using PairFlyRaw = PairFlyT<false>;   // bare primitive pairs (twin-shell blocks apply several coefficient sets)


// decode Cartesian component k of shell l -> (lx, ly, lz), libcint order
MQC_HD void cart_lmn(int l, int k, int& lx, int& ly, int& lz)
{
    int r = 0;
    while ((r + 1) * (r + 2) / 2 <= k) ++r;
    lz = k - r * (r + 1) / 2;
    ly = r - lz;
    lx = l - r;
}

// classes whose Cartesian block is at most this many numbers are fully unrolled into registers;
// larger ones keep rolled component loops (tables addressed at run time) to bound code size
constexpr int ERI_UNROLL_LIMIT = 324;

// ---------------------------------------------------------------------------------------
// Twin s shells.  Dunning-type sets give an atom two s functions contracted over the SAME primitives
// (cc-pVDZ oxygen: 9 primitives, two coefficient columns); a segmented treatment forms every primitive
// integral over them twice per index.  A twin block forms the primitive integrals once and adds them
// into one accumulator set per member combination with that combination's coefficient product
// (libcint's general-contraction loop does the same on the CPU: cint2e.c CINT2e_loop_nopt).
// c?[0] = coefficients of the first member, c?[1] = of the second (any valid pointer with f? = 0 when that
// position is an ordinary shell: branch-free inner loops).  Only s positions can be twins (multiplicity 2), others have multiplicity 1.
#ifndef MQC_PLAIN_TK
#define MQC_PLAIN_TK 2
#endif
struct NoTwin {
    static constexpr bool enabled = false;
};
struct TwinCoefs {
    static constexpr bool enabled = true;
    const double* ca[2];
    const double* cb[2];
    const double* cc[2];
    const double* cd[2];
    double fa, fb, fc, fd;    // 1.0 where the position is a twin, 0.0 where c?[1] merely repeats c?[0]
};
MQC_HD constexpr int twin_mult(bool enabled, int l) { return (enabled && l == 0) ? 2 : 1; }
MQC_HD double twin_coef(const double* const* c, double f, int m, int k) { return m == 0 ? c[0][k] : f * c[1][k]; }

// ---------------------------------------------------------------------------------------
// Hermite intermediate (bra-outer form).  A contracted integral is
//   (ab|cd) = sum_P K_P sum_{tuv} Eab_{tuv}(P) [ sum_Q K_Q sum_{t'u'v'} (-1)^{t'+u'+v'} Ecd_{t'u'v'}(Q) R_{t+t',u+u',v+v'}(P,Q) ]
// and the bracket H[cd][tuv] -- nherm(LA+LB) numbers per ket component -- knows nothing of the bra COMPONENT.  The
// ket primitive loop therefore only accumulates H (a handful of FMAs per ket component), and the bra Hermite
// coefficients are built and applied ONCE per bra primitive pair, after that loop.  The one-quartet-at-a-time form
// applied them inside it: for a (pp| or (dp| bra that was 60-80 % of the arithmetic of every primitive quartet, and
// the deep contractions of Dunning sets sit on the ket side (s shells of 3-9 primitives) of most classes.  The bra
// tables are not live inside the ket loop either, which is what had pushed the d classes into scratch memory.
#ifndef MQC_BRA_OUTER
#define MQC_BRA_OUTER 1
#endif
#ifndef MQC_BRA_OUTER_HMAX
#define MQC_BRA_OUTER_HMAX 72
#endif
// bra primitive pairs held at a time (one Hermite intermediate each): 4 while an intermediate is at most this many
// numbers, 2 up to the second bound, else 1
#ifndef MQC_BRA_BLOCK_H4
#define MQC_BRA_BLOCK_H4 12
#endif
#ifndef MQC_BRA_BLOCK_H2
#define MQC_BRA_BLOCK_H2 30
#endif
MQC_HD constexpr bool eri_bra_outer(int la, int lb, int lc, int ld)
{
    const int nab = ncart(la) * ncart(lb), ncd = ncart(lc) * ncart(ld);
    // (ss|ss) and (ps|ss) have no bra contraction worth moving; H must stay in registers
    return MQC_BRA_OUTER != 0 && nab * ncd > 3 && ncd * nherm(la + lb) <= MQC_BRA_OUTER_HMAX;
}

// H[icd][h] += sum_{t'u'v'} (-1)^{t'+u'+v'} Ecd[icd][t'u'v'] R[h + (t'u'v')], all Cartesian ket components
template <int LAB, int LC, int LD>
MQC_HD void ket_into_hermite(const E1D<LC, LD>& fx, const E1D<LC, LD>& fy, const E1D<LC, LD>& fz, const double* R, double* H)
{
    constexpr int NHAB = nherm(LAB);
    int icd = 0;
#pragma unroll
    for (int cx = LC; cx >= 0; --cx) {
#pragma unroll
        for (int cy = LC - cx; cy >= 0; --cy) {
            const int cz = LC - cx - cy;
#pragma unroll
            for (int dx = LD; dx >= 0; --dx) {
#pragma unroll
                for (int dy = LD - dx; dy >= 0; --dy) {
                    const int dz = LD - dx - dy;
#pragma unroll
                    for (int tt = 0; tt <= cx + dx; ++tt) {
#pragma unroll
                        for (int uu = 0; uu <= cy + dy; ++uu) {
#pragma unroll
                            for (int ww = 0; ww <= cz + dz; ++ww) {
                                double f = fx.get(cx, dx, tt) * fy.get(cy, dy, uu) * fz.get(cz, dz, ww);
                                if ((tt + uu + ww) & 1) f = -f;
#pragma unroll
                                for (int N = 0; N <= LAB; ++N) {
#pragma unroll
                                    for (int t = N; t >= 0; --t) {
#pragma unroll
                                        for (int u = N - t; u >= 0; --u) {
                                            const int v = N - t - u;
                                            H[icd * NHAB + hidx(t, u, v)] += f * R[hidx(t + tt, u + uu, v + ww)];
                                        }
                                    }
                                }
                            }
                        }
                    }
                    ++icd;
                }
            }
        }
    }
}

// out[iab * NK + k] += kp * sum_{tuv} Eab[iab][tuv] H[k][tuv] for k < NK (ostride between consecutive out entries)
template <int LA, int LB, int NK>
MQC_HD void bra_from_hermite(const E1D<LA, LB>& ex, const E1D<LA, LB>& ey, const E1D<LA, LB>& ez, const double* H, double kp,
                             double* out, int ostride = 1)
{
    constexpr int NHAB = nherm(LA + LB);
    int iab = 0;
#pragma unroll
    for (int ax = LA; ax >= 0; --ax) {
#pragma unroll
        for (int ay = LA - ax; ay >= 0; --ay) {
            const int az = LA - ax - ay;
#pragma unroll
            for (int bx = LB; bx >= 0; --bx) {
#pragma unroll
                for (int by = LB - bx; by >= 0; --by) {
                    const int bz = LB - bx - by;
                    double sm[NK];
#pragma unroll
                    for (int k = 0; k < NK; ++k) sm[k] = 0.0;
#pragma unroll
                    for (int t = 0; t <= ax + bx; ++t) {
#pragma unroll
                        for (int u = 0; u <= ay + by; ++u) {
#pragma unroll
                            for (int v = 0; v <= az + bz; ++v) {
                                const double e = ex.get(ax, bx, t) * ey.get(ay, by, u) * ez.get(az, bz, v);
#pragma unroll
                                for (int k = 0; k < NK; ++k) sm[k] += e * H[k * NHAB + hidx(t, u, v)];
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < NK; ++k) out[(iab * NK + k) * ostride] += kp * sm[k];
                    ++iab;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Contracted Cartesian ERI block (ab|cd), out[((ia*NCB+ib)*NCC+ic)*NCD+id] (accumulated
// into a zeroed buffer by this routine).
template <int LA, int LB, int LC, int LD, class Bra, class Ket, class Twin = NoTwin>
MQC_HD void eri_cart_block_src(const Bra& bra, const Ket& ket, const double* __restrict__ boys_table, double* out,
                               const Twin& tw = Twin(), int bra_first = 0, int bra_step = 0)
{
    constexpr int NCA = ncart(LA), NCB = ncart(LB), NCC = ncart(LC), NCD = ncart(LD);
    constexpr int LAB = LA + LB, LCD = LC + LD, L = LAB + LCD;
    constexpr int NHAB = nherm(LAB);
    constexpr bool UNROLLED = (NCA * NCB * NCC * NCD <= ERI_UNROLL_LIMIT);
    constexpr int MA = twin_mult(Twin::enabled, LA), MB = twin_mult(Twin::enabled, LB);
    constexpr int MC = twin_mult(Twin::enabled, LC), MD = twin_mult(Twin::enabled, LD);
    constexpr int NCOMB = MA * MB * MC * MD, NC = NCA * NCB * NCC * NCD;
    (void)LCD; (void)tw;
    if constexpr (UNROLLED) {
#pragma unroll
        for (int i = 0; i < NCOMB * NC; ++i) out[i] = 0.0;
    } else {
        for (int i = 0; i < NCOMB * NC; ++i) out[i] = 0.0;
    }

    constexpr double TWO_PI_25 = 34.986836655249725693;   // 2 pi^(5/2)

    // one primitive quartet: bra record + Hermite tables, ket record
    auto quartet = [&](const PrimPair& P, const E1D<LA, LB>& ex, const E1D<LA, LB>& ey, const E1D<LA, LB>& ez,
                       const PrimPair& Qp, auto&& emit) {
            const double p = P.p, px = P.px, py = P.py, pz = P.pz;
            const double q = Qp.p, qx = Qp.px, qy = Qp.py, qz = Qp.pz;
            E1D<LC, LD> fx, fy, fz;
            fx.build(qx - ket.ax(), qx - ket.bx(), Qp.hp);
            fy.build(qy - ket.ay(), qy - ket.by(), Qp.hp);
            fz.build(qz - ket.az(), qz - ket.bz(), Qp.hp);
            const double rs = fast_rsqrt(p + q);
            const double rpq = rs * rs;
            const double alpha = p * q * rpq;
            const double pref = TWO_PI_25 * rs * P.kp * Qp.kp;
            double R[nherm(L)];
            hermite_r<L>(alpha, px - qx, py - qy, pz - qz, boys_table, R);

            if constexpr (UNROLLED) {
                // loop over ket components; for each, G[h] over bra Hermite indices
                int icd = 0;
#pragma unroll
                for (int cx = LC; cx >= 0; --cx) {
#pragma unroll
                    for (int cy = LC - cx; cy >= 0; --cy) {
                        const int cz = LC - cx - cy;
#pragma unroll
                        for (int dx = LD; dx >= 0; --dx) {
#pragma unroll
                            for (int dy = LD - dx; dy >= 0; --dy) {
                                const int dz = LD - dx - dy;
                                double G[NHAB];
#pragma unroll
                                for (int h = 0; h < NHAB; ++h) G[h] = 0.0;
#pragma unroll
                                for (int tt = 0; tt <= cx + dx; ++tt) {
#pragma unroll
                                    for (int uu = 0; uu <= cy + dy; ++uu) {
#pragma unroll
                                        for (int ww = 0; ww <= cz + dz; ++ww) {
                                            double f = fx.get(cx, dx, tt) * fy.get(cy, dy, uu) * fz.get(cz, dz, ww);
                                            if ((tt + uu + ww) & 1) f = -f;
#pragma unroll
                                            for (int N = 0; N <= LAB; ++N) {
#pragma unroll
                                                for (int t = N; t >= 0; --t) {
#pragma unroll
                                                    for (int u = N - t; u >= 0; --u) {
                                                        const int v = N - t - u;
                                                        G[hidx(t, u, v)] += f * R[hidx(t + tt, u + uu, v + ww)];
                                                    }
                                                }
                                            }
                                        }
                                    }
                                }
                                // out[ab][cd] += pref * sum_h Eab[h] G[h]
                                int iab = 0;
#pragma unroll
                                for (int ax = LA; ax >= 0; --ax) {
#pragma unroll
                                    for (int ay = LA - ax; ay >= 0; --ay) {
                                        const int az = LA - ax - ay;
#pragma unroll
                                        for (int bx = LB; bx >= 0; --bx) {
#pragma unroll
                                            for (int by = LB - bx; by >= 0; --by) {
                                                const int bz = LB - bx - by;
                                                double s = 0.0;
#pragma unroll
                                                for (int t = 0; t <= ax + bx; ++t) {
#pragma unroll
                                                    for (int u = 0; u <= ay + by; ++u) {
#pragma unroll
                                                        for (int v = 0; v <= az + bz; ++v) {
                                                            s += ex.get(ax, bx, t) * ey.get(ay, by, u) * ez.get(az, bz, v) *
                                                                 G[hidx(t, u, v)];
                                                        }
                                                    }
                                                }
                                                emit(iab * (NCC * NCD) + icd, pref * s);
                                                ++iab;
                                            }
                                        }
                                    }
                                }
                                ++icd;
#if defined(__HIP_DEVICE_COMPILE__)
                                // keep the scheduler from interleaving ket components: the temporaries of
                                // one component die before the next starts (register pressure, not order)
                                __builtin_amdgcn_sched_barrier(0);
#endif
                            }
                        }
                    }
                }
            } else {
                // rolled form for the large classes: same arithmetic, run-time component indices
#pragma unroll 1
                for (int icd = 0; icd < NCC * NCD; ++icd) {
                    int cx, cy, cz, dx, dy, dz;
                    cart_lmn(LC, icd / NCD, cx, cy, cz);
                    cart_lmn(LD, icd % NCD, dx, dy, dz);
                    double G[NHAB];
                    for (int h = 0; h < NHAB; ++h) G[h] = 0.0;
                    for (int tt = 0; tt <= cx + dx; ++tt)
                        for (int uu = 0; uu <= cy + dy; ++uu)
                            for (int ww = 0; ww <= cz + dz; ++ww) {
                                double f = fx.get(cx, dx, tt) * fy.get(cy, dy, uu) * fz.get(cz, dz, ww);
                                if ((tt + uu + ww) & 1) f = -f;
                                for (int N = 0; N <= LAB; ++N)
                                    for (int t = N; t >= 0; --t)
                                        for (int u = N - t; u >= 0; --u) {
                                            const int v = N - t - u;
                                            G[hidx(t, u, v)] += f * R[hidx(t + tt, u + uu, v + ww)];
                                        }
                            }
#pragma unroll 1
                    for (int iab = 0; iab < NCA * NCB; ++iab) {
                        int ax, ay, az, bx, by, bz;
                        cart_lmn(LA, iab / NCB, ax, ay, az);
                        cart_lmn(LB, iab % NCB, bx, by, bz);
                        double s = 0.0;
                        for (int t = 0; t <= ax + bx; ++t)
                            for (int u = 0; u <= ay + by; ++u)
                                for (int v = 0; v <= az + bz; ++v)
                                    s += ex.get(ax, bx, t) * ey.get(ay, by, u) * ez.get(az, bz, v) * G[hidx(t, u, v)];
                        emit(iab * (NCC * NCD) + icd, pref * s);
                    }
                }
            }
    };

    // Small classes (block <= 27 numbers): ket records are formed TK at a time and held in registers
    // while ALL bra records stream past, so the ket-side exp/reciprocal/centre arithmetic -- a third of
    // the inner loop of (ss|ss) -- runs once per record instead of once per primitive quartet.
    constexpr int TK = !(UNROLLED && NCA * NCB * NCC * NCD <= 27) ? 1 : (NCA * NCB * NCC * NCD <= 3 ? 4 : MQC_PLAIN_TK);
    auto plain = [&](int i, double v) { out[i] += v; };
    (void)plain;
    constexpr bool BRA_OUTER = UNROLLED && !Twin::enabled && eri_bra_outer(LA, LB, LC, LD);
    if constexpr (BRA_OUTER) {
        // Bra primitive pairs outside, TB of them at a time with one Hermite intermediate each; ket primitive pairs
        // inside: a ket record (exponent sum, centre, exp() factor) and its Hermite tables are formed once per bra BLOCK
        // and meet TB independent Boys / R_tuv evaluations, whose table loads overlap.  The bra tables exist only
        // after the ket loop.
        constexpr int NK = NCC * NCD, NKH = NK * NHAB;
        constexpr int TB = NKH <= MQC_BRA_BLOCK_H4 ? 4 : (NKH <= MQC_BRA_BLOCK_H2 ? 2 : 1);
        const int nkl = ket.npairs(), nb = bra.npairs();
        int bi = 0, bj = 0;
        for (int b0 = 0; b0 < nb; b0 += TB) {
            PrimPair Pb[TB];
            bool any = false;
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                if (b0 + t < nb) {
                    Pb[t] = bra.get(bi, bj);
                    if (++bj == bra.npb()) { bj = 0; ++bi; }
                } else {
                    Pb[t] = PrimPair{};
                    Pb[t].p = 1.0; Pb[t].kp = 0.0;
                }
                any = any || (Pb[t].kp != 0.0);
            }
            if (!any) continue;
            double H[TB][NKH];
#pragma unroll
            for (int t = 0; t < TB; ++t)
#pragma unroll
                for (int h = 0; h < NKH; ++h) H[t][h] = 0.0;
            PrimPair Qn = ket.get(0, 0);
            int kc = 0, kd = 0;
            for (int kl = 0; kl < nkl; ++kl) {
                const PrimPair Qp = Qn;
                if (++kd == ket.npb()) { kd = 0; ++kc; }
                if (kl + 1 < nkl) Qn = ket.get(kc, kd);
                if (Qp.kp == 0.0) continue;
                E1D<LC, LD> fx, fy, fz;
                fx.build(Qp.px - ket.ax(), Qp.px - ket.bx(), Qp.hp);
                fy.build(Qp.py - ket.ay(), Qp.py - ket.by(), Qp.hp);
                fz.build(Qp.pz - ket.az(), Qp.pz - ket.bz(), Qp.hp);
                const double kq = TWO_PI_25 * Qp.kp;
#pragma unroll
                for (int t = 0; t < TB; ++t) {
                    if (TB > 1 && Pb[t].kp == 0.0) continue;
                    const double rs = fast_rsqrt(Pb[t].p + Qp.p);
                    const double alpha = Pb[t].p * Qp.p * rs * rs;
                    double R[nherm(L)];
                    hermite_r<L>(alpha, Pb[t].px - Qp.px, Pb[t].py - Qp.py, Pb[t].pz - Qp.pz, boys_table, R, kq * rs);
                    ket_into_hermite<LAB, LC, LD>(fx, fy, fz, R, H[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                if (Pb[t].kp == 0.0) continue;
                E1D<LA, LB> ex, ey, ez;
                ex.build(Pb[t].px - bra.ax(), Pb[t].px - bra.bx(), Pb[t].hp);
                ey.build(Pb[t].py - bra.ay(), Pb[t].py - bra.by(), Pb[t].hp);
                ez.build(Pb[t].pz - bra.az(), Pb[t].pz - bra.bz(), Pb[t].hp);
                bra_from_hermite<LA, LB, NK>(ex, ey, ez, H[t], Pb[t].kp, out);
            }
        }
    } else if constexpr (Twin::enabled) {
      if (bra_step > 0) {
        // PARTIAL twin block over the bra primitive pairs bra_first, bra_first + bra_step, ... (eri_twin_wave_kernel: the
        // 64 lanes of a wave share ONE deeply contracted entry of a small batch and add their parts up afterwards).
        // Bra pairs outside, one Hermite intermediate per ket member combination, as in the bra-outer form above: the
        // ket loop only accumulates H[mcd][cd][tuv]; bra tables, bra members' weights and the accumulator sets come in
        // once per bra pair.
        constexpr int NK = NCC * NCD, MCD = MC * MD, HSZ = MCD * NK * NHAB;
        const int nkl = ket.npairs(), nb = bra.npairs();
        for (int b = bra_first; b < nb; b += bra_step) {
            const int pi = b / bra.npb(), pj = b - pi * bra.npb();
            const PrimPair P = bra.get(pi, pj);
            if (P.kp == 0.0) continue;
            double H[HSZ];
#pragma unroll
            for (int h = 0; h < HSZ; ++h) H[h] = 0.0;
            PrimPair Qn = ket.get(0, 0);
            int kc = 0, kd = 0;
            for (int kl = 0; kl < nkl; ++kl) {
                const PrimPair Qp = Qn;
                double wcd[MCD];
#pragma unroll
                for (int mc = 0; mc < MC; ++mc)
#pragma unroll
                    for (int md = 0; md < MD; ++md) wcd[mc * MD + md] = twin_coef(tw.cc, tw.fc, mc, kc) * twin_coef(tw.cd, tw.fd, md, kd);
                if (++kd == ket.npb()) { kd = 0; ++kc; }
                if (kl + 1 < nkl) Qn = ket.get(kc, kd);
                if (Qp.kp == 0.0) continue;
                E1D<LC, LD> fx, fy, fz;
                fx.build(Qp.px - ket.ax(), Qp.px - ket.bx(), Qp.hp);
                fy.build(Qp.py - ket.ay(), Qp.py - ket.by(), Qp.hp);
                fz.build(Qp.pz - ket.az(), Qp.pz - ket.bz(), Qp.hp);
                const double rs = fast_rsqrt(P.p + Qp.p);
                const double alpha = P.p * Qp.p * rs * rs;
                double R[nherm(L)];
                hermite_r<L>(alpha, P.px - Qp.px, P.py - Qp.py, P.pz - Qp.pz, boys_table, R, TWO_PI_25 * Qp.kp * rs);
                double G[NK * NHAB];
#pragma unroll
                for (int h = 0; h < NK * NHAB; ++h) G[h] = 0.0;
                ket_into_hermite<LAB, LC, LD>(fx, fy, fz, R, G);
#pragma unroll
                for (int m = 0; m < MCD; ++m)
#pragma unroll
                    for (int h = 0; h < NK * NHAB; ++h) H[m * NK * NHAB + h] += wcd[m] * G[h];
            }
            E1D<LA, LB> ex, ey, ez;
            ex.build(P.px - bra.ax(), P.px - bra.bx(), P.hp);
            ey.build(P.py - bra.ay(), P.py - bra.by(), P.hp);
            ez.build(P.pz - bra.az(), P.pz - bra.bz(), P.hp);
#pragma unroll
            for (int ma = 0; ma < MA; ++ma)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const double wab = P.kp * twin_coef(tw.ca, tw.fa, ma, pi) * twin_coef(tw.cb, tw.fb, mb, pj);
#pragma unroll
                    for (int m = 0; m < MCD; ++m)
                        bra_from_hermite<LA, LB, NK>(ex, ey, ez, H + m * NK * NHAB, wab, out + ((ma * MB + mb) * MCD + m) * NC);
                }
        }
      } else {
        static_assert(!Twin::enabled || TK > 1, "twin blocks exist for the small register classes only");
        constexpr int TKT = (NC == 1) ? 2 : 1;   // ket records held at a time (register budget: NCOMB accumulator sets)
        const int nkl = ket.npairs();
        int kc = 0, kd = 0;
        for (int kl0 = 0; kl0 < nkl; kl0 += TKT) {
            PrimPair Qt[TKT];
            double wcd[TKT][MC * MD];
#pragma unroll
            for (int t = 0; t < TKT; ++t) {
                if (kl0 + t < nkl) {
                    Qt[t] = ket.get(kc, kd);
#pragma unroll
                    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
                        for (int md = 0; md < MD; ++md) wcd[t][mc * MD + md] = twin_coef(tw.cc, tw.fc, mc, kc) * twin_coef(tw.cd, tw.fd, md, kd);
                    if (++kd == ket.npb()) { kd = 0; ++kc; }
                } else {
                    Qt[t] = PrimPair{};
                    Qt[t].p = 1.0; Qt[t].kp = 0.0;
#pragma unroll
                    for (int m = 0; m < MC * MD; ++m) wcd[t][m] = 0.0;
                }
            }
            for (int ip = 0; ip < bra.npa(); ++ip) {
                for (int jp = 0; jp < bra.npb(); ++jp) {
                    const PrimPair P = bra.get(ip, jp);
                    if (P.kp == 0.0) continue;
                    double wab[MA * MB];
#pragma unroll
                    for (int ma = 0; ma < MA; ++ma)
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) wab[ma * MB + mb] = twin_coef(tw.ca, tw.fa, ma, ip) * twin_coef(tw.cb, tw.fb, mb, jp);
                    E1D<LA, LB> ex, ey, ez;
                    ex.build(P.px - bra.ax(), P.px - bra.bx(), P.hp);
                    ey.build(P.py - bra.ay(), P.py - bra.by(), P.hp);
                    ez.build(P.pz - bra.az(), P.pz - bra.bz(), P.hp);
#pragma unroll
                    for (int t = 0; t < TKT; ++t) {
                        if (Qt[t].kp != 0.0) {
                            quartet(P, ex, ey, ez, Qt[t], [&](int i, double v) {
#pragma unroll
                                for (int mcd = 0; mcd < MC * MD; ++mcd) {
                                    const double vc = wcd[t][mcd] * v;
#pragma unroll
                                    for (int mab = 0; mab < MA * MB; ++mab) out[(mab * (MC * MD) + mcd) * NC + i] += wab[mab] * vc;
                                }
                            });
                        }
                    }
                }
            }
        }
      }
    } else if constexpr (TK > 1) {
        const int nkl = ket.npairs();
        int kc = 0, kd = 0;
        for (int kl0 = 0; kl0 < nkl; kl0 += TK) {
            PrimPair Qt[TK];
#pragma unroll
            for (int t = 0; t < TK; ++t) {
                if (kl0 + t < nkl) {
                    Qt[t] = ket.get(kc, kd);
                    if (++kd == ket.npb()) { kd = 0; ++kc; }
                } else {
                    Qt[t] = PrimPair{};
                    Qt[t].p = 1.0; Qt[t].kp = 0.0;
                }
            }
            for (int ip = 0; ip < bra.npa(); ++ip) {
                for (int jp = 0; jp < bra.npb(); ++jp) {
                    const PrimPair P = bra.get(ip, jp);
                    if (P.kp == 0.0) continue;
                    E1D<LA, LB> ex, ey, ez;
                    ex.build(P.px - bra.ax(), P.px - bra.bx(), P.hp);
                    ey.build(P.py - bra.ay(), P.py - bra.by(), P.hp);
                    ez.build(P.pz - bra.az(), P.pz - bra.bz(), P.hp);
#pragma unroll
                    for (int t = 0; t < TK; ++t) {
                        if (Qt[t].kp != 0.0) quartet(P, ex, ey, ez, Qt[t], plain);
                    }
                }
            }
        }
    } else {
        for (int ip = 0; ip < bra.npa(); ++ip) {
            for (int jp = 0; jp < bra.npb(); ++jp) {
                const PrimPair P = bra.get(ip, jp);
                if (P.kp == 0.0) continue;          // screened primitive pair (wave-uniform for rigid monomers)
                E1D<LA, LB> ex, ey, ez;
                ex.build(P.px - bra.ax(), P.px - bra.bx(), P.hp);
                ey.build(P.py - bra.ay(), P.py - bra.by(), P.hp);
                ez.build(P.pz - bra.az(), P.pz - bra.bz(), P.hp);
                // ket primitive pairs as ONE loop with the next record formed ahead of the arithmetic
                const int nkl = ket.npairs();
                PrimPair Qn = ket.get(0, 0);
                int kc = 0, kd = 0;
                for (int kl = 0; kl < nkl; ++kl) {
                    const PrimPair Qp = Qn;
                    if (++kd == ket.npb()) { kd = 0; ++kc; }
                    if (kl + 1 < nkl) Qn = ket.get(kc, kd);
                    if (Qp.kp == 0.0) continue;
                    quartet(P, ex, ey, ez, Qp, plain);
                }
            }
        }
    }
}

template <int LA, int LB, int LC, int LD>
MQC_HD void eri_cart_block(const ShellRef& A, const ShellRef& B, const ShellRef& C, const ShellRef& D,
                           const double* __restrict__ boys_table, double* out)
{
    const PairFly bra(A, B), ket(C, D);
    eri_cart_block_src<LA, LB, LC, LD>(bra, ket, boys_table, out);
}

// Twin block: out[combo][NC] with combo = ((ma*MB + mb)*MC + mc)*MD + md over the member multiplicities
// (2 on s positions, 1 elsewhere); A..D are the FIRST members, tw carries both coefficient columns.
constexpr bool eri_has_twin_block(int la, int lb, int lc, int ld)
{
    // The block routine covers every class of at most 27 Cartesian components with an s position; the engine
    // uses it where it measured faster than the segmented kernels on MI355X -- (ss|ss) and (ps|ss), where the
    // extra accumulator sets are few.  With more components the NCOMB x NC accumulate per primitive quartet and
    // the register pressure (one wave per SIMD) cancel the saved primitive work.
    return la <= 1 && lb == 0 && lc == 0 && ld == 0;
}
constexpr int eri_twin_combos(int la, int lb, int lc, int ld)
{
    return twin_mult(true, la) * twin_mult(true, lb) * twin_mult(true, lc) * twin_mult(true, ld);
}
template <int LA, int LB, int LC, int LD>
MQC_HD void eri_cart_block_twin(const ShellRef& A, const ShellRef& B, const ShellRef& C, const ShellRef& D, const TwinCoefs& tw,
                                const double* __restrict__ boys_table, double* out, int bra_first = 0, int bra_step = 0)
{
    const PairFlyRaw bra(A, B), ket(C, D);
    eri_cart_block_src<LA, LB, LC, LD, PairFlyRaw, PairFlyRaw, TwinCoefs>(bra, ket, boys_table, out, tw, bra_first, bra_step);
}

// ---------------------------------------------------------------------------------------
// Pass-structured ERI block for the classes whose accumulators do not fit the register file.
//
//   * the ket index is transformed to real solid harmonics ON THE FLY (the c2s weight is folded
//     into the ket Hermite coefficient), so only NCA*NCB x (ket spherical components) numbers
//     are accumulated and no ket-side post-transform exists;
//   * the ket spherical components are processed in PASSES of at most CH components; within a
//     pass the accumulators acc[(iab*CH + s)*stride] live wherever the caller put them -- the
//     gfx950 kernels pass a wave-private LDS slab ([accumulator][lane], stride 64: conflict-free,
//     one ds_read/ds_write pair per update), the host harness a plain array (stride 1);
//   * after a pass the bra indices are transformed and every finished value is handed to
//     `sink(i, j, k, l, value)` (spherical indices inside the shell block).
// Each pass recomputes the Hermite tables and R_tuv for its primitive quartets: more arithmetic,
// no scratch memory -- which is what the profile said these classes were waiting on.
template <int LA, int LB, int LC, int LD, int CH, int PASS, class Bra, class Ket, class Sink>
MQC_HD void eri_pass(const Bra& bra, const Ket& ket,
                     const double* __restrict__ boys_table, const double* __restrict__ c2s,
                     double* __restrict__ acc, int stride, Sink& sink)
{
    constexpr int NCA = ncart(LA), NCB = ncart(LB), NCC = ncart(LC), NCD = ncart(LD);
    constexpr int NSA = nsph(LA), NSB = nsph(LB), NSC = nsph(LC), NSD = nsph(LD);
    constexpr int LAB = LA + LB, L = LA + LB + LC + LD;
    constexpr int NHAB = nherm(LAB);
    constexpr int S0 = PASS * CH, S1 = (S0 + CH < NSC * NSD) ? S0 + CH : NSC * NSD;
    constexpr int NS = S1 - S0;
    (void)NSC;
#pragma unroll
    for (int i = 0; i < NCA * NCB * NS; ++i) acc[i * stride] = 0.0;

    constexpr double TWO_PI_25 = 34.986836655249725693;

#ifndef MQC_PASS_HMAX
#define MQC_PASS_HMAX 96
#endif
    if constexpr (MQC_BRA_OUTER != 0 && NS * NHAB <= MQC_PASS_HMAX) {
        // Hermite-intermediate form (see eri_bra_outer): per primitive quartet only H[s][tuv] of the pass's ket
        // components is touched (registers); the bra tables are built after the ket loop and the LDS accumulators
        // are updated once per bra primitive pair instead of once per primitive quartet
        const int nkl = ket.npairs();
        for (int ip = 0; ip < bra.npa(); ++ip) {
            for (int jp = 0; jp < bra.npb(); ++jp) {
                const PrimPair P = bra.get(ip, jp);
                if (P.kp == 0.0) continue;
                double H[NS * NHAB];
#pragma unroll
                for (int h = 0; h < NS * NHAB; ++h) H[h] = 0.0;
                PrimPair Qn = ket.get(0, 0);
                int kc = 0, kd = 0;
                for (int kl = 0; kl < nkl; ++kl) {
                    const PrimPair Qp = Qn;
                    if (++kd == ket.npb()) { kd = 0; ++kc; }
                    if (kl + 1 < nkl) Qn = ket.get(kc, kd);
                    if (Qp.kp == 0.0) continue;
                    E1D<LC, LD> fx, fy, fz;
                    fx.build(Qp.px - ket.ax(), Qp.px - ket.bx(), Qp.hp);
                    fy.build(Qp.py - ket.ay(), Qp.py - ket.by(), Qp.hp);
                    fz.build(Qp.pz - ket.az(), Qp.pz - ket.bz(), Qp.hp);
                    const double rs = fast_rsqrt(P.p + Qp.p);
                    const double alpha = P.p * Qp.p * rs * rs;
                    double R[nherm(L)];
                    hermite_r<L>(alpha, P.px - Qp.px, P.py - Qp.py, P.pz - Qp.pz, boys_table, R, TWO_PI_25 * rs * Qp.kp);
#pragma unroll
                    for (int s = S0; s < S1; ++s) {
                        const int mc = s / NSD, md = s - mc * NSD;
#pragma unroll
                        for (int kcc = 0; kcc < NCC; ++kcc) {
                            const double wc = c2s_coef<LC>(c2s, mc, kcc);
                            if (wc == 0.0) continue;
                            int cx = 0, cy = 0, cz = 0;
                            cart_lmn(LC, kcc, cx, cy, cz);
#pragma unroll
                            for (int kdd = 0; kdd < NCD; ++kdd) {
                                const double wd = c2s_coef<LD>(c2s, md, kdd);
                                if (wd == 0.0) continue;
                                int dx = 0, dy = 0, dz = 0;
                                cart_lmn(LD, kdd, dx, dy, dz);
#pragma unroll
                                for (int tt = 0; tt <= cx + dx; ++tt)
#pragma unroll
                                    for (int uu = 0; uu <= cy + dy; ++uu)
#pragma unroll
                                        for (int ww = 0; ww <= cz + dz; ++ww) {
                                            double f = (wc * wd) * fx.get(cx, dx, tt) * fy.get(cy, dy, uu) * fz.get(cz, dz, ww);
                                            if ((tt + uu + ww) & 1) f = -f;
#pragma unroll
                                            for (int N = 0; N <= LAB; ++N)
#pragma unroll
                                                for (int t = N; t >= 0; --t)
#pragma unroll
                                                    for (int u = N - t; u >= 0; --u)
                                                        H[(s - S0) * NHAB + hidx(t, u, N - t - u)] += f * R[hidx(t + tt, u + uu, N - t - u + ww)];
                                        }
                            }
                        }
                    }
                }
                E1D<LA, LB> ex, ey, ez;
                ex.build(P.px - bra.ax(), P.px - bra.bx(), P.hp);
                ey.build(P.py - bra.ay(), P.py - bra.by(), P.hp);
                ez.build(P.pz - bra.az(), P.pz - bra.bz(), P.hp);
                bra_from_hermite<LA, LB, NS>(ex, ey, ez, H, P.kp, acc, stride);
            }
        }
    } else
    for (int ip = 0; ip < bra.npa(); ++ip) {
        for (int jp = 0; jp < bra.npb(); ++jp) {
            const PrimPair P = bra.get(ip, jp);
            if (P.kp == 0.0) continue;          // screened primitive pair (wave-uniform for rigid monomers)
            const double p = P.p, px = P.px, py = P.py, pz = P.pz;
            E1D<LA, LB> ex, ey, ez;
            ex.build(px - bra.ax(), px - bra.bx(), P.hp);
            ey.build(py - bra.ay(), py - bra.by(), P.hp);
            ez.build(pz - bra.az(), pz - bra.bz(), P.hp);
            const int nkl = ket.npairs();
            PrimPair Qn = ket.get(0, 0);
            int kc = 0, kd = 0;
            for (int kl = 0; kl < nkl; ++kl) {
                {
                    ex.pin(); ey.pin(); ez.pin();
                    const PrimPair Qp = Qn;
                    if (++kd == ket.npb()) { kd = 0; ++kc; }
                    if (kl + 1 < nkl) Qn = ket.get(kc, kd);
                    if (Qp.kp == 0.0) continue;
                    const double q = Qp.p, qx = Qp.px, qy = Qp.py, qz = Qp.pz;
                    E1D<LC, LD> fx, fy, fz;
                    fx.build(qx - ket.ax(), qx - ket.bx(), Qp.hp);
                    fy.build(qy - ket.ay(), qy - ket.by(), Qp.hp);
                    fz.build(qz - ket.az(), qz - ket.bz(), Qp.hp);
                    const double rs = fast_rsqrt(p + q);
                    const double rpq = rs * rs;
                    const double alpha = p * q * rpq;
                    const double pref = TWO_PI_25 * rs * P.kp * Qp.kp;
                    double R[nherm(L)];
                    hermite_r<L>(alpha, px - qx, py - qy, pz - qz, boys_table, R);
#pragma unroll
                    for (int s = S0; s < S1; ++s) {
                        const int mc = s / NSD, md = s - mc * NSD;
                        double G[NHAB];
#pragma unroll
                        for (int h = 0; h < NHAB; ++h) G[h] = 0.0;
#pragma unroll
                        for (int kc = 0; kc < NCC; ++kc) {
                            const double wc = c2s_coef<LC>(c2s, mc, kc);
                            if (wc == 0.0) continue;
                            int cx = 0, cy = 0, cz = 0;
                            cart_lmn(LC, kc, cx, cy, cz);
#pragma unroll
                            for (int kd = 0; kd < NCD; ++kd) {
                                const double wd = c2s_coef<LD>(c2s, md, kd);
                                if (wd == 0.0) continue;
                                int dx = 0, dy = 0, dz = 0;
                                cart_lmn(LD, kd, dx, dy, dz);
#pragma unroll
                                for (int tt = 0; tt <= cx + dx; ++tt)
#pragma unroll
                                    for (int uu = 0; uu <= cy + dy; ++uu)
#pragma unroll
                                        for (int ww = 0; ww <= cz + dz; ++ww) {
                                            double f = (wc * wd) * fx.get(cx, dx, tt) * fy.get(cy, dy, uu) * fz.get(cz, dz, ww);
                                            if ((tt + uu + ww) & 1) f = -f;
#pragma unroll
                                            for (int N = 0; N <= LAB; ++N)
#pragma unroll
                                                for (int t = N; t >= 0; --t)
#pragma unroll
                                                    for (int u = N - t; u >= 0; --u)
                                                        G[hidx(t, u, N - t - u)] += f * R[hidx(t + tt, u + uu, N - t - u + ww)];
                                        }
                            }
                        }
                        int iab = 0;
#pragma unroll
                        for (int ax = LA; ax >= 0; --ax)
#pragma unroll
                            for (int ay = LA - ax; ay >= 0; --ay) {
                                const int az = LA - ax - ay;
#pragma unroll
                                for (int bx = LB; bx >= 0; --bx)
#pragma unroll
                                    for (int by = LB - bx; by >= 0; --by) {
                                        const int bz = LB - bx - by;
                                        double sm = 0.0;
#pragma unroll
                                        for (int t = 0; t <= ax + bx; ++t)
#pragma unroll
                                            for (int u = 0; u <= ay + by; ++u)
#pragma unroll
                                                for (int v = 0; v <= az + bz; ++v)
                                                    sm += ex.get(ax, bx, t) * ey.get(ay, by, u) * ez.get(az, bz, v) * G[hidx(t, u, v)];
                                        acc[(iab * NS + (s - S0)) * stride] += pref * sm;
                                        ++iab;
                                    }
                            }
                    }
                }
            }
        }
    }
    // bra indices to solid harmonics, hand every finished value to the sink
#pragma unroll
    for (int s = S0; s < S1; ++s) {
        const int mc = s / NSD, md = s - mc * NSD;
#pragma unroll
        for (int i = 0; i < NSA; ++i)
#pragma unroll
            for (int j = 0; j < NSB; ++j) {
                double v = 0.0;
#pragma unroll
                for (int ia = 0; ia < NCA; ++ia) {
                    const double wa = c2s_coef<LA>(c2s, i, ia);
                    if (wa == 0.0) continue;
#pragma unroll
                    for (int ib = 0; ib < NCB; ++ib) {
                        const double wb = c2s_coef<LB>(c2s, j, ib);
                        if (wb == 0.0) continue;
                        v += (wa * wb) * acc[((ia * NCB + ib) * NS + (s - S0)) * stride];
                    }
                }
                sink(i, j, mc, md, v);
            }
    }
}

template <int LA, int LB, int LC, int LD, int CH, int PASS, class Bra, class Ket, class Sink>
MQC_HD void eri_passes_src(const Bra& bra, const Ket& ket,
                           const double* __restrict__ boys_table, const double* __restrict__ c2s,
                           double* __restrict__ acc, int stride, Sink& sink)
{
    constexpr int NPASS = (nsph(LC) * nsph(LD) + CH - 1) / CH;
    eri_pass<LA, LB, LC, LD, CH, PASS>(bra, ket, boys_table, c2s, acc, stride, sink);
    if constexpr (PASS + 1 < NPASS) eri_passes_src<LA, LB, LC, LD, CH, PASS + 1>(bra, ket, boys_table, c2s, acc, stride, sink);
}

template <int LA, int LB, int LC, int LD, int CH, int PASS, class Sink>
MQC_HD void eri_passes_from(const ShellRef& A, const ShellRef& B, const ShellRef& C, const ShellRef& D,
                            const double* __restrict__ boys_table, const double* __restrict__ c2s,
                            double* __restrict__ acc, int stride, Sink& sink)
{
    const PairFly bra(A, B), ket(C, D);
    eri_passes_src<LA, LB, LC, LD, CH, PASS>(bra, ket, boys_table, c2s, acc, stride, sink);
}

// ket components per pass: as many as keep the accumulator slab at or under 64 entries per lane
MQC_HD constexpr int eri_pass_chunk(int la, int lb, int lc, int ld)
{
    const int nab = ncart(la) * ncart(lb), ncd = nsph(lc) * nsph(ld);
    int ch = 64 / nab;
    if (ch < 1) ch = 1;
    if (nab == 36 && ncd > 1) ch = 2;      // (dd| bra: two components per pass (72 accumulators)
    if (ch > ncd) ch = ncd;
    return ch;
}

// Schwarz diagonal blocks (ab|ab): the one-shot kernel keeps a (dd|dd) block in scratch (4 KB per lane, 4.7 ms for
// 6 k threads), so the d classes use passes too; their instantiations sit in translation units of their own
MQC_HD constexpr bool schwarz_uses_passes(int la, int lb)
{
    return (la == 1 && lb == 1) || la == 2;
}

// classes that go through the pass kernel (everything that spilled in the one-shot register kernel)
MQC_HD constexpr bool eri_uses_passes(int la, int lb, int lc, int ld)
{
    return ncart(la) * ncart(lb) * ncart(lc) * ncart(ld) > 30 && (la + lb + lc + ld) >= 4;
}

}  // namespace mqc
