// kern_xc.hip -- exchange-correlation quadrature on the Becke-partitioned Lebedev/Treutler grid.
//
// Replaces cuestMolecularGridCreate + cuestXCPotentialRKSCompute as called from
// backends/cuest/backend/mqc_cuest_integrals.f90:917-953,1368-1421; numerics follow the CPU path
// the goldens were produced with (backends/libcint/mqc_libcint_xc.F90:796-927, :1379-1455;
// backends/libcint/mqc_libcint_ao.f90:69-448; src/methods/mqc_dft_partition.f90:82-179,324-375):
//     rho = rowdot(chi D, chi), grad rho = 2 rowdot(chi D, grad chi), sigma = |grad rho|^2
//     E_xc = sum w f(rho, sigma),  V += A + A^T,
//     A = (w (v_rho/2 chi + 2 v_sigma grad rho . grad chi))^T chi
// The functionals are this repo's own implementation of the published closed forms libxc 7.1.2
// evaluates (lda_x, lda_c_vwn, lda_c_vwn_rpa, gga_x_b88, gga_c_lyp, gga_x_pbe, gga_c_pbe with
// lda_c_pw_mod), unpolarised, with first derivatives by forward-mode dual numbers.
//
// Mapping: a workgroup walks tiles of PT grid points of ONE fragment.  Per tile: AO values (and
// gradients) for all n functions go to LDS as [function][point]; X = D chi is formed from LDS;
// the density, the functional and the per-point coefficient vector a follow; the n x n update
// A += a chi^T accumulates in REGISTERS across all tiles of the workgroup and is flushed once.
#include "engine.hpp"
#include "md_integrals.hpp"
#include <cstdlib>
#include <string>

namespace mqc {

constexpr double XC_DENS_THRESHOLD = 1.0e-20;
constexpr double XC_EXP_CUTOFF = 46.0;

// ------------------------------------------------------------------ dual numbers (value, d/drho, d/dsigma)
struct Dual {
    double v, r, s;
};
__device__ __forceinline__ Dual mk(double v) { return {v, 0.0, 0.0}; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.r + b.r, a.s + b.s}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.r - b.r, a.s - b.s}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.r, -a.s}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.r * b.v + a.v * b.r, a.s * b.v + a.v * b.s}; }
// hardware-seeded reciprocal and rsqrt + two Newton steps (<= 1 ulp): the functionals are division-heavy and are
// evaluated once per wave and 16 points, so their instruction count is paid in full
__device__ __forceinline__ double xc_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double xc_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    y = fma(y, fma(-hx * y, y, 0.5), y);
    return y;
}
__device__ __forceinline__ Dual operator/(Dual a, Dual b)
{
    const double inv = xc_rcp(b.v), q = a.v * inv;
    return {q, (a.r - q * b.r) * inv, (a.s - q * b.s) * inv};
}
__device__ __forceinline__ Dual operator+(Dual a, double b) { return {a.v + b, a.r, a.s}; }
__device__ __forceinline__ Dual operator+(double b, Dual a) { return {a.v + b, a.r, a.s}; }
__device__ __forceinline__ Dual operator-(Dual a, double b) { return {a.v - b, a.r, a.s}; }
__device__ __forceinline__ Dual operator-(double b, Dual a) { return {b - a.v, -a.r, -a.s}; }
__device__ __forceinline__ Dual operator*(Dual a, double b) { return {a.v * b, a.r * b, a.s * b}; }
__device__ __forceinline__ Dual operator*(double b, Dual a) { return {a.v * b, a.r * b, a.s * b}; }
__device__ __forceinline__ Dual operator/(Dual a, double b) { const double i = xc_rcp(b); return {a.v * i, a.r * i, a.s * i}; }
__device__ __forceinline__ Dual operator/(double b, Dual a) { return mk(b) / a; }
__device__ __forceinline__ Dual chain(Dual x, double f, double df) { return {f, df * x.r, df * x.s}; }
#if defined(XC_MATH_PROBE)      /* measurement only: single-precision transcendentals, wrong results */
#define XC_EXP(x) ((double)__expf((float)(x)))
#define XC_LOG(x) ((double)__logf((float)(x)))
#define XC_ATAN(x) ((double)atanf((float)(x)))
#define XC_ASINH(x) ((double)__logf((float)(x) + sqrtf((float)(x) * (float)(x) + 1.0f)))
#define XC_CBRT(x) ((double)__powf((float)(x), 0.33333334f))
#else
#define XC_EXP(x) exp(x)
#define XC_LOG(x) log(x)
#define XC_ATAN(x) atan(x)
#define XC_ASINH(x) asinh(x)
#define XC_CBRT(x) cbrt(x)
#endif
__device__ __forceinline__ Dual dexp(Dual x) { const double f = XC_EXP(x.v); return chain(x, f, f); }
__device__ __forceinline__ Dual dlog(Dual x) { return chain(x, XC_LOG(x.v), xc_rcp(x.v)); }
__device__ __forceinline__ Dual dsqrt(Dual x) { const double i = xc_rsqrt(x.v); return chain(x, x.v * i, 0.5 * i); }
__device__ __forceinline__ Dual datan(Dual x) { return chain(x, XC_ATAN(x.v), xc_rcp(1.0 + x.v * x.v)); }
__device__ __forceinline__ Dual dasinh(Dual x) { return chain(x, XC_ASINH(x.v), xc_rsqrt(1.0 + x.v * x.v)); }
__device__ __forceinline__ Dual dpow(Dual x, double p) { const double f = pow(x.v, p); return chain(x, f, p * f / x.v); }
__device__ __forceinline__ Dual dcbrt(Dual x) { const double f = XC_CBRT(x.v); return chain(x, f, f * xc_rcp(3.0 * x.v)); }

// ------------------------------------------------------------------ functionals: energy per volume
__device__ __forceinline__ Dual f_lda_x(Dual rho, Dual r13)
{
    return -0.7385587663820224 * (rho * r13);                   // -(3/4)(3/pi)^(1/3) rho^(4/3)
}

__device__ __forceinline__ Dual f_vwn(Dual rho, Dual r13, double A, double x0, double b, double c)
{
    const Dual rs = 0.6203504908994001 / r13;            // (3/(4 pi))^(1/3) rho^(-1/3)
    const Dual x = dsqrt(rs);
    const Dual X = x * x + b * x + c;
    const double X0 = x0 * x0 + b * x0 + c;
    const double Q = sqrt(4.0 * c - b * b);
    const Dual at = datan(Q / (2.0 * x + b));
    const Dual xm = x - x0;
    const Dual ec = A * (dlog(x * x / X) + (2.0 * b / Q) * at
                         - (b * x0 / X0) * (dlog(xm * xm / X) + (2.0 * (b + 2.0 * x0) / Q) * at));
    return rho * ec;
}

__device__ __forceinline__ Dual f_b88(Dual rho, Dual r13, Dual sigma)
{
    const double beta = 0.0042, cx = 0.9305257363491;             // (3/2)(3/(4 pi))^(1/3)
    const Dual rh = 0.5 * rho;
    const Dual r43 = rh * (0.7937005259840998 * r13);            // cbrt(rho/2) = 2^(-1/3) cbrt(rho)
    const Dual x = dsqrt(0.25 * sigma) / r43;
    const Dual e = -cx * r43 - beta * r43 * x * x / (1.0 + 6.0 * beta * x * dasinh(x));
    return 2.0 * e;
}

__device__ __forceinline__ Dual f_lyp(Dual rho, Dual r13, Dual sigma)
{
    const double a = 0.04918, b = 0.132, c = 0.2533, d = 0.349;
    const double cf = 2.871234000188191;                          // (3/10)(3 pi^2)^(2/3)
    const Dual rm13 = 1.0 / r13;
    const Dual den = 1.0 + d * rm13;
    const Dual delta = c * rm13 + d * rm13 / den;
    const Dual rm53 = rm13 * rm13 * rm13 * rm13 * rm13;
    return -a * rho / den - a * b * dexp(-c * rm13) / den * (cf * rho - rm53 * sigma * ((3.0 + 7.0 * delta) * (1.0 / 72.0)));
}

constexpr double PBE_BETA = 0.06672455060314922;
constexpr double PBE_GAMMA = 0.031090690869654895;                // (1 - ln 2)/pi^2
constexpr double PBE_MU = 0.2195149727645171;
constexpr double PBE_KAPPA = 0.804;

__device__ __forceinline__ Dual f_pbe_x(Dual rho, Dual r13, Dual sigma)
{
    const Dual kf = 3.0936677262801355 * r13;                     // (3 pi^2)^(1/3) rho^(1/3)
    const Dual s2 = sigma / (4.0 * kf * kf * rho * rho);
    const Dual fx = (1.0 + PBE_KAPPA) - PBE_KAPPA / (1.0 + (PBE_MU / PBE_KAPPA) * s2);
    return f_lda_x(rho, r13) * fx;
}

__device__ __forceinline__ Dual f_pbe_c(Dual rho, Dual r13, Dual sigma)
{
    const double A = 0.0310907, a1 = 0.21370, b1 = 7.5957, b2 = 3.5876, b3 = 1.6382, b4 = 0.49294;   // lda_c_pw_mod
    const Dual rs = 0.6203504908994001 / r13;
    const Dual srs = dsqrt(rs);
    const Dual q = 2.0 * A * (b1 * srs + b2 * rs + b3 * rs * srs + b4 * rs * rs);
    const Dual ec = -2.0 * A * (1.0 + a1 * rs) * dlog(1.0 + 1.0 / q);
    const Dual kf = 3.0936677262801355 * r13;
    const Dual ks2 = (4.0 / M_PI) * kf;
    const Dual t2 = sigma / (4.0 * ks2 * rho * rho);
    const Dual Aa = (PBE_BETA / PBE_GAMMA) / (dexp(-ec / PBE_GAMMA) - 1.0);
    const Dual at2 = Aa * t2;
    const Dual H = PBE_GAMMA * dlog(1.0 + (PBE_BETA / PBE_GAMMA) * t2 * (1.0 + at2) / (1.0 + at2 + at2 * at2));
    return rho * (ec + H);
}

// f = rho * eps_xc per volume and its derivatives, zero below the density threshold; components k0, k0 + kstep, ...
// (the tiled kernel deals the components of a functional to its waves: B3LYP's four run side by side)
__device__ __forceinline__ void eval_functional(const XcSpec& xc, double rho, double sigma, double& f, double& vr, double& vs,
                                                int k0 = 0, int kstep = 1)
{
    f = 0.0; vr = 0.0; vs = 0.0;
    if (!(rho > XC_DENS_THRESHOLD) || k0 >= xc.ncomp) return;
    const Dual R = {rho, 1.0, 0.0};
    const Dual S_all = {fmax(sigma, 1.0e-40), 0.0, 1.0};
    const Dual R13_all = dcbrt(R);          // every component needs rho^(1/3): formed once
    for (int k = k0; k < xc.ncomp; k += kstep) {
        Dual d;
        // The cases are pure functions of loop invariants: left alone, the compiler hoists ALL SEVEN functionals out of the
        // loop (speculative execution) and the loop only selects among the results -- every tile paid for PBE, VWN5 and
        // the rest whatever the functional was.  An opaque copy of rho per iteration keeps the switch a real branch.
        double rv = R.v, cv = R13_all.v, cr = R13_all.r, sv = S_all.v;
        asm volatile("" : "+v"(rv), "+v"(cv), "+v"(cr), "+v"(sv));
        const Dual Rk = {rv, 1.0, 0.0}, R13 = {cv, cr, 0.0}, S = {sv, 0.0, 1.0};
        switch (xc.id[k]) {
            case XC_LDA_X: d = f_lda_x(Rk, R13); break;
            case XC_LDA_C_VWN: d = f_vwn(Rk, R13, 0.0310907, -0.10498, 3.72744, 12.9352); break;
            case XC_LDA_C_VWN_RPA: d = f_vwn(Rk, R13, 0.0310907, -0.409286, 13.0720, 42.7198); break;
            case XC_GGA_X_B88: d = f_b88(Rk, R13, S); break;
            case XC_GGA_C_LYP: d = f_lyp(Rk, R13, S); break;
            case XC_GGA_X_PBE: d = f_pbe_x(Rk, R13, S); break;
            case XC_GGA_C_PBE: d = f_pbe_c(Rk, R13, S); break;
            default: d = mk(0.0);
        }
        f += xc.w[k] * d.v; vr += xc.w[k] * d.r; vs += xc.w[k] * d.s;
    }
}

// ------------------------------------------------------------------ spin-polarised forms (unrestricted Kohn-Sham)
// value + derivatives with respect to (rho_a, rho_b, sigma_aa, sigma_ab, sigma_bb): the layout of libxc's polarised
// calls as xc_add_potential_uks feeds them (mqc_libcint_xc.F90:938-951).  Closed forms: the spin-scaling relations
// of exchange, VWN's zeta interpolations, Miehlich's gradient-only LYP for two spin densities, PBE correlation with
// phi(zeta) and the spin-polarised PW92 ("pw_mod") local part.
struct D5 {
    double v, d[5];
};
__device__ __forceinline__ D5 mk5(double v) { return {v, {0.0, 0.0, 0.0, 0.0, 0.0}}; }
__device__ __forceinline__ D5 var5(double v, int i) { D5 r = mk5(v); r.d[i] = 1.0; return r; }
__device__ __forceinline__ D5 operator+(D5 a, D5 b) { D5 r; r.v = a.v + b.v; for (int i = 0; i < 5; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ D5 operator-(D5 a, D5 b) { D5 r; r.v = a.v - b.v; for (int i = 0; i < 5; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ D5 operator-(D5 a) { D5 r; r.v = -a.v; for (int i = 0; i < 5; ++i) r.d[i] = -a.d[i]; return r; }
__device__ __forceinline__ D5 operator*(D5 a, D5 b) { D5 r; r.v = a.v * b.v; for (int i = 0; i < 5; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ D5 operator/(D5 a, D5 b)
{
    const double inv = xc_rcp(b.v), q = a.v * inv;
    D5 r; r.v = q;
    for (int i = 0; i < 5; ++i) r.d[i] = (a.d[i] - q * b.d[i]) * inv;
    return r;
}
__device__ __forceinline__ D5 operator+(D5 a, double b) { a.v += b; return a; }
__device__ __forceinline__ D5 operator+(double b, D5 a) { a.v += b; return a; }
__device__ __forceinline__ D5 operator-(D5 a, double b) { a.v -= b; return a; }
__device__ __forceinline__ D5 operator-(double b, D5 a) { return mk5(b) - a; }
__device__ __forceinline__ D5 operator*(D5 a, double b) { a.v *= b; for (int i = 0; i < 5; ++i) a.d[i] *= b; return a; }
__device__ __forceinline__ D5 operator*(double b, D5 a) { return a * b; }
__device__ __forceinline__ D5 operator/(D5 a, double b) { return a * xc_rcp(b); }
__device__ __forceinline__ D5 operator/(double b, D5 a) { return mk5(b) / a; }
__device__ __forceinline__ D5 chain5(D5 x, double f, double df) { D5 r; r.v = f; for (int i = 0; i < 5; ++i) r.d[i] = df * x.d[i]; return r; }
__device__ __forceinline__ D5 exp5(D5 x) { const double f = exp(x.v); return chain5(x, f, f); }
__device__ __forceinline__ D5 log5(D5 x) { return chain5(x, log(x.v), xc_rcp(x.v)); }
__device__ __forceinline__ D5 sqrt5(D5 x) { const double i = xc_rsqrt(x.v); return chain5(x, x.v * i, 0.5 * i); }
__device__ __forceinline__ D5 atan5(D5 x) { return chain5(x, atan(x.v), xc_rcp(1.0 + x.v * x.v)); }
__device__ __forceinline__ D5 asinh5(D5 x) { return chain5(x, asinh(x.v), xc_rsqrt(1.0 + x.v * x.v)); }
__device__ __forceinline__ D5 pow5(D5 x, double p) { const double f = pow(x.v, p); return chain5(x, f, p * f * xc_rcp(x.v)); }
__device__ __forceinline__ D5 cbrt5(D5 x) { const double f = cbrt(x.v); return chain5(x, f, f * xc_rcp(3.0 * x.v)); }

__device__ __forceinline__ D5 zeta_f5(D5 opz13, D5 omz13, D5 z)
{
    // f(zeta) = [(1+z)^(4/3) + (1-z)^(4/3) - 2] / (2^(4/3) - 2), from the cube roots of 1 +- zeta
    return ((1.0 + z) * opz13 + (1.0 - z) * omz13 - 2.0) * (1.0 / 0.5198420997897464);
}
__device__ __forceinline__ D5 vwn_aux5(D5 x /* sqrt(rs) */, double A, double x0, double b, double c)
{
    const D5 X = x * x + b * x + c;
    const double X0 = x0 * x0 + b * x0 + c;
    const double Q = sqrt(4.0 * c - b * b);
    const D5 at = atan5(Q / (2.0 * x + b));
    const D5 xm = x - x0;
    return A * (log5(x * x / X) + (2.0 * b / Q) * at - (b * x0 / X0) * (log5(xm * xm / X) + (2.0 * (b + 2.0 * x0) / Q) * at));
}
__device__ __forceinline__ D5 b88_spin5(D5 r, D5 s)
{
    const double beta = 0.0042, cx = 0.9305257363491;
    const D5 r43 = r * cbrt5(r);
    const D5 x = sqrt5(s) / r43;
    return -cx * r43 - beta * r43 * x * x / (1.0 + 6.0 * beta * x * asinh5(x));
}
__device__ __forceinline__ D5 pbe_x_unpol5(D5 rho, D5 sigma)
{
    const D5 r13 = cbrt5(rho);
    const D5 kf = 3.0936677262801355 * r13;
    const D5 s2 = sigma / (4.0 * kf * kf * rho * rho);
    const D5 fx = (1.0 + PBE_KAPPA) - PBE_KAPPA / (1.0 + (PBE_MU / PBE_KAPPA) * s2);
    return -0.7385587663820224 * (rho * r13) * fx;
}
__device__ __forceinline__ D5 pw_mod_g5(D5 rs, D5 srs, double A, double a1, double b1, double b2, double b3, double b4)
{
    const D5 q = 2.0 * A * (b1 * srs + b2 * rs + b3 * rs * srs + b4 * rs * rs);
    return -2.0 * A * (1.0 + a1 * rs) * log5(1.0 + 1.0 / q);
}

constexpr double XC_SPIN_FLOOR = 1.0e-30;        // a spin density below this is held there: zeta stays inside (-1, 1)

// f per volume and its five derivatives; zero where the TOTAL density is below the threshold
__device__ __forceinline__ void eval_functional_pol(const XcSpec& xc, double ra_in, double rb_in, double saa, double sab, double sbb,
                                                    double& f, double* dv)
{
    f = 0.0;
    for (int i = 0; i < 5; ++i) dv[i] = 0.0;
    if (!(ra_in + rb_in > XC_DENS_THRESHOLD)) return;
    const D5 ra_all = var5(fmax(ra_in, XC_SPIN_FLOOR), 0), rb_all = var5(fmax(rb_in, XC_SPIN_FLOOR), 1);
    const D5 Saa = var5(fmax(saa, 1.0e-40), 2), Sab = var5(sab, 3), Sbb = var5(fmax(sbb, 1.0e-40), 4);
    const D5 rho_all = ra_all + rb_all;
    const D5 r13_all = cbrt5(rho_all);
    const D5 z = (ra_all - rb_all) / rho_all;
    const D5 opz13 = cbrt5(1.0 + z), omz13 = cbrt5(1.0 - z);
    const D5 rs_all = 0.6203504908994001 / r13_all;
    const D5 sig_all = Saa + 2.0 * Sab + Sbb;
    for (int k = 0; k < xc.ncomp; ++k) {
        D5 d;
        // opaque per iteration, so that the cases stay behind their branch instead of being hoisted out of the loop
        // all seven at once (see eval_functional)
        D5 rho = rho_all, ra = ra_all, rb = rb_all, r13 = r13_all, rs = rs_all, sig = sig_all;
        asm volatile("" : "+v"(rho.v), "+v"(ra.v), "+v"(rb.v), "+v"(r13.v), "+v"(rs.v), "+v"(sig.v));
        switch (xc.id[k]) {
            case XC_LDA_X:
                d = -0.7385587663820224 * 1.2599210498948732 * (ra * cbrt5(ra) + rb * cbrt5(rb));
                break;
            case XC_LDA_C_VWN: {
                const D5 x = sqrt5(rs);
                const D5 eP = vwn_aux5(x, 0.0310907, -0.10498, 3.72744, 12.9352);
                const D5 eF = vwn_aux5(x, 0.01554535, -0.32500, 7.06042, 18.0578);
                const D5 aC = vwn_aux5(x, -0.016886863940389628 /* -1/(6 pi^2) */, -0.0047584, 1.13107, 13.0045);
                const D5 fz = zeta_f5(opz13, omz13, z);
                const D5 z2 = z * z, z4 = z2 * z2;
                d = rho * (eP + aC * fz * (1.0 - z4) * (1.0 / 1.7099209341613653) + (eF - eP) * fz * z4);
                break;
            }
            case XC_LDA_C_VWN_RPA: {
                const D5 x = sqrt5(rs);
                const D5 eP = vwn_aux5(x, 0.0310907, -0.409286, 13.0720, 42.7198);
                const D5 eF = vwn_aux5(x, 0.01554535, -0.743294, 20.1231, 101.578);
                const D5 fz = zeta_f5(opz13, omz13, z);
                d = rho * (eP * (1.0 - fz) + eF * fz);
                break;
            }
            case XC_GGA_X_B88: d = b88_spin5(ra, Saa) + b88_spin5(rb, Sbb); break;
            case XC_GGA_C_LYP: {
                const double a = 0.04918, b = 0.132, c = 0.2533, dd = 0.349, cf = 2.871234000188191;
                const D5 rm13 = 1.0 / r13;
                const D5 den = 1.0 + dd * rm13;
                const D5 rm113 = rm13 * rm13 / (rho * rho * rho);              // rho^(-11/3)
                const D5 omega = exp5(-c * rm13) / den * rm113;
                const D5 delta = c * rm13 + dd * rm13 / den;
                const D5 rab = ra * rb;
                const D5 ra13 = cbrt5(ra), rb13 = cbrt5(rb);
                const D5 ra83 = ra * ra * ra13 * ra13, rb83 = rb * rb * rb13 * rb13;
                const D5 t1 = (12.699208415745595 * cf) * (ra83 + rb83);        // 2^(11/3) C_F (ra^(8/3) + rb^(8/3))
                const D5 t2 = (47.0 / 18.0 - (7.0 / 18.0) * delta) * sig;
                const D5 t3 = (2.5 - delta * (1.0 / 18.0)) * (Saa + Sbb);
                const D5 t4 = (delta - 11.0) * (1.0 / 9.0) * (ra * Saa + rb * Sbb) / rho;
                const D5 r2 = (2.0 / 3.0) * rho * rho;
                const D5 br = rab * (t1 + t2 - t3 - t4) - r2 * sig + (r2 - ra * ra) * Sbb + (r2 - rb * rb) * Saa;
                d = -4.0 * a / den * rab / rho - (a * b) * omega * br;
                break;
            }
            case XC_GGA_X_PBE: d = 0.5 * (pbe_x_unpol5(2.0 * ra, 4.0 * Saa) + pbe_x_unpol5(2.0 * rb, 4.0 * Sbb)); break;
            case XC_GGA_C_PBE: {
                const D5 srs = sqrt5(rs);
                const D5 g0 = pw_mod_g5(rs, srs, 0.0310907, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294);
                const D5 g1 = pw_mod_g5(rs, srs, 0.01554535, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517);
                const D5 g2 = pw_mod_g5(rs, srs, 0.0168869, 0.11125, 10.357, 3.6231, 0.88026, 0.49671);       // = -alpha_c
                const D5 fz = zeta_f5(opz13, omz13, z);
                const D5 z2 = z * z, z4 = z2 * z2;
                const D5 ec = g0 - g2 * fz * (1.0 - z4) * (1.0 / 1.709920934161365617563962776245) + (g1 - g0) * fz * z4;
                const D5 phi = 0.5 * (opz13 * opz13 + omz13 * omz13);
                const D5 phi3 = phi * phi * phi;
                const D5 kf = 3.0936677262801355 * r13;
                const D5 ks2 = (4.0 / M_PI) * kf;
                const D5 t2 = sig / (4.0 * phi * phi * ks2 * rho * rho);
                const D5 Aa = (PBE_BETA / PBE_GAMMA) / (exp5(-ec / (PBE_GAMMA * phi3)) - 1.0);
                const D5 at2 = Aa * t2;
                const D5 H = PBE_GAMMA * phi3 * log5(1.0 + (PBE_BETA / PBE_GAMMA) * t2 * (1.0 + at2) / (1.0 + at2 + at2 * at2));
                d = rho * (ec + H);
                break;
            }
            default: d = mk5(0.0);
        }
        f += xc.w[k] * d.v;
        for (int i = 0; i < 5; ++i) dv[i] += xc.w[k] * d.d[i];
    }
}


// ------------------------------------------------------------------ meta-GGA: value + N first derivatives
// N = 3: (rho, sigma, tau), restricted.  N = 7: (rho_a, rho_b, sigma_aa, sigma_ab, sigma_bb, tau_a, tau_b), unrestricted.
template <int N>
struct DualN {
    double v, d[N];
};
template <int N> __device__ __forceinline__ DualN<N> mkn(double v) { DualN<N> r; r.v = v; for (int i = 0; i < N; ++i) r.d[i] = 0.0; return r; }
template <int N> __device__ __forceinline__ DualN<N> varn(double v, int k) { DualN<N> r = mkn<N>(v); r.d[k] = 1.0; return r; }
template <int N> __device__ __forceinline__ DualN<N> operator+(DualN<N> a, DualN<N> b) { DualN<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> __device__ __forceinline__ DualN<N> operator-(DualN<N> a, DualN<N> b) { DualN<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> __device__ __forceinline__ DualN<N> operator*(DualN<N> a, DualN<N> b) { DualN<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> __device__ __forceinline__ DualN<N> operator/(DualN<N> a, DualN<N> b)
{
    const double inv = 1.0 / b.v, q = a.v * inv;
    DualN<N> r; r.v = q;
    for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - q * b.d[i]) * inv;
    return r;
}
template <int N> __device__ __forceinline__ DualN<N> operator+(DualN<N> a, double b) { a.v += b; return a; }
template <int N> __device__ __forceinline__ DualN<N> operator+(double b, DualN<N> a) { a.v += b; return a; }
template <int N> __device__ __forceinline__ DualN<N> operator-(DualN<N> a, double b) { a.v -= b; return a; }
template <int N> __device__ __forceinline__ DualN<N> operator-(double b, DualN<N> a) { return mkn<N>(b) - a; }
template <int N> __device__ __forceinline__ DualN<N> operator*(DualN<N> a, double b) { a.v *= b; for (int i = 0; i < N; ++i) a.d[i] *= b; return a; }
template <int N> __device__ __forceinline__ DualN<N> operator*(double b, DualN<N> a) { return a * b; }
template <int N> __device__ __forceinline__ DualN<N> operator/(DualN<N> a, double b) { return a * (1.0 / b); }
template <int N> __device__ __forceinline__ DualN<N> operator/(double b, DualN<N> a) { return mkn<N>(b) / a; }
template <int N> __device__ __forceinline__ DualN<N> chainn(DualN<N> x, double f, double df) { DualN<N> r; r.v = f; for (int i = 0; i < N; ++i) r.d[i] = df * x.d[i]; return r; }
template <int N> __device__ __forceinline__ DualN<N> expn(DualN<N> x) { const double f = exp(x.v); return chainn(x, f, f); }
template <int N> __device__ __forceinline__ DualN<N> logn(DualN<N> x) { return chainn(x, log(x.v), 1.0 / x.v); }
template <int N> __device__ __forceinline__ DualN<N> sqrtn(DualN<N> x) { const double f = sqrt(x.v); return chainn(x, f, 0.5 / f); }
template <int N> __device__ __forceinline__ DualN<N> cbrtn(DualN<N> x) { const double f = cbrt(x.v); return chainn(x, f, f / (3.0 * x.v)); }
template <int N> __device__ __forceinline__ DualN<N> maxn(DualN<N> a, DualN<N> b) { return a.v > b.v ? a : b; }

template <int N> __device__ __forceinline__ DualN<N> pw_mod_gn(DualN<N> rs, DualN<N> srs, double A, double a1, double b1, double b2, double b3, double b4)
{
    const DualN<N> q = 2.0 * A * (b1 * srs + b2 * rs + b3 * rs * srs + b4 * rs * rs);
    return -2.0 * A * (1.0 + a1 * rs) * logn(1.0 + 1.0 / q);
}

// the gradient correction H of PBE correlation on top of a uniform-gas energy ec, for a spin-scaling factor phi
template <int N> __device__ __forceinline__ DualN<N> pbe_c_h(DualN<N> ec, DualN<N> phi, DualN<N> r13, DualN<N> rho, DualN<N> sigma)
{
    const DualN<N> phi3 = phi * phi * phi;
    const DualN<N> ks2 = (4.0 / M_PI) * 3.0936677262801355 * r13;
    const DualN<N> t2 = sigma / (4.0 * phi * phi * ks2 * rho * rho);
    const DualN<N> Aa = (PBE_BETA / PBE_GAMMA) / (expn(-1.0 * ec / (PBE_GAMMA * phi3)) - 1.0);
    const DualN<N> at2 = Aa * t2;
    return PBE_GAMMA * phi3 * logn(1.0 + (PBE_BETA / PBE_GAMMA) * t2 * (1.0 + at2) / (1.0 + at2 + at2 * at2));
}

// PBE correlation energy per particle at zeta = 0 or zeta = 1 (lda_c_pw_mod inside), the two ends TPSS needs
template <int N> __device__ __forceinline__ DualN<N> pbe_c_eps_fixed_zeta(DualN<N> rho, DualN<N> sigma, bool ferro)
{
    const DualN<N> r13 = cbrtn(rho);
    const DualN<N> rs = 0.6203504908994001 / r13;
    const DualN<N> srs = sqrtn(rs);
    const DualN<N> ec = ferro ? pw_mod_gn(rs, srs, 0.01554535, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517)
                              : pw_mod_gn(rs, srs, 0.0310907, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294);
    return ec + pbe_c_h(ec, mkn<N>(ferro ? 0.7937005259840998 : 1.0), r13, rho, sigma);
}

// PBE correlation energy per particle for two spin densities (gga_c_pbe with lda_c_pw_mod, as eval_functional_pol)
template <int N> __device__ __forceinline__ DualN<N> pbe_c_eps_pol(DualN<N> rho, DualN<N> z, DualN<N> sigma)
{
    const DualN<N> r13 = cbrtn(rho);
    const DualN<N> rs = 0.6203504908994001 / r13;
    const DualN<N> srs = sqrtn(rs);
    const DualN<N> g0 = pw_mod_gn(rs, srs, 0.0310907, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294);
    const DualN<N> g1 = pw_mod_gn(rs, srs, 0.01554535, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517);
    const DualN<N> g2 = pw_mod_gn(rs, srs, 0.0168869, 0.11125, 10.357, 3.6231, 0.88026, 0.49671);       // = -alpha_c
    const DualN<N> opz13 = cbrtn(1.0 + z), omz13 = cbrtn(1.0 - z);
    const DualN<N> fz = ((1.0 + z) * opz13 + (1.0 - z) * omz13 - 2.0) * (1.0 / 0.5198420997897464);
    const DualN<N> z2 = z * z, z4 = z2 * z2;
    const DualN<N> ec = g0 - g2 * fz * (1.0 - z4) * (1.0 / 1.709920934161365617563962776245) + (g1 - g0) * fz * z4;
    const DualN<N> phi = 0.5 * (opz13 * opz13 + omz13 * omz13);
    return ec + pbe_c_h(ec, phi, r13, rho, sigma);
}

// libxc mgga_x_tpss for a spin-unpolarised density, energy per volume (PRL 91, 146401 eqs. 5-10)
template <int N> __device__ __forceinline__ DualN<N> tpss_x_unpol(DualN<N> rho, DualN<N> sigma, DualN<N> tau)
{
    const double b = 0.40, c = 1.59096, e = 1.537, kappa = 0.804, mu = 0.21951, mu_ge = 10.0 / 81.0, se = 1.2397580409095959;
    const DualN<N> r13 = cbrtn(rho);
    const DualN<N> r23 = r13 * r13;
    const DualN<N> z = sigma / (8.0 * rho * tau);
    const DualN<N> z2 = z * z;
    const DualN<N> p = sigma / ((4.0 * 9.570780000627305) * (r23 * rho * rho));            // 4 (3 pi^2)^(2/3) rho^(8/3)
    const DualN<N> tau_unif = (0.3 * 9.570780000627305) * (r23 * rho);
    const DualN<N> alpha = (tau - sigma / (8.0 * rho)) / tau_unif;
    const DualN<N> qb = 0.45 * (alpha - 1.0) / sqrtn(1.0 + b * alpha * (alpha - 1.0)) + (2.0 / 3.0) * p;
    const DualN<N> opz2 = 1.0 + z2;
    const DualN<N> num = (mu_ge + c * z2 / (opz2 * opz2)) * p + (146.0 / 2025.0) * qb * qb
                       - (73.0 / 405.0) * qb * sqrtn(0.5 * (0.36 * z2 + p * p)) + (mu_ge * mu_ge / kappa) * p * p
                       + (2.0 * se * mu_ge * 0.36) * z2 + (e * mu) * p * p * p;
    const DualN<N> den = (1.0 + se * p) * (1.0 + se * p);
    const DualN<N> x = num / den;
    const DualN<N> fx = (1.0 + kappa) - kappa / (1.0 + x / kappa);
    return -0.7385587663820224 * (rho * r13) * fx;
}

// libxc mgga_x_tpss + mgga_c_tpss, unpolarised (Tao, Perdew, Staroverov, Scuseria, PRL 91, 146401 eqs. 5-14);
// f per volume and d/d(rho, sigma, tau), zero below the density threshold.  mqc_xc_spec.f90:142-166 names the pair.
__device__ __forceinline__ void eval_functional_mgga(const XcSpec& xc, double rho_in, double sigma_in, double tau_in, double& f, double* dv)
{
    typedef DualN<3> D3;
    f = 0.0;
    dv[0] = dv[1] = dv[2] = 0.0;
    if (!(rho_in > XC_DENS_THRESHOLD)) return;
    const D3 rho = varn<3>(rho_in, 0), sigma = varn<3>(fmax(sigma_in, 1.0e-40), 1), tau = varn<3>(fmax(tau_in, 1.0e-20), 2);
    for (int k = 0; k < xc.ncomp; ++k) {
        D3 d = mkn<3>(0.0);
        if (xc.id[k] == XC_MGGA_X_TPSS) {
            d = tpss_x_unpol(rho, sigma, tau);
        } else if (xc.id[k] == XC_MGGA_C_TPSS) {
            const double dd = 2.8, C0 = 0.53;
            const D3 z = sigma / (8.0 * rho * tau);
            const D3 z2 = z * z;
            const D3 e_pbe = pbe_c_eps_fixed_zeta(rho, sigma, false);
            const D3 e_til = maxn(pbe_c_eps_fixed_zeta(0.5 * rho, 0.25 * sigma, true), e_pbe);
            const D3 e_rev = e_pbe * (1.0 + C0 * z2) - (1.0 + C0) * z2 * e_til;
            d = rho * e_rev * (1.0 + dd * e_rev * z2 * z);
        }
        f += xc.w[k] * d.v;
        for (int i = 0; i < 3; ++i) dv[i] += xc.w[k] * d.d[i];
    }
}

// the same pair for two spin densities: exchange by spin scaling, correlation with C(zeta, xi) (eq. 14);
// dv = d/d(rho_a, rho_b, sigma_aa, sigma_ab, sigma_bb, tau_a, tau_b)
__device__ __forceinline__ void eval_functional_mgga_pol(const XcSpec& xc, double ra_in, double rb_in, double saa, double sab, double sbb,
                                                         double ta_in, double tb_in, double& f, double* dv)
{
    typedef DualN<7> D7;
    f = 0.0;
    for (int i = 0; i < 7; ++i) dv[i] = 0.0;
    if (!(ra_in + rb_in > XC_DENS_THRESHOLD)) return;
    const D7 ra = varn<7>(fmax(ra_in, XC_SPIN_FLOOR), 0), rb = varn<7>(fmax(rb_in, XC_SPIN_FLOOR), 1);
    const D7 Saa = varn<7>(fmax(saa, 1.0e-40), 2), Sab = varn<7>(sab, 3), Sbb = varn<7>(fmax(sbb, 1.0e-40), 4);
    const D7 ta = varn<7>(fmax(ta_in, 1.0e-20), 5), tb = varn<7>(fmax(tb_in, 1.0e-20), 6);
    for (int k = 0; k < xc.ncomp; ++k) {
        D7 d = mkn<7>(0.0);
        if (xc.id[k] == XC_MGGA_X_TPSS) {
            d = 0.5 * (tpss_x_unpol(2.0 * ra, 4.0 * Saa, 2.0 * ta) + tpss_x_unpol(2.0 * rb, 4.0 * Sbb, 2.0 * tb));
        } else if (xc.id[k] == XC_MGGA_C_TPSS) {
            const double dd = 2.8;
            const D7 rho = ra + rb;
            const D7 zeta = (ra - rb) / rho;
            const D7 sig = Saa + 2.0 * Sab + Sbb;
            const D7 z = sig / (8.0 * rho * (ta + tb));
            const D7 z2 = z * z;
            const D7 e_pbe = pbe_c_eps_pol(rho, zeta, sig);
            const D7 et_a = maxn(pbe_c_eps_fixed_zeta(ra, Saa, true), e_pbe);
            const D7 et_b = maxn(pbe_c_eps_fixed_zeta(rb, Sbb, true), e_pbe);
            const D7 omz = 1.0 - zeta, opz = 1.0 + zeta;
            const D7 gz2 = (omz * omz * Saa - 2.0 * omz * opz * Sab + opz * opz * Sbb) / (rho * rho);       // |grad zeta|^2
            const D7 r13 = cbrtn(rho);
            const D7 xi2 = gz2 / ((4.0 * 9.570780000627305) * (r13 * r13));
            const D7 zeta2 = zeta * zeta;
            const D7 c0 = 0.53 + 0.87 * zeta2 + 0.50 * zeta2 * zeta2 + 2.26 * zeta2 * zeta2 * zeta2;
            const D7 opz13 = cbrtn(opz), omz13 = cbrtn(omz);
            const D7 den = 1.0 + 0.5 * xi2 * (1.0 / (opz * opz13) + 1.0 / (omz * omz13));
            const D7 den2 = den * den;
            const D7 C = c0 / (den2 * den2);
            const D7 e_rev = e_pbe * (1.0 + C * z2) - (1.0 + C) * z2 * (ra * et_a + rb * et_b) / rho;
            d = rho * e_rev * (1.0 + dd * e_rev * z2 * z);
        }
        f += xc.w[k] * d.v;
        for (int i = 0; i < 7; ++i) dv[i] += xc.w[k] * d.d[i];
    }
}

// ------------------------------------------------------------------ Becke / Treutler partition weights
__device__ __forceinline__ double becke_cutoff(double nu)
{
    double f = nu;
    f = 0.5 * f * (3.0 - f * f);
    f = 0.5 * f * (3.0 - f * f);
    f = 0.5 * f * (3.0 - f * f);
    return 0.5 * (1.0 - f);
}

constexpr int BECKE_MAX_ATOMS = 64;

__global__ void __launch_bounds__(256) becke_weights_kernel(BatchView bv)
{
    const int f = blockIdx.y;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const GridDev& gd = bv.grid;
    if (g >= gd.npts) return;
    const int na = bv.topo.natoms;
    const double* xyz = bv.xyz + (size_t)f * na * 3;
    const int owner = gd.pt_atom[g], it = gd.pt_tmpl[g];
    const double px = xyz[3 * owner] + gd.tmpl_xyz[3 * it], py = xyz[3 * owner + 1] + gd.tmpl_xyz[3 * it + 1],
                 pz = xyz[3 * owner + 2] + gd.tmpl_xyz[3 * it + 2];
    double w = gd.tmpl_w[it];
    if (na > 1) {
        double dist[BECKE_MAX_ATOMS], cell[BECKE_MAX_ATOMS];
        for (int i = 0; i < na; ++i) {
            const double dx = px - xyz[3 * i], dy = py - xyz[3 * i + 1], dz = pz - xyz[3 * i + 2];
            dist[i] = sqrt(dx * dx + dy * dy + dz * dz);
            cell[i] = 1.0;
        }
        for (int i = 0; i < na; ++i)
            for (int j = i + 1; j < na; ++j) {
                const double dx = xyz[3 * i] - xyz[3 * j], dy = xyz[3 * i + 1] - xyz[3 * j + 1], dz = xyz[3 * i + 2] - xyz[3 * j + 2];
                const double mu = (dist[i] - dist[j]) / sqrt(dx * dx + dy * dy + dz * dz);
                const double chi = gd.sqrt_bragg[i] / gd.sqrt_bragg[j];
                double a = 0.25 * (1.0 / chi - chi);
                a = fmax(-0.5, fmin(0.5, a));
                const double nu = mu + a * (1.0 - mu * mu);
                const double s = becke_cutoff(nu);
                cell[i] *= s;
                cell[j] *= (1.0 - s);
            }
        double tot = 0.0;
        for (int i = 0; i < na; ++i) tot += cell[i];
        w = tot > 0.0 ? w * cell[owner] / tot : 0.0;
    }
    gd.weights[(size_t)f * gd.npts + g] = w;
}

void launch_becke_weights(const BatchView& bv, hipStream_t s)
{
    dim3 grid((bv.grid.npts + 255) / 256, bv.nfrag), block(256);
    hipLaunchKernelGGL(becke_weights_kernel, grid, block, 0, s, bv);
}

// ------------------------------------------------------------------ the per-iteration quadrature kernel
constexpr int XC_NT = 256;

// angular part of one shell at one point from its radial value and derivative factor
template <bool GGA, bool LF = true>
__device__ __forceinline__ void emit_shell(int l, int ao, double dx, double dy, double dz, double rad, double drad,
                                           double* __restrict__ chi, double* __restrict__ gx, double* __restrict__ gy,
                                           double* __restrict__ gz, int ptp, int p, const double* __restrict__ c2s)
{
    if (l == 0) {
        chi[ao * ptp + p] = rad;
        if (GGA) { gx[ao * ptp + p] = dx * drad; gy[ao * ptp + p] = dy * drad; gz[ao * ptp + p] = dz * drad; }
    } else if (l == 1) {
        chi[ao * ptp + p] = dx * rad; chi[(ao + 1) * ptp + p] = dy * rad; chi[(ao + 2) * ptp + p] = dz * rad;
        if (GGA) {
            gx[ao * ptp + p] = rad + dx * dx * drad; gy[ao * ptp + p] = dx * dy * drad; gz[ao * ptp + p] = dx * dz * drad;
            gx[(ao + 1) * ptp + p] = dy * dx * drad; gy[(ao + 1) * ptp + p] = rad + dy * dy * drad; gz[(ao + 1) * ptp + p] = dy * dz * drad;
            gx[(ao + 2) * ptp + p] = dz * dx * drad; gy[(ao + 2) * ptp + p] = dz * dy * drad; gz[(ao + 2) * ptp + p] = rad + dz * dz * drad;
        }
    } else if (LF && l == 3) {
        // f shell, the seven real solid harmonics written out: value and gradient of the ten cubic monomials (libcint
        // order xxx, xxy, xxz, xyy, xyz, xzz, yyy, yyz, yzz, zzz), combined with the l = 3 rows of the c2s table
        // (host_setup.cpp: build_c2s_tables; zeros drop out at compile time).  The table-driven loop below takes
        // ~2 000 instructions and 70 global loads per item and was the longest part of a def2-TZVP slab.
        constexpr double T3[7][10] = {
            {0, 1.7701307697799307, 0, 0, 0, 0, -0.59004358992664352, 0, 0, 0},
            {0, 0, 0, 0, 2.8906114426405543, 0, 0, 0, 0, 0},
            {0, -0.45704579946446572, 0, 0, 0, 0, -0.45704579946446572, 0, 1.8281831978578629, 0},
            {0, 0, -1.1195289977703462, 0, 0, 0, 0, -1.1195289977703462, 0, 0.7463526651802308},
            {-0.45704579946446572, 0, 0, -0.45704579946446572, 0, 1.8281831978578629, 0, 0, 0, 0},
            {0, 0, 1.4453057213202771, 0, 0, 0, 0, -1.4453057213202771, 0, 0},
            {0.59004358992664352, 0, 0, -1.7701307697799307, 0, 0, 0, 0, 0, 0}};
        const double x2 = dx * dx, y2 = dy * dy, z2 = dz * dz, xy = dx * dy, xz = dx * dz, yz = dy * dz;
        const double cv[10] = {x2 * dx, x2 * dy, x2 * dz, dx * y2, xy * dz, dx * z2, y2 * dy, y2 * dz, dy * z2, z2 * dz};
        const double cgx[10] = {3.0 * x2, 2.0 * xy, 2.0 * xz, y2, yz, z2, 0.0, 0.0, 0.0, 0.0};
        const double cgy[10] = {0.0, x2, 0.0, 2.0 * xy, xz, 0.0, 3.0 * y2, 2.0 * yz, z2, 0.0};
        const double cgz[10] = {0.0, 0.0, x2, 0.0, xy, 2.0 * xz, 0.0, y2, 2.0 * yz, 3.0 * z2};
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            double v = 0.0, ax = 0.0, ay = 0.0, az = 0.0;
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                if (T3[m][k] != 0.0) { v += T3[m][k] * cv[k]; ax += T3[m][k] * cgx[k]; ay += T3[m][k] * cgy[k]; az += T3[m][k] * cgz[k]; }
            }
            chi[(ao + m) * ptp + p] = v * rad;
            if (GGA) {
                gx[(ao + m) * ptp + p] = ax * rad + v * dx * drad;
                gy[(ao + m) * ptp + p] = ay * rad + v * dy * drad;
                gz[(ao + m) * ptp + p] = az * rad + v * dz * drad;
            }
        }
    } else if (LF && l >= 4) {
        // l = 4: Cartesian monomials x^a y^b z^c in libcint order, transformed with the packed c2s table
        const int nc = ncart(l), nsp = nsph(l);
        auto ipow = [](double x, int k) { double r = 1.0; for (int i = 0; i < k; ++i) r *= x; return r; };   // k <= 4, no arrays (no scratch)
        const double* tab = c2s + c2s_table_offset(l);
        for (int m = 0; m < nsp; ++m) {
            double v = 0.0, ax = 0.0, ay = 0.0, az = 0.0;
            int k = 0;
            for (int a = l; a >= 0; --a)
                for (int b = l - a; b >= 0; --b, ++k) {
                    const int c = l - a - b;
                    const double w = tab[m * nc + k];
                    if (w == 0.0) continue;
                    const double xa = ipow(dx, a), yb = ipow(dy, b), zc = ipow(dz, c);
                    v += w * xa * yb * zc;
                    if (GGA) {
                        if (a) ax += w * a * ipow(dx, a - 1) * yb * zc;
                        if (b) ay += w * b * xa * ipow(dy, b - 1) * zc;
                        if (c) az += w * c * xa * yb * ipow(dz, c - 1);
                    }
                }
            chi[(ao + m) * ptp + p] = v * rad;
            if (GGA) {
                gx[(ao + m) * ptp + p] = ax * rad + v * dx * drad;
                gy[(ao + m) * ptp + p] = ay * rad + v * dy * drad;
                gz[(ao + m) * ptp + p] = az * rad + v * dz * drad;
            }
        }
    } else {
        // l = 2: Cartesian xx,xy,xz,yy,yz,zz -> libcint's xy,yz,z2,xz,x2-y2
        const double cv[6] = {dx * dx, dx * dy, dx * dz, dy * dy, dy * dz, dz * dz};
        const double cgx[6] = {2.0 * dx, dy, dz, 0.0, 0.0, 0.0};
        const double cgy[6] = {0.0, dx, 0.0, 2.0 * dy, dz, 0.0};
        const double cgz[6] = {0.0, 0.0, dx, 0.0, dy, 2.0 * dz};
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            double v = 0.0, ax = 0.0, ay = 0.0, az = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double w = c2s_coef<2>(nullptr, m, k);
                v += w * cv[k]; ax += w * cgx[k]; ay += w * cgy[k]; az += w * cgz[k];
            }
            chi[(ao + m) * ptp + p] = v * rad;
            if (GGA) {
                gx[(ao + m) * ptp + p] = ax * rad + v * dx * drad;
                gy[(ao + m) * ptp + p] = ay * rad + v * dy * drad;
                gz[(ao + m) * ptp + p] = az * rad + v * dz * drad;
            }
        }
    }
}

// one (point, radial group) work item: the exponentials of the group's primitives are formed once and contracted
// with every member shell's coefficient row (TopologyDev::grp_*), then each member gets its angular part
// the radial-group tables a workgroup keeps in LDS (the tiled kernel): per group a packed descriptor
// (first shell's AO offset | l << 12 | atom << 16 | shell count << 24 | primitive count << 26), exponent / coefficient
// offsets, and the exponent and coefficient arrays themselves -- global loads in the primitive loop were a dependent
// chain of L2 latencies per (point, group) item
struct GroupTables {
    const int* desc;        // [ngroup][4]: packed, poff, coff, ao offsets of shells 1..2 packed (aoff1 | aoff2 << 16)
    const double* exps;
    const double* coefs;
};

template <bool GGA>
__device__ __forceinline__ void eval_group_lds(const GroupTables& gt, const double* __restrict__ xyz, int g, double px, double py, double pz,
                                               double* __restrict__ chi, double* __restrict__ gx, double* __restrict__ gy,
                                               double* __restrict__ gz, int ptp, int p, const double* __restrict__ c2s);

template <bool GGA>
__device__ __forceinline__ void eval_group(const TopologyDev& tp, const double* __restrict__ xyz, int g, double px, double py, double pz,
                                           double* __restrict__ chi, double* __restrict__ gx, double* __restrict__ gy,
                                           double* __restrict__ gz, int ptp, int p, const double* __restrict__ c2s)
{
    const int sh0 = tp.grp_first[g], nc = tp.grp_count[g], np = tp.grp_nprim[g];
    const int l = tp.sh_l[sh0], at = tp.sh_atom[sh0];
    const double dx = px - xyz[3 * at], dy = py - xyz[3 * at + 1], dz = pz - xyz[3 * at + 2];
    const double r2 = dx * dx + dy * dy + dz * dz;
    const double* e = tp.gexps + tp.grp_poff[g];
    const double* c = tp.gcoefs + tp.grp_coff[g];
    double rad[XC_GROUP_MAX], drad[XC_GROUP_MAX];
#pragma unroll
    for (int k = 0; k < XC_GROUP_MAX; ++k) { rad[k] = 0.0; drad[k] = 0.0; }
    for (int i = 0; i < np; ++i) {
        const double ar2 = e[i] * r2;
        if (ar2 < XC_EXP_CUTOFF) {           // exp(-46) = 1e-20: tight primitives vanish a fraction of a bohr from their nucleus
            const double ex = exp(-ar2), m2e = -2.0 * e[i];
#pragma unroll
            for (int k = 0; k < XC_GROUP_MAX; ++k) {
                if (k < nc) {
                    const double t = c[k * np + i] * ex;
                    rad[k] += t; drad[k] += m2e * t;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < XC_GROUP_MAX; ++k)
        if (k < nc) emit_shell<GGA>(l, tp.sh_aoff[sh0 + k], dx, dy, dz, rad[k], drad[k], chi, gx, gy, gz, ptp, p, c2s);
}

template <bool GGA>
__device__ __forceinline__ void eval_group_lds(const GroupTables& gt, const double* __restrict__ xyz, int g, double px, double py, double pz,
                                               double* __restrict__ chi, double* __restrict__ gx, double* __restrict__ gy,
                                               double* __restrict__ gz, int ptp, int p, const double* __restrict__ c2s)
{
    const int pk = gt.desc[6 * g], poff = gt.desc[6 * g + 1], coff = gt.desc[6 * g + 2], ao12 = gt.desc[6 * g + 3];
    const int ao0 = pk & 0xfff, l = (pk >> 12) & 0xf, at = (pk >> 16) & 0xff, nc = (pk >> 24) & 0x3, np = (pk >> 26) & 0x3f;
    const double dx = px - xyz[3 * at], dy = py - xyz[3 * at + 1], dz = pz - xyz[3 * at + 2];
    const double r2 = dx * dx + dy * dy + dz * dz;
    const double* e = gt.exps + poff;
    const double* c = gt.coefs + coff;
    double rad[XC_GROUP_MAX], drad[XC_GROUP_MAX];
#pragma unroll
    for (int k = 0; k < XC_GROUP_MAX; ++k) { rad[k] = 0.0; drad[k] = 0.0; }
    for (int i = 0; i < np; ++i) {
        const double ei = e[i];
        const double ar2 = ei * r2;
        if (ar2 < XC_EXP_CUTOFF) {
            const double ex = exp(-ar2), m2e = -2.0 * ei;
#pragma unroll
            for (int k = 0; k < XC_GROUP_MAX; ++k) {
                if (k < nc) {
                    const double t = c[k * np + i] * ex;
                    rad[k] += t; drad[k] += m2e * t;
                }
            }
        }
    }
    emit_shell<GGA>(l, ao0, dx, dy, dz, rad[0], drad[0], chi, gx, gy, gz, ptp, p, c2s);
    if (nc > 1) emit_shell<GGA>(l, ao12 & 0xffff, dx, dy, dz, rad[1], drad[1], chi, gx, gy, gz, ptp, p, c2s);
    if (nc > 2) emit_shell<GGA>(l, ao12 >> 16, dx, dy, dz, rad[2], drad[2], chi, gx, gy, gz, ptp, p, c2s);
}

// the same item from the radial cache: rad / drad of the group's shells are loaded (rad_pt values of a tile are
// contiguous, lanes along the points), only the angular part is formed
template <bool GGA>
__device__ __forceinline__ void eval_group_cached(const GroupTables& gt, const double* __restrict__ xyz, int g, double px, double py, double pz,
                                                  const double* __restrict__ radt /* this tile: [nshell][2][PT] */, int pt_stride,
                                                  double* __restrict__ chi, double* __restrict__ gx, double* __restrict__ gy,
                                                  double* __restrict__ gz, int ptp, int p, const double* __restrict__ c2s)
{
    const int pk = gt.desc[6 * g], ao12 = gt.desc[6 * g + 3], sh0 = gt.desc[6 * g + 4];
    const int ao0 = pk & 0xfff, l = (pk >> 12) & 0xf, at = (pk >> 16) & 0xff, nc = (pk >> 24) & 0x3;
    const double dx = px - xyz[3 * at], dy = py - xyz[3 * at + 1], dz = pz - xyz[3 * at + 2];
    const double* r0 = radt + (size_t)sh0 * 2 * pt_stride + p;
    emit_shell<GGA>(l, ao0, dx, dy, dz, r0[0], r0[pt_stride], chi, gx, gy, gz, ptp, p, c2s);
    if (nc > 1) emit_shell<GGA>(l, ao12 & 0xffff, dx, dy, dz, r0[2 * pt_stride], r0[3 * pt_stride], chi, gx, gy, gz, ptp, p, c2s);
    if (nc > 2) emit_shell<GGA>(l, ao12 >> 16, dx, dy, dz, r0[4 * pt_stride], r0[5 * pt_stride], chi, gx, gy, gz, ptp, p, c2s);
}

// fills the radial cache: thread = (radial group, point) of one fragment, the exponentials of eval_group once per SCF
__global__ void __launch_bounds__(256) xc_radial_cache_kernel(BatchView bv)
{
    const int f = blockIdx.y;
    const TopologyDev& tp = bv.topo;
    const GridDev& gd = bv.grid;
    const int PTC = gd.rad_pt;
    const int ntiles = (gd.npts + PTC - 1) / PTC;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)tp.ngroup * ntiles * PTC;
    if (idx >= total) return;
    const int rg = (int)(idx / ((long)ntiles * PTC));
    const int gp = (int)(idx - (long)rg * ntiles * PTC);          // padded point index
    const int tile = gp / PTC, p = gp - tile * PTC;
    const int sh0 = tp.grp_first[rg], nc = tp.grp_count[rg], np = tp.grp_nprim[rg];
    double rad[XC_GROUP_MAX], drad[XC_GROUP_MAX];
#pragma unroll
    for (int k = 0; k < XC_GROUP_MAX; ++k) { rad[k] = 0.0; drad[k] = 0.0; }
    if (gp < gd.npts) {
        const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
        const int oa = gd.pt_atom[gp], it = gd.pt_tmpl[gp], at = tp.sh_atom[sh0];
        const double dx = xyz[3 * oa] + gd.tmpl_xyz[3 * it] - xyz[3 * at], dy = xyz[3 * oa + 1] + gd.tmpl_xyz[3 * it + 1] - xyz[3 * at + 1],
                     dz = xyz[3 * oa + 2] + gd.tmpl_xyz[3 * it + 2] - xyz[3 * at + 2];
        const double r2 = dx * dx + dy * dy + dz * dz;
        const double* e = tp.gexps + tp.grp_poff[rg];
        const double* c = tp.gcoefs + tp.grp_coff[rg];
        for (int i = 0; i < np; ++i) {
            const double ei = e[i], ar2 = ei * r2;
            if (ar2 < XC_EXP_CUTOFF) {
                const double ex = exp(-ar2), m2e = -2.0 * ei;
#pragma unroll
                for (int k = 0; k < XC_GROUP_MAX; ++k)
                    if (k < nc) { const double t = c[k * np + i] * ex; rad[k] += t; drad[k] += m2e * t; }
            }
        }
    }
    double* base = gd.rad + ((size_t)f * ntiles + tile) * tp.nshell * 2 * PTC;
#pragma unroll
    for (int k = 0; k < XC_GROUP_MAX; ++k)
        if (k < nc) {
            double* r0 = base + (size_t)(sh0 + k) * 2 * PTC + p;
            r0[0] = rad[k]; r0[PTC] = drad[k];
        }
}

void launch_xc_radial_cache(const BatchView& bv, hipStream_t s)
{
    const int PTC = bv.grid.rad_pt;
    const long ntiles = (bv.grid.npts + PTC - 1) / PTC;
    const long total = (long)bv.topo.ngroup * ntiles * PTC;
    hipLaunchKernelGGL(xc_radial_cache_kernel, dim3((unsigned)((total + 255) / 256), bv.nfrag), dim3(256), 0, s, bv);
}

// zero rows of a group for a point beyond the grid
template <bool GGA>
__device__ __forceinline__ void zero_group(const TopologyDev& tp, int g, double* __restrict__ chi, double* __restrict__ gx,
                                           double* __restrict__ gy, double* __restrict__ gz, int ptp, int p)
{
    const int sh0 = tp.grp_first[g], nc = tp.grp_count[g], l = tp.sh_l[sh0];
    for (int k = 0; k < nc; ++k) {
        const int ao = tp.sh_aoff[sh0 + k];
        for (int m = 0; m < 2 * l + 1; ++m) {
            chi[(ao + m) * ptp + p] = 0.0;
            if (GGA) { gx[(ao + m) * ptp + p] = 0.0; gy[(ao + m) * ptp + p] = 0.0; gz[(ao + m) * ptp + p] = 0.0; }
        }
    }
}

// PT points per tile; NV = ceil(n*n / 256) register accumulators per thread
typedef double v4f64 __attribute__((ext_vector_type(4)));      // accumulator / result registers of v_mfma_f64_16x16x4

// ------------------------------------------------------------------ tiled workgroup kernel, any n (round 2)
// (The round-1 kernels -- a wave-private MFMA kernel that kept the whole density matrix and all accumulator tiles in one
// wave's 400 registers, and a VALU kernel above n = 48 -- were removed in round 3.)  Here a WORKGROUP of XV_NW waves owns a tile of PT
// points: the AO slab [function][point] is shared in LDS, and the two GEMM-shaped contractions are cut into
// 16 x 16 MFMA jobs dealt round-robin to the waves --
//     X = D chi      jobs (row tile, point tile): D as A-fragments straight from global memory (one fragment's D is
//                    n^2 * 8 B and stays in L1/L2), chi as B-fragments from LDS; the job's partial rho / grad rho
//                    are reduced over its 16 rows in registers and added into LDS with ds_add_f64;
//     A += a chi^T   jobs (row tile, column tile): a wave keeps ITS tiles' accumulators (ceil(NT16^2 / XV_NW) of them)
//                    in registers across all tiles of the workgroup and flushes them once.
// A wave then needs < 128 registers at n = 48 (3 workgroups = 12 waves per CU), the functional runs once per point
// (lane = point on wave 0 while the other workgroups of the CU compute), and n up to 144 uses the matrix cores
// (benzene/cc-pVDZ n = 114, def2-TZVP water dimer n = 86).  Same arithmetic as mqc_libcint_xc.F90:796-927.
constexpr int XV_NW = 4;
// MQC_HIP_XC_PROBE (timing experiments only, results are then meaningless): bit 0 skips the AO slab evaluation, bit 1 the
// X = D chi jobs, bit 2 the functional (constants instead), bit 3 the accumulation A += a chi^T
__device__ int g_xc_probe = 0;
// In-kernel phase stamps (build with -DXC_STAMPS=1; never in the shipped library): every wave adds the cycles it spent
// working in and waiting at the end of each phase of a tile; launch_xc prints the running totals to stderr.
#ifndef XC_STAMPS
#define XC_STAMPS 0
#endif
#if XC_STAMPS
__device__ unsigned long long g_xc_stamps[16];
#define XC_ST_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define XC_ST(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_t; st_t = t_; }
#else
#define XC_ST_DECL
#define XC_ST(k)
#endif

template <bool GGA, int PT, int JMAX, int OCC, bool DREG, int NTC, bool FAST = false, int NWV = XV_NW>
__global__ void __launch_bounds__(64 * NWV, OCC) xc_tile_kernel(BatchView bv, int only_active)
{
    // One tile = PT points: AO slab -> X = D chi, rho / grad rho -> functional (one lane per point) -> a -> A += a chi^T.
    // (A 256-point super-tile in two passes -- so that all 256 threads evaluate the functional -- rebuilt the slab and was
    // slower, 2.09 s against 1.99 s per evaluation at the time; removed)
    extern __shared__ double lds[];
    const int f = blockIdx.y;
    // FAST: the lean build of the slab phase for the common case -- radial tile in LDS, no f shells: one item per
    // (SHELL, point) with a single copy of the angular code.  The general slab (radial groups with up to three
    // contracted shells each, cached / uncached paths, f and g shells) is ~4 500 instructions, the whole tile loop
    // ~80 KB of code against a 64 KB instruction cache shared by two CUs: the eight waves of a CU, each in its own
    // phase, were waiting on instruction fetches (making MORE waves run different code at once -- the functional's
    // components dealt to four waves -- made the kernel slower, twice).
    const bool rad_in_lds = FAST || (only_active & 2) != 0;   // bit 1 of the flag word: radial tile staged in LDS
    if ((only_active & 1) && bv.istate[4 * f] == ST_DONE) return;
    constexpr int RS = PT + 1, PT16 = PT / 16, NTHR = 64 * NWV;
    const int n = bv.n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lo = lane & 15, hi = lane >> 4;
    // NTC > 0: the number of 16-function tiles is a template constant (n <= 64), so that every k-loop and tile test
    // below unrolls without a branch per step -- with a runtime count hipcc wrapped each LDS read and each MFMA of the
    // X = D chi loop in its own scalar branch and spilled the loop's scalars into VGPR lanes
    const int NT16 = NTC > 0 ? NTC : ((n + 15) >> 4), NP = NT16 << 4, KS = NP >> 2;
    const TopologyDev& tp = bv.topo;
    const GridDev& gd = bv.grid;
    double* chi = lds;                                        // [NP][RS]
    double* gx = chi + (size_t)NP * RS;                       // GGA: grad chi, x then overwritten by a;  LDA: a
    double* gy = gx + (size_t)NP * RS;
    double* gz = gy + (GGA ? (size_t)NP * RS : 0);
    double* red = gz + (GGA ? (size_t)NP * RS : 0);           // [PT][4] rho, grad rho sums of the tile
    double* coef = red + 4 * PT;                              // [PT][4] w v_rho / 2, 2 w v_sigma grad rho
    double* axyz = coef + 4 * PT;                             // [64][3] this fragment's atoms
    double* pxyz = axyz + 3 * 64;                             // [2][PT][3] grid points of the current / next tile
    double* tab = pxyz + 6 * PT;                              // radial-group tables: desc ints, exponents, coefficients
    const int ng = tp.ngroup, ngp = tp.gprim_total, ngc = tp.gcoef_total;
    int* tdesc = (int*)tab;
    double* texps = tab + 3 * ng;
    double* tcoefs = texps + ngp;
    int* sdesc = (int*)(tab + 3 * ng + ngp + ngc + 8);        // [nshell] ao | l << 12 | atom << 16 (FAST slab)
    double* radl = tab + (((size_t)(3 * ng + ngp + ngc + 8 + (tp.nshell + 1) / 2) + 1) & ~(size_t)1);      // [nshell][2][PT] radial tile (16-byte aligned)
    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const double* __restrict__ D = bv.D + (size_t)f * n * n;
    const double* __restrict__ wts = gd.weights + (size_t)f * gd.npts;

    // rows n..NP-1 stay zero for the whole kernel; the sums start at zero
    for (int idx = tid; idx < (GGA ? 4 : 2) * NP * RS + 8 * PT; idx += NTHR) lds[idx] = 0.0;
    for (int g = tid; g < ng; g += NTHR) {
        const int sh0 = tp.grp_first[g], nc = tp.grp_count[g], npg = tp.grp_nprim[g];
        tdesc[6 * g] = tp.sh_aoff[sh0] | (tp.sh_l[sh0] << 12) | (tp.sh_atom[sh0] << 16) | (nc << 24) | (npg << 26);
        tdesc[6 * g + 1] = tp.grp_poff[g];
        tdesc[6 * g + 2] = tp.grp_coff[g];
        tdesc[6 * g + 3] = (nc > 1 ? tp.sh_aoff[sh0 + 1] : 0) | ((nc > 2 ? tp.sh_aoff[sh0 + 2] : 0) << 16);
        tdesc[6 * g + 4] = sh0;
        tdesc[6 * g + 5] = 0;
    }
    if (FAST) for (int sh = tid; sh < tp.nshell; sh += NTHR) sdesc[sh] = tp.sh_aoff[sh] | (tp.sh_l[sh] << 12) | (tp.sh_atom[sh] << 16);
    for (int idx = tid; idx < ngp; idx += NTHR) texps[idx] = tp.gexps[idx];
    for (int idx = tid; idx < ngc; idx += NTHR) tcoefs[idx] = tp.gcoefs[idx];
    const GroupTables gt{tdesc, texps, tcoefs};
    for (int idx = tid; idx < 3 * tp.natoms; idx += NTHR) axyz[idx] = xyz[idx];
    // grid point coordinates of a tile into LDS (buffer b): one pair of dependent global loads per POINT, issued a tile
    // ahead -- every (radial group, point) item used to walk pt_atom -> pt_tmpl -> tmpl_xyz itself, and the slab phase
    // was a chain of L2 latencies
    auto load_points = [&](int g0, int b, int q) {
        const int g = g0 + q;
        double x = 0.0, y = 0.0, z = 0.0;
        if (g < gd.npts) {
            const int oa = gd.pt_atom[g], it = gd.pt_tmpl[g];
            x = xyz[3 * oa] + gd.tmpl_xyz[3 * it]; y = xyz[3 * oa + 1] + gd.tmpl_xyz[3 * it + 1]; z = xyz[3 * oa + 2] + gd.tmpl_xyz[3 * it + 2];
        }
        double* pp = pxyz + 3 * (b * PT + q);
        pp[0] = x; pp[1] = y; pp[2] = z;
    };
    if (tid < PT) load_points(blockIdx.x * PT, 0, tid);
    int pbuf = 0;
    // radial cache of this fragment (tiles of PT points), when the engine filled one with this tile size
    const double* __restrict__ radf = (gd.rad && gd.rad_pt == PT)
                                          ? gd.rad + (size_t)f * ((size_t)(gd.npts + PT - 1) / PT) * tp.nshell * 2 * PT : nullptr;
    // the radial values of a tile are one contiguous block of nshell x 2 x PT doubles in HBM: LDS-DMA moves it into
    // LDS without registers (1 KiB per wave-instruction), issued as soon as the slab of the current tile has been
    // built and landing under the MFMA phase -- read straight from HBM by the slab items, every item waited out an
    // HBM round trip and the cache saved nothing over recomputing the exponentials
    const int rad_bytes = tp.nshell * 2 * PT * 8;
    auto stage_radial = [&](int tile) {
        if (!rad_in_lds || tile * PT >= gd.npts) return;
        const char* src = (const char*)(radf + (size_t)tile * tp.nshell * 2 * PT);
        for (int c = wave; c * 1024 < rad_bytes; c += NWV) {
            const int off = c * 1024 + lane * 16;
            if (off < rad_bytes)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                                 (__attribute__((address_space(3))) void*)((char*)radl + c * 1024), 16, 0, 0);
        }
    };
    stage_radial(blockIdx.x);
    int iter = 0;
    // DREG (n <= 64): the X = D chi jobs of a wave are the same (row tile, point tile) pairs in every tile, so the wave
    // keeps its density fragments in registers for the whole kernel -- reading them from L2 per tile made the phase a
    // chain of load latencies (probes: 30 % of the kernel for 10 % of its MFMA work)
    static_assert(!DREG || NTC > 0, "register-resident density fragments need a compile-time tile count");
    constexpr int DJ = DREG ? (NTC * PT16 + NWV - 1) / NWV : 1, DK = DREG ? 4 * NTC : 1;      // X jobs per wave, k-steps per job
    double dfrag[DJ][DK];
    if (DREG) {
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const int job = wave + NWV * j;
            const int mt = job / PT16;
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) {
                const int mu = 16 * mt + lo, nu = 4 * ks + hi;
                dfrag[j][ks] = (job < NT16 * PT16 && mu < n && nu < n) ? D[(size_t)mu * n + nu] : 0.0;
            }
        }
    }
    v4f64 vacc[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) vacc[j] = (v4f64){0.0, 0.0, 0.0, 0.0};
    double e_acc = 0.0, n_acc = 0.0;
    __syncthreads();
    XC_ST_DECL

    const int probe = g_xc_probe;
    // n > 144: the 16 x 16 output tiles are dealt to blockIdx.z groups of NWV * JMAX (the slab, the densities and the
    // functional are formed in every group; the energy counts in group 0)
    const int tz0 = (int)blockIdx.z * NWV * JMAX;
    auto ao_slab = [&](int g0) {
        if (probe & 1) return;
        if constexpr (FAST) {
            // (shell, point) items, point fastest; beyond the grid's end the radial values are zero and so is the shell
            for (int idx = tid; idx < tp.nshell * PT; idx += NTHR) {
                const int sh = idx / PT, p = idx - sh * PT;
                const int sd = sdesc[sh];
                const int ao = sd & 0xfff, l = (sd >> 12) & 0xf, at = sd >> 16;
                const double* pp = pxyz + 3 * (pbuf * PT + p);
                const double dx = pp[0] - axyz[3 * at], dy = pp[1] - axyz[3 * at + 1], dz = pp[2] - axyz[3 * at + 2];
                const bool in = g0 + p < gd.npts;
                const double* r0 = radl + (size_t)sh * 2 * PT + p;
                emit_shell<GGA, false>(l, ao, dx, dy, dz, in ? r0[0] : 0.0, in ? r0[PT] : 0.0, chi, gx, gy, gz, RS, p, bv.c2s);
            }
            return;
        }
        // AO values (and gradients) of PT points: (radial group, point) items, point fastest
        for (int idx = tid; idx < tp.ngroup * PT; idx += NTHR) {
            const int rg = idx / PT, p = idx - rg * PT;
            const int g = g0 + p;
            if (g < gd.npts) {
                const double* pp = pxyz + 3 * (pbuf * PT + p);
                const double ptx = pp[0], pty = pp[1], ptz = pp[2];
                if (rad_in_lds)
                    eval_group_cached<GGA>(gt, axyz, rg, ptx, pty, ptz, radl, PT, chi, gx, gy, gz, RS, p, bv.c2s);
                else if (radf)
                    eval_group_cached<GGA>(gt, axyz, rg, ptx, pty, ptz, radf + (size_t)(g0 / PT) * tp.nshell * 2 * PT, PT,
                                           chi, gx, gy, gz, RS, p, bv.c2s);
                else
                    eval_group_lds<GGA>(gt, axyz, rg, ptx, pty, ptz, chi, gx, gy, gz, RS, p, bv.c2s);
            } else {
                zero_group<GGA>(tp, rg, chi, gx, gy, gz, RS, p);
            }
        }
    };

    const int ntile = (gd.npts + PT - 1) / PT;
    for (int st = blockIdx.x; st < ntile; st += gridDim.x) {
        const int s0 = st * PT, g0 = s0;
        // ---- the slab and the densities of the tile's points
        {
            XC_ST(11)
            ao_slab(g0);
            XC_ST(0)
            __syncthreads();
            XC_ST(1)
            stage_radial(st + (int)gridDim.x);      // lands under the MFMA phase below
            // X = D chi, job = (row tile mt, point tile pt); rho and grad rho from the accumulator rows
#pragma unroll 2
            for (int jj = 0; jj < (DREG ? DJ : 64); ++jj) {       // DREG: DJ <= 2, unrolled
                const int job = wave + NWV * jj;
                if (job >= NT16 * PT16 || (probe & 2)) break;
                const int mt = job / PT16, pt = job - mt * PT16;
                v4f64 xacc = (v4f64){0.0, 0.0, 0.0, 0.0};
                const int mu_a = 16 * mt + lo;
                const double* __restrict__ drow = D + (size_t)mu_a * n;
                const bool row_ok = mu_a < n;
                if (DREG) {
                    double bw[DK];
#pragma unroll
                    for (int ks = 0; ks < DK; ++ks) bw[ks] = chi[(4 * ks + hi) * RS + 16 * pt + lo];
#pragma unroll
                    for (int ks = 0; ks < DK; ++ks)
                        xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(dfrag[jj < DJ ? jj : 0][ks], bw[ks], xacc, 0, 0, 0);
                } else
                // eight k-steps at a time: the density elements (global, L1/L2) and the AO values (LDS) of a batch are all
                // in flight before the MFMAs that consume them -- one load per MFMA made the loop a chain of L2 latencies
                for (int k0 = 0; k0 < KS; k0 += 8) {
                    double av[8], bw[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int nu = 4 * (k0 + u) + hi;
                        const bool ok = (k0 + u < KS) && row_ok && nu < n;
                        av[u] = ok ? drow[nu] : 0.0;
                        bw[u] = (k0 + u < KS) ? chi[nu * RS + 16 * pt + lo] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bw[u], xacc, 0, 0, 0);
                }
                double rho = 0.0, rx = 0.0, ry = 0.0, rz = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = (16 * mt + hi + 4 * r) * RS + 16 * pt + lo;
                    const double x = xacc[r];
                    rho += x * chi[o];
                    if (GGA) { rx += x * gx[o]; ry += x * gy[o]; rz += x * gz[o]; }
                }
                rho += __shfl_xor(rho, 16, 64); rho += __shfl_xor(rho, 32, 64);
                if (GGA) {
                    rx += __shfl_xor(rx, 16, 64); rx += __shfl_xor(rx, 32, 64);
                    ry += __shfl_xor(ry, 16, 64); ry += __shfl_xor(ry, 32, 64);
                    rz += __shfl_xor(rz, 16, 64); rz += __shfl_xor(rz, 32, 64);
                }
                if (hi == 0) {
                    double* rp = red + 4 * (16 * pt + lo);
                    atomicAdd(&rp[0], rho);
                    if (GGA) { atomicAdd(&rp[1], rx); atomicAdd(&rp[2], ry); atomicAdd(&rp[3], rz); }
                }
            }
            XC_ST(2)
            __syncthreads();
            XC_ST(3)
        }
        // ---- the functional: lane = point, on ONE wave -- a long dual-number instruction stream that keeps its SIMD's
        // FP64 pipe busy by itself.  The wave ROTATES over the tiles: wave w of every workgroup sits on SIMD w, so with
        // a fixed wave the functional phases of a CU's workgroups all queued on one SIMD while three idled.  (Dealing
        // the functional's components to the four waves was measured slower: 1.39 s against 1.30 s per evaluation.)
        {
            if (wave == ((iter + (int)blockIdx.x) & (NWV - 1)) && lane < PT) {
                const int p = lane;
                double* rp = red + 4 * p;
                const double rho = rp[0], rx = 2.0 * rp[1], ry = 2.0 * rp[2], rz = 2.0 * rp[3];
                rp[0] = 0.0; rp[1] = 0.0; rp[2] = 0.0; rp[3] = 0.0;
                const double sigma = GGA ? rx * rx + ry * ry + rz * rz : 0.0;
                double fx, vr, vs;
                if (probe & 4) { fx = -rho; vr = -1.0; vs = 0.0; }
                else eval_functional(bv.xc, rho, sigma, fx, vr, vs);
                const double w = (s0 + p < gd.npts) ? wts[s0 + p] : 0.0;
                e_acc += w * fx;
                n_acc += w * rho;
                double* cp = coef + 4 * p;
                cp[0] = 0.5 * w * vr;
                const double t2 = 2.0 * w * vs;
                cp[1] = t2 * rx; cp[2] = t2 * ry; cp[3] = t2 * rz;
            }
        }
        // ---- a and the accumulation (the slab is still in LDS)
        {
            XC_ST(4)
            __syncthreads();          // coef written
            XC_ST(5)
            // the next tile's grid points, a tile ahead (last PT threads)
            if (tid >= NTHR - PT) load_points((st + (int)gridDim.x) * PT, pbuf ^ 1, tid - (NTHR - PT));
            // a[mu][p] = w v_rho / 2 chi + 2 w v_sigma grad rho . grad chi, written over gx
            {       // a thread's point is the same for all its elements (NTHR is a multiple of PT): its four weights once
                const int p = tid % PT;
                const double* cp = coef + 4 * p;
                const double c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
#pragma unroll 3
                for (int mu = tid / PT; mu < n; mu += NTHR / PT) {
                    const int o = mu * RS + p;
                    double a = c0 * chi[o];
                    if (GGA) a += c1 * gx[o] + c2 * gy[o] + c3 * gz[o];
                    gx[o] = a;
                }
            }
            XC_ST(6)
            __syncthreads();
            XC_ST(7)
            // A += a chi^T: this wave's tiles t = wave, wave + NWV, ...
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                const int t = tz0 + wave + NWV * j;
                if (t < NT16 * NT16 && !(probe & 8)) {
                    const int mt = t / NT16, nt = t - mt * NT16;
                    const double* __restrict__ ar = gx + (size_t)(16 * mt + lo) * RS + hi;
                    const double* __restrict__ br = chi + (size_t)(16 * nt + lo) * RS + hi;
#pragma unroll
                    for (int ks = 0; ks < PT / 4; ++ks) vacc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[4 * ks], br[4 * ks], vacc[j], 0, 0, 0);
                }
            }
        }
        pbuf ^= 1;
        ++iter;
        XC_ST(8)
        __syncthreads();
        XC_ST(9)
    }
#if XC_STAMPS
    XC_ST(11)
    if (lane == 0)
        for (int k = 0; k < 12; ++k) atomicAdd(&g_xc_stamps[k], st_acc[k]);
#endif
    // flush: lane holds A[mu = 16 mt + hi + 4 r][nu = 16 nt + lo]
    double* Vx = bv.Vxc + (size_t)f * n * n;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int t = tz0 + wave + NWV * j;
        if (t < NT16 * NT16) {
            const int mt = t / NT16, nt = t - mt * NT16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mu = 16 * mt + hi + 4 * r, nu = 16 * nt + lo;
                const double v = vacc[j][r];
                if (mu < n && nu < n && v != 0.0) atomicAdd(&Vx[mu * n + nu], v);
            }
        }
    }
    // E_xc and N_e: every thread holds partial sums
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { e_acc += __shfl_down(e_acc, off, 64); n_acc += __shfl_down(n_acc, off, 64); }
    if (lane == 0 && blockIdx.z == 0) {
        atomicAdd(&bv.scal[(size_t)f * 8 + 5], e_acc);
        atomicAdd(&bv.scal[(size_t)f * 8 + 6], n_acc);
    }
}

template <bool GGA, int PT, int JMAX, int OCC, bool DREG, int NTC, int NWV = XV_NW>
static void xc_tile_launch(const BatchView& bv, int oa, hipStream_t s)
{
    const int np = ((bv.n + 15) / 16) * 16;
    // + radial-group tables: 2 doubles of descriptor per group, the exponents and up to XC_GROUP_MAX coefficient rows;
    // groups <= shells <= n, primitives per group <= 63 (descriptor field); bounded by the topology's own totals
    const size_t tab = 3 * (size_t)bv.topo.ngroup + (size_t)bv.topo.gprim_total + (size_t)bv.topo.gcoef_total + 8 + ((size_t)bv.topo.nshell + 1) / 2;
    size_t lds = sizeof(double) * ((size_t)(GGA ? 4 : 2) * np * (PT + 1) + 8 * PT + 3 * 64 + 6 * PT + ((tab + 1) & ~(size_t)1));
    // the radial tile rides in LDS when that does not cost a resident workgroup (OCC of them share 160 KB)
    const size_t rad_lds = sizeof(double) * (size_t)bv.topo.nshell * 2 * PT;
    static const bool rad_lds_on = [] { const char* e = std::getenv("MQC_HIP_XC_RADIAL_LDS"); return !(e && e[0] == '0'); }();
    const size_t lds_cap = (size_t)160 * 1024 / (OCC < 1 ? 1 : OCC) - 512;
    if (rad_lds_on && bv.grid.rad && bv.grid.rad_pt == PT && (lds + rad_lds <= lds_cap || (lds > lds_cap && lds + rad_lds <= 156 * 1024))) {
        lds += rad_lds;
        oa |= 2;
    }
    auto kern = xc_tile_kernel<GGA, PT, JMAX, OCC, DREG, NTC, false, NWV>;
    if constexpr (DREG) {
        static const bool fast_on = [] { const char* e = std::getenv("MQC_HIP_XC_FAST_SLAB"); return !(e && e[0] == '0'); }();
        if (fast_on && (oa & 2) && bv.topo.lmax <= 2) kern = xc_tile_kernel<GGA, PT, JMAX, OCC, DREG, NTC, true, NWV>;
    }
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int tile_pts = PT;
    const int ntiles = (bv.grid.npts + tile_pts - 1) / tile_pts;
    int gx = (6144 + bv.nfrag - 1) / bv.nfrag;       // ~3 workgroups per CU x 8 in flight over the batch; many tiles each
    if (gx > ntiles) gx = ntiles;
    if (gx < 1) gx = 1;
    const int nt16 = (bv.n + 15) / 16;
    const int nz = (nt16 * nt16 + NWV * JMAX - 1) / (NWV * JMAX);
    hipLaunchKernelGGL(kern, dim3(gx, bv.nfrag, nz), dim3(64 * NWV), lds, s, bv, oa);
}

template <bool GGA>
static bool xc_tile_dispatch(const BatchView& bv, int oa, hipStream_t s)
{
    // 32-point tiles at two workgroups (eight waves) per CU: 256 registers, no scratch.  (A 168-register build for
    // three waves per SIMD spilled ~300 B per lane in the functional and measured slower; removed.)
    const int nt = (bv.n + 15) / 16, jobs = (nt * nt + XV_NW - 1) / XV_NW;
    if (nt <= 4) {                       // n <= 64
        if (nt == 1) xc_tile_launch<GGA, 32, 1, 2, true, 1>(bv, oa, s);
        else if (nt == 2) xc_tile_launch<GGA, 32, 1, 2, true, 2>(bv, oa, s);
        else if (nt == 3) xc_tile_launch<GGA, 32, 3, 2, true, 3>(bv, oa, s);
        else xc_tile_launch<GGA, 32, 4, 2, true, 4>(bv, oa, s);
        return true;
    }
    // 64 < n <= 96 (def2-TZVP water dimer, n = 86): still 32-point tiles with the density fragments in registers, ONE
    // workgroup per CU (101 KB of slab) and the whole register file for its four waves -- against 16-point tiles the
    // functional runs on twice the lanes and the density is not re-read from L2 per tile (MQC_HIP_XC_WIDE_TILE=0: off)
    static const bool wide_tile = [] { const char* e = std::getenv("MQC_HIP_XC_WIDE_TILE"); return !(e && e[0] == '0'); }();
    if (wide_tile && nt <= 6) {
        // (eight waves per workgroup -- half the accumulators per wave -- measured the same: 3.94 against 3.92 s)
        if (nt == 5) xc_tile_launch<GGA, 32, 7, 1, true, 5>(bv, oa, s);
        else xc_tile_launch<GGA, 32, 9, 1, true, 6>(bv, oa, s);
        return true;
    }
    if (jobs <= 9) xc_tile_launch<GGA, 16, 9, 1, false, 0>(bv, oa, s);             // n <= 96
    else if (jobs <= 16) xc_tile_launch<GGA, 16, 16, 1, false, 0>(bv, oa, s);      // n <= 128
    else if (jobs <= 21) xc_tile_launch<GGA, 16, 21, 1, false, 0>(bv, oa, s);      // n <= 144
    else if (bv.n <= 256) xc_tile_launch<GGA, 16, 16, 1, false, 0>(bv, oa, s);     // above: 64 output tiles per z group
    else return false;
    return true;
}

// ------------------------------------------------------------------ the pipelined quadrature kernel (n <= 64)
// The tile kernel above parks three of its four waves at a barrier while ONE wave evaluates the functional -- 47 % of a
// workgroup's timeline (profiles/r02_xc_phase_stamps.txt).  Here a workgroup is EIGHT waves with specialised roles and
// TWO slab buffers, and the tiles move through a software pipeline:
//     waves 0..5  workers:     S1 slab(t+1) -> buffer B    S2 X = D chi, rho, grad rho of t+1     S3 A += a chi^T of tile t
//     wave  6     exchange:    S1 / S2 the exchange components of the functional on tile t (lane = point)
//     wave  7     correlation: S1 / S2 the correlation components of tile t; both leave partial v_rho, v_sigma in LDS
// so that the long dual-number instruction streams run UNDER the workers' slab and MFMA phases instead of in front of
// three idle waves.  Three barriers per tile instead of five.  The coefficient vector a = w v_rho / 2 chi + 2 w v_sigma
// grad rho . grad chi is not stored: in S3 a wave owns the pair (row tile mt, half kh of the tile's points), forms its
// A-operand fragments from the slab on the fly and feeds them to the MFMAs of all column tiles nt -- one pass over the
// slab less and one barrier less.  Same arithmetic as xc_tile_kernel (mqc_libcint_xc.F90:796-927); the summation order
// inside rho and A differs only through the halves' atomic adds.
// Conditions (else the tile kernel runs): radial cache filled with 32-point tiles, s/p/d shells, restricted, LDS fits.
constexpr int XP_NW = 8, XP_WORK = 6, XP_PT = 32;
#if XC_STAMPS
__device__ unsigned long long g_xp_stamps[24];      // [role: worker, exchange, correlation][S1 work, wait, S2 work, wait, S3 work, wait] + prologue
#define XP_ST_DECL unsigned long long xp_t = __builtin_amdgcn_s_memtime(), xp_acc[7] = {0, 0, 0, 0, 0, 0, 0};
#define XP_ST(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); xp_acc[k] += t_ - xp_t; xp_t = t_; }
#else
#define XP_ST_DECL
#define XP_ST(k)
#endif

__host__ __device__ __forceinline__ bool xc_is_exchange(int id) { return id == XC_LDA_X || id == XC_GGA_X_B88 || id == XC_GGA_X_PBE; }

__device__ __forceinline__ Dual eval_component(int id, Dual R, Dual R13, Dual S)
{
    switch (id) {
        case XC_LDA_X: return f_lda_x(R, R13);
        case XC_LDA_C_VWN: return f_vwn(R, R13, 0.0310907, -0.10498, 3.72744, 12.9352);
        case XC_LDA_C_VWN_RPA: return f_vwn(R, R13, 0.0310907, -0.409286, 13.0720, 42.7198);
        case XC_GGA_X_B88: return f_b88(R, R13, S);
        case XC_GGA_C_LYP: return f_lyp(R, R13, S);
        case XC_GGA_X_PBE: return f_pbe_x(R, R13, S);
        case XC_GGA_C_PBE: return f_pbe_c(R, R13, S);
        default: return mk(0.0);
    }
}

template <bool GGA, int NTC>
__global__ void __launch_bounds__(64 * XP_NW, 1) xc_pipe_kernel(BatchView bv, int only_active)
{
    extern __shared__ double lds[];
    const int f = blockIdx.y;
    if ((only_active & 1) && bv.istate[4 * f] == ST_DONE) return;
    constexpr int PT = XP_PT, RS = PT + 1, PT16 = PT / 16, NTHR = 64 * XP_NW, NWT = 64 * XP_WORK;
    constexpr int NP = 16 * NTC, ARR = GGA ? 4 : 1, SL = ARR * NP * RS;
    constexpr int NJOB = NTC * PT16, DJ = (NJOB + XP_WORK - 1) / XP_WORK, DK = 4 * NTC;      // X jobs, jobs per worker, k-steps
    constexpr int NPAIR = 2 * NTC;                                                         // (row tile, point half) pairs of S3
    const int n = bv.n, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const bool worker = wave < XP_WORK;
    const TopologyDev& tp = bv.topo;
    const GridDev& gd = bv.grid;
    double* slab0 = lds;                                      // two slabs: [ARR][NP][RS] = chi, (gx, gy, gz)
    double* red = lds + 2 * SL;                               // [2][PT][4] rho, grad rho / 2 sums
    double* fc = red + 8 * PT;                                // [2][PT][8] c0x, t2x, rx, ry, rz, c0c, t2c, -
    double* axyz = fc + 16 * PT;                              // [64][3]
    double* pxyz = axyz + 3 * 64;                             // [2][PT][3]
    double* xcw = pxyz + 6 * PT;                              // [8] weights of the functional's components
    int* xcid = (int*)(xcw + 8);                              // [8] their ids (4 doubles)
    int* sdesc = (int*)(xcw + 12);                            // [nshell] ao | l << 12 | atom << 16
    double* radl = xcw + 12 + ((((size_t)(tp.nshell + 1) / 2) + 1) & ~(size_t)1);          // [nshell][2][PT], 16-byte aligned
    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const double* __restrict__ D = bv.D + (size_t)f * n * n;
    const double* __restrict__ wts = gd.weights + (size_t)f * gd.npts;
    const double* __restrict__ radf = gd.rad + (size_t)f * ((size_t)(gd.npts + PT - 1) / PT) * tp.nshell * 2 * PT;

    for (int idx = tid; idx < 2 * SL + 24 * PT; idx += NTHR) lds[idx] = 0.0;      // rows n..NP-1 of both slabs stay zero
    for (int sh = tid; sh < tp.nshell; sh += NTHR) sdesc[sh] = tp.sh_aoff[sh] | (tp.sh_l[sh] << 12) | (tp.sh_atom[sh] << 16);
    for (int idx = tid; idx < 3 * tp.natoms; idx += NTHR) axyz[idx] = xyz[idx];
    // the functional's component table moves to LDS once: the kernel-argument segment it comes from is host-coherent
    // memory, and a scalar load from it inside the tile loop (rolled loop over the components) cost microseconds per tile
    if (tid < 6) { xcid[tid] = tid < bv.xc.ncomp ? bv.xc.id[tid] : 0; xcw[tid] = tid < bv.xc.ncomp ? bv.xc.w[tid] : 0.0; }
    const int ntile = (gd.npts + PT - 1) / PT;
    const int stride = (int)gridDim.x;
    auto load_points = [&](int tile, int b, int q) {
        const int g = tile * PT + q;
        double x = 0.0, y = 0.0, z = 0.0;
        if (g < gd.npts) {
            const int oa = gd.pt_atom[g], it = gd.pt_tmpl[g];
            x = xyz[3 * oa] + gd.tmpl_xyz[3 * it]; y = xyz[3 * oa + 1] + gd.tmpl_xyz[3 * it + 1]; z = xyz[3 * oa + 2] + gd.tmpl_xyz[3 * it + 2];
        }
        double* pp = pxyz + 3 * (b * PT + q);
        pp[0] = x; pp[1] = y; pp[2] = z;
    };
    const int rad_bytes = tp.nshell * 2 * PT * 8;
    auto stage_radial = [&](int tile) {       // workers: the tile's radial block HBM -> LDS by LDS-DMA (no registers)
        if (tile >= ntile) return;
        const char* src = (const char*)(radf + (size_t)tile * tp.nshell * 2 * PT);
        for (int c = wave; c * 1024 < rad_bytes; c += XP_WORK) {
            const int off = c * 1024 + lane * 16;
            if (off < rad_bytes)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                                 (__attribute__((address_space(3))) void*)((char*)radl + c * 1024), 16, 0, 0);
        }
    };
    auto slab = [&](int tile, int b) {        // workers: (shell, point) items, point fastest
        double* chi = slab0 + (size_t)b * SL;
        double* gx = chi + NP * RS; double* gy = gx + (GGA ? NP * RS : 0); double* gz = gy + (GGA ? NP * RS : 0);
        const int g0 = tile * PT;
        for (int idx = tid; idx < tp.nshell * PT; idx += NWT) {
            const int sh = idx / PT, p = idx - sh * PT;
            const int sd = sdesc[sh];
            const int ao = sd & 0xfff, l = (sd >> 12) & 0xf, at = sd >> 16;
            const double* pp = pxyz + 3 * (b * PT + p);
            const double dx = pp[0] - axyz[3 * at], dy = pp[1] - axyz[3 * at + 1], dz = pp[2] - axyz[3 * at + 2];
            const bool in = g0 + p < gd.npts;
            const double* r0 = radl + (size_t)sh * 2 * PT + p;
            emit_shell<GGA, false>(l, ao, dx, dy, dz, in ? r0[0] : 0.0, in ? r0[PT] : 0.0, chi, gx, gy, gz, RS, p, bv.c2s);
        }
    };
    // density fragments of this worker's X jobs, resident for the whole kernel
    double dfrag[DJ][DK];
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
        const int job = wave + XP_WORK * j, mt = job / PT16;
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) {
            const int mu = 16 * mt + lo, nu = 4 * ks + hi;
            dfrag[j][ks] = (worker && job < NJOB && mu < n && nu < n) ? D[(size_t)mu * n + nu] : 0.0;
        }
    }
    auto density = [&](int b) {               // workers: X = D chi of the slab in buffer b, rho and grad rho into red[b]
        const double* chi = slab0 + (size_t)b * SL;
        const double* gx = chi + NP * RS; const double* gy = gx + (GGA ? NP * RS : 0); const double* gz = gy + (GGA ? NP * RS : 0);
#pragma unroll
        for (int jj = 0; jj < DJ; ++jj) {
            const int job = wave + XP_WORK * jj;
            if (job >= NJOB) break;
            const int mt = job / PT16, pt = job - mt * PT16;
            v4f64 xacc = (v4f64){0.0, 0.0, 0.0, 0.0};
            double bw[DK];
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) bw[ks] = chi[(4 * ks + hi) * RS + 16 * pt + lo];
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(dfrag[jj][ks], bw[ks], xacc, 0, 0, 0);
            double rho = 0.0, rx = 0.0, ry = 0.0, rz = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = (16 * mt + hi + 4 * r) * RS + 16 * pt + lo;
                const double x = xacc[r];
                rho += x * chi[o];
                if (GGA) { rx += x * gx[o]; ry += x * gy[o]; rz += x * gz[o]; }
            }
            rho += __shfl_xor(rho, 16, 64); rho += __shfl_xor(rho, 32, 64);
            if (GGA) {
                rx += __shfl_xor(rx, 16, 64); rx += __shfl_xor(rx, 32, 64);
                ry += __shfl_xor(ry, 16, 64); ry += __shfl_xor(ry, 32, 64);
                rz += __shfl_xor(rz, 16, 64); rz += __shfl_xor(rz, 32, 64);
            }
            if (hi == 0) {
                double* rp = red + 4 * (b * PT + 16 * pt + lo);
                atomicAdd(&rp[0], rho);
                if (GGA) { atomicAdd(&rp[1], rx); atomicAdd(&rp[2], ry); atomicAdd(&rp[3], rz); }
            }
        }
    };
    v4f64 vacc[NTC];
#pragma unroll
    for (int j = 0; j < NTC; ++j) vacc[j] = (v4f64){0.0, 0.0, 0.0, 0.0};
    double e_acc = 0.0, n_acc = 0.0;
    // which components this functional wave runs, and in which half of the tile's two overlapped phases.  Everything
    // here is wave-uniform (scalar loads of the kernel argument with a uniform index): a per-lane index into bv.xc made
    // the compiler fetch the ids through VECTOR loads from the kernel-argument segment -- host-coherent memory, several
    // microseconds per access, once per component and tile (profiles/r03_xc_pipe_stamps.txt, first build)
    const bool xwave = wave == XP_WORK, cwave = wave == XP_WORK + 1;
    const int ncomp = bv.xc.ncomp;
    int nmine = 0;
    for (int k = 0; k < ncomp; ++k) nmine += (xc_is_exchange(bv.xc.id[k]) == xwave) ? 1 : 0;      // once, before the tile loop
    const int first_half = (nmine + 1) / 2;
    auto run_components = [&](bool second, Dual R, Dual R13, Dual S, double& fx, double& vr, double& vs) {
        int cnt = 0;
#pragma unroll 1
        for (int k = 0; k < ncomp; ++k) {
            const int id = __builtin_amdgcn_readfirstlane(xcid[k]);
            if (xc_is_exchange(id) != xwave) continue;
            const bool in_second = cnt >= first_half;
            ++cnt;
            if (in_second != second) continue;
            double rv = R.v, cv = R13.v, cr = R13.r, sv = S.v;
            asm volatile("" : "+v"(rv), "+v"(cv), "+v"(cr), "+v"(sv));      // opaque per iteration: see eval_functional
            const Dual Rk = {rv, 1.0, 0.0}, R13k = {cv, cr, 0.0}, Sk = {sv, 0.0, 1.0};
            const Dual d = eval_component(id, Rk, R13k, Sk);
            const double wk = xcw[k];
            fx += wk * d.v; vr += wk * d.r; vs += wk * d.s;
        }
    };

    XP_ST_DECL
    // ---- prologue: tile 0 through slab and density
    int tile = blockIdx.x;
    if (tid < PT) load_points(tile, 0, tid);
    if (worker) stage_radial(tile);
    __syncthreads();
    if (worker) slab(tile, 0);
    __syncthreads();
    if (worker) {
        density(0);
        stage_radial(tile + stride);
        if (wave == 0 && lane < PT) load_points(tile + stride, 1, lane);
    }
    __syncthreads();
    XP_ST(6)
    // the functional waves fetch the quadrature weights of a tile one tile ahead
    double w_next = (!worker && lane < PT && tile * PT + lane < gd.npts) ? wts[tile * PT + lane] : 0.0;

    int cur = 0;
    for (; tile < ntile; tile += stride, cur ^= 1) {
        const int nxt = cur ^ 1, next = tile + stride;
        const bool has_next = next < ntile;
        const int s0 = tile * PT;
        // ---- S1: slab of the next tile | functional, first half
        Dual R = mk(0.0), S = mk(0.0), R13 = mk(0.0);
        double fx = 0.0, vr = 0.0, vs = 0.0, rho = 0.0, rx = 0.0, ry = 0.0, rz = 0.0, w = 0.0;
        bool live = false;
        if (worker) {
            if (has_next) slab(next, nxt);
        } else if (lane < PT) {
            const double* rp = red + 4 * (cur * PT + lane);
            rho = rp[0]; rx = 2.0 * rp[1]; ry = 2.0 * rp[2]; rz = 2.0 * rp[3];
            w = w_next;
            w_next = (has_next && next * PT + lane < gd.npts) ? wts[next * PT + lane] : 0.0;
            live = rho > XC_DENS_THRESHOLD && !(g_xc_probe & 32);
            if (live) {
                R = {rho, 1.0, 0.0};
                S = {fmax(GGA ? rx * rx + ry * ry + rz * rz : 0.0, 1.0e-40), 0.0, 1.0};
                R13 = dcbrt(R);
                if (!(g_xc_probe & 16)) run_components(false, R, R13, S, fx, vr, vs);
            }
        }
        XP_ST(0)
        __syncthreads();
        XP_ST(1)
        // ---- S2: densities of the next tile | functional, second half -> partial coefficients
        if (worker) {
            if (has_next) {
                stage_radial(next + stride);
                double px = 0.0, py = 0.0, pz = 0.0;
                const bool pts = wave == XP_WORK - 1 && lane < PT;
                if (pts) {        // issued before the MFMA job, stored after it: the dependent global loads ride under it
                    const int g = (next + stride) * PT + lane;
                    if (g < gd.npts) {
                        const int oa = gd.pt_atom[g], it = gd.pt_tmpl[g];
                        px = xyz[3 * oa] + gd.tmpl_xyz[3 * it]; py = xyz[3 * oa + 1] + gd.tmpl_xyz[3 * it + 1]; pz = xyz[3 * oa + 2] + gd.tmpl_xyz[3 * it + 2];
                    }
                }
                density(nxt);
                if (pts) { double* pp = pxyz + 3 * (cur * PT + lane); pp[0] = px; pp[1] = py; pp[2] = pz; }
            }
        } else if (lane < PT) {
            if (cwave) { double* rp = red + 4 * (cur * PT + lane); rp[0] = 0.0; rp[1] = 0.0; rp[2] = 0.0; rp[3] = 0.0; }
            if (live && !(g_xc_probe & 16)) run_components(true, R, R13, S, fx, vr, vs);
            e_acc += w * fx;
            double* cp = fc + 8 * (cur * PT + lane);
            if (xwave) {
                n_acc += w * rho;
                cp[0] = 0.5 * w * vr; cp[1] = 2.0 * w * vs; cp[2] = rx; cp[3] = ry; cp[4] = rz;
            } else {
                cp[5] = 0.5 * w * vr; cp[6] = 2.0 * w * vs;
            }
        }
        XP_ST(2)
        __syncthreads();
        XP_ST(3)
        // ---- S3: A += a chi^T of the current tile; wave = (row tile mt, half kh of the points), a formed on the fly
        if (wave < NPAIR) {
            const int mt = wave >> 1, kh = wave & 1;
            const double* chi = slab0 + (size_t)cur * SL;
            const double* gx = chi + NP * RS; const double* gy = gx + (GGA ? NP * RS : 0); const double* gz = gy + (GGA ? NP * RS : 0);
            double av[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int p = 16 * kh + 4 * ks + hi;
                const double* cp = fc + 8 * (cur * PT + p);
                const int o = (16 * mt + lo) * RS + p;
                double a = (cp[0] + cp[5]) * chi[o];
                if (GGA) {
                    const double t2 = cp[1] + cp[6];
                    a += t2 * (cp[2] * gx[o] + cp[3] * gy[o] + cp[4] * gz[o]);
                }
                av[ks] = a;
            }
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt) {
                const double* __restrict__ br = chi + (size_t)(16 * nt + lo) * RS + 16 * kh + hi;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) vacc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], br[4 * ks], vacc[nt], 0, 0, 0);
            }
        }
        XP_ST(4)
        __syncthreads();
        XP_ST(5)
    }
#if XC_STAMPS
    if (lane == 0) {
        const int role = worker ? 0 : (xwave ? 1 : 2);
        for (int k = 0; k < 6; ++k) atomicAdd(&g_xp_stamps[6 * role + k], xp_acc[k]);
        atomicAdd(&g_xp_stamps[18], xp_acc[6]);
    }
#endif
    // flush: lane holds A[mu = 16 mt + hi + 4 r][nu = 16 nt + lo] of its (mt, kh) pair
    double* Vx = bv.Vxc + (size_t)f * n * n;
    if (wave < NPAIR) {
        const int mt = wave >> 1;
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mu = 16 * mt + hi + 4 * r, nu = 16 * nt + lo;
                const double v = vacc[nt][r];
                if (mu < n && nu < n && v != 0.0) atomicAdd(&Vx[mu * n + nu], v);
            }
    }
    if (!worker) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { e_acc += __shfl_down(e_acc, off, 64); n_acc += __shfl_down(n_acc, off, 64); }
        if (lane == 0) {
            atomicAdd(&bv.scal[(size_t)f * 8 + 5], e_acc);
            if (xwave) atomicAdd(&bv.scal[(size_t)f * 8 + 6], n_acc);
        }
    }
}

template <bool GGA, int NTC>
static bool xc_pipe_launch(const BatchView& bv, int oa, hipStream_t s)
{
    constexpr int PT = XP_PT;
    const size_t sl = (size_t)(GGA ? 4 : 1) * 16 * NTC * (PT + 1);
    const size_t doubles = 2 * sl + 24 * PT + 3 * 64 + 6 * PT + 12 + (((((size_t)bv.topo.nshell + 1) / 2) + 1) & ~(size_t)1) + (size_t)bv.topo.nshell * 2 * PT;
    const size_t lds = sizeof(double) * doubles;
    if (lds > (size_t)160 * 1024 - 256) return false;
    auto kern = xc_pipe_kernel<GGA, NTC>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int ntiles = (bv.grid.npts + PT - 1) / PT;
    int gx = (2048 + bv.nfrag - 1) / bv.nfrag;        // one workgroup per CU: ~8 rounds of 256 over the batch, many tiles each
    if (gx > ntiles) gx = ntiles;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(kern, dim3(gx, bv.nfrag), dim3(64 * XP_NW), lds, s, bv, oa);
#if XC_STAMPS
    (void)hipStreamSynchronize(s);
    unsigned long long h[24];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_xp_stamps), sizeof(h));
    const char* role[3] = {"workers (6 waves)", "exchange wave", "correlation wave"};
    for (int r = 0; r < 3; ++r)
        std::fprintf(stderr, "xc pipe stamps n=%d nfrag=%d %s: S1 %llu wait %llu | S2 %llu wait %llu | S3 %llu wait %llu\n", bv.n, bv.nfrag, role[r],
                     h[6 * r], h[6 * r + 1], h[6 * r + 2], h[6 * r + 3], h[6 * r + 4], h[6 * r + 5]);
#endif
    return true;
}

// n <= 64, restricted, s/p/d shells, radial cache at 32-point tiles: the pipelined kernel (measured level with the tile
// kernel once the functional's cases stayed behind their branch; kept for A/B runs, MQC_HIP_XC_PIPE=1 turns it on)
template <bool GGA>
static bool xc_pipe_dispatch(const BatchView& bv, int oa, hipStream_t s)
{
    static const bool on = [] { const char* e = std::getenv("MQC_HIP_XC_PIPE"); return e && e[0] == '1'; }();
    if (!on || bv.uhf || bv.n > 64 || bv.topo.lmax > 2 || !bv.grid.rad || bv.grid.rad_pt != XP_PT) return false;
    bool has_x = false, has_c = false;
    for (int k = 0; k < bv.xc.ncomp; ++k) { if (xc_is_exchange(bv.xc.id[k])) has_x = true; else has_c = true; }
    if (!has_x) return false;             // the exchange wave also carries the gradient and the electron count
    (void)has_c;
    const int nt = (bv.n + 15) / 16;
    if (nt == 1) return xc_pipe_launch<GGA, 1>(bv, oa, s);
    if (nt == 2) return xc_pipe_launch<GGA, 2>(bv, oa, s);
    if (nt == 3) return xc_pipe_launch<GGA, 3>(bv, oa, s);
    return xc_pipe_launch<GGA, 4>(bv, oa, s);
}

// ------------------------------------------------------------------ the split quadrature (n <= 64): three kernels
// One tile kernel has to hold the slab, the density fragments, the A accumulators AND the functional's dual-number
// stream: 230-256 registers and 50-125 KB of LDS, so eight waves per CU, and every phase is exposed latency (MFMA pipe
// 17 % busy, profiles/r02_pmc_summary.json).  Cut at the two points where only four numbers per grid point cross:
//   xc_density_kernel     slab -> X = D chi on the matrix cores -> rho, grad rho / 2 per point   -> pt4[f][point][4]
//   xc_functional_kernel  one THREAD per point, all 64 lanes of every wave busy: f, v_rho, v_sigma -> E_xc, N_e and the
//                         coefficients (w v_rho / 2, 2 w v_sigma grad rho) over the same four numbers
//   xc_potential_kernel   slab again, but only chi and a = c0 chi + c . grad chi are kept (25 KB of LDS at n = 48) ->
//                         A += a chi^T on the matrix cores, accumulators resident across the workgroup's tiles
// Each kernel needs a fraction of the registers and LDS (three to five workgroups per CU), no wave ever waits for a
// single-wave phase, and the functional runs at full width.  Price: the slab's angular part is formed twice and the
// radial cache is read twice.  Same arithmetic as the tile kernel (mqc_libcint_xc.F90:796-927).
constexpr int XS_PT = 32, XS_SI = 4;      // tile points; slab items per thread prefetched (threads: 64 x NW, a multiple of XS_PT)
// row stride of the potential kernel's two LDS arrays.  33: the MFMA operand reads (row <- lane & 15, point <- lane >> 4)
// of lanes (row, point) and (row + 1, point - 1) share a bank (SQ_LDS_BANK_CONFLICT = 29 % of the LDS cycles); 34 =
// 2 (mod 32) is conflict-free for them -- and measured no faster on one box (805 / 807 against 795 / 796 ms per B3LYP
// evaluation, -DXS_RS3=34): the kernel is not bound by the LDS array
#ifndef XS_RS3
#define XS_RS3 33
#endif

// value and a = c0 chi + c . grad chi of one shell at one point (l <= 2; l = 3 with LF), from its radial value and
// derivative factor
template <bool GGA, bool LF = false>
__device__ __forceinline__ void emit_shell_a(int l, int ao, double dx, double dy, double dz, double rad, double drad,
                                             double c0, double c1, double c2, double c3,
                                             double* __restrict__ chi, double* __restrict__ av, int ptp, int p)
{
    if (rad == 0.0 && drad == 0.0) {
        // every primitive of the shell was below the exponent cutoff at this point (the radial cache stores exact zeros):
        // the functions and a are zero -- written without the angular arithmetic
        const int nf = 2 * l + 1;
        for (int m = 0; m < nf; ++m) { chi[(ao + m) * ptp + p] = 0.0; av[(ao + m) * ptp + p] = 0.0; }
        return;
    }
    const double cd = GGA ? (c1 * dx + c2 * dy + c3 * dz) * drad : 0.0;        // c . r  R'/r
    if (l == 0) {
        chi[ao * ptp + p] = rad;
        av[ao * ptp + p] = c0 * rad + cd;
    } else if (l == 1) {
        const double vx = dx * rad, vy = dy * rad, vz = dz * rad;
        chi[ao * ptp + p] = vx; chi[(ao + 1) * ptp + p] = vy; chi[(ao + 2) * ptp + p] = vz;
        av[ao * ptp + p] = c0 * vx + (GGA ? c1 * rad + dx * cd : 0.0);
        av[(ao + 1) * ptp + p] = c0 * vy + (GGA ? c2 * rad + dy * cd : 0.0);
        av[(ao + 2) * ptp + p] = c0 * vz + (GGA ? c3 * rad + dz * cd : 0.0);
    } else if (LF && l == 3) {
        // the seven real solid harmonics of an f shell over the ten cubic monomials (same table as emit_shell)
        constexpr double T3[7][10] = {
            {0, 1.7701307697799307, 0, 0, 0, 0, -0.59004358992664352, 0, 0, 0},
            {0, 0, 0, 0, 2.8906114426405543, 0, 0, 0, 0, 0},
            {0, -0.45704579946446572, 0, 0, 0, 0, -0.45704579946446572, 0, 1.8281831978578629, 0},
            {0, 0, -1.1195289977703462, 0, 0, 0, 0, -1.1195289977703462, 0, 0.7463526651802308},
            {-0.45704579946446572, 0, 0, -0.45704579946446572, 0, 1.8281831978578629, 0, 0, 0, 0},
            {0, 0, 1.4453057213202771, 0, 0, 0, 0, -1.4453057213202771, 0, 0},
            {0.59004358992664352, 0, 0, -1.7701307697799307, 0, 0, 0, 0, 0, 0}};
        const double x2 = dx * dx, y2 = dy * dy, z2 = dz * dz, xy = dx * dy, xz = dx * dz, yz = dy * dz;
        const double cv[10] = {x2 * dx, x2 * dy, x2 * dz, dx * y2, xy * dz, dx * z2, y2 * dy, y2 * dz, dy * z2, z2 * dz};
        const double cgx[10] = {3.0 * x2, 2.0 * xy, 2.0 * xz, y2, yz, z2, 0.0, 0.0, 0.0, 0.0};
        const double cgy[10] = {0.0, x2, 0.0, 2.0 * xy, xz, 0.0, 3.0 * y2, 2.0 * yz, z2, 0.0};
        const double cgz[10] = {0.0, 0.0, x2, 0.0, xy, 2.0 * xz, 0.0, y2, 2.0 * yz, 3.0 * z2};
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            double v = 0.0, g = 0.0;
#pragma unroll
            for (int k = 0; k < 10; ++k)
                if (T3[m][k] != 0.0) {
                    v += T3[m][k] * cv[k];
                    if (GGA) g += T3[m][k] * (c1 * cgx[k] + c2 * cgy[k] + c3 * cgz[k]);
                }
            chi[(ao + m) * ptp + p] = v * rad;
            av[(ao + m) * ptp + p] = c0 * v * rad + (GGA ? g * rad + v * cd : 0.0);
        }
    } else {
        const double cv[6] = {dx * dx, dx * dy, dx * dz, dy * dy, dy * dz, dz * dz};
        const double cgx[6] = {2.0 * dx, dy, dz, 0.0, 0.0, 0.0};
        const double cgy[6] = {0.0, dx, 0.0, 2.0 * dy, dz, 0.0};
        const double cgz[6] = {0.0, 0.0, dx, 0.0, dy, 2.0 * dz};
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            double v = 0.0, g = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double w = c2s_coef<2>(nullptr, m, k);
                v += w * cv[k];
                if (GGA) g += w * (c1 * cgx[k] + c2 * cgy[k] + c3 * cgz[k]);
            }
            chi[(ao + m) * ptp + p] = v * rad;
            av[(ao + m) * ptp + p] = c0 * v * rad + (GGA ? g * rad + v * cd : 0.0);
        }
    }
}

template <bool GGA, int NTC, int NW, bool LF>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : (NTC >= 4 ? 2 : 3)) xc_density_kernel(BatchView bv, int only_active)
{
    extern __shared__ double lds[];
    const int f = blockIdx.y;
    if ((only_active & 1) && bv.istate[4 * f] == ST_DONE) return;
    constexpr int XS_NT = 64 * NW, KPART = NW / 2;           // threads; parts the functions are cut into (2 or 4)
    constexpr int PT = XS_PT, RS = PT + 1, NP = 16 * NTC, ARR = GGA ? 4 : 1, SL = ARR * NP * RS, HK = 4 * NTC / KPART;   // HK: k-steps of one part of the functions
    const int n = bv.n, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const TopologyDev& tp = bv.topo;
    const GridDev& gd = bv.grid;
    double* chi = lds;
    double* gx = chi + NP * RS; double* gy = gx + (GGA ? NP * RS : 0); double* gz = gy + (GGA ? NP * RS : 0);
    double* red = lds + SL;                                   // [PT][4]
    double* pxyz = red + 4 * PT;                              // [2][PT][3]
    double* axyz = pxyz + 6 * PT;                             // [natoms][3]
    int* sdesc = (int*)(axyz + ((3 * tp.natoms + 1) & ~1));   // [nshell]
    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const double* __restrict__ D = bv.D + (size_t)f * n * n;
    const int ntile = (gd.npts + PT - 1) / PT;
    const double* __restrict__ radf = gd.rad + (size_t)f * ntile * tp.nshell * 2 * PT;
    double* __restrict__ out4 = gd.pt4 + (size_t)f * ntile * PT * 4;

    for (int idx = tid; idx < SL + 4 * PT; idx += XS_NT) lds[idx] = 0.0;          // rows n..NP-1 stay zero
    for (int sh = tid; sh < tp.nshell; sh += XS_NT) sdesc[sh] = tp.sh_aoff[sh] | (tp.sh_l[sh] << 12) | (tp.sh_atom[sh] << 16);
    for (int idx = tid; idx < 3 * tp.natoms; idx += XS_NT) axyz[idx] = xyz[idx];
    // this wave's part of X = D chi: point tile pt, part kh of the functions (halves with four waves, quarters with
    // eight), every row tile mt -- the B operand
    // (AO values of its points) is then the same for all its jobs, and rho / grad rho of its 16 points add up in
    // registers over the row tiles: one reduction per tile and wave
    const int pt = (wave / KPART) & 1, kh = wave % KPART;
    double dfrag[NTC][HK];
#pragma unroll
    for (int mt = 0; mt < NTC; ++mt)
#pragma unroll
        for (int ks = 0; ks < HK; ++ks) {
            const int mu = 16 * mt + lo, nu = 4 * (kh * HK + ks) + hi;
            dfrag[mt][ks] = (mu < n && nu < n) ? D[(size_t)mu * n + nu] : 0.0;
        }
    auto fetch_point = [&](int tile, int q, double& x, double& y, double& z) {
        const int g = tile * PT + q;
        x = 0.0; y = 0.0; z = 0.0;
        if (g < gd.npts) {
            const int oa = gd.pt_atom[g], it = gd.pt_tmpl[g];
            x = xyz[3 * oa] + gd.tmpl_xyz[3 * it]; y = xyz[3 * oa + 1] + gd.tmpl_xyz[3 * it + 1]; z = xyz[3 * oa + 2] + gd.tmpl_xyz[3 * it + 2];
        }
    };
    const int stride = (int)gridDim.x;
    int tile = blockIdx.x;
    const bool ptthr = tid >= XS_NT - PT;                     // the last PT threads carry the grid points a tile ahead
    const int q = tid - (XS_NT - PT);
    double nx = 0.0, ny = 0.0, nz = 0.0;
    if (ptthr) { fetch_point(tile, q, nx, ny, nz); double* pp = pxyz + 3 * q; pp[0] = nx; pp[1] = ny; pp[2] = nz; }
    // radial values of this thread's slab items, fetched a TILE AHEAD into registers (issued after the slab, landing
    // under the MFMA phase): read inside the slab loop every item waited out an HBM round trip
    const int nitem = tp.nshell * PT;
    double rv[XS_SI], dv[XS_SI];
    auto radial_fetch = [&](int t) {
#pragma unroll
        for (int k = 0; k < XS_SI; ++k) {
            const int idx = tid + XS_NT * k;
            rv[k] = 0.0; dv[k] = 0.0;
            if (idx < nitem && t < ntile) {
                const int sh = idx / PT, p = idx - sh * PT;
                if (t * PT + p < gd.npts) {
                    const double* r0 = radf + ((size_t)t * tp.nshell + sh) * 2 * PT + p;
                    rv[k] = r0[0];
                    if (GGA) dv[k] = r0[PT];
                }
            }
        }
    };
    radial_fetch(tile);
    __syncthreads();
    int pbuf = 0;
    for (; tile < ntile; tile += stride, pbuf ^= 1) {
        if (ptthr) fetch_point(tile + stride, q, nx, ny, nz);       // in flight under the slab
        // ---- slab: (shell, point) items, point fastest (lanes along the points)
        {
            const int g0 = tile * PT;
            // XS_NT is a multiple of PT: all items of a thread sit on ONE grid point (p = tid mod PT), shells 8 apart
            const int p0 = tid & (PT - 1);
            const double* pp0 = pxyz + 3 * (pbuf * PT + p0);
            const double ptx = pp0[0], pty = pp0[1], ptz = pp0[2];
#pragma unroll
            for (int k = 0; k < XS_SI; ++k) {
                const int sh = (tid >> 5) + (XS_NT / PT) * k;
                if (sh < tp.nshell) {
                    const int sd = sdesc[sh];
                    const int ao = sd & 0xfff, l = (sd >> 12) & 0xf, at = sd >> 16;
                    const double dx = ptx - axyz[3 * at], dy = pty - axyz[3 * at + 1], dz = ptz - axyz[3 * at + 2];
                    emit_shell<GGA, LF>(l, ao, dx, dy, dz, rv[k], dv[k], chi, gx, gy, gz, RS, p0, bv.c2s);
                }
            }
            const double* __restrict__ radt = radf + (size_t)tile * tp.nshell * 2 * PT;
            for (int idx = tid + XS_NT * XS_SI; idx < nitem; idx += XS_NT) {        // more than XS_SI items per thread: direct
                const int sh = idx / PT, p = idx - sh * PT;
                const int sd = sdesc[sh];
                const int ao = sd & 0xfff, l = (sd >> 12) & 0xf, at = sd >> 16;
                const double* pp = pxyz + 3 * (pbuf * PT + p);
                const double dx = pp[0] - axyz[3 * at], dy = pp[1] - axyz[3 * at + 1], dz = pp[2] - axyz[3 * at + 2];
                const bool in = g0 + p < gd.npts;
                const double* r0 = radt + (size_t)sh * 2 * PT + p;
                emit_shell<GGA, LF>(l, ao, dx, dy, dz, in ? r0[0] : 0.0, (GGA && in) ? r0[PT] : 0.0, chi, gx, gy, gz, RS, p, bv.c2s);
            }
        }
        radial_fetch(tile + stride);
        if (ptthr) { double* pp = pxyz + 3 * ((pbuf ^ 1) * PT + q); pp[0] = nx; pp[1] = ny; pp[2] = nz; }
        __syncthreads();
        // ---- X = D chi for this wave's (pt, kh), all row tiles; rho and grad rho of its 16 points
        {
            double bw[HK];
#pragma unroll
            for (int ks = 0; ks < HK; ++ks) bw[ks] = chi[(4 * (kh * HK + ks) + hi) * RS + 16 * pt + lo];
            double rho = 0.0, rx = 0.0, ry = 0.0, rz = 0.0;
            // the row tiles' accumulation chains are independent: issued interleaved (k-step outer), a chain's next MFMA
            // is NTC instructions behind its predecessor instead of waiting out its latency
            v4f64 xacc[NTC];
#pragma unroll
            for (int mt = 0; mt < NTC; ++mt) xacc[mt] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < HK; ++ks)
#pragma unroll
                for (int mt = 0; mt < NTC; ++mt) xacc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(dfrag[mt][ks], bw[ks], xacc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < NTC; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = (16 * mt + hi + 4 * r) * RS + 16 * pt + lo;
                    const double x = xacc[mt][r];
                    rho += x * chi[o];
                    if (GGA) { rx += x * gx[o]; ry += x * gy[o]; rz += x * gz[o]; }
                }
            rho += __shfl_xor(rho, 16, 64); rho += __shfl_xor(rho, 32, 64);
            if (GGA) {
                rx += __shfl_xor(rx, 16, 64); rx += __shfl_xor(rx, 32, 64);
                ry += __shfl_xor(ry, 16, 64); ry += __shfl_xor(ry, 32, 64);
                rz += __shfl_xor(rz, 16, 64); rz += __shfl_xor(rz, 32, 64);
            }
            if (hi == 0) {
                double* rp = red + 4 * (16 * pt + lo);
                atomicAdd(&rp[0], rho);
                if (GGA) { atomicAdd(&rp[1], rx); atomicAdd(&rp[2], ry); atomicAdd(&rp[3], rz); }
            }
        }
        __syncthreads();
        // ---- the tile's four numbers per point go out (1 KiB, contiguous); the sums start at zero again
        if (tid < 4 * PT) {
            out4[(size_t)tile * PT * 4 + tid] = red[tid];
            red[tid] = 0.0;
        }
    }
}

template <bool GGA>
__global__ void __launch_bounds__(256) xc_functional_kernel(BatchView bv, int only_active)
{
    const int f = blockIdx.y;
    if ((only_active & 1) && bv.istate[4 * f] == ST_DONE) return;
    const GridDev& gd = bv.grid;
    const int npad = ((gd.npts + XS_PT - 1) / XS_PT) * XS_PT;
    double* __restrict__ v4 = gd.pt4 + (size_t)f * npad * 4;
    const double* __restrict__ wts = gd.weights + (size_t)f * gd.npts;
    double e_acc = 0.0, n_acc = 0.0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npad; p += gridDim.x * blockDim.x) {
        const double2 a = *(const double2*)(v4 + 4 * (size_t)p), b = *(const double2*)(v4 + 4 * (size_t)p + 2);
        const double rho = a.x, rx = 2.0 * a.y, ry = 2.0 * b.x, rz = 2.0 * b.y;
        const double w = p < gd.npts ? wts[p] : 0.0;
        const double sigma = GGA ? rx * rx + ry * ry + rz * rz : 0.0;
        double fx, vr, vs;
        eval_functional(bv.xc, rho, sigma, fx, vr, vs);
        e_acc += w * fx; n_acc += w * rho;
        const double t2 = 2.0 * w * vs;
        *(double2*)(v4 + 4 * (size_t)p) = make_double2(0.5 * w * vr, t2 * rx);
        *(double2*)(v4 + 4 * (size_t)p + 2) = make_double2(t2 * ry, t2 * rz);
    }
    __shared__ double part[2][4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { e_acc += __shfl_down(e_acc, off, 64); n_acc += __shfl_down(n_acc, off, 64); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = e_acc; part[1][threadIdx.x >> 6] = n_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&bv.scal[(size_t)f * 8 + 5], part[0][0] + part[0][1] + part[0][2] + part[0][3]);
        atomicAdd(&bv.scal[(size_t)f * 8 + 6], part[1][0] + part[1][1] + part[1][2] + part[1][3]);
    }
}

template <bool GGA, int NTC, int NW, bool LF>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 2 : ((NTC >= 4 || LF) ? 2 : 3)) xc_potential_kernel(BatchView bv, int only_active)
{
    extern __shared__ double lds[];
    const int f = blockIdx.y;
    if ((only_active & 1) && bv.istate[4 * f] == ST_DONE) return;
    constexpr int XS_NT = 64 * NW;
    constexpr int PT = XS_PT, RS = XS_RS3, NP = 16 * NTC, SL = 2 * NP * RS;
    constexpr int NU = 2 * NTC * NTC, JU = (NU + NW - 1) / NW;   // (output tile, half of the points) units, units per wave
    const int n = bv.n, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const TopologyDev& tp = bv.topo;
    const GridDev& gd = bv.grid;
    double* chi = lds;                                        // [NP][RS]
    double* av = chi + NP * RS;                               // [NP][RS]  a = c0 chi + c . grad chi
    double* coef = lds + SL;                                  // [2][PT][4]
    double* pxyz = coef + 8 * PT;                             // [2][PT][3]
    double* axyz = pxyz + 6 * PT;                             // [natoms][3]
    int* sdesc = (int*)(axyz + ((3 * tp.natoms + 1) & ~1));   // [nshell]
    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const int ntile = (gd.npts + PT - 1) / PT;
    const double* __restrict__ radf = gd.rad + (size_t)f * ntile * tp.nshell * 2 * PT;
    const double* __restrict__ c4 = gd.pt4 + (size_t)f * ntile * PT * 4;

    for (int idx = tid; idx < SL; idx += XS_NT) lds[idx] = 0.0;
    for (int sh = tid; sh < tp.nshell; sh += XS_NT) sdesc[sh] = tp.sh_aoff[sh] | (tp.sh_l[sh] << 12) | (tp.sh_atom[sh] << 16);
    for (int idx = tid; idx < 3 * tp.natoms; idx += XS_NT) axyz[idx] = xyz[idx];
    auto fetch_point = [&](int tile, int q, double& x, double& y, double& z) {
        const int g = tile * PT + q;
        x = 0.0; y = 0.0; z = 0.0;
        if (g < gd.npts) {
            const int oa = gd.pt_atom[g], it = gd.pt_tmpl[g];
            x = xyz[3 * oa] + gd.tmpl_xyz[3 * it]; y = xyz[3 * oa + 1] + gd.tmpl_xyz[3 * it + 1]; z = xyz[3 * oa + 2] + gd.tmpl_xyz[3 * it + 2];
        }
    };
    v4f64 vacc[JU];
#pragma unroll
    for (int j = 0; j < JU; ++j) vacc[j] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const int stride = (int)gridDim.x;
    int tile = blockIdx.x;
    // a tile ahead: threads 0 .. 4 PT - 1 carry the coefficients, the last PT threads the grid points
    const bool cthr = tid < 4 * PT, ptthr = tid >= XS_NT - PT;
    const int q = tid - (XS_NT - PT);
    double cnext = 0.0, nx = 0.0, ny = 0.0, nz = 0.0;
    if (cthr) coef[tid] = c4[(size_t)tile * PT * 4 + tid];
    if (ptthr) { fetch_point(tile, q, nx, ny, nz); double* pp = pxyz + 3 * q; pp[0] = nx; pp[1] = ny; pp[2] = nz; }
    const int nitem = tp.nshell * PT;
    double rv[XS_SI], dv[XS_SI];
    auto radial_fetch = [&](int t) {          // this thread's slab items of tile t, a tile ahead (see xc_density_kernel)
#pragma unroll
        for (int k = 0; k < XS_SI; ++k) {
            const int idx = tid + XS_NT * k;
            rv[k] = 0.0; dv[k] = 0.0;
            if (idx < nitem && t < ntile) {
                const int sh = idx / PT, p = idx - sh * PT;
                if (t * PT + p < gd.npts) {
                    const double* r0 = radf + ((size_t)t * tp.nshell + sh) * 2 * PT + p;
                    rv[k] = r0[0];
                    if (GGA) dv[k] = r0[PT];
                }
            }
        }
    };
    radial_fetch(tile);
    __syncthreads();
    int buf = 0;
    for (; tile < ntile; tile += stride, buf ^= 1) {
        const int next = tile + stride;
        if (cthr && next < ntile) cnext = c4[(size_t)next * PT * 4 + tid];
        if (ptthr) fetch_point(next, q, nx, ny, nz);
        // ---- slab: chi and a of every (shell, point)
        {
            const int g0 = tile * PT;
            // all items of a thread sit on ONE grid point: its coordinates and coefficients are read once per tile
            const int p0 = tid & (PT - 1);
            const double* pp0 = pxyz + 3 * (buf * PT + p0);
            const double ptx = pp0[0], pty = pp0[1], ptz = pp0[2];
            const double* cp = coef + 4 * (buf * PT + p0);
            const double c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
#pragma unroll
            for (int k = 0; k < XS_SI; ++k) {
                const int sh = (tid >> 5) + (XS_NT / PT) * k;
                if (sh < tp.nshell) {
                    const int sd = sdesc[sh];
                    const int ao = sd & 0xfff, l = (sd >> 12) & 0xf, at = sd >> 16;
                    const double dx = ptx - axyz[3 * at], dy = pty - axyz[3 * at + 1], dz = ptz - axyz[3 * at + 2];
                    emit_shell_a<GGA, LF>(l, ao, dx, dy, dz, rv[k], dv[k], c0, c1, c2, c3, chi, av, RS, p0);
                }
            }
            const double* __restrict__ radt = radf + (size_t)tile * tp.nshell * 2 * PT;
            for (int idx = tid + XS_NT * XS_SI; idx < nitem; idx += XS_NT) {
                const int sh = idx / PT, p = idx - sh * PT;
                const int sd = sdesc[sh];
                const int ao = sd & 0xfff, l = (sd >> 12) & 0xf, at = sd >> 16;
                const double* pp = pxyz + 3 * (buf * PT + p);
                const double dx = pp[0] - axyz[3 * at], dy = pp[1] - axyz[3 * at + 1], dz = pp[2] - axyz[3 * at + 2];
                const bool in = g0 + p < gd.npts;
                const double* r0 = radt + (size_t)sh * 2 * PT + p;
                const double* cp = coef + 4 * (buf * PT + p);
                emit_shell_a<GGA, LF>(l, ao, dx, dy, dz, in ? r0[0] : 0.0, (GGA && in) ? r0[PT] : 0.0, cp[0], cp[1], cp[2], cp[3], chi, av, RS, p);
            }
        }
        radial_fetch(next);
        if (cthr) coef[4 * ((buf ^ 1) * PT) + tid] = cnext;
        if (ptthr) { double* pp = pxyz + 3 * ((buf ^ 1) * PT + q); pp[0] = nx; pp[1] = ny; pp[2] = nz; }
        __syncthreads();
        // ---- A += a chi^T: unit = (output tile t = mt NTC + nt, half kh of the tile's points)
#pragma unroll
        for (int j = 0; j < JU; ++j) {
            const int u = wave + NW * j;
            if (u < NU) {
                const int t = u >> 1, kh = u & 1, mt = t / NTC, nt = t - mt * NTC;
                const double* __restrict__ ar = av + (size_t)(16 * mt + lo) * RS + 16 * kh + hi;
                const double* __restrict__ br = chi + (size_t)(16 * nt + lo) * RS + 16 * kh + hi;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) vacc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[4 * ks], br[4 * ks], vacc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    double* Vx = bv.Vxc + (size_t)f * n * n;
#pragma unroll
    for (int j = 0; j < JU; ++j) {
        const int u = wave + NW * j;
        if (u < NU) {
            const int t = u >> 1, mt = t / NTC, nt = t - mt * NTC;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mu = 16 * mt + hi + 4 * r, nu = 16 * nt + lo;
                const double v = vacc[j][r];
                if (mu < n && nu < n && v != 0.0) atomicAdd(&Vx[mu * n + nu], v);
            }
        }
    }
}

template <bool GGA, int NTC, int NW, bool LF>
static bool xc_split_launch(const BatchView& bv, int oa, hipStream_t s)
{
    constexpr int PT = XS_PT, NTHR = 64 * NW;
    const size_t misc = 6 * PT + (((size_t)3 * bv.topo.natoms + 1) & ~(size_t)1) + ((size_t)bv.topo.nshell + 1) / 2;
    const size_t lds1 = sizeof(double) * ((size_t)(GGA ? 4 : 1) * 16 * NTC * (PT + 1) + 4 * PT + misc);
    const size_t lds3 = sizeof(double) * ((size_t)2 * 16 * NTC * XS_RS3 + 8 * PT + misc);
    if (lds1 > (size_t)160 * 1024 - 256 || lds3 > (size_t)160 * 1024 - 256) return false;
    auto k1 = xc_density_kernel<GGA, NTC, NW, LF>;
    auto k3 = xc_potential_kernel<GGA, NTC, NW, LF>;
    (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    (void)hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
    const int ntiles = (bv.grid.npts + PT - 1) / PT;
    // density: no state carried from tile to tile, any number of workgroups; ~8 rounds of the resident set over the batch
    int g1 = ((NW == 8 ? 2048 : 6144) + bv.nfrag - 1) / bv.nfrag;
    if (g1 > ntiles) g1 = ntiles;
    if (g1 < 1) g1 = 1;
    // potential: the accumulators are flushed once per workgroup (n^2 atomics), so a workgroup takes many tiles
    int g3 = (3072 + bv.nfrag - 1) / bv.nfrag;
    if (g3 > (ntiles + 15) / 16) g3 = (ntiles + 15) / 16;
    if (g3 < 1) g3 = 1;
    const int npad = ntiles * PT;
    int g2 = (npad + 255) / 256;
    if (g2 > 64) g2 = 64;
    hipLaunchKernelGGL(k1, dim3(g1, bv.nfrag), dim3(NTHR), lds1, s, bv, oa);
    hipLaunchKernelGGL(xc_functional_kernel<GGA>, dim3(g2, bv.nfrag), dim3(256), 0, s, bv, oa);
    hipLaunchKernelGGL(k3, dim3(g3, bv.nfrag), dim3(NTHR), lds3, s, bv, oa);
    return true;
}

// n <= 96, restricted, s/p/d (f from n > 32) shells, radial cache at 32-point tiles, point buffer there
// (MQC_HIP_XC_SPLIT=0: off).  Four waves per workgroup up to n = 64, eight above (def2-TZVP water dimer n = 86,
// cc-pVDZ trimers n = 72): the functions of X = D chi are then cut into quarters and the (tile, half) units of the
// potential kernel are dealt to eight waves, so that the density fragments and accumulators still fit the registers.
template <bool GGA>
static bool xc_split_dispatch(const BatchView& bv, int oa, hipStream_t s)
{
    static const bool on = [] { const char* e = std::getenv("MQC_HIP_XC_SPLIT"); return !(e && e[0] == '0'); }();
    if (!on || bv.uhf || bv.n > 96 || bv.topo.lmax > 3 || !bv.grid.rad || bv.grid.rad_pt != XS_PT || !bv.grid.pt4) return false;
    const int nt = (bv.n + 15) / 16;
    const bool f = bv.topo.lmax == 3;
    if (f && nt <= 2) return false;
    if (nt == 1) return xc_split_launch<GGA, 1, 4, false>(bv, oa, s);
    if (nt == 2) return xc_split_launch<GGA, 2, 4, false>(bv, oa, s);
    if (nt == 3) return f ? xc_split_launch<GGA, 3, 4, true>(bv, oa, s) : xc_split_launch<GGA, 3, 4, false>(bv, oa, s);
    if (nt == 4) return f ? xc_split_launch<GGA, 4, 4, true>(bv, oa, s) : xc_split_launch<GGA, 4, 4, false>(bv, oa, s);
    if (nt == 5) return f ? xc_split_launch<GGA, 5, 8, true>(bv, oa, s) : xc_split_launch<GGA, 5, 8, false>(bv, oa, s);
    return f ? xc_split_launch<GGA, 6, 8, true>(bv, oa, s) : xc_split_launch<GGA, 6, 8, false>(bv, oa, s);
}

__global__ void xc_reset_kernel(BatchView bv)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < bv.nfrag) { bv.scal[(size_t)f * 8 + 5] = 0.0; bv.scal[(size_t)f * 8 + 6] = 0.0; }
}

template <bool GGA, int PT, int NV>
__global__ void __launch_bounds__(XC_NT) xc_uks_kernel(BatchView bv, int flags)
{
    extern __shared__ double lds[];
    const int f = blockIdx.y;
    const bool beta = (flags & 2) != 0;
    if ((flags & 1) && bv.istate[4 * f] == ST_DONE) return;
    const int n = bv.n, tid = threadIdx.x;
    const TopologyDev& tp = bv.topo;
    const GridDev& gd = bv.grid;
    constexpr int PTP = PT + 1;
    double* chi = lds;
    double* gx = chi + (size_t)n * PTP;
    double* gy = gx + (GGA ? (size_t)n * PTP : 0);
    double* gz = gy + (GGA ? (size_t)n * PTP : 0);
    double* Xa = gz + (GGA ? (size_t)n * PTP : 0);
    double* Xb = Xa + (size_t)n * PTP;
    double* A = Xb + (size_t)n * PTP;
    double* pw = A + (size_t)n * PTP;                 // [PT] weights
    double* pc = pw + PT;                             // [4][PT] this spin's v_rho w / 2 and gradient coefficient vector

    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const double* __restrict__ Da = bv.D + (size_t)f * n * n;
    const double* __restrict__ Db = bv.Db + (size_t)f * n * n;
    const double* __restrict__ wts = gd.weights + (size_t)f * gd.npts;

    double acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.0;
    double e_acc = 0.0, n_acc = 0.0;

    const int ntiles = (gd.npts + PT - 1) / PT;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int g0 = tile * PT;
        for (int idx = tid; idx < tp.ngroup * PT; idx += XC_NT) {
            const int rg = idx / PT, p = idx - rg * PT;
            const int g = g0 + p;
            if (g < gd.npts) {
                const int oa = gd.pt_atom[g], it = gd.pt_tmpl[g];
                eval_group<GGA>(tp, xyz, rg, xyz[3 * oa] + gd.tmpl_xyz[3 * it], xyz[3 * oa + 1] + gd.tmpl_xyz[3 * it + 1],
                                xyz[3 * oa + 2] + gd.tmpl_xyz[3 * it + 2], chi, gx, gy, gz, PTP, p, bv.c2s);
            } else {
                zero_group<GGA>(tp, rg, chi, gx, gy, gz, PTP, p);
            }
        }
        if (tid < PT) pw[tid] = (g0 + tid < gd.npts) ? wts[g0 + tid] : 0.0;
        __syncthreads();
        for (int idx = tid; idx < n * PT; idx += XC_NT) {
            const int mu = idx / PT, p = idx - mu * PT;
            const double* __restrict__ ra = Da + (size_t)mu * n;
            const double* __restrict__ rb = Db + (size_t)mu * n;
            double sa = 0.0, sb = 0.0;
            for (int nu = 0; nu < n; ++nu) { const double c = chi[nu * PTP + p]; sa += ra[nu] * c; sb += rb[nu] * c; }
            Xa[mu * PTP + p] = sa; Xb[mu * PTP + p] = sb;
        }
        __syncthreads();
        if (tid < PT) {
            const int p = tid;
            double ra = 0.0, rb = 0.0, ga[3] = {0.0, 0.0, 0.0}, gb[3] = {0.0, 0.0, 0.0};
            for (int mu = 0; mu < n; ++mu) {
                const double xa = Xa[mu * PTP + p], xb = Xb[mu * PTP + p], c = chi[mu * PTP + p];
                ra += xa * c; rb += xb * c;
                if (GGA) {
                    const double dx = gx[mu * PTP + p], dy = gy[mu * PTP + p], dz = gz[mu * PTP + p];
                    ga[0] += xa * dx; ga[1] += xa * dy; ga[2] += xa * dz;
                    gb[0] += xb * dx; gb[1] += xb * dy; gb[2] += xb * dz;
                }
            }
            for (int d = 0; d < 3; ++d) { ga[d] *= 2.0; gb[d] *= 2.0; }
            const double saa = ga[0] * ga[0] + ga[1] * ga[1] + ga[2] * ga[2];
            const double sab = ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2];
            const double sbb = gb[0] * gb[0] + gb[1] * gb[1] + gb[2] * gb[2];
            double fx, dv[5];
            eval_functional_pol(bv.xc, ra, rb, saa, sab, sbb, fx, dv);
            const double w = pw[p];
            e_acc += w * fx;
            n_acc += w * (ra + rb);
            pc[p] = 0.5 * w * (beta ? dv[1] : dv[0]);
            if (GGA) {
                const double vss = beta ? dv[4] : dv[2], vab = dv[3];
                for (int d = 0; d < 3; ++d) pc[(1 + d) * PT + p] = w * (2.0 * vss * (beta ? gb[d] : ga[d]) + vab * (beta ? ga[d] : gb[d]));
            }
        }
        __syncthreads();
        for (int idx = tid; idx < n * PT; idx += XC_NT) {
            const int mu = idx / PT, p = idx - mu * PT;
            double a = pc[p] * chi[mu * PTP + p];
            if (GGA) a += pc[PT + p] * gx[mu * PTP + p] + pc[2 * PT + p] * gy[mu * PTP + p] + pc[3 * PT + p] * gz[mu * PTP + p];
            A[mu * PTP + p] = a;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int idx = tid + XC_NT * k;
            if (idx < n * n) {
                const int mu = idx / n, nu = idx - mu * n;
                const double* __restrict__ ar = A + mu * PTP;
                const double* __restrict__ cr = chi + nu * PTP;
                double sum = 0.0;
#pragma unroll 8
                for (int p = 0; p < PT; ++p) sum += ar[p] * cr[p];
                acc[k] += sum;
            }
        }
        __syncthreads();
    }
    double* Vx = bv.Vxc + ((size_t)(beta ? bv.nfrag : 0) + f) * n * n;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int idx = tid + XC_NT * k;
        if (idx < n * n && acc[k] != 0.0) atomicAdd(&Vx[idx], acc[k]);
    }
    if (!beta) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { e_acc += __shfl_down(e_acc, off, 64); n_acc += __shfl_down(n_acc, off, 64); }
        if (tid == 0) {
            atomicAdd(&bv.scal[(size_t)f * 8 + 5], e_acc);
            atomicAdd(&bv.scal[(size_t)f * 8 + 6], n_acc);
        }
    }
}

template <bool GGA, int PT, int NV>
static void xc_uks_launch(const BatchView& bv, int oa, hipStream_t s)
{
    const int n = bv.n;
    const size_t lds = sizeof(double) * ((size_t)(GGA ? 7 : 4) * n * (PT + 1) + 5 * PT + 16);
    auto kern = xc_uks_kernel<GGA, PT, NV>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int ntiles = (bv.grid.npts + PT - 1) / PT;
    int gx = (8192 + bv.nfrag - 1) / bv.nfrag;
    if (gx > ntiles) gx = ntiles;
    if (gx < 1) gx = 1;
    for (int spin = 0; spin < 2; ++spin)
        hipLaunchKernelGGL(kern, dim3(gx, bv.nfrag), dim3(XC_NT), lds, s, bv, oa | (spin << 1));
}

// Meta-GGA quadrature (restricted, or unrestricted with both spin densities per launch), parity-first on the vector units (the reference's meta-GGA lives on its CPU backend:
// eval_rho's tau, mqc_libcint_ao.f90:374-417; accumulate_xc_matrix's third term, mqc_libcint_xc.F90:1436-1448).
// Per tile of PT points: chi and grad chi in LDS, X = D chi -> rho, grad rho; X = D d_d chi for d = x, y, z ->
// tau = 1/2 sum_d X_d . d_d chi; then A = (w v_rho / 2) chi + w c . grad chi and
//     acc += A^T chi + (w v_tau / 4) sum_d (d_d chi)^T d_d chi        (V_xc = acc + acc^T)
template <int PT, int NV, bool UKS>
__global__ void __launch_bounds__(XC_NT) xc_mgga_kernel(BatchView bv, int flags)
{
    extern __shared__ double lds[];
    const int f = blockIdx.y;
    const bool beta = UKS && (flags & 2) != 0;           // unrestricted: one launch per spin potential, both densities in each
    if ((flags & 1) && bv.istate[4 * f] == ST_DONE) return;
    const int n = bv.n, tid = threadIdx.x;
    const TopologyDev& tp = bv.topo;
    const GridDev& gd = bv.grid;
    constexpr int PTP = PT + 1, NS = UKS ? 2 : 1;
    double* chi = lds;
    double* gx = chi + (size_t)n * PTP;
    double* gy = gx + (size_t)n * PTP;
    double* gz = gy + (size_t)n * PTP;
    double* X = gz + (size_t)n * PTP;
    double* A = X + (size_t)n * PTP;
    double* pw = A + (size_t)n * PTP;                 // [PT] weights
    double* pc = pw + PT;                             // [5][PT]: w v_rho / 2, w c_x, w c_y, w c_z, w v_tau / 4 (this spin's)

    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const double* __restrict__ Dsp[2] = {bv.D + (size_t)f * n * n, UKS ? bv.Db + (size_t)f * n * n : nullptr};
    const double* __restrict__ wts = gd.weights + (size_t)f * gd.npts;

    double acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.0;
    double e_acc = 0.0, n_acc = 0.0;

    const int ntiles = (gd.npts + PT - 1) / PT;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int g0 = tile * PT;
        for (int idx = tid; idx < tp.ngroup * PT; idx += XC_NT) {
            const int rg = idx / PT, p = idx - rg * PT;
            const int g = g0 + p;
            if (g < gd.npts) {
                const int oa = gd.pt_atom[g], it = gd.pt_tmpl[g];
                eval_group<true>(tp, xyz, rg, xyz[3 * oa] + gd.tmpl_xyz[3 * it], xyz[3 * oa + 1] + gd.tmpl_xyz[3 * it + 1],
                                 xyz[3 * oa + 2] + gd.tmpl_xyz[3 * it + 2], chi, gx, gy, gz, PTP, p, bv.c2s);
            } else {
                zero_group<true>(tp, rg, chi, gx, gy, gz, PTP, p);
            }
        }
        if (tid < PT) pw[tid] = (g0 + tid < gd.npts) ? wts[g0 + tid] : 0.0;
        double rho[NS], gr[NS][3], tau[NS];
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) { rho[sp] = 0.0; tau[sp] = 0.0; gr[sp][0] = gr[sp][1] = gr[sp][2] = 0.0; }
#pragma unroll
        for (int sp = 0; sp < NS; ++sp)
            for (int pass = 0; pass < 4; ++pass) {
                const double* __restrict__ src = pass == 0 ? chi : pass == 1 ? gx : pass == 2 ? gy : gz;
                const double* __restrict__ D = Dsp[sp];
                __syncthreads();
                for (int idx = tid; idx < n * PT; idx += XC_NT) {
                    const int mu = idx / PT, p = idx - mu * PT;
                    const double* __restrict__ dr = D + (size_t)mu * n;
                    double sum = 0.0;
                    for (int nu = 0; nu < n; ++nu) sum += dr[nu] * src[nu * PTP + p];
                    X[mu * PTP + p] = sum;
                }
                __syncthreads();
                if (tid < PT) {
                    const int p = tid;
                    if (pass == 0) {
                        for (int mu = 0; mu < n; ++mu) {
                            const double x = X[mu * PTP + p];
                            rho[sp] += x * chi[mu * PTP + p];
                            gr[sp][0] += x * gx[mu * PTP + p]; gr[sp][1] += x * gy[mu * PTP + p]; gr[sp][2] += x * gz[mu * PTP + p];
                        }
                    } else {
                        double t = 0.0;
                        for (int mu = 0; mu < n; ++mu) t += X[mu * PTP + p] * src[mu * PTP + p];
                        tau[sp] += 0.5 * t;
                    }
                }
            }
        if (tid < PT) {
            const int p = tid;
            const double w = pw[p];
#pragma unroll
            for (int sp = 0; sp < NS; ++sp)
                for (int d = 0; d < 3; ++d) gr[sp][d] *= 2.0;
            if (UKS) {
                const double* ga = gr[0]; const double* gb = gr[NS - 1];
                const double saa = ga[0] * ga[0] + ga[1] * ga[1] + ga[2] * ga[2];
                const double sab = ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2];
                const double sbb = gb[0] * gb[0] + gb[1] * gb[1] + gb[2] * gb[2];
                double fx, dv[7];
                eval_functional_mgga_pol(bv.xc, rho[0], rho[NS - 1], saa, sab, sbb, tau[0], tau[NS - 1], fx, dv);
                e_acc += w * fx;
                n_acc += w * (rho[0] + rho[NS - 1]);
                pc[p] = 0.5 * w * (beta ? dv[1] : dv[0]);
                const double vss = beta ? dv[4] : dv[2], vab = dv[3];
                for (int d = 0; d < 3; ++d) pc[(1 + d) * PT + p] = w * (2.0 * vss * (beta ? gb[d] : ga[d]) + vab * (beta ? ga[d] : gb[d]));
                pc[4 * PT + p] = 0.25 * w * (beta ? dv[6] : dv[5]);
            } else {
                const double sigma = gr[0][0] * gr[0][0] + gr[0][1] * gr[0][1] + gr[0][2] * gr[0][2];
                double fx, dv[3];
                eval_functional_mgga(bv.xc, rho[0], sigma, tau[0], fx, dv);
                e_acc += w * fx;
                n_acc += w * rho[0];
                pc[p] = 0.5 * w * dv[0];
                for (int d = 0; d < 3; ++d) pc[(1 + d) * PT + p] = w * 2.0 * dv[1] * gr[0][d];
                pc[4 * PT + p] = 0.25 * w * dv[2];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < n * PT; idx += XC_NT) {
            const int mu = idx / PT, p = idx - mu * PT;
            A[mu * PTP + p] = pc[p] * chi[mu * PTP + p] + pc[PT + p] * gx[mu * PTP + p] + pc[2 * PT + p] * gy[mu * PTP + p]
                            + pc[3 * PT + p] * gz[mu * PTP + p];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int idx = tid + XC_NT * k;
            if (idx < n * n) {
                const int mu = idx / n, nu = idx - mu * n;
                const double* __restrict__ ar = A + mu * PTP;
                const double* __restrict__ cr = chi + nu * PTP;
                double sum = 0.0;
#pragma unroll 4
                for (int p = 0; p < PT; ++p)
                    sum += ar[p] * cr[p] + pc[4 * PT + p] * (gx[mu * PTP + p] * gx[nu * PTP + p] + gy[mu * PTP + p] * gy[nu * PTP + p]
                                                             + gz[mu * PTP + p] * gz[nu * PTP + p]);
                acc[k] += sum;
            }
        }
        __syncthreads();
    }
    double* Vx = bv.Vxc + ((size_t)(beta ? bv.nfrag : 0) + f) * n * n;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int idx = tid + XC_NT * k;
        if (idx < n * n && acc[k] != 0.0) atomicAdd(&Vx[idx], acc[k]);
    }
    if (!beta) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { e_acc += __shfl_down(e_acc, off, 64); n_acc += __shfl_down(n_acc, off, 64); }
        if (tid == 0) {
            atomicAdd(&bv.scal[(size_t)f * 8 + 5], e_acc);
            atomicAdd(&bv.scal[(size_t)f * 8 + 6], n_acc);
        }
    }
}

template <int PT, int NV>
static void xc_mgga_launch(const BatchView& bv, int oa, hipStream_t s)
{
    const int n = bv.n;
    const size_t lds = sizeof(double) * ((size_t)6 * n * (PT + 1) + 6 * PT + 16);
    const int ntiles = (bv.grid.npts + PT - 1) / PT;
    int gx = (8192 + bv.nfrag - 1) / bv.nfrag;
    if (gx > ntiles) gx = ntiles;
    if (gx < 1) gx = 1;
    if (bv.uhf) {
        auto kern = xc_mgga_kernel<PT, NV, true>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int spin = 0; spin < 2; ++spin)
            hipLaunchKernelGGL(kern, dim3(gx, bv.nfrag), dim3(XC_NT), lds, s, bv, oa | (spin << 1));
    } else {
        auto kern = xc_mgga_kernel<PT, NV, false>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(gx, bv.nfrag), dim3(XC_NT), lds, s, bv, oa);
    }
}

void launch_xc(const BatchView& bv, bool only_active, hipStream_t s)
{
    const int n = bv.n, oa = only_active ? 1 : 0;
    (void)hipMemsetAsync(bv.Vxc, 0, sizeof(double) * (size_t)bv.nfrag * n * n * (bv.uhf ? 2 : 1), s);
    hipLaunchKernelGGL(xc_reset_kernel, dim3((bv.nfrag + 255) / 256), dim3(256), 0, s, bv);
    const bool gga = bv.xc.gga != 0;
    if (bv.xc.gga == 2) {
        // meta-GGA (restricted or unrestricted, n <= 140: validate_options refuses the rest): 16-point tiles, LDS 6 n 17 doubles <= 114 KB
        const int nv = (n * n + XC_NT - 1) / XC_NT;
        if (nv <= 10) xc_mgga_launch<16, 10>(bv, oa, s);
        else if (nv <= 29) xc_mgga_launch<16, 29>(bv, oa, s);
        else if (nv <= 54) xc_mgga_launch<16, 54>(bv, oa, s);
        else xc_mgga_launch<16, 77>(bv, oa, s);
        return;
    }
    if (bv.uhf) {
        // n <= 116 (the in-core path UHF runs on): 16-point tiles, GGA LDS 7 n 17 doubles <= 110 KB
        const int nv = (n * n + XC_NT - 1) / XC_NT;
        if (nv <= 10) { if (gga) xc_uks_launch<true, 16, 10>(bv, oa, s); else xc_uks_launch<false, 16, 10>(bv, oa, s); }
        else if (nv <= 29) { if (gga) xc_uks_launch<true, 16, 29>(bv, oa, s); else xc_uks_launch<false, 16, 29>(bv, oa, s); }
        else if (nv <= 54) { if (gga) xc_uks_launch<true, 16, 54>(bv, oa, s); else xc_uks_launch<false, 16, 54>(bv, oa, s); }
        else { if (gga) xc_uks_launch<true, 16, 77>(bv, oa, s); else xc_uks_launch<false, 16, 77>(bv, oa, s); }      // n <= 140
        return;
    }
    static const bool probed = [] {
        if (const char* e = std::getenv("MQC_HIP_XC_PROBE")) { const int v = std::atoi(e); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xc_probe), &v, sizeof(int)); }
        return true;
    }();
    (void)probed;
    if (gga ? xc_split_dispatch<true>(bv, oa, s) : xc_split_dispatch<false>(bv, oa, s)) return;
    if (gga ? xc_pipe_dispatch<true>(bv, oa, s) : xc_pipe_dispatch<false>(bv, oa, s)) return;
    if (gga ? xc_tile_dispatch<true>(bv, oa, s) : xc_tile_dispatch<false>(bv, oa, s)) {
#if XC_STAMPS
        (void)hipStreamSynchronize(s);
        unsigned long long h[16];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_xc_stamps), sizeof(h));
        std::fprintf(stderr, "xc stamps n=%d nfrag=%d: slab %llu wait %llu | X %llu wait %llu | functional %llu wait %llu | a %llu wait %llu | A %llu wait %llu | other %llu\n",
                     bv.n, bv.nfrag, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[11]);
#endif
        return;
    }
    // not reached: validate_options refuses fragments above n_ao = 256, the tile kernel covers everything below
    std::fprintf(stderr, "mqc_hip: no quadrature kernel for n = %d\n", n);
}


// ------------------------------------------------------------------ exchange-correlation gradient
// xc_gradient (backends/libcint/mqc_libcint_gradient.f90:331-549), per point g with quadrature weight q, partition
// P and w = q P, channel density D (total when restricted, one spin otherwise), Xc = D chi, Xg_j = D d_j chi:
//     S_k(A) = sum_{mu in A} [ w v_rho d_k chi_mu Xc_mu + sum_j w c_j (d_k d_j chi_mu Xc_mu + d_k chi_mu Xg_j,mu) ]
//     grad[A] -= 2 S(A)                 the functions of A move with A
//     grad[own(g)] += 2 sum_A S(A)      the point moves with the atom that owns it
//     grad[A] += q f dP/dR_A,  grad[own] -= q f sum_A dP/dR_A      the partition weights move (becke_xc_grad_kernel)
// with c = dE/d(grad rho) = 2 v_sigma grad rho (restricted) or 2 v_ss grad rho_s + v_ab grad rho_s' (unrestricted).
// Parity-first kernels on the vector units: a gradient is one evaluation per geometry step, not per SCF iteration.
constexpr int XG_PT = 8, XG_NT = 256;

// value, gradient and packed Hessian (xx, xy, xz, yy, yz, zz) of P(x, y, z) R(r^2) from the polynomial's own
// derivatives and R0 = R, R1 = R'/r, R2 = (R'/r)'/r:  d_i d_j (P R) = P_ij R0 + (P_i x_j + P_j x_i) R1 + P (R2 x_i x_j + R1 d_ij)
template <bool GGA>
__device__ __forceinline__ void put_function(double P, const double* Pi, const double* Pij, const double* x, double R0, double R1, double R2,
                                             double* __restrict__ base, size_t S, int o)
{
    base[o] = P * R0;
    for (int i = 0; i < 3; ++i) base[(1 + i) * S + o] = Pi[i] * R0 + P * R1 * x[i];
    if (GGA) {
        int m = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = i; j < 3; ++j, ++m)
                base[(4 + m) * S + o] = Pij[m] * R0 + (Pi[i] * x[j] + Pi[j] * x[i]) * R1 + P * (R2 * x[i] * x[j] + (i == j ? R1 : 0.0));
    }
}

template <bool GGA>
__device__ void emit_shell_d2(int l, int ao, double dx, double dy, double dz, double R0, double R1, double R2,
                              double* __restrict__ base, size_t S, int ptp, int p, const double* __restrict__ c2s)
{
    const double x[3] = {dx, dy, dz};
    if (l == 0) {
        const double Pi[3] = {0.0, 0.0, 0.0}, Pij[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        put_function<GGA>(1.0, Pi, Pij, x, R0, R1, R2, base, S, ao * ptp + p);
    } else if (l == 1) {
        for (int k = 0; k < 3; ++k) {
            double Pi[3] = {0.0, 0.0, 0.0};
            const double Pij[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            Pi[k] = 1.0;
            put_function<GGA>(x[k], Pi, Pij, x, R0, R1, R2, base, S, (ao + k) * ptp + p);
        }
    } else if (l == 2) {
        // l = 2: the five real solid harmonics as combinations of xx, xy, xz, yy, yz, zz (libcint order xy, yz, z2, xz, x2-y2)
        const int ca[6] = {0, 0, 0, 1, 1, 2}, cb[6] = {0, 1, 2, 1, 2, 2};
        for (int m = 0; m < 5; ++m) {
            double P = 0.0, Pi[3] = {0.0, 0.0, 0.0}, Pij[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            for (int c = 0; c < 6; ++c) {
                const double w = c2s_coef<2>(nullptr, m, c);
                if (w == 0.0) continue;
                const int a = ca[c], b = cb[c];
                P += w * x[a] * x[b];
                Pi[a] += w * x[b]; Pi[b] += w * x[a];
                // packed index of (a, b), a <= b: xx 0, xy 1, xz 2, yy 3, yz 4, zz 5 -- the same order as the Cartesian list
                Pij[c] += (a == b) ? 2.0 * w : w;
            }
            put_function<GGA>(P, Pi, Pij, x, R0, R1, R2, base, S, (ao + m) * ptp + p);
        }
    } else {
        // l >= 3: the real solid harmonics from the cart -> sph table over the monomials x^i y^j z^k and their derivatives
        double pw[3][LMAX_AO + 1];
        for (int d = 0; d < 3; ++d) { pw[d][0] = 1.0; for (int k = 1; k <= l; ++k) pw[d][k] = pw[d][k - 1] * x[d]; }
        const int nc = ncart(l);
        for (int m = 0; m < 2 * l + 1; ++m) {
            double P = 0.0, Pi[3] = {0.0, 0.0, 0.0}, Pij[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            for (int c = 0; c < nc; ++c) {
                const double w = c2s[c2s_table_offset(l) + m * nc + c];
                if (w == 0.0) continue;
                int e[3];
                cart_lmn(l, c, e[0], e[1], e[2]);
                // value and first derivative factor of every axis: f_d = x_d^e_d, f_d' = e_d x_d^(e_d - 1), f_d'' = e_d (e_d - 1) x_d^(e_d - 2)
                double f0[3], f1[3], f2[3];
                for (int d = 0; d < 3; ++d) {
                    f0[d] = pw[d][e[d]];
                    f1[d] = e[d] > 0 ? e[d] * pw[d][e[d] - 1] : 0.0;
                    f2[d] = e[d] > 1 ? e[d] * (e[d] - 1) * pw[d][e[d] - 2] : 0.0;
                }
                P += w * f0[0] * f0[1] * f0[2];
                Pi[0] += w * f1[0] * f0[1] * f0[2]; Pi[1] += w * f0[0] * f1[1] * f0[2]; Pi[2] += w * f0[0] * f0[1] * f1[2];
                Pij[0] += w * f2[0] * f0[1] * f0[2]; Pij[1] += w * f1[0] * f1[1] * f0[2]; Pij[2] += w * f1[0] * f0[1] * f1[2];
                Pij[3] += w * f0[0] * f2[1] * f0[2]; Pij[4] += w * f0[0] * f1[1] * f1[2]; Pij[5] += w * f0[0] * f0[1] * f2[2];
            }
            put_function<GGA>(P, Pi, Pij, x, R0, R1, R2, base, S, (ao + m) * ptp + p);
        }
    }
}

template <bool GGA, bool UKS>
__global__ void __launch_bounds__(XG_NT) xc_grad_kernel(BatchView bv, double* __restrict__ d_grad, double* __restrict__ fbuf)
{
    extern __shared__ double lds[];
    const int f = blockIdx.y, tid = threadIdx.x, n = bv.n;
    const TopologyDev& tp = bv.topo;
    const GridDev& gd = bv.grid;
    constexpr int PT = XG_PT, PTP = PT + 1, NA = GGA ? 10 : 4, NXS = GGA ? 4 : 1, NSP = UKS ? 2 : 1;
    const size_t S = (size_t)n * PTP;
    double* ao = lds;                                   // [NA][n][PTP]: chi, d chi (3), d d chi (6)
    double* X = ao + NA * S;                            // [NSP][NXS][n][PTP]: D chi, D d_j chi
    double* pwv = X + (size_t)NSP * NXS * S;            // [NSP][4][PT]: w v_rho, w c_j
    double* gacc = pwv + NSP * 4 * PT;                  // [natoms][3]
    int* aoff = (int*)(gacc + 3 * tp.natoms);           // [natoms + 1] first function of every atom
    int* own = aoff + tp.natoms + 1;                    // [PT]
    const double* xyz = bv.xyz + (size_t)f * tp.natoms * 3;
    const double* Dm[2] = {bv.D + (size_t)f * n * n, UKS ? bv.Db + (size_t)f * n * n : nullptr};
    const double* wts = gd.weights + (size_t)f * gd.npts;

    for (int i = tid; i < 3 * tp.natoms; i += XG_NT) gacc[i] = 0.0;
    if (tid == 0) {
        // shells are ordered by atom: the first function of every atom
        for (int a = 0; a <= tp.natoms; ++a) aoff[a] = n;
        for (int sh = tp.nshell - 1; sh >= 0; --sh) aoff[tp.sh_atom[sh]] = tp.sh_aoff[sh];
        for (int a = tp.natoms - 1; a >= 0; --a) if (aoff[a] > aoff[a + 1]) aoff[a] = aoff[a + 1];       // atoms without functions
    }
    __syncthreads();
    const int ntiles = (gd.npts + PT - 1) / PT;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int g0 = tile * PT;
        // 1. functions, first and second derivatives: (shell, point) items
        for (int idx = tid; idx < tp.nshell * PT; idx += XG_NT) {
            const int sh = idx / PT, p = idx - sh * PT, g = g0 + p;
            const int l = tp.sh_l[sh], at = tp.sh_atom[sh];
            double dx = 0.0, dy = 0.0, dz = 0.0, R0 = 0.0, R1 = 0.0, R2 = 0.0;
            if (g < gd.npts) {
                const int oa = gd.pt_atom[g], it = gd.pt_tmpl[g];
                dx = xyz[3 * oa] + gd.tmpl_xyz[3 * it] - xyz[3 * at];
                dy = xyz[3 * oa + 1] + gd.tmpl_xyz[3 * it + 1] - xyz[3 * at + 1];
                dz = xyz[3 * oa + 2] + gd.tmpl_xyz[3 * it + 2] - xyz[3 * at + 2];
                const double r2 = dx * dx + dy * dy + dz * dz;
                const double* e = tp.exps + tp.sh_poff[sh];
                const double* c = tp.coefs + tp.sh_poff[sh];
                for (int k = 0; k < tp.sh_nprim[sh]; ++k) {
                    const double a = e[k], ar2 = a * r2;
                    if (ar2 < XC_EXP_CUTOFF) {
                        const double t = c[k] * exp(-ar2);
                        R0 += t; R1 -= 2.0 * a * t; R2 += 4.0 * a * a * t;
                    }
                }
            }
            emit_shell_d2<GGA>(l, tp.sh_aoff[sh], dx, dy, dz, R0, R1, R2, ao, S, PTP, p, bv.c2s);
        }
        if (tid < PT) own[tid] = (g0 + tid < gd.npts) ? gd.pt_atom[g0 + tid] : 0;
        __syncthreads();
        // 2. X = D chi and, for a gradient-corrected functional, D d_j chi, per spin
        for (int idx = tid; idx < n * PT; idx += XG_NT) {
            const int mu = idx / PT, p = idx - mu * PT;
            for (int sp = 0; sp < NSP; ++sp) {
                const double* __restrict__ dr = Dm[sp] + (size_t)mu * n;
                double acc[NXS];
                for (int k = 0; k < NXS; ++k) acc[k] = 0.0;
                for (int nu = 0; nu < n; ++nu) {
                    const double d = dr[nu];
                    for (int k = 0; k < NXS; ++k) acc[k] += d * ao[k * S + nu * PTP + p];
                }
                for (int k = 0; k < NXS; ++k) X[((size_t)sp * NXS + k) * S + mu * PTP + p] = acc[k];
            }
        }
        __syncthreads();
        // 3. densities, the functional, the per-point weights; the energy density goes to fbuf for the partition term
        if (tid < PT) {
            const int p = tid, g = g0 + p;
            double rho[2] = {0.0, 0.0}, gr[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
            for (int sp = 0; sp < NSP; ++sp)
                for (int mu = 0; mu < n; ++mu) {
                    const double xc_ = X[(size_t)sp * NXS * S + mu * PTP + p];
                    rho[sp] += xc_ * ao[mu * PTP + p];
                    if (GGA) for (int j = 0; j < 3; ++j) gr[sp][j] += 2.0 * xc_ * ao[(1 + j) * S + mu * PTP + p];
                }
            const double w = (g < gd.npts) ? wts[g] : 0.0;
            double fx = 0.0;
            if (!UKS) {
                const double sigma = GGA ? gr[0][0] * gr[0][0] + gr[0][1] * gr[0][1] + gr[0][2] * gr[0][2] : 0.0;
                double vr, vs;
                eval_functional(bv.xc, rho[0], sigma, fx, vr, vs);
                pwv[p] = w * vr;
                for (int j = 0; j < 3; ++j) pwv[(1 + j) * PT + p] = GGA ? w * 2.0 * vs * gr[0][j] : 0.0;
            } else {
                double saa = 0.0, sab = 0.0, sbb = 0.0, dv[5];
                for (int j = 0; j < 3; ++j) { saa += gr[0][j] * gr[0][j]; sab += gr[0][j] * gr[1][j]; sbb += gr[1][j] * gr[1][j]; }
                eval_functional_pol(bv.xc, rho[0], rho[1], saa, sab, sbb, fx, dv);
                pwv[p] = w * dv[0]; pwv[4 * PT + p] = w * dv[1];
                for (int j = 0; j < 3; ++j) {
                    pwv[(1 + j) * PT + p] = GGA ? w * (2.0 * dv[2] * gr[0][j] + dv[3] * gr[1][j]) : 0.0;
                    pwv[(5 + j) * PT + p] = GGA ? w * (2.0 * dv[4] * gr[1][j] + dv[3] * gr[0][j]) : 0.0;
                }
            }
            if (g < gd.npts) fbuf[(size_t)f * gd.npts + g] = fx;
        }
        __syncthreads();
        // 4. block sums per (point, atom): the functions of the atom move with it, the point with its owner
        for (int idx = tid; idx < PT * tp.natoms; idx += XG_NT) {
            const int A = idx / PT, p = idx - A * PT;
            double Sk[3] = {0.0, 0.0, 0.0};
            for (int sp = 0; sp < NSP; ++sp) {
                const double wv = pwv[(sp * 4) * PT + p];
                const double c0 = pwv[(sp * 4 + 1) * PT + p], c1 = pwv[(sp * 4 + 2) * PT + p], c2 = pwv[(sp * 4 + 3) * PT + p];
                const double* Xs = X + (size_t)sp * NXS * S;
                for (int mu = aoff[A]; mu < aoff[A + 1]; ++mu) {
                    const int o = mu * PTP + p;
                    const double xc_ = Xs[o];
                    const double g0_ = ao[S + o], g1_ = ao[2 * S + o], g2_ = ao[3 * S + o];
                    Sk[0] += wv * g0_ * xc_; Sk[1] += wv * g1_ * xc_; Sk[2] += wv * g2_ * xc_;
                    if (GGA) {
                        const double hxx = ao[4 * S + o], hxy = ao[5 * S + o], hxz = ao[6 * S + o], hyy = ao[7 * S + o], hyz = ao[8 * S + o], hzz = ao[9 * S + o];
                        const double xg = c0 * Xs[S + o] + c1 * Xs[2 * S + o] + c2 * Xs[3 * S + o];        // sum_j c_j (D d_j chi)_mu
                        Sk[0] += (c0 * hxx + c1 * hxy + c2 * hxz) * xc_ + g0_ * xg;
                        Sk[1] += (c0 * hxy + c1 * hyy + c2 * hyz) * xc_ + g1_ * xg;
                        Sk[2] += (c0 * hxz + c1 * hyz + c2 * hzz) * xc_ + g2_ * xg;
                    }
                }
            }
            const int o = own[p];
            for (int k = 0; k < 3; ++k) {
                if (Sk[k] != 0.0) { atomicAdd(&gacc[3 * A + k], -2.0 * Sk[k]); atomicAdd(&gacc[3 * o + k], 2.0 * Sk[k]); }
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < 3 * tp.natoms; i += XG_NT)
        if (gacc[i] != 0.0) atomicAdd(&d_grad[(size_t)f * tp.natoms * 3 + i], gacc[i]);
}

// d(f3(f2(f1(nu))))/d nu of Becke's cell function s = (1 - f3) / 2, f(x) = x (3 - x^2) / 2
__device__ __forceinline__ void becke_cutoff_d(double nu, double& s, double& ds)
{
    const double f1 = 0.5 * nu * (3.0 - nu * nu), f2 = 0.5 * f1 * (3.0 - f1 * f1), f3 = 0.5 * f2 * (3.0 - f2 * f2);
    s = 0.5 * (1.0 - f3);
    ds = -0.5 * (1.5 * (1.0 - f2 * f2)) * (1.5 * (1.0 - f1 * f1)) * (1.5 * (1.0 - nu * nu));
}

// The partition term: P_O = cell_O / sum_C cell_C with cell_C = prod_{B != C} s(nu_CB); point held fixed,
//     dP_O/dR_A = P_O sum_C (delta_CO - P_C) d ln cell_C / dR_A,
// and the pair (i < j) touches d ln cell_i and d ln cell_j with respect to R_i and R_j only.
__global__ void __launch_bounds__(256) becke_xc_grad_kernel(BatchView bv, const double* __restrict__ fbuf, double* __restrict__ d_grad)
{
    __shared__ double gblk[3 * BECKE_MAX_ATOMS];
    const int f = blockIdx.y;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const GridDev& gd = bv.grid;
    const int na = bv.topo.natoms;
    for (int i = threadIdx.x; i < 3 * na; i += blockDim.x) gblk[i] = 0.0;
    __syncthreads();
    if (g < gd.npts && na > 1) {
        const double* xyz = bv.xyz + (size_t)f * na * 3;
        const int owner = gd.pt_atom[g], it = gd.pt_tmpl[g];
        const double px = xyz[3 * owner] + gd.tmpl_xyz[3 * it], py = xyz[3 * owner + 1] + gd.tmpl_xyz[3 * it + 1],
                     pz = xyz[3 * owner + 2] + gd.tmpl_xyz[3 * it + 2];
        const double qf = gd.tmpl_w[it] * fbuf[(size_t)f * gd.npts + g];
        if (qf != 0.0) {
            double dist[BECKE_MAX_ATOMS], cell[BECKE_MAX_ATOMS], G[3 * BECKE_MAX_ATOMS];
            for (int i = 0; i < na; ++i) {
                const double dx = px - xyz[3 * i], dy = py - xyz[3 * i + 1], dz = pz - xyz[3 * i + 2];
                dist[i] = sqrt(dx * dx + dy * dy + dz * dz);
                cell[i] = 1.0;
                G[3 * i] = 0.0; G[3 * i + 1] = 0.0; G[3 * i + 2] = 0.0;
            }
            for (int i = 0; i < na; ++i)
                for (int j = i + 1; j < na; ++j) {
                    const double dx = xyz[3 * i] - xyz[3 * j], dy = xyz[3 * i + 1] - xyz[3 * j + 1], dz = xyz[3 * i + 2] - xyz[3 * j + 2];
                    const double mu = (dist[i] - dist[j]) / sqrt(dx * dx + dy * dy + dz * dz);
                    const double chi = gd.sqrt_bragg[i] / gd.sqrt_bragg[j];
                    double a = 0.25 * (1.0 / chi - chi);
                    a = fmax(-0.5, fmin(0.5, a));
                    const double s = becke_cutoff(mu + a * (1.0 - mu * mu));
                    cell[i] *= s;
                    cell[j] *= (1.0 - s);
                }
            double tot = 0.0;
            for (int i = 0; i < na; ++i) tot += cell[i];
            if (tot > 0.0 && cell[owner] > 0.0) {
                const double itot = 1.0 / tot, PO = cell[owner] * itot;
                for (int i = 0; i < na; ++i)
                    for (int j = i + 1; j < na; ++j) {
                        const double rx = xyz[3 * i] - xyz[3 * j], ry = xyz[3 * i + 1] - xyz[3 * j + 1], rz = xyz[3 * i + 2] - xyz[3 * j + 2];
                        const double Rij = sqrt(rx * rx + ry * ry + rz * rz), iR = 1.0 / Rij;
                        const double mu = (dist[i] - dist[j]) * iR;
                        const double chi = gd.sqrt_bragg[i] / gd.sqrt_bragg[j];
                        double a = 0.25 * (1.0 / chi - chi);
                        a = fmax(-0.5, fmin(0.5, a));
                        double s, ds;
                        becke_cutoff_d(mu + a * (1.0 - mu * mu), s, ds);
                        const double c = ds * (1.0 - 2.0 * a * mu);                 // d s / d mu
                        // K = (delta_iO - P_i) c / s - (delta_jO - P_j) c / (1 - s); a vanished factor means a vanished cell
                        double K = 0.0;
                        if (s > 1.0e-300) K += ((i == owner ? 1.0 : 0.0) - cell[i] * itot) * c / s;
                        if (1.0 - s > 1.0e-300) K -= ((j == owner ? 1.0 : 0.0) - cell[j] * itot) * c / (1.0 - s);
                        if (K == 0.0) continue;
                        // d mu / dR_i = -u_i / R_ij - mu (R_i - R_j) / R_ij^2,  d mu / dR_j = u_j / R_ij + mu (R_i - R_j) / R_ij^2
                        const double ui[3] = {(px - xyz[3 * i]) / fmax(dist[i], 1.0e-300), (py - xyz[3 * i + 1]) / fmax(dist[i], 1.0e-300), (pz - xyz[3 * i + 2]) / fmax(dist[i], 1.0e-300)};
                        const double uj[3] = {(px - xyz[3 * j]) / fmax(dist[j], 1.0e-300), (py - xyz[3 * j + 1]) / fmax(dist[j], 1.0e-300), (pz - xyz[3 * j + 2]) / fmax(dist[j], 1.0e-300)};
                        const double rv[3] = {rx, ry, rz};
                        for (int k = 0; k < 3; ++k) {
                            const double t = mu * rv[k] * iR * iR;
                            G[3 * i + k] += K * (-ui[k] * iR - t);
                            G[3 * j + k] += K * (uj[k] * iR + t);
                        }
                    }
                double sum[3] = {0.0, 0.0, 0.0};
                for (int i = 0; i < na; ++i)
                    for (int k = 0; k < 3; ++k) {
                        const double v = qf * PO * G[3 * i + k];
                        if (v != 0.0) atomicAdd(&gblk[3 * i + k], v);
                        sum[k] += v;
                    }
                for (int k = 0; k < 3; ++k) if (sum[k] != 0.0) atomicAdd(&gblk[3 * owner + k], -sum[k]);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * na; i += blockDim.x)
        if (gblk[i] != 0.0) atomicAdd(&d_grad[(size_t)f * na * 3 + i], gblk[i]);
}

static DevicePool g_xcgrad_pool[2];

bool launch_xc_gradient(const BatchView& bv, double* d_grad, hipStream_t s, std::string& err)
{
    const int n = bv.n;
    const bool gga = bv.xc.gga != 0, uks = bv.uhf != 0;
    double* fbuf = (double*)g_xcgrad_pool[bv.slot & 1].ensure(sizeof(double) * (size_t)bv.nfrag * bv.grid.npts + 256);
    if (!fbuf) { err = "out of device memory (XC gradient)"; return false; }
    const int NA = gga ? 10 : 4, NX = (gga ? 4 : 1) * (uks ? 2 : 1);
    const size_t lds = sizeof(double) * ((size_t)(NA + NX) * n * (XG_PT + 1) + 8 * XG_PT + 3 * bv.topo.natoms + 8) + sizeof(int) * (bv.topo.natoms + 2 + XG_PT);
    if (lds > 160 * 1024) { err = "XC gradient: fragment too large for the LDS slab"; return false; }
    const int ntiles = (bv.grid.npts + XG_PT - 1) / XG_PT;
    int gx = (8192 + bv.nfrag - 1) / bv.nfrag;
    if (gx > ntiles) gx = ntiles;
    if (gx < 1) gx = 1;
#define XG_LAUNCH(G, U)                                                                                     \
    do {                                                                                                    \
        (void)hipFuncSetAttribute((const void*)xc_grad_kernel<G, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((xc_grad_kernel<G, U>), dim3(gx, bv.nfrag), dim3(XG_NT), lds, s, bv, d_grad, fbuf); \
    } while (0)
    if (gga) { if (uks) XG_LAUNCH(true, true); else XG_LAUNCH(true, false); }
    else { if (uks) XG_LAUNCH(false, true); else XG_LAUNCH(false, false); }
#undef XG_LAUNCH
    hipLaunchKernelGGL(becke_xc_grad_kernel, dim3((bv.grid.npts + 255) / 256, bv.nfrag), dim3(256), 0, s, bv, fbuf, d_grad);
    return true;
}

}  // namespace mqc
