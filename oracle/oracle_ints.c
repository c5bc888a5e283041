/*
 * oracle_ints.c -- CPU restatement of the integral layer behind metalquicha's
 * libcint SCF path.  TEST INFRASTRUCTURE ONLY: nothing in the shipped product
 * (metalquicha_amd/, the C-ABI library) may link, import or call this file.
 * It is the checker for the HIP kernels (tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg).
 *
 * What it restates (reference file:line, /root/reference):
 *   - one-electron S, T, V:  backends/libcint/mqc_libcint_integrals.F90:843-911
 *     (one_electron -> libcint_1e_{ovlp,kin,nuc}_sph)
 *   - four-centre ERIs:      backends/libcint/mqc_libcint_integrals.F90:1449
 *     (molecule_eris -> libcint_2e_sph)
 *   - three/two-centre ERIs: backends/libcint/mqc_libcint_integrals.F90:1144,1331-1447
 *   - Schwarz bounds:        backends/libcint/mqc_libcint_direct.f90:105-153
 *   - AO values on a grid:   backends/libcint/mqc_libcint_ao.f90:69 (eval_ao_block)
 *
 * The integral arithmetic itself lives in a third-party dependency that is
 * absent from /root/reference: libfint v0.1.1 / libcint fork @3c78069
 * (CMakeLists.txt:457-460,505-524).  Its published conventions are restated
 * here: real solid harmonics r^l Y_lm with libcint's AO order (p = x,y,z;
 * d = xy,yz,z2,xz,x2-y2; ...), radial normalisation gto_norm, Cartesian
 * component order x^(l-i) y^(i-j) z^j.  The algorithm is McMurchie-Davidson
 * (Hermite Gaussians + Boys function) -- values are what matter, and they are
 * pinned by the reference's known-answer energies (tests/test_oracle_golden.py).
 *
 * Inputs are flat arrays (no structs cross the ctypes boundary):
 *   nshell, sh_l[], sh_nprim[], sh_poff[] (offset into exps/coefs),
 *   sh_xyz[3*nshell], exps[], coefs[] (already normalised radial coefficients),
 *   sh_aoff[] (spherical AO offset of each shell), nao.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LMAX 4              /* g functions; matches the reference's c2s table (l <= 4) */
#define NCART(l) (((l) + 1) * ((l) + 2) / 2)
/* Cartesian mode (orc_set_cartesian): shells above p keep their NCART Cartesian components, libcint's
 * x^(l-i) y^(i-j) z^j order with the radial normalisation only -- what the reference's CPU path runs for a
 * basis whose d shells are marked gto_cartesian (6-31G*, mqc_libcint_integrals.F90:132-148).  Converged SCF
 * energies do not depend on how the individual components are normalised. */
static int g_cartesian = 0;
void orc_set_cartesian(int flag) { g_cartesian = flag; }
#define NSPH(l) ((g_cartesian && (l) >= 2) ? NCART(l) : 2 * (l) + 1)
#define MAXCART NCART(LMAX)
#define LSUM_MAX (4 * LMAX)
#define NHERM_MAX(L) (((L) + 1) * ((L) + 2) * ((L) + 3) / 6)

typedef struct {
    int nshell;
    const int *l, *nprim, *poff, *aoff;
    const double *xyz, *exps, *coefs;
    int nao;
} basis_t;

/* ---------------------------------------------------------------- Boys */
static void boys(int nmax, double T, double *F)
{
    if (T < 1e-13) {
        for (int n = 0; n <= nmax; ++n) F[n] = 1.0 / (2 * n + 1) - T / (2 * n + 3);
        return;
    }
    if (T < 36.0 + 1.5 * nmax) {
        /* series for the top order, then downward recursion (stable) */
        double et = exp(-T);
        double term = 1.0 / (2 * nmax + 1), sum = term;
        for (int k = 1; k < 400; ++k) {
            term *= 2.0 * T / (2 * nmax + 2 * k + 1);
            sum += term;
            if (term < 1e-17 * sum) break;
        }
        F[nmax] = et * sum;
        for (int n = nmax; n > 0; --n) F[n - 1] = (2.0 * T * F[n] + et) / (2 * n - 1);
    } else {
        double et = exp(-T);
        F[0] = 0.5 * sqrt(M_PI / T) * erf(sqrt(T));
        for (int n = 0; n < nmax; ++n) F[n + 1] = ((2 * n + 1) * F[n] - et) / (2.0 * T);
    }
}

/* ------------------------------------------------- Hermite expansion E */
/* E[i][j][t], 0<=i<=la, 0<=j<=lb, 0<=t<=i+j; 1-D, includes exp(-mu X_AB^2). */
#define EDIM (LMAX + 3)   /* +2 so the kinetic operator can raise j by 2 */
typedef double etab_t[EDIM][EDIM][2 * EDIM];

static void hermite_e(int la, int lb, double a, double b, double xa, double xb, etab_t E)
{
    double p = a + b, mu = a * b / p, xab = xa - xb;
    double xp = (a * xa + b * xb) / p, xpa = xp - xa, xpb = xp - xb;
    double hp = 0.5 / p;
    memset(E, 0, sizeof(etab_t));
    E[0][0][0] = exp(-mu * xab * xab);
    for (int i = 0; i <= la; ++i) {
        if (i > 0) {
            for (int t = 0; t <= i; ++t) {
                double v = xpa * E[i - 1][0][t];
                if (t > 0) v += hp * E[i - 1][0][t - 1];
                if (t + 1 <= i - 1) v += (t + 1) * E[i - 1][0][t + 1];
                E[i][0][t] = v;
            }
        }
        for (int j = 1; j <= lb; ++j) {
            for (int t = 0; t <= i + j; ++t) {
                double v = xpb * E[i][j - 1][t];
                if (t > 0) v += hp * E[i][j - 1][t - 1];
                if (t + 1 <= i + j - 1) v += (t + 1) * E[i][j - 1][t + 1];
                E[i][j][t] = v;
            }
        }
    }
}

/* ------------------------------------------------ Hermite Coulomb R_tuv */
/* R[t][u][v] for t+u+v <= L, from Boys values; alpha = reduced exponent. */
#define RDIM (LSUM_MAX + 1)
static void hermite_r(int L, double alpha, double X, double Y, double Z, double *Rout /* RDIM^3 */)
{
    /* work[n][t][u][v] with n descending; small L so do the naive 4-index thing */
    static __thread double *work = NULL;
    if (!work) work = (double *)malloc(sizeof(double) * (RDIM + 1) * RDIM * RDIM * RDIM);
#define W(n, t, u, v) work[(((n) * RDIM + (t)) * RDIM + (u)) * RDIM + (v)]
    double F[LSUM_MAX + 2];
    boys(L, alpha * (X * X + Y * Y + Z * Z), F);
    double m2a = 1.0;
    for (int n = 0; n <= L; ++n) { W(n, 0, 0, 0) = m2a * F[n]; m2a *= -2.0 * alpha; }
    for (int N = 1; N <= L; ++N) {           /* total order t+u+v = N */
        for (int n = 0; n <= L - N; ++n) {
            for (int t = 0; t <= N; ++t)
                for (int u = 0; u <= N - t; ++u) {
                    int v = N - t - u;
                    double val;
                    if (t > 0) {
                        val = X * W(n + 1, t - 1, u, v);
                        if (t > 1) val += (t - 1) * W(n + 1, t - 2, u, v);
                    } else if (u > 0) {
                        val = Y * W(n + 1, t, u - 1, v);
                        if (u > 1) val += (u - 1) * W(n + 1, t, u - 2, v);
                    } else {
                        val = Z * W(n + 1, t, u, v - 1);
                        if (v > 1) val += (v - 1) * W(n + 1, t, u, v - 2);
                    }
                    W(n, t, u, v) = val;
                }
        }
    }
    for (int t = 0; t <= L; ++t)
        for (int u = 0; u <= L - t; ++u)
            for (int v = 0; v <= L - t - u; ++v)
                Rout[(t * RDIM + u) * RDIM + v] = W(0, t, u, v);
#undef W
}

/* ---------------------------------------------- Cartesian components */
static void cart_components(int l, int (*lmn)[3])
{
    int k = 0;
    for (int lx = l; lx >= 0; --lx)
        for (int ly = l - lx; ly >= 0; --ly) {
            lmn[k][0] = lx; lmn[k][1] = ly; lmn[k][2] = l - lx - ly; ++k;
        }
}

/* ------------------------------------- real solid harmonics r^l Y_lm */
static double binom(int n, int k)
{
    if (k < 0 || k > n) return 0.0;
    double r = 1.0;
    for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return r;
}
static double fact(int n) { double r = 1.0; for (int i = 2; i <= n; ++i) r *= i; return r; }

/* c2s[l] is NSPH x NCART, libcint order: m = -l..l, except l=1 which is x,y,z. */
static double C2S[LMAX + 1][2 * LMAX + 1][MAXCART];
static int c2s_ready = 0;

static void build_c2s(void)
{
    if (c2s_ready) return;
    memset(C2S, 0, sizeof(C2S));
    for (int l = 0; l <= LMAX; ++l) {
        int lmn[MAXCART][3];
        cart_components(l, lmn);
        int nc = NCART(l);
        double racah_to_y = sqrt((2.0 * l + 1.0) / (4.0 * M_PI));
        for (int m = -l; m <= l; ++m) {
            int am = abs(m);
            double nlm = 1.0 / (pow(2.0, am) * fact(l)) *
                         sqrt(2.0 * fact(l + am) * fact(l - am) / (m == 0 ? 2.0 : 1.0));
            int row = m + l;
            /* v runs over integers (m>=0) or half-integers (m<0): use 2v */
            int twov0 = (m >= 0) ? 0 : 1;
            for (int t = 0; t <= (l - am) / 2; ++t)
                for (int u = 0; u <= t; ++u)
                    for (int twov = twov0; twov <= am; twov += 2) {
                        /* (-1)^(t + v - v_m), v - v_m integer */
                        int sgn_pow = t + (twov - twov0) / 2;
                        double c = ((sgn_pow & 1) ? -1.0 : 1.0) * pow(0.25, t) * binom(l, t) *
                                   binom(l - t, am + t) * binom(t, u) * binom(am, twov);
                        int ex = 2 * t + am - 2 * u - twov;
                        int ey = 2 * u + twov;
                        int ez = l - 2 * t - am;
                        if (ex < 0 || ey < 0 || ez < 0) continue;
                        for (int k = 0; k < nc; ++k)
                            if (lmn[k][0] == ex && lmn[k][1] == ey && lmn[k][2] == ez)
                                C2S[l][row][k] += nlm * c * racah_to_y;
                    }
        }
        if (l == 1) {
            /* libcint keeps p functions in x,y,z order (m = +1,-1,0) */
            double tmp[3][MAXCART];
            memcpy(tmp, C2S[1], sizeof(tmp));
            memcpy(C2S[1][0], tmp[2], sizeof(tmp[0]));  /* m=+1 -> x */
            memcpy(C2S[1][1], tmp[0], sizeof(tmp[0]));  /* m=-1 -> y */
            memcpy(C2S[1][2], tmp[1], sizeof(tmp[0]));  /* m= 0 -> z */
        }
    }
    c2s_ready = 1;
}

void orc_c2s(int l, double *out /* NSPH(l) x NCART(l) row-major */)
{
    build_c2s();
    for (int r = 0; r < NSPH(l); ++r)
        for (int c = 0; c < NCART(l); ++c) out[r * NCART(l) + c] = C2S[l][r][c];
}

/* transform one index of a tensor: in[pre][nc][post] -> out[pre][ns][post]; l<0: identity 1x1 */
static void c2s_index(int l, int pre, int post, const double *in, double *out)
{
    if (l < 0) { memcpy(out, in, sizeof(double) * pre * post); return; }
    int nc = NCART(l), ns = NSPH(l);
    if (g_cartesian && l >= 2) { memcpy(out, in, sizeof(double) * pre * nc * post); return; }
    for (int a = 0; a < pre; ++a)
        for (int s = 0; s < ns; ++s)
            for (int b = 0; b < post; ++b) {
                double v = 0.0;
                for (int c = 0; c < nc; ++c) v += C2S[l][s][c] * in[(a * nc + c) * post + b];
                out[(a * ns + s) * post + b] = v;
            }
}

/* -------------------------------------------------- one-electron ints */
/* S, T, V (nao x nao, row-major, spherical).  V uses charges Z at atom xyz. */
void orc_int1e(int nshell, const int *sh_l, const int *sh_nprim, const int *sh_poff,
               const int *sh_aoff, const double *sh_xyz, const double *exps, const double *coefs,
               int nao, int natom, const double *zq, const double *axyz,
               double *S, double *T, double *V)
{
    build_c2s();
    memset(S, 0, sizeof(double) * nao * nao);
    memset(T, 0, sizeof(double) * nao * nao);
    memset(V, 0, sizeof(double) * nao * nao);
    double *R = (double *)malloc(sizeof(double) * RDIM * RDIM * RDIM);
    for (int A = 0; A < nshell; ++A)
        for (int B = 0; B < nshell; ++B) {
            int la = sh_l[A], lb = sh_l[B];
            int nca = NCART(la), ncb = NCART(lb);
            int lmna[MAXCART][3], lmnb[MAXCART][3];
            cart_components(la, lmna); cart_components(lb, lmnb);
            double sc[MAXCART * MAXCART] = {0}, tc[MAXCART * MAXCART] = {0}, vc[MAXCART * MAXCART] = {0};
            const double *ra = sh_xyz + 3 * A, *rb = sh_xyz + 3 * B;
            for (int ip = 0; ip < sh_nprim[A]; ++ip)
                for (int jp = 0; jp < sh_nprim[B]; ++jp) {
                    double a = exps[sh_poff[A] + ip], b = exps[sh_poff[B] + jp];
                    double cc = coefs[sh_poff[A] + ip] * coefs[sh_poff[B] + jp];
                    double p = a + b;
                    etab_t E[3];
                    for (int d = 0; d < 3; ++d) hermite_e(la, lb + 2, a, b, ra[d], rb[d], E[d]);
                    double s3 = pow(M_PI / p, 1.5);
                    double P[3];
                    for (int d = 0; d < 3; ++d) P[d] = (a * ra[d] + b * rb[d]) / p;
                    for (int ia = 0; ia < nca; ++ia)
                        for (int ib = 0; ib < ncb; ++ib) {
                            const int *u = lmna[ia], *w = lmnb[ib];
                            double s1[3], t1[3];
                            for (int d = 0; d < 3; ++d) {
                                int i = u[d], j = w[d];
                                s1[d] = E[d][i][j][0];
                                double tt = -2.0 * b * (2 * j + 1) * E[d][i][j][0] + 4.0 * b * b * E[d][i][j + 2][0];
                                if (j >= 2) tt += j * (j - 1) * E[d][i][j - 2][0];
                                t1[d] = -0.5 * tt;
                            }
                            sc[ia * ncb + ib] += cc * s3 * s1[0] * s1[1] * s1[2];
                            tc[ia * ncb + ib] += cc * s3 * (t1[0] * s1[1] * s1[2] + s1[0] * t1[1] * s1[2] + s1[0] * s1[1] * t1[2]);
                        }
                    int L = la + lb;
                    for (int at = 0; at < natom; ++at) {
                        if (zq[at] == 0.0) continue;
                        hermite_r(L, p, P[0] - axyz[3 * at], P[1] - axyz[3 * at + 1], P[2] - axyz[3 * at + 2], R);
                        double pref = -zq[at] * 2.0 * M_PI / p * cc;
                        for (int ia = 0; ia < nca; ++ia)
                            for (int ib = 0; ib < ncb; ++ib) {
                                const int *u = lmna[ia], *w = lmnb[ib];
                                double v = 0.0;
                                for (int t = 0; t <= u[0] + w[0]; ++t)
                                    for (int uu = 0; uu <= u[1] + w[1]; ++uu)
                                        for (int vv = 0; vv <= u[2] + w[2]; ++vv)
                                            v += E[0][u[0]][w[0]][t] * E[1][u[1]][w[1]][uu] * E[2][u[2]][w[2]][vv] *
                                                 R[(t * RDIM + uu) * RDIM + vv];
                                vc[ia * ncb + ib] += pref * v;
                            }
                    }
                }
            /* cart -> sph on both indices, scatter */
            double tmp[MAXCART * MAXCART], sph[MAXCART * MAXCART];
            double *mats[3] = {sc, tc, vc};
            double *outs[3] = {S, T, V};
            int nsa = NSPH(la), nsb = NSPH(lb);
            for (int k = 0; k < 3; ++k) {
                c2s_index(la, 1, ncb, mats[k], tmp);
                c2s_index(lb, nsa, 1, tmp, sph);
                for (int i = 0; i < nsa; ++i)
                    for (int j = 0; j < nsb; ++j)
                        outs[k][(sh_aoff[A] + i) * nao + sh_aoff[B] + j] = sph[i * nsb + j];
            }
        }
    free(R);
}

/* ----------------------------------------------------------- ERI core */
typedef struct {
    int l, nprim;
    const double *xyz, *exps, *coefs;
} shell_t;

static const double ORIGIN[3] = {0.0, 0.0, 0.0};
static const double ZERO_EXP[1] = {0.0};
static const double UNIT_COEF[1] = {1.0};

/* A "unit" shell (l = -1): exponent 0, coefficient 1, no angular factor.  Used to
 * write three- and two-centre integrals as four-centre ones. */
static shell_t unit_shell(void)
{
    shell_t s; s.l = -1; s.nprim = 1; s.xyz = ORIGIN; s.exps = ZERO_EXP; s.coefs = UNIT_COEF; return s;
}

static int shell_ncart(const shell_t *s) { return s->l < 0 ? 1 : NCART(s->l); }
static int shell_nsph(const shell_t *s) { return s->l < 0 ? 1 : NSPH(s->l); }

/* out: spherical block [nsa][nsb][nsc][nsd], caller provides buffers */
static void eri_quartet(const shell_t *A, const shell_t *B, const shell_t *C, const shell_t *D,
                        double *out, double *cart, double *tmp, double *R)
{
    int la = A->l < 0 ? 0 : A->l, lb = B->l < 0 ? 0 : B->l;
    int lc = C->l < 0 ? 0 : C->l, ld = D->l < 0 ? 0 : D->l;
    int nca = NCART(la), ncb = NCART(lb), ncc = NCART(lc), ncd = NCART(ld);
    int lmna[MAXCART][3], lmnb[MAXCART][3], lmnc[MAXCART][3], lmnd[MAXCART][3];
    cart_components(la, lmna); cart_components(lb, lmnb);
    cart_components(lc, lmnc); cart_components(ld, lmnd);
    int ntot = nca * ncb * ncc * ncd;
    memset(cart, 0, sizeof(double) * ntot);
    int L = la + lb + lc + ld;
    for (int ip = 0; ip < A->nprim; ++ip)
        for (int jp = 0; jp < B->nprim; ++jp) {
            double a = A->exps[ip], b = B->exps[jp], p = a + b;
            etab_t Eab[3];
            double P[3];
            for (int d = 0; d < 3; ++d) {
                hermite_e(la, lb, a, b, A->xyz[d], B->xyz[d], Eab[d]);
                P[d] = (a * A->xyz[d] + b * B->xyz[d]) / p;
            }
            double cab = A->coefs[ip] * B->coefs[jp];
            for (int kp = 0; kp < C->nprim; ++kp)
                for (int lp = 0; lp < D->nprim; ++lp) {
                    double c = C->exps[kp], d_ = D->exps[lp], q = c + d_;
                    etab_t Ecd[3];
                    double Q[3];
                    for (int d = 0; d < 3; ++d) {
                        hermite_e(lc, ld, c, d_, C->xyz[d], D->xyz[d], Ecd[d]);
                        Q[d] = (c * C->xyz[d] + d_ * D->xyz[d]) / q;
                    }
                    double alpha = p * q / (p + q);
                    double pref = 2.0 * pow(M_PI, 2.5) / (p * q * sqrt(p + q)) * cab * C->coefs[kp] * D->coefs[lp];
                    hermite_r(L, alpha, P[0] - Q[0], P[1] - Q[1], P[2] - Q[2], R);
                    int idx = 0;
                    for (int ia = 0; ia < nca; ++ia)
                        for (int ib = 0; ib < ncb; ++ib) {
                            const int *ua = lmna[ia], *ub = lmnb[ib];
                            int tx = ua[0] + ub[0], ty = ua[1] + ub[1], tz = ua[2] + ub[2];
                            for (int ic = 0; ic < ncc; ++ic)
                                for (int id = 0; id < ncd; ++id, ++idx) {
                                    const int *uc = lmnc[ic], *ud = lmnd[id];
                                    int sx = uc[0] + ud[0], sy = uc[1] + ud[1], sz = uc[2] + ud[2];
                                    double v = 0.0;
                                    for (int t = 0; t <= tx; ++t) {
                                        double ex = Eab[0][ua[0]][ub[0]][t];
                                        for (int u = 0; u <= ty; ++u) {
                                            double exy = ex * Eab[1][ua[1]][ub[1]][u];
                                            for (int w = 0; w <= tz; ++w) {
                                                double e3 = exy * Eab[2][ua[2]][ub[2]][w];
                                                double inner = 0.0;
                                                for (int tt = 0; tt <= sx; ++tt) {
                                                    double fx = Ecd[0][uc[0]][ud[0]][tt];
                                                    for (int uu = 0; uu <= sy; ++uu) {
                                                        double fxy = fx * Ecd[1][uc[1]][ud[1]][uu];
                                                        for (int ww = 0; ww <= sz; ++ww) {
                                                            double sgn = ((tt + uu + ww) & 1) ? -1.0 : 1.0;
                                                            inner += sgn * fxy * Ecd[2][uc[2]][ud[2]][ww] *
                                                                     R[((t + tt) * RDIM + (u + uu)) * RDIM + (w + ww)];
                                                        }
                                                    }
                                                }
                                                v += e3 * inner;
                                            }
                                        }
                                    }
                                    cart[idx] += pref * v;
                                }
                        }
                }
        }
    /* cart -> sph, one index at a time */
    int nsa = shell_nsph(A), nsb = shell_nsph(B), nsc = shell_nsph(C), nsd = shell_nsph(D);
    c2s_index(A->l, 1, ncb * ncc * ncd, cart, tmp);
    c2s_index(B->l, nsa, ncc * ncd, tmp, cart);
    c2s_index(C->l, nsa * nsb, ncd, cart, tmp);
    c2s_index(D->l, nsa * nsb * nsc, 1, tmp, out);
    (void)shell_ncart;
}

static shell_t get_shell(int i, const int *sh_l, const int *sh_nprim, const int *sh_poff,
                         const double *sh_xyz, const double *exps, const double *coefs)
{
    shell_t s;
    s.l = sh_l[i]; s.nprim = sh_nprim[i]; s.xyz = sh_xyz + 3 * i;
    s.exps = exps + sh_poff[i]; s.coefs = coefs + sh_poff[i];
    return s;
}

#define QBUF (MAXCART * MAXCART * MAXCART * MAXCART)

/* Full (nao^4) ERI tensor, chemists' notation eri[i][j][k][l] = (ij|kl), using the
 * 8-fold shell symmetry. */
void orc_eri4(int nshell, const int *sh_l, const int *sh_nprim, const int *sh_poff,
              const int *sh_aoff, const double *sh_xyz, const double *exps, const double *coefs,
              int nao, double *eri)
{
    build_c2s();
    long n = nao;
    int npair = nshell * (nshell + 1) / 2;
    int *pa = (int *)malloc(sizeof(int) * npair), *pb = (int *)malloc(sizeof(int) * npair);
    int k = 0;
    for (int a = 0; a < nshell; ++a) for (int b = 0; b <= a; ++b) { pa[k] = a; pb[k] = b; ++k; }
#pragma omp parallel
    {
        double *out = (double *)malloc(sizeof(double) * QBUF);
        double *cart = (double *)malloc(sizeof(double) * QBUF);
        double *tmp = (double *)malloc(sizeof(double) * QBUF);
        double *R = (double *)malloc(sizeof(double) * RDIM * RDIM * RDIM);
#pragma omp for schedule(dynamic, 1)
        for (int ij = npair - 1; ij >= 0; --ij)
            for (int kl = 0; kl <= ij; ++kl) {
                int a = pa[ij], b = pb[ij], c = pa[kl], d = pb[kl];
                shell_t A = get_shell(a, sh_l, sh_nprim, sh_poff, sh_xyz, exps, coefs);
                shell_t B = get_shell(b, sh_l, sh_nprim, sh_poff, sh_xyz, exps, coefs);
                shell_t C = get_shell(c, sh_l, sh_nprim, sh_poff, sh_xyz, exps, coefs);
                shell_t D = get_shell(d, sh_l, sh_nprim, sh_poff, sh_xyz, exps, coefs);
                eri_quartet(&A, &B, &C, &D, out, cart, tmp, R);
                int na = NSPH(A.l), nb = NSPH(B.l), nc = NSPH(C.l), nd = NSPH(D.l);
                int idx = 0;
                for (int i = 0; i < na; ++i)
                    for (int j = 0; j < nb; ++j)
                        for (int kk = 0; kk < nc; ++kk)
                            for (int l = 0; l < nd; ++l, ++idx) {
                                long I = sh_aoff[a] + i, J = sh_aoff[b] + j, K = sh_aoff[c] + kk, Lx = sh_aoff[d] + l;
                                double v = out[idx];
                                eri[((I * n + J) * n + K) * n + Lx] = v;
                                eri[((J * n + I) * n + K) * n + Lx] = v;
                                eri[((I * n + J) * n + Lx) * n + K] = v;
                                eri[((J * n + I) * n + Lx) * n + K] = v;
                                eri[((K * n + Lx) * n + I) * n + J] = v;
                                eri[((Lx * n + K) * n + I) * n + J] = v;
                                eri[((K * n + Lx) * n + J) * n + I] = v;
                                eri[((Lx * n + K) * n + J) * n + I] = v;
                            }
            }
        free(out); free(cart); free(tmp); free(R);
    }
    free(pa); free(pb);
}

/* Schwarz shell-pair bounds Q[A][B] = sqrt(max |(ab|ab)|)  (mqc_libcint_direct.f90:129-146) */
void orc_schwarz(int nshell, const int *sh_l, const int *sh_nprim, const int *sh_poff,
                 const double *sh_xyz, const double *exps, const double *coefs, double *Q)
{
    build_c2s();
#pragma omp parallel
    {
        double *out = (double *)malloc(sizeof(double) * QBUF);
        double *cart = (double *)malloc(sizeof(double) * QBUF);
        double *tmp = (double *)malloc(sizeof(double) * QBUF);
        double *R = (double *)malloc(sizeof(double) * RDIM * RDIM * RDIM);
#pragma omp for schedule(dynamic, 1) collapse(2)
        for (int a = 0; a < nshell; ++a)
            for (int b = 0; b < nshell; ++b) {
                if (b > a) continue;
                shell_t A = get_shell(a, sh_l, sh_nprim, sh_poff, sh_xyz, exps, coefs);
                shell_t B = get_shell(b, sh_l, sh_nprim, sh_poff, sh_xyz, exps, coefs);
                eri_quartet(&A, &B, &A, &B, out, cart, tmp, R);
                int na = NSPH(A.l), nb = NSPH(B.l);
                double m = 0.0;
                for (int i = 0; i < na; ++i)
                    for (int j = 0; j < nb; ++j) {
                        double v = fabs(out[((i * nb + j) * na + i) * nb + j]);
                        if (v > m) m = v;
                    }
                /* the reference takes the max over the whole (ab|ab) block */
                for (int i = 0; i < na * nb * na * nb; ++i) if (fabs(out[i]) > m) m = fabs(out[i]);
                Q[a * nshell + b] = Q[b * nshell + a] = sqrt(m);
            }
        free(out); free(cart); free(tmp); free(R);
    }
}

/* (ab|P): out[nao][nao][naux];  basis 1 = orbital, basis 2 = auxiliary */
void orc_eri3c(int nshell, const int *sh_l, const int *sh_nprim, const int *sh_poff,
               const int *sh_aoff, const double *sh_xyz, const double *exps, const double *coefs, int nao,
               int xshell, const int *x_l, const int *x_nprim, const int *x_poff,
               const int *x_aoff, const double *x_xyz, const double *x_exps, const double *x_coefs, int naux,
               double *out3)
{
    build_c2s();
#pragma omp parallel
    {
        double *out = (double *)malloc(sizeof(double) * QBUF);
        double *cart = (double *)malloc(sizeof(double) * QBUF);
        double *tmp = (double *)malloc(sizeof(double) * QBUF);
        double *R = (double *)malloc(sizeof(double) * RDIM * RDIM * RDIM);
        shell_t U = unit_shell();
#pragma omp for schedule(dynamic, 1)
        for (int a = 0; a < nshell; ++a)
            for (int b = 0; b <= a; ++b)
                for (int P = 0; P < xshell; ++P) {
                    shell_t A = get_shell(a, sh_l, sh_nprim, sh_poff, sh_xyz, exps, coefs);
                    shell_t B = get_shell(b, sh_l, sh_nprim, sh_poff, sh_xyz, exps, coefs);
                    shell_t C = get_shell(P, x_l, x_nprim, x_poff, x_xyz, x_exps, x_coefs);
                    eri_quartet(&A, &B, &C, &U, out, cart, tmp, R);
                    int na = NSPH(A.l), nb = NSPH(B.l), nc = NSPH(C.l);
                    for (int i = 0; i < na; ++i)
                        for (int j = 0; j < nb; ++j)
                            for (int k = 0; k < nc; ++k) {
                                long I = sh_aoff[a] + i, J = sh_aoff[b] + j, K = x_aoff[P] + k;
                                double v = out[(i * nb + j) * nc + k];
                                out3[(I * nao + J) * naux + K] = v;
                                out3[(J * nao + I) * naux + K] = v;
                            }
                }
        free(out); free(cart); free(tmp); free(R);
    }
}

/* (P|Q): out[naux][naux] */
void orc_eri2c(int xshell, const int *x_l, const int *x_nprim, const int *x_poff,
               const int *x_aoff, const double *x_xyz, const double *x_exps, const double *x_coefs, int naux,
               double *out2)
{
    build_c2s();
#pragma omp parallel
    {
        double *out = (double *)malloc(sizeof(double) * QBUF);
        double *cart = (double *)malloc(sizeof(double) * QBUF);
        double *tmp = (double *)malloc(sizeof(double) * QBUF);
        double *R = (double *)malloc(sizeof(double) * RDIM * RDIM * RDIM);
        shell_t U = unit_shell();
#pragma omp for schedule(dynamic, 1)
        for (int P = 0; P < xshell; ++P)
            for (int Qs = 0; Qs <= P; ++Qs) {
                shell_t A = get_shell(P, x_l, x_nprim, x_poff, x_xyz, x_exps, x_coefs);
                shell_t C = get_shell(Qs, x_l, x_nprim, x_poff, x_xyz, x_exps, x_coefs);
                eri_quartet(&A, &U, &C, &U, out, cart, tmp, R);
                int na = NSPH(A.l), nc = NSPH(C.l);
                for (int i = 0; i < na; ++i)
                    for (int k = 0; k < nc; ++k) {
                        long I = x_aoff[P] + i, K = x_aoff[Qs] + k;
                        out2[I * naux + K] = out2[K * naux + I] = out[i * nc + k];
                    }
            }
        free(out); free(cart); free(tmp); free(R);
    }
}

/* ------------------------------------------------- AO values on a grid */
/* ao[npts][nao] and (optionally) grad[3][npts][nao]; follows eval_ao_block's
 * (n_points, n_ao) layout (mqc_libcint_ao.f90:69-91). */
void orc_eval_ao(int nshell, const int *sh_l, const int *sh_nprim, const int *sh_poff,
                 const int *sh_aoff, const double *sh_xyz, const double *exps, const double *coefs,
                 int nao, long npts, const double *pts /* [npts][3] */, double *ao, double *grad)
{
    build_c2s();
#pragma omp parallel for schedule(static)
    for (long g = 0; g < npts; ++g) {
        for (int A = 0; A < nshell; ++A) {
            int l = sh_l[A], nc = NCART(l), ns = NSPH(l);
            int lmn[MAXCART][3];
            cart_components(l, lmn);
            double dx = pts[3 * g] - sh_xyz[3 * A], dy = pts[3 * g + 1] - sh_xyz[3 * A + 1], dz = pts[3 * g + 2] - sh_xyz[3 * A + 2];
            double r2 = dx * dx + dy * dy + dz * dz;
            double rad = 0.0, drad = 0.0;   /* sum c e^{-a r2}; sum -2a c e^{-a r2} */
            for (int ip = 0; ip < sh_nprim[A]; ++ip) {
                double a = exps[sh_poff[A] + ip], e = coefs[sh_poff[A] + ip] * exp(-a * r2);
                rad += e; drad += -2.0 * a * e;
            }
            double px[LMAX + 2], py[LMAX + 2], pz[LMAX + 2];
            px[0] = py[0] = pz[0] = 1.0;
            for (int i = 1; i <= l + 1; ++i) { px[i] = px[i - 1] * dx; py[i] = py[i - 1] * dy; pz[i] = pz[i - 1] * dz; }
            double vc[MAXCART], gc[3][MAXCART];
            for (int k = 0; k < nc; ++k) {
                int a = lmn[k][0], b = lmn[k][1], c = lmn[k][2];
                double mono = px[a] * py[b] * pz[c];
                vc[k] = mono * rad;
                gc[0][k] = (a ? a * px[a - 1] * py[b] * pz[c] : 0.0) * rad + mono * dx * drad;
                gc[1][k] = (b ? b * px[a] * py[b - 1] * pz[c] : 0.0) * rad + mono * dy * drad;
                gc[2][k] = (c ? c * px[a] * py[b] * pz[c - 1] : 0.0) * rad + mono * dz * drad;
            }
            for (int s = 0; s < ns; ++s) {
                double v = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
                for (int k = 0; k < nc; ++k) {
                    double cs = (g_cartesian && l >= 2) ? (s == k ? 1.0 : 0.0) : C2S[l][s][k];
                    v += cs * vc[k]; g0 += cs * gc[0][k]; g1 += cs * gc[1][k]; g2 += cs * gc[2][k];
                }
                ao[g * nao + sh_aoff[A] + s] = v;
                if (grad) {
                    grad[(0 * npts + g) * nao + sh_aoff[A] + s] = g0;
                    grad[(1 * npts + g) * nao + sh_aoff[A] + s] = g1;
                    grad[(2 * npts + g) * nao + sh_aoff[A] + s] = g2;
                }
            }
        }
    }
}

void orc_boys(int nmax, double T, double *F) { boys(nmax, T, F); }
