// kern_df.hip -- density-fitted Coulomb and exchange for a whole batch.
//
// Replaces cuestDFIntPlanCreate / cuestDFCoulombCompute / cuestDFSymmetricExchangeCompute
// (backends/cuest/backend/mqc_cuest_integrals.f90:531-736,1636-1748); numerics follow the CPU
// path: build_df_tensor (backends/libcint/mqc_libcint_integrals.F90:913-1038: (mu nu|P) and (P|Q)
// over the same kernel, B = (mu nu|Q) J^{-1/2}) and build_fock_df
// (backends/libcint/mqc_libcint_rhf.f90:1576-1646: c_P = sum B D, J = sum_P B_P c_P,
// W_P = B_P C_occ, K = 2 sum_P W_P W_P^T).
//
// J and K depend on the fitted tensor only through B B^T, which is the same for ANY square root of
// the metric; the engine uses the Cholesky factor (B = L^{-1} (P|mu nu)) instead of the symmetric
// J^{-1/2} the reference forms by eigen-decomposition.  The reference drops metric eigenvalues
// below 1e-10; a metric that is that close to singular is REFUSED here (pivot test) rather than
// silently treated differently.
//
// Layouts per fragment:  A3 / Bfit [naux][npair] (auxiliary index major, packed AO pairs fastest:
// every pass streams contiguous rows), metric / Linv [naux][naux].
#include "engine.hpp"
#include "md_integrals.hpp"
#include <cstdlib>

typedef double v4f64 __attribute__((ext_vector_type(4)));

namespace mqc {

constexpr int DF_NT = 256;
constexpr double DF_PIVOT_FLOOR = 1.0e-10;    // metric_inverse_sqrt's eigen-threshold (mqc_libcint_integrals.F90:1002)

__device__ __forceinline__ ShellRef df_shell(const TopologyDev& tp, const double* xyz, int s)
{
    ShellRef r;
    r.nprim = tp.sh_nprim[s];
    r.exps = tp.exps + tp.sh_poff[s];
    r.coefs = tp.coefs + tp.sh_poff[s];
    const int at = tp.sh_atom[s];
    r.x = xyz[3 * at]; r.y = xyz[3 * at + 1]; r.z = xyz[3 * at + 2];
    return r;
}

__device__ __forceinline__ ShellRef unit_shell(const double* unit)
{
    // exponent 0, coefficient 1: turns a four-centre routine into a three- or two-centre one
    ShellRef r;
    r.nprim = 1; r.exps = unit; r.coefs = unit + 1; r.x = 0.0; r.y = 0.0; r.z = 0.0;
    return r;
}

// cart -> sph of one index, rolled (DF blocks are not the hot path)
template <int L>
__device__ __forceinline__ void df_c2s(const double* c2s, int pre, int post, const double* in, double* out)
{
    constexpr int NC = ncart(L), NS = nsph(L);
    for (int a = 0; a < pre; ++a)
        for (int s = 0; s < NS; ++s)
            for (int r = 0; r < post; ++r) {
                double v = 0.0;
                for (int c = 0; c < NC; ++c) v += c2s_coef<L>(c2s, s, c) * in[(a * NC + c) * post + r];
                out[(a * NS + s) * post + r] = v;
            }
}

// (ab|P): thread = (task, fragment); task = (A, B, P) with la >= lb
template <int LA, int LB, int LC>
__global__ void __launch_bounds__(64) df3c_kernel(BatchView bv, const int* __restrict__ tasks, int ntask)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long)ntask * bv.nfrag) return;
    const int it = (int)(tid / bv.nfrag), f = (int)(tid % bv.nfrag);
    const int A = tasks[3 * it], B = tasks[3 * it + 1], P = tasks[3 * it + 2];
    const double* xyz = bv.xyz + (size_t)f * bv.topo.natoms * 3;
    ShellRef a = df_shell(bv.topo, xyz, A), b = df_shell(bv.topo, xyz, B), c = df_shell(bv.aux, xyz, P);
    ShellRef u = unit_shell(bv.unit);
    constexpr int NC = ncart(LA) * ncart(LB) * ncart(LC);
    constexpr int NSA = nsph(LA), NSB = nsph(LB), NSC = nsph(LC);
    double cart[NC], tmp[NC];
    eri_cart_block<LA, LB, LC, 0>(a, b, c, u, bv.boys, cart);
    df_c2s<LA>(bv.c2s, 1, ncart(LB) * ncart(LC), cart, tmp);
    df_c2s<LB>(bv.c2s, NSA, ncart(LC), tmp, cart);
    df_c2s<LC>(bv.c2s, NSA * NSB, 1, cart, tmp);
    const int oa = bv.topo.sh_aoff[A], ob = bv.topo.sh_aoff[B], oc = bv.aux.sh_aoff[P];
    const size_t np = (size_t)bv.npair;
    double* A3 = bv.df_a3 + (size_t)f * bv.naux * np;
    for (int i = 0; i < NSA; ++i)
        for (int j = 0; j < NSB; ++j) {
            if (A == B && j > i) continue;
            const int I = oa + i, J = ob + j;
            const size_t pr = I >= J ? (size_t)I * (I + 1) / 2 + J : (size_t)J * (J + 1) / 2 + I;
            for (int k = 0; k < NSC; ++k) A3[(size_t)(oc + k) * np + pr] = tmp[(i * NSB + j) * NSC + k];
        }
}

// (P|Q): thread = (aux shell pair P >= Q ordered lp >= lq, fragment)
template <int LP, int LQ>
__global__ void __launch_bounds__(64) df2c_kernel(BatchView bv, const int* __restrict__ tasks, int ntask)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long)ntask * bv.nfrag) return;
    const int it = (int)(tid / bv.nfrag), f = (int)(tid % bv.nfrag);
    const int P = tasks[2 * it], Q = tasks[2 * it + 1];
    const double* xyz = bv.xyz + (size_t)f * bv.topo.natoms * 3;
    ShellRef p = df_shell(bv.aux, xyz, P), q = df_shell(bv.aux, xyz, Q);
    ShellRef u = unit_shell(bv.unit);
    constexpr int NC = ncart(LP) * ncart(LQ);
    constexpr int NSP = nsph(LP), NSQ = nsph(LQ);
    double cart[NC], tmp[NC];
    eri_cart_block<LP, 0, LQ, 0>(p, u, q, u, bv.boys, cart);
    df_c2s<LP>(bv.c2s, 1, ncart(LQ), cart, tmp);
    df_c2s<LQ>(bv.c2s, NSP, 1, tmp, cart);
    const int op = bv.aux.sh_aoff[P], oq = bv.aux.sh_aoff[Q];
    const int na = bv.naux;
    double* M = bv.df_metric + (size_t)f * na * na;
    for (int i = 0; i < NSP; ++i)
        for (int k = 0; k < NSQ; ++k) {
            const double v = cart[i * NSQ + k];
            M[(size_t)(op + i) * na + oq + k] = v;
            M[(size_t)(oq + k) * na + op + i] = v;
        }
}

// In-place Cholesky M = L L^T (lower) and Linv = L^{-1}, one workgroup per fragment, matrices in global memory (L2).
// Step k: column k goes to LDS (scaled), then the trailing update runs row-wise -- a wave per row, lanes along the
// row -- so every access to the matrix is a contiguous row segment (the round-1 version walked (i, j) by integer
// division with a strided column read per element: 0.8 GB of L2 transactions per fragment, 38 ms for 2016 dimers).
//
// Which fit is it?  The reference forms J^{-1/2} = U s^{-1/2} U^T over the eigenvalues ABOVE 1e-10
// (metric_inverse_sqrt, mqc_libcint_integrals.F90:992-1038).  J and K depend on the fitted tensor through B^T B only,
// and B^T B = A3^T J^{-1} A3 for ANY square root -- so when no eigenvalue is cut, the Cholesky fit gives the
// reference's J and K exactly.  No eigenvalue is cut when lambda_min > 1e-10, and lambda_min = 1/||J^{-1}||_2 >=
// 1/||L^{-1}||_F^2: the kernel checks that bound (and the pivots) and flags the fragment otherwise (scal[7] = 1);
// flagged fragments take the eigen-decomposition path below, which applies the cut as the reference does.
__global__ void __launch_bounds__(DF_NT) df_cholesky_kernel(BatchView bv)
{
    extern __shared__ double colbuf[];       // [na] column k, then reduction scratch
    const int f = blockIdx.x, tid = threadIdx.x, na = bv.naux;
    const int lane = tid & 63, wave = tid >> 6;
    double* M = bv.df_metric + (size_t)f * na * na;
    double* Li = bv.df_linv + (size_t)f * na * na;
    double* save = bv.df_work + (size_t)f * na * na;     // the diagonal survives here (first na entries)
    __shared__ double s_d, s_red[DF_NT / 64];
    __shared__ int s_bad;
    if (tid == 0) s_bad = 0;
    for (int i = tid; i < na; i += DF_NT) save[i] = M[(size_t)i * na + i];
    __syncthreads();
    for (int k = 0; k < na; ++k) {
        if (tid == 0) {
            const double akk = M[(size_t)k * na + k];
            if (!(akk > DF_PIVOT_FLOOR)) s_bad = 1;
            s_d = sqrt(akk > DF_PIVOT_FLOOR ? akk : 1.0);
        }
        __syncthreads();
        const double d = s_d, rd = 1.0 / d;
        for (int i = k + tid; i < na; i += DF_NT) {
            const double v = (i == k) ? d : M[(size_t)i * na + k] * rd;
            M[(size_t)i * na + k] = v;
            colbuf[i] = v;
        }
        __syncthreads();
        for (int i = k + 1 + wave; i < na; i += DF_NT / 64) {
            const double ci = colbuf[i];
            double* __restrict__ row = M + (size_t)i * na;
            for (int j = k + 1 + lane; j <= i; j += 64) row[j] -= ci * colbuf[j];
        }
        __syncthreads();
    }
    // Linv by forward substitution, one column per thread: L X = I  (row r of L through LDS)
    double fro = 0.0;
    for (int c0 = 0; c0 < na; c0 += DF_NT) {
        const int c = c0 + tid;
        for (int r = 0; r < na; ++r) {
            __syncthreads();
            for (int t = tid; t <= r; t += DF_NT) colbuf[t] = M[(size_t)r * na + t];
            __syncthreads();
            if (c < na) {
                double v = 0.0;
                if (r >= c) {
                    double s = (r == c) ? 1.0 : 0.0;
                    for (int t = c; t < r; ++t) s -= colbuf[t] * Li[(size_t)t * na + c];
                    v = s / colbuf[r];
                    fro += v * v;
                }
                Li[(size_t)r * na + c] = v;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) fro += __shfl_xor(fro, off, 64);
    if (lane == 0) s_red[wave] = fro;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < DF_NT / 64; ++w) tot += s_red[w];
        // lambda_min >= 1 / ||L^-1||_F^2 must clear the reference's cut with a margin for the rounding of the factor
        const bool safe = !s_bad && tot > 0.0 && (1.0 / tot) > 2.0 * DF_PIVOT_FLOOR;
        bv.scal[(size_t)f * 8 + 7] = safe ? 0.0 : 1.0;
        s_bad = safe ? 0 : 1;
    }
    __syncthreads();
    if (s_bad) {
        // hand the eigen path a pristine metric: the strict upper triangle was never touched, the diagonal was saved
        for (int i = wave; i < na; i += DF_NT / 64)
            for (int j = lane; j < i; j += 64) M[(size_t)i * na + j] = M[(size_t)j * na + i];
        __syncthreads();
        for (int i = tid; i < na; i += DF_NT) M[(size_t)i * na + i] = save[i];
    }
}

// The same factorisation and inverse, blocked 16 x 16 on the FP64 matrix cores (default; MQC_HIP_DF_CHOL_V1=1 runs the
// kernel above).  Left-looking by block columns kb:
//   panel   P[rt] = M[rt][kb] - sum_{jb<kb} L[rt][jb] L[kb][jb]^T      MFMA jobs (row tile rt >= kb), result to LDS
//   factor  the 16 x 16 diagonal block in LDS (16 rank-1 steps), its triangular inverse Dinv (one column per lane)
//   solve   L[rt][kb] = P[rt] Dinv^T for rt > kb                        MFMA, 4 k-steps per tile
// then L^{-1} by block rows:  X[rb][cb] = -Dinv[rb] sum_{cb<=sb<rb} L[rb][sb] X[sb][cb]  -- the accumulator of the inner
// product is already in B-operand layout for the product with Dinv (C/D row = hi + 4r  <->  B k = 4 ks + hi).
// Rows and columns beyond na behave as an identity block, so the last, ragged block needs no special code.
// The unblocked kernel re-reads the trailing matrix from L2 na times (87 GB for 2016 dimers) and its inverse walks
// O(na^3) dependent loads: 28 ms; this one moves each block O(nb) times.
__global__ void __launch_bounds__(DF_NT) df_cholesky_mfma_kernel(BatchView bv)
{
    extern __shared__ double lds[];
    const int f = blockIdx.x, tid = threadIdx.x, na = bv.naux;
    const int lane = tid & 63, wave = tid >> 6, lo = lane & 15, hi = lane >> 4;
    const int nb = (na + 15) >> 4, NP = nb << 4;
    double* M = bv.df_metric + (size_t)f * na * na;
    double* Li = bv.df_linv + (size_t)f * na * na;
    double* save = bv.df_work + (size_t)f * na * na;
    double* P = lds;                   // [NP][16] panel of the current block column
    double* Dg = P + (size_t)NP * 16;  // [16][17] diagonal block
    double* Di = Dg + 16 * 17;         // [16][17] its inverse
    __shared__ double s_red[DF_NT / 64];
    __shared__ int s_bad;
    if (tid == 0) s_bad = 0;
    for (int i = tid; i < na; i += DF_NT) save[i] = M[(size_t)i * na + i];
    __syncthreads();
    auto Lat = [&](int r, int c) -> double { return (r < na && c < na) ? M[(size_t)r * na + c] : (r == c ? 1.0 : 0.0); };

    for (int kb = 0; kb < nb; ++kb) {
        // ---- panel update
        for (int rt = kb + wave; rt < nb; rt += DF_NT / 64) {
            v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
            for (int jb = 0; jb < kb; ++jb) {
                double a[4], b[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    a[ks] = Lat(16 * rt + lo, 16 * jb + 4 * ks + hi);
                    b[ks] = Lat(16 * kb + lo, 16 * jb + 4 * ks + hi);
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rt + hi + 4 * r, col = 16 * kb + lo;
                // block column kb of M is still the original metric (left-looking: it is written only after this step), and
                // so is the upper triangle, which nothing ever writes: the diagonal tile reads symmetric values
                const double m = Lat(row, col);
                P[(size_t)row * 16 + lo] = m - acc[r];
            }
        }
        __syncthreads();
        // ---- diagonal block: copy (lower part is authoritative), factor, invert
        for (int idx = tid; idx < 256; idx += DF_NT) {
            const int i = idx >> 4, j = idx & 15;
            Dg[i * 17 + j] = (i >= j) ? P[(size_t)(16 * kb + i) * 16 + j] : 0.0;
            Di[i * 17 + j] = 0.0;
        }
        __syncthreads();
        for (int k = 0; k < 16; ++k) {
            if (tid == 0) {
                const double akk = Dg[k * 17 + k];
                const bool pad = 16 * kb + k >= na;
                if (!pad && !(akk > DF_PIVOT_FLOOR)) s_bad = 1;
                Dg[k * 17 + k] = sqrt((akk > DF_PIVOT_FLOOR) ? akk : 1.0);
            }
            __syncthreads();
            const double dk = Dg[k * 17 + k];
            if (tid < 16 && tid > k) Dg[tid * 17 + k] /= dk;
            __syncthreads();
            {
                const int i = tid >> 4, j = tid & 15;
                if (i >= j && j > k) Dg[i * 17 + j] -= Dg[i * 17 + k] * Dg[j * 17 + k];
            }
            __syncthreads();
        }
        if (tid < 16) {
            // column tid of Dinv by forward substitution
            const int c = tid;
            for (int r = c; r < 16; ++r) {
                double sum = (r == c) ? 1.0 : 0.0;
                for (int t = c; t < r; ++t) sum -= Dg[r * 17 + t] * Di[t * 17 + c];
                Di[r * 17 + c] = sum / Dg[r * 17 + r];
            }
        }
        __syncthreads();
        // diagonal block of L and of L^{-1} to global (lower parts only)
        for (int idx = tid; idx < 256; idx += DF_NT) {
            const int i = idx >> 4, j = idx & 15;
            const int row = 16 * kb + i, col = 16 * kb + j;
            if (i >= j && row < na && col < na) { M[(size_t)row * na + col] = Dg[i * 17 + j]; Li[(size_t)row * na + col] = Di[i * 17 + j]; }
        }
        // ---- solve the rows below: L[rt][kb] = P[rt] Dinv^T
        for (int rt = kb + 1 + wave; rt < nb; rt += DF_NT / 64) {
            v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double a = P[(size_t)(16 * rt + lo) * 16 + 4 * ks + hi];
                const double b = Di[lo * 17 + 4 * ks + hi];          // B[k][j] = Dinv[j][k]
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rt + hi + 4 * r, col = 16 * kb + lo;
                if (row < na && col < na) M[(size_t)row * na + col] = acc[r];
            }
        }
        __syncthreads();
    }
    // ---- L^{-1} by block rows (diagonal blocks are in place)
    double fro = 0.0;
    for (int rb = 0; rb < nb; ++rb) {
        for (int cb = wave; cb < rb; cb += DF_NT / 64) {
            v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
            for (int sb = cb; sb < rb; ++sb) {
                double a[4], b[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    a[ks] = Lat(16 * rb + lo, 16 * sb + 4 * ks + hi);
                    const int xr = 16 * sb + 4 * ks + hi, xc = 16 * cb + lo;
                    // X[sb][cb]: lower-triangular inverse; inside the diagonal block only xr >= xc is stored
                    b[ks] = (xr < na && xc < na && xr >= xc) ? Li[(size_t)xr * na + xc] : ((xr == xc) ? 1.0 : 0.0);
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
            }
            v4f64 x = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int dr = 16 * rb + lo, dc = 16 * rb + 4 * ks + hi;
                const double a = (dr < na && dc < na && dr >= dc) ? Li[(size_t)dr * na + dc] : ((dr == dc) ? 1.0 : 0.0);
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(a, -acc[ks], x, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rb + hi + 4 * r, col = 16 * cb + lo;
                if (row < na && col < na) { Li[(size_t)row * na + col] = x[r]; fro += x[r] * x[r]; }
            }
        }
        __syncthreads();
    }
    // diagonal blocks' share of the Frobenius norm
    for (int idx = tid; idx < nb * 256; idx += DF_NT) {
        const int b_ = idx >> 8, i = (idx >> 4) & 15, j = idx & 15;
        const int row = 16 * b_ + i, col = 16 * b_ + j;
        if (i >= j && row < na && col < na) { const double v = Li[(size_t)row * na + col]; fro += v * v; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) fro += __shfl_xor(fro, off, 64);
    if (lane == 0) s_red[wave] = fro;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < DF_NT / 64; ++w) tot += s_red[w];
        const bool safe = !s_bad && tot > 0.0 && (1.0 / tot) > 2.0 * DF_PIVOT_FLOOR;
        bv.scal[(size_t)f * 8 + 7] = safe ? 0.0 : 1.0;
        s_bad = safe ? 0 : 1;
    }
    __syncthreads();
    if (s_bad) {
        for (int i = wave; i < na; i += DF_NT / 64)
            for (int j = lane; j < i; j += 64) M[(size_t)i * na + j] = M[(size_t)j * na + i];
        __syncthreads();
        for (int i = tid; i < na; i += DF_NT) M[(size_t)i * na + i] = save[i];
    }
}

// Eigen path for flagged fragments: the reference's J^{-1/2} = U s^{-1/2} U^T over eigenvalues > 1e-10.
// One-sided (Hestenes) Jacobi on the ROWS of B = V^T M: plane rotations make the rows orthogonal; then row p of V^T is
// eigenvector p and lambda_p = b_p . v_p.  Rows are contiguous, a wave owns a pair of rows, the pairs of one round of
// the round-robin ordering are disjoint.  Rare (a well-conditioned auxiliary basis never comes here), so the kernel
// is written for robustness, not speed.
__global__ void __launch_bounds__(DF_NT) df_metric_eig_kernel(BatchView bv)
{
    const int f = blockIdx.x, tid = threadIdx.x, na = bv.naux;
    if (bv.scal[(size_t)f * 8 + 7] == 0.0) return;
    const int lane = tid & 63, wave = tid >> 6;
    double* Bm = bv.df_metric + (size_t)f * na * na;      // becomes V^T M
    double* Vt = bv.df_work + (size_t)f * na * na;        // V^T
    double* T = bv.df_linv + (size_t)f * na * na;         // result: J^{-1/2}
    __shared__ double s_off[DF_NT / 64];
    __shared__ int s_done;
    for (int idx = tid; idx < na * na; idx += DF_NT) Vt[idx] = (idx / na == idx % na) ? 1.0 : 0.0;
    __syncthreads();
    const int m = (na + 1) & ~1;                          // players of the tournament (one bye when na is odd)
    for (int sweep = 0; sweep < 40; ++sweep) {
        double worst = 0.0;
        for (int round = 0; round < m - 1; ++round) {
            for (int k = wave; k < m / 2; k += DF_NT / 64) {
                int p, q;
                if (k == 0) { p = m - 1; q = round; }
                else { p = (round + k) % (m - 1); q = (round - k + (m - 1)) % (m - 1); }
                if (p >= na || q >= na) continue;
                double* bp = Bm + (size_t)p * na; double* bq = Bm + (size_t)q * na;
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int j = lane; j < na; j += 64) { const double x = bp[j], y = bq[j]; al += x * x; be += y * y; ga += x * y; }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { al += __shfl_xor(al, off, 64); be += __shfl_xor(be, off, 64); ga += __shfl_xor(ga, off, 64); }
                const double lim = sqrt(al * be);
                if (!(fabs(ga) > 1.0e-15 * lim) || lim == 0.0) continue;
                worst = fmax(worst, fabs(ga) / lim);
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                double* vp = Vt + (size_t)p * na; double* vq = Vt + (size_t)q * na;
                for (int j = lane; j < na; j += 64) {
                    const double x = bp[j], y = bq[j];
                    bp[j] = c * x - sn * y; bq[j] = sn * x + c * y;
                    const double u = vp[j], w = vq[j];
                    vp[j] = c * u - sn * w; vq[j] = sn * u + c * w;
                }
            }
            __syncthreads();
        }
        if (lane == 0) s_off[wave] = worst;
        __syncthreads();
        if (tid == 0) {
            double w = 0.0;
            for (int k = 0; k < DF_NT / 64; ++k) w = fmax(w, s_off[k]);
            s_done = w < 1.0e-14;
        }
        __syncthreads();
        if (s_done) break;
    }
    // T = sum over kept p of v_p v_p^T / sqrt(lambda_p), lambda_p = b_p . v_p; the first row of B keeps 1/sqrt(lambda) (0 = cut)
    double* isl = Bm;      // reuse: inverse square roots go to a scratch row after the eigenvalues are known
    __syncthreads();
    double lam_keep = 0.0;
    (void)lam_keep;
    // eigenvalues first (rows of B are still needed), then overwrite row storage
    for (int p = wave; p < na; p += DF_NT / 64) {
        double s = 0.0;
        for (int j = lane; j < na; j += 64) s += Bm[(size_t)p * na + j] * Vt[(size_t)p * na + j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) T[p] = (s > DF_PIVOT_FLOOR) ? 1.0 / sqrt(s) : 0.0;      // parked in T's first row for a moment
    }
    __syncthreads();
    for (int p = tid; p < na; p += DF_NT) isl[p] = T[p];
    __syncthreads();
    for (int idx = tid; idx < na * na; idx += DF_NT) {
        const int i = idx / na, j = idx - i * na;
        double s = 0.0;
        for (int p = 0; p < na; ++p) s += isl[p] * Vt[(size_t)p * na + i] * Vt[(size_t)p * na + j];
        T[idx] = s;
    }
    if (tid == 0) bv.scal[(size_t)f * 8 + 7] = 2.0;       // fitted with the full symmetric J^{-1/2}
}

// Bfit = Linv x A3 on the FP64 matrix cores: job = (16 auxiliary rows, 16 packed-pair columns); the lower-triangular
// factor stops the k loop at the diagonal block, the symmetric J^{-1/2} of the eigen path (flag 2) runs it in full.
__global__ void __launch_bounds__(DF_NT) df_fit_mfma_kernel(BatchView bv)
{
    const int f = blockIdx.y, na = bv.naux;
    const int np = bv.npair;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lo = lane & 15, hi = lane >> 4;
    const int nrt = (na + 15) >> 4, nct = (np + 15) >> 4;
    const bool full = bv.scal[(size_t)f * 8 + 7] == 2.0;
    const double* __restrict__ A3 = bv.df_a3 + (size_t)f * na * (size_t)np;
    double* __restrict__ Bf = bv.df_b + (size_t)f * na * (size_t)np;
    const double* __restrict__ Li = bv.df_linv + (size_t)f * na * na;
    for (int job = blockIdx.x * (DF_NT / 64) + wave; job < nrt * nct; job += gridDim.x * (DF_NT / 64)) {
        const int rt = job / nct, ct = job - rt * nct;
        const int R = 16 * rt + lo, col = 16 * ct + lo;
        const int kend = full ? na : min(na, 16 * (rt + 1));
        v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
        for (int s0 = 0; s0 < kend; s0 += 4) {
            const int sidx = s0 + hi;
            const double a = (R < na && sidx < kend && (full || sidx <= R)) ? Li[(size_t)R * na + sidx] : 0.0;
            const double b = (sidx < kend && col < np) ? A3[(size_t)sidx * np + col] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * rt + hi + 4 * r;
            if (row < na && col < np) Bf[(size_t)row * np + col] = acc[r];
        }
    }
}

__device__ __forceinline__ void df_unpack(int idx, int& k, int& l)
{
    k = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
    while ((k + 1) * (k + 2) / 2 <= idx) ++k;
    while (k * (k + 1) / 2 > idx) --k;
    l = idx - k * (k + 1) / 2;
}

// J: c_R = sum_pairs B[R][pair] (2 - delta) D[pair];  J[pair] = sum_R B[R][pair] c_R.  One workgroup per fragment.
__global__ void __launch_bounds__(DF_NT) df_j_kernel(BatchView bv, int only_active)
{
    extern __shared__ double lds[];
    const int f = blockIdx.x, tid = threadIdx.x, n = bv.n, na = bv.naux;
    if (only_active && bv.istate[4 * f] == ST_DONE) return;
    const size_t np = (size_t)bv.npair;
    const double* __restrict__ Bf = bv.df_b + (size_t)f * na * np;
    const double* __restrict__ D = bv.D + (size_t)f * n * n;
    double* __restrict__ J = bv.J + (size_t)f * n * n;
    double* Dp = lds;             // np
    double* c = lds + np;         // na
    for (int idx = tid; idx < (int)np; idx += DF_NT) {
        int k, l;
        df_unpack(idx, k, l);
        const double d = D[k * n + l];
        Dp[idx] = k == l ? d : 2.0 * d;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int R = wave; R < na; R += DF_NT / 64) {
        const double* __restrict__ row = Bf + (size_t)R * np;
        double s = 0.0;
        for (int idx = lane; idx < (int)np; idx += 64) s += row[idx] * Dp[idx];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) c[R] = s;
    }
    __syncthreads();
    for (int idx = tid; idx < (int)np; idx += DF_NT) {
        double s = 0.0;
        for (int R = 0; R < na; ++R) s += Bf[(size_t)R * np + idx] * c[R];
        int k, l;
        df_unpack(idx, k, l);
        J[k * n + l] = s; J[l * n + k] = s;
    }
}

// K = 2 sum_R W_R W_R^T, W_R = B_R C_occ.  grid = (R-chunks, fragment); K accumulates in registers
// over the chunk and is flushed with atomics (K is zeroed by the launcher).
template <int NV, bool BLDS>
__global__ void __launch_bounds__(DF_NT) df_k_kernel(BatchView bv, int only_active)
{
    extern __shared__ double lds[];
    const int f = blockIdx.y, tid = threadIdx.x, n = bv.n, na = bv.naux, o = bv.nocc;
    if (only_active && bv.istate[4 * f] == ST_DONE) return;
    const size_t np = (size_t)bv.npair;
    const double* __restrict__ Bf = bv.df_b + (size_t)f * na * np;
    const double* __restrict__ C = bv.C + (size_t)f * n * n;
    double* Bs = lds;                       // n x (n+1) unpacked B_R (BLDS only)
    double* Co = Bs + (BLDS ? (size_t)n * (n + 1) : 0);  // n x o occupied orbitals
    double* W = Co + (size_t)n * o;         // n x (o+1)
    const int ldb = n + 1, ldw = o + 1;
    for (int idx = tid; idx < n * o; idx += DF_NT) { const int r = idx / o, i = idx - r * o; Co[idx] = C[r * n + i]; }
    double acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.0;
    __syncthreads();
    for (int R = blockIdx.x; R < na; R += gridDim.x) {
        const double* __restrict__ row = Bf + (size_t)R * np;
        if (BLDS) {
            for (int idx = tid; idx < (int)np; idx += DF_NT) {
                int k, l;
                df_unpack(idx, k, l);
                const double v = row[idx];
                Bs[k * ldb + l] = v; Bs[l * ldb + k] = v;
            }
            __syncthreads();
        }
        for (int idx = tid; idx < n * o; idx += DF_NT) {
            const int mu = idx / o, i = idx - mu * o;
            double s = 0.0;
            if (BLDS) {
                for (int la = 0; la < n; ++la) s += Bs[mu * ldb + la] * Co[la * o + i];
            } else {
                // large fragments: the packed row is read straight from global memory (L2)
                const int mbase = mu * (mu + 1) / 2;
                for (int la = 0; la <= mu; ++la) s += row[mbase + la] * Co[la * o + i];
                for (int la = mu + 1; la < n; ++la) s += row[la * (la + 1) / 2 + mu] * Co[la * o + i];
            }
            W[mu * ldw + i] = s;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int idx = tid + DF_NT * k;
            if (idx < n * n) {
                const int mu = idx / n, nu = idx - mu * n;
                double s = 0.0;
                for (int i = 0; i < o; ++i) s += W[mu * ldw + i] * W[nu * ldw + i];
                acc[k] += s;
            }
        }
        __syncthreads();
    }
    double* K = bv.K + (size_t)f * n * n;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int idx = tid + DF_NT * k;
        if (idx < n * n && acc[k] != 0.0) atomicAdd(&K[idx], 2.0 * acc[k]);
    }
}

// ------------------------------------------------------------------------------------------
// J and K in ONE pass over the fitted tensor, contractions on the FP64 matrix cores (round 2).
//
// build_fock_df (mqc_libcint_rhf.f90:1576-1646): c_R = sum B_R D, J = sum_R B_R c_R, W_R = B_R C_occ,
// K = 2 sum_R W_R W_R^T.  SURVEY 8d: J needs 8 n^2 A bytes "if J and K share a pass" -- they do here: a workgroup
// walks auxiliary rows R of one fragment; row R (npair doubles, contiguous) is loaded ONCE from HBM into registers
// (the next row is in flight while the current one is consumed), parked in LDS, and used for
//   * c_R            (each wave reduces the row against the packed density in LDS),
//   * J += B_R c_R   (lane-private accumulators over the packed pairs the thread loaded),
//   * W_R = B_R C    as 16 x 16 MFMA jobs (A-fragments gathered from the packed row, B-fragments = C_occ in LDS),
//   * K += W_R W_R^T as MFMA jobs on the lower-triangle tiles, accumulators resident in the waves' registers
// across all rows of the workgroup; one atomic flush at the end.  J is finished by the second sweep c -> J inside
// the same loop because c_R is complete once its row has been read: no second pass over B.
constexpr int DJ_NW = 4;

__device__ __forceinline__ int dj_pidx(int a, int b) { return a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }

template <int JMAX, int NLD, bool WITH_K, bool DPG>
__global__ void __launch_bounds__(64 * DJ_NW) df_jk_mfma_kernel(BatchView bv, int only_active)
{
    extern __shared__ double lds[];
    const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (only_active && bv.istate[4 * f] == ST_DONE) return;
    const int lo = lane & 15, hi = lane >> 4;
    const int n = bv.n, na = bv.naux, o = bv.nocc;
    const int np = bv.npair;
    const int NT16 = (n + 15) >> 4, NP = NT16 << 4, OT = (o + 15) >> 4, OP = OT << 4, WS = OP + 1;
    const double* __restrict__ Bf = bv.df_b + (size_t)f * na * (size_t)np;
    const double* __restrict__ D = bv.D + (size_t)f * n * n;
    const double* __restrict__ C = bv.C + (size_t)f * n * n;
    // (round 3: the packed density no longer sits in LDS or in the SCF workspace -- each thread keeps the elements that
    // pair with the row elements it loads in registers, see dpk below; DPG is kept as a template parameter only so that
    // the instantiation list of round 2 stands)
    double* row = lds;                                // [np]  the current row of B
    double* Co = row + np;                            // [NP][OP] occupied orbitals, zero padded
    double* W = Co + (WITH_K ? (size_t)NP * OP : 0);  // [NP][WS]

    if (WITH_K) {
        for (int idx = tid; idx < NP * OP; idx += 64 * DJ_NW) {
            const int r = idx / OP, i = idx - r * OP;
            Co[idx] = (r < n && i < o) ? C[r * n + i] : 0.0;
        }
        for (int idx = tid; idx < NP * WS; idx += 64 * DJ_NW) W[idx] = 0.0;
    }
    double jacc[NLD], nxt[NLD], dpk[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) jacc[k] = 0.0;
    // this thread's packed-density elements (off-diagonal doubled): they pair with the row elements it loads, so c_R is
    // formed from registers -- five products, a wave reduction and four numbers through LDS -- instead of every wave
    // walking the whole row in LDS (18 steps of two LDS reads per lane and row)
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int idx = tid + 64 * DJ_NW * k;
        dpk[k] = 0.0;
        if (idx < np) {
            int a, b;
            df_unpack(idx, a, b);
            const double d = D[a * n + b];
            dpk[k] = a == b ? d : 2.0 * d;
        }
    }
    __shared__ double cpart[2][DJ_NW];
    int cbuf = 0;
    v4f64 kacc[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) kacc[j] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const int ntile = NT16 * (NT16 + 1) / 2;
    // W_R = B_R C_occ gathers its A operand from the PACKED row: element (mu, la) sits at dj_pidx(mu, la).  The offsets
    // do not depend on the row: those of the wave's first job are formed once (a job per wave covers n <= 64 with one
    // orbital tile); further jobs compute them on the fly
    constexpr int WOFF = 16;                        // k-steps covered by the cached offsets (NP <= 64)
    int woff[WOFF];
    {
        const int job = wave, mt = job / OT, mu = 16 * mt + lo;
#pragma unroll
        for (int ks = 0; ks < WOFF; ++ks) {
            const int la = 4 * ks + hi;
            woff[ks] = (job < NT16 * OT && ks < (NP >> 2) && mu < n && la < n) ? dj_pidx(mu, la) : -1;
        }
    }

    int R = blockIdx.x;
    if (R < na) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) { const int idx = tid + 64 * DJ_NW * k; nxt[k] = idx < np ? Bf[(size_t)R * np + idx] : 0.0; }
    }
    __syncthreads();
    for (; R < na; R += gridDim.x) {
        double cur[NLD];
        double cp = 0.0;
#pragma unroll
        for (int k = 0; k < NLD; ++k) { cur[k] = nxt[k]; const int idx = tid + 64 * DJ_NW * k; if (idx < np) row[idx] = cur[k]; cp += cur[k] * dpk[k]; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cp += __shfl_xor(cp, off, 64);
        if (lane == 0) cpart[cbuf][wave] = cp;
        const int Rn = R + gridDim.x;
        if (Rn < na) {
#pragma unroll
            for (int k = 0; k < NLD; ++k) { const int idx = tid + 64 * DJ_NW * k; nxt[k] = idx < np ? Bf[(size_t)Rn * np + idx] : 0.0; }
        }
        __syncthreads();
        // c_R = sum of the waves' partial dot products (written before the barrier above; the buffer alternates so that
        // the next row's partials do not overwrite values a slower wave still reads), then J += B_R c_R on the thread's pairs
        double c = 0.0;
#pragma unroll
        for (int w = 0; w < DJ_NW; ++w) c += cpart[cbuf][w];
        cbuf ^= 1;
#pragma unroll
        for (int k = 0; k < NLD; ++k) jacc[k] += cur[k] * c;
        if (WITH_K) {
            // W_R = B_R C_occ: jobs (row tile, orbital tile)
            for (int job = wave; job < NT16 * OT; job += DJ_NW) {
                const int mt = job / OT, ot = job - mt * OT;
                const int mu = 16 * mt + lo;
                v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
                if (job == wave && (NP >> 2) <= WOFF) {
#pragma unroll
                    for (int ks = 0; ks < WOFF; ++ks) {
                        if (ks < (NP >> 2)) {
                            const double a = woff[ks] >= 0 ? row[woff[ks]] : 0.0;
                            const double b = Co[(4 * ks + hi) * OP + 16 * ot + lo];
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                        }
                    }
                } else
                for (int ks = 0; ks < (NP >> 2); ++ks) {
                    const int la = 4 * ks + hi;
                    const double a = (mu < n && la < n) ? row[dj_pidx(mu, la)] : 0.0;
                    const double b = Co[la * OP + 16 * ot + lo];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) W[(16 * mt + hi + 4 * r) * WS + 16 * ot + lo] = acc[r];
            }
            __syncthreads();
            // K += W W^T on the lower-triangle tiles, dealt round-robin
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                const int t = wave + DJ_NW * j;
                if (t < ntile) {
                    int mt = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
                    while ((mt + 1) * (mt + 2) / 2 <= t) ++mt;
                    while (mt * (mt + 1) / 2 > t) --mt;
                    const int nt = t - mt * (mt + 1) / 2;
                    const double* __restrict__ ar = W + (size_t)(16 * mt + lo) * WS + hi;
                    const double* __restrict__ br = W + (size_t)(16 * nt + lo) * WS + hi;
                    for (int ks = 0; ks < (OP >> 2); ++ks) kacc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[4 * ks], br[4 * ks], kacc[j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // flush J (packed pairs of this thread) and K (this wave's tiles)
    double* J = bv.J + (size_t)f * n * n;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int idx = tid + 64 * DJ_NW * k;
        if (idx < np && jacc[k] != 0.0) {
            int a, b;
            df_unpack(idx, a, b);
            atomicAdd(&J[a * n + b], jacc[k]);
            if (a != b) atomicAdd(&J[b * n + a], jacc[k]);
        }
    }
    if (WITH_K) {
        double* K = bv.K + (size_t)f * n * n;
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int t = wave + DJ_NW * j;
            if (t < ntile) {
                int mt = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
                while ((mt + 1) * (mt + 2) / 2 <= t) ++mt;
                while (mt * (mt + 1) / 2 > t) --mt;
                const int nt = t - mt * (mt + 1) / 2;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mu = 16 * mt + hi + 4 * r, nu = 16 * nt + lo;
                    const double v = 2.0 * kacc[j][r];
                    if (mu < n && nu < n && v != 0.0) {
                        if (mt != nt) { atomicAdd(&K[mu * n + nu], v); atomicAdd(&K[nu * n + mu], v); }
                        else atomicAdd(&K[mu * n + nu], v);
                    }
                }
            }
        }
    }
}

template <int JMAX, int NLD, bool WITH_K, bool DPG>
static void df_jk_mfma_launch(const BatchView& bv, int oa, hipStream_t s)
{
    const int np16 = ((bv.n + 15) / 16) * 16, op = ((bv.nocc + 15) / 16) * 16;
    const size_t lds = sizeof(double) * ((size_t)bv.npair + (WITH_K ? (size_t)np16 * op + (size_t)np16 * (op + 1) : 0) + 8);
    auto kern = df_jk_mfma_kernel<JMAX, NLD, WITH_K, DPG>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int gx = (3072 + bv.nfrag - 1) / bv.nfrag;
    if (gx > bv.naux) gx = bv.naux;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(kern, dim3(gx, bv.nfrag), dim3(64 * DJ_NW), lds, s, bv, oa);
}

// ------------------------------------------------------------------------------------------
// Round 3: the same pass, but every WAVE walks its own auxiliary rows from start to end -- no workgroup barrier inside
// the row loop.  The kernel above synchronises the workgroup three times per row (row parked | W complete | K done)
// and its 60 MFMAs per row are cut into 3 + 6 jobs for 4 waves; 22 % of the matrix-core cycles were busy and the J/K
// stage read its tensor at 0.14 of the HBM peak (profiles/r03_pmc_summary.json).  Here a wave owns rows R = wave,
// wave + 4 G, ...: it parks its row in a wave-private LDS buffer (the gather of the W operands needs random access to
// the packed row), forms c_R by a wave reduction against the packed density (shared, read-only LDS), updates its J
// accumulators, runs ALL W = B_R C_occ jobs with interleaved accumulators, writes W to a wave-private buffer and runs
// ALL lower-triangle K tiles with interleaved accumulators; the only workgroup barriers are the one after the
// prologue and the ones of the final reduction of the four waves' J and K into one flush.  n <= 48, occupied
// orbitals <= 16 (one orbital tile): the MBE fragment sizes; everything else keeps the kernel above.
template <int NTC, int NLW>
__global__ void __launch_bounds__(64 * DJ_NW, 2) df_jk_wave_kernel(BatchView bv, int only_active)
{
    extern __shared__ double lds[];
    const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (only_active && bv.istate[4 * f] == ST_DONE) return;
    const int lo = lane & 15, hi = lane >> 4;
    constexpr int NP = 16 * NTC, OP = 16, WS = OP + 1, KS = NP / 4, NTILE = NTC * (NTC + 1) / 2;
    const int n = bv.n, na = bv.naux, o = bv.nocc, np = bv.npair;
    const double* __restrict__ Bf = bv.df_b + (size_t)f * na * (size_t)np;
    const double* __restrict__ D = bv.D + (size_t)f * n * n;
    const double* __restrict__ C = bv.C + (size_t)f * n * n;
    double* Dp = lds;                                   // [np]  packed density, off-diagonal doubled (shared)
    double* Co = Dp + np;                               // [NP][OP] occupied orbitals, zero padded (shared)
    double* rowbuf = Co + NP * OP + (size_t)wave * (np + NP * WS);   // this wave's row [np] ...
    double* W = rowbuf + np;                            // ... and its W [NP][WS]
    for (int idx = tid; idx < np; idx += 64 * DJ_NW) {
        int k, l;
        df_unpack(idx, k, l);
        const double d = D[k * n + l];
        Dp[idx] = k == l ? d : 2.0 * d;
    }
    for (int idx = tid; idx < NP * OP; idx += 64 * DJ_NW) {
        const int r = idx / OP, i = idx - r * OP;
        Co[idx] = (r < n && i < o) ? C[r * n + i] : 0.0;
    }
    // gather offsets of the W operands (row-invariant): element (mu = 16 mt + lo, la = 4 ks + hi) of the packed row
    // (two 16-bit offsets per register: npair <= 1176; 0xffff = outside the matrix)
    unsigned woff[NTC][KS / 2];
#pragma unroll
    for (int mt = 0; mt < NTC; ++mt)
#pragma unroll
        for (int kp = 0; kp < KS / 2; ++kp) {
            const int mu = 16 * mt + lo, la0 = 4 * (2 * kp) + hi, la1 = 4 * (2 * kp + 1) + hi;
            const unsigned o0 = (mu < n && la0 < n) ? (unsigned)dj_pidx(mu, la0) : 0xffffu;
            const unsigned o1 = (mu < n && la1 < n) ? (unsigned)dj_pidx(mu, la1) : 0xffffu;
            woff[mt][kp] = o0 | (o1 << 16);
        }
    double jacc[NLW], nxt[NLW];
#pragma unroll
    for (int k = 0; k < NLW; ++k) jacc[k] = 0.0;
    v4f64 kacc[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) kacc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const int stride = (int)gridDim.x * DJ_NW;
    int R = (int)blockIdx.x * DJ_NW + wave;
    if (R < na) {
#pragma unroll
        for (int k = 0; k < NLW; ++k) { const int idx = lane + 64 * k; nxt[k] = idx < np ? Bf[(size_t)R * np + idx] : 0.0; }
    }
    __syncthreads();                                    // Dp, Co complete
    for (; R < na; R += stride) {
        // ---- the row: registers -> wave-private LDS; c_R; the next row goes in flight
        double c = 0.0;
#pragma unroll
        for (int k = 0; k < NLW; ++k) {
            const int idx = lane + 64 * k;
            if (idx < np) { rowbuf[idx] = nxt[k]; c += nxt[k] * Dp[idx]; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
#pragma unroll
        for (int k = 0; k < NLW; ++k) jacc[k] += nxt[k] * c;
        const int Rn = R + stride;
        if (Rn < na) {
#pragma unroll
            for (int k = 0; k < NLW; ++k) { const int idx = lane + 64 * k; nxt[k] = idx < np ? Bf[(size_t)Rn * np + idx] : 0.0; }
        }
        // ---- W_R = B_R C_occ: NTC row tiles, accumulators interleaved over the k-steps (wave-private data: the LDS
        // writes above are ordered before these reads by the wave's own lgkmcnt)
        v4f64 wacc[NTC];
#pragma unroll
        for (int mt = 0; mt < NTC; ++mt) wacc[mt] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const double cob = Co[(4 * ks + hi) * OP + lo];      // B operand of the step's jobs: C_occ row 4 ks + hi, orbital lo
#pragma unroll
            for (int mt = 0; mt < NTC; ++mt) {
                const unsigned off = (ks & 1) ? (woff[mt][ks >> 1] >> 16) : (woff[mt][ks >> 1] & 0xffffu);
                const double a = off != 0xffffu ? rowbuf[off] : 0.0;
                wacc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, cob, wacc[mt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int mt = 0; mt < NTC; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) W[(16 * mt + hi + 4 * r) * WS + lo] = wacc[mt][r];
        // ---- K += W W^T on the lower-triangle tiles
        double wa[NTC][4];
#pragma unroll
        for (int mt = 0; mt < NTC; ++mt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wa[mt][ks] = W[(16 * mt + lo) * WS + 4 * ks + hi];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            int t = 0;
#pragma unroll
            for (int mt = 0; mt < NTC; ++mt)
#pragma unroll
                for (int nt = 0; nt <= mt; ++nt, ++t) kacc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[mt][ks], wa[nt][ks], kacc[t], 0, 0, 0);
        }
    }
    // ---- the four waves' J and K meet in LDS (the row / W buffers are free now), one flush per workgroup
    __syncthreads();
    double* Jsum = Co + NP * OP;                        // [np] over the wave buffers
    double* Ksum = Jsum + np;                           // [NP][NP]
    for (int idx = tid; idx < np + NP * NP; idx += 64 * DJ_NW) Jsum[idx] = 0.0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NLW; ++k) { const int idx = lane + 64 * k; if (idx < np && jacc[k] != 0.0) atomicAdd(&Jsum[idx], jacc[k]); }
    {
        int t = 0;
#pragma unroll
        for (int mt = 0; mt < NTC; ++mt)
#pragma unroll
            for (int nt = 0; nt <= mt; ++nt, ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = kacc[t][r];
                    if (v != 0.0) atomicAdd(&Ksum[(16 * mt + hi + 4 * r) * NP + 16 * nt + lo], v);
                }
    }
    __syncthreads();
    double* J = bv.J + (size_t)f * n * n;
    for (int idx = tid; idx < np; idx += 64 * DJ_NW) {
        const double v = Jsum[idx];
        if (v != 0.0) {
            int a, b;
            df_unpack(idx, a, b);
            atomicAdd(&J[a * n + b], v);
            if (a != b) atomicAdd(&J[b * n + a], v);
        }
    }
    double* K = bv.K + (size_t)f * n * n;
    for (int idx = tid; idx < NP * NP; idx += 64 * DJ_NW) {
        const int mu = idx / NP, nu = idx - mu * NP;
        // tiles (mt, nt) with mt >= nt were formed: a diagonal tile holds its whole 16 x 16 block, an off-diagonal one
        // stands for its transpose as well
        if (mu < n && nu < n && (mu >> 4) >= (nu >> 4)) {
            const double v = 2.0 * Ksum[idx];
            if (v != 0.0) {
                atomicAdd(&K[mu * n + nu], v);
                if ((mu >> 4) != (nu >> 4)) atomicAdd(&K[nu * n + mu], v);
            }
        }
    }
}

template <int NTC, int NLW>
static bool df_jk_wave_launch(const BatchView& bv, int oa, hipStream_t s)
{
    const int NP = 16 * NTC;
    const size_t lds = sizeof(double) * ((size_t)bv.npair + (size_t)NP * 16 + DJ_NW * ((size_t)bv.npair + (size_t)NP * 17) + 8);
    if (lds > (size_t)80 * 1024) return false;          // two workgroups per CU
    if ((size_t)bv.npair + (size_t)NP * NP > DJ_NW * ((size_t)bv.npair + (size_t)NP * 17)) return false;   // final reduction fits the wave buffers
    auto kern = df_jk_wave_kernel<NTC, NLW>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int gx = (2048 + bv.nfrag - 1) / bv.nfrag;           // workgroups per fragment: every wave should see several rows
    const int maxg = (bv.naux + 4 * DJ_NW - 1) / (4 * DJ_NW);
    if (gx > maxg) gx = maxg;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(kern, dim3(gx, bv.nfrag), dim3(64 * DJ_NW), lds, s, bv, oa);
    return true;
}

// n <= 48, one orbital tile, exchange wanted (MQC_HIP_DF_WAVE=0: the workgroup kernel)
static bool df_jk_wave_dispatch(const BatchView& bv, int oa, hipStream_t s)
{
    static const bool on = [] { const char* e = std::getenv("MQC_HIP_DF_WAVE"); return !(e && e[0] == '0'); }();
    if (!on || bv.exx == 0.0 || bv.n > 48 || bv.nocc > 16) return false;
    const int nt = (bv.n + 15) / 16, nlw = (bv.npair + 63) / 64;
    if (nt == 1 && nlw <= 3) return df_jk_wave_launch<1, 3>(bv, oa, s);
    if (nt == 2 && nlw <= 9) return df_jk_wave_launch<2, 9>(bv, oa, s);
    if (nt == 3 && nlw <= 19) return df_jk_wave_launch<3, 19>(bv, oa, s);
    return false;
}

// false: no instantiation covers this size (the round-1 kernels take it)
static bool df_jk_mfma_dispatch(const BatchView& bv, int oa, hipStream_t s)
{
    const int nt = (bv.n + 15) / 16, ntile = nt * (nt + 1) / 2, jobs = (ntile + DJ_NW - 1) / DJ_NW;
    const int nld = (bv.npair + 64 * DJ_NW - 1) / (64 * DJ_NW);
    const bool k = bv.exx != 0.0;
    const int np16 = nt * 16, op = ((bv.nocc + 15) / 16) * 16;
    const size_t body = (k ? (size_t)np16 * op + (size_t)np16 * (op + 1) : 0) + 8;
    const bool dpg = false;      // round 2: packed density in global memory for large fragments; unused since the density sits in registers
    if (sizeof(double) * ((size_t)bv.npair + body) > 150 * 1024) return false;
#define DJ(JM, NL)                                                                                        \
    do {                                                                                                  \
        if (k) { if (dpg) df_jk_mfma_launch<JM, NL, true, true>(bv, oa, s); else df_jk_mfma_launch<JM, NL, true, false>(bv, oa, s); } \
        else { if (dpg) df_jk_mfma_launch<1, NL, false, true>(bv, oa, s); else df_jk_mfma_launch<1, NL, false, false>(bv, oa, s); }   \
        return true;                                                                                      \
    } while (0)
    if (jobs <= 1 && nld <= 2) DJ(1, 2);            // n <= 16..31
    if (jobs <= 2 && nld <= 5) DJ(2, 5);            // n <= 48 (npair 1176)
    if (jobs <= 3 && nld <= 9) DJ(3, 9);            // n <= 64
    if (jobs <= 6 && nld <= 19) DJ(6, 19);          // n <= 96
    if (jobs <= 9 && nld <= 33) DJ(9, 33);          // n <= 128
#undef DJ
    return false;
}

// ------------------------------------------------------------------------------------------
template <int LA, int LB, int LC>
static void df3c_launch(const BatchView& bv, const std::vector<int>& t, int* d, hipStream_t s)
{
    const int nt = (int)t.size() / 3;
    if (!nt) return;
    (void)hipMemcpyAsync(d, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice, s);
    const long total = (long)nt * bv.nfrag;
    hipLaunchKernelGGL((df3c_kernel<LA, LB, LC>), dim3((int)((total + 63) / 64)), dim3(64), 0, s, bv, d, nt);
}

template <int LP, int LQ>
static void df2c_launch(const BatchView& bv, const std::vector<int>& t, int* d, hipStream_t s)
{
    const int nt = (int)t.size() / 2;
    if (!nt) return;
    (void)hipMemcpyAsync(d, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice, s);
    const long total = (long)nt * bv.nfrag;
    hipLaunchKernelGGL((df2c_kernel<LP, LQ>), dim3((int)((total + 63) / 64)), dim3(64), 0, s, bv, d, nt);
}

void launch_df_build(const BatchView& bv, const Topology& topo, const Topology& aux, hipStream_t s)
{
    static DevicePool lists_slot[2];
    DevicePool& lists = lists_slot[bv.slot & 1];
    const int ns = (int)topo.shells.size(), nx = (int)aux.shells.size();
    // bucket tasks
    std::vector<int> t3[3][3][4], t2[4][4];
    std::vector<int> t3f[4][4];              // an f shell in the bra: (l_b, l_P) -> entries (A, B, P, -) for the general kernel
    for (int A = 0; A < ns; ++A)
        for (int B = 0; B <= A; ++B) {
            int a = A, b = B;
            if (topo.shells[a].l < topo.shells[b].l) std::swap(a, b);
            for (int P = 0; P < nx; ++P) {
                if (topo.shells[a].l == 3) {
                    auto& v = t3f[topo.shells[b].l][aux.shells[P].l];
                    v.push_back(a); v.push_back(b); v.push_back(P); v.push_back(0);
                    continue;
                }
                auto& v = t3[topo.shells[a].l][topo.shells[b].l][aux.shells[P].l];
                v.push_back(a); v.push_back(b); v.push_back(P);
            }
        }
    for (int P = 0; P < nx; ++P)
        for (int Q = 0; Q <= P; ++Q) {
            int p = P, q = Q;
            if (aux.shells[p].l < aux.shells[q].l) std::swap(p, q);
            auto& v = t2[aux.shells[p].l][aux.shells[q].l];
            v.push_back(p); v.push_back(q);
        }
    size_t tot = 0;
    for (auto& x : t3) for (auto& y : x) for (auto& z : y) tot += z.size();
    for (auto& x : t2) for (auto& y : x) tot += y.size();
    for (auto& x : t3f) for (auto& y : x) tot += y.size();
    int* d = (int*)lists.ensure((tot + 64) * sizeof(int));
    size_t off = 0;
    for (int lb = 0; lb < 4; ++lb)
        for (int lp = 0; lp < 4; ++lp) {
            auto& v = t3f[lb][lp];
            if (v.empty()) continue;
            (void)hipMemcpyAsync(d + off, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, s);
            if (!launch_df3c_general(bv, 3, lb, lp, d + off, (int)(v.size() / 4), s))
                std::fprintf(stderr, "mqc_hip: three-centre class (f %d | %d) does not fit the general kernel\n", lb, lp);
            off += v.size();
        }
#define DF3(a, b, c) df3c_launch<a, b, c>(bv, t3[a][b][c], d + off, s); off += t3[a][b][c].size();
#define DF3_ALLC(a, b) DF3(a, b, 0) DF3(a, b, 1) DF3(a, b, 2) DF3(a, b, 3)
    DF3_ALLC(0, 0) DF3_ALLC(1, 0) DF3_ALLC(1, 1) DF3_ALLC(2, 0) DF3_ALLC(2, 1) DF3_ALLC(2, 2)
#undef DF3_ALLC
#undef DF3
#define DF2(p, q) df2c_launch<p, q>(bv, t2[p][q], d + off, s); off += t2[p][q].size();
    DF2(0, 0) DF2(1, 0) DF2(1, 1) DF2(2, 0) DF2(2, 1) DF2(2, 2) DF2(3, 0) DF2(3, 1) DF2(3, 2) DF2(3, 3)
#undef DF2
    {
        static const bool v1 = [] { const char* e = std::getenv("MQC_HIP_DF_CHOL_V1"); return e && e[0] == '1'; }();
        const size_t npad = (size_t)((bv.naux + 15) / 16) * 16;
        const size_t lds_m = sizeof(double) * (npad * 16 + 2 * 16 * 17 + 8);
        if (v1 || lds_m > 150 * 1024) {
            hipLaunchKernelGGL(df_cholesky_kernel, dim3(bv.nfrag), dim3(DF_NT), sizeof(double) * (size_t)(bv.naux + 8), s, bv);
        } else {
            (void)hipFuncSetAttribute((const void*)df_cholesky_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
            hipLaunchKernelGGL(df_cholesky_mfma_kernel, dim3(bv.nfrag), dim3(DF_NT), lds_m, s, bv);
        }
    }
    hipLaunchKernelGGL(df_metric_eig_kernel, dim3(bv.nfrag), dim3(DF_NT), 0, s, bv);      // flagged fragments only
    {
        const int jobs = ((bv.naux + 15) / 16) * ((bv.npair + 15) / 16);
        int gx = (jobs + DF_NT / 64 - 1) / (DF_NT / 64);
        if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(df_fit_mfma_kernel, dim3(gx, bv.nfrag), dim3(DF_NT), 0, s, bv);
    }
}

size_t df_k_lds_bytes(int n, int o, bool blds) { return sizeof(double) * ((blds ? (size_t)n * (n + 1) : 0) + (size_t)n * o + (size_t)n * (o + 1) + 8); }

void launch_df_jk(const BatchView& bv, bool only_active, hipStream_t s)
{
    const int n = bv.n, oa = only_active ? 1 : 0;
    (void)hipMemsetAsync(bv.K, 0, sizeof(double) * (size_t)bv.nfrag * n * n, s);
    // MQC_HIP_DF_V1=1: the round-1 kernels (J as one workgroup per fragment with two passes over B, K on the vector units)
    static const bool v1 = [] { const char* e = std::getenv("MQC_HIP_DF_V1"); return e && e[0] == '1'; }();
    if (!v1) {
        (void)hipMemsetAsync(bv.J, 0, sizeof(double) * (size_t)bv.nfrag * n * n, s);
        if (df_jk_wave_dispatch(bv, oa, s)) return;
        if (df_jk_mfma_dispatch(bv, oa, s)) return;
    }
    const size_t ldsj = sizeof(double) * ((size_t)bv.npair + bv.naux + 8);
    (void)hipFuncSetAttribute((const void*)df_j_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsj);
    hipLaunchKernelGGL(df_j_kernel, dim3(bv.nfrag), dim3(DF_NT), ldsj, s, bv, oa);
    const bool blds = df_k_lds_bytes(n, bv.nocc, true) <= 150 * 1024;
    const size_t ldsk = df_k_lds_bytes(n, bv.nocc, blds);
    int gx = (2048 + bv.nfrag - 1) / bv.nfrag;
    if (gx > bv.naux) gx = bv.naux;
    if (gx < 1) gx = 1;
    const int nv = (n * n + DF_NT - 1) / DF_NT;
#define DFK(NVV, BL)                                                                                 \
    do {                                                                                             \
        (void)hipFuncSetAttribute((const void*)df_k_kernel<NVV, BL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsk); \
        hipLaunchKernelGGL((df_k_kernel<NVV, BL>), dim3(gx, bv.nfrag), dim3(DF_NT), ldsk, s, bv, oa); \
    } while (0)
    if (nv <= 10) DFK(10, true); else if (nv <= 29) DFK(29, true); else if (blds) DFK(54, true); else DFK(77, false);
#undef DFK
}


__global__ void scale_kernel(double* __restrict__ p, size_t count, double f)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] *= f;
}

// p[0 .. count) *= f   (unrestricted density fitting: the exchange kernels return 2 W W^T, the closed-shell convention)
void launch_scale(double* p, size_t count, double f, hipStream_t s)
{
    if (count) hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, p, count, f);
}

}  // namespace mqc
