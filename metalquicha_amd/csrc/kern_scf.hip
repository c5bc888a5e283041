// kern_scf.hip -- the linear algebra of one SCF iteration, one workgroup per fragment.
//
// What the cuEST loop does with ~25 cuBLAS/cuSOLVER launches and 3 blocking scalar
// fetches per iteration per fragment (backends/cuest/backend/mqc_cuest_scf.f90:444-553,
// mqc_cuest_integrals.f90:1841-1947, mqc_diis_device.f90:122-222), this file does in ONE
// launch for the whole batch: each 256-thread workgroup owns a fragment and performs
//   F = H + J - (exx/2) K,  E = 1/2 sum D (H + F)          (assemble_fock + matrix_dot)
//   e = X^T (F D S - S D F) X                              (commutator_device)
//   DIIS push / coefficients / extrapolation               (diis_device_t, mqc_diis.f90)
//   F' = X^T F X, eigen-decomposition, C = X C', D = 2 C_o C_o^T   (diagonalize_fock_device)
//   dE, rms dD, convergence state machine                  (mqc_libcint_rhf.f90:626-662)
// with the reference CPU path's semantics (dE and rms(dD) test for iteration > 1, then a
// final full rebuild of F and E from the converged density).
//
// The eigen-solver is a cyclic parallel Jacobi held in LDS (the matrices are 24..116 wide:
// a LAPACK-style tridiagonalisation would be launch- and latency-bound here, whereas one CU
// with the matrix in its 160 KB LDS does n/2 independent rotations per step).
#include "engine.hpp"

// This file is compiled twice: with 256 threads per fragment (the throughput build: eight or more fragments per CU) and,
// from kern_scf_wide.hip, with 512 (the latency build for batches that leave CUs idle -- one rank's share of an 8-way
// split, single-fragment calls -- where a rotation set's update loops are the critical path).  Everything device-side
// has internal linkage so that the two builds do not meet at link time.
#ifndef MQC_SCF_NT
#define MQC_SCF_NT 256
#endif

namespace mqc {
namespace {

constexpr int NT = MQC_SCF_NT;
// In-kernel phase stamps of the restricted SCF step (build with -DSCF_STAMPS=1; never in the shipped library)
#ifndef SCF_STAMPS
#define SCF_STAMPS 0
#endif
#if SCF_STAMPS
__device__ unsigned long long g_scf_stamps[8];
#define SCF_ST_DECL unsigned long long sst_t = __builtin_amdgcn_s_memtime();
#define SCF_ST(k) { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_scf_stamps[k], t_ - sst_t); } sst_t = __builtin_amdgcn_s_memtime(); }
#else
#define SCF_ST_DECL
#define SCF_ST(k)
#endif
constexpr double OVERLAP_EIG_TOL = 1.0e-7;   // src/scf/mqc_scf_common.f90:33
constexpr double GWH_K = 1.75;               // src/scf/mqc_scf_common.f90:39
constexpr double PIVOT_FLOOR = 1.0e-14;      // src/methods/mqc_diis.f90:30
constexpr double JACOBI_SKIP = 1.0e-13;      // rotation skipped when |a_pq| <= this * max|a_ii| (1e-15: one more sweep in ~1 of 5 diagonalisations, SCF step 19.0 -> 17.6 ms per evaluation at 1e-13 with unchanged iteration counts and 7e-12 Eh parity; 1e-12 gains nothing more and quadruples the parity error)
constexpr int JACOBI_MAX_SWEEPS = 60;

__device__ __forceinline__ double block_sum(double v, double* red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) t += red[k];
    return t;
}

__device__ __forceinline__ double block_max(double v, double* red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double t = red[0];
#pragma unroll
    for (int k = 1; k < NT / 64; ++k) t = fmax(t, red[k]);
    return t;
}

// C[M x N] (ldc) = op(A) * B;  A is [M x K] (lda) or, if TA, [K x M];  B is [K x N] (ldb).
template <bool TA>
__device__ void wg_gemm(int M, int N, int K, const double* __restrict__ A, int lda,
                        const double* __restrict__ B, int ldb, double* __restrict__ C, int ldc)
{
    for (int idx = threadIdx.x; idx < M * N; idx += NT) {
        const int i = idx / N, j = idx - i * N;
        double s = 0.0;
        if (TA) {
            for (int k = 0; k < K; ++k) s += A[k * lda + i] * B[k * ldb + j];
        } else {
            for (int k = 0; k < K; ++k) s += A[i * lda + k] * B[k * ldb + j];
        }
        C[i * ldc + j] = s;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------
// Workgroup GEMM on the FP64 matrix cores: C = op(A) op(B), one 16 x 16 output tile per wave at a time
// (v_mfma_f64_16x16x4_f64; operand layout of the guide: A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15], C[row = (lane>>4) + 4 r][col = lane&15]).  Operands are read straight from
// where they live (global/L2 or LDS): every lane fetches one element per operand and k-step, sixteen k at a
// time so that eight loads are in flight before the four MFMAs that consume them.  Edges are zero-filled.
//   TA: A is stored K x M (A[k*lda + i]) instead of M x K;  TB: B is stored N x K (B[j*ldb + k]).
//   store(i, j, value) receives every element of the M x N result once.
// The scalar loops this replaces spent ~17 VALU instructions per multiply-add on addressing.
typedef double v4f64_t __attribute__((ext_vector_type(4)));

template <bool TA, bool TB, class Store>
__device__ __forceinline__ void wg_gemm_mfma(int M, int N, int K, const double* A, int lda, const double* B, int ldb, Store store)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lo = lane & 15, hi = lane >> 4;
    const int mt = (M + 15) >> 4, nt = (N + 15) >> 4;
    for (int t = wave; t < mt * nt; t += NT / 64) {
        const int ti = t / nt, tj = t - ti * nt;
        const int i = 16 * ti + lo, j = 16 * tj + lo;
        const bool iok = i < M, jok = j < N;
        const int ic = iok ? i : 0, jc = jok ? j : 0;
        v4f64_t acc = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < K; k0 += 16) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 4 * u + hi;
                const bool kok = k < K;
                const int kc = kok ? k : 0;
                const double av = TA ? A[kc * lda + ic] : A[ic * lda + kc];
                const double bv_ = TB ? B[jc * ldb + kc] : B[kc * ldb + jc];
                a[u] = (iok && kok) ? av : 0.0;
                b[u] = (jok && kok) ? bv_ : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * ti + hi + 4 * r, col = 16 * tj + lo;
            if (row < M && col < N) store(row, col, acc[r]);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ double jacobi_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double jacobi_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    y = fma(y, fma(-hx * y, y, 0.5), y);
    return y;
}

// 1: rotation sets in which no pair exceeds the threshold skip their update phases (MQC_HIP_JACOBI_SKIP=0 turns it off)
__device__ int g_jacobi_skip_idle_sets = 1;
// measurement switch (MQC_HIP_JACOBI_MAX_SWEEPS): caps the sweeps; wrong eigenvectors below ~6
__device__ int g_jacobi_max_sweeps = 1000;

struct JacobiLds {
    double* A;      // mp x lda
    double* V;      // mp x lda (LDS) or global m x ldv
    double* rc;     // mp/2
    double* rs;     // mp/2
    int* rp;        // mp/2
    int* rq;        // mp/2
    double* red;    // 8
    int* flag;      // 4: sweep flag + three rotating per-set flags
    int* rflag;     // mp: one flag per rotation set of the sweep (deferred eigenvectors, mode 1)
    int lda;
    // mode 1 with deferred eigenvectors (see jacobi_eig): global scratch for the matrix while V sits in LDS (m * m) and
    // for the (c, s) of one sweep's rotations ((mp - 1) * mp doubles); null = V rotated in global memory set by set
    double* spill = nullptr;
    double* rlog = nullptr;
};

__host__ __device__ inline int even_up(int m) { return (m + 1) & ~1; }

__host__ __device__ inline size_t jacobi_lds_doubles(int m, bool v_in_lds, bool a_in_lds = true)
{
    const int mp = even_up(m), lda = mp + 1;
    size_t d = a_in_lds ? (size_t)mp * lda : 0;   // A
    if (v_in_lds) d += (size_t)mp * lda;     // V
    d += 2 * (mp / 2);                       // rc, rs
    d += mp / 2 + 1;                         // rp, rq as ints packed in doubles
    d += 8 + 2;                              // red, flag
    d += mp / 2 + 1;                         // rflag (mp ints)
    return d;
}

// mode 2: A and V in LDS; 1: A in LDS, V in global memory; 0: both in global memory (a_global: (m+2)^2 doubles of the
// fragment's workspace) -- matrices too wide for the CU's 160 KB, rotated in the L2 instead
__device__ inline JacobiLds carve_jacobi(double* lds, int m, int mode, double* v_global, double* a_global = nullptr)
{
    JacobiLds j;
    const int mp = even_up(m);
    const bool v_in_lds = mode == 2;
    j.lda = mp + 1;
    if (mode == 0) j.A = a_global;
    else { j.A = lds; lds += (size_t)mp * j.lda; }
    if (v_in_lds) { j.V = lds; lds += (size_t)mp * j.lda; } else { j.V = v_global; }
    j.rc = lds; lds += mp / 2;
    j.rs = lds; lds += mp / 2;
    j.rp = (int*)lds; j.rq = j.rp + mp / 2; lds += mp / 2 + 1;
    j.red = lds; lds += 8;
    j.flag = (int*)lds; lds += 2;
    j.rflag = (int*)lds;
    return j;
}

// Symmetric eigen-decomposition of the m x m matrix already stored in jl.A (padded to even mp
// with a zero row/column).  On exit jl.A's diagonal holds the eigenvalues and V's columns the
// eigenvectors (V addressed as V[row * ldv + col]).
template <int JM>
__device__ void jacobi_eig(JacobiLds& jl, int m, int ldv, const double* __restrict__ v0 = nullptr, int ldv0 = 0)
{
    constexpr bool VLDS = JM == 2;      // JM: 2 = A and V in LDS, 1 = A in LDS and V in global memory, 0 = both global (n > 140)
    const int mp = even_up(m), lda = jl.lda, half = mp / 2;
    const int tid = threadIdx.x;
    double* A = jl.A;
    double* V = jl.V;
    // V = I, or the basis the matrix was pre-rotated into (warm start)
    for (int idx = tid; idx < mp * mp; idx += NT) {
        const int r = idx / mp, c = idx - r * mp;
        if (VLDS || (r < m && c < m)) {
            double v = (r == c) ? 1.0 : 0.0;
            if (v0 != nullptr && r < m && c < m) v = v0[r * ldv0 + c];
            V[r * ldv + c] = v;
        }
    }
    double dmax = 0.0;
    for (int i = tid; i < m; i += NT) dmax = fmax(dmax, fabs(A[i * lda + i]));
    double omax = 0.0;
    for (int idx = tid; idx < m * m; idx += NT) {
        const int r = idx / m, c = idx - r * m;
        if (r != c) omax = fmax(omax, fabs(A[r * lda + c]));
    }
    dmax = block_max(fmax(dmax, omax), jl.red);
    const double thresh = JACOBI_SKIP * dmax;
    __syncthreads();

    const int max_sweeps = min(JACOBI_MAX_SWEEPS, g_jacobi_max_sweeps);
    // (k, col) of this thread's first work item and the step between its items: no division inside the sweeps
    const int k0 = tid / mp, c0 = tid - k0 * mp, dk = NT / mp, dc = NT - dk * mp;
    const int kb0 = tid / half, lb0 = tid - kb0 * half, dkb = NT / half, dlb = NT - dkb * half;      // same for the 2 x 2 blocks
    if constexpr (JM == 1) {
        if (jl.rlog != nullptr && jl.spill != nullptr) {
            // DEFERRED EIGENVECTORS (96 < n <= 140: the matrix fits the LDS, matrix + vectors do not).  The sweeps never read
            // V, so a sweep first runs on A alone and logs its rotations' (c, s) -- the pairs (p, q) of set r follow from r --
            // then A steps aside into global scratch, V comes into the same LDS, the logged sets are replayed on it
            // there, V goes back and A returns.  Rotating V in global memory set by set made every set wait for its
            // stores to become visible at the barrier (12.7 us per set at n = 114 against 5 us with both in LDS).
            double* const Vl = jl.A;          // V's place in LDS during a replay: rows < m, row stride lda
            auto pair_of = [&](int r, int k, int& pp, int& qq) {
                if (k == 0) { pp = mp - 1; qq = r; }
                else {
                    pp = r + k; if (pp >= mp - 1) pp -= mp - 1;
                    qq = r - k; if (qq < 0) qq += mp - 1;
                }
                if (pp > qq) { const int t = pp; pp = qq; qq = t; }
            };
            for (int sweep = 0; sweep < max_sweeps; ++sweep) {
                if (tid == 0) jl.flag[0] = 0;
                for (int i = tid; i < mp; i += NT) jl.rflag[i] = 0;
                __syncthreads();
                for (int r = 0; r < mp - 1; ++r) {
                    if (tid < half) {
                        int p, q;
                        pair_of(r, tid, p, q);
                        const double apq = A[p * lda + q];
                        double c = 1.0, sn = 0.0;
                        if (fabs(apq) > thresh) {
                            const double app = A[p * lda + p], aqq = A[q * lda + q];
                            const double theta = (aqq - app) * jacobi_rcp(2.0 * apq);
                            const double h2 = theta * theta + 1.0;
                            const double t = (theta >= 0.0 ? 1.0 : -1.0) * jacobi_rcp(fabs(theta) + h2 * jacobi_rsqrt(h2));
                            c = jacobi_rsqrt(t * t + 1.0);
                            sn = t * c;
                            jl.flag[0] = 1;
                            jl.rflag[r] = 1;
                        }
                        jl.rc[tid] = c; jl.rs[tid] = sn; jl.rp[tid] = p; jl.rq[tid] = q;
                        jl.rlog[((size_t)r * half + tid) * 2] = c;
                        jl.rlog[((size_t)r * half + tid) * 2 + 1] = sn;
                    }
                    __syncthreads();
                    if (jl.rflag[r] == 0) continue;
                    for (int idx = tid, k = kb0, l = lb0; idx < half * half; idx += NT) {
                        const double s1 = jl.rs[k], s2 = jl.rs[l];
                        if (s1 != 0.0 || s2 != 0.0) {
                            const double c1 = jl.rc[k], c2 = jl.rc[l];
                            const int p1 = jl.rp[k], q1 = jl.rq[k], p2 = jl.rp[l], q2 = jl.rq[l];
                            const double b00 = A[p1 * lda + p2], b01 = A[p1 * lda + q2];
                            const double b10 = A[q1 * lda + p2], b11 = A[q1 * lda + q2];
                            const double t00 = c1 * b00 - s1 * b10, t01 = c1 * b01 - s1 * b11;
                            const double t10 = s1 * b00 + c1 * b10, t11 = s1 * b01 + c1 * b11;
                            A[p1 * lda + p2] = c2 * t00 - s2 * t01;
                            A[p1 * lda + q2] = s2 * t00 + c2 * t01;
                            A[q1 * lda + p2] = c2 * t10 - s2 * t11;
                            A[q1 * lda + q2] = s2 * t10 + c2 * t11;
                        }
                        k += dkb; l += dlb;
                        if (l >= half) { l -= half; ++k; }
                    }
                    __syncthreads();
                }
                if (jl.flag[0] == 0) break;
                // ---- replay the sweep on V: A out, V in
                for (int idx = tid; idx < m * m; idx += NT) { const int i = idx / m, j = idx - i * m; jl.spill[idx] = A[i * lda + j]; }
                __syncthreads();
                for (int idx = tid; idx < m * m; idx += NT) { const int i = idx / m, j = idx - i * m; Vl[i * lda + j] = V[i * ldv + j]; }
                __syncthreads();
                for (int r = 0; r < mp - 1; ++r) {
                    if (jl.rflag[r] == 0) continue;
                    if (tid < half) {
                        int p, q;
                        pair_of(r, tid, p, q);
                        jl.rc[tid] = jl.rlog[((size_t)r * half + tid) * 2];
                        jl.rs[tid] = jl.rlog[((size_t)r * half + tid) * 2 + 1];
                        jl.rp[tid] = p; jl.rq[tid] = q;
                    }
                    __syncthreads();
                    for (int idx = tid, k = k0, row = c0; idx < half * mp; idx += NT, k += dk, row += dc) {
                        if (row >= mp) { row -= mp; ++k; }
                        const double sn = jl.rs[k];
                        if (sn != 0.0 && row < m) {
                            const double c = jl.rc[k];
                            const int p = jl.rp[k], q = jl.rq[k];
                            const double vp = Vl[row * lda + p], vq = Vl[row * lda + q];
                            Vl[row * lda + p] = c * vp - sn * vq;
                            Vl[row * lda + q] = sn * vp + c * vq;
                        }
                    }
                    __syncthreads();
                }
                // ---- V out, A back (its padding row and column are zero)
                for (int idx = tid; idx < m * m; idx += NT) { const int i = idx / m, j = idx - i * m; V[i * ldv + j] = Vl[i * lda + j]; }
                __syncthreads();
                for (int idx = tid; idx < mp * lda; idx += NT) A[idx] = 0.0;
                __syncthreads();
                for (int idx = tid; idx < m * m; idx += NT) { const int i = idx / m, j = idx - i * m; A[i * lda + j] = jl.spill[idx]; }
                __syncthreads();
            }
            __syncthreads();
            return;
        }
    }
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        if (tid == 0) { jl.flag[0] = 0; jl.flag[1] = 0; jl.flag[2] = 0; jl.flag[3] = 0; }
        __syncthreads();
        for (int r = 0; r < mp - 1; ++r) {
            // flag[1 + r % 3]: does this rotation set rotate anything?  With a warm start most sets of the late
            // sweeps do not, and their two update phases (and barriers) are skipped.  Three slots in rotation:
            // the slot reset here was last read two sets ago, i.e. before a barrier every thread has passed.
            if (tid == 0) jl.flag[1 + (r + 1) % 3] = 0;
            if (tid < half) {
                int p, q;
                if (tid == 0) { p = mp - 1; q = r; }
                else {      // (r + tid) mod (mp - 1) and (r - tid) mod (mp - 1) without integer division
                    p = r + tid; if (p >= mp - 1) p -= mp - 1;
                    q = r - tid; if (q < 0) q += mp - 1;
                }
                if (p > q) { const int t = p; p = q; q = t; }
                const double apq = A[p * lda + q];
                double c = 1.0, s = 0.0;
                if (fabs(apq) > thresh) {
                    const double app = A[p * lda + p], aqq = A[q * lda + q];
                    // hardware-seeded reciprocal / rsqrt + two Newton steps (<= 1 ulp): this serial prologue of every
                    // rotation set costs as much as the update it precedes when written with IEEE division and sqrt
                    const double theta = (aqq - app) * jacobi_rcp(2.0 * apq);
                    const double h2 = theta * theta + 1.0;
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) * jacobi_rcp(fabs(theta) + h2 * jacobi_rsqrt(h2));
                    c = jacobi_rsqrt(t * t + 1.0);
                    s = t * c;
                    jl.flag[0] = 1;
                    jl.flag[1 + r % 3] = 1;
                }
                jl.rc[tid] = c; jl.rs[tid] = s; jl.rp[tid] = p; jl.rq[tid] = q;
            }
            __syncthreads();
            if (g_jacobi_skip_idle_sets && jl.flag[1 + r % 3] == 0) continue;
            // A <- J^T A J in ONE phase: the rotation pairs partition the indices, so the 2 x 2 blocks
            // A[{p_k, q_k}][{p_l, q_l}] are disjoint and each is transformed from both sides by one thread
            // (one read and one write per element, no barrier between the row and the column half)
            for (int idx = tid, k = kb0, l = lb0; idx < half * half; idx += NT) {
                const double s1 = jl.rs[k], s2 = jl.rs[l];
                if (s1 != 0.0 || s2 != 0.0) {
                    const double c1 = jl.rc[k], c2 = jl.rc[l];
                    const int p1 = jl.rp[k], q1 = jl.rq[k], p2 = jl.rp[l], q2 = jl.rq[l];
                    const double b00 = A[p1 * lda + p2], b01 = A[p1 * lda + q2];
                    const double b10 = A[q1 * lda + p2], b11 = A[q1 * lda + q2];
                    const double t00 = c1 * b00 - s1 * b10, t01 = c1 * b01 - s1 * b11;
                    const double t10 = s1 * b00 + c1 * b10, t11 = s1 * b01 + c1 * b11;
                    A[p1 * lda + p2] = c2 * t00 - s2 * t01;
                    A[p1 * lda + q2] = s2 * t00 + c2 * t01;
                    A[q1 * lda + p2] = c2 * t10 - s2 * t11;
                    A[q1 * lda + q2] = s2 * t10 + c2 * t11;
                }
                k += dkb; l += dlb;
                if (l >= half) { l -= half; ++k; }
            }
            // V <- V J
            for (int idx = tid, k = k0, row = c0; idx < half * mp; idx += NT, k += dk, row += dc) {
                if (row >= mp) { row -= mp; ++k; }
                const double s = jl.rs[k];
                if (s != 0.0 && (VLDS || row < m)) {
                    const double c = jl.rc[k];
                    const int p = jl.rp[k], q = jl.rq[k];
                    const double vp = V[row * ldv + p], vq = V[row * ldv + q];
                    V[row * ldv + p] = c * vp - s * vq;
                    V[row * ldv + q] = s * vp + c * vq;
                }
            }
            __syncthreads();
        }
        if (jl.flag[0] == 0) break;
        __syncthreads();
    }
    __syncthreads();
}

// rank of eigenvalue i among the first m diagonal entries (ascending, ties by index)
__device__ __forceinline__ int eig_rank(const double* A, int lda, int m, int i)
{
    const double wi = A[i * lda + i];
    int r = 0;
    for (int j = 0; j < m; ++j) {
        const double wj = A[j * lda + j];
        r += (wj < wi || (wj == wi && j < i)) ? 1 : 0;
    }
    return r;
}

struct FragPtrs {
    double *S, *H, *X, *F, *D, *C, *J, *K, *W, *eps, *scal;
    double *diis_f, *diis_e, *diis_b;
    int *diis_state, *istate;
};

__device__ __forceinline__ FragPtrs frag_ptrs(const BatchView& bv, int f)
{
    const size_t nn = (size_t)bv.n * bv.n;
    FragPtrs p;
    p.S = bv.S + f * nn; p.H = bv.H + f * nn; p.X = bv.X + f * nn; p.F = bv.F + f * nn;
    p.D = bv.D + f * nn; p.C = bv.C + f * nn; p.J = bv.J + f * nn; p.K = bv.K + f * nn;
    p.W = bv.W + f * 6 * nn; p.eps = bv.eps + (size_t)f * bv.n; p.scal = bv.scal + (size_t)f * 8;
    p.diis_f = bv.diis_f + f * DIIS_MAX * nn; p.diis_e = bv.diis_e + f * DIIS_MAX * nn;
    p.diis_b = bv.diis_b + (size_t)f * DIIS_MAX * DIIS_MAX;
    p.diis_state = bv.diis_state + 2 * f; p.istate = bv.istate + 4 * f;
    return p;
}

// ---------------------------------------------------------------------------------------
// X = U s^{-1/2} over overlap eigenvalues > 1e-7, ascending (build_orthogonalizer,
// src/scf/mqc_scf_common.f90:43-83).  nmo goes to istate[2].
template <int JM>
__global__ void __launch_bounds__(NT) orthogonalizer_kernel(BatchView bv)
{
    constexpr bool VLDS = JM == 2;      // JM: 2 = A and V in LDS, 1 = A in LDS and V in global memory, 0 = both global (n > 140)
    extern __shared__ double lds[];
    const int f = blockIdx.x, n = bv.n, tid = threadIdx.x;
    FragPtrs p = frag_ptrs(bv, f);
    JacobiLds jl = carve_jacobi(lds, n, JM, p.W, p.W + 4 * (size_t)n * n + n);
    if (JM == 1 && even_up(n) == n) { jl.spill = p.W + 2 * (size_t)n * n; jl.rlog = p.W + 4 * (size_t)n * n + n; }   // W2, and W4 behind the rank ints: (n - 1) n doubles
    const int mp = even_up(n), lda = jl.lda;
    const int ldv = VLDS ? lda : n;
    for (int idx = tid; idx < mp * lda; idx += NT) jl.A[idx] = 0.0;
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += NT) jl.A[(idx / n) * lda + (idx % n)] = p.S[idx];
    __syncthreads();
    jacobi_eig<JM>(jl, n, ldv);
    // count dropped modes
    int drop = 0;
    for (int i = 0; i < n; ++i) drop += (jl.A[i * lda + i] > OVERLAP_EIG_TOL) ? 0 : 1;
    const int nmo = n - drop;
    for (int idx = tid; idx < n * n; idx += NT) p.X[idx] = 0.0;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        const double w = jl.A[i * lda + i];
        if (!(w > OVERLAP_EIG_TOL)) continue;
        const int col = eig_rank(jl.A, lda, n, i) - drop;
        const double sc = 1.0 / sqrt(w);
        for (int r = tid; r < n; r += NT) p.X[r * n + col] = jl.V[r * ldv + i] * sc;
    }
    if (tid == 0) p.istate[2] = nmo;
}

// F (n x n, global) -> eps, C, D.  Uses W[2] (n x m) as scratch and the Jacobi LDS block.
// WARM: the previous iteration's eigenvectors Vp (orthogonal basis, bv.Vprev) pre-rotate F' so
// that the Jacobi sweeps start from a nearly diagonal matrix; V then starts from Vp, so the final
// V is the full eigenvector matrix and no extra product is needed.
// vprev / nocc / dfac: spin channel of an unrestricted run (defaults: the closed-shell values)
template <int JM, bool WARM>
__device__ void diagonalize_and_density(const BatchView& bv, FragPtrs& p, JacobiLds& jl, int m, double* vprev = nullptr, int nocc = -1,
                                        double dfac = 2.0)
{
    constexpr bool VLDS = JM == 2;      // JM: 2 = A and V in LDS, 1 = A in LDS and V in global memory, 0 = both global (n > 140)
    const int n = bv.n, tid = threadIdx.x;
    const size_t nn = (size_t)n * n;
    double* T = p.W + 2 * nn;       // F X  (n x m), then F' Vp (m x m)
    double* Vg = p.W + 3 * nn;      // global eigenvectors when they do not fit in LDS
    double* Vp = (vprev ? vprev : bv.Vprev) + (size_t)blockIdx.x * nn;    // m x m, ld n
    if (nocc < 0) nocc = bv.nocc;
    const int mp = even_up(m), lda = jl.lda;
    const int ldv = VLDS ? lda : m;
    if (!VLDS) jl.V = Vg;
    // deferred eigenvectors (jacobi_eig, mode 1): T is dead once F' sits in LDS, W4 behind the rank ints is unused in this mode
    if (JM == 1 && even_up(m) <= n) { jl.spill = T; jl.rlog = p.W + 4 * nn + n; }
    wg_gemm_mfma<false, false>(n, m, n, p.F, n, p.X, n, [&](int i, int j, double v) { T[i * n + j] = v; });
    for (int idx = tid; idx < mp * lda; idx += NT) jl.A[idx] = 0.0;
    __syncthreads();
    // F' = X^T (F X) straight into LDS
    double* const Alds = jl.A;
    wg_gemm_mfma<true, false>(m, m, n, p.X, n, T, n, [&](int i, int j, double v) { Alds[i * lda + j] = v; });
    if (WARM) {
        // T = F' Vp  (m x m, ld n), then F'' = Vp^T T back into LDS
        wg_gemm_mfma<false, false>(m, m, m, Alds, lda, Vp, n, [&](int i, int j, double v) { T[i * n + j] = v; });
        wg_gemm_mfma<true, false>(m, m, m, Vp, n, T, n, [&](int i, int j, double v) { Alds[i * lda + j] = v; });
    }
    // symmetrise (F' is symmetric up to rounding; Jacobi assumes exact symmetry)
    for (int idx = tid; idx < m * m; idx += NT) {
        const int i = idx / m, j = idx - i * m;
        if (i < j) {
            const double a = 0.5 * (jl.A[i * lda + j] + jl.A[j * lda + i]);
            jl.A[i * lda + j] = a; jl.A[j * lda + i] = a;
        }
    }
    __syncthreads();
#if SCF_STAMPS
    unsigned long long sst_t = __builtin_amdgcn_s_memtime();
    SCF_ST(3)          /* (re-based: counts only the barrier) */
#endif
    jacobi_eig<JM>(jl, m, ldv, WARM ? Vp : nullptr, n);
    SCF_ST(4)
    // keep the eigenvectors for the next iteration's warm start
    for (int idx = tid; idx < m * m; idx += NT) {
        const int i = idx / m, j = idx - i * m;
        Vp[i * n + j] = jl.V[i * ldv + j];
    }
    // sorted eigenvalues + C = X C'
    int* rank = (int*)(p.W + 4 * nn);   // m ints
    for (int i = tid; i < m; i += NT) {
        const int r = eig_rank(jl.A, lda, m, i);
        rank[i] = r;
        p.eps[r] = jl.A[i * lda + i];
    }
    __syncthreads();
    // C[r][rank[i]] = sum_k X[r][k] V[k][i]
    double* const Cg = p.C;
    wg_gemm_mfma<false, false>(n, m, m, p.X, n, jl.V, ldv, [&](int r, int i, double v) { Cg[r * n + rank[i]] = v; });
    // D = 2 C_occ C_occ^T
    double* const Dg = p.D;
    wg_gemm_mfma<false, true>(n, n, nocc, Cg, n, Cg, n, [&](int i, int j, double v) { Dg[i * n + j] = dfac * v; });
}

// the beta spin's view of a fragment: same S, H, X, W, state; its own F, D, C, J, K, eps and histories
__device__ __forceinline__ FragPtrs frag_ptrs_beta(const BatchView& bv, int f)
{
    const size_t nn = (size_t)bv.n * bv.n;
    FragPtrs p = frag_ptrs(bv, f);
    p.F = bv.Fb + f * nn; p.D = bv.Db + f * nn; p.C = bv.Cb + f * nn; p.J = bv.Jb + f * nn; p.K = bv.Kb + f * nn;
    p.eps = bv.epsb + (size_t)f * bv.n;
    p.diis_f = bv.diis_fb + f * DIIS_MAX * nn; p.diis_e = bv.diis_eb + f * DIIS_MAX * nn;
    return p;
}

// Starting Fock (core or GWH, guess_fock mqc_libcint_rhf.f90:1354-1380), then the first density.
template <int JM>
__global__ void __launch_bounds__(NT) guess_kernel(BatchView bv, int guess_kind)
{
    constexpr bool VLDS = JM == 2;      // JM: 2 = A and V in LDS, 1 = A in LDS and V in global memory, 0 = both global (n > 140)
    extern __shared__ double lds[];
    const int f = blockIdx.x, n = bv.n, tid = threadIdx.x;
    FragPtrs p = frag_ptrs(bv, f);
    const int m = p.istate[2];
    JacobiLds jl = carve_jacobi(lds, m, JM, nullptr, (JM == 0) ? bv.W + ((size_t)blockIdx.x * 6 + 4) * n * n + n : nullptr);
    for (int idx = tid; idx < n * n; idx += NT) {
        const int i = idx / n, j = idx - i * n;
        double v = p.H[idx];
        if (guess_kind == MQC_HIP_GUESS_GWH && i != j) v = 0.5 * GWH_K * p.S[idx] * (p.H[i * n + i] + p.H[j * n + j]);
        // superposed atoms: the Hartree-Fock Fock matrix of the guess density, J and K built by the engine just before
        // (atomic_guess_fock, mqc_libcint_rhf.f90:1382-1411: full exchange whatever the functional)
        if (guess_kind == MQC_HIP_GUESS_SAD) v += p.J[idx] - 0.5 * p.K[idx];
        p.F[idx] = v;
    }
    __syncthreads();
    if (bv.uhf) {
        // symmetric guess: both spins get the orbitals of the same starting Fock, the occupations separate them
        // (run_libcint_uhf, mqc_libcint_rhf.f90:838-866)
        diagonalize_and_density<JM, false>(bv, p, jl, m, nullptr, bv.nalpha, 1.0);
        FragPtrs pb = frag_ptrs_beta(bv, f);
        const size_t nn = (size_t)n * n;
        double* Vpa = bv.Vprev + (size_t)f * nn; double* Vpb = bv.Vprevb + (size_t)f * nn;
        for (int idx = tid; idx < n * n; idx += NT) { pb.C[idx] = p.C[idx]; Vpb[idx] = Vpa[idx]; pb.F[idx] = p.F[idx]; }
        for (int i = tid; i < n; i += NT) pb.eps[i] = p.eps[i];
        __syncthreads();
        double* const Dbg = pb.D; const double* Cbg = pb.C;
        wg_gemm_mfma<false, true>(n, n, bv.nbeta, Cbg, n, Cbg, n, [&](int i, int j, double v) { Dbg[i * n + j] = v; });
    } else {
        diagonalize_and_density<JM, false>(bv, p, jl, m);
    }
    if (tid == 0) {
        p.istate[0] = ST_ITER; p.istate[1] = 0; p.istate[3] = 0;
        p.diis_state[0] = 0; p.diis_state[1] = 0;
        p.scal[0] = 0.0; p.scal[1] = 0.0; p.scal[2] = 0.0; p.scal[3] = 0.0; p.scal[4] = 0.0;
    }
}

// slot (1-based) of the age-th oldest entry, age = 1..n_stored (diis_slot_of_age, mqc_diis.f90:89-103)
__host__ __device__ inline int diis_slot_of_age(int newest, int n_stored, int max_vectors, int age)
{
    int v = (newest - n_stored + age - 1) % max_vectors;
    if (v < 0) v += max_vectors;
    return v + 1;
}

// DIIS weights, oldest first, from the slot-indexed overlap cache: the reference's
// diis_coefficients + solve_diis (mqc_diis.f90:164-273) transcribed operation for operation.
// aug: (DIIS_MAX + 1) x (DIIS_MAX + 2) doubles of workspace supplied by the caller (LDS in the kernels: a private array
// indexed at run time would live in scratch memory)
constexpr int DIIS_AUG = (DIIS_MAX + 1) * (DIIS_MAX + 2);
__host__ __device__ inline bool diis_solve(const double* overlap /* [maxv*maxv] slot coords */, int newest,
                                           int n_stored, int max_vectors, double* coef /* n_stored+1 */, double* aug)
{
    if (n_stored < 2) return false;
    const int n = n_stored, N = n + 1;
    const int ld = N + 1;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) aug[i * ld + j] = -1.0;
    aug[n * ld + n] = 0.0;
    double scale = 0.0;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            const int si = diis_slot_of_age(newest, n_stored, max_vectors, i + 1) - 1;
            const int sj = diis_slot_of_age(newest, n_stored, max_vectors, j + 1) - 1;
            const double v = overlap[si * max_vectors + sj];
            aug[i * ld + j] = v;
            scale = fmax(scale, fabs(v));
        }
    if (scale > 0.0)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) aug[i * ld + j] /= scale;
    for (int i = 0; i < N; ++i) aug[i * ld + N] = 0.0;
    aug[(N - 1) * ld + N] = -1.0;
    for (int i = 0; i < N; ++i) {
        int piv = i;
        for (int j = i + 1; j < N; ++j)
            if (fabs(aug[j * ld + i]) > fabs(aug[piv * ld + i])) piv = j;
        if (piv != i)
            for (int c = 0; c <= N; ++c) { const double t = aug[i * ld + c]; aug[i * ld + c] = aug[piv * ld + c]; aug[piv * ld + c] = t; }
        const double pivot = aug[i * ld + i];
        if (fabs(pivot) < PIVOT_FLOOR) return false;
        for (int j = i + 1; j < N; ++j) {
            const double factor = aug[j * ld + i] / pivot;
            for (int c = i; c <= N; ++c) aug[j * ld + c] -= factor * aug[i * ld + c];
        }
    }
    for (int i = N - 1; i >= 0; --i) {
        double s = aug[i * ld + N];
        for (int c = i + 1; c < N; ++c) s -= aug[i * ld + c] * coef[c];
        coef[i] = s / aug[i * ld + i];
    }
    return true;
}

template <int JM>
__global__ void __launch_bounds__(NT) scf_step_kernel(BatchView bv)
{
    constexpr bool VLDS = JM == 2;      // JM: 2 = A and V in LDS, 1 = A in LDS and V in global memory, 0 = both global (n > 140)
    extern __shared__ double lds[];
    const int f = blockIdx.x, n = bv.n, tid = threadIdx.x;
    FragPtrs p = frag_ptrs(bv, f);
    const int state = p.istate[0];
    if (state == ST_DONE) return;
    const int m = p.istate[2];
    const size_t nn = (size_t)n * n;
    JacobiLds jl = carve_jacobi(lds, m, JM, nullptr, (JM == 0) ? bv.W + ((size_t)blockIdx.x * 6 + 4) * n * n + n : nullptr);
    SCF_ST_DECL

    // ---- Fock assembly and energy (assemble_fock :985-990,1206-1228; electronic_energy :1691)
    // The energy is taken from the Fock matrix BEFORE V_xc is added, plus E_xc
    // (mqc_libcint_rhf.f90:985-990,1206-1228).
    double e = 0.0;
    const double* Ax = bv.Vxc ? bv.Vxc + (size_t)f * nn : nullptr;
    for (int idx = tid; idx < n * n; idx += NT) {
        const double h = p.H[idx];
        double fk = h + p.J[idx] - 0.5 * bv.exx * p.K[idx];
        e += p.D[idx] * (h + fk);
        if (Ax) { const int i = idx / n, j = idx - i * n; fk += Ax[idx] + Ax[j * n + i]; }
        p.F[idx] = fk;
    }
    e = 0.5 * block_sum(e, jl.red);
    if (Ax) e += p.scal[5];
    if (state == ST_FINAL) {
        if (tid == 0) { p.scal[4] = e; p.istate[0] = ST_DONE; }
        return;
    }

    SCF_ST(0)
    // ---- DIIS error e = X^T (F D S - S D F) X   (commutator :1326-1352)
    double* W0 = p.W; double* W1 = p.W + nn; double* W2 = p.W + 2 * nn; double* Err = p.W + 5 * nn;
    wg_gemm_mfma<false, false>(n, n, n, p.F, n, p.D, n, [&](int i, int j, double v) { W0[i * n + j] = v; });     // F D
    wg_gemm_mfma<false, false>(n, n, n, W0, n, p.S, n, [&](int i, int j, double v) { W1[i * n + j] = v; });       // F D S
    for (int idx = tid; idx < n * n; idx += NT) {            // A - A^T  (S D F = (F D S)^T)
        const int i = idx / n, j = idx - i * n;
        W0[idx] = W1[idx] - W1[j * n + i];
    }
    __syncthreads();
    wg_gemm_mfma<false, false>(n, m, n, W0, n, p.X, n, [&](int i, int j, double v) { W2[i * n + j] = v; });       // (.) X      n x m
    wg_gemm_mfma<true, false>(m, m, n, p.X, n, W2, n, [&](int i, int j, double v) { Err[i * m + j] = v; });       // X^T (.)    m x m, ld m

    SCF_ST(1)
    // ---- DIIS push / extrapolate (mqc_diis.f90:113-162; RHF extrapolates from the first iteration)
    const int maxv = bv.diis_size;
    if (maxv > 0) {
        int n_stored = p.diis_state[0], newest = p.diis_state[1];
        newest = newest % maxv + 1;
        if (n_stored < maxv) n_stored += 1;
        const int slot = newest - 1;
        double* fh = p.diis_f + (size_t)slot * nn;
        double* eh = p.diis_e + (size_t)slot * nn;
        for (int idx = tid; idx < n * n; idx += NT) fh[idx] = p.F[idx];
        for (int idx = tid; idx < m * m; idx += NT) eh[idx] = Err[idx];
        __syncthreads();
        // the new row of the overlap matrix: all n_stored inner products in ONE pass over the newest error vector and one
        // block reduction (a loop of block sums cost two barriers per stored vector)
        {
            __shared__ double bpart[NT / 64][DIIS_MAX];
            double sv[DIIS_MAX];
#pragma unroll
            for (int a = 0; a < DIIS_MAX; ++a) sv[a] = 0.0;
            for (int idx = tid; idx < m * m; idx += NT) {
                const double en = eh[idx];
#pragma unroll
                for (int a = 0; a < DIIS_MAX; ++a)
                    if (a < n_stored) sv[a] += en * p.diis_e[(size_t)(diis_slot_of_age(newest, n_stored, maxv, a + 1) - 1) * nn + idx];
            }
#pragma unroll
            for (int a = 0; a < DIIS_MAX; ++a) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) sv[a] += __shfl_down(sv[a], off, 64);
            }
            if ((tid & 63) == 0) {
#pragma unroll
                for (int a = 0; a < DIIS_MAX; ++a) bpart[tid >> 6][a] = sv[a];
            }
            __syncthreads();
            if (tid < n_stored) {
                double t = 0.0;
                for (int w = 0; w < NT / 64; ++w) t += bpart[w][tid];
                const int other = diis_slot_of_age(newest, n_stored, maxv, tid + 1) - 1;
                p.diis_b[slot * maxv + other] = t; p.diis_b[other * maxv + slot] = t;
            }
        }
        __syncthreads();
        double* coef = jl.rc;     // reuse LDS (>= 9 doubles: mp/2 >= 9 needs m >= 18; fall back to red otherwise)
        __shared__ double coef_s[DIIS_MAX + 2];
        __shared__ int ok_s;
        __shared__ double aug_s[DIIS_AUG];
        if (tid == 0) {
            const bool ok = diis_solve(p.diis_b, newest, n_stored, maxv, coef_s, aug_s);
            ok_s = ok ? 1 : 0;
            p.diis_state[0] = n_stored; p.diis_state[1] = newest;
        }
        (void)coef;
        __syncthreads();
        if (ok_s) {
            for (int idx = tid; idx < n * n; idx += NT) {
                double s = 0.0;
                for (int i = 0; i < n_stored; ++i) {
                    const int sl = diis_slot_of_age(newest, n_stored, maxv, i + 1) - 1;
                    s += coef_s[i] * p.diis_f[(size_t)sl * nn + idx];
                }
                p.F[idx] = s;
            }
        }
        __syncthreads();
    }

    SCF_ST(2)
    // ---- keep the old density, diagonalise, new density
    double* Dold = p.W + nn;    // W1 is free again
    for (int idx = tid; idx < n * n; idx += NT) Dold[idx] = p.D[idx];
    __syncthreads();
    diagonalize_and_density<JM, true>(bv, p, jl, m);
    SCF_ST(6)

    double d2 = 0.0;
    for (int idx = tid; idx < n * n; idx += NT) { const double d = p.D[idx] - Dold[idx]; d2 += d * d; }
    d2 = block_sum(d2, jl.red);
    if (tid == 0) {
        const double e_old = p.scal[1];
        const double de = fabs(e - e_old);
        const double drms = sqrt(d2 / (double)(n * n));
        const int iter = p.istate[1] + 1;
        p.scal[0] = e; p.scal[1] = e; p.scal[2] = de; p.scal[3] = drms;
        p.istate[1] = iter;
        if (iter > 1 && de < bv.e_tol && drms < bv.d_tol) { p.istate[3] = 1; p.istate[0] = ST_FINAL; }
        else if (iter >= bv.max_iter) { p.istate[3] = 0; p.istate[0] = ST_FINAL; }
    }
}

// ---------------------------------------------------------------------------------------
// Unrestricted step (run_libcint_uhf, mqc_libcint_rhf.f90:868-931): F_s = H + J[D_a + D_b] - exx K[D_s],
// E = 1/2 sum_s D_s (H + F_s); the commutators of both spins laid end to end form ONE DIIS error vector, the two
// Fock matrices one DIIS Fock vector (a single coefficient set moves both spins), extrapolation starts at
// iteration 4 (DEFAULT_UHF_DIIS_START, :51); both spins are diagonalised; dE and the rms over BOTH density changes
// decide convergence; the final state rebuilds the energy from the converged densities.
constexpr int UHF_DIIS_START = 4;

template <int JM>
__global__ void __launch_bounds__(NT) scf_step_uhf_kernel(BatchView bv)
{
    constexpr bool VLDS = JM == 2;      // JM: 2 = A and V in LDS, 1 = A in LDS and V in global memory, 0 = both global (n > 140)
    extern __shared__ double lds[];
    const int f = blockIdx.x, n = bv.n, tid = threadIdx.x;
    FragPtrs pa = frag_ptrs(bv, f);
    FragPtrs pb = frag_ptrs_beta(bv, f);
    const int state = pa.istate[0];
    if (state == ST_DONE) return;
    const int m = pa.istate[2];
    const size_t nn = (size_t)n * n;
    JacobiLds jl = carve_jacobi(lds, m, JM, nullptr, (JM == 0) ? bv.W + ((size_t)blockIdx.x * 6 + 4) * n * n + n : nullptr);

    // unrestricted Kohn-Sham: exchange scaled by the functional's fraction, the energy from the Fock matrices BEFORE
    // the spin potentials are added, plus E_xc; V_s = A_s + A_s^T with A_b behind A_a (kern_xc.hip, xc_uks_kernel)
    double e = 0.0;
    const double* Axa = bv.Vxc ? bv.Vxc + (size_t)f * nn : nullptr;
    const double* Axb = bv.Vxc ? bv.Vxc + ((size_t)bv.nfrag + f) * nn : nullptr;
    for (int idx = tid; idx < n * n; idx += NT) {
        const double h = pa.H[idx];
        const double j = pa.J[idx] + pb.J[idx];
        double fa = h + j - bv.exx * pa.K[idx];
        double fb = h + j - bv.exx * pb.K[idx];
        e += pa.D[idx] * (h + fa) + pb.D[idx] * (h + fb);
        if (Axa) {
            const int i = idx / n, jj = idx - i * n;
            fa += Axa[idx] + Axa[jj * n + i];
            fb += Axb[idx] + Axb[jj * n + i];
        }
        pa.F[idx] = fa; pb.F[idx] = fb;
    }
    e = 0.5 * block_sum(e, jl.red);
    if (Axa) e += pa.scal[5];
    if (state == ST_FINAL) {
        if (tid == 0) { pa.scal[4] = e; pa.istate[0] = ST_DONE; }
        return;
    }
    const int iter = pa.istate[1] + 1;

    // commutators straight into the newest history slot (the slot is claimed even without DIIS: one slot of scratch)
    const int maxv = bv.diis_size;
    int n_stored = pa.diis_state[0], newest = pa.diis_state[1];
    if (maxv > 0) { newest = newest % maxv + 1; if (n_stored < maxv) n_stored += 1; }
    const int slot = maxv > 0 ? newest - 1 : 0;
    double* W0 = pa.W; double* W1 = pa.W + nn; double* W2 = pa.W + 2 * nn;
    for (int spin = 0; spin < 2; ++spin) {
        FragPtrs& p = spin ? pb : pa;
        double* Err = p.diis_e + (size_t)slot * nn;
        wg_gemm_mfma<false, false>(n, n, n, p.F, n, p.D, n, [&](int i, int j, double v) { W0[i * n + j] = v; });
        wg_gemm_mfma<false, false>(n, n, n, W0, n, p.S, n, [&](int i, int j, double v) { W1[i * n + j] = v; });
        for (int idx = tid; idx < n * n; idx += NT) {
            const int i = idx / n, j = idx - i * n;
            W0[idx] = W1[idx] - W1[j * n + i];
        }
        __syncthreads();
        wg_gemm_mfma<false, false>(n, m, n, W0, n, p.X, n, [&](int i, int j, double v) { W2[i * n + j] = v; });
        wg_gemm_mfma<true, false>(m, m, n, p.X, n, W2, n, [&](int i, int j, double v) { Err[i * m + j] = v; });
        if (maxv > 0) {
            double* fh = p.diis_f + (size_t)slot * nn;
            for (int idx = tid; idx < n * n; idx += NT) fh[idx] = p.F[idx];
        }
        __syncthreads();
    }
    if (maxv > 0) {
        const double* ea = pa.diis_e + (size_t)slot * nn;
        const double* eb = pb.diis_e + (size_t)slot * nn;
        for (int age = 1; age <= n_stored; ++age) {
            const int other = diis_slot_of_age(newest, n_stored, maxv, age) - 1;
            const double* oa = pa.diis_e + (size_t)other * nn;
            const double* ob = pb.diis_e + (size_t)other * nn;
            double s = 0.0;
            for (int idx = tid; idx < m * m; idx += NT) s += ea[idx] * oa[idx] + eb[idx] * ob[idx];
            s = block_sum(s, jl.red);
            if (tid == 0) { pa.diis_b[slot * maxv + other] = s; pa.diis_b[other * maxv + slot] = s; }
        }
        __syncthreads();
        __shared__ double coef_u[DIIS_MAX + 2];
        __shared__ int ok_u;
        __shared__ double aug_u[DIIS_AUG];
        if (tid == 0) {
            const bool ok = (iter >= UHF_DIIS_START) && diis_solve(pa.diis_b, newest, n_stored, maxv, coef_u, aug_u);
            ok_u = ok ? 1 : 0;
            pa.diis_state[0] = n_stored; pa.diis_state[1] = newest;
        }
        __syncthreads();
        if (ok_u) {
            for (int idx = tid; idx < n * n; idx += NT) {
                double sa = 0.0, sb = 0.0;
                for (int i = 0; i < n_stored; ++i) {
                    const int sl = diis_slot_of_age(newest, n_stored, maxv, i + 1) - 1;
                    sa += coef_u[i] * pa.diis_f[(size_t)sl * nn + idx];
                    sb += coef_u[i] * pb.diis_f[(size_t)sl * nn + idx];
                }
                pa.F[idx] = sa; pb.F[idx] = sb;
            }
        }
        __syncthreads();
    }

    // old densities to W0 / W1, both spins diagonalised, new densities
    for (int idx = tid; idx < n * n; idx += NT) { W0[idx] = pa.D[idx]; W1[idx] = pb.D[idx]; }
    __syncthreads();
    diagonalize_and_density<JM, true>(bv, pa, jl, m, bv.Vprev, bv.nalpha, 1.0);
    __syncthreads();
    diagonalize_and_density<JM, true>(bv, pb, jl, m, bv.Vprevb, bv.nbeta, 1.0);

    double d2 = 0.0;
    for (int idx = tid; idx < n * n; idx += NT) {
        const double da = pa.D[idx] - W0[idx], db = pb.D[idx] - W1[idx];
        d2 += da * da + db * db;
    }
    d2 = block_sum(d2, jl.red);
    if (tid == 0) {
        const double e_old = pa.scal[1];
        const double de = fabs(e - e_old);
        const double drms = sqrt(d2 / (double)(2 * n * n));
        pa.scal[0] = e; pa.scal[1] = e; pa.scal[2] = de; pa.scal[3] = drms;
        pa.istate[1] = iter;
        if (iter > 1 && de < bv.e_tol && drms < bv.d_tol) { pa.istate[3] = 1; pa.istate[0] = ST_FINAL; }
        else if (iter >= bv.max_iter) { pa.istate[3] = 0; pa.istate[0] = ST_FINAL; }
    }
}

__global__ void count_active_kernel(BatchView bv)
{
    int c = 0;
    for (int f = threadIdx.x; f < bv.nfrag; f += blockDim.x) c += (bv.istate[4 * f] != ST_DONE) ? 1 : 0;
    __shared__ int tot;
    if (threadIdx.x == 0) tot = 0;
    __syncthreads();
    atomicAdd(&tot, c);
    __syncthreads();
    if (threadIdx.x == 0) bv.counters[0] = tot;
}

// ---------------------------------------------------------------------------------------
}  // anonymous namespace

static int jacobi_mode(int n)
{
    if (jacobi_lds_doubles(n, true) * sizeof(double) <= 150 * 1024) return 2;
    if (jacobi_lds_doubles(n, false) * sizeof(double) <= 158 * 1024) return 1;      // n <= 140
    return 0;
}

#if MQC_SCF_NT == 256
size_t scf_lds_bytes(int n)
{
    const int mode = jacobi_mode(n);
    return jacobi_lds_doubles(n, mode == 2, mode != 0) * sizeof(double);
}
void launch_orthogonalizer_wide(const BatchView& bv, hipStream_t s);
void launch_guess_wide(const BatchView& bv, int guess_kind, hipStream_t s);
void launch_scf_step_wide(const BatchView& bv, hipStream_t s);
// batches of at most this many fragments take the 512-thread build (MQC_HIP_SCF_WIDE_MAX; 0 turns it off)
static int scf_wide_max()
{
    static const int v = [] { const char* e = std::getenv("MQC_HIP_SCF_WIDE_MAX"); return e ? std::atoi(e) : 640; }();
    return v;
}
#define MQC_SCF_PUBLIC(name) name
#else
#define MQC_SCF_PUBLIC(name) name##_wide
#endif

#define MQC_JACOBI_DISPATCH(KERNEL, N, ...)                                                   \
    do {                                                                                      \
        const int jm_ = jacobi_mode(N);                                                       \
        if (jm_ == 2) launch_wg(KERNEL<2>, __VA_ARGS__);                                      \
        else if (jm_ == 1) launch_wg(KERNEL<1>, __VA_ARGS__);                                 \
        else launch_wg(KERNEL<0>, __VA_ARGS__);                                               \
    } while (0)

template <typename K, typename... Args>
static void launch_wg(K kern, int nfrag, size_t lds, hipStream_t s, Args... args)
{
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(nfrag), dim3(NT), lds, s, args...);
}

void MQC_SCF_PUBLIC(launch_orthogonalizer)(const BatchView& bv, hipStream_t s)
{
#if MQC_SCF_NT == 256
    if (bv.nfrag <= scf_wide_max()) return launch_orthogonalizer_wide(bv, s);
#endif
    const size_t lds = scf_lds_bytes(bv.n);
    MQC_JACOBI_DISPATCH(orthogonalizer_kernel, bv.n, bv.nfrag, lds, s, bv);
}

void MQC_SCF_PUBLIC(launch_guess)(const BatchView& bv, int guess_kind, hipStream_t s)
{
#if MQC_SCF_NT == 256
    if (bv.nfrag <= scf_wide_max()) return launch_guess_wide(bv, guess_kind, s);
#endif
    const size_t lds = scf_lds_bytes(bv.n);
    MQC_JACOBI_DISPATCH(guess_kernel, bv.n, bv.nfrag, lds, s, bv, guess_kind);
}

static void apply_jacobi_env()
{
    static const bool done = [] {
        const char* e = std::getenv("MQC_HIP_JACOBI_SKIP");
        if (e && e[0] == '0') { const int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_jacobi_skip_idle_sets), &z, sizeof(int)); }
        if (const char* m = std::getenv("MQC_HIP_JACOBI_MAX_SWEEPS")) { const int v = std::atoi(m); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_jacobi_max_sweeps), &v, sizeof(int)); }
        return true;
    }();
    (void)done;
}

#if MQC_SCF_NT == 256
__global__ void broadcast_kernel(double* dst, const double* src, size_t count)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[(size_t)blockIdx.y * count + i] = src[i];
}

void launch_broadcast(double* dst, const double* src, size_t count, int nfrag, hipStream_t s)
{
    hipLaunchKernelGGL(broadcast_kernel, dim3((unsigned)((count + 255) / 256), nfrag), dim3(256), 0, s, dst, src, count);
}

#endif

void MQC_SCF_PUBLIC(launch_scf_step)(const BatchView& bv, hipStream_t s)
{
#if MQC_SCF_NT == 256
    if (bv.nfrag <= scf_wide_max()) return launch_scf_step_wide(bv, s);
#endif
    apply_jacobi_env();
    const size_t lds = scf_lds_bytes(bv.n);
    if (bv.uhf) MQC_JACOBI_DISPATCH(scf_step_uhf_kernel, bv.n, bv.nfrag, lds, s, bv);
    else MQC_JACOBI_DISPATCH(scf_step_kernel, bv.n, bv.nfrag, lds, s, bv);
    hipLaunchKernelGGL(count_active_kernel, dim3(1), dim3(256), 0, s, bv);
#if SCF_STAMPS
    (void)hipStreamSynchronize(s);
    unsigned long long h[8];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_scf_stamps), sizeof(h));
    std::fprintf(stderr, "scf stamps nfrag=%d n=%d: fock %llu | commutator %llu | diis %llu | jacobi %llu | diag total (gemms + jacobi + C, D) %llu\n",
                 bv.nfrag, bv.n, h[0], h[1], h[2], h[4], h[6]);
#endif
}

#if MQC_SCF_NT == 256
// ---- stage-level entry points ---------------------------------------------------------
template <int JM>
__global__ void __launch_bounds__(NT) syev_kernel(int n, const double* A, double* w, double* V, double* Aglb)
{
    constexpr bool VLDS = JM == 2;      // JM: 2 = A and V in LDS, 1 = A in LDS and V in global memory, 0 = both global (n > 140)
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    JacobiLds jl = carve_jacobi(lds, n, JM, V, Aglb);
    const int mp = even_up(n), lda = jl.lda, ldv = VLDS ? lda : n;
    if (JM == 1 && Aglb != nullptr) { jl.spill = Aglb; jl.rlog = Aglb + (size_t)n * n; }      // mode 1: scratch for the deferred eigenvectors
    for (int idx = tid; idx < mp * lda; idx += NT) jl.A[idx] = 0.0;
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += NT) jl.A[(idx / n) * lda + (idx % n)] = A[idx];
    __syncthreads();
    jacobi_eig<JM>(jl, n, ldv);
    __shared__ int rank_s[256];
    for (int i = tid; i < n; i += NT) { rank_s[i] = eig_rank(jl.A, lda, n, i); w[rank_s[i]] = jl.A[i * lda + i]; }
    __syncthreads();
    if (VLDS) {
        for (int idx = tid; idx < n * n; idx += NT) {
            const int r = idx / n, i = idx - r * n;
            V[r * n + rank_s[i]] = jl.V[r * ldv + i];
        }
    } else {
        // in-place column permutation of the global V through registers, one row at a time
        for (int r = 0; r < n; ++r) {
            double v = 0.0;
            if (tid < n) v = V[r * n + tid];
            __syncthreads();
            if (tid < n) V[r * n + rank_s[tid]] = v;
            __syncthreads();
        }
    }
}

void launch_syev(int n, double* dA, double* dw, double* dV, hipStream_t s)
{
    const size_t lds = scf_lds_bytes(n);
    const int mode = jacobi_mode(n);
    if (mode == 2) {
        (void)hipFuncSetAttribute((const void*)syev_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(syev_kernel<2>, dim3(1), dim3(NT), lds, s, n, dA, dw, dV, (double*)nullptr);
    } else if (mode == 1) {
        // scratch of the deferred eigenvectors: the matrix (n * n) and one sweep's rotations ((n + 1) (n + 2))
        double* aw = nullptr;
        if (hipMalloc(&aw, sizeof(double) * ((size_t)n * n + (size_t)(n + 2) * (n + 2))) != hipSuccess) aw = nullptr;
        (void)hipFuncSetAttribute((const void*)syev_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(syev_kernel<1>, dim3(1), dim3(NT), lds, s, n, dA, dw, dV, aw);
        (void)hipStreamSynchronize(s);
        if (aw) (void)hipFree(aw);
    } else {
        // stage-level call on a matrix too wide for LDS: the padded working copy lives in a temporary
        double* aw = nullptr;
        if (hipMalloc(&aw, sizeof(double) * (size_t)(n + 2) * (n + 2)) != hipSuccess) return;
        (void)hipFuncSetAttribute((const void*)syev_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(syev_kernel<0>, dim3(1), dim3(NT), lds, s, n, dA, dw, dV, aw);
        (void)hipStreamSynchronize(s);
        (void)hipFree(aw);
    }
}

__global__ void diis_coeff_kernel(int n_stored, const double* overlap, double* coef, int* ok)
{
    __shared__ double aug_s[DIIS_AUG];
    __shared__ double cf[DIIS_MAX + 2];
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        // the caller's matrix is already age-ordered: newest = n_stored, max_vectors = n_stored
        const bool good = diis_solve(overlap, n_stored, n_stored, n_stored, cf, aug_s);
        *ok = good ? 1 : 0;
        for (int i = 0; i < n_stored; ++i) coef[i] = good ? cf[i] : 0.0;
    }
}

void launch_diis_coeff(int n_stored, const double* d_overlap, double* d_coef, int* d_ok, hipStream_t s)
{
    hipLaunchKernelGGL(diis_coeff_kernel, dim3(1), dim3(64), 0, s, n_stored, d_overlap, d_coef, d_ok);
}

#endif      // MQC_SCF_NT == 256

}  // namespace mqc
