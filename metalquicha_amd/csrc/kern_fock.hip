// kern_fock.hip -- J[D] and K[D] from the HBM-resident packed ERI matrix, whole batch per launch.
//
// This is the per-iteration hot kernel of the exact-exchange path: it replaces the reference's
// in-core contraction (build_fock, backends/libcint/mqc_libcint_rhf.f90:1491-1574) and plays
// the role cuestDFCoulombCompute + cuestDFSymmetricExchangeCompute play in the cuEST loop
// (backends/cuest/backend/mqc_cuest_integrals.f90:1636-1748).
//
// Roofline: HBM-bound.  Per fragment and iteration it streams the pair matrix
// M[pair(i,j)][pair(k,l)] exactly once (npair^2 * 8 bytes; 11.06 MB for n = 48) and performs
// 5 FMAs per element (1 for J, 4 for the two symmetric mat-vecs of K) = 1.25 flop/byte, far
// below the FP64 ridge (~10 flop/byte), so the design goal is full-rate coalesced streaming
// with enough bytes in flight per CU to cover HBM latency:
//   * a wave owns one row pair(i,j); its 64 lanes read the row contiguously, ALL of the row's
//     loads issued back to back into registers (up to 19 x 512 B in flight per wave), and the
//     NEXT row's loads are issued before the current row is consumed (register double buffer);
//   * the row is staged in the wave's private LDS buffer (shell-pair-blocked data staged through
//     LDS, as the north star asks); no workgroup barrier is involved: a wave only ever reads
//     its own buffer, and DS operations of one wave execute in order;
//   * J_ij is the dot product of the row with the packed density (2 - delta_kl) D_kl;
//   * the row is the packed lower triangle of the symmetric matrix V^{ij}_{kl} = (ij|kl);
//     K[i,:] += V^{ij} D[:,j] and K[j,:] += V^{ij} D[:,i] are evaluated from LDS with lane <-> k,
//     so the 2x expansion packed -> square never touches HBM;
//   * K is accumulated per workgroup in LDS (ds_add_f64) and flushed once with global atomics.
#include "engine.hpp"
#include <cmath>
#include <cstdlib>

namespace mqc {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ void unpack_pair(int idx, int& k, int& l)
{
    k = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
    while ((k + 1) * (k + 2) / 2 <= idx) ++k;
    while (k * (k + 1) / 2 > idx) --k;
    l = idx - k * (k + 1) / 2;
}

__device__ __forceinline__ double readlane_f64(double x, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

// One wave's K work for the row held in `rowbuf` (packed lower triangle of V^{ij}).
// DREG (n <= 64): lane l holds D[i,l] and D[j,l] in registers and the loop broadcasts them with
// v_readlane; otherwise the two density rows are read through the pointers Di, Dj.
template <int KCH, bool DREG, bool FULL8>
__device__ __forceinline__ void row_exchange(const double* __restrict__ rowbuf, const double* __restrict__ Di,
                                             const double* __restrict__ Dj, int n, int lane, double* acc_i, double* acc_j)
{
#pragma unroll
    for (int c = 0; c < KCH; ++c) { acc_i[c] = 0.0; acc_j[c] = 0.0; }
    // DREG: Di, Dj are wave-uniform global rows (the caller made i, j scalar): D[i,l], D[j,l] arrive through
    // the scalar cache (s_load) and enter the FMAs as SGPR operands -- no VALU work, no v_readlane.
    // l runs in blocks of U: the U LDS reads of a block are issued back to back and only then consumed
    // (the compiler leaves a rolled loop with one read and a full lgkmcnt(0) wait per l otherwise, and the
    // kernel spends its time on LDS latency instead of streaming).  Lanes beyond n read a valid (clamped)
    // address and are discarded at the flush; l beyond n reads element (n-1, .) and is weighted by zero.
    constexpr int U = 8;
    int kk[KCH], tri[KCH];
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        const int k = lane + 64 * c;
        kk[c] = k < n ? k : n - 1;
        tri[c] = kk[c] * (kk[c] + 1) / 2;
    }
    for (int l0 = 0; l0 < n; l0 += U) {
        double v[U][KCH];
        const bool full = FULL8 || l0 + U <= n;                 // uniform; FULL8: n is a multiple of 8, no tail code at all
        if (full) {
            // scalar bookkeeping kept to one add per l: tri(l+1) = tri(l) + l + 1
            int lbase = l0 * (l0 + 1) / 2;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int l = l0 + u;
#pragma unroll
                for (int c = 0; c < KCH; ++c) {
                    const int idx = kk[c] >= l ? tri[c] + l : lbase + kk[c];
                    v[u][c] = rowbuf[idx];
                }
                lbase += l + 1;
            }
        } else if constexpr (!FULL8) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int l = (l0 + u < n) ? l0 + u : n - 1;       // uniform
                const int lbase = l * (l + 1) / 2;                  // uniform
#pragma unroll
                for (int c = 0; c < KCH; ++c) {
                    const int idx = kk[c] >= l ? tri[c] + l : lbase + kk[c];
                    v[u][c] = rowbuf[idx];
                }
            }
        }
        double dil[U], djl[U];
        if constexpr (DREG) {
            // constant address space + uniform address = s_load (one s_load_dwordx16 per row and block):
            // D is read-only for the whole kernel and the scalar cache is invalidated at kernel start
            typedef double double8 __attribute__((ext_vector_type(8)));
            typedef const double8 __attribute__((address_space(4))) * scalar_ptr8;
            typedef const double __attribute__((address_space(4))) * scalar_ptr;
            if (full) {
                const double8 a = *(scalar_ptr8)(Di + l0), b = *(scalar_ptr8)(Dj + l0);
#pragma unroll
                for (int u = 0; u < U; ++u) { dil[u] = a[u]; djl[u] = b[u]; }
            } else if constexpr (!FULL8) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool live = l0 + u < n;
                    const int lr = live ? l0 + u : n - 1;
                    dil[u] = live ? ((scalar_ptr)Di)[lr] : 0.0;
                    djl[u] = live ? ((scalar_ptr)Dj)[lr] : 0.0;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool live = l0 + u < n;
                const int lr = live ? l0 + u : n - 1;
                dil[u] = live ? Di[lr] : 0.0;
                djl[u] = live ? Dj[lr] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                acc_i[c] += v[u][c] * djl[u];
                acc_j[c] += v[u][c] * dil[u];
            }
        }
    }
}

// MAXU > 0: register-prefetch variant, a row is at most MAXU*64 doubles.
// MAXU == 0: generic variant, the row is loaded in chunks of 16 x 64.
// KCH = ceil(n / 64).  KLDS: the K accumulator lives in LDS; DREG: density rows through registers
// (n <= 64), else a copy of D sits in LDS next to K (KLDS) or is read from global memory.
template <int KCH, bool KLDS, int NW, int MAXU, bool DREG, bool FULL8 = false>
__global__ void __launch_bounds__(64 * NW) jk_incore_kernel(BatchView bv, int only_active)
{
    extern __shared__ double lds[];
    const int f = blockIdx.y;
    if ((only_active & 1) && bv.istate[4 * f] == ST_DONE) return;
    const int n = bv.n, np = bv.npair;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NTH = 64 * NW;
    const double* __restrict__ Dg = bv.D + (size_t)f * n * n;
    double* __restrict__ Jg = bv.J + (size_t)f * n * n;
    double* __restrict__ Kg = bv.K + (size_t)f * n * n;
    const double* __restrict__ M = bv.eri + (size_t)f * np * np;

    // FULL8 (the tuned dimer variant): row buffers padded to MAXU*64 so that staging needs no bounds test,
    // and the packed density this lane multiplies with sits in registers for the whole workgroup lifetime
    constexpr bool TUNED = FULL8 && MAXU > 0;
    constexpr int MAXU2 = (MAXU + 1) / 2;             // TUNED: 16-byte loads, chunks of 128 doubles (np is even)
    const int rs = TUNED ? MAXU2 * 128 : np;
    double* rowbuf = lds + (size_t)wave * rs;        // NW private row buffers
    double* Dp = lds + (size_t)NW * rs;               // packed (2 - delta) D
    double* Dl = Dp + np;                             // n*n (KLDS && !DREG)
    double* Kl = Dl + ((KLDS && !DREG) ? n * n : 0);  // n*n (KLDS)

    for (int idx = tid; idx < np; idx += NTH) {
        int k, l;
        unpack_pair(idx, k, l);
        const double d = Dg[k * n + l];
        Dp[idx] = (k == l) ? d : 2.0 * d;
    }
    if (KLDS) {
        for (int idx = tid; idx < n * n; idx += NTH) { if (!DREG) Dl[idx] = Dg[idx]; Kl[idx] = 0.0; }
    }
    const double* Dsrc = (KLDS && !DREG) ? Dl : Dg;
    __syncthreads();

    const int stride = gridDim.x * NW;
    int row = blockIdx.x * NW + wave;

    if constexpr (MAXU > 0) {
        typedef double double2v __attribute__((ext_vector_type(2)));
        double v[TUNED ? 1 : MAXU];
        double2v v2[TUNED ? MAXU2 : 1], dp2[TUNED ? MAXU2 : 1];
        auto load_row = [&](int r) {
            const double* __restrict__ src = M + (size_t)r * np;
            if constexpr (TUNED) {
#pragma unroll
                for (int u = 0; u < MAXU2; ++u) {
                    const int idx = 2 * lane + 128 * u;
                    v2[u] = idx < np ? *(const double2v*)(src + idx) : (double2v){0.0, 0.0};
                }
            } else {
#pragma unroll
                for (int u = 0; u < MAXU; ++u) { const int idx = lane + 64 * u; v[u] = idx < np ? src[idx] : 0.0; }
            }
        };
        if constexpr (TUNED) {
#pragma unroll
            for (int u = 0; u < MAXU2; ++u) {
                const int idx = 2 * lane + 128 * u;
                dp2[u] = idx < np ? *(const double2v*)(Dp + idx) : (double2v){0.0, 0.0};
            }
        }
        if (row < np) load_row(row);
        while (row < np) {
            int i, j;
            unpack_pair(row, i, j);
            double accj = 0.0;
            if constexpr (TUNED) {
#pragma unroll
                for (int u = 0; u < MAXU2; ++u) {
                    *(double2v*)(rowbuf + 2 * lane + 128 * u) = v2[u];
                    accj += v2[u][0] * dp2[u][0] + v2[u][1] * dp2[u][1];
                }
            } else {
#pragma unroll
                for (int u = 0; u < MAXU; ++u) {
                    const int idx = lane + 64 * u;
                    if (idx < np) { rowbuf[idx] = v[u]; accj += v[u] * Dp[idx]; }
                }
            }
            // issue the next row's loads now; they complete while this row is being contracted
            const int nrow = row + stride;
            if (nrow < np) load_row(nrow);
            accj = wave_sum(accj);
            if (lane == 0) { Jg[i * n + j] = accj; Jg[j * n + i] = accj; }
            double acc_i[KCH], acc_j[KCH];
            if (only_active & 2) {       // measurement switch (MQC_HIP_JK_SKIP_EXCHANGE=1): stream + J only, K is wrong
#pragma unroll
                for (int c = 0; c < KCH; ++c) { acc_i[c] = 0.0; acc_j[c] = 0.0; }
            } else {
                const int iu = __builtin_amdgcn_readfirstlane(i), ju = __builtin_amdgcn_readfirstlane(j);
                row_exchange<KCH, DREG, FULL8>(rowbuf, Dsrc + iu * n, Dsrc + ju * n, n, lane, acc_i, acc_j);
            }
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const int k = lane + 64 * c;
                if (k < n) {
                    double* Kt = KLDS ? Kl : Kg;
                    atomicAdd(&Kt[i * n + k], acc_i[c]);
                    if (i != j) atomicAdd(&Kt[j * n + k], acc_j[c]);
                }
            }
            row = nrow;
        }
    } else {
        constexpr int CH = 16;
        for (; row < np; row += stride) {
            int i, j;
            unpack_pair(row, i, j);
            const double* __restrict__ src = M + (size_t)row * np;
            double accj = 0.0;
            for (int base = 0; base < np; base += 64 * CH) {
                double v[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) { const int idx = base + lane + 64 * u; v[u] = idx < np ? src[idx] : 0.0; }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int idx = base + lane + 64 * u;
                    if (idx < np) { rowbuf[idx] = v[u]; accj += v[u] * Dp[idx]; }
                }
            }
            accj = wave_sum(accj);
            if (lane == 0) { Jg[i * n + j] = accj; Jg[j * n + i] = accj; }
            double acc_i[KCH], acc_j[KCH];
            row_exchange<KCH, DREG, false>(rowbuf, Dsrc + i * n, Dsrc + j * n, n, lane, acc_i, acc_j);
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const int k = lane + 64 * c;
                if (k < n) {
                    double* Kt = KLDS ? Kl : Kg;
                    atomicAdd(&Kt[i * n + k], acc_i[c]);
                    if (i != j) atomicAdd(&Kt[j * n + k], acc_j[c]);
                }
            }
        }
    }
    if (KLDS) {
        __syncthreads();
        for (int idx = tid; idx < n * n; idx += NTH) {
            const double kv = Kl[idx];
            if (kv != 0.0) atomicAdd(&Kg[idx], kv);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Long rows (64 < n <= ~88: def2-TZVP water dimer, n = 86, rows of 30 KB): a wave-private row buffer per wave no longer
// fits more than four waves per CU and the stream starves (32 KB in flight per CU, ~1 TB/s).  Here the WHOLE workgroup
// (eight waves) owns one row at a time: every thread keeps its slice of the next three rows in flight in registers
// (90 KB per CU), the current row sits in one of two LDS buffers, J is a block reduction, and the exchange mat-vecs
// split the l range over the waves (blocks of eight l, wave w takes blocks w, w + 8, ...) with lane <-> k and LDS
// atomics into the workgroup's K accumulator.  One barrier per row.
constexpr int JKC_NW = 8, JKC_NT = 64 * JKC_NW;

template <int KCH, int RPT>       // RPT = doubles per thread per row = ceil(np / 512)
__global__ void __launch_bounds__(JKC_NT) jk_rowcoop_kernel(BatchView bv, int only_active)
{
    extern __shared__ double lds[];
    const int f = blockIdx.y;
    if ((only_active & 1) && bv.istate[4 * f] == ST_DONE) return;
    const int n = bv.n, np = bv.npair;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NPAD = RPT * JKC_NT;
    const double* __restrict__ Dg = bv.D + (size_t)f * n * n;
    double* __restrict__ Jg = bv.J + (size_t)f * n * n;
    double* __restrict__ Kg = bv.K + (size_t)f * n * n;
    const double* __restrict__ M = bv.eri + (size_t)f * np * np;
    double* rowbuf = lds;                       // [2][NPAD]
    double* Dp = rowbuf + 2 * NPAD;             // packed (2 - delta) D, padded with zeros to NPAD
    double* Kl = Dp + NPAD;                     // n * n
    double* jred = Kl + (size_t)n * n;          // [2][JKC_NW]
    for (int idx = tid; idx < NPAD; idx += JKC_NT) {
        double d = 0.0;
        if (idx < np) { int k, l; unpack_pair(idx, k, l); d = Dg[k * n + l]; if (k != l) d *= 2.0; }
        Dp[idx] = d;
    }
    for (int idx = tid; idx < n * n; idx += JKC_NT) Kl[idx] = 0.0;
    __syncthreads();
    double dp[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) dp[u] = Dp[tid + JKC_NT * u];

    const int stride = gridDim.x;
    int row = blockIdx.x;
    double va[RPT], vb[RPT], vc[RPT];
    auto load_row = [&](int r, double* v) {
        const double* __restrict__ src = M + (size_t)r * np;
#pragma unroll
        for (int u = 0; u < RPT; ++u) { const int idx = tid + JKC_NT * u; v[u] = (r < np && idx < np) ? src[idx] : 0.0; }
    };
    load_row(row, va);
    load_row(row + stride, vb);
    load_row(row + 2 * stride, vc);
    int t = 0, prev_i = 0, prev_j = 0;
    bool have_prev = false;
    while (row < np) {
        double* buf = rowbuf + (t & 1) * NPAD;
        double accj = 0.0;
#pragma unroll
        for (int u = 0; u < RPT; ++u) { buf[tid + JKC_NT * u] = va[u]; accj += va[u] * dp[u]; }
        // rotate the register ring and put the row after next in flight
#pragma unroll
        for (int u = 0; u < RPT; ++u) { va[u] = vb[u]; vb[u] = vc[u]; }
        load_row(row + 3 * stride, vc);
        accj = wave_sum(accj);
        if (lane == 0) jred[(t & 1) * JKC_NW + wave] = accj;
        __syncthreads();
        int i, j;
        unpack_pair(row, i, j);
        if (tid == 0) {
            // this row's J from the partial sums just published; the barrier above ordered them
            double sj = 0.0;
#pragma unroll
            for (int w = 0; w < JKC_NW; ++w) sj += jred[(t & 1) * JKC_NW + w];
            Jg[i * n + j] = sj; Jg[j * n + i] = sj;
        }
        (void)have_prev; (void)prev_i; (void)prev_j;
        if (!(only_active & 2)) {
            const int iu = __builtin_amdgcn_readfirstlane(i), ju = __builtin_amdgcn_readfirstlane(j);
            const double* __restrict__ Di = Dg + (size_t)iu * n;
            const double* __restrict__ Dj = Dg + (size_t)ju * n;
            double acc_i[KCH], acc_j[KCH];
            int kk[KCH], tri[KCH];
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                acc_i[c] = 0.0; acc_j[c] = 0.0;
                const int k = lane + 64 * c;
                kk[c] = k < n ? k : n - 1;
                tri[c] = kk[c] * (kk[c] + 1) / 2;
            }
            typedef const double __attribute__((address_space(4))) * scalar_ptr;
            for (int l0 = 8 * wave; l0 < n; l0 += 8 * JKC_NW) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool live = l0 + u < n;                  // uniform
                    const int l = live ? l0 + u : n - 1;
                    const int lbase = l * (l + 1) / 2;
                    const double dil = live ? ((scalar_ptr)Di)[l] : 0.0, djl = live ? ((scalar_ptr)Dj)[l] : 0.0;
#pragma unroll
                    for (int c = 0; c < KCH; ++c) {
                        const double v = buf[kk[c] >= l ? tri[c] + l : lbase + kk[c]];
                        acc_i[c] += v * djl;
                        acc_j[c] += v * dil;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const int k = lane + 64 * c;
                if (k < n) {
                    atomicAdd(&Kl[i * n + k], acc_i[c]);
                    if (i != j) atomicAdd(&Kl[j * n + k], acc_j[c]);
                }
            }
        }
        row += stride;
        ++t;
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += JKC_NT) {
        const double kv = Kl[idx];
        if (kv != 0.0) atomicAdd(&Kg[idx], kv);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Triangular tensor (BatchView::eri_tri): only the elements col <= row of the symmetric pair matrix are stored,
// T[f][row (row + 1) / 2 + col] -- half the bytes of the square per fragment and iteration, half the zero fill, half
// the copy of shared blocks.  Every stored element plays both of its roles in ONE visit:
//   J[row] += v D'[col]  (the dot product along the row, as before)   and   J[col] += v D'[row]  for col < row;
//   K: a row is still the packed lower triangle of V^{ij}, cut off after (k, l) = (i, j).  The two symmetric mat-vecs
//      run over what is there and give Kh; K = Kh + Kh^T at the flush -- the transposed role of an element
//      contributes the transpose of what its stored role contributes, because D is symmetric -- and the diagonal
//      element (ij|ij), which is its own transpose, enters Kh with weight 1/2.
// Rows of a triangle hold 1 ... npair elements, so a wave takes them in PAIRS of constant length: row npair - 1 - t
// (long) with row t (short), npair + 1 elements, the same 9.4 KB per step as one row of the square (n = 48).  Lane <->
// stream position p = lane + 64 u: the long row first (its column IS p, so a lane's scattered J sums live in
// registers), then the short row (column p - length of the long row: density from LDS, scattered J
// through LDS atomics, a quarter of the elements on average).  In the wave's LDS buffer each row is completed with
// zeros to the end of its last shell row, so that the mat-vec loops need no masks: [long triangle, padded | short
// triangle, padded], at most jk_tri_buffer(npair) numbers.
__device__ __forceinline__ void row_exchange_tri(const double* __restrict__ buf, const double* __restrict__ Di,
                                                 const double* __restrict__ Dj, int i, int lane, double& acc_i, double& acc_j)
{
    typedef double double8 __attribute__((ext_vector_type(8)));
    typedef const double8 __attribute__((address_space(4))) * scalar_ptr8;
    acc_i = 0.0; acc_j = 0.0;
    const int kk = lane <= i ? lane : i;          // lanes beyond the triangle read lane i's elements and are dropped at the flush
    const int trik = kk * (kk + 1) / 2;
    // full blocks of eight l: no clamps, one scalar add per l (tri(l + 1) = tri(l) + l + 1), as in row_exchange
    int l0 = 0, lbase = 0;
    for (; l0 + 8 <= i + 1; l0 += 8) {             // i is wave-uniform; n is a multiple of 8, so l0 + 8 <= n
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int l = l0 + u;
            v[u] = buf[kk >= l ? trik + l : lbase + kk];
            lbase += l + 1;
        }
        const double8 a = *(scalar_ptr8)(Di + l0), b = *(scalar_ptr8)(Dj + l0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc_i += v[u] * b[u];
            acc_j += v[u] * a[u];
        }
    }
    if (l0 <= i) {                                  // the last, partial block: l beyond i re-reads l = i with weight zero
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int l = (l0 + u <= i) ? l0 + u : i;                  // uniform
            const int lb = l * (l + 1) / 2;
            v[u] = buf[kk >= l ? trik + l : lb + kk];
        }
        const double8 a = *(scalar_ptr8)(Di + l0), b = *(scalar_ptr8)(Dj + l0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool live = l0 + u <= i;
            acc_i += v[u] * (live ? b[u] : 0.0);
            acc_j += v[u] * (live ? a[u] : 0.0);
        }
    }
}

template <int NW, int MAXU>
__global__ void __launch_bounds__(64 * NW) jk_tri_kernel(BatchView bv, int only_active, int rs)
{
    extern __shared__ double lds[];
    const int f = blockIdx.y;
    if ((only_active & 1) && bv.istate[4 * f] == ST_DONE) return;
    const int n = bv.n, np = bv.npair;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NTH = 64 * NW;
    const double* __restrict__ Dg = bv.D + (size_t)f * n * n;
    double* __restrict__ Jg = bv.J + (size_t)f * n * n;
    double* __restrict__ Kg = bv.K + (size_t)f * n * n;
    const double* __restrict__ T = bv.eri + (size_t)f * bv.eri_stride;
    double* rowbuf = lds + (size_t)wave * rs;          // NW private buffers
    double* Dp = lds + (size_t)NW * rs;                 // packed (2 - delta) D
    double* Jl = Dp + np;                               // packed J of this workgroup
    double* Kl = Jl + np;                               // n * n: Kh of this workgroup

    for (int idx = tid; idx < np; idx += NTH) {
        int k, l;
        unpack_pair(idx, k, l);
        const double d = Dg[k * n + l];
        Dp[idx] = (k == l) ? d : 2.0 * d;
        Jl[idx] = 0.0;
    }
    for (int idx = tid; idx < n * n; idx += NTH) Kl[idx] = 0.0;
    __syncthreads();

    // (the density factors of the long row come from LDS as well: with them in registers next to the scattered sums
    // and the row in flight the twelve-wave workgroup spills)
    double jsc[MAXU], v[MAXU];
#pragma unroll
    for (int u = 0; u < MAXU; ++u) jsc[u] = 0.0;
    const int npairs = (np + 1) / 2;                    // odd npair: the middle row comes alone
    const int stride = gridDim.x * NW;
    int t = blockIdx.x * NW + wave;
    auto load_pair = [&](int tt) {
        const int rl = np - 1 - tt, llen = rl + 1, slen = (tt < rl) ? tt + 1 : 0;
        const double* __restrict__ srcl = T + (size_t)rl * (rl + 1) / 2;
        const double* __restrict__ srcs = T + (size_t)tt * (tt + 1) / 2;
#pragma unroll
        for (int u = 0; u < MAXU; ++u) {
            const int p = lane + 64 * u, q = p - llen;
            const double* a = p < llen ? srcl + p : srcs + (q < slen ? q : 0);
            v[u] = (p < llen + slen) ? *a : 0.0;
        }
    };
    if (t < npairs) load_pair(t);
    while (t < npairs) {
        const int rl = np - 1 - t, llen = rl + 1, slen = (t < rl) ? t + 1 : 0;
        int il, jl, is = 0, js = 0;
        unpack_pair(rl, il, jl);
        if (slen > 0) unpack_pair(t, is, js);
        il = __builtin_amdgcn_readfirstlane(il); jl = __builtin_amdgcn_readfirstlane(jl);
        is = __builtin_amdgcn_readfirstlane(is); js = __builtin_amdgcn_readfirstlane(js);
        const int sb = (il + 1) * (il + 2) / 2;         // the short triangle starts behind the padded long one
        const double dpl = Dp[rl], dps = Dp[t];
        // stage the pair (diagonal elements halved for the exchange), J along the rows, J scattered to the columns
        double accl = 0.0, accs = 0.0;
        const int lq = lane - llen;                     // column of a lane inside the short row, chunk 0
        double* const sbuf = rowbuf + sb;
#pragma unroll
        for (int u = 0; u < MAXU; ++u) {
            const int base = 64 * u;                    // the three cases below are wave-uniform (scalar branches)
            const double x = v[u];
            if (base + 64 <= llen - 1) {
                // the whole chunk inside the long row and off its diagonal element: no per-lane tests
                accl += x * Dp[lane + base];
                jsc[u] += x * dpl;
                rowbuf[lane + base] = x;
            } else if (base >= llen && base + 64 <= llen + slen - 1) {
                // the whole chunk inside the short row and off its diagonal element
                const int q = lq + base;
                accs += x * Dp[q];
                atomicAdd(&Jl[q], x * dps);
                sbuf[q] = x;
            } else if (base < llen + slen) {
                // a chunk with an end of a row in it (at most three per pair): the general form
                const int p = lane + base, q = lq + base;
                if (p < llen) {
                    accl += x * Dp[p];
                    jsc[u] += (p < llen - 1) ? x * dpl : 0.0;
                    rowbuf[p] = (p == llen - 1) ? 0.5 * x : x;
                } else if (q < slen) {
                    accs += x * Dp[q];
                    if (q < slen - 1) atomicAdd(&Jl[q], x * dps);
                    sbuf[q] = (q == slen - 1) ? 0.5 * x : x;
                }
            }
        }
        if (lane < sb - llen) rowbuf[llen + lane] = 0.0;                                  // rest of shell row il
        if (slen > 0 && lane < (is + 1) * (is + 2) / 2 - slen) rowbuf[sb + slen + lane] = 0.0;   // rest of shell row is
        // the next pair's loads go out now and complete while this pair is contracted
        const int nt = t + stride;
        if (nt < npairs) load_pair(nt);
        accl = wave_sum(accl);
        accs = wave_sum(accs);
        if (lane == 0) { atomicAdd(&Jl[rl], accl); if (slen > 0) atomicAdd(&Jl[t], accs); }
        double ai, aj;
        row_exchange_tri(rowbuf, Dg + il * n, Dg + jl * n, il, lane, ai, aj);
        if (lane <= il) {
            atomicAdd(&Kl[il * n + lane], ai);
            if (il != jl) atomicAdd(&Kl[jl * n + lane], aj);
        }
        if (slen > 0) {
            row_exchange_tri(rowbuf + sb, Dg + is * n, Dg + js * n, is, lane, ai, aj);
            if (lane <= is) {
                atomicAdd(&Kl[is * n + lane], ai);
                if (is != js) atomicAdd(&Kl[js * n + lane], aj);
            }
        }
        t = nt;
    }
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
        const int p = lane + 64 * u;
        if (p < np && jsc[u] != 0.0) atomicAdd(&Jl[p], jsc[u]);
    }
    __syncthreads();
    for (int idx = tid; idx < np; idx += NTH) {
        const double jv = Jl[idx];
        if (jv == 0.0) continue;
        int k, l;
        unpack_pair(idx, k, l);
        atomicAdd(&Jg[k * n + l], jv);
        if (k != l) atomicAdd(&Jg[l * n + k], jv);
    }
    for (int idx = tid; idx < n * n; idx += NTH) {
        const double kv = Kl[idx];
        if (kv == 0.0) continue;
        const int a = idx / n, b = idx - a * n;
        atomicAdd(&Kg[idx], kv);                      // K = Kh + Kh^T
        atomicAdd(&Kg[b * n + a], kv);
    }
}

// numbers a wave's LDS buffer must hold: the padded long triangle plus the padded short one, worst row pair
static int jk_tri_buffer(int np)
{
    auto shell_row = [](int idx) { int k = (int)((std::sqrt(8.0 * idx + 1.0) - 1.0) * 0.5); while ((k + 1) * (k + 2) / 2 <= idx) ++k; while (k * (k + 1) / 2 > idx) --k; return k; };
    int worst = 0;
    for (int t = 0; t < (np + 1) / 2; ++t) {
        const int rl = np - 1 - t, il = shell_row(rl), is = shell_row(t);
        const int need = (il + 1) * (il + 2) / 2 + (t < rl ? (is + 1) * (is + 2) / 2 : 0);
        if (need > worst) worst = need;
    }
    return (worst + 1) & ~1;
}

constexpr int JK_TRI_NW = 12, JK_TRI_MAXU = 19;
static size_t jk_tri_lds_bytes(int n, int np) { return sizeof(double) * ((size_t)JK_TRI_NW * jk_tri_buffer(np) + 2 * (size_t)np + (size_t)n * n); }

// The triangular layout is taken for the batches the tuned square kernel served: restricted, dimer-sized fragments
// (n <= 64, a multiple of 8; 640 < npair, npair + 1 <= 19 * 64) in batches of at least 64.  MQC_HIP_ERI_TRI=0: square.
bool jk_tri_layout(int n, int np, int nfrag, bool uhf)
{
    static const bool on = [] { const char* e = std::getenv("MQC_HIP_ERI_TRI"); return !(e && e[0] == '0'); }();
    if (!on || uhf || nfrag < 64 || n > 64 || n % 8 != 0 || np <= 10 * 64 || np + 1 > JK_TRI_MAXU * 64) return false;
    return jk_tri_lds_bytes(n, np) <= (size_t)160 * 1024 - 1024;
}

static int jk_grid_x(const BatchView& bv, int nw)
{
    // enough workgroups to cover 256 CUs several times over, but few enough that the per-workgroup
    // D load and K flush stay amortised over many rows
    int want = (4096 + bv.nfrag - 1) / bv.nfrag;
    int maxx = (bv.npair + nw - 1) / nw;
    if (want < 1) want = 1;
    if (want > maxx) want = maxx;
    return want;
}

template <int KCH, bool KLDS, int NW, int MAXU, bool DREG, bool FULL8 = false>
static void jk_launch(const BatchView& bv, int oa, size_t lds, hipStream_t s)
{
    auto kern = jk_incore_kernel<KCH, KLDS, NW, MAXU, DREG, FULL8>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(jk_grid_x(bv, NW), bv.nfrag), block(64 * NW);
    hipLaunchKernelGGL(kern, grid, block, lds, s, bv, oa);
}

static bool jk_rowcoop_on()
{
    static const bool on = [] { const char* e = std::getenv("MQC_HIP_JK_ROWCOOP"); return !(e && e[0] == '0'); }();
    return on;
}

void launch_jk_incore(const BatchView& bv, bool only_active, hipStream_t s)
{
    const int n = bv.n, np = bv.npair;
    (void)hipMemsetAsync(bv.K, 0, sizeof(double) * (size_t)bv.nfrag * n * n, s);
    if (bv.eri_tri) {
        // J is summed over the workgroups of a fragment here (row and column roles), so it starts from zero like K
        (void)hipMemsetAsync(bv.J, 0, sizeof(double) * (size_t)bv.nfrag * n * n, s);
        const size_t lds = jk_tri_lds_bytes(n, np);
        auto kern = jk_tri_kernel<JK_TRI_NW, JK_TRI_MAXU>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int gx = (4096 + bv.nfrag - 1) / bv.nfrag;
        const int maxx = ((np + 1) / 2 + JK_TRI_NW - 1) / JK_TRI_NW;
        if (gx < 1) gx = 1;
        if (gx > maxx) gx = maxx;
        hipLaunchKernelGGL(kern, dim3(gx, bv.nfrag), dim3(64 * JK_TRI_NW), lds, s, bv, only_active ? 1 : 0, jk_tri_buffer(np));
        return;
    }
    const int kch = (n + 63) / 64;
    static const int skip_exchange = [] { const char* e = std::getenv("MQC_HIP_JK_SKIP_EXCHANGE"); return (e && e[0] == '1') ? 2 : 0; }();
    const int oa = (only_active ? 1 : 0) | skip_exchange;
    const size_t LDS_MAX = 160 * 1024 - 1024;
    const size_t lds_reg = sizeof(double) * ((size_t)5 * np + (size_t)n * n);      // 4 waves, K in LDS, D in registers
    const size_t lds4k = sizeof(double) * ((size_t)5 * np + 2 * (size_t)n * n);   // 4 waves, D and K in LDS
    const size_t lds4 = sizeof(double) * ((size_t)5 * np);                        // 4 waves, D and K global
    const size_t lds2 = sizeof(double) * ((size_t)3 * np);                        // 2 waves, D and K global
    const size_t lds_reg12 = sizeof(double) * ((size_t)13 * np + (size_t)n * n);   // 12 waves share D' and the K accumulator
    if (n <= 64 && np <= 19 * 64 && np > 5 * 64 && lds_reg12 <= LDS_MAX && bv.nfrag >= 64) {
        // dimer-sized fragments in a large batch: one 12-wave workgroup per CU (141 KB of LDS for n = 48)
        // keeps 12 x 9.4 KB of row loads in flight per CU instead of 8 x
        if (np <= 10 * 64) jk_launch<1, true, 12, 10, true>(bv, oa, lds_reg12, s);
        else if (n % 8 == 0 && np % 2 == 0 && sizeof(double) * ((size_t)12 * 10 * 128 + np + (size_t)n * n) <= LDS_MAX)
            jk_launch<1, true, 12, 19, true, true>(bv, oa, sizeof(double) * ((size_t)12 * 10 * 128 + np + (size_t)n * n), s);
        else jk_launch<1, true, 12, 19, true>(bv, oa, lds_reg12, s);
    } else if (n <= 64 && lds_reg <= LDS_MAX) {
        // the fragment sizes of an MBE run (n = 48: 65.5 KB -> two workgroups per CU)
        if (np <= 5 * 64) jk_launch<1, true, 4, 5, true>(bv, oa, lds_reg, s);
        else if (np <= 10 * 64) jk_launch<1, true, 4, 10, true>(bv, oa, lds_reg, s);
        else if (np <= 19 * 64) jk_launch<1, true, 4, 19, true>(bv, oa, lds_reg, s);
        else jk_launch<1, true, 4, 0, true>(bv, oa, lds_reg, s);
    } else if (lds4k <= LDS_MAX && kch <= 2) {
        jk_launch<2, true, 4, 0, false>(bv, oa, lds4k, s);
    } else if (kch == 2 && np <= 8 * JKC_NT && sizeof(double) * ((size_t)3 * 8 * JKC_NT + (size_t)n * n + 2 * JKC_NW) <= LDS_MAX && jk_rowcoop_on()) {
        // 64 < n <= 89: rows up to 32 KB, the workgroup-cooperative kernel
        const size_t lds = sizeof(double) * ((size_t)3 * 8 * JKC_NT + (size_t)n * n + 2 * JKC_NW);
        auto kern = jk_rowcoop_kernel<2, 8>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int gx = (2048 + bv.nfrag - 1) / bv.nfrag;
        if (gx < 1) gx = 1;
        if (gx > np) gx = np;
        hipLaunchKernelGGL(kern, dim3(gx, bv.nfrag), dim3(JKC_NT), lds, s, bv, oa);
    } else if (lds4 <= LDS_MAX) {
        if (kch <= 2) jk_launch<2, false, 4, 0, false>(bv, oa, lds4, s);
        else if (kch == 3) jk_launch<3, false, 4, 0, false>(bv, oa, lds4, s);
        else jk_launch<4, false, 4, 0, false>(bv, oa, lds4, s);
    } else {
        if (kch <= 2) jk_launch<2, false, 2, 0, false>(bv, oa, lds2, s);
        else if (kch == 3) jk_launch<3, false, 2, 0, false>(bv, oa, lds2, s);
        else jk_launch<4, false, 2, 0, false>(bv, oa, lds2, s);
    }
}

}  // namespace mqc
