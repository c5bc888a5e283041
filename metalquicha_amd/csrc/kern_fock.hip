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
#include <vector>
#include <cstdlib>

namespace mqc {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Wave sum through the DPP lane network: quad swaps, half-row and row mirrors, then the two row broadcasts -- 18 VALU
// instructions and no LDS round trip (a shuffle is two ds_bpermute_b32 per step and a wait for each).  The total is
// valid in lanes 48..63 only.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_take(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_top(double v)
{
    v += dpp_take<0xB1, 0xf>(v);        // quad_perm [1, 0, 3, 2]
    v += dpp_take<0x4E, 0xf>(v);        // quad_perm [2, 3, 0, 1]
    v += dpp_take<0x141, 0xf>(v);       // row_half_mirror
    v += dpp_take<0x140, 0xf>(v);       // row_mirror: every lane of a row of 16 holds the row's sum
    v += dpp_take<0x142, 0xa>(v);       // row_bcast15 into rows 1 and 3
    v += dpp_take<0x143, 0xc>(v);       // row_bcast31 into rows 2 and 3: row 3 holds the wave's sum
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
                    const int ia = tri[c] + l, ib = lbase + kk[c];
                    const int idx = ia > ib ? ia : ib;      // tri(max) + min is the larger of the two forms (see row_exchange_tri)
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
                    const int ia = tri[c] + l, ib = lbase + kk[c];
                    const int idx = ia > ib ? ia : ib;      // tri(max) + min is the larger of the two forms (see row_exchange_tri)
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
                        const int ia = tri[c] + l, ib = lbase + kk[c];
                        const double v = buf[ia > ib ? ia : ib];      // tri(max) + min is the larger of the two forms
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
// Triangular tensor (BatchView::eri_tri): only the elements col <= row of the symmetric pair matrix are stored -- half
// the bytes of the square per fragment and iteration, half the zero fill, half the copy of shared blocks.  Every stored
// element plays both of its roles in ONE visit:
//   J[row] += v D'[col]  (the dot product along the row, as before)   and   J[col] += v D'[row];
//   K: a row is still the packed lower triangle of V^{ij}, cut off after (k, l) = (i, j).  The two symmetric mat-vecs
//      run over what is there and give Kh; K = Kh + Kh^T at the flush -- the transposed role of an element
//      contributes the transpose of what its stored role contributes, because D is symmetric.
//   The diagonal element (ij|ij) is its own transpose: the integral kernels store it HALVED (PairStore::put), and with
//   that it needs no special case here, neither in J (row role + column role = the whole) nor in K.
// Rows of a triangle hold 1 ... npair elements, so they are stored and streamed in PAIRS of constant length: block t
// = [row npair - 1 - t | row t], each completed with zeros to the end of its last shell row (the mat-vec loops then
// need no masks; the zeros come from the zero fill of the tensor), eri_tri_pb doubles per block -- 1240 for n = 48
// against 1176 for a row of the square.  A wave streams a block exactly as the square kernel streams a row (16-byte
// loads, the next block in flight in registers, the block parked in the wave's LDS buffer as it lies in HBM).
__device__ __forceinline__ void row_exchange_tri(const double* __restrict__ buf, const double* __restrict__ Di,
                                                 const double* __restrict__ Dj, int i, int lane, double& acc_i, double& acc_j)
{
    typedef double double8 __attribute__((ext_vector_type(8)));
    typedef const double8 __attribute__((address_space(4))) * scalar_ptr8;
    acc_i = 0.0; acc_j = 0.0;
    const int kk = lane <= i ? lane : i;          // lanes beyond the triangle read lane i's elements and are dropped at the flush
    const int trik = kk * (kk + 1) / 2;
    // full blocks of eight l: no clamps, one scalar add per l (tri(l + 1) = tri(l) + l + 1), as in row_exchange.
    // Element (k, l) of the packed triangle sits at tri(max) + min = max(tri(k) + l, tri(l) + k) -- the form that is not
    // the valid one is the smaller of the two, (k - l)(k + l + 1) / 2 >= k - l -- so the index needs no compare and
    // select; with u taken out of both forms the read is [max(tri(k) + l0, tri(l) + k - u)] + u, u in the instruction's
    // offset field: two integer instructions per element instead of five (the loop is bound by instruction issue)
    // (all of it in LDS byte addresses, so that no shift is left either)
    typedef const double __attribute__((address_space(3))) * lds_cptr;
    const unsigned base = (unsigned)(size_t)(lds_cptr)buf;
    unsigned rowbase = base + 8u * (unsigned)trik, colbase = base + 8u * (unsigned)kk;
    asm volatile("" : "+v"(rowbase), "+v"(colbase));      // opaque: otherwise the compiler re-associates the shift back into the loop
    int l0 = 0, lbase = 0;
    for (; l0 + 8 <= i + 1; l0 += 8) {             // i is wave-uniform; n is a multiple of 8, so l0 + 8 <= n
        double v[8];
        const unsigned rowaddr = rowbase + 8u * (unsigned)l0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int l = l0 + u;
            const unsigned coladdr = colbase + 8u * (unsigned)(lbase - u);
            const unsigned sel = rowaddr > coladdr ? rowaddr : coladdr;
            v[u] = *(lds_cptr)(size_t)(sel + 8u * (unsigned)u);
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
            // scalar on purpose (readfirstlane): otherwise the compiler builds the two index forms under exec masks
            const int l = __builtin_amdgcn_readfirstlane((l0 + u <= i) ? l0 + u : i);
            const int lb = __builtin_amdgcn_readfirstlane(l * (l + 1) / 2);
            const unsigned ra = rowbase + 8u * (unsigned)l, ca = colbase + 8u * (unsigned)lb;
            v[u] = *(lds_cptr)(size_t)(ra > ca ? ra : ca);
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

template <int NW, int MAXU2>       // MAXU2 chunks of 128 doubles cover a block
__global__ void __launch_bounds__(64 * NW) jk_tri_kernel(BatchView bv, int only_active)
{
    extern __shared__ double lds[];
    typedef double double2v __attribute__((ext_vector_type(2)));
    const int f = blockIdx.y;
    if ((only_active & 1) && bv.istate[4 * f] == ST_DONE) return;
    const int n = bv.n, np = bv.npair, pb = bv.eri_tri_pb;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NTH = 64 * NW;
    const double* __restrict__ Dg = bv.D + (size_t)f * n * n;
    double* __restrict__ Jg = bv.J + (size_t)f * n * n;
    double* __restrict__ Kg = bv.K + (size_t)f * n * n;
    const double* __restrict__ T = bv.eri + (size_t)f * bv.eri_stride;
    double* rowbuf = lds + (size_t)wave * pb;          // NW private buffers, one block each
    double* Dp = lds + (size_t)NW * pb;                 // packed (2 - delta) D, two zeros behind it
    double* Jl = Dp + np + 2;                           // packed J of this workgroup (+ two slots the padding may touch)
    double* Kl = Jl + np + 2;                           // n * n: Kh of this workgroup
    unsigned char* rz = (unsigned char*)(Kl + (size_t)n * n);          // [np] 1 = pair row known to be all zeros
    int* ao2sh = (int*)(rz + ((np + 15) & ~15));                        // [64] shell of every function
    double* wred = (double*)(ao2sh + 64);                               // [NW]

    // Zero rows.  A screened build leaves a pair row (ij) untouched -- all zeros from the fill -- when Q_ij times the
    // fragment's largest bound is below the threshold: every quartet on that bra pair was dropped.  For the far-apart
    // dimers of an MBE list that is every row with i and j on different monomers, half of the tensor; such rows are
    // neither loaded nor contracted.  (Twin blocks are kept or dropped as a whole, so a row this test calls zero can
    // hold integrals below the threshold: dropping them is what the threshold means.)
    const bool screen = bv.jk_q != nullptr;
    if (screen) {
        const TopologyDev& tp = bv.topo;
        const int ns = tp.nshell;
        const double* __restrict__ Qf = bv.jk_q + (size_t)f * ns * ns;
        for (int sh = tid; sh < ns; sh += NTH) {
            const int o = tp.sh_aoff[sh], nfun = 2 * tp.sh_l[sh] + 1;
            for (int m = 0; m < nfun; ++m) ao2sh[o + m] = sh;
        }
        double qm = 0.0;
        for (int idx = tid; idx < ns * ns; idx += NTH) qm = fmax(qm, Qf[idx]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) qm = fmax(qm, __shfl_xor(qm, off, 64));
        if (lane == 0) wred[wave] = qm;
        __syncthreads();
        double qmax = 0.0;
        for (int w = 0; w < NW; ++w) qmax = fmax(qmax, wred[w]);
        for (int idx = tid; idx < np; idx += NTH) {
            int k, l;
            unpack_pair(idx, k, l);
            rz[idx] = (Qf[ao2sh[k] * ns + ao2sh[l]] * qmax < bv.jk_qthresh) ? 1 : 0;
        }
    } else {
        for (int idx = tid; idx < np; idx += NTH) rz[idx] = 0;
    }

    for (int idx = tid; idx < np + 2; idx += NTH) {
        double d = 0.0;
        if (idx < np) {
            int k, l;
            unpack_pair(idx, k, l);
            d = Dg[k * n + l];
            if (k != l) d *= 2.0;
        }
        Dp[idx] = d;
        Jl[idx] = 0.0;
    }
    for (int idx = tid; idx < n * n; idx += NTH) Kl[idx] = 0.0;
    // slot 0 of the function -> shell map doubles as the workgroup's count of chunks read: function 0 sits in shell 0, so
    // the slot already holds 0 where the map was built, and is set to 0 where it was not
    if (tid == 0) ao2sh[0] = 0;
    __syncthreads();

    double2v v2[MAXU2], jsc2[MAXU2];
#pragma unroll
    for (int u = 0; u < MAXU2; ++u) jsc2[u] = (double2v){0.0, 0.0};
    const int npairs = (np + 1) / 2;                    // odd npair: the middle row has a block to itself
    const int stride = gridDim.x * NW;
    int t = blockIdx.x * NW + wave;
    const int* __restrict__ shell_row = bv.eri_tri_sb + npairs;
    int chunks_read = 0;          // wave-uniform: chunks of 128 doubles this wave asked HBM for
    auto load_block = [&](int tt) {
        const double* __restrict__ src = T + (size_t)tt * pb;
        // chunks that hold nothing but rows known to be zero are not read
        const int rln = np - 1 - tt;
        const bool zln = __builtin_amdgcn_readfirstlane((int)rz[rln]) != 0;
        const bool zsn = (tt < rln) ? (__builtin_amdgcn_readfirstlane((int)rz[tt]) != 0) : true;
        const int iln = __builtin_amdgcn_readfirstlane(shell_row[rln]), isn = __builtin_amdgcn_readfirstlane(shell_row[tt]);
        const int sbn = (iln + 1) * (iln + 2) / 2, sen = (tt < rln) ? sbn + (isn + 1) * (isn + 2) / 2 : sbn;
        // first and one-past-last chunk to read (wave-uniform, scalar): [0, ceil(sb / 128)) for the long row, [sb / 128,
        // ceil(se / 128)) for the short one; the union is one interval because the short row starts where the long ends
        int c0 = zln ? sbn >> 7 : 0, c1 = zsn ? (sbn + 127) >> 7 : (sen + 127) >> 7;
        if (zln && zsn) { c0 = 0; c1 = 0; }
        c0 = __builtin_amdgcn_readfirstlane(c0); c1 = __builtin_amdgcn_readfirstlane(c1);
        chunks_read += (c1 < MAXU2 ? c1 : MAXU2) - c0;
        const double* __restrict__ srcl = src + 2 * lane;
#pragma unroll
        for (int u = 0; u < MAXU2; ++u) {
            if (u >= c0 && u < c1) {          // scalar branch (kept one by the empty asm: as a select the load needs a second register set)
                asm volatile("" ::: "memory");
                v2[u] = (2 * lane + 128 * u < pb) ? *(const double2v*)(srcl + 128 * u) : (double2v){0.0, 0.0};
            } else {
                v2[u] = (double2v){0.0, 0.0};
            }
        }
    };
    if (t < npairs) load_block(t);
    while (t < npairs) {
        const int rl = np - 1 - t;
        const bool two = t < rl;                        // the block holds a short row too
        const bool zl = __builtin_amdgcn_readfirstlane((int)rz[rl]) != 0;
        const bool zs = two ? (__builtin_amdgcn_readfirstlane((int)rz[t]) != 0) : true;
        // shell rows of the two pair rows from the table behind eri_tri_sb (uniform addresses: scalar loads)
        const int il = __builtin_amdgcn_readfirstlane(shell_row[rl]), jl = rl - il * (il + 1) / 2;
        const int is = two ? __builtin_amdgcn_readfirstlane(shell_row[t]) : 0, js = t - is * (is + 1) / 2;
        const int sb = (il + 1) * (il + 2) / 2;         // = eri_tri_sb[t]: the short row starts behind the padded long one
        const int se = two ? sb + (is + 1) * (is + 2) / 2 : sb;   // end of the padded short row
        const double dpl = Dp[rl], dps = Dp[t];
        double accl = 0.0, accs = 0.0;
        // a block whose rows are both known to be zero was not loaded and is not staged (its registers hold zeros)
        if (!(zl && zs))
#pragma unroll
        for (int u = 0; u < MAXU2; ++u) {
            const int base = 128 * u, idx = 2 * lane + base;   // the cases below are wave-uniform
            const double2v x = v2[u];
            if (base + 128 <= pb) *(double2v*)(rowbuf + idx) = x;
            else if (idx < pb) *(double2v*)(rowbuf + idx) = x;
            if (base + 128 <= sb) {
                // the whole chunk inside the (padded) long row: column = position
                const double2v d = *(const double2v*)(Dp + idx);
                accl += x[0] * d[0] + x[1] * d[1];
                jsc2[u] += x * dpl;
            } else if (base >= sb) {
                if (base < se) {
                    // the whole chunk inside the (padded) short row: column = position - sb
                    const int q = idx - sb;
                    if (idx < se) {
                        accs += x[0] * Dp[q] + x[1] * Dp[q + 1];
                        atomicAdd(&Jl[q], x[0] * dps);
                        atomicAdd(&Jl[q + 1], x[1] * dps);
                    }
                }
            } else {
                // the chunk with the boundary in it
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int p = idx + e;
                    if (p < sb) {
                        accl += x[e] * Dp[p];
                        jsc2[u][e] += x[e] * dpl;
                    } else if (p < se) {
                        accs += x[e] * Dp[p - sb];
                        atomicAdd(&Jl[p - sb], x[e] * dps);
                    }
                }
            }
        }
        // the next block's loads go out now and complete while this one is contracted
        const int nt = t + stride;
        if (nt < npairs) load_block(nt);
        accl = wave_sum_top(accl);
        accs = wave_sum_top(accs);
        if (lane == 63) { if (!zl) atomicAdd(&Jl[rl], accl); if (!zs) atomicAdd(&Jl[t], accs); }
        double ai, aj;
        if (!zl) {
            row_exchange_tri(rowbuf, Dg + il * n, Dg + jl * n, il, lane, ai, aj);
            if (lane <= il) {
                atomicAdd(&Kl[il * n + lane], ai);
                if (il != jl) atomicAdd(&Kl[jl * n + lane], aj);
            }
        }
        if (!zs) {
            row_exchange_tri(rowbuf + sb, Dg + is * n, Dg + js * n, is, lane, ai, aj);
            if (lane <= is) {
                atomicAdd(&Kl[is * n + lane], ai);
                if (is != js) atomicAdd(&Kl[js * n + lane], aj);
            }
        }
        t = nt;
    }
#pragma unroll
    for (int u = 0; u < MAXU2; ++u) {
        const int idx = 2 * lane + 128 * u;
        if (idx < np && jsc2[u][0] != 0.0) atomicAdd(&Jl[idx], jsc2[u][0]);
        if (idx + 1 < np && jsc2[u][1] != 0.0) atomicAdd(&Jl[idx + 1], jsc2[u][1]);
    }
    if (lane == 0 && chunks_read > 0) atomicAdd((int*)&ao2sh[0], chunks_read);      // ao2sh is done with: slot 0 becomes the workgroup's count
    __syncthreads();
    if (tid == 0 && bv.jk_loaded) {
        if (gridDim.x == 1) bv.jk_loaded[f] = ao2sh[0];
        else atomicAdd(&bv.jk_loaded[f], ao2sh[0]);
    }
    if (gridDim.x == 1) {
        // one workgroup per fragment (large batches): J and K = Kh + Kh^T are complete here -- plain stores, and the
        // launcher zeroes nothing
        for (int idx = tid; idx < np; idx += NTH) {
            int k, l;
            unpack_pair(idx, k, l);
            const double jv = Jl[idx];
            Jg[k * n + l] = jv;
            Jg[l * n + k] = jv;
        }
        for (int idx = tid; idx < n * n; idx += NTH) {
            const int a = idx / n, b = idx - a * n;
            Kg[idx] = Kl[idx] + Kl[b * n + a];
        }
        return;
    }
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

// doubles per block: the padded long triangle plus the padded short one, worst row pair (even, so that blocks stay
// 16-byte aligned); optionally where the short row starts in every block
int jk_tri_block(int np, std::vector<int>* short_row_start)
{
    auto shell_row = [](int idx) { int k = (int)((std::sqrt(8.0 * idx + 1.0) - 1.0) * 0.5); while ((k + 1) * (k + 2) / 2 <= idx) ++k; while (k * (k + 1) / 2 > idx) --k; return k; };
    const int npairs = (np + 1) / 2;
    // table: [npairs] start of the short row in every block, then [np] the shell row i of every pair row (ij)
    if (short_row_start) {
        short_row_start->assign((size_t)npairs + np, 0);
        for (int r = 0; r < np; ++r) (*short_row_start)[(size_t)npairs + r] = shell_row(r);
    }
    int worst = 0;
    for (int t = 0; t < npairs; ++t) {
        const int rl = np - 1 - t, il = shell_row(rl), is = shell_row(t);
        const int sb = (il + 1) * (il + 2) / 2;
        const int need = sb + (t < rl ? (is + 1) * (is + 2) / 2 : 0);
        if (short_row_start) (*short_row_start)[t] = sb;
        if (need > worst) worst = need;
    }
    return (worst + 1) & ~1;
}

constexpr int JK_TRI_NW = 12, JK_TRI_MAXU2 = 10;
static size_t jk_tri_lds_bytes(int n, int np)
{
    // row buffers, packed D' and J (+ 2), Kh; zero-row flags, function -> shell map, per-wave maxima
    return sizeof(double) * ((size_t)JK_TRI_NW * jk_tri_block(np) + 2 * ((size_t)np + 2) + (size_t)n * n) + (((size_t)np + 15) & ~(size_t)15) + 64 * sizeof(int) + JK_TRI_NW * sizeof(double);
}

// The triangular layout is taken for the batches the tuned square kernel served: restricted, dimer-sized fragments
// (n <= 64, a multiple of 8; 640 < npair; a block within ten chunks of 128) in batches of at least 64.
// MQC_HIP_ERI_TRI=0: the square.
bool jk_tri_layout(int n, int np, int nfrag, bool uhf)
{
    static const bool on = [] { const char* e = std::getenv("MQC_HIP_ERI_TRI"); return !(e && e[0] == '0'); }();
    if (!on || uhf || nfrag < 64 || n > 64 || n % 8 != 0 || np <= 10 * 64) return false;
    if (jk_tri_block(np) > JK_TRI_MAXU2 * 128) return false;
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
    if (bv.eri_tri) {
        const size_t lds = jk_tri_lds_bytes(n, np);
        auto kern = jk_tri_kernel<JK_TRI_NW, JK_TRI_MAXU2>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        // Workgroups per fragment: every workgroup pays a prologue (packed density, 23 KB of LDS zeroed) and a flush, so
        // as few as keep three rounds of workgroups on the card -- ONE from 768 fragments on (2016 dimers, same box:
        // 3.12 ms per launch with one, 3.27 with three, 3.55 with six).  MQC_HIP_JK_TRI_WG overrides (measurements).
        static const int gx_env = [] { const char* e = std::getenv("MQC_HIP_JK_TRI_WG"); return e ? std::atoi(e) : 0; }();
        int gx = gx_env > 0 ? gx_env : (768 + bv.nfrag - 1) / bv.nfrag;
        const int maxx = ((np + 1) / 2 + JK_TRI_NW - 1) / JK_TRI_NW;
        if (gx < 1) gx = 1;
        if (gx > maxx) gx = maxx;
        if (gx > 1 && bv.jk_loaded) (void)hipMemsetAsync(bv.jk_loaded, 0, sizeof(int) * (size_t)bv.nfrag, s);
        if (gx > 1) {
            // several workgroups add into J and K (atomics): both start from zero
            (void)hipMemsetAsync(bv.K, 0, sizeof(double) * (size_t)bv.nfrag * n * n, s);
            (void)hipMemsetAsync(bv.J, 0, sizeof(double) * (size_t)bv.nfrag * n * n, s);
        }
        hipLaunchKernelGGL(kern, dim3(gx, bv.nfrag), dim3(64 * JK_TRI_NW), lds, s, bv, only_active ? 1 : 0);
        return;
    }
    (void)hipMemsetAsync(bv.K, 0, sizeof(double) * (size_t)bv.nfrag * n * n, s);
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
