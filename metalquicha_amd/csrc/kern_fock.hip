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
// below the FP64 ridge (~10 flop/byte), so the design goal is coalesced full-rate streaming:
//   * a wave owns one row pair(i,j): 64 lanes read it contiguously from HBM into LDS
//     (shell-pair-blocked data staged through LDS, as the north star asks);
//   * J_ij is the dot product of that row with the packed density (2 - delta_kl) D_kl;
//   * the row is the packed lower triangle of the symmetric n x n matrix V^{ij}_{kl} = (ij|kl);
//     K[i,:] += V^{ij} D[:,j] and K[j,:] += V^{ij} D[:,i] are evaluated from LDS with lane <-> k,
//     so the 2x expansion packed -> square never touches HBM;
//   * K is accumulated per workgroup in LDS (ds_add_f64) and flushed once with global atomics.
// A workgroup (4 waves) keeps D of its fragment in LDS and walks a strided set of rows.
#include "engine.hpp"

namespace mqc {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// KCH = ceil(n / 64): how many k values a lane owns.  DLDS: D, packed D' and the K accumulator live in LDS.
template <int KCH, bool DLDS, int NW>
__global__ void __launch_bounds__(64 * NW) jk_incore_kernel(BatchView bv, int only_active)
{
    extern __shared__ double lds[];
    const int f = blockIdx.y;
    if (only_active && bv.istate[4 * f] == ST_DONE) return;
    const int n = bv.n, np = bv.npair;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double* __restrict__ Dg = bv.D + (size_t)f * n * n;
    double* __restrict__ Jg = bv.J + (size_t)f * n * n;
    double* __restrict__ Kg = bv.K + (size_t)f * n * n;
    const double* __restrict__ M = bv.eri + (size_t)f * np * np;

    // LDS carve-up
    constexpr int NTH = 64 * NW;
    double* rowbuf = lds + (size_t)wave * np;        // NW row buffers
    double* Dp = lds + (size_t)NW * np;               // packed (2 - delta) D, np
    double* Dl = Dp + np;                            // n*n   (DLDS only)
    double* Kl = Dl + (DLDS ? n * n : 0);            // n*n   (DLDS only)

    for (int idx = tid; idx < np; idx += NTH) {
        // unpack idx -> (k,l), k >= l
        int k = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
        while ((k + 1) * (k + 2) / 2 <= idx) ++k;
        while (k * (k + 1) / 2 > idx) --k;
        const int l = idx - k * (k + 1) / 2;
        const double d = Dg[k * n + l];
        Dp[idx] = (k == l) ? d : 2.0 * d;
    }
    if (DLDS) {
        for (int idx = tid; idx < n * n; idx += NTH) { Dl[idx] = Dg[idx]; Kl[idx] = 0.0; }
    }
    __syncthreads();

    const int rows_per_iter = gridDim.x * NW;
    const int iters = (np + rows_per_iter - 1) / rows_per_iter;
    for (int it = 0; it < iters; ++it) {
        const int row = it * rows_per_iter + blockIdx.x * NW + wave;
        const bool active = row < np;
        int i = 0, j = 0;
        double accj = 0.0;
        if (active) {
            i = (int)((sqrt(8.0 * row + 1.0) - 1.0) * 0.5);
            while ((i + 1) * (i + 2) / 2 <= row) ++i;
            while (i * (i + 1) / 2 > row) --i;
            j = row - i * (i + 1) / 2;
            const double* __restrict__ src = M + (size_t)row * np;
            for (int idx = lane; idx < np; idx += 64) {
                const double v = src[idx];
                rowbuf[idx] = v;
                accj += v * Dp[idx];
            }
        }
        __syncthreads();
        if (active) {
            accj = wave_sum(accj);
            if (lane == 0) { Jg[i * n + j] = accj; Jg[j * n + i] = accj; }
            double acc_i[KCH], acc_j[KCH];
            int kbase[KCH];
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                acc_i[c] = 0.0; acc_j[c] = 0.0;
                const int k = lane + 64 * c;
                kbase[c] = k * (k + 1) / 2;
            }
            const double* Di = DLDS ? (Dl + i * n) : (Dg + i * n);
            const double* Dj = DLDS ? (Dl + j * n) : (Dg + j * n);
            for (int l = 0; l < n; ++l) {
                const double dil = Di[l], djl = Dj[l];
                const int lbase = l * (l + 1) / 2;
#pragma unroll
                for (int c = 0; c < KCH; ++c) {
                    const int k = lane + 64 * c;
                    if (k < n) {
                        const double v = rowbuf[k >= l ? kbase[c] + l : lbase + k];
                        acc_i[c] += v * djl;
                        acc_j[c] += v * dil;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const int k = lane + 64 * c;
                if (k < n) {
                    if (DLDS) {
                        atomicAdd(&Kl[i * n + k], acc_i[c]);
                        if (i != j) atomicAdd(&Kl[j * n + k], acc_j[c]);
                    } else {
                        atomicAdd(&Kg[i * n + k], acc_i[c]);
                        if (i != j) atomicAdd(&Kg[j * n + k], acc_j[c]);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (DLDS) {
        for (int idx = tid; idx < n * n; idx += NTH) {
            const double v = Kl[idx];
            if (v != 0.0) atomicAdd(&Kg[idx], v);
        }
    }
}

static int jk_grid_x(const BatchView& bv, int nw)
{
    // enough workgroups to cover 256 CUs a few times over, but few enough that the per-workgroup
    // D load and K flush stay amortised over many rows
    int want = (2048 + bv.nfrag - 1) / bv.nfrag;
    int maxx = (bv.npair + nw - 1) / nw;
    if (want < 1) want = 1;
    if (want > maxx) want = maxx;
    return want;
}

void launch_jk_incore(const BatchView& bv, bool only_active, hipStream_t s)
{
    const int n = bv.n, np = bv.npair;
    (void)hipMemsetAsync(bv.K, 0, sizeof(double) * (size_t)bv.nfrag * n * n, s);
    const int kch = (n + 63) / 64;
    const size_t lds_small = sizeof(double) * ((size_t)5 * np + 2 * (size_t)n * n);
    const size_t lds_mid = sizeof(double) * ((size_t)5 * np);     // 4 waves, D and K in global memory
    const size_t lds_big = sizeof(double) * ((size_t)3 * np);     // 2 waves, D and K in global memory
    const size_t LDS_MAX = 160 * 1024 - 512;
    const int mode = lds_small <= 150 * 1024 ? 0 : (lds_mid <= LDS_MAX ? 1 : 2);
    const size_t lds = mode == 0 ? lds_small : (mode == 1 ? lds_mid : lds_big);
    const int nw = mode == 2 ? 2 : 4;
    dim3 grid(jk_grid_x(bv, nw), bv.nfrag), block(64 * nw);
    const int oa = only_active ? 1 : 0;
#define JK_LAUNCH(KC, DL, NWV)                                                                     \
    do {                                                                                           \
        auto kern = jk_incore_kernel<KC, DL, NWV>;                                                 \
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kern, grid, block, lds, s, bv, oa);                                     \
    } while (0)
    if (mode == 0) {
        if (kch == 1) JK_LAUNCH(1, true, 4);
        else JK_LAUNCH(2, true, 4);
    } else if (mode == 1) {
        if (kch == 1) JK_LAUNCH(1, false, 4);
        else JK_LAUNCH(2, false, 4);
    } else {
        if (kch <= 2) JK_LAUNCH(2, false, 2);
        else if (kch == 3) JK_LAUNCH(3, false, 2);
        else JK_LAUNCH(4, false, 2);
    }
#undef JK_LAUNCH
}

}  // namespace mqc
