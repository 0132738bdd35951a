// Decode-regime GEMM for gfx950: Y[M,N] = X[M,K] W[N,K]^T for 9 <= M <= 256 rows (rows = questions x branches).
//
// At these M the product is still bound by streaming W from HBM (M=192, 7B: 78 GFLOP vs 405 MB per layer), but a
// generic library GEMM tiles M and N for reuse it does not need and reaches ~370 TF/s / <2 TB/s here (rocprofv3,
// profiles/r01_e2e_Q96_kernel_stats.txt).  This kernel is organised around the weight stream instead:
//   * block = 256 threads = 4 waves, output tile = ALL M rows x 64 columns; wave w owns columns 16w..16w+15;
//   * each wave streams its own 16 W rows straight into MFMA B fragments (lane (n = l&15, g = l>>4) loads 16 B of
//     W[n][k + 8g ..], k-contiguous): W is read from HBM exactly once, never staged, prefetched 4 K-steps ahead;
//   * the X tile [M x 64] of a K-step is staged ONCE per block in LDS (register-staged, double-buffered, one barrier
//     per K-step, rows padded to 144 B so the 16 rows of an A-fragment read land on distinct 16-B bank slots) and
//     shared by the 4 waves, so X traffic out of L2 is N/64 x |X| instead of N/16 x |X|;
//   * 16x16x32 bf16 MFMAs, accumulators MT x 4 VGPRs per lane;
//   * N = 4096 projections fill the chip through split-K: gridDim.y slices write fp32 slabs [S][M][N] that the
//     following RMSNorm sums (no atomics, no extra pass).
// Bound: HBM (weight streaming) up to M ~ 256; MFMA only beyond.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vdd_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ uint32_t f2bf(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

constexpr int BK = 64;          // K-step
constexpr int XLD = 72;         // padded LDS row: 64 + 8 elements = 144 B
constexpr int PF = 4;           // W prefetch depth in K-steps

template <int MT, bool SLAB>
__global__ void __launch_bounds__(256) mid_gemm_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W,
                                                       uint16_t* __restrict__ Y, float* __restrict__ Yslab, int M, int N, int K,
                                                       long long ldx, long long ldy) {
    constexpr int BM = 16 * MT;
    constexpr int XCH = BM * 8 / 256;               // 16-B chunks of the X tile per thread (MT even -> integer)
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];   // 2 x BM x XLD
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 15, g = lane >> 4;
    const int nsplit = gridDim.y, kslice = K / nsplit, kbeg = blockIdx.y * kslice, nsteps = kslice / BK;
    const int n0 = blockIdx.x * 64 + wave * 16;
    int wrow = n0 + ln; if (wrow >= N) wrow = N - 1;
    const uint16_t* wp = W + (size_t)wrow * K + kbeg + g * 8;

    // X staging assignment: chunk c = tid + 256 i  ->  row c / 8, 16-B slot c % 8
    const uint16_t* xsrc[XCH];
    int xdst[XCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int c = tid + 256 * i, r = c >> 3, sl = c & 7;
        int rr = r < M ? r : M - 1;
        xsrc[i] = X + (size_t)rr * ldx + kbeg + sl * 8;
        xdst[i] = r * XLD + sl * 8;
    }
    f32x4_t acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // prologue: X tile 0 -> LDS buffer 0; W fragments of steps 0 .. PF-1
    uint4 xs[XCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) xs[i] = *reinterpret_cast<const uint4*>(xsrc[i]);
    bf16x8_t wf[PF][2];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int st = p < nsteps ? p : nsteps - 1;
            wf[p][ks] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)st * BK + ks * 32);
        }
#pragma unroll
    for (int i = 0; i < XCH; ++i) *reinterpret_cast<uint4*>(&lds[xdst[i]]) = xs[i];
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const uint16_t* cur = lds + (step & 1) * BM * XLD;
        uint16_t* nxt = lds + ((step + 1) & 1) * BM * XLD;
        const bool more = step + 1 < nsteps;
        if (more) {
#pragma unroll
            for (int i = 0; i < XCH; ++i) xs[i] = *reinterpret_cast<const uint4*>(xsrc[i] + (size_t)(step + 1) * BK);
        }
        // W fragments of this step leave the ring; the load for step + PF enters it
        const bf16x8_t b0 = wf[0][0], b1 = wf[0][1];
#pragma unroll
        for (int p = 0; p + 1 < PF; ++p) { wf[p][0] = wf[p + 1][0]; wf[p][1] = wf[p + 1][1]; }
        {
            const int st = step + PF < nsteps ? step + PF : nsteps - 1;
            wf[PF - 1][0] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)st * BK);
            wf[PF - 1][1] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)st * BK + 32);
        }
        const uint16_t* ap = cur + ln * XLD + g * 8;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(ap + t * 16 * XLD);
            const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(ap + t * 16 * XLD + 32);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[t], 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < XCH; ++i) *reinterpret_cast<uint4*>(&nxt[xdst[i]]) = xs[i];
        }
        __syncthreads();
    }
    // epilogue: C/D map col = ln, row = 4 g + r
    const int col = n0 + ln;
    if (col < N) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + g * 4 + r;
                if (row < M) {
                    if constexpr (SLAB) Yslab[((size_t)blockIdx.y * M + row) * N + col] = acc[t][r];
                    else Y[(size_t)row * ldy + col] = (uint16_t)f2bf(acc[t][r]);
                }
            }
    }
}

template <int MT>
int launch(const uint16_t* X, const uint16_t* W, uint16_t* Y, float* Ys, int M, int N, int K, long long ldx, long long ldy, int nsplit,
           hipStream_t st) {
    const size_t lds = 2 * (size_t)(16 * MT) * XLD * 2;
    dim3 grid((N + 63) / 64, nsplit), block(256);
    if (Ys != nullptr) {
        static int a = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&mid_gemm_kernel<MT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)a;
        hipLaunchKernelGGL((mid_gemm_kernel<MT, true>), grid, block, lds, st, X, W, Y, Ys, M, N, K, ldx, ldy);
    } else {
        static int a = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&mid_gemm_kernel<MT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)a;
        hipLaunchKernelGGL((mid_gemm_kernel<MT, false>), grid, block, lds, st, X, W, Y, Ys, M, N, K, ldx, ldy);
    }
    return hipGetLastError() == hipSuccess ? VDD_OK : VDD_ERR_LAUNCH;
}

}  // namespace

extern "C" int vdd_mid_gemm(const void* X, const void* W, void* Y, float* Y_slabs, int M, int N, int K, int64_t ldx, int64_t ldy,
                            int n_split, void* stream) {
    if (M <= 0 || N <= 0) return VDD_OK;
    if (!X || !W || (!Y && !Y_slabs) || M > 256 || n_split < 1 || K % (BK * n_split) != 0 || (ldx % 8) != 0) return VDD_ERR_INVALID_ARG;
    if (!Y_slabs && n_split != 1) return VDD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    auto x = (const uint16_t*)X; auto w = (const uint16_t*)W; auto y = (uint16_t*)Y;
    if (M <= 32) return launch<2>(x, w, y, Y_slabs, M, N, K, ldx, ldy, n_split, st);
    if (M <= 64) return launch<4>(x, w, y, Y_slabs, M, N, K, ldx, ldy, n_split, st);
    if (M <= 96) return launch<6>(x, w, y, Y_slabs, M, N, K, ldx, ldy, n_split, st);
    if (M <= 128) return launch<8>(x, w, y, Y_slabs, M, N, K, ldx, ldy, n_split, st);
    if (M <= 192) return launch<12>(x, w, y, Y_slabs, M, N, K, ldx, ldy, n_split, st);
    return launch<16>(x, w, y, Y_slabs, M, N, K, ldx, ldy, n_split, st);
}
