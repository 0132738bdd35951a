// Inner-loop experiment (not part of the library): 256 x 256 tile, K-tile 32, FOUR LDS stages, so that an LDS-DMA load has three
// tile periods to land instead of one (the library kernel: K-tile 64, two stages).  Y = X W^T, bf16, one tile per workgroup.
// hipcc --offload-arch=gfx950 -O3 -o gemm_bk32_probe gemm_bk32_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <type_traits>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NW = 8, TM = 128, TN = 64, MI = 4, NI = 2;
constexpr int STAGE = (BM + BN) * 64;          // 32 KiB: rows of 32 k = 64 B
constexpr int NST = 4;
constexpr int IMG = (BM + BN) / 16;            // 1-KiB images (16 rows x 64 B) per stage = 32 -> 4 per wave

__global__ void __launch_bounds__(512) gemm_bk32(const uint16_t* X, const uint16_t* W, uint16_t* Y, int M, int N, int K, int Nt) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    // tile order: 8 row tiles x all column tiles, row tile fastest (as the library kernel)
    const int L = blockIdx.x, Mt = M / BM, per_group = 8 * Nt, gid = L / per_group, in_g = L - gid * per_group;
    const int tm = gid * 8 + in_g % 8, tn = in_g / 8;
    const int m0 = tm * BM, n0 = tn * BN;
    (void)Mt;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(X + (size_t)m0 * K), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)n0 * K), 0, 0x7fffffff, 0x00020000);
    // image i of a stage: rows 16 i .. 16 i + 15 (X rows 0..255 then W rows 0..255), lane -> (row = lane / 4, slot = lane % 4), source chunk = slot ^ ((row >> 2) & 3)
    uint32_t off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int img = j * NW + wave, row = (img & 15) * 16 + (lane >> 2), slot = lane & 3;
        const int c = slot ^ ((row >> 2) & 3);
        off[j] = (uint32_t)row * (uint32_t)(K * 2) + c * 16;
    }
    auto stage = [&](int t, int st) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int img = j * NW + wave;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(img < 16 ? rx : rw, (lds_ptr_t)(lds + st * STAGE + img * 1024), 16, off[j], t * 64, 0, 0);
        }
    };
    const int frow = (lane & 31) * 64, fsw = ((lane & 31) >> 2) & 3, kh = lane >> 5;
    f32x16_t acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nk = K / 32;
    // Pipeline: the fragments of tile t are in registers when iteration t starts (read during iteration t - 1); iteration t reads
    // tile t + 1's fragments behind its MFMAs.  So at barrier(t) every wave is done with tile t's LDS stage (free for tile t + 4) and
    // tile t + 1 must have landed - it was issued at barrier(t - 3): three tile periods of flight.
    bf16x8_t xa[2][MI], wa[2][NI], xb[2][MI], wb[2][NI];
    auto rd = [&](int st, bf16x8_t (&xg)[2][MI], bf16x8_t (&wg)[2][NI]) {
        const char* base = lds + st * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < NI; ++i) wg[kk][i] = *reinterpret_cast<const bf16x8_t*>(base + (BM + wc * TN + i * 32) * 64 + frow + (((2 * kk + kh) ^ fsw) * 16));
#pragma unroll
            for (int j = 0; j < MI; ++j) xg[kk][j] = *reinterpret_cast<const bf16x8_t*>(base + (wr * TM + j * 32) * 64 + frow + (((2 * kk + kh) ^ fsw) * 16));
        }
    };
    auto mm = [&](bf16x8_t (&xg)[2][MI], bf16x8_t (&wg)[2][NI]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wg[kk][i], xg[kk][j], acc[i][j], 0, 0, 0);
    };
    auto interleave = [&](bool dma, bool reads) {          // MFMA, [DMA], [read], ...
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (dma && i < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if (reads && i < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    };
    stage(0, 0); stage(1, 1); stage(2, 2); stage(3, 3);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");       // tile 0 landed
    __builtin_amdgcn_s_barrier();
    rd(0, xa, wa);
    auto iter = [&](int t, bf16x8_t (&xc)[2][MI], bf16x8_t (&wc_)[2][NI], bf16x8_t (&xn)[2][MI], bf16x8_t (&wn)[2][NI], auto st_c, auto nx_c) {
        constexpr bool ST = decltype(st_c)::value, NX = decltype(nx_c)::value;
        // tile t + 1 landed (tiles t + 2, t + 3 may still fly); own reads of tile t done
        if constexpr (ST) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ST) stage(t + 4, t & 3);
        if constexpr (NX) rd((t + 1) & 3, xn, wn);
        mm(xc, wc_);
        interleave(ST, NX);
        __builtin_amdgcn_sched_barrier(0);
    };
    using T_ = std::true_type; using F_ = std::false_type;
    int t = 0;
    for (; t + 4 < nk; t += 2) { iter(t, xa, wa, xb, wb, T_{}, T_{}); iter(t + 1, xb, wb, xa, wa, T_{}, T_{}); }      // nk % 4 == 0: the stage index below stays literal-friendly
    iter(t, xa, wa, xb, wb, F_{}, T_{}); iter(t + 1, xb, wb, xa, wa, F_{}, T_{});
    iter(t + 2, xa, wa, xb, wb, F_{}, T_{}); iter(t + 3, xb, wb, xa, wa, F_{}, F_{});
    const int mrow = m0 + wr * TM + (lane & 31);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wc * TN + i * 32 + q * 8 + 4 * (lane >> 5);
                const f32x4_t v = {acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
                *reinterpret_cast<uint2*>(Y + (size_t)(mrow + j * 32) * N + n) = __builtin_bit_cast(uint2, __builtin_convertvector(v, bf16x4_t));
            }
#endif
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int M = 8192, N = 8192, K = 8192;
    std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.f - 0.5f; };
    for (auto& v : hx) v = f2bf(rnd());
    for (auto& v : hw) v = f2bf(rnd() * 0.1f);
    uint16_t *X, *W, *Y;
    hipMalloc(&X, hx.size() * 2); hipMalloc(&W, hw.size() * 2); hipMalloc(&Y, (size_t)M * N * 2);
    hipMemcpy(X, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    const int smem = NST * STAGE;
    hipFuncSetAttribute((const void*)gemm_bk32, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    const int Nt = N / BN, tiles = (M / BM) * Nt;
    auto launch = [&]() { hipLaunchKernelGGL(gemm_bk32, dim3(tiles), dim3(512), smem, 0, X, W, Y, M, N, K, Nt); };
    launch(); hipDeviceSynchronize();
    std::vector<uint16_t> hy((size_t)M * N);
    hipMemcpy(hy.data(), Y, hy.size() * 2, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int t = 0; t < 64; ++t) {
        const int m = (t * 1237) % M, n = (t * 7919 + 13) % N;
        double r = 0; for (int k = 0; k < K; ++k) r += (double)bf2f(hx[(size_t)m * K + k]) * bf2f(hw[(size_t)n * K + k]);
        maxerr = std::max(maxerr, fabs(r - bf2f(hy[(size_t)m * N + n])));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int i = 0; i < 12; ++i) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    printf("gemm_bk32 8192^3: %.1f us = %.0f TF/s (max |err| on 64 samples %.3f)\n", ts[6] * 1e3, 2.0 * M * N * K / (ts[6] * 1e-3) / 1e12, maxerr);
    return 0;
}
