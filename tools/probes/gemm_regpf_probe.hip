// Inner-loop experiment (not part of the library): the 4-wave / 128 x 128-per-wave / AGPR-accumulator GEMM with a REGISTER prefetch
// stage in front of the LDS (global -> VGPR two tiles ahead -> ds_write one tile ahead -> ds_read -> MFMA), i.e. the structure of the
// vendor library's MT256x256x64 kernels, to see whether it lifts the LDS-bandwidth ceiling of the library kernel (8 waves of
// 128 x 64 with LDS-DMA: 192 KiB of fragment reads + 64 KiB of DMA writes per K-tile against 2,062 MFMA clocks).
// Here: 128 KiB of fragment reads + 64 KiB of ds_write per K-tile.  Y = X W^T, bf16, M, N % 256 == 0, K % 64 == 0, one tile per
// workgroup round (persistent grid-stride over tiles).
//   hipcc --offload-arch=gfx950 -O3 -o gemm_regpf_probe gemm_regpf_probe.hip && ./gemm_regpf_probe [M N K]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <type_traits>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

constexpr int BM = 256, BN = 256, BK = 64, NW = 4, NT = 256, TM = 128, TN = 128, MI = 4, NI = 4;
constexpr int BUF = (BM + BN) * 128;            // 64 KiB per K-tile: rows of 64 k = 128 B
constexpr int LPT = (BM + BN) * 8 / NT;         // 16-byte chunks per thread per K-tile = 16 (8 of X, 8 of W)

__global__ void __launch_bounds__(256) gemm_regpf(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W, uint16_t* __restrict__ Y,
                                                  int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int Mt = M / BM, Nt = N / BN, T = Mt * Nt, nk = K / BK;
    // fragment read addresses (as the library kernel): row * 128 + ((kk * 2 + (lane >> 5)) ^ ((row >> 1) & 7)) * 16, row & 31 = lane & 31
    const int c0 = ((lane >> 5) ^ ((lane >> 1) & 7)) * 16;
    const int xrow = (wr * TM + (lane & 31)) * 128 + c0;
    const int wrow = (BM + wc * TN + (lane & 31)) * 128 + c0;
    // global -> LDS mapping of this thread's 16 chunks: chunk j covers row (j * 32 + tid / 8) of the X tile (j < 8) or of the W tile
    // (j >= 8), 16-byte column tid % 8 (a wave instruction = 8 rows x 128 B: whole lines); LDS slot = column ^ ((row >> 1) & 7),
    // which does not depend on j: ONE per-thread global offset and ONE LDS offset, everything else is immediates / scalars.
    const uint32_t goff = (uint32_t)(tid >> 3) * (uint32_t)(K * 2) + (tid & 7) * 16;
    const uint32_t loff = (uint32_t)(tid >> 3) * 128 + (((tid & 7) ^ ((tid >> 4) & 7)) * 16);
    for (int L = blockIdx.x; L < T; L += gridDim.x) {
        // tile order: groups of 8 row tiles x all column tiles, row tile fastest
        const int per_group = 8 * Nt, gid = L / per_group, first_m = gid * 8, gsz = min(Mt - first_m, 8), in_g = L - gid * per_group;
        const int tm = first_m + in_g % gsz, tn = in_g / gsz;
        const int m0 = tm * BM, n0 = tn * BN;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(X + (size_t)m0 * K), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)n0 * K), 0, 0x7fffffff, 0x00020000);
        const int rowstep = 32 * K * 2;             // bytes between a thread's consecutive chunks (32 rows)
        f32x16_t acc[NI][MI];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        u32x4_t ga[LPT], gb[LPT];                   // tile t+1 (being written into LDS) and tile t+2 (in flight)
        // loads as inline asm: the compiler's own s_waitcnt insertion treats the loop-carried register stage as "the newest loads" and
        // ends up at vmcnt(0) inside every iteration (prefetch distance one tile instead of two); the waits are placed by hand below
        auto gload = [&](u32x4_t (&g)[LPT], int t) {
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                const int so = (j & 7) * rowstep + t * (BK * 2);
                if (j < 8) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(g[j]) : "v"(goff), "s"(rx), "s"(so) : "memory");
                else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(g[j]) : "v"(goff), "s"(rw), "s"(so) : "memory");
            }
        };
        auto lwrite = [&](const u32x4_t (&g)[LPT], int buf, int j0, int j1) {
#pragma unroll
            for (int j = 0; j < LPT; ++j) if (j >= j0 && j < j1) *reinterpret_cast<u32x4_t*>(lds + buf * BUF + j * 4096 + loff) = g[j];
        };
        bf16x8_t xg[2][MI], wg[2][NI];
        auto rd = [&](int buf, int kk, int s) {
#pragma unroll
            for (int i = 0; i < NI; ++i) wg[s][i] = *reinterpret_cast<const bf16x8_t*>(lds + buf * BUF + ((wrow + i * 4096) ^ (kk << 5)));
#pragma unroll
            for (int i = 0; i < MI; ++i) xg[s][i] = *reinterpret_cast<const bf16x8_t*>(lds + buf * BUF + ((xrow + i * 4096) ^ (kk << 5)));
        };
        auto mm = [&](int s) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wg[s][i], xg[s][j], acc[i][j], 0, 0, 0);
        };
        // prologue: tile 0 -> LDS buffer 0, tile 1 -> registers
        gload(ga, 0);
        gload(gb, 1);
        __syncthreads();                           // (previous tile's fragment reads are done)
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        lwrite(ga, 0, 0, LPT);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        rd(0, 0, 0);
        auto iter = [&](int t, int buf, u32x4_t (&cur)[LPT], u32x4_t (&nxt)[LPT]) {
            // registers: `cur` holds tile t+1 (loaded one iteration ago), `nxt` receives tile t+2
            // (loads are issued unconditionally - past the end they re-read the last tile, clamped - so that vmcnt arithmetic is static)
            gload(nxt, t + 2 < nk ? t + 2 : nk - 1);
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");        // tile t+1's 16 loads (issued an iteration ago) have landed
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                rd(buf, kk + 1, (kk + 1) & 1);
                lwrite(cur, buf ^ 1, kk == 0 ? 0 : (kk == 1 ? 6 : 11), kk == 0 ? 6 : (kk == 1 ? 11 : 16));   // tile t+1 into the other buffer over three k-steps
                mm(kk & 1);
                // one LDS operation behind each MFMA: the 8 fragment reads of the next k-step, then this k-step's tile writes
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    else if (i < 14) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // every read of tile t and every write of tile t+1 has been issued: the barrier sits in front of the LAST k-step's MFMAs,
            // which then run while the first fragments of tile t+1 are read
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            rd(buf ^ 1, 0, 0);
            mm(1);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int t = 0; t < nk; t += 2) {          // nk is even (K % 128 == 0): static register roles
            iter(t, 0, gb, ga);
            iter(t + 1, 1, ga, gb);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // epilogue: acc[i][j][e]: n = wc*TN + i*32 + (e&3) + 8*(e>>2) + 4*(lane>>5), m = wr*TM + j*32 + (lane&31)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const int m = m0 + wr * TM + j * 32 + (lane & 31);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wc * TN + i * 32 + q * 8 + 4 * (lane >> 5);
                    const f32x4_t v = {acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
                    *reinterpret_cast<uint2*>(Y + (size_t)m * N + n) = __builtin_bit_cast(uint2, __builtin_convertvector(v, bf16x4_t));
                }
            }
    }
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const int M = argc > 3 ? atoi(argv[1]) : 8192, N = argc > 3 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 8192;
    if (M % 256 || N % 256 || K % 128) { printf("M, N %% 256, K %% 128\n"); return 1; }
    const int n_rot = (int)((600e6 / ((double)N * K * 2)) + 2);
    std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hx) v = f2bf(rnd());
    for (auto& v : hw) v = f2bf(rnd() * 0.1f);
    uint16_t *dx, *dy; std::vector<uint16_t*> dw(n_rot);
    hipMalloc(&dx, hx.size() * 2); hipMalloc(&dy, (size_t)M * N * 2);
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    for (auto& p : dw) { hipMalloc(&p, hw.size() * 2); hipMemcpy(p, hw.data(), hw.size() * 2, hipMemcpyHostToDevice); }
    hipFuncSetAttribute((const void*)gemm_regpf, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
    int ncu = 256; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int T = (M / 256) * (N / 256), grid = T < ncu ? T : ncu;
    auto run = [&](int i) { hipLaunchKernelGGL(gemm_regpf, dim3(grid), dim3(256), 2 * BUF, 0, dx, dw[i % n_rot], dy, M, N, K); };
    run(0);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    std::vector<uint16_t> hy((size_t)M * N);
    hipMemcpy(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int trial = 0; trial < 2000; ++trial) {
        s = s * 1664525u + 1013904223u; const int m = (s >> 4) % M; s = s * 1664525u + 1013904223u; const int n = (s >> 4) % N;
        double r = 0;
        for (int k = 0; k < K; ++k) r += (double)bf2f(hx[(size_t)m * K + k]) * bf2f(hw[(size_t)n * K + k]);
        maxerr = fmax(maxerr, fabs(r - bf2f(hy[(size_t)m * N + n]))); maxref = fmax(maxref, fabs(r));
    }
    printf("check: max |err| %.4f at max |ref| %.2f -> %s\n", maxerr, maxref, maxerr <= 0.01 * maxref + 0.02 ? "OK" : "WRONG");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) run(i);
    hipEventRecord(e0);
    const int iters = 20;
    for (int i = 0; i < iters; ++i) run(i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("{\"M\": %d, \"N\": %d, \"K\": %d, \"us\": %.1f, \"PFs\": %.3f}\n", M, N, K, ms / iters * 1e3, 2.0 * M * N * K / (ms / iters * 1e-3) / 1e15);
    return 0;
}
