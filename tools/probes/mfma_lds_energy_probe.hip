// What is an LDS fragment read worth in a GEMM K loop on this part?  (VERDICT r4 #2d: "a wave layout that reads <= 0.5 KiB of LDS per
// MFMA" - before building such a kernel, price the lever by itself.)
// Every wave of a full grid (256 CUs x 8 waves, two per SIMD - the library kernel's occupancy) runs the library kernel's wave tile: 128 x 64
// as 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16, K-tiles of 4 k-steps, fragments read one k-step ahead of their MFMAs, one LDS read
// issued behind each MFMA (sched_group_barrier), the X-outer boustrophedon walk.  No global traffic, no LDS writes, no barriers in the loop:
// only the MFMAs and the fragment reads differ between the variants:
//   reads = 6   4 X + 2 W fragments per k-step from LDS              (0.75 KiB per MFMA: the library kernel)
//   reads = 4   4 X fragments from LDS, W from 8 resident registers   (0.50 KiB per MFMA: W global -> registers, or a 128 x 128 wave tile)
//   reads = 2   2 fragments from LDS                                  (0.25)
//   reads = 0   everything from registers                             (the bare pipe on this operand pattern)
// LDS holds random bf16 bit patterns (or zeros); every k-step reads a different KiB, so the operands toggle as in a real K loop.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_lds_energy_probe.hip -o mfma_lds_energy_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;

constexpr int LDS_BYTES = 64 * 1024;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// DMA: every wave also issues the library kernel's LDS-DMA volume - 8 x 1 KiB per K-tile (its share of a 256 x 256 tile's X and W images),
// L2-resident source, into a second 64-KiB region nobody reads, one piece behind each of the first MFMAs of a k-step pair.
// BAR: one workgroup barrier per K-tile (behind s_waitcnt vmcnt(0), as the library kernel's double buffer needs it).
template <int READS, bool DMA = false, bool BAR = false>
__global__ void __launch_bounds__(512) kloop(const uint4* __restrict__ src, float* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < LDS_BYTES / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = src[(size_t)(blockIdx.x & 63) * (LDS_BYTES / 16) + i];
    __syncthreads();
    // register-resident fragments (used where a variant does not read from LDS)
    bf8 xr[4][4], wr[4][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xr[kk][j] = *reinterpret_cast<const bf8*>(lds + ((kk * 6 + j) * 1024 + lane * 16));
#pragma unroll
        for (int i = 0; i < 2; ++i) wr[kk][i] = *reinterpret_cast<const bf8*>(lds + ((kk * 6 + 4 + i) * 1024 + lane * 16 + 32768));
    }
    f16v acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf8 xg[2][4], wg[2][2];
    // fragment f (0-3: X, 4-5: W) of k-step s lives at KiB (s * 6 + f) mod 64 of the LDS image: a lane-linear, conflict-free ds_read_b128
    auto rd = [&](int s, int set) {
        const int base = (s * 6 * 1024) & (LDS_BYTES - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (READS == 6 || READS == 4 || (READS == 2 && j < 2))
                xg[set][j] = *reinterpret_cast<const bf8*>(lds + ((base + j * 1024) & (LDS_BYTES - 1)) + lane * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (READS == 6) wg[set][i] = *reinterpret_cast<const bf8*>(lds + ((base + (4 + i) * 1024) & (LDS_BYTES - 1)) + lane * 16);
    };
    auto mm = [&](int kk, int set) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = (j & 1) ? 1 - ii : ii;
                const bf8 x = (READS == 6 || READS == 4 || (READS == 2 && j < 2)) ? xg[set][j] : xr[kk][j];
                const bf8 w = READS == 6 ? wg[set][i] : wr[kk][i];
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, acc[i][j], 0, 0, 0);
            }
    };
    auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < READS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    };
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)(blockIdx.x & 63) * (LDS_BYTES / 16)), 0, LDS_BYTES, 0x00020000);
    auto dma = [&](int it, int half) {               // 4 of the K-tile's 8 pieces
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + LDS_BYTES + (wave * 8 + half * 4 + j) * 1024), 16, lane * 16,
                                                     ((it * 8 + half * 4 + j) * 1024) & (LDS_BYTES - 1), 0, 0);
    };
    auto interleave_dma = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if (i < READS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    };
    int s = 0;
    rd(s, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (BAR && kk == 2) {                    // (the library kernel's barrier sits in the middle of a tile's MFMAs)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (DMA && kk >= 2) dma(it, kk - 2);
            rd(s + 1, (kk + 1) & 1);                 // the next k-step's fragments under this k-step's MFMAs
            mm(kk, kk & 1);
            if (DMA && kk >= 2) interleave_dma(); else interleave();
            __builtin_amdgcn_sched_barrier(0);
            ++s;
        }
    }
    float total = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) total += acc[i][j][e];
    if (total == 12345.678f) sink[blockIdx.x * 512 + tid] = total;
}

static uint16_t to_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }

template <int READS, bool DMA = false, bool BAR = false>
static double run(const uint4* d_src, float* d_sink, int blocks, int iters) {
    constexpr int SM = DMA ? 2 * LDS_BYTES : LDS_BYTES;
    (void)hipFuncSetAttribute((const void*)kloop<READS, DMA, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, SM);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kloop<READS, DMA, BAR><<<blocks, 512, SM>>>(d_src, d_sink, iters / 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) kloop<READS, DMA, BAR><<<blocks, 512, SM>>>(d_src, d_sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const double total = 4.0 * blocks * 8.0 * iters * 32.0 * (2.0 * 32 * 32 * 16);
    return total / (ms * 1e-3) / 1e15;
}

int main() {
    int ncu = 256;
    hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) == hipSuccess) ncu = p.multiProcessorCount;
    const size_t n16 = (size_t)64 * LDS_BYTES / 16;
    uint4* h = (uint4*)malloc(n16 * 16);
    uint4* d_src; float* d_sink;
    hipMalloc(&d_src, n16 * 16); hipMalloc(&d_sink, (size_t)ncu * 512 * 4);
    const int iters = 20000;
    for (int zero = 0; zero < 2; ++zero) {
        srand(7);
        uint16_t* e = (uint16_t*)h;
        for (size_t i = 0; i < n16 * 8; ++i) {
            float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
            e[i] = zero ? 0 : to_bf16(0.05f * sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2));
        }
        hipMemcpy(d_src, h, n16 * 16, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep)
            printf("{\"operands\": \"%s\", \"waves_per_simd\": 2, \"PFs_6_reads_0.75KiB_per_mfma\": %.3f, \"PFs_4_reads_0.50KiB\": %.3f, \"PFs_2_reads_0.25KiB\": %.3f, "
                   "\"PFs_0_reads_registers\": %.3f}\n", zero ? "zeros" : "random bf16 N(0, 0.05)", run<6>(d_src, d_sink, ncu, iters), run<4>(d_src, d_sink, ncu, iters),
                   run<2>(d_src, d_sink, ncu, iters), run<0>(d_src, d_sink, ncu, iters));
        for (int rep = 0; rep < 2; ++rep)
            printf("{\"operands\": \"%s\", \"PFs_6_reads\": %.3f, \"PFs_6_reads_barrier\": %.3f, \"PFs_6_reads_lds_dma\": %.3f, \"PFs_6_reads_lds_dma_barrier\": %.3f, "
                   "\"PFs_4_reads_half_the_lds_dma_barrier\": null}\n", zero ? "zeros" : "random bf16 N(0, 0.05)", run<6>(d_src, d_sink, ncu, iters),
                   run<6, false, true>(d_src, d_sink, ncu, iters), run<6, true, false>(d_src, d_sink, ncu, iters), run<6, true, true>(d_src, d_sink, ncu, iters));
        fflush(stdout);
    }
    return 0;
}
