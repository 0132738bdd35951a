// What does the MFMA pipe sustain when NOTHING else runs - and does it depend on the data and on the instruction shape?
// Every wave of a full grid (256 CUs x 8 waves, two per SIMD) issues back-to-back MFMAs on register operands loaded once from memory:
// eight independent accumulator chains (no dependency stalls), operands rotated between instructions so that the inputs of consecutive
// MFMAs differ (switching activity of a real K loop).  Variants: v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 / the f16 forms,
// on N(0,1)-like random bit patterns and on zeros.  Prints PF/s per variant (dense flops: 2 M N K per instruction).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_power_probe.hip -o mfma_power_probe          
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;

template <int SHAPE, bool F16, int ORDER = 0>
__global__ void __launch_bounds__(512) mfma_loop(const uint4* __restrict__ src, float* __restrict__ sink, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(size_t)tid * 8 + i]; b[i] = src[(size_t)tid * 8 + 4 + i]; }
    float total = 0.f;
    if constexpr (SHAPE == 32) {
        f16v acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (F16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[i & 3]), __builtin_bit_cast(h8, b[(i + it) & 3]), acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a[ORDER == 0 ? (i & 3) : ORDER == 1 ? (i >> 2) : ORDER == 3 ? ((i + it) & 3) : ORDER == 4 ? (((i + 1) >> 1) & 1) : 0]), __builtin_bit_cast(bf8, b[ORDER == 2 ? 0 : ORDER == 3 ? (i >> 2) : ORDER == 4 ? ((i >> 1) & 3) : ((i + it) & 3)]), acc[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) total += acc[i][e];
    } else {
        f4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (F16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a[i & 3]), __builtin_bit_cast(h8, b[(i + it) & 3]), acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a[i & 3]), __builtin_bit_cast(bf8, b[(i + it) & 3]), acc[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    if (total == 12345.678f) sink[tid] = total;          // keeps the chains alive
}

static uint16_t to_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static uint16_t to_f16(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }

template <int SHAPE, bool F16, int ORDER = 0>
double run(const uint4* d_src, float* d_sink, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<SHAPE, F16, ORDER><<<blocks, 512>>>(d_src, d_sink, iters / 8);          // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) mfma_loop<SHAPE, F16, ORDER><<<blocks, 512>>>(d_src, d_sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const double flops_per = SHAPE == 32 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
    const double total = 4.0 * blocks * 8.0 * iters * 8.0 * flops_per;        // launches x blocks x waves x iterations x MFMAs per iteration
    return total / (ms * 1e-3) / 1e15;
}

int main() {
    int ncu = 256;
    hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) == hipSuccess) ncu = p.multiProcessorCount;
    const int blocks = ncu, threads = blocks * 512;
    const size_t n16 = (size_t)threads * 8 * 8;                                   // 8 uint4 per thread
    uint16_t* h = (uint16_t*)malloc(n16 * 2);
    uint4* d_src; float* d_sink;
    hipMalloc(&d_src, n16 * 2); hipMalloc(&d_sink, (size_t)threads * 4);
    const int iters = 40000;                                                       // ~5 ms per launch at 1.5 PF/s
    struct { const char* name; int kind; } data[] = {{"random", 0}, {"random x 0.02 (weights)", 1}, {"zeros", 2}};
    for (auto& dset : data) {
        for (int f16 = 0; f16 < 2; ++f16) {
            srand(1);
            for (size_t i = 0; i < n16; ++i) {
                float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
                float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
                float v = dset.kind == 2 ? 0.f : (dset.kind == 1 && ((i / 32) & 1)) ? g * 0.02f : g;       // kind 1: b operands small (a weight matrix)
                h[i] = f16 ? to_f16(v) : to_bf16(v);
            }
            hipMemcpy(d_src, h, n16 * 2, hipMemcpyHostToDevice);
            double r32 = f16 ? run<32, true>(d_src, d_sink, blocks, iters) : run<32, false>(d_src, d_sink, blocks, iters);
            double r16 = f16 ? run<16, true>(d_src, d_sink, blocks, iters) : run<16, false>(d_src, d_sink, blocks, iters);
            printf("{\"data\": \"%s\", \"dtype\": \"%s\", \"mfma_32x32x16_PFs\": %.3f, \"mfma_16x16x32_PFs\": %.3f", dset.name, f16 ? "fp16" : "bf16", r32, r16);
            if (!f16) printf(", \"32x32x16_A_held_for_4\": %.3f, \"32x32x16_same_A_B_every_time\": %.3f, \"32x32x16_B_held_for_4\": %.3f, \"32x32x16_B_held_for_2_A_alternates_snake\": %.3f", run<32, false, 1>(d_src, d_sink, blocks, iters), run<32, false, 2>(d_src, d_sink, blocks, iters), run<32, false, 3>(d_src, d_sink, blocks, iters), run<32, false, 4>(d_src, d_sink, blocks, iters));
            printf("}\n");
            fflush(stdout);
        }
    }
    return 0;
}
