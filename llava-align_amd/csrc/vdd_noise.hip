// Forward-diffusion noising of the VCD branch's image (vcd_utils/vcd_add_noise.py:18-22):
//   x_t = fl(fl(sqrt(abar_t) * x_0) + fl(sqrt(1 - abar_t) * eps)),   eps ~ N(0, 1)
// Elementwise and HBM-bound (12 B/element for fp32 with explicit eps, 8 B with the in-kernel
// Philox + Box-Muller generator).  The 1000-step schedule itself (:7-16) is a host-side
// table; the two scalars arrive as arguments.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vdd_hip.h"

namespace {

__device__ __forceinline__ void philox4(uint64_t seed, uint64_t ctr, uint32_t out[4]) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x5eed, c3 = 0;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

template <typename T> __device__ __forceinline__ float ld(const T* p, long long i);
template <> __device__ __forceinline__ float ld<float>(const float* p, long long i) { return p[i]; }
template <> __device__ __forceinline__ float ld<_Float16>(const _Float16* p, long long i) { return (float)p[i]; }
template <> __device__ __forceinline__ float ld<__bf16>(const __bf16* p, long long i) { return (float)p[i]; }

// 4 elements per thread: one Philox block -> two Box-Muller pairs
template <typename T>
__global__ void __launch_bounds__(256) vdd_noise_kernel(const T* __restrict__ x, T* __restrict__ y, long long n,
                                                        float a, float b, const float* __restrict__ eps,
                                                        uint64_t seed, uint64_t offset) {
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (long long i0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i0 < n; i0 += stride) {
        float z[4];
        if (eps != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = (i0 + j < n) ? eps[i0 + j] : 0.f;
        } else {
            uint32_t r[4];
            philox4(seed, offset + (uint64_t)(i0 >> 2), r);
            const float inv = 1.0f / 4294967296.0f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float u1 = ((float)r[2 * h] + 0.5f) * inv, u2 = ((float)r[2 * h + 1] + 0.5f) * inv;
                float rad = sqrtf(-2.0f * __logf(u1));
                float s, c;
                __sincosf(6.28318530717958647692f * u2, &s, &c);
                z[2 * h] = rad * c; z[2 * h + 1] = rad * s;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i0 + j < n) {
                float xv = ld<T>(x, i0 + j);
                T t0 = (T)__fmul_rn(a, xv);               // each torch op rounds into the image dtype
                T t1 = (T)__fmul_rn(b, (float)(T)z[j]);
                y[i0 + j] = (T)__fadd_rn((float)t0, (float)t1);
            }
        }
    }
}

}  // namespace

extern "C" int vdd_add_diffusion_noise(const void* x, void* y, int64_t n, int dtype, float sqrt_abar,
                                       float sqrt_one_minus_abar, const float* eps, uint64_t seed,
                                       uint64_t offset, void* hip_stream) {
    if (n < 0 || (n > 0 && (!x || !y))) return VDD_ERR_INVALID_ARG;
    if (n == 0) return VDD_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    long long blocks = (n + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    dim3 g((unsigned)blocks), b(256);
    switch (dtype) {
        case VDD_F32: hipLaunchKernelGGL(vdd_noise_kernel<float>, g, b, 0, st, (const float*)x, (float*)y, (long long)n, sqrt_abar, sqrt_one_minus_abar, eps, seed, offset); break;
        case VDD_F16: hipLaunchKernelGGL(vdd_noise_kernel<_Float16>, g, b, 0, st, (const _Float16*)x, (_Float16*)y, (long long)n, sqrt_abar, sqrt_one_minus_abar, eps, seed, offset); break;
        case VDD_BF16: hipLaunchKernelGGL(vdd_noise_kernel<__bf16>, g, b, 0, st, (const __bf16*)x, (__bf16*)y, (long long)n, sqrt_abar, sqrt_one_minus_abar, eps, seed, offset); break;
        default: return VDD_ERR_INVALID_ARG;
    }
    return hipGetLastError() == hipSuccess ? VDD_OK : VDD_ERR_LAUNCH;
}
