// Read-bandwidth probe for the decode-attention prefix pass (not part of the library): how fast can 2048 waves stream
// 320 KiB each in 32-KiB bursts, against a plain grid-stride read of the same bytes?   hipcc --offload-arch=gfx950 -O3 -o hbm_stream_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// mode 0: every wave owns a contiguous stream of `per_wave` bytes at base + wave_id * stride; bursts of NB x 1 KiB loads, all
// issued before any is consumed (the prefix pass's pattern).  rot: start the walk at a per-wave rotated burst.
template <int NB, int SLEEP = 0, bool PIPE = false, bool NT = false>
__global__ void __launch_bounds__(256, 2) stream_kernel(const char* base, size_t stride, int per_wave, int rot_on, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const char* p = base + (size_t)wave * stride;
    const int nb = per_wave / (NB * 1024);
    const int rot = rot_on ? (wave * 5) % nb : 0;
    u32x4 acc = {0, 0, 0, 0};
    for (int b = 0; b < nb; ++b) {
        int bb = b + rot; if (bb >= nb) bb -= nb;
        const char* q = p + (size_t)bb * NB * 1024 + lane * 16;
        u32x4 v[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) v[i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(q + i * 1024)) : *reinterpret_cast<const u32x4*>(q + i * 1024);
        if constexpr (PIPE) {            // consume the first half, sleep (the "compute"), consume the rest: the second half stays in flight over the gap
#pragma unroll
            for (int i = 0; i < NB / 2; ++i) acc ^= v[i];
            if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
#pragma unroll
            for (int i = NB / 2; i < NB; ++i) acc ^= v[i];
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) acc ^= v[i];
            if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);      // a compute phase with nothing in flight for this wave
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// the prefix pass's grid: (88 items, 8 head groups); items 0..63 are whole 320-KiB streams of (slot = item, head = 4 y + wave),
// items 64..87 one burst each; optional start-up chain of dependent loads and a partial-row write per wave
template <int MODE>
__global__ void __launch_bounds__(256, 2) grid2d_kernel(const char* base, const int* chain, float* ws, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, item = blockIdx.x, head = blockIdx.y * 4 + wave;
    int slot = item < 64 ? item : 64;
    if (MODE & 1) { int a = chain[item]; int b = chain[128 + a]; int c = chain[256 + b + lane % 16]; slot += c; }      // zeros: three dependent loads
    const int nb = item < 64 ? 10 : 1;
    const char* p = base + ((size_t)slot * 32 + head) * (320 * 1024);
    const int rot = (head * 5 + item * 3) % nb;
    u32x4 acc = {0, 0, 0, 0};
    for (int b = 0; b < nb; ++b) {
        int bb = b + rot; if (bb >= nb) bb -= nb;
        const char* q = p + (size_t)bb * 32768 + lane * 16;
        u32x4 v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = *reinterpret_cast<const u32x4*>(q + i * 1024);
#pragma unroll
        for (int i = 0; i < 32; ++i) acc ^= v[i];
        __builtin_amdgcn_s_sleep(40);
    }
    if (MODE & 2) {        // 6 (or 16) rows x 130 floats per wave, as the partial rows
        const int rows = item < 64 ? 6 : 16;
        for (int r = 0; r < rows; ++r) {
            float* w = ws + (((size_t)(item * 16 + r) * 32 + head) * 11) * 130;
            w[lane] = (float)acc.x; w[64 + lane] = (float)acc.y; if (lane < 2) w[128 + lane] = 0.f;
        }
    }
    if (MODE & 12) {       // 512-byte aligned O records written as 16 B per lane (+ MODE 8: (m, l) pairs in a separate [head][row] array)
        const int rows = item < 64 ? 6 : 16;
        for (int r = 0; r < rows; ++r) {
            float* w = ws + ((size_t)(item * 16 + r) * 32 + head) * 128;
            if (lane < 32) *reinterpret_cast<float4*>(w + lane * 4) = make_float4((float)acc.x, (float)acc.y, 0.f, 0.f);
        }
        if ((MODE & 8) && lane < rows) *reinterpret_cast<float2*>(ws + (size_t)88 * 16 * 32 * 128 + ((size_t)head * 88 * 16 + item * 16 + lane) * 2) = make_float2(1.f, 2.f);
    }
    if (MODE & 16) {       // 528-byte records (O + m, l, pad), 16 B per lane
        const int rows = item < 64 ? 6 : 16;
        for (int r = 0; r < rows; ++r) {
            float* w = ws + ((size_t)(item * 16 + r) * 32 + head) * 132;
            if (lane < 33) *reinterpret_cast<float4*>(w + lane * 4) = make_float4((float)acc.x, (float)acc.y, 0.f, 0.f);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

__global__ void __launch_bounds__(256) flat_kernel(const char* base, size_t bytes, uint32_t* sink) {
    const size_t n = bytes / 16, step = (size_t)gridDim.x * 256;
    u32x4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < n; i += 4 * step) {
        u32x4 a = reinterpret_cast<const u32x4*>(base)[i], b = reinterpret_cast<const u32x4*>(base)[i + step];
        u32x4 c = reinterpret_cast<const u32x4*>(base)[i + 2 * step], d = reinterpret_cast<const u32x4*>(base)[i + 3 * step];
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n; i += step) acc ^= reinterpret_cast<const u32x4*>(base)[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <class F>
double time_us(F f, int iters = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> t;
    for (int i = 0; i < iters; ++i) { hipEventRecord(e0); f(i); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2] * 1e3;
}

int main() {
    const size_t pool = (size_t)24 << 30;            // rotate through 24 GiB so nothing stays in the 256-MiB Infinity Cache
    char* buf; hipMalloc(&buf, pool); hipMemset(buf, 1, pool);
    uint32_t* sink; hipMalloc(&sink, 4);
    const int waves = 2048, per_wave = 320 * 1024;
    const size_t bytes = (size_t)waves * per_wave;
    auto off = [&](int i, size_t span) { return (size_t)(i % (int)(pool / span)) * span; };
    printf("flat grid-stride read of %zu MB: ", bytes >> 20);
    for (int g : {1024, 2048, 4096, 16384}) { double us = time_us([&](int i) { hipLaunchKernelGGL(flat_kernel, dim3(g), dim3(256), 0, 0, buf + off(i, bytes), bytes, sink); }); printf("grid %d: %.1f us = %.2f TB/s;  ", g, us, bytes / us / 1e6); }
    printf("\n");
    for (int rot = 0; rot < 2; ++rot)
        for (size_t stride : {(size_t)per_wave, (size_t)160 * 1024, (size_t)164 * 1024 + 4096}) {      // packed; the pool's head stride (K then V apart); padded
            const size_t span = stride * waves + per_wave;
            double u32 = time_us([&](int i) { hipLaunchKernelGGL(stream_kernel<32>, dim3(waves / 4), dim3(256), 0, 0, buf + off(i, span), stride, per_wave, rot, sink); });
            double u16 = time_us([&](int i) { hipLaunchKernelGGL(stream_kernel<16>, dim3(waves / 4), dim3(256), 0, 0, buf + off(i, span), stride, per_wave, rot, sink); });
            double u8 = time_us([&](int i) { hipLaunchKernelGGL(stream_kernel<8>, dim3(waves / 4), dim3(256), 0, 0, buf + off(i, span), stride, per_wave, rot, sink); });
            printf("streams stride %zu KiB rot %d: burst 32 KiB %.1f us = %.2f TB/s | 16 KiB %.1f us = %.2f | 8 KiB %.1f us = %.2f\n", stride >> 10, rot, u32, bytes / u32 / 1e6, u16, bytes / u16 / 1e6, u8, bytes / u8 / 1e6);
        }
    {
        const size_t stride = per_wave, span = stride * waves + per_wave;
        double a = time_us([&](int i) { hipLaunchKernelGGL((stream_kernel<32, 0>), dim3(waves / 4), dim3(256), 0, 0, buf + off(i, span), stride, per_wave, 1, sink); });
        double b = time_us([&](int i) { hipLaunchKernelGGL((stream_kernel<32, 20>), dim3(waves / 4), dim3(256), 0, 0, buf + off(i, span), stride, per_wave, 1, sink); });
        double c = time_us([&](int i) { hipLaunchKernelGGL((stream_kernel<32, 40>), dim3(waves / 4), dim3(256), 0, 0, buf + off(i, span), stride, per_wave, 1, sink); });
        double d = time_us([&](int i) { hipLaunchKernelGGL((stream_kernel<32, 80>), dim3(waves / 4), dim3(256), 0, 0, buf + off(i, span), stride, per_wave, 1, sink); });
        double e = time_us([&](int i) { hipLaunchKernelGGL((stream_kernel<32, 40, true>), dim3(waves / 4), dim3(256), 0, 0, buf + off(i, span), stride, per_wave, 1, sink); });
        printf("32-KiB bursts + a gap per burst with nothing in flight: none %.1f us | 0.5 us %.1f | 1.1 us %.1f | 2.1 us %.1f | 1.1 us in mid-burst %.1f\n", a, b, c, d, e);
    }
    {   // nontemporal loads: the stream takes no L2 line
        const size_t stride = per_wave, span = stride * waves + per_wave;
        double a = time_us([&](int i) { hipLaunchKernelGGL((stream_kernel<32, 0, false, false>), dim3(waves / 4), dim3(256), 0, 0, buf + off(i, span), stride, per_wave, 1, sink); });
        double b = time_us([&](int i) { hipLaunchKernelGGL((stream_kernel<32, 0, false, true>), dim3(waves / 4), dim3(256), 0, 0, buf + off(i, span), stride, per_wave, 1, sink); });
        printf("32-KiB bursts, plain loads %.1f us = %.2f TB/s | nontemporal loads %.1f us = %.2f TB/s\n", a, bytes / a / 1e6, b, bytes / b / 1e6);
    }
    {   // the same stream launch alternating with a different streaming kernel (as the prefix pass alternates with the own pass)
        const size_t stride = per_wave, span = stride * waves + per_wave;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        std::vector<float> t;
        for (int i = 0; i < 20; ++i) {
            hipLaunchKernelGGL(flat_kernel, dim3(4096), dim3(256), 0, 0, buf + off(2 * i + 1, span), (size_t)700 << 20, sink);
            hipEventRecord(e0);
            hipLaunchKernelGGL((stream_kernel<32, 40>), dim3(waves / 4), dim3(256), 0, 0, buf + off(2 * i, span), stride, per_wave, 1, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("alternating with a 700-MB flat read: stream launch %.1f us\n", t[10] * 1e3);
        int* chain; hipMalloc(&chain, 4096); hipMemset(chain, 0, 4096);
        float* ws; hipMalloc(&ws, (size_t)88 * 16 * 32 * 11 * 130 * 4);
        const size_t span2 = (size_t)65 * 32 * 320 * 1024;
        double a0 = time_us([&](int i) { hipLaunchKernelGGL(grid2d_kernel<0>, dim3(88, 8), dim3(256), 0, 0, buf + off(i, span2), chain, ws, sink); });
        double a1 = time_us([&](int i) { hipLaunchKernelGGL(grid2d_kernel<1>, dim3(88, 8), dim3(256), 0, 0, buf + off(i, span2), chain, ws, sink); });
        double a2 = time_us([&](int i) { hipLaunchKernelGGL(grid2d_kernel<2>, dim3(88, 8), dim3(256), 0, 0, buf + off(i, span2), chain, ws, sink); });
        double a3 = time_us([&](int i) { hipLaunchKernelGGL(grid2d_kernel<3>, dim3(88, 8), dim3(256), 0, 0, buf + off(i, span2), chain, ws, sink); });
        double a4 = time_us([&](int i) { hipLaunchKernelGGL(grid2d_kernel<4>, dim3(88, 8), dim3(256), 0, 0, buf + off(i, span2), chain, ws, sink); });
        double a8 = time_us([&](int i) { hipLaunchKernelGGL(grid2d_kernel<8>, dim3(88, 8), dim3(256), 0, 0, buf + off(i, span2), chain, ws, sink); });
        double a16 = time_us([&](int i) { hipLaunchKernelGGL(grid2d_kernel<16>, dim3(88, 8), dim3(256), 0, 0, buf + off(i, span2), chain, ws, sink); });
        printf("aligned 512-B records, 16 B per lane: %.1f us | + (m, l) in a [head][row] array %.1f | 528-B records %.1f\n", a4, a8, a16);
        printf("prefix-pass grid (88 x 8, 192 short workgroups): %.1f us | + start-up chain %.1f | + partial-row writes %.1f | both %.1f\n", a0, a1, a2, a3);
    }
    return 0;
}
