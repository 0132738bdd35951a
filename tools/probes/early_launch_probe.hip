// Early-launch probe (not part of the library): what does a chain of weight-streaming launches gain when launch n + 1 is already
// resident - its first loads in flight - while launch n drains?  The one-question decode step is such a chain (5 launches per layer,
// ~4.5 us of boundary each: DESIGN_APPENDIX.md "the few-row launch chain").
//   serial      the chain on ONE stream (what the engine does today, captured in a graph)
//   two-stream  launch i on stream i % 2; the only dependency of launch i on launch i - 1 is a device flag: every workgroup issues
//               its first batch of loads, THEN waits (bounded spin) until all workgroups of launch i - 1 have arrived, then goes on.
//               Launches i and i - 2 share a stream, so at most two launches are resident at a time.
// Every launch streams `bytes` of its own matrix (rotating through > 256 MiB so nothing stays in the Infinity Cache) and arrives on
// its flag with a relaxed agent-scope atomic after its (tiny) result store.
//   hipcc --offload-arch=gfx950 -O3 -o early_launch_probe early_launch_probe.hip && ./early_launch_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int NB = 16;                                   // 16-byte loads per lane and batch (two register stages of 8 in the real kernels)
__global__ void __launch_bounds__(256) stream_launch(const char* w, size_t bytes, const int* wait_flag, int expect, int* set_flag, uint32_t* sink,
                                                     int* timeouts, int variant, int* relay, int epoch, int* relay_set) {
    const int tid = threadIdx.x;
    const size_t per_block = bytes / gridDim.x;           // contiguous slice per workgroup, walked in batches of NB x 4 KiB
    const char* p = w + (size_t)blockIdx.x * per_block + tid * 16;
    const int nbat = (int)(per_block / (NB * 4096));
    u32x4 v[NB], acc = {0, 0, 0, 0};
    const bool late = (variant & 1) != 0;                 // variant bit 0: load only after the wait (no prefetch: the wait alone)
    if (!late) {
#pragma unroll
        for (int i = 0; i < NB; ++i) v[i] = *reinterpret_cast<const u32x4*>(p + i * 4096);      // first batch: in flight before the wait
    }
    if (wait_flag != nullptr) {
        if (tid == 0) {
            int spins = 0;
            if (variant & 8) {
                // two levels: 8 leader workgroups poll the arrival counter and each raises one relay word (256 B apart: its own memory
                // channel); everybody else polls the relay word of its group - 8 pollers on the hot word instead of 512
                int* relay_w = relay + (blockIdx.x & 7) * 64;
                if (blockIdx.x < 8) {
                    while (__hip_atomic_load(wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1 << 20)) { atomicAdd(timeouts, 1); break; }
                    }
                    __hip_atomic_store(relay_w, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    while (__hip_atomic_load(relay_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins > (1 << 20)) { atomicAdd(timeouts, 1); break; }
                    }
                }
            } else {
                while (__hip_atomic_load(wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
                    if (variant & 4) __builtin_amdgcn_s_sleep(64); else __builtin_amdgcn_s_sleep(4);
                    if (++spins > (1 << 20)) { atomicAdd(timeouts, 1); break; }                   // never hang the box
                }
            }
        }
        __syncthreads();
    }
    if (late) {
#pragma unroll
        for (int i = 0; i < NB; ++i) v[i] = *reinterpret_cast<const u32x4*>(p + i * 4096);
    }
    for (int b = 0; b < nbat; ++b) {
#pragma unroll
        for (int i = 0; i < NB; ++i) acc ^= v[i];
        if (b + 1 < nbat) {
#pragma unroll
            for (int i = 0; i < NB; ++i) v[i] = *reinterpret_cast<const u32x4*>(p + (size_t)(b + 1) * NB * 4096 + i * 4096);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
    __syncthreads();
    if (tid == 0 && set_flag != nullptr) {
        if (variant & 16) {
            // hierarchical arrival: 8 group counters (256 B apart); the last arriver of a group adds the group's size to the launch's counter
            int* grp = relay_set + 8 * 64 + (blockIdx.x & 7) * 64;
            const int per = gridDim.x / 8;
            if (__hip_atomic_fetch_add(grp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == per - 1)
                __hip_atomic_fetch_add(set_flag, per, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_fetch_add(set_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char** argv) {
    const int chain = argc > 2 ? atoi(argv[2]) : 5;       // 5: the one-question layer (five launches); 9: the 17 - 256-row layer (norm, qkv, rope, attention, o, norm, gate/up, down + one more small launch)
    const int variant = argc > 1 ? atoi(argv[1]) : 0;     // bit 0: no prefetch before the wait; bit 1: wait for the FIRST arrival only; bit 2: long poll sleep; bit 3: two-level poll (8 leaders + relay words); bit 4: hierarchical arrival (8 group counters)
    const size_t pool = (size_t)4 << 30;
    char* buf; hipMalloc(&buf, pool); hipMemset(buf, 1, pool);
    uint32_t* sink; hipMalloc(&sink, 4);
    int *flags, *timeouts, *relay; hipMalloc(&flags, 4096 * 4); hipMalloc(&timeouts, 4); hipMalloc(&relay, 4096 * 1024 * 4);
    hipStream_t s[2]; hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking); hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
    // a 7B decoder layer's five launches: qkv 100 MB, (attention: 21 MB of K/V), o 33.5, gate/up 180, down 90
    // (chain 9: the elementwise launches of the nine-launch layer move a few MB each: here 32 MB = one 64-KiB batch per workgroup, ~5 us - their cost IS the boundary)
    const size_t sizes5[5] = {(size_t)100 << 20, (size_t)21 << 20, (size_t)33 << 20, (size_t)180 << 20, (size_t)90 << 20};
    const size_t sizes9[9] = {(size_t)32 << 20, (size_t)100 << 20, (size_t)32 << 20, (size_t)64 << 20, (size_t)33 << 20, (size_t)32 << 20, (size_t)180 << 20, (size_t)32 << 20, (size_t)90 << 20};
    const size_t* sizes = chain == 9 ? sizes9 : sizes5;
    const int NL = chain == 9 ? 9 : 5;
    const int grid = 512, layers = 32, n = NL * layers;
    // both forms are captured into a hipGraph (the engine replays its decode step as a graph: no host launch cost in the comparison);
    // the two-stream form forks s[1] off s[0] at the start of the capture and joins it at the end - between launches the flags are
    // the only dependency across the streams
    auto build = [&](int mode) {
        hipGraph_t g; hipGraphExec_t ge;
        hipEvent_t fork, join; hipEventCreate(&fork); hipEventCreate(&join);
        hipStreamBeginCapture(s[0], hipStreamCaptureModeGlobal);
        if (mode >= 1) { hipEventRecord(fork, s[0]); hipStreamWaitEvent(s[1], fork, 0); }
        size_t off = 0;
        for (int i = 0; i < n; ++i) {
            size_t bytes = sizes[i % NL] / (grid * NB * 4096) * (grid * NB * 4096);
            if (off + bytes > pool) off = 0;
            const int* wf = (mode == 1 && i > 0) ? flags + (i - 1) : nullptr;      // mode 2: two streams, NO dependency at all (what the two queues alone cost)
            hipLaunchKernelGGL(stream_launch, dim3(grid), dim3(256), 0, mode >= 1 ? s[i & 1] : s[0], buf + off, bytes, wf, (variant & 2) ? 1 : grid, flags + i, sink, timeouts, variant, relay + (size_t)(i > 0 ? i - 1 : 0) * 1024, 1, relay + (size_t)i * 1024);
            off += bytes;
        }
        if (mode >= 1) { hipEventRecord(join, s[1]); hipStreamWaitEvent(s[0], join, 0); }
        hipStreamEndCapture(s[0], &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        return ge;
    };
    hipGraphExec_t ge[3] = {build(0), build(1), build(2)};
    auto run = [&](int mode) {
        hipMemset(flags, 0, 4096 * 4); hipMemset(timeouts, 0, 4); hipMemset(relay, 0, 4096 * 1024 * 4);
        hipDeviceSynchronize();
        auto t0 = std::chrono::high_resolution_clock::now();
        hipGraphLaunch(ge[mode], s[0]);
        hipStreamSynchronize(s[0]);
        auto t1 = std::chrono::high_resolution_clock::now();
        int to = 0; hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost);
        return std::make_pair(std::chrono::duration<double, std::micro>(t1 - t0).count() / layers, to);
    };
    size_t layer_bytes = 0; for (int i = 0; i < NL; ++i) layer_bytes += sizes[i] / (grid * NB * 4096) * (grid * NB * 4096);
    for (int rep = 0; rep < 3; ++rep) {
        auto a = run(0); auto b = run(1); auto c = run(2);
        printf("{\"launches_per_layer\": %d, \"variant\": %d, \"layer_MB\": %.1f, \"serial_us_per_layer\": %.1f, \"serial_TBps\": %.2f, \"two_stream_flag_us_per_layer\": %.1f, \"two_stream_TBps\": %.2f, \"flag_timeouts\": %d, "
               "\"two_stream_independent_us_per_layer\": %.1f}\n",
               NL, variant, layer_bytes / 1e6, a.first, layer_bytes / a.first / 1e6, b.first, layer_bytes / b.first / 1e6, b.second, c.first);
    }
    return 0;
}
