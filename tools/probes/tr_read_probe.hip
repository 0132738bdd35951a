// What does ds_read_b64_tr_b16 return?  Hypothesis: within a 16-lane group, lane i gets elem j = the (i % 4)-th 16-bit element of the
// 8 bytes addressed by lane 4 j + i / 4 of the same group (a 4 x 16 block transposed, row r supplied by lanes 4 r .. 4 r + 3).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const int* addr_in, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)lds + (uint32_t)addr_in[threadIdx.x];
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = r.x & 0xFFFF; out[threadIdx.x * 4 + 1] = r.x >> 16;
    out[threadIdx.x * 4 + 2] = r.y & 0xFFFF; out[threadIdx.x * 4 + 3] = r.y >> 16;
}
int main() {
    int h_addr[64]; uint16_t h_out[256];
    int* d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, 256); hipMalloc(&d_out, 512);
    for (int test = 0; test < 2; ++test) {
        for (int l = 0; l < 64; ++l) h_addr[l] = test == 0 ? l * 8 : ((l * 37 + 11) % 500) * 8;      // canonical / scattered 8-B aligned
        hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, 512, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int grp = l & ~15, i = l & 15, src = grp + 4 * j + i / 4;
                const int want = h_addr[src] / 2 + (i % 4);
                if (h_out[l * 4 + j] != want) ++bad;
            }
        printf("test %d: %d mismatches vs hypothesis\n", test, bad);
        if (bad) for (int l = 0; l < 20; ++l) printf("lane %d (addr elem %d): %d %d %d %d\n", l, h_addr[l] / 2, h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
    }
    return 0;
}
