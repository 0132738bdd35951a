// Do the two fp32 -> fp16 conversions the fp16 instantiation of the model kernels uses round alike?  v_cvt_f16_f32 (f2e) vs
// v_cvt_pk_f16_f32 (cvt_pk / pack, csrc/vdd_elem.h) vs a software round-to-nearest-even, over 2^28 bit patterns spread over the whole
// fp32 range (denormal results and ties included).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off cvt_f16_probe.hip -o cvt_f16_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef __attribute__((ext_vector_type(2))) float f2;
__device__ uint32_t soft_rne(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f), s = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return s | 0x7e00u;
    if (a >= 0x477ff000u) return s | 0x7c00u;                       // >= 65520 rounds to inf
    if (a < 0x33000001u) return s;                                  // <= 2^-25: rounds to zero (2^-25 itself is a tie -> even = 0)
    int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;
    int shift = e >= -14 ? 13 : 13 + (-14 - e);
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1))) ++q;
    uint32_t h = e >= -14 ? (((uint32_t)(e + 15) << 10) + (q - 0x400u)) : q;
    return s | h;
}
__global__ void probe(unsigned long long* counts, uint32_t* examples) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t bits = (uint32_t)(i << 4) | (uint32_t)((i * 2654435761u) & 15u);
    const float x = __builtin_bit_cast(float, bits);
    const uint32_t a = __builtin_bit_cast(uint16_t, (_Float16)x);
    const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{x, -x}, h2));
    const uint32_t b = pk & 0xffffu, c = soft_rne(x);
    const bool nan = (bits & 0x7fffffffu) > 0x7f800000u;
    if (!nan && a != b) { unsigned long long k = atomicAdd(&counts[0], 1ull); if (k < 8) { examples[k * 4] = bits; examples[k * 4 + 1] = a; examples[k * 4 + 2] = b; examples[k * 4 + 3] = c; } }
    if (!nan && a != c) atomicAdd(&counts[1], 1ull);
    if (!nan && b != c) atomicAdd(&counts[2], 1ull);
    if (!nan && (pk >> 16) != (c ^ 0x8000u)) atomicAdd(&counts[3], 1ull);
}
int main() {
    unsigned long long* counts; uint32_t* ex;
    hipMalloc(&counts, 32); hipMalloc(&ex, 8 * 16); hipMemset(counts, 0, 32); hipMemset(ex, 0, 128);
    probe<<<(1u << 28) / 256, 256>>>(counts, ex);
    unsigned long long h[4]; uint32_t he[32];
    hipMemcpy(h, counts, 32, hipMemcpyDeviceToHost); hipMemcpy(he, ex, 128, hipMemcpyDeviceToHost);
    printf("{\"patterns\": %u, \"cvt_vs_pk\": %llu, \"cvt_vs_soft_rne\": %llu, \"pk_lo_vs_soft_rne\": %llu, \"pk_hi_vs_soft_rne\": %llu}\n", 1u << 28, h[0], h[1], h[2], h[3]);
    for (int k = 0; k < 8 && k < (int)h[0]; ++k) printf("  x=0x%08x (%g) cvt=0x%04x pk=0x%04x soft=0x%04x\n", he[k * 4], __builtin_bit_cast(float, he[k * 4]), he[k * 4 + 1], he[k * 4 + 2], he[k * 4 + 3]);
    return 0;
}
