// Element type of the model kernels (language model, ViT, projector): the reference runs them in the checkpoint's dtype -
// fp16 for every released LLaVA-1.5 / InstructBLIP / Qwen-VL driver (experiments/llava/model/builder.py:40 `torch_dtype=torch.float16`,
// experiments/eval/calibrate/llava_calibrate.py:163 `.half().cuda()`), bf16 for BASELINE config #2.
//
// vdd_llm_kernels.hip, vdd_prefill_kernels.hip and vdd_gemm.hip are each compiled ONCE PER STORAGE TYPE (`-DVDD_ELEM=2` bf16,
// `-DVDD_ELEM=1` fp16: the vdd_dtype values): the kernels address tensors as 16-bit words and go through the few functions below
// for everything that depends on the encoding - widening, the round-to-nearest-even narrowing every torch op of the reference
// applies, the packed pair forms, the dot-product and MFMA opcodes.  Accumulation is fp32 for both.  An instantiation lives in
// its own namespace (kernel symbols `vdd_bf16::gemm_kernel<...>` / `vdd_f16::gemm_kernel<...>` in a rocprof trace) and exports its C
// entries as `<name>_bf16` / `<name>_f16` with hidden visibility; vdd_model_dispatch.hip holds the public `int dtype` switch.
#ifndef VDD_ELEM_H
#define VDD_ELEM_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vdd_hip.h"

#ifndef VDD_ELEM
#error "compile the model kernels with -DVDD_ELEM=1 (fp16) or -DVDD_ELEM=2 (bf16)"
#endif

#if VDD_ELEM == 2
#define VDD_ELEM_NS vdd_bf16
#define VDD_IMPL(name) name##_bf16
#define VDD_ELEM_IS_BF16 1
#elif VDD_ELEM == 1
#define VDD_ELEM_NS vdd_f16
#define VDD_IMPL(name) name##_f16
#define VDD_ELEM_IS_BF16 0
#else
#error "VDD_ELEM must be 1 (VDD_F16) or 2 (VDD_BF16)"
#endif
#define VDD_HIDDEN __attribute__((visibility("hidden")))

namespace vdd_elem {

#if VDD_ELEM_IS_BF16
typedef __bf16 elem_t;
#else
typedef _Float16 elem_t;
#endif
typedef __attribute__((ext_vector_type(2))) elem_t ex2_t;
typedef __attribute__((ext_vector_type(4))) elem_t ex4_t;
typedef __attribute__((ext_vector_type(8))) elem_t ex8_t;      // one MFMA A / B operand: 8 elements = 16 bytes per lane
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// ---- widening: the low 16 bits of `b` / the two halves of a packed dword
__device__ __forceinline__ float e2f(uint32_t b) {
#if VDD_ELEM_IS_BF16
    return __builtin_bit_cast(float, b << 16);
#else
    return (float)__builtin_bit_cast(_Float16, (uint16_t)b);
#endif
}
__device__ __forceinline__ float lo(uint32_t w) {
#if VDD_ELEM_IS_BF16
    return __builtin_bit_cast(float, w << 16);
#else
    return (float)__builtin_bit_cast(ex2_t, w)[0];
#endif
}
__device__ __forceinline__ float hi(uint32_t w) {
#if VDD_ELEM_IS_BF16
    return __builtin_bit_cast(float, w & 0xFFFF0000u);
#else
    return (float)__builtin_bit_cast(ex2_t, w)[1];              // v_cvt_f32_f16 ... src0_sel:WORD_1
#endif
}

// ---- narrowing, round-to-nearest-even (what `tensor.to(dtype)` and every op of a half-precision torch model do); fp16 saturates
// to +-inf beyond 65504 exactly as torch does
__device__ __forceinline__ uint32_t f2e(float f) {
#if VDD_ELEM_IS_BF16
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
#else
    return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)f);          // v_cvt_f16_f32 (RNE under the default mode register)
#endif
}
__device__ __forceinline__ float rnd(float f) { return (float)(elem_t)f; }   // round to the element type, keep as float
// two fp32 -> one packed dword in ONE instruction (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32; NOT the round-toward-zero v_cvt_pkrtz)
__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, ex2_t));
}
__device__ __forceinline__ uint32_t pack(float a, float b) {
#if VDD_ELEM_IS_BF16
    return f2e(a) | (f2e(b) << 16);
#else
    return cvt_pk(a, b);
#endif
}
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
    const f32x4_t v = {a, b, c, d};
    return __builtin_bit_cast(uint2, __builtin_convertvector(v, ex4_t));
}

// ---- fp32 += a.x * b.x + a.y * b.y over one packed pair (v_dot2c_f32_bf16 / v_dot2c_f32_f16; the products are exact in fp32)
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc) {
#if VDD_ELEM_IS_BF16
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ex2_t, a), __builtin_bit_cast(ex2_t, b), acc, false);
#else
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(ex2_t, a), __builtin_bit_cast(ex2_t, b), acc, false);
#endif
}

// ---- matrix cores: same shapes, rates and fragment layouts for both encodings
template <class V>
__device__ __forceinline__ f32x4_t mfma16(V a, V b, f32x4_t c) {          // v_mfma_f32_16x16x32_{bf16,f16}
    static_assert(sizeof(V) == 16, "an MFMA operand is 8 x 16 bit per lane");
#if VDD_ELEM_IS_BF16
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ex8_t, a), __builtin_bit_cast(ex8_t, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ex8_t, a), __builtin_bit_cast(ex8_t, b), c, 0, 0, 0);
#endif
}
template <class V>
__device__ __forceinline__ f32x16_t mfma32(V a, V b, f32x16_t c) {        // v_mfma_f32_32x32x16_{bf16,f16}
    static_assert(sizeof(V) == 16, "an MFMA operand is 8 x 16 bit per lane");
#if VDD_ELEM_IS_BF16
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ex8_t, a), __builtin_bit_cast(ex8_t, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ex8_t, a), __builtin_bit_cast(ex8_t, b), c, 0, 0, 0);
#endif
}

}  // namespace vdd_elem

#endif /* VDD_ELEM_H */
