// Public C entries of the laboratory library (vdd_lost.h): the dtype switch over the two instantiations of vdd_skinny_slab.hip /
// vdd_layer_persistent.hip, as csrc/vdd_model_dispatch.hip does for the product.  Host code only.
#include <stdint.h>

#include "vdd_lost.h"

#define VDD_HIDDEN __attribute__((visibility("hidden")))
#define VDD_P(...) __VA_ARGS__
// one model entry: the two instantiations' prototypes (signature without `dtype`) and the public switch
#define VDD_MODEL_FN(name, params, args)                                      \
    VDD_HIDDEN int name##_bf16(params, void* stream);                         \
    VDD_HIDDEN int name##_f16(params, void* stream);                          \
    int name(params, int dtype, void* stream) {                               \
        if (dtype == VDD_BF16) return name##_bf16(args, stream);              \
        if (dtype == VDD_F16) return name##_f16(args, stream);                \
        return VDD_ERR_INVALID_ARG; /* fp32 models are not served */          \
    }

extern "C" {

VDD_MODEL_FN(vdd_skinny_slab,
             VDD_P(const void* X, const float* ss, int nss, const void* ln_w, float eps, const void* W, const void* R, void* Y, float* ss_out, int M, int N,
                   int K, int64_t ldx, int64_t ldr, int64_t ldy, int swiglu, void* workspace, int64_t workspace_bytes),
             VDD_P(X, ss, nss, ln_w, eps, W, R, Y, ss_out, M, N, K, ldx, ldr, ldy, swiglu, workspace, workspace_bytes))
VDD_MODEL_FN(vdd_decode_layers,
             VDD_P(const vdd_layer_desc* layers, int n_layers, const void* resid_in, void* resid_out, float* ss_out, const int32_t* pos,
                   const int32_t* cpos, const int32_t* slot, const float* cos_sin, const int32_t* rows, int M, int d, int H, int Hkv, int F, int D,
                   float eps, float scale, int64_t slot_stride, int t_max, int64_t prefix_stride, int prefix_tmax, int has_qkv_bias,
                   void* workspace, int64_t workspace_bytes),
             VDD_P(layers, n_layers, resid_in, resid_out, ss_out, pos, cpos, slot, cos_sin, rows, M, d, H, Hkv, F, D, eps, scale, slot_stride, t_max,
                   prefix_stride, prefix_tmax, has_qkv_bias, workspace, workspace_bytes))
VDD_HIDDEN int vdd_decode_layers_max_rows_bf16(int d, int H, int F, int D, int n_layers);
VDD_HIDDEN int64_t vdd_decode_layers_workspace_bytes_bf16(int M, int d, int H, int F, int D);
VDD_HIDDEN int vdd_decode_layers_ss_cols_bf16(int d, int H, int F, int D);
int vdd_decode_layers_ss_cols(int d, int H, int F, int D, int dtype) {
    return (dtype == VDD_BF16 || dtype == VDD_F16) ? vdd_decode_layers_ss_cols_bf16(d, H, F, D) : 0;
}
int vdd_decode_layers_max_rows(int d, int H, int F, int D, int n_layers, int dtype) {
    return (dtype == VDD_BF16 || dtype == VDD_F16) ? vdd_decode_layers_max_rows_bf16(d, H, F, D, n_layers) : 0;
}
int64_t vdd_decode_layers_workspace_bytes(int M, int d, int H, int F, int D, int dtype) {
    return (dtype == VDD_BF16 || dtype == VDD_F16) ? vdd_decode_layers_workspace_bytes_bf16(M, d, H, F, D) : 0;
}

VDD_HIDDEN int64_t vdd_skinny_slab_workspace_bytes_bf16(int M, int N, int K, int swiglu);
VDD_HIDDEN int64_t vdd_skinny_slab_status_offset_bf16(void);
int64_t vdd_skinny_slab_workspace_bytes(int M, int N, int K, int swiglu) { return vdd_skinny_slab_workspace_bytes_bf16(M, N, K, swiglu); }
int64_t vdd_skinny_slab_status_offset(void) { return vdd_skinny_slab_status_offset_bf16(); }

}  // extern "C"
