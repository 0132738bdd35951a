// Public C entries of the model kernels (include/vdd_hip.h): a switch on `dtype` over the two instantiations of
// vdd_llm_kernels.hip / vdd_prefill_kernels.hip / vdd_gemm.hip (compiled per storage type, csrc/vdd_elem.h), whose own entries are
// `<name>_bf16` / `<name>_f16` with hidden visibility.  Host code only.  The reference selects the dtype once, when it loads the
// checkpoint (experiments/llava/model/builder.py:40 fp16; BASELINE config #2 bf16); here it is an argument of every call.
#include <stdint.h>

#include "vdd_hip.h"

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

VDD_MODEL_FN(vdd_rmsnorm,
             VDD_P(const void* x, const void* delta, const float* delta_slabs, int n_slabs, const void* w, void* y, void* resid_out, int M, int d, float eps),
             VDD_P(x, delta, delta_slabs, n_slabs, w, y, resid_out, M, d, eps))
VDD_MODEL_FN(vdd_rope_kv_write,
             VDD_P(const void* qkv, const int* pos, const int* cpos, const int* slot, const float* cos_sin, void* q_out, void* k_cache, void* v_cache,
                   int M, int Hq, int Hkv, int D, int64_t slot_stride, int t_max),
             VDD_P(qkv, pos, cpos, slot, cos_sin, q_out, k_cache, v_cache, M, Hq, Hkv, D, slot_stride, t_max))
VDD_MODEL_FN(vdd_silu_mul, VDD_P(const void* gate_up, void* out, int64_t M, int F), VDD_P(gate_up, out, M, F))
VDD_MODEL_FN(vdd_embed, VDD_P(const int64_t* ids, const void* table, void* out, int M, int d, int vocab), VDD_P(ids, table, out, M, d, vocab))
VDD_MODEL_FN(vdd_embed_scatter, VDD_P(const int32_t* ids, const int32_t* rows, const void* table, void* out, int M, int d, int vocab),
             VDD_P(ids, rows, table, out, M, d, vocab))
VDD_MODEL_FN(vdd_skinny_gemm,
             VDD_P(const void* X, const void* W, const void* R, void* Y, float* Y_slabs, int n_split, int M, int N, int K, int64_t ldx, int64_t ldr, int64_t ldy),
             VDD_P(X, W, R, Y, Y_slabs, n_split, M, N, K, ldx, ldr, ldy))
VDD_MODEL_FN(vdd_skinny_swiglu, VDD_P(const void* X, const void* W_gate_up, void* act, int M, int F, int K, int64_t ldx),
             VDD_P(X, W_gate_up, act, M, F, K, ldx))
VDD_MODEL_FN(vdd_skinny_gemm_resid_ss,
             VDD_P(const void* X, const void* W, const void* R, void* Y, float* ss_out, int M, int N, int K, int64_t ldx, int64_t ldr, int64_t ldy),
             VDD_P(X, W, R, Y, ss_out, M, N, K, ldx, ldr, ldy))
VDD_MODEL_FN(vdd_skinny_gemm_normed,
             VDD_P(const void* H, const float* ss, int nss, const void* ln_w, float eps, const void* W, void* Y, int M, int N, int K, int64_t ldh, int64_t ldy),
             VDD_P(H, ss, nss, ln_w, eps, W, Y, M, N, K, ldh, ldy))
VDD_MODEL_FN(vdd_skinny_swiglu_normed,
             VDD_P(const void* H, const float* ss, int nss, const void* ln_w, float eps, const void* W_gate_up, void* act, int M, int F, int K, int64_t ldh),
             VDD_P(H, ss, nss, ln_w, eps, W_gate_up, act, M, F, K, ldh))
VDD_MODEL_FN(vdd_gemm,
             VDD_P(const void* X, const void* W, void* Y, const void* bias, const void* resid, int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldy,
                   int64_t ldr, int epilogue, int config, void* workspace, int64_t workspace_bytes),
             VDD_P(X, W, Y, bias, resid, M, N, K, ldx, ldw, ldy, ldr, epilogue, config, workspace, workspace_bytes))
VDD_MODEL_FN(vdd_decode_attention,
             VDD_P(const void* q, const void* k_cache, const void* v_cache, const void* k_prefix, const void* v_prefix, const int32_t* rows, void* out,
                   void* workspace, int M, int H, int Hkv, int D, int64_t slot_stride, int t_max, int64_t prefix_stride, int prefix_tmax, int max_len,
                   float scale),
             VDD_P(q, k_cache, v_cache, k_prefix, v_prefix, rows, out, workspace, M, H, Hkv, D, slot_stride, t_max, prefix_stride, prefix_tmax, max_len, scale))
VDD_MODEL_FN(vdd_decode_attention_fused,
             VDD_P(const void* qkv, const int32_t* pos, const int32_t* cpos, const int32_t* slot, const float* cos_sin, void* k_cache, void* v_cache,
                   const void* k_prefix, const void* v_prefix, const int32_t* rows, void* out, int M, int H, int Hkv, int D, int64_t slot_stride, int t_max,
                   int64_t prefix_stride, int prefix_tmax, float scale),
             VDD_P(qkv, pos, cpos, slot, cos_sin, k_cache, v_cache, k_prefix, v_prefix, rows, out, M, H, Hkv, D, slot_stride, t_max, prefix_stride,
                   prefix_tmax, scale))
VDD_MODEL_FN(vdd_decode_attention_fused_split,
             VDD_P(const void* qkv, const int32_t* pos, const int32_t* cpos, const int32_t* slot, const float* cos_sin, void* k_cache, void* v_cache,
                   const void* k_prefix, const void* v_prefix, const int32_t* rows, void* out, int M, int H, int Hkv, int D, int64_t slot_stride, int t_max,
                   int64_t prefix_stride, int prefix_tmax, float scale, void* workspace, int n_split),
             VDD_P(qkv, pos, cpos, slot, cos_sin, k_cache, v_cache, k_prefix, v_prefix, rows, out, M, H, Hkv, D, slot_stride, t_max, prefix_stride,
                   prefix_tmax, scale, workspace, n_split))
VDD_MODEL_FN(vdd_decode_attention_grouped,
             VDD_P(const void* q, const void* k_cache, const void* v_cache, const void* k_prefix, const void* v_prefix, const void* prefix_frag,
                   const int32_t* rows, const int32_t* groups, const int32_t* group_rows, const int32_t* items, int n_items, void* out, void* workspace,
                   int M, int H, int Hkv, int D, int64_t slot_stride, int t_max, int64_t prefix_stride, int prefix_tmax, int max_prefix_len,
                   int max_own_len, int prefix_chunks_per_item, float scale),
             VDD_P(q, k_cache, v_cache, k_prefix, v_prefix, prefix_frag, rows, groups, group_rows, items, n_items, out, workspace, M, H, Hkv, D,
                   slot_stride, t_max, prefix_stride, prefix_tmax, max_prefix_len, max_own_len, prefix_chunks_per_item, scale))
VDD_MODEL_FN(vdd_prefix_fragments,
             VDD_P(const void* k_prefix, const void* v_prefix, void* prefix_frag, const int32_t* prefix_len_of_slot, int n_slots, int Hkv, int t_max, int D),
             VDD_P(k_prefix, v_prefix, prefix_frag, prefix_len_of_slot, n_slots, Hkv, t_max, D))
VDD_MODEL_FN(vdd_flash_attention,
             VDD_P(const void* q, const void* k_cache, const void* v_cache, const void* k_prefix, const void* v_prefix, const int32_t* seqs, void* out,
                   int n_seq, int max_tq, int H, int Hkv, int D, int64_t slot_stride, int t_max, int64_t prefix_stride, int prefix_tmax, float scale,
                   int causal),
             VDD_P(q, k_cache, v_cache, k_prefix, v_prefix, seqs, out, n_seq, max_tq, H, Hkv, D, slot_stride, t_max, prefix_stride, prefix_tmax, scale, causal))
VDD_MODEL_FN(vdd_flash_attention_packed,
             VDD_P(const void* q, const void* k_cache, const void* v_cache, const void* k_prefix, const void* v_prefix, const int32_t* seqs,
                   const int32_t* packs, void* out, int n_packs, int H, int Hkv, int D, int64_t slot_stride, int t_max, int64_t prefix_stride,
                   int prefix_tmax, float scale),
             VDD_P(q, k_cache, v_cache, k_prefix, v_prefix, seqs, packs, out, n_packs, H, Hkv, D, slot_stride, t_max, prefix_stride, prefix_tmax, scale))
VDD_MODEL_FN(vdd_attention_probs,
             VDD_P(const void* q, const void* k_cache, const void* k_prefix, const int32_t* seq, void* out, int H, int Hkv, int D, int64_t slot_stride,
                   int t_max, int64_t prefix_stride, int prefix_tmax, float scale),
             VDD_P(q, k_cache, k_prefix, seq, out, H, Hkv, D, slot_stride, t_max, prefix_stride, prefix_tmax, scale))
VDD_MODEL_FN(vdd_vit_im2col, VDD_P(const void* images, int image_dtype, void* patches, int n, int S, int P, int Kp),
             VDD_P(images, image_dtype, patches, n, S, P, Kp))
VDD_MODEL_FN(vdd_vit_assemble, VDD_P(const void* emb, const void* cls, const void* pos, void* out, int n, int T, int width),
             VDD_P(emb, cls, pos, out, n, T, width))
VDD_MODEL_FN(vdd_vit_qkv_split,
             VDD_P(const void* qkv, void* q, void* k_cache, void* v_cache, int n, int T, int H, int D, int64_t slot_stride, int t_max, int parts),
             VDD_P(qkv, q, k_cache, v_cache, n, T, H, D, slot_stride, t_max, parts))
VDD_MODEL_FN(vdd_add, VDD_P(const void* a, const void* b, void* out, int64_t n), VDD_P(a, b, out, n))
VDD_MODEL_FN(vdd_layernorm, VDD_P(const void* x, const void* w, const void* b, void* y, int M, int d, float eps), VDD_P(x, w, b, y, M, d, eps))
VDD_MODEL_FN(vdd_bias_act, VDD_P(const void* x, const void* bias, void* y, int64_t M, int d, int act), VDD_P(x, bias, y, M, d, act))

// workspace sizes do not depend on the storage type (fp32 partials, int32 counters): one instantiation answers
VDD_HIDDEN int64_t vdd_gemm_workspace_bytes_bf16(int M, int N);
VDD_HIDDEN int64_t vdd_decode_attention_workspace_bytes_bf16(int M, int H, int D, int max_len);
VDD_HIDDEN int64_t vdd_decode_attention_fused_split_workspace_bytes_bf16(int M, int H, int n_split);
int64_t vdd_gemm_workspace_bytes(int M, int N) { return vdd_gemm_workspace_bytes_bf16(M, N); }
int64_t vdd_decode_attention_workspace_bytes(int M, int H, int D, int max_len) { return vdd_decode_attention_workspace_bytes_bf16(M, H, D, max_len); }
int64_t vdd_decode_attention_fused_split_workspace_bytes(int M, int H, int n_split) {
    return vdd_decode_attention_fused_split_workspace_bytes_bf16(M, H, n_split);
}

}  // extern "C"
