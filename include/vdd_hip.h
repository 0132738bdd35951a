/*
 * vdd_hip.h — C ABI of the MI355X-native visual-contrastive-decoding core (libvdd_hip.so).
 *
 * Drop-in boundary for the hot path of yfzhang114/LLaVA-Align (reference paths are
 * relative to that repo).  The reference is pure Python with no FFI, so these entry
 * points are what a maintainer would bind (ctypes stub: INTEGRATION.md) to replace:
 *
 *   vdd_add_diffusion_noise  vcd_utils/vcd_add_noise.py:18-22 (q(x_t | x_0) of the noisy-image branch)
 *   vdd_contrast_sample      vcd_utils/vcd_sample.py:185-207,257-260,285-288
 *                            (both-branch average, contrast, adaptive-plausibility
 *                             mask, HF temperature/top-k/top-p warpers, softmax,
 *                             multinomial, pad-after-EOS, unfinished update) and
 *                            experiments/utils/metrics.py:102-104 (softmax -> top-k
 *                             probabilities of the step-0 scores row).
 *
 * Conventions: plain pointers and sizes only (no torch types); every pointer is a
 * DEVICE pointer owned by the caller unless stated otherwise; the library never
 * allocates, frees or synchronises; all work is enqueued on the given hipStream_t
 * (passed as void*); functions return 0 on success or a negative vdd_status code and
 * never throw or exit.  Re-entrant; no global mutable state.
 */
#ifndef VDD_HIP_H
#define VDD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDD_ABI_VERSION 3

typedef enum vdd_status {
    VDD_OK = 0,
    VDD_ERR_INVALID_ARG = -1,   /* null/negative/misaligned argument, see vdd_last_error() */
    VDD_ERR_UNSUPPORTED = -2,   /* reserved */
    VDD_ERR_LAUNCH = -3         /* hipLaunchKernel failed */
} vdd_status;

typedef enum vdd_dtype { VDD_F32 = 0, VDD_F16 = 1, VDD_BF16 = 2 } vdd_dtype;

/* flags */
#define VDD_PICK_ARGMAX        (1u << 0) /* token = first index of the row maximum of the final scores
                                            (deterministic stand-in for multinomial when one survivor) */
#define VDD_CUTOFF_F32_SCALAR  (1u << 1) /* cutoff = fl(max + log_beta) with log_beta kept in fp32 (torch-GPU
                                            CPU-scalar path); default demotes log_beta to dtype first
                                            (torch-CPU, what the golden vectors pin) */
#define VDD_TEMP_RECIPROCAL    (1u << 2) /* scores * (1/T) (torch-GPU div-by-scalar) instead of scores / T */
#define VDD_TOPP_FP32_MASS     (1u << 4) /* top-p by integrating the fp32 softmax mass (equal scores kept or dropped together)
                                            instead of the reference's model-dtype softmax -> sequential cumsum arithmetic;
                                            rows keeping more than vdd_topp_exact_max() candidates always take this form */
#define VDD_NO_SAMPLE          (1u << 3) /* only produce scores_out (used when a Python logits_processor
                                            must run between contrast and warp) */

/* row_status values */
#define VDD_ROW_OK 0
#define VDD_ROW_EMPTY 1   /* no finite score left or NaN present: the reference's torch.multinomial raises */

typedef struct vdd_sample_params {
    uint32_t abi_version;        /* = VDD_ABI_VERSION */
    uint32_t flags;
    /* ---- inputs: last-position logits of each branch, row-major [B, V] ---- */
    const void* logit_v;         /* main (image) branch                       vcd_sample.py:119 */
    const void* logit_cd;        /* <unk> / noisy-image branch, NULL = plain path      :169    */
    const void* logit_dd;        /* image-token-dropped branch, NULL unless both modes :184    */
    int64_t stride_v, stride_cd, stride_dd;   /* row strides in ELEMENTS */
    int32_t B, V;
    int32_t dtype;               /* vdd_dtype: the model dtype; all arithmetic is rounded in it */
    int32_t min_keep;            /* HF min_tokens_to_keep (>=1) */
    double alpha;                /* cd_alpha                                           :188    */
    double log_beta;             /* fp32 value of log(cd_beta) as torch.log(torch.tensor(beta)) :191 */
    double temperature;          /* <=0 or ==1: no temperature warper */
    double top_p;                /* >=1 or <0: no top-p warper */
    int32_t top_k;               /* <=0: no top-k warper */
    int32_t n_eos;
    /* ---- sampling ---- */
    uint64_t philox_seed, philox_offset;   /* u_row = philox4x32-10(seed; offset, row) */
    const float* uniforms;       /* optional [B]: explicit u in [0,1) overriding philox */
    /* ---- EOS / pad bookkeeping (vcd_sample.py:257-260,285-288) ---- */
    const int64_t* eos_ids;      /* [n_eos] or NULL */
    int64_t pad_id;              /* used iff unfinished != NULL && n_eos > 0 */
    int64_t* unfinished;         /* optional in/out [B] (1 = still generating) */
    /* ---- outputs ---- */
    int64_t* next_tokens;        /* [B] (may be NULL iff VDD_NO_SAMPLE); row r at next_tokens[r*stride_tokens] */
    int64_t stride_tokens;       /* >=1; lets the caller point at column `cur_len` of its [B, max_len] id buffer */
    void* scores_out;            /* optional [B, V] dtype: post-warp scores (what output_scores returns) */
    int64_t stride_scores;
    float* top_prob;             /* optional [B, n_top] softmax(scores) top-n, descending */
    int64_t* top_tok;            /* optional [B, n_top] */
    int32_t n_top;               /* <= 16 */
    int32_t _pad0;
    int32_t* row_status;         /* optional [B] */
    /* ---- scratch ---- */
    void* workspace;             /* optional [B, V] dtype: working row for V > vdd_lds_row_capacity(dtype)
                                    when scores_out is NULL (rows that fit LDS never touch it) */
    int64_t stride_workspace;
    const uint64_t* philox_offset_ptr;   /* optional device counter ADDED to philox_offset at run time, so a launch
                                            captured in a HIP graph draws fresh numbers on every replay */
    /* ---- logits-processor stage: runs where vcd_sample.py:197 calls `logits_processor(input_ids, scores)`, i.e. after the
     *      contrast + plausibility mask (or on the raw logits of the plain path, :204) and BEFORE the warpers ---- */
    const int32_t* eos_min_step; /* optional [B]: while (step + *step_ptr) < eos_min_step[row] every id of eos_ids scores -inf:
                                    HF MinNewTokensLengthLogitsProcessor (min_new_tokens, run_qwen.py:194) and
                                    MinLengthLogitsProcessor (min_length - prompt_length, blip2_vicuna_instruct.py:397) */
    int64_t step;                /* new tokens this row set has produced so far (0 at the prefill step) */
    const int64_t* step_ptr;     /* optional device counter ADDED to step at run time (graph replay) */
    const int32_t* force_eos;    /* optional [B] flags: where non-zero, scores[row, force_eos_id] = force_eos_value - the Qwen
                                    StopWordsLogitsProcessor (qwen_generation_utils.py:352-359; flags from vdd_stop_words_match);
                                    applied after eos_min_step, as HF orders custom processors behind its own */
    int64_t force_eos_id;
    double force_eos_value;      /* the reference uses float(2**15) */
} vdd_sample_params;

/* Fused per-step contrastive sampling tail; one launch for all B rows. */
int vdd_contrast_sample(const vdd_sample_params* p, void* hip_stream);

/* Stop-sequence matcher of the Qwen StopWordsLogitsProcessor (qwen_generation_utils.py:361-385): force_out[row] = 1 iff the ids of
 * the row so far END with one of the stop sequences.  The ids of a row are its prompt tail (prompt_tail [B, tail_len] int64, the
 * LAST tail_len prompt tokens, left-padded with -1) followed by its (step + *step_ptr) generated tokens gen[row * ld_gen + 0..].
 * stop_flat / stop_off: the n_stop sequences back to back and their offsets (n_stop + 1 entries).  One thread per row. */
int vdd_stop_words_match(const int64_t* gen, int64_t ld_gen, int64_t step, const int64_t* step_ptr, const int64_t* prompt_tail,
                         int tail_len, const int64_t* stop_flat, const int32_t* stop_off, int n_stop, int32_t* force_out, int B,
                         void* hip_stream);

/* HF RepetitionPenaltyLogitsProcessor on a [B, V] scores matrix in place (blip2_vicuna_instruct.py:400 passes repetition_penalty):
 * for every DISTINCT id in the row's history (prompt_ids [B, prompt_len] int64, -1 = padding, then (step + *step_ptr) generated
 * tokens of gen) score = score < 0 ? score * penalty : score / penalty, computed from the un-penalised value and rounded in
 * `dtype`.  flags & VDD_TEMP_RECIPROCAL: the division as a multiply by fl32(1 / penalty) (torch-GPU), else a true division
 * (torch-CPU).  History length (prompt_len + step) <= 8192. */
int vdd_repetition_penalty(void* scores, int64_t stride, int dtype, int B, int V, const int64_t* prompt_ids, int prompt_len,
                           const int64_t* gen, int64_t ld_gen, int64_t step, const int64_t* step_ptr, float penalty,
                           uint32_t flags, void* hip_stream);

/* VCD branch input: x_t = sqrt(abar_t) x_0 + sqrt(1-abar_t) eps   (vcd_utils/vcd_add_noise.py:18-22).
 * x, y: n elements of `dtype` (may alias).  eps: optional explicit fp32 noise [n]; NULL draws
 * N(0,1) from philox4x32-10(seed; offset + i/4) + Box-Muller.  The two scalars are row t of the
 * reference's sigmoid schedule (:7-16), tabulated by the host. */
int vdd_add_diffusion_noise(const void* x, void* y, int64_t n, int dtype, float sqrt_abar,
                            float sqrt_one_minus_abar, const float* eps, uint64_t seed,
                            uint64_t offset, void* hip_stream);

/* ---------------------------------------------------------------------------------------------
 * Branch-batched language-model step (device pointers, row-major).
 * These replace what the reference runs per branch and per token through HF's eager LlamaModel
 * (experiments/llava/model/language_model/llava_llama.py:88-103; vcd_sample.py:109,163,178):
 * rows of one call = (question, branch) pairs, so weights stream from HBM once per step.
 *
 * `dtype` (the argument in front of hip_stream of every model entry): VDD_BF16 or VDD_F16 - the storage type of EVERY 16-bit tensor
 * of the call (activations, weights, KV caches); VDD_F32 is refused.  The reference runs the model in the checkpoint's dtype:
 * fp16 in every released driver (experiments/llava/model/builder.py:40 `torch_dtype=torch.float16`, llava_calibrate.py:163
 * `.half().cuda()`), bf16 in BASELINE config #2.  Accumulation is fp32 for both; "bf16(...)" in the comments below marks a
 * round-to-nearest-even to the call's dtype (where the eager torch op of the reference rounds).  Both instantiations are the same
 * source compiled per storage type (csrc/vdd_elem.h): same tile shapes, same summation orders, MFMA / dot opcodes of the type.
 * ------------------------------------------------------------------------------------------- */

/* h = x (+ delta); resid_out = h (optional); y = bf16(bf16(h * rsqrt(mean h^2 + eps)) * w).  d % 8 == 0, d <= 8192.
 * delta is either bf16 [M, d] or, as delta_slabs, the n_slabs fp32 split-K partials [n_slabs][M][d] of vdd_skinny_gemm
 * (summed, rounded to bf16, then added - the same roundings as a bf16 GEMM output followed by the residual add). */
int vdd_rmsnorm(const void* x, const void* delta, const float* delta_slabs, int n_slabs, const void* w, void* y, void* resid_out,
                int M, int d, float eps, int dtype, void* hip_stream);

/* qkv [M, (Hq+2Hkv)*D] -> q_out [M, Hq, D] with rotary embedding at pos[row] (HF rotate_half pairing;
 * cos_sin fp32 [max_pos, D/2, 2]); k (rotated) and v are written to cache[slot[row]][kv_head][cpos[row]][D]
 * (cpos = index inside the slot: pos for a prefix slot, pos - prefix_len for a compact own slot;
 * cache slot stride in elements; t_max tokens per slot). */
int vdd_rope_kv_write(const void* qkv, const int* pos, const int* cpos, const int* slot, const float* cos_sin, void* q_out,
                      void* k_cache, void* v_cache, int M, int Hq, int Hkv, int D, int64_t slot_stride, int t_max, int dtype, void* hip_stream);

/* out[m, f] = silu(gate_up[m, f]) * gate_up[m, F + f]. */
int vdd_silu_mul(const void* gate_up, void* out, int64_t M, int F, int dtype, void* hip_stream);

int vdd_embed(const int64_t* ids, const void* table, void* out, int M, int d, int vocab, int dtype, void* hip_stream);

/* Token-embedding gather written straight into the packed prefill matrix: out[rows[m], :] = table[ids[m], :] (int32 ids and
 * rows; the splice of llava_arch.py:122-163 - text chunks embedded around the image features - without an intermediate
 * [M, d] tensor and an index_put). */
int vdd_embed_scatter(const int32_t* ids, const int32_t* rows, const void* table, void* out, int M, int d, int vocab, int dtype, void* hip_stream);

/* Y[M,N] = X[M,K] W[N,K]^T (+ R[M,N]); M <= 64, K % (128 * n_split) == 0; W is read from HBM exactly once.
 * n_split > 1 (with Y_slabs fp32 [n_split][M][N], Y may be NULL): split-K across blocks too; the slabs are summed by
 * vdd_rmsnorm's delta_slabs input.  With n_split = 1 and N <= 8192 (N = 4096 gives only 256 column blocks) a block runs eight
 * waves that split K eight ways inside it. */
int vdd_skinny_gemm(const void* X, const void* W, const void* R, void* Y, float* Y_slabs, int n_split, int M, int N, int K,
                    int64_t ldx, int64_t ldr, int64_t ldy, int dtype, void* hip_stream);

/* act[M,F] = silu(X Wg^T) * (X Wu^T) for W_gate_up = [Wg; Wu] ([2F, K] row-major), M <= 16 with K % 128 == 0 (8 features per
 * workgroup) or 17 <= M <= 64 with K % 256 == 0 (16 features per workgroup, gate + up tiles on 16x16x32 MFMAs): the gate/up
 * projection and SiLU*mul of the Llama MLP (HF LlamaMLP.forward [ext] under llava_llama.py:88-103) in one weight-streaming
 * launch; replaces vdd_skinny_gemm(N = 2F) + vdd_silu_mul with the same bf16 rounding points. */
int vdd_skinny_swiglu(const void* X, const void* W_gate_up, void* act, int M, int F, int K, int64_t ldx, int dtype, void* hip_stream);

/* Small-M (M <= 16: one or a few questions in flight, the reference's own B = 1 regime) fusion of the decoder layer's two RMSNorm
 * launches into the weight-streaming projections around them (HF LlamaDecoderLayer [ext] under llava_llama.py:88-103):
 *   vdd_skinny_gemm_resid_ss   Y = bf16(bf16(X W^T) + R): the d-wide projection (attention output / MLP down) writes the NEW residual
 *                              stream and ss_out[row * ceil(N/16) + block] = the sum of squares of the block's 16 columns of that row;
 *   vdd_skinny_gemm_normed     Y = rmsnorm(H) W^T with rmsnorm(H)[k] = bf16(bf16(H[k] * rstd) * ln_w[k]), rstd = rsqrt(sum(ss[row][0..nss))
 *                              / K + eps): H is the un-normalised residual stream (row length K), normalised ONCE per workgroup into
 *                              LDS.  K % 256 == 0, K <= 8192, nss % 4 == 0, nss <= 512, M * 2 K <= 142 KiB (VDD_ERR_UNSUPPORTED beyond:
 *                              the caller takes vdd_rmsnorm + vdd_skinny_gemm);
 *   vdd_skinny_swiglu_normed   the same input form for the gate/up projection + SiLU*mul (vdd_skinny_swiglu).
 * The partial sums are added in a fixed order (deterministic); same bf16 rounding points as vdd_rmsnorm + the plain projections. */
int vdd_skinny_gemm_resid_ss(const void* X, const void* W, const void* R, void* Y, float* ss_out, int M, int N, int K, int64_t ldx,
                             int64_t ldr, int64_t ldy, int dtype, void* hip_stream);
int vdd_skinny_gemm_normed(const void* H, const float* ss, int nss, const void* ln_w, float eps, const void* W, void* Y, int M, int N,
                           int K, int64_t ldh, int64_t ldy, int dtype, void* hip_stream);
int vdd_skinny_swiglu_normed(const void* H, const float* ss, int nss, const void* ln_w, float eps, const void* W_gate_up, void* act,
                             int M, int F, int K, int64_t ldh, int dtype, void* hip_stream);

/* Row-batched projection GEMM, any M above the skinny regime (csrc/vdd_gemm.hip): Y[M,N] = epilogue(X[M,K] W[N,K]^T), bf16 in,
 * fp32 accumulate (32x32x16 MFMA, both operands LDS-DMA'd into swizzled LDS tiles, persistent stream-K over one workgroup
 * per CU), bf16 out.  K % 128 == 0, N % 4 == 0, ldx/ldw % 8 == 0 (elements).  Replaces the eager nn.Linear calls of HF
 * LlamaModel / CLIPVisionModel / the mlp2x_gelu projector (llava_llama.py:88-103, clip_encoder.py:39-51,
 * multimodal_projector/builder.py:33-46) at batch.
 * epilogue: VDD_GEMM_NONE; _BIAS y = bf16(acc + bias[n]); _BIAS_QUICK_GELU / _BIAS_GELU act(bf16(acc + bias));
 *   _SWIGLU  W = [Wgate; Wup] (2N rows), Y[M, N] = bf16(bf16(silu(bf16 gate)) * bf16 up), N % 128 == 0;
 *   _BIAS_RESID y = bf16(bf16(acc + bias) + resid[m, n]).
 * workspace: caller-owned device scratch of >= vdd_gemm_workspace_bytes(M, N) bytes (the same for every shape: a 4-MiB arrival-
 *   counter region, one int32 per output tile - at most 2^20 tiles per launch - followed by one fp32 partial tile per workgroup)
 *   whose counter region is ZERO before the first call (counters of tiles cut across workgroups; every completed call leaves
 *   them zero again; after a launch that did NOT complete - device fault, abort - zero the region before the next call).  One
 *   buffer per stream: launches on one stream may share it, launches that can run concurrently must not.
 * config: low 4 bits = macro tile (0 = 256x256; 1..8 = 256x256, 128x256, 256x128, 192x256, 256x192, 192x192, 192x128 (these
 *         three: no SwiGLU), 64x256 (a few dozen rows: the launch is a W stream); 9..11 = 128x128 (no SwiGLU), 128x128 and 192x128 on four
 *         waves: W streams for 65 - 256 rows with 3 - 5 K-tile buffers in flight; 12..15 = 64x128 on 4 / 8 waves, 32x128 on 4 / 2 waves (13, 14:
 *         no SwiGLU): up to 64 / 32 rows, six / eight buffers); bits 6-7 = the tile id's high bits (id 16 = 96x128 on 4 waves, no SwiGLU:
 *         batches that are a multiple of 96 rows rather than of 128); bits 4-5 = schedule (0 hybrid: whole-tile rounds + stream-K remainder, 1 data-parallel only,
 *         2 stream-K only).  Every choice writes the same result up to the fp32 summation order of a K-split tile.
 *         Schedule 3 = split-K SLABS (epilogue NONE only): bits 8-15 = S (1 .. K / 128); Y is then an fp32 buffer [S][M][N] (ldy = N,
 *         16-byte aligned) receiving the S partial products of every output tile - one (tile, K part) per workgroup, no fix-up, no
 *         waiting - for a consumer that adds them (vdd_rmsnorm's delta_slabs: the attention-output / MLP-down projections of a few
 *         dozen rows, whose 16 - 32 output tiles would otherwise be finished by one workgroup each reading 15 partial tiles in turn). */
#define VDD_GEMM_NONE 0
#define VDD_GEMM_BIAS 1
#define VDD_GEMM_BIAS_QUICK_GELU 2
#define VDD_GEMM_BIAS_GELU 3
#define VDD_GEMM_SWIGLU 4
#define VDD_GEMM_BIAS_RESID 5
int64_t vdd_gemm_workspace_bytes(int M, int N);
int vdd_gemm(const void* X, const void* W, void* Y, const void* bias, const void* resid, int M, int N, int K,
             int64_t ldx, int64_t ldw, int64_t ldy, int64_t ldr, int epilogue, int config, void* workspace, int64_t workspace_bytes,
             int dtype, void* hip_stream);

/* One query per (row, head) over that row's KV: rows[m] = {slot, len, prefix_slot, prefix_len} (int32 x4);
 * tokens [0, prefix_len) are read from the PREFIX pool (k_prefix/v_prefix, slot prefix_slot, index t: a shared
 * prompt prefix), tokens [prefix_len, len) from the OWN pool (k_cache/v_cache, slot `slot`, index t - prefix_len).
 * The two pools may be the same buffer. D == 128.
 * Split-KV: the key range is cut in 64-key chunks processed by independent waves (partials in `workspace`,
 * vdd_decode_attention_workspace_bytes(M, H, D, max_len) bytes, max_len >= every rows[m].len) and merged. */
int vdd_decode_attention(const void* q, const void* k_cache, const void* v_cache, const void* k_prefix, const void* v_prefix,
                         const int32_t* rows, void* out, void* workspace, int M, int H, int Hkv, int D, int64_t slot_stride, int t_max,
                         int64_t prefix_stride, int prefix_tmax, int max_len, float scale, int dtype, void* hip_stream);
int64_t vdd_decode_attention_workspace_bytes(int M, int H, int D, int max_len);

/* Small-M decode attention with RoPE, the KV-cache write and the chunk merge fused into ONE launch (the reference's own
 * operating point: one question = 2-3 branch rows per step, where the step is launch-latency bound).  `qkv` is the
 * un-rotated projection [M, (H + 2 Hkv) * D]; the new token's K/V are rotated in registers, stored at index cpos[row]
 * of slot[row] and attended from registers; older tokens come from the prefix / own pools exactly as in
 * vdd_decode_attention.  rows[i].len counts the new token.  Replaces vdd_rope_kv_write + vdd_decode_attention
 * (HF LlamaAttention.forward [ext] under llava_llama.py:88-103) for M <= ~16. */
int vdd_decode_attention_fused(const void* qkv, const int32_t* pos, const int32_t* cpos, const int32_t* slot, const float* cos_sin,
                               void* k_cache, void* v_cache, const void* k_prefix, const void* v_prefix, const int32_t* rows, void* out,
                               int M, int H, int Hkv, int D, int64_t slot_stride, int t_max, int64_t prefix_stride, int prefix_tmax,
                               float scale, int dtype, void* hip_stream);

/* The same launch with the old keys of every (row, head) cut into n_split (2 or 4) slices, one workgroup each: for one or two
 * questions in flight H x M workgroups leave most of the chip idle and each of them walks its context in three dependent fetch
 * rounds.  A slice leaves an un-normalised partial in `workspace`; the workgroup that finishes last merges the n_split partials
 * in slice order (deterministic) and writes the output row.  `workspace`: vdd_decode_attention_fused_split_workspace_bytes(M, H,
 * n_split) bytes, ZEROED once by the caller (the per-(row, head) tickets behind the partials return to zero after every launch);
 * one workspace per stream. */
int64_t vdd_decode_attention_fused_split_workspace_bytes(int M, int H, int n_split);
int vdd_decode_attention_fused_split(const void* qkv, const int32_t* pos, const int32_t* cpos, const int32_t* slot, const float* cos_sin,
                                     void* k_cache, void* v_cache, const void* k_prefix, const void* v_prefix, const int32_t* rows, void* out,
                                     int M, int H, int Hkv, int D, int64_t slot_stride, int t_max, int64_t prefix_stride, int prefix_tmax,
                                     float scale, void* workspace, int n_split, int dtype, void* hip_stream);

/* Same result as vdd_decode_attention when every row with prefix_len > 0 is listed in exactly one group of rows
 * sharing (prefix_slot, prefix_len): groups[g] = {row_off, n_rows, prefix_slot, prefix_len} (int32 x4) indexes
 * group_rows[]; items[i] = {group, first_row_of_16_row_slice, item index inside the prefix, 0} (int32 x4) is the
 * host-built work list of the prefix pass (one wave per item and head); item j of a group covers keys
 * [j * 64 * prefix_chunks_per_item, (j + 1) * 64 * prefix_chunks_per_item) of its prefix.
 * The prefix pass runs on MFMA from prefix_frag (the vdd_prefix_fragments image of k_prefix / v_prefix; required when
 * n_items > 0): the rows of a slice are the M dimension, each prefix byte leaves HBM once per group, and an item walks its
 * 64-key chunks with an online softmax so that it leaves ONE partial per (row, head).  Own tokens: one wave per (row, head)
 * that also merges the prefix partials (own ranges <= 256 keys), else the split-KV kernel + one combine.
 * Workspace: vdd_decode_attention_workspace_bytes(M, H, D, round_up(max_prefix_len, 64) + round_up(max_own_len, 64)). */
int vdd_decode_attention_grouped(const void* q, const void* k_cache, const void* v_cache, const void* k_prefix, const void* v_prefix,
                                 const void* prefix_frag /* vdd_prefix_fragments image, [n_slots][Hkv][2 * prefix_tmax][D] */,
                                 const int32_t* rows, const int32_t* groups, const int32_t* group_rows, const int32_t* items, int n_items,
                                 void* out, void* workspace, int M, int H, int Hkv, int D, int64_t slot_stride, int t_max,
                                 int64_t prefix_stride, int prefix_tmax, int max_prefix_len, int max_own_len,
                                 int prefix_chunks_per_item, float scale, int dtype, void* hip_stream);

/* Fragment-major copy of the prefix pool for the MFMA prefix pass: prefix_frag[slot][kv_head][chunk] = one 32-KiB block per
 * 64-key chunk (t_max % 64 == 0; twice the bytes and slot stride of k_prefix), 16 K fragments then 16 V^T fragments, each
 * the 1-KiB lane-linear image of one 16x16x32 MFMA operand; keys >= prefix_len_of_slot[slot] are zero-filled.  Built once
 * after the prefix prefill; every fragment load of the decode pass is then one contiguous KiB straight from HBM. */
int vdd_prefix_fragments(const void* k_prefix, const void* v_prefix, void* prefix_frag, const int32_t* prefix_len_of_slot, int n_slots,
                         int Hkv, int t_max, int D, int dtype, void* hip_stream);

/* Prefill attention (MFMA, flash-style).  q/out [Ttot, H*D] packed by sequence; seqs[s] = {q_row0, Tq, pos0,
 * slot, prefix_slot, prefix_len} (int32 x6): query i of sequence s sits at position pos0+i and attends keys
 * [0, pos0+i] (causal) or [0, pos0+Tq) (non-causal) read from the caches (prefix pool below prefix_len, own
 * pool at index t - prefix_len above, as in vdd_decode_attention), whose K/V for the new tokens must already be
 * written.  D in {64, 128}.
 * Replaces the eager attention of LlamaModel / CLIPVisionModel at step 0 (llava_arch.py:82-204). */
int vdd_flash_attention(const void* q, const void* k_cache, const void* v_cache, const void* k_prefix, const void* v_prefix,
                        const int32_t* seqs, void* out, int n_seq, int max_tq, int H, int Hkv, int D, int64_t slot_stride, int t_max,
                        int64_t prefix_stride, int prefix_tmax, float scale, int causal, int dtype, void* hip_stream);

/* The same attention for SHORT sequences (Tq <= 32 each, causal, D = 128) that continue shared prefixes - the suffix pass of the
 * prefill: packs[p] = 4 sequence indices (int32 x4, -1 = none) with the SAME (prefix_slot, prefix_len); a workgroup takes one
 * pack and head, stages the 64-key tiles inside the prefix once for its four waves and the rest of each sequence's keys one
 * sequence after the other.  Same result as vdd_flash_attention on the same descriptors. */
int vdd_flash_attention_packed(const void* q, const void* k_cache, const void* v_cache, const void* k_prefix, const void* v_prefix,
                               const int32_t* seqs, const int32_t* packs, void* out, int n_packs, int H, int Hkv, int D,
                               int64_t slot_stride, int t_max, int64_t prefix_stride, int prefix_tmax, float scale, int dtype, void* hip_stream);

/* The attention PROBABILITIES of one prompt, materialised: out [H, Tq, Tk] (Tk = pos0 + Tq) of the call's dtype,
 * out[h][i][t] = softmax over t <= pos0 + i of (q_i . k_t) * scale computed in fp32 and rounded (HF's eager attention), 0 behind the
 * diagonal.  q [.., H*D] rotated queries (row q_row0 + i), keys from [prefix slot | own slot] as in vdd_flash_attention; `seq` is a
 * HOST array {q_row0, Tq, pos0, slot, prefix_slot, prefix_len}.  D == 128, Tk <= 16384.  This is what the reference's POPE driver
 * reads of generate(output_attentions=True): model_outputs['attentions'][0][-1], step 0 / last layer (llava_calibrate.py:175,180-182);
 * the flash kernels of the path never build the matrix. */
int vdd_attention_probs(const void* q, const void* k_cache, const void* k_prefix, const int32_t* seq /* host */, void* out, int H, int Hkv,
                        int D, int64_t slot_stride, int t_max, int64_t prefix_stride, int prefix_tmax, float scale, int dtype, void* hip_stream);

/* ViT front-end glue around the patch-embed GEMM (HF CLIPVisionEmbeddings / CLIPAttention as run by clip_encoder.py:39-51):
 * im2col of the stride-P patch convolution (images [n,3,S,S] of vdd_dtype `image_dtype`: fp32 / fp16 / bf16 -> patches of the model `dtype` [n*(S/P)^2, Kp], zero
 * padded from 3*P*P to Kp columns); class token + position embeddings (h[i,t] = (t ? emb[i*(T-1)+t-1] : cls) + pos[t]);
 * the fused qkv projection [n*T, 3, H, D] split into q [n*T, H*D] and the K / V caches [image][H][t_max][D]. */
int vdd_vit_im2col(const void* images, int image_dtype, void* patches, int n, int S, int P, int Kp, int dtype, void* hip_stream);
int vdd_vit_assemble(const void* emb, const void* cls, const void* pos, void* out, int n, int T, int width, int dtype, void* hip_stream);
int vdd_vit_qkv_split(const void* qkv, void* q, void* k_cache, void* v_cache, int n, int T, int H, int D, int64_t slot_stride, int t_max,
                      int parts /* 3: [q,k,v]; 2: a fused [k,v] projection (cross-attention), q unused */, int dtype, void* hip_stream);

/* out = a + b elementwise (bf16, n % 8 == 0 elements): word + position embeddings of the InstructBLIP Q-Former's text input
 * (lavis Qformer.py:95-99). */
int vdd_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* hip_stream);

/* CLIP ViT LayerNorm (with bias); d % 8 == 0, d <= 4096. */
int vdd_layernorm(const void* x, const void* w, const void* b, void* y, int M, int d, float eps, int dtype, void* hip_stream);

/* y = act(x + bias): act 0 none, 1 quick_gelu (CLIP MLP), 2 gelu-erf (mlp2x_gelu projector, builder.py:33-46). */
int vdd_bias_act(const void* x, const void* bias, void* y, int64_t M, int d, int act, int dtype, void* hip_stream);

/* Most candidates (finite scores left after top-k) a row may keep for the exact top-p arithmetic. */
int vdd_topp_exact_max(void);

/* Largest V whose working row stays in LDS; larger V need scores_out or workspace. */
int vdd_lds_row_capacity(int dtype);

/* Name of the dominant kernel symbol launched for (dtype, V) — for matching rocprof rows. */
const char* vdd_kernel_name(int dtype, int V);

int vdd_abi_version(void);

/* Thread-local description of the last non-zero status returned on this thread. */
const char* vdd_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* VDD_HIP_H */
