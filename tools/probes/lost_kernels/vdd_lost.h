/* vdd_lost.h - C entries of two kernels that were built, tested, measured SLOWER than what ships and taken out of libvdd_hip.so
 * (round 6; DESIGN.md section 6, profiles/r05_slab_probe.jsonl, profiles/r05_persistent_layer_timeline.jsonl).  They live on as a
 * laboratory: `python tools/probes/lost_kernels/lost_ops.py` builds tools/probes/lost_kernels/libvdd_lost.so from the two sources
 * next to this header; nothing in the package, bench.py or the default test suite loads it.  Same conventions as include/vdd_hip.h
 * (device pointers + sizes + hipStream_t as void*, int status, `int dtype` = vdd_dtype in front of the stream). */
#ifndef VDD_LOST_H
#define VDD_LOST_H
#include <stdint.h>

#include "vdd_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The same three projections for 17 - 64 rows in flight (the per-rank batch of an 8-GPU split of LLaVA-Bench / a 4-GPU split of POPE:
 * experiments/eval/MME/run_llava.py:32-40, experiments/eval/llava_sampling.py:100-116), csrc/vdd_skinny_slab.hip: the grid is the CU
 * count, a workgroup owns a range of 16-column tiles of W and ONE slab of K, stages X[:, slab] once into LDS (normalised on the way
 * when `ss` is given), streams W global -> registers -> MFMA, and the last of a tile's KS arrivals adds the fp32 slabs in slab order
 * and runs the epilogue (deterministic; nobody waits for anybody).
 *   ss / nss / ln_w / eps   NULL / 0: X is the input as is.  Else X is the un-normalised residual stream H (row length K) and the
 *                           staging applies bf16(bf16(h * rstd) * ln_w[k]), rstd = rsqrt(sum(ss[row][0 .. nss)) / K + eps); nss % 16 == 0.
 *   swiglu = 0              Y[M, N] = bf16(X W^T) (+ R, rounded again: the new residual stream); ss_out (or NULL) [M, ceil(N / 16)]:
 *                           sums of squares of the 16 columns of each tile of the rows of Y - the `ss` of the next call.
 *   swiglu = 1              W = [Wg; Wu] ([2 N, K]), Y[M, N] = silu(X Wg^T) * (X Wu^T) with the rounding points of vdd_skinny_swiglu;
 *                           R and ss_out must be NULL.
 * K % 32 == 0, ldx % 8 == 0, 1 <= M <= 64.  workspace: vdd_skinny_slab_workspace_bytes(M, N, K, swiglu) bytes (-1: shape not served,
 * the call would return VDD_ERR_UNSUPPORTED), ZEROED once by the caller (tile tickets, left zero by every completed launch); launches
 * that may overlap (different streams) need their own. */
int64_t vdd_skinny_slab_workspace_bytes(int M, int N, int K, int swiglu);
int vdd_skinny_slab(const void* X, const float* ss, int nss, const void* ln_w, float eps, const void* W, const void* R, void* Y,
                    float* ss_out, int M, int N, int K, int64_t ldx, int64_t ldr, int64_t ldy, int swiglu, void* workspace,
                    int64_t workspace_bytes, int dtype, void* hip_stream);

/* ALL decoder layers of one decode step for 1 - 4 rows (one or two questions x their branches in flight: the reference's own
 * operating point, llava_calibrate.py:130,161-177 / llava_llama.py:88-103) as ONE persistent launch (csrc/vdd_layer_persistent.hip):
 * per CU four weight-streaming waves (a four-batch register pipeline that runs ahead across op and layer boundaries) and four
 * gather waves that bring every op's input vector on chip through 8-byte {tag, data} granules; replaces n_layers x
 * (vdd_skinny_gemm_normed + vdd_decode_attention_fused + vdd_skinny_gemm_resid_ss + vdd_skinny_swiglu_normed + vdd_skinny_gemm_resid_ss).
 * layers: DEVICE array of n_layers descriptors (weights [N, K] K-contiguous as everywhere; k_own / v_own / k_pre / v_pre the layer's
 *   KV pools with the strides given: exactly the arguments of vdd_decode_attention_fused).  MHA only (Hkv == H), D == 128, d == 128 H.
 * resid_in [M, d]: the embeddings; resid_out [M, d] + ss_out [M, vdd_decode_layers_ss_cols] (partial sums of squares of a row): what
 *   vdd_skinny_gemm_normed takes for the final norm + lm_head.  pos / cpos / slot / cos_sin / rows: as vdd_decode_attention_fused.
 * workspace: >= vdd_decode_layers_workspace_bytes bytes, 256-byte aligned, ZEROED ONCE by the caller and then left alone (it holds the
 *   launch counter the granule epochs derive from - graph replays included - and, at byte 4, the first give-up code: every spin
 *   of the kernel is bounded; a non-zero word there means a wait timed out and the step's results are garbage).
 *   A workspace of >= that + 256 * n_layers * (#workgroups <= 256) bytes also receives a [workgroup][layer][32] timeline of 100 MHz
 *   ticks behind the exchange buffers (tools/persistent_probe.py --timeline); the engine never allocates it.
 * has_qkv_bias: every descriptor carries bqkv (Qwen).  vdd_decode_layers_max_rows: rows a launch takes for this shape and depth on
 *   this device (0: not served; the LDS holds M (d + F) elements + the descriptor table). */
typedef struct vdd_layer_desc {
    const void* ln1; const void* wqkv; const void* bqkv /* or NULL */; const void* wo; const void* ln2; const void* wgu; const void* wd;
    void* k_own; void* v_own; const void* k_pre; const void* v_pre;
} vdd_layer_desc;
int vdd_decode_layers_max_rows(int d, int H, int F, int D, int n_layers, int dtype);
int64_t vdd_decode_layers_workspace_bytes(int M, int d, int H, int F, int D, int dtype);
int vdd_decode_layers_ss_cols(int d, int H, int F, int D, int dtype);     /* columns of ss_out (one per workgroup and 16-column block) */
int vdd_decode_layers(const vdd_layer_desc* layers, int n_layers, const void* resid_in, void* resid_out, float* ss_out,
                      const int32_t* pos, const int32_t* cpos, const int32_t* slot, const float* cos_sin, const int32_t* rows,
                      int M, int d, int H, int Hkv, int F, int D, float eps, float scale, int64_t slot_stride, int t_max,
                      int64_t prefix_stride, int prefix_tmax, int has_qkv_bias, void* workspace, int64_t workspace_bytes, int dtype,
                      void* hip_stream);

/* Byte offset, inside a vdd_skinny_slab workspace, of the int32 give-up word: 0, or 1 + the index of the first workgroup whose bounded
 * wait for its team's arrival counter timed out (the launch's outputs are garbage then; re-zero the workspace). */
int64_t vdd_skinny_slab_status_offset(void);

#ifdef __cplusplus
}
#endif
#endif
