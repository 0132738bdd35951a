// Logits processors that need a row's token history (run where the reference calls `logits_processor(input_ids, scores)`,
// vcd_utils/vcd_sample.py:197,204): the stop-sequence matcher behind the Qwen StopWordsLogitsProcessor
// (experiments/Qwen_VL/qwen_generation_utils.py:305-385) and HF's RepetitionPenaltyLogitsProcessor.  The history of a row is
// [prompt ids | generated ids]; both kernels read the generated part straight from the decode loop's device-resident id matrix and
// the step counter from device memory, so they sit inside a captured decode step.  Tiny, latency-bound launches (a few ids per row).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "vdd_hip.h"

namespace {

__global__ void stop_words_kernel(const long long* gen, long long ld_gen, long long step, const long long* step_ptr, const long long* tail,
                                  int tail_len, const long long* stop_flat, const int* stop_off, int n_stop, int* force_out, int B) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B) return;
    const long long n_gen = step + (step_ptr ? *step_ptr : 0ll);
    auto id_from_end = [&](int j, bool& ok) -> long long {          // j = 0: the last id of the row so far
        const long long t = n_gen - 1 - j;
        if (t >= 0) { ok = true; return gen[(long long)row * ld_gen + t]; }
        const long long u = (long long)tail_len + t;                 // t < 0: inside the prompt tail
        if (u < 0) { ok = false; return -1; }
        const long long v = tail[(long long)row * tail_len + u];
        ok = v >= 0;                                                 // -1 = left padding: the row is shorter than the sequence
        return v;
    };
    int hit = 0;
    for (int s = 0; s < n_stop && !hit; ++s) {
        const int o = stop_off[s], len = stop_off[s + 1] - o;
        bool match = true;                                           // (_tokens_match: an empty sequence always matches, :363-365)
        for (int j = 0; j < len && match; ++j) {
            bool ok;
            const long long id = id_from_end(j, ok);
            match = ok && id == stop_flat[o + len - 1 - j];
        }
        hit = match ? 1 : 0;
    }
    force_out[row] = hit;
}

constexpr int RP_BLOCK = 256, RP_PER = 32;      // history entries per thread held between the read and the write phase

template <int DT> __device__ __forceinline__ float ld_score(const void* p, long long i) {
    if constexpr (DT == VDD_F32) return ((const float*)p)[i];
    else if constexpr (DT == VDD_F16) return (float)((const _Float16*)p)[i];
    else return __builtin_bit_cast(float, (uint32_t)((const uint16_t*)p)[i] << 16);
}
template <int DT> __device__ __forceinline__ void st_score(void* p, long long i, float f) {
    if constexpr (DT == VDD_F32) ((float*)p)[i] = f;
    else if constexpr (DT == VDD_F16) ((_Float16*)p)[i] = (_Float16)f;
    else ((uint16_t*)p)[i] = __builtin_bit_cast(uint16_t, (__bf16)f);
}

// One workgroup per row.  Phase 1 reads the score of every history id, phase 2 writes the penalised value: every occurrence of an
// id computes the same result from the same un-penalised score (gather -> where -> scatter of the reference), so duplicates are benign.
template <int DT>
__global__ void __launch_bounds__(RP_BLOCK) rep_penalty_kernel(void* scores, long long stride, int V, const long long* prompt, int prompt_len,
                                                               const long long* gen, long long ld_gen, long long step,
                                                               const long long* step_ptr, float penalty, float inv_penalty, int recip) {
    const int row = blockIdx.x, tid = threadIdx.x;
    const long long n_gen = step + (step_ptr ? *step_ptr : 0ll);
    const long long n = (long long)prompt_len + n_gen;
    char* base = (char*)scores;
    const long long ro = (long long)row * stride;
    float val[RP_PER];
    long long id[RP_PER];
#pragma unroll
    for (int k = 0; k < RP_PER; ++k) {
        const long long e = (long long)tid + (long long)k * RP_BLOCK;
        id[k] = -1;
        if (e < n) {
            const long long t = e < prompt_len ? prompt[(long long)row * prompt_len + e] : gen[(long long)row * ld_gen + (e - prompt_len)];
            if (t >= 0 && t < V) { id[k] = t; val[k] = ld_score<DT>(base, ro + t); }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RP_PER; ++k) {
        if (id[k] >= 0) {
            const float x = val[k];
            const float y = x < 0.f ? __fmul_rn(x, penalty) : (recip ? __fmul_rn(x, inv_penalty) : __fdiv_rn(x, penalty));
            st_score<DT>(base, ro + id[k], y);
        }
    }
}

thread_local char g_err2[160] = "";

}  // namespace

extern "C" {

int vdd_stop_words_match(const int64_t* gen, int64_t ld_gen, int64_t step, const int64_t* step_ptr, const int64_t* prompt_tail,
                         int tail_len, const int64_t* stop_flat, const int32_t* stop_off, int n_stop, int32_t* force_out, int B,
                         void* hip_stream) {
    if (B < 0 || n_stop < 0 || tail_len < 0 || !force_out || (n_stop > 0 && (!stop_flat || !stop_off)) || (tail_len > 0 && !prompt_tail) ||
        (!gen && (step > 0 || step_ptr)))
        return VDD_ERR_INVALID_ARG;
    if (B == 0) return VDD_OK;
    hipLaunchKernelGGL(stop_words_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)hip_stream, (const long long*)gen, (long long)ld_gen,
                       (long long)step, (const long long*)step_ptr, (const long long*)prompt_tail, tail_len, (const long long*)stop_flat,
                       stop_off, n_stop, force_out, B);
    return hipGetLastError() == hipSuccess ? VDD_OK : VDD_ERR_LAUNCH;
}

int vdd_repetition_penalty(void* scores, int64_t stride, int dtype, int B, int V, const int64_t* prompt_ids, int prompt_len,
                           const int64_t* gen, int64_t ld_gen, int64_t step, const int64_t* step_ptr, float penalty,
                           uint32_t flags, void* hip_stream) {
    if (!scores || B < 0 || V <= 0 || prompt_len < 0 || (prompt_len > 0 && !prompt_ids) || !(penalty > 0.f) ||
        (!gen && (step > 0 || step_ptr)) || (dtype != VDD_F32 && dtype != VDD_F16 && dtype != VDD_BF16))
        return VDD_ERR_INVALID_ARG;
    if ((int64_t)prompt_len + step + (step_ptr ? ld_gen : 0) > (int64_t)RP_BLOCK * RP_PER) return VDD_ERR_UNSUPPORTED;   // history bound
    if (B == 0) return VDD_OK;
    const int recip = (flags & VDD_TEMP_RECIPROCAL) ? 1 : 0;
    const float inv = 1.0f / penalty;
    hipStream_t st = (hipStream_t)hip_stream;
#define RP_LAUNCH(DT) hipLaunchKernelGGL((rep_penalty_kernel<DT>), dim3(B), dim3(RP_BLOCK), 0, st, scores, (long long)stride, V, \
        (const long long*)prompt_ids, prompt_len, (const long long*)gen, (long long)ld_gen, (long long)step, (const long long*)step_ptr, penalty, inv, recip)
    if (dtype == VDD_F32) RP_LAUNCH(VDD_F32); else if (dtype == VDD_F16) RP_LAUNCH(VDD_F16); else RP_LAUNCH(VDD_BF16);
#undef RP_LAUNCH
    return hipGetLastError() == hipSuccess ? VDD_OK : VDD_ERR_LAUNCH;
}

}  // extern "C"
