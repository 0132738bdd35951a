// Prefill-side kernels for gfx950: flash-style MFMA attention for the LLM prompt and the
// CLIP ViT, LayerNorm, and bias + activation.  They replace the eager HF modules the
// reference runs at step 0 of every branch (experiments/llava/model/llava_arch.py:82-204 ->
// CLIPVisionModel / LlamaModel [ext]; clip_encoder.py:39-51; multimodal_projector/builder.py:33-46).
//
// vdd_flash_attention: one 256-thread block = 64 query rows of one (sequence, head); each of
// its 4 waves owns 16 rows.  Per 32-key tile a wave issues S = Q K^T as 16x16x32 bf16 MFMAs with
// the K fragments loaded straight from the KV cache (lane (key = l&15, g = l>>4) reads 16 B of
// K[key][32 ks + 8 g ..], k-contiguous, no staging), runs the online softmax in the C layout
// (row statistics via 4 intra-16-lane shuffles), re-lays P out as an A fragment through a
// 1.25-KiB per-wave LDS patch, and multiplies by V, which the block stages once per tile in
// LDS ([32][D+8] bf16) because the MFMA B operand wants key-contiguous data while the cache
// is dim-contiguous.  Keys come from [prefix slot | own slot] like the decode kernel.
// Bound: MFMA (dense contraction), but attention is ~2.5 % of prefill FLOPs at T=635, d=4096.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vdd_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ float bf2f(uint32_t b) { return __builtin_bit_cast(float, b << 16); }
__device__ __forceinline__ uint32_t f2bf(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float lo(uint32_t w) { return bf2f(w & 0xFFFFu); }
__device__ __forceinline__ float hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }
__device__ __forceinline__ uint32_t pack(float a, float b) { return f2bf(a) | (f2bf(b) << 16); }

struct SeqDesc { int q_row0, Tq, pos0, slot, pslot, plen; };

constexpr float NEG_BIG = -1.0e30f;

template <int D, bool CAUSAL>
__global__ void __launch_bounds__(256, 4) flash_attn_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc,
                                                         const uint16_t* __restrict__ vc, const uint16_t* __restrict__ kpre,
                                                         const uint16_t* __restrict__ vpre, const SeqDesc* __restrict__ seqs,
                                                         uint16_t* __restrict__ out, int H, int Hkv, long long slot_stride,
                                                         int t_max, long long pre_stride, int pre_tmax, float scale) {
    constexpr int KS = D / 32;        // k-steps of the QK^T contraction
    constexpr int NT = D / 16;        // 16-wide output tiles over the head dim
    constexpr int VLD = D + 8;        // padded V row (elements)
    constexpr int PLD = 40;           // padded P row (elements)
    __shared__ __attribute__((aligned(16))) uint16_t v_lds[32 * VLD];
    __shared__ __attribute__((aligned(16))) uint16_t p_lds[4][16 * PLD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 15, g = lane >> 4;
    const SeqDesc sd = seqs[blockIdx.z];
    const int head = blockIdx.y, kvh = head / (H / Hkv);
    const int qt0 = blockIdx.x * 64;
    if (qt0 >= sd.Tq) return;
    const int r0 = qt0 + wave * 16;                           // this wave's first query row (within the sequence)
    const int Tk = sd.pos0 + sd.Tq;                           // keys that exist
    const int last_row = min(qt0 + 63, sd.Tq - 1);
    const int kend = CAUSAL ? min(Tk, sd.pos0 + last_row + 1) : Tk;   // block-uniform key bound

    // Q fragments (A operand): lane (m = ln, g) holds Q[r0 + m][32 ks + 8 g .. +7]
    bf16x8_t qf[KS];
    {
        int qr = r0 + ln; if (qr >= sd.Tq) qr = sd.Tq - 1;
        const uint16_t* qp = q + ((size_t)(sd.q_row0 + qr) * H + head) * D + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 32);
    }
    // own pool: token t at index t - plen (compact slots); prefix pool: token t at index t
    const size_t head_off = (size_t)kvh * t_max * D, pre_off = (size_t)kvh * pre_tmax * D;
    const uint16_t* kbase_own = kc + (size_t)sd.slot * slot_stride + head_off - (size_t)sd.plen * D;
    const uint16_t* vbase_own = vc + (size_t)sd.slot * slot_stride + head_off - (size_t)sd.plen * D;
    const uint16_t* kbase_pre = kpre + (size_t)sd.pslot * pre_stride + pre_off;
    const uint16_t* vbase_pre = vpre + (size_t)sd.pslot * pre_stride + pre_off;

    f32x4_t o[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float mrow[4] = {NEG_BIG, NEG_BIG, NEG_BIG, NEG_BIG}, lrow[4] = {0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < kend; kt += 32) {
        // ---- stage V[kt .. kt+31][0..D) into LDS (whole block) ----
        __syncthreads();
        for (int i = tid; i < 32 * (D / 8); i += 256) {
            const int key = i / (D / 8), dd = (i % (D / 8)) * 8;
            int t = kt + key; if (t >= Tk) t = Tk - 1;
            const uint16_t* vp = (t < sd.plen ? vbase_pre : vbase_own) + (size_t)t * D + dd;
            *reinterpret_cast<uint4*>(&v_lds[key * VLD + dd]) = *reinterpret_cast<const uint4*>(vp);
        }
        // ---- S = Q K^T for 2 x 16 keys ----
        f32x4_t s[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int t = kt + 16 * j + ln; if (t >= Tk) t = Tk - 1;
            const uint16_t* kp = (t < sd.plen ? kbase_pre : kbase_own) + (size_t)t * D + g * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kp + ks * 32);
                s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], kf, s[j], 0, 0, 0);
            }
        }
        // ---- mask + online softmax (C layout: row = 4 g + r, col = key ln (+16 j)) ----
        float p[2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qpos = sd.pos0 + r0 + g * 4 + r;
            float mx = NEG_BIG;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int key = kt + 16 * j + ln;
                float v = s[j][r] * scale;
                if (key >= Tk || (CAUSAL && key > qpos)) v = NEG_BIG;
                p[j][r] = v;
                mx = fmaxf(mx, v);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2));
            mx = fmaxf(mx, __shfl_xor(mx, 4)); mx = fmaxf(mx, __shfl_xor(mx, 8));
            const float mn = fmaxf(mrow[r], mx);
            const float corr = __expf(mrow[r] - mn);
            float ps = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) { p[j][r] = (p[j][r] <= NEG_BIG * 0.5f) ? 0.f : __expf(p[j][r] - mn); ps += p[j][r]; }
            lrow[r] = lrow[r] * corr + ps;                    // per-lane partial row sum (reduced at the end)
            mrow[r] = mn;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) o[nt][r] *= corr;
        }
        // ---- P (C layout) -> LDS -> A fragment ----
        uint16_t* pw = p_lds[wave];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 2; ++j) pw[(g * 4 + r) * PLD + 16 * j + ln] = (uint16_t)f2bf(p[j][r]);
        __syncthreads();                                      // V tile staged (all waves) and P patch written
        const bf16x8_t pf = *reinterpret_cast<const bf16x8_t*>(&pw[ln * PLD + g * 8]);
        // ---- O += P V : B fragment lane (dim = 16 nt + ln, g) holds V[kt + 8 g + i][dim], i < 8 ----
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bf16x8_t vf;
#pragma unroll
            for (int i = 0; i < 8; ++i) vf[i] = (short)v_lds[(g * 8 + i) * VLD + nt * 16 + ln];
            o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, o[nt], 0, 0, 0);
        }
    }
    // ---- finish: reduce row sums over the 16 lanes of a group, normalise, store ----
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float l = lrow[r];
        l += __shfl_xor(l, 1); l += __shfl_xor(l, 2); l += __shfl_xor(l, 4); l += __shfl_xor(l, 8);
        const int qr = r0 + g * 4 + r;
        if (qr < sd.Tq) {
            const float inv = 1.f / l;
            uint16_t* op = out + ((size_t)(sd.q_row0 + qr) * H + head) * D + ln;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) op[nt * 16] = (uint16_t)f2bf(o[nt][r] * inv);
        }
    }
}

// ------------------------------------------------------------------ LayerNorm (CLIP ViT)
__global__ void __launch_bounds__(256) layernorm_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                        const uint16_t* __restrict__ b, uint16_t* __restrict__ y, int d, float eps) {
    __shared__ float red[2][4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const size_t off = (size_t)row * d;
    constexpr int VPT = 2;                     // d <= 4096
    uint4 h[VPT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int e = (i * 256 + tid) * 8;
        if (e < d) {
            uint4 a = *reinterpret_cast<const uint4*>(x + off + e);
            h[i] = a;
            const float f[8] = {lo(a.x), hi(a.x), lo(a.y), hi(a.y), lo(a.z), hi(a.z), lo(a.w), hi(a.w)};
#pragma unroll
            for (int k = 0; k < 8; ++k) { s1 += f[k]; s2 += f[k] * f[k]; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = s2; }
    __syncthreads();
    const float mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)d;
    const float var = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)d - mean * mean;
    const float rstd = rsqrtf(fmaxf(var, 0.f) + eps);
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int e = (i * 256 + tid) * 8;
        if (e < d) {
            uint4 a = h[i], gw = *reinterpret_cast<const uint4*>(w + e), gb = *reinterpret_cast<const uint4*>(b + e), o;
            auto f = [&](uint32_t hv, uint32_t wv, uint32_t bv) {
                return pack((lo(hv) - mean) * rstd * lo(wv) + lo(bv), (hi(hv) - mean) * rstd * hi(wv) + hi(bv));
            };
            o.x = f(a.x, gw.x, gb.x); o.y = f(a.y, gw.y, gb.y); o.z = f(a.z, gw.z, gb.z); o.w = f(a.w, gw.w, gb.w);
            *reinterpret_cast<uint4*>(y + off + e) = o;
        }
    }
}

// ------------------------------------------------------------------ bias + activation
// act: 0 none, 1 quick_gelu x*sigmoid(1.702x) (CLIP), 2 gelu erf (mlp2x_gelu projector)
__global__ void __launch_bounds__(256) bias_act_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ bias,
                                                       uint16_t* __restrict__ y, long long M, int d, int act) {
    const long long n8 = M * (d / 8);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const int e = (int)(i % (d / 8)) * 8;
        uint4 a = *reinterpret_cast<const uint4*>(x + i * 8);
        uint4 bb = bias ? *reinterpret_cast<const uint4*>(bias + e) : make_uint4(0, 0, 0, 0);
        auto f = [&](float v, float b) {
            v = bf2f(f2bf(v + b));
            if (act == 1) v = v / (1.f + __expf(-1.702f * v));
            else if (act == 2) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
            return v;
        };
        uint4 o;
        o.x = pack(f(lo(a.x), lo(bb.x)), f(hi(a.x), hi(bb.x))); o.y = pack(f(lo(a.y), lo(bb.y)), f(hi(a.y), hi(bb.y)));
        o.z = pack(f(lo(a.z), lo(bb.z)), f(hi(a.z), hi(bb.z))); o.w = pack(f(lo(a.w), lo(bb.w)), f(hi(a.w), hi(bb.w)));
        *reinterpret_cast<uint4*>(y + i * 8) = o;
    }
}

inline int ok() { return hipGetLastError() == hipSuccess ? VDD_OK : VDD_ERR_LAUNCH; }

}  // namespace

extern "C" {

int vdd_flash_attention(const void* q, const void* k_cache, const void* v_cache, const void* k_prefix, const void* v_prefix,
                        const int32_t* seqs, void* out, int n_seq, int max_tq, int H, int Hkv, int D, int64_t slot_stride, int t_max,
                        int64_t prefix_stride, int prefix_tmax, float scale, int causal, void* stream) {
    if (n_seq <= 0 || max_tq <= 0) return VDD_OK;
    if (!q || !k_cache || !v_cache || !k_prefix || !v_prefix || !seqs || !out || (D != 128 && D != 64) || H % Hkv != 0) return VDD_ERR_INVALID_ARG;
    dim3 grid((max_tq + 63) / 64, H, n_seq), block(256);
    hipStream_t st = (hipStream_t)stream;
    auto Q = (const uint16_t*)q; auto K = (const uint16_t*)k_cache; auto V = (const uint16_t*)v_cache; auto O = (uint16_t*)out;
    auto S = (const SeqDesc*)seqs; auto KP = (const uint16_t*)k_prefix; auto VP = (const uint16_t*)v_prefix;
    const long long ps = (long long)prefix_stride;
    if (D == 128) {
        if (causal) hipLaunchKernelGGL((flash_attn_kernel<128, true>), grid, block, 0, st, Q, K, V, KP, VP, S, O, H, Hkv, (long long)slot_stride, t_max, ps, prefix_tmax, scale);
        else hipLaunchKernelGGL((flash_attn_kernel<128, false>), grid, block, 0, st, Q, K, V, KP, VP, S, O, H, Hkv, (long long)slot_stride, t_max, ps, prefix_tmax, scale);
    } else {
        if (causal) hipLaunchKernelGGL((flash_attn_kernel<64, true>), grid, block, 0, st, Q, K, V, KP, VP, S, O, H, Hkv, (long long)slot_stride, t_max, ps, prefix_tmax, scale);
        else hipLaunchKernelGGL((flash_attn_kernel<64, false>), grid, block, 0, st, Q, K, V, KP, VP, S, O, H, Hkv, (long long)slot_stride, t_max, ps, prefix_tmax, scale);
    }
    return ok();
}

int vdd_layernorm(const void* x, const void* w, const void* b, void* y, int M, int d, float eps, void* stream) {
    if (M <= 0) return VDD_OK;
    if (!x || !w || !b || !y || d % 8 != 0 || d > 4096) return VDD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(layernorm_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)w,
                       (const uint16_t*)b, (uint16_t*)y, d, eps);
    return ok();
}

int vdd_bias_act(const void* x, const void* bias, void* y, int64_t M, int d, int act, void* stream) {
    if (M <= 0) return VDD_OK;
    if (!x || !y || d % 8 != 0 || act < 0 || act > 2) return VDD_ERR_INVALID_ARG;
    long long n8 = (long long)M * (d / 8);
    int blocks = (int)((n8 + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bias_act_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)bias,
                       (uint16_t*)y, (long long)M, d, act);
    return ok();
}

}  // extern "C"
