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

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two fp32 -> packed bf16, round-to-nearest-even, in ONE instruction (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

struct SeqDesc { int q_row0, Tq, pos0, slot, pslot, plen; };

constexpr float NEG_BIG = -1.0e30f;

// Per 32-key step and wave (16 query rows):
//   S^T = K Q^T   A = K fragment straight from the cache, with the key -> MFMA-row assignment
//                 key(i) = kt + 8 (i / 4) + (i % 4) (+ 4 for the second tile), B = Q fragment (held in registers);
//                 in the C layout lane (query ln, g) then owns the scores of keys kt + 8 g .. + 7: exactly the A fragment
//                 of the PV product, so P never leaves registers (no LDS patch, no barrier for it);
//   softmax       per query column: 8 local scores + 2 cross-group shuffles; the rescale factor of O row 4 g + r is
//                 fetched from the lanes of column 4 g + r;
//   O += P V      B = V fragment = 8 consecutive keys of one dim: one 16-byte LDS read from a TRANSPOSED, double-buffered
//                 V tile [dim][32 keys] (conflict-free with the 80-byte row pitch) that the block stages once per step.
// One barrier per step (the tile hand-off); 24 LDS instructions per lane and step instead of 75.
template <int D, bool CAUSAL>
__global__ void __launch_bounds__(256, 4) flash_attn_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc,
                                                            const uint16_t* __restrict__ vc, const uint16_t* __restrict__ kpre,
                                                            const uint16_t* __restrict__ vpre, const SeqDesc* __restrict__ seqs,
                                                            uint16_t* __restrict__ out, int H, int Hkv, long long slot_stride,
                                                            int t_max, long long pre_stride, int pre_tmax, float scale) {
    constexpr int KS = D / 32;        // k-steps of the QK^T contraction
    constexpr int NT = D / 16;        // 16-wide output tiles over the head dim
    constexpr int VTLD = 40;          // pitch of a transposed V row: 32 keys + 8 pad (80 B: 16-B aligned, conflict-free b128 reads)
    __shared__ __attribute__((aligned(16))) uint16_t vt_lds[2][D * VTLD];

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

    // Q fragments (B operand of S^T): lane (n = query ln, g) holds Q[r0 + ln][32 ks + 8 g .. +7]
    bf16x8_t qf[KS];
    {
        int qr = r0 + ln; if (qr >= sd.Tq) qr = sd.Tq - 1;
        const uint16_t* qp = q + ((size_t)(sd.q_row0 + qr) * H + head) * D + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 32);
    }
    const int qpos = sd.pos0 + r0 + ln;                       // position of this lane's query column
    // own pool: token t at index t - plen (compact slots); prefix pool: token t at index t
    const size_t head_off = (size_t)kvh * t_max * D, pre_off = (size_t)kvh * pre_tmax * D;
    const uint16_t* kbase_own = kc + (size_t)sd.slot * slot_stride + head_off - (size_t)sd.plen * D;
    const uint16_t* vbase_own = vc + (size_t)sd.slot * slot_stride + head_off - (size_t)sd.plen * D;
    const uint16_t* kbase_pre = kpre + (size_t)sd.pslot * pre_stride + pre_off;
    const uint16_t* vbase_pre = vpre + (size_t)sd.pslot * pre_stride + pre_off;

    f32x4_t o[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float mq = NEG_BIG, lq = 0.f;                             // running max / sum of query column ln (replicated over g)

    // staging map: consecutive lanes take consecutive KEYS of one 8-dim slab (2-byte-consecutive transposed LDS writes)
    constexpr int SLABS = D / 8, PER = 32 * SLABS / 256;      // uint4 per thread per tile (D=128: 2, D=64: 1)
    auto stage = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = tid + i * 256;
            const int key = e & 31, dd = (e >> 5) * 8;
            int t = kt + key; if (t >= Tk) t = Tk - 1;
            const uint4 v4 = *reinterpret_cast<const uint4*>((t < sd.plen ? vbase_pre : vbase_own) + (size_t)t * D + dd);
            uint16_t* dst = &vt_lds[buf][dd * VTLD + key];
            dst[0 * VTLD] = (uint16_t)(v4.x & 0xFFFFu); dst[1 * VTLD] = (uint16_t)(v4.x >> 16);
            dst[2 * VTLD] = (uint16_t)(v4.y & 0xFFFFu); dst[3 * VTLD] = (uint16_t)(v4.y >> 16);
            dst[4 * VTLD] = (uint16_t)(v4.z & 0xFFFFu); dst[5 * VTLD] = (uint16_t)(v4.z >> 16);
            dst[6 * VTLD] = (uint16_t)(v4.w & 0xFFFFu); dst[7 * VTLD] = (uint16_t)(v4.w >> 16);
        }
    };
    stage(0, 0);
    int buf = 0;
    // a wave whose 16 query rows lie past the end of the sequence (suffixes are ~25 tokens: half of the block) only helps
    // staging the V tiles; a wave whose rows all precede a key tile (causal) has nothing to add from it either
    const bool wave_has_rows = r0 < sd.Tq;
    for (int kt = 0; kt < kend; kt += 32, buf ^= 1) {
        const bool active = wave_has_rows && !(CAUSAL && kt > sd.pos0 + min(r0 + 15, sd.Tq - 1));
        // ---- S^T = K Q^T for 2 x 16 keys (MFMA row i of tile j is key kt + 8 (i / 4) + (i % 4) + 4 j) ----
        f32x4_t s[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
        if (active) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int t = kt + (ln >> 2) * 8 + (ln & 3) + 4 * j; if (t >= Tk) t = Tk - 1;
            const uint16_t* kp = (t < sd.plen ? kbase_pre : kbase_own) + (size_t)t * D + g * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kp + ks * 32);
                s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[j], 0, 0, 0);
            }
        }
        }
        __syncthreads();                                      // tile `buf` staged by everyone; tile buf^1 no longer read
        if (kt + 32 < kend) stage(kt + 32, buf ^ 1);          // next tile's loads fly under this step's arithmetic
        if (!active) continue;
        // ---- mask + online softmax of query column ln: this lane's scores are keys kt + 8 g + 4 j + r ----
        // masking is only needed on the causal diagonal and on the ragged last tile (wave-uniform test)
        const bool need_mask = (kt + 32 > Tk) || (CAUSAL && kt + 31 > sd.pos0 + r0);
        float mx = NEG_BIG;
        if (need_mask) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt + 8 * g + 4 * j + r;
                    float v = s[j][r] * scale;
                    if (key >= Tk || (CAUSAL && key > qpos)) v = NEG_BIG;
                    s[j][r] = v; mx = fmaxf(mx, v);
                }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[j][r] *= scale; mx = fmaxf(mx, s[j][r]); }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn = fmaxf(mq, mx);
        float pv[8], ps = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[j][r] - mn);            // masked scores: exp(-1e30 - mn) = 0
                pv[4 * j + r] = e; ps += e;
            }
        bf16x8_t pf;
        {
            const uint4 pk = make_uint4(cvt_pk_bf16(pv[0], pv[1]), cvt_pk_bf16(pv[2], pv[3]), cvt_pk_bf16(pv[4], pv[5]), cvt_pk_bf16(pv[6], pv[7]));
            pf = __builtin_bit_cast(bf16x8_t, pk);
        }
        ps += __shfl_xor(ps, 16); ps += __shfl_xor(ps, 32);
        // the running maximum moves on few steps: rescale (l, O) lazily behind a wave-uniform test.  O rows are queries
        // 4 g + r: their factor lives in the lanes of column 4 g + r.
        if (__any(mn > mq)) {
            const float corr = __expf(mq - mn);
            lq *= corr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float cr = __shfl(corr, 4 * g + r);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) o[nt][r] *= cr;
            }
        }
        lq += ps;
        mq = mn;
        // ---- O += P V : B fragment lane (dim = 16 nt + ln, g) = V[kt + 8 g .. + 7][dim] = 16 B of the transposed tile ----
        const uint16_t* vt = &vt_lds[buf][(size_t)ln * VTLD + g * 8];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(vt + nt * 16 * VTLD);
            o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, o[nt], 0, 0, 0);
        }
    }
    // ---- finish: normalise (row sum of query 4 g + r from the lanes of that column), store ----
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float l = __shfl(lq, 4 * g + r);
        const int qr = r0 + g * 4 + r;
        if (qr < sd.Tq) {
            const float inv = 1.f / l;
            uint16_t* op = out + ((size_t)(sd.q_row0 + qr) * H + head) * D + ln;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) op[nt * 16] = (uint16_t)(cvt_pk_bf16(o[nt][r] * inv, 0.f) & 0xFFFFu);
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


// ------------------------------------------------------------------ ViT front-end glue (clip_encoder.py:39-51 -> HF CLIPVisionEmbeddings)
// im2col of the stride-P patch convolution: images [n, 3, S, S] (fp32 / fp16 / bf16) -> patches [n * G * G, Kp] bf16, column
// (c * P + py) * P + px, zero-padded to Kp (the K of the patch-embed GEMM: a multiple of 128).  One block per patch.
template <typename T>
__global__ void __launch_bounds__(256) vit_im2col_kernel(const T* __restrict__ img, uint16_t* __restrict__ out, int S, int P, int G, int Kp) {
    const int patch = blockIdx.x, gi = patch / (G * G), gy = (patch / G) % G, gx = patch % G;
    const int K = 3 * P * P;
    const T* base = img + (size_t)gi * 3 * S * S + (size_t)(gy * P) * S + gx * P;
    for (int col = threadIdx.x; col < Kp; col += 256) {
        float v = 0.f;
        if (col < K) {
            const int c = col / (P * P), r = col - c * P * P, py = r / P, px = r - py * P;
            v = (float)base[(size_t)c * S * S + (size_t)py * S + px];
        }
        out[(size_t)patch * Kp + col] = (uint16_t)f2bf(v);
    }
}

// h[i, t, :] = (t == 0 ? cls : emb[i * (T - 1) + t - 1, :]) + pos[t, :]      (class token + position embedding, bf16 add)
__global__ void __launch_bounds__(256) vit_assemble_kernel(const uint16_t* __restrict__ emb, const uint16_t* __restrict__ cls,
                                                           const uint16_t* __restrict__ pos, uint16_t* __restrict__ out, int T, int w) {
    const int row = blockIdx.x, i = row / T, t = row - i * T;
    const uint16_t* src = t == 0 ? cls : emb + (size_t)(i * (T - 1) + t - 1) * w;
    for (int e = threadIdx.x * 8; e < w; e += 256 * 8) {
        const uint4 a = *reinterpret_cast<const uint4*>(src + e), b = *reinterpret_cast<const uint4*>(pos + (size_t)t * w + e);
        uint4 o;
        o.x = pack(lo(a.x) + lo(b.x), hi(a.x) + hi(b.x)); o.y = pack(lo(a.y) + lo(b.y), hi(a.y) + hi(b.y));
        o.z = pack(lo(a.z) + lo(b.z), hi(a.z) + hi(b.z)); o.w = pack(lo(a.w) + lo(b.w), hi(a.w) + hi(b.w));
        *reinterpret_cast<uint4*>(out + (size_t)row * w + e) = o;
    }
}

// qkv [n * T, 3, H, D] (one fused projection) -> q [n * T, H * D] and the K / V caches [slot = image][H][t][D] the attention reads;
// parts = 2: the input is a fused [k, v] projection only (cross-attention keys / values)
__global__ void __launch_bounds__(256) vit_qkv_split_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ q,
                                                            uint16_t* __restrict__ kc, uint16_t* __restrict__ vc, int T, int H, int D,
                                                            long long slot_stride, int t_max, int parts) {
    const int row = blockIdx.x, i = row / T, t = row - i * T, hd = H * D;
    for (int e = threadIdx.x * 8; e < parts * hd; e += 256 * 8) {
        const uint4 v = *reinterpret_cast<const uint4*>(qkv + (size_t)row * parts * hd + e);
        const int part = e / hd, which = part + (3 - parts), r = e - part * hd, h = r / D, d = r - h * D;
        uint16_t* dst = which == 0 ? q + (size_t)row * hd + r
                                   : (which == 1 ? kc : vc) + (size_t)i * slot_stride + ((size_t)h * t_max + t) * D + d;
        *reinterpret_cast<uint4*>(dst) = v;
    }
}

// out = a + b (bf16, one rounding): word + position embeddings of the Q-Former's text input
__global__ void __launch_bounds__(256) add_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, uint16_t* __restrict__ out, long long n8) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const uint4 x = reinterpret_cast<const uint4*>(a)[i], y = reinterpret_cast<const uint4*>(b)[i];
        uint4 o;
        o.x = pack(lo(x.x) + lo(y.x), hi(x.x) + hi(y.x)); o.y = pack(lo(x.y) + lo(y.y), hi(x.y) + hi(y.y));
        o.z = pack(lo(x.z) + lo(y.z), hi(x.z) + hi(y.z)); o.w = pack(lo(x.w) + lo(y.w), hi(x.w) + hi(y.w));
        reinterpret_cast<uint4*>(out)[i] = o;
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

int vdd_vit_im2col(const void* images, int dtype, void* patches, int n, int S, int P, int Kp, void* stream) {
    if (n <= 0) return VDD_OK;
    if (!images || !patches || P <= 0 || S % P != 0 || Kp < 3 * P * P) return VDD_ERR_INVALID_ARG;
    const int G = S / P;
    const dim3 grid(n * G * G), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VDD_F32) hipLaunchKernelGGL(vit_im2col_kernel<float>, grid, block, 0, st, (const float*)images, (uint16_t*)patches, S, P, G, Kp);
    else if (dtype == VDD_F16) hipLaunchKernelGGL(vit_im2col_kernel<_Float16>, grid, block, 0, st, (const _Float16*)images, (uint16_t*)patches, S, P, G, Kp);
    else if (dtype == VDD_BF16) hipLaunchKernelGGL(vit_im2col_kernel<__bf16>, grid, block, 0, st, (const __bf16*)images, (uint16_t*)patches, S, P, G, Kp);
    else return VDD_ERR_INVALID_ARG;
    return ok();
}

int vdd_vit_assemble(const void* emb, const void* cls, const void* pos, void* out, int n, int T, int width, void* stream) {
    if (n <= 0) return VDD_OK;
    if (!emb || !cls || !pos || !out || width % 8 != 0 || T < 2) return VDD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(n * T), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)emb, (const uint16_t*)cls,
                       (const uint16_t*)pos, (uint16_t*)out, T, width);
    return ok();
}

int vdd_vit_qkv_split(const void* qkv, void* q, void* k_cache, void* v_cache, int n, int T, int H, int D, int64_t slot_stride, int t_max,
                      int parts, void* stream) {
    if (n <= 0) return VDD_OK;
    if (!qkv || (!q && parts == 3) || !k_cache || !v_cache || D % 8 != 0 || T > t_max || (parts != 2 && parts != 3)) return VDD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(vit_qkv_split_kernel, dim3(n * T), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)qkv, (uint16_t*)q,
                       (uint16_t*)k_cache, (uint16_t*)v_cache, T, H, D, (long long)slot_stride, t_max, parts);
    return ok();
}

int vdd_add(const void* a, const void* b, void* out, int64_t n, void* stream) {
    if (n <= 0) return VDD_OK;
    if (!a || !b || !out || n % 8 != 0) return VDD_ERR_INVALID_ARG;
    const long long n8 = n / 8;
    int blocks = (int)((n8 + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a, (const uint16_t*)b, (uint16_t*)out, n8);
    return ok();
}

}  // extern "C"
