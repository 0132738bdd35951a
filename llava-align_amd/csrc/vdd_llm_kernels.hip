// Decode-step kernels of the branch-batched LLaVA language model for gfx950 (MI355X).
//
// They replace what the reference executes, per branch and per token, through HF's eager
// LlamaModel (experiments/llava/model/language_model/llava_llama.py:88-103 -> transformers
// LlamaDecoderLayer [ext]).  Rows of one step = (question, branch) pairs, so the weights are
// streamed from HBM once per step for ALL branches (the reference streams them once per
// branch: vcd_sample.py:109,163,178).  All tensors bf16, accumulation fp32.
//
//   vdd_rmsnorm            h = residual(+delta); y = h * rsqrt(mean h^2 + eps) * w      HBM-bound
//   vdd_rope_kv_write      RoPE(q,k) at per-row positions; k,v -> per-row KV slot        HBM-bound
//   vdd_silu_mul           silu(gate) * up                                               HBM-bound
//   vdd_embed              token embedding gather                                        HBM-bound
//   vdd_skinny_gemm        Y[M,N] = X[M,K] W[N,K]^T (+R), M <= 64: weight streaming      HBM-bound
//                          (W read once; MFMA 16x16x32 does the k-reduction)
//   vdd_decode_attention   one query per (row, head) over a ragged, prefix-shared KV      HBM-bound

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vdd_elem.h"

namespace {
namespace VDD_ELEM_NS {
using namespace vdd_elem;

typedef __attribute__((ext_vector_type(8))) short frag8_t;    // MFMA A/B fragment: 8 elements (bf16 or fp16 bit patterns)
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_nt;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
// data that crosses the chip once (an activation read by one kernel, a cache line written for a later step): nontemporal, so it
// takes no L2 line from data that IS re-read (the GEMMs' shared operand panels, the next GEMM's X)
__device__ __forceinline__ uint4 ld_stream(const uint16_t* p) { return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p))); }
__device__ __forceinline__ void st_stream(uint16_t* p, uint4 v) { __builtin_nontemporal_store(__builtin_bit_cast(u32x4_nt, v), reinterpret_cast<u32x4_nt*>(p)); }
// a weight fragment of the weight-streaming projections: every byte of W is read by ONE wave, once per launch.  Default cache policy:
// nontemporal weight loads measured SLOWER here (one-question layer chain 90.2 vs 84.8 us)
__device__ __forceinline__ frag8_t ld_w(const uint16_t* p) { return *reinterpret_cast<const frag8_t*>(p); }

// q.k over 8 packed pairs with v_dot2c_f32_{bf16,f16} (fp32 accumulate; the 16-bit x 16-bit products are exact in fp32)
__device__ __forceinline__ float dot8(const uint4& a, const uint4& b) {
    float s = dot2(a.x, b.x, 0.f);
    s = dot2(a.y, b.y, s);
    s = dot2(a.z, b.z, s);
    s = dot2(a.w, b.w, s);
    return s;
}
// One key of the online softmax for a 16-lane group: s is already reduced over the group.  The running maximum only
// moves on a small fraction of the keys, so the rescale of (l, acc) is taken lazily behind a wave-uniform test and
// the common path is 8 converts + 8 FMAs.
#define ATT_ONLINE_STEP(s, vv, m, l, acc)                                                                      \
    do {                                                                                                       \
        if (__any((s) > (m))) {                                                                                \
            const float mn_ = fmaxf((m), (s));                                                                 \
            const float corr_ = __expf((m) - mn_);                                                             \
            (l) *= corr_;                                                                                      \
            _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) (acc)[e_] *= corr_;                                \
            (m) = mn_;                                                                                         \
        }                                                                                                      \
        const float p_ = __expf((s) - (m));                                                                    \
        (l) += p_;                                                                                             \
        (acc)[0] += p_ * lo((vv).x); (acc)[1] += p_ * hi((vv).x); (acc)[2] += p_ * lo((vv).y); (acc)[3] += p_ * hi((vv).y); \
        (acc)[4] += p_ * lo((vv).z); (acc)[5] += p_ * hi((vv).z); (acc)[6] += p_ * lo((vv).w); (acc)[7] += p_ * hi((vv).w); \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------ RMSNorm (+ residual add)
// One 256-thread block per row; the row stays in registers between the two passes.
// HF rounding order: h = bf16(resid + delta); y = bf16( bf16(h * rstd) * w ).
template <int VPT>   // uint4 (8 x bf16) per thread
__global__ void __launch_bounds__(256) rmsnorm_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ delta,
                                                      const float* __restrict__ slabs, int n_slabs, long long slab_stride,
                                                      const uint16_t* __restrict__ w, uint16_t* __restrict__ y,
                                                      uint16_t* __restrict__ resid_out, int d, float eps, int stream) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const size_t off = (size_t)row * d;
    uint4 h[VPT], dl[VPT], wv[VPT];
    float ss = 0.f;
    // every load of the row leaves before anything is consumed (x, the delta, the ln weights: ONE memory round trip), from clamped,
    // never predicated addresses: a load inside `if (e < d)` is waited for inside that branch, which made the VPT chunks of a thread
    // - and the weights behind them - a chain of round trips (13 us for 25 MB at 1,536 rows)
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int e = (i * 256 + tid) * 8, ec = e < d ? e : 0;
        h[i] = stream ? ld_stream(x + off + ec) : *reinterpret_cast<const uint4*>(x + off + ec);
        if (delta != nullptr) dl[i] = stream ? ld_stream(delta + off + ec) : *reinterpret_cast<const uint4*>(delta + off + ec);
        wv[i] = *reinterpret_cast<const uint4*>(w + ec);
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int e = (i * 256 + tid) * 8;
        if (e < d) {
            uint4 a = h[i];
            if (delta != nullptr) {
                const uint4 b = dl[i];
                a.x = pack(lo(a.x) + lo(b.x), hi(a.x) + hi(b.x)); a.y = pack(lo(a.y) + lo(b.y), hi(a.y) + hi(b.y));
                a.z = pack(lo(a.z) + lo(b.z), hi(a.z) + hi(b.z)); a.w = pack(lo(a.w) + lo(b.w), hi(a.w) + hi(b.w));
            }
            if (slabs != nullptr) {      // delta = bf16(sum of the split-K fp32 partial slabs of the producing GEMM)
                float dsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                // the slabs are added in slab order (the bits of round 4), but their loads go out FOUR slabs at a time: one slab per iteration
                // was one memory round trip per slab (S = 8 - 27: most of this launch's time at a few dozen rows)
                int sl = 0;
                for (; sl + 4 <= n_slabs; sl += 4) {
                    float4 q0[4], q1[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        q0[u] = *reinterpret_cast<const float4*>(slabs + (size_t)(sl + u) * slab_stride + off + e);
                        q1[u] = *reinterpret_cast<const float4*>(slabs + (size_t)(sl + u) * slab_stride + off + e + 4);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        dsum[0] += q0[u].x; dsum[1] += q0[u].y; dsum[2] += q0[u].z; dsum[3] += q0[u].w;
                        dsum[4] += q1[u].x; dsum[5] += q1[u].y; dsum[6] += q1[u].z; dsum[7] += q1[u].w;
                    }
                }
                for (; sl < n_slabs; ++sl) {
                    const float4 p0 = *reinterpret_cast<const float4*>(slabs + (size_t)sl * slab_stride + off + e);
                    const float4 p1 = *reinterpret_cast<const float4*>(slabs + (size_t)sl * slab_stride + off + e + 4);
                    dsum[0] += p0.x; dsum[1] += p0.y; dsum[2] += p0.z; dsum[3] += p0.w;
                    dsum[4] += p1.x; dsum[5] += p1.y; dsum[6] += p1.z; dsum[7] += p1.w;
                }
                auto addr = [&](uint32_t hv, float d0, float d1) { return pack(lo(hv) + e2f(f2e(d0)), hi(hv) + e2f(f2e(d1))); };
                a.x = addr(a.x, dsum[0], dsum[1]); a.y = addr(a.y, dsum[2], dsum[3]);
                a.z = addr(a.z, dsum[4], dsum[5]); a.w = addr(a.w, dsum[6], dsum[7]);
            }
            if (resid_out != nullptr) { if (stream) st_stream(resid_out + off + e, a); else *reinterpret_cast<uint4*>(resid_out + off + e) = a; }
            h[i] = a;
            float f;
            f = lo(a.x); ss += f * f; f = hi(a.x); ss += f * f; f = lo(a.y); ss += f * f; f = hi(a.y); ss += f * f;
            f = lo(a.z); ss += f * f; f = hi(a.z); ss += f * f; f = lo(a.w); ss += f * f; f = hi(a.w); ss += f * f;
        }
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int e = (i * 256 + tid) * 8;
        if (e < d) {
            uint4 a = h[i], g = wv[i], o;
            auto nrm = [&](uint32_t hv, uint32_t gv) {
                float n0 = e2f(f2e(lo(hv) * rstd)), n1 = e2f(f2e(hi(hv) * rstd));
                return pack(n0 * lo(gv), n1 * hi(gv));
            };
            o.x = nrm(a.x, g.x); o.y = nrm(a.y, g.y); o.z = nrm(a.z, g.z); o.w = nrm(a.w, g.w);
            *reinterpret_cast<uint4*>(y + off + e) = o;
        }
    }
}

// ------------------------------------------------------------------ RoPE + KV-cache write
// qkv: [M, (Hq + 2 Hkv) * D] (q | k | v).  One block per row, one 16-lane group per head
// chunk.  rotate_half pairing (HF Llama): (i, i + D/2).  cos/sin table: fp32 [max_pos, D/2].
// K/V go to cache[slot][kv_head][pos][D].
__global__ void __launch_bounds__(256) rope_kv_kernel(const uint16_t* __restrict__ qkv, const int* __restrict__ pos,
                                                      const int* __restrict__ cpos, const int* __restrict__ slot,
                                                      const float* __restrict__ cs_table,
                                                      uint16_t* __restrict__ q_out, uint16_t* __restrict__ k_cache,
                                                      uint16_t* __restrict__ v_cache, int Hq, int Hkv, int D,
                                                      long long slot_stride, int t_max, int stream) {
    const int row = blockIdx.x, tid = threadIdx.x;
    const int p = pos[row], s = slot[row], cp = cpos[row];     // rotary position vs index inside the cache slot
    const int half = D / 2, h8 = half / 8;                      // a work item = 8 consecutive pairs (i, i + D/2) of one head
    const uint16_t* src = qkv + (size_t)row * (size_t)((Hq + 2 * Hkv) * D);
    const float* cs = cs_table + (size_t)p * half * 2;
    // All loads of a pass leave before the first result is used: work items tid, tid + 256, ... in batches of IT, read from clamped
    // (always valid) indices, consumed under the `valid` predicate.  As `for (i = tid; i < n; i += 256) { load; rotate; store; }` every
    // trip was a memory round trip of its own (two for q / k, two for v: 17 us per launch at 1,536 rows).
    constexpr int IT = 2;                                     // 64 q / k heads x 8 items = 512 = one batch
    const int n_qk = (Hq + Hkv) * h8;
    for (int i0 = tid; i0 < n_qk; i0 += 256 * IT) {
        uint4 a4[IT], b4[IT];
        float4 c[IT][4];
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = i0 + 256 * k, ic = i < n_qk ? i : 0;
            const int head = ic / h8, pi = (ic % h8) * 8;
            const uint16_t* hsrc = src + (size_t)head * D;           // q heads then k heads are contiguous in qkv
            a4[k] = stream ? ld_stream(hsrc + pi) : *reinterpret_cast<const uint4*>(hsrc + pi);
            b4[k] = stream ? ld_stream(hsrc + pi + half) : *reinterpret_cast<const uint4*>(hsrc + pi + half);
#pragma unroll
            for (int q = 0; q < 4; ++q) c[k][q] = *reinterpret_cast<const float4*>(cs + pi * 2 + 4 * q);
        }
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = i0 + 256 * k;
            if (i < n_qk) {
                const int head = i / h8, pi = (i % h8) * 8;
                const bool is_k = head >= Hq;
                auto rot = [](uint32_t av, uint32_t bv, float cA, float sA, float cB, float sB, uint32_t& r0, uint32_t& r1) {
                    const float a0 = lo(av), a1 = hi(av), b0 = lo(bv), b1 = hi(bv);
                    r0 = pack(a0 * cA - b0 * sA, a1 * cB - b1 * sB);
                    r1 = pack(b0 * cA + a0 * sA, b1 * cB + a1 * sB);
                };
                uint4 r0, r1;
                rot(a4[k].x, b4[k].x, c[k][0].x, c[k][0].y, c[k][0].z, c[k][0].w, r0.x, r1.x);
                rot(a4[k].y, b4[k].y, c[k][1].x, c[k][1].y, c[k][1].z, c[k][1].w, r0.y, r1.y);
                rot(a4[k].z, b4[k].z, c[k][2].x, c[k][2].y, c[k][2].z, c[k][2].w, r0.z, r1.z);
                rot(a4[k].w, b4[k].w, c[k][3].x, c[k][3].y, c[k][3].z, c[k][3].w, r0.w, r1.w);
                uint16_t* dst = is_k ? k_cache + (size_t)s * slot_stride + ((size_t)(head - Hq) * t_max + cp) * D
                                     : q_out + (size_t)row * Hq * D + (size_t)head * D;
                if (is_k && stream) { st_stream(dst + pi, r0); st_stream(dst + pi + half, r1); }
                else { *reinterpret_cast<uint4*>(dst + pi) = r0; *reinterpret_cast<uint4*>(dst + pi + half) = r1; }
            }
        }
    }
    const uint16_t* vsrc = src + (size_t)(Hq + Hkv) * D;
    const int n_v = Hkv * D / 8;
    for (int i0 = tid; i0 < n_v; i0 += 256 * IT) {
        uint4 vv[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = i0 + 256 * k, ic = i < n_v ? i : 0;
            vv[k] = stream ? ld_stream(vsrc + (size_t)ic * 8) : *reinterpret_cast<const uint4*>(vsrc + (size_t)ic * 8);
        }
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = i0 + 256 * k;
            if (i < n_v) {
                const int head = (i * 8) / D, dd = (i * 8) % D;
                uint16_t* vd = v_cache + (size_t)s * slot_stride + ((size_t)head * t_max + cp) * D + dd;
                if (stream) st_stream(vd, vv[k]); else *reinterpret_cast<uint4*>(vd) = vv[k];
            }
        }
    }
}

// ------------------------------------------------------------------ SiLU(gate) * up
__global__ void __launch_bounds__(256) silu_mul_kernel(const uint16_t* __restrict__ gu, uint16_t* __restrict__ out,
                                                       long long M, int F) {
    // grid: (ceil(F / 8 / 256), M): a thread owns 8 consecutive features of one row
    const int f = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (f >= F) return;
    const size_t m = blockIdx.y;
    const uint4 g = *reinterpret_cast<const uint4*>(gu + m * 2 * F + f);
    const uint4 u = *reinterpret_cast<const uint4*>(gu + m * 2 * F + F + f);
    auto act = [](uint32_t gv, uint32_t uv) {
        const float g0 = lo(gv), g1 = hi(gv);
        const float s0 = e2f(f2e(g0 / (1.f + __expf(-g0)))), s1 = e2f(f2e(g1 / (1.f + __expf(-g1))));
        return pack(s0 * lo(uv), s1 * hi(uv));
    };
    uint4 o; o.x = act(g.x, u.x); o.y = act(g.y, u.y); o.z = act(g.z, u.z); o.w = act(g.w, u.w);
    *reinterpret_cast<uint4*>(out + m * F + f) = o;
}

// ------------------------------------------------------------------ embedding gather
// ids outside [0, vocab) read row 0 instead of wild memory: a row whose distribution had no finite score gets token -1
// from the sampling kernel (its status is what the host raises on), and the captured decode step embeds it before any check
__global__ void __launch_bounds__(256) embed_kernel(const long long* __restrict__ ids, const uint16_t* __restrict__ table,
                                                    uint16_t* __restrict__ out, int d, int vocab) {
    const int row = blockIdx.x;
    long long id = ids[row];
    if (id < 0 || id >= vocab) id = 0;
    for (int e = threadIdx.x * 8; e < d; e += 256 * 8)
        *reinterpret_cast<uint4*>(out + (size_t)row * d + e) = *reinterpret_cast<const uint4*>(table + (size_t)id * d + e);
}

__global__ void __launch_bounds__(256) embed_scatter_kernel(const int* __restrict__ ids, const int* __restrict__ rows,
                                                            const uint16_t* __restrict__ table, uint16_t* __restrict__ out, int d, int vocab) {
    long long id = ids[blockIdx.x];
    if (id < 0 || id >= vocab) id = 0;
    const size_t row = (size_t)rows[blockIdx.x];
    for (int e = threadIdx.x * 8; e < d; e += 256 * 8)
        *reinterpret_cast<uint4*>(out + row * d + e) = *reinterpret_cast<const uint4*>(table + (size_t)id * d + e);
}

// ------------------------------------------------------------------ skinny GEMM (weight streaming)
// Y[M,N] = X[M,K] * W[N,K]^T (+ R[M,N]).  One 256-thread block owns 16 output columns; its 4
// waves split K four ways and each streams its quarter of the 16 W rows straight into MFMA B
// fragments (lane (n = l&15, g = l>>4) holds W[n0+n][k + 8g .. +7]: 16 B, k-contiguous), so W
// is read from HBM exactly once and never staged.  X fragments (A operand, same lane map over
// rows) are L2-resident re-reads.  MT = number of 16-row M tiles (M <= 16*MT); the four wave
// partials are summed through LDS.
// NW = 4 or 8 waves per block (8: the N <= 8192 projections, whose d/16 column blocks alone would leave one 4-wave block per CU -
// eight-way K split inside the block, no fp32 slabs for the consumer to sum).
// Small-M decoder-layer fusion of the two RMSNorm launches (a one-question step is a chain of ~7 launches per layer of which the
// two norms are 6.8 us each for 16 KB of data):
//   SSOUT  the d-wide projections (attention output / MLP down): Y = bf16(bf16(X W^T) + R) IS the new residual stream, and the block
//          also leaves the sum of squares of its 16 columns per row in ss_out[row][blockIdx.x] (summed by the consumer in a fixed
//          order: deterministic);
//   NORM   the projections that read a normalised input (qkv, lm_head; gate/up in skinny_swiglu_kernel): X is the un-normalised
//          residual stream H; every lane sums its row's partials (issued behind the first W loads: off the weight stream's
//          critical path), rstd = rsqrt(sum / K + eps), and the X fragments are normalised as they are loaded,
//          bf16(bf16(h * rstd) * ln_w[k]) - the roundings of rmsnorm_kernel.
struct NormIn { const float* ss; int nss; const uint16_t* lnw; float eps; };

typedef __attribute__((ext_vector_type(2))) float f32x2_hw;
// two elements (packed in a dword) -> rnd(rnd(h * rstd) * g) with the hardware's RNE pack conversion (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32)
__device__ __forceinline__ uint32_t norm_pair(uint32_t hv, uint32_t gv, float rstd) {
    const uint32_t nb = cvt_pk(lo(hv) * rstd, hi(hv) * rstd);
    return cvt_pk(lo(nb) * lo(gv), hi(nb) * hi(gv));
}
__device__ __forceinline__ uint4 norm_chunk(uint4 a, uint4 w, float rstd) {
    return make_uint4(norm_pair(a.x, w.x, rstd), norm_pair(a.y, w.y, rstd), norm_pair(a.z, w.z, rstd), norm_pair(a.w, w.w, rstd));
}

// rstd of row `r` from the producer's per-block partial sums ss[r][0 .. nss) (nss % 4 == 0): thread-local, fixed order - 16-byte
// loads issued in batches of eight (one memory round trip per 32 partials), then added in index order
__device__ __forceinline__ float row_rstd(const NormIn& ni, int r, int K) {
    const float4* p = reinterpret_cast<const float4*>(ni.ss + (size_t)r * ni.nss);
    const int n4 = ni.nss >> 2;
    float s = 0.f;
    for (int base = 0; base < n4; base += 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = base + j < n4 ? p[base + j] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    return rsqrtf(s / (float)K + ni.eps);
}

// Persistent form of the weight-streaming projections for a NORMALISED input: a block normalises the (<= 16) rows of H ONCE into LDS
// - bf16(bf16(h * rstd) * ln_w[k]), the roundings of rmsnorm_kernel - behind its first batch of W loads, then walks its column
// blocks (blockIdx.x, + gridDim.x, ...) with the A fragments read from LDS.  (Normalising per column block - in every one of the
// 768 - 2000 blocks of the plain kernels - costs more VALU time than the two RMSNorm launches it replaces: 6.2 vs 3.4 ms per
// one-question step.)  LDS image: row r at byte 2 K r, its 16-byte chunk q at slot q ^ (r & 15): a ds_read_b128 is served in four
// 16-lane groups ({0-3, 12-15, 20-27}, ...: 16 DISTINCT rows at two adjacent chunks) and the XOR puts those on 16 distinct slots of the
// 256-byte bank row (a padded stride of 2 K + 64 left them 2-way conflicted: 16 rows cost 33 us where 2 rows - broadcast - cost 21).
// SWIGLU: a column block is 8 features (gate rows | up rows of W = [Wg; Wu]) with the SiLU * mul epilogue of skinny_swiglu_kernel.
template <int NW, bool SWIGLU, int RMAX>          // RMAX: M rounded up to 2 / 4 / 8 / 16 (the prologue is straight-line code over RMAX rows)
__global__ void __launch_bounds__(NW * 64) skinny_normed_kernel(const uint16_t* __restrict__ Hs, const uint16_t* __restrict__ W,
                                                               uint16_t* __restrict__ Y, int M, int N, int K, long long ldh,
                                                               long long ldy, NormIn ni, int n_cb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xs[];           // [M][2 K] bytes, chunk-swizzled
    __shared__ float part2[2][NW][64][4];                                         // double-buffered: ONE barrier per column block
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 15, g = lane >> 4;
    const int kq = K / NW, kbeg = wave * kq;
    const int rstride = 2 * K;
    constexpr int U = 8;
    // a wave's K slice as nb batches of U 32-deep k-steps; a ragged last batch (K = 5120 over 8 waves: 640 = 2 batches + 128) loads
    // from clamped addresses and multiplies the k-steps beyond the slice by a zero W fragment
    const int nb = (kq + 32 * U - 1) / (32 * U);
    const bool ragged = (kq % (32 * U)) != 0;
    auto wptr = [&](int cb) {
        if constexpr (SWIGLU) { int f = cb * 8 + (ln & 7); if (f >= N) f = N - 1; return W + ((size_t)(ln < 8 ? 0 : N) + f) * K + kbeg + g * 8; }
        else { int nrow = cb * 16 + ln; if (nrow >= N) nrow = N - 1; return W + (size_t)nrow * K + kbeg + g * 8; }
    };
    frag8_t b0[U], b1[U];
    auto ldw = [&](frag8_t (&b)[U], const uint16_t* wp, int i) {
        const int kk = i * 32 * U;
        if (ragged && i == nb - 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) b[u] = ld_w(wp + (kk + 32 * u < kq ? kk + 32 * u : kk));
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) b[u] = ld_w(wp + kk + 32 * u);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    int cb = blockIdx.x;
    const uint16_t* wp = wptr(cb < n_cb ? cb : 0);
    if (nb > 0) ldw(b0, wp, 0);                                                // the weight stream starts before anything else:
    bool b1_ahead = false;                                                     // BOTH register stages of the first column block are
    if (nb > 1) { ldw(b1, wp, 1); b1_ahead = true; }                           // in flight under the normalisation prologue
    // ---- normalise the rows into LDS (once per block).  Sum of squares of a row = its nss per-block partials: thread t takes
    // partials t, t + NT, ... (one load per row for d <= 4096), butterfly over the wave, the NW wave sums in a fixed tree: one
    // memory round trip for all rows together and deterministic.  (A per-thread loop over the partials of a row is a chain of
    // dependent round trips: 10 us per launch.)
    __shared__ float red[NW][16];
    {
        // everything the normalisation reads goes out in ONE memory round trip, as STRAIGHT-LINE code: the per-block partial sums of
        // all rows, the ln weights and the first RB rows of H (further row batches: one round trip each).  Loads are never
        // predicated - a row / chunk / partial that does not exist is read at a clamped, valid address and dropped: a load inside an
        // `if` or a run-time loop ends in its own s_waitcnt, which made the prologue a chain of M round trips (16 rows: 12 us).
        constexpr int NT = NW * 64, CMAX = 16 / NW, RB = RMAX < 16 / CMAX ? RMAX : 16 / CMAX, PMAX = 512 / NT;   // K <= 8192, nss <= 512
        float pv[RMAX][PMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const int rc = r < M ? r : M - 1;
#pragma unroll
            for (int q = 0; q < PMAX; ++q) {
                const int i = tid + q * NT;
                pv[r][q] = ni.ss[(size_t)rc * ni.nss + (i < ni.nss ? i : 0)];
            }
        }
        uint4 gw[CMAX], hv[RB][CMAX];
        auto ld_rows = [&](int r0) {
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int rc = r0 + j < M ? r0 + j : M - 1;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) {
                    const int e = (c * NT + tid) * 8;
                    hv[j][c] = *reinterpret_cast<const uint4*>(Hs + (size_t)rc * ldh + (e < K ? e : 0));
                }
            }
        };
#pragma unroll
        for (int c = 0; c < CMAX; ++c) { const int e = (c * NT + tid) * 8; gw[c] = *reinterpret_cast<const uint4*>(ni.lnw + (e < K ? e : 0)); }
        ld_rows(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < PMAX; ++q) v += (tid + q * NT < ni.nss) ? pv[r][q] : 0.f;
            const float sw = wave_sum(v);
            if (lane == 0) red[wave][r] = sw;
        }
        __syncthreads();
        for (int r0 = 0; r0 < M; r0 += RB) {
            if (r0 > 0) { ld_rows(r0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int r = r0 + j;
                float tot = (red[0][r & 15] + red[1][r & 15]) + (red[2][r & 15] + red[3][r & 15]);
                if constexpr (NW == 8) tot += (red[4][r & 15] + red[5][r & 15]) + (red[6][r & 15] + red[7][r & 15]);
                const float rstd = rsqrtf(tot / (float)K + ni.eps);
#pragma unroll
                for (int c = 0; c < CMAX; ++c) {
                    const int e = (c * NT + tid) * 8;
                    const uint4 nv = norm_chunk(hv[j][c], gw[c], rstd);
                    if (r < M && e < K) *reinterpret_cast<uint4*>(xs + (size_t)r * rstride + ((e * 2) ^ ((r & 15) << 4))) = nv;
                }
            }
        }
    }
    __syncthreads();
    int rr = ln; if (rr >= M) rr = M - 1;
    const unsigned char* xrow = xs + (size_t)rr * rstride;
    const int xk0 = (kbeg + g * 8) * 2, xsw = (rr & 15) << 4;
    auto xfrag = [&](int k) { return *reinterpret_cast<const frag8_t*>(xrow + ((xk0 + k * 2) ^ xsw)); };
    auto mm = [&](const frag8_t (&b)[U], int i, f32x4_t& acc) {
        const int kk = i * 32 * U;
        if (ragged && i == nb - 1) {
            const frag8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool in = kk + 32 * u < kq;
                acc = mfma16(xfrag(in ? kk + 32 * u : kk), in ? b[u] : zero, acc);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc = mfma16(xfrag(kk + 32 * u), b[u], acc);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    int pbuf = 0;
    for (; cb < n_cb; cb += gridDim.x) {
        f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
        int it = 0;
        for (; it + 2 <= nb; it += 2) {
            if (!b1_ahead) ldw(b1, wp, it + 1);
            b1_ahead = false;
            mm(b0, it, acc);
            if (it + 2 < nb) ldw(b0, wp, it + 2);
            mm(b1, it + 1, acc);
        }
        if (it < nb) mm(b0, it, acc);
        const int cb_next = cb + gridDim.x;
        const uint16_t* wp_next = wptr(cb_next < n_cb ? cb_next : cb);
        // BOTH register stages of the next column block go out before the barrier: the weight stream does not drain while the block
        // meets.  The partial sums alternate between two LDS buffers: the waves that do not finish the tile go straight on to the
        // next one (a buffer is rewritten two column blocks later, behind the barrier in between).
        if (cb_next < n_cb && nb > 0) { ldw(b0, wp_next, 0); if (nb > 1) { ldw(b1, wp_next, 1); b1_ahead = true; } }
        float (*part)[64][4] = part2[pbuf];
        pbuf ^= 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][lane][r] = acc[r];
        __syncthreads();
        if constexpr (SWIGLU) {
            if (tid < 128) {           // C map: col = lane & 15, row = (lane >> 4) * 4 + r.  (row = tid / 8, feature = tid % 8)
                const int row = tid >> 3, c = tid & 7;
                if (row < M && cb * 8 + c < N) {
                    const int lg = (row >> 2) * 16 + c, lu = lg + 8, q = row & 3;
                    float gs = 0.f, us = 0.f;
                    gs = (part[0][lg][q] + part[1][lg][q]) + (part[2][lg][q] + part[3][lg][q]);
                    us = (part[0][lu][q] + part[1][lu][q]) + (part[2][lu][q] + part[3][lu][q]);
                    if constexpr (NW == 8) { gs += (part[4][lg][q] + part[5][lg][q]) + (part[6][lg][q] + part[7][lg][q]);
                                             us += (part[4][lu][q] + part[5][lu][q]) + (part[6][lu][q] + part[7][lu][q]); }
                    const float gb = e2f(f2e(gs)), ub = e2f(f2e(us));
                    const float sl = e2f(f2e(gb / (1.f + __expf(-gb))));
                    Y[(size_t)row * ldy + cb * 8 + c] = (uint16_t)f2e(sl * ub);
                }
            }
        } else {
            if (wave == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = g * 4 + r, col = cb * 16 + ln;
                    float sacc = (part[0][lane][r] + part[1][lane][r]) + (part[2][lane][r] + part[3][lane][r]);
                    if constexpr (NW == 8) sacc += (part[4][lane][r] + part[5][lane][r]) + (part[6][lane][r] + part[7][lane][r]);
                    if (row < M && col < N) Y[(size_t)row * ldy + col] = (uint16_t)f2e(sacc);
                }
            }
        }
        wp = wp_next;
    }
}

template <int MT, int NW, bool NORM = false, bool SSOUT = false>
__global__ void __launch_bounds__(NW * 64) skinny_gemm_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W,
                                                             const uint16_t* __restrict__ R, uint16_t* __restrict__ Y,
                                                             float* __restrict__ Yslab, int M, int N, int K, long long ldx,
                                                             long long ldr, long long ldy, NormIn ni, float* __restrict__ ss_out) {
    static_assert(!NORM || MT == 1, "normalise-on-load is the <= 16-row path");
    __shared__ float part[NW][MT][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16;
    const int ln = lane & 15, g = lane >> 4;
    // gridDim.y > 1: split-K across blocks as well; slice s writes the fp32 slab Yslab[s][M][N], summed by the RMSNorm that consumes it
    const int kq = K / (NW * (int)gridDim.y);
    const int kbeg = ((int)blockIdx.y * NW + wave) * kq;
    int nrow = n0 + ln; if (nrow >= N) nrow = N - 1;
    const uint16_t* wp = W + (size_t)nrow * K + kbeg + g * 8;
    constexpr int WS = 1;
    const uint16_t* xp[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) { int r = t * 16 + ln; if (r >= M) r = M - 1; xp[t] = X + (size_t)r * ldx + kbeg + g * 8; }
    f32x4_t acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // Two register stages: the loads of batch i + 1 (8 x 16 B of W per lane) go out before the MFMAs of batch i, so a wave's K
    // slice (1,024 - 2,752 elements = 4 - 10 batches) costs ONE memory round trip plus streaming instead of one per batch (a
    // one-question step is a chain of these launches: see DESIGN.md section 5).  Same MFMAs in the same order.
    constexpr int U = MT > 2 ? 4 : 8;                  // k-steps per register stage (33 - 64 rows: 4, so that TWO stages of 16 - 20 fragments fit)
    const int nit = kq / (32 * U);
    frag8_t b0[U], a0[U][MT], b1[U], a1[U][MT];
    constexpr int UG = NORM ? U : 1;
    frag8_t g0[UG], g1[UG];                             // NORM: the ln weights of a batch's k positions
    const uint16_t* gp = NORM ? ni.lnw + kbeg + g * 8 : nullptr;
    float rstd = 1.f;
    auto ld = [&](frag8_t (&b)[U], frag8_t (&a)[U][MT], frag8_t (&gw)[UG], int kk) {
#pragma unroll
        for (int u = 0; u < U; ++u) b[u] = ld_w(wp + (size_t)(kk + 32 * u) * WS);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t) a[u][t] = *reinterpret_cast<const frag8_t*>(xp[t] + kk + 32 * u);
        if constexpr (NORM) {
#pragma unroll
            for (int u = 0; u < U; ++u) gw[u] = *reinterpret_cast<const frag8_t*>(gp + kk + 32 * u);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mm = [&](const frag8_t (&b)[U], const frag8_t (&a)[U][MT], const frag8_t (&gw)[UG]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                frag8_t av = a[u][t];
                if constexpr (NORM) av = __builtin_bit_cast(frag8_t, norm_chunk(__builtin_bit_cast(uint4, av), __builtin_bit_cast(uint4, gw[u]), rstd));
                acc[t] = mfma16(av, b[u], acc[t]);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    uint16_t rpre[4] = {0, 0, 0, 0};                    // MT == 1: wave 0's residual entries, fetched under the weight stream
    {   // Two register stages at every row count (MT = 2: 2 x (32 + 64) registers; MT = 3 / 4 with U = 4: 2 x (16 + 64)).  A single stage
        // paid one memory round trip per batch - 11 in a row for the down projection's 2,752-deep wave slices: 33.3 -> 26.6 us for its
        // 90 MB at 17 rows, 34.3 -> 33.8 at 32; the attention output at 40 / 48 / 64 rows 19.5 / 20.7 / 23.5 -> 17.1 / 18.3 / 20.9 us.
        // (Eight waves per block with two stages at 17 - 32 rows: 29.6 / 37.6 - slower; tools/skinny_crossover_probe.py.)
        if (nit > 0) ld(b0, a0, g0, 0);
        if (MT == 1 && R != nullptr && wave == 0 && Yslab == nullptr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int row = g * 4 + r, col = n0 + ln; if (row < M && col < N) rpre[r] = R[(size_t)row * ldr + col]; }
        }
        if constexpr (NORM) { int r = ln; if (r >= M) r = M - 1; rstd = row_rstd(ni, r, K); }
        int it = 0;
        for (; it + 2 <= nit; it += 2) {
            ld(b1, a1, g1, (it + 1) * 32 * U);
            mm(b0, a0, g0);
            if (it + 2 < nit) ld(b0, a0, g0, (it + 2) * 32 * U);
            mm(b1, a1, g1);
        }
        if (it < nit) mm(b0, a0, g0);
    }
    // the K slice's remainder (K = 11008 over 8 waves: 1376 = 5 batches + 96): ONE more batch whose loads all go out together, the
    // k-steps beyond the slice with a zero W fragment (+0 to the accumulators; X is read at a clamped, valid address) - a loop of
    // load -> MFMA is a chain of memory round trips at the end of every block
    const int krem = nit * 32 * U;
    if (krem < kq) {
        const frag8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = krem + 32 * u, kc = kk < kq ? kk : krem;
            b0[u] = ld_w(wp + (size_t)kc * WS);
#pragma unroll
            for (int t = 0; t < MT; ++t) a0[u][t] = *reinterpret_cast<const frag8_t*>(xp[t] + kc);
            if constexpr (NORM) g0[u] = *reinterpret_cast<const frag8_t*>(gp + kc);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) if (krem + 32 * u >= kq) b0[u] = zero;
        mm(b0, a0, g0);
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][t][lane][r] = acc[t][r];
    __syncthreads();
    // C/D map of 16x16x32: col = lane & 15, row = (lane >> 4) * 4 + r.  Wave w finishes tile(s) t = w, w+4, ...
    for (int t = wave; t < MT; t += NW) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = t * 16 + g * 4 + r, col = n0 + ln;
            float s = (part[0][t][lane][r] + part[1][t][lane][r]) + (part[2][t][lane][r] + part[3][t][lane][r]);
            if constexpr (NW == 8) s += (part[4][t][lane][r] + part[5][t][lane][r]) + (part[6][t][lane][r] + part[7][t][lane][r]);
            float sq = 0.f;
            if (row < M && col < N) {
                if (Yslab != nullptr) { Yslab[((size_t)blockIdx.y * M + row) * N + col] = s; continue; }
                float o = e2f(f2e(s));
                if (R != nullptr) o = o + e2f(MT == 1 ? rpre[r] : R[(size_t)row * ldr + col]);
                const uint32_t ob = f2e(o);
                Y[(size_t)row * ldy + col] = (uint16_t)ob;
                if constexpr (SSOUT) { const float h = e2f(ob); sq = h * h; }
            }
            if constexpr (SSOUT) {      // the 16 lanes ln = 0..15 of a lane group hold the block's 16 columns of one row
                sq += __shfl_xor(sq, 1); sq += __shfl_xor(sq, 2); sq += __shfl_xor(sq, 4); sq += __shfl_xor(sq, 8);
                if (ln == 0 && row < M) ss_out[(size_t)row * gridDim.x + blockIdx.x] = sq;
            }
        }
    }
}

// ------------------------------------------------------------------ skinny gate/up GEMM with the SwiGLU epilogue
// act[M,F] = silu(X Wg^T) * (X Wu^T), W = [Wg; Wu] ([2F, K]).  The 16 MFMA columns of a block are 8 gate columns
// (B-fragment lanes ln < 8 -> row f0 + ln) and the 8 matching up columns (ln >= 8 -> row F + f0 + ln - 8), so the
// epilogue finds gate and up of one feature in the same LDS tile: no [M, 2F] round trip and no silu_mul launch.
// Rounding as the unfused pair: gate, up -> bf16; silu(gate) -> bf16; product -> bf16.
template <bool NORM>
__global__ void __launch_bounds__(256) skinny_swiglu_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W,
                                                            uint16_t* __restrict__ A, int M, int F, int K, long long ldx, NormIn ni) {
    __shared__ float part[4][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f0 = blockIdx.x * 8;
    const int ln = lane & 15, g = lane >> 4;
    const int kq = K / 4, kbeg = wave * kq;
    int f = f0 + (ln & 7); if (f >= F) f = F - 1;
    const uint16_t* wp = W + ((size_t)(ln < 8 ? 0 : F) + f) * K + kbeg + g * 8;
    constexpr int WS = 1;
    int r = ln; if (r >= M) r = M - 1;
    const uint16_t* xp = X + (size_t)r * ldx + kbeg + g * 8;
    f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int U = 8;             // two register stages, as in skinny_gemm_kernel
    const int nit = kq / (32 * U);
    frag8_t b0[U], a0[U], b1[U], a1[U];
    constexpr int UG = NORM ? U : 1;
    frag8_t g0[UG], g1[UG];
    const uint16_t* gp = NORM ? ni.lnw + kbeg + g * 8 : nullptr;
    float rstd = 1.f;
    auto ld = [&](frag8_t (&b)[U], frag8_t (&a)[U], frag8_t (&gw)[UG], int kk) {
#pragma unroll
        for (int u = 0; u < U; ++u) b[u] = ld_w(wp + (size_t)(kk + 32 * u) * WS);
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = *reinterpret_cast<const frag8_t*>(xp + kk + 32 * u);
        if constexpr (NORM) {
#pragma unroll
            for (int u = 0; u < U; ++u) gw[u] = *reinterpret_cast<const frag8_t*>(gp + kk + 32 * u);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mm = [&](const frag8_t (&b)[U], const frag8_t (&a)[U], const frag8_t (&gw)[UG]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            frag8_t av = a[u];
            if constexpr (NORM) av = __builtin_bit_cast(frag8_t, norm_chunk(__builtin_bit_cast(uint4, av), __builtin_bit_cast(uint4, gw[u]), rstd));
            acc = mfma16(av, b[u], acc);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    if (nit > 0) ld(b0, a0, g0, 0);
    if constexpr (NORM) rstd = row_rstd(ni, r, K);
    int it = 0;
    for (; it + 2 <= nit; it += 2) {
        ld(b1, a1, g1, (it + 1) * 32 * U);
        mm(b0, a0, g0);
        if (it + 2 < nit) ld(b0, a0, g0, (it + 2) * 32 * U);
        mm(b1, a1, g1);
    }
    if (it < nit) mm(b0, a0, g0);
    int k = nit * 32 * U;
    for (; k < kq; k += 32) {
        frag8_t a = *reinterpret_cast<const frag8_t*>(xp + k);
        if constexpr (NORM) a = __builtin_bit_cast(frag8_t, norm_chunk(__builtin_bit_cast(uint4, a), *reinterpret_cast<const uint4*>(gp + k), rstd));
        acc = mfma16(a, *reinterpret_cast<const frag8_t*>(wp + (size_t)k * WS), acc);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) part[wave][lane][rr] = acc[rr];
    __syncthreads();
    // C map: col = lane & 15, row = (lane >> 4) * 4 + rr.  Threads 0..127: (row = tid / 8, feature = tid % 8)
    if (tid < 128) {
        const int row = tid >> 3, c = tid & 7;
        if (row < M && f0 + c < F) {
            const int lg = (row >> 2) * 16 + c, lu = lg + 8, rr = row & 3;
            const float gs = (part[0][lg][rr] + part[1][lg][rr]) + (part[2][lg][rr] + part[3][lg][rr]);
            const float us = (part[0][lu][rr] + part[1][lu][rr]) + (part[2][lu][rr] + part[3][lu][rr]);
            const float gb = e2f(f2e(gs)), ub = e2f(f2e(us));
            const float sl = e2f(f2e(gb / (1.f + __expf(-gb))));
            A[(size_t)row * F + f0 + c] = (uint16_t)f2e(sl * ub);
        }
    }
}

// ------------------------------------------------------------------ weight streaming for 17 - 64 rows
// A few dozen rows in flight (an 8-GPU split of config #3 leaves 34 rows per rank; config #5 on 4 GPUs ~190): too many for the
// 16-column blocks above - every block re-reads ALL of X from L2, 4 X fragments per W fragment at 64 rows, and that traffic (400 MB
// for the qkv projection against 100 MB of weights) is what the launch then moves - and too few for the MFMA GEMM, whose 64 x 256
// tiles cover a fraction of the CUs and pay a serial stream-K fix-up (30 - 43 us per projection whatever M <= 64).
// Here a block owns 32 output columns - two MFMA column tiles sharing every X fragment - its NW waves split K, and the loads of
// batch i + 1 (U k-steps: 2 U W fragments + MT U X fragments per lane) are in flight under the MFMAs of batch i.  W is still read
// from HBM exactly once.  SWIGLU: the two column tiles are 16 gate columns and the 16 matching up columns of W = [Wg; Wu], so the
// epilogue pairs them in registers: act = rnd(rnd(silu(rnd(gate))) * rnd(up)), the rounding points of the unfused pair.
template <int MT, int NW, bool SWIGLU>
__global__ void __launch_bounds__(NW * 64) skinny_wide_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W,
                                                             const uint16_t* __restrict__ R, uint16_t* __restrict__ Y, int M, int N, int K,
                                                             long long ldx, long long ldr, long long ldy) {
    extern __shared__ __attribute__((aligned(16))) float wide_part[];            // [NW][MT][2][64][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * (SWIGLU ? 16 : 32);                               // SWIGLU: N = F features, 16 per block
    const int kq = K / NW, kbeg = wave * kq;
    const uint16_t* wp[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        int row = SWIGLU ? n0 + ln : n0 + 16 * c + ln;
        if (row >= N) row = N - 1;
        wp[c] = W + ((size_t)(SWIGLU && c == 1 ? N : 0) + row) * K + kbeg + g * 8;
    }
    const uint16_t* xp[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) { int r = t * 16 + ln; if (r >= M) r = M - 1; xp[t] = X + (size_t)r * ldx + kbeg + g * 8; }
    f32x4_t acc[MT][2];
#pragma unroll
    for (int t = 0; t < MT; ++t) { acc[t][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    constexpr int U = MT >= 4 ? 2 : 4;           // k-steps per register stage (64 rows: 2 x (2 + 4) fragments x 2 stages + 32 accumulators fill the file)
    const int nit = kq / (32 * U);
    frag8_t b0[U][2], a0[U][MT], b1[U][2], a1[U][MT];
    auto ld = [&](frag8_t (&b)[U][2], frag8_t (&a)[U][MT], int kk) {
#pragma unroll
        for (int u = 0; u < U; ++u) { b[u][0] = ld_w(wp[0] + kk + 32 * u); b[u][1] = ld_w(wp[1] + kk + 32 * u); }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t) a[u][t] = *reinterpret_cast<const frag8_t*>(xp[t] + kk + 32 * u);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mm = [&](const frag8_t (&b)[U][2], const frag8_t (&a)[U][MT]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                acc[t][0] = mfma16(a[u][t], b[u][0], acc[t][0]);
                acc[t][1] = mfma16(a[u][t], b[u][1], acc[t][1]);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    if (nit > 0) ld(b0, a0, 0);
    int it = 0;
    for (; it + 2 <= nit; it += 2) {
        ld(b1, a1, (it + 1) * 32 * U);
        mm(b0, a0);
        if (it + 2 < nit) ld(b0, a0, (it + 2) * 32 * U);
        mm(b1, a1);
    }
    if (it < nit) mm(b0, a0);
    const int krem = nit * 32 * U;
    if (krem < kq) {            // the slice's ragged tail (11008 / 8 = 1376 = 10 batches + 96): one more batch, k-steps beyond it with zero W
        const frag8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = krem + 32 * u, kc = kk < kq ? kk : krem;
            b0[u][0] = ld_w(wp[0] + kc); b0[u][1] = ld_w(wp[1] + kc);
#pragma unroll
            for (int t = 0; t < MT; ++t) a0[u][t] = *reinterpret_cast<const frag8_t*>(xp[t] + kc);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) if (krem + 32 * u >= kq) { b0[u][0] = zero; b0[u][1] = zero; }
        mm(b0, a0);
    }
    auto part = [&](int w, int t, int c) { return wide_part + ((((size_t)w * MT + t) * 2 + c) * 64) * 4; };
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) *reinterpret_cast<f32x4_t*>(part(wave, t, c) + lane * 4) = acc[t][c];
    __syncthreads();
    // C/D map of 16x16x32: col = lane & 15, row = (lane >> 4) * 4 + r.  The NW partials of an entry are added in wave order (fixed).
    auto total = [&](int t, int c) {
        f32x4_t s = *reinterpret_cast<const f32x4_t*>(part(0, t, c) + lane * 4);
#pragma unroll
        for (int w = 1; w < NW; ++w) { const f32x4_t v = *reinterpret_cast<const f32x4_t*>(part(w, t, c) + lane * 4); s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3]; }
        return s;
    };
    if constexpr (SWIGLU) {
        for (int t = wave; t < MT; t += NW) {
            const f32x4_t gs = total(t, 0), us = total(t, 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + g * 4 + r, col = n0 + ln;
                if (row < M && col < N) {
                    const float gb = e2f(f2e(gs[r])), ub = e2f(f2e(us[r]));
                    const float sl = e2f(f2e(gb / (1.f + __expf(-gb))));
                    Y[(size_t)row * ldy + col] = (uint16_t)f2e(sl * ub);
                }
            }
        }
    } else {
        for (int tc = wave; tc < MT * 2; tc += NW) {
            const int t = tc >> 1, c = tc & 1;
            const f32x4_t sv = total(t, c);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + g * 4 + r, col = n0 + 16 * c + ln;
                if (row < M && col < N) {
                    float o = e2f(f2e(sv[r]));
                    if (R != nullptr) o = o + e2f(R[(size_t)row * ldr + col]);
                    Y[(size_t)row * ldy + col] = (uint16_t)f2e(o);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ decode attention
// One wave per (row, head).  Lane group g = lane>>4 takes keys t = 4i + g, lane j = lane&15 the
// dims 8j..8j+7: per iteration a wave reads 4 consecutive K rows (1 KiB contiguous) and the 4
// matching V rows.  Each group runs its own online softmax; the 4 partial states merge at the
// end.  Context of a row = [prefix slot: tokens 0..plen) ++ [own slot: tokens plen..len): rows
// that share a prompt prefix (same image + system prompt) point at ONE physical copy.
struct AttnRow { int slot, len, pslot, plen; };

// Split-KV form: blockIdx.z = key chunk of CH keys; every (row, head, chunk) wave leaves an
// un-normalised partial (acc[128], m, l) in `ws`, and decode_attn_combine merges the chunks.
// The split turns one ~650-iteration latency chain per (row, head) into <= CH/4 iterations with
// 8 x 16-B loads in flight per lane, and multiplies the number of independent waves by T/CH.
constexpr int ATT_CH = 64;
// a partial of the split-KV attention: D un-normalised output floats, then (running max, sum); records are 528 B so that every
// store of one is 16-byte aligned (520-byte records cost the prefix pass 10 us of partial-line write traffic: tools/probes/hbm_stream_probe.hip)
constexpr int ATT_PS = 128 + 4;

template <int D>   // head dim 128
__global__ void __launch_bounds__(256) decode_attn_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc,
                                                          const uint16_t* __restrict__ vc, const uint16_t* __restrict__ kpre,
                                                          const uint16_t* __restrict__ vpre, const AttnRow* __restrict__ rows,
                                                          float* __restrict__ ws, int H, int Hkv, long long slot_stride,
                                                          int t_max, long long pre_stride, int pre_tmax, float scale, int nchunk,
                                                          int own_only, int chunk_base) {
    static_assert(D == 128, "lane map assumes 16 lanes x 8 dims");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int head = blockIdx.x * 4 + wave, row = blockIdx.y, chunk = blockIdx.z;
    if (head >= H) return;
    const int g = lane >> 4, j = lane & 15;
    const AttnRow ar = rows[row];
    float* wsp = ws + (((size_t)row * H + head) * nchunk + chunk_base + chunk) * ATT_PS;
    // own_only: the shared prefix [0, plen) is handled by the grouped MFMA kernel; this one starts at plen
    const int k0 = (own_only ? ar.plen : 0) + chunk * ATT_CH, k1 = min(ar.len, k0 + ATT_CH);
    if (k0 >= ar.len) {                                   // empty chunk: neutral partial
        if (lane == 0) { wsp[D] = -INFINITY; wsp[D + 1] = 0.f; }
        return;
    }
    const int kvh = head / (H / Hkv);
    const uint4 qv = *reinterpret_cast<const uint4*>(q + ((size_t)row * H + head) * D + j * 8);
    // own pool stores token t at index t - plen (compact slots); the prefix pool at index t
    const size_t hoff = (size_t)kvh * t_max * D + j * 8, poff = (size_t)kvh * pre_tmax * D + j * 8;
    const uint16_t* k_own = kc + (size_t)ar.slot * slot_stride + hoff - (size_t)ar.plen * D;
    const uint16_t* v_own = vc + (size_t)ar.slot * slot_stride + hoff - (size_t)ar.plen * D;
    const uint16_t* k_pre = kpre + (size_t)ar.pslot * pre_stride + poff;
    const uint16_t* v_pre = vpre + (size_t)ar.pslot * pre_stride + poff;
    float m = -INFINITY, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;
    for (int t0 = k0 + g; t0 < k1; t0 += 4 * U) {
        uint4 kv[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + 4 * u;
            const int tt = t < k1 ? t : k1 - 1;
            const bool pre = tt < ar.plen;
            kv[u] = ld_stream((pre ? k_pre : k_own) + (size_t)tt * D);          // each K / V row is read by one wave of the launch
            vv[u] = ld_stream((pre ? v_pre : v_own) + (size_t)tt * D);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float s = dot8(qv, kv[u]);
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
            s *= scale;
            if (t0 + 4 * u < k1) ATT_ONLINE_STEP(s, vv[u], m, l, acc);
        }
    }
    // merge the 4 lane groups (xor 16, 32)
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const float mo = __shfl_xor(m, o), lo_ = __shfl_xor(l, o);
        const float mn = fmaxf(m, mo);
        const float c0 = (m == -INFINITY) ? 0.f : __expf(m - mn), c1 = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
        l = l * c0 + lo_ * c1;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float ao = __shfl_xor(acc[e], o); acc[e] = acc[e] * c0 + ao * c1; }
        m = mn;
    }
    if (g == 0) {
        *reinterpret_cast<float4*>(wsp + j * 8) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(wsp + j * 8 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        if (j == 0) { wsp[D] = m; wsp[D + 1] = l; }
    }
}

// Grouped mode, short own ranges (a question's own prompt tokens + what it generated so far: <= 256 keys): ONE wave per
// (row, head) walks the whole own range [plen, len), then folds in the row's prefix partials left by the MFMA prefix pass
// and writes the normalised bf16 output - no own partials, no separate combine launch.
template <int D>
__global__ void __launch_bounds__(256) decode_attn_own_merge_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc,
                                                                    const uint16_t* __restrict__ vc, const AttnRow* __restrict__ rows,
                                                                    const float* __restrict__ ws, uint16_t* __restrict__ out, int H, int Hkv,
                                                                    long long slot_stride, int t_max, float scale, int nchunk, int npre,
                                                                    int pre_keys) {
    static_assert(D == 128, "lane map assumes 16 lanes x 8 dims");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int head = blockIdx.x * 4 + wave, row = blockIdx.y;
    if (head >= H) return;
    const int g = lane >> 4, j = lane & 15;
    const AttnRow ar = rows[row];
    const int kvh = head / (H / Hkv);
    const uint4 qv = *reinterpret_cast<const uint4*>(q + ((size_t)row * H + head) * D + j * 8);
    const size_t hoff = (size_t)kvh * t_max * D + j * 8;
    const uint16_t* k_own = kc + (size_t)ar.slot * slot_stride + hoff;       // compact slot: own token i at index i
    const uint16_t* v_own = vc + (size_t)ar.slot * slot_stride + hoff;
    const int n_own = ar.len - ar.plen;
    float m = -INFINITY, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int U = 8;
    for (int t0 = g; t0 < n_own; t0 += 4 * U) {
        uint4 kv[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + 4 * u;
            const int tt = t < n_own ? t : n_own - 1;
            kv[u] = ld_stream(k_own + (size_t)tt * D);
            vv[u] = ld_stream(v_own + (size_t)tt * D);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float s = dot8(qv, kv[u]);
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
            s *= scale;
            if (t0 + 4 * u < n_own) ATT_ONLINE_STEP(s, vv[u], m, l, acc);
        }
    }
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const float mo = __shfl_xor(m, o), lo_ = __shfl_xor(l, o);
        const float mn = fmaxf(m, mo);
        const float c0 = (m == -INFINITY) ? 0.f : __expf(m - mn), c1 = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
        l = l * c0 + lo_ * c1;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float ao = __shfl_xor(acc[e], o); acc[e] = acc[e] * c0 + ao * c1; }
        m = mn;
    }
    // fold in the prefix partials (every lane of a 16-lane group reads its 8 dims; group 0 stores)
    const int used_pre = npre > 0 ? (ar.plen + pre_keys - 1) / pre_keys : 0;
    const float* base = ws + ((size_t)row * H + head) * nchunk * ATT_PS;
    for (int c = 0; c < used_pre; ++c) {
        const float* pp = base + (size_t)c * ATT_PS;
        const float mo = pp[D], lo_ = pp[D + 1];
        const float mn = fmaxf(m, mo);
        const float c0 = (m == -INFINITY) ? 0.f : __expf(m - mn), c1 = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
        const float4 a0 = *reinterpret_cast<const float4*>(pp + j * 8), a1 = *reinterpret_cast<const float4*>(pp + j * 8 + 4);
        l = l * c0 + lo_ * c1;
        acc[0] = acc[0] * c0 + a0.x * c1; acc[1] = acc[1] * c0 + a0.y * c1; acc[2] = acc[2] * c0 + a0.z * c1; acc[3] = acc[3] * c0 + a0.w * c1;
        acc[4] = acc[4] * c0 + a1.x * c1; acc[5] = acc[5] * c0 + a1.y * c1; acc[6] = acc[6] * c0 + a1.z * c1; acc[7] = acc[7] * c0 + a1.w * c1;
        m = mn;
    }
    if (g == 0) {
        const float inv = 1.f / l;
        uint4 o4;
        o4.x = pack(acc[0] * inv, acc[1] * inv); o4.y = pack(acc[2] * inv, acc[3] * inv);
        o4.z = pack(acc[4] * inv, acc[5] * inv); o4.w = pack(acc[6] * inv, acc[7] * inv);
        *reinterpret_cast<uint4*>(out + ((size_t)row * H + head) * D + j * 8) = o4;
    }
}

// one wave per (row, head): lane owns dims 2*lane, 2*lane+1.  Chunk space: [0, npre) prefix chunks (grouped
// kernel; only when grouped), then own/whole-context chunks.
template <int D>
__global__ void __launch_bounds__(256) decode_attn_combine_kernel(const float* __restrict__ ws, const AttnRow* __restrict__ rows,
                                                                  uint16_t* __restrict__ out, int H, int nchunk, int npre,
                                                                  int pre_keys) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int head = blockIdx.x * 4 + wave, row = blockIdx.y;
    if (head >= H) return;
    const AttnRow ar = rows[row];
    const int used_pre = npre > 0 ? (ar.plen + pre_keys - 1) / pre_keys : 0;
    const int own_len = npre > 0 ? ar.len - ar.plen : ar.len;
    const int used_own = min(nchunk - npre, (own_len + ATT_CH - 1) / ATT_CH);
    const float* base = ws + ((size_t)row * H + head) * nchunk * ATT_PS;
    float M = -INFINITY;
    for (int c = 0; c < used_pre; ++c) M = fmaxf(M, base[c * ATT_PS + D]);
    for (int c = 0; c < used_own; ++c) M = fmaxf(M, base[(npre + c) * ATT_PS + D]);
    float L = 0.f, a0 = 0.f, a1 = 0.f;
    auto add = [&](int c) {
        const float* p = base + c * ATT_PS;
        const float w = __expf(p[D] - M);
        L += w * p[D + 1];
        const float2 v = *reinterpret_cast<const float2*>(p + 2 * lane);
        a0 += w * v.x; a1 += w * v.y;
    };
    for (int c = 0; c < used_pre; ++c) add(c);
    for (int c = 0; c < used_own; ++c) add(npre + c);
    const float inv = 1.f / L;
    reinterpret_cast<uint32_t*>(out + ((size_t)row * H + head) * D)[lane] = pack(a0 * inv, a1 * inv);
}

// ------------------------------------------------------------------ small-M fused decode attention
// For a handful of rows (the reference's own operating point: ONE question = 2-3 branch rows) the step is a chain of
// latency-bound launches, so RoPE, the KV-cache write, the whole-context attention and the merge run as ONE kernel:
// block = (head, row), NW waves; key t belongs to 16-lane group (t mod 4 NW); the new token's K/V never round-trip
// through the cache (they are rotated in registers, written once per KV head, and attended from registers).
// KSPLIT > 1 (one or two questions in flight: H x M blocks would leave most CUs idle and each block three dependent fetch rounds
// deep): blockIdx.z takes the z-th slice of the old keys, leaves an un-normalised partial (128 sums, max, weight) in `ws` with
// write-through stores and draws a ticket from the (row, head) counter; the block that draws the last ticket merges the KSPLIT
// partials in slice order (deterministic) and writes the output row.  Same hand-off as the GEMM's stream-K fix-up: write-through
// (sc1) stores + s_waitcnt before a relaxed agent-scope atomic, agent-scope (sc1) loads behind it, no fences; the counter is left
// at zero for the next launch.
constexpr int ATT_FS = 128 + 4;          // floats per partial of the split form: 128 sums, max, weight, 2 pad (16-byte rows)
template <int D, int NW, int KSPLIT>
__global__ void __launch_bounds__(NW * 64) decode_attn_fused_kernel(const uint16_t* __restrict__ qkv, const int* __restrict__ pos,
                                                                    const int* __restrict__ cpos, const int* __restrict__ slot,
                                                                    const float* __restrict__ cs_table, uint16_t* __restrict__ kc,
                                                                    uint16_t* __restrict__ vc, const uint16_t* __restrict__ kpre,
                                                                    const uint16_t* __restrict__ vpre, const AttnRow* __restrict__ rows,
                                                                    uint16_t* __restrict__ out, int H, int Hkv, long long slot_stride,
                                                                    int t_max, long long pre_stride, int pre_tmax, float scale,
                                                                    float* __restrict__ ws, int* __restrict__ tickets) {
    static_assert(D == 128, "lane map assumes 16 lanes x 8 dims");
    __shared__ float part[NW][D + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    const int head = blockIdx.x, row = blockIdx.y, zs = KSPLIT > 1 ? (int)blockIdx.z : 0;
    const AttnRow ar = rows[row];
    const int p = pos[row], cp = cpos[row];
    const int kvh = head / (H / Hkv);
    const uint16_t* src = qkv + (size_t)row * (size_t)((H + 2 * Hkv) * D);
    // context pointers and the FIRST batch of K/V loads go out before the RoPE arithmetic (they do not depend on q)
    const size_t hoff = (size_t)kvh * t_max * D + j * 8, poff = (size_t)kvh * pre_tmax * D + j * 8;
    const uint16_t* k_own = kc + (size_t)ar.slot * slot_stride + hoff - (size_t)ar.plen * D;
    const uint16_t* v_own = vc + (size_t)ar.slot * slot_stride + hoff - (size_t)ar.plen * D;
    const uint16_t* k_pre = kpre + (size_t)ar.pslot * pre_stride + poff;
    const uint16_t* v_pre = vpre + (size_t)ar.pslot * pre_stride + poff;
    constexpr int U = 4, STEP = NW * 4;
    const int n_all = ar.len - 1;                       // keys already in the cache; the new token is attended from registers
    // this block's slice [k_lo, n_old) of them (whole STEP-key rounds per slice); the last slice also takes the new token
    const int per = KSPLIT > 1 ? ((n_all + KSPLIT * STEP - 1) / (KSPLIT * STEP)) * STEP : 0;
    const int k_lo = KSPLIT > 1 ? min(zs * per, n_all) : 0;
    const int n_old = KSPLIT > 1 ? min(k_lo + per, n_all) : n_all;
    const bool last_slice = zs == KSPLIT - 1;
    const int kl = k_lo + wave * 4 + g;
    uint4 kn_[U], vn_[U];
    auto fetch = [&](int t0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + STEP * u;
            const int tt = t < n_old ? t : n_old - 1;
            const bool pre = tt < ar.plen;
            kn_[u] = *reinterpret_cast<const uint4*>((pre ? k_pre : k_own) + (size_t)tt * D);
            vn_[u] = *reinterpret_cast<const uint4*>((pre ? v_pre : v_own) + (size_t)tt * D);
        }
    };
    if (kl < n_old) fetch(kl);
    // rotate_half RoPE: lane j < 8 holds dims 8j.. (first half), lane j + 8 the partner dims; same rounding as rope_kv_kernel
    const float* cs = cs_table + ((size_t)p * (D / 2) + (j & 7) * 8) * 2;
    const float4 c0 = *reinterpret_cast<const float4*>(cs), c1 = *reinterpret_cast<const float4*>(cs + 4),
                 c2 = *reinterpret_cast<const float4*>(cs + 8), c3 = *reinterpret_cast<const float4*>(cs + 12);
    const float sign = j < 8 ? -1.f : 1.f;
    auto rope = [&](const uint16_t* hsrc) {
        const uint4 own = *reinterpret_cast<const uint4*>(hsrc + j * 8);
        uint4 oth;
        oth.x = __shfl_xor(own.x, 8); oth.y = __shfl_xor(own.y, 8); oth.z = __shfl_xor(own.z, 8); oth.w = __shfl_xor(own.w, 8);
        auto r2 = [&](uint32_t a, uint32_t b, float cA, float sA, float cB, float sB) {
            return pack(lo(a) * cA + (sign * lo(b)) * sA, hi(a) * cB + (sign * hi(b)) * sB);
        };
        uint4 r;
        r.x = r2(own.x, oth.x, c0.x, c0.y, c0.z, c0.w); r.y = r2(own.y, oth.y, c1.x, c1.y, c1.z, c1.w);
        r.z = r2(own.z, oth.z, c2.x, c2.y, c2.z, c2.w); r.w = r2(own.w, oth.w, c3.x, c3.y, c3.z, c3.w);
        return r;
    };
    const uint4 qv = rope(src + (size_t)head * D);
    const uint4 kn = rope(src + (size_t)(H + kvh) * D);
    const uint4 vn = *reinterpret_cast<const uint4*>(src + (size_t)(H + Hkv + kvh) * D + j * 8);
    if (last_slice && wave == 0 && g == 0 && head % (H / Hkv) == 0) {      // one writer per KV head
        const size_t o = (size_t)slot[row] * slot_stride + ((size_t)kvh * t_max + cp) * D + j * 8;
        *reinterpret_cast<uint4*>(kc + o) = kn;
        *reinterpret_cast<uint4*>(vc + o) = vn;
    }
    float m = -INFINITY, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t0 = kl; t0 < n_old; t0 += STEP * U) {
        uint4 kv[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { kv[u] = kn_[u]; vv[u] = vn_[u]; }
        if (t0 + STEP * U < n_old) fetch(t0 + STEP * U);      // next batch in flight under this batch's arithmetic
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float sc = dot8(qv, kv[u]);
            sc += __shfl_xor(sc, 1); sc += __shfl_xor(sc, 2); sc += __shfl_xor(sc, 4); sc += __shfl_xor(sc, 8);
            sc *= scale;
            if (t0 + STEP * u < n_old) ATT_ONLINE_STEP(sc, vv[u], m, l, acc);
        }
    }
    {   // the new token, from registers (last wave's last group is the least loaded)
        float sc = dot8(qv, kn);
        sc += __shfl_xor(sc, 1); sc += __shfl_xor(sc, 2); sc += __shfl_xor(sc, 4); sc += __shfl_xor(sc, 8);
        sc *= scale;
        if (last_slice && wave == NW - 1 && g == 3) ATT_ONLINE_STEP(sc, vn, m, l, acc);
    }
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const float mo = __shfl_xor(m, o), lo_ = __shfl_xor(l, o);
        const float mn = fmaxf(m, mo);
        const float e0 = (m == -INFINITY) ? 0.f : __expf(m - mn), e1 = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
        l = l * e0 + lo_ * e1;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float ao = __shfl_xor(acc[e], o); acc[e] = acc[e] * e0 + ao * e1; }
        m = mn;
    }
    if (g == 0) {
        *reinterpret_cast<float4*>(&part[wave][j * 8]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(&part[wave][j * 8 + 4]) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        if (j == 0) { part[wave][D] = m; part[wave][D + 1] = l; }
    }
    __syncthreads();
    if (wave == 0) {
        float Mx = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) Mx = fmaxf(Mx, part[w][D]);
        float L = 0.f, a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float wgt = part[w][D] == -INFINITY ? 0.f : __expf(part[w][D] - Mx);       // empty waves (and a whole empty slice)
            L += wgt * part[w][D + 1];
            a0 += wgt * part[w][2 * lane]; a1 += wgt * part[w][2 * lane + 1];
        }
        if constexpr (KSPLIT == 1) {
            const float inv = 1.f / L;
            reinterpret_cast<uint32_t*>(out + ((size_t)row * H + head) * D)[lane] = pack(a0 * inv, a1 * inv);
        } else {
            const size_t rh = (size_t)row * H + head;
            float* mine = ws + (rh * KSPLIT + zs) * ATT_FS;
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, ATT_FS * 4, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, f32x2_hw{a0, a1}), rp, lane * 8, 0, 16);
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, f32x2_hw{Mx, L}), rp, D * 4, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int ticket = 0;
            if (lane == 0) ticket = __hip_atomic_fetch_add(tickets + rh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ticket = __builtin_amdgcn_readfirstlane(ticket);
            if (ticket == KSPLIT - 1) {                  // every other slice has published: merge in slice order
                if (lane == 0) __hip_atomic_store(tickets + rh, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(ws + rh * KSPLIT * ATT_FS), 0, KSPLIT * ATT_FS * 4, 0x00020000);
                f32x2_hw av[KSPLIT], ml[KSPLIT];
#pragma unroll
                for (int k = 0; k < KSPLIT; ++k) {
                    av[k] = __builtin_bit_cast(f32x2_hw, __builtin_amdgcn_raw_buffer_load_b64(rq, (k * ATT_FS) * 4 + lane * 8, 0, 16));
                    ml[k] = __builtin_bit_cast(f32x2_hw, __builtin_amdgcn_raw_buffer_load_b64(rq, (k * ATT_FS + D) * 4, 0, 16));
                }
                float Mg = -INFINITY;
#pragma unroll
                for (int k = 0; k < KSPLIT; ++k) Mg = fmaxf(Mg, ml[k][0]);
                float Lg = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
                for (int k = 0; k < KSPLIT; ++k) {
                    const float wgt = ml[k][0] == -INFINITY ? 0.f : __expf(ml[k][0] - Mg);     // an empty slice weighs nothing
                    Lg += wgt * ml[k][1]; b0 += wgt * av[k][0]; b1 += wgt * av[k][1];
                }
                const float inv = 1.f / Lg;
                reinterpret_cast<uint32_t*>(out + rh * D)[lane] = pack(b0 * inv, b1 * inv);
            }
        }
    }
}

// ------------------------------------------------------------------ prefix-grouped decode attention
// Rows that share a prompt prefix (the 6 POPE questions of one image; ALL image-free branch rows) form a group: each
// prefix K/V byte leaves HBM/L2 once per group instead of once per row.
struct GroupDesc { int row_off, n_rows, pslot, plen; };

// ------------------------------------------------------------------ prefix pass on the matrix cores
// K and V of a shared prefix are static during decoding, so a second copy is kept in FRAGMENT-MAJOR order, built once per
// generate by prefix_fragments_kernel: per (slot, kv head) and 64-key chunk one 32-KiB block = 16 K fragments + 16 V^T
// fragments, each the 1-KiB lane-linear image of one MFMA operand (lane l at byte 16 l).  Every fragment load of the pass is
// then one contiguous KiB (8 whole 128-byte lines) instead of 16 half-lines (K rows) or 4 x 256 B (a key-blocked V^T), both
// contractions take their operands straight from HBM, and the pass needs no LDS and no barrier:
//   S^T = K Q^T   A = K fragment (t, ks): lane (i, g) = 16 B of K[key_t(i)][32 ks + 8 g ..], B = Q fragment (lane (query, g))
//                 C layout: column = query, row i = 4 g + r.  The key assigned to MFMA row i is chosen as
//                 key_a(i) = k0 + 8 (i/4) + i%4 for tile a and key_b(i) = key_a(i) + 4 for tile b, so that the 8
//                 probabilities a lane holds after the softmax are the 8 CONSECUTIVE keys k0 + 8 g .. + 7:
//   O = P V       A = P fragment = those 8 registers as they are (no re-layout), B = V^T fragment (kk, nt): lane (j, g) =
//                 V[k0 + 32 kk + 8 g .. + 7][8 j + nt]: column j of output tile nt is head dim 8 j + nt, so a lane ends up with
//                 8 CONSECUTIVE dims of its 4 query rows and a partial row leaves as 16 lanes x 32 B = one contiguous 512 B.
// One wave = one (16-row slice of a group, head, item of 64-key chunks): 16 + 16 MFMAs per chunk, online softmax across the
// chunks, un-normalised partial to the workspace (merged by the own pass / decode_attn_combine_kernel).
constexpr int FRAG_CHUNK_ELEMS = 2 * ATT_CH * 128;      // bf16 elements of one chunk's block (K then V^T)
__global__ void __launch_bounds__(256) prefix_fragments_kernel(const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                                                               uint16_t* __restrict__ frag, int t_max, const int* __restrict__ plen_of_slot) {
    // grid: (t_max / 64, n_kv_heads, n_slots); block 256: one 64-key chunk, D = 128
    constexpr int D = 128;
    const int chunk = blockIdx.x, head = blockIdx.y, slot = blockIdx.z, plen = plen_of_slot[slot], k0 = chunk * ATT_CH;
    if (k0 >= plen) return;
    const size_t base = ((size_t)slot * gridDim.y + head) * (size_t)t_max * D;
    uint16_t* out = frag + 2 * base + (size_t)chunk * FRAG_CHUNK_ELEMS;
    const uint4 zero = make_uint4(0, 0, 0, 0);          // keys past the prefix: they meet p = 0 in the MFMA and must be finite
    __shared__ uint16_t tile[ATT_CH][D + 8];
    for (int p = threadIdx.x; p < ATT_CH * D / 8; p += 256) {
        const int key = p / (D / 8), c = p % (D / 8);
        *reinterpret_cast<uint4*>(&tile[key][c * 8]) = k0 + key < plen ? *reinterpret_cast<const uint4*>(v + base + (size_t)(k0 + key) * D + c * 8) : zero;
    }
    for (int p = threadIdx.x; p < 1024; p += 256) {     // K fragments: 16-byte pieces of K rows, re-ordered
        const int t = p >> 8, ks = (p >> 6) & 3, lane = p & 63, ln = lane & 15, g = lane >> 4;
        const int key = k0 + (t >> 1) * 32 + (ln >> 2) * 8 + (t & 1) * 4 + (ln & 3);
        st_stream(out + (size_t)p * 8, key < plen ? *reinterpret_cast<const uint4*>(k + base + (size_t)key * D + ks * 32 + g * 8) : zero);      // read again only by the decode steps
    }
    __syncthreads();
    for (int p = threadIdx.x; p < 1024; p += 256) {     // V^T fragments: 8 keys of one dim per lane
        const int kk = p >> 9, nt = (p >> 6) & 7, lane = p & 63, ln = lane & 15, g = lane >> 4;
        uint16_t e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = tile[kk * 32 + g * 8 + i][ln * 8 + nt];
        uint4 o;
        o.x = e[0] | ((uint32_t)e[1] << 16); o.y = e[2] | ((uint32_t)e[3] << 16); o.z = e[4] | ((uint32_t)e[5] << 16); o.w = e[6] | ((uint32_t)e[7] << 16);
        st_stream(out + ATT_CH * D + (size_t)p * 8, o);
    }
}

// waves-per-SIMD hint 4: left alone the compiler hoists all 16 K (then V) fragment loads of a chunk and lands at 140
// registers = 3 waves per SIMD; capped at 128 the pass is 10 % faster (tools/attn_probe.py: 314 -> 274 us per layer).
template <int D>
__global__ void __launch_bounds__(256, 2) decode_attn_prefix_mfma_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ frag,
                                                                      const GroupDesc* __restrict__ groups,
                                                                      const int* __restrict__ group_rows, const int4* __restrict__ items,
                                                                      float* __restrict__ ws, int H, int Hkv, long long pre_stride,
                                                                      int pre_tmax, float scale, int nchunk, int sub) {
    static_assert(D == 128, "");
    constexpr int KS = D / 32, NT = D / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ln = lane & 15, g = lane >> 4;
    const int head = blockIdx.x * 4 + wave;               // head groups fastest: the (long-first) item list is walked in order
    if (head >= H) return;
    const int4 item = items[blockIdx.y];
    const GroupDesc gd = groups[item.x];
    // an item = `sub` consecutive 64-key chunks (online softmax across them): one partial per (row, head, item)
    const int r0 = item.y, part = item.z, kbeg = part * ATT_CH * sub;
    if (r0 >= gd.n_rows || kbeg >= gd.plen) return;
    const int k1 = min(gd.plen, kbeg + ATT_CH * sub);
    const int kvh = head / (H / Hkv);
    // B operand of S^T: this lane's query row
    int rq = r0 + ln; if (rq >= gd.n_rows) rq = gd.n_rows - 1;
    const int qrow = group_rows[gd.row_off + rq];
    frag8_t qf[KS];
    {
        const uint16_t* qp = q + ((size_t)qrow * H + head) * D + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const frag8_t*>(qp + ks * 32);
    }
    const uint16_t* fb = frag + 2 * ((size_t)gd.pslot * pre_stride + (size_t)kvh * pre_tmax * D) + (size_t)lane * 8;
    f32x4_t o[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lrun = 0.f;          // of query ln (replicated over g)
    // Chunks in a ROTATED order (the online softmax does not care): every (group, head) stream starts at t = 0 of a region that
    // is a multiple of 32 KiB away from the others', so waves walking in step would all sit on the same HBM channels
    const int nch_i = (k1 - kbeg + ATT_CH - 1) / ATT_CH;
    const int rot = (int)((unsigned)(head * 5 + item.x * 3 + r0) % (unsigned)nch_i);
    for (int ci = 0; ci < nch_i; ++ci) {
        const int cc = ci + rot < nch_i ? ci + rot : ci + rot - nch_i;
        const int k0 = kbeg + cc * ATT_CH;
        // The pass is a pure stream (one wave reads the chunk's 32-KiB block and nobody else does), so what matters is bytes in
        // flight: ALL 32 fragment loads of the chunk are issued before the first MFMA.  Issued a few at a time behind their
        // consumers (as the compiler schedules them under a 128-VGPR cap) the pass ran at 4.2 TB/s with 1-2 loads in flight
        // per wave in its P V half.
        frag8_t kf[4][KS], vf[2][NT];
        const uint16_t* cb = fb + (size_t)(k0 / ATT_CH) * FRAG_CHUNK_ELEMS;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[t][ks] = __builtin_nontemporal_load(reinterpret_cast<const frag8_t*>(cb + (t * KS + ks) * 512));
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) vf[kk][nt] = __builtin_nontemporal_load(reinterpret_cast<const frag8_t*>(cb + ATT_CH * D + (kk * NT + nt) * 512));
        __builtin_amdgcn_sched_barrier(0);
        // S^T: 4 tiles (a0, b0 | a1, b1); MFMA row i = ln of the A operand
        f32x4_t s[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            s[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s[t] = mfma16(kf[t][ks], qf[ks], s[t]);
        }
        // column = query ln; this lane's rows 4 g + r are keys k0 + 32 (t>>1) + 8 g + 4 (t&1) + r
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + (t >> 1) * 32 + g * 8 + (t & 1) * 4 + r;
                const float v = key < k1 ? s[t][r] * scale : -INFINITY;
                s[t][r] = v; mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mnew = fmaxf(mrun, mx);                      // finite: every chunk holds >= 1 valid key
        const float corr = __expf(mrun - mnew);                  // exp(-inf) = 0 on the first chunk
        float lsum = 0.f;
        frag8_t pf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pv = __expf(s[kk * 2 + (e >> 2)][e & 3] - mnew);      // exp(-inf) = 0 for masked keys
                lsum += pv;
                pf[kk][e] = (short)f2e(pv);
            }
        lsum += __shfl_xor(lsum, 16); lsum += __shfl_xor(lsum, 32);
        lrun = lrun * corr + lsum;
        mrun = mnew;
        if (ci > 0) {           // O rows are queries 4 g + r: their rescale factor lives in the lanes of column 4 g + r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float cr = __shfl(corr, 4 * g + r);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) o[nt][r] *= cr;
            }
        }
        // O += P V
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (k0 + kk * 32 < k1) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) o[nt] = mfma16(pf[kk], vf[kk][nt], o[nt]);
            }
        }
    }
    // O's C layout: column ln of tile nt = dim 8 ln + nt, row = query 4 g + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rr = r0 + g * 4 + r;
        if (rr < gd.n_rows) {
            const int orow = group_rows[gd.row_off + rr];
            float* wsp = ws + (((size_t)orow * H + head) * nchunk + part) * ATT_PS + ln * 8;
            *reinterpret_cast<float4*>(wsp) = make_float4(o[0][r], o[1][r], o[2][r], o[3][r]);
            *reinterpret_cast<float4*>(wsp + 4) = make_float4(o[4][r], o[5][r], o[6][r], o[7][r]);
        }
    }
    if (g == 0 && r0 + ln < gd.n_rows) {          // (m, l) of query ln live in the lanes of column ln
        float* wsp = ws + (((size_t)qrow * H + head) * nchunk + part) * ATT_PS;
        *reinterpret_cast<float2*>(wsp + D) = make_float2(mrun, lrun);
    }
}


inline int ok(hipError_t) { return hipGetLastError() == hipSuccess ? VDD_OK : VDD_ERR_LAUNCH; }

// Persistent blocks of the normalise-once projections: two 4-wave blocks per CU while two copies of the normalised rows fit in the
// 160 KiB of LDS; beyond that (>= 10 rows of 4096) ONE 8-wave block per CU, so that as many W bytes are in flight per CU as before
// (12 rows, gate/up: 48.6 us with one 4-wave block, 64 us with 512 blocks in two rounds - each round repeats the prologue).
static constexpr size_t NORMED_LDS_CAP = 142 * 1024;                 // + 16.5 KiB static (two partial-sum buffers of 8 waves)
struct NormedPlan { int grid; bool wide; };
static NormedPlan normed_plan(int n_cb, size_t lds_bytes) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    const bool two = 2 * (lds_bytes + 9 * 1024) <= 160 * 1024;
    const int g = (two ? 2 : 1) * n;
    return NormedPlan{n_cb < g ? n_cb : g, !two};
}
template <bool SWIGLU>
static int normed_launch(const void* H, const float* ss, int nss, const void* ln_w, float eps, const void* W, void* Y, int M, int N, int K,
                         int64_t ldh, int64_t ldy, int n_cb, void* stream) {
    const size_t lds = (size_t)M * 2 * (size_t)K;
    if (lds > NORMED_LDS_CAP || K > 8192 || nss > 512) return VDD_ERR_UNSUPPORTED;
    const NormIn ni{ss, nss, (const uint16_t*)ln_w, eps};
    const NormedPlan pl = normed_plan(n_cb, lds);
#define VDD_NORMED(NW, R)                                                                                                              \
    do {                                                                                                                               \
        static bool attr = false;                                                                                                      \
        if (!attr) { (void)hipFuncSetAttribute((const void*)skinny_normed_kernel<NW, SWIGLU, R>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)NORMED_LDS_CAP); attr = true; }                                                    \
        hipLaunchKernelGGL((skinny_normed_kernel<NW, SWIGLU, R>), dim3(pl.grid), dim3(NW * 64), lds, (hipStream_t)stream,              \
                           (const uint16_t*)H, (const uint16_t*)W, (uint16_t*)Y, M, N, K, (long long)ldh, (long long)ldy, ni, n_cb);   \
    } while (0)
    if (pl.wide) { if (M <= 8) VDD_NORMED(8, 8); else VDD_NORMED(8, 16); }
    else if (M <= 2) VDD_NORMED(4, 2);
    else if (M <= 4) VDD_NORMED(4, 4);
    else if (M <= 8) VDD_NORMED(4, 8);
    else VDD_NORMED(4, 16);
#undef VDD_NORMED
    return ok(hipSuccess);
}

}  // namespace VDD_ELEM_NS
}  // namespace

using namespace VDD_ELEM_NS;

extern "C" {

VDD_HIDDEN int VDD_IMPL(vdd_rmsnorm)(const void* x, const void* delta, const float* delta_slabs, int n_slabs, const void* w, void* y, void* resid_out,
                int M, int d, float eps, void* stream) {
    if (M <= 0) return VDD_OK;
    if (!x || !w || !y || d % 8 != 0 || d > 8192 || (delta && delta_slabs) || (delta_slabs && n_slabs < 1)) return VDD_ERR_INVALID_ARG;
    const float* Sl = delta_slabs; const long long ss = (long long)M * d;
    hipStream_t st = (hipStream_t)stream;
    auto X = (const uint16_t*)x; auto Dl = (const uint16_t*)delta; auto Wt = (const uint16_t*)w; auto Y = (uint16_t*)y; auto Ro = (uint16_t*)resid_out;
    const int vpt = (d / 8 + 255) / 256;
    // decode-size batches: the residual stream crosses the chip once per kernel (nontemporal); at prefill size the next kernel still
    // finds part of it in the Infinity Cache, and there the hint costs 2 % of the prefill
    const int stream_io = M <= 8192 ? 1 : 0;
    if (vpt <= 1) hipLaunchKernelGGL(rmsnorm_kernel<1>, dim3(M), dim3(256), 0, st, X, Dl, Sl, n_slabs, ss, Wt, Y, Ro, d, eps, stream_io);
    else if (vpt <= 2) hipLaunchKernelGGL(rmsnorm_kernel<2>, dim3(M), dim3(256), 0, st, X, Dl, Sl, n_slabs, ss, Wt, Y, Ro, d, eps, stream_io);
    else hipLaunchKernelGGL(rmsnorm_kernel<4>, dim3(M), dim3(256), 0, st, X, Dl, Sl, n_slabs, ss, Wt, Y, Ro, d, eps, stream_io);
    return ok(hipSuccess);
}

VDD_HIDDEN int VDD_IMPL(vdd_rope_kv_write)(const void* qkv, const int* pos, const int* cpos, const int* slot, const float* cos_sin, void* q_out, void* k_cache,
                      void* v_cache, int M, int Hq, int Hkv, int D, int64_t slot_stride, int t_max, void* stream) {
    if (M <= 0) return VDD_OK;
    if (!qkv || !pos || !cpos || !slot || !cos_sin || !q_out || !k_cache || !v_cache || D % 16 != 0) return VDD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rope_kv_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)qkv, pos, cpos, slot, cos_sin,
                       (uint16_t*)q_out, (uint16_t*)k_cache, (uint16_t*)v_cache, Hq, Hkv, D, (long long)slot_stride, t_max, M <= 8192 ? 1 : 0);
    return ok(hipSuccess);
}

VDD_HIDDEN int VDD_IMPL(vdd_silu_mul)(const void* gate_up, void* out, int64_t M, int F, void* stream) {
    if (M <= 0) return VDD_OK;
    if (!gate_up || !out || F % 8 != 0) return VDD_ERR_INVALID_ARG;
    if (M > 65535) {        // gridDim.y limit: launch in row slabs
        for (int m0 = 0; m0 < M; m0 += 65535) {
            const int mm = M - m0 < 65535 ? M - m0 : 65535;
            hipLaunchKernelGGL(silu_mul_kernel, dim3((F / 8 + 255) / 256, mm), dim3(256), 0, (hipStream_t)stream,
                               (const uint16_t*)gate_up + (size_t)m0 * 2 * F, (uint16_t*)out + (size_t)m0 * F, (long long)mm, F);
        }
        return ok(hipSuccess);
    }
    hipLaunchKernelGGL(silu_mul_kernel, dim3((F / 8 + 255) / 256, M), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)gate_up, (uint16_t*)out, (long long)M, F);
    return ok(hipSuccess);
}

VDD_HIDDEN int VDD_IMPL(vdd_embed)(const int64_t* ids, const void* table, void* out, int M, int d, int vocab, void* stream) {
    if (M <= 0) return VDD_OK;
    if (!ids || !table || !out || d % 8 != 0 || vocab <= 0) return VDD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(embed_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const long long*)ids, (const uint16_t*)table, (uint16_t*)out, d, vocab);
    return ok(hipSuccess);
}

VDD_HIDDEN int VDD_IMPL(vdd_embed_scatter)(const int32_t* ids, const int32_t* rows, const void* table, void* out, int M, int d, int vocab, void* stream) {
    if (M <= 0) return VDD_OK;
    if (!ids || !rows || !table || !out || d % 8 != 0 || vocab <= 0) return VDD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(embed_scatter_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, ids, rows, (const uint16_t*)table, (uint16_t*)out, d, vocab);
    return ok(hipSuccess);
}

// Eight waves per 16-column block (K split eight ways inside the block) are ONE resident block per CU: right when the column blocks
// fill the chip in whole rounds (N = 4096: 256 blocks on 256 CUs).  N = 5120 (LLaVA-1.5-13B) is 320 blocks - a second round for a
// quarter of the chip, 0.62 of the stream rate (o-projection 23.9 us for 52 MB, down 48.9 us for 141 MB at 12 rows); four-wave
// blocks are small enough for three per CU, so all 320 stream at once.
static bool eight_wave_blocks(int N) {
    static int n_cu = 0;
    if (n_cu == 0) { int dev = 0, n = 0; n_cu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256; }
    const int blocks = (N + 15) / 16;
    return blocks <= n_cu || blocks % n_cu == 0;
}

static int skinny_gemm_launch(const void* X, const void* W, const void* R, void* Y, float* Y_slabs, int n_split, int M, int N, int K,
                              int64_t ldx, int64_t ldr, int64_t ldy, void* stream) {
    if (M <= 0 || N <= 0) return VDD_OK;
    if (!X || !W || (!Y && !Y_slabs) || n_split < 1 || K % (128 * n_split) != 0 || M > 64 || (ldx % 8) != 0) return VDD_ERR_INVALID_ARG;
    if (!Y_slabs && n_split != 1) return VDD_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((N + 15) / 16, n_split);
    auto x = (const uint16_t*)X; auto w = (const uint16_t*)W; auto r = (const uint16_t*)R; auto y = (uint16_t*)Y;
#define VDD_SKINNY(MT, NW) hipLaunchKernelGGL((skinny_gemm_kernel<MT, NW>), grid, dim3(NW * 64), 0, st, x, w, r, y, Y_slabs, M, N, K, (long long)ldx, (long long)ldr, (long long)ldy, NormIn{}, (float*)nullptr)
    // few column blocks (N = d projections): eight waves per block split K eight ways instead of fp32 slabs across blocks
    if (M > 16 && n_split == 1 && Y != nullptr && K % 256 == 0 && N > 8192) {          // 17 - 64 rows, wide outputs: 32 columns per block
        const int mt = (M + 15) / 16;
        const dim3 g2((N + 31) / 32);
#define VDD_WIDE(MT, NW)                                                                                                               \
        do {                                                                                                                           \
            constexpr int smem = NW * MT * 2 * 64 * 4 * (int)sizeof(float);                                                            \
            static bool attr = false;                                                                                                  \
            if (!attr) { (void)hipFuncSetAttribute((const void*)skinny_wide_kernel<MT, NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; } \
            hipLaunchKernelGGL((skinny_wide_kernel<MT, NW, false>), g2, dim3(NW * 64), smem, st, x, w, r, y, M, N, K, (long long)ldx, (long long)ldr, (long long)ldy); \
        } while (0)
        if (mt == 2) VDD_WIDE(2, 4); else if (mt == 3) VDD_WIDE(3, 4); else VDD_WIDE(4, 4);
#undef VDD_WIDE
        return ok(hipSuccess);
    }
    if (M <= 16 && n_split == 1 && N <= 8192 && K % 256 == 0 && eight_wave_blocks(N)) VDD_SKINNY(1, 8);
    else if (M <= 16) VDD_SKINNY(1, 4);
    else if (M <= 32) VDD_SKINNY(2, 4); else VDD_SKINNY(4, 4);
#undef VDD_SKINNY
    return ok(hipSuccess);
}

VDD_HIDDEN int VDD_IMPL(vdd_skinny_gemm)(const void* X, const void* W, const void* R, void* Y, float* Y_slabs, int n_split, int M, int N, int K,
                    int64_t ldx, int64_t ldr, int64_t ldy, void* stream) {
    return skinny_gemm_launch(X, W, R, Y, Y_slabs, n_split, M, N, K, ldx, ldr, ldy, stream);
}

static int skinny_swiglu_launch(const void* X, const void* W, void* act, int M, int F, int K, int64_t ldx, void* stream) {
    if (M <= 0 || F <= 0) return VDD_OK;
    if (!X || !W || !act || M > 64 || K % 128 != 0 || (ldx % 8) != 0 || (M > 16 && K % 256 != 0)) return VDD_ERR_INVALID_ARG;
    if (M > 16) {                                                           // 17 - 64 rows: 16 features (gate + up tiles) per block
        const int mt = (M + 15) / 16;
        const dim3 g2((F + 15) / 16);
        hipStream_t st = (hipStream_t)stream;
#define VDD_WIDE(MT, NW)                                                                                                               \
        do {                                                                                                                           \
            constexpr int smem = NW * MT * 2 * 64 * 4 * (int)sizeof(float);                                                            \
            static bool attr = false;                                                                                                  \
            if (!attr) { (void)hipFuncSetAttribute((const void*)skinny_wide_kernel<MT, NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; } \
            hipLaunchKernelGGL((skinny_wide_kernel<MT, NW, true>), g2, dim3(NW * 64), smem, st, (const uint16_t*)X, (const uint16_t*)W, (const uint16_t*)nullptr, \
                               (uint16_t*)act, M, F, K, (long long)ldx, 0LL, (long long)F);                                            \
        } while (0)
        if (mt == 2) VDD_WIDE(2, 4); else if (mt == 3) VDD_WIDE(3, 4); else VDD_WIDE(4, 4);
#undef VDD_WIDE
        return ok(hipSuccess);
    }
    hipLaunchKernelGGL(skinny_swiglu_kernel<false>, dim3((F + 7) / 8), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)X,
                            (const uint16_t*)W, (uint16_t*)act, M, F, K, (long long)ldx, NormIn{});
    return ok(hipSuccess);
}

// ---- small-M decoder-layer fusions of the RMSNorm launches (see skinny_gemm_kernel)
VDD_HIDDEN int VDD_IMPL(vdd_skinny_gemm_resid_ss)(const void* X, const void* W, const void* R, void* Y, float* ss_out, int M, int N, int K, int64_t ldx,
                             int64_t ldr, int64_t ldy, void* stream) {
    if (M <= 0 || N <= 0) return VDD_OK;
    if (!X || !W || !R || !Y || !ss_out || M > 16 || K % 128 != 0 || (ldx % 8) != 0) return VDD_ERR_INVALID_ARG;
    dim3 grid((N + 15) / 16, 1);
    hipStream_t st = (hipStream_t)stream;
    if (N <= 8192 && K % 256 == 0 && eight_wave_blocks(N))
        hipLaunchKernelGGL((skinny_gemm_kernel<1, 8, false, true>), grid, dim3(512), 0, st, (const uint16_t*)X, (const uint16_t*)W, (const uint16_t*)R,
                           (uint16_t*)Y, (float*)nullptr, M, N, K, (long long)ldx, (long long)ldr, (long long)ldy, NormIn{}, ss_out);
    else
        hipLaunchKernelGGL((skinny_gemm_kernel<1, 4, false, true>), grid, dim3(256), 0, st, (const uint16_t*)X, (const uint16_t*)W, (const uint16_t*)R,
                           (uint16_t*)Y, (float*)nullptr, M, N, K, (long long)ldx, (long long)ldr, (long long)ldy, NormIn{}, ss_out);
    return ok(hipSuccess);
}

VDD_HIDDEN int VDD_IMPL(vdd_skinny_gemm_normed)(const void* H, const float* ss, int nss, const void* ln_w, float eps, const void* W, void* Y, int M, int N,
                           int K, int64_t ldh, int64_t ldy, void* stream) {
    if (M <= 0 || N <= 0) return VDD_OK;
    if (!H || !ss || nss <= 0 || (nss % 4) != 0 || !ln_w || !W || !Y || M > 16 || K % 256 != 0 || (ldh % 8) != 0) return VDD_ERR_INVALID_ARG;
    return normed_launch<false>(H, ss, nss, ln_w, eps, W, Y, M, N, K, ldh, ldy, (N + 15) / 16, stream);
}

VDD_HIDDEN int VDD_IMPL(vdd_skinny_swiglu_normed)(const void* H, const float* ss, int nss, const void* ln_w, float eps, const void* W_gate_up, void* act,
                             int M, int F, int K, int64_t ldh, void* stream) {
    if (M <= 0 || F <= 0) return VDD_OK;
    if (!H || !ss || nss <= 0 || (nss % 4) != 0 || !ln_w || !W_gate_up || !act || M > 16 || K % 256 != 0 || (ldh % 8) != 0) return VDD_ERR_INVALID_ARG;
    return normed_launch<true>(H, ss, nss, ln_w, eps, W_gate_up, act, M, F, K, ldh, (int64_t)F, (F + 7) / 8, stream);
}
VDD_HIDDEN int VDD_IMPL(vdd_skinny_swiglu)(const void* X, const void* W_gate_up, void* act, int M, int F, int K, int64_t ldx, void* stream) {
    return skinny_swiglu_launch(X, W_gate_up, act, M, F, K, ldx, stream);
}

VDD_HIDDEN int VDD_IMPL(vdd_decode_attention)(const void* q, const void* k_cache, const void* v_cache, const void* k_prefix, const void* v_prefix,
                         const int32_t* rows, void* out, void* workspace, int M, int H, int Hkv, int D, int64_t slot_stride, int t_max,
                         int64_t prefix_stride, int prefix_tmax, int max_len, float scale, void* stream) {
    if (M <= 0) return VDD_OK;
    if (!q || !k_cache || !v_cache || !k_prefix || !v_prefix || !rows || !out || !workspace || D != 128 || H % Hkv != 0 || max_len <= 0) return VDD_ERR_INVALID_ARG;
    const int nchunk = (max_len + ATT_CH - 1) / ATT_CH;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(decode_attn_kernel<128>, dim3((H + 3) / 4, M, nchunk), dim3(256), 0, st, (const uint16_t*)q,
                       (const uint16_t*)k_cache, (const uint16_t*)v_cache, (const uint16_t*)k_prefix, (const uint16_t*)v_prefix,
                       (const AttnRow*)rows, (float*)workspace, H, Hkv, (long long)slot_stride, t_max, (long long)prefix_stride,
                       prefix_tmax, scale, nchunk, 0, 0);
    hipLaunchKernelGGL(decode_attn_combine_kernel<128>, dim3((H + 3) / 4, M), dim3(256), 0, st, (const float*)workspace,
                       (const AttnRow*)rows, (uint16_t*)out, H, nchunk, 0, ATT_CH);
    return ok(hipSuccess);
}

VDD_HIDDEN int VDD_IMPL(vdd_decode_attention_fused)(const void* qkv, const int32_t* pos, const int32_t* cpos, const int32_t* slot, const float* cos_sin,
                               void* k_cache, void* v_cache, const void* k_prefix, const void* v_prefix, const int32_t* rows, void* out,
                               int M, int H, int Hkv, int D, int64_t slot_stride, int t_max, int64_t prefix_stride, int prefix_tmax,
                               float scale, void* stream) {
    if (M <= 0) return VDD_OK;
    if (!qkv || !pos || !cpos || !slot || !cos_sin || !k_cache || !v_cache || !k_prefix || !v_prefix || !rows || !out || D != 128 ||
        H % Hkv != 0 || M > 65535) return VDD_ERR_INVALID_ARG;
    hipLaunchKernelGGL((decode_attn_fused_kernel<128, 16, 1>), dim3(H, M), dim3(1024), 0, (hipStream_t)stream, (const uint16_t*)qkv, pos, cpos,
                       slot, cos_sin, (uint16_t*)k_cache, (uint16_t*)v_cache, (const uint16_t*)k_prefix, (const uint16_t*)v_prefix,
                       (const AttnRow*)rows, (uint16_t*)out, H, Hkv, (long long)slot_stride, t_max, (long long)prefix_stride, prefix_tmax,
                       scale, (float*)nullptr, (int*)nullptr);
    return ok(hipSuccess);
}

VDD_HIDDEN int64_t VDD_IMPL(vdd_decode_attention_fused_split_workspace_bytes)(int M, int H, int n_split) {
    if (M <= 0 || H <= 0 || n_split < 1) return 0;
    return ((int64_t)M * H * n_split * ATT_FS + (int64_t)M * H) * 4;
}

VDD_HIDDEN int VDD_IMPL(vdd_decode_attention_fused_split)(const void* qkv, const int32_t* pos, const int32_t* cpos, const int32_t* slot, const float* cos_sin,
                                     void* k_cache, void* v_cache, const void* k_prefix, const void* v_prefix, const int32_t* rows, void* out,
                                     int M, int H, int Hkv, int D, int64_t slot_stride, int t_max, int64_t prefix_stride, int prefix_tmax,
                                     float scale, void* workspace, int n_split, void* stream) {
    if (M <= 0) return VDD_OK;
    if (!qkv || !pos || !cpos || !slot || !cos_sin || !k_cache || !v_cache || !k_prefix || !v_prefix || !rows || !out || !workspace || D != 128 ||
        H % Hkv != 0 || M > 65535 || (n_split != 2 && n_split != 4)) return VDD_ERR_INVALID_ARG;
    float* ws = (float*)workspace;
    int* tickets = (int*)(ws + (size_t)M * H * n_split * ATT_FS);
#define VDD_FUSED_SPLIT(KS)                                                                                                            \
    hipLaunchKernelGGL((decode_attn_fused_kernel<128, 16, KS>), dim3(H, M, KS), dim3(1024), 0, (hipStream_t)stream, (const uint16_t*)qkv, pos, \
                       cpos, slot, cos_sin, (uint16_t*)k_cache, (uint16_t*)v_cache, (const uint16_t*)k_prefix, (const uint16_t*)v_prefix,  \
                       (const AttnRow*)rows, (uint16_t*)out, H, Hkv, (long long)slot_stride, t_max, (long long)prefix_stride, prefix_tmax, \
                       scale, ws, tickets)
    if (n_split == 2) VDD_FUSED_SPLIT(2); else VDD_FUSED_SPLIT(4);
#undef VDD_FUSED_SPLIT
    return ok(hipSuccess);
}

VDD_HIDDEN int VDD_IMPL(vdd_decode_attention_grouped)(const void* q, const void* k_cache, const void* v_cache, const void* k_prefix, const void* v_prefix,
                                 const void* prefix_frag, const int32_t* rows, const int32_t* groups, const int32_t* group_rows,
                                 const int32_t* items, int n_items, void* out, void* workspace, int M, int H, int Hkv, int D,
                                 int64_t slot_stride, int t_max, int64_t prefix_stride, int prefix_tmax, int max_prefix_len,
                                 int max_own_len, int prefix_chunks_per_item, float scale, void* stream) {
    if (M <= 0) return VDD_OK;
    if (!q || !k_cache || !v_cache || !k_prefix || !v_prefix || !rows || !groups || !group_rows || !items || !out || !workspace || D != 128 ||
        H % Hkv != 0 || max_own_len <= 0 || max_prefix_len < 0 || n_items < 0 || n_items > 65535 || prefix_chunks_per_item < 1) return VDD_ERR_INVALID_ARG;
    if (n_items > 0 && (prefix_frag == nullptr || prefix_tmax % ATT_CH != 0)) return VDD_ERR_INVALID_ARG;     // the MFMA prefix pass reads the fragment-major image
    const int sub = prefix_chunks_per_item;                                     // it leaves one partial per item of `sub` 64-key chunks
    const int pre_keys = ATT_CH * sub;
    const int npre = (max_prefix_len + pre_keys - 1) / pre_keys, nown = (max_own_len + ATT_CH - 1) / ATT_CH;
    const int nchunk = npre + nown;
    hipStream_t st = (hipStream_t)stream;
    if (n_items > 0 && npre > 0) {
        hipLaunchKernelGGL(decode_attn_prefix_mfma_kernel<128>, dim3((H + 3) / 4, n_items), dim3(256), 0, st,
                           (const uint16_t*)q, (const uint16_t*)prefix_frag, (const GroupDesc*)groups, group_rows,
                           (const int4*)items, (float*)workspace, H, Hkv, (long long)prefix_stride, prefix_tmax, scale, nchunk, sub);
    }
    if (max_own_len <= 256) {        // short own ranges: one wave per (row, head) finishes the row (own keys + prefix partials)
        hipLaunchKernelGGL(decode_attn_own_merge_kernel<128>, dim3((H + 3) / 4, M), dim3(256), 0, st, (const uint16_t*)q,
                           (const uint16_t*)k_cache, (const uint16_t*)v_cache, (const AttnRow*)rows, (const float*)workspace,
                           (uint16_t*)out, H, Hkv, (long long)slot_stride, t_max, scale, nchunk, npre, pre_keys);
        return ok(hipSuccess);
    }
    hipLaunchKernelGGL(decode_attn_kernel<128>, dim3((H + 3) / 4, M, nown), dim3(256), 0, st, (const uint16_t*)q,
                       (const uint16_t*)k_cache, (const uint16_t*)v_cache, (const uint16_t*)k_prefix, (const uint16_t*)v_prefix,
                       (const AttnRow*)rows, (float*)workspace, H, Hkv, (long long)slot_stride, t_max, (long long)prefix_stride,
                       prefix_tmax, scale, nchunk, 1, npre);
    hipLaunchKernelGGL(decode_attn_combine_kernel<128>, dim3((H + 3) / 4, M), dim3(256), 0, st, (const float*)workspace,
                       (const AttnRow*)rows, (uint16_t*)out, H, nchunk, npre, pre_keys);
    return ok(hipSuccess);
}

VDD_HIDDEN int VDD_IMPL(vdd_prefix_fragments)(const void* k_prefix, const void* v_prefix, void* prefix_frag, const int32_t* prefix_len_of_slot, int n_slots,
                         int Hkv, int t_max, int D, void* stream) {
    if (n_slots <= 0) return VDD_OK;
    if (!k_prefix || !v_prefix || !prefix_frag || !prefix_len_of_slot || D != 128 || t_max % ATT_CH != 0) return VDD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(prefix_fragments_kernel, dim3(t_max / ATT_CH, Hkv, n_slots), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)k_prefix, (const uint16_t*)v_prefix, (uint16_t*)prefix_frag, t_max, prefix_len_of_slot);
    return ok(hipSuccess);
}

VDD_HIDDEN int64_t VDD_IMPL(vdd_decode_attention_workspace_bytes)(int M, int H, int D, int max_len) {
    return (int64_t)M * H * ((max_len + ATT_CH - 1) / ATT_CH) * ATT_PS * 4;
}

}  // extern "C"
